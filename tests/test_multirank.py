"""The N > 1 path on CPU: two gloo ranks shard a list of images round-robin, each produces its
"latents", and rank 0 gathers and reassembles them in dataset order (the one collective of the job)."""
import os
import socket
import subprocess
import sys

import torch

from style_transfer_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from style_transfer_amd import sharding
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_items = {n_items}
mine = sharding.shard_indices(n_items, rank, world)
# stand-in for the per-image hot path: latent i is filled with i + 0.25
local = torch.stack([torch.full((4, 8, 8), i + 0.25) for i in mine]) if mine else torch.zeros((0, 4, 8, 8))
local, n_valid = sharding.pad_to_common(local, n_items, world)
dist.barrier()
stacked = sharding.gather_latents(local, dist, dst=0)
if rank == 0:
    full = sharding.reassemble(stacked, n_items)
    assert full.shape == (n_items, 4, 8, 8), full.shape
    for i in range(n_items):
        assert torch.all(full[i] == i + 0.25), i
    print("GATHER_OK", n_items, world)
else:
    assert stacked is None
dist.barrier()
dist.destroy_process_group()
"""


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_indices_cover_every_item_once():
    for n in (0, 1, 5, 64, 90):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in sharding.shard_indices(n, r, world))
            assert seen == list(range(n))
            sizes = [len(sharding.shard_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_shard_by_content_keeps_a_content_image_on_one_rank():
    """The partition of runs that keep per-content state: every item once, all items of a content key on one
    rank, the same answer whoever computes it, loads balanced to within the largest group."""
    import random
    rnd = random.Random(7)
    for trial in range(30):
        n_keys, world = rnd.randint(1, 12), rnd.choice((1, 2, 3, 8))
        keys = [("c%d" % rnd.randrange(n_keys)) for _ in range(rnd.randint(0, 40))]
        lists = sharding.shard_by_content(keys, world)
        assert len(lists) == world and sorted(i for l in lists for i in l) == list(range(len(keys)))
        owner = {}
        for r, l in enumerate(lists):
            for i in l:
                assert owner.setdefault(keys[i], r) == r
            runs = [keys[i] for j, i in enumerate(l) if j == 0 or keys[i] != keys[l[j - 1]]]
            assert len(runs) == len(set(runs))                 # the items of a content key sit next to each other
        assert lists == sharding.shard_by_content(list(keys), world)
        biggest = max([keys.count(k) for k in set(keys)] or [0])
        assert max(len(l) for l in lists) - min(len(l) for l in lists) <= max(biggest, 1)
    # one content per line (the reference's dataset.txt): the same loads as round-robin
    lists = sharding.shard_by_content(["c%d" % i for i in range(10)], 4)
    assert sorted(len(l) for l in lists) == [2, 2, 3, 3]
    # gather of a ragged, non-round-robin partition comes back in dataset order
    keys = ["a", "b", "a", "a", "c", "b"]
    lists = sharding.shard_by_content(keys, 2)
    per = max(len(l) for l in lists)
    blocks = []
    for l in lists:
        x = torch.stack([torch.full((2, 3), float(i)) for i in l]) if l else torch.zeros((0, 2, 3))
        blocks.append(sharding.pad_to_common(x, len(keys), 2, per=per)[0])
    full = sharding.reassemble_lists(torch.stack(blocks), lists)
    assert [float(full[i, 0, 0]) for i in range(len(keys))] == [0., 1., 2., 3., 4., 5.]


def test_single_process_gather_is_identity():
    x = torch.arange(24.0).reshape(3, 2, 4)
    assert torch.equal(sharding.reassemble(sharding.gather_latents(x), 3), x)


def run_world(n_items, world=2):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                   LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER.format(root=ROOT, n_items=n_items)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    return outs


def test_two_ranks_gloo_gather_even_and_ragged():
    assert "GATHER_OK 6 2" in run_world(6)[0]
    assert "GATHER_OK 5 2" in run_world(5)[0]      # rank 1 holds one item fewer: padded for the gather


def test_bench_starts_its_own_ranks_when_the_driver_does_not():
    """`python bench.py --gpus N` without WORLD_SIZE must run N ranks, not one: the launcher starts N fresh
    processes with the rank environment torch.distributed.run would give them (probed here without a GPU)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--rank-probe"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    seen = sorted((json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")), key=lambda d: d["rank"])
    assert [d["rank"] for d in seen] == [0, 1, 2] and all(d["world"] == 3 and d["local_rank"] == d["rank"] for d in seen)
    assert len({d["port"] for d in seen}) == 1 and seen[0]["master"] == "127.0.0.1"
    # a WORLD_SIZE that disagrees with --gpus is an error, not a silent one-rank run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rank-probe"],
                       env=dict(env, WORLD_SIZE="4", RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
