"""Feature-store loader, packed side-car, single-style driver path and producer command on the GPU."""
import json
import os
import types

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_masks, rel_err
from style_transfer_amd import feature_store as fs
from style_transfer_amd import ops
from style_transfer_amd.loader import FeatureLoader

pytestmark = pytest.mark.gpu


def case():
    return json.load(open(os.path.join(GOLDEN, "unet_case.json")))


def feature_list(g, steps=50):
    out = []
    for _ in range(steps):
        e = {"config": {"T": 1.5}, "z_enc": torch.randn(1, 4, 16, 16, generator=g)}
        for b in range(6, 12):
            n, d = (64, 40) if b < 9 else (256, 20)
            for t in "qkv":
                e["output_block_%d_self_attn_%s" % (b, t)] = torch.randn(8, n, d, generator=g)
        out.append(e)
    return out


def make_root(tmp_path, lines, masks=()):
    root, pre = tmp_path / "data", tmp_path / "pre"
    for sub in ("cnt", "sty/char", "sty/back"):
        os.makedirs(root / sub, exist_ok=True)
    os.makedirs(pre, exist_ok=True)
    g = torch.Generator().manual_seed(11)
    done = set()
    for ln in lines:
        names = ln.split()
        for sub, name, suffix in zip(("cnt", "sty/char", "sty/back"), names, ("_cnt.pkl", "_sty.pkl", "_sty.pkl")):
            (root / sub / name).write_bytes(b"x")
            f = os.path.splitext(name)[0] + suffix
            if f not in done:
                done.add(f)
                fs.save_feature_file(str(pre / f), feature_list(g))
    (root / "dataset.txt").write_text("\n".join(lines) + "\n")
    for n in masks:
        np.save(str(root / "cnt" / (n + "_mask.npy")), load_masks()[case()["mask"]])
    cfg = dict(case()["cfg"]); cfg["model_channels"] = 160
    (tmp_path / "unet.json").write_text(json.dumps(cfg))
    return root, pre


def cli_args(tmp_path, root, pre, out):
    return ["--data_root", str(root), "--precomputed", str(pre), "--output_path", str(tmp_path / out), "--random_unet",
            "--unet_config", str(tmp_path / "unet.json"), "--H", "128", "--W", "128", "--start_step", "4",
            "--gamma", "0.75", "--T", "1.5"]


def same_store(a, b):
    """Two stores hold the same bytes: tensors equal, key/value images equal."""
    for la, lb in zip(a.lists(), b.lists()):
        assert len(la) == len(lb)
        for x, y in zip(la, lb):
            assert set(x) <= set(y) and {k for k in y if not k.endswith("_self_attn_q")} <= set(x)
            for k in x:
                if k == "config":
                    assert x[k] == y[k]
                elif isinstance(y[k], ops.PackedKV):
                    assert (x[k].B, x[k].H, x[k].N, x[k].d, x[k].compute) == (y[k].B, y[k].H, y[k].N, y[k].d, y[k].compute)
                    assert torch.equal(x[k].image, y[k].image), k
                else:
                    assert torch.equal(x[k], y[k]), k


@pytest.mark.parametrize("compute", ["f16", "bf16"])
def test_sidecar_holds_the_bytes_pack_produces(tmp_path, compute):
    """Route 1: pickle -> host -> HBM -> sid_pack_kv on the loader's stream (and the side-car written);
    route 2: side-car -> pinned host -> HBM, no conversion.  Both equal FeatureStore.load(...).pack()
    byte for byte, and the side-car of a style file is a third of its pickle."""
    root, pre = make_root(tmp_path, ["c1.jpg a.png b.png"])
    want = fs.FeatureStore.load(str(pre), "c1", "a", "b", "cuda:0").pack(8, compute, T=1.5)
    torch.cuda.synchronize()
    ld = FeatureLoader(str(pre), "cuda:0", heads=8, compute=compute, T=1.5, sidecar="write", prefetch=True)
    ld.request(("c1", "a", "b"))
    got = ld.get(("c1", "a", "b"))
    torch.cuda.synchronize()
    assert ld.stats["routes"] == {"pkl": 3, "sidecar": 0}
    for role, base in (("cnt", "c1"), ("sty", "a"), ("sty", "b")):
        assert os.path.isfile(str(pre / ("%s_%s.sidpack" % (base, role))))
    # (a style list from the loader holds no queries: compare what it has)
    same_store(got, want)
    assert "output_block_9_self_attn_q" not in got.char[3] and "output_block_9_self_attn_q" in got.cnt[3]
    ld.close()
    ld2 = FeatureLoader(str(pre), "cuda:0", heads=8, compute=compute, T=1.5, sidecar="use", prefetch=False)
    again = ld2.get(("c1", "a", "b"))
    torch.cuda.synchronize()
    assert ld2.stats["routes"] == {"pkl": 0, "sidecar": 3}
    same_store(again, want)
    # (at these small head dims the operand padding is at its worst; at SD-v1 sizes the ratio is 0.28)
    assert os.path.getsize(str(pre / "a_sty.sidpack")) < 0.6 * os.path.getsize(str(pre / "a_sty.pkl"))
    # another arithmetic, or a rewritten pickle, makes the side-car stale: the pickle is read again
    other = "bf16" if compute == "f16" else "f16"
    ld3 = FeatureLoader(str(pre), "cuda:0", heads=8, compute=other, T=1.5, sidecar="use", prefetch=False)
    ld3.get(("c1", "a", "b"))
    assert ld3.stats["routes"]["pkl"] == 3
    # the merged dicts of a loader store drive the attention like those of a plain store
    opt = types.SimpleNamespace(gamma=0.75, T=1.5)
    m1, m2 = again.merged(opt, 49)[5], want.merged(opt, 49)[5]
    assert set(m1) == {k for k in m2 if not (k.endswith("_q_sty1") or k.endswith("_q_sty2"))}


def test_prefetch_and_sidecar_change_no_latent(tmp_path):
    """The CLI over a sweep that revisits contents and styles: reading ahead on the worker thread, serving
    lists from HBM, and reading side-cars instead of pickles all hand the sampler the same bytes
    (test_sidecar_holds_the_bytes_pack_produces compares them bit for bit), so the latents are those of a
    run that reads every file in turn, the reference's order.  The host network's stock convolutions are
    not bit-reproducible from run to run, so "the same" is measured against two identical runs: bit-equal
    if those are, else within a few times their spread."""
    from style_transfer_amd import run as cli
    lines = ["c1.jpg a.png b.png", "c2.jpg a.png b.png", "c1.jpg a2.png b.png", "c2.jpg a2.png b.png"]
    root, pre = make_root(tmp_path, lines, masks=("c1", "c2"))
    sync = ["--no_prefetch", "--sidecar", "off"]
    cli.main(cli_args(tmp_path, root, pre, "warm") + sync)           # the convolution library picks its kernels
    load = lambda name: torch.load(str(tmp_path / name / "latents_all.pt"), weights_only=True)
    cli.main(cli_args(tmp_path, root, pre, "plain") + sync)
    cli.main(cli_args(tmp_path, root, pre, "plain2") + sync)
    want = load("plain")
    assert tuple(want.shape) == (4, 4, 16, 16) and torch.isfinite(want).all()
    spread = (load("plain2") - want).abs().max().item()
    runs = {"prefetch": ["--sidecar", "off"], "write": ["--sidecar", "write"], "use": ["--sidecar", "use"],
            "small_cache": ["--sidecar", "use", "--store_cache_gb", "0.001"]}
    for name, extra in runs.items():
        cli.main(cli_args(tmp_path, root, pre, name) + extra)
        diff = (load(name) - want).abs().max().item()
        print("%s: max |latent difference| %.2e (two identical runs: %.2e)" % (name, diff, spread))
        assert diff <= max(4 * spread, 1e-6 * want.abs().max().item()), (name, diff, spread)
    assert os.path.isfile(str(pre / "c1_cnt.sidpack")) and os.path.isfile(str(pre / "a2_sty.sidpack"))


def test_single_style_line_takes_the_no_mask_path(tmp_path, monkeypatch):
    """A `cnt sty` line: initial latent adain(cnt, sty), single-style merge, the no-mask StyleID attention
    branch (ldm/modules/attention.py:569-579) -- against the same steps made by hand."""
    from style_transfer_amd import run as cli
    from style_transfer_amd.features import adain, feat_merge
    from style_transfer_amd.sampler import DDIMSampler
    root, pre = make_root(tmp_path, ["c1.jpg a.png"])
    calls = {"plain": 0, "injected": 0}
    real_plain, real_inj = ops.plain_attention, ops.injected_attention
    monkeypatch.setattr(ops, "plain_attention", lambda *a, **k: (calls.__setitem__("plain", calls["plain"] + 1), real_plain(*a, **k))[1])
    monkeypatch.setattr(ops, "injected_attention", lambda *a, **k: (calls.__setitem__("injected", calls["injected"] + 1), real_inj(*a, **k))[1])
    args = cli_args(tmp_path, root, pre, "single")
    cli.main(args)
    got = torch.load(str(tmp_path / "single" / "c1_a.pt"), weights_only=True)
    assert tuple(got.shape) == (1, 4, 16, 16) and torch.isfinite(got).all()
    assert calls["injected"] == 0 and calls["plain"] > 0
    # by hand
    opt = cli.build_parser().parse_args(args)
    dev = torch.device("cuda:0")
    model = cli.load_unet(opt, dev)
    model.set_content_name(str(root / "cnt" / "c1.jpg"))
    store = fs.FeatureStore.load(str(pre), "c1", "a", None, dev)
    cz, sz = store.z_enc()
    merged = feat_merge(opt, store.cnt, store.char, start_step=opt.start_step)
    assert "output_block_9_self_attn_q_sty" in merged[49] and "output_block_9_self_attn_k_sty2" not in merged[49]
    s = DDIMSampler(model)
    s.make_schedule(50, 0.0)
    with torch.no_grad():
        want = s.sample(adain(cz, sz), torch.zeros(1, 77, 768, device=dev), injected_features=merged, start_step=opt.start_step)
    assert rel_err(got.numpy(), want.cpu().numpy()) < 1e-4
    # the style matters and the injection matters
    with torch.no_grad():
        bare = s.sample(adain(cz, sz), torch.zeros(1, 77, 768, device=dev), start_step=opt.start_step)
    assert rel_err(bare.cpu().numpy(), want.cpu().numpy()) > 1e-2
    # a two-name line whose content has a mask file is refused, as the reference's mask branch would fail on it
    np.save(str(root / "cnt" / "c1_mask.npy"), load_masks()[case()["mask"]])
    with pytest.raises(ValueError, match="has a mask file"):
        cli.main(cli_args(tmp_path, root, pre, "refused"))


def test_producer_command_writes_the_store_the_consumer_reads(tmp_path):
    """`python -m style_transfer_amd.produce`: latents -> `{cnt}_cnt.pkl`, `{sty}_sty.pkl`,
    `{cnt}_residuals_all.pkl` in the reference's layout (generate_pkl_only.py:181-250, :310-316); existing files
    are skipped; the stylisation command then runs on what was written, high-frequency injection included."""
    from style_transfer_amd import produce, run as cli
    from style_transfer_amd.sampler import extract_features
    root, pre = tmp_path / "data", tmp_path / "pre"
    lat = tmp_path / "lat"
    for sub in ("cnt", "sty/char", "sty/back"):
        os.makedirs(root / sub)
    os.makedirs(lat)
    for sub, name in (("cnt", "c1.jpg"), ("sty/char", "a.png"), ("sty/back", "b.png")):
        (root / sub / name).write_bytes(b"x")
    (root / "dataset.txt").write_text("c1.jpg a.png b.png\n")
    np.save(str(root / "cnt" / "c1_mask.npy"), load_masks()[case()["mask"]])
    g = torch.Generator().manual_seed(3)
    for n in ("c1", "a", "b"):
        torch.save(torch.randn(1, 4, 16, 16, generator=g), str(lat / (n + ".pt")))
    (tmp_path / "unet.json").write_text(json.dumps(case()["cfg"]))
    common = ["--data_root", str(root), "--precomputed", str(pre), "--random_unet", "--unet_config", str(tmp_path / "unet.json"),
              "--H", "128", "--W", "128"]
    written = produce.main(common + ["--latents", str(lat)])
    names = sorted(os.path.basename(w) for w in written)
    assert names == ["a_sty.pkl", "b_sty.pkl", "c1_cnt.pkl", "c1_residuals_all.pkl"]
    feats = fs.load_feature_file(str(pre / "c1_cnt.pkl"), "cuda:0")
    assert len(feats) == 50 and feats[0]["config"] == {"T": 1.5}
    assert tuple(feats[7]["output_block_11_self_attn_k"].shape) == (8, 256, 20)
    res = fs.load_residuals_file(str(pre / "c1_residuals_all.pkl"), "cpu")
    assert sorted(res) == list(range(1, 1000, 20)) and len(res[1]) == 18
    # what the file holds is what extract_features gives for that latent with that model
    opt = produce.build_parser().parse_args(common + ["--latents", str(lat)])
    model = cli.load_unet(opt, torch.device("cuda:0"))
    x0 = torch.load(str(lat / "c1.pt"), weights_only=True).to("cuda:0")
    with torch.no_grad():
        want = extract_features(model, x0, torch.zeros(1, 77, 768, device="cuda:0"))
    for i in (0, 20, 49):
        for k in ("z_enc", "output_block_6_self_attn_q", "output_block_11_self_attn_v"):
            assert rel_err(feats[i][k].cpu().numpy(), want[i][k].cpu().numpy()) < 1e-3, (i, k)
    # second run: everything exists, nothing is rewritten (the reference's resume, generate_pkl_only.py:196)
    stamp = os.path.getmtime(str(pre / "c1_cnt.pkl"))
    assert produce.main(common + ["--latents", str(lat)]) == []
    assert os.path.getmtime(str(pre / "c1_cnt.pkl")) == stamp
    # a missing latent is an error that names the file
    (root / "dataset.txt").write_text("c1.jpg a.png b.png\nc1.jpg a.png b.png\n")
    os.remove(str(pre / "a_sty.pkl")); os.remove(str(lat / "a.pt"))
    with pytest.raises(FileNotFoundError, match="a.pt"):
        produce.main(common + ["--latents", str(lat)])
    torch.save(torch.randn(1, 4, 16, 16, generator=g), str(lat / "a.pt"))
    produce.main(common + ["--latents", str(lat)])
    # the consumer on the produced store: 49 evaluations, the filter active (a residual file exists)
    (root / "dataset.txt").write_text("c1.jpg a.png b.png\n")
    cli.main(common + ["--output_path", str(tmp_path / "out"), "--ratio", "0.5"])
    z = torch.load(str(tmp_path / "out" / "c1_a_b.pt"), weights_only=True)
    assert tuple(z.shape) == (1, 4, 16, 16) and torch.isfinite(z).all()


def test_lanes_give_the_latents_of_one_at_a_time(tmp_path):
    """`--lanes 3`: three images in flight, each on its own stream with its own graphed evaluation, slots and
    workspace, contents with different masks in one batch; same latents as one line at a time (to the
    run-to-run spread of the host network's convolutions)."""
    from style_transfer_amd import run as cli
    lines = ["c1.jpg a.png b.png", "c2.jpg a.png b.png", "c1.jpg a2.png b.png", "c2.jpg a2.png b2.png", "c1.jpg a.png b2.png"]
    root, pre = make_root(tmp_path, lines, masks=("c1",))
    np.save(str(root / "cnt" / "c2_mask.npy"), 1.0 - load_masks()[case()["mask"]])       # another mask for the second content
    load = lambda name: torch.load(str(tmp_path / name / "latents_all.pt"), weights_only=True)
    cli.main(cli_args(tmp_path, root, pre, "warm"))
    cli.main(cli_args(tmp_path, root, pre, "one") + ["--graph_unet"])
    cli.main(cli_args(tmp_path, root, pre, "one2") + ["--graph_unet"])
    want = load("one")
    spread = (load("one2") - want).abs().max().item()
    cli.main(cli_args(tmp_path, root, pre, "lanes") + ["--lanes", "3"])
    got = load("lanes")
    assert tuple(got.shape) == (5, 4, 16, 16) and torch.isfinite(got).all()
    diff = (got - want).abs().max().item()
    print("lanes: max |latent difference| %.2e (two identical runs: %.2e)" % (diff, spread))
    # (the lanes run the least-work decomposition of the attention calls: another cut of the key runs, so the
    # difference is that of two decompositions, 4e-3 of the largest output per call at most)
    assert diff <= max(4 * spread, 5e-3 * want.abs().max().item())
    assert not torch.equal(got[0], got[2]) and not torch.equal(got[0], got[1])


def test_network_autocast_flag(tmp_path):
    """`--network_autocast`: the host network's convolutions and linears in bf16, the attention kernels on bf16
    activations; two-style lines (one at a time and in lanes) and a single-style line stay within 40 dB of the
    full-precision run (the reference-loop pin of this mode is in test_gpu_unet)."""
    from style_transfer_amd import run as cli
    root, pre = make_root(tmp_path, ["c1.jpg a.png b.png", "c2.jpg a.png b.png"], masks=("c1", "c2"))
    load = lambda name: torch.load(str(tmp_path / name / "latents_all.pt"), weights_only=True).numpy()
    psnr = lambda ref, got: 10 * np.log10(float(ref.max() - ref.min()) ** 2 / max(float(((ref - got) ** 2).mean()), 1e-300))
    cli.main(cli_args(tmp_path, root, pre, "full"))
    cli.main(cli_args(tmp_path, root, pre, "auto") + ["--network_autocast"])
    cli.main(cli_args(tmp_path, root, pre, "auto_lanes") + ["--network_autocast", "--lanes", "2"])
    want = load("full")
    for name in ("auto", "auto_lanes"):
        got = load(name)
        assert got.shape == want.shape and np.isfinite(got).all()
        db = [psnr(want[i], got[i]) for i in range(len(want))]
        print("%s: PSNR to the full-precision run %s dB" % (name, ["%.1f" % x for x in db]))
        assert min(db) >= 40.0
        assert not np.array_equal(got, want)                           # the flag did something
    # a single-style line (no mask file: the StyleID branch) under autocast
    sub = tmp_path / "s"
    root2, pre2 = make_root(sub, ["c1.jpg a.png"])
    cli.main(cli_args(sub, root2, pre2, "full"))
    cli.main(cli_args(sub, root2, pre2, "auto") + ["--network_autocast"])
    one = lambda name: torch.load(str(sub / name / "latents_all.pt"), weights_only=True).numpy()
    assert psnr(one("full")[0], one("auto")[0]) >= 40.0


def test_sweep_of_16_lines_over_two_ranks_with_lanes(tmp_path):
    """The data_vis sweep in small: 16 lines = 4 contents x 2 character x 2 background styles, started the way
    the driver starts ranks (here both ranks on the one GPU of the box, the gather through the host), two
    images in flight per rank, side-cars written by a first pass and read by the second.  Rank 0's gathered
    stack equals the single-process run line by line, and a rank reads each of its files once."""
    import re
    import socket
    import subprocess
    import sys
    from style_transfer_amd import run as cli
    lines = ["c%d.jpg a%d.png b%d.png" % (c, a, b) for a in (1, 2) for b in (1, 2) for c in (1, 2, 3, 4)]
    root, pre = make_root(tmp_path, lines, masks=("c1", "c2", "c3", "c4"))
    args = cli_args(tmp_path, root, pre, "one")
    cli.main(args + ["--sidecar", "write"])
    single = torch.load(str(tmp_path / "one" / "latents_all.pt"), weights_only=True)
    assert tuple(single.shape) == (16, 4, 16, 16) and torch.isfinite(single).all()
    assert len([n for n in os.listdir(str(pre)) if n.endswith(".sidpack")]) == 8
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SID_DIST_BACKEND="gloo", SID_ONE_DEVICE="1", PYTHONPATH=repo + os.pathsep + env.get("PYTHONPATH", ""))
    two = cli_args(tmp_path, root, pre, "two")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), "-m", "style_transfer_amd.run"] + two + ["--lanes", "2"],
                       env=env, cwd=repo, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    m = re.search(r"16 triples on 2 GPU\(s\); rank 0 read (\d+) files \(([\d.]+) GB, (\d+) from side-cars\), (\d+) served from HBM", r.stdout)
    assert m, r.stdout[-500:]
    files, _, from_side, hits = int(m.group(1)), float(m.group(2)), int(m.group(3)), int(m.group(4))
    # rank 0 takes lines 0, 2, 4, ...: contents c1 and c3, all four styles: 6 files, each read once, all from side-cars
    assert files == 6 and from_side == 6 and hits == 8 * 3 - 6
    both = torch.load(str(tmp_path / "two" / "latents_all.pt"), weights_only=True)
    assert tuple(both.shape) == (16, 4, 16, 16)
    assert (both - single).abs().max().item() < 5e-3 * single.abs().max().item()
    for ln in lines:
        assert os.path.isfile(str(tmp_path / "two" / ("_".join(os.path.splitext(n)[0] for n in ln.split()) + ".pt")))


def test_loader_and_graph_capture_take_turns(tmp_path):
    """The loader's thread reads, uploads and packs while the sampling thread captures UNet evaluations over and
    over.  Left to overlap, a blocking copy of the loader fails now and then with "operation not permitted when
    stream is capturing" (once in a few thousand copies, whatever capture mode the threads are in), so the two
    take turns through ops.capture_lock: GraphedUNet holds it while it captures, the loader takes it one list
    entry at a time.  Every load must succeed and hand out the bytes a plain load gives."""
    from style_transfer_amd import unet as U
    from style_transfer_amd.graphed import GraphedUNet
    root, pre = make_root(tmp_path, ["c1.jpg a.png b.png"])
    want = fs.FeatureStore.load(str(pre), "c1", "a", "b", "cuda:0").pack(8, "f16", T=1.5)
    model = U.UNetModel(**case()["cfg"]).eval().to("cuda:0")
    model.set_content_name(str(root / "cnt" / "nomask.jpg"))
    x, t = torch.randn(1, 4, 16, 16, device="cuda:0"), torch.tensor([481], device="cuda:0")
    ctx = torch.randn(1, 77, 768, device="cuda:0")
    ld = FeatureLoader(str(pre), "cuda:0", heads=8, compute="f16", T=1.5, sidecar="off", prefetch=True, cache_bytes=0)
    loads = 0
    with torch.no_grad():
        ld.request(("c1", "a", "b"))
        import time
        for i in range(40):
            GraphedUNet(model)(x, t, ctx)                        # warm-up + capture, under the lock
            time.sleep(0.2)                                      # (the sampling of an image: the loader's turn)
            if all(f.done() for f in ld._pending.values()):
                got = ld.get(("c1", "a", "b"))                   # raises if any call of the load failed
                loads += 1
                if loads == 1:
                    torch.cuda.synchronize()
                    same_store(got, want)
                del got
                ld.request(("c1", "a", "b"))                     # cache_bytes = 0: the files are read again
    ld.get(("c1", "a", "b"))
    ld.close()
    assert loads >= 2 and ld.stats["files"] == 3 * (loads + 1)
