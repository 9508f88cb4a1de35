#!/usr/bin/env python3
"""Feature-store routes on SD-v1-sized synthetic files: bytes per second from file to packed lists in HBM.

    python tools/loader_bench.py [--dir /dev/shm/sid_loader_bench] [--keep]

Writes one content and one style file with save_feature_file (the reference's pickle recipe, 3.54 GB each)
and times (1) the plain reader + pack, as run.py did before the loader existed, (2) the loader's pickle
route (host unpickle, pinned staging, upload and pack on a side stream), (3) the side-car route."""
import argparse
import json
import os
import shutil
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_amd import feature_store as fs  # noqa: E402
from style_transfer_amd.loader import FeatureLoader  # noqa: E402


def sd_feature_list(dev, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    out = []
    for _ in range(50):
        e = {"config": {"T": 1.5}, "z_enc": torch.randn(1, 4, 64, 64, device=dev, generator=g)}
        for b in range(6, 12):
            n, d = (1024, 80) if b < 9 else (4096, 40)
            for t in "qkv":
                e["output_block_%d_self_attn_%s" % (b, t)] = torch.randn(8, n, d, device=dev, generator=g)
        out.append(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="/dev/shm/sid_loader_bench")
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    os.makedirs(args.dir, exist_ok=True)
    rec = {"dir": args.dir}
    t0 = time.perf_counter()
    for name, seed in (("c_cnt.pkl", 1), ("a_sty.pkl", 2), ("b_sty.pkl", 3)):
        fs.save_feature_file(os.path.join(args.dir, name), sd_feature_list(dev, seed))
    rec["write_s"] = time.perf_counter() - t0
    size = {n: os.path.getsize(os.path.join(args.dir, n)) for n in ("c_cnt.pkl", "a_sty.pkl", "b_sty.pkl")}
    total = sum(size.values())
    rec["pickle_GB_per_triple"] = total / 1e9

    def timed(fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t, out

    dt, store = timed(lambda: fs.FeatureStore.load(args.dir, "c", "a", "b", dev).pack(8, "f16", T=1.5))
    rec["plain_reader_then_pack"] = {"s": dt, "GB_per_s": total / dt / 1e9}
    del store
    torch.cuda.empty_cache()
    for mode, sidecar in (("loader_pickle_route", "write"), ("loader_sidecar_route", "use")):
        ld = FeatureLoader(args.dir, dev, heads=8, compute="f16", T=1.5, sidecar=sidecar, prefetch=False)
        if mode == "loader_pickle_route":
            ld.sidecar = "off"
        dt, store = timed(lambda: ld.get(("c", "a", "b")))
        nbytes = ld.stats["file_bytes"]
        rec[mode] = {"s": dt, "file_GB": nbytes / 1e9, "GB_per_s": nbytes / dt / 1e9, "pickle_equivalent_GB_per_s": total / dt / 1e9,
                     "hbm_GB": ld.resident_bytes() / 1e9}
        if mode == "loader_pickle_route":       # now write the side-cars (not timed above)
            ld2 = FeatureLoader(args.dir, dev, heads=8, compute="f16", T=1.5, sidecar="write", prefetch=False)
            dt2, _ = timed(lambda: ld2.get(("c", "a", "b")))
            rec["pickle_route_plus_sidecar_write_s"] = dt2
            ld2.close()
        del store
        ld.close()
        torch.cuda.empty_cache()
    rec["sidecar_GB_per_triple"] = sum(os.path.getsize(os.path.join(args.dir, n)) for n in os.listdir(args.dir) if n.endswith(".sidpack")) / 1e9
    print(json.dumps(rec))
    if not args.keep:
        shutil.rmtree(args.dir, ignore_errors=True)


if __name__ == "__main__":
    main()
