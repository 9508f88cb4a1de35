"""Feature-store loader: the three feature files of a triple on their way from disk to HBM.

The reference unpickles three 3.5 GB files for every triple, one after the other, before it samples
(run_ori.py:168-222).  At 10.6 GB per triple that, not the sampling, bounds images per second, so this
loader takes the files off the sampling path:

  * a worker thread reads the NEXT triple's files while the current one samples: file -> HBM on a side
    stream (a side-car through two pinned staging slots) -> `sid_pack_kv` on that stream; the sampling
    stream only waits for the event that marks a file ready.  The worker touches the GPU one list entry at
    a time under `ops.capture_lock`, which the sampling thread holds while it captures a HIP graph;
  * every loaded (and packed) list stays in HBM under an LRU bound, keyed by file: a sweep that pairs one
    style with many contents (or one content with many style pairs) reads each file once;
  * a packed side-car (`{base}_cnt.sidpack` / `{base}_sty.sidpack`, feature_store.write_sidecar) holds what
    the kernels read, as they read it: no unpickling, no conversion, 0.43 of the bytes for a style file
    (measured on SD-v1-sized files in tmpfs: 12 GB/s from file to HBM against 2.7 GB/s through the pickle).

Results do not depend on which route a list took: the side-car is the bytes `pack` produced.
"""
import os
import threading
import time
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor

import torch

from . import feature_store as fs
from . import ops


class _Loaded:
    def __init__(self, feats, blocks, event, hbm_bytes, file_bytes, seconds, route):
        self.feats, self.blocks, self.event = feats, blocks, event
        self.hbm_bytes, self.file_bytes, self.seconds, self.route = hbm_bytes, file_bytes, seconds, route


class FeatureLoader:
    """`request(bases)` starts loading what `get(bases)` will hand out; bases = (content, char style,
    back style or None) file stems.  `get` returns a FeatureStore whose lists are ready on the calling
    stream."""

    def __init__(self, precomputed, device, heads=8, compute="f16", T=None, dtype=None, packed=True,
                 cache_content=False, cache_bytes=96 << 30, sidecar="use", prefetch=True, staging_bytes=96 << 20):
        if sidecar not in ("off", "use", "write"):
            raise ValueError("sidecar must be 'off', 'use' or 'write'")
        self.precomputed, self.device = precomputed, torch.device(device)
        self.heads, self.compute, self.T, self.dtype = heads, compute, T, dtype
        self.packed, self.cache_content = packed, cache_content
        self.cache_bytes, self.sidecar, self.prefetch = cache_bytes, sidecar, prefetch
        self._cache = OrderedDict()                 # (role, base) -> _Loaded, least recently used first
        self._pending = {}                          # (role, base) -> Future
        self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="sid-loader") if prefetch else None
        self._stream = None
        self._staging = None
        self._staging_bytes = staging_bytes
        self._lock = threading.Lock()               # one load at a time touches the staging buffers
        self.stats = {"files": 0, "hits": 0, "file_bytes": 0, "load_seconds": 0.0, "wait_seconds": 0.0,
                      "routes": {"pkl": 0, "sidecar": 0}, "evicted": 0}

    # ---- what to load -------------------------------------------------------------------------
    @staticmethod
    def _keys(bases):
        cnt, char, back = bases
        keys = [("cnt", cnt), ("sty", char)]
        if back is not None and back != char:
            keys.append(("sty", back))
        return keys

    def _path(self, key):
        role, base = key
        return os.path.join(self.precomputed, base + ("_cnt.pkl" if role == "cnt" else "_sty.pkl"))

    def request(self, bases):
        """Start loading the files of `bases` that are neither resident nor on their way."""
        if self._pool is None:
            return
        for key in self._keys(bases):
            if key not in self._cache and key not in self._pending:
                self._pending[key] = self._pool.submit(self._load, key, bool(bases[2] is not None))

    def get(self, bases):
        two_styles = bases[2] is not None
        got = {}
        t0 = time.perf_counter()
        for key in self._keys(bases):
            ent = self._cache.get(key)
            if ent is not None:
                self.stats["hits"] += 1
                self._cache.move_to_end(key)
            else:
                fut = self._pending.pop(key, None)
                ent = fut.result() if fut is not None else self._load(key, two_styles)
                self._cache[key] = ent
                self.stats["files"] += 1
                self.stats["file_bytes"] += ent.file_bytes
                self.stats["load_seconds"] += ent.seconds
                self.stats["routes"][ent.route] += 1
            got[key] = ent
        self.stats["wait_seconds"] += time.perf_counter() - t0
        cur = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        for ent in got.values():
            if cur is not None and ent.event is not None:
                cur.wait_event(ent.event)
                for b in ent.blocks:          # allocated on the loader's stream, read on this one from here on
                    b.record_stream(cur)
        self._evict(set(got))
        cnt, char, back = bases
        return fs.FeatureStore(got[("cnt", cnt)].feats, got[("sty", char)].feats,
                               got[("sty", back)].feats if two_styles else None)

    def _evict(self, keep):
        if self.cache_bytes <= 0:              # keep nothing: every line reads its files again (the reference's order)
            self.stats["evicted"] += len(self._cache)
            self._cache.clear()
            return
        total = sum(e.hbm_bytes for e in self._cache.values())
        for key in list(self._cache):
            if total <= self.cache_bytes:
                break
            if key in keep:
                continue
            total -= self._cache.pop(key).hbm_bytes
            self.stats["evicted"] += 1

    def resident_bytes(self):
        return sum(e.hbm_bytes for e in self._cache.values())

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        self._pending.clear()

    # ---- one file -----------------------------------------------------------------------------
    def _ensure_stream(self):
        if self.device.type == "cuda" and self._stream is None:
            with ops.capture_lock:
                self._stream = torch.cuda.Stream(device=self.device)
                # two pinned staging slots: slot k is refilled while slot k^1 is still on its way to HBM
                self._staging = [[torch.empty(self._staging_bytes, dtype=torch.uint8).pin_memory(), None] for _ in range(2)]
            self._slot = 0

    def _load(self, key, two_styles):
        """Runs on the worker thread (or on the caller's when prefetch is off): returns a _Loaded whose
        event marks the last upload and pack launch on the loader's stream."""
        role, base = key
        pkl = self._path(key)
        want_packed = self.packed and two_styles
        t0 = time.perf_counter()
        with self._lock:
            if self.device.type == "cuda":
                torch.cuda.set_device(self.device)
            self._ensure_stream()
            side = fs.sidecar_path(pkl)
            feats = blocks = None
            route = "pkl"
            if want_packed and self.sidecar != "off" and os.path.isfile(side):
                try:
                    rd = fs.SidecarReader(side)
                    if rd.matches(pkl, role, self.heads, self.compute, self.T):
                        feats, blocks, nfile = self._read_sidecar(rd)
                        route = "sidecar"
                except (ValueError, OSError, KeyError):
                    feats = None                       # unreadable or stale: the pickle is the truth
            if feats is None:
                if not os.path.isfile(pkl):
                    raise FileNotFoundError("feature file not found: %s" % pkl)
                feats, blocks, nfile = self._read_pickle(pkl, role, want_packed)
            event = None
            if self._stream is not None:
                # entry by entry, each under the capture lock: a graph capture on the sampling thread waits for
                # one entry's launches at most, and this thread stays off the GPU while it captures
                for e in feats:
                    with ops.capture_lock, torch.cuda.stream(self._stream):
                        if want_packed and route == "pkl":
                            fs.pack_feature_list([e], self.heads, self.compute, self.T)
                        if want_packed and role == "cnt" and self.cache_content:
                            fs.cache_content_blocks([e], self.heads)
                with ops.capture_lock:
                    event = torch.cuda.Event()
                    event.record(self._stream)
                if want_packed and route == "pkl" and self.sidecar == "write":
                    with ops.capture_lock:
                        self._stream.synchronize()
                        fs.write_sidecar(side, feats, role, self.heads, self.compute, pkl)
            blocks = self._live_blocks(feats)
            hbm = sum(b.untyped_storage().nbytes() for b in blocks)
        return _Loaded(feats, blocks or [], event, hbm, nfile, time.perf_counter() - t0, route)

    @staticmethod
    def _live_blocks(feats):
        """The allocations a list keeps alive (one per distinct storage)."""
        seen, out = set(), []
        for e in feats:
            for v in e.values():
                ts = [v] if torch.is_tensor(v) else ([v.image] + ([v.partial.data] if v.partial is not None else [])
                                                     if isinstance(v, ops.PackedKV) else [])
                for t in ts:
                    sp = t.untyped_storage().data_ptr()
                    if sp not in seen:
                        seen.add(sp)
                        out.append(t)
        return out

    def _upload(self, pieces, total):
        """`pieces`: [(offset, cpu uint8 tensor or (file, file offset, nbytes))] laid out in a buffer of `total`
        bytes.  Returns the device buffer; the copy runs on the loader's stream."""
        if self._stream is None:                     # CPU store (tests): no staging
            flat = torch.empty(total, dtype=torch.uint8)
            for off, src in pieces:
                if torch.is_tensor(src):
                    flat[off:off + src.numel()] = src
                else:
                    f, fo, n = src
                    f.seek(fo)
                    f.readinto(memoryview(flat[off:off + n].numpy()))
            return flat
        if total > self._staging_bytes:
            self._staging_bytes = (total + (16 << 20)) // (16 << 20) * (16 << 20)
            with ops.capture_lock:
                self._stream.synchronize()
                self._staging = [[torch.empty(self._staging_bytes, dtype=torch.uint8).pin_memory(), None] for _ in range(2)]
        slot = self._staging[self._slot]
        self._slot ^= 1
        if slot[1] is not None:
            with ops.capture_lock:
                slot[1].synchronize()                # the slot's previous upload has left it
        host = slot[0]
        for off, src in pieces:
            if torch.is_tensor(src):
                host[off:off + src.numel()].copy_(src)
            else:
                f, fo, n = src
                f.seek(fo)
                got = f.readinto(memoryview(host[off:off + n].numpy()))
                if got != n:
                    raise OSError("short read: %d of %d bytes" % (got, n))
        with ops.capture_lock, torch.cuda.stream(self._stream):
            flat = torch.empty(total, dtype=torch.uint8, device=self.device)
            flat.copy_(host[:total], non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(self._stream)
        return flat

    def _read_sidecar(self, rd):
        feats = []
        with open(rd.path, "rb", buffering=0) as f:
            for i in range(len(rd.entries)):
                fo, n = rd.entry_range(i)
                flat = self._upload([(0, (f, fo, n))], n) if n else torch.empty(0, dtype=torch.uint8, device=self.device)
                entry = rd.build_entry(i, flat)
                if self.dtype is not None:
                    self._cast(entry)
                feats.append(entry)
        return feats, None, rd.start + rd.header["payload_bytes"]

    def _cast(self, entry):
        for k, v in entry.items():
            if torch.is_tensor(v) and v.is_floating_point() and k != "z_enc" and v.dtype != self.dtype:
                if self._stream is not None:
                    with ops.capture_lock, torch.cuda.stream(self._stream):
                        entry[k] = v.to(self.dtype)
                else:
                    entry[k] = v.to(self.dtype)

    def _read_pickle(self, pkl, role, want_packed):
        """The reference's file through the allow-list unpickler onto the host (no GPU call: this part runs
        whatever the sampling thread does), then entry by entry into HBM.  A packed style list never uploads its
        queries (nothing reads them); keys and values are freed when they have been packed."""
        host = fs.load_feature_file(pkl, "cpu", None)
        feats = []
        for e in host:
            out = {}
            keep = [(k, v) for k, v in e.items()
                    if not (want_packed and role == "sty" and torch.is_tensor(v) and k.endswith("_self_attn_q"))]
            if self._stream is not None:
                with ops.capture_lock, torch.cuda.stream(self._stream):
                    for k, v in keep:
                        out[k] = v.to(self.device).contiguous() if torch.is_tensor(v) else v
            else:
                for k, v in keep:
                    out[k] = v.to(self.device).contiguous() if torch.is_tensor(v) else v
            if self.dtype is not None:
                self._cast(out)
            feats.append(out)
        return feats, None, os.path.getsize(pkl)
