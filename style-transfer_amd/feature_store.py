"""Feature store: the `precomputed_feats_k/{base}_cnt.pkl` / `{base}_sty.pkl` files.

Layout (generate_pkl_only.py:101-142, run_ori.py:168-222): a pickled `list[50]`; entry i belongs to
DDIM timestep 981 - 20*i and is a dict with 'config' ({'T': 1.5}), 'z_enc' (1,4,64,64) and the 18
tensors `output_block_{6..11}_self_attn_{q,k,v}` of shape (heads, N, d) — (8,1024,80) for blocks
6-8, (8,4096,40) for 9-11.  The reference writes them with plain `pickle.dump` of CUDA tensors and
reads them with `pickle.load`, which executes whatever the file says.  This loader reads the same
bytes with an allow-list unpickler: only tensor reconstruction is permitted, and tensor storages
are decoded by `torch.load(..., weights_only=True)` onto the requested device, so a file written on
a CUDA machine loads on ROCm (or CPU) and a file that is not a feature list is refused.
"""
import io
import json
import os
import pickle
import struct

import torch

N_STEPS = 50
INJECT_BLOCKS = tuple(range(6, 12))

_DTYPES = {"float32", "float16", "bfloat16", "float64", "int64", "int32", "int16", "int8", "uint8", "bool"}
_STORAGES = {"FloatStorage", "HalfStorage", "BFloat16Storage", "DoubleStorage", "LongStorage", "IntStorage",
             "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage", "UntypedStorage"}


class _SafeUnpickler(pickle.Unpickler):
    def __init__(self, file, device):
        super().__init__(file)
        self._device = device

    def _load_storage(self, b):
        return torch.load(io.BytesIO(b), map_location=self._device, weights_only=True)

    def find_class(self, module, name):
        if module == "torch._utils" and name in ("_rebuild_tensor_v2", "_rebuild_tensor"):
            return getattr(torch._utils, name)
        if module == "torch.storage" and name == "_load_from_bytes":
            return self._load_storage
        if module == "torch" and (name in _DTYPES or name in _STORAGES or name in ("Size", "device")):
            return getattr(torch, name)
        if module == "collections" and name == "OrderedDict":
            import collections
            return collections.OrderedDict
        raise pickle.UnpicklingError("feature file refers to %s.%s, which a feature list does not need; refusing to load it"
                                     % (module, name))


def load_feature_file(path, device="cuda", dtype=None):
    """Read one `*_cnt.pkl` / `*_sty.pkl` into `list[dict]` with every tensor on `device`
    (optionally cast to `dtype`, e.g. torch.bfloat16 to halve the HBM the store occupies)."""
    with open(path, "rb") as f:
        feats = _SafeUnpickler(f, torch.device(device)).load()
    validate_layout(feats)
    dev = torch.device(device)
    for entry in feats:
        for k, v in entry.items():
            if torch.is_tensor(v):
                v = v.to(dev)
                if dtype is not None and v.is_floating_point() and k != "z_enc":
                    v = v.to(dtype)
                entry[k] = v.contiguous()
    return feats


def save_feature_file(path, feats):
    """Write a feature list in the reference's layout (plain pickle of the list of dicts,
    generate_pkl_only.py:209-210), readable by the reference's `pickle.load` and by load_feature_file."""
    validate_layout(feats)
    with open(path, "wb") as f:
        pickle.dump(feats, f)


def validate_layout(feats):
    """Raise ValueError unless `feats` looks like a feature list: a list of dicts whose tensor entries
    are `z_enc` (4-D) or `output_block_{i}_self_attn_{q,k,v}` (3-D, the three of a block alike)."""
    if not isinstance(feats, list) or not feats:
        raise ValueError("a feature file holds a non-empty list (one dict per DDIM step), got %s" % type(feats).__name__)
    for i, entry in enumerate(feats):
        if not isinstance(entry, dict):
            raise ValueError("step %d is %s, not a dict" % (i, type(entry).__name__))
        shapes = {}
        for k, v in entry.items():
            if k == "config":
                if not isinstance(v, dict):
                    raise ValueError("step %d: 'config' must be a dict" % i)
                continue
            if not torch.is_tensor(v):
                raise ValueError("step %d: entry %r is %s, not a tensor" % (i, k, type(v).__name__))
            if k == "z_enc":
                if v.dim() != 4:
                    raise ValueError("step %d: z_enc must be (1,C,H,W), got %s" % (i, tuple(v.shape)))
                continue
            parts = k.rsplit("_self_attn_", 1)
            if len(parts) != 2 or parts[1] not in ("q", "k", "v") or not parts[0].startswith("output_block_"):
                raise ValueError("step %d: unexpected key %r" % (i, k))
            if v.dim() != 3:
                raise ValueError("step %d: %s must be (heads, N, d), got %s" % (i, k, tuple(v.shape)))
            shapes.setdefault(parts[0], set()).add(tuple(v.shape))
        for blk, sh in shapes.items():
            if len(sh) != 1:
                raise ValueError("step %d: q/k/v of %s disagree in shape: %s" % (i, blk, sorted(sh)))


RESIDUAL_SUFFIXES = ("_cnt_h", "_cnt_skip", "_residual")


def validate_residuals(residuals_all):
    """Raise ValueError unless `residuals_all` has the layout of `{base}_residuals_all.pkl`
    (generate_pkl_only.py:144-178, 310-316): `{int timestep: {"output_block_{i}_cnt_h" |
    "..._cnt_skip" | "..._residual": (1,C,H,W) tensor}}`."""
    if not isinstance(residuals_all, dict):
        raise ValueError("a residuals file holds a dict keyed by timestep, got %s" % type(residuals_all).__name__)
    for t, entry in residuals_all.items():
        if not isinstance(t, int) or not isinstance(entry, dict):
            raise ValueError("residuals: key %r must be an int timestep mapping to a dict" % (t,))
        for k, v in entry.items():
            if not (k.startswith("output_block_") and k.endswith(RESIDUAL_SUFFIXES)):
                raise ValueError("residuals[%d]: unexpected key %r" % (t, k))
            if not torch.is_tensor(v) or v.dim() != 4:
                raise ValueError("residuals[%d][%r] must be a (B,C,H,W) tensor" % (t, k))


def save_residuals_file(path, residuals_all):
    """Write `{base}_residuals_all.pkl` as the reference does (plain pickle of the dict of CPU tensors,
    generate_pkl_only.py:310-316)."""
    validate_residuals(residuals_all)
    with open(path, "wb") as f:
        pickle.dump({int(t): {k: v.detach().cpu() for k, v in e.items()} for t, e in residuals_all.items()}, f)


def load_residuals_file(path, device="cuda"):
    """Read `{base}_residuals_all.pkl` with the allow-list unpickler; tensors land on `device` (the
    reference keeps them on the CPU and moves one per ResBlock call, high_frequency_final.py:76)."""
    with open(path, "rb") as f:
        residuals = _SafeUnpickler(f, torch.device(device)).load()
    validate_residuals(residuals)
    dev = torch.device(device)
    return {t: {k: v.to(dev).contiguous() for k, v in e.items()} for t, e in residuals.items()}


class FeatureStore:
    """The three feature lists of one (content, character style, background style) triple, resident
    on one device, and the merged per-step dicts the UNet consumes (run_ori.py:36-75, 235).
    `back` None: a (content, style) pair, the single-style path (hf_run.py:40-68, 383-455)."""

    def __init__(self, cnt, char, back=None):
        self.cnt, self.char, self.back = cnt, char, back

    def lists(self):
        return tuple(f for f in (self.cnt, self.char, self.back) if f is not None)

    @classmethod
    def load(cls, precomputed_dir, cnt_base, char_base, back_base, device="cuda", dtype=None, cnt=None):
        """`cnt`: the content list of an earlier triple with the same content image, as that triple's store
        left it (loaded, possibly packed and carrying its cached content blocks), instead of the file."""
        rd = lambda name: load_feature_file(os.path.join(precomputed_dir, name), device, dtype)
        return cls(rd(cnt_base + "_cnt.pkl") if cnt is None else cnt, rd(char_base + "_sty.pkl"),
                   rd(back_base + "_sty.pkl") if back_base is not None else None)

    def z_enc(self):
        """The fully inverted latents (entry 0), cloned as run_ori.py:172,191,210 does."""
        return tuple(torch.clone(f[0]["z_enc"]) for f in self.lists())

    def merged(self, opt, start_step=49):
        from .features import feat_merge, feat_merge_2sty
        if self.back is None:
            return feat_merge(opt, self.cnt, self.char, start_step=start_step)
        return feat_merge_2sty(opt, self.cnt, self.char, self.back, start_step=start_step)

    def pack(self, heads=8, compute="f16", cache_content=False, T=None):
        """Convert every key/value pair of the three lists to the attention kernels' operand layout
        (ops.pack_kv), in place: `..._self_attn_k` and `..._self_attn_v` of a block then hold ONE
        PackedKV object and the two tensors are dropped.  Done once per feature file, it removes the
        per-call conversion of the six K/V tensors from the 49 x 6 attention calls of an image and
        shrinks the store (bf16 operands: 112 bytes per token and head at head_dim 40, against 320
        for the two fp32 tensors).  The queries stay tensors: the kernel blends them per call.
        cache_content: also compute the content block of every (timestep, layer) of the content list
        (ops.content_partial: it reads q, k, v of the content file only) and attach it to the content
        pair; the attention calls then compute the two style blocks alone.  Worth it when the content
        list is reused for several style pairs (pass it to the next FeatureStore.load as `cnt`): the
        partials cost what one image's content blocks cost, 0.66 GB per content image at SD-v1 sizes.
        T: the temperature the calls will use (default: the file's own config['T']); with compute 'f16' a
        layer whose logit scales the half-precision mode refuses is packed for bf16 operands instead."""
        if self.back is None:
            raise ValueError("pre-packed key/value pairs serve the mask-based two-style attention; a single-style "
                             "store stays unpacked")
        for feats in self.lists():
            pack_feature_list(feats, heads, compute, T)
        if cache_content:
            cache_content_blocks(self.cnt, heads)
        return self

    def nbytes(self):
        seen, total = set(), 0
        for f in self.lists():
            for e in f:
                for v in e.values():
                    if torch.is_tensor(v):
                        total += v.numel() * v.element_size()
                    elif hasattr(v, "image") and id(v) not in seen:
                        seen.add(id(v))
                        total += v.nbytes() + (v.partial.nbytes() if v.partial is not None else 0)
        return total


def pack_feature_list(feats, heads=8, compute="f16", T=None):
    """FeatureStore.pack for one list, in place (an already packed entry is left alone)."""
    from . import ops
    from .attention import resolve_injected_compute
    for entry in feats:
        for key in [k for k in entry if k.endswith("_self_attn_k") and torch.is_tensor(entry[k])]:
            vkey = key[:-1] + "v"
            n, d = entry[key].shape[-2:]
            temp = T if T is not None else (entry.get("config") or {}).get("T", 1.0)
            mode = resolve_injected_compute(compute, n, d, temp, d ** -0.5)
            kv = ops.pack_kv(entry[key].contiguous(), entry[vkey].contiguous(), heads, mode)
            entry[key] = entry[vkey] = kv
    return feats


def cache_content_blocks(cnt_feats, heads=8):
    """Attach the content block (ops.content_partial) of every (timestep, layer) to the packed content list."""
    from . import ops
    jobs = []
    for entry in cnt_feats:
        for key in [k for k in entry if k.endswith("_self_attn_k") and isinstance(entry[k], ops.PackedKV)]:
            kv, qkey = entry[key], key[:-1] + "q"
            if kv.partial is None:
                # the partial remembers the tensor it was made from: the call must pass this very one
                q = entry[qkey] = entry[qkey].contiguous()
                jobs.append((q, kv, q.shape[-1] ** -0.5))
    for (_, kv, _), partial in zip(jobs, ops.content_partials(jobs, heads)):
        kv.partial = partial
    return cnt_feats


# ------------------------------------------------------------------------------------------------
# Packed side-car: `{base}_cnt.sidpack` / `{base}_sty.sidpack` beside the pickle
# ------------------------------------------------------------------------------------------------
# What a packed feature list holds in HBM, written as it lies there: a JSON header and one flat payload,
# every item 256-byte aligned, every list entry a contiguous byte range.  Reading it executes nothing and
# converts nothing: a range goes from the file into pinned host memory and from there into one device
# buffer that the entry's tensors and key/value images are views of.  A style list keeps only what the
# attention reads (its key/value images and the inverted latent of entry 0): a third of the pickle.
# No counterpart in the reference (run_ori.py:168-222 unpickles 3.5 GB per file for every triple).
SIDECAR_MAGIC = b"SIDPACK1"
SIDECAR_ALIGN = 256
_TORCH_DTYPES = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}


def sidecar_path(pkl_path):
    return os.path.splitext(pkl_path)[0] + ".sidpack"


def _source_stamp(pkl_path):
    st = os.stat(pkl_path)
    return {"name": os.path.basename(pkl_path), "size": st.st_size, "mtime_ns": st.st_mtime_ns}


def write_sidecar(path, feats, role, heads, compute, pkl_path=None):
    """`feats`: a feature list as pack_feature_list leaves it (key/value pairs as PackedKV).  role "sty"
    drops the queries.  Written to a temporary name and renamed, so a reader never sees half a file."""
    from . import ops
    entries, blobs, off = [], [], 0
    for entry in feats:
        meta, seen = {"config": entry.get("config"), "items": {}, "begin": off}, {}
        for k, v in entry.items():
            if k == "config" or (role == "sty" and k.endswith("_self_attn_q")):
                continue
            if isinstance(v, ops.PackedKV):
                if id(v) in seen:
                    meta["items"][k] = {"kind": "same", "as": seen[id(v)]}
                    continue
                seen[id(v)] = k
                raw = v.image
                item = {"kind": "kv", "B": v.B, "H": v.H, "N": v.N, "d": v.d, "compute": int(v.compute)}
            elif torch.is_tensor(v):
                raw = v.contiguous()
                item = {"kind": "tensor", "dtype": str(v.dtype).replace("torch.", ""), "shape": list(v.shape)}
                if item["dtype"] not in _TORCH_DTYPES:
                    raise ValueError("side-car: %s has dtype %s" % (k, v.dtype))
            else:
                raise ValueError("side-car: entry %r is neither a tensor nor a packed key/value pair" % k)
            nbytes = raw.numel() * raw.element_size()
            item["offset"], item["nbytes"] = off, nbytes
            meta["items"][k] = item
            blobs.append((off, raw))
            off = (off + nbytes + SIDECAR_ALIGN - 1) // SIDECAR_ALIGN * SIDECAR_ALIGN
        meta["end"] = off
        entries.append(meta)
    header = {"version": 1, "role": role, "heads": heads, "compute": compute, "entries": entries, "payload_bytes": off,
              "source": _source_stamp(pkl_path) if pkl_path else None}
    hb = json.dumps(header).encode()
    start = (len(SIDECAR_MAGIC) + 8 + len(hb) + 4095) // 4096 * 4096
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(SIDECAR_MAGIC + struct.pack("<Q", len(hb)) + hb)
        for o, raw in blobs:
            f.seek(start + o)
            f.write(memoryview(raw.detach().cpu().contiguous().view(torch.uint8).reshape(-1).numpy()))
        f.truncate(start + off)
    os.replace(tmp, path)
    return start + off


class SidecarReader:
    """Header of a side-car file and the means to turn byte ranges of its payload into list entries."""

    def __init__(self, path):
        self.path = path
        with open(path, "rb") as f:
            if f.read(len(SIDECAR_MAGIC)) != SIDECAR_MAGIC:
                raise ValueError("%s is not a packed side-car" % path)
            (n,) = struct.unpack("<Q", f.read(8))
            if n > 64 << 20:
                raise ValueError("%s: header of %d bytes" % (path, n))
            self.header = json.loads(f.read(n).decode())
        if self.header.get("version") != 1:
            raise ValueError("%s: side-car version %r" % (path, self.header.get("version")))
        self.start = (len(SIDECAR_MAGIC) + 8 + n + 4095) // 4096 * 4096
        self.entries = self.header["entries"]
        if os.path.getsize(path) < self.start + self.header["payload_bytes"]:
            raise ValueError("%s is truncated" % path)

    def matches(self, pkl_path, role, heads, compute, T=None):
        """Made from this very pickle (name, size, mtime), for this role, head count and arithmetic: the mode of
        every key/value image must be the one a fresh pack would choose now."""
        from . import ops
        from .attention import resolve_injected_compute
        h = self.header
        if h.get("role") != role or h.get("heads") != heads or h.get("compute") != compute:
            return False
        if pkl_path is not None and os.path.exists(pkl_path) and h.get("source") != _source_stamp(pkl_path):
            return False
        for e in self.entries:
            temp = T if T is not None else (e.get("config") or {}).get("T", 1.0)
            for item in e["items"].values():
                if item["kind"] == "kv":
                    want = ops._compute_code(resolve_injected_compute(compute, item["N"], item["d"], temp, item["d"] ** -0.5))
                    if item["compute"] != want:
                        return False
        return True

    def entry_range(self, i):
        """(file offset, bytes) of entry i's contiguous payload range."""
        e = self.entries[i]
        return self.start + e["begin"], e["end"] - e["begin"]

    def build_entry(self, i, flat):
        """Entry i as a dict over `flat`, a uint8 tensor holding the entry's payload range."""
        from . import ops
        e = self.entries[i]
        out = {"config": dict(e["config"])} if e.get("config") is not None else {}
        for k, item in e["items"].items():
            if item["kind"] == "same":
                continue
            raw = flat[item["offset"] - e["begin"]: item["offset"] - e["begin"] + item["nbytes"]]
            if item["kind"] == "kv":
                out[k] = ops.PackedKV(raw, item["B"], item["H"], item["N"], item["d"], item["compute"])
            else:
                out[k] = raw.view(_TORCH_DTYPES[item["dtype"]]).view(item["shape"])
        for k, item in e["items"].items():
            if item["kind"] == "same":
                out[k] = out[item["as"]]
        return out
