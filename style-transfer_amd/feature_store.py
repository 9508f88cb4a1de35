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
import pickle

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
    on one device, and the merged per-step dicts the UNet consumes (run_ori.py:36-75, 235)."""

    def __init__(self, cnt, char, back):
        self.cnt, self.char, self.back = cnt, char, back

    @classmethod
    def load(cls, precomputed_dir, cnt_base, char_base, back_base, device="cuda", dtype=None, cnt=None):
        """`cnt`: the content list of an earlier triple with the same content image, as that triple's store
        left it (loaded, possibly packed and carrying its cached content blocks), instead of the file."""
        import os
        rd = lambda name: load_feature_file(os.path.join(precomputed_dir, name), device, dtype)
        return cls(rd(cnt_base + "_cnt.pkl") if cnt is None else cnt, rd(char_base + "_sty.pkl"), rd(back_base + "_sty.pkl"))

    def z_enc(self):
        """The fully inverted latents (entry 0), cloned as run_ori.py:172,191,210 does."""
        return tuple(torch.clone(f[0]["z_enc"]) for f in (self.cnt, self.char, self.back))

    def merged(self, opt, start_step=49):
        from .features import feat_merge_2sty
        return feat_merge_2sty(opt, self.cnt, self.char, self.back, start_step=start_step)

    def pack(self, heads=8, compute="bf16", cache_content=False):
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
        partials cost what one image's content blocks cost, 0.66 GB per content image at SD-v1 sizes."""
        from . import ops
        for feats in (self.cnt, self.char, self.back):
            for entry in feats:
                for key in [k for k in entry if k.endswith("_self_attn_k") and torch.is_tensor(entry[k])]:
                    vkey = key[:-1] + "v"
                    d = entry[key].shape[-1]
                    # half-precision operands exist for head dims 33-48 and 65-80; other layers take bf16
                    mode = "bf16" if compute == "f16" and not (33 <= d <= 48 or 65 <= d <= 80) else compute
                    kv = ops.pack_kv(entry[key].contiguous(), entry[vkey].contiguous(), heads, mode)
                    entry[key] = entry[vkey] = kv
        if cache_content:
            jobs = []
            for entry in self.cnt:
                for key in [k for k in entry if k.endswith("_self_attn_k") and isinstance(entry[k], ops.PackedKV)]:
                    kv, qkey = entry[key], key[:-1] + "q"
                    if kv.partial is None:
                        # the partial remembers the tensor it was made from: the call must pass this very one
                        q = entry[qkey] = entry[qkey].contiguous()
                        jobs.append((q, kv, q.shape[-1] ** -0.5))
            for (_, kv, _), partial in zip(jobs, ops.content_partials(jobs, heads)):
                kv.partial = partial
        return self

    def nbytes(self):
        seen, total = set(), 0
        for f in (self.cnt, self.char, self.back):
            for e in f:
                for v in e.values():
                    if torch.is_tensor(v):
                        total += v.numel() * v.element_size()
                    elif hasattr(v, "image") and id(v) not in seen:
                        seen.add(id(v))
                        total += v.nbytes() + (v.partial.nbytes() if v.partial is not None else 0)
        return total
