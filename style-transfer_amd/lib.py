"""ctypes binding of libsid.so, the C ABI declared in include/sid.h.

The library is the product path: there is no CPU or PyTorch fallback.  If libsid.so is missing
or an entry point fails, the caller gets an exception that says so.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SID_LIB_OVERRIDE") or os.path.join(HERE, "libsid.so")   # override: kernel experiments only

SID_OK = 0
DTYPE_F32, DTYPE_BF16 = 0, 1
COMPUTE_BF16, COMPUTE_F32X3, COMPUTE_F16QK = 0, 1, 2
LAYOUT_BHND, LAYOUT_BNHD = 0, 1
STATIC_SCRATCH_BYTES = 64 << 20

_c = ctypes
_vp, _i, _f, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t

# name -> (restype, argtypes): every symbol include/sid.h declares
SIGNATURES = {
    "sid_last_error": (_c.c_char_p, []),
    "sid_version": (_i, []),
    "sid_injected_attn_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "sid_injected_attn_fwd": (_i, [_vp] * 8 + [_vp, _vp] + [_i] * 4 + [_f] * 4 + [_i] * 3 + [_vp, _sz, _vp]),
    "sid_injected_attn_fwd_static": (_i, [_vp] * 8 + [_vp, _vp] + [_i] * 4 + [_f] * 4 + [_i] * 3 + [_vp]),
    "sid_kv_image_bytes": (_sz, [_i] * 5),
    "sid_pack_kv": (_i, [_vp, _vp, _vp, _sz] + [_i] * 6 + [_vp]),
    "sid_injected_attn_packed_workspace_bytes": (_sz, [_i] * 5),
    "sid_injected_attn_fwd_packed": (_i, [_vp] * 5 + [_vp, _vp] + [_i] * 4 + [_f] * 4 + [_i] * 3 + [_vp, _sz, _vp]),
    "sid_content_partial_bytes": (_sz, [_i] * 5),
    "sid_content_partial_workspace_bytes": (_sz, [_i] * 5),
    "sid_content_partial": (_i, [_vp, _vp, _vp, _sz] + [_i] * 4 + [_f] + [_i] * 2 + [_vp, _sz, _vp]),
    "sid_injected_attn_cached_workspace_bytes": (_sz, [_i] * 5),
    "sid_injected_attn_fwd_cached": (_i, [_vp] * 5 + [_vp, _vp] + [_i] * 4 + [_f] * 4 + [_i] * 3 + [_vp, _sz, _vp]),
    "sid_injected_attn_skip_counts_offset": (_c.c_longlong, [_i] * 7),
    "sid_attn_plan": (_i, [_i] * 8 + [_c.POINTER(_c.c_int)]),
    "sid_plain_attn_workspace_bytes": (_sz, [_i] * 6),
    "sid_plain_attn_fwd": (_i, [_vp] * 4 + [_i] * 5 + [_f] + [_i] * 5 + [_vp, _sz, _vp]),
    "sid_mask_downsample": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "sid_hf_filter": (_i, [_vp] * 4 + [_i] * 4 + [_f, _i, _vp]),
    "sid_hf_filter_radius": (_i, [_vp] * 4 + [_i] * 4 + [_i, _i, _vp]),
    "sid_residual_sum": (_i, [_vp] * 4 + [_c.c_longlong, _i, _vp]),
    "sid_masked_adain": (_i, [_vp] * 4 + [_i, _i, _vp] + [_i] * 5 + [_vp]),
}


class SidError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libsid error %d: %s" % (code, message))
        self.code = code


_lib = None


def load():
    """Load libsid.so (once) and declare the argument types of every entry point."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python -m style_transfer_amd.build` "
                "(hipcc --offload-arch=gfx950); this package has no fallback path" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is None:
                if os.environ.get("SID_LIB_OVERRIDE"):
                    continue          # an older experimental build may predate an entry point
                raise AttributeError("%s does not export %s" % (LIB_PATH, name))
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(code):
    if code != SID_OK:
        raise SidError(code, load().sid_last_error().decode("utf-8", "replace"))
