"""Counter-based synthetic data: bit-identical on every machine.

Golden fixtures store only seeds for their inputs, so the container that ran the
reference and the GPU box that runs the parity tests must regenerate the very same
float32 values.  Library RNGs (numpy / torch) do not promise that across versions or
devices, and transcendental-based transforms (Box-Muller) can differ in the last bit
between libm builds.  Here a value depends only on (seed, flat index) through integer
hashing plus one exact float64 multiply:

    bits = splitmix64(seed * GOLDEN + index)          # 64-bit integer hash
    s    = sum of the four 16-bit fields of `bits`    # Irwin-Hall(4), integer
    z    = float32((s - 131070) * sqrt(3) / 65536)    # mean 0, variance ~1, |z| <= 3.47
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_SCALE = float(np.sqrt(3.0) / 65536.0)


def _splitmix64(x):
    x = (x + _GOLDEN).astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * _M1
    x = (x ^ (x >> np.uint64(27))) * _M2
    return x ^ (x >> np.uint64(31))


def hash_u64(seed, n, offset=0):
    """n 64-bit hashes for flat indices offset .. offset+n-1 of stream `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        base = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) * _GOLDEN
        return _splitmix64(base + idx)


def normal(seed, shape, sigma=1.0):
    """float32 array of `shape`, approximately N(0, sigma^2), from stream `seed`."""
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    chunk = 1 << 22
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        b = hash_u64(seed, m, lo)
        s = ((b & np.uint64(0xFFFF)) + ((b >> np.uint64(16)) & np.uint64(0xFFFF))
             + ((b >> np.uint64(32)) & np.uint64(0xFFFF)) + (b >> np.uint64(48)))
        z = (s.astype(np.int64) - 131070).astype(np.float64) * (_SCALE * float(sigma))
        out[lo:lo + m] = z.astype(np.float32)
    return out.reshape(shape)


def uniform(seed, shape):
    """float32 array of `shape`, uniform on [0, 1) with 24-bit resolution."""
    n = int(np.prod(shape))
    b = hash_u64(seed, n)
    return ((b >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32).reshape(shape)


def disc_mask(side=512, radius_frac=0.3):
    """Binary float32 (side, side) mask: 1 inside a centred disc of radius radius_frac*side."""
    y, x = np.mgrid[0:side, 0:side]
    c = (side - 1) / 2.0
    return (((y - c) ** 2 + (x - c) ** 2) <= (radius_frac * side) ** 2).astype(np.float32)


def half_plane_mask(side=512):
    """Binary float32 (side, side) mask: 1 on the left half."""
    m = np.zeros((side, side), dtype=np.float32)
    m[:, : side // 2] = 1.0
    return m


def attention_inputs(seed, heads, n_tokens, head_dim, sigma_qk=1.0, sigma_v=1.0, batch=1):
    """The eight tensors of one injected-attention call.

    Returns a dict with `xq` of shape (batch, N, heads*head_dim) -- the layout a Linear
    query projection produces -- and `q_cnt, k1, v1, k2, v2, k_cnt, v_cnt` of shape
    (batch*heads, N, head_dim), the layout of the precomputed feature lists.
    """
    bh = batch * heads
    names = ["q_cnt", "k1", "v1", "k2", "v2", "k_cnt", "v_cnt"]
    out = {"xq": normal(seed * 16 + 0, (batch, n_tokens, heads * head_dim), sigma_qk)}
    for i, nm in enumerate(names):
        sig = sigma_v if nm.startswith("v") else sigma_qk
        out[nm] = normal(seed * 16 + 1 + i, (bh, n_tokens, head_dim), sig)
    return out


def param_init(name, shape):
    """Deterministic parameter values from the parameter's NAME (so two implementations of one
    network that agree on names get identical weights without a checkpoint): matrices and
    convolutions ~ N(0, 1/fan_in), normalisation scales ~ 1 + 0.1 N(0,1), biases ~ 0.05 N(0,1)."""
    import zlib
    seed = zlib.crc32(name.encode()) & 0x7FFFFFFF
    shape = tuple(shape)
    if name.endswith("weight") and len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return normal(seed, shape, sigma=fan_in ** -0.5)
    if name.endswith("weight"):
        return (1.0 + normal(seed, shape, sigma=0.1)).astype(np.float32)
    return normal(seed, shape, sigma=0.05)


def stub_eps(x, t, c, injected_features=None):
    """Closed-form stand-in for a UNet noise prediction (torch tensors in and out): a fixed linear map
    of the latent plus terms in the timestep, the conditioning and the step's feature dict, so sampler
    loops can be compared between implementations without network weights."""
    import torch
    e = 0.6 * torch.roll(x, 1, -1) - 0.25 * x + 0.05 * torch.sin(t.to(x.dtype) / 100.0).reshape(-1, 1, 1, 1)
    if c is not None:
        e = e + 0.1 * c.to(x.dtype).mean()
    if injected_features is not None:
        e = e + float(injected_features["shift"])
    return e
