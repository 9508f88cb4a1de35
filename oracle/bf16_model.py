"""Error model of bf16 matrix-product operands for the injected attention (numpy, float64).

TEST INFRASTRUCTURE ONLY, like everything under oracle/.  This is NOT a restatement of the reference:
it is the reference arithmetic (sid_oracle._rows_impl, ldm/modules/attention.py:370-546) with every
operand of the two matrix products rounded to bfloat16 and everything else exact.  ANY implementation
that feeds bf16 operands to its matrix cores (this library's SID_COMPUTE_BF16 mode, or the reference
itself run under bf16 autocast) differs from the fp32 reference by about what this model differs by;
tests use it to tell that irreducible operand-rounding error from a kernel defect.
"""
import numpy as np

LOG2E = np.float32(1.4426950408889634)


def bf16_round(a):
    """Round float32 values to the nearest bfloat16 (ties to even), returned as float64."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) << np.uint64(16)
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def f16_round(a):
    """Round float32 values to the nearest IEEE half (saturating at +-65504), returned as float64."""
    a = np.clip(np.asarray(a, dtype=np.float32), -65504.0, 65504.0)
    return a.astype(np.float16).astype(np.float64)


def _split_heads(x, heads):
    n, c = x.shape
    return np.ascontiguousarray(x.reshape(n, heads, c // heads).transpose(1, 0, 2))


def injected_attention_rows(rows, xq, q_cnt, k1, v1, k2, v2, k_cnt, v_cnt, m, gamma, T, scale, q_self_coef=0.5,
                            qk="bf16"):
    """Rows `rows` of the injected attention with rounded product operands -> (len(rows), H*d).
    qk: rounding of the scaled queries and the keys, "bf16" or "f16" (SID_COMPUTE_F16QK); the probabilities
    and values are rounded to bf16 in both."""
    f32 = np.float32
    rq = bf16_round if qk == "bf16" else f16_round
    rows = np.asarray(rows)
    H, N, d = q_cnt.shape
    qc = np.asarray(q_cnt, f32)[:, rows]
    q_ = _split_heads(np.asarray(xq, f32)[rows], H)
    q = (qc * f32(gamma)).astype(f32) + (q_ * f32(q_self_coef)).astype(f32)         # f32 blend
    c2s = f32(f32(T) * f32(scale)) * LOG2E
    c2c = f32(scale) * LOG2E
    qs = rq((q * c2s).astype(f32))                                                    # query images carry the logit scale
    qcs = rq((qc * c2c).astype(f32))
    S1 = np.einsum("hid,hjd->hij", qs, rq(k1))
    S2 = np.einsum("hid,hjd->hij", qs, rq(k2))
    SC = np.einsum("hid,hjd->hij", qcs, rq(k_cnt))
    mm = np.asarray(m, np.float64).reshape(-1)[rows].reshape(1, -1, 1)
    min_cc = SC.min(-1, keepdims=True) / float(c2c) * float(scale)
    b1 = (min_cc - S1.max(-1, keepdims=True) / float(c2s)) * (-mm) * float(LOG2E)
    b2 = (min_cc - S2.max(-1, keepdims=True) / float(c2s)) * (mm) * float(LOG2E)
    L = np.concatenate([S1 + b1, S2 + b2, SC], axis=2)
    P = np.exp2(L - L.max(-1, keepdims=True))
    P = bf16_round(P.astype(f32))
    V = np.concatenate([bf16_round(v1), bf16_round(v2), bf16_round(v_cnt)], axis=1)
    o = np.einsum("hij,hjd->hid", P, V) / P.sum(-1, keepdims=True)
    h, n, dd = o.shape
    return np.ascontiguousarray(o.transpose(1, 0, 2).reshape(n, h * dd))


def plain_attention_rows(rows, q, k, v, logit_scale, qk="bf16"):
    """Rows of softmax((q k^T) * logit_scale) v with rounded operands.  q (H,Nq,d), k, v (H,Nk,d)."""
    f32 = np.float32
    rows = np.asarray(rows)
    rq = bf16_round if qk == "bf16" else f16_round
    c2 = f32(logit_scale) * LOG2E
    qs = rq((np.asarray(q, f32)[:, rows] * c2).astype(f32))
    S = np.einsum("hid,hjd->hij", qs, rq(k))
    P = bf16_round(np.exp2(S - S.max(-1, keepdims=True)).astype(f32))
    o = np.einsum("hij,hjd->hid", P, bf16_round(v)) / P.sum(-1, keepdims=True)
    h, n, dd = o.shape
    return np.ascontiguousarray(o.transpose(1, 0, 2).reshape(n, h * dd))
