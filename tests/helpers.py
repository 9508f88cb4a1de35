"""Shared fixture access for the test suite (no GPU, no reference access)."""
import json
import os

import numpy as np

from style_transfer_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_masks = None


def load_masks():
    """name -> (512, 512) float32 binary mask, unpacked from the bit-packed fixture."""
    global _masks
    if _masks is None:
        z = np.load(os.path.join(GOLDEN, "masks_packed.npz"))
        _masks = {k: np.unpackbits(z[k]).reshape(512, 512).astype(np.float32) for k in z.files}
    return _masks


def mask_for(kind):
    if kind == "disc":
        return synth.disc_mask()
    if kind == "half":
        return synth.half_plane_mask()
    if kind == "none":
        return None
    return load_masks()[kind]


def attention_cases():
    with open(os.path.join(GOLDEN, "attention_cases.json")) as f:
        return json.load(f)


def attention_golden():
    return np.load(os.path.join(GOLDEN, "attention.npz"))


def hf_cases():
    with open(os.path.join(GOLDEN, "hf_cases.json")) as f:
        return json.load(f)


def case_inputs(case):
    return synth.attention_inputs(case["seed"], case["heads"], case["N"], case["d"],
                                  case["sigma_qk"], case["sigma_v"])


def module_weights(case):
    """Seeded to_q/to_k/to_v/to_out weights of the whole-module golden cases."""
    C = case["heads"] * case["d"]
    s = 1.0 / np.sqrt(C)
    w = case["seed"] * 100
    return dict(to_q=synth.normal(w + 1, (C, C), s), to_k=synth.normal(w + 2, (C, C), s),
                to_v=synth.normal(w + 3, (C, C), s), to_out_w=synth.normal(w + 4, (C, C), s),
                to_out_b=synth.normal(w + 5, (C,), 0.1))


def rel_err(a, b):
    """max |a-b| / max |b| : the 'relative' of the parity bars (1e-3 fp32, 2e-2 bf16)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
