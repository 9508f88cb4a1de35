"""Content mask handling for the injected attention.

The reference reloads `<content>_mask.npy` from disk, thresholds it and resizes it inside every
attention call (ldm/modules/attention.py:270-284; 12 calls per UNet evaluation).  The result
depends only on the file and the token grid, so it is computed once per (file, grid, device)
with the sid_mask_downsample kernel and kept on the GPU.
"""
import os

import numpy as np
import torch

from . import ops


def mask_path_for(cnt_name):
    """`<content path without extension>_mask.npy` (ldm/modules/attention.py:410-411).
    Raises TypeError for cnt_name=None, like os.path.splitext does in the reference."""
    base_name, _ = os.path.splitext(cnt_name)
    return base_name + "_mask.npy"


class MaskCache:
    def __init__(self):
        self._raw = {}      # (path, mtime, device) -> (Hin, Win) float32 tensor
        self._down = {}     # (path, mtime, device, h, w) -> (h*w,) signed float32 tensor

    def clear(self):
        self._raw.clear()
        self._down.clear()

    def signed_mask(self, mask_path, h, w, device):
        """(h*w,) float32 on `device`: +1 character ... -1 background (sign channel ch=+1)."""
        mtime = os.path.getmtime(mask_path)
        key = (mask_path, mtime, str(device), h, w)
        m = self._down.get(key)
        if m is None:
            rkey = key[:3]
            raw = self._raw.get(rkey)
            if raw is None:
                arr = np.load(mask_path)
                if arr.ndim != 2:
                    raise ValueError("%s: expected a 2-D mask, got shape %s" % (mask_path, arr.shape))
                raw = torch.tensor(arr, dtype=torch.float32).to(device).contiguous()
                self._raw[rkey] = raw
            m = ops.mask_downsample(raw, h, w, 1.0)
            self._down[key] = m
        return m


default_cache = MaskCache()
