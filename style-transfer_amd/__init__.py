"""MI355X-native mask-based multi-style injection path (gfx950 HIP kernels behind a C ABI).

Submodules:
  lib             ctypes binding of libsid.so (the C ABI declared in include/sid.h)
  ops             tensor-level entry points (torch tensors in, kernels enqueued on the current stream)
  attention       drop-in `CrossAttention` with the injected-features forward signature
  high_frequency  `high_freq_filter`, schedule and ResBlock patch
  mask            content-mask loading, thresholding and per-resolution cache
  features        `feat_merge_2sty` and the feature-list layout
  synth           deterministic synthetic inputs
"""
__version__ = "0.1.0"
