"""One UNet evaluation as a HIP graph.

The host network around the path is some 700 small PyTorch kernels per evaluation, and launched one by
one it is bound by the host's launch path, not by the GPU (an SD-v1-size evaluation takes 17 ms eager
against a few ms of GPU work).  `GraphedUNet` captures one evaluation -- network, injected attention,
routing -- as a HIP graph over static input slots and replays it for every DDIM step: per step the
latent, the timestep, the context and that step's merged feature dict are copied into the slots
(device-to-device, about 115 MB per step for an SD-v1 triple: tens of microseconds) and the graph is
replayed.  Everything the evaluation decides on the host (which attention branch a layer takes, the
mask file of the content image, gamma and T) is frozen at capture, so a graph serves one content image
and one (gamma, T); `for_step` re-captures when they change.  The reference has no counterpart: it runs
the network eagerly (ldm/models/diffusion/ddim.py:286-295).
"""
import torch

from . import ops


def _clone_slot(v):
    if torch.is_tensor(v):
        return v.clone()
    if isinstance(v, ops.PackedKV):
        slot = ops.PackedKV(v.image.clone(), v.B, v.H, v.N, v.d, v.compute)
        if v.partial is not None:           # a content pair with its cached content block (its q slot is tied in _capture)
            p = v.partial
            slot.partial = ops.ContentPartial(p.data.clone(), p.B, p.H, p.N, p.d, p.compute, p.scale, None)
        return slot
    if isinstance(v, dict):
        return dict(v)
    return v


def _copy_into(slot, v):
    if torch.is_tensor(slot):
        slot.copy_(v, non_blocking=True)
    elif isinstance(slot, ops.PackedKV):
        slot.image.copy_(v.image, non_blocking=True)
        if slot.partial is not None:
            slot.partial.data.copy_(v.partial.data, non_blocking=True)


class GraphedUNet:
    """Callable with the UNet's signature `(x, timesteps, context, injected_features=None)`."""

    def __init__(self, unet):
        self.unet = unet
        self._key = None
        self._graph = None
        self.content_name = None       # set by a caller that runs several of these over one network: this lane's content image

    def _signature(self, x, context, feats):
        cnt = tuple(getattr(m, "cnt_name", None) for m in self.unet.attention_modules()[:1])
        cfg = None if feats is None else (feats["config"].get("T"), feats["config"].get("gamma"))
        # (a key is marked when its pair carries a cached content block: the evaluation then takes another entry point)
        keys = None if feats is None else tuple(sorted(k + ("+" if getattr(v, "partial", None) is not None else "")
                                                       for k, v in feats.items() if k != "config"))
        return (tuple(x.shape), x.dtype, None if context is None else tuple(context.shape), cnt, cfg, keys)

    def _capture(self, x, timesteps, context, feats):
        with ops.capture_lock:           # the feature loader's thread stays off the GPU meanwhile
            self._capture_locked(x, timesteps, context, feats)

    def _capture_locked(self, x, timesteps, context, feats):
        self.x, self.t = x.clone(), timesteps.clone()
        self.ctx = None if context is None else context.clone()
        self.slots = None
        if feats is not None:
            self.slots = {}
            seen = {}
            for k, v in feats.items():          # a PackedKV sits under its k and v keys: one slot for both
                if id(v) not in seen:
                    seen[id(v)] = _clone_slot(v)
                self.slots[k] = seen[id(v)]
            # a cached content block belongs to the content query it was made from: in the slots that is the
            # query slot of the same layer (the attention call checks the address)
            for k, slot in self.slots.items():
                if isinstance(slot, ops.PackedKV) and slot.partial is not None and k.endswith("_k_cnt"):
                    slot.partial.q_ptr = self.slots[k[:-len("_k_cnt")] + "_q"].data_ptr()
        self.stream = torch.cuda.Stream(device=x.device)
        self.stream.wait_stream(torch.cuda.current_stream(x.device))
        for t in (x, timesteps, context):           # the clones above were made on the caller's stream
            if t is not None:
                t.record_stream(self.stream)
        with torch.cuda.stream(self.stream), torch.no_grad():
            for _ in range(2):                  # warm-up: workspaces, mask cache, library kernel selection
                self.unet(self.x, self.t, self.ctx, injected_features=self.slots)
        torch.cuda.synchronize(x.device)
        self._graph = torch.cuda.CUDAGraph()
        # thread_local: the feature loader's worker thread keeps reading, allocating and packing the next
        # triple while this thread captures (in the default global mode its runtime calls invalidate the capture)
        with torch.cuda.graph(self._graph, stream=self.stream, capture_error_mode="thread_local"), torch.no_grad():
            self.out = self.unet(self.x, self.t, self.ctx, injected_features=self.slots)

    def __call__(self, x, timesteps=None, context=None, injected_features=None):
        if self.content_name is not None:
            mods = self.unet.attention_modules()
            if mods and mods[0].cnt_name != self.content_name:
                self.unet.set_content_name(self.content_name)
        key = self._signature(x, context, injected_features)
        if key != self._key:
            self._capture(x, timesteps, context, injected_features)
            self._key = key
        self.x.copy_(x, non_blocking=True)
        self.t.copy_(timesteps, non_blocking=True)
        if context is not None:
            self.ctx.copy_(context, non_blocking=True)
        if injected_features is not None:
            done = set()
            for k, v in injected_features.items():
                if k != "config" and id(v) not in done:
                    done.add(id(v))
                    _copy_into(self.slots[k], v)
        self._graph.replay()
        return self.out.clone()

    # the sampler and the CLI reach through to the network for these
    def __getattr__(self, name):
        return getattr(self.unet, name)
