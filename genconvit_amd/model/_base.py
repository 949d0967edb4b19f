"""Shared plumbing of the HIP-backed GenConViT modules: reference-named parameters held as ordinary
``nn.Parameter``s (so ``state_dict`` / ``load_state_dict`` / ``.to()`` / ``.half()`` / ``.parameters()``
behave like the reference's modules), plus a lazily (re)built :class:`genconvit_amd._lib.Handle`
that owns the packed device copy the kernels read.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib, synth


def build_param_tree(root: nn.Module, entries, init: str, seed: int, tag: str):
    """Register ``entries`` (name, shape, kind) under ``root`` with dotted names -> nested modules,
    so ``root.state_dict()`` has exactly the reference's keys."""
    for name, shape, kind in entries:
        parts = name.split(".")
        node = root
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, nn.Module())
            node = node._modules[p]
        if init == "synthetic":
            t = synth.make_param(tag + name, shape, kind, seed)
        elif init == "empty":
            t = torch.zeros(shape) if kind in ("bn_mean",) else torch.ones(shape) if kind == "bn_var" else torch.empty(shape)
        else:
            raise ValueError(f"init must be 'synthetic' or 'empty', got {init!r}")
        if kind in ("bn_mean", "bn_var"):
            node.register_buffer(parts[-1], t)
        else:
            node.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))


# state_dict keys of the published checkpoints that never take part in forward (SURVEY.md §0.4,
# Appendix A.3): the Swin "embedder" (registered twice) and HybridEmbed.proj, BN counters.
OFF_PATH_MARKERS = ("embedder.", "patch_embed.", "num_batches_tracked")


# Parameters at least this large (elements) never move to the device as nn.Parameters: the VAE's two 25088 x 12544
# Linear weights are 1.26 GB each in fp32 and the kernels read the handle's packed copy, not the module's tensors.
# They stay host-resident (``.to(device)`` / ``.cuda()`` change only their dtype), are packed into the handle
# tensor by tensor (descriptor ``on_device = 0``), and ``state_dict()`` keeps returning them under the reference's keys.
HOST_RESIDENT_NUMEL = 1 << 24
MAX_HANDLE_BATCH = 512          # gcv_create's upper bound for max_batch


class HipModule(nn.Module):
    """Base class: parameter tree + handle lifecycle."""

    _default_max_batch = 32

    def __init__(self):
        super().__init__()
        self._handle = None
        self._handle_key = None
        self._dirty = True
        self._max_batch = self._default_max_batch
        self._sig_params = None
        self._loaded_sig = None

    # nn.Module._apply is what .to()/.half()/.float()/.cuda() go through
    def _apply(self, fn, *a, **k):
        self._dirty = True
        big = [(m, n, p) for m in self.modules() for n, p in m._parameters.items()
               if p is not None and p.numel() >= HOST_RESIDENT_NUMEL]
        if not big:
            return super()._apply(fn, *a, **k)
        # what does fn do to a floating tensor?  (dtype is kept for the host-resident ones, the device is not)
        probe = fn(torch.empty(1, dtype=big[0][2].dtype, device=big[0][2].device))
        for m, n, p in big:
            del m._parameters[n]
        try:
            super()._apply(fn, *a, **k)
        finally:
            for m, n, p in big:
                if p.dtype != probe.dtype and probe.is_floating_point():
                    p = nn.Parameter(p.detach().to("cpu", probe.dtype), requires_grad=False)
                m._parameters[n] = p
        return self

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        own = set(self.state_dict().keys())
        filtered = {k: v for k, v in state_dict.items()
                    if k in own or not any(m in k for m in OFF_PATH_MARKERS)}
        self._dirty = True
        return super().load_state_dict(filtered, strict=strict, **kw)

    def _param_device_dtype(self):
        p = next(q for q in self.parameters() if q.numel() < HOST_RESIDENT_NUMEL)
        return p.device, p.dtype

    def _load_into(self, handle):   # subclasses push their weights
        raise NotImplementedError

    def invalidate(self):
        """Force the packed device copy to be rebuilt at the next forward: needed after edits the version check cannot
        see — writes through ``p.data`` and re-assigned Parameters (``module.fc.weight = nn.Parameter(...)``)."""
        self._dirty = True

    def _weights_signature(self):
        # in-place edits (p.mul_(), optimizer-style writes) bump ``_version``.  The parameter list is cached between packs,
        # so a RE-ASSIGNED Parameter (module.fc.weight = nn.Parameter(...)) is not seen here: invalidate() is the contract
        # for that (docstring above)
        ps = self._sig_params
        if ps is None:
            ps = self._sig_params = list(self.parameters()) + list(self.buffers())
        return (sum(p._version for p in ps), sum(p.data_ptr() for p in ps) & 0xFFFFFFFFFFFF)

    def _get_handle(self, batch: int):
        batch = min(batch, MAX_HANDLE_BATCH)
        device, dtype = self._param_device_dtype()
        if device.type != "cuda":
            raise _lib.GenConViTHipError(
                "model parameters are on CPU: the GenConViT HIP path has no CPU fallback — call .to('cuda') "
                "on a machine with an MI355X")
        if batch > self._max_batch:
            self._max_batch = 1 << (batch - 1).bit_length()
        key = (device.index if device.index is not None else torch.cuda.current_device(), dtype, self._max_batch)
        if self._handle is None or self._handle_key != key:
            if self._handle is not None:
                self._handle.close()
            self._handle = _lib.Handle(key[0], dtype, self._max_batch)
            self._handle_key = key
            self._dirty = True
        if not self._dirty and self._weights_signature() != self._loaded_sig:
            self._dirty = True                         # a parameter was edited in place or replaced since the last pack
        if self._dirty:
            self._sig_params = None
            self._load_into(self._handle)
            self._dirty = False
            self._loaded_sig = self._weights_signature()
        return self._handle

    def _chunks(self, n: int):
        """Batches beyond one handle's workspace (512 frames) run as consecutive chunks: the reference accepts any B."""
        return [(i, min(i + MAX_HANDLE_BATCH, n)) for i in range(0, n, MAX_HANDLE_BATCH)]

    def reserve(self, max_batch: int):
        """Size the workspace for batches up to ``max_batch`` ahead of the first call."""
        self._max_batch = min(max(int(max_batch), 1), MAX_HANDLE_BATCH)
        return self

    def _prep_input(self, x):
        device, dtype = self._param_device_dtype()
        if x.device != device:
            x = x.to(device)
        if x.dtype != dtype:      # reference bug (prediction.py:248-249: df.half() result dropped) fixed here
            x = x.to(dtype)
        return x.contiguous()
