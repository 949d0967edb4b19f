"""Swin-T "embedder" on the HIP path (row A6 of SURVEY.md §8a).

The reference builds ``timm.create_model('swin_tiny_patch4_window7_224')`` (model/genconvit_ed.py:69,
model/genconvit_vae.py:96), wraps it in ``HybridEmbed`` and never executes it in forward;
it runs once in ``HybridEmbed.__init__`` to probe output dims (model/model_embedder.py:22).  This module
holds the Swin-T parameters under timm's key names and evaluates ``gcv_swin_forward``.
"""
from __future__ import annotations

import torch

from .. import spec, synth
from ._base import HipModule, build_param_tree


class SwinTinyEmbedder(HipModule):
    num_features = 768

    def __init__(self, init="synthetic", seed=synth.DEFAULT_SEED):
        super().__init__()
        build_param_tree(self, spec.swin_tiny_spec(""), init, seed, "swin/")

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        # timm checkpoints also carry derived buffers (relative_position_index, attn_mask): recomputed in-kernel
        own = set(self.state_dict().keys())
        filtered = {k: v for k, v in state_dict.items()
                    if k in own or not ("relative_position_index" in k or "attn_mask" in k)}
        self._dirty = True
        return torch.nn.Module.load_state_dict(self, filtered, strict=strict, **kw)

    def _load_into(self, handle):
        handle.load_swin(self.state_dict(), "")

    @torch.no_grad()
    def forward(self, x):
        x = self._prep_input(x)
        return self._get_handle(x.shape[0]).swin_forward(x)
