"""GenConViTED — network A of GenConViT on the MI355X HIP path.

Mirror of the reference's ``model/genconvit_ed.py:64-88`` (same class name, constructor arguments,
state_dict keys and ``forward(images) -> (B,2)`` contract).  The arithmetic — AE encoder/decoder,
two ConvNeXt-T passes over one 2B-image token stream, GELU-MLP head — runs in
``libgenconvit_hip.so`` (``gcv_ed_forward``); nothing here computes on the CPU.
"""
from __future__ import annotations

import torch

from .. import spec, synth
from ._base import HipModule, build_param_tree


class GenConViTED(HipModule):
    def __init__(self, config, pretrained=True, init="synthetic", seed=synth.DEFAULT_SEED):
        """``pretrained`` is accepted for signature compatibility: the reference uses it to fetch
        ImageNet weights through timm (genconvit_ed.py:68-69), which needs a network.  Here the
        parameters start from the deterministic synthetic generator (``init='synthetic'``) or
        uninitialised (``init='empty'``, when a checkpoint is loaded right after)."""
        super().__init__()
        self.config = config
        if config["model"]["backbone"] != "convnext_tiny":
            raise ValueError("only the convnext_tiny backbone is built (reference default, model/config.yaml:2)")
        build_param_tree(self, spec.ed_spec(), init, seed, "ed/")
        self.num_features = spec.NUM_BACKBONE_CLASSES * 2          # genconvit_ed.py:72

    def _load_into(self, handle):
        handle.load_ed(self.state_dict())

    @torch.no_grad()
    def forward(self, images):
        images = self._prep_input(images)
        if images.shape[0] == 0:                      # an empty batch is an empty result, as with the reference's nn.Modules
            return torch.empty((0, 2), dtype=torch.float32, device=images.device)
        if images.shape[0] > 512:
            return torch.cat([self._get_handle(hi - lo).ed_forward(images[lo:hi]) for lo, hi in self._chunks(images.shape[0])])
        return self._get_handle(images.shape[0]).ed_forward(images)

    def backbone_forward(self, images):
        """ConvNeXt-T alone (timm ``convnext_tiny`` forward, call site genconvit_ed.py:82-83)."""
        images = self._prep_input(images)
        return self._get_handle(images.shape[0]).convnext_forward(0, images)
