"""GenConViT ensemble wrapper — mirror of the reference's ``model/genconvit.py:7-75``.

Same constructor ``GenConViT(config, ed, vae, net, fp16)``, same weight-file convention
(``weight/{name}.pth`` relative to the cwd, raw state_dict or ``{'state_dict': ...}``), same error
text when a file is missing, same forward: ``'ed'`` -> (B,2), ``'vae'`` -> (B,2), anything else ->
``cat((ed, vae), dim=0)`` -> (2B,2).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .. import _lib
from .genconvit_ed import GenConViTED
from .genconvit_vae import GenConViTVAE


def _read_checkpoint(name):
    path = name if os.path.isabs(name) or name.endswith(".pth") else os.path.join("weight", f"{name}.pth")
    try:       # zip-format checkpoints are memory-mapped: the 2.6 GB VAE file is paged in tensor by tensor
        ckpt = torch.load(path, map_location=torch.device("cpu"), mmap=True, weights_only=True)
    except (RuntimeError, ValueError, TypeError):          # legacy (non-zip) files cannot be mapped
        ckpt = torch.load(path, map_location=torch.device("cpu"))
    return ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt


class GenConViT(nn.Module):
    concurrent = True        # run ED and VAE on two streams when net is the ensemble (class-level switch)
    # The HIP path always computes fp32 logits.  The reference returns them in the model's dtype (model/genconvit.py:59-61:
    # after ``.half()`` the networks' last Linear emits fp16, and pred_vid's sigmoid / mean then run in fp16).  Setting
    # this to True rounds the returned logits to the parameters' dtype, so that ``out.dtype`` and ``y_val`` follow the
    # reference's --fp16 pipeline; the default keeps the more accurate fp32 tensor.
    reference_logits_dtype = False

    def __init__(self, config, ed, vae, net, fp16):
        super().__init__()
        self.net = net
        self.fp16 = fp16
        want_ed = net != "vae"
        want_vae = net != "ed"
        try:
            if want_ed:
                self.model_ed = GenConViTED(config, init="empty")
                self.model_ed.load_state_dict(_read_checkpoint(ed))
                self.model_ed.eval()
            if want_vae:
                self.model_vae = GenConViTVAE(config, init="empty")
                self.model_vae.load_state_dict(_read_checkpoint(vae))
                self.model_vae.eval()
                self.model_vae.keep_kl_weights = False     # forward() below drops kl / recon like the reference (:70,73)
        except FileNotFoundError:
            if net == "ed":
                raise Exception(f"Error: weight/{ed}.pth file not found.")
            if net == "vae":
                raise Exception(f"Error: weight/{vae}.pth file not found.")
            raise Exception("Error: Model weights file not found.")
        if self.fp16:
            self.half()

    @classmethod
    def from_modules(cls, model_ed=None, model_vae=None, net="genconvit", fp16=False):
        """Build the wrapper around already constructed networks (synthetic weights, tests, bench)."""
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self.net, self.fp16 = net, fp16
        if net != "vae":
            self.model_ed = model_ed
        if net != "ed":
            self.model_vae = model_vae
        if fp16:
            self.half()
        return self

    @torch.no_grad()
    def forward(self, x, eps=None):
        out = self._forward_fp32(x, eps)
        if self.reference_logits_dtype:
            m = self.model_ed if self.net != "vae" else self.model_vae
            out = out.to(m._param_device_dtype()[1])
        return out

    def _forward_fp32(self, x, eps=None):
        if self.net == "ed":
            return self.model_ed(x)
        if self.net == "vae":
            return self.model_vae(x, eps=eps, want_recon=False)[0]
        if self.concurrent and next(self.model_ed.parameters()).is_cuda:
            # ED and VAE are independent until the concat: gcv_genconvit_forward runs them on two streams inside the
            # library (each network has its own handle / workspace) and joins them back into the current stream, so
            # the GPU overlaps one network's small-grid kernels (14x14 and 7x7 stages, heads) with the other's
            x = self.model_ed._prep_input(x)
            B = x.shape[0]
            if B == 0:
                return torch.empty((0, 2), dtype=torch.float32, device=x.device)
            if eps is None:
                eps = torch.randn((B, self.model_vae.latent_dims), dtype=torch.float32, device=x.device,
                                  generator=self.model_vae._generator)
            else:
                eps = eps.to(device=x.device, dtype=torch.float32)
            if B <= 512:
                return _lib.genconvit_forward(self.model_ed._get_handle(B), self.model_vae._get_handle(B), x, eps)
            parts = [_lib.genconvit_forward(self.model_ed._get_handle(hi - lo), self.model_vae._get_handle(hi - lo),
                                            x[lo:hi], eps[lo:hi]) for lo, hi in self.model_ed._chunks(B)]
            return torch.cat([p[:p.shape[0] // 2] for p in parts] + [p[p.shape[0] // 2:] for p in parts])
        x1 = self.model_ed(x)
        x2 = self.model_vae(x, eps=eps, want_recon=False)[0]
        return torch.cat((x1, x2), dim=0)
