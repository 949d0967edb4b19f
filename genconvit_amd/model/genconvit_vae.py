"""GenConViTVAE — network B of GenConViT on the MI355X HIP path.

Mirror of the reference's ``model/genconvit_vae.py:91-116``: same class name, constructor arguments,
state_dict keys and ``forward(x) -> (logits (B,2), resized reconstruction (B,3,224,224))`` tuple.
The encoder's only RNG draw (``torch.randn_like``, genconvit_vae.py:46) is an explicit ``eps``
argument here (drawn with ``torch.randn`` when omitted) so results are reproducible and checkable.
"""
from __future__ import annotations

import torch

from .. import spec, synth
from ._base import HipModule, build_param_tree


class GenConViTVAE(HipModule):
    def __init__(self, config, pretrained=True, init="synthetic", seed=synth.DEFAULT_SEED):
        super().__init__()
        self.config = config
        self.latent_dims = config["model"]["latent_dims"]
        if self.latent_dims != spec.LATENT_DIMS:
            raise ValueError("latent_dims must be 12544 = 256*7*7 (decoder Unflatten, genconvit_vae.py:81)")
        if config["model"]["backbone"] != "convnext_tiny":
            raise ValueError("only the convnext_tiny backbone is built (reference default, model/config.yaml:2)")
        build_param_tree(self, spec.vae_spec(include_unused=True), init, seed, "vae/")
        self.num_feature = spec.NUM_BACKBONE_CLASSES * 2
        self.kl = None          # Encoder.kl side effect (genconvit_vae.py:58), filled when want_kl
        self.mse = None
        self._generator = None
        self.keep_kl_weights = True   # False: encoder.var is not packed (inference never reads it, genconvit.py:70,73)

    def set_generator(self, generator):
        """torch.Generator used for eps when none is passed (device generator of the model device)."""
        self._generator = generator

    def _load_into(self, handle):
        handle.load_vae(self.state_dict(), with_var=self.keep_kl_weights)

    @torch.no_grad()
    def forward(self, x, eps=None, want_recon=True, want_mse=False, want_kl=False):
        x = self._prep_input(x)
        B = x.shape[0]
        if B == 0:                                    # empty batch -> empty outputs (no RNG draw, no launch)
            self.kl, self.mse = None, None
            recon = torch.empty((0, 3, 224, 224), dtype=x.dtype, device=x.device) if want_recon else None
            return torch.empty((0, 2), dtype=torch.float32, device=x.device), recon
        if eps is None:
            eps = torch.randn((B, self.latent_dims), dtype=torch.float32, device=x.device, generator=self._generator)
        else:
            eps = eps.to(device=x.device, dtype=torch.float32)
        if B > 512:                                   # beyond one handle's workspace: consecutive chunks (kl: last chunk's)
            parts = [self._get_handle(hi - lo).vae_forward(x[lo:hi], eps[lo:hi], want_recon, want_mse, want_kl)
                     for lo, hi in self._chunks(B)]
            cat = lambda i: torch.cat([p[i] for p in parts]) if parts[0][i] is not None else None
            logits, recon, mse, kl = cat(0), cat(1), cat(2), parts[-1][3]
        else:
            logits, recon, mse, kl = self._get_handle(B).vae_forward(x, eps, want_recon, want_mse, want_kl)
        self.kl = kl[0] if kl is not None else None
        self.mse = mse
        return logits, recon

    def backbone_forward(self, images):
        images = self._prep_input(images)
        return self._get_handle(images.shape[0]).convnext_forward(1, images)
