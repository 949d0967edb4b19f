"""HybridEmbed — mirror of the reference's ``model/model_embedder.py:4-44``.

In GenConViT the wrapped backbone is Swin-T; the reference attaches the module as
``backbone.patch_embed`` but timm's ConvNeXt forward never calls it, so it only
matters for its constructor probe (one Swin forward on zeros to read the output dims) and for
state_dict compatibility.  Here the probe runs on the HIP Swin-T path when the wrapped module
provides it; ``forward`` keeps the reference semantics (backbone -> 1x1 proj -> flatten -> transpose).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class HybridEmbed(nn.Module):
    def __init__(self, backbone, img_size=224, patch_size=1, feature_size=None, in_chans=3, embed_dim=768):
        super().__init__()
        if not isinstance(backbone, nn.Module):
            raise TypeError("backbone must be an nn.Module")
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.backbone = backbone
        if feature_size is None:
            with torch.no_grad():
                was_training = backbone.training
                backbone.eval()
                p = next(backbone.parameters(), None)
                probe = torch.zeros(1, in_chans, img_size, img_size,
                                    device=p.device if p is not None else None,
                                    dtype=p.dtype if p is not None else None)
                o = backbone(probe)
                if isinstance(o, (list, tuple)):
                    o = o[-1]
                feature_size = tuple(o.shape[-2:])
                feature_dim = o.shape[1]
                backbone.train(was_training)
        else:
            feature_size = (feature_size, feature_size)
            feature_dim = (backbone.feature_info.channels()[-1] if hasattr(backbone, "feature_info")
                           else backbone.num_features)
        if feature_size[0] % patch_size or feature_size[1] % patch_size:
            raise ValueError("feature map not divisible by patch size")
        self.grid_size = (feature_size[0] // patch_size, feature_size[1] // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(feature_dim, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)

    def forward(self, x):
        x = self.backbone(x)
        if isinstance(x, (list, tuple)):
            x = x[-1]
        return self.proj(x).flatten(2).transpose(1, 2)   # raises for 2-D logits, exactly like the reference
