"""Parameter inventory of the GenConViT `ed` / `vae` networks on the hot path.

Names and shapes follow the reference state_dict layout so that the published
``weight/{ed,vae}.pth`` checkpoints map 1:1:

* ED   : /root/reference/model/genconvit_ed.py:8-75
* VAE  : /root/reference/model/genconvit_vae.py:10-105
* ConvNeXt-T (third-party timm==0.6.5 ``convnext_tiny``; call sites
  genconvit_ed.py:68,82-83 and genconvit_vae.py:97,111-112; spec SURVEY Appendix A.1)

Each entry is ``(name, shape, kind)``; ``kind`` selects the synthetic
distribution in :mod:`genconvit_amd.synth` and is ignored when real
checkpoints are loaded.
"""
from __future__ import annotations

CONVNEXT_DIMS = (96, 192, 384, 768)
CONVNEXT_DEPTHS = (3, 3, 9, 3)
NUM_BACKBONE_CLASSES = 1000
LATENT_DIMS = 12544          # model/config.yaml:4
VAE_FLAT = 128 * 14 * 14     # genconvit_vae.py:36


def convnext_tiny_spec(prefix: str):
    """timm 0.6.5 ``convnext_tiny`` parameters under ``prefix`` (e.g. ``backbone.``)."""
    out = []
    p = prefix
    out.append((p + "stem.0.weight", (96, 3, 4, 4), "conv"))
    out.append((p + "stem.0.bias", (96,), "bias"))
    out.append((p + "stem.1.weight", (96,), "ln_w"))
    out.append((p + "stem.1.bias", (96,), "bias"))
    for i, (dim, depth) in enumerate(zip(CONVNEXT_DIMS, CONVNEXT_DEPTHS)):
        if i > 0:
            din = CONVNEXT_DIMS[i - 1]
            out.append((p + f"stages.{i}.downsample.0.weight", (din,), "ln_w"))
            out.append((p + f"stages.{i}.downsample.0.bias", (din,), "bias"))
            out.append((p + f"stages.{i}.downsample.1.weight", (dim, din, 2, 2), "conv"))
            out.append((p + f"stages.{i}.downsample.1.bias", (dim,), "bias"))
        for j in range(depth):
            b = p + f"stages.{i}.blocks.{j}."
            out.append((b + "conv_dw.weight", (dim, 1, 7, 7), "conv"))
            out.append((b + "conv_dw.bias", (dim,), "bias"))
            out.append((b + "norm.weight", (dim,), "ln_w"))
            out.append((b + "norm.bias", (dim,), "bias"))
            out.append((b + "mlp.fc1.weight", (4 * dim, dim), "linear"))
            out.append((b + "mlp.fc1.bias", (4 * dim,), "bias"))
            out.append((b + "mlp.fc2.weight", (dim, 4 * dim), "linear"))
            out.append((b + "mlp.fc2.bias", (dim,), "bias"))
            out.append((b + "gamma", (dim,), "gamma"))
    out.append((p + "head.norm.weight", (768,), "ln_w"))
    out.append((p + "head.norm.bias", (768,), "bias"))
    out.append((p + "head.fc.weight", (NUM_BACKBONE_CLASSES, 768), "linear"))
    out.append((p + "head.fc.bias", (NUM_BACKBONE_CLASSES,), "bias"))
    return out


def ed_spec():
    """GenConViTED parameters that take part in forward (genconvit_ed.py:64-88)."""
    out = []
    chans = [3, 16, 32, 64, 128, 256]
    for li, idx in enumerate((0, 3, 6, 9, 12)):              # genconvit_ed.py:14-32
        out.append((f"encoder.features.{idx}.weight", (chans[li + 1], chans[li], 3, 3), "conv"))
        out.append((f"encoder.features.{idx}.bias", (chans[li + 1],), "bias"))
    dch = [256, 128, 64, 32, 16, 3]
    for li, idx in enumerate((0, 2, 4, 6, 8)):               # genconvit_ed.py:44-57
        out.append((f"decoder.features.{idx}.weight", (dch[li], dch[li + 1], 2, 2), "convT"))
        out.append((f"decoder.features.{idx}.bias", (dch[li + 1],), "bias"))
    out += convnext_tiny_spec("backbone.")
    out.append(("fc.weight", (500, 2000), "linear"))          # genconvit_ed.py:72-74
    out.append(("fc.bias", (500,), "bias"))
    out.append(("fc2.weight", (2, 500), "linear"))
    out.append(("fc2.bias", (2,), "bias"))
    return out


def vae_spec(include_unused: bool = True):
    """GenConViTVAE parameters (genconvit_vae.py:91-105).

    ``include_unused`` adds ``encoder.fc1/fc2`` and ``fc3`` which exist in the
    checkpoints but never run in forward.
    """
    out = []
    chans = [3, 16, 32, 64, 128]
    for li, idx in enumerate((0, 3, 6, 9)):                  # genconvit_vae.py:15-31
        c = chans[li + 1]
        out.append((f"encoder.features.{idx}.weight", (c, chans[li], 3, 3), "conv"))
        out.append((f"encoder.features.{idx}.bias", (c,), "bias"))
        out.append((f"encoder.features.{idx + 1}.weight", (c,), "ln_w"))
        out.append((f"encoder.features.{idx + 1}.bias", (c,), "bias"))
        out.append((f"encoder.features.{idx + 1}.running_mean", (c,), "bn_mean"))
        out.append((f"encoder.features.{idx + 1}.running_var", (c,), "bn_var"))
    out.append(("encoder.mu.weight", (LATENT_DIMS, VAE_FLAT), "linear"))   # :36
    out.append(("encoder.mu.bias", (LATENT_DIMS,), "bias"))
    out.append(("encoder.var.weight", (LATENT_DIMS, VAE_FLAT), "linear"))  # :37
    out.append(("encoder.var.bias", (LATENT_DIMS,), "bias"))
    if include_unused:
        out.append(("encoder.fc1.weight", (256, VAE_FLAT), "linear"))     # :34
        out.append(("encoder.fc1.bias", (256,), "bias"))
        out.append(("encoder.fc2.weight", (128, 256), "linear"))          # :35
        out.append(("encoder.fc2.bias", (128,), "bias"))
    dch = [256, 64, 32, 16, 3]
    for li, idx in enumerate((0, 2, 4, 6)):                  # genconvit_vae.py:68-78
        out.append((f"decoder.features.{idx}.weight", (dch[li], dch[li + 1], 2, 2), "convT"))
        out.append((f"decoder.features.{idx}.bias", (dch[li + 1],), "bias"))
    out += convnext_tiny_spec("convnext_backbone.")
    out.append(("fc.weight", (500, 2000), "linear"))          # :101
    out.append(("fc.bias", (500,), "bias"))
    out.append(("fc2.weight", (2, 500), "linear"))            # :103
    out.append(("fc2.bias", (2,), "bias"))
    if include_unused:
        out.append(("fc3.weight", (500, 1000), "linear"))     # :102
        out.append(("fc3.bias", (500,), "bias"))
    return out


# ---- Swin-T (timm 0.6.5 swin_tiny_patch4_window7_224) ----
SWIN_DIMS = (96, 192, 384, 768)
SWIN_DEPTHS = (2, 2, 6, 2)
SWIN_HEADS = (3, 6, 12, 24)


def swin_tiny_spec(prefix: str):
    """Learnable parameters of Swin-T (buffers such as relative_position_index /
    attn_mask are derived, not stored here)."""
    out = []
    p = prefix
    out.append((p + "patch_embed.proj.weight", (96, 3, 4, 4), "conv"))
    out.append((p + "patch_embed.proj.bias", (96,), "bias"))
    out.append((p + "patch_embed.norm.weight", (96,), "ln_w"))
    out.append((p + "patch_embed.norm.bias", (96,), "bias"))
    for i, (dim, depth, nh) in enumerate(zip(SWIN_DIMS, SWIN_DEPTHS, SWIN_HEADS)):
        for j in range(depth):
            b = p + f"layers.{i}.blocks.{j}."
            out.append((b + "norm1.weight", (dim,), "ln_w"))
            out.append((b + "norm1.bias", (dim,), "bias"))
            out.append((b + "attn.relative_position_bias_table", (169, nh), "relpos"))
            out.append((b + "attn.qkv.weight", (3 * dim, dim), "linear"))
            out.append((b + "attn.qkv.bias", (3 * dim,), "bias"))
            out.append((b + "attn.proj.weight", (dim, dim), "linear"))
            out.append((b + "attn.proj.bias", (dim,), "bias"))
            out.append((b + "norm2.weight", (dim,), "ln_w"))
            out.append((b + "norm2.bias", (dim,), "bias"))
            out.append((b + "mlp.fc1.weight", (4 * dim, dim), "linear"))
            out.append((b + "mlp.fc1.bias", (4 * dim,), "bias"))
            out.append((b + "mlp.fc2.weight", (dim, 4 * dim), "linear"))
            out.append((b + "mlp.fc2.bias", (dim,), "bias"))
        if i < 3:
            d = p + f"layers.{i}.downsample."
            out.append((d + "norm.weight", (4 * dim,), "ln_w"))
            out.append((d + "norm.bias", (4 * dim,), "bias"))
            out.append((d + "reduction.weight", (2 * dim, 4 * dim), "linear"))
    out.append((p + "norm.weight", (768,), "ln_w"))
    out.append((p + "norm.bias", (768,), "bias"))
    out.append((p + "head.weight", (1000, 768), "linear"))
    out.append((p + "head.bias", (1000,), "bias"))
    return out
