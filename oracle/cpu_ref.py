"""CPU oracle: functional fp32 restatement of the GenConViT ``ed``/``vae`` forward.

TEST INFRASTRUCTURE ONLY.  Nothing under ``genconvit_amd/`` imports this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may, and only as the checker / reported CPU baseline.

Pinning status (see DESIGN.md "Oracle"):
  * ED Encoder/Decoder, VAE Encoder/Decoder (incl. the reparameterisation quirk),
    the ED/VAE/GenConViT forward glue (cat order, activations, head) and
    pred_func's vote are PINNED: ``tests/golden/make_golden.py`` runs the
    reference's own classes (imported from /root/reference in the build
    container) on the same synthetic weights and the outputs match this file
    bit-for-bit; the vectors are committed under ``tests/golden/``.
  * The ConvNeXt-T / Swin-T arithmetic lives in third-party ``timm==0.6.5``
    (requirements.txt:5), which is absent here and has no tests in the
    reference: **parity unpinned** by the reference for that part.  It is
    restated from the published architecture and
    cross-checked against the independent Hugging Face ``transformers``
    implementation of the same architectures (tests/test_oracle.py).

Every function cites the reference lines it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# Architecture constants, written out here on purpose: the oracle shares no table with the product it checks
# (genconvit_amd/spec.py holds the product's copy; tests/test_oracle.py asserts the two agree).
CONVNEXT_DEPTHS = (3, 3, 9, 3)          # timm convnext_tiny
CONVNEXT_DIMS = (96, 192, 384, 768)
SWIN_DEPTHS = (2, 2, 6, 2)              # timm swin_tiny_patch4_window7_224
SWIN_DIMS = (96, 192, 384, 768)
SWIN_HEADS = (3, 6, 12, 24)

# --------------------------------------------------------------------------- same-dtype restatement (16-bit storage)
# "for 16-bit report the delta vs the fp32 oracle and vs a same-dtype CPU restatement".  Inside
# ``with storage_dtype(torch.float16 | torch.bfloat16):`` the functions below round (a) the input frames, (b) every
# weight the HIP path keeps in the storage dtype (the GEMM / conv-as-GEMM weights; biases, LayerNorm affine, layer
# scale, depthwise taps (except those of the 56-pixel C = 96 maps in launches of 64 images and more, where they are an MFMA operand: csrc/dwconv_mfma.h, ``Launch`` below), the
# first 3->16 convs, the last 16->3 transposed convs and the 500->2 layer stay fp32, as in genconvit_amd/csrc/net_impl.h) and (c) every activation at the points where the HIP path stores it in HBM or
# feeds it to a 16-bit MFMA operand (not: the output of the last block of stage 1 and, in launches of 65536 tokens and more, of
# stage 0, whose LayerNorm2d runs in that block's epilogue: ``Launch`` below) — all arithmetic stays fp32, like the kernels' accumulators.  With no storage
# dtype set (the default) ``_q`` returns its argument untouched: the fp32 oracle is bit-for-bit what it was.
_STORE = None


class storage_dtype:
    def __init__(self, dtype):
        self.dtype = None if dtype in (None, torch.float32) else dtype

    def __enter__(self):
        global _STORE
        self.prev, _STORE = _STORE, self.dtype
        return self

    def __exit__(self, *exc):
        global _STORE
        _STORE = self.prev


def _q(t):
    return t if _STORE is None else t.to(_STORE).float()


LN_EPS_CONVNEXT = 1e-6   # timm ConvNeXt LayerNorm/LayerNorm2d eps
LN_EPS_SWIN = 1e-5       # timm Swin nn.LayerNorm default
BN_EPS = 1e-5            # nn.BatchNorm2d default, genconvit_vae.py:16
LEAKY = 0.01             # nn.LeakyReLU default, genconvit_vae.py:17


# --------------------------------------------------------------------------- ED AE
def ed_encoder(sd, x, taps=None):
    """model/genconvit_ed.py:13-36 — 5x[Conv2d 3x3 s1 p1 -> ReLU -> MaxPool 2x2]."""
    for li, idx in enumerate((0, 3, 6, 9, 12)):
        w = sd[f"encoder.features.{idx}.weight"]
        x = F.conv2d(x, w if li == 0 else _q(w), sd[f"encoder.features.{idx}.bias"], stride=1, padding=1)
        x = _q(F.max_pool2d(F.relu(x), kernel_size=2, stride=2))
        if taps is not None:
            taps[f"ed_enc{li}"] = x
    return x


def ed_decoder(sd, x, taps=None):
    """model/genconvit_ed.py:43-61 — 5x[ConvTranspose2d 2x2 s2 -> ReLU]."""
    for li, idx in enumerate((0, 2, 4, 6, 8)):
        w = sd[f"decoder.features.{idx}.weight"]
        x = _q(F.relu(F.conv_transpose2d(x, w if li == 4 else _q(w), sd[f"decoder.features.{idx}.bias"], stride=2)))
        if taps is not None:
            taps[f"ed_dec{li}"] = x
    return x


# --------------------------------------------------------------------------- ConvNeXt-T
def _ln2d(x, w, b, eps):
    """timm LayerNorm2d: LN over the channel dim of NCHW (biased var, eps in sqrt)."""
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, eps).permute(0, 3, 1, 2)


# Launch-geometry thresholds of the HIP path that decide WHERE it rounds in 16-bit storage.  They are the oracle's own copy
# (this file shares no table with the product); tests/test_host_cpu.py parses the product headers and asserts they agree.
FUSED_LNP_MIN_TOKENS = 65536     # csrc/fused_mlp.h fused_mlp_res_applies: C = 96 launches of at least this many tokens run the
                                 # LDS-resident MLP, the only C = 96 kernel with the LayerNorm-patchify epilogue
DW_MFMA_MIN_IMAGE_ROWS = 14 * 256   # csrc/dwconv_roll_impl.h long_bands: images x rows of a C = 96 / 56-pixel dw launch from
                                    # which the taps run on the matrix pipe (as a 16-bit MFMA operand)


class Launch:
    """What one ConvNeXt pass of the HIP path looks like from the launch side (csrc/net_impl.h run_convnext): the passes of
    one network that share weights are concatenated on the token axis, so the dispatch rules above see the totals, not the
    single pass the functional restatement evaluates.  ``stage0_tokens``: tokens of ALL segments of the launch at stage 0;
    ``dw_images``: images in the depthwise launch this pass's geometry belongs to (neighbouring segments of one geometry
    share a launch)."""

    def __init__(self, stage0_tokens, dw_images):
        self.stage0_tokens, self.dw_images = int(stage0_tokens), int(dw_images)


def convnext_block(sd, p, x, store=True, mfma_taps=False):
    """timm 0.6.5 ConvNeXtBlock.forward: dw7x7 -> LN(NHWC)
    -> fc1 -> exact GELU -> fc2 -> * gamma -> + shortcut.  ``store=False``: the block's output is not a storage point
    (the HIP path applies the next stage's LayerNorm2d in this block's epilogue, on the fp32 values).  ``mfma_taps``: the
    HIP path runs this block's taps on the matrix pipe (csrc/dwconv_mfma.h), i.e. as an MFMA operand in the storage dtype."""
    c = x.shape[1]
    w_dw = sd[p + "conv_dw.weight"]
    if mfma_taps:
        w_dw = _q(w_dw)
    y = F.conv2d(x, w_dw, sd[p + "conv_dw.bias"], padding=3, groups=c)
    y = y.permute(0, 2, 3, 1)
    y = _q(F.layer_norm(y, (c,), sd[p + "norm.weight"], sd[p + "norm.bias"], LN_EPS_CONVNEXT))
    y = F.linear(y, _q(sd[p + "mlp.fc1.weight"]), sd[p + "mlp.fc1.bias"])
    y = _q(F.gelu(y))
    if _STORE is not None and c == 384:
        # the C = 384 kernel pair folds the layer scale into the packed fc2 (csrc/mlp_pair.h pack_w2_frag_kernel): the MFMA
        # operand is gamma * W2 rounded once to the storage dtype, the bias term gamma * b2 stays fp32
        g = sd[p + "gamma"]
        y = F.linear(y, _q(g[:, None] * sd[p + "mlp.fc2.weight"]), g * sd[p + "mlp.fc2.bias"])
        y = y.permute(0, 3, 1, 2)
    else:
        y = F.linear(y, _q(sd[p + "mlp.fc2.weight"]), sd[p + "mlp.fc2.bias"])
        y = y.permute(0, 3, 1, 2)
        y = y * sd[p + "gamma"].reshape(1, -1, 1, 1)
    return _q(y + x) if store else y + x


def convnext_tiny(sd, prefix, x, taps=None, store_out=True, launch=None):
    """timm 0.6.5 ``convnext_tiny`` forward: stem -> 4 stages -> (norm_pre=Identity)
    -> head(global avg pool, LayerNorm2d, flatten, fc).  Called by the reference at
    model/genconvit_ed.py:82-83 and model/genconvit_vae.py:111-112.  ``launch`` (16-bit restatement only): the geometry of
    the HIP launch this pass is a segment of; default = the pass on its own (gcv_convnext_forward)."""
    p = prefix
    if launch is None:
        launch = Launch(x.shape[0] * (x.shape[2] // 4) * (x.shape[3] // 4), x.shape[0])
    x = F.conv2d(x, sd[p + "stem.0.weight"], sd[p + "stem.0.bias"], stride=4)
    x = _q(_ln2d(x, sd[p + "stem.1.weight"], sd[p + "stem.1.bias"], LN_EPS_CONVNEXT))
    if taps is not None:
        taps["stem"] = x
    for i, depth in enumerate(CONVNEXT_DEPTHS):
        if i > 0:
            x = _q(_ln2d(x, sd[p + f"stages.{i}.downsample.0.weight"],
                         sd[p + f"stages.{i}.downsample.0.bias"], LN_EPS_CONVNEXT))
            x = _q(F.conv2d(x, _q(sd[p + f"stages.{i}.downsample.1.weight"]),
                            sd[p + f"stages.{i}.downsample.1.bias"], stride=2))
        for j in range(depth):
            # the last block of stages 0 and 1 hands its fp32 output to the stage boundary's LayerNorm (epilogue fusion in
            # csrc/fused_mlp_res.h — only for launches of 65536 tokens and more — and csrc/xs_mlp.h)
            fused = (j == depth - 1 and (i == 1 or (i == 0 and launch.stage0_tokens >= FUSED_LNP_MIN_TOKENS))
                     and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)
            mfma = i == 0 and x.shape[-1] == 56 and launch.dw_images * x.shape[2] >= DW_MFMA_MIN_IMAGE_ROWS
            x = convnext_block(sd, p + f"stages.{i}.blocks.{j}.", x, store=not fused, mfma_taps=mfma)
        if taps is not None:
            taps[f"stage{i}"] = x
    x = x.mean((2, 3), keepdim=True)
    x = _q(_ln2d(x, sd[p + "head.norm.weight"], sd[p + "head.norm.bias"], LN_EPS_CONVNEXT))
    x = torch.flatten(x, 1)
    x = F.linear(x, _q(sd[p + "head.fc.weight"]), sd[p + "head.fc.bias"])
    return _q(x) if store_out else x       # (ED / VAE store it after the head's first activation)


# --------------------------------------------------------------------------- ED
def ed_forward(sd, images, taps=None):
    """GenConViTED.forward, model/genconvit_ed.py:77-88.  cat order is
    [backbone(recon), backbone(orig)]; ``self.relu`` is exact-erf nn.GELU (:75)."""
    images = _q(images)
    encimg = ed_encoder(sd, images, taps)
    decimg = ed_decoder(sd, encimg, taps)
    # both passes are one 2B-image launch of one geometry (net_impl.h ed_forward)
    la = Launch(2 * images.shape[0] * (images.shape[2] // 4) * (images.shape[3] // 4), 2 * images.shape[0])
    x1 = convnext_tiny(sd, "backbone.", decimg, store_out=False, launch=la)
    x2 = convnext_tiny(sd, "backbone.", images, taps, store_out=False, launch=la)
    x = torch.cat((x1, x2), dim=1)
    if taps is not None:
        taps["ed_feat"] = x
    x = _q(F.gelu(x))
    x = _q(F.gelu(F.linear(x, _q(sd["fc.weight"]), sd["fc.bias"])))
    return F.linear(x, sd["fc2.weight"], sd["fc2.bias"])


# --------------------------------------------------------------------------- VAE
def vae_encoder_features(sd, x, taps=None):
    """model/genconvit_vae.py:14-31,52-53 — 4x[Conv2d 3x3 s2 p1 -> BatchNorm2d(eval)
    -> LeakyReLU(0.01)] then flatten (C-major)."""
    for li, idx in enumerate((0, 3, 6, 9)):
        b = f"encoder.features.{idx + 1}."
        if _STORE is None:
            x = F.conv2d(x, sd[f"encoder.features.{idx}.weight"], sd[f"encoder.features.{idx}.bias"],
                         stride=2, padding=1)
            x = F.batch_norm(x, sd[b + "running_mean"], sd[b + "running_var"], sd[b + "weight"],
                             sd[b + "bias"], training=False, eps=BN_EPS)
        else:   # the HIP path folds the eval-mode BatchNorm into the conv, then rounds the folded GEMM weight
            sc = sd[b + "weight"] / torch.sqrt(sd[b + "running_var"] + BN_EPS)
            w = sd[f"encoder.features.{idx}.weight"] * sc.view(-1, 1, 1, 1)
            bias = (sd[f"encoder.features.{idx}.bias"] - sd[b + "running_mean"]) * sc + sd[b + "bias"]
            x = F.conv2d(x, w if li == 0 else _q(w), bias, stride=2, padding=1)
        x = _q(F.leaky_relu(x, LEAKY))
        if taps is not None:
            taps[f"vae_enc{li}"] = x
    return torch.flatten(x, start_dim=1)


def vae_encoder(sd, x, eps, as_written=False, want_kl=False, taps=None):
    """Encoder.forward + reparameterize, model/genconvit_vae.py:43-60.

    ``z = eps * exp(0.5*mu(x)) + mu(x)`` — the reference uses ``mu`` (not ``var``)
    for the std (:45).  ``eps`` replaces ``torch.randn_like`` (:46).  With
    ``as_written`` the three redundant ``mu`` GEMMs and the ``var`` GEMM are
    executed like the reference does (CPU-baseline fidelity); results are equal."""
    f = vae_encoder_features(sd, x, taps)
    mu = F.linear(f, _q(sd["encoder.mu.weight"]), sd["encoder.mu.bias"])
    kl = None
    if as_written or want_kl:
        var = F.linear(f, sd["encoder.var.weight"], sd["encoder.var.bias"])
        kl = 0.5 * torch.mean(-0.5 * torch.sum(1 + var - mu ** 2 - var.exp(), dim=1), dim=0)  # :58
    if as_written:
        std = torch.exp(0.5 * F.linear(f, sd["encoder.mu.weight"], sd["encoder.mu.bias"]))
        z = eps * std + F.linear(f, sd["encoder.mu.weight"], sd["encoder.mu.bias"])
    else:
        z = _q(eps * torch.exp(0.5 * mu) + mu)
    if taps is not None:
        taps["vae_mu"] = mu
        taps["vae_z"] = z
    return z, kl


def vae_decoder(sd, z, taps=None):
    """Decoder.forward, model/genconvit_vae.py:82-88 — Unflatten(256,7,7) then
    4x[ConvTranspose2d 2x2 s2 -> LeakyReLU]."""
    x = z.reshape(z.shape[0], 256, 7, 7)
    for li, idx in enumerate((0, 2, 4, 6)):
        w = sd[f"decoder.features.{idx}.weight"]
        x = _q(F.leaky_relu(F.conv_transpose2d(x, w if li == 3 else _q(w), sd[f"decoder.features.{idx}.bias"],
                                               stride=2), LEAKY))
        if taps is not None:
            taps[f"vae_dec{li}"] = x
    return x


def resize224(x_hat):
    """transforms.Resize((224,224), antialias=True) on a float tensor
    (genconvit_vae.py:105,116) == bilinear, align_corners=False; antialias is a
    no-op when upsampling."""
    return F.interpolate(x_hat, size=(224, 224), mode="bilinear", align_corners=False, antialias=True)


def vae_forward(sd, x, eps, as_written=False, want_kl=False, taps=None, merged=False):
    """GenConViTVAE.forward, model/genconvit_vae.py:107-116.  cat order is
    [backbone(orig @224), backbone(x_hat @112)]; activation is ReLU (:104).
    Returns (logits, resized reconstruction, kl or None)."""
    x = _q(x)
    z, kl = vae_encoder(sd, x, eps, as_written, want_kl, taps)
    x_hat = vae_decoder(sd, z, taps)
    # 16-bit restatement: gcv_vae_forward on its own runs the two passes as two launches (backbone(x) on a side stream);
    # inside gcv_genconvit_forward they are the two segments of one launch (``merged``: csrc/net_impl.h vae_forward)
    B, t224, t112 = x.shape[0], (x.shape[2] // 4) * (x.shape[3] // 4), (x_hat.shape[2] // 4) * (x_hat.shape[3] // 4)
    la1 = Launch(B * (t224 + t112), B) if merged else None
    x1 = convnext_tiny(sd, "convnext_backbone.", x, store_out=False, launch=la1)
    x2 = convnext_tiny(sd, "convnext_backbone.", x_hat, store_out=False, launch=la1)
    f = torch.cat((x1, x2), dim=1)
    if taps is not None:
        taps["vae_feat"] = f
    f = _q(F.relu(f))
    f = _q(F.relu(F.linear(f, _q(sd["fc.weight"]), sd["fc.bias"])))
    logits = F.linear(f, sd["fc2.weight"], sd["fc2.bias"])
    return logits, resize224(x_hat), kl


def mse_per_frame(recons, images):
    """Per-frame mean squared error; its mean over frames is the reference's
    ``nn.MSELoss()(recons, images)`` (train/train_vae.py:24,76; train.py:57)."""
    return ((recons - images) ** 2).flatten(1).mean(dim=1)


# --------------------------------------------------------------------------- ensemble + vote
def genconvit_forward(sd_ed, sd_vae, x, eps, net="genconvit", as_written=False):
    """GenConViT.forward, model/genconvit.py:66-75: 'ed' -> (B,2); 'vae' -> (B,2);
    anything else -> cat((ed, vae), dim=0) -> (2B,2)."""
    if net == "ed":
        return ed_forward(sd_ed, x)
    if net == "vae":
        return vae_forward(sd_vae, x, eps, as_written)[0]
    x1 = ed_forward(sd_ed, x)
    x2 = vae_forward(sd_vae, x, eps, as_written, merged=True)[0]
    return torch.cat((x1, x2), dim=0)


def max_prediction_value(y_pred):
    """model/pred_func.py:123-131."""
    mean_val = torch.mean(y_pred, dim=0)
    return (torch.argmax(mean_val).item(),
            mean_val[0].item() if mean_val[0] > mean_val[1] else abs(1 - mean_val[1]).item())


def vote(logits):
    """pred_vid's tail, model/pred_func.py:120: sigmoid -> mean over rows -> argmax."""
    return max_prediction_value(torch.sigmoid(logits.squeeze()))


def preprocess_frame(frames_u8):
    """model/pred_func.py:95-108 with dataset/loader.py:63-65,77: uint8 NHWC ->
    float NCHW -> /255 -> Normalize(mean, std)."""
    x = torch.as_tensor(frames_u8).float().permute(0, 3, 1, 2) / 255.0
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return (x - mean) / std


# --------------------------------------------------------------------------- Swin-T (A6)
def _swin_rel_index(ws=7):
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij"))
    cf = coords.flatten(1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)                                   # (49,49)


def _swin_attn_mask(H, W, ws, shift):
    img = torch.zeros(1, H, W, 1)
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, h, w, :] = cnt
            cnt += 1
    mw = img.view(1, H // ws, ws, W // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)   # (nW,49,49)


def swin_block(sd, p, x, H, W, nh, shift, ws=7):
    """timm 0.6.5 SwinTransformerBlock."""
    B, L, C = x.shape
    sc = x
    y = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS_SWIN).view(B, H, W, C)
    if shift:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    yw = y.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    Bw, N = yw.shape[0], ws * ws
    qkv = F.linear(yw, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(Bw, N, 3, nh, C // nh)
    q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)
    attn = (q * (C // nh) ** -0.5) @ k.transpose(-2, -1)
    bias = sd[p + "attn.relative_position_bias_table"][_swin_rel_index(ws).view(-1)].view(N, N, nh)
    attn = attn + bias.permute(2, 0, 1).unsqueeze(0)
    if shift:
        m = _swin_attn_mask(H, W, ws, shift)
        nW = m.shape[0]
        attn = attn.view(Bw // nW, nW, nh, N, N) + m.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, nh, N, N)
    attn = attn.softmax(dim=-1)
    yw = (attn @ v).transpose(1, 2).reshape(Bw, N, C)
    yw = F.linear(yw, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    y = yw.view(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
    if shift:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    x = sc + y.reshape(B, L, C)
    y = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS_SWIN)
    y = F.linear(F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])),
                 sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + y


def swin_tiny(sd, prefix, x):
    """timm 0.6.5 ``swin_tiny_patch4_window7_224`` forward -> (N,1000).  The
    reference only ever runs it once at construction (model/model_embedder.py:22)."""
    p = prefix
    x = F.conv2d(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=4)
    x = x.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (96,), sd[p + "patch_embed.norm.weight"], sd[p + "patch_embed.norm.bias"], LN_EPS_SWIN)
    H = W = 56
    for i, (dim, depth, nh) in enumerate(zip(SWIN_DIMS, SWIN_DEPTHS, SWIN_HEADS)):
        for j in range(depth):
            shift = 0 if (j % 2 == 0 or H <= 7) else 3
            x = swin_block(sd, p + f"layers.{i}.blocks.{j}.", x, H, W, nh, shift)
        if i < 3:
            B, L, C = x.shape
            y = x.view(B, H, W, C)
            y = torch.cat([y[:, 0::2, 0::2], y[:, 1::2, 0::2], y[:, 0::2, 1::2], y[:, 1::2, 1::2]], -1)
            y = y.view(B, -1, 4 * C)
            d = p + f"layers.{i}.downsample."
            y = F.layer_norm(y, (4 * C,), sd[d + "norm.weight"], sd[d + "norm.bias"], LN_EPS_SWIN)
            x = F.linear(y, sd[d + "reduction.weight"])
            H, W = H // 2, W // 2
    x = F.layer_norm(x, (768,), sd[p + "norm.weight"], sd[p + "norm.bias"], LN_EPS_SWIN)
    x = x.mean(dim=1)
    return F.linear(x, sd[p + "head.weight"], sd[p + "head.bias"])
