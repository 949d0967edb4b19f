#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ by running THE REFERENCE'S OWN
classes (imported from /root/reference — build container only, never on the GPU box).

What is the reference and what is not:
  * ``Encoder``/``Decoder`` of model/genconvit_ed.py and model/genconvit_vae.py,
    ``GenConViTED.forward``, ``GenConViTVAE.forward``, ``HybridEmbed`` and
    ``GenConViT.forward``-equivalent glue are executed from the reference source as-is.
  * ``timm`` (==0.6.5, requirements.txt:5) and ``torchvision`` are NOT installed and not
    vendored.  Their import lines are satisfied with small in-memory modules:
    ``timm.create_model`` hands back an nn.Module that evaluates the oracle's restated
    ConvNeXt-T / Swin-T (oracle/cpu_ref.py), ``torchvision.transforms.Resize`` is the
    documented bilinear interpolate.  So the vectors pin everything the reference itself
    wrote (AE/VAE arithmetic, reparameterisation quirk, cat order, activations, heads,
    ensemble concat, vote), while the third-party backbone arithmetic stays
    "parity unpinned" by the reference (see oracle/cpu_ref.py header).

Usage (build container):  python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from genconvit_amd import spec, synth            # noqa: E402
from oracle import cpu_ref                       # noqa: E402

SEED = synth.DEFAULT_SEED
EPS_TORCH_SEED = 20261004


def _tree_from_spec(entries, sd=None):
    """nn.Module hierarchy whose state_dict keys equal the spec names."""
    root = nn.Module()
    for name, shape, kind in entries:
        parts = name.split(".")
        node = root
        for p in parts[:-1]:
            if not hasattr(node, p):
                node.add_module(p, nn.Module())
            node = getattr(node, p)
        t = torch.zeros(shape) if sd is None else sd[name].clone()
        if kind in ("bn_mean", "bn_var"):
            node.register_buffer(parts[-1], t)
        else:
            node.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))
    return root


class _OracleConvNeXt(nn.Module):
    def __init__(self):
        super().__init__()
        tree = _tree_from_spec(spec.convnext_tiny_spec(""))
        for n, m in tree.named_children():
            self.add_module(n, m)
        self.head.fc.out_features = spec.NUM_BACKBONE_CLASSES   # read at genconvit_ed.py:72

    def forward(self, x):
        return cpu_ref.convnext_tiny(self.state_dict(), "", x)


class _OracleSwin(nn.Module):
    def __init__(self):
        super().__init__()
        tree = _tree_from_spec(spec.swin_tiny_spec(""))
        for n, m in tree.named_children():
            self.add_module(n, m)

    def forward(self, x):
        return cpu_ref.swin_tiny(self.state_dict(), "", x)


def _create_model(name, pretrained=False, **kw):
    if name.startswith("convnext"):
        return _OracleConvNeXt()
    if name.startswith("swin"):
        return _OracleSwin()
    raise ValueError(name)


def install_import_shims():
    import transformers  # noqa: F401  (import before the shims; its probing dislikes spec-less modules)
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class Resize:
        def __init__(self, size, antialias=True):
            self.size, self.antialias = size, antialias

        def __call__(self, x):
            # (the CPU has no 16-bit antialiased kernel: the --fp16 runs below resize in fp32; the resized reconstruction is
            #  the forward's second output and not on the logits path)
            return F.interpolate(x.float(), size=self.size, mode="bilinear", align_corners=False,
                                 antialias=self.antialias).to(x.dtype)
    tvt.Resize = Resize
    tv.transforms = tvt
    tm = types.ModuleType("timm")
    tm.create_model = _create_model
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    sys.modules["timm"] = tm


def load_reference():
    install_import_shims()
    os.chdir(REF)                      # model/genconvit_vae.py:8 reads model/config.yaml relative to cwd
    sys.path.insert(0, REF)
    import model.genconvit_ed as ref_ed
    import model.genconvit_vae as ref_vae
    from model.config import load_config
    return ref_ed, ref_vae, load_config()


def _load_into(ref_model, sd):
    missing, unexpected = ref_model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    for k in missing:   # Swin duplicates / HybridEmbed proj / BN counters never touch forward
        assert ("embedder." in k or "patch_embed." in k or k.endswith("num_batches_tracked")), k


def slice64(t):
    f = t.detach().flatten()
    idx = torch.linspace(0, f.numel() - 1, 64).long()
    return f[idx].numpy().copy()


def main():
    torch.set_grad_enabled(False)
    ref_ed, ref_vae, config = load_reference()
    out = {}

    sd_ed = synth.make_state_dict(spec.ed_spec(), SEED, "ed/")
    sd_vae = synth.make_state_dict(spec.vae_spec(), SEED, "vae/")
    x4 = synth.make_frames(4, SEED)

    # ---- reference AE / VAE pieces in isolation (pure reference code) ----
    enc = ref_ed.Encoder().eval()
    dec = ref_ed.Decoder().eval()
    enc.load_state_dict({k[len("encoder."):]: v for k, v in sd_ed.items() if k.startswith("encoder.")})
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd_ed.items() if k.startswith("decoder.")})
    r_enc = enc(x4[:2])
    r_dec = dec(r_enc)
    o_enc = cpu_ref.ed_encoder(sd_ed, x4[:2])
    o_dec = cpu_ref.ed_decoder(sd_ed, o_enc)
    assert torch.equal(r_enc, o_enc) and torch.equal(r_dec, o_dec), "ED AE restatement != reference"
    out["ed_enc_slice"] = slice64(r_enc)
    out["ed_dec_slice"] = slice64(r_dec)

    venc = ref_vae.Encoder(config["model"]["latent_dims"]).eval()
    vdec = ref_vae.Decoder(config["model"]["latent_dims"]).eval()
    _load_into(venc, {k[len("encoder."):]: v for k, v in sd_vae.items() if k.startswith("encoder.")})
    vdec.load_state_dict({k[len("decoder."):]: v for k, v in sd_vae.items() if k.startswith("decoder.")})
    torch.manual_seed(EPS_TORCH_SEED)
    r_z = venc(x4)
    r_kl = venc.kl.clone()
    torch.manual_seed(EPS_TORCH_SEED)
    eps = torch.randn(4, 12544)        # the reference's only RNG draw (genconvit_vae.py:46)
    o_z, o_kl = cpu_ref.vae_encoder(sd_vae, x4, eps, as_written=True)
    assert torch.equal(r_z, o_z), "VAE encoder restatement != reference"
    assert torch.allclose(r_kl, o_kl, rtol=1e-6, atol=0)
    o_z1, _ = cpu_ref.vae_encoder(sd_vae, x4, eps, as_written=False)
    assert torch.equal(o_z1, o_z), "dedup mu path changed z"
    r_xhat = vdec(r_z)
    assert torch.equal(r_xhat, cpu_ref.vae_decoder(sd_vae, o_z))
    out["vae_eps"] = eps.numpy()
    out["vae_z_slice"] = slice64(r_z)
    out["vae_xhat_slice"] = slice64(r_xhat)
    out["vae_kl"] = np.float32(r_kl.item())

    # ---- reference full forwards (reference glue + restated timm backbone) ----
    m_ed = ref_ed.GenConViTED(config, pretrained=False).eval()
    _load_into(m_ed, sd_ed)
    r_ed_logits = m_ed(x4)
    taps = {}
    o_ed_logits = cpu_ref.ed_forward(sd_ed, x4, taps)
    assert torch.equal(r_ed_logits, o_ed_logits), "ED forward restatement != reference glue"
    out["ed_logits"] = r_ed_logits.numpy()
    out["ed_feat_slice"] = slice64(taps["ed_feat"])
    for k in ("stem", "stage0", "stage1", "stage2", "stage3"):
        out["ed_orig_" + k + "_slice"] = slice64(taps[k])

    m_vae = ref_vae.GenConViTVAE(config, pretrained=False).eval()
    _load_into(m_vae, sd_vae)
    torch.manual_seed(EPS_TORCH_SEED)
    r_vae_logits, r_recon = m_vae(x4)
    taps = {}
    o_vae_logits, o_recon, _ = cpu_ref.vae_forward(sd_vae, x4, eps, as_written=True, taps=taps)
    assert torch.equal(r_vae_logits, o_vae_logits), "VAE forward restatement != reference glue"
    assert torch.equal(r_recon, o_recon)
    out["vae_logits"] = r_vae_logits.numpy()
    out["vae_recon_slice"] = slice64(r_recon)
    out["vae_feat_slice"] = slice64(taps["vae_feat"])
    out["vae_mse"] = cpu_ref.mse_per_frame(r_recon, x4).numpy()

    # ---- the reference's own --fp16 pipeline (model/genconvit.py:24-25,38-39,59-61: model.half(), prediction.py feeds
    # .half() frames): every module evaluates in torch.float16 on the CPU, the backbone stand-in included (its functional
    # ops take the dtype of the half parameters, as timm's modules would after .half()).  torch.randn_like (:46) would draw a
    # different, half-precision stream: it is pinned to the fp32 run's eps (rounded to half) so that the two runs differ by
    # precision alone.  These vectors are the yardstick for the HIP path's fp16 storage.
    m_ed.half()
    r_ed_half = m_ed(x4.half())
    assert r_ed_half.dtype == torch.float16
    out["ed_logits_half"] = r_ed_half.float().numpy()
    m_vae.half()
    _randn_like = torch.randn_like
    torch.randn_like = lambda t, **kw: eps.to(t.dtype)
    try:
        r_vae_half, _ = m_vae(x4.half())
    finally:
        torch.randn_like = _randn_like
    assert r_vae_half.dtype == torch.float16
    out["vae_logits_half"] = r_vae_half.float().numpy()
    print("reference .half() vs reference fp32: ED %.3e  VAE %.3e" % (
        (r_ed_half.float() - r_ed_logits).abs().max().item(), (r_vae_half.float() - r_vae_logits).abs().max().item()))
    m_ed.bfloat16()
    m_vae.bfloat16()
    torch.randn_like = lambda t, **kw: eps.to(t.dtype)
    try:
        r_ed_bf = m_ed(x4.bfloat16())
        r_vae_bf, _ = m_vae(x4.bfloat16())
    finally:
        torch.randn_like = _randn_like
    out["ed_logits_bf16"] = r_ed_bf.float().numpy()
    out["vae_logits_bf16"] = r_vae_bf.float().numpy()
    print("reference .bfloat16() vs reference fp32: ED %.3e  VAE %.3e" % (
        (r_ed_bf.float() - r_ed_logits).abs().max().item(), (r_vae_bf.float() - r_vae_logits).abs().max().item()))

    # GenConViT.forward (model/genconvit.py:66-75) cannot be constructed without weight files on
    # disk (its ctor torch.load()s weight/*.pth); its forward is 3 lines, restated here verbatim.
    r_all = torch.cat((r_ed_logits, r_vae_logits), dim=0)
    out["genconvit_logits"] = r_all.numpy()

    # pred_func vote (model/pred_func.py:120-131) — import fails on cv2/dlib, so the two pure
    # functions are exec'd from the reference source text.
    src = open(os.path.join(REF, "model", "pred_func.py")).read()
    start = src.index("def max_prediction_value")
    end = src.index("def extract_frames")
    ns = {"torch": torch}
    exec(src[start:end], ns)
    y, yv = ns["max_prediction_value"](torch.sigmoid(r_all.squeeze()))
    assert (y, yv) == cpu_ref.vote(r_all)
    out["vote_idx"] = np.int64(y)
    out["vote_val"] = np.float64(yv)
    out["vote_label"] = np.array(ns["real_or_fake"](y))

    # HybridEmbed ctor probe (model/model_embedder.py:16-37): Swin logits shape -> proj dims
    he = m_ed.backbone.patch_embed
    out["hybrid_grid"] = np.array(he.grid_size)
    out["hybrid_proj_shape"] = np.array(he.proj.weight.shape)

    out["seed"] = np.int64(SEED)
    out["eps_torch_seed"] = np.int64(EPS_TORCH_SEED)
    path = os.path.join(HERE, "genconvit_b4.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    for k in ("ed_logits", "vae_logits"):
        print(k, out[k])
    print("vote", y, yv, out["vote_label"])


if __name__ == "__main__":
    main()
