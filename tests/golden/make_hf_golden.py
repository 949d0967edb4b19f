#!/usr/bin/env python3
"""Generate tests/golden/hf_backbones.npz: outputs of an INDEPENDENT implementation of the two third-party backbones
(Hugging Face ``transformers`` ConvNeXt / Swin, instantiated from config objects, no download) on the synthetic
weights and frames the GPU parity tests use.

Why: the ConvNeXt-T / Swin-T arithmetic lives in timm==0.6.5 (requirements.txt:5 of the reference), which is absent,
and the reference holds no tests — the oracle's restatement of it is "parity unpinned" by the reference.  The
CPU-only cross-check in tests/test_oracle.py runs in the build container; this fixture lets the GPU box compare the HIP
path with numbers that did not come from oracle/cpu_ref.py's backbone code:
  cnx224 / cnx112   ConvNeXt-T (B,1000) logits at 224 and 112 px, ED backbone weights
  swin              Swin-T (B,1000) logits
  ed_logits_hf      GenConViTED.forward with the backbone evaluated by Hugging Face (glue: oracle, pinned bit-exact
  vae_logits_hf     to the reference's classes by make_golden.py); likewise GenConViTVAE.forward with the golden eps

Usage (build container; `transformers` is not needed on the GPU box):  python tests/golden/make_hf_golden.py [out.npz]
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from genconvit_amd import spec, synth            # noqa: E402
from oracle import cpu_ref                       # noqa: E402
from tests.test_oracle import _hf_convnext, _hf_swin   # noqa: E402  (key maps timm -> transformers)

torch.set_grad_enabled(False)


def strip(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def main(out):
    sd_ed = synth.make_state_dict(spec.ed_spec(), synth.DEFAULT_SEED, "ed/")
    sd_vae = synth.make_state_dict(spec.vae_spec(), synth.DEFAULT_SEED, "vae/")
    hf_ed = _hf_convnext(strip(sd_ed, "backbone."))
    hf_vae = _hf_convnext(strip(sd_vae, "convnext_backbone."))
    res = {}
    x224 = synth.make_frames(3, name="cnx224")
    x112 = torch.nn.functional.avg_pool2d(synth.make_frames(3, name="cnx112"), 2)
    res["cnx224"] = hf_ed(pixel_values=x224).logits.numpy()
    res["cnx112"] = hf_ed(pixel_values=x112).logits.numpy()
    for k, x in (("cnx224", x224), ("cnx112", x112)):
        d = np.abs(res[k] - cpu_ref.convnext_tiny(sd_ed, "backbone.", x).numpy()).max()
        print(f"{k}: |HF - oracle restatement| = {d:.2e}")
        assert d < 5e-5

    sd_sw = synth.make_state_dict(spec.swin_tiny_spec(""), synth.DEFAULT_SEED, "swin/")
    xs = synth.make_frames(3, name="swin")
    res["swin"] = _hf_swin(sd_sw)(pixel_values=xs).logits.numpy()
    d = np.abs(res["swin"] - cpu_ref.swin_tiny(sd_sw, "", xs).numpy()).max()
    print(f"swin: |HF - oracle restatement| = {d:.2e}")
    assert d < 1e-4

    # whole ED / VAE forward with the backbone swapped for the independent implementation
    gold = dict(np.load(os.path.join(HERE, "genconvit_b4.npz"), allow_pickle=False))
    x = synth.make_frames(4)
    eps = torch.from_numpy(gold["vae_eps"])
    orig = cpu_ref.convnext_tiny
    try:
        cpu_ref.convnext_tiny = lambda sd, prefix, xx, taps=None, store_out=True, launch=None: \
            (hf_ed if prefix == "backbone." else hf_vae)(pixel_values=xx).logits
        res["ed_logits_hf"] = cpu_ref.ed_forward(sd_ed, x).numpy()
        res["vae_logits_hf"] = cpu_ref.vae_forward(sd_vae, x, eps)[0].numpy()
    finally:
        cpu_ref.convnext_tiny = orig
    print("ed  |HF-backed - reference golden| =", np.abs(res["ed_logits_hf"] - gold["ed_logits"]).max())
    print("vae |HF-backed - reference golden| =", np.abs(res["vae_logits_hf"] - gold["vae_logits"]).max())
    np.savez_compressed(out, **{k: v.astype(np.float32) for k, v in res.items()})
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "hf_backbones.npz"))
