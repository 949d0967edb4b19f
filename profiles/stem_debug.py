#!/usr/bin/env python3
"""debug helper: stem conv4x4+LN kernel vs torch, error map.  usage: stem_debug.py <f16|bf16> <nchw|nhwc> res"""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genconvit_amd import _lib
dtype = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1]]
layout, res, n = sys.argv[2], int(sys.argv[3]), 2
dev = torch.device("cuda", 0)
lib = _lib.load()
g = torch.Generator().manual_seed(1)
R = lambda s, sc: (torch.rand(s, generator=g) * 2 - 1) * sc
x = R((n, 3, res, res), 2.0).to(dtype).float()
w = R((96, 3, 4, 4), 0.2).to(dtype).float()
b, lw, lb = R((96,), 0.1), R((96,), 0.5) + 1.0, R((96,), 0.1)
y = F.conv2d(x, w, b, stride=4).permute(0, 2, 3, 1)
want = F.layer_norm(y, (96,), lw, lb, 1e-6)
wp = w.reshape(96, 48).t().contiguous().to(dev)
if layout == "nchw":
    xd = x.to(dev, dtype); st = (3 * res * res, res * res, res, 1)
else:
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev, dtype); st = (res * res * 3, 1, res * 3, 3)
out = torch.zeros((n, res // 4, res // 4, 96), dtype=dtype, device=dev)
bd, lwd, lbd = b.to(dev), lw.to(dev), lb.to(dev)
_lib.check(lib.gcv_k_stem_ln(_lib.dtype_code(dtype), xd.data_ptr(), *st, wp.data_ptr(), bd.data_ptr(), lwd.data_ptr(),
                             lbd.data_ptr(), out.data_ptr(), n, res // 4, res // 4, 1e-6, _lib.current_stream_ptr(dev)), "stem")
torch.cuda.synchronize()
err = (out.float().cpu() - want).abs()
print("max err", err.max().item(), "mean", err.mean().item())
print("per image", err.amax(dim=(1, 2, 3)).tolist())
print("per row (first 16)", [round(v, 3) for v in err.amax(dim=(0, 2, 3)).tolist()[:16]])
print("per col (first 16)", [round(v, 3) for v in err.amax(dim=(0, 1, 3)).tolist()[:16]])
print("per channel", [round(v, 2) for v in err.amax(dim=(0, 1, 2)).tolist()])
print("token 0 got ", [round(v, 3) for v in out[0, 0, 0, :12].float().cpu().tolist()])
print("token 0 want", [round(v, 3) for v in want[0, 0, 0, :12].tolist()])
et = err.amax(dim=3).reshape(-1)
print("bad tokens:", int((et > 0.05).sum()), "of", et.numel())
bylane = [round(et[l::32].max().item(), 2) for l in range(32)]
print("max err by token % 32:", bylane)
ntile = et.numel() // 32
print("max err by tile (first 24):", [round(et[32 * t:32 * t + 32].max().item(), 2) for t in range(min(ntile, 24))])
ec = err.reshape(-1, 96)
bad = (ec > 0.05)
print("bad count by channel%32:", [int(bad[:, c::32].sum()) for c in range(32)])
g2 = ((out.float().cpu() - lb) / lw).reshape(-1, 96)
yy = y.reshape(-1, 96)
for t in (0, 1, 33, 100):
    A = torch.stack([yy[t], torch.ones(96)], 1)
    sol = torch.linalg.lstsq(A, g2[t].unsqueeze(1)).solution.squeeze()
    res_ = (A @ sol - g2[t]).abs().max().item()
    print(f"token {t}: got ~ {sol[0]:.4f} * conv + {sol[1]:.4f}, residual {res_:.4f}; true rstd {1/yy[t].std(unbiased=False):.4f} mean {yy[t].mean():.4f}")
