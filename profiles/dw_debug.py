#!/usr/bin/env python3
"""debug helper: dwconv7+LN kernel vs torch on one shape, error map by position.  usage: dw_debug.py C H n dtype"""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genconvit_amd import _lib
C, H, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dtype = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[4]]
dev = torch.device("cuda", 0)
lib = _lib.load()
g = torch.Generator().manual_seed(1)
x = ((torch.rand((n, C, H, H), generator=g) * 2 - 1) * 2).to(dtype).float()
w = (torch.rand((C, 1, 7, 7), generator=g) * 2 - 1) * 0.25
b, lw, lb = torch.rand(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5, torch.rand(C, generator=g) * 0.1
want = F.layer_norm(F.conv2d(x, w, b, padding=3, groups=C).permute(0, 2, 3, 1), (C,), lw, lb, 1e-6)
xd = x.permute(0, 2, 3, 1).contiguous().to(dev, dtype)
wdw = w.reshape(C, 49).t().contiguous().to(dev)
out = torch.zeros((n, H, H, C), dtype=dtype, device=dev)
D = lambda t: t.to(dev)
bd, lwd, lbd = D(b), D(lw), D(lb)
_lib.check(lib.gcv_k_dwconv7_ln(_lib.dtype_code(dtype), xd.data_ptr(), wdw.data_ptr(), bd.data_ptr(), lwd.data_ptr(),
                                 lbd.data_ptr(), out.data_ptr(), n, H, H, C, 1e-6, _lib.current_stream_ptr(dev)), "dw")
torch.cuda.synchronize()
err = (out.float().cpu() - want).abs()
print("max err", err.max().item())
print("per image", err.amax(dim=(1, 2, 3)).tolist())
print("per row", [round(v, 4) for v in err.amax(dim=(0, 2, 3)).tolist()])
print("per col", [round(v, 4) for v in err.amax(dim=(0, 1, 3)).tolist()])
ec = err.amax(dim=(0, 1, 2))
print("per channel (first 32)", [round(v, 4) for v in ec[:32].tolist()], "max at", ec.argmax().item())
