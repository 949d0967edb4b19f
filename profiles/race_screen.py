#!/usr/bin/env python3
"""Race screen for the ring-pipelined MLP kernels (xs_pw1 + pw2f at C = 384, xs_mlp at C = 192): the same launch repeated
under varying load must give bit-identical results (a ring WAR / RAW slip shows up as a run-to-run difference long before
it shows up as a tolerance failure), and the first result must match the fp32 reference.  One process, bounded."""
import math, os, sys, threading
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genconvit_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
R = lambda *s: (torch.rand(*s, device=dev) * 2 - 1)
bad = 0
for dt, code, tol in ((torch.float16, _lib.GCV_F16, 8e-3), (torch.bfloat16, _lib.GCV_BF16, 6e-2)):
    for C, M in ((384, 50176), (384, 31360), (384, 7840), (384, 999), (192, 200704), (192, 125440), (192, 31360), (192, 4099)):
        x, res = R(M, C).to(dt), R(M, C).to(dt)
        w1 = (R(4 * C, C) / math.sqrt(C)).to(dt); w2 = R(C, 4 * C) / math.sqrt(4 * C)
        b1, b2, g = R(4 * C) * 0.1, R(C) * 0.1, R(C) * 0.5
        side = torch.cuda.Stream()
        noise = R(64 << 20 // 4)
        outs = []
        for it in range(12):
            out = res.clone()
            if it % 3 == 1:                       # uneven load: a copy kernel on another stream beside the launch
                with torch.cuda.stream(side):
                    noise.mul_(1.0001)
            _lib.check(lib.gcv_k_fused_mlp(code, C, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                           g.data_ptr(), out.data_ptr(), out.data_ptr(), M, _lib.current_stream_ptr(dev)), "mlp")
            torch.cuda.synchronize()
            outs.append(out)
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        n = min(M, 4096)
        h = F.gelu(x[:n].float() @ w1.float().t() + b1).to(dt).float()
        want = res[:n].float() + g * (h @ w2.to(dt).float().t() + b2)
        err = (outs[0][:n].float() - want).abs().max().item()
        ok = same and err <= tol
        bad += not ok
        print(f"{str(dt):15s} C={C} M={M:7d}: 12 runs bit-identical: {same}; max err vs fp32 math on the first {n} rows {err:.2e} {'ok' if ok else 'FAIL'}")
sys.exit(1 if bad else 0)
