"""Per-kernel parity on the MI355X: every HIP kernel (called through the C ABI's gcv_k_* entry
points) against a plain PyTorch fp32 CPU reference of the same op.  fp32 storage must agree to
~1e-5; 16-bit storage is checked against the same fp32 math on inputs rounded to that dtype."""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

from genconvit_amd import _lib
from tests import kutil
from tests.kutil import DTYPES, dev, gemm, ptr, rnd, tol

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

ALL = ["f32", "bf16", "f16"]


_KEEP = []


def D(t, dtype=None):
    """device copy that stays alive for the duration of the test (the caching allocator would
    otherwise hand a freed temporary's block to the next one before the kernel has run)"""
    d = t.to(dev()) if dtype is None else t.to(dev(), dtype)
    _KEEP.append(d)
    if len(_KEEP) > 64:
        torch.cuda.synchronize()
        del _KEEP[:32]
    return d


def q(t, dtype):
    """round a fp32 CPU tensor through the storage dtype"""
    return t.to(dtype).float()


def act_ref(x, act):
    return {0: lambda v: v, 1: F.relu, 2: F.gelu, 3: lambda v: F.leaky_relu(v, 0.01)}[act](x)


def assert_close(got, want, atol, what=""):
    err = (got.float().cpu() - want).abs().max().item()
    assert err <= atol, f"{what}: max abs err {err:.3e} > {atol:.3e}"


# ----------------------------------------------------------------------------- GEMM, plain A
@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("M,N,K,act", [
    (300, 384, 96, 2),      # 128x96 tiles, M tail, GELU (ConvNeXt pw1 shape class)
    (200, 1000, 768, 0),    # 128x128 tiles with N tail (head fc)
    (40, 500, 2000, 2),     # 64x128 tiles, N tail (GenConViT fc)
    (8, 1000, 768, 1),      # 32x128 tiles
    (130, 192, 1536, 0),    # downsample-conv shape class, long K
    (256, 96, 48, 3),       # K shorter than one 16-bit K tile
    (9000, 384, 384, 2),    # pw1 at K = 384 on the LDS-DMA kernel: 142 tiles + an M tail, GELU
    (8200, 192, 384, 0),    # ... one N tile, no activation
    (70000, 1536, 384, 2),  # ... the stage-2 shape class at full N: several rounds of workgroups
])
def test_gemm_bias_act(dt, M, N, K, act):
    dtype = DTYPES[dt]
    A, W = q(rnd((M, K), 1), dtype), q(rnd((N, K), 2, 1 / math.sqrt(K)), dtype)
    bias = rnd((N,), 3, 0.1)
    ldc = N + 8
    C = torch.full((M, ldc), 7.0, dtype=dtype, device=dev())
    gemm(dtype, _lib.A_PLAIN, _lib.EPI_BIAS_ACT, A.to(dev(), dtype), W.to(dev(), dtype), C, M, N, K, lda=K, ldc=ldc,
         bias=bias.to(dev()), act=act)
    want = act_ref(A @ W.t() + bias, act)
    assert_close(C[:, :N], want, tol(dtype, 2.0), "gemm")
    assert torch.all(C[:, N:].float() == 7.0), "padding columns were overwritten"


@pytest.mark.parametrize("dt", ALL)
def test_gelu_nan_behaviour_is_documented(dt):
    """A NaN pre-activation (here: injected through the bias) in the GELU epilogue.  fp32 storage (exact erf, float max)
    propagates every NaN.  The 16-bit paths take max(x, 0) as an integer max of the bit pattern (csrc/gemm.h GeluH16): a
    NaN with a clear sign bit still propagates, one with the sign bit set comes out as a finite number — accepted and
    documented there; this test pins that behaviour so that a change of it is a decision."""
    dtype = DTYPES[dt]
    M, N, K = 300, 384, 96
    A, W = q(rnd((M, K), 1), dtype), q(rnd((N, K), 2, 1 / math.sqrt(K)), dtype)
    bias = rnd((N,), 3, 0.1)
    bias[5] = float("nan")
    bias[7] = torch.tensor([0xFFC00000 - (1 << 32)], dtype=torch.int32).view(torch.float32)[0]   # NaN, sign bit set
    C = torch.zeros((M, N), dtype=dtype, device=dev())
    gemm(dtype, _lib.A_PLAIN, _lib.EPI_BIAS_ACT, D(A, dtype), D(W, dtype), C, M, N, K, lda=K, ldc=N, bias=D(bias), act=2)
    out = C.float().cpu()
    clean = [c for c in range(N) if c not in (5, 7)]
    assert torch.isfinite(out[:, clean]).all()
    assert torch.isnan(out[:, 5]).all(), "a positive NaN must propagate in every dtype"
    if dtype == torch.float32:
        assert torch.isnan(out[:, 7]).all(), "fp32 storage propagates every NaN"
    else:
        assert torch.isfinite(out[:, 7]).all(), "documented: sign-bit NaN -> finite in the packed 16-bit GELU"


@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("M,N,K", [(300, 96, 384), (150, 256, 64), (129, 768, 3072)])
def test_gemm_layerscale_residual(dt, M, N, K):
    dtype = DTYPES[dt]
    A, W = q(rnd((M, K), 1), dtype), q(rnd((N, K), 2, 1 / math.sqrt(K)), dtype)
    bias, gamma = rnd((N,), 3, 0.1), rnd((N,), 4, 0.5)
    X = q(rnd((M, N), 5), dtype)
    Xd = X.to(dev(), dtype)
    gemm(dtype, _lib.A_PLAIN, _lib.EPI_RESID, A.to(dev(), dtype), W.to(dev(), dtype), Xd, M, N, K, lda=K, ldc=N,
         bias=bias.to(dev()), gamma=gamma.to(dev()), resid=Xd)          # in place, like the forward does
    want = X + gamma * (A @ W.t() + bias)
    assert_close(Xd, want, tol(dtype, 2.0), "gemm resid")


@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("M,N,K,splitk,kps", [(8, 256, 512, 4, 128), (40, 384, 1024, 2, 512), (130, 128, 640, 3, 256)])
def test_gemm_splitk(dt, M, N, K, splitk, kps):
    dtype = DTYPES[dt]
    A, W = q(rnd((M, K), 1), dtype), q(rnd((N, K), 2, 1 / math.sqrt(K)), dtype)
    part = torch.zeros((splitk, M, N), dtype=torch.float32, device=dev())
    gemm(dtype, _lib.A_PLAIN, _lib.EPI_SPLITK, A.to(dev(), dtype), W.to(dev(), dtype), None, M, N, K, lda=K,
         partial=part, splitk=splitk, k_per_split=kps)
    assert_close(part.sum(0), A @ W.t(), 2e-5 if dt == "f32" else 1e-4, "splitk")


# ----------------------------------------------------------------------------- conv as GEMM
@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("n,H,cin,cout", [(2, 16, 16, 32), (3, 8, 32, 64), (2, 14, 128, 256), (1, 28, 64, 128),
                                          (3, 112, 16, 32)])     # 37632 rows: 294 tiles (256-row tiles were tried for such launches: 99 vs 79 us at B=128)
def test_conv3x3_relu_maxpool(dt, n, H, cin, cout):
    """ED encoder layers 2-5 (model/genconvit_ed.py:18-32): conv3x3 s1 p1 -> ReLU -> maxpool2."""
    dtype = DTYPES[dt]
    x = q(rnd((n, cin, H, H), 1), dtype)
    w = q(rnd((cout, cin, 3, 3), 2, 1 / math.sqrt(9 * cin)), dtype)
    b = rnd((cout,), 3, 0.1)
    want = F.max_pool2d(F.relu(F.conv2d(x, w, b, padding=1)), 2).permute(0, 2, 3, 1)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev(), dtype)
    wt = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(dev(), dtype)
    out = torch.zeros((n, H // 2, H // 2, cout), dtype=dtype, device=dev())
    gemm(dtype, _lib.A_IM2COL3_POOL, _lib.EPI_POOL4, x_nhwc, wt, out, n * H * H, cout, 9 * cin, ldc=cout,
         bias=b.to(dev()), act=1, H=H, W=H, cin_log2=int(math.log2(cin)))
    assert_close(out, want, tol(dtype, 2.0), "conv3 pool")


@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("n,H,cin,cout", [(2, 16, 16, 32), (3, 8, 32, 64), (2, 28, 64, 128)])
def test_conv3x3_stride2_leaky(dt, n, H, cin, cout):
    """VAE encoder layers 2-4 (model/genconvit_vae.py:19-30), BatchNorm folded by the caller."""
    dtype = DTYPES[dt]
    x = q(rnd((n, cin, H, H), 1), dtype)
    w = q(rnd((cout, cin, 3, 3), 2, 1 / math.sqrt(9 * cin)), dtype)
    b = rnd((cout,), 3, 0.1)
    want = F.leaky_relu(F.conv2d(x, w, b, stride=2, padding=1), 0.01).permute(0, 2, 3, 1)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev(), dtype)
    wt = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(dev(), dtype)
    out = torch.zeros((n, H // 2, H // 2, cout), dtype=dtype, device=dev())
    gemm(dtype, _lib.A_IM2COL3_S2, _lib.EPI_BIAS_ACT, x_nhwc, wt, out, n * (H // 2) ** 2, cout, 9 * cin, ldc=cout,
         bias=b.to(dev()), act=3, H=H, W=H, cin_log2=int(math.log2(cin)))
    assert_close(out, want, tol(dtype, 2.0), "conv3 s2")


@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("n,H,cin,cout,act", [(2, 7, 256, 128, 1), (3, 14, 64, 32, 3), (2, 28, 32, 16, 1)])
def test_conv_transpose_2x2(dt, n, H, cin, cout, act):
    """ED / VAE decoder ConvTranspose2d(k=2,s=2) layers (genconvit_ed.py:44-55, genconvit_vae.py:68-76)."""
    dtype = DTYPES[dt]
    x = q(rnd((n, cin, H, H), 1), dtype)
    w = q(rnd((cin, cout, 2, 2), 2, 1 / math.sqrt(cin)), dtype)
    b = rnd((cout,), 3, 0.1)
    want = act_ref(F.conv_transpose2d(x, w, b, stride=2), act).permute(0, 2, 3, 1)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev(), dtype)
    wt = w.permute(2, 3, 1, 0).reshape(4 * cout, cin).contiguous().to(dev(), dtype)    # ((dy,dx,co), ci)
    out = torch.zeros((n, 2 * H, 2 * H, cout), dtype=dtype, device=dev())
    gemm(dtype, _lib.A_PLAIN, _lib.EPI_CONVT, x_nhwc, wt, out, n * H * H, 4 * cout, cin, lda=cin, bias=b.to(dev()),
         act=act, H=H, W=H, cout_log2=int(math.log2(cout)))
    assert_close(out, want, tol(dtype, 2.0), "convT")


# ----------------------------------------------------------------------------- ConvNeXt pieces
@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("layout,res", [("nchw", 224), ("nhwc", 112), ("nchw", 32), ("nchw", 20), ("nhwc", 20)])   # 20: a 25-token tail tile
def test_stem_conv4x4_layernorm(dt, layout, res):
    dtype = DTYPES[dt]
    n = 2
    x = q(rnd((n, 3, res, res), 1, 2.0), dtype)
    w = q(rnd((96, 3, 4, 4), 2, 0.2), dtype)          # 16-bit storage runs the stem on the matrix pipe: 16-bit weights
    b, lw, lb = rnd((96,), 3, 0.1), rnd((96,), 4, 0.5) + 1.0, rnd((96,), 5, 0.1)
    y = F.conv2d(x, w, b, stride=4).permute(0, 2, 3, 1)
    want = F.layer_norm(y, (96,), lw, lb, 1e-6)
    wp = w.reshape(96, 48).t().contiguous().to(dev())
    if layout == "nchw":
        xd = x.to(dev(), dtype)
        st = (3 * res * res, res * res, res, 1)
    else:
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev(), dtype)
        st = (res * res * 3, 1, res * 3, 3)
    out = torch.zeros((n, res // 4, res // 4, 96), dtype=dtype, device=dev())
    kutil.call("gcv_k_stem_ln", _lib.dtype_code(dtype), ptr(xd), *st, ptr(wp), ptr(D(b)), ptr(D(lw)),
               ptr(D(lb)), ptr(out), n, res // 4, res // 4, 1e-6)
    assert_close(out, want, tol(dtype, 3.0), "stem")


@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("C,H,n", [(96, 56, 2), (96, 28, 3), (192, 28, 2), (192, 14, 1), (384, 14, 2), (384, 7, 3),
                                   (768, 7, 2), (768, 3, 3),
                                   (768, 1, 5), (768, 2, 3), (768, 4, 2),      # whole-map kernel of the tiny stage-3 maps (with (768, 3, 3))
                                   # launches of more than 128 seven-row bands keep seven-row bands (the large-batch rule);
                                   # the few-image cases above run the two- to four-row bands of small launches
                                   (96, 56, 17), (384, 14, 70),
                                   # 16-bit storage, 56-pixel C = 96 maps in bands of 14 rows and more: the matrix-pipe kernel
                                   # (dwconv_mfma.h; fp32 storage stays on the VALU kernel): four 14-row bands, ragged 19/19/18
                                   (96, 56, 64), (96, 56, 100)])
def test_dwconv7x7_layernorm(dt, C, H, n):
    """ConvNeXt block front half: depthwise 7x7 p3 + LayerNorm(C, eps 1e-6), NHWC."""
    dtype = DTYPES[dt]
    x = q(rnd((n, C, H, H), 1, 2.0), dtype)
    w = rnd((C, 1, 7, 7), 2, 0.25)
    b, lw, lb = rnd((C,), 3, 0.1), rnd((C,), 4, 0.5) + 1.0, rnd((C,), 5, 0.1)
    y = F.conv2d(x, w, b, padding=3, groups=C).permute(0, 2, 3, 1)
    want = F.layer_norm(y, (C,), lw, lb, 1e-6)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev(), dtype)
    wdw = w.reshape(C, 49).t().contiguous().to(dev())
    out = torch.zeros((n, H, H, C), dtype=dtype, device=dev())
    kutil.call("gcv_k_dwconv7_ln", _lib.dtype_code(dtype), ptr(xd), ptr(wdw), ptr(D(b)), ptr(D(lw)),
               ptr(D(lb)), ptr(out), n, H, H, C, 1e-6)
    assert_close(out, want, tol(dtype, 3.0), "dwconv_ln")


@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("C,H", [(96, 56), (192, 28), (384, 14), (384, 7), (192, 14)])
def test_layernorm2d_space_to_depth(dt, C, H):
    """Downsample front half: LayerNorm2d + 2x2 patch gather; odd sizes drop the last row/col (7 -> 3)."""
    dtype = DTYPES[dt]
    n = 2
    x = q(rnd((n, H, H, C), 1, 2.0), dtype)
    lw, lb = rnd((C,), 4, 0.5) + 1.0, rnd((C,), 5, 0.1)
    y = F.layer_norm(x, (C,), lw, lb, 1e-6)
    Ho = H // 2
    y = y[:, :2 * Ho, :2 * Ho]
    want = y.reshape(n, Ho, 2, Ho, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(n, Ho, Ho, 4 * C)
    out = torch.zeros((n, Ho, Ho, 4 * C), dtype=dtype, device=dev())
    kutil.call("gcv_k_ln_patchify", _lib.dtype_code(dtype), ptr(D(x, dtype)), ptr(D(lw)),
               ptr(D(lb)), ptr(out), n, H, H, C, 1e-6)
    assert_close(out, want, tol(dtype, 3.0), "ln_patchify")


@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("HW", [49, 9])
def test_avgpool_layernorm(dt, HW):
    dtype = DTYPES[dt]
    n, C = 5, 768
    x = q(rnd((n, HW, C), 1, 2.0), dtype)
    lw, lb = rnd((C,), 4, 0.5) + 1.0, rnd((C,), 5, 0.1)
    want = F.layer_norm(x.mean(1), (C,), lw, lb, 1e-6)
    out = torch.zeros((n, C), dtype=dtype, device=dev())
    kutil.call("gcv_k_pool_ln", _lib.dtype_code(dtype), ptr(D(x, dtype)), ptr(D(lw)), ptr(D(lb)),
               ptr(out), n, HW, C, 1e-6)
    assert_close(out, want, tol(dtype, 3.0), "pool_ln")


@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("C", [96, 384, 1536])
def test_layernorm_rows(dt, C):
    dtype = DTYPES[dt]
    rows = 37
    x = q(rnd((rows, C), 1, 2.0), dtype)
    lw, lb = rnd((C,), 4, 0.5) + 1.0, rnd((C,), 5, 0.1)
    want = F.layer_norm(x, (C,), lw, lb, 1e-5)
    out = torch.zeros((rows, C), dtype=dtype, device=dev())
    kutil.call("gcv_k_layernorm_rows", _lib.dtype_code(dtype), ptr(D(x, dtype)), ptr(D(lw)),
               ptr(D(lb)), ptr(out), rows, C, 1e-5)
    assert_close(out, want, tol(dtype, 3.0), "layernorm_rows")


# ----------------------------------------------------------------------------- AE / VAE small kernels
@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("pool", [True, False])
@pytest.mark.parametrize("H", [32, 224, 20])          # 20: not a multiple of 8 -> the VALU kernel for every dtype
def test_first_conv_3_to_16(dt, pool, H):
    dtype = DTYPES[dt]
    n = 2
    x = q(rnd((n, 3, H, H), 1, 2.0), dtype)
    w = q(rnd((16, 3, 3, 3), 2, 0.3), dtype)          # 16-bit storage runs on the matrix pipe: 16-bit weights
    b = rnd((16,), 3, 0.1)
    if pool:
        want = F.max_pool2d(F.relu(F.conv2d(x, w, b, padding=1)), 2)
        act = 1
    else:
        want = F.leaky_relu(F.conv2d(x, w, b, stride=2, padding=1), 0.01)
        act = 3
    want = want.permute(0, 2, 3, 1)
    wp = w.permute(2, 3, 1, 0).reshape(27, 16).contiguous().to(dev())
    out = torch.zeros((n, H // 2, H // 2, 16), dtype=dtype, device=dev())
    kutil.call("gcv_k_conv3_first", _lib.dtype_code(dtype), ptr(D(x, dtype)), 3 * H * H, H * H, H, 1, ptr(wp),
               ptr(D(b)), ptr(out), n, H, H, int(pool), act)
    assert_close(out, want, tol(dtype, 3.0), "conv3_first")


@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("act", [1, 3])
def test_last_conv_transpose_16_to_3(dt, act):
    dtype = DTYPES[dt]
    n, H = 2, 12
    x = q(rnd((n, 16, H, H), 1, 2.0), dtype)
    w = rnd((16, 3, 2, 2), 2, 0.3)
    b = rnd((3,), 3, 0.1)
    want = act_ref(F.conv_transpose2d(x, w, b, stride=2), act).permute(0, 2, 3, 1)
    wp = w.permute(0, 2, 3, 1).reshape(16, 12).contiguous().to(dev())
    out = torch.zeros((n, 2 * H, 2 * H, 3), dtype=dtype, device=dev())
    kutil.call("gcv_k_convt2_small", _lib.dtype_code(dtype), ptr(D(x.permute(0, 2, 3, 1).contiguous(), dtype)),
               ptr(wp), ptr(D(b)), ptr(out), n, H, H, act)
    assert_close(out, want, tol(dtype, 3.0), "convt2_small")


@pytest.mark.parametrize("dt", ALL)
def test_reparameterise_from_splitk(dt):
    """z = eps*exp(0.5*mu) + mu with std taken from mu (model/genconvit_vae.py:45-47), written NHWC."""
    dtype = DTYPES[dt]
    B, N, S = 3, 12544, 4
    part = rnd((S, B, N), 1, 0.4)
    bias, eps = rnd((N,), 2, 0.1), torch.randn((B, N), generator=torch.Generator().manual_seed(3))
    mu = part.sum(0) + bias
    z = eps * torch.exp(0.5 * mu) + mu
    want = z.reshape(B, 256, 49).permute(0, 2, 1).reshape(B, N)
    mu_out = torch.zeros((B, N), dtype=torch.float32, device=dev())
    zout = torch.zeros((B, N), dtype=dtype, device=dev())
    kutil.call("gcv_k_reparam", _lib.dtype_code(dtype), ptr(D(part)), S, ptr(D(bias)), ptr(D(eps)),
               ptr(mu_out), ptr(zout), B, N)
    assert_close(mu_out, mu, 1e-5, "mu")
    assert_close(zout, want, tol(dtype, 8.0), "z")


@pytest.mark.parametrize("dt", ALL)
def test_head_tail(dt):
    dtype = DTYPES[dt]
    B, K = 9, 500
    h = q(rnd((B, K), 1), dtype)
    w, b = rnd((2, K), 2, 0.05), rnd((2,), 3, 0.1)
    out = torch.zeros((B, 2), dtype=torch.float32, device=dev())
    kutil.call("gcv_k_head_tail", _lib.dtype_code(dtype), ptr(D(h, dtype)), ptr(D(w)), ptr(D(b)),
               ptr(out), B, K)
    assert_close(out, h @ w.t() + b, 1e-5, "head tail")


@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("act", [1, 2])
def test_head_tail_with_splitk_reduce(dt, act):
    """The head as the networks run it: fc's split-K partials summed, + bias, ReLU (vae) / exact GELU (ed), rounded to the
    storage dtype, then fc2 (model/genconvit_ed.py:87, model/genconvit_vae.py:114)."""
    dtype = DTYPES[dt]
    B, K, S = 37, 500, 8
    part = rnd((S, B, K), 1, 0.5)
    b1 = rnd((K,), 2, 0.1)
    w, b = rnd((2, K), 3, 0.05), rnd((2,), 4, 0.1)
    h = q(act_ref(part.sum(0) + b1, act), dtype)
    out = torch.zeros((B, 2), dtype=torch.float32, device=dev())
    kutil.call("gcv_k_head_tail_splitk", _lib.dtype_code(dtype), ptr(D(part)), S, ptr(D(b1)), act, ptr(D(w)), ptr(D(b)),
               ptr(out), B, K)
    # (a hidden value that sits on a rounding boundary of the storage dtype may round the other way: one ulp of h times w)
    assert_close(out, h @ w.t() + b, {torch.float32: 1e-5, torch.float16: 2e-4, torch.bfloat16: 2e-3}[dtype], "head tail split-K")


@pytest.mark.parametrize("dt", ALL)
def test_bilinear_resize_and_mse(dt):
    """transforms.Resize((224,224)) of x_hat (genconvit_vae.py:116) + per-frame MSE (train/train_vae.py:24)."""
    dtype = DTYPES[dt]
    B = 2
    xhat = q(rnd((B, 3, 112, 112), 1, 2.0), dtype)
    img = q(rnd((B, 3, 224, 224), 2, 2.0), dtype)
    want = F.interpolate(xhat, size=(224, 224), mode="bilinear", align_corners=False, antialias=True)
    recon = torch.zeros((B, 3, 224, 224), dtype=dtype, device=dev())
    msepart = torch.zeros((B, 196), dtype=torch.float32, device=dev())
    mse = torch.zeros((B,), dtype=torch.float32, device=dev())
    kutil.call("gcv_k_resize_mse", _lib.dtype_code(dtype), ptr(D(xhat.permute(0, 2, 3, 1).contiguous(), dtype)),
               ptr(D(img, dtype)), ptr(recon), ptr(msepart), ptr(mse), B)
    assert_close(recon, want, tol(dtype, 3.0), "resize")
    want_mse = ((want - img) ** 2).flatten(1).mean(1)
    assert_close(mse, want_mse, 1e-4 * float(want_mse.max()), "mse")


def test_vote_sigmoid_mean():
    logits = rnd((37, 2), 1, 3.0)
    got = _lib.vote(logits.to(dev()))
    torch.cuda.synchronize()
    assert_close(got, torch.sigmoid(logits).mean(0), 1e-6, "vote")


# ----------------------------------------------------------------------------- error behaviour
def test_errors_are_raised_not_swallowed():
    with pytest.raises(_lib.GenConViTHipError, match="multiple of the 16-byte chunk"):
        A = torch.zeros((4, 6), device=dev())
        gemm(torch.float32, _lib.A_PLAIN, _lib.EPI_BIAS_ACT, A, A, A, 4, 4, 6, lda=6, ldc=4)
    with pytest.raises(_lib.GenConViTHipError, match="C must be one of"):
        x = torch.zeros((1, 7, 7, 100), device=dev())
        kutil.call("gcv_k_dwconv7_ln", 0, ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), 1, 7, 7, 100, 1e-6)


# ----------------------------------------------------------------------------- Swin-T pieces (row A6)
@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("H,C,nH,shift", [(14, 96, 3, 0), (14, 96, 3, 3), (28, 192, 6, 3), (7, 768, 24, 0)])
def test_swin_window_attention(dt, H, C, nH, shift):
    """W-MSA / SW-MSA (timm 0.6.5 WindowAttention + cyclic shift + mask) on a (B,H,W,3C) qkv tensor."""
    from oracle.cpu_ref import _swin_attn_mask, _swin_rel_index
    dtype = DTYPES[dt]
    B, ws, N = 2, 7, 49
    qkv = q(rnd((B, H, H, 3 * C), 1, 1.5), dtype)
    table = rnd((169, nH), 2, 0.5)
    y = qkv
    if shift:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    yw = y.view(B, H // ws, ws, H // ws, ws, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(-1, N, 3, nH, C // nH)
    qq, kk, vv = yw.permute(2, 0, 3, 1, 4).unbind(0)
    attn = (qq * (C // nH) ** -0.5) @ kk.transpose(-2, -1)
    attn = attn + table[_swin_rel_index(ws).view(-1)].view(N, N, nH).permute(2, 0, 1).unsqueeze(0)
    if shift:
        m = _swin_attn_mask(H, H, ws, shift)
        attn = (attn.view(B, m.shape[0], nH, N, N) + m.unsqueeze(1).unsqueeze(0)).view(-1, nH, N, N)
    o = (attn.softmax(-1) @ vv).transpose(1, 2).reshape(-1, N, C)
    o = o.view(B, H // ws, H // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, H, C)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    out = torch.zeros((B, H, H, C), dtype=dtype, device=dev())
    kutil.call("gcv_k_swin_window_attn", _lib.dtype_code(dtype), ptr(D(qkv, dtype)), ptr(D(table)), ptr(out), B, H, H, C,
               nH, shift)
    assert_close(out, o, tol(dtype, 2.0), "window attention")


@pytest.mark.parametrize("dt", ALL)
@pytest.mark.parametrize("C,H", [(96, 56), (192, 28), (384, 14)])
def test_swin_patch_merging_layernorm(dt, C, H):
    dtype = DTYPES[dt]
    n = 2
    x = q(rnd((n, H, H, C), 1, 2.0), dtype)
    lw, lb = rnd((4 * C,), 4, 0.5) + 1.0, rnd((4 * C,), 5, 0.1)
    y = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    want = F.layer_norm(y, (4 * C,), lw, lb, 1e-5)
    out = torch.zeros((n, H // 2, H // 2, 4 * C), dtype=dtype, device=dev())
    kutil.call("gcv_k_patch_merge_ln", _lib.dtype_code(dtype), ptr(D(x, dtype)), ptr(D(lw)), ptr(D(lb)), ptr(out), n, H,
               H, C, 1e-5)
    assert_close(out, want, tol(dtype, 3.0), "patch merging")


@pytest.mark.parametrize("dt", ALL)
def test_mean_over_tokens(dt):
    dtype = DTYPES[dt]
    x = q(rnd((3, 49, 768), 1, 2.0), dtype)
    out = torch.zeros((3, 768), dtype=dtype, device=dev())
    kutil.call("gcv_k_mean_tokens", _lib.dtype_code(dtype), ptr(D(x, dtype)), ptr(out), 3, 49, 768)
    assert_close(out, x.mean(1), tol(dtype, 1.0), "mean tokens")


# ----------------------------------------------------------------------------- fused ConvNeXt MLP
@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("C,M", [(96, 256), (96, 1000), (192, 300), (192, 37),
                                 (96, 70013),       # >= 65536 tokens at C=96: the LDS-resident persistent kernel
                                 (192, 70013),      # C=192 x-stationary kernel: 274 passes on 256 workgroups, i.e. a second pass (ring wrap-around, x reload) in some
                                 (384, 128), (384, 1000), (384, 6272),      # C=384 kernel pair, small launches: 128-token pw2 tiles, pw1 hidden range split up to 24 ways
                                 (384, 40000),      # 157 tiles (hidden range split, 256x192 pw2 tiles)
                                 (384, 60013)])     # ... 235 token tiles: unsplit pw1, full-width 256x384 pw2 tiles, ragged last block
def test_fused_mlp_layerscale_residual(dt, C, M):
    """timm ConvNeXtBlock tail: fc1 -> exact GELU -> fc2 -> * gamma -> + shortcut, hidden kept on chip."""
    dtype = DTYPES[dt]
    x = q(rnd((M, C), 1, 1.5), dtype)
    w1 = q(rnd((4 * C, C), 2, 1 / math.sqrt(C)), dtype)
    w2 = q(rnd((C, 4 * C), 3, 1 / math.sqrt(4 * C)), dtype)
    b1, b2, gamma = rnd((4 * C,), 4, 0.1), rnd((C,), 5, 0.1), rnd((C,), 6, 0.5)
    res = q(rnd((M, C), 7), dtype)
    h = q(F.gelu(x @ w1.t() + b1), dtype)          # the kernel rounds the hidden activation to T for GEMM2
    want = res + gamma * (h @ w2.t() + b2)
    out = D(res, dtype).clone()
    kutil.call("gcv_k_fused_mlp", _lib.dtype_code(dtype), C, ptr(D(x, dtype)), ptr(D(w1, dtype)), ptr(D(b1)),
               ptr(D(w2)), ptr(D(b2)), ptr(D(gamma)), ptr(out), ptr(out), M)
    assert_close(out, want, tol(dtype, 2.0), "fused mlp")


# segments: (images, H, W) per segment, in token order
_LNP_CASES = [
    (96, [(22, 56, 56)]),                          # one segment, 68 992 tokens: 2 156 wave tiles, none ragged
    (96, [(21, 56, 56), (7, 28, 28)]),             # 65 856 + 5 488 tokens: the boundary is a tile edge (2 058 * 32)
    (96, [(21, 56, 56), (9, 28, 28), (3, 14, 14)]),   # three geometries, last tile ragged (73 500 tokens)
    (96, [(5, 56, 56), (67, 28, 28)]),             # 15 680 + 52 528: odd image count behind a boundary at token 490 * 32
    (96, [(3, 30, 30), (23, 56, 54)]),             # non-square maps, boundary at 2 700 = 84 * 32 + 12: a tile straddles it
    (192, [(3, 28, 28)]),                          # 2 352 tokens: ragged last pass (9.2 passes of 256)
    (192, [(5, 28, 28), (5, 14, 14)]),             # 3 920 + 980: boundary at 122 * 32 + 16, inside a wave tile
    (192, [(1, 14, 14), (2, 28, 28), (3, 6, 10)]),   # boundary at 196 = 6 * 32 + 4, non-square tail segment, 1 944 tokens
    (192, [(67, 28, 28), (67, 14, 14)]),           # 65 660 tokens: 257 passes on 256 persistent workgroups (second pass)
]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("C,segs", _LNP_CASES)
def test_fused_mlp_layernorm_patchify_epilogue(dt, C, segs):
    """Last block of ConvNeXt stage 0 / 1 as the network runs it in 16-bit storage: MLP + layer scale + shortcut, then the
    stage boundary's LayerNorm2d and the 2x2 space-to-depth of its stride-2 conv in the same kernel's epilogue (timm
    ConvNeXtStage.downsample; fused_mlp_res_kernel<T, true> at C = 96, xs_mlp_kernel<T, 192, true>).  Checked element-wise
    against plain torch: F.layer_norm of the fp32 MLP output, then the patch gather, per segment."""
    dtype = DTYPES[dt]
    M = sum(n * h * w for n, h, w in segs)
    x = q(rnd((M, C), 1, 1.5), dtype)
    w1 = q(rnd((4 * C, C), 2, 1 / math.sqrt(C)), dtype)
    w2 = q(rnd((C, 4 * C), 3, 1 / math.sqrt(4 * C)), dtype)
    b1, b2, gamma = rnd((4 * C,), 4, 0.1), rnd((C,), 5, 0.1), rnd((C,), 6, 0.5)
    res = q(rnd((M, C), 7), dtype)
    lw, lb = rnd((C,), 8, 0.5) + 1.0, rnd((C,), 9, 0.1)
    h = q(F.gelu(x @ w1.t() + b1), dtype)
    y = res + gamma * (h @ w2.t() + b2)            # the residual stream the kernel keeps in registers (fp32)
    yn = F.layer_norm(y, (C,), lw, lb, 1e-6)
    want, tok0, hw, wd, out0, t = [], [], [], [], [], 0
    for n, H, W in segs:
        tok0.append(t); hw.append(H * W); wd.append(W); out0.append(t // 4)
        v = yn[t:t + n * H * W].reshape(n, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5)
        want.append(v.reshape(n * (H // 2) * (W // 2), 4 * C))
        t += n * H * W
    want = torch.cat(want)
    out = torch.full((M // 4 + 3, 4 * C), 7.0, dtype=dtype, device=dev())
    arr = lambda v: (ctypes.c_int * 4)(*(v + [0] * (4 - len(v))))
    a_tok0, a_hw, a_wd, a_out0 = arr(tok0), arr(hw), arr(wd), arr(out0)
    resd = D(res, dtype)
    kutil.call("gcv_k_fused_mlp_lnp", _lib.dtype_code(dtype), C, ptr(D(x, dtype)), ptr(D(w1, dtype)), ptr(D(b1)),
               ptr(D(w2)), ptr(D(b2)), ptr(D(gamma)), ptr(resd), ptr(D(lw)), ptr(D(lb)), 1e-6, len(segs), a_tok0, a_hw, a_wd,
               a_out0, ptr(out), M)
    assert torch.equal(resd.cpu(), res.to(dtype)), "the residual stream must not be written"
    assert (out[M // 4:].float() == 7.0).all(), "rows behind the last patch row were written"
    # y ~ U(-1, 1) + small: LayerNorm output of magnitude ~3 (lw up to 1.5, |z| up to ~2.5) -> scale 4
    assert_close(out[:M // 4], want, tol(dtype, 4.0), "fused mlp + LN-patchify")


# ----------------------------------------------------------------------------- LDS-DMA pipelined GEMM
@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K,act", [
    (700, 1536, 384, 2),     # ConvNeXt stage-2 pw1 class (GELU)
    (513, 192, 32, 0),       # one K tile  (pipeline shorter than the ring)
    (256, 192, 64, 0),       # two K tiles
    (300, 384, 96, 2),       # three K tiles, M tail
    (384, 768, 128, 0),      # exactly ring depth
    (1000, 384, 3072, 0),    # long K
])
def test_gemm_lds_dma_pipeline_bias_act(dt, M, N, K, act):
    dtype = DTYPES[dt]
    A, W = q(rnd((M, K), 1), dtype), q(rnd((N, K), 2, 1 / math.sqrt(K)), dtype)
    bias = rnd((N,), 3, 0.1)
    C = torch.full((M, N), 7.0, dtype=dtype, device=dev())
    gemm(dtype, _lib.A_PLAIN, _lib.EPI_BIAS_ACT, D(A, dtype), D(W, dtype), C, M, N, K, lda=K, ldc=N, bias=D(bias), act=act)
    assert_close(C, act_ref(A @ W.t() + bias, act), tol(dtype, 2.0), "glds gemm")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(1000, 384, 1536), (257, 768, 3072), (640, 192, 768)])
def test_gemm_lds_dma_pipeline_residual(dt, M, N, K):
    dtype = DTYPES[dt]
    A, W = q(rnd((M, K), 1), dtype), q(rnd((N, K), 2, 1 / math.sqrt(K)), dtype)
    bias, gamma = rnd((N,), 3, 0.1), rnd((N,), 4, 0.5)
    X = q(rnd((M, N), 5), dtype)
    Xd = D(X, dtype).clone()
    gemm(dtype, _lib.A_PLAIN, _lib.EPI_RESID, D(A, dtype), D(W, dtype), Xd, M, N, K, lda=K, ldc=N, bias=D(bias),
         gamma=D(gamma), resid=Xd)
    assert_close(Xd, X + gamma * (A @ W.t() + bias), tol(dtype, 2.0), "glds gemm resid")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K,act", [(8229, 1536, 384, "gelu"), (12544, 3072, 768, "gelu"), (16400, 768, 128, "none")])
def test_gemm_lds_dma_many_tiles_bias_act(dt, M, N, K, act):
    """Several waves of tiles per CU (the B=128 shapes of stages 2/3); M has a ragged last tile."""
    dtype = DTYPES[dt]
    act = {"none": _lib.ACT_NONE, "gelu": _lib.ACT_GELU}[act]
    A, W = q(rnd((M, K), 1), dtype), q(rnd((N, K), 2, 1 / math.sqrt(K)), dtype)
    bias = rnd((N,), 3, 0.1)
    C = torch.full((M + 8, N), 7.0, dtype=dtype, device=dev())
    gemm(dtype, _lib.A_PLAIN, _lib.EPI_BIAS_ACT, D(A, dtype), D(W, dtype), C, M, N, K, lda=K, ldc=N, bias=D(bias), act=act)
    assert_close(C[:M], act_ref(A @ W.t() + bias, act), tol(dtype, 2.0), "glds gemm (many tiles)")
    assert (C[M:].float() == 7.0).all(), "rows past M must not be written"


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(32868, 384, 1536), (16500, 768, 3072)])
def test_gemm_lds_dma_many_tiles_residual(dt, M, N, K):
    dtype = DTYPES[dt]
    A, W = q(rnd((M, K), 1), dtype), q(rnd((N, K), 2, 1 / math.sqrt(K)), dtype)
    bias, gamma = rnd((N,), 3, 0.1), rnd((N,), 4, 0.5)
    X = q(rnd((M, N), 5), dtype)
    Xd = D(X, dtype).clone()
    gemm(dtype, _lib.A_PLAIN, _lib.EPI_RESID, D(A, dtype), D(W, dtype), Xd, M, N, K, lda=K, ldc=N, bias=D(bias),
         gamma=D(gamma), resid=Xd)
    assert_close(Xd, X + gamma * (A @ W.t() + bias), tol(dtype, 2.0), "glds gemm resid (many tiles)")


def test_preprocess_frame_on_device_matches_reference_semantics():
    """Row N1: uint8 NHWC -> normalised NCHW (model/pred_func.py:95-108, dataset/loader.py:63-65,77)."""
    from genconvit_amd import synth
    from genconvit_amd.model import pred_func
    from oracle import cpu_ref
    u8 = synth.make_uint8_frames(3)
    want = cpu_ref.preprocess_frame(u8.numpy())
    got = _lib.preprocess(u8.to(dev()))
    assert got.dtype == torch.float32 and got.shape == (3, 3, 224, 224)
    assert_close(got, want, 1e-6, "preprocess fp32")
    assert_close(_lib.preprocess(u8.to(dev()), torch.float16), want, 2e-3, "preprocess fp16")
    assert_close(pred_func.preprocess_frame(u8.numpy()), want, 1e-6, "preprocess_frame drop-in")
