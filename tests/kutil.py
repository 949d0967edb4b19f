"""Test helpers: call the per-kernel C-ABI entry points (gcv_k_*) with torch device tensors."""
import ctypes

import torch

from genconvit_amd import _lib

DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def dev():
    return torch.device("cuda", 0)


def stream():
    return _lib.current_stream_ptr(dev())


def ptr(t):
    return None if t is None else t.data_ptr()


def gemm(dtype, a_mode, epi, A, Wt, C, M, N, K, lda=0, ldc=0, bias=None, gamma=None, resid=None, partial=None,
         act=0, splitk=1, k_per_split=0, H=0, W=0, cin_log2=0, cout_log2=0):
    lib = _lib.load()
    g = _lib.GemmArgs(ptr(A), ptr(Wt), ptr(C), ptr(bias), ptr(gamma), ptr(resid), ptr(partial), M, N, K, lda, ldc,
                      act, splitk, k_per_split, H, W, cin_log2, cout_log2)
    _lib.check(lib.gcv_k_gemm(_lib.dtype_code(dtype), a_mode, epi, ctypes.byref(g), stream()), "gcv_k_gemm")
    torch.cuda.synchronize()


def call(name, *args):
    lib = _lib.load()
    _lib.check(getattr(lib, name)(*args, stream()), name)
    torch.cuda.synchronize()


def tol(dtype, scale=1.0):
    """(rtol-free) absolute tolerance for outputs of magnitude ~scale."""
    return {torch.float32: 2e-5, torch.bfloat16: 3e-2, torch.float16: 4e-3}[dtype] * scale


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale
