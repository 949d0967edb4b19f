"""ctypes binding of ``libgenconvit_hip.so`` (C ABI: include/genconvit_hip.h).

The north-star asks for cffi; cffi is not installed in this image, ctypes (stdlib) binds the same
C ABI.  There is NO CPU fallback: if the library is missing, or there is no gfx950 device, the
calls raise.

The library is linked without a hard dependency on a particular HIP runtime (``-no-hip-rt``):
PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` and a second runtime instance in the same
process would not understand torch's streams.  So the runtime already used by the process is made
globally visible first (torch's bundled copy when torch is importable, /opt/rocm's otherwise).
"""
from __future__ import annotations

import ctypes
import ctypes.util
import json
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GCV_LIB_PATH") or os.path.join(_HERE, "lib", "libgenconvit_hip.so")   # override: diagnostic builds

GCV_F32, GCV_BF16, GCV_F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU, ACT_LEAKY = 0, 1, 2, 3
A_PLAIN, A_IM2COL3_POOL, A_IM2COL3_S2 = 0, 1, 2
EPI_BIAS_ACT, EPI_RESID, EPI_POOL4, EPI_CONVT, EPI_SPLITK = 0, 1, 2, 3, 4

c_void_p, c_int, c_int64, c_float, c_char_p, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                                         ctypes.c_float, ctypes.c_char_p, ctypes.c_size_t)


class TensorDesc(ctypes.Structure):
    _fields_ = [("name", c_char_p), ("data", c_void_p), ("numel", c_int64), ("on_device", c_int)]


class GemmArgs(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("Wt", c_void_p), ("C", c_void_p), ("bias", c_void_p), ("gamma", c_void_p),
                ("resid", c_void_p), ("partial", c_void_p)] + \
               [(n, c_int) for n in ("M", "N", "K", "lda", "ldc", "act", "splitk", "k_per_split", "H", "W",
                                     "cin_log2", "cout_log2")]


# every symbol include/genconvit_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "gcv_last_error": (c_char_p, []),
    "gcv_create": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int]),
    "gcv_destroy": (None, [c_void_p]),
    "gcv_workspace_bytes": (c_size_t, [c_void_p]),
    "gcv_load_ed": (c_int, [c_void_p, ctypes.POINTER(TensorDesc), c_int]),
    "gcv_load_vae": (c_int, [c_void_p, ctypes.POINTER(TensorDesc), c_int]),
    "gcv_load_swin": (c_int, [c_void_p, ctypes.POINTER(TensorDesc), c_int, c_char_p]),
    "gcv_ed_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gcv_vae_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gcv_genconvit_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gcv_comm_available": (c_int, []),
    "gcv_comm_count": (c_int, [c_void_p]),
    "gcv_comm_unique_id": (c_int, [c_void_p]),
    "gcv_comm_create": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_void_p, c_int]),
    "gcv_comm_destroy": (None, [c_void_p]),
    "gcv_allgather_logits": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gcv_convnext_forward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "gcv_swin_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gcv_vote": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "gcv_preprocess": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gcv_face_crop_resize": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "gcv_vote_segments": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "gcv_profile_enable": (c_int, [c_void_p, c_int]),
    "gcv_profile_report": (c_char_p, [c_void_p]),
    "gcv_k_gemm": (c_int, [c_int, c_int, c_int, ctypes.POINTER(GemmArgs), c_void_p]),
    "gcv_k_stem_ln": (c_int, [c_int, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "gcv_k_dwconv7_ln": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                 c_int, c_int, c_float, c_void_p]),
    "gcv_k_ln_patchify": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                  c_void_p]),
    "gcv_k_layernorm_rows": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "gcv_k_pool_ln": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "gcv_k_conv3_first": (c_int, [c_int, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gcv_k_convt2_small": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "gcv_k_reparam": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gcv_k_head_tail": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gcv_k_head_tail_splitk": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                       c_void_p]),
    "gcv_k_resize_mse": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gcv_k_swin_window_attn": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_void_p]),
    "gcv_k_patch_merge_ln": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                     c_void_p]),
    "gcv_k_mean_tokens": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gcv_k_fused_mlp": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_int, c_void_p]),
    "gcv_k_fused_mlp_lnp": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_void_p]),
    "gcv_k_fused_mlp_timed": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
}

_lib = None
_lock = threading.Lock()


class GenConViTHipError(RuntimeError):
    pass


def _preload_hip_runtime():
    cands = []
    try:
        import torch
        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:   # torch not importable: plain C-ABI use
        pass
    cands += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    last = None
    for c in cands:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            return ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
        except OSError as e:   # try the next candidate
            last = e
    raise GenConViTHipError(f"cannot load a HIP runtime (libamdhip64.so): {last}")


def load():
    """Load (once) and return the ctypes library with all signatures set.  Raises if absent."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise GenConViTHipError(
                f"{LIB_PATH} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C genconvit_amd/csrc`). There is no CPU fallback.")
        _preload_hip_runtime()
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)    # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def last_error() -> str:
    return (load().gcv_last_error() or b"").decode(errors="replace")


def check(rc: int, what: str):
    if rc != 0:
        raise GenConViTHipError(f"{what} failed (rc={rc}): {last_error()}")


def dtype_code(torch_dtype) -> int:
    import torch
    try:
        return {torch.float32: GCV_F32, torch.bfloat16: GCV_BF16, torch.float16: GCV_F16}[torch_dtype]
    except KeyError:
        raise GenConViTHipError(f"unsupported dtype {torch_dtype}; use float32, bfloat16 or float16") from None


def current_stream_ptr(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream


class Handle:
    """Owns one ``gcv_handle`` (packed weights + workspace) on one device / dtype."""

    def __init__(self, device_index: int, torch_dtype, max_batch: int):
        import torch
        self.lib = load()
        if not torch.cuda.is_available():
            raise GenConViTHipError("no HIP device visible: the GenConViT HIP path needs an MI355X (gfx950); "
                                    "there is no CPU fallback")
        self.device_index = int(device_index)
        self.dtype = torch_dtype
        self.max_batch = int(max_batch)
        self._h = c_void_p()
        check(self.lib.gcv_create(ctypes.byref(self._h), self.device_index, dtype_code(torch_dtype), self.max_batch),
              "gcv_create")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.gcv_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown
            pass

    # -- weights ---------------------------------------------------------------------------
    @staticmethod
    def _descs(state_dict, skip_prefixes=()):
        import torch
        keep = []
        names = []
        for k, v in state_dict.items():
            if not torch.is_tensor(v) or not v.is_floating_point() or any(k.startswith(p) or p in k for p in skip_prefixes):
                continue
            t = v.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            keep.append(t)
            names.append(k.encode())
        arr = (TensorDesc * len(keep))()
        for i, (n, t) in enumerate(zip(names, keep)):
            arr[i] = TensorDesc(n, t.data_ptr(), t.numel(), 1 if t.is_cuda else 0)
        return arr, keep, names

    _OFF_PATH = ("embedder.", "patch_embed.", "encoder.fc1.", "encoder.fc2.", "fc3.", "num_batches_tracked")

    def _settle(self):
        """The library reads the tensors with synchronous copies / null-stream kernels: whatever stream produced them
        (a non-blocking ``.to()``, a cast on a side stream) must have finished first."""
        import torch
        torch.cuda.synchronize(self.device_index)

    def load_ed(self, state_dict):
        self._settle()
        arr, keep, _ = self._descs(state_dict, self._OFF_PATH)
        check(self.lib.gcv_load_ed(self._h, arr, len(keep)), "gcv_load_ed")

    def load_vae(self, state_dict, with_var=True):
        """``with_var=False`` leaves ``encoder.var`` (1.26 GB fp32, only read for the optional KL output) unpacked."""
        self._settle()
        arr, keep, _ = self._descs(state_dict, self._OFF_PATH + (() if with_var else ("encoder.var.",)))
        check(self.lib.gcv_load_vae(self._h, arr, len(keep)), "gcv_load_vae")

    def load_swin(self, state_dict, prefix=""):
        self._settle()
        arr, keep, _ = self._descs(state_dict)
        check(self.lib.gcv_load_swin(self._h, arr, len(keep), prefix.encode()), "gcv_load_swin")

    # -- forwards --------------------------------------------------------------------------
    def _check_x(self, x, res=224):
        import torch
        if not (torch.is_tensor(x) and x.is_cuda and x.device.index == self.device_index):
            raise GenConViTHipError(f"input must be a CUDA(HIP) tensor on device {self.device_index}")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != res or x.shape[3] != res:
            raise GenConViTHipError(f"input must be (B,3,{res},{res}), got {tuple(x.shape)}")
        if x.dtype != self.dtype:
            raise GenConViTHipError(f"input dtype {x.dtype} != handle dtype {self.dtype}")
        if x.shape[0] < 1 or x.shape[0] > self.max_batch:
            raise GenConViTHipError(f"batch {x.shape[0]} outside [1,{self.max_batch}]")
        return x.contiguous()

    def ed_forward(self, x):
        import torch
        x = self._check_x(x)
        out = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device)
        check(self.lib.gcv_ed_forward(self._h, x.data_ptr(), x.shape[0], out.data_ptr(), current_stream_ptr(x.device)),
              "gcv_ed_forward")
        return out

    def vae_forward(self, x, eps, want_recon=True, want_mse=False, want_kl=False):
        import torch
        x = self._check_x(x)
        B = x.shape[0]
        if not (torch.is_tensor(eps) and eps.is_cuda and eps.device.index == self.device_index and tuple(eps.shape) == (B, 12544)):
            raise GenConViTHipError(f"eps must be a tensor of shape ({B},12544) on device {self.device_index}")
        eps = eps.float().contiguous()
        out = torch.empty((B, 2), dtype=torch.float32, device=x.device)
        recon = torch.empty((B, 3, 224, 224), dtype=self.dtype, device=x.device) if want_recon else None
        mse = torch.empty((B,), dtype=torch.float32, device=x.device) if want_mse else None
        kl = torch.empty((1,), dtype=torch.float32, device=x.device) if want_kl else None
        p = lambda t: t.data_ptr() if t is not None else None
        check(self.lib.gcv_vae_forward(self._h, x.data_ptr(), eps.data_ptr(), B, out.data_ptr(), p(recon), p(mse), p(kl),
                                       current_stream_ptr(x.device)), "gcv_vae_forward")
        return out, recon, mse, kl

    def convnext_forward(self, which, x):
        import torch
        x = self._check_x(x, res=x.shape[-1])
        out = torch.empty((x.shape[0], 1000), dtype=self.dtype, device=x.device)
        check(self.lib.gcv_convnext_forward(self._h, which, x.data_ptr(), x.shape[0], x.shape[-1], out.data_ptr(),
                                            current_stream_ptr(x.device)), "gcv_convnext_forward")
        return out

    def swin_forward(self, x):
        import torch
        x = self._check_x(x)
        out = torch.empty((x.shape[0], 1000), dtype=self.dtype, device=x.device)
        check(self.lib.gcv_swin_forward(self._h, x.data_ptr(), x.shape[0], out.data_ptr(), current_stream_ptr(x.device)),
              "gcv_swin_forward")
        return out

    def workspace_bytes(self) -> int:
        return int(self.lib.gcv_workspace_bytes(self._h))

    def profile_enable(self, on=True):
        check(self.lib.gcv_profile_enable(self._h, 1 if on else 0), "gcv_profile_enable")

    def profile_report(self):
        return json.loads((self.lib.gcv_profile_report(self._h) or b"[]").decode())


def genconvit_forward(h_ed: "Handle", h_vae: "Handle", x, eps):
    """``GenConViT.forward`` for net='genconvit' (model/genconvit.py:66-75) through ``gcv_genconvit_forward``: ED and VAE
    on two streams inside the library, joined back into the current stream; returns the (2B,2) fp32 logits."""
    import torch
    x = h_ed._check_x(x)
    B = x.shape[0]
    if h_vae.device_index != h_ed.device_index or h_vae.dtype != h_ed.dtype:
        raise GenConViTHipError("ED and VAE handles must share device and dtype")
    if not (torch.is_tensor(eps) and eps.is_cuda and eps.device.index == h_ed.device_index and tuple(eps.shape) == (B, 12544)):
        raise GenConViTHipError(f"eps must be a tensor of shape ({B},12544) on device {h_ed.device_index}")
    eps = eps.float().contiguous()
    out = torch.empty((2 * B, 2), dtype=torch.float32, device=x.device)
    check(h_ed.lib.gcv_genconvit_forward(h_ed._h, h_vae._h, x.data_ptr(), eps.data_ptr(), B, out.data_ptr(),
                                         current_stream_ptr(x.device)), "gcv_genconvit_forward")
    return out


class Comm:
    """RCCL communicator of the C ABI (``gcv_comm_*``): one per process group, used for the logit all-gather."""

    @staticmethod
    def _share_torch_rccl():
        """Every rank must bind the SAME RCCL build (the one torch.distributed already uses in this process): point the
        library's run-time loader at torch's bundled copy unless the caller chose one."""
        try:
            import torch
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            if os.path.exists(path):
                os.environ.setdefault("GCV_RCCL_PATH", path)
        except Exception:
            pass

    def __init__(self, world: int, rank: int, unique_id: bytes, device_index: int):
        self._share_torch_rccl()
        self.lib = load()
        self.world, self.rank, self.device_index = int(world), int(rank), int(device_index)
        self._c = c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        check(self.lib.gcv_comm_create(ctypes.byref(self._c), self.world, self.rank, buf, self.device_index), "gcv_comm_create")

    @staticmethod
    def available() -> bool:
        """RCCL and its entry points can be bound in this process (dlopen + dlsym; starts nothing)."""
        Comm._share_torch_rccl()
        return bool(load().gcv_comm_available())

    def count(self) -> int:
        """Ranks in the communicator as RCCL itself reports them (ncclCommCount)."""
        return int(self.lib.gcv_comm_count(self._c))

    @staticmethod
    def unique_id() -> bytes:
        Comm._share_torch_rccl()
        buf = ctypes.create_string_buffer(128)
        check(load().gcv_comm_unique_id(buf), "gcv_comm_unique_id")
        return buf.raw

    def allgather(self, local):
        """local: fp32 device tensor (same numel on every rank) -> (world, *local.shape)."""
        import torch
        local = local.float().contiguous()
        out = torch.empty((self.world,) + tuple(local.shape), dtype=torch.float32, device=local.device)
        check(self.lib.gcv_allgather_logits(self._c, local.data_ptr(), local.numel(), out.data_ptr(),
                                            current_stream_ptr(local.device)), "gcv_allgather_logits")
        return out

    def close(self):
        if getattr(self, "_c", None) is not None and self._c:
            self.lib.gcv_comm_destroy(self._c)
            self._c = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def preprocess(frames_u8, dtype=None):
    """Device-side ``preprocess_frame`` (model/pred_func.py:95-108): uint8 (N,H,W,3) device tensor ->
    normalised (N,3,H,W) tensor of ``dtype`` (fp32 by default, like the reference)."""
    import torch
    lib = load()
    dtype = dtype or torch.float32
    if not (frames_u8.is_cuda and frames_u8.dtype == torch.uint8 and frames_u8.dim() == 4 and frames_u8.shape[3] == 3):
        raise GenConViTHipError("preprocess expects a uint8 device tensor of shape (N,H,W,3)")
    frames_u8 = frames_u8.contiguous()
    n, h, w, _ = frames_u8.shape
    out = torch.empty((n, 3, h, w), dtype=dtype, device=frames_u8.device)
    check(lib.gcv_preprocess(dtype_code(dtype), frames_u8.data_ptr(), out.data_ptr(), n, h, w,
                             current_stream_ptr(frames_u8.device)), "gcv_preprocess")
    return out


def face_crop_resize(frames_u8, boxes, size=224):
    """Row N4: ``cv2.resize(frame[top:bottom, left:right], (size, size), interpolation=cv2.INTER_AREA)`` of face_rec
    (model/pred_func.py:79-85) for all boxes in one launch.  ``frames_u8``: (F,H,W,3) uint8 device tensor (RGB);
    ``boxes``: (n,5) integers (frame index, top, right, bottom, left).  Returns (n,size,size,3) uint8 on the device.
    Boxes outside their frame are an error here (the kernel itself would write zeros for them)."""
    import torch
    lib = load()
    if not (frames_u8.is_cuda and frames_u8.dtype == torch.uint8 and frames_u8.dim() == 4 and frames_u8.shape[3] == 3):
        raise GenConViTHipError("face_crop_resize expects a uint8 device tensor of shape (F,H,W,3)")
    frames_u8 = frames_u8.contiguous()
    nf, h, w, _ = frames_u8.shape
    b = torch.as_tensor(boxes, dtype=torch.int32).reshape(-1, 5).cpu()
    if b.numel():
        f, top, right, bottom, left = b.unbind(1)
        ok = (f >= 0) & (f < nf) & (top >= 0) & (left >= 0) & (bottom <= h) & (right <= w) & (top < bottom) & (left < right)
        if not bool(ok.all()):
            raise GenConViTHipError(f"face_crop_resize: box {int((~ok).nonzero()[0])} lies outside its {h}x{w} frame")
    out = torch.empty((b.shape[0], size, size, 3), dtype=torch.uint8, device=frames_u8.device)
    if b.shape[0] == 0:
        return out
    bd = b.to(frames_u8.device)
    check(lib.gcv_face_crop_resize(frames_u8.data_ptr(), nf, h, w, bd.data_ptr(), b.shape[0], out.data_ptr(), size,
                                   current_stream_ptr(frames_u8.device)), "gcv_face_crop_resize")
    return out


def vote_segments(logits, batch, nets, offsets):
    """Per-video ``mean(sigmoid(logits), dim=0)`` for several videos batched in one forward (row N3):
    ``offsets`` = int32 device tensor of n_videos+1 frame offsets; returns (n_videos, 2) fp32."""
    import torch
    lib = load()
    logits = logits.float().contiguous()
    offsets = offsets.to(device=logits.device, dtype=torch.int32).contiguous()
    nvid = offsets.numel() - 1
    out = torch.empty((nvid, 2), dtype=torch.float32, device=logits.device)
    check(lib.gcv_vote_segments(logits.data_ptr(), int(batch), int(nets), offsets.data_ptr(), nvid, out.data_ptr(),
                                current_stream_ptr(logits.device)), "gcv_vote_segments")
    return out


def vote(logits):
    """Device-side ``mean(sigmoid(logits), dim=0)`` (model/pred_func.py:120,125) -> (2,) fp32 tensor."""
    import torch
    lib = load()
    logits = logits.float().contiguous().reshape(-1, 2)
    out = torch.empty((2,), dtype=torch.float32, device=logits.device)
    check(lib.gcv_vote(logits.data_ptr(), logits.shape[0], out.data_ptr(), current_stream_ptr(logits.device)), "gcv_vote")
    return out
