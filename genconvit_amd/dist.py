"""Frame-sharded multi-GPU inference — new capability, the reference is single device.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU
tests).  Frames are independent (no cross-frame op on the path; BatchNorm is in eval mode), so each
rank runs the forward on a contiguous shard with replicated weights and the only exchange is one
all-gather of the per-frame logits (B_local x nets x 2 fp32 — 2 KiB per rank at 128 frames/GPU,
latency-bound).  Rows are then re-ordered to the reference's ``cat((ed, vae), dim=0)`` layout
(model/genconvit.py:74) so the vote (model/pred_func.py:120-131) sees exactly the unsharded tensor.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

_COMMS = {}       # process group -> genconvit_amd._lib.Comm (RCCL through the C ABI)


def _c_comm(group, device):
    """The C-ABI RCCL communicator of ``group`` (created on first use: rank 0's ncclUniqueId travels over the
    torch.distributed group).  None when the group is not an RCCL ("nccl") group of device tensors, or when
    GCV_DIST_TORCH=1 asks for torch.distributed's own all_gather."""
    if device.type != "cuda" or os.environ.get("GCV_DIST_TORCH") == "1" or dist.get_backend(group) != "nccl":
        return None
    key = id(group) if group is not None else 0
    if key not in _COMMS:
        from . import _lib
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        # ncclCommInitRank blocks until every rank has joined: first make sure, over the torch group, that RCCL could be
        # bound on ALL ranks (gcv_comm_available: dlopen + symbol lookup, nothing is started); otherwise every rank keeps
        # torch's all_gather.  Only rank 0 then draws the ncclUniqueId (each ncclGetUniqueId starts a bootstrap listener).
        try:
            ok = 1 if _lib.Comm.available() else 0
            why = _lib.last_error() if not ok else ""
        except Exception as e:      # library missing on this rank
            ok, why = 0, str(e)
        if not ok:
            print(f"[genconvit_amd.dist] rank {rank}: RCCL not bound through the C ABI ({why}); using torch.distributed", flush=True)
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            _COMMS[key] = None
        else:
            box = [_lib.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            # ncclCommInitRank through the C ABI; if it fails on ANY rank (agreed over the torch group again), every rank
            # drops its communicator and the gather stays on torch.distributed's own RCCL communicator
            try:
                c = _lib.Comm(world, rank, box[0], device.index if device.index is not None else torch.cuda.current_device())
                ok, why = 1, ""
            except Exception as e:
                c, ok, why = None, 0, str(e)
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 0:
                if not ok:
                    print(f"[genconvit_amd.dist] rank {rank}: gcv_comm_create failed ({why}); using torch.distributed", flush=True)
                if c is not None:
                    c.close()
                c = None
            _COMMS[key] = c
    return _COMMS[key]


def comm_info(group=None):
    """How the logit all-gather of ``group`` travels: {"path": "c_abi" | "torch", "ranks": ranks RCCL / the group reports}."""
    key = id(group) if group is not None else 0
    c = _COMMS.get(key)
    if c is not None:
        return {"path": "c_abi(gcv_allgather_logits -> ncclAllGather)", "ranks": c.count()}
    return {"path": "torch.distributed.all_gather", "ranks": dist.get_world_size(group) if dist.is_initialized() else 1}


def close_comms():
    for c in _COMMS.values():
        if c is not None:
            c.close()
    _COMMS.clear()


def shard_bounds(n_frames: int, world_size: int, rank: int):
    """Contiguous, balanced shard [lo, hi) of ``n_frames`` for ``rank`` (ragged tails allowed)."""
    base, rem = divmod(n_frames, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_logits(local_logits: torch.Tensor, n_frames: int, nets: int, group=None) -> torch.Tensor:
    """All-gather per-rank logits of shape (nets*B_local, 2) (rows ``[ed rows; vae rows]`` of the
    local shard) into the unsharded (nets*n_frames, 2) tensor in the reference's row order."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(n_frames, world, r) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    lo, hi = sizes[rank]
    bl = hi - lo
    assert local_logits.shape == (nets * bl, 2), (local_logits.shape, nets, bl)
    # fixed-size slab per rank (all_gather needs equal shapes): (nets, bmax, 2)
    slab = torch.zeros((nets, bmax, 2), dtype=torch.float32, device=local_logits.device)
    if bl:
        slab[:, :bl] = local_logits.float().reshape(nets, bl, 2)
    comm = _c_comm(group, slab.device)
    if comm is not None:                      # gcv_allgather_logits: RCCL through the C ABI, on the current stream
        out = comm.allgather(slab)
    else:                                     # gloo (CPU tests) or GCV_DIST_TORCH=1
        out = [torch.empty_like(slab) for _ in range(world)]
        dist.all_gather(out, slab, group=group)
    full = torch.empty((nets, n_frames, 2), dtype=torch.float32, device=local_logits.device)
    for r, (rlo, rhi) in enumerate(sizes):
        if rhi > rlo:
            full[:, rlo:rhi] = out[r][:, :rhi - rlo]
    return full.reshape(nets * n_frames, 2)


def sharded_forward(model_fn, frames: torch.Tensor, eps, nets: int, group=None) -> torch.Tensor:
    """Run ``model_fn(frames_shard, eps_shard) -> (nets*B_local, 2)`` on this rank's shard and return
    the unsharded logits on every rank.  ``frames``/``eps`` are the full batch (or already this
    rank's shard when ``frames.shape[0]`` equals the shard size is NOT assumed: pass full tensors)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = frames.shape[0]
    lo, hi = shard_bounds(n, world, rank)
    if hi > lo:
        local = model_fn(frames[lo:hi], None if eps is None else eps[lo:hi])
    else:
        local = torch.empty((0, 2), dtype=torch.float32, device=frames.device)
    return gather_logits(local, n, nets, group)
