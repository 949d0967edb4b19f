#!/usr/bin/env python3
"""bench.py — frames/sec of the GenConViT (ed+vae) forward on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic frames already resident in HBM:
    genconvit forward (ED + VAE, model/genconvit.py:66-75) on this rank's 128-frame shard
    -> (N>1) RCCL all-gather of the per-frame logits -> sigmoid/mean vote on device.
Workload at N=1 = BASELINE.json configs[3] ("genconvit, batch=128, 1xMI355X, fp16"), at N>1 =
configs[4] (128 frames per GPU, weak scaling).  Weights are random-init of the reference
architecture from the in-repo deterministic generator; frames are synthetic (no datasets here).

Rank 0 prints ONE JSON line: metric/value (whole-job frames/s), ms_per_step, plus
  "roofline"     — the dominant kernel family of the step, timed live with HIP events on the
                   launch stream (gcv_profile_*), algorithmic FLOPs/bytes from the launch shapes
  "cpu_baseline" — the CPU oracle (a port of the reference's PyTorch CPU path; the reference itself
                   cannot run without timm) timed on this host's cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from genconvit_amd import _lib, dist as gdist, spec, synth          # noqa: E402
from genconvit_amd.model.config import load_config                  # noqa: E402
from genconvit_amd.model.genconvit import GenConViT                 # noqa: E402
from genconvit_amd.model.genconvit_ed import GenConViTED            # noqa: E402
from genconvit_amd.model.genconvit_vae import GenConViTVAE          # noqa: E402

# What the matrix pipe sustains with DATA in the operands, measured on this pool (profiles/micro/mfma_shape_clock.hip: the
# x-stationary MLP inner loop, LDS fragment reads + v_mfma_f32_32x32x16 only, random fp16 operands: 1.38 PFLOP/s at an in-kernel
# clock of 1.47 GHz; 2.08 PFLOP/s at 2.36 GHz on zeros).  Reported beside `peak` for 16-bit runs, never instead of it.
SUSTAINED_MFMA_RANDOM = {"value": 1382.0, "unit": "TFLOP/s", "in_kernel_clock_ghz": 1.47,
                         "source": "profiles/r03_micro/mfma_shape_clock.txt"}
PEAK = {  # /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks, HBM3E spec
    "mfma": {"f32": 157.3, "f16": 2500.0, "bf16": 2500.0},   # TFLOP/s
    "hbm": 8000.0,                                             # GB/s
}
DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
GFLOP_PER_FRAME = {"ed": 18.39, "vae": 11.78, "genconvit": 30.17}   # BASELINE.md §2 (algorithmic)


def host_cores() -> int:
    """CPU cores this process may actually use: cgroup quota if set (the GPU box exposes 256 logical
    CPUs but grants a 16-core share), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return min(n, int(os.environ.get("GCV_HOST_CORES", "32")))


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def measured_traffic(family: str, net: str, batch: int, dtype: str, launches_per_step: int):
    """HBM bytes per launch of a kernel family from a committed rocprofv3 PMC run of THIS configuration
    (profiles/r*_traffic*.json, written by profiles/pmc_bench_traffic.sh: FETCH_SIZE x2 + WRITE_SIZE in separate
    passes, with the net / batch / dtype it was recorded on).  PMC counters cannot be collected from inside the
    process, so this is the newest committed measurement of the same workload — or (None, None): a number from
    another workload is not this run's traffic."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), reverse=True):
        try:
            d = json.load(open(f))
            c = d.get("config", {})
            if (c.get("net"), c.get("batch"), c.get("dtype")) != (net, batch, dtype):
                continue
            fam = d["families"][family]
            # the PMC file must have been recorded on the same launch structure: its dispatch count is steps x this
            # run's launches per step, or it describes other kernels (a stale file after the kernels changed, or a
            # family classification that swept in other GEMMs) and is refused
            if fam.get("dispatches") != fam.get("steps", 0) * launches_per_step:
                continue
            return round(fam["hbm_bytes_per_launch"]), os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--net", default="genconvit", choices=["ed", "vae", "genconvit"])
    ap.add_argument("--batch", type=int, default=128, help="frames per GPU")
    ap.add_argument("--dtype", default="f16", choices=["f32", "f16", "bf16"])
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-concurrent", action="store_true", help="run ED and VAE back to back on one stream (A/B switch)")
    ap.add_argument("--swin", action="store_true",
                    help="also time the Swin-T embedder (row A6: constructed but never executed by the reference forward)")
    return ap.parse_args()


def build_models(net, dtype, batch, device):
    cfg = load_config()
    ed = vae = None
    sds = {}
    if net != "vae":
        sds["ed"] = synth.make_state_dict(spec.ed_spec(), synth.DEFAULT_SEED, "ed/", device=device)
        ed = GenConViTED(cfg, init="empty")
        ed.load_state_dict(sds["ed"])
        ed = ed.to(device).to(dtype).eval().reserve(batch)
    if net != "ed":
        sds["vae"] = synth.make_state_dict(spec.vae_spec(include_unused=False), synth.DEFAULT_SEED, "vae/", device=device)
        vae = GenConViTVAE(cfg, init="empty")
        vae.load_state_dict(sds["vae"], strict=False)
        vae = vae.to(device).to(dtype).eval().reserve(batch)
    return GenConViT.from_modules(ed, vae, net=net), sds


def cpu_baseline(net, sds, seconds):
    """Oracle (CPU port of the reference's PyTorch path, as written: 3x mu + var GEMMs) on the host
    cores, reference config[0] shape (batch 4).  Bounded to ~`seconds` of CPU work."""
    from oracle import cpu_ref
    torch.set_num_threads(host_cores())
    cpu = {k: {n: t.float().cpu() for n, t in sd.items()} for k, sd in sds.items()}
    if "vae" in cpu and "encoder.var.weight" not in cpu["vae"]:
        cpu["vae"]["encoder.var.weight"] = cpu["vae"]["encoder.mu.weight"]
        cpu["vae"]["encoder.var.bias"] = cpu["vae"]["encoder.mu.bias"]
    B = 4
    x = synth.make_frames(B, name="cpu_baseline")
    eps = synth.make_eps(B)
    def timed(as_written, budget):
        fwd = lambda: cpu_ref.genconvit_forward(cpu.get("ed"), cpu.get("vae"), x, eps, net=net, as_written=as_written)
        with torch.no_grad():
            fwd()                                    # warm-up
            times = []
            t_end = time.perf_counter() + budget
            while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 50):
                t0 = time.perf_counter()
                fwd()
                times.append(time.perf_counter() - t0)
        times.sort()
        return times[len(times) // 2], len(times)
    med, n = timed(True, seconds * 0.6)
    med_d, n_d = timed(False, seconds * 0.4)     # the deduplicated number beside the as-written one
    return {"value": round(B / med, 3), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "value_deduplicated": round(B / med_d, 3),
            "sample": f"{net} forward, batch {B}, fp32, oracle/cpu_ref.py; `value` as the reference writes it (3x mu + var "
                      f"GEMMs, model/genconvit_vae.py:45-56), median of {n} iterations after 1 warm-up; "
                      f"`value_deduplicated` with mu computed once and no var GEMM (what the HIP path computes), "
                      f"median of {n_d}"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("for --gpus N>1 launch with torch.distributed.run --nproc-per-node N")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the HIP path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # GCV_BENCH_FORCE_DIST=1 takes the RCCL path (init, gather, barrier, max-reduce) even with one rank: a way to
    # rehearse the multi-GPU code on a one-GPU box
    dist_on = world > 1 or os.environ.get("GCV_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        torch.distributed.init_process_group("nccl", device_id=device)
    torch.set_grad_enabled(False)
    torch.set_num_threads(host_cores())
    log(f"host cores usable: {host_cores()} (os.cpu_count() = {os.cpu_count()})")

    dtype = DT[a.dtype]
    GenConViT.concurrent = not a.no_concurrent
    if a.no_concurrent:
        os.environ["GCV_VAE_SPLIT"] = "0"      # one stream for everything (also the library's default since round 3)
    nets = 2 if a.net == "genconvit" else 1
    model, sds = build_models(a.net, dtype, a.batch, device)
    log("models built (synthetic weights generated on device)")
    # every rank holds its own 128-frame shard of the global batch (weak scaling)
    n_global = a.batch * world
    lo, hi = gdist.shard_bounds(n_global, world, rank)
    x = synth.make_frames(hi - lo, name=f"bench_frames_r{rank}").to(device).to(dtype)
    eps = synth.make_eps(hi - lo, name=f"bench_eps_r{rank}").to(device)

    def step():
        logits = model(x, eps=eps)
        if dist_on:
            logits = gdist.gather_logits(logits, n_global, nets)
        return _lib.vote(logits)

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    log("warm-up done")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out).all()
    log(f"timed region: {a.steps} steps in {dt:.3f} s")
    # diagnostic, NOT `value`: the same step with a device synchronisation behind every step — what a caller that reads each
    # result before submitting the next batch (pred_vid) sees.  The timed region above lets the host run ahead, so every
    # step's launches are queued before the GPU reaches them; here the enqueue time of a step is exposed.
    t1 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
        torch.cuda.synchronize()
    synced_ms = (time.perf_counter() - t1) / a.steps * 1e3
    log(f"synchronised after every step: {synced_ms:.3f} ms per step")

    # ---- roofline of the dominant kernel family: HIP events around every launch (separate steps so
    #      the event records do not perturb the timed region) --------------------------------------
    roof = None
    roof_families = []
    if rank == 0:
        handles = [m._handle for m in (getattr(model, "model_ed", None), getattr(model, "model_vae", None)) if m is not None]
        for h in handles:
            h.profile_enable(True)
        GenConViT.concurrent = False     # per-kernel durations are taken with the two networks back to back:
        agg = {}                         # overlapped launches would time each kernel while it shares the GPU
        for ps in range(max(a.profile_steps, 1) + 1):
            model(x, eps=eps)            # forward only: rank 0 is alone here, no collective may be issued
            torch.cuda.synchronize()
            for h in handles:
                recs = h.profile_report()
                if ps == 0:              # untimed: the first profiled step creates the event pool and switches schedule
                    continue
                for r in recs:
                    g = agg.setdefault(r["tag"], {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
                    for k in ("launches", "ms", "flops", "bytes"):
                        g[k] += r[k]
        for h in handles:
            h.profile_enable(False)
        GenConViT.concurrent = not a.no_concurrent
        total_ms = sum(v["ms"] for v in agg.values())
        # kernel families: every pointwise-MLP contraction of the ConvNeXt blocks runs on the MFMA GEMM
        # templates (gemm_glds_kernel / gemm_kernel with plain A, and the fused two-GEMM MLP kernel)
        MFMA_TAGS = ("cnx.pw1_gelu", "cnx.pw2_scale_res", "cnx.fused_mlp")
        fam = {}
        for tag, v in agg.items():
            f = "mfma_gemm(cnx.pw1_gelu+cnx.pw2_scale_res+cnx.fused_mlp)" if tag in MFMA_TAGS else tag
            g = fam.setdefault(f, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            for k in g:
                g[k] += v[k]
        def entry(name, d, bound, tkey):
            avg_ms = d["ms"] / d["launches"]
            if bound == "mfma":
                achieved = d["flops"] / d["launches"] / (avg_ms * 1e-3) / 1e12
                peak, unit = PEAK["mfma"][a.dtype], "TFLOP/s"
            else:
                achieved = d["bytes"] / d["launches"] / (avg_ms * 1e-3) / 1e9
                peak, unit = PEAK["hbm"], "GB/s"
            lps = d["launches"] // max(a.profile_steps, 1)
            traffic, traffic_src = measured_traffic(tkey, a.net, a.batch, a.dtype, lps) if tkey else (None, None)
            e = {"kernel": name, "bound": bound, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
                 "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)",
                 "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
                 "launches_per_step": d["launches"] // max(a.profile_steps, 1), "avg_launch_ms": round(avg_ms, 4),
                 "share_of_step": round(d["ms"] / total_ms, 3)}
            if bound == "mfma" and a.dtype in ("f16", "bf16"):
                e["peak_sustained_random_operands"] = dict(SUSTAINED_MFMA_RANDOM,
                                                           frac=round(achieved / SUSTAINED_MFMA_RANDOM["value"], 4))
            return e
        mfma_name = "mfma_gemm(cnx.pw1_gelu+cnx.pw2_scale_res+cnx.fused_mlp)"
        name, d = max(fam.items(), key=lambda kv: kv[1]["ms"])
        bound = "mfma" if name.startswith("mfma_gemm") or "gemm" in name else "hbm"
        roof = entry(name, d, bound, "mfma_gemm" if name == mfma_name else ("dwconv7_ln" if "dwconv" in name else None))
        roof["breakdown_ms_per_step"] = {k: round(v["ms"] / max(a.profile_steps, 1), 3)
                                         for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:16]}
        # north_star: "achieved HBM GB/s for the depthwise/window-attention kernels and MFMA utilisation for the
        # pointwise GEMMs": one entry per kernel family of the step (the dominant one is `roofline` above)
        families = []
        if mfma_name in fam:
            families.append(entry(mfma_name, fam[mfma_name], "mfma", "mfma_gemm"))
        if "cnx.dwconv7_ln" in fam:
            families.append(entry("cnx.dwconv7_ln", fam["cnx.dwconv7_ln"], "hbm", "dwconv7_ln"))
        if "vae.mu_gemm_splitk" in fam:          # weight-streaming GEMM: 25088 x 12544 weights read once per batch
            families.append(entry("vae.mu_gemm_splitk", fam["vae.mu_gemm_splitk"], "hbm", "mu_gemm"))
        roof_families = families

    log("kernel profile pass done")
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.net, sds, a.cpu_seconds)

    swin = None
    if rank == 0 and a.swin:
        from genconvit_amd.model.swin import SwinTinyEmbedder
        sw = SwinTinyEmbedder(init="empty")
        sw.load_state_dict(synth.make_state_dict(spec.swin_tiny_spec(""), synth.DEFAULT_SEED, "swin/", device=device))
        sw = sw.to(device).to(dtype).eval().reserve(a.batch)
        for _ in range(max(a.warmup, 1)):
            sw(x)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            sw(x)
        torch.cuda.synchronize()
        t_sw = (time.perf_counter() - t1) / a.steps
        sw._handle.profile_enable(True)
        sw(x)
        torch.cuda.synchronize()
        wa = [r for r in sw._handle.profile_report() if r["tag"] == "swin.window_attn"]
        sw._handle.profile_enable(False)
        if wa:
            ms = sum(r["ms"] for r in wa); nl = sum(r["launches"] for r in wa); by = sum(r["bytes"] for r in wa)
            roof_families.append({"kernel": "swin.window_attn", "bound": "hbm", "achieved": round(by / (ms * 1e-3) / 1e9, 2),
                                  "peak": PEAK["hbm"], "unit": "GB/s", "frac": round(by / (ms * 1e-3) / 1e9 / PEAK["hbm"], 4),
                                  "traffic": None, "algorithmic_bytes_per_launch": round(by / nl), "launches_per_step": nl,
                                  "avg_launch_ms": round(ms / nl, 4)})
        swin = {"note": "Swin-T embedder forward alone; NOT part of `value` (the reference never executes it in forward, "
                        "SURVEY.md section 0.4)", "frames_per_s": round(a.batch / t_sw, 1), "ms_per_step": round(t_sw * 1e3, 3),
                "frames_per_s_genconvit_plus_swin": round(a.batch / (t_sw + dt / a.steps), 1),
                "algorithmic_gflop_per_frame": 8.98}

    # evidence for the multi-GPU path: how the logit all-gather travelled, how many ranks RCCL itself counts, and which
    # device every rank ran on (gathered over the process group; one rank: just this device)
    comm = None
    if dist_on:
        names = [None] * world
        torch.distributed.all_gather_object(names, f"{torch.cuda.get_device_name(device)} (cuda:{local_rank})")
        info = gdist.comm_info()
        comm = {"backend": "nccl (RCCL)", "allgather_path": info["path"], "rccl_ranks": info["ranks"], "devices": names}
    if rank == 0:
        fps = n_global * a.steps / dt
        line = {
            "metric": "frames/sec (224x224, ed+vae forward)" if a.net == "genconvit" else f"frames/sec (224x224, {a.net} forward)",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": ("genconvit (ed+vae)" if a.net == "genconvit" else a.net) +
                                   f" forward, {a.batch} frames/GPU x {world} GPU, 224x224x3, "
                                   f"{a.dtype} storage / fp32 accumulate" + (", RCCL logit all-gather + vote" if world > 1 else ", vote"),
                       "net": a.net, "frames_per_gpu": a.batch, "global_batch": n_global,
                       "parallelism": f"frame-shard x{world}", "algorithmic_gflop_per_frame": GFLOP_PER_FRAME[a.net]},
            "roofline": roof, "roofline_families": roof_families, "cpu_baseline": cpu,
            "ms_per_step_synchronised": round(synced_ms, 4),    # diagnostic: a device sync behind every step (not `value`)
            "rccl_ranks": comm["rccl_ranks"] if comm else 0,
            "comm": comm if comm else {"backend": None, "allgather_path": "none (one process, no collective)", "rccl_ranks": 0,
                                       "devices": [f"{torch.cuda.get_device_name(device)} (cuda:{local_rank})"]},
        }
        if swin is not None:
            line["swin_embedder"] = swin
        print(json.dumps(line), flush=True)
    if dist_on:
        torch.distributed.barrier()
        gdist.close_comms()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
