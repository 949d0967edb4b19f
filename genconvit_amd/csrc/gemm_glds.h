// Pipelined MFMA GEMM for the large pointwise layers (16-bit storage, plain A operand):
//     C[m][n] = epilogue( sum_k A[m][k] * Wt[n][k] ),   N % 192 == 0, K % 32 == 0
//
// Why a second GEMM kernel: the register-staged kernel in gemm.h keeps ONE K tile per workgroup in
// flight; at the ConvNeXt stage-2/3 shapes a K tile is ~400 MFMA cycles of work but ~1.8 us of
// memory latency (the 4C-wide hidden operand streams from HBM), and its 128x96 tile moves 0.018 B of
// operand per FLOP through L2.  This kernel
//   * loads operands with LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write) into a ring of K-tile
//     stages and waits with a COUNTED vmcnt — one raw s_barrier per K tile.  The fill rate of a CU does not depend on
//     the ring depth (profiles/micro/lds_dma_rate.hip), only on how many workgroups issue and on the piece size, so
//     the ring is double-buffered: 64-byte rows 2 x 20 KB (three workgroups per CU), 128-byte rows 2 x 40 KB (two)
//   * uses a 128 x 192 tile, 4 waves as 2(M) x 2(N), 64 x 96 per wave (2x3 32x32 accumulators):
//     0.013 B/FLOP from L2 and 0.83 LDS fragment reads per MFMA
//   * LDS image is lane-linear per DMA instruction (16 rows x 64 B); the bank swizzle is applied to the
//     SOURCE chunk each lane fetches and to the fragment read (guide §5.4 rule 21)
// Out-of-range rows are clamped to the last valid row when loading (their results are never stored);
// K has no tail by construction.  Epilogue as in gemm.h (tokens on lanes, LDS-staged coalesced
// stores); the layer-scale residual is added in registers before staging so the stage stays 16-bit.
#pragma once
#include "gemm.h"

namespace gcv {

#ifndef GCV_GLDS_STAGES
#define GCV_GLDS_STAGES 2     // 64-byte-row ring: 2 x 20 KB leaves room for THREE workgroups per CU (-5 % at K = 384 vs 4 stages)
#endif
constexpr int kGldsBM = 128, kGldsBN = 192;
// ring geometry by bytes per LDS row: 64 B (32 k) x 2 stages (40 KB: the epilogue staging then sets the footprint, 53 KB,
// three workgroups per CU), or 128 B (64 k, a whole 128-byte line per row) x 2 stages (80 KB, two per CU); whole-line
// fetches fill LDS 1.3-1.4x faster (profiles/micro/lds_dma_rate.hip; the fill rate does not depend on the ring depth), so
// 128 is used when K % 64 == 0 and K >= 512
template <int BKB> struct GldsRing { static constexpr int stages = BKB == 64 ? GCV_GLDS_STAGES : 2; };

template <typename T, int BKB = 64> struct GldsSmem {
  static constexpr int kStage = (kGldsBM + kGldsBN) * BKB;                 // 20480 / 40960
  static constexpr int kMain = GldsRing<BKB>::stages * kStage;             // 81920
  static constexpr int kEpiRow = kGldsBN / 2 + 4;                          // dwords per staged row: 16-B aligned rows
  static constexpr int kEpiBG = kGldsBM * kEpiRow * 4;                      // byte offset of the bias|gamma broadcast rows
  static constexpr int kEpi = kEpiBG + 2 * kGldsBN * 4;
  static constexpr int bytes = kMain > kEpi ? kMain : kEpi;
};

template <typename T, int EPI, int ACT, int BKB = 64>
__global__ void __launch_bounds__(256, 2) gemm_glds_kernel(const GemmArgs g) {
  static_assert(sizeof(T) == 2, "LDS-DMA GEMM is built for 16-bit storage");
  static_assert(EPI == EPI_BIAS_ACT || EPI == EPI_RESID, "epilogues: bias+act, layer-scale residual");
  static_assert(BKB == 64 || BKB == 128, "LDS rows are 64 or 128 bytes");
  constexpr int BM = kGldsBM, BN = kGldsBN, S = GldsRing<BKB>::stages;
  constexpr int EPC = 8, CPR = BKB / 16, BK = CPR * EPC;          // 4 (8) chunks, 32 (64) k per tile
  constexpr int RPI = 64 / CPR;                                   // rows per DMA wave-instruction: 16 (8)
  constexpr int STAGE = GldsSmem<T, BKB>::kStage;
  // bank swizzle key of a row: the 16 lanes of a ds_read_b128 group must land on 16 distinct 16-byte units mod 16
  auto swz = [](int row) { return CPR == 4 ? ((row >> 2) & 3) : ((row >> 1) & 7); };
  constexpr int A_BYTES = BM * BKB;
  constexpr int NQ = STAGE / 1024;                                // DMA wave-instructions per stage (20 / 40)
  constexpr int QPW = NQ / 4;                                     // per wave (5 / 10)
  constexpr int MI = 2, NI = 3;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the DMA's LDS base (M0) and the tile offsets
                                                                  // stay in SGPRs instead of a v_readfirstlane per DMA
  const int lr = lane & 31, lh = lane >> 5;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 96;

  GLDS_STAMP(0);
  GLDS_STAMP_HWID();
  const int ntn = g.N / BN;
  const int ntm = (g.M + BM - 1) / BM;
  const int bid = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (bid / ntn) * BM;
  const int n0 = (bid % ntn) * BN;
  const int nkt = g.K / BK;

  // ---- per-lane DMA sources: instruction q = wave + 4 i covers LDS rows [16 q, 16 q + 16) of the stage
  const T* src[QPW];
#pragma unroll
  for (int i = 0; i < QPW; ++i) {
    const int q = wave + 4 * i;
    const int rl = q * RPI + lane / CPR;                 // row inside the stage image
    const int phys = lane % CPR;
    if (rl < BM) {
      const int c = phys ^ swz(rl);
      const int m = min(m0 + rl, g.M - 1);
      src[i] = (const T*)g.A + (int64_t)m * g.lda + c * EPC;
    } else {
      const int rb = rl - BM;
      const int c = phys ^ swz(rb);
      const int n = min(n0 + rb, g.N - 1);
      src[i] = (const T*)g.Wt + (int64_t)n * g.K + c * EPC;
    }
  }
  auto issue = [&](int kt) {
    unsigned char* base = smem + (kt % S) * STAGE;
#pragma unroll
    for (int i = 0; i < QPW; ++i) {
      const int q = wave + 4 * i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (int64_t)kt * BK),
                                       (__attribute__((address_space(3))) void*)(base + q * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  auto compute = [&](int kt) {
    const unsigned char* sA = smem + (kt % S) * STAGE;
    const unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int t = 0; t < CPR / 2; ++t) {
      const int c = 2 * t + lh;
      u32x4 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm0 + i * 32 + lr;
        af[i] = *(const u32x4*)(sA + row * BKB + ((c ^ swz(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wn0 + j * 32 + lr;
        bf[j] = *(const u32x4*)(sB + row * BKB + ((c ^ swz(row)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) Mfma<T>::run(bf[j], af[i], acc[i][j]);
    }
  };

  // bias / layer-scale for this tile's 192 channels: fetched now by 96 lanes (one f32x4 each) so the latency hides
  // under the main loop; broadcast through LDS once the ring is free.  (Older than every DMA, so the counted
  // vmcnt waits below are unaffected.)
  f32x4 pre = {0.f, 0.f, 0.f, 0.f};
  if (tid < BN / 4) {
    if (g.bias) pre = *(const f32x4*)(g.bias + n0 + 4 * tid);
  } else if (EPI == EPI_RESID && tid < BN / 2) {
    pre = *(const f32x4*)(g.gamma + n0 + 4 * (tid - BN / 4));
  }

  // the residual rows of the tile (EPI_RESID: 24 independent 8-byte loads per lane) are issued here as well, ahead of every
  // DMA: they land under the K loop instead of costing the epilogue a round trip to HBM (48 registers held until then; the
  // kernel runs two workgroups per CU at K >= 512, where this epilogue is used, so 256 are available)
  typedef T t4 __attribute__((ext_vector_type(4)));
  t4 rres[EPI == EPI_RESID ? MI : 1][EPI == EPI_RESID ? NI : 1][4];
  auto load_resid = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm0 + i * 32 + lr;
      const int64_t mm = m < g.M ? m : g.M - 1;
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rres[i][j][q] = *(const t4*)((const T*)g.resid + mm * g.ldc + n0 + wn0 + j * 32 + 8 * q + 4 * lh);
    }
  };
  // (the 64-byte-row ring runs three workgroups per CU, 168 registers each: there the loads stay in the epilogue)
  constexpr bool RESID_EARLY = EPI == EPI_RESID && BKB == 128;
  if (RESID_EARLY && !(GCV_GLDS_ABLATE & 4)) load_resid();

  // ---- pipeline: stages kt+1 .. kt+S-2 stay in flight while stage kt is consumed ----
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nkt) issue(s);
  const int n_steady = nkt - (S - 2) > 0 ? nkt - (S - 2) : 0;
  for (int kt = 0; kt < n_steady; ++kt) {
    // my DMAs for stage kt have landed once at most (S-2) younger stages (QPW each) are outstanding
    // (one asm statement so no LDS access can be scheduled between the wait and the barrier)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((S - 2) * QPW) : "memory");   // stage kt landed everywhere; all left kt-1
    if (kt == 0) GLDS_STAMP(1);
    if (!(GCV_GLDS_ABLATE & 2) && kt + S - 1 < nkt) issue(kt + S - 1); // refill the buffer stage kt-1 used
    if (!(GCV_GLDS_ABLATE & 1)) compute(kt);
  }
  for (int kt = n_steady; kt < nkt; ++kt) {  // drain: fewer stages outstanding, wait for all of them
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (!(GCV_GLDS_ABLATE & 1)) compute(kt);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // all waves done reading the ring before it is reused
  GLDS_STAMP(2);

  // ---- epilogue: token on the lane, 4 consecutive channels in 4 consecutive registers ----
  // Every global read the epilogue needs is in flight at once (the residual: 24 independent 8-byte loads per lane,
  // issued before the barrier) or already in registers (bias / gamma), so it pays ONE memory latency, not one
  // per 4-channel group.
  constexpr int SROW = GldsSmem<T, BKB>::kEpiRow;
  uint32_t* sC = reinterpret_cast<uint32_t*>(smem);
  float* sBG = reinterpret_cast<float*>(smem + GldsSmem<T, BKB>::kEpiBG);
  T* Cp = (T*)g.C;
  if (EPI == EPI_RESID && !RESID_EARLY && !(GCV_GLDS_ABLATE & 4)) load_resid();
  if (tid < BN / 2) *(f32x4*)(sBG + 4 * tid) = pre;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NI; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = wn0 + j * 32 + 8 * q + 4 * lh;
      const f32x4 bv = *(const f32x4*)(sBG + nl);
      f32x4 gv = {1.f, 1.f, 1.f, 1.f};
      if (EPI == EPI_RESID) gv = *(const f32x4*)(sBG + BN + nl);
      float v[MI][4];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = acc[i][j][4 * q + e];
      if (!(GCV_GLDS_ABLATE & 4)) bias_act4n<ACT, T, MI>(v, bv);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int ml = wm0 + i * 32 + lr;
        if (EPI == EPI_RESID && !(GCV_GLDS_ABLATE & 4)) {
          const t4 r = rres[i][j][q];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][e] = fmaf(v[i][e], gv[e], to_f(r[e]));
        }
        t4 o = {from_f<T>(v[i][0]), from_f<T>(v[i][1]), from_f<T>(v[i][2]), from_f<T>(v[i][3])};
        *(t4*)(sC + ml * SROW + (nl >> 1)) = o;
      }
    }
  }
  GLDS_STAMP(3);
  __syncthreads();
  // coalesced write-out: 16 bytes per lane, 24 lanes per 384-byte tile row
  constexpr int PPR = BN / 8;
#pragma unroll 4
  for (int idx = tid; idx < BM * PPR; idx += 256) {
    const int rl = idx / PPR, pc = idx - rl * PPR;
    const int m = m0 + rl;
    if (m < g.M) *(u32x4*)(Cp + (int64_t)m * g.ldc + n0 + 8 * pc) = *(const u32x4*)(sC + rl * SROW + 4 * pc);
  }
  GLDS_STAMP(4);      // stores issued
#if GCV_GLDS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GLDS_STAMP(6);      // stores acknowledged
#endif
}

template <typename T> int launch_gemm_glds(const GemmArgs& g, int epi, hipStream_t s);
// true when (g, a_mode, epi) is a shape the LDS-DMA kernel covers
template <typename T> bool gemm_glds_applicable(const GemmArgs& g, int a_mode, int epi);

}  // namespace gcv
