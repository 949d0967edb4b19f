// Fused ConvNeXt MLP for the narrow stages (C = 96, 192), 16-bit storage:
//     out = resid + gamma * ( W2 . GELU( W1 . x_ln + b1 ) + b2 )            (timm ConvNeXtBlock)
// The 4C-wide hidden activation never leaves the CU: unfused, stage 0 moves 13*M*C*s bytes per block
// (the hidden tensor is written and read back once) and is HBM-bound in 16-bit; fused it moves 3*M*C*s.
//
// Workgroup = 8 waves = 256 tokens, each wave owns 32 tokens for the whole kernel:
//   * the wave's x_ln rows are loaded once, straight into MFMA B-operand fragments (registers)
//   * the hidden dimension is walked in chunks of HC = 96: W1[chunk] (HC x C) and W2[:, chunk] (C x HC)
//     are streamed through LDS (double buffered, register-staged, one barrier per chunk) and shared by
//     the 8 waves
//   * GEMM1 accumulators have the hidden index on registers and the token on the lane, so after
//     bias + GELU (branch-free erf) they are packed to 16-bit and fed directly as the B operand of
//     GEMM2 (guide: "an accumulator tile as the next MFMA's operand"); W2's hidden axis is
//     pre-permuted at pack time (bits 2 and 3 of the index swapped inside every 16) so its A-operand
//     fragments are plain 16-byte chunks
//   * epilogue: (acc + b2) * gamma staged through LDS per wave, residual added, 8-byte coalesced stores
// LDS rows are padded by 16 B (row stride / 16 odd) so ds_read_b128 fragment reads are conflict-free.
#pragma once
#include "gemm.h"

namespace gcv {


struct MlpArgs {
  const void* X;       // (M, C) LayerNorm'ed dw-conv output
  const void* W1;      // (4C, C) row-major (nn.Linear layout) — chunk ch = rows [ch*HC, (ch+1)*HC)
  const float* b1;     // (4C)
  const void* W2c;     // [4C/HC][C][HC] chunk-major, hidden axis permuted (pack_w2_chunks)
  const float* b2;     // (C)
  const float* gamma;  // (C)
  const void* resid;   // (M, C) block input (may alias out)
  void* out;           // (M, C)
  int M;
  // Optional (fused_mlp_res_kernel only; the LAST block of stage 0): the stage boundary's LayerNorm2d + 2x2 space-to-depth
  // (timm ConvNeXt `downsample`: LayerNorm2d -> Conv2d(k=2, s=2)) applied to the output rows, which are needed by
  // nothing else: `out` is then the patchified (M/4, 4C) operand of the down-sampling GEMM and the (M, C) residual stream is
  // not written at all.  Tokens [tok0[s], tok0[s+1]) are images of hw[s] = H*W pixels, W = w[s]; their patch rows start
  // at out0[s].
  const float* lnp_w = nullptr;
  const float* lnp_b = nullptr;
  float lnp_eps = 0.0f;
  int lnp_nseg = 0;
  int lnp_tok0[4] = {0, 0, 0, 0};
  int lnp_hw[4] = {1, 1, 1, 1};
  int lnp_wd[4] = {1, 1, 1, 1};
  int lnp_out0[4] = {0, 0, 0, 0};
};

constexpr int kMlpHC = 96;
// whether launch_fused_mlp runs the LDS-resident kernel (fused_mlp_res.h: the only one with the LN-patchify epilogue)
static inline bool fused_mlp_res_applies(int C, int64_t M) { return C == 96 && M >= 256 * 8 * 32; }

template <typename T, int C, int NW> struct MlpSmem {
  static constexpr int kRow1 = C * (int)sizeof(T) + 16;          // W1 chunk row (C elements) + pad
  static constexpr int kRow2 = kMlpHC * (int)sizeof(T) + 16;     // W2 chunk row (HC elements) + pad
  static constexpr int kBuf = kMlpHC * kRow1 + C * kRow2;        // one chunk pair
  static constexpr int kBias = 4 * C * 4;                        // b1 in LDS
  static constexpr int kStage = NW * 32 * (96 + 4) * 4;          // per-wave fp32 staging of 96 columns
  static constexpr int kMain = 2 * kBuf + kBias;
  static constexpr int bytes = kMain > kStage ? kMain : kStage;
};

template <typename T, int C, int NW>
__global__ void __launch_bounds__(NW * 64, 2) fused_mlp_kernel(const MlpArgs a) {
  static_assert(sizeof(T) == 2, "fused MLP is built for 16-bit storage");
  static_assert(C == 96 || C == 192, "fused MLP covers the C=96 and C=192 stages");
  constexpr int HC = kMlpHC;
  constexpr int NCH = 4 * C / HC;             // hidden chunks
  constexpr int KP1 = C / 16;                 // k-steps of GEMM1 (K = C)
  constexpr int NJ = HC / 32;                 // 32-wide hidden groups per chunk
  constexpr int NO = C / 32;                  // 32-wide output-channel tiles
  constexpr int NT = NW * 64;                 // threads
  constexpr int ROW1 = MlpSmem<T, C, NW>::kRow1, ROW2 = MlpSmem<T, C, NW>::kRow2, BUF = MlpSmem<T, C, NW>::kBuf;
  constexpr int P1 = HC * C * 2 / 16;         // 16-byte pieces of a W1 chunk
  constexpr int P2 = C * HC * 2 / 16;         // ... of a W2 chunk
  constexpr int PCS = (P1 + P2 + NT - 1) / NT;  // pieces per thread

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sB1 = reinterpret_cast<float*>(smem + 2 * BUF);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t m_wave = (int64_t)blockIdx.x * (NW * 32) + wave * 32;
  const int64_t m = m_wave + lr;
  const int64_t mc = m < a.M ? m : (int64_t)a.M - 1;      // clamp: tail rows compute garbage, store nothing

  const T* __restrict__ Xp = (const T*)a.X;
  const T* __restrict__ W1p = (const T*)a.W1;
  const T* __restrict__ W2p = (const T*)a.W2c;

  GCV_STAMP(0);
  for (int i = tid; i < 4 * C; i += NT) sB1[i] = a.b1[i];

  // x_ln fragments: k-step p, lane (token lr, half lh) holds k = 16p + 8lh .. +7
  u32x4 xf[KP1];
#pragma unroll
  for (int p = 0; p < KP1; ++p) xf[p] = *(const u32x4*)(Xp + mc * C + 16 * p + 8 * lh);

  // The next chunk is fetched global -> registers at the top of an iteration and written to the idle
  // LDS buffer at its end, so a whole iteration of MFMA + GELU covers the load latency.  SPLIT (C=192,
  // register pressure) fetches in two halves: before GEMM1 / before GEMM2.
  constexpr bool SPLIT = (C == 192);
  constexpr int PH = SPLIT ? (PCS + 1) / 2 : PCS;
  u32x4 stage_reg[PH];
  auto fetch = [&](int ch, int part) {
    const unsigned char* g1 = (const unsigned char*)(W1p + (int64_t)ch * HC * C);
    const unsigned char* g2 = (const unsigned char*)(W2p + (int64_t)ch * C * HC);
#pragma unroll
    for (int i = 0; i < PH; ++i) {
      const int idx = tid + (part * PH + i) * NT;
      if (idx < P1) stage_reg[i] = *(const u32x4*)(g1 + (int64_t)idx * 16);
      else if (idx < P1 + P2) stage_reg[i] = *(const u32x4*)(g2 + (int64_t)(idx - P1) * 16);
    }
  };
  auto stash = [&](int buf, int part) {
    unsigned char* s1 = smem + buf * BUF;
    unsigned char* s2 = s1 + HC * ROW1;
    constexpr int RP1 = C * 2 / 16, RP2 = HC * 2 / 16;    // pieces per row
    int t = tid;
    if (C == 192) asm volatile("" : "+v"(t));             // recompute the LDS addresses here: hoisted out of the chunk loop
                                                          // they were ten more live registers, spilled and reloaded per chunk
#pragma unroll
    for (int i = 0; i < PH; ++i) {
      const int idx = t + (part * PH + i) * NT;
      if (idx < P1) {
        const int row = idx / RP1, c = idx - row * RP1;
        *(u32x4*)(s1 + row * ROW1 + c * 16) = stage_reg[i];
      } else if (idx < P1 + P2) {
        const int k = idx - P1;
        const int row = k / RP2, c = k - row * RP2;
        *(u32x4*)(s2 + row * ROW2 + c * 16) = stage_reg[i];
      }
    }
  };

  f32x16 acc2[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[o][r] = 0.0f;

  fetch(0, 0);
  stash(0, 0);
  if (SPLIT) {
    fetch(0, 1);
    stash(0, 1);
  }
  __syncthreads();
  GCV_STAMP(1);

  for (int ch = 0; ch < NCH; ++ch) {
    if (!(GCV_MLP_ABLATE & 2) && ch + 1 < NCH) fetch(ch + 1, 0);
    const unsigned char* s1 = smem + (ch & 1) * BUF;
    const unsigned char* s2 = s1 + HC * ROW1;

    // ---- GEMM1: hidden[HC x 32 tokens] = W1[chunk] . x_ln^T --------------------------------
    f32x16 acc1[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[j][r] = 0.0f;
    // W1 fragments are read in batches of GB k-steps (GB*NJ ds_read_b128) and consumed by GB*NJ MFMAs;
    // sched_barriers fence the batches.  Left to itself hipcc emits ds_read -> s_waitcnt -> v_mfma
    // chains: at 2 waves per SIMD every one of the 144 MFMAs of a token tile then waits ~150 cycles
    // for its own LDS read (10 us of a 26 us workgroup).
    {
      constexpr int GB = (C == 96) ? 3 : 2;
      static_assert(KP1 % GB == 0, "k-steps per batch");
#pragma unroll
      for (int p0 = 0; p0 < KP1; p0 += GB) {
        u32x4 wf[GB][NJ];
#pragma unroll
        for (int pp = 0; pp < GB; ++pp)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            wf[pp][j] = *(const u32x4*)(s1 + (32 * j + lr) * ROW1 + (2 * (p0 + pp) + lh) * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pp = 0; pp < GB; ++pp)
#pragma unroll
          for (int j = 0; j < NJ; ++j) Mfma<T>::run(wf[pp][j], xf[p0 + pp], acc1[j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!(GCV_MLP_ABLATE & 2) && SPLIT && ch + 1 < NCH) {
      stash((ch + 1) & 1, 0);
      fetch(ch + 1, 1);
    }
    if (ch == 0) GCV_STAMP(7);
    // ---- bias + GELU in registers, pack to 16-bit B-operand fragments, GEMM2 -----------------
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      u32x4 hf[2];
      constexpr int QB = (C == 96) ? 2 : 1;            // channel groups per batch: 2*QB independent GELU chains
#pragma unroll
      for (int q0 = 0; q0 < 4; q0 += QB) {             // (C = 192 has no registers to spare for four)
        float hv[QB][4];
#pragma unroll
        for (int qq = 0; qq < QB; ++qq) {
          const int q = q0 + qq;
          const f32x4 bv = *(const f32x4*)(sB1 + ch * HC + 32 * j + 8 * q + 4 * lh);
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[qq][e] = acc1[j][4 * q + e] + bv[e];
        }
        if (!(GCV_MLP_ABLATE & 1)) act4n<ACT_GELU, T, QB>(hv);
#pragma unroll
        for (int qq = 0; qq < QB; ++qq) {
          const int q = q0 + qq;
          typedef T t4 __attribute__((ext_vector_type(4)));
          const t4 h4 = {from_f<T>(hv[qq][0]), from_f<T>(hv[qq][1]), from_f<T>(hv[qq][2]), from_f<T>(hv[qq][3])};
          const uint2 pk = __builtin_bit_cast(uint2, h4);
          hf[q >> 1][2 * (q & 1)] = pk.x;
          hf[q >> 1][2 * (q & 1) + 1] = pk.y;
        }
      }
      {
        // SB k-steps of this 32-hidden group per batch: SB*NO fragment reads, then SB*NO MFMAs
        constexpr int SB = (C == 96) ? 2 : 1;
#pragma unroll
        for (int s0 = 0; s0 < 2; s0 += SB) {
          u32x4 w2f[SB][NO];
#pragma unroll
          for (int s = 0; s < SB; ++s)
#pragma unroll
            for (int o = 0; o < NO; ++o)
              w2f[s][o] = *(const u32x4*)(s2 + (32 * o + lr) * ROW2 + (2 * (2 * j + s0 + s) + lh) * 16);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 0; s < SB; ++s)
#pragma unroll
            for (int o = 0; o < ((GCV_MLP_ABLATE & 4) ? 1 : NO); ++o) Mfma<T>::run(w2f[s][o], hf[s0 + s], acc2[o]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (ch == 0) GCV_STAMP(8);
    if (!(GCV_MLP_ABLATE & 2) && ch + 1 < NCH) stash((ch + 1) & 1, SPLIT ? 1 : 0);
    if (ch == 0) GCV_STAMP(9);
    __syncthreads();
    if (ch < 4) GCV_STAMP(2 + ch);
  }

  // ---- epilogue: (acc2 + b2) * gamma -> per-wave LDS tile [32 tokens][96 cols] fp32 -> + resid -> store
  float* sC = reinterpret_cast<float*>(smem) + wave * 32 * 100;
  const T* Rp = (const T*)a.resid;
  T* Op = (T*)a.out;
#pragma unroll
  for (int half = 0; half < NO / 3; ++half) {
#pragma unroll
    for (int oo = 0; oo < 3; ++oo) {
      const int o = half * 3 + oo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = 32 * o + 8 * q + 4 * lh;
        const f32x4 bv = *(const f32x4*)(a.b2 + n);
        const f32x4 gv = *(const f32x4*)(a.gamma + n);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (acc2[o][4 * q + e] + bv[e]) * gv[e];
        *(f32x4*)(sC + lr * 100 + 32 * oo + 8 * q + 4 * lh) = v;
      }
    }
    __syncthreads();
    typedef T t4 __attribute__((ext_vector_type(4)));
    // residual rows first, as 12 independent loads from clamped (always valid) addresses: inside the
    // `row < M` branch each load waited out its own HBM latency (12 x ~1.2k cycles per workgroup)
    t4 rres[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int idx = lane + 64 * i;            // 32 rows x 24 four-channel pieces
      const int row = idx / 24, pc = idx - row * 24;
      const int64_t mm = m_wave + row;
      const int64_t mmc = mm < a.M ? mm : (int64_t)a.M - 1;
      rres[i] = *(const t4*)(Rp + mmc * C + half * 96 + 4 * pc);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int idx = lane + 64 * i;
      const int row = idx / 24, pc = idx - row * 24;
      const int64_t mm = m_wave + row;
      const f32x4 v = *(const f32x4*)(sC + row * 100 + 4 * pc);
      t4 o4;
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = from_f<T>(v[e] + to_f(rres[i][e]));
      if (mm < a.M) *(t4*)(Op + mm * C + half * 96 + 4 * pc) = o4;
    }
    __syncthreads();
  }
  GCV_STAMP(6);
}

template <typename T> int launch_fused_mlp(const MlpArgs& a, int C, hipStream_t s);
// W2 (C,4C) fp32 on device -> chunk-major, hidden-permuted T (see header)
template <typename T> int launch_pack_w2_chunks(const float* w2_dev, T* out, int C, hipStream_t s);

}  // namespace gcv
