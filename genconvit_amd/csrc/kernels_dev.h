// Non-GEMM kernels of the GenConViT path: HBM-bound NHWC kernels with
// fp32 math, 64-lane wave reductions and LDS-staged cross-channel statistics.
//   K3  stem conv4x4s4 + LayerNorm2d            stem_ln_kernel
//   K4  depthwise 7x7 + LayerNorm               dwconv7_ln_kernel
//   K6  LayerNorm2d + 2x2 space-to-depth        ln_patchify_kernel   (conv2x2s2 itself is a GEMM)
//   K7  global-avg-pool + LayerNorm2d           pool_ln_kernel       (fc is a GEMM)
//   K1/K9 first 3->16 conv (Cin=3 is not MFMA-shaped)   conv3_first_kernel
//   K2/K12 last 16->3 ConvTranspose2d           convt2_small_kernel
//   K10/K11 split-K reduce + bias + reparameterise      reparam_kernel, kl_rows_kernel
//   K8  500->2 head tail                        head_tail_kernel
//   K13/K14 bilinear 112->224 (+ per-frame MSE) resize_mse_kernel
//   K15 sigmoid -> mean over rows               vote_kernel
#pragma once
#include "common.h"
#include "gemm.h"

namespace gcv {

// 16-bit storage values as raw bits (packed-pair math, LDS images)
template <typename T> __device__ __forceinline__ uint32_t bits16(float v) {
  const T t = from_f<T>(v);
  return (uint32_t)__builtin_bit_cast(unsigned short, t);
}
template <typename T> __device__ __forceinline__ float from_bits16(uint32_t u) {
  return to_f(__builtin_bit_cast(T, (unsigned short)u));
}

// ------------------------------------------------------------------ K3: stem
// conv 4x4 stride 4 (3 -> 96) + LayerNorm over the 96 channels, output NHWC tokens.
// Input addressed by element strides so NCHW frames and the NHWC reconstruction both work.
// wp: [48][96] fp32 with k = ci*16 + ky*4 + kx.
constexpr int kStemTok = 128;          // tokens per workgroup
template <typename T>
__global__ void __launch_bounds__(256) stem_ln_kernel(const T* __restrict__ x, int64_t sb, int64_t sc, int64_t sy,
                                                      int64_t sx, const float* __restrict__ wp,
                                                      const float* __restrict__ bias, const float* __restrict__ lnw,
                                                      const float* __restrict__ lnb, T* __restrict__ out, int nimg,
                                                      int Ho, int Wo, float eps) {
  __shared__ __attribute__((aligned(16))) float sW[48 * 96];
  __shared__ float sIn[kStemTok][49];
  const int tid = threadIdx.x;
  for (int i = tid; i < 48 * 96 / 4; i += 256) reinterpret_cast<float4*>(sW)[i] = reinterpret_cast<const float4*>(wp)[i];
  const int64_t total = (int64_t)nimg * Ho * Wo;
  const int64_t p0 = (int64_t)blockIdx.x * kStemTok;
  // patch staging.  Both layouts the path uses keep 4 consecutive patch elements contiguous in memory — NCHW: the 4 kx
  // of one (ci, ky); NHWC: 4 of the 12 (kx, ci) of one ky — so a work item is one 8-byte load (per-element staging
  // spent ~20 integer instructions and one 2-byte load on each of the 48 patch elements).
  const bool al = (reinterpret_cast<uintptr_t>(x) & 7u) == 0 && ((sb | sy) & 3) == 0;
  const bool nchw4 = sizeof(T) == 2 && al && sx == 1 && (sc & 3) == 0;
  const bool nhwc4 = sizeof(T) == 2 && al && sc == 1 && sx == 3;
  if constexpr (sizeof(T) == 2) {
   if (nchw4 || nhwc4) {
    for (int e = tid; e < kStemTok * 12; e += 256) {
      const int p = e / 12, r = e - p * 12;              // NCHW: r = ci*4 + ky ; NHWC: r = ky*3 + third
      const int64_t gp = p0 + p;
      uint2 raw = {0u, 0u};
      const int ky = nchw4 ? (r & 3) : (r / 3);
      const int sub = nchw4 ? (r >> 2) : (r - 3 * ky);   // NCHW: ci ; NHWC: which third of the 12 (kx, ci) elements
      if (gp < total) {
        const int xo = (int)(gp % Wo);
        const int64_t t = gp / Wo;
        const int yo = (int)(t % Ho);
        const int64_t b = t / Ho;
        const int64_t base = b * sb + (int64_t)(4 * yo + ky) * sy +
                             (nchw4 ? (int64_t)sub * sc + 4 * xo : (int64_t)12 * xo + 4 * sub);
        raw = *reinterpret_cast<const uint2*>(x + base);
      }
      const uint32_t h[4] = {raw.x & 0xffffu, raw.x >> 16, raw.y & 0xffffu, raw.y >> 16};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = from_bits16<T>(h[j]);
        if (nchw4) {
          sIn[p][sub * 16 + ky * 4 + j] = v;             // k = ci*16 + ky*4 + kx
        } else {
          const int el = 4 * sub + j, kx = el / 3, ci = el - 3 * kx;
          sIn[p][ci * 16 + ky * 4 + kx] = v;
        }
      }
    }
   }
  }
  if (!(nchw4 || nhwc4)) {
    for (int e = tid; e < kStemTok * 48; e += 256) {
      const int p = e / 48, k = e - p * 48;
      const int64_t gp = p0 + p;
      float v = 0.0f;
      if (gp < total) {
        const int xo = (int)(gp % Wo);
        const int64_t t = gp / Wo;
        const int yo = (int)(t % Ho);
        const int64_t b = t / Ho;
        const int ci = k >> 4, ky = (k >> 2) & 3, kx = k & 3;
        v = to_f(x[b * sb + ci * sc + (int64_t)(4 * yo + ky) * sy + (int64_t)(4 * xo + kx) * sx]);
      }
      sIn[p][k] = v;
    }
  }
  __syncthreads();
  // thread = (4 tokens, 12-channel group): three float4 of weights from LDS feed 48 FMAs (one token per thread issued
  // 13 LDS reads per 12 FMAs)
  const int pq = tid >> 3, cg = tid & 7;               // tokens 4 pq .. 4 pq + 3, channels 12 cg .. 12 cg + 11
  float acc[4][12];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[t][i] = bias[cg * 12 + i];
#pragma unroll 2
  for (int k = 0; k < 48; ++k) {
    const float4* wr = reinterpret_cast<const float4*>(sW + k * 96 + cg * 12);
    float w[12];
#pragma unroll
    for (int g4 = 0; g4 < 3; ++g4) {
      const float4 v4 = wr[g4];
      w[4 * g4] = v4.x; w[4 * g4 + 1] = v4.y; w[4 * g4 + 2] = v4.z; w[4 * g4 + 3] = v4.w;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float xv = sIn[4 * pq + t][k];
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[t][i] = fmaf(xv, w[i], acc[t][i]);
    }
  }
  float lw[12], lb[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) { lw[i] = lnw[cg * 12 + i]; lb[i] = lnb[cg * 12 + i]; }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += acc[t][i];
    s = group8_sum(s);
    const float mean = s * (1.0f / 96.0f);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { const float d = acc[t][i] - mean; q = fmaf(d, d, q); }
    q = group8_sum(q);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 96.0f) + eps);
    const int64_t gp = p0 + 4 * pq + t;
    if (gp < total) {
      T* o = out + gp * 96 + cg * 12;
#pragma unroll
      for (int i = 0; i < 12; ++i) o[i] = from_f<T>((acc[t][i] - mean) * rstd * lw[i] + lb[i]);
    }
  }
}

// ------------------------------------------------------------------ K3 on the matrix pipe (16-bit storage)
// The stem is a [tokens x 48] x [48 x 96] contraction: 4608 MACs per token cost the vector pipe more than the
// 288 bytes the token moves.  Here a wave owns 32-token tiles: the patch IS the MFMA operand — a lane's 8 k values of
// a k-step are two 8-byte pieces of the frame (NCHW: the 4 kx of rows ky, ky+1 of one channel; NHWC: two thirds of
// the 12 (kx, ci) values of a row), fetched straight into registers one tile ahead — 3 k-steps x 3 channel tiles =
// 9 MFMAs per tile; the 16-bit weight fragments (k permuted to the frame's memory order) are built once per
// workgroup in LDS.  LayerNorm runs on the accumulator (token on the lane, the two half-waves hold 48 channels
// each; two-pass statistics, one v_permlane32_swap per partial), and the store is six 16-byte pieces per lane.
template <typename T>
__global__ void __launch_bounds__(256, 3) stem_ln_mfma_kernel(const T* __restrict__ x, int64_t sb, int64_t sc, int64_t sy,
                                                           int nhwc, const float* __restrict__ wp,
                                                           const float* __restrict__ bias, const float* __restrict__ lnw,
                                                           const float* __restrict__ lnb, T* __restrict__ out, int total,
                                                           int Ho, int Wo, float eps) {
  static_assert(sizeof(T) == 2, "matrix-pipe stem is built for 16-bit storage");
  __shared__ __attribute__((aligned(16))) T sWf[9 * 64 * 8];          // [o][p][lane][8]
  __shared__ __attribute__((aligned(16))) float sPar[3 * 96];         // bias | ln weight | ln bias
  const int tid = threadIdx.x;
  for (int idx = tid; idx < 48 * 96; idx += 256) {
    const int k = idx / 96, n = idx - k * 96;
    const int ci = k >> 4, ky = (k >> 2) & 3, kx = k & 3;
    const int kk = nhwc ? ky * 12 + kx * 3 + ci : k;                  // position in the frame's memory order
    const int o = n >> 5, lr = n & 31, p = kk >> 4, lh = (kk >> 3) & 1, i = kk & 7;
    sWf[((o * 3 + p) * 64 + lh * 32 + lr) * 8 + i] = from_f<T>(wp[idx]);
  }
  for (int i = tid; i < 3 * 96; i += 256) sPar[i] = i < 96 ? bias[i] : (i < 192 ? lnw[i - 96] : lnb[i - 192]);
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  u32x4 wf[3][3];
#pragma unroll
  for (int o = 0; o < 3; ++o)
#pragma unroll
    for (int p = 0; p < 3; ++p) wf[o][p] = *reinterpret_cast<const u32x4*>(sWf + ((o * 3 + p) * 64 + lane) * 8);

  const int ntiles = (total + 31) / 32;
  const int stride = (int)gridDim.x * 4;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef T t4 __attribute__((ext_vector_type(4)));
  u32x2 xf[3][2];
  auto load_x = [&](int tile) {
    const int gp = min(tile * 32 + lr, total - 1);         // tail rows recompute the last token, store nothing
    const int xo = gp % Wo, t = gp / Wo;
    const int yo = t % Ho, b = t / Ho;
    const T* base = x + (int64_t)b * sb + (int64_t)(4 * yo) * sy;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const T* src;
        if (nhwc) {
          const int j = 4 * p + 2 * lh + h;                // piece j of the 12: row ky = j / 3, third j % 3
          const int ky = j / 3, th = j - 3 * ky;
          src = base + (int64_t)ky * sy + 12 * xo + 4 * th;
        } else {
          src = base + (int64_t)p * sc + (int64_t)(2 * lh + h) * sy + 4 * xo;
        }
        xf[p][h] = *reinterpret_cast<const u32x2*>(src);
      }
  };
  int tile = (int)blockIdx.x * 4 + wave;
  if (tile < ntiles) load_x(tile);
  for (; tile < ntiles; tile += stride) {
    const float* sp = sPar;                                // (through an empty asm: the parameter reads stay in the loop
    asm volatile("" : "+v"(sp));                           //  instead of 144 hoisted registers)
    f32x16 acc[3];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(sp + 32 * o + 8 * q + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[o][4 * q + e] = bv[e];
      }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const u32x4 xv = {xf[p][0][0], xf[p][0][1], xf[p][1][0], xf[p][1][1]};
#pragma unroll
      for (int o = 0; o < 3; ++o) Mfma<T>::run(wf[o][p], xv, acc[o]);
    }
    if (tile + stride < ntiles) load_x(tile + stride);      // lands under the LayerNorm and the stores
    // LayerNorm over the token's 96 channels: 48 on this lane, 48 on lane ^ 32
    // (ds_bpermute, not v_permlane32_swap: with one value as both operands hipcc 7.2 folds the two results into one
    // register and adds it to itself)
    auto both = [&](float v) { return v + __shfl_xor(v, 32, 64); };
    float s = 0.0f;
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[o][r];
    const float mean = both(s) * (1.0f / 96.0f);
    float qq = 0.0f;
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = acc[o][r] - mean; qq = fmaf(d, d, qq); }
    const float rstd = 1.0f / sqrtf(both(qq) * (1.0f / 96.0f) + eps);
    const int m = tile * 32 + lr;
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int q = 0; q < 4; q += 2) {
        uint2 pk[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int n = 32 * o + 8 * (q + d) + 4 * lh;
          const f32x4 lw4 = *reinterpret_cast<const f32x4*>(sp + 96 + n);
          const f32x4 lb4 = *reinterpret_cast<const f32x4*>(sp + 192 + n);
          t4 o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) o4[e] = from_f<T>(fmaf((acc[o][4 * (q + d) + e] - mean) * rstd, lw4[e], lb4[e]));
          pk[d] = __builtin_bit_cast(uint2, o4);
        }
        // lanes 32-63 of piece q <-> lanes 0-31 of piece q+1: every lane ends up with 16 contiguous bytes
        const auto sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
        const auto sy2 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
        const u32x4 w = {sx[0], sy2[0], sx[1], sy2[1]};
        if (m < total) *reinterpret_cast<u32x4*>(out + (int64_t)m * 96 + 32 * o + 8 * q + 8 * lh) = w;
      }
  }
}

// ------------------------------------------------------------------ K4: dwconv7x7 + LN (generic tile kernel)
// The ConvNeXt-T shapes with W % 7 == 0 run the rolling-strip kernel of dwconv_roll.h; this one covers the rest
// (the 3x3 map of the 112-px pass).
// NHWC.  One thread per channel of a 7x7 output tile: the 49 taps of that channel live in
// registers, every input value of the 13x13 halo window is loaded once (lanes = consecutive
// channels -> coalesced) and scattered into the <=49 accumulators it touches.  LayerNorm over
// channels: conv outputs staged in LDS [pixel][C], 32-lane groups reduce a pixel each
// (two-pass mean / variance), every thread then normalises its own registers.
// C=96 packs two tiles per workgroup (192 threads) so no lanes idle.
template <typename T, int C>
__global__ void __launch_bounds__((C == 96) ? 192 : C, (C == 768) ? 3 : 2)
dwconv7_ln_kernel(const T* __restrict__ x, const float* __restrict__ wdw /*[49][C]*/,
                  const float* __restrict__ bdw, const float* __restrict__ lnw, const float* __restrict__ lnb,
                  T* __restrict__ y, int nimg, int H, int W, float eps) {
  constexpr int TILES = (C == 96) ? 2 : 1;
  constexpr int TPB = C * TILES;
  constexpr int NP = TILES * 49;
  extern __shared__ __attribute__((aligned(16))) float dw_lds[];   // [NP][C] values, then [NP][2] stats
  float* stats = dw_lds + NP * C;

  const int tid = threadIdx.x;
  const int tslot = tid / C;
  const int c = tid - tslot * C;
  const int tiles_x = (W + 6) / 7, tiles_y = (H + 6) / 7;
  const int total = nimg * tiles_x * tiles_y;
  const int tile_raw = xcd_remap(blockIdx.x, gridDim.x) * TILES + tslot;   // contiguous tile runs per XCD
  const bool tile_ok = tile_raw < total;
  const int tile = tile_ok ? tile_raw : total - 1;   // idle slot recomputes the last tile, stores nothing
  const int tx = tile % tiles_x, t2 = tile / tiles_x;
  const int ty = t2 % tiles_y, b = t2 / tiles_y;
  const int x0 = tx * 7, y0 = ty * 7;

  float w[49];
#pragma unroll
  for (int t = 0; t < 49; ++t) w[t] = wdw[t * C + c];
  const float bv = bdw[c];

  // 13x13 halo window, fully unrolled so acc[]/w[] stay in registers; loads are issued one halo
  // row ahead and fenced per row (sched_barrier) so the compiler cannot hoist all 169 of them.
  float acc[49];
#pragma unroll
  for (int t = 0; t < 49; ++t) acc[t] = bv;
  const T* xb = x + (int64_t)b * H * W * C + c;
  // branch-free halo loads: clamp the address into the image, zero the value by select
  // (multiplying by a 0/1 mask instead of selecting keeps the loads unconditional: one basic block)
  int xoff[13];
  float xmask[13];
#pragma unroll
  for (int s = 0; s < 13; ++s) {
    const int ix = x0 + s - 3;
    xmask[s] = (ix >= 0 && ix < W) ? 1.0f : 0.0f;
    xoff[s] = min(max(ix, 0), W - 1) * C;
  }
  // The halo window is loaded in batches of RB rows (5+5+3, or 2-row batches at C=768 where 12 waves per
  // workgroup cap the register file at 168): all loads of a batch are independent and
  // in flight together, then its FMAs run.  (A row-by-row prefetch exposed one memory latency per row:
  // at the 14x14 / 7x7 stages there are only 512 / 128 workgroups, nothing else hides it.)
  constexpr int RB = (C == 768) ? 2 : 5;
#pragma unroll
  for (int rb = 0; rb < 13; rb += RB) {
    float v[RB][13];
#pragma unroll
    for (int rr = 0; rr < RB; ++rr) {
      const int r = rb + rr;
      if (r < 13) {
        const int iy = y0 + r - 3;
        const float rmask = (iy >= 0 && iy < H) ? 1.0f : 0.0f;
        const T* rp = xb + (int64_t)min(max(iy, 0), H - 1) * W * C;
#pragma unroll
        for (int s = 0; s < 13; ++s) v[rr][s] = to_f(rp[xoff[s]]) * (rmask * xmask[s]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rr = 0; rr < RB; ++rr) {
      const int r = rb + rr;
      if (r < 13) {
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
          for (int sx = 0; sx < 13; ++sx) {
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
              const int oy = r - ky, ox = sx - kx;
              if (oy >= 0 && oy < 7 && ox >= 0 && ox < 7)
                acc[oy * 7 + ox] = fmaf(v[rr][sx], w[ky * 7 + kx], acc[oy * 7 + ox]);
            }
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float* sval = dw_lds + tslot * 49 * C;
#pragma unroll
  for (int p = 0; p < 49; ++p) sval[p * C + c] = acc[p];
  __syncthreads();

  const int grp = tid >> 5, gl = tid & 31;
  constexpr int NG = TPB / 32;
  for (int p = grp; p < NP; p += NG) {
    const float* row = dw_lds + p * C;
    float rv[C / 32];                                // the pixel's values stay in registers for both passes
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < C / 32; ++k) { rv[k] = row[gl + 32 * k]; s += rv[k]; }
    s = group32_sum(s);
    const float mean = s * (1.0f / C);
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < C / 32; ++k) { const float d = rv[k] - mean; q = fmaf(d, d, q); }
    q = group32_sum(q);
    if (gl == 0) {
      stats[2 * p] = mean;
      stats[2 * p + 1] = 1.0f / sqrtf(q * (1.0f / C) + eps);
    }
  }
  __syncthreads();

  if (tile_ok) {
    const float lw = lnw[c], lb = lnb[c];
    T* yb = y + (int64_t)b * H * W * C + c;
#pragma unroll
    for (int p = 0; p < 49; ++p) {
      const int oy = y0 + p / 7, ox = x0 + p % 7;
      if (oy < H && ox < W) {
        const float mean = stats[2 * (tslot * 49 + p)], rstd = stats[2 * (tslot * 49 + p) + 1];
        yb[((int64_t)oy * W + ox) * C] = from_f<T>((acc[p] - mean) * rstd * lw + lb);
      }
    }
  }
}

// ------------------------------------------------------------------ K4, tiny maps
// dw7x7 + LayerNorm on S x S maps with S <= 4 (the 3 x 3 stage-3 map of the 112-pixel pass: 7 // 2 = 3): the 7x7 window
// of EVERY output covers the whole map, so a channel's S*S outputs are S*S dot products over its S*S inputs with the taps
// (2S-1)^2 of the 49 that can reach a pixel.  One workgroup per image, one thread per channel, inputs, outputs and taps in
// registers, LayerNorm statistics (two passes) through per-wave partial sums.  The generic tile kernel walked a 13 x 13
// masked halo window for these nine pixels: 28 us per launch at 128 images, against 3 here.
template <typename T, int C, int S>
__global__ void __launch_bounds__(C) dwconv7_ln_tiny_kernel(const T* __restrict__ x, const float* __restrict__ wdw /*[49][C]*/,
                                                            const float* __restrict__ bdw, const float* __restrict__ lnw,
                                                            const float* __restrict__ lnb, T* __restrict__ y, float eps) {
  static_assert(S >= 1 && S <= 4 && C % 64 == 0, "tiny maps");
  constexpr int NPX = S * S, NW = C / 64;
  __shared__ float part[2][NW][NPX];
  const int c = threadIdx.x, wave = c >> 6, lane = c & 63;
  const T* xb = x + (int64_t)blockIdx.x * NPX * C + c;
  float xin[NPX], acc[NPX];
#pragma unroll
  for (int p = 0; p < NPX; ++p) xin[p] = to_f(xb[p * C]);
  const float bv = bdw[c];
#pragma unroll
  for (int p = 0; p < NPX; ++p) acc[p] = bv;
#pragma unroll
  for (int dy = -(S - 1); dy <= S - 1; ++dy)
#pragma unroll
    for (int dx = -(S - 1); dx <= S - 1; ++dx) {
      const float w = wdw[((dy + 3) * 7 + dx + 3) * C + c];        // out[oy][ox] += x[oy + dy][ox + dx] * w[dy + 3][dx + 3]
#pragma unroll
      for (int oy = 0; oy < S; ++oy)
#pragma unroll
        for (int ox = 0; ox < S; ++ox)
          if (oy + dy >= 0 && oy + dy < S && ox + dx >= 0 && ox + dx < S)
            acc[oy * S + ox] = fmaf(xin[(oy + dy) * S + ox + dx], w, acc[oy * S + ox]);
    }
  // LayerNorm over the C channels of each pixel: mean, then the centred second moment
  float mean[NPX], rstd[NPX];
#pragma unroll
  for (int p = 0; p < NPX; ++p) {
    const float sp = wave_sum(acc[p]);
    if (lane == 0) part[0][wave][p] = sp;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NPX; ++p) {
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += part[0][w][p];
    mean[p] = t * (1.0f / C);
    const float d = acc[p] - mean[p];
    const float qp = wave_sum(d * d);
    if (lane == 0) part[1][wave][p] = qp;
  }
  __syncthreads();
  const float lw = lnw[c], lb = lnb[c];
  T* yb = y + (int64_t)blockIdx.x * NPX * C + c;
#pragma unroll
  for (int p = 0; p < NPX; ++p) {
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += part[1][w][p];
    rstd[p] = rsqrtf(t * (1.0f / C) + eps);
    yb[p * C] = from_f<T>((acc[p] - mean[p]) * rstd[p] * lw + lb);
  }
}

// ------------------------------------------------------------------ K6: LN2d + space-to-depth
// x (nimg,H,W,C) -> out (nimg,H/2,W/2,4C), K index (dy*2+dx)*C + c; floor(H/2): an odd last
// row/col is dropped exactly as Conv2d(k=2,s=2) drops it (7 -> 3 in the 112-px pass).
template <typename T>
__global__ void __launch_bounds__(256) ln_patchify_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bvec, T* __restrict__ out,
                                                          int nimg, int H, int W, int C, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t pix = (int64_t)blockIdx.x * 4 + wave;
  const int64_t total = (int64_t)nimg * H * W;
  if (pix >= total) return;
  const int ix = (int)(pix % W);
  const int64_t t = pix / W;
  const int iy = (int)(t % H);
  const int64_t b = t / H;
  const int Ho = H >> 1, Wo = W >> 1;
  if (iy >= 2 * Ho || ix >= 2 * Wo) return;
  const T* src = x + pix * C;
  float v[12];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const int c = lane + 64 * k;
    v[k] = (c < C) ? to_f(src[c]) : 0.0f;
    s += v[k];
  }
  s = wave_sum(s);
  const float mean = s / (float)C;
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const int c = lane + 64 * k;
    const float d = (c < C) ? v[k] - mean : 0.0f;
    q = fmaf(d, d, q);
  }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q / (float)C + eps);
  T* dst = out + ((b * Ho + (iy >> 1)) * Wo + (ix >> 1)) * 4 * C + ((iy & 1) * 2 + (ix & 1)) * C;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const int c = lane + 64 * k;
    if (c < C) dst[c] = from_f<T>((v[k] - mean) * rstd * w[c] + bvec[c]);
  }
}

// K6 vectorised: C/6 lanes per pixel (16 / 32 / 64 for C = 96 / 192 / 384), each lane owns 6 contiguous
// channels (three 2-channel words for 16-bit storage), so a wave normalises 4 / 2 / 1 pixels with all
// lanes busy and 4-byte accesses instead of one wave per pixel with 2-byte accesses at 75 % lane use.
template <typename T, int C>
__global__ void __launch_bounds__(256) ln_patchify_vec_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bvec, T* __restrict__ out,
                                                              int nimg, int H, int W, float eps) {
  static_assert(sizeof(T) == 2, "vectorised LN-patchify is built for 16-bit storage");
  constexpr int LPP = C / 6;                    // lanes per pixel
  constexpr int PPB = 256 / LPP;                // pixels per block
  const int tid = threadIdx.x;
  const int sub = tid / LPP, l = tid - sub * LPP;
  const int64_t pix = (int64_t)blockIdx.x * PPB + sub;
  const int64_t total = (int64_t)nimg * H * W;
  const bool live = pix < total;
  const int64_t pc = live ? pix : total - 1;
  const int ix = (int)(pc % W);
  const int64_t t = pc / W;
  const int iy = (int)(t % H);
  const int64_t b = t / H;
  const int Ho = H >> 1, Wo = W >> 1;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(x + pc * C + 6 * l);
  float v[6];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint32_t u = src[k];
    v[2 * k] = from_bits16<T>(u & 0xffffu);
    v[2 * k + 1] = from_bits16<T>(u >> 16);
    s += v[2 * k] + v[2 * k + 1];
  }
  s = group_sum<LPP>(s);
  const float mean = s * (1.0f / C);
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < 6; ++k) { const float d = v[k] - mean; q = fmaf(d, d, q); }
  q = group_sum<LPP>(q);
  const float rstd = 1.0f / sqrtf(q * (1.0f / C) + eps);
  if (!live || iy >= 2 * Ho || ix >= 2 * Wo) return;
  uint32_t* dst = reinterpret_cast<uint32_t*>(out + ((b * Ho + (iy >> 1)) * Wo + (ix >> 1)) * 4 * C +
                                              ((iy & 1) * 2 + (ix & 1)) * C + 6 * l);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int c = 6 * l + 2 * k;
    const uint32_t lo = bits16<T>((v[2 * k] - mean) * rstd * w[c] + bvec[c]);
    const uint32_t hi = bits16<T>((v[2 * k + 1] - mean) * rstd * w[c + 1] + bvec[c + 1]);
    dst[k] = lo | (hi << 16);
  }
}

// ------------------------------------------------------------------ generic row LayerNorm (Swin)
template <typename T>
__global__ void __launch_bounds__(256) layernorm_rows_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bvec, T* __restrict__ out,
                                                             int64_t rows, int C, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const T* src = x + row * C;
  float v[24];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 24; ++k) {
    const int c = lane + 64 * k;
    v[k] = (c < C) ? to_f(src[c]) : 0.0f;
    s += v[k];
  }
  s = wave_sum(s);
  const float mean = s / (float)C;
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < 24; ++k) {
    const int c = lane + 64 * k;
    const float d = (c < C) ? v[k] - mean : 0.0f;
    q = fmaf(d, d, q);
  }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q / (float)C + eps);
  T* dst = out + row * C;
#pragma unroll
  for (int k = 0; k < 24; ++k) {
    const int c = lane + 64 * k;
    if (c < C) dst[c] = from_f<T>((v[k] - mean) * rstd * w[c] + bvec[c]);
  }
}

// ------------------------------------------------------------------ K16: Swin window attention
// One wave per (window, head): 49 tokens x head_dim 32.  K and V of the window are staged in LDS as
// fp32; lane i < 49 owns query row i: scores against the 49 keys (+ relative-position bias looked up
// from the (169, nH) table, + the -100 shift mask between regions), fp32 softmax in registers, then
// P.V.  The cyclic shift and the window partition are index arithmetic on the (B,H,W,3C) qkv tensor;
// the result is written back at the token's original position (reverse shift folded in).
template <typename T>
__global__ void __launch_bounds__(64) swin_window_attn_kernel(const T* __restrict__ qkv, const float* __restrict__ rpb,
                                                              T* __restrict__ out, int H, int W, int C, int nH,
                                                              int shift, float scale) {
  __shared__ __attribute__((aligned(16))) float sK[49 * 32];
  __shared__ __attribute__((aligned(16))) float sV[49 * 32];
  const int lane = threadIdx.x;
  const int head = blockIdx.y;
  const int nWx = W / 7, nWy = H / 7;
  const int win = blockIdx.x % (nWx * nWy);
  const int b = blockIdx.x / (nWx * nWy);
  const int wy = win / nWx, wx = win - wy * nWx;
  const bool live = lane < 49;
  const int ty = live ? lane / 7 : 0, tx = live ? lane - (lane / 7) * 7 : 0;
  const int ys = wy * 7 + ty, xs = wx * 7 + tx;                    // coordinates in the shifted frame
  const int yo = (ys + shift) % H, xo = (xs + shift) % W;          // original position of that token
  const int64_t tok = ((int64_t)b * H + yo) * W + xo;
  const T* base = qkv + tok * 3 * C + head * 32;
  float q[32];
  if (live) {
#pragma unroll
    for (int d = 0; d < 32; ++d) {
      q[d] = to_f(base[d]) * scale;
      sK[lane * 32 + d] = to_f(base[C + d]);
      sV[lane * 32 + d] = to_f(base[2 * C + d]);
    }
  } else {
#pragma unroll
    for (int d = 0; d < 32; ++d) q[d] = 0.0f;
  }
  __syncthreads();
  // region id of this token for the shift mask (timm: slices (0,-ws), (-ws,-shift), (-shift,None))
  const int ry = ys < H - 7 ? 0 : (ys < H - shift ? 1 : 2);
  const int rx = xs < W - 7 ? 0 : (xs < W - shift ? 1 : 2);
  const int myreg = ry * 3 + rx;
  float sc[49];
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < 49; ++j) {
    const int jy = j / 7, jx = j % 7;
    float a = 0.0f;
#pragma unroll
    for (int d = 0; d < 32; ++d) a = fmaf(q[d], sK[j * 32 + d], a);
    const int ridx = (ty - jy + 6) * 13 + (tx - jx + 6);
    a += rpb[ridx * nH + head];
    if (shift) {
      const int ysj = wy * 7 + jy, xsj = wx * 7 + jx;
      const int rj = (ysj < H - 7 ? 0 : (ysj < H - shift ? 1 : 2)) * 3 + (xsj < W - 7 ? 0 : (xsj < W - shift ? 1 : 2));
      if (rj != myreg) a -= 100.0f;
    }
    sc[j] = a;
    mx = fmaxf(mx, a);
  }
  float sum = 0.0f;
  float o[32];
#pragma unroll
  for (int d = 0; d < 32; ++d) o[d] = 0.0f;
#pragma unroll
  for (int j = 0; j < 49; ++j) {
    const float p = __expf(sc[j] - mx);
    sum += p;
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = fmaf(p, sV[j * 32 + d], o[d]);
  }
  if (live) {
    const float inv = 1.0f / sum;
    T* dst = out + tok * C + head * 32;
#pragma unroll
    for (int d = 0; d < 32; ++d) dst[d] = from_f<T>(o[d] * inv);
  }
}

// ------------------------------------------------------------------ K16 (16-bit): window attention on MFMA
// One wave per (window, head); the 49 tokens are padded to 64.
//   S^T = K . Q^T   : v_mfma_f32_32x32x16, A = K fragments (rows = keys), B = Q fragments (cols = queries)
//                     -> key on registers, QUERY ON THE LANE (2x2 tiles of 32x32, 8 MFMAs)
//   softmax         : scale, + relative-position bias (LDS table) + shift mask, row max / sum = in-lane over
//                     32 registers + one cross-half exchange; padded keys masked out
//   O^T = V^T . P^T : the probability accumulators are packed to 16-bit and used directly as the B operand
//                     (guide: "an accumulator tile as the next MFMA's operand"); V^T fragments come from an LDS
//                     image [d][key] whose key axis is pre-permuted to the accumulator's k order (8 MFMAs)
// K and Q tiles live in LDS as [token][32] rows (64 B) read with ds_read_b128.
template <typename T>
__global__ void __launch_bounds__(64, 3) swin_window_attn_mfma_kernel(const T* __restrict__ qkv,
                                                                  const float* __restrict__ rpb, T* __restrict__ out,
                                                                  int H, int W, int C, int nH, int shift,
                                                                  float scale) {
  static_assert(sizeof(T) == 2, "MFMA window attention is built for 16-bit storage");
  __shared__ __attribute__((aligned(16))) unsigned short sQ[64 * 32];
  __shared__ __attribute__((aligned(16))) unsigned short sK[64 * 32];
  __shared__ __attribute__((aligned(16))) unsigned short sVt[32 * 64];    // [d][permuted key]
  __shared__ float sB[169];
  const int lane = threadIdx.x;
  const int lr = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y;
  const int nWx = W / 7, nWy = H / 7;
  const int win = blockIdx.x % (nWx * nWy);
  const int b = blockIdx.x / (nWx * nWy);
  const int wy = win / nWx, wx = win - wy * nWx;

  for (int i = lane; i < 169; i += 64) sB[i] = rpb[i * nH + head];
  // ---- stage Q, K rows and V^T (token = lane; padded tokens 49..63 are zero) ----
  {
    const bool live = lane < 49;
    const int ty = live ? lane / 7 : 0, tx = live ? lane - (lane / 7) * 7 : 0;
    const int yo = (wy * 7 + ty + shift) % H, xo = (wx * 7 + tx + shift) % W;
    const int64_t tok = ((int64_t)b * H + yo) * W + xo;
    const T* base = qkv + tok * 3 * C + head * 32;
    u32x4 q4[4], k4[4], v4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      q4[c] = live ? *(const u32x4*)(base + 8 * c) : z;
      k4[c] = live ? *(const u32x4*)(base + C + 8 * c) : z;
      v4[c] = live ? *(const u32x4*)(base + 2 * C + 8 * c) : z;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      *(u32x4*)(sQ + lane * 32 + 8 * c) = q4[c];
      *(u32x4*)(sK + lane * 32 + 8 * c) = k4[c];
    }
    // V^T with the key axis in accumulator order: key = 32 kt + 16 s + 8 a + 4 h + e  ->  32 kt + 16 s + 8 h + 4 a + e
    const int kt = lane >> 5, s = (lane >> 4) & 1, a = (lane >> 3) & 1, hh = (lane >> 2) & 1, e = lane & 3;
    const int pk = 32 * kt + 16 * s + 8 * hh + 4 * a + e;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        sVt[(8 * c + 2 * w) * 64 + pk] = (unsigned short)(v4[c][w] & 0xffffu);
        sVt[(8 * c + 2 * w + 1) * 64 + pk] = (unsigned short)(v4[c][w] >> 16);
      }
  }
  __syncthreads();

  // ---- S^T = K . Q^T ----
  f32x16 sacc[2][2];               // [key tile][query tile]
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kt][qt][r] = 0.0f;
#pragma unroll
  for (int st = 0; st < 2; ++st) {             // d = 16 st + 8 lh .. +7
    u32x4 kf[2], qf[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      kf[t] = *(const u32x4*)(sK + (32 * t + lr) * 32 + 16 * st + 8 * lh);
      qf[t] = *(const u32x4*)(sQ + (32 * t + lr) * 32 + 16 * st + 8 * lh);
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) Mfma<T>::run(kf[kt], qf[qt], sacc[kt][qt]);
  }

  // ---- softmax over keys, per query (query = 32 qt + lr on this lane; keys split over registers and halves) ----
  u32x4 pf[2][2][2];               // [query tile][key tile][k-step] packed probabilities (B operand of P.V)
  float inv_sum[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = 32 * qt + lr;
    const int qy = qi / 7, qx = qi - qy * 7;                    // (garbage for padded queries, never stored)
    const int qys = wy * 7 + qy, qxs = wx * 7 + qx;
    const int qreg = (qys < H - 7 ? 0 : (qys < H - shift ? 1 : 2)) * 3 + (qxs < W - 7 ? 0 : (qxs < W - shift ? 1 : 2));
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kj = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = sacc[kt][qt][r] * scale;
        if (kj < 49 && qi < 49) {
          const int ky = kj / 7, kx = kj - ky * 7;
          v += sB[(qy - ky + 6) * 13 + (qx - kx + 6)];
          if (shift) {
            const int kys = wy * 7 + ky, kxs = wx * 7 + kx;
            const int kreg = (kys < H - 7 ? 0 : (kys < H - shift ? 1 : 2)) * 3 + (kxs < W - 7 ? 0 : (kxs < W - shift ? 1 : 2));
            if (kreg != qreg) v -= 100.0f;
          }
        } else if (kj >= 49) {
          v = -3.0e38f;                                          // padded key: exp() -> 0
        }
        sacc[kt][qt][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.0f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __expf(sacc[kt][qt][r] - mx);
        sacc[kt][qt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    inv_sum[qt] = 1.0f / sum;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint32_t lo = bits16<T>(sacc[kt][qt][8 * s + 2 * w]);
          const uint32_t hi = bits16<T>(sacc[kt][qt][8 * s + 2 * w + 1]);
          pf[qt][kt][s][w] = lo | (hi << 16);
        }
  }

  // ---- O^T = V^T . P^T : rows = d (registers), cols = query (lane) ----
  f32x16 oacc[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[qt][r] = 0.0f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const u32x4 vf = *(const u32x4*)(sVt + lr * 64 + 32 * kt + 16 * s + 8 * lh);   // A: row d = lr, 8 permuted keys
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) Mfma<T>::run(vf, pf[qt][kt][s], oacc[qt]);
    }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = 32 * qt + lr;
    if (qi < 49) {
      const int qy = qi / 7, qx = qi - qy * 7;
      const int yo = (wy * 7 + qy + shift) % H, xo = (wx * 7 + qx + shift) % W;
      T* dst = out + (((int64_t)b * H + yo) * W + xo) * C + head * 32;
      typedef T t4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        t4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = from_f<T>(oacc[qt][4 * g4 + e] * inv_sum[qt]);
        *(t4*)(dst + 8 * g4 + 4 * lh) = o;
      }
    }
  }
}

// ------------------------------------------------------------------ Swin PatchMerging front half
// x (n,H,W,C) -> LN over the 4C concat [x(0::2,0::2), x(1::2,0::2), x(0::2,1::2), x(1::2,1::2)] -> (n,H/2,W/2,4C)
template <typename T>
__global__ void __launch_bounds__(256) patch_merge_ln_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bvec, T* __restrict__ out,
                                                             int nimg, int H, int W, int C, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int Ho = H >> 1, Wo = W >> 1;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= (int64_t)nimg * Ho * Wo) return;
  const int xo = (int)(row % Wo);
  const int64_t t = row / Wo;
  const int yo = (int)(t % Ho);
  const int64_t b = t / Ho;
  const int C4 = 4 * C;
  float v[24];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 24; ++k) {
    const int c4 = lane + 64 * k;
    float val = 0.0f;
    if (c4 < C4) {
      const int part = c4 / C, c = c4 - part * C;       // part = dx*2 + dy
      const int dy = part & 1, dx = part >> 1;
      val = to_f(x[((b * H + 2 * yo + dy) * W + 2 * xo + dx) * C + c]);
    }
    v[k] = val;
    s += val;
  }
  s = wave_sum(s);
  const float mean = s / (float)C4;
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < 24; ++k) {
    const float d = (lane + 64 * k < C4) ? v[k] - mean : 0.0f;
    q = fmaf(d, d, q);
  }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q / (float)C4 + eps);
  T* dst = out + row * C4;
#pragma unroll
  for (int k = 0; k < 24; ++k) {
    const int c4 = lane + 64 * k;
    if (c4 < C4) dst[c4] = from_f<T>((v[k] - mean) * rstd * w[c4] + bvec[c4]);
  }
}

// mean over the L tokens of each image: x (n, L, C) -> (n, C)
template <typename T>
__global__ void __launch_bounds__(256) mean_tokens_kernel(const T* __restrict__ x, T* __restrict__ out, int L, int C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.0f;
    for (int p = 0; p < L; ++p) s += to_f(x[((int64_t)b * L + p) * C + c]);
    out[(int64_t)b * C + c] = from_f<T>(s / (float)L);
  }
}

// ------------------------------------------------------------------ K7: avg-pool + LN(768)
template <typename T>
__global__ void __launch_bounds__(256) pool_ln_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bvec, T* __restrict__ out, int HW,
                                                      int C /*768*/, float eps, int seg_n, int row_stride, int row0) {
  // image b of the launch (its passes one after the other, seg_n images each) -> output row
  // (b % seg_n) * row_stride + row0 + b / seg_n: with row_stride = the network's number of passes the rows of one frame's
  // passes are neighbours, and the classifier GEMM over all of them writes the (B, passes * 1000) feature matrix in one launch
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int orow = (b % seg_n) * row_stride + row0 + b / seg_n;
  const T* src = x + (int64_t)b * HW * C;
  float m[3] = {0.f, 0.f, 0.f};
  for (int p = 0; p < HW; ++p) {
#pragma unroll
    for (int k = 0; k < 3; ++k) m[k] += to_f(src[(int64_t)p * C + tid + 256 * k]);
  }
  const float inv = 1.0f / (float)HW;
#pragma unroll
  for (int k = 0; k < 3; ++k) m[k] *= inv;
  float s = wave_sum(m[0] + m[1] + m[2]);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)C;
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) { const float d = m[k] - mean; q = fmaf(d, d, q); }
  q = wave_sum(q);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((red[4] + red[5] + red[6] + red[7]) / (float)C + eps);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int c = tid + 256 * k;
    out[(int64_t)orow * C + c] = from_f<T>((m[k] - mean) * rstd * w[c] + bvec[c]);
  }
}

// ------------------------------------------------------------------ K1/K9 first conv (Cin = 3)
// POOL=true : Conv2d(3,16,3,s1,p1) -> ReLU -> MaxPool2 (ED encoder layer 1), thread = pooled pixel
// POOL=false: Conv2d(3,16,3,s2,p1) (+folded BN) -> LeakyReLU (VAE encoder layer 1)
// wp: [27][16] fp32 with k = (ky*3+kx)*3 + ci; output NHWC (nimg,Ho,Wo,16).
template <typename T, bool POOL>
__global__ void __launch_bounds__(256) conv3_first_kernel(const T* __restrict__ x, int64_t sb, int64_t sc, int64_t sy,
                                                          int64_t sx, const float* __restrict__ wp,
                                                          const float* __restrict__ bias, T* __restrict__ out,
                                                          int nimg, int H, int W, int act) {
  // 27x16 taps live in LDS and are read as wave-uniform (broadcast) float4s (in SGPRs they overflowed
  // the scalar file: the spill code ran at 3 % of the VALU rate).  The thread's input patch is parked in
  // LDS as [element][thread] so the 27-tap loop can stay a real loop: fully unrolled, hipcc hoists all
  // 432 weight reads ahead of the FMAs and the kernel needs 480 VGPRs.
  constexpr int PW = POOL ? 4 : 3;      // input patch width/height
  constexpr int NPOS = POOL ? 4 : 1;
  __shared__ __attribute__((aligned(16))) float sW[27 * 16];
  __shared__ T sIn[PW * PW * 3][256];      // parked in the storage dtype: 24 KB instead of 48 KB for 16-bit -> twice the workgroups per CU
  const int tid = threadIdx.x;
  for (int i = tid; i < 27 * 16; i += 256) sW[i] = wp[i];
  const int Ho = H >> 1, Wo = W >> 1;
  const int64_t total = (int64_t)nimg * Ho * Wo;
  const int64_t gp = (int64_t)blockIdx.x * 256 + tid;
  const bool live = gp < total;
  const int64_t gpc = live ? gp : total - 1;
  const int xo = (int)(gpc % Wo);
  const int64_t t = gpc / Wo;
  const int yo = (int)(t % Ho);
  const int64_t b = t / Ho;
  const int iy0 = 2 * yo - 1, ix0 = 2 * xo - 1;
#pragma unroll
  for (int r = 0; r < PW; ++r)
#pragma unroll
    for (int s = 0; s < PW; ++s) {
      const int iy = iy0 + r, ix = ix0 + s;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
        sIn[(r * PW + s) * 3 + ci][tid] = ok ? x[b * sb + ci * sc + (int64_t)iy * sy + (int64_t)ix * sx] : from_f<T>(0.0f);
    }
  __syncthreads();
  float acc[NPOS][16];
#pragma unroll
  for (int q = 0; q < NPOS; ++q)
#pragma unroll
    for (int co = 0; co < 16; ++co) acc[q][co] = 0.0f;
#pragma unroll 1
  for (int k = 0; k < 27; ++k) {          // k = (ky*3 + kx)*3 + ci
    const int tap = k / 3, ci = k - tap * 3;
    const int ky = tap / 3, kx = tap - ky * 3;
    const float4* wr = reinterpret_cast<const float4*>(sW + k * 16);
    float w[16];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 v4 = wr[g4];
      w[4 * g4] = v4.x; w[4 * g4 + 1] = v4.y; w[4 * g4 + 2] = v4.z; w[4 * g4 + 3] = v4.w;
    }
#pragma unroll
    for (int q = 0; q < NPOS; ++q) {
      const float v = to_f(sIn[(((q >> 1) + ky) * PW + (q & 1) + kx) * 3 + ci][tid]);
#pragma unroll
      for (int co = 0; co < 16; ++co) acc[q][co] = fmaf(v, w[co], acc[q][co]);
    }
  }
  if (!live) return;
  T* o = out + gp * 16;
#pragma unroll
  for (int co = 0; co < 16; ++co) {
    float r = acc[0][co];
#pragma unroll
    for (int q = 1; q < NPOS; ++q) r = fmaxf(r, acc[q][co]);
    o[co] = from_f<T>(apply_act(r + bias[co], act));
  }
}

// ------------------------------------------------------------------ K1/K9 first conv on the matrix pipe (16-bit storage)
// 16 output channels x 27 taps is one v_mfma_f32_16x16x32 per 16 pixels (K = ci*9 + ky*3 + kx, padded to 32 with zero
// weights): D[channel][pixel], a lane ends up with 4 consecutive channels of one pixel.  A workgroup stages the input
// rows of an 8-row band of one frame in LDS (16-byte pieces, 8-element zero aprons left and right, zero rows outside the
// frame); a lane gathers its 8 patch values with ds_read_u16 from 8 per-lane base addresses (they differ by lane
// group, i.e. by which 8 of the 32 k it feeds) plus compile-time column offsets.
//   POOL  (ED layer 1: conv s1 + ReLU + maxpool 2): the 16 pixels of an MFMA are a 2 x 8 block laid out so that a pool
//         window is a lane quad (two DPP max); four MFMAs = 2 x 32 conv pixels = 16 pooled pixels x 16 channels = 512
//         contiguous output bytes, and quad lane r stores the window of MFMA r: 8 bytes per lane, fully coalesced.
//   !POOL (VAE layer 1: conv s2 + folded BN + LeakyReLU): the 16 pixels are 16 consecutive output columns.
template <typename T> struct Mfma16;
template <> struct Mfma16<half_t> {
  __device__ static __forceinline__ f32x4 run(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mfma16<bf16_t> {
  __device__ static __forceinline__ f32x4 run(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
constexpr int kConv3MaxW = 224;

template <typename T, bool POOL>
__global__ void __launch_bounds__(256) conv3_first_mfma_kernel(const T* __restrict__ x, int64_t sb, int64_t sc, int64_t sy,
                                                               const float* __restrict__ wp, const float* __restrict__ bias,
                                                               T* __restrict__ out, int nimg, int H, int W, int act) {
  static_assert(sizeof(T) == 2, "matrix-pipe first conv is built for 16-bit storage");
  constexpr int RWS = POOL ? 10 : 9;                 // staged input rows of an 8-row band (one apron row above, one below)
  constexpr int MPI = POOL ? 4 : 1;                  // MFMAs per work item
  __shared__ __attribute__((aligned(16))) T sIn[3 * RWS * (kConv3MaxW + 16)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pitch = W + 16;
  const int bands = H >> 3;
  const int img = blockIdx.x / bands, band = blockIdx.x - img * bands;
  const int row0 = 8 * band - 1;
  const int Ho = H >> 1, Wo = W >> 1;
  {   // ---- stage the band: (3 * RWS) rows of (W / 8 + 2) 16-byte pieces
    const int ppr = (W >> 3) + 2;
    const T* xi = x + (int64_t)img * sb;
    for (int e = tid; e < 3 * RWS * ppr; e += 256) {
      const int rr = e / ppr, pc = e - rr * ppr;
      const int ci = rr / RWS, r = rr - ci * RWS;
      const int iy = row0 + r;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (pc >= 1 && pc <= (W >> 3) && iy >= 0 && iy < H)
        v = *reinterpret_cast<const u32x4*>(xi + (int64_t)ci * sc + (int64_t)iy * sy + 8 * (pc - 1));
      *reinterpret_cast<u32x4*>(sIn + rr * pitch + 8 * pc) = v;
    }
  }
  const int g = lane >> 4, j = lane & 15;
  // weights: lane (g, channel j) holds k = 8g .. 8g+7 of that channel;  wp is [27][16] with k = (ky*3 + kx)*3 + ci
  u32x4 wa;
  int off[8];
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    uint32_t pk = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 8 * g + i + h;
      const int ci = k / 9, ky = (k - 9 * ci) / 3, kx = k - 9 * ci - 3 * ky;
      const float wv = k < 27 ? wp[((ky * 3 + kx) * 3 + ci) * 16 + j] : 0.0f;
      pk |= bits16<T>(wv) << (16 * h);
      off[i + h] = k < 27 ? ((ci * RWS + ky) * pitch + kx) : 0;        // elements
    }
    wa[i >> 1] = pk;
  }
  const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + 4 * g);
  const int px = POOL ? (((j >> 1) & 1) * pitch + 2 * (j >> 2) + (j & 1) + 7) : (2 * j + 7);
  const T* ap[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ap[i] = sIn + px + off[i];
  __syncthreads();

  const int ncg = W >> 5;
  for (int it = wave; it < 4 * ncg; it += 4) {
    const int rr = it / ncg, cg = it - rr * ncg;
    const int ioff = 2 * rr * pitch + 32 * cg;
    f32x4 acc[MPI];
#pragma unroll
    for (int m = 0; m < MPI; ++m) {
      uint32_t v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const unsigned short*>(ap[i] + ioff + 8 * m);
      u32x4 pb = {v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
      if (g == 3) { pb[1] &= 0xffffu; pb[2] = 0u; pb[3] = 0u; }        // k = 27 .. 31 do not exist
      acc[m] = Mfma16<T>::run(wa, pb, b4);
    }
    typedef T t4 __attribute__((ext_vector_type(4)));
    if constexpr (POOL) {
      f32x4 sel = acc[0];
#pragma unroll
      for (int m = 0; m < MPI; ++m) {
        f32x4 pm;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v = acc[m][c];
          v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)));
          v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)));
          pm[c] = v;
        }
        if (m == 0) sel = pm;
        else if ((j & 3) == m) sel = pm;
      }
      t4 o4;
#pragma unroll
      for (int c = 0; c < 4; ++c) o4[c] = from_f<T>(apply_act(sel[c], act));
      const int yo = 4 * band + rr, xo = 16 * cg + 4 * (j & 3) + (j >> 2);
      *reinterpret_cast<t4*>(out + (((int64_t)img * Ho + yo) * Wo + xo) * 16 + 4 * g) = o4;
    } else {
      t4 o4;
#pragma unroll
      for (int c = 0; c < 4; ++c) o4[c] = from_f<T>(apply_act(acc[0][c], act));
      const int yo = 4 * band + rr, xo = 16 * cg + j;
      *reinterpret_cast<t4*>(out + (((int64_t)img * Ho + yo) * Wo + xo) * 16 + 4 * g) = o4;
    }
  }
}

// ------------------------------------------------------------------ K2/K12 last ConvTranspose2d 16 -> 3
// x NHWC (nimg,H,W,16) -> out NHWC (nimg,2H,2W,3); wp: [16][2][2][3] fp32 (ci,dy,dx,co).
template <typename T>
__global__ void __launch_bounds__(256) convt2_small_kernel(const T* __restrict__ x, const float* __restrict__ wp,
                                                           const float* __restrict__ bias, T* __restrict__ out,
                                                           int nimg, int H, int W, int act) {
  __shared__ __attribute__((aligned(16))) float sW[16 * 12];
  if (threadIdx.x < 16 * 12) sW[threadIdx.x] = wp[threadIdx.x];
  __syncthreads();
  const int64_t total = (int64_t)nimg * H * W;
  const int64_t gp = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gp >= total) return;
  const int xi = (int)(gp % W);
  const int64_t t = gp / W;
  const int yi = (int)(t % H);
  const int64_t b = t / H;
  const T* src = x + gp * 16;
  float acc[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) acc[j] = bias[j % 3];
  if constexpr (sizeof(T) == 2) {
    // the pixel's 16 channels are 32 contiguous bytes: two 16-byte loads (sixteen 2-byte loads, 32 bytes apart from lane to
    // lane, and twelve 2-byte stores per thread made this launch 45 us for 90 MB), then four passes over four channels each
    // (a real loop: unrolled, hipcc keeps all 192 weights in registers); the pass's two dwords come out of the eight by select
    const u32x4 lo = *reinterpret_cast<const u32x4*>(src), hi = *reinterpret_cast<const u32x4*>(src + 8);
#pragma unroll 1
    for (int h = 0; h < 4; ++h) {
      const uint32_t d0 = h == 0 ? lo[0] : (h == 1 ? lo[2] : (h == 2 ? hi[0] : hi[2]));
      const uint32_t d1 = h == 0 ? lo[1] : (h == 1 ? lo[3] : (h == 2 ? hi[1] : hi[3]));
      const float v[4] = {from_bits16<T>(d0 & 0xffffu), from_bits16<T>(d0 >> 16), from_bits16<T>(d1 & 0xffffu), from_bits16<T>(d1 >> 16)};
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const float4* wr = reinterpret_cast<const float4*>(sW + (4 * h + c4) * 12);
#pragma unroll
        for (int g4 = 0; g4 < 3; ++g4) {
          const float4 w4 = wr[g4];
          acc[4 * g4] = fmaf(v[c4], w4.x, acc[4 * g4]);
          acc[4 * g4 + 1] = fmaf(v[c4], w4.y, acc[4 * g4 + 1]);
          acc[4 * g4 + 2] = fmaf(v[c4], w4.z, acc[4 * g4 + 2]);
          acc[4 * g4 + 3] = fmaf(v[c4], w4.w, acc[4 * g4 + 3]);
        }
      }
    }
    // an output row of this pixel is 2 pixels x 3 channels = 12 contiguous bytes (4-byte aligned: 12 xi + 12 W rows): one
    // 12-byte store per row, consecutive lanes write consecutive pieces
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      T* o = out + ((b * 2 * H + 2 * yi + dy) * 2 * W + 2 * xi) * 3;
      uint32_t pk[3];
#pragma unroll
      for (int j = 0; j < 3; ++j)
        pk[j] = bits16<T>(apply_act(acc[dy * 6 + 2 * j], act)) | (bits16<T>(apply_act(acc[dy * 6 + 2 * j + 1], act)) << 16);
      typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
      *reinterpret_cast<u32x3*>(o) = u32x3{pk[0], pk[1], pk[2]};
    }
  } else {
#pragma unroll 1
    for (int ci = 0; ci < 16; ++ci) {     // a real loop: unrolled, hipcc keeps all 192 weights in registers
      const float v = to_f(src[ci]);
      const float4* wr = reinterpret_cast<const float4*>(sW + ci * 12);
#pragma unroll
      for (int g4 = 0; g4 < 3; ++g4) {
        const float4 w4 = wr[g4];
        acc[4 * g4] = fmaf(v, w4.x, acc[4 * g4]);
        acc[4 * g4 + 1] = fmaf(v, w4.y, acc[4 * g4 + 1]);
        acc[4 * g4 + 2] = fmaf(v, w4.z, acc[4 * g4 + 2]);
        acc[4 * g4 + 3] = fmaf(v, w4.w, acc[4 * g4 + 3]);
      }
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      T* o = out + ((b * 2 * H + 2 * yi + dy) * 2 * W + 2 * xi) * 3;
#pragma unroll
      for (int j = 0; j < 6; ++j) o[j] = from_f<T>(apply_act(acc[dy * 6 + j], act));
    }
  }
}

// ------------------------------------------------------------------ K10/K11 split-K reduce + reparam
// mu[b][n] = sum_s partial[s][b][n] + bias[n];  z = eps*exp(0.5*mu) + mu  (reference quirk:
// std from mu, model/genconvit_vae.py:45).  z is written in NHWC order for the decoder's
// Unflatten(256,7,7): n = c*49 + hw  ->  hw*256 + c.
template <typename T>
__global__ void __launch_bounds__(256) reparam_kernel(const float* __restrict__ partial, int splitk,
                                                      const float* __restrict__ bias, const float* __restrict__ eps,
                                                      float* __restrict__ mu_out, T* __restrict__ z_nhwc, int B,
                                                      int N /*12544*/) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * N;
  if (i >= total) return;
  const int n = (int)(i % N);
  const int64_t b = i / N;
  float mu = bias[n];
  for (int s = 0; s < splitk; ++s) mu += partial[(int64_t)s * total + i];
  if (mu_out) mu_out[i] = mu;
  const float z = fmaf(eps[i], expf(0.5f * mu), mu);
  const int c = n / 49, hw = n - c * 49;
  z_nhwc[b * N + hw * 256 + c] = from_f<T>(z);
}

// rowsum[b] = sum_n (1 + var - mu^2 - exp(var)), var from split-K slabs (genconvit_vae.py:58)
static __global__ void __launch_bounds__(256) kl_rows_kernel(const float* __restrict__ partial, int splitk,
                                                      const float* __restrict__ bias, const float* __restrict__ mu,
                                                      float* __restrict__ rowsum, int B, int N) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t total = (int64_t)B * N;
  float s = 0.0f;
  for (int n = tid; n < N; n += 256) {
    const int64_t i = (int64_t)b * N + n;
    float var = bias[n];
    for (int k = 0; k < splitk; ++k) var += partial[(int64_t)k * total + i];
    const float m = mu[i];
    s += 1.0f + var - m * m - expf(var);
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) rowsum[b] = red[0] + red[1] + red[2] + red[3];
}

// kl = 0.5 * mean_b(-0.5 * rowsum[b])   (kl_weight = 0.5, genconvit_vae.py:40,58)
static __global__ void kl_finish_kernel(const float* __restrict__ rowsum, float* __restrict__ kl, int B) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.0f;
    for (int b = 0; b < B; ++b) s += -0.5f * rowsum[b];
    *kl = 0.5f * s / (float)B;
  }
}

// ------------------------------------------------------------------ K8 tail: 500 -> 2
template <typename T>
__global__ void __launch_bounds__(256) head_tail_kernel(const T* __restrict__ h, const float* __restrict__ w /*[2][K]*/,
                                                        const float* __restrict__ bias, float* __restrict__ logits,
                                                        int B, int K) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  float a0 = 0.0f, a1 = 0.0f;
  for (int k = lane; k < K; k += 64) {
    const float v = to_f(h[(int64_t)b * K + k]);
    a0 = fmaf(v, w[k], a0);
    a1 = fmaf(v, w[K + k], a1);
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  if (lane == 0) {
    logits[2 * b] = a0 + bias[0];
    logits[2 * b + 1] = a1 + bias[1];
  }
}

// K8 with the hidden layer's split-K reduce folded in: h = act(sum_s partial[s] + b1) (rounded to T, as the stored hidden
// activation was), logits = h . W2^T + b2.  The 2000 -> 500 layer of <= 512 rows is a chain of 32 K tiles on a handful of
// workgroups when it runs as one GEMM (37 us at 128 rows); split eight ways it is 4 tiles deep, and its reduction costs
// nothing here.  (model/genconvit_ed.py:87, model/genconvit_vae.py:114: fc2(act(fc(act(x)))).)
template <typename T, int ACT>
__global__ void __launch_bounds__(256) head_tail_splitk_kernel(const float* __restrict__ partial, int S,
                                                               const float* __restrict__ b1, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ logits,
                                                               int B, int K) {
  // one workgroup per row: every thread owns hidden units tid, tid + 256, .. and sums their S partials with independent
  // loads (a wave per row walked 8 x 8 dependent loads: 22 us for 128 rows)
  __shared__ float red[2][4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float a0 = 0.0f, a1 = 0.0f;
  for (int k = tid; k < K; k += 256) {
    float v = b1[k];
    const float* p = partial + (int64_t)b * K + k;
#pragma unroll 8
    for (int s = 0; s < S; ++s) v += p[(int64_t)s * B * K];
    v = to_f(from_f<T>(act_fn<ACT>(v)));
    a0 = fmaf(v, w[k], a0);
    a1 = fmaf(v, w[K + k], a1);
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  if ((tid & 63) == 0) { red[0][tid >> 6] = a0; red[1][tid >> 6] = a1; }
  __syncthreads();
  if (tid < 2) logits[2 * b + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3] + bias[tid];
}

// ------------------------------------------------------------------ K13/K14 bilinear x2 + MSE
// x_hat NHWC (B,112,112,3) -> recon NCHW (B,3,224,224) (nullable) ; msepart[b][blk] partial sums of
// (recon - img)^2 over the block's pixels (nullable), img NCHW (B,3,224,224).
template <typename T>
__global__ void __launch_bounds__(256) resize_mse_kernel(const T* __restrict__ xhat, const T* __restrict__ img,
                                                         T* __restrict__ recon, float* __restrict__ msepart) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;           // 224*224 = 196 * 256
  const int oy = pix / 224, ox = pix - oy * 224;
  // align_corners=False, scale 0.5: src = (dst + 0.5) * 0.5 - 0.5, clamped below at 0
  float sy = fmaxf((oy + 0.5f) * 0.5f - 0.5f, 0.0f), sx = fmaxf((ox + 0.5f) * 0.5f - 0.5f, 0.0f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = min(y0 + 1, 111), x1 = min(x0 + 1, 111);
  const float ly = sy - y0, lx = sx - x0;
  const T* base = xhat + (int64_t)b * 112 * 112 * 3;
  float err = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v00 = to_f(base[(y0 * 112 + x0) * 3 + c]), v01 = to_f(base[(y0 * 112 + x1) * 3 + c]);
    const float v10 = to_f(base[(y1 * 112 + x0) * 3 + c]), v11 = to_f(base[(y1 * 112 + x1) * 3 + c]);
    const float top = v00 + (v01 - v00) * lx, bot = v10 + (v11 - v10) * lx;
    const float v = top + (bot - top) * ly;
    const int64_t o = (((int64_t)b * 3 + c) * 224 + oy) * 224 + ox;
    if (recon) recon[o] = from_f<T>(v);
    if (msepart) { const float d = v - to_f(img[o]); err = fmaf(d, d, err); }
  }
  if (msepart) {
    err = wave_sum(err);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = err;
    __syncthreads();
    if (threadIdx.x == 0) msepart[b * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}

static __global__ void __launch_bounds__(256) mse_finish_kernel(const float* __restrict__ msepart, float* __restrict__ mse,
                                                         int nblk, float inv_count) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float s = 0.0f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += msepart[b * nblk + i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) mse[b] = (red[0] + red[1] + red[2] + red[3]) * inv_count;
}

// ------------------------------------------------------------------ N1: fused preprocess_frame
// uint8 NHWC (n,224,224,3) -> normalised NCHW in T: ((u8 / 255) - mean) / std, same op order as the reference
// (model/pred_func.py:95-108: .float(), / 255.0, then transforms.Normalize of dataset/loader.py:63-65,77).
template <typename T>
__global__ void __launch_bounds__(256) preprocess_kernel(const unsigned char* __restrict__ u8, T* __restrict__ out,
                                                         int64_t npix_total, int hw) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // pixel index over n*H*W
  if (i >= npix_total) return;
  const int64_t n = i / hw;
  const int p = (int)(i - n * hw);
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = (float)u8[i * 3 + c] / 255.0f;
    out[(n * 3 + c) * hw + p] = from_f<T>((v - mean[c]) / stdv[c]);
  }
}

// ------------------------------------------------------------------ K15 vote: mean over rows of sigmoid
static __global__ void __launch_bounds__(256) vote_kernel(const float* __restrict__ logits, int rows,
                                                   float* __restrict__ mean2) {
  __shared__ float red[8];
  float s0 = 0.0f, s1 = 0.0f;
  for (int r = threadIdx.x; r < rows; r += 256) {
    s0 += 1.0f / (1.0f + expf(-logits[2 * r]));
    s1 += 1.0f / (1.0f + expf(-logits[2 * r + 1]));
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s0; red[4 + (threadIdx.x >> 6)] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mean2[0] = (red[0] + red[1] + red[2] + red[3]) / (float)rows;
    mean2[1] = (red[4] + red[5] + red[6] + red[7]) / (float)rows;
  }
}

// ------------------------------------------------------------------ N3: per-video vote over a batched forward
// Several videos' frames are concatenated into one batch; logits rows are [net0 frames 0..B-1; net1 frames ...]
// (model/genconvit.py:74).  Video v owns frames [off[v], off[v+1]): mean2[v][c] = mean over its frames and nets of
// sigmoid(logit[.][c])  == max_prediction_value's mean(dim=0) applied per video (model/pred_func.py:120,125).
static __global__ void __launch_bounds__(64) vote_segments_kernel(const float* __restrict__ logits, int B, int nets,
                                                                  const int* __restrict__ off, float* __restrict__ mean2) {
  const int v = blockIdx.x, lane = threadIdx.x;
  const int lo = off[v], hi = off[v + 1];
  float s0 = 0.0f, s1 = 0.0f;
  for (int n = 0; n < nets; ++n)
    for (int f = lo + lane; f < hi; f += 64) {
      const int r = n * B + f;
      s0 += 1.0f / (1.0f + expf(-logits[2 * r]));
      s1 += 1.0f / (1.0f + expf(-logits[2 * r + 1]));
    }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (lane == 0) {
    const float cnt = (float)(nets * (hi - lo));
    mean2[2 * v] = cnt > 0 ? s0 / cnt : 0.5f;
    mean2[2 * v + 1] = cnt > 0 ? s1 / cnt : 0.5f;
  }
}

}  // namespace gcv
