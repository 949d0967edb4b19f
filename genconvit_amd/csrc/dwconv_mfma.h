// K4 (round 3): depthwise 7x7 + LayerNorm with the taps on the MATRIX pipe, 16-bit storage, C = 96 / 192.
//
// Same operation and call sites as dwconv_roll.h (timm ConvNeXtBlock conv_dw + LayerNorm; reference
// model/genconvit_ed.py:82-83, model/genconvit_vae.py:111-112), same workgroup shape (a band of rows of one image over its
// whole width, walked down one input row per step, one barrier per row, staging / LayerNorm waves unchanged), but the 343
// FMAs per thread and row are gone: the tap waves issue v_mfma_f32_4x4x4_16B_{f16,bf16}.
//
// Why that instruction.  A depthwise convolution has no contraction over channels, so on the big MFMA shapes a channel
// can only be a row or column of a block-diagonal operand (1/16 or 1/32 of the pipe).  The 4x4x4 form computes SIXTEEN
// independent 4x4x4 products per instruction: a block is a channel.  Within a block the horizontal taps are a banded
// (Toeplitz) 4 x 12 matrix: four neighbouring outputs x = 4 xb + m need the inputs x' = 4 xb - 4 .. 4 xb + 7 (12 values, 10
// of them used), i.e. three k-steps of four:
//     A[m][k] (k-step ks, tap row ky) = w[ky][4 ks + k - m - 1]  (0 outside 0 .. 6)      2 VGPRs per (ky, ks): 42 per 16 channels
//     B[k][n]                         = in[row][4 (xb0 + n + ks) - 4 + k]  for the four x blocks xb0 + n of the task
//     D[m][n]                         = lane (channel, n), register m: four consecutive outputs of one channel
// 28 of the 48 MACs of a block row are real taps (58 %); the instruction retires in 8 cycles (profiles/micro/mfma4x4_probe:
// 128 MAC / cycle / SIMD against 16 - 27 for v_fma_f32), so a row of 56 pixels x 96 channels is 504 MFMAs = 1008 cycles
// per SIMD where the VALU kernel spends ~3100 cycles on FMAs alone.
//
// Operands.  The input ring stays NHWC but 16-bit (the staging waves copy 16-byte pieces as two 8-byte LDS writes, no
// widening), with a pixel pitch of 2 C + 8 bytes: a lane gathers the four x' of a B operand with four ds_read_b32 (immediate
// offsets k * pitch; two neighbouring channels share a dword) and two v_perm_b32, and the pitch puts the four x blocks of a
// half wave on different banks.  Walking DOWN the band, an input row is read ONCE and feeds the seven output rows that are
// still open (seven accumulator sets per task, rotated at compile time like dwconv_roll.h's): 12 gathers for 42 MFMAs per
// wave and row, task 0's operands of the NEXT row fetched under task 1's MFMAs (with every wave gathering right behind
// the barrier the matrix pipe idled for the 1100+ cycles the 288 LDS reads of a step take).  Taps are rounded to the storage
// dtype (they are an MFMA operand); accumulation, bias and LayerNorm are fp32 as before.  fp32 storage keeps dwconv_roll.h.
//
// Measured (profiles/r03_micro/dw_mfma_measurements.txt): 144 -> 108 us at 256 images in
// the single-kernel loop, 85 -> 80 us per launch inside the step, + 1 % whole-step throughput.  A workgroup's set-up (ring
// zeroing, 84 tap loads per lane) is only paid back by bands of 14 rows and more: the launcher keeps shorter bands (batches
// of 32) on the VALU kernel.  Registers are the limit: 42 tap operands + 56 accumulators + 12 gathered operands leave no
// slack under the 128 of a 16-wave workgroup (a straight-line interior path spilled 220), and C = 384 would need 84 tap
// operands per wave.
#pragma once
#include "dwconv_roll.h"

namespace gcv {


template <typename T, int C, int NS> struct DwMfmaLds {
  static constexpr int W = 7 * NS;
  static constexpr int NXG = (W + 15) / 16;            // x groups of 16 outputs (the last one may be partly empty)
  static constexpr int RP = 16 * NXG + 8;              // staged pixels per row: 4 zero pixels, W data, zeros up to 16 NXG + 7
  static constexpr int PITCH = 2 * C + 8;              // bytes per staged pixel.  A gather (ds_read_u16, 32 banks) touches, per
                                                       // half wave, 8 channels (4 dwords) of the four x blocks 4 pixels apart:
                                                       // 4 PITCH bytes = 8 banks mod 32 puts them on banks 0-3 / 8-11 / 16-19 / 24-27
                                                       // (2 C + 16 left a 2-way conflict; the price is 8-byte staging writes)
  static constexpr int SLOT = RP * PITCH;
  static constexpr int IN_BYTES = 3 * SLOT;
  static constexpr int SP = C + 4;                     // floats per pixel of a finished row (the four x blocks of a quad of lanes
                                                       // write 4 pixels apart: 4 SP = 16 banks mod 32, a free 2-way conflict)
  static constexpr int SVAL_BYTES = 2 * W * SP * 4;
  static constexpr int bytes = IN_BYTES + SVAL_BYTES;
  static constexpr int NTASK = (C / 16) * NXG;         // (16-channel group, x group)
  static constexpr int NMW = 12;                       // MFMA waves
  static constexpr int TPW = NTASK / NMW;              // tasks per wave; a wave's tasks share their channel group
  static constexpr int NCONV = NMW * 64;
  static constexpr int NLN = 256;                      // LayerNorm / staging threads
  static constexpr int NT = NCONV + NLN;
  static_assert(NTASK % NMW == 0 && TPW >= 1 && NXG % TPW == 0, "a wave's tasks are x groups of one channel group");
  static_assert(PITCH % 8 == 0 && PITCH % 32 == 8, "pitch");
};

typedef float dwm_f32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct Mfma4;
template <> struct Mfma4<half_t> {
  typedef _Float16 v4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ dwm_f32x4 run(uint2 a, uint2 b, dwm_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(v4, a), __builtin_bit_cast(v4, b), c, 0, 0, 0);
  }
};
template <> struct Mfma4<bf16_t> {
  typedef short v4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ dwm_f32x4 run(uint2 a, uint2 b, dwm_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(v4, a), __builtin_bit_cast(v4, b), c, 0, 0, 0);
  }
};

template <typename T, int C, int NS>
__global__ void __launch_bounds__((DwMfmaLds<T, C, NS>::NT), 4)
dwconv7_ln_mfma_kernel(const T* __restrict__ x, const float* __restrict__ wdw /*[49][C]*/,
                       const float* __restrict__ bdw, const float* __restrict__ lnw, const float* __restrict__ lnb,
                       T* __restrict__ y, int H, int band_rows, int nbands, float eps) {
  static_assert(sizeof(T) == 2, "16-bit storage (fp32 storage: dwconv_roll.h)");
  typedef DwElem<T> EL;
  typedef DwMfmaLds<T, C, NS> LY;
  constexpr int NCONV = LY::NCONV, NLN = LY::NLN;
  constexpr int W = LY::W, P = W, NXG = LY::NXG, TPW = LY::TPW, PITCH = LY::PITCH, SLOT = LY::SLOT, SP = LY::SP;
  constexpr int EPC = EL::EPC;
  extern __shared__ __attribute__((aligned(16))) unsigned char dwm_lds[];
  unsigned char* const in_ring = dwm_lds;
  float* const sval_ring = reinterpret_cast<float*>(dwm_lds + LY::IN_BYTES);

  const int tid = threadIdx.x;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int band = wg % nbands, img = wg / nbands;
  const int ob = band * band_rows;
  const int nrows = min(band_rows, H - ob);
  const int r0 = ob - 3;
  const int nit = nrows + 6;
  const int it0 = max(0, -r0);
  constexpr int row_bytes = W * C * (int)sizeof(T);
  const int64_t img_elems = (int64_t)H * W * C;
  auto row_ok = [&](int it) { const int r = r0 + it; return it < nit && r >= 0 && r < H; };

  // the whole ring is zeroed once: the pad pixels are never written again
  for (int i = tid; i < LY::IN_BYTES / 16; i += LY::NT) reinterpret_cast<u32x4*>(in_ring)[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();

  if (tid < NCONV) {
    // ================================================================== MFMA waves
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = wave * TPW;                         // first task of this wave
    const int cg = t0 / NXG, xg0 = t0 % NXG;
    const int cb = lane >> 2, n = lane & 3;            // block (channel of the group) / x block of the group (B, D) or m (A)
    const int c = 16 * cg + cb;
    // Toeplitz tap operands of this channel: A[ky][ks], lane (channel, m = n)
    uint2 ta[7][3];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        typedef T t4 __attribute__((ext_vector_type(4)));
        t4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int dx = 4 * ks + k - n - 1;
          v[k] = from_f<T>(dx >= 0 && dx < 7 ? wdw[(ky * 7 + dx) * C + c] : 0.0f);
        }
        ta[ky][ks] = __builtin_bit_cast(uint2, v);
      }
    const float bv = bdw[c];
    dwm_f32x4 acc[7][TPW];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[i][t] = dwm_f32x4{bv, bv, bv, bv};
    // B gather base: pixel 4 (4 xg0 + n) of the padded row, channel c; task, k-step and k are compile-time offsets
    // (dword reads: two neighbouring channels share a dword and a v_perm_b32 picks this lane's halves of two pixels;
    //  ds_read_u16 gathers took 1100 - 2800 cycles per step for 288 wave instructions, profiles/dw_stamps.py)
    const unsigned char* const in_base = in_ring + (16 * xg0 + 4 * n) * PITCH + 2 * (c & ~1);
    const uint32_t psel = (c & 1) ? 0x07060302u : 0x05040100u;
    float* const sv_base = sval_ring + (16 * xg0 + 4 * n) * SP + c;
    GCV_LDS_BARRIER();                                 // P1: rows it0 and it0 + 1 are staged
    int cslot = 0;                                     // ring slot of input row `it`
    // B operands of task t from the row at `ra`: 3 k-steps x 4 pixels (dword reads: two neighbouring channels share a
    // dword, v_perm_b32 picks this lane's halves of two pixels)
    auto gather = [&](const unsigned char* ra, const int t, uint2 (&bo)[3]) {
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        uint32_t e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = *reinterpret_cast<const uint32_t*>(ra + (16 * t + 4 * ks + k) * PITCH);
        bo[ks] = uint2{__builtin_amdgcn_perm(e[1], e[0], psel), __builtin_amdgcn_perm(e[3], e[2], psel)};
      }
    };
    uint2 b0[3] = {};
    if (it0 < nit && r0 + it0 >= 0 && r0 + it0 < H) gather(in_base, 0, b0);
    for (int base = 0; base < nit; base += 7) {
#pragma unroll
      for (int ph = 0; ph < 7; ++ph) {
        const int it = base + ph;
        if (it >= it0 && it < nit) {
          const int r = r0 + it;
          const bool rvalid = r >= 0 && r < H;
#define DWM_STAMP3(a, b, c) DW_STAMP(tid == 0 && it == GCV_DW_STAMP_IT, a); DW_STAMP(tid == 320 && it == GCV_DW_STAMP_IT, b); DW_STAMP(tid == NCONV - 64 && it == GCV_DW_STAMP_IT, c)
          DWM_STAMP3(0, 16, 21);
          const unsigned char* const ra = in_base + cslot * SLOT;
          cslot = (cslot == 2) ? 0 : cslot + 1;
          // Operands: task 0's were gathered during the PREVIOUS step (b0: the row has been in the ring since the step
          // before that), the other tasks' are gathered now and land under task 0's MFMAs; then task 0's operands of the
          // NEXT row are gathered under the last task's MFMAs.  (With every wave gathering right behind the barrier the
          // matrix pipe sat idle for the 1100+ cycles the 288 LDS reads of a step take, profiles/dw_stamps.py.)
          const bool nvalid = it + 1 < nit && r + 1 >= 0 && r + 1 < H;
          const unsigned char* const rn = in_base + cslot * SLOT;      // cslot already points at row it + 1
          uint2 bt[3];
          if (TPW > 1 && rvalid && !(GCV_DWM_ABLATE & 4)) gather(ra, 1, bt);
          float* const sv = sv_base + (it & 1) * (P * SP);
          auto finish_row = [&](const int t) {         // output row it - 6 of task t -> LDS, its accumulator back to the bias
            const int sd = (ph + 1) % 7;
#pragma unroll
            for (int m = 0; m < 4; ++m)
              if (16 * (xg0 + t) + 4 * n + m < W) sv[(16 * t + m) * SP] = acc[sd][t][m];
            acc[sd][t] = dwm_f32x4{bv, bv, bv, bv};
          };
          auto mf = [&](const int ky, const int ks, const uint2 (&bo)[3], const int t) {
            const int slot = (ph - ky + 7) % 7;        // compile-time after unrolling
            if (!(GCV_DWM_ABLATE & 1)) acc[slot][t] = Mfma4<T>::run(ta[ky][ks], bo[ks], acc[slot][t]);
            else asm volatile("" ::"v"(bo[ks]), "v"(ta[ky][ks]));
          };
          auto taps = [&](const uint2 (&bo)[3], const int t) {
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
              const int oi = it - ky;                // tap row ky of input row `it` feeds output row oi (relative to the band)
              if (oi >= 0 && oi < nrows) {
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) mf(ky, ks, bo, t);
              }
            }
          };
          if (rvalid && !(GCV_DWM_ABLATE & 4)) {
            taps(b0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (TPW == 1 && nvalid) gather(rn, 0, b0);
            if (TPW > 1) {
              static_assert(TPW <= 2, "one prefetched and one in-step task");
              if (nvalid) gather(rn, 0, b0);
              __builtin_amdgcn_sched_barrier(0);
              taps(bt, 1);
            }
          } else if (nvalid && !(GCV_DWM_ABLATE & 4)) {
            gather(rn, 0, b0);
          }
#if GCV_DW_STAMPS
#pragma unroll
          for (int i = 0; i < 7; ++i) asm volatile("" : "+v"(acc[i][0]));
          DWM_STAMP3(2, 18, 23);
#endif
          if (it >= 6) {                               // input row r completes output row r - 3 (tap row 6)
#pragma unroll
            for (int t = 0; t < TPW; ++t) finish_row(t);
          }
          DWM_STAMP3(3, 19, 24);
          GCV_LDS_BARRIER();
          DWM_STAMP3(4, 20, 25);
#if GCV_DW_STAMPS
          if (tid == 0 && blockIdx.x < 64 && (it == GCV_DW_STAMP_IT || it == GCV_DW_STAMP_IT + 10))
            gcv_dw_stamps[blockIdx.x * 32 + (it == GCV_DW_STAMP_IT ? 5 : 7)] = __builtin_amdgcn_s_memrealtime();
          DW_STAMP(tid == 0 && it == GCV_DW_STAMP_IT + 10, 6);
#endif
        }
      }
    }
  } else {
    // ================================================================== staging + LayerNorm waves (as in dwconv_roll.h)
    __builtin_amdgcn_s_setprio(3);
    const int lid = tid - NCONV;
    constexpr int L = C / 24;                          // lanes per pixel: 6 pieces of 4 channels each
    static_assert(L == 4 || L == 8, "C = 96 / 192");
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(
        y + (int64_t)img * img_elems, 0, (int)(img_elems * (int64_t)sizeof(T)), 0x00020000);
    const int p = lid / L, g = lid - p * L;
    const bool ln_on = lid < P * L;
    float lwv[24], lbv[24];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float4 a = *reinterpret_cast<const float4*>(lnw + 4 * (g + L * j));
      const float4 b = *reinterpret_cast<const float4*>(lnb + 4 * (g + L * j));
      lwv[4 * j] = a.x; lwv[4 * j + 1] = a.y; lwv[4 * j + 2] = a.z; lwv[4 * j + 3] = a.w;
      lbv[4 * j] = b.x; lbv[4 * j + 1] = b.y; lbv[4 * j + 2] = b.z; lbv[4 * j + 3] = b.w;
    }
    // staging: row it + 3 is loaded during step it and written, as it is, into ring slot (it + 2) % 3 during step it + 1
    // (a whole step for the load to land, two barriers before the row is read); 8-byte LDS writes, the pitch is 8 mod 16
    constexpr int PPP = C / EPC;                       // 16-byte pieces per pixel
    constexpr int ROWP = P * PPP;                      // ... of one image row
    constexpr int NPT = (ROWP + NLN - 1) / NLN;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(x + (int64_t)img * img_elems), 0, (int)(img_elems * (int64_t)sizeof(T)), 0x00020000);
    u32x4 stg[NPT];
    int soff[NPT];                                     // ring offset of piece k of this thread: pixel 4 + q / PPP, piece q % PPP
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int q = lid + k * NLN;
      soff[k] = (4 + q / PPP) * PITCH + (q % PPP) * 16;
    }
    auto stage_load = [&](int it) {
      const int row = r0 + it;
#pragma unroll
      for (int k = 0; k < NPT; ++k)
        if (NPT * NLN == ROWP || lid + k * NLN < ROWP)
          stg[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (lid + k * NLN) * 16, row * row_bytes, 0));
    };
    auto stage_write = [&](int slot) {
#pragma unroll
      for (int k = 0; k < NPT; ++k)
        if (NPT * NLN == ROWP || lid + k * NLN < ROWP) {
          unsigned char* d = in_ring + slot * SLOT + soff[k];
          *reinterpret_cast<uint2*>(d) = uint2{stg[k][0], stg[k][1]};
          *reinterpret_cast<uint2*>(d + 8) = uint2{stg[k][2], stg[k][3]};
        }
    };
    auto ln_row = [&](int orow, int slot) {
      if (!ln_on || (GCV_DWM_ABLATE & 2)) return;
      const float* sv = sval_ring + slot * (P * SP) + p * SP + 4 * g;
      float v[24];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float4 a = *reinterpret_cast<const float4*>(sv + 4 * L * j);
        v[4 * j] = a.x; v[4 * j + 1] = a.y; v[4 * j + 2] = a.z; v[4 * j + 3] = a.w;
      }
#if GCV_DW_STAMPS
#pragma unroll
      for (int e = 0; e < 24; ++e) asm volatile("" : "+v"(v[e]));
      DW_STAMP(lid == 0 && orow == ob + GCV_DW_STAMP_IT - 7, 11);
#endif
      float s0 = 0.0f, s1 = 0.0f, q0 = 0.0f, q1 = 0.0f;
#pragma unroll
      for (int e = 0; e < 24; e += 2) {
        s0 += v[e]; s1 += v[e + 1];
        q0 = fmaf(v[e], v[e], q0); q1 = fmaf(v[e + 1], v[e + 1], q1);
      }
      const float mean = dw_group_sum<L>(s0 + s1) * (1.0f / C);
      const float ex2 = dw_group_sum<L>(q0 + q1) * (1.0f / C);
      const float rstd = __builtin_amdgcn_rsqf(fmaxf(fmaf(-mean, mean, ex2), 0.0f) + eps);
      const float nmr = -mean * rstd;
#if GCV_DW_STAMPS
      asm volatile("" ::"v"(nmr), "v"(rstd));
      DW_STAMP(lid == 0 && orow == ob + GCV_DW_STAMP_IT - 7, 12);
#endif
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      u32x2 pk[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(fmaf(v[4 * j + e], rstd, nmr), lwv[4 * j + e], lbv[4 * j + e]);
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef T t16x2 __attribute__((ext_vector_type(2)));
        const t16x2 lo = __builtin_convertvector(f32x2{o[0], o[1]}, t16x2), hi = __builtin_convertvector(f32x2{o[2], o[3]}, t16x2);
        pk[j] = u32x2{__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
      }
      __builtin_amdgcn_sched_barrier(0);               // every piece finished before the first store (dwconv_roll.h)
#pragma unroll
      for (int j = 0; j < 6; ++j)
        __builtin_amdgcn_raw_buffer_store_b64(pk[j], rsy, (p * C + 4 * (g + L * j)) * (int)sizeof(T), orow * row_bytes, 0);
      __builtin_amdgcn_sched_barrier(0);
    };

    if (row_ok(it0)) { stage_load(it0); stage_write(0); }
    if (row_ok(it0 + 1)) { stage_load(it0 + 1); stage_write(1); }
    if (row_ok(it0 + 2)) stage_load(it0 + 2);
    GCV_LDS_BARRIER();                                 // P1
    int wslot = 2;                                     // ring slot row it + 2 goes to ((it - it0 + 2) % 3)
    for (int it = it0; it < nit; ++it) {
      if (row_ok(it + 2)) stage_write(wslot);
      wslot = (wslot == 2) ? 0 : wslot + 1;
      if (row_ok(it + 3)) stage_load(it + 3);
      DW_STAMP(lid == 0 && it == GCV_DW_STAMP_IT, 8); DW_STAMP(lid == 0 && it == GCV_DW_STAMP_IT, 9); DW_STAMP(lid == 0 && it == GCV_DW_STAMP_IT, 10);
      DW_STAMP(lid == NLN - 64 && it == GCV_DW_STAMP_IT, 26);
      if (it >= 7) ln_row(ob + it - 7, (it - 1) & 1);
      DW_STAMP(lid == 0 && it == GCV_DW_STAMP_IT, 13); DW_STAMP(lid == NLN - 64 && it == GCV_DW_STAMP_IT, 27);
      GCV_LDS_BARRIER();
      DW_STAMP(lid == 0 && it == GCV_DW_STAMP_IT, 14); DW_STAMP(lid == NLN - 64 && it == GCV_DW_STAMP_IT, 28);
    }
    ln_row(ob + nit - 7, (nit - 1) & 1);
  }
}

}  // namespace gcv
