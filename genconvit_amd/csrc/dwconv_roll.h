// K4 (round 2): depthwise 7x7 + LayerNorm as a ROLLING STRIP kernel.
//
// Replaces timm ConvNeXtBlock's front half — conv_dw (Conv2d(dim, dim, 7, padding=3, groups=dim)) followed by
// LayerNorm(dim, eps 1e-6) over channels — as called from /root/reference/model/genconvit_ed.py:82-83 and
// model/genconvit_vae.py:111-112.
//
// Why the round-1 kernels sat at 14 % of the HBM roofline with traffic already algorithmic: one workgroup per 7x7
// tile re-reads a 13x13 halo (3.45x) and pays 104 LDS reads (+ as many packs) for 343 dot2 per thread, five
// barriers per tile.  This kernel removes the vertical halo altogether:
//   * one workgroup = a band of rows of one image over its whole width; one TAP THREAD = one channel of a
//     7-pixel-wide column strip.  It walks DOWN the band, one input row per step, holding the 7 output rows that are
//     still open as 49 fp32 accumulators and the 49 taps in registers: an input row of 13 values feeds
//     7 x 7 x 7 = 343 FMAs (26 FMAs per value read, against 6.6);
//   * rows above / below the band and tap rows that fall outside it are skipped by wave-uniform branches, so no FMA
//     is spent on vertical padding; a strip that spans the whole image width (W = 7) also skips the horizontal
//     padding taps at compile time (FULLW); horizontal padding otherwise = a zeroed 3-pixel apron of the LDS ring;
//   * specialised waves, ONE workgroup barrier per row.  Threads [0, NS * C) run the taps; a further third of that
//     many ("staging waves") move every byte across the vector-memory pipe exactly once, 16 bytes per lane: they
//     load input row it + 3 (a whole step to land), widen it to fp32 into a three-slot LDS ring, and LayerNorm +
//     store the output row the tap waves finished one step earlier (24 values per lane from a two-slot fp32 LDS
//     ring, C / 24 lanes per pixel, DPP group reduction).  (A first version read its 13 values per row straight from
//     global memory, 2 bytes per lane: 156 loads + 84 stores per row and CU took longer than the 343 FMAs —
//     profiles/r02_dw/v1_*.)
//   * the tap waves read the 13 values of the NEXT row with inline-asm ds_read_b32 while the current row's FMAs run
//     and rotate their priority (s_setprio) as they advance through a row's seven tap rows: three waves share a SIMD
//     and its arbiter would otherwise serve two of them and leave the third to run alone at half rate
//     (profiles/micro/fmac_banks.hip).
// fp32 accumulation, exact fp32 taps, LayerNorm statistics in fp32 for every storage dtype.  DESIGN.md section 4 has
// the measurements and the dead ends.
#pragma once
#include <type_traits>

#include "common.h"

namespace gcv {

// per-dtype pieces: EPC elements per 16-byte piece; lo / hi widen the two 16-bit halves of a dword to fp32
template <typename T> struct DwElem;
template <> struct DwElem<float> {
  static constexpr int EPC = 4;
  __device__ static __forceinline__ u32x4 pack(const float (&v)[4]) {
    return u32x4{__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1]),
                 __builtin_bit_cast(uint32_t, v[2]), __builtin_bit_cast(uint32_t, v[3])};
  }
};
template <> struct DwElem<half_t> {
  static constexpr int EPC = 8;
  __device__ static __forceinline__ float lo(uint32_t v) { return (float)__builtin_bit_cast(_Float16, (uint16_t)v); }
  __device__ static __forceinline__ float hi(uint32_t v) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(v >> 16)); }
};
template <> struct DwElem<bf16_t> {
  static constexpr int EPC = 8;
  __device__ static __forceinline__ float lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
  __device__ static __forceinline__ float hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
};

template <int L> __device__ __forceinline__ float dw_group_sum(float v) {
  static_assert(L == 4 || L == 8 || L == 16 || L == 32 || L == 64, "LayerNorm groups: 4 .. 64 lanes");
  if constexpr (L == 4) { v += GCV_DPP_F32(v, 0xB1); return v + GCV_DPP_F32(v, 0x4E); }   // quad: xor 1, xor 2
  else if constexpr (L == 8) return group8_sum(v);
  else if constexpr (L == 16) return row16_sum(v);
  else if constexpr (L == 32) return group32_sum(v);
  else return wave_sum(v);
}

#define GCV_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---- inline-asm LDS reads of one staged (fp32) input row; the offsets are instruction immediates ----
template <int OFF> __device__ __forceinline__ void dw_lds_rd(float& dst, uint32_t addr) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int C, int S, int S1> __device__ __forceinline__ void dw_read_row(float (&nxt)[13], uint32_t addr) {
  if constexpr (S < S1) {
    static_assert(S * C * 4 < 65536, "ds offset field");
    dw_lds_rd<S * C * 4>(nxt[S], addr);
    dw_read_row<C, S + 1, S1>(nxt, addr);
  }
}

template <typename T, int C, int NS> struct DwRollLds {
  static constexpr int P = 7 * NS;
  static constexpr int HP = (NS == 1) ? 0 : 3;         // zero apron (pixels) left and right of a staged input row
  static constexpr int IN_ROW = (P + 2 * HP) * C;      // elements
  static constexpr int IN_BYTES = 3 * IN_ROW * 4;      // three-slot ring of staged input rows, converted to fp32
  static constexpr int SVAL_BYTES = 2 * P * C * 4;
  static constexpr int bytes = IN_BYTES + SVAL_BYTES;
  static constexpr int NCONV = NS * C;                 // tap threads: one per (strip, channel)
  static constexpr int NLN = NCONV / 3;                // LayerNorm / staging threads (one wave per SIMD at NCONV = 768)
  static constexpr int NT = NCONV + NLN;
};

// grid: nimg * nbands workgroups of (NS * C) * 4 / 3 threads, NS = W / 7; workgroup = (image, band of `band_rows`
// output rows).  Waves are specialised: threads [0, NS * C) run the taps, the last third stages input rows and does
// LayerNorm + the stores of the row the tap waves finished one step earlier; the two meet at one barrier per row.
template <typename T, int C, int NS>
__global__ void __launch_bounds__((DwRollLds<T, C, NS>::NT), 4)
dwconv7_ln_roll_kernel(const T* __restrict__ x, const float* __restrict__ wdw /*[49][C]*/,
                       const float* __restrict__ bdw, const float* __restrict__ lnw, const float* __restrict__ lnb,
                       T* __restrict__ y, int H, int band_rows, int nbands, float eps) {
  typedef DwElem<T> EL;
  typedef DwRollLds<T, C, NS> LY;
  constexpr bool FULLW = (NS == 1);
  constexpr int NCONV = LY::NCONV, NLN = LY::NLN;
  constexpr int W = 7 * NS;
  constexpr int P = LY::P, HP = LY::HP, IN_ROW = LY::IN_ROW;
  constexpr int EPC = EL::EPC;
  constexpr int S0 = FULLW ? 3 : 0, S1 = FULLW ? 10 : 13;   // halo columns that can hold data
  extern __shared__ __attribute__((aligned(16))) unsigned char dwr_lds[];
  float* const in_ring = reinterpret_cast<float*>(dwr_lds);
  float* const sval_ring = reinterpret_cast<float*>(dwr_lds + LY::IN_BYTES);

  const int tid = threadIdx.x;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);    // an XCD walks neighbouring bands / images: apron rows hit its L2
  const int band = wg % nbands, img = wg / nbands;
  const int ob = band * band_rows;
  const int nrows = min(band_rows, H - ob);
  const int r0 = ob - 3;
  const int nit = nrows + 6;
  const int it0 = max(0, -r0);                        // steps whose input row lies above the image are not run at all
  constexpr int row_bytes = W * C * (int)sizeof(T);
  const int64_t img_elems = (int64_t)H * W * C;

  // Timeline (iteration `it` handles input row r0 + it; barrier `it` closes it):
  //   staging waves, iteration it : write row it+2 (loaded one iteration earlier) into ring slot (it+2)%3, issue the
  //                                 loads of row it+3, LayerNorm + store the output row the tap waves finished in it-1
  //   tap waves, iteration it     : read row `it` from slot it%3, 343 FMAs, write the finished output row to sval[it&1]
  // so every global load has a whole iteration to land before anything waits for it.
  auto row_ok = [&](int it) { const int r = r0 + it; return it < nit && r >= 0 && r < H; };

  if (tid < NCONV) {
    // ================================================================== tap waves
    const int sl = tid / C, c = tid - sl * C;
    // ES0 / ES1: halo columns of this strip that can hold data.  Two strips of 7 at multiples of 64 channels (the 14-pixel
    // maps at C = 192 / 384: whole waves per strip) know at compile time that their outer three columns are the zero apron:
    // 6 of a tap row's 49 FMAs and 3 of its 13 LDS reads are padding there, as in the single full-width strip.
    auto tap_waves = [&](auto s0c, auto s1c) {
    constexpr int ES0 = decltype(s0c)::value, ES1 = decltype(s1c)::value;
    float w[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) w[k] = wdw[k * C + c];
    const float bv = bdw[c];
    float acc[7][7];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[i][j] = bv;
    float* const sv_base = sval_ring + sl * 7 * C + c;
    // the 13 values of the NEXT input row are read from LDS while this row's FMAs run: inline-asm reads that nothing
    // waits for until the `s_waitcnt lgkmcnt(0)` of the step's barrier, which names the registers so that no use can
    // move above it.  (The ring holds fp32: 16-bit values would need 13 conversions and 13 more live registers here.)
    float nxt[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) nxt[i] = 0.0f;
    const uint32_t in_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)dwr_lds +
                             (uint32_t)(((sl * 7 + HP - 3) * C + c) * 4);
#define GCV_DW_BARRIER_NXT()                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]),    \
                 "+v"(nxt[4]), "+v"(nxt[5]), "+v"(nxt[6]), "+v"(nxt[7]), "+v"(nxt[8]), "+v"(nxt[9]), "+v"(nxt[10]), \
                 "+v"(nxt[11]), "+v"(nxt[12])::"memory")
    GCV_LDS_BARRIER();                                 // P1: rows 0 and 1 are staged
    dw_read_row<C, ES0, ES1>(nxt, in_addr);              // row it0 (slot 0)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]), "+v"(nxt[4]), "+v"(nxt[5]),
                 "+v"(nxt[6]), "+v"(nxt[7]), "+v"(nxt[8]), "+v"(nxt[9]), "+v"(nxt[10]), "+v"(nxt[11]), "+v"(nxt[12])::"memory");
    int pslot = 1;                                     // ring slot of input row it + 1
    for (int base = 0; base < nit; base += 7) {
#pragma unroll
      for (int ph = 0; ph < 7; ++ph) {
        const int it = base + ph;
        if (it >= it0 && it < nit) {
          const int r = r0 + it;
          float cur[13];
          DW_STAMP(tid == 0 && it == GCV_DW_STAMP_IT, 0); DW_STAMP(tid == 320 && it == GCV_DW_STAMP_IT, 16); DW_STAMP(tid == NCONV - 64 && it == GCV_DW_STAMP_IT, 21);
#pragma unroll
          for (int s = ES0; s < ES1; ++s) cur[s] = nxt[s];
          if (!(GCV_DWR_ABLATE & 1)) dw_read_row<C, ES0, ES1>(nxt, in_addr + (uint32_t)(pslot * IN_ROW * 4));
          pslot = (pslot == 2) ? 0 : pslot + 1;
          // tap rows in the order 6, 0, 1 .. 5: row 6 completes an output row, which is written to LDS at once so that
          // the 84 ds_write_b32 of the workgroup (64 B/clk: ~340 cycles) drain under the other six rows' FMAs
          const bool rvalid = !(GCV_DWR_ABLATE & 1) && r >= 0 && r < H;
          auto tap_row = [&](int ky, int pos) {
            const int oi = it - ky;                    // output row (relative to the band) this tap row feeds
            if (rvalid && oi >= 0 && oi < nrows) {
              const int slot = (ph - ky + 7) % 7;      // compile-time after unrolling
              // Three tap waves share a SIMD, and its arbiter (priority, then age) lets the two oldest issue at their
              // full rate while the third waits, then leaves that one to run ALONE at half the SIMD's rate
              // (profiles/micro/fmac_banks.hip: 3 waves/SIMD take the time of 4).  Falling priority as a wave
              // advances through its seven tap rows keeps the three within a row of each other instead.
              if (!(GCV_DWR_ABLATE & 32)) {
                switch (pos) {                         // (the builtin wants a literal; the switch folds after unrolling)
                  case 0: case 1: __builtin_amdgcn_s_setprio(3); break;
                  case 2: case 3: __builtin_amdgcn_s_setprio(2); break;
                  case 4: case 5: __builtin_amdgcn_s_setprio(1); break;
                  default: __builtin_amdgcn_s_setprio(0); break;
                }
              }
#pragma unroll
              for (int ox = 0; ox < 7; ++ox)
#pragma unroll
                for (int kx = 0; kx < 7; ++kx)
                  if (ox + kx >= ES0 && ox + kx < ES1)
                    acc[slot][ox] = fmaf(cur[ox + kx], w[ky * 7 + kx], acc[slot][ox]);
            }
          };
          tap_row(6, 0);
          if (it >= 6) {                               // input row r completes output row r - 3
            const int sd = (ph + 1) % 7;
            float* sv = sv_base + (it & 1) * (P * C);
#pragma unroll
            for (int j = 0; j < 7; ++j) { sv[j * C] = acc[sd][j]; acc[sd][j] = bv; }
          }
#pragma unroll
          for (int ky = 0; ky < 6; ++ky) tap_row(ky, ky + 1);
#if GCV_DW_STAMPS
          asm volatile("" : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[3][0]), "+v"(acc[4][0]), "+v"(acc[5][0]), "+v"(acc[6][0]));
          DW_STAMP(tid == 0 && it == GCV_DW_STAMP_IT, 2); DW_STAMP(tid == 320 && it == GCV_DW_STAMP_IT, 18); DW_STAMP(tid == NCONV - 64 && it == GCV_DW_STAMP_IT, 23);
#endif
          DW_STAMP(tid == 0 && it == GCV_DW_STAMP_IT, 3); DW_STAMP(tid == 320 && it == GCV_DW_STAMP_IT, 19); DW_STAMP(tid == NCONV - 64 && it == GCV_DW_STAMP_IT, 24);
          if (!(GCV_DWR_ABLATE & 16)) GCV_DW_BARRIER_NXT();
          DW_STAMP(tid == 0 && it == GCV_DW_STAMP_IT, 4); DW_STAMP(tid == 320 && it == GCV_DW_STAMP_IT, 20); DW_STAMP(tid == NCONV - 64 && it == GCV_DW_STAMP_IT, 25);
#if GCV_DW_STAMPS
          if (tid == 0 && blockIdx.x < 64 && (it == GCV_DW_STAMP_IT || it == GCV_DW_STAMP_IT + 10))
            gcv_dw_stamps[blockIdx.x * 32 + (it == GCV_DW_STAMP_IT ? 5 : 7)] = __builtin_amdgcn_s_memrealtime();
          DW_STAMP(tid == 0 && it == GCV_DW_STAMP_IT + 10, 6);
#endif
        }
      }
    }
#undef GCV_DW_BARRIER_NXT
    };
    if constexpr (NS == 2 && C % 64 == 0) {
      if (__builtin_amdgcn_readfirstlane(sl) == 0) tap_waves(std::integral_constant<int, 3>{}, std::integral_constant<int, 13>{});
      else tap_waves(std::integral_constant<int, 0>{}, std::integral_constant<int, 10>{});
    } else {
      tap_waves(std::integral_constant<int, S0>{}, std::integral_constant<int, S1>{});
    }
  } else {
    // ================================================================== staging + LayerNorm waves
    __builtin_amdgcn_s_setprio(3);                     // one latency-bound wave per SIMD beside three FMA-bound ones
    const int lid = tid - NCONV;
    constexpr int L = C / 24;                          // lanes per pixel: 6 pieces of 4 channels each
    static_assert(L == 4 || L == 8 || L == 16 || L == 32, "C = 96 / 192 / 384 / 768");
    constexpr int ROWP = P * C / EPC;                  // 16-byte pieces of one image row
    constexpr int NPT = (ROWP + NLN - 1) / NLN;        // staged pieces per thread (3 for 16-bit, 6 for fp32)
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(x + (int64_t)img * img_elems), 0, (int)(img_elems * (int64_t)sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(
        y + (int64_t)img * img_elems, 0, (int)(img_elems * (int64_t)sizeof(T)), 0x00020000);
    const int p = lid / L, g = lid - p * L;            // LayerNorm role: pixel p of the row, channels 4 (g + L j) ..+4
    const bool ln_on = lid < P * L;
    float lwv[24], lbv[24];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float4 a = *reinterpret_cast<const float4*>(lnw + 4 * (g + L * j));
      const float4 b = *reinterpret_cast<const float4*>(lnb + 4 * (g + L * j));
      lwv[4 * j] = a.x; lwv[4 * j + 1] = a.y; lwv[4 * j + 2] = a.z; lwv[4 * j + 3] = a.w;
      lbv[4 * j] = b.x; lbv[4 * j + 1] = b.y; lbv[4 * j + 2] = b.z; lbv[4 * j + 3] = b.w;
    }
    // zero the aprons of the three ring slots once (they are never written again)
    if constexpr (HP > 0) {
      constexpr int APR = HP * C / 4;                  // float4 pieces per apron
      for (int i = lid; i < 6 * APR; i += NLN) {
        const int slot = i / (2 * APR), j = i - slot * (2 * APR);
        const int off = slot * IN_ROW + (j < APR ? j * 4 : (HP + P) * C + (j - APR) * 4);
        *reinterpret_cast<float4*>(in_ring + off) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    }
    u32x4 stg[NPT];
    auto stage_load = [&](int it) {                    // row_ok(it) checked by the caller
      const int row = r0 + it;
#pragma unroll
      for (int k = 0; k < NPT; ++k)
        if (NPT * NLN == ROWP || lid + k * NLN < ROWP)
          stg[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (lid + k * NLN) * 16, row * row_bytes, 0));
    };
    auto stage_write = [&](int slot) {                 // 16-bit storage is widened to fp32 on the way into the ring
#pragma unroll
      for (int k = 0; k < NPT; ++k)
        if (NPT * NLN == ROWP || lid + k * NLN < ROWP) {
          float* dst = in_ring + slot * IN_ROW + HP * C + (lid + k * NLN) * EPC;
          if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<u32x4*>(dst) = stg[k];
          } else {
            float f[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[2 * e] = EL::lo(stg[k][e]); f[2 * e + 1] = EL::hi(stg[k][e]); }
            *reinterpret_cast<float4*>(dst) = make_float4(f[0], f[1], f[2], f[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(f[4], f[5], f[6], f[7]);
          }
        }
    };
    // LayerNorm + store of one finished output row: every lane keeps its 24 values in registers
    auto ln_row = [&](int orow, int slot) {
      if (!ln_on) return;
      const float* sv = sval_ring + slot * (P * C) + p * C + 4 * g;
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
      float mean, rstd;
      if constexpr (sizeof(T) == 4) {                  // fp32 storage: two-pass statistics (local mean / M2, Chan's combination)
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < 24; ++e) s += v[e];
        const float m = s * (1.0f / 24.0f);
        float qq = 0.0f;
#pragma unroll
        for (int e = 0; e < 24; ++e) qq = fmaf(v[e] - m, v[e] - m, qq);
        mean = dw_group_sum<L>(s) * (1.0f / C);
        const float dm = m - mean;
        const float var = dw_group_sum<L>(fmaf(24.0f * dm, dm, qq)) * (1.0f / C);
        rstd = 1.0f / sqrtf(var + eps);
      } else {                                         // 16-bit storage: sum / sum of squares in one pass, v_rsq_f32
        float s0 = 0.0f, s1 = 0.0f, q0 = 0.0f, q1 = 0.0f;
#pragma unroll
        for (int e = 0; e < 24; e += 2) {
          s0 += v[e]; s1 += v[e + 1];
          q0 = fmaf(v[e], v[e], q0); q1 = fmaf(v[e + 1], v[e + 1], q1);
        }
        mean = dw_group_sum<L>(s0 + s1) * (1.0f / C);
        const float ex2 = dw_group_sum<L>(q0 + q1) * (1.0f / C);
        rstd = __builtin_amdgcn_rsqf(fmaxf(fmaf(-mean, mean, ex2), 0.0f) + eps);
      }
#if GCV_DW_STAMPS
      asm volatile("" : "+v"(rstd), "+v"(mean));
      DW_STAMP(lid == 0 && orow == ob + GCV_DW_STAMP_IT - 7, 12);
#endif
      if (GCV_DWR_ABLATE & 4) return;
      // all six pieces are finished in registers of their own BEFORE the first store is issued: a 16-byte store whose
      // data registers the very next VALU instructions overwrite came out with a stale first dword in lanes 12-15 of
      // each 16-lane row (seen with fp32 storage, run to run different rows) — no register is reused behind a store
      const float nmr = -mean * rstd;
      u32x4 pk[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(fmaf(v[4 * j + e], rstd, nmr), lwv[4 * j + e], lbv[4 * j + e]);
        if constexpr (sizeof(T) == 4) {
          pk[j] = EL::pack(o);
        } else {                                       // vector conversion: v_fma_mixlo/hi_f16, v_cvt_pk_bf16_f32
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          typedef T t16x2 __attribute__((ext_vector_type(2)));
          const t16x2 lo = __builtin_convertvector(f32x2{o[0], o[1]}, t16x2), hi = __builtin_convertvector(f32x2{o[2], o[3]}, t16x2);
          pk[j] = u32x4{__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi), 0u, 0u};
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int off = (p * C + 4 * (g + L * j)) * (int)sizeof(T);
        // 8-byte stores only.  A 16-byte buffer store with an SGPR soffset reads its data registers late and the
        // compiler's hazard recogniser does not guard that form: VALU writes a few instructions behind it gave stale
        // dwords in lanes 12-15 of each 16-lane row (fp32 storage; first seen with register reuse right behind the store,
        // again - dwords 2-3 of the first piece - in an experiment without the row barrier behind the stores).
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk[j][0], pk[j][1]}, rsy, off, orow * row_bytes, 0);
        if constexpr (sizeof(T) == 4)
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk[j][2], pk[j][3]}, rsy, off + 8, orow * row_bytes, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };

    if (row_ok(it0)) { stage_load(it0); stage_write(0); }
    if (row_ok(it0 + 1)) { stage_load(it0 + 1); stage_write(1); }
    if (!(GCV_DWR_ABLATE & 8) && row_ok(it0 + 2)) stage_load(it0 + 2);
    GCV_LDS_BARRIER();                                 // P1
    int wslot = 2;                                     // ring slot row it + 2 goes to ((it - it0 + 2) % 3)
    for (int it = it0; it < nit; ++it) {
      DW_STAMP(lid == 0 && it == GCV_DW_STAMP_IT, 8); DW_STAMP(lid == NLN - 64 && it == GCV_DW_STAMP_IT, 26);
      if (!(GCV_DWR_ABLATE & 8) && row_ok(it + 2)) stage_write(wslot);
      wslot = (wslot == 2) ? 0 : wslot + 1;
      DW_STAMP(lid == 0 && it == GCV_DW_STAMP_IT, 9);
      if (!(GCV_DWR_ABLATE & 8) && row_ok(it + 3)) stage_load(it + 3);
      DW_STAMP(lid == 0 && it == GCV_DW_STAMP_IT, 10);
      if (!(GCV_DWR_ABLATE & 2) && it >= 7) ln_row(ob + it - 7, (it - 1) & 1);
      DW_STAMP(lid == 0 && it == GCV_DW_STAMP_IT, 13); DW_STAMP(lid == NLN - 64 && it == GCV_DW_STAMP_IT, 27);
      if (!(GCV_DWR_ABLATE & 16)) GCV_LDS_BARRIER();
      DW_STAMP(lid == 0 && it == GCV_DW_STAMP_IT, 14); DW_STAMP(lid == NLN - 64 && it == GCV_DW_STAMP_IT, 28);
    }
    if (!(GCV_DWR_ABLATE & 2)) ln_row(ob + nit - 7, (nit - 1) & 1);
  }
}

}  // namespace gcv
