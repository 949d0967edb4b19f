// LDS-tiled MFMA GEMM family for gfx950:  C[m][n] = epilogue( sum_k A[m][k] * Wt[n][k] ).
//
// One kernel template covers every dense contraction on the GenConViT path
// (SURVEY.md §2.1 K1,K2,K5,K6,K7,K8,K9,K10,K12):
//   A operand   : plain row-major tokens, or implicit im2col of a 3x3 conv over NHWC
//   Wt operand  : (N,K) row-major == nn.Linear layout; conv weights pre-packed to (Cout, ky,kx,ci)
//   epilogues   : bias+activation | bias,*gamma,+residual (ConvNeXt layer-scale) |
//                 bias+act+2x2 max-pool | ConvTranspose2d(k=s=2) pixel-shuffle store | split-K slab
//
// Tiling: 256 threads = 4 waves, each wave owns WMxWN of the BMxBN tile as (WM/32)x(WN/32)
// 32x32 MFMA accumulators.  Operands are staged global -> registers -> LDS (double buffered, one
// barrier per K tile) as 16-byte chunks along K; a lane's MFMA fragment is exactly one chunk:
//   16-bit : v_mfma_f32_32x32x16_{f16,bf16}   lane (r=l&31,h=l>>5) holds k = 8h..8h+7 of row r
//   fp32   : v_mfma_f32_32x32x2_f32 x4        lane half h supplies k = 4h+j in sub-step j
// (the k permutation inside a chunk pair is the same for A and Wt, so the sum is unchanged).
// LDS rows are BKB bytes (64 for fp32, 128 for 16-bit, 64 for 16-bit GEMMs whose K is 32 mod 64) with
// the chunk index XOR-swizzled by the row so the 16-lane groups of ds_read_b128 touch all 64 banks once.
// Epilogue: the Wt fragment is fed as the MFMA A operand, so a lane owns one TOKEN and 4 consecutive
// registers are 4 consecutive output channels; bias/activation/layer-scale run in registers, the tile
// is staged through LDS and written back with 8-16 B per lane over contiguous channels.
#pragma once
#include <type_traits>

#include "common.h"

namespace gcv {

enum { A_PLAIN = 0, A_IM2COL3_POOL = 1, A_IM2COL3_S2 = 2 };
enum { EPI_BIAS_ACT = 0, EPI_RESID = 1, EPI_POOL4 = 2, EPI_CONVT = 3, EPI_SPLITK = 4 };

struct GemmArgs {
  const void* A;        // A_PLAIN: (M, lda) ; im2col modes: NHWC input (nimg, H, W, Cin)
  const void* Wt;       // (N, K) row-major
  void* C;              // output, T
  const float* bias;    // (N) — EPI_CONVT: (Cout); may be null
  const float* gamma;   // EPI_RESID: (N)
  const void* resid;    // EPI_RESID: (M, ldc) T (may alias C)
  float* partial;       // EPI_SPLITK: (splitk, M, N) fp32
  int M, N, K;
  int lda, ldc;
  int act;
  int splitk;           // >= 1 (grid.y)
  int k_per_split;      // elements, multiple of the K tile
  int H, W;             // im2col: conv input dims; EPI_CONVT: input dims of the transposed conv
  int cin_log2;         // im2col: log2(Cin)
  int cout_log2;        // EPI_CONVT: log2(Cout); N = 4*Cout ordered (dy,dx,co)
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Mfma;
template <> struct Mfma<float> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& acc) {
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf[j], acc, 0, 0, 0);
  }
};
template <> struct Mfma<half_t> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  }
};
template <> struct Mfma<bf16_t> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};

// Branch-free erf (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7 analytically, 6e-7 in fp32): the
// libm erff is ~100 branchy instructions and made the GELU epilogue 5x the MFMA time of a K=96 GEMM.
__device__ __forceinline__ float erf_fast(float x) {
  const float a = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(a * a * -1.4426950408889634f);
  return copysignf(fmaf(-p, e, 1.0f), x);
}

template <int ACT> __device__ __forceinline__ float act_fn(float x) {
  if (ACT == ACT_RELU) return fmaxf(x, 0.0f);
  if (ACT == ACT_GELU) { const float h = 0.5f * x; return fmaf(h, erf_fast(x * 0.70710678118654752440f), h); }
  if (ACT == ACT_LEAKY) return fmaxf(x, 0.0f) + 0.01f * fminf(x, 0.0f);
  return x;
}

/// GELU for the 16-bit storage paths, four values at a time on the packed fp32 pipe (v_pk_fma_f32: two lanes of
// work per issue slot).  The A&S form above costs ~13 VALU + 2 quarter-rate transcendentals per value and made the
// hidden-layer epilogues VALU-bound; here
//     gelu(x) = max(x, 0) - h(min(|x|, 4.5)),   h(a) = a * 0.5 * erfc(a / sqrt 2)
// with h a degree-10 minimax polynomial in t = a * (2/4.5) - 1 (fit in profiles/gelu_fit.py; |error| <= 1.5e-5
// over all x, i.e. 30x below the fp16 rounding step of the stored result; h(4.5) = 1.5e-5 is the tail that is cut).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float c) { return (f32x2){c, c}; }
// NCH independent Horner chains advance in lockstep: a dependent v_pk_fma_f32 needs wait states, so one chain at a
// time leaves the packed pipe half idle (the compiler pads it with s_nop); 4 chains keep it issuing back to back.
template <int NCH> __device__ __forceinline__ void gelu_pk_n(f32x2 (&x)[NCH]) {
  constexpr float kC[11] = {2.749713404e-02f, -1.330395067e-01f, 2.465923971e-01f, -1.472158060e-01f,
                            -2.029683018e-01f, 4.347813707e-01f, -2.049071560e-01f, -1.763150062e-01f,
                            1.763803063e-01f, 2.178248281e-02f, -4.258673483e-02f};
  f32x2 t[NCH], p[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const f32x2 a = {fminf(fabsf(x[c][0]), 4.5f), fminf(fabsf(x[c][1]), 4.5f)};
    t[c] = __builtin_elementwise_fma(a, splat2(0.44444444444f), splat2(-1.0f));
    p[c] = __builtin_elementwise_fma(splat2(kC[10]), t[c], splat2(kC[9]));
  }
#pragma unroll
  for (int k = 8; k >= 0; --k) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) p[c] = __builtin_elementwise_fma(p[c], t[c], splat2(kC[k]));
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const f32x2 r = {fmaxf(x[c][0], 0.0f), fmaxf(x[c][1], 0.0f)};
    x[c] = r - p[c];
  }
}
__device__ __forceinline__ f32x2 gelu_pk(f32x2 x) {
  f32x2 v[1] = {x};
  gelu_pk_n<1>(v);
  return v[0];
}

// The same polynomial on plain v_fma_f32, NCH independent chains (bit-identical results: every lane of a v_pk_fma_f32 is an
// IEEE fma).  Alone on a SIMD the packed form is ~6 % faster (profiles/micro/gelu_rate.hip: 18.7 vs 19.9 ns per element
// with two waves), but beside MFMAs a v_pk_fma_f32 stalls the matrix pipe (~22 cycles each, MI355X guide "price of one
// filler beside MFMAs"): kernels that interleave the GELU with MFMAs use this one and are built with -fno-slp-vectorize
// (hipcc otherwise re-packs the chains).
template <int NCH> __device__ __forceinline__ void gelu_fma_n(float (&x)[NCH]) {
  constexpr float kC[11] = {2.749713404e-02f, -1.330395067e-01f, 2.465923971e-01f, -1.472158060e-01f,
                            -2.029683018e-01f, 4.347813707e-01f, -2.049071560e-01f, -1.763150062e-01f,
                            1.763803063e-01f, 2.178248281e-02f, -4.258673483e-02f};
  float t[NCH], p[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    t[c] = fmaf(fminf(fabsf(x[c]), 4.5f), 0.44444444444f, -1.0f);
    p[c] = fmaf(kC[10], t[c], kC[9]);
  }
#pragma unroll
  for (int k = 8; k >= 0; --k)
#pragma unroll
    for (int c = 0; c < NCH; ++c) p[c] = fmaf(p[c], t[c], kC[k]);
#pragma unroll
  for (int c = 0; c < NCH; ++c) x[c] = fmaxf(x[c], 0.0f) - p[c];
}

// GELU with the polynomial on the packed-fp16 pipe (16-bit storage paths).  Measured on gfx950
// (profiles/micro/mfma_valu_overlap.hip): while MFMAs are in flight on a SIMD every vector instruction costs ~4-5 cycles
// of issue whatever it is and however many waves share the SIMD (one wave: 15 + 4 N cycles per MFMA + N instructions),
// so beside MFMAs the only lever is fewer instructions — and a v_pk_fma_f16 advances TWO Horner chains for the price
// of one v_fma_f32.  Only h(a) = a/2 * erfc(a / sqrt 2) <= 0.17 is evaluated in fp16; |x| is clamped and converted
// first, and the final max(x, 0) - h stays in fp32, so the stored 16-bit result moves by a fraction of its own rounding
// step: rms error of the fp16 hidden activation 2.37e-4 against 2.12e-4 with the fp32 polynomial (exact GELU rounded to
// fp16: 2.12e-4), maximum 2.1e-3 against 2.0e-3; unchanged to three digits for bf16 storage (profiles/gelu_fit.py --h16).
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
// Round 3, second pass (profiles/gelu_fit.py --h16): the fp16 evaluation itself carries ~1e-4 of noise on h <= 0.17, so a
// degree-8 fit on [0, 4] (|fit error| 1.1e-4; h(4) = 1.3e-4 is what the clamp cuts off) leaves the stored fp16 activation
// where the degree-10 fit on [0, 4.5] had it (rms 2.40e-4 against 2.37e-4; exact GELU rounded to fp16: 2.12e-4) for one
// vector instruction per value less.  For fp16 STORAGE the last step also stays on the packed pipe (finish_pk:
// y = fma(p, 2, max(fp16(x), 0)), one v_pk_max_i16 + one v_pk_fma_f16 per PAIR instead of v_max_i32 + v_fma_mix per value
// and a v_cvt_pk afterwards): x is rounded to fp16 before the ReLU part, as the reference's own .half() pipeline does with
// the Linear output (rms 3.25e-4).  7 vector instructions per hidden value instead of 9.5; bf16 storage: 8.5.
// NaN: max(x, 0) is an integer max of the bit pattern (v_pk_max_i16 / v_max_i32), so a NaN with the sign bit set comes out as
// the finite value 2 p(4) while a positive NaN propagates (|x| clamps to 4.0, the ReLU part stays NaN).  Accepted on this path:
// a NaN in a hidden activation means the input or the weights were already corrupt, and the fp32-storage path (exact erf, float
// max) propagates every NaN — compare the two dtypes when chasing one (tests/test_kernels_gpu.py test_gelu_nan_behaviour_is_documented).
struct GeluH16 {                       // polynomial state of NP pairs of values between the three phases
  static constexpr int DEG = 8;
  template <int NP> struct State { h16x2 xh[NP], t[NP], p[NP]; };
  static __device__ __forceinline__ h16x2 k2(float c) { return (h16x2){(_Float16)c, (_Float16)c}; }
  // the polynomial carries -h/2 (every coefficient times -0.5: exact in fp16), so that the last step is
  // y = fma(p, 2, max(x, 0)) with the fp16 -> fp32 widening of p inside the instruction (v_fma_mix_f32; a multiplier
  // of -1 is folded into a subtraction first and then costs a separate v_cvt_f32_f16 per value)
  static constexpr float kScale = 2.0f;
  static constexpr float kC[DEG + 1] = {-0.5f * 4.543376254e-02f, 0.5f * 1.712937983e-01f, -0.5f * 2.188785784e-01f,
                                        -0.5f * 1.159314756e-02f, 0.5f * 3.094936844e-01f, -0.5f * 2.450228537e-01f,
                                        -0.5f * 3.058419957e-02f, 0.5f * 8.537015778e-02f, -0.5f * 1.466048626e-02f};
  // phase A: a = min(|x|, 4) -> fp16, t = a/2 - 1, Horner levels DEG .. LAST (inclusive)
  template <int NP, int LAST> static __device__ __forceinline__ void begin(const float* x, State<NP>& st) {
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      const f32x2 xv = {x[2 * c], x[2 * c + 1]};                                  // v_cvt_pk_f16_f32, then |.| of both halves
      st.xh[c] = __builtin_convertvector(xv, h16x2);
      const uint32_t ab = __builtin_bit_cast(uint32_t, st.xh[c]) & 0x7fff7fffu;   // with one v_and_b32
      typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
      // non-negative fp16 values order like their bit patterns: the clamp is a v_pk_min_u16 against 4.0 = 0x4400 (a float
      // minimum would first canonicalise its operand with one more instruction; a NaN clamps to 4)
      const h16x2 a = __builtin_bit_cast(h16x2, __builtin_elementwise_min(__builtin_bit_cast(u16x2, ab), (u16x2){0x4400, 0x4400}));
      st.t[c] = __builtin_elementwise_fma(a, k2(0.5f), k2(-1.0f));
      st.p[c] = __builtin_elementwise_fma(k2(kC[DEG]), st.t[c], k2(kC[DEG - 1]));
    }
    horner<NP, DEG - 2, LAST>(st);
  }
  template <int NP, int FROM, int TO> static __device__ __forceinline__ void horner(State<NP>& st) {
#pragma unroll
    for (int k = FROM; k >= TO; --k)
#pragma unroll
      for (int c = 0; c < NP; ++c) st.p[c] = __builtin_elementwise_fma(st.p[c], st.t[c], k2(kC[k]));
  }
  // phase C: y = max(x, 0) - h, in fp32
  template <int NP> static __device__ __forceinline__ void finish(const float* x, const State<NP>& st, float* y) {
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      // max(x, 0) as a signed-integer maximum of the bit pattern (negative floats are negative integers): one instruction,
      // where fmaxf canonicalises its operand first; the fp16 -> fp32 widening of h rides on the fma (v_fma_mix_f32)
      const float r0 = __builtin_bit_cast(float, max(__builtin_bit_cast(int, x[2 * c]), 0));
      const float r1 = __builtin_bit_cast(float, max(__builtin_bit_cast(int, x[2 * c + 1]), 0));
      y[2 * c] = __builtin_fmaf((float)st.p[c][0], kScale, r0);
      y[2 * c + 1] = __builtin_fmaf((float)st.p[c][1], kScale, r1);
    }
  }
  // phase C on the packed pipe (fp16 storage): dword c = the fp16 pair (2c, 2c+1), ready to be an MFMA operand or stored
  template <int NP> static __device__ __forceinline__ void finish_pk(const State<NP>& st, uint32_t* out) {
    typedef short i16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      const h16x2 r = __builtin_bit_cast(h16x2, __builtin_elementwise_max(__builtin_bit_cast(i16x2, st.xh[c]), (i16x2){0, 0}));
      out[c] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(st.p[c], k2(kScale), r));
    }
  }
  // phase C for storage type T: NP dwords of 16-bit pairs
  template <typename T, int NP> static __device__ __forceinline__ void finish_frag(const float* x, const State<NP>& st, uint32_t* out) {
    if constexpr (std::is_same<T, half_t>::value) {
      finish_pk<NP>(st, out);
    } else {
      float y[2 * NP];
      finish<NP>(x, st, y);
      typedef T t2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int c = 0; c < NP; ++c) out[c] = __builtin_bit_cast(uint32_t, (t2){from_f<T>(y[2 * c]), from_f<T>(y[2 * c + 1])});
    }
  }
};
template <int NV> __device__ __forceinline__ void gelu_h16_n(float (&x)[NV]) {
  static_assert(NV % 2 == 0, "pairs");
  GeluH16::State<NV / 2> st;
  GeluH16::begin<NV / 2, 0>(x, st);
  GeluH16::finish<NV / 2>(x, st, x);
}
// GELU of NV values -> NV/2 dwords of 16-bit pairs in storage type T (the MFMA operand / store format)
template <typename T, int NV> __device__ __forceinline__ void gelu_h16_frag(const float (&x)[NV], uint32_t (&out)[NV / 2]) {
  static_assert(NV % 2 == 0, "pairs");
  GeluH16::State<NV / 2> st;
  GeluH16::begin<NV / 2, 0>(x, st);
  GeluH16::finish_frag<T, NV / 2>(x, st, out);
}

#ifndef GCV_GELU_H16
#define GCV_GELU_H16 1        // act4n evaluates the 16-bit GELU with gelu_h16_n (0: the packed-fp32 polynomial, gelu_pk_n)
#endif
#ifndef GCV_GELU_SCALAR
#define GCV_GELU_SCALAR 0     // 1: act4n evaluates the 16-bit GELU with gelu_fma_n (for translation units built with -fno-slp-vectorize)
#endif

// activation of NG groups of four consecutive channels (bias already added); for 16-bit GELU the 2*NG packed
// pairs are the independent chains of gelu_pk_n
template <int ACT, typename T, int NG> __device__ __forceinline__ void act4n(float (&v)[NG][4]) {
#ifndef GCV_GELU_EXACT
  if (ACT == ACT_GELU && sizeof(T) == 2 && GCV_GELU_H16 && !GCV_GELU_SCALAR) {
    float x[4 * NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) x[4 * g + e] = v[g][e];
    gelu_h16_n<4 * NG>(x);
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[g][e] = x[4 * g + e];
    return;
  }
  if (ACT == ACT_GELU && sizeof(T) == 2 && GCV_GELU_SCALAR) {
    float x[4 * NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) x[4 * g + e] = v[g][e];
    gelu_fma_n<4 * NG>(x);
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[g][e] = x[4 * g + e];
    return;
  }
  if (ACT == ACT_GELU && sizeof(T) == 2) {
    f32x2 x[2 * NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      x[2 * g] = (f32x2){v[g][0], v[g][1]};
      x[2 * g + 1] = (f32x2){v[g][2], v[g][3]};
    }
    gelu_pk_n<2 * NG>(x);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      v[g][0] = x[2 * g][0]; v[g][1] = x[2 * g][1]; v[g][2] = x[2 * g + 1][0]; v[g][3] = x[2 * g + 1][1];
    }
    return;
  }
#endif
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[g][e] = act_fn<ACT>(v[g][e]);
}
// NG groups that share one bias vector (the MI token blocks of a wave)
template <int ACT, typename T, int NG> __device__ __forceinline__ void bias_act4n(float (&v)[NG][4], const f32x4 b) {
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[g][e] += b[e];
  act4n<ACT, T, NG>(v);
}

template <int ACT, typename T> __device__ __forceinline__ void bias_act4(float (&v)[4], const f32x4 b) {
  float w[1][4] = {{v[0], v[1], v[2], v[3]}};
  bias_act4n<ACT, T, 1>(w, b);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = w[0][e];
}

// staged-row length (dwords) of the epilogue tile: BN elements + a pad that makes the 32 token rows a
// wave writes land on distinct banks (fp32: stride = 4*odd for ds_write_b128, 16-bit: 2*odd for b64)
template <int BN, bool F32> struct StageRow { static constexpr int dwords = F32 ? BN + 4 : BN / 2 + 2; };

template <typename T, int BM, int BN, int BKB, int EPI> struct GemmSmem {
  static constexpr bool kStageF32 = (sizeof(T) == 4) || EPI == EPI_RESID || EPI == EPI_SPLITK;
  static constexpr int kMain = 2 * (BM + BN) * BKB;
  static constexpr int kEpiBG = BM * StageRow<BN, kStageF32>::dwords * 4;     // bias | gamma broadcast rows live here
  static constexpr int kEpi = kEpiBG + 2 * BN * 4;
  static constexpr int bytes = kMain > kEpi ? kMain : kEpi;
};

template <typename T, int BM, int BN, int WM, int WN, int BKB, int AMODE, int EPI, int ACT>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmArgs g) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int CPR = BKB / 16;                      // 16-byte chunks per LDS row
  constexpr int BK = CPR * EPC;                      // K elements per tile
  constexpr int SW_SHIFT = (CPR == 4) ? 2 : 1;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per workgroup");
  static_assert(BKB == 64 || BKB == 128, "LDS rows are 64 or 128 bytes");
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int A_CH = BM * CPR, B_CH = BN * CPR;
  constexpr int A_SLOTS = (A_CH + 255) / 256, B_SLOTS = (B_CH + 255) / 256;
  constexpr int A_BYTES = BM * BKB, B_BYTES = BN * BKB;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WM;
  const int wn0 = (wave % WAVES_N) * WN;

  const int ntn = (g.N + BN - 1) / BN;
  const int ntm = (g.M + BM - 1) / BM;
  const int bid = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (bid / ntn) * BM;
  const int n0 = (bid % ntn) * BN;

  const int kbeg = (EPI == EPI_SPLITK) ? blockIdx.y * g.k_per_split : 0;
  const int kend = (EPI == EPI_SPLITK) ? min(g.K, kbeg + g.k_per_split) : g.K;
  const int nkt = (kend - kbeg + BK - 1) / BK;

  const T* __restrict__ Ap = (const T*)g.A;
  const T* __restrict__ Wp = (const T*)g.Wt;

  // ---- per-slot constants (row/chunk of each 16-byte piece this thread stages) ----
  int a_coff[A_SLOTS];
  bool a_ok[A_SLOTS];
  int64_t a_base[A_SLOTS];        // PLAIN: row*lda ; im2col: image base (b*H*W)
  int a_y[A_SLOTS], a_x[A_SLOTS]; // im2col: conv-output coordinates
#pragma unroll
  for (int s = 0; s < A_SLOTS; ++s) {
    const int idx = tid + s * 256;
    const int row = idx / CPR;
    a_coff[s] = (idx % CPR) * EPC;
    const int m = m0 + row;
    a_ok[s] = (idx < A_CH) && (m < g.M);
    a_base[s] = 0; a_y[s] = 0; a_x[s] = 0;
    if (AMODE == A_PLAIN) {
      a_base[s] = (int64_t)m * g.lda;
    } else if (AMODE == A_IM2COL3_POOL) {
      // m = ((b*Hp + yo)*Wp + xo)*4 + (dy*2+dx): 4 consecutive rows = one 2x2 pool window
      const int Hp = g.H >> 1, Wp2 = g.W >> 1;
      const int q = m & 3, p = m >> 2;
      const int xo = p % Wp2, t = p / Wp2;
      const int yo = t % Hp, b = t / Hp;
      a_y[s] = 2 * yo + (q >> 1);
      a_x[s] = 2 * xo + (q & 1);
      a_base[s] = (int64_t)b * g.H * g.W;
    } else {  // A_IM2COL3_S2: m = (b*Ho + yo)*Wo + xo, input row = 2*yo + ky - 1
      const int Ho = g.H >> 1, Wo = g.W >> 1;
      const int xo = m % Wo, t = m / Wo;
      const int yo = t % Ho, b = t / Ho;
      a_y[s] = 2 * yo;
      a_x[s] = 2 * xo;
      a_base[s] = (int64_t)b * g.H * g.W;
    }
  }
  int b_coff[B_SLOTS];
  bool b_ok[B_SLOTS];
  int64_t b_base[B_SLOTS];
#pragma unroll
  for (int s = 0; s < B_SLOTS; ++s) {
    const int idx = tid + s * 256;
    const int row = idx / CPR;
    b_coff[s] = (idx % CPR) * EPC;
    const int n = n0 + row;
    b_ok[s] = (idx < B_CH) && (n < g.N);
    b_base[s] = (int64_t)n * g.K;
  }

  u32x4 areg[A_SLOTS], breg[B_SLOTS];

  // bias / layer-scale of this tile's BN channels: one f32x4 per lane of the first BN/2 lanes, fetched now so the
  // latency hides under the main loop; broadcast through LDS in the epilogue (a per-group global load there is a
  // dependent ~1 us latency, 4*NI*MI of them per tile)
  f32x4 pre = {0.f, 0.f, 0.f, 0.f};
  if (EPI != EPI_SPLITK) {
    if (tid < BN / 4) {
      const int n = n0 + 4 * tid;
      const int bi = (EPI == EPI_CONVT) ? (n & ((1 << g.cout_log2) - 1)) : n;
      if (g.bias && n < g.N) pre = *(const f32x4*)(g.bias + bi);
    } else if (EPI == EPI_RESID && tid < BN / 2) {
      const int n = n0 + 4 * (tid - BN / 4);
      if (n < g.N) pre = *(const f32x4*)(g.gamma + n);
    }
  }
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  auto fetch = [&](int kt) {
    const int k0 = kbeg + kt * BK;
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) {
      const int k = k0 + a_coff[s];
      u32x4 v = zero4;
      if (a_ok[s] && k < kend) {
        if (AMODE == A_PLAIN) {
          v = *(const u32x4*)(Ap + a_base[s] + k);
        } else {
          const int tap = k >> g.cin_log2;
          const int ci = k & ((1 << g.cin_log2) - 1);
          const int ky = (tap * 11) >> 5;          // tap / 3 for tap in 0..8
          const int kx = tap - 3 * ky;
          const int iy = a_y[s] + ky - 1, ix = a_x[s] + kx - 1;
          if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
            v = *(const u32x4*)(Ap + (((a_base[s] + (int64_t)iy * g.W + ix) << g.cin_log2) + ci));
        }
      }
      areg[s] = v;
    }
#pragma unroll
    for (int s = 0; s < B_SLOTS; ++s) {
      const int k = k0 + b_coff[s];
      u32x4 v = zero4;
      if (b_ok[s] && k < kend) v = *(const u32x4*)(Wp + b_base[s] + k);
      breg[s] = v;
    }
  };

  auto stage = [&](int buf) {
    unsigned char* sA = smem + buf * (A_BYTES + B_BYTES);
    unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) {
      const int idx = tid + s * 256;
      if (A_CH % 256 == 0 || idx < A_CH) {
        const int row = idx / CPR, c = idx % CPR;
        *(u32x4*)(sA + row * BKB + ((c ^ ((row >> SW_SHIFT) & (CPR - 1))) << 4)) = areg[s];
      }
    }
#pragma unroll
    for (int s = 0; s < B_SLOTS; ++s) {
      const int idx = tid + s * 256;
      if (B_CH % 256 == 0 || idx < B_CH) {
        const int row = idx / CPR, c = idx % CPR;
        *(u32x4*)(sB + row * BKB + ((c ^ ((row >> SW_SHIFT) & (CPR - 1))) << 4)) = breg[s];
      }
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  if (nkt > 0) {
    fetch(0);
    stage(0);
  }
  __syncthreads();

  const int lr = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) fetch(kt + 1);
    const unsigned char* sA = smem + (kt & 1) * (A_BYTES + B_BYTES);
    const unsigned char* sB = sA + A_BYTES;
    // fragments of chunk pair t+1 are read while the MFMAs of pair t run (register double buffer)
    u32x4 af[2][MI], bf[2][NI];
    auto read_frags = [&](int t, int slot) {
      const int c = 2 * t + lh;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm0 + i * 32 + lr;
        af[slot][i] = *(const u32x4*)(sA + row * BKB + ((c ^ ((row >> SW_SHIFT) & (CPR - 1))) << 4));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wn0 + j * 32 + lr;
        bf[slot][j] = *(const u32x4*)(sB + row * BKB + ((c ^ ((row >> SW_SHIFT) & (CPR - 1))) << 4));
      }
    };
    read_frags(0, 0);
#pragma unroll
    for (int t = 0; t < CPR / 2; ++t) {
      if (t + 1 < CPR / 2) read_frags(t + 1, (t + 1) & 1);
      // Wt fragment is the MFMA "A" operand: the accumulator then has the TOKEN on the lane and 4
      // consecutive output channels in 4 consecutive registers (D row = channel, D col = token).
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) Mfma<T>::run(bf[t & 1][j], af[t & 1][i], acc[i][j]);
    }
    if (kt + 1 < nkt) stage((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------------------------
  // acc[i][j][r]: token  m = m0 + wm0 + 32 i + (lane & 31)
  //               channel n = n0 + wn0 + 32 j + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  // 1) bias / activation / layer-scale in registers, 2) stage the tile in LDS as [token][channel],
  // 3) read it back row-wise so every global store is 8-16 B per lane over contiguous channels.
  constexpr bool STAGE_F32 = GemmSmem<T, BM, BN, BKB, EPI>::kStageF32;
  constexpr int SROW = StageRow<BN, STAGE_F32>::dwords;       // dwords per staged row
  uint32_t* sC = reinterpret_cast<uint32_t*>(smem);
  float* sBG = reinterpret_cast<float*>(smem + GemmSmem<T, BM, BN, BKB, EPI>::kEpiBG);
  if (EPI != EPI_SPLITK) {
    if (tid < BN / 2) *(f32x4*)(sBG + 4 * tid) = pre;
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = wn0 + j * 32 + 8 * q + 4 * lh;
      float v[MI][4];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = acc[i][j][4 * q + e];
      if (EPI != EPI_SPLITK) {
        const f32x4 bv = *(const f32x4*)(sBG + nl);
        bias_act4n<ACT, T, MI>(v, bv);
        if (EPI == EPI_RESID) {
          const f32x4 gv = *(const f32x4*)(sBG + BN + nl);
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] *= gv[e];
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int ml = wm0 + i * 32 + lr;
        if (STAGE_F32) {
          f32x4 o = {v[i][0], v[i][1], v[i][2], v[i][3]};
          *(f32x4*)(sC + ml * SROW + nl) = o;
        } else {
          typedef T t4 __attribute__((ext_vector_type(4)));
          t4 o = {from_f<T>(v[i][0]), from_f<T>(v[i][1]), from_f<T>(v[i][2]), from_f<T>(v[i][3])};
          *(t4*)(sC + ml * SROW + (nl >> 1)) = o;
        }
      }
    }
  }
  __syncthreads();

  constexpr int PPR = BN / 4;                       // 4-channel pieces per row
  constexpr int ROWS = (EPI == EPI_POOL4) ? BM / 4 : BM;
  typedef T t4 __attribute__((ext_vector_type(4)));
  T* Cp = (T*)g.C;
  // the layer-scale residual is fetched first for all of this thread's pieces, from clamped (always
  // valid) addresses, so the loads are independent instead of one exposed latency per guarded iteration
  constexpr int NPIECES = (ROWS * PPR + 255) / 256;
  t4 rres[EPI == EPI_RESID ? NPIECES : 1];
  if (EPI == EPI_RESID) {
#pragma unroll
    for (int it = 0; it < NPIECES; ++it) {
      const int idx = tid + it * 256;
      const int rl = min(idx / PPR, ROWS - 1), pc = idx % PPR;
      const int m = min(m0 + rl, g.M - 1), n = min(n0 + 4 * pc, g.N - 4);
      rres[it] = *(const t4*)((const T*)g.resid + (int64_t)m * g.ldc + n);
    }
  }
#pragma unroll
  for (int it = 0; it < NPIECES; ++it) {
    const int idx = tid + it * 256;
    if (idx >= ROWS * PPR) break;
    const int rl = idx / PPR, pc = idx - rl * PPR;
    const int n = n0 + 4 * pc;
    const int m = m0 + ((EPI == EPI_POOL4) ? 4 * rl : rl);
    if (m >= g.M || n >= g.N) continue;
    f32x4 v;
    if (EPI == EPI_POOL4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f32x4 w;
        if (STAGE_F32) {
          w = *(const f32x4*)(sC + (4 * rl + k) * SROW + 4 * pc);
        } else {
          const t4 h = *(const t4*)(sC + (4 * rl + k) * SROW + 2 * pc);
          w = f32x4{to_f(h[0]), to_f(h[1]), to_f(h[2]), to_f(h[3])};
        }
        if (k == 0) v = w;
        else
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], w[e]);
      }
    } else if (STAGE_F32) {
      v = *(const f32x4*)(sC + rl * SROW + 4 * pc);
    }
    int64_t o;
    if (EPI == EPI_CONVT) {
      const int xx = m % g.W, tt = m / g.W;
      const int yy = tt % g.H, bb = tt / g.H;
      const int dq = n >> g.cout_log2, co = n & ((1 << g.cout_log2) - 1);
      o = ((((int64_t)bb * 2 * g.H + 2 * yy + (dq >> 1)) * 2 * g.W + 2 * xx + (dq & 1)) << g.cout_log2) + co;
    } else if (EPI == EPI_POOL4) {
      o = (int64_t)(m >> 2) * g.ldc + n;
    } else {
      o = (int64_t)m * g.ldc + n;
    }
    if (EPI == EPI_SPLITK) {
      *(f32x4*)(g.partial + ((int64_t)blockIdx.y * g.M + m) * g.N + n) = v;
      continue;
    }
    if (EPI == EPI_RESID) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += to_f(rres[it][e]);
    }
    if (STAGE_F32 || EPI == EPI_POOL4) {
      t4 out = {from_f<T>(v[0]), from_f<T>(v[1]), from_f<T>(v[2]), from_f<T>(v[3])};
      *(t4*)(Cp + o) = out;
    } else {
      *(t4*)(Cp + o) = *(const t4*)(sC + rl * SROW + 2 * pc);
    }
  }
}

// Host-side launcher: validates shapes, picks a tile config, launches on `stream`.
// Returns 0 on success (error text via gcv::get_error()).
template <typename T> int launch_gemm(const GemmArgs& g, int a_mode, int epi, hipStream_t stream);
// name of the kernel family a (dtype, a_mode, epi) launch resolves to — for profiling tables
const char* gemm_family_name(int a_mode, int epi);

}  // namespace gcv
