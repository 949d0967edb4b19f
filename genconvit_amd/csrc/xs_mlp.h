// Fused ConvNeXt MLP for the narrow stages (C = 96, 192), 16-bit storage, x-stationary with an LDS-DMA weight ring:
//     out = resid + gamma * ( W2 . GELU( W1 . x_ln + b1 ) + b2 )            (timm ConvNeXtBlock; call sites
//     model/genconvit_ed.py:82-83, model/genconvit_vae.py:111-112 of the reference)
//
// One persistent workgroup of 8 waves per CU.  A pass = 256 tokens, 32 per wave: the wave's x_ln rows are MFMA B fragments
// in registers (C/4 VGPRs), its 32 x C output tile is C/32 accumulators (C/2 VGPRs), and the 4C hidden channels are walked
// in chunks of 32.  A chunk's weights are ONE contiguous record, packed at load time in the order the fragments are read
// (pack_xs_mlp): [W1 rows 32kc .. +31 as C/16 A fragments | W2[:, chunk] as 2 x C/32 A fragments, hidden axis permuted for
// the accumulator-as-operand hand-off], 12 KB (C = 96) / 24 KB (C = 192).  Records stream through a ring of D slots by
// LDS-DMA (whole lines, no swizzle: a fragment read is `base + immediate`, lane-linear) and the walk simply continues into
// the next pass, so the ring never drains.
//
// Software pipeline of a wave, per step g (one barrier):
//     acc1' = b1[g] + W1[g] . x                    C/16   MFMAs   (fragments from slot g)
//     h(g-1) = GELU(acc1)  -> two B fragments      ~170 vector instructions (polynomial on the packed-fp16 pipe, gemm.h)
//     acc2  += W2[:, g-2] . h(g-2)                 2 C/32 MFMAs   (fragments from slot g-2, which therefore lives two
//                                                                  steps longer than in a plain GEMM ring)
// issued in fenced sub-blocks of four MFMAs, each with the fragment reads of the next sub-block and its share of the GELU
// (sched_group_barrier 1 MFMA : 1 LDS read : n vector instructions): measured on gfx950, vector instructions beside
// MFMAs cost ~4-5 issue cycles each however many waves share the SIMD (profiles/micro/mfma_valu_overlap.hip), so what
// matters is that no wave ever sits in a vector-only or a matrix-only phase.
//
// What it replaces: fused_mlp_kernel (C = 192: register-staged 96-wide chunks, __syncthreads per chunk, 261 us at 256
// images) and fused_mlp_res_kernel (C = 96: weights resident in LDS).
#pragma once
#include <type_traits>

#include "gemm.h"

namespace gcv {

struct XsMlpArgs {
  const void* X;        // (M, C) LayerNorm'ed dw-conv output, token-major
  const void* Wp;       // packed records, pack_xs_mlp: [4C/32][(C/16 + 2 C/32) * 512] elements
  const float* b1;      // (4C)
  const float* b2;      // (C)
  const float* gamma;   // (C)
  const void* resid;    // (M, C) block input (may alias out)
  void* out;            // (M, C)
  int M;
  // Optional (last block of a stage): the stage boundary's LayerNorm2d + 2x2 space-to-depth in the epilogue, as in
  // MlpArgs (fused_mlp.h): `out` is then the patchified (M/4, 4C) operand of the down-sampling GEMM
  const float* lnp_w = nullptr;
  const float* lnp_b = nullptr;
  float lnp_eps = 0.0f;
  int lnp_nseg = 0;
  int lnp_tok0[4] = {0, 0, 0, 0};
  int lnp_hw[4] = {1, 1, 1, 1};
  int lnp_wd[4] = {1, 1, 1, 1};
  int lnp_out0[4] = {0, 0, 0, 0};
};

// W1 (4C, C) and W2 (C, 4C) row-major of type S on the device -> records of T.  Fragment f of a record is 512 elements
// [kh = 2][r = 32][e = 8]: f < C/16: W1[32kc + r][16f + 8kh + e]; f = C/16 + s * C/32 + nb: W2[32nb + r][32kc + hid] with
// hid = 16s + 8(e >> 2) + 4kh + (e & 3) — the order in which a lane's GELU'd accumulator registers supply k
template <typename T, typename S>
__global__ void __launch_bounds__(256) pack_xs_mlp_kernel(const T* __restrict__ w1, const S* __restrict__ w2,
                                                          T* __restrict__ out, int C) {
  const int KP1 = C / 16, NO = C / 32, NF = KP1 + 2 * NO;
  const int64_t total = (int64_t)(4 * C / 32) * NF * 512;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i & 7), r = (int)((i >> 3) & 31), kh = (int)((i >> 8) & 1);
  const int64_t t = i >> 9;
  const int f = (int)(t % NF), kc = (int)(t / NF);
  float v;
  if (f < KP1) {
    v = (float)w1[(int64_t)(32 * kc + r) * C + 16 * f + 8 * kh + e];
  } else {
    const int s = (f - KP1) / NO, nb = (f - KP1) % NO;
    const int hid = 16 * s + 8 * (e >> 2) + 4 * kh + (e & 3);
    v = (float)w2[(int64_t)(32 * nb + r) * 4 * C + 32 * kc + hid];
  }
  out[i] = from_f<T>(v);
}

template <int C> struct XsMlpCfg {
  static constexpr int KP1 = C / 16, NO = C / 32, NM = KP1 + 2 * NO;    // MFMAs (= fragments) per chunk: 12 / 24
  static constexpr int REC = NM * 1024;                                 // record bytes
  static constexpr int NKC = 4 * C / 32;                                // chunks: 12 / 24
  static constexpr int D = C == 96 ? 10 : 6;                            // ring slots (120 KB / 144 KB)
  static constexpr int PPW = 3;                                         // DMA pieces per loader wave and chunk
  static constexpr int NLW = NM / PPW;                                  // loader waves: 4 / 8
  static constexpr int kB1 = D * REC;                                   // b1 (4C floats)
  static constexpr int kBG = kB1 + 4 * C * 4;                           // b2 | gamma (2C floats)
  static constexpr int kLn = kBG + 2 * C * 4;                           // LNP variant: LayerNorm weight | bias (2C floats)
  static constexpr int bytes = kLn + 2 * C * 4;
  static constexpr int NSB = NM / 4;                                    // sub-blocks of four MFMAs per step: 3 / 6
};


#define GCV_XM_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory")

template <typename T, int C, bool LNP = false>
__global__ void __launch_bounds__(512, 2) xs_mlp_kernel(const XsMlpArgs a, const int npass) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  static_assert(C == 96 || C == 192, "narrow ConvNeXt stages");
  typedef XsMlpCfg<C> CF;
  constexpr int KP1 = CF::KP1, NO = CF::NO, NM = CF::NM, REC = CF::REC, NKC = CF::NKC, D = CF::D, PPW = CF::PPW;
  constexpr int NLW = CF::NLW, NSB = CF::NSB;
  // Ring, TWO steps per barrier: chunk n is needed at step n (W1 part) and at step n + 2 (W2 part).  At the head of the pair
  // (n, n+1), n even, chunks n-2 .. n+1 are live, the slots of n-4 and n-3 are dead (W2[n-3] was consumed at step n-1) and
  // take chunks n+2 and n+3; the only DMAs outstanding before that refill are those of chunks n and n+1, issued one pair
  // earlier (~6000 cycles): vmcnt(0).  4 live + 2 incoming = 6 slots.
  static_assert(D >= 6 && NKC % 2 == 0, "ring: four live chunks and two incoming; passes are whole pairs of steps");
  typedef T t4 __attribute__((ext_vector_type(4)));

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;

  const int my_passes = (npass - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (my_passes <= 0) return;

  // ---- biases / layer scale -> LDS (before any DMA is in flight)
  {
    float* sf = reinterpret_cast<float*>(smem + CF::kB1);
    for (int i = tid; i < 6 * C; i += 512) sf[i] = i < 4 * C ? a.b1[i] : (i < 5 * C ? a.b2[i - 4 * C] : a.gamma[i - 5 * C]);
    if (LNP) {
      float* sl = reinterpret_cast<float*>(smem + CF::kLn);
      for (int i = tid; i < 2 * C; i += 512) sl[i] = i < C ? a.lnp_w[i] : a.lnp_b[i - C];
    }
  }
  __syncthreads();

  // ---- ring: loader wave w issues pieces w, w + NLW, w + 2 NLW of a record
  const bool loader = wave < NLW;
  const unsigned char* const wsrc = (const unsigned char*)a.Wp + wave * 1024;
  const unsigned lane16 = (unsigned)lane * 16u;
  int gw = 0, slot_w = 0;                                  // chunk (mod NKC) and slot of the next refill
  auto issue = [&]() {
    if (loader) {
      const unsigned char* src = wsrc + (int64_t)gw * REC;
      unsigned char* dst = smem + slot_w * REC + wave * 1024;
#pragma unroll
      for (int i = 0; i < PPW; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * NLW * 1024 + lane16),
                                         (__attribute__((address_space(3))) void*)(dst + i * NLW * 1024), 16, 0, 0);
    }
    gw = gw + 1 == NKC ? 0 : gw + 1;
    slot_w = slot_w + 1 == D ? 0 : slot_w + 1;
  };
  issue();
  issue();                                                 // chunks 0 and 1

  const float* const sb1 = reinterpret_cast<const float*>(smem + CF::kB1) + 4 * lh;
  const float* const sB2_ = reinterpret_cast<const float*>(smem + CF::kBG);
  const float* const sG_ = sB2_ + C;
  const T* __restrict__ Xp = (const T*)a.X;
  const T* Rp = (const T*)a.resid;
  T* Op = (T*)a.out;

  int slot1 = 0;                                           // slot of the chunk whose W1 part this step reads
  auto slot_dec = [&](int s, int k) { return s - k < 0 ? s - k + D : s - k; };

  u32x4 xf[KP1];
  auto load_x = [&](int64_t mc) {
    const T* xp = Xp + mc * C + 8 * lh;
#pragma unroll
    for (int p = 0; p < KP1; ++p)
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(xf[p]) : "v"(xp), "n"(p * 32) : "memory");
  };
  auto row_of = [&](int it) { return ((int64_t)((int)blockIdx.x + it * (int)gridDim.x) * 8 + wave) * 32 + lr; };
  {
    const int64_t m0 = row_of(0);
    load_x(m0 < a.M ? m0 : (int64_t)a.M - 1);
  }
  XM_STAMP(0);

  for (int it = 0; it < my_passes; ++it) {
    const int64_t m = row_of(it);
    const int64_t mc = m < a.M ? m : (int64_t)a.M - 1;     // clamp: tail rows compute garbage, store nothing
    // the x loads of this pass were issued by hand (before the loop / at the end of the previous epilogue): wait for them
    // here.  (vmcnt(0) also drains the ring's DMAs once per pass; the counted waits below resume from there.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int p = 0; p < KP1; ++p) asm volatile("" : "+v"(xf[p]));

    f32x16 acc2[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[o][r] = 0.0f;

    // ---- GELU of 8 of a lane's 16 hidden values (half h: registers 8h .. 8h+7) in three parts -> one B fragment
    GeluH16::State<4> gst;
    float gx[8];
    auto gelu_a = [&](const f32x16& acc, int half) {
#pragma unroll
      for (int c = 0; c < 8; ++c) gx[c] = acc[8 * half + c];
      if (!(GCV_XM_ABLATE & 1)) GeluH16::begin<4, 5>(gx, gst);
    };
    auto gelu_b = [&]() { if (!(GCV_XM_ABLATE & 1)) GeluH16::horner<4, 4, 0>(gst); };
    auto gelu_c = [&](u32x4& hf) {
      uint32_t hw[4];
      if (!(GCV_XM_ABLATE & 1)) GeluH16::finish_frag<T, 4>(gx, gst, hw);
      else {
        typedef T t2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int c = 0; c < 4; ++c) hw[c] = __builtin_bit_cast(uint32_t, (t2){from_f<T>(gx[2 * c]), from_f<T>(gx[2 * c + 1])});
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) hf[c] = hw[c];
    };
    auto gelu_piece = [&](const f32x16& cur1, u32x4 (&hfc)[2], const int pc) {   // pc = 0 .. 5: A0 B0 C0 A1 B1 C1
      if (pc % 3 == 0) gelu_a(cur1, pc / 3);
      if (pc % 3 == 1) gelu_b();
      if (pc % 3 == 2) gelu_c(hfc[pc / 3]);
    };

    // ---- one step.  G1: GEMM1 of chunk g into nxt1; GL: GELU of cur1 into hfc; G2: GEMM2 of chunk g-2 from hfp.
    // MFMA op i of a step: i < KP1 -> GEMM1 k-step i; else GEMM2 (s, nb) = ((i - KP1) / NO, (i - KP1) % NO); its A
    // fragment is fragment i of the record in slot g (GEMM1) or slot g-2 (GEMM2).
    u32x4 wf[4];
    auto frag_ptr = [&](const int i, const unsigned char* s1, const unsigned char* s2) {
      return (const u32x4*)((i < KP1 ? s1 : s2) + i * 1024);
    };
    auto step = [&](f32x16& cur1, f32x16& nxt1, u32x4 (&hfp)[2], u32x4 (&hfc)[2], int kc, auto g1c, auto glc, auto g2c,
                    auto ringc) {
      constexpr bool G1 = decltype(g1c)::value, GL = decltype(glc)::value, G2 = decltype(g2c)::value;
      constexpr int RING = decltype(ringc)::value;                      // 0: drain step, 1: head of a pair, 2: its second step
      constexpr int I0 = G1 ? 0 : KP1, I1 = G2 ? NM : KP1;              // MFMA ops of this step
      if (RING == 1) {
        GCV_XM_WAIT(0);
        if (!(GCV_XM_ABLATE & 2)) { issue(); issue(); }
      }
      const unsigned char* s1 = smem + slot1 * REC + lane16;
      const unsigned char* s2 = smem + slot_dec(slot1, 2) * REC + lane16;
      if (G1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bv = *(const f32x4*)(sb1 + kc * 32 + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) nxt1[4 * q + e] = bv[e];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (I0 + i < I1) wf[i] = *frag_ptr(I0 + i, s1, s2);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int NB = (I1 - I0 + 3) / 4;                             // sub-blocks in this step
      constexpr int PPB = GL ? (6 + NB - 1) / NB : 0;                   // GELU pieces per sub-block
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        u32x4 w0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w0[i] = wf[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int nx = I0 + 4 * (j + 1) + i;
          if (nx < I1) wf[i] = *frag_ptr(nx, s1, s2);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int op = I0 + 4 * j + i;
          if (op < I1 && !(GCV_XM_ABLATE & 8)) {
            if (op < KP1) Mfma<T>::run(w0[i], xf[op], nxt1);
            else Mfma<T>::run(w0[i], hfp[(op - KP1) / NO], acc2[(op - KP1) % NO]);
          }
        }
        if (GL) {
#pragma unroll
          for (int pc = 0; pc < PPB; ++pc)
            if (j * PPB + pc < 6) gelu_piece(cur1, hfc, j * PPB + pc);
          constexpr int VPM = PPB * 8;                                  // ~30 vector instructions per GELU piece
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (RING) slot1 = slot1 + 1 == D ? 0 : slot1 + 1;
    };
    using TT = std::true_type;
    using FF = std::false_type;
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;

    f32x16 accA, accB;
    u32x4 hfA[2], hfB[2];
    if (it == 0) XM_STAMP(1);
    step(accA, accA, hfA, hfA, 0, TT{}, FF{}, FF{}, R1{});               // g = 0: GEMM1(0) -> A
    step(accA, accB, hfA, hfA, 1, TT{}, TT{}, FF{}, R2{});               // g = 1: GEMM1(1) -> B, h(0) -> hfA
    if (it == 0) XM_STAMP(2);
    // steady state, two steps per iteration: (cur A->hfB | GEMM2 hfA) then (cur B->hfA | GEMM2 hfB)
#pragma unroll 1
    for (int g = 2; g < NKC; g += 2) {
      step(accB, accA, hfA, hfB, g, TT{}, TT{}, TT{}, R1{});             // GEMM1(g) -> A, h(g-1) = GELU(B) -> hfB, GEMM2(g-2) hfA
      step(accA, accB, hfB, hfA, g + 1, TT{}, TT{}, TT{}, R2{});         // GEMM1(g+1) -> B, h(g) = GELU(A) -> hfA, GEMM2(g-1) hfB
    }
    if (it == 0) XM_STAMP(3);
    // residual rows into the (now dead) x registers: 16-byte pieces, lanes 0-31 channels 16p .. 16p+7, lanes 32-63 the
    // next eight; in flight under the two drain steps
    static_assert(NKC % 2 == 0, "two steps per iteration");
    u32x4 rres[KP1];
#pragma unroll
    for (int p = 0; p < KP1; ++p) rres[p] = *(const u32x4*)(Rp + mc * C + 16 * p + 8 * lh);
    // drain (the ring is not touched: no barrier): h(NKC-1) = GELU(B) -> hfB with GEMM2(NKC-2) hfA, then GEMM2(NKC-1) hfB.
    // slot1 already points one past the last chunk: GEMM2's "slot g-2" is slot1 - 2 for chunk NKC-2, slot1 - 1 for NKC-1
    step(accB, accA, hfA, hfB, 0, FF{}, TT{}, TT{}, R0{});
    slot1 = slot1 + 1 == D ? 0 : slot1 + 1;
    step(accB, accA, hfB, hfA, 0, FF{}, FF{}, TT{}, R0{});
    slot1 = slot_dec(slot1, 1);
    if (it == 0) XM_STAMP(4);

    // ---- epilogue: (acc2 + b2) * gamma + resid -> 16-bit, 16-byte pieces (v_permlane32_swap converts between a row's
    // 16-byte piece and the accumulator's (8q + 4lh) halves, it is its own inverse)
    {
      const float* sB2 = sB2_;
      const float* sG = sG_;
      asm volatile("" : "+v"(sB2), "+v"(sG));              // keeps the LDS reads inside the pass loop, unspilled
      if constexpr (LNP) {
        // out = LayerNorm2d(resid + gamma * (acc2 + b2)) at the patch position of token m (see fused_mlp_res.h)
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const u32x4 rw = rres[2 * o + pr];
            const auto rx = __builtin_amdgcn_permlane32_swap(rw[0], rw[2], false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(rw[1], rw[3], false, false);
            const t4 rq[2] = {__builtin_bit_cast(t4, uint2{rx[0], ry[0]}), __builtin_bit_cast(t4, uint2{rx[1], ry[1]})};
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              const int q = 2 * pr + d;
              const int n = 32 * o + 8 * q + 4 * lh;
              const f32x4 bv = *(const f32x4*)(sB2 + n);
              const f32x4 gv = *(const f32x4*)(sG + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float v = fmaf(acc2[o][4 * q + e] + bv[e], gv[e], to_f(rq[d][e]));
                acc2[o][4 * q + e] = v;
                s1 += v;
                s2 = fmaf(v, v, s2);
              }
            }
          }
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        const float mean = s1 * (1.0f / C);
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(fmaf(-mean, mean, s2 * (1.0f / C)), 0.0f) + a.lnp_eps);
        const float nmr = -mean * rstd;
        int sg = 0;
#pragma unroll
        for (int t = 1; t < 4; ++t) sg += (t < a.lnp_nseg && (int)mc >= a.lnp_tok0[t]) ? 1 : 0;
        const int local = (int)mc - a.lnp_tok0[sg];
        const int hw = a.lnp_hw[sg], wd = a.lnp_wd[sg];
        const int img = local / hw, rem = local - img * hw;
        const int py = rem / wd, px = rem - py * wd;
        const int64_t prow = (int64_t)a.lnp_out0[sg] + ((int64_t)img * (hw / wd / 2) + (py >> 1)) * (wd >> 1) + (px >> 1);
        T* const dst = Op + prow * (4 * C) + ((py & 1) * 2 + (px & 1)) * C;
        const float* sLw = reinterpret_cast<const float*>(smem + CF::kLn);
        asm volatile("" : "+v"(sLw));
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            uint2 pk[2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              const int q = 2 * pr + d;
              const int n = 32 * o + 8 * q + 4 * lh;
              const f32x4 wv = *(const f32x4*)(sLw + n);
              const f32x4 cv = *(const f32x4*)(sLw + C + n);
              t4 o4;
#pragma unroll
              for (int e = 0; e < 4; ++e) o4[e] = from_f<T>(fmaf(fmaf(acc2[o][4 * q + e], rstd, nmr), wv[e], cv[e]));
              pk[d] = __builtin_bit_cast(uint2, o4);
            }
            const auto sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
            const u32x4 w = {sx[0], sy[0], sx[1], sy[1]};
            if (m < a.M) *(u32x4*)(dst + 32 * o + 16 * pr + 8 * lh) = w;
          }
      } else
#pragma unroll
      for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const u32x4 rw = rres[2 * o + pr];
          const auto rx = __builtin_amdgcn_permlane32_swap(rw[0], rw[2], false, false);
          const auto ry = __builtin_amdgcn_permlane32_swap(rw[1], rw[3], false, false);
          const t4 rq[2] = {__builtin_bit_cast(t4, uint2{rx[0], ry[0]}), __builtin_bit_cast(t4, uint2{rx[1], ry[1]})};
          uint2 pk[2];
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const int q = 2 * pr + d;
            const int n = 32 * o + 8 * q + 4 * lh;
            const f32x4 bv = *(const f32x4*)(sB2 + n);
            const f32x4 gv = *(const f32x4*)(sG + n);
            t4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = from_f<T>(fmaf(acc2[o][4 * q + e] + bv[e], gv[e], to_f(rq[d][e])));
            pk[d] = __builtin_bit_cast(uint2, o4);
          }
          const auto sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
          const u32x4 w = {sx[0], sy[0], sx[1], sy[1]};
          if (m < a.M) *(u32x4*)(Op + m * C + 32 * o + 16 * pr + 8 * lh) = w;
        }
    }
    // next pass's x rows (the wait at the top of the loop covers them)
    if (it + 1 < my_passes) {
      const int64_t mn = row_of(it + 1);
      load_x(mn < a.M ? mn : (int64_t)a.M - 1);
    }
    if (it == 0) XM_STAMP(5);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no DMA may outlive the workgroup's LDS
  XM_STAMP(6);
}

template <typename T> int launch_xs_mlp(const XsMlpArgs& a, int C, hipStream_t s);
// W1 (4C, C) of T and W2 (C, 4C) of S (T or float) on the device -> the packed records
template <typename T, typename S> int launch_pack_xs_mlp(const T* w1, const S* w2, T* out, int C, hipStream_t s);
// C = 96 works (and is what the template was first written for) but measures 270 us at 256 images against 215 us for the
// LDS-resident fused_mlp_res_kernel: twelve steps of 12 MFMAs each pay a barrier and a ring refill per step.  It is
// instantiated in GCV_EXPERIMENTS builds only; the product path uses this kernel at C = 192.
#ifdef GCV_EXPERIMENTS
static inline bool xs_mlp_supported(int C) { return C == 96 || C == 192; }
#else
static inline bool xs_mlp_supported(int C) { return C == 192; }
#endif
static inline bool xs_mlp_default(int C) { return C == 192 || (C == 96 && exp_env("GCV_XS_MLP96") != nullptr); }
static inline size_t xs_mlp_packed_elems(int C) { return (size_t)(4 * C / 32) * (size_t)(C / 16 + 2 * (C / 32)) * 512; }

}  // namespace gcv
