// Fused ConvNeXt MLP with LDS-DMA weight rings, C = 192 / 384 (16-bit storage):
//     out = resid + gamma * ( W2 . GELU( W1 . x_ln + b1 ) + b2 ),   W1 (4C, C), W2 (C, 4C)
//
// Both instances are opt-in experiments (GCV_MLP_RING192=1, GCV_FUSED_MLP384=1), correct and unit-tested, slower than
// what they would replace.  C = 192: 4-wave workgroups of 128 tokens, 78 KB of LDS -> two workgroups per CU, 96-register
// output tile; 288 us vs 279 us for the streaming kernel (fill-latency bound: two groups of prefetch are ~one L2->LDS
// latency).  C = 384 (text below): one workgroup per CU.
//
// Why: as two LDS-DMA GEMMs (pw1 + GELU, pw2 * gamma + res) this stage is bound by the L2 -> LDS fill rate of a CU —
// every 128-token tile pulls its share of both weight matrices once per 192 output columns (31 KB of operands per
// token) — and the 1536-wide hidden activation makes a round trip through HBM.  Fused, a workgroup owns 128 tokens for
// the whole MLP: 18 KB of weights per token, no hidden tensor, one launch instead of two.
//
// STATUS: correct (tests/test_kernels_gpu.py) but NOT used by default: at 256 images it takes 309 us against 216 us
// for the two GEMMs (without GELU still 257 us: with one wave per SIMD nothing covers the LDS latency in front of each
// batch of MFMAs; the matrix pipe is ~25 % busy).  Opt-in with GCV_FUSED_MLP384=1.  What it would need: fragment
// double-buffering in registers it does not have, or two waves per SIMD, which the 192-register output tile forbids.
//
// Shape of the kernel (one persistent 4-wave workgroup per CU, one wave per SIMD, 512 registers per lane):
//   * each wave owns 32 tokens: their x_ln rows are 24 MFMA B fragments in registers (96 VGPRs), the (32 x 384)
//     output tile is 12 accumulators (192 registers)
//   * the hidden dimension is walked in 48 groups of 32.  Group g needs W1[32g .. 32g+31, :] (24.6 KB) and
//     W2[:, 32g .. 32g+31] (24.6 KB, pre-packed per group with the hidden axis permuted for the accumulator-as-operand
//     trick).  Both stream through 3-slot LDS rings by LDS-DMA, two groups ahead, shared by the four waves: one counted
//     vmcnt wait + one raw barrier per group
//   * per group and wave:  N = b1[g+1] + W1[g+1] . x   (24 MFMAs, interleaved 1 : 10 with the vector instructions of)
//                          h = GELU(C)                  (packed to two 16-byte B fragments)
//                          acc2 += W2[:, g] . h         (24 MFMAs)
//     so the vector work of group g hides under the matrix work of group g+1
//   * W1 rows are 768 B: the 16-byte chunk index is XOR-swizzled by (row & 15); W2 group rows are 64 B with the
//     (row >> 2) & 3 swizzle of the GEMM ring.  Both are applied to the DMA source chunk.
//   * epilogue: the x registers are dead after the last GEMM1 of a tile and take the residual rows (48 independent
//     8-byte loads); (acc2 + b2) * gamma + resid is stored as 8-byte pieces straight from the accumulator layout
#pragma once
#include "../fused_mlp.h"

namespace gcv {

template <int C> struct MlpRingSmem {
  static constexpr int kW1Slot = 32 * C * 2;                // one hidden group of W1: 32 rows x C k
  static constexpr int kW2Slot = C * 64;                    // one hidden group of W2: C rows x 32 hidden
  static constexpr int kSlots = 3;
  static constexpr int kW1 = 0;
  static constexpr int kW2 = kSlots * kW1Slot;
  static constexpr int kB1 = kW2 + kSlots * kW2Slot;        // 4C floats
  static constexpr int kB2 = kB1 + 4 * C * 4;               // C floats
  static constexpr int kG = kB2 + C * 4;                    // C floats
  static constexpr int bytes = kG + C * 4;                  // 78336 (C = 192) / 156672 (C = 384)
};

#define GCV_M384_WAIT(N)  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory")

template <typename T, int C>
__global__ void __launch_bounds__(256, C == 192 ? 2 : 1) fused_mlp_ring_kernel(const MlpArgs a, const int ntiles) {
  static_assert(sizeof(T) == 2, "fused MLP is built for 16-bit storage");
  static_assert(C == 192 || C == 384, "ring kernel: C = 192 / 384");
  typedef MlpRingSmem<C> Mlp384Smem;
  constexpr int NG = 4 * C / 32, KP1 = C / 16, NO = C / 32;            // hidden groups of 32, GEMM1 k-steps, output tiles
  constexpr int W1ROW = C * 2;                                         // bytes per W1 row
  constexpr int NI = Mlp384Smem::kW1Slot / 4096;                       // DMA instructions per wave, ring part and stage (3 / 6)
  constexpr int W1S = Mlp384Smem::kW1Slot, W2S = Mlp384Smem::kW2Slot;
  constexpr int FB = (C == 192) ? 3 : 6;                               // fragments read per batch (register budget: 256 / 512)
  // swizzle of a W1 row's 16-byte chunk index: the 16 lanes of a ds_read_b128 group must hit 16 distinct units mod 16
  auto swz1 = [](int row, int chunk) {
    return C == 384 ? ((chunk & ~15) | ((chunk & 15) ^ (row & 15))) : ((chunk & ~7) | ((chunk & 7) ^ ((row >> 1) & 7)));
  };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const float* sB1 = reinterpret_cast<const float*>(smem + Mlp384Smem::kB1);
  const float* sB2 = reinterpret_cast<const float*>(smem + Mlp384Smem::kB2);
  const float* sG = reinterpret_cast<const float*>(smem + Mlp384Smem::kG);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // wave-uniform: DMA destinations live in SGPRs / M0
  const int lr = lane & 31, lh = lane >> 5;
  typedef T t4 __attribute__((ext_vector_type(4)));

  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (my_tiles <= 0) return;

  // ---- per-lane DMA source offsets (bytes inside one group's W1 rows / packed W2 block) ----
  int off1[NI], off2[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int q = wave + 4 * i;
    const int o = 1024 * q + 16 * lane;                                // byte offset in the lane-linear LDS image
    const int row = o / W1ROW, phys = (o - row * W1ROW) >> 4;
    off1[i] = row * W1ROW + (swz1(row, phys) << 4);
    const int r2 = 16 * q + (lane >> 2), p2 = lane & 3;
    off2[i] = r2 * 64 + ((p2 ^ ((r2 >> 2) & 3)) << 4);
  }
  const unsigned char* gW1 = (const unsigned char*)a.W1;
  const unsigned char* gW2 = (const unsigned char*)a.W2c;
  auto issue_w1 = [&](int n) {                                         // W1 rows of global step n's group -> slot n % 3
    const unsigned char* src = gW1 + (int64_t)(n % NG) * W1S;
    unsigned char* dst = smem + Mlp384Smem::kW1 + (n % 3) * W1S;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off1[i]),
                                       (__attribute__((address_space(3))) void*)(dst + (wave + 4 * i) * 1024), 16, 0, 0);
  };
  auto issue_w2 = [&](int n) {
    const unsigned char* src = gW2 + (int64_t)(n % NG) * W2S;
    unsigned char* dst = smem + Mlp384Smem::kW2 + (n % 3) * W2S;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off2[i]),
                                       (__attribute__((address_space(3))) void*)(dst + (wave + 4 * i) * 1024), 16, 0, 0);
  };

  // ---- biases / layer scale -> LDS; ring prologue: W1(0) | W1(1), W2(0) | W1(2), W2(1)  (12 DMAs per wave each) ----
  {
    float* sf = reinterpret_cast<float*>(smem + Mlp384Smem::kB1);
    for (int i = tid; i < 6 * C; i += 256)
      sf[i] = i < 4 * C ? a.b1[i] : (i < 5 * C ? a.b2[i - 4 * C] : a.gamma[i - 5 * C]);
  }
  issue_w1(0);
  issue_w1(0);                                                         // (twice: keeps every issue point at 2*NI DMAs per wave)
  issue_w1(1); issue_w2(0);
  issue_w1(2); issue_w2(1);
  __syncthreads();                                                     // biases visible (this also drains the prologue DMAs)

  const T* __restrict__ Xp = (const T*)a.X;
  const T* Rp = (const T*)a.resid;
  T* Op = (T*)a.out;
  const int key2 = (lr >> 2) & 3;                                      // W2 swizzle key (32 o + lr: o adds 0 mod 4 to row >> 2)

  int n0 = 0;                                                          // global step of the current tile's group 0
  for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x, n0 += NG) {
    const int64_t m = (int64_t)tile * 128 + wave * 32 + lr;
    const int64_t mc = m < a.M ? m : (int64_t)a.M - 1;                 // clamp: tail rows compute garbage, store nothing

    u32x4 xf[KP1];                                                     // x_ln fragments: k-step p, lane (token lr, half lh)
#pragma unroll
    for (int p = 0; p < KP1; ++p) xf[p] = *(const u32x4*)(Xp + mc * C + 16 * p + 8 * lh);

    f32x16 acc2[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[o][r] = 0.0f;

    auto load_b1 = [&](int g, f32x16& acc) {                          // accumulator := b1 of group g (token-on-lane layout)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *(const f32x4*)(sB1 + g * 32 + 8 * q + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * q + e] = bv[e];
      }
    };
    // FB k-steps of GEMM1 for the group in W1 slot `slot`: acc += W1[group][:, 16 p0 ..] . x
    auto gemm1_part = [&](int slot, int p0, f32x16& acc) {
      const unsigned char* base = smem + Mlp384Smem::kW1 + slot * W1S + lr * W1ROW;
      u32x4 wf[FB];
#pragma unroll
      for (int p = 0; p < FB; ++p) wf[p] = *(const u32x4*)(base + (swz1(lr, 2 * (p0 + p) + lh) << 4));
#pragma unroll
      for (int p = 0; p < FB; ++p) Mfma<T>::run(wf[p], xf[p0 + p], acc);
    };
    // GELU of 8 of the 16 hidden values a lane holds -> one 16-byte B fragment (C = 192: four values at a time)
    auto gelu_half = [&](const f32x16& acc, int half, u32x4& hf) {
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        float hv[1][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[0][e] = acc[4 * (2 * half + qq) + e];
        act4n<ACT_GELU, T, 1>(hv);
        const t4 h4 = {from_f<T>(hv[0][0]), from_f<T>(hv[0][1]), from_f<T>(hv[0][2]), from_f<T>(hv[0][3])};
        const uint2 pk = __builtin_bit_cast(uint2, h4);
        hf[2 * qq] = pk.x;
        hf[2 * qq + 1] = pk.y;
      }
    };
    // GEMM2, k-step s (16 of the group's 32 hidden), output tiles o0 .. o0+FB-1: acc2[o] += W2[32 o .., group][:, 16 s ..] . h_s
    auto gemm2_part = [&](int slot, int s, int o0, const u32x4& hf) {
      const unsigned char* base = smem + Mlp384Smem::kW2 + slot * W2S + lr * 64 + (((2 * s + lh) ^ key2) << 4);
      u32x4 w2f[FB];
#pragma unroll
      for (int o = 0; o < FB; ++o) w2f[o] = *(const u32x4*)(base + (o0 + o) * 32 * 64);
#pragma unroll
      for (int o = 0; o < FB; ++o) Mfma<T>::run(w2f[o], hf, acc2[o0 + o]);
    };
    // one step: C holds b1 + W1[g] . x ; computes N for g+1 (unless last), h(g), acc2 += W2[:, g] . h(g)
    auto step = [&](int g, f32x16& Cacc, f32x16& Nacc) {
      const int n = n0 + g;
      GCV_M384_WAIT(2 * NI);                                           // W1(n+1), W2(n) landed everywhere; step n-1 is over
      issue_w1(n + 3);                                                 // (past the last tile these fetch groups nobody reads:
      issue_w2(n + 2);                                                 //  the count stays at 2*NI DMAs per wave and step)
      u32x4 hf[2];
      if (g + 1 < NG) {
        load_b1(g + 1, Nacc);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int part = 0; part < KP1 / 2 / FB; ++part) {
            gemm1_part((n + 1) % 3, (KP1 / 2) * half + FB * part, Nacc);
            __builtin_amdgcn_sched_barrier(0);
          }
          gelu_half(Cacc, half, hf[half]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        gelu_half(Cacc, 0, hf[0]);
        gelu_half(Cacc, 1, hf[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int o0 = 0; o0 < NO; o0 += FB) {
          gemm2_part(n % 3, s, o0, hf[s]);
          __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- group 0 of this tile: its W1 rows are resident (prologue / the wait of the previous tile's last step) ----
    f32x16 accA, accB;
    load_b1(0, accA);
#pragma unroll
    for (int p0 = 0; p0 < KP1; p0 += FB) {
      gemm1_part(n0 % 3, p0, accA);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int g = 0; g < NG - 2; g += 2) {
      step(g, accA, accB);
      step(g + 1, accB, accA);
    }
    step(NG - 2, accA, accB);
    // the x registers are dead: they take the residual rows, in flight under the last group's GELU and GEMM2
    t4 rres[NO][4];
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int q = 0; q < 4; ++q) rres[o][q] = *(const t4*)(Rp + mc * C + 32 * o + 8 * q + 4 * lh);
    step(NG - 1, accB, accA);

    // ---- epilogue: (acc2 + b2) * gamma + resid -> 16-bit, 8-byte pieces ----
    // (the b2 / gamma pointers go through an empty asm so that their 384 loop-invariant LDS reads are not hoisted out of
    //  the tile loop and spilled)
    const float* sB2t = sB2;
    const float* sGt = sG;
    asm volatile("" : "+v"(sB2t), "+v"(sGt));
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nn = 32 * o + 8 * q + 4 * lh;
        const f32x4 bv = *(const f32x4*)(sB2t + nn);
        const f32x4 gv = *(const f32x4*)(sGt + nn);
        t4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = from_f<T>(fmaf(acc2[o][4 * q + e] + bv[e], gv[e], to_f(rres[o][q][e])));
        if (m < a.M) *(t4*)(Op + m * C + nn) = o4;
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // no DMA may outlive the workgroup's LDS
}

template <typename T> int launch_fused_mlp_ring(const MlpArgs& a, int C, hipStream_t s);

}  // namespace gcv
