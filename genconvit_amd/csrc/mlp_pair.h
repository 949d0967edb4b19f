// ConvNeXt MLP at C = 384 (stage 2) as a PAIR of kernels that share a fragment-major hidden tensor (16-bit storage):
//     hidden = GELU( W1 . x_ln + b1 )                     xs_pw1_kernel   (x-stationary, W1 streamed through an LDS ring)
//     out    = resid + gamma * ( W2 . hidden + b2 )       pw2f_kernel     (256-token tiles, both operands by linear LDS-DMA)
// (timm ConvNeXtBlock mlp.fc1 -> GELU -> mlp.fc2 -> gamma -> + shortcut; call sites model/genconvit_ed.py:82-83,
//  model/genconvit_vae.py:111-112 of the reference.)
//
// Why not the tile GEMM of gemm_glds.h: at K = 384 its 128x192 tile is six K steps followed by a GELU epilogue that is as
// long as the K loop, the two phases run back to back in the same waves, and the L2 -> LDS fill rate of a CU
// (~28 B/clk with 64-byte row pieces) bounds the K loop at ~half the matrix rate.  Here
//   * pw1 keeps a wave's 32 tokens of x_ln in registers as 24 MFMA B fragments for the whole kernel and streams W1 once
//     per 256 tokens: 24 KB per 32 hidden channels, i.e. 16 B/clk at the full matrix rate, fetched as whole lines because
//     the packed W1 is contiguous in the order the fragments are read.  The GELU of hidden chunk g-1 (~150 vector
//     instructions: polynomial on the packed-fp16 pipe, gemm.h GeluH16) rides between the 24 MFMAs of chunk g in fenced
//     sub-blocks, so there is no epilogue phase at all.  (Measured, DESIGN.md section 4.0: beside MFMAs every vector
//     instruction costs ~4 issue cycles whatever the arrangement, so what counts is how few there are: 8.5 per MFMA here,
//     12 in the tile GEMM's epilogue.)
//   * the hidden tensor is stored FRAGMENT-MAJOR: block (token block tb of 32, hidden chunk kc of 32) is 2 KB laid out
//     [kq = 4][token = 32][8 hidden] — one 16-byte piece per lane, exactly an MFMA operand fragment.  pw1 writes a block
//     with two contiguous 1 KB store instructions; pw2 fetches blocks with linear 1 KB LDS-DMA pieces (whole lines at a
//     K step of 32) and reads fragments with `base + immediate` ds_read_b128, conflict-free without a swizzle.
//   * W1 / W2 are packed once at load time in the same fragment order (pack_w1_frag / pack_w2_frag).
//   * rings are sized in time, not in stages: a DMA takes 1-2 us from issue to landing under load, so pw1 runs 6 x 24 KB
//     (four chunks ahead), pw2 3 x 40 KB or 5 x 28 KB; one workgroup of 8 waves per CU.
// Measured at 256 images (rocprofv3): pw1 80 us (tile GEMM 106), pw2 83 us (tile GEMM ~100); timelines and ablations:
// profiles/xs_stamps.py, p2_stamps.py, GCV_XS_ABLATE / GCV_P2_ABLATE builds, DESIGN.md section 4.0.
#pragma once
#include <type_traits>

#include "gemm.h"

namespace gcv {

struct MlpPairArgs {
  const void* X;        // (M, C) LayerNorm'ed dw-conv output, token-major
  const void* W1f;      // packed W1: [4C/32][C/16][2][32][8]
  const float* b1;      // (4C)
  const void* W2f;      // packed gamma * W2: [4C/32][C/32][4][32][8] (pack_w2_frag_kernel folds the layer scale in)
  const float* b2;      // (C)
  const float* gamma;   // (C)
  const void* resid;    // (M, C) block input (may alias out)
  void* out;            // (M, C)
  void* hidden;         // workspace: ceil(M/32) * (4C/32) * 2048 bytes
  int M;
};

// ---------------------------------------------------------------------------------------------- weight packers
// W1 (4C, C) row-major T -> [g = n/32][p = k/16][kh][r = n%32][e = k%8]: chunk g is 32*C contiguous elements and the
// MFMA A fragment of k-step p is the 1 KB at (g*KP + p) * 1024 B, lane (r, kh) at + (kh*32 + r) * 16 B
template <typename T, typename S>
__global__ void __launch_bounds__(256) pack_w1_frag_kernel(const S* __restrict__ w1, T* __restrict__ out, int C) {
  const int64_t total = (int64_t)4 * C * C;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i & 7), r = (int)((i >> 3) & 31), kh = (int)((i >> 8) & 1);
  const int64_t t = i >> 9;
  const int KP = C / 16;
  const int p = (int)(t % KP), g = (int)(t / KP);
  out[i] = from_f<T>((float)w1[(int64_t)(32 * g + r) * C + 16 * p + 8 * kh + e]);
}
// W2 (C, 4C) row-major -> [kc = k/32][nb = n/32][kq = (k%32)/8][r = n%32][e = k%8], with the block's layer scale folded
// in: the packed weight is T(gamma[n] * W2[n][k]) (rounded once, from the source precision), so that pw2 accumulates
// gamma * (W2 . h) directly and its accumulator can start at gamma * b2 and take the residual at any time
template <typename T, typename S>
__global__ void __launch_bounds__(256) pack_w2_frag_kernel(const S* __restrict__ w2, const float* __restrict__ gamma,
                                                           T* __restrict__ out, int C) {
  const int64_t total = (int64_t)4 * C * C;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i & 7), r = (int)((i >> 3) & 31), kq = (int)((i >> 8) & 3);
  const int64_t t = i >> 10;
  const int NB = C / 32;
  const int nb = (int)(t % NB), kc = (int)(t / NB);
  out[i] = from_f<T>(gamma[32 * nb + r] * (float)w2[(int64_t)(32 * nb + r) * 4 * C + 32 * kc + 8 * kq + e]);
}

// ---------------------------------------------------------------------------------------------- pw1: x-stationary
template <int C, int NW, int D> struct XsPw1Smem {
  static constexpr int kSlot = 32 * C * 2;                 // one hidden chunk of W1 (24576 at C = 384)
  static constexpr int kBias = D * kSlot;                  // b1 of this workgroup's hidden range (<= 4C floats)
  static constexpr int bytes = kBias + 4 * C * 4;
};

#define GCV_XS_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory")


// grid = (ceil(M / (32 NW)), nsplit): workgroup (bx, by) owns tokens [32 NW bx, +32 NW) and hidden chunks
// [by * nch_split, +nch_split) of the 4C/32
template <typename T, int C, int NW, int D>
__global__ void __launch_bounds__(NW * 64, NW / 4) xs_pw1_kernel(const MlpPairArgs a, const int nch_split) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  typedef XsPw1Smem<C, NW, D> SM;
  constexpr int KP = C / 16;                               // k-steps of 16
  constexpr int SLOT = SM::kSlot;
  constexpr int NPC = SLOT / 1024;                         // DMA pieces per chunk
  static_assert(NPC % NW == 0, "pieces per wave");
  constexpr int PPW = NPC / NW;
  constexpr int WAITN = (D - 2) * PPW;                     // see the count at step()
  constexpr int NKC = 4 * C / 32;                          // hidden chunks in all

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int ntb = (a.M + 31) >> 5;
  const int tb = (int)blockIdx.x * NW + wave;              // this wave's token block
  const int64_t m = (int64_t)tb * 32 + lr;
  const int64_t mc = m < a.M ? m : (int64_t)a.M - 1;       // clamp: rows past M repeat the last row, their block is
                                                           // either padding of the hidden tensor or dropped (tb >= ntb)
  const int g0 = (int)blockIdx.y * nch_split;
  const int nch = (NKC - g0) < nch_split ? (NKC - g0) : nch_split;

  // ---- x_ln fragments: k-step p, lane (token lr, half lh) holds k = 16p + 8lh .. +7.  Issued by hand so that they stay
  // in flight under the ring prologue (hipcc would wait vmcnt(0) for an ordinary load before the first DMA)
  u32x4 xf[KP];
  {
    const T* xp = (const T*)a.X + mc * C + 8 * lh;
#pragma unroll
    for (int p = 0; p < KP; ++p)
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(xf[p]) : "v"(xp), "n"(p * 32) : "memory");
  }
  // ---- b1 of the hidden range -> LDS by DMA (1 KB = 8 chunks per piece; sources clamped into the array)
  {
    const int npieces = (nch * 128 + 1023) >> 10;
    for (int pc = wave; pc < npieces; pc += NW) {
      int fo = g0 * 32 + pc * 256 + lane * 4;              // first of this lane's four floats
      fo = fo < 4 * C - 4 ? fo : 4 * C - 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.b1 + fo),
                                       (__attribute__((address_space(3))) void*)(smem + SM::kBias + pc * 1024), 16, 0, 0);
    }
  }
  // ---- W1 ring: chunk i of this workgroup's range -> slot i % D; the walk wraps around so that every step issues
  // exactly PPW pieces per wave (the last D-1 fetches are never read)
  const unsigned char* const wsrc = (const unsigned char*)a.W1f + (int64_t)g0 * SLOT + wave * 1024 + lane * 16;
  int gw = 0, slot_w = 0;
  auto issue_piece = [&](int i) {                          // piece wave + NW * i of chunk gw -> slot slot_w
    const unsigned char* src = wsrc + (int64_t)gw * SLOT;
    unsigned char* dst = smem + slot_w * SLOT + wave * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * NW * 1024),
                                     (__attribute__((address_space(3))) void*)(dst + i * NW * 1024), 16, 0, 0);
  };
  auto issue_advance = [&]() {
    gw = gw + 1 == nch ? 0 : gw + 1;
    slot_w = slot_w + 1 == D ? 0 : slot_w + 1;
  };
  auto issue = [&]() {
#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_piece(i);
    issue_advance();
  };
  // (Spreading a step's pieces over its sub-blocks, the two waves of a SIMD taking turns, was measured and is worse:
  //  86 us against 80 us at 256 images.  An LDS-DMA instruction costs its SIMD the same issue time wherever it sits.)
  XS_STAMP(0);
#pragma unroll
  for (int s = 0; s < (GCV_XS_PAIR ? D - 2 : D - 1); ++s) issue();

  // hidden tensor as a buffer: blocks of token blocks >= ntb are out of range and their stores are dropped
  const __amdgpu_buffer_rsrc_t rsh =
      __builtin_amdgcn_make_buffer_rsrc(a.hidden, 0, (int)((int64_t)ntb * NKC * 2048), 0x00020000);
  // (the product ntb * NKC * 2048 < 2^31 is checked by the launcher)
  const unsigned hbase = ((unsigned)tb * NKC + (unsigned)g0) * 2048u + (unsigned)lane * 16u;

  const float* const sb = reinterpret_cast<const float*>(smem + SM::kBias) + 4 * lh;
  int slot_r = 0;

  // ---- one hidden chunk = 24 MFMAs on one accumulation chain (accumulator preloaded with b1), issued in six fenced
  // sub-blocks of four.  Each sub-block also carries one third of the GELU of half of the PREVIOUS chunk (8 of a lane's
  // 16 values = four packed pairs that advance in lockstep) and the four fragment reads of the next sub-block, and
  // sched_group_barrier spaces them 1 MFMA : 1 LDS read : ~7 vector instructions.  (A single pipeline over the whole
  // chunk, with the LDS reads inside it, is not honoured by hipcc 7.2: it emits the 24 MFMAs as read -> wait -> MFMA
  // pairs and the GELU behind them, one dependent v_pk_fma chain at a time.)
  GeluH16::State<4> gst;                                   // GELU state of the half chunk (8 values) in flight
  float gx[8];
  auto gelu_a = [&](const f32x16& acc, int half) {         // |x| clamp -> fp16, t, Horner levels 8 .. 6
#pragma unroll
    for (int c = 0; c < 8; ++c) gx[c] = acc[8 * half + c];
    GeluH16::begin<4, 6>(gx, gst);
  };
  auto gelu_b = [&]() { GeluH16::horner<4, 5, 2>(gst); };  // Horner levels 5 .. 2
  // max(x, 0) - h -> 16-bit -> one 1 KB half of the block.  Lane (token, lh) holds channels 8q + 4lh + e; one
  // v_permlane32_swap per dword of a (q, q+1) pair gives lanes 0-31 the 16 bytes of piece q and lanes 32-63 those of
  // piece q+1, i.e. the wave stores [kq][token][8] in lane order.
  auto gelu_c = [&](const f32x16& acc, int half, int kc) {
    uint32_t hw[4];
    GeluH16::horner<4, 1, 0>(gst);                         // Horner levels 1, 0
    GeluH16::finish_frag<T, 4>(gx, gst, hw);
    const uint2 pk[2] = {uint2{hw[0], hw[1]}, uint2{hw[2], hw[3]}};
    u32x4 w;
    if (!(GCV_XS_ABLATE & 32)) {
      const auto sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
      const auto sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
      w = (u32x4){sx[0], sy[0], sx[1], sy[1]};
    } else {
      w = (u32x4){pk[0].x, pk[0].y, pk[1].x, pk[1].y};
    }
    if (GCV_XS_ABLATE & 64) asm volatile("" ::"v"(w));      // keeps the whole GELU alive without the store
    else if (!(GCV_XS_ABLATE & 4) || w[0] == 0x12345u)
      __builtin_amdgcn_raw_buffer_store_b128(w, rsh, hbase + (unsigned)kc * 2048u + (unsigned)half * 1024u, 0, 0);
  };

  u32x4 wf[4];                                             // W1 fragments of the sub-block about to issue
  const unsigned char* sw = smem + lane * 16;              // + slot * SLOT + p * 1024
  auto read_frags = [&](int p0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = *(const u32x4*)(sw + (p0 + i) * 1024);
  };
  auto read_bias = [&](f32x16& acc, int kc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *(const f32x4*)(sb + kc * 32 + 8 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[4 * q + e] = bv[e];
    }
  };
  // sub-block j of chunk kc: MFMAs 4j .. 4j+3 into nxt, GELU part (j % 3) of half (j / 3) of chunk kc-1 (in cur)
  auto sub_block = [&](f32x16& cur, f32x16& nxt, int kc, auto jc, auto gc) {
    constexpr int j = decltype(jc)::value;
    constexpr bool gelu_prev = decltype(gc)::value;
    u32x4 w0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w0[i] = wf[i];
    if (j < 5 && !(GCV_XS_ABLATE & 16)) read_frags(4 * j + 4);
    if (!(GCV_XS_ABLATE & 8)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) Mfma<T>::run(w0[i], xf[4 * j + i], nxt);
    }
    if (gelu_prev) {
      if (j % 3 == 0 && !(GCV_XS_ABLATE & 1)) gelu_a(cur, j / 3);
      if (j % 3 == 1 && !(GCV_XS_ABLATE & 1)) gelu_b();
      if (j % 3 == 2) gelu_c(cur, j / 3, kc - 1);
#if GCV_XS_SGB == 1
      if (j == 0) __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);    // vector work first: the first fragments are
#pragma unroll                                                          // still on their way from LDS
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, j == 0 ? 6 : 9, 0);
      }
#elif GCV_XS_SGB == 2                                                   // diagnostic: the four MFMAs, then the vector work
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 40, 0);
#elif GCV_XS_SGB == 3                                                   // diagnostic: the vector work, then the four MFMAs
      __builtin_amdgcn_sched_group_barrier(0x002, 40, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    if (gelu_prev && (kc == 10 || kc == 11)) XS_STAMP(32 + 8 * (kc - 10) + j);
  };
#define GCV_XS_SUBBLOCKS(cur, nxt, kc, G)                                                      \
  sub_block(cur, nxt, kc, std::integral_constant<int, 0>{}, std::integral_constant<bool, G>{}); \
  sub_block(cur, nxt, kc, std::integral_constant<int, 1>{}, std::integral_constant<bool, G>{}); \
  sub_block(cur, nxt, kc, std::integral_constant<int, 2>{}, std::integral_constant<bool, G>{}); \
  sub_block(cur, nxt, kc, std::integral_constant<int, 3>{}, std::integral_constant<bool, G>{}); \
  sub_block(cur, nxt, kc, std::integral_constant<int, 4>{}, std::integral_constant<bool, G>{}); \
  sub_block(cur, nxt, kc, std::integral_constant<int, 5>{}, std::integral_constant<bool, G>{})
  static_assert(KP == 24, "the sub-block schedule is written for K = 384");
  auto gelu_tail = [&](const f32x16& acc, int kcl) {       // the last chunk's GELU has no MFMAs to hide under
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      gelu_a(acc, half);
      gelu_b();
      gelu_c(acc, half, kcl);
    }
  };
  // the MFMAs of chunk kc into `nxt` with the GELU of chunk kc-1 (in `cur`) in their shadow; the chunk's slot is current
  auto chunk = [&](f32x16& cur, f32x16& nxt, int kc, auto gc) {
    sw = smem + slot_r * SLOT + lane * 16;
    slot_r = slot_r + 1 == D ? 0 : slot_r + 1;
    read_bias(nxt, kc);
    read_frags(0);
    __builtin_amdgcn_sched_barrier(0);
    GCV_XS_SUBBLOCKS(cur, nxt, kc, decltype(gc)::value);
  };
  f32x16 accA, accB;
#if GCV_XS_PAIR
  // TWO chunks per barrier (nch is even: launcher): at the head of pair (kc, kc+1) the ring holds chunks kc .. kc+3, the
  // slots of kc-2 and kc-1 are free; both needed chunks have landed when at most the 2 * PPW pieces of kc+2, kc+3 are
  // outstanding (the stores of the previous pair are younger still: the count is conservative by them)
  constexpr int WAITP = 2 * PPW;
  GCV_XS_WAIT(WAITP);
#pragma unroll
  for (int p = 0; p < KP; ++p) asm volatile("" : "+v"(xf[p]));   // no use of xf may move above the wait
  issue();
  issue();
  chunk(accB, accA, 0, std::false_type{});
  chunk(accA, accB, 1, std::true_type{});
#pragma unroll 1
  for (int kc = 2; kc < nch; kc += 2) {
    if (kc >= 8 && kc < 16) XS_STAMP(2 * kc);
    GCV_XS_WAIT(WAITP);
    if (kc >= 8 && kc < 16) XS_STAMP(2 * kc + 1);
    if (!(GCV_XS_ABLATE & 2)) { issue(); issue(); }
    chunk(accB, accA, kc, std::true_type{});
    chunk(accA, accB, kc + 1, std::true_type{});
  }
  XS_STAMP(1);
  gelu_tail(accB, nch - 1);
#else
  // One chunk per barrier.  Wait count: the DMAs of chunk kc were issued D-1 steps ago; younger than them are the
  // (D-2) * PPW DMAs of chunks kc+1 .. kc+D-2 and the 2 stores of each step since; vmcnt((D-2) * PPW) ignores the stores
  auto step = [&](f32x16& cur, f32x16& nxt, int kc) {
    if (kc >= 8 && kc < 16) XS_STAMP(2 * kc);
    GCV_XS_WAIT(WAITN);
    if (kc >= 8 && kc < 16) XS_STAMP(2 * kc + 1);
    if (!(GCV_XS_ABLATE & 2)) issue();
    chunk(cur, nxt, kc, std::true_type{});
  };
  GCV_XS_WAIT(WAITN);
#pragma unroll
  for (int p = 0; p < KP; ++p) asm volatile("" : "+v"(xf[p]));   // no use of xf may move above the wait
  issue();
  chunk(accA, accA, 0, std::false_type{});
  int kc = 1;
#pragma unroll 1
  for (; kc + 1 < nch; kc += 2) {
    step(accA, accB, kc);
    step(accB, accA, kc + 1);
  }
  XS_STAMP(1);
  if (kc < nch) {                                          // nch even: one more chunk, it ends up in accB
    step(accA, accB, kc);
    gelu_tail(accB, kc);
  } else {
    gelu_tail(accA, kc - 1);
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no DMA may outlive the workgroup's LDS
  XS_STAMP(2);
}

// ---------------------------------------------------------------------------------------------- pw2: fragment-major operands
template <int C, int BN, int D, int TB = 8> struct Pw2fSmem {
  static constexpr int kStage = (TB + BN / 32) * 2048;     // TB token blocks + BN/32 channel blocks, one 32-deep K chunk
  static constexpr int kBG = D * kStage;                   // gamma * b2 of the tile's BN channels
  // residual rows on their way into the accumulators (TRICKLE, below): D - 1 slots of 1 KB per wave
  static constexpr int kRes = kBG + BN * 4;
  static constexpr int bytes = kRes + 8 * (D - 1) * 1024;
  static_assert(bytes <= 160 * 1024, "LDS of a CU");
};

// f(integral_constant<int, I>) for I = B .. E-1, in order (a loop whose index is a constant expression in its body)
template <int B, int E, typename F> __device__ __forceinline__ void pw2f_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    pw2f_static_for<B + 1, E>(f);
  }
}

// grid = ceil(M / (32 TB)) * (C / BN); 8 waves: BN = 384 -> 2 (M) x 4 (N), wave tile 128 tokens x 96 channels;
//                                               BN = 192 -> 4 (M) x 2 (N), wave tile  64 tokens x 96 channels (TB = 8)
// or 32 x 96 (TB = 4: 128-token tiles for launches of a few thousand tokens, where 256-token tiles leave most CUs idle and
// every workgroup walks the whole K = 4C on its own: the vae B = 32 configuration ran 50 workgroups of 37 us each)
template <typename T, int C, int BN, int D, int TB = 8>
__global__ void __launch_bounds__(512, 2) pw2f_kernel(const MlpPairArgs a) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  static_assert(BN == 384 || BN == 192, "tile widths");
  static_assert(TB == 8 || (TB == 4 && BN == 192), "token blocks per tile");
  typedef Pw2fSmem<C, BN, D, TB> SM;
  constexpr int NKC = 4 * C / 32;                          // K chunks of 32
  constexpr int NB = BN / 32;                              // channel blocks per tile
  constexpr int CB = C / 32;                               // channel blocks in all
  constexpr int WN = BN / 96, WM = 8 / WN;                 // waves along N / M
  constexpr int MI = TB / WM, NI = 3;                      // token blocks / channel blocks per wave
  constexpr int HP = 2 * TB;                               // 1 KB pieces of the hidden operand per stage (8 per DMA round)
  static_assert(HP % 8 == 0 && MI >= 1, "hidden pieces fill whole rounds of the eight waves");
  constexpr int STAGE = SM::kStage;
  constexpr int NPC = STAGE / 1024;                        // 40 / 28 pieces per stage
  constexpr int PPW = (NPC + 7) / 8;                       // 5 / 4 (BN = 192: waves 4-7 repeat the last piece)
  constexpr int WAITN = (D - 2) * PPW;
  typedef T t4 __attribute__((ext_vector_type(4)));

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int ntb = (a.M + 31) >> 5;
  constexpr int NTN = C / BN;
  const int ntiles = ((a.M + 32 * TB - 1) / (32 * TB)) * NTN;
  const int bid = xcd_remap(blockIdx.x, ntiles);
  const int tile_m = bid / NTN, tile_n = bid - tile_m * NTN;
  const int tb0 = tile_m * TB;
  const int nb0 = tile_n * NB;

  // ---- per-lane DMA sources of this wave's pieces q = wave + 8 i: i < HP/8 are halves of hidden blocks (tb0 + q/2, kc),
  // the others are 1 KB pieces of the NB contiguous W2 blocks (kc, nb0 ..).  (Which kind a piece is depends on i alone, so
  // that the per-piece state lives in named registers, not in an indexed array.)
  const unsigned char *src0, *src1, *src2, *src3, *src4 = nullptr;
  auto init_src = [&](const int i) -> const unsigned char* {
    if (i < HP / 8) {
      int tbq = tb0 + 4 * i + (wave >> 1);
      tbq = tbq < ntb ? tbq : ntb - 1;
      return (const unsigned char*)a.hidden + (int64_t)tbq * NKC * 2048 + (wave & 1) * 1024;
    }
    int q = wave + 8 * i;
    q = q < NPC ? q : NPC - 1;
    return (const unsigned char*)a.W2f + (int64_t)nb0 * 2048 + (q - HP) * 1024;
  };
  // (wave-uniform bases in SGPRs; the lane's 16-byte offset is the 32-bit VGPR operand of the saddr form)
  const unsigned lane16 = (unsigned)lane * 16u;
  src0 = init_src(0); src1 = init_src(1); src2 = init_src(2); src3 = init_src(3);
  if (PPW > 4) src4 = init_src(4);
  int slot_w = 0;
  auto issue_piece = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < PPW) {
      const unsigned char*& sp = i == 0 ? src0 : (i == 1 ? src1 : (i == 2 ? src2 : (i == 3 ? src3 : src4)));
      unsigned char* dst = smem + slot_w * STAGE;
      int q = wave + 8 * i;
      q = q < NPC ? q : NPC - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sp + lane16),
                                       (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
      sp += i < HP / 8 ? 2048 : CB * 2048;                // next K chunk: the block after / the next row of W2 blocks
    }
  };
#define GCV_P2_PIECE(I) issue_piece(std::integral_constant<int, I>{})
  auto issue_advance = [&]() { slot_w = slot_w + 1 == D ? 0 : slot_w + 1; };
  auto issue = [&]() {
    GCV_P2_PIECE(0); GCV_P2_PIECE(1); GCV_P2_PIECE(2); GCV_P2_PIECE(3); GCV_P2_PIECE(4);
    issue_advance();
  };
  // gamma * b2 -> LDS (ordinary loads, before any DMA is in flight): W2f carries the layer scale, so the accumulator of
  // out = resid + gamma * (W2 . h + b2) starts at gamma * b2 and the residual is simply added to it
  {
    float* sbg = reinterpret_cast<float*>(smem + SM::kBG);
    for (int i = tid; i < BN; i += 512) sbg[i] = a.b2[nb0 * 32 + i] * a.gamma[nb0 * 32 + i];
  }
  __syncthreads();
  P2_STAMP(40);
#pragma unroll
  for (int s = 0; s < D - 1; ++s) issue();                 // K chunks 0 .. D-2 (NKC >= D - 1 by construction)

  f32x16 acc[MI][NI];
  {
    const float* sbg = reinterpret_cast<const float*>(smem + SM::kBG) + wn * 96 + 4 * lh;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *(const f32x4*)(sbg + 32 * j + 8 * q);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = bv[e];
      }
  }

  // ---- TRICKLE: the residual rows do not wait for the epilogue.  Round 3 read them there, four dependent round trips to
  // HBM per wave with nothing beside them: 20 of the 256 x 384 kernel's 83 us, all 196 workgroups at once.
  // Here piece r = (token block i, channel block j, half pr) of the wave's residual tile — one 16-byte piece per lane, the
  // layout the accumulator exchange below wants — is fetched by LDS-DMA into one of D - 1 private 1 KB slots at K step r
  // (registers: none; it is issued BEFORE the step's ring pieces, so the counted wait of step r + D - 1, which retires the
  // ring pieces issued at step r, retires it too) and added into acc[i][j] at step r + D - 1.  2 MI NI pieces (24 at the
  // 128 x 96 wave tile), NKC = 48 steps.
  constexpr int NPR = MI * NI * 2;
  constexpr int RD = D - 1;                                // steps between a piece's issue and its use = its slots
  static_assert(NPR + RD <= NKC - (D - 1), "the trickle ends inside the steady-state part of the K loop");
  const T* const Rp = (const T*)a.resid;
  unsigned char* const sres = smem + SM::kRes + wave * (RD * 1024);
  const int ncol0 = nb0 * 32 + wn * 96;                    // first channel of this wave
  auto row_of = [&](int i) { return (int64_t)(tb0 + wm * MI + i) * 32 + lr; };
  auto issue_res = [&](int r) {
    const int i = r / (2 * NI), j = (r >> 1) % NI, pr = r & 1;
    const int64_t mm = row_of(i);
    const int64_t mmc = mm < a.M ? mm : (int64_t)a.M - 1;
    const T* src = Rp + mmc * C + ncol0 + 32 * j + 16 * pr + 8 * lh;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(sres + (r % RD) * 1024), 16, 0, 0);
  };
  auto take_res = [&](auto rc) {                           // acc[i][j][8 pr .. 8 pr + 7] += the landed piece
    constexpr int r = decltype(rc)::value;
    constexpr int i = r / (2 * NI), j = (r >> 1) % NI, pr = r & 1;
    const u32x4 rw = *(const u32x4*)(sres + (r % RD) * 1024 + lane * 16);
    const auto rx = __builtin_amdgcn_permlane32_swap(rw[0], rw[2], false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(rw[1], rw[3], false, false);
    const t4 rq[2] = {__builtin_bit_cast(t4, uint2{rx[0], ry[0]}), __builtin_bit_cast(t4, uint2{rx[1], ry[1]})};
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][4 * (2 * pr + d) + e] += to_f(rq[d][e]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slot is refilled right behind this
  };

  int slot_r = 0;
  // One K chunk = 2 k-steps x MI x NI MFMAs, issued in four fenced groups (k-step s = g / 2, token blocks of half g % 2).
  // With REFILL the wave's DMA pieces of the ring refill go between the groups, one or two at a time, and the two waves of
  // a SIMD use different slots (an LDS-DMA instruction holds its wave for 60-180 cycles: all eight waves issuing their five
  // right behind the barrier cost 450-750 cycles per chunk with the matrix pipe idle; profiles/p2_stamps.py).
  auto compute = [&](auto refill) {
    constexpr bool REFILL = decltype(refill)::value;
    const unsigned char* st = smem + ((GCV_P2_ABLATE & 4) ? 0 : slot_r * STAGE) + lane * 16;
    // registers: the token-block fragments of one k-step at a time (the second k-step's are read once the first's last
    // MFMA has issued), the three channel-block fragments of both
    u32x4 hf[MI], wf[2][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) hf[i] = *(const u32x4*)(st + (wm * MI + i) * 2048);
#pragma unroll
    for (int j = 0; j < NI; ++j) wf[0][j] = *(const u32x4*)(st + TB * 2048 + (wn * NI + j) * 2048);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int sk = g >> 1;
      if (g == 0) {
#pragma unroll
        for (int j = 0; j < NI; ++j) wf[1][j] = *(const u32x4*)(st + TB * 2048 + (wn * NI + j) * 2048 + 1024);
      }
      if (g == 2) {
#pragma unroll
        for (int i = 0; i < MI; ++i) hf[i] = *(const u32x4*)(st + (wm * MI + i) * 2048 + 1024);
      }
#pragma unroll
      for (int i = (MI * (g & 1) + 1) / 2; i < (MI * ((g & 1) + 1) + 1) / 2; ++i)   // MI = 1: all in the first group
#pragma unroll
        for (int j = 0; j < NI; ++j)
          if (!(GCV_P2_ABLATE & 1)) Mfma<T>::run(wf[sk][j], hf[i], acc[i][j]);
          else asm volatile("" ::"v"(wf[sk][j]), "v"(hf[i]));
      __builtin_amdgcn_sched_barrier(0);
      if (REFILL && !(GCV_P2_ABLATE & 2)) {
        // pieces per gap: {1, 1, 2, 1} (with four pieces the last gap is empty)
        // (the same pieces in the same gaps for every wave: a wave-uniform branch that touched different source
        //  pointers on its two sides made hipcc keep all five in scratch; the two waves of a SIMD are skewed by the
        //  matrix pipe anyway, one runs its group while the other waits for it)
        if (g == 0) GCV_P2_PIECE(0);
        if (g == 1) GCV_P2_PIECE(1);
        if (g == 2) { GCV_P2_PIECE(2); GCV_P2_PIECE(3); }
        if (g == 3) GCV_P2_PIECE(4);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (REFILL) issue_advance();
    slot_r = slot_r + 1 == D ? 0 : slot_r + 1;
  };

  // steady state: chunk kc has landed when at most (D-2) younger stages are outstanding; its slot's previous tenant
  // (chunk kc-1's neighbour in the ring) was left by every wave before this barrier, so the refill goes there
  constexpr int NSTEADY = NKC - (D - 1);
  // steps 0 .. NPR + RD - 1, straight-line (which accumulator block a step feeds must be a constant: a run-time switch over
  // the 24 blocks made hipcc spill 761 registers): a residual piece rides in front of the ring pieces of steps 0 .. NPR - 1,
  // so the D - 2 younger stages a step leaves in flight hold PPW operations each plus one for every such step among them
  pw2f_static_for<0, NPR + RD>([&](auto kcc) {
    constexpr int kc = decltype(kcc)::value;
    constexpr int lo = kc - (D - 2) > 0 ? kc - (D - 2) : 0, hi = kc < NPR ? kc : NPR;   // younger steps [lo, hi) issued a piece
    constexpr int WAITR = (D - 2) * PPW + (hi > lo ? hi - lo : 0);
    GCV_XS_WAIT(WAITR);
    if constexpr (kc >= RD) take_res(std::integral_constant<int, kc - RD>{});
    if constexpr (kc < NPR) issue_res(kc);
    __builtin_amdgcn_sched_barrier(0);
    compute(std::true_type{});
  });
#pragma unroll 1
  for (int kc = NPR + RD; kc < NSTEADY; ++kc) {
    if (kc >= 32 && kc < 40) P2_STAMP(4 * (kc - 32));     // (diagnostic builds: steps 32 .. 39 are behind the trickle)
    GCV_XS_WAIT(WAITN);
    if (kc >= 32 && kc < 40) { P2_STAMP(4 * (kc - 32) + 1); P2_STAMP(4 * (kc - 32) + 2); }
    compute(std::true_type{});
    if (kc >= 32 && kc < 40) P2_STAMP(4 * (kc - 32) + 3);
  }
#pragma unroll
  for (int kc = NSTEADY; kc < NKC; ++kc) {                 // drain: nothing left to issue
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    compute(std::false_type{});
  }

  P2_STAMP(41);
  // ---- epilogue, token-major rows: the accumulator already is the result (it started at gamma * b2, W2f carries gamma, the
  // residual came in under the K loop).  A row's 16-byte piece (channels 16 pr + 8 lh .. + 7 of block j) is stored whole;
  // v_permlane32_swap converts the accumulator's (8q + 4lh) halves into it.
  T* Op = (T*)a.out;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int64_t mm = row_of(i);
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        uint2 pk[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          t4 o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) o4[e] = from_f<T>(acc[i][j][4 * (2 * pr + d) + e]);
          pk[d] = __builtin_bit_cast(uint2, o4);
        }
        const auto sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
        const u32x4 w = {sx[0], sy[0], sx[1], sy[1]};
        if (mm < a.M) *(u32x4*)(Op + mm * C + ncol0 + 32 * j + 16 * pr + 8 * lh) = w;
      }
  }
  P2_STAMP(42);
}

// the two halves (separately, so that a caller can time them) and both in stream order
template <typename T> int launch_xs_pw1(const MlpPairArgs& a, int C, hipStream_t s);
template <typename T> int launch_pw2f(const MlpPairArgs& a, int C, hipStream_t s);
template <typename T> int launch_mlp_pair(const MlpPairArgs& a, int C, hipStream_t s);
// (4C, C) / (C, 4C) row-major weights of type S (T or float) on the device -> fragment-major T
template <typename T, typename S> int launch_pack_w1_frag(const S* w1, T* out, int C, hipStream_t s);
template <typename T, typename S> int launch_pack_w2_frag(const S* w2, const float* gamma, T* out, int C, hipStream_t s);
static inline bool mlp_pair_supported(int C) { return C == 384; }
// bytes of the hidden workspace launch_mlp_pair needs for M tokens
static inline size_t mlp_pair_hidden_bytes(int64_t M, int C) { return (size_t)((M + 31) / 32) * (size_t)(4 * C / 32) * 2048; }

}  // namespace gcv
