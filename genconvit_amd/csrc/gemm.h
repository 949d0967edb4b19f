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
// LDS rows are BKB bytes (64 for fp32, 128 for 16-bit) with the chunk index XOR-swizzled by
// the row so the 16-lane groups of ds_read_b128 touch all 64 banks exactly once.
#pragma once
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
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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

template <typename T, int BM, int BN, int WM, int WN, int AMODE, int EPI>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmArgs g) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int BKB = (sizeof(T) == 4) ? 64 : 128;   // bytes of K per LDS row
  constexpr int CPR = BKB / 16;                      // 16-byte chunks per row
  constexpr int BK = CPR * EPC;                      // K elements per tile (16 fp32 / 64 16-bit)
  constexpr int SW_SHIFT = (CPR == 4) ? 2 : 1;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per workgroup");
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int A_CH = BM * CPR, B_CH = BN * CPR;
  constexpr int A_SLOTS = (A_CH + 255) / 256, B_SLOTS = (B_CH + 255) / 256;
  constexpr int A_BYTES = BM * BKB, B_BYTES = BN * BKB;

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

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
  int a_coff[A_SLOTS];            // element offset of the chunk inside the K tile
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
      const int Hp = g.H >> 1, Wp = g.W >> 1;
      const int q = m & 3, p = m >> 2;
      const int xo = p % Wp, t = p / Wp;
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
#pragma unroll
    for (int t = 0; t < CPR / 2; ++t) {
      u32x4 af[MI], bf[NI];
      const int c = 2 * t + lh;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm0 + i * 32 + lr;
        af[i] = *(const u32x4*)(sA + row * BKB + ((c ^ ((row >> SW_SHIFT) & (CPR - 1))) << 4));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wn0 + j * 32 + lr;
        bf[j] = *(const u32x4*)(sB + row * BKB + ((c ^ ((row >> SW_SHIFT) & (CPR - 1))) << 4));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) Mfma<T>::run(af[i], bf[j], acc[i][j]);
    }
    if (kt + 1 < nkt) stage((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: acc[i][j][r] is C[row][col], col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  T* __restrict__ Cp = (T*)g.C;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn0 + j * 32 + lr;
      const bool n_ok = n < g.N;
      const int mb = m0 + wm0 + i * 32 + 4 * lh;
      if (EPI == EPI_BIAS_ACT) {
        const float bv = (g.bias && n_ok) ? g.bias[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if (n_ok && m < g.M) Cp[(int64_t)m * g.ldc + n] = from_f<T>(apply_act(acc[i][j][r] + bv, g.act));
        }
      } else if (EPI == EPI_RESID) {
        const float bv = (g.bias && n_ok) ? g.bias[n] : 0.0f;
        const float gv = n_ok ? g.gamma[n] : 0.0f;
        const T* __restrict__ Rp = (const T*)g.resid;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if (n_ok && m < g.M) {
            const int64_t o = (int64_t)m * g.ldc + n;
            Cp[o] = from_f<T>(to_f(Rp[o]) + gv * (acc[i][j][r] + bv));
          }
        }
      } else if (EPI == EPI_POOL4) {
        const float bv = (g.bias && n_ok) ? g.bias[n] : 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = mb + 8 * q;                      // first of 4 consecutive rows
          float v = fmaxf(fmaxf(acc[i][j][4 * q], acc[i][j][4 * q + 1]),
                          fmaxf(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]));
          v = apply_act(v + bv, g.act);                  // act is monotone (ReLU): max commutes
          if (n_ok && m < g.M) Cp[(int64_t)(m >> 2) * g.ldc + n] = from_f<T>(v);
        }
      } else if (EPI == EPI_CONVT) {
        const int cout = 1 << g.cout_log2;
        const int co = n & (cout - 1), dq = n >> g.cout_log2;
        const int dy = dq >> 1, dx = dq & 1;
        const float bv = (g.bias && n_ok) ? g.bias[co] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if (n_ok && m < g.M) {
            const int x = m % g.W, t = m / g.W;
            const int y = t % g.H, b = t / g.H;
            const int64_t o = ((((int64_t)b * 2 * g.H + 2 * y + dy) * 2 * g.W + 2 * x + dx) << g.cout_log2) + co;
            Cp[o] = from_f<T>(apply_act(acc[i][j][r] + bv, g.act));
          }
        }
      } else {  // EPI_SPLITK
        float* __restrict__ Pp = g.partial + (int64_t)blockIdx.y * g.M * g.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if (n_ok && m < g.M) Pp[(int64_t)m * g.N + n] = acc[i][j][r];
        }
      }
    }
  }
}

// Host-side launcher: validates shapes, picks a tile config, launches on `stream`.
// Returns 0 on success (error text via gcv::get_error()).
template <typename T> int launch_gemm(const GemmArgs& g, int a_mode, int epi, hipStream_t stream);
// name of the kernel family a (dtype, a_mode, epi) launch resolves to — for profiling tables
const char* gemm_family_name(int a_mode, int epi);

}  // namespace gcv
