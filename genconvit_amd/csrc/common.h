// Common device/host helpers for the gfx950 (MI355X, CDNA4) GenConViT kernels.
// Wave = 64 lanes everywhere; no portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace gcv {

// ---- storage dtypes -------------------------------------------------------
// Activations and GEMM weights are stored as T in {float, half_t, bf16_t};
// all accumulation, normalisation statistics and epilogue math are fp32.
typedef _Float16 half_t;
typedef __bf16 bf16_t;

enum { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };

template <typename T> struct DT;
template <> struct DT<float>  { static constexpr int id = DT_F32;  static constexpr int EPC = 4; };
template <> struct DT<bf16_t> { static constexpr int id = DT_BF16; static constexpr int EPC = 8; };
template <> struct DT<half_t> { static constexpr int id = DT_F16;  static constexpr int EPC = 8; };
// EPC = elements per 16-byte chunk

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v) { return (T)v; }

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_LEAKY = 3 };

__device__ __forceinline__ float gelu_erf(float x) {
  // exact-erf GELU (nn.GELU default), reference model/genconvit_ed.py:75 and timm Mlp
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case ACT_RELU:  return fmaxf(x, 0.0f);
    case ACT_GELU:  return gelu_erf(x);
    case ACT_LEAKY: return x > 0.0f ? x : 0.01f * x;   // nn.LeakyReLU default slope
    default:        return x;
  }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- 16-byte vector helpers ------------------------------------------------
struct alignas(16) Chunk { uint32_t w[4]; };

template <typename T> struct Vec16;   // T[EPC] viewed as a 16-byte chunk
template <> struct Vec16<float>  { float v[4]; };
template <> struct Vec16<half_t> { half_t v[8]; };
template <> struct Vec16<bf16_t> { bf16_t v[8]; };

// wave-level reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- host side -------------------------------------------------------------
void set_error(const std::string& msg);
const char* get_error();

#define GCV_CHECK_HIP(expr)                                                         \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      ::gcv::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));          \
      return -1;                                                                    \
    }                                                                               \
  } while (0)

#define GCV_REQUIRE(cond, msg)                                                      \
  do {                                                                              \
    if (!(cond)) {                                                                  \
      ::gcv::set_error(std::string("requirement failed: ") + #cond + " — " + (msg)); \
      return -2;                                                                    \
    }                                                                               \
  } while (0)

#define GCV_TRY(x) do { int _rc = (x); if (_rc) return _rc; } while (0)
#define GCV_UP(dst, store, vec)                                                     \
  do {                                                                              \
    dst = (store).upload(vec);                                                      \
    if (!(dst)) { ::gcv::set_error("hipMalloc/upload failed for " #dst); return -5; } \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// XCD-aware block remap (8 XCDs, blocks dealt round-robin): give each XCD a
// contiguous range of logical tile ids so tiles that share an operand panel
// hit the same L2.  Bijective for any grid size (guide §5 "XCD swizzle").
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + local;
}

}  // namespace gcv
