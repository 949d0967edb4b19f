// Common device/host helpers for the gfx950 (MI355X, CDNA4) GenConViT kernels.
// Wave = 64 lanes everywhere; no portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "diag/diag.h"     // diagnostic switches: all off in the product build

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

// ---- cross-lane reductions --------------------------------------------------------------------------
// Inside a 16-lane DPP row the exchange is a modifier of the v_add / v_max itself (no LDS traffic); __shfl_xor
// always compiles to ds_bpermute_b32, an LDS-pipe instruction with its own address register — the LayerNorm
// statistics loops issued almost as many of those as the depthwise taps issued real LDS reads.
#define GCV_DPP_F32(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
// xor-1 / xor-2 inside quads (quad_perm [1,0,3,2] / [2,3,0,1]), then row_half_mirror and row_mirror: every lane of the
// row ends up with the reduction over the row's 16 lanes
__device__ __forceinline__ float row16_sum(float v) {
  v += GCV_DPP_F32(v, 0xB1);
  v += GCV_DPP_F32(v, 0x4E);
  v += GCV_DPP_F32(v, 0x141);
  v += GCV_DPP_F32(v, 0x140);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, GCV_DPP_F32(v, 0xB1));
  v = fmaxf(v, GCV_DPP_F32(v, 0x4E));
  v = fmaxf(v, GCV_DPP_F32(v, 0x141));
  v = fmaxf(v, GCV_DPP_F32(v, 0x140));
  return v;
}
// aligned groups of 4 / 8 lanes (quads, half rows)
__device__ __forceinline__ float group8_sum(float v) {
  v += GCV_DPP_F32(v, 0xB1);
  v += GCV_DPP_F32(v, 0x4E);
  v += GCV_DPP_F32(v, 0x141);
  return v;
}
// aligned groups of 32 lanes: two rows, one cross-row exchange through the LDS crossbar
__device__ __forceinline__ float group32_sum(float v) {
  v = row16_sum(v);
  return v + __shfl_xor(v, 16, 64);
}
// whole wave (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
  v = group32_sum(v);
  return v + __shfl_xor(v, 32, 64);
}
// aligned groups of N lanes, N = 16 / 32 / 64
template <int N> __device__ __forceinline__ float group_sum(float v) {
  static_assert(N == 16 || N == 32 || N == 64, "group_sum: 16 / 32 / 64 lanes");
  return N == 16 ? row16_sum(v) : (N == 32 ? group32_sum(v) : wave_sum(v));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}

// ---- A/B switches ---------------------------------------------------------
// The shipped library reads two environment variables: GCV_VAE_SPLIT (schedule of gcv_vae_forward, net_impl.h) and
// GCV_RCCL_PATH (api.hip).  Every other switch between a default kernel and a slower / older alternative exists only
// in builds with -DGCV_EXPERIMENTS (profiles/build_variant.sh <name> "-DGCV_EXPERIMENTS" ...): exp_env() is the one
// place they are read, and it is a constant in the product build, so the alternatives are dead code there.
#ifdef GCV_EXPERIMENTS
#include <cstdlib>
static inline const char* exp_env(const char* name) { return std::getenv(name); }
#else
static inline const char* exp_env(const char*) { return nullptr; }
#endif

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

// hipFuncAttributeMaxDynamicSharedMemorySize is an attribute of a (function, device) pair: set it once per device,
// under a lock (several handles on several GPUs may live in one process; launchers can be called from any thread)
int ensure_dynamic_lds(const void* fn, int bytes);
#define GCV_ENSURE_LDS(fn, bytes) GCV_TRY(::gcv::ensure_dynamic_lds((const void*)(fn), (int)(bytes)))

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
