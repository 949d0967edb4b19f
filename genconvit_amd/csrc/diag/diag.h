// Diagnostic switches of the kernels, in one place.  A product build (the Makefile's) defines none of them: every
// GCV_*_ABLATE mask is 0, every *_STAMP(...) expands to nothing and no stamp buffer or reader exists in the library.
// Only the variant builds of profiles/build_variant.sh (-DGCV_XS_STAMPS=1, -DGCV_P2_ABLATE=2, -DGCV_EXPERIMENTS, ...) turn
// anything on; their outputs are read by profiles/*_stamps.py.  Also here: diag/fused_mlp_ring.h, the LDS-DMA ring MLP that
// was measured slower twice and is compiled in GCV_EXPERIMENTS builds only.
#pragma once

// ---- ablation masks (bit set = that part of the kernel is compiled out; results are then wrong on purpose)
#ifndef GCV_XS_ABLATE
#define GCV_XS_ABLATE 0    // xs_pw1_kernel: 1 no GELU arithmetic, 2 no DMA after the prologue, 4 no stores, 8 no MFMA,
#endif                     //                16 no fragment reads after the first, 32 no permlane exchange, 64 GELU kept alive without the store
#ifndef GCV_P2_ABLATE
#define GCV_P2_ABLATE 0    // pw2f_kernel: 1 no MFMA, 2 no DMA in the K loop, 4 fragment reads of chunk 0 only, 8 no epilogue loads / stores
#endif
#ifndef GCV_XM_ABLATE
#define GCV_XM_ABLATE 0    // xs_mlp_kernel: 1 no GELU arithmetic, 2 no DMA after the prologue, 8 no MFMA
#endif
#ifndef GCV_MLP_ABLATE
#define GCV_MLP_ABLATE 0   // fused_mlp_kernel / fused_mlp_res_kernel: 1 no GELU, 2 no weight streaming after chunk 0, 4 no GEMM2, 8 x rows of tile 0 only
#endif
#ifndef GCV_GLDS_ABLATE
#define GCV_GLDS_ABLATE 0  // gemm_glds_kernel: 1 no MFMA / fragment reads, 2 no steady-state loads, 4 no epilogue math
#endif
#ifndef GCV_DWR_ABLATE
#define GCV_DWR_ABLATE 0   // dwconv7_ln_roll_kernel: 1 no tap FMAs, 2 no LN reduction, 4 no normalise / stores, 8 no input loads, 16 no barrier, 32 no priority rotation
#endif
#ifndef GCV_DWM_ABLATE
#define GCV_DWM_ABLATE 0   // dwconv7_ln_mfma_kernel: 1 no MFMAs, 2 no LayerNorm (reads, math, stores), 4 no operand gathers
#endif
// ---- schedule variants of xs_pw1_kernel that were measured against the product's (mlp_pair.h)
#ifndef GCV_XS_PAIR
#define GCV_XS_PAIR 1      // two hidden chunks per barrier (0: one, the first version of the kernel)
#endif
#ifndef GCV_XS_SGB
#define GCV_XS_SGB 1       // 1 = 1 MFMA : 1 LDS read : n vector instructions (product); 0 / 2 / 3 diagnostics, see sub_block
#endif

// ---- s_memtime stamps: <FAMILY>_STAMP(...) writes the cycle counter of chosen waves of the first workgroups into a
// __device__ buffer nothing else reads; gcv_debug_read_*_stamps copies it out (exported by the family's f16 TU).
#ifndef GCV_XS_STAMPS
#define GCV_XS_STAMPS 0
#endif
#ifndef GCV_XS_STAMP_WAVE
#define GCV_XS_STAMP_WAVE 0
#endif
#ifndef GCV_P2_STAMPS
#define GCV_P2_STAMPS 0
#endif
#ifndef GCV_P2_STAMP_WAVE
#define GCV_P2_STAMP_WAVE 0
#endif
#ifndef GCV_XM_STAMPS
#define GCV_XM_STAMPS 0
#endif
#ifndef GCV_MLP_STAMPS
#define GCV_MLP_STAMPS 0
#endif
#ifndef GCV_GLDS_STAMPS
#define GCV_GLDS_STAMPS 0
#endif
#ifndef GCV_DW_STAMPS
#define GCV_DW_STAMPS 0
#endif

#define GCV_DIAG_STAMP(buf, cond, idx)                                                \
  do {                                                                                \
    if (cond) {                                                                       \
      unsigned long long _t;                                                          \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");       \
      buf[idx] = _t;                                                                  \
    }                                                                                 \
  } while (0)
#define GCV_DIAG_READER(fn, buf)                                                                          \
  extern "C" __attribute__((visibility("default"))) int fn(unsigned long long* host, int n) {             \
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gcv::buf), sizeof(unsigned long long) * n);          \
  }
#define GCV_DIAG_NOP do { } while (0)

#if GCV_XS_STAMPS
namespace gcv { __device__ unsigned long long gcv_xs_stamps[64 * 64]; }
#define XS_STAMP(i) GCV_DIAG_STAMP(gcv_xs_stamps, blockIdx.x < 64 && blockIdx.y == 0 && threadIdx.x == 64 * GCV_XS_STAMP_WAVE, blockIdx.x * 64 + (i))
#define GCV_XS_STAMP_READER GCV_DIAG_READER(gcv_debug_read_xs_stamps, gcv_xs_stamps)
#else
#define XS_STAMP(i) GCV_DIAG_NOP
#define GCV_XS_STAMP_READER
#endif

#if GCV_P2_STAMPS
namespace gcv { __device__ unsigned long long gcv_p2_stamps[64 * 64]; }
#define P2_STAMP(i) GCV_DIAG_STAMP(gcv_p2_stamps, blockIdx.x < 64 && threadIdx.x == 64 * GCV_P2_STAMP_WAVE, blockIdx.x * 64 + (i))
#define GCV_P2_STAMP_READER GCV_DIAG_READER(gcv_debug_read_p2_stamps, gcv_p2_stamps)
#else
#define P2_STAMP(i) GCV_DIAG_NOP
#define GCV_P2_STAMP_READER
#endif

#if GCV_XM_STAMPS
namespace gcv { __device__ unsigned long long gcv_xm_stamps[64 * 64]; }
#define XM_STAMP(i) GCV_DIAG_STAMP(gcv_xm_stamps, blockIdx.x < 64 && threadIdx.x == 0, blockIdx.x * 64 + (i))
#define GCV_XM_STAMP_READER GCV_DIAG_READER(gcv_debug_read_xm_stamps, gcv_xm_stamps)
#else
#define XM_STAMP(i) GCV_DIAG_NOP
#define GCV_XM_STAMP_READER
#endif

#if GCV_MLP_STAMPS
namespace gcv { __device__ unsigned long long gcv_mlp_stamps[64 * 16]; }
#define GCV_STAMP(i) GCV_DIAG_STAMP(gcv_mlp_stamps, blockIdx.x < 64 && threadIdx.x == 0, blockIdx.x * 16 + (i))
// the resident kernel reports wave 0 in slots i and wave 7 in slots i + 3
#define RES_STAMP(slot)                                                                                   \
  do {                                                                                                    \
    if (blockIdx.x < 64 && (threadIdx.x & 63) == 0) {                                                     \
      unsigned long long _t;                                                                              \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                           \
      if (threadIdx.x == 0) gcv_mlp_stamps[blockIdx.x * 16 + (slot)] = _t;                                \
      if (threadIdx.x == 448) gcv_mlp_stamps[blockIdx.x * 16 + (slot) + 3] = _t;                          \
    }                                                                                                     \
  } while (0)
#define GCV_MLP_STAMP_READER GCV_DIAG_READER(gcv_debug_read_stamps, gcv_mlp_stamps)
#else
#define GCV_STAMP(i) GCV_DIAG_NOP
#define RES_STAMP(slot) GCV_DIAG_NOP
#define GCV_MLP_STAMP_READER
#endif

#if GCV_GLDS_STAMPS
namespace gcv { __device__ unsigned long long gcv_glds_stamps[4096 * 8]; }
#define GLDS_STAMP(i) GCV_DIAG_STAMP(gcv_glds_stamps, blockIdx.x < 4096 && threadIdx.x == 0, blockIdx.x * 8 + (i))
// slot 5: HW_ID | XCC_ID of the workgroup
#define GLDS_STAMP_HWID()                                                                                         \
  do {                                                                                                            \
    if (blockIdx.x < 4096 && threadIdx.x == 0)                                                                    \
      gcv_glds_stamps[blockIdx.x * 8 + 5] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) |       \
                                            ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32); \
  } while (0)
#define GCV_GLDS_STAMP_READER GCV_DIAG_READER(gcv_debug_read_glds_stamps, gcv_glds_stamps)
#else
#define GLDS_STAMP(i) GCV_DIAG_NOP
#define GLDS_STAMP_HWID() GCV_DIAG_NOP
#define GCV_GLDS_STAMP_READER
#endif

#if GCV_DW_STAMPS
// (profiles/dw_stamps.py): tap wave 0 (slots 0..7) and staging wave 0 (slots 8..15) of workgroups 0..63 in step GCV_DW_STAMP_IT
namespace gcv { __device__ unsigned long long gcv_dw_stamps[64 * 32]; }
#define GCV_DW_STAMP_IT 30
#define DW_STAMP(cond, slot) GCV_DIAG_STAMP(gcv_dw_stamps, (cond) && blockIdx.x < 64, blockIdx.x * 32 + (slot))
#define GCV_DW_STAMP_READER GCV_DIAG_READER(gcv_debug_read_dw_stamps, gcv_dw_stamps)
#else
#define DW_STAMP(cond, slot) GCV_DIAG_NOP
#define GCV_DW_STAMP_READER
#endif
