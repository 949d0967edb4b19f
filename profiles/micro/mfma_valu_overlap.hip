// How many independent v_fma_f32 hide under one v_mfma_f32_32x32x16_f16, by where the accumulator lives (ArchVGPR / AccVGPR),
// by chain shape (one dependent accumulation chain / two alternating accumulators) and by waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
// Prints shader cycles per MFMA slot seen by wave 0 (s_memtime) and wall ns per slot.
#include <hip/hip_runtime.h>
#include <cstdio>

#define V1(c) "v_fma_f32 v" #c ", v40, v41, v" #c "\n"
#define VALU0 ""
#define VALU4 V1(24) V1(25) V1(26) V1(27)
#define VALU8 VALU4 V1(28) V1(29) V1(30) V1(31)
#define VALU12 VALU8 V1(32) V1(33) V1(34) V1(35)
#define VALU16 VALU12 V1(36) V1(37) V1(38) V1(39)
#define PK1(c) "v_pk_fma_f32 v[" #c ":" #c "+1], v[40:41], v[42:43], v[" #c ":" #c "+1]\n"
#define PK4 PK1(24) PK1(26) PK1(28) PK1(30)
#define H1(c) "v_pk_fma_f16 v" #c ", v40, v41, v" #c "\n"
#define HALF4 H1(24) H1(25) H1(26) H1(27)
#define HALF8 HALF4 H1(28) H1(29) H1(30) H1(31)
#define HALF16 HALF8 H1(32) H1(33) H1(34) H1(35) H1(36) H1(37) H1(38) H1(39)
#define M1(c) "v_pk_max_f16 v" #c ", v40, v" #c "\n"
#define MAX8 M1(24) M1(25) M1(26) M1(27) M1(28) M1(29) M1(30) M1(31)
#define C1(c) "v_cvt_pk_f16_f32 v" #c ", v40, v41\n"
#define CVT8 C1(24) C1(25) C1(26) C1(27) C1(28) C1(29) C1(30) C1(31)
#define MF_V "v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], v[0:15]\n"
#define MF_A "v_mfma_f32_32x32x16_f16 a[0:15], v[16:19], v[20:23], a[0:15]\n"
#define MF_A2 "v_mfma_f32_32x32x16_f16 a[16:31], v[16:19], v[20:23], a[16:31]\n"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23", \
             "v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43", \
             "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15", \
             "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31"

#define KERNEL(NAME, BODY)                                                                     \
  __global__ void __launch_bounds__(512) NAME(long long* cyc, float* out) {                    \
    asm volatile("v_mov_b32 v40, 1.0\nv_mov_b32 v41, 0.5\nv_mov_b32 v42, 1.0\nv_mov_b32 v43, 0.5\n" ::: CLOB); \
    long long t0 = __builtin_amdgcn_s_memtime();                                               \
    _Pragma("unroll 1") for (int it = 0; it < 500; ++it) {                                     \
      asm volatile(BODY BODY BODY BODY ::: CLOB);                                              \
    }                                                                                          \
    long long t1 = __builtin_amdgcn_s_memtime();                                               \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                   \
    if (t1 == 12345) out[threadIdx.x] = 1.0f;                                                  \
  }

KERNEL(k_v0, MF_V VALU0)
KERNEL(k_v4, MF_V VALU4)
KERNEL(k_v8, MF_V VALU8)
KERNEL(k_v12, MF_V VALU12)
KERNEL(k_v16, MF_V VALU16)
KERNEL(k_a0, MF_A VALU0)
KERNEL(k_a4, MF_A VALU4)
KERNEL(k_a8, MF_A VALU8)
KERNEL(k_a12, MF_A VALU12)
KERNEL(k_a16, MF_A VALU16)
KERNEL(k_aa8, MF_A VALU8 MF_A2 VALU8)        // two accumulators alternating: 2 MFMA slots per BODY
KERNEL(k_apk4, MF_A PK4)                     // 4 packed fp32 FMAs (= 8 issue slots) per MFMA
KERNEL(k_ah8, MF_A HALF8)
KERNEL(k_ah16, MF_A HALF16)
KERNEL(k_am8, MF_A MAX8)
KERNEL(k_ac8, MF_A CVT8)
KERNEL(k_xh8, HALF8)
KERNEL(k_xh16, HALF16)
KERNEL(k_x8, VALU8)                          // no MFMA
KERNEL(k_x16, VALU16)

template <typename K> void run(const char* name, K kern, int slots_per_body, int threads) {
  long long* cyc; float* out;
  (void)hipMalloc(&cyc, 8); (void)hipMalloc(&out, 4096);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, cyc, out);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, cyc, out);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double slots = 500.0 * 4 * slots_per_body;
  printf("%-44s waves/SIMD %d: %7.1f cycles per slot (wave 0) | wall %6.1f ns per slot per wave\n", name, threads / 256,
         h / slots, ms * 1e6 / slots);
  (void)hipFree(cyc); (void)hipFree(out);
}

int main() {
  for (int threads : {256, 512}) {
    run("VGPR acc, MFMA only", k_v0, 1, threads);
    run("VGPR acc, MFMA + 4 v_fma", k_v4, 1, threads);
    run("VGPR acc, MFMA + 8 v_fma", k_v8, 1, threads);
    run("VGPR acc, MFMA + 12 v_fma", k_v12, 1, threads);
    run("VGPR acc, MFMA + 16 v_fma", k_v16, 1, threads);
    run("AGPR acc, MFMA only", k_a0, 1, threads);
    run("AGPR acc, MFMA + 4 v_fma", k_a4, 1, threads);
    run("AGPR acc, MFMA + 8 v_fma", k_a8, 1, threads);
    run("AGPR acc, MFMA + 12 v_fma", k_a12, 1, threads);
    run("AGPR acc, MFMA + 16 v_fma", k_a16, 1, threads);
    run("AGPR, two accumulators, (MFMA + 8 v_fma) x2", k_aa8, 2, threads);
    run("AGPR acc, MFMA + 4 v_pk_fma_f32", k_apk4, 1, threads);
    run("AGPR acc, MFMA + 8 v_pk_fma_f16", k_ah8, 1, threads);
    run("AGPR acc, MFMA + 16 v_pk_fma_f16", k_ah16, 1, threads);
    run("AGPR acc, MFMA + 8 v_pk_max_f16", k_am8, 1, threads);
    run("AGPR acc, MFMA + 8 v_cvt_pk_f16_f32", k_ac8, 1, threads);
    run("no MFMA, 8 v_pk_fma_f16 per slot", k_xh8, 1, threads);
    run("no MFMA, 16 v_pk_fma_f16 per slot", k_xh16, 1, threads);
    run("no MFMA, 8 v_fma per slot", k_x8, 1, threads);
    run("no MFMA, 16 v_fma per slot", k_x16, 1, threads);
  }
  return 0;
}
