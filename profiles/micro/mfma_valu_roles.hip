// Role-specialised waves on one SIMD: waves 0-3 of a workgroup (one per SIMD) issue only MFMAs, waves 4.. (1, 2 or 3 more per
// SIMD) issue only v_fma_f32.  Every wave reports its own cycle count; the question is whether matrix and vector work of
// DIFFERENT waves overlap where the same work interleaved inside one wave does not (mfma_valu_overlap.hip: 15 + 4 N cycles
// per MFMA + N v_fma, for one wave or two).
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/mfma_valu_roles.hip -o /tmp/mvr && /tmp/mvr
#include <hip/hip_runtime.h>
#include <cstdio>

#define V1(c) "v_fma_f32 v" #c ", v40, v41, v" #c "\n"
#define VALU16 V1(24) V1(25) V1(26) V1(27) V1(28) V1(29) V1(30) V1(31) V1(32) V1(33) V1(34) V1(35) V1(36) V1(37) V1(38) V1(39)
#define MF_A "v_mfma_f32_32x32x16_f16 a[0:15], v[16:19], v[20:23], a[0:15]\n"
#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23", \
             "v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41", \
             "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15"

// n_mfma / n_valu16: loop counts of the two roles (4 MFMAs / 64 v_fma per iteration); a count of 0 leaves the role idle
__global__ void __launch_bounds__(1024) roles(long long* cyc, float* out, int n_mfma, int n_valu16, int prio_mfma, int prio_valu) {
  const int wave = threadIdx.x >> 6;
  asm volatile("v_mov_b32 v40, 1.0\nv_mov_b32 v41, 0.5\n" ::: CLOB);
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (prio_mfma) __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
    for (int it = 0; it < n_mfma; ++it) asm volatile(MF_A MF_A MF_A MF_A ::: CLOB);
  } else {
    if (prio_valu) __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
    for (int it = 0; it < n_valu16; ++it) asm volatile(VALU16 VALU16 VALU16 VALU16 ::: CLOB);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
  if (t1 == 12345) out[threadIdx.x] = 1.0f;
}

void run(const char* name, int threads, int n_mfma, int n_valu16, int pm, int pv) {
  long long* cyc; float* out;
  (void)hipMalloc(&cyc, 8 * 16); (void)hipMalloc(&out, 8192);
  (void)hipMemset(cyc, 0, 8 * 16);
  hipLaunchKernelGGL(roles, dim3(256), dim3(threads), 0, 0, cyc, out, n_mfma, n_valu16, pm, pv);
  (void)hipDeviceSynchronize();
  long long h[16]; (void)hipMemcpy(h, cyc, 8 * 16, hipMemcpyDeviceToHost);
  const int nw = threads / 64;
  double vmax = 0;
  for (int w = 4; w < nw; ++w) vmax = h[w] > vmax ? h[w] : vmax;
  printf("%-52s MFMA wave: %6.1f cycles per MFMA", name, n_mfma ? h[0] / (4.0 * n_mfma) : 0.0);
  if (nw > 4 && n_valu16)
    printf(" | VALU waves: %5.2f cycles per v_fma per wave, SIMD retires one per %5.2f cycles", vmax / (64.0 * n_valu16),
           vmax / (64.0 * n_valu16) / ((nw - 4) / 4));
  printf("\n");
  (void)hipFree(cyc); (void)hipFree(out);
}

int main() {
  run("MFMA wave alone", 256, 2000, 0, 0, 0);
  run("1 VALU wave alone (no MFMA wave running)", 512, 0, 800, 0, 0);
  run("2 VALU waves alone", 768, 0, 800, 0, 0);
  run("3 VALU waves alone", 1024, 0, 800, 0, 0);
  // balanced so that both roles run for about the same time if they overlap: 8000 MFMAs x 32 = 256k cycles
  run("MFMA wave + 1 VALU wave  (6 v_fma per MFMA)", 512, 2000, 750, 0, 0);
  run("MFMA wave + 2 VALU waves (5 v_fma per MFMA each)", 768, 2000, 625, 0, 0);
  run("MFMA wave + 3 VALU waves (4 v_fma per MFMA each)", 1024, 2000, 500, 0, 0);
  run("MFMA wave(prio 3) + 1 VALU wave", 512, 2000, 750, 1, 0);
  run("MFMA wave(prio 3) + 2 VALU waves", 768, 2000, 625, 1, 0);
  run("MFMA wave + 2 VALU waves(prio 3)", 768, 2000, 625, 0, 1);
  return 0;
}
