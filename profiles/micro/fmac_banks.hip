// Does v_fmac_f32 (three VGPR reads: acc, a, b) issue at 2 cycles per wave on gfx950 whatever the registers, or does
// it depend on the VGPR banks (index mod 4) of its operands?  Explicit physical registers through inline asm.
//   hipcc --offload-arch=gfx950 -O3 fmac_banks.hip -o /tmp/fmac_banks && /tmp/fmac_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>

#define REP8(x) x x x x x x x x
// pattern macros: 8 independent accumulators per group, 8 groups = 64 v_fmac per loop trip
#define LOOP(body)                                                                                        \
  asm volatile("s_mov_b32 s20, 512\n"                                                                      \
               "1:\n" body "s_sub_u32 s20, s20, 1\n"                                                       \
               "s_cmp_lg_u32 s20, 0\n"                                                                     \
               "s_cbranch_scc1 1b\n" ::                                                                    \
                   : "s20", "scc", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", \
                     "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", \
                     "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", \
                     "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63")

template <int MODE> __global__ void k(long long* cyc, float* out) {
  long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {  // acc bank 0, a bank 1, b bank 2 (all different)
    LOOP(REP8("v_fmac_f32 v8, v41, v50\n v_fmac_f32 v12, v45, v54\n v_fmac_f32 v16, v49, v58\n v_fmac_f32 v20, v53, v62\n"
              "v_fmac_f32 v24, v41, v50\n v_fmac_f32 v28, v45, v54\n v_fmac_f32 v32, v49, v58\n v_fmac_f32 v36, v53, v62\n"));
  } else if (MODE == 1) {  // a and b same bank (1), acc bank 0
    LOOP(REP8("v_fmac_f32 v8, v41, v45\n v_fmac_f32 v12, v45, v49\n v_fmac_f32 v16, v49, v53\n v_fmac_f32 v20, v53, v57\n"
              "v_fmac_f32 v24, v41, v45\n v_fmac_f32 v28, v45, v49\n v_fmac_f32 v32, v49, v53\n v_fmac_f32 v36, v53, v57\n"));
  } else if (MODE == 2) {  // acc and a same bank (0), b bank 2
    LOOP(REP8("v_fmac_f32 v8, v40, v50\n v_fmac_f32 v12, v44, v54\n v_fmac_f32 v16, v48, v58\n v_fmac_f32 v20, v52, v62\n"
              "v_fmac_f32 v24, v40, v50\n v_fmac_f32 v28, v44, v54\n v_fmac_f32 v32, v48, v58\n v_fmac_f32 v36, v52, v62\n"));
  } else if (MODE == 3) {  // all three in bank 0
    LOOP(REP8("v_fmac_f32 v8, v40, v44\n v_fmac_f32 v12, v44, v48\n v_fmac_f32 v16, v48, v52\n v_fmac_f32 v20, v52, v56\n"
              "v_fmac_f32 v24, v40, v44\n v_fmac_f32 v28, v44, v48\n v_fmac_f32 v32, v48, v52\n v_fmac_f32 v36, v52, v56\n"));
  } else if (MODE == 4) {  // what hipcc emitted for the dw kernel: consecutive acc, consecutive a, one b
    LOOP(REP8("v_fmac_f32 v8, v40, v52\n v_fmac_f32 v9, v41, v52\n v_fmac_f32 v10, v42, v52\n v_fmac_f32 v11, v43, v52\n"
              "v_fmac_f32 v12, v44, v52\n v_fmac_f32 v13, v45, v52\n v_fmac_f32 v14, v46, v52\n v_fmac_f32 v8, v41, v53\n"));
  } else if (MODE == 10 || MODE == 11 || MODE == 12) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (MODE == 10) {           // static priority by position on the SIMD (waves w, w+4, w+8 share one)
      if (wave >= 8) __builtin_amdgcn_s_setprio(0); else if (wave >= 4) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
    }
    if (MODE == 12) {           // younger waves get the HIGHER priority
      if (wave >= 8) __builtin_amdgcn_s_setprio(2); else if (wave >= 4) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
    if (MODE == 11 && wave >= 12) return;   // 1024-thread launch: three busy waves and one that has left per SIMD
    LOOP(REP8("v_fmac_f32 v8, v40, v52\n v_fmac_f32 v9, v41, v52\n v_fmac_f32 v10, v42, v52\n v_fmac_f32 v11, v43, v52\n"
              "v_fmac_f32 v12, v44, v52\n v_fmac_f32 v13, v45, v52\n v_fmac_f32 v14, v46, v52\n v_fmac_f32 v8, v41, v53\n"));
  } else if (MODE == 5) {  // scalar b operand (tap in an SGPR): two VGPR reads only
    LOOP(REP8("v_fmac_f32 v8, s4, v40\n v_fmac_f32 v9, s4, v41\n v_fmac_f32 v10, s4, v42\n v_fmac_f32 v11, s4, v43\n"
              "v_fmac_f32 v12, s5, v44\n v_fmac_f32 v13, s5, v45\n v_fmac_f32 v14, s5, v46\n v_fmac_f32 v15, s5, v41\n"));
  } else if (MODE == 6) {  // v_pk_fma_f32, operands in different banks
    LOOP(REP8("v_pk_fma_f32 v[8:9], v[42:43], v[52:53], v[8:9]\n v_pk_fma_f32 v[12:13], v[46:47], v[56:57], v[12:13]\n"
              "v_pk_fma_f32 v[16:17], v[42:43], v[52:53], v[16:17]\n v_pk_fma_f32 v[20:21], v[46:47], v[56:57], v[20:21]\n"));
  } else if (MODE == 7) {  // v_dot2c_f32_f16 acc bank0, a bank1, b bank2
    LOOP(REP8("v_dot2c_f32_f16 v8, v41, v50\n v_dot2c_f32_f16 v12, v45, v54\n v_dot2c_f32_f16 v16, v49, v58\n v_dot2c_f32_f16 v20, v53, v62\n"
              "v_dot2c_f32_f16 v24, v41, v50\n v_dot2c_f32_f16 v28, v45, v54\n v_dot2c_f32_f16 v32, v49, v58\n v_dot2c_f32_f16 v36, v53, v62\n"));
  } else if (MODE == 8) {  // v_fma_mix_f32 (f16 a, f32 b, f32 acc), different banks
    LOOP(REP8("v_fma_mix_f32 v8, v41, v50, v8 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v12, v45, v54, v12 op_sel_hi:[1,0,0]\n"
              "v_fma_mix_f32 v16, v49, v58, v16 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v20, v53, v62, v20 op_sel_hi:[1,0,0]\n"
              "v_fma_mix_f32 v24, v41, v50, v24 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v28, v45, v54, v28 op_sel_hi:[1,0,0]\n"
              "v_fma_mix_f32 v32, v49, v58, v32 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v36, v53, v62, v36 op_sel_hi:[1,0,0]\n"));
  } else if (MODE == 9) {  // v_pk_fma_f16
    LOOP(REP8("v_pk_fma_f16 v8, v41, v50, v8\n v_pk_fma_f16 v12, v45, v54, v12\n v_pk_fma_f16 v16, v49, v58, v16\n v_pk_fma_f16 v20, v53, v62, v20\n"
              "v_pk_fma_f16 v24, v41, v50, v24\n v_pk_fma_f16 v28, v45, v54, v28\n v_pk_fma_f16 v32, v49, v58, v32\n v_pk_fma_f16 v36, v53, v62, v36\n"));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
  if (out && blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = (float)(t1 - t0);
}

template <int MODE> void run(const char* name) {
  long long* cyc; hipMalloc(&cyc, 8);
  float* outd; hipMalloc(&outd, 64);
  for (int threads : {256, 512, 768, 1024}) {
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, cyc, outd);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, cyc, outd);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 512.0 * (MODE == 6 ? 32 : 64);
    const double wps = threads / 256.0;
    printf("%-44s %d waves/SIMD: %6.2f ticks per instr per wave | wall %7.1f us -> %5.2f ns per instr per SIMD\n", name,
           (int)wps, h / n, ms * 1e3, ms * 1e6 / (n * wps));
    float pw[16]; hipMemcpy(pw, outd, sizeof(pw), hipMemcpyDeviceToHost);
    printf("      per-wave ticks per instr (block 0):");
    for (int w = 0; w < threads / 64; ++w) printf(" %.2f", pw[w] / n);
    printf("\n");
  }
  hipFree(cyc);
}

int main() {
  run<4>("v_fmac_f32 hipcc-like (consecutive regs)");
  run<10>("hipcc-like, priority 2/1/0 by wave age");
  run<12>("hipcc-like, priority 0/1/2 by wave age");
  run<11>("hipcc-like, waves 12-15 exit at once");
  return 0;
  run<0>("v_fmac_f32 acc/a/b in banks 0/1/2");
  run<1>("v_fmac_f32 a,b same bank");
  run<2>("v_fmac_f32 acc,a same bank");
  run<3>("v_fmac_f32 all same bank");
  run<4>("v_fmac_f32 hipcc-like (consecutive regs)");
  run<5>("v_fmac_f32 SGPR tap");
  run<6>("v_pk_fma_f32 (2 FMAs per lane)");
  run<7>("v_dot2c_f32_f16");
  run<8>("v_fma_mix_f32");
  run<9>("v_pk_fma_f16");
  return 0;
}
