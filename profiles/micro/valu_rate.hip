// Issue-rate probe: cycles per wave-instruction for v_fma_f32, v_pk_fma_f32 (independent / dependent chains),
// one wave per SIMD and two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int CHAINS> __global__ void k(float* out, long long* cyc, float seed) {
  f32x2 p[CHAINS], t[CHAINS];
  for (int c = 0; c < CHAINS; ++c) { p[c] = (f32x2){seed + c, seed - c}; t[c] = (f32x2){0.999f + 1e-4f * threadIdx.x, 1.0001f}; }
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (MODE == 0) {            // scalar fma on .x only
          p[c][0] = __builtin_fmaf(p[c][0], t[c][0], 0.5f);
        } else if (MODE == 1) {     // packed fma
          p[c] = __builtin_elementwise_fma(p[c], t[c], (f32x2){0.5f, 0.25f});
        } else if (MODE == 2) {     // two scalar fmas (same flops as packed)
          p[c][0] = __builtin_fmaf(p[c][0], t[c][0], 0.5f);
          p[c][1] = __builtin_fmaf(p[c][1], t[c][1], 0.25f);
        } else if (MODE == 3) {     // v_dot2_f32_f16 (fp32 accumulate)
          typedef _Float16 h2 __attribute__((ext_vector_type(2)));
          p[c][0] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, t[c][0]), __builtin_bit_cast(h2, t[c][1]), p[c][0], false);
        } else if (MODE == 4) {     // v_pk_fma_f16
          typedef _Float16 h2 __attribute__((ext_vector_type(2)));
          h2 a = __builtin_bit_cast(h2, p[c][0]);
          a = __builtin_elementwise_fma(a, __builtin_bit_cast(h2, t[c][0]), __builtin_bit_cast(h2, t[c][1]));
          p[c][0] = __builtin_bit_cast(float, a);
        } else if (MODE == 5) {     // v_exp_f32
          p[c][0] = __builtin_amdgcn_exp2f(p[c][0]);
        } else if (MODE == 6) {     // v_cvt_pk_f16_f32-ish round trip: cvt f32->f16->f32
          p[c][0] = (float)(_Float16)p[c][0] + 1.0f;
        } else if (MODE == 7) {     // v_dot2_f32_bf16
          typedef __bf16 b2 __attribute__((ext_vector_type(2)));
          typedef short s2 __attribute__((ext_vector_type(2)));
          p[c][0] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, t[c][0]), __builtin_bit_cast(b2, t[c][1]), p[c][0], false);
        }
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) s += p[c][0] + p[c][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int CHAINS> void run(const char* name, int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&cyc, 8);
  hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = 256.0 * 8 * CHAINS * (MODE == 2 ? 2 : 1);
  const double waves_per_simd = threads / 256.0;
  // wall: ns per wave-instruction per SIMD = ms / (n * waves_per_simd)
  printf("%-28s chains %d threads %4d: %6.2f ticks/wave-instr (own clock) | wall %7.1f us -> %5.2f ns per wave-instr per SIMD, tick = %.3f ns\n",
         name, CHAINS, threads, h / n, ms * 1e3, ms * 1e6 / (n * waves_per_simd), ms * 1e6 / h);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int threads : {256, 1024}) {
    run<0, 1>("v_fma_f32 dependent", threads);
    run<0, 4>("v_fma_f32", threads);
    run<0, 8>("v_fma_f32", threads);
    run<1, 1>("v_pk_fma_f32 dependent", threads);
    run<1, 2>("v_pk_fma_f32", threads);
    run<1, 4>("v_pk_fma_f32", threads);
    run<1, 8>("v_pk_fma_f32", threads);
    run<2, 4>("2x v_fma_f32", threads);
    run<3, 1>("v_dot2_f32_f16 dependent", threads);
    run<3, 4>("v_dot2_f32_f16", threads);
    run<3, 8>("v_dot2_f32_f16", threads);
    run<7, 8>("v_dot2_f32_bf16", threads);
    run<4, 1>("v_pk_fma_f16 dependent", threads);
    run<4, 4>("v_pk_fma_f16", threads);
    run<4, 8>("v_pk_fma_f16", threads);
    run<5, 8>("v_exp_f32", threads);
    run<6, 8>("cvt f32->f16->f32 + add (3 instr)", threads);
  }
  return 0;
}
