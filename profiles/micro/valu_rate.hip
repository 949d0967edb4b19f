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
        } else {                    // two scalar fmas (same flops as packed)
          p[c][0] = __builtin_fmaf(p[c][0], t[c][0], 0.5f);
          p[c][1] = __builtin_fmaf(p[c][1], t[c][1], 0.25f);
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
  hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = 256.0 * 8 * CHAINS * (MODE == 2 ? 2 : 1);
  printf("%-28s chains %d threads %4d: %6.2f cycles per wave-instruction\n", name, CHAINS, threads, h / n);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int threads : {256, 512}) {
    run<0, 1>("v_fma_f32 dependent", threads);
    run<0, 4>("v_fma_f32", threads);
    run<0, 8>("v_fma_f32", threads);
    run<1, 1>("v_pk_fma_f32 dependent", threads);
    run<1, 2>("v_pk_fma_f32", threads);
    run<1, 4>("v_pk_fma_f32", threads);
    run<1, 8>("v_pk_fma_f32", threads);
    run<2, 4>("2x v_fma_f32", threads);
  }
  return 0;
}
