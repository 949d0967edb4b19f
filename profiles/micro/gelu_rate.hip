// Cost of one GELU per hidden element for several formulations (ticks per element per wave, and wall ns per element per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -I genconvit_amd/csrc profiles/micro/gelu_rate.hip -o /tmp/gelu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "gemm.h"
using namespace gcv;

// scalar-fma variant of the same polynomial, NCH independent chains
template <int NCH> __device__ __forceinline__ void gelu_fma_n(float (&x)[NCH]) {
  constexpr float kC[11] = {2.749713404e-02f, -1.330395067e-01f, 2.465923971e-01f, -1.472158060e-01f,
                            -2.029683018e-01f, 4.347813707e-01f, -2.049071560e-01f, -1.763150062e-01f,
                            1.763803063e-01f, 2.178248281e-02f, -4.258673483e-02f};
  float t[NCH], p[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    t[c] = fmaf(fminf(fabsf(x[c]), 4.5f), 0.44444444444f, -1.0f);
    p[c] = fmaf(kC[10], t[c], kC[9]);
  }
#pragma unroll
  for (int k = 8; k >= 0; --k)
#pragma unroll
    for (int c = 0; c < NCH; ++c) p[c] = fmaf(p[c], t[c], kC[k]);
#pragma unroll
  for (int c = 0; c < NCH; ++c) x[c] = fmaxf(x[c], 0.0f) - p[c];
}

template <int MODE> __global__ void k(float* out, long long* cyc, float seed) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed * (i - 7.5f) * 0.3f + 1e-3f * threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
    if (MODE == 0) {                      // exact A&S erf, scalar
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = act_fn<ACT_GELU>(v[i]) + 0.25f;
    } else if (MODE == 1) {               // packed polynomial, 4 chains (as in the kernels)
#pragma unroll
      for (int g = 0; g < 16; g += 8) {
        f32x2 x[4] = {{v[g], v[g + 1]}, {v[g + 2], v[g + 3]}, {v[g + 4], v[g + 5]}, {v[g + 6], v[g + 7]}};
        gelu_pk_n<4>(x);
#pragma unroll
        for (int c = 0; c < 4; ++c) { v[g + 2 * c] = x[c][0] + 0.25f; v[g + 2 * c + 1] = x[c][1] + 0.25f; }
      }
    } else if (MODE == 2) {               // packed polynomial, 8 chains
      f32x2 x[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) x[c] = (f32x2){v[2 * c], v[2 * c + 1]};
      gelu_pk_n<8>(x);
#pragma unroll
      for (int c = 0; c < 8; ++c) { v[2 * c] = x[c][0] + 0.25f; v[2 * c + 1] = x[c][1] + 0.25f; }
    } else if (MODE == 3) {               // scalar-fma polynomial, 8 chains
#pragma unroll
      for (int g = 0; g < 16; g += 8) {
        float x[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = v[g + c];
        gelu_fma_n<8>(x);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[g + c] = x[c] + 0.25f;
      }
    } else {                              // scalar-fma polynomial, 16 chains
      gelu_fma_n<16>(v);
#pragma unroll
      for (int c = 0; c < 16; ++c) v[c] += 0.25f;
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE> void run(const char* name, int threads) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 4 * 1024 * 1024); (void)hipMalloc(&cyc, 8);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double elems = 256.0 * 16;                       // per lane
  printf("%-34s waves/SIMD %d: %6.1f ticks per element (own clock) | wall %6.2f ns per element per SIMD\n", name, threads / 256,
         h / elems, ms * 1e6 / (elems * (threads / 256.0)));
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  for (int threads : {256, 512}) {
    run<0>("erf (A&S 7.1.26), scalar", threads);
    run<1>("poly deg 10, v_pk_fma x4 chains", threads);
    run<2>("poly deg 10, v_pk_fma x8 chains", threads);
    run<3>("poly deg 10, v_fma x8 chains", threads);
    run<4>("poly deg 10, v_fma x16 chains", threads);
  }
  return 0;
}
