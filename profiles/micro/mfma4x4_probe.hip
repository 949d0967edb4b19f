// v_mfma_f32_4x4x4_16B_f16 (16 independent 4x4x4 blocks per instruction: the only MFMA shape whose blocks can be the
// CHANNELS of a depthwise convolution): operand / result layout probe and issue rate on gfx950.
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/mfma4x4_probe.hip -o /tmp/m44 && /tmp/m44
// Layout hypothesis checked: A[i][k] of block b in lane 4b+i, element k of the lane's four halves; B[k][j] in lane 4b+j,
// element k; D[i][j] in lane 4b+j, register i.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout(const _Float16* A, const _Float16* B, float* D) {   // A, B: [64 lanes][4]
  const int l = threadIdx.x;
  h4 a = {A[l * 4], A[l * 4 + 1], A[l * 4 + 2], A[l * 4 + 3]};
  h4 b = {B[l * 4], B[l * 4 + 1], B[l * 4 + 2], B[l * 4 + 3]};
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[l * 4 + i] = c[i];
}
__global__ void layout_bf(const __bf16* A, const __bf16* B, float* D) {
  const int l = threadIdx.x;
  b4 a = {A[l * 4], A[l * 4 + 1], A[l * 4 + 2], A[l * 4 + 3]};
  b4 b = {B[l * 4], B[l * 4 + 1], B[l * 4 + 2], B[l * 4 + 3]};
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s4, a), __builtin_bit_cast(s4, b), c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[l * 4 + i] = c[i];
}

#define MF(c) "v_mfma_f32_4x4x4_16b_f16 v[" #c ":" #c "+3], v[40:41], v[42:43], v[" #c ":" #c "+3]\n"
#define MF7 MF(0) MF(4) MF(8) MF(12) MF(16) MF(20) MF(24)
#define V1(c) "v_fma_f32 v" #c ", v44, v45, v" #c "\n"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23", \
             "v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45"
#define KERNEL(NAME, BODY, NMF)                                                                \
  __global__ void __launch_bounds__(1024) NAME(long long* cyc, float* out) {                   \
    asm volatile("v_mov_b32 v40, 0\nv_mov_b32 v41, 0\nv_mov_b32 v42, 0\nv_mov_b32 v43, 0\nv_mov_b32 v44, 1.0\nv_mov_b32 v45, 0.5\n" ::: CLOB); \
    long long t0 = __builtin_amdgcn_s_memtime();                                               \
    _Pragma("unroll 1") for (int it = 0; it < 500; ++it) {                                     \
      asm volatile(BODY BODY BODY BODY ::: CLOB);                                              \
    }                                                                                          \
    long long t1 = __builtin_amdgcn_s_memtime();                                               \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                   \
    if (t1 == 12345) out[threadIdx.x] = 1.0f;                                                  \
  }
KERNEL(k_mf7, MF7, 7)                                            // seven independent accumulators, as the dw walk has
KERNEL(k_mf7_v2, MF7 V1(32) V1(33), 7)                           // + 2 vector instructions per 7 MFMAs
KERNEL(k_mf7_v7, MF7 V1(32) V1(33) V1(34) V1(35) V1(36) V1(37) V1(38), 7)
KERNEL(k_mf1, MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0), 7)      // one dependent chain

int main() {
  _Float16 hA[256], hB[256];
  __bf16 bA[256], bB[256];
  srand(1);
  for (int i = 0; i < 256; ++i) {
    hA[i] = (_Float16)((rand() % 9) - 4); hB[i] = (_Float16)((rand() % 9) - 4);
    bA[i] = (__bf16)(float)hA[i]; bB[i] = (__bf16)(float)hB[i];
  }
  void *dA, *dB; float* dD; long long* dc;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 1024 * 4); hipMalloc(&dc, 8);
  float hD[256];
  for (int bf = 0; bf < 2; ++bf) {
    hipMemcpy(dA, bf ? (void*)bA : (void*)hA, 512, hipMemcpyHostToDevice);
    hipMemcpy(dB, bf ? (void*)bB : (void*)hB, 512, hipMemcpyHostToDevice);
    if (bf) layout_bf<<<1, 64>>>((const __bf16*)dA, (const __bf16*)dB, dD);
    else layout<<<1, 64>>>((const _Float16*)dA, (const _Float16*)dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 16; ++b)
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          float want = 0;
          for (int k = 0; k < 4; ++k) want += (float)hA[(4 * b + i) * 4 + k] * (float)hB[(4 * b + j) * 4 + k];
          const float got = hD[(4 * b + j) * 4 + i];
          if (fabsf(got - want) > 1e-3f) ++bad;
        }
    printf("%s layout (A[i][k]: lane 4b+i elem k; B[k][j]: lane 4b+j elem k; D[i][j]: lane 4b+j reg i): %s (%d mismatches)\n",
           bf ? "bf16_1k" : "f16", bad ? "WRONG" : "confirmed", bad);
  }
  struct { const char* name; void (*k)(long long*, float*); int nmf; } ks[] = {
      {"7 independent accumulators", k_mf7, 7}, {"7 MFMA + 2 v_fma", k_mf7_v2, 7}, {"7 MFMA + 7 v_fma", k_mf7_v7, 7},
      {"one dependent chain", k_mf1, 7}};
  for (int waves = 4; waves <= 16; waves *= 2)
    for (auto& k : ks) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      k.k<<<256, waves * 64>>>(dc, dD);
      hipEventRecord(e0);
      k.k<<<256, waves * 64>>>(dc, dD);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
      const double n = 500.0 * 4 * k.nmf;
      printf("%2d waves/CU  %-28s %6.2f cycles per MFMA (wave 0), %6.2f ns per MFMA and SIMD (wall)\n", waves, k.name,
             c / n, ms * 1e6 / (n * waves / 4.0));
    }
  return 0;
}
