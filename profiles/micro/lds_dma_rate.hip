// L2 -> LDS fill rate of the LDS-DMA ring used by gemm_glds_kernel, as a function of the bytes fetched per row and
// stage (64 B = half a 128-B line, 128 B = a whole line) and of the ring depth.  No MFMA: the loop only waits.
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/lds_dma_rate.hip -o /tmp/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int BKB, int S, int ROWS>
__global__ void __launch_bounds__(256) fill(const unsigned char* __restrict__ A, const unsigned char* __restrict__ W,
                                            int M, int Kbytes, int nkt, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = ROWS * BKB;
  constexpr int LPR = BKB / 16;                 // lanes per row
  constexpr int RPI = 64 / LPR;                 // rows per DMA wave-instruction
  constexpr int NQ = ROWS / RPI;                // instructions per stage
  constexpr int QPW = NQ / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = (blockIdx.x % (M / 128)) * 128;
  const unsigned char* src[QPW];
#pragma unroll
  for (int i = 0; i < QPW; ++i) {
    const int q = wave + 4 * i;
    const int rl = q * RPI + lane / LPR;
    const int c = lane % LPR;
    src[i] = rl < 128 ? A + (int64_t)(m0 + rl) * Kbytes + c * 16 : W + (int64_t)(rl - 128) * Kbytes + c * 16;
  }
  auto issue = [&](int kt) {
    unsigned char* base = smem + (kt % S) * STAGE;
#pragma unroll
    for (int i = 0; i < QPW; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (int64_t)kt * BKB),
                                       (__attribute__((address_space(3))) void*)(base + (wave + 4 * i) * 1024), 16, 0, 0);
  };
  for (int s = 0; s < S - 1; ++s) issue(s);
  float acc = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + S - 1 < nkt) {
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((S - 2) * QPW) : "memory");
      issue(kt + S - 1);
    } else {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    acc += *(const float*)(smem + (kt % S) * STAGE + tid * 4);   // touch the stage
  }
  if (acc == 12345.f) sink[0] = acc;
}

template <int BKB, int S, int ROWS> void run(int wgs_per_cu, int Kbytes) {
  const int M = 128 * 392;                           // the pw1 C=384 activation matrix (50176 rows)
  unsigned char *A, *W; float* sink;
  (void)hipMalloc(&A, (size_t)M * Kbytes); (void)hipMalloc(&W, (size_t)(ROWS - 128) * Kbytes); (void)hipMalloc(&sink, 4);
  (void)hipMemset(A, 1, (size_t)M * Kbytes); (void)hipMemset(W, 1, (size_t)(ROWS - 128) * Kbytes);
  constexpr int SMEM = S * ROWS * BKB;
  (void)hipFuncSetAttribute((const void*)fill<BKB, S, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  const int nkt = Kbytes / BKB;
  const int grid = 256 * wgs_per_cu * 6;             // six rounds of workgroups
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((fill<BKB, S, ROWS>), dim3(grid), dim3(256), SMEM, 0, A, W, M, Kbytes, nkt, sink);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((fill<BKB, S, ROWS>), dim3(grid), dim3(256), SMEM, 0, A, W, M, Kbytes, nkt, sink);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  const double bytes = (double)grid * nkt * ROWS * BKB;
  printf("row piece %3d B, ring %d x %2d KB (%3d KB/WG), K = %4d B: %7.1f us  %6.2f TB/s into LDS  = %5.1f B/clk/CU at 2.0 GHz\n",
         BKB, S, ROWS * BKB / 1024, SMEM / 1024, Kbytes, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.0e9 / 1e3 * 1e3 / 1e3);
  (void)hipFree(A); (void)hipFree(W); (void)hipFree(sink);
}

int main() {
  for (int Kb : {768, 3072}) {
    run<64, 4, 320>(2, Kb);      // gemm_glds_kernel today: 64-B pieces, 4 x 20 KB
    run<128, 2, 320>(2, Kb);     // whole lines, double buffer
    run<128, 3, 320>(1, Kb);     // whole lines, 3 x 40 KB, one workgroup per CU
    run<64, 3, 320>(2, Kb);
    run<64, 2, 320>(2, Kb);
    run<64, 2, 320>(3, Kb);      // today's K = 384 configuration: three workgroups per CU
    run<64, 4, 320>(1, Kb);      // ONE workgroup per CU (a persistent kernel): how deep must the ring be?
    run<64, 6, 320>(1, Kb);
    run<64, 8, 320>(1, Kb);
  }
  return 0;
}
