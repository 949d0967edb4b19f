// Does a VALU write to the data registers of a 16-byte buffer store need wait states on gfx950 when the store carries an
// SGPR soffset?  The CDNA ISA hazard table (and LLVM's GCNHazardRecognizer::createsVALUHazard) say the ">64-bit VMEM store
// followed by a VALU write of its data VGPRs" hazard exists only when soffset is NOT a register.  dwconv_roll.h saw stale
// dwords in lanes 12-15 of each 16-lane row with exactly that form (x4 store, SGPR soffset, data registers rewritten right
// behind it), so this reproduces the sequence in isolation:
//     buffer_store_dwordx{2,4} v[0:3], v4, s[rsrc], <soffset> offen ; [s_nop n] ; v_mov_b32 v0..v3, POISON
// and counts stored dwords that came out as POISON (or anything but the value the lane held when the store issued).
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/store_hazard.hip -o /tmp/sh && /tmp/sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kPoison = 0xDEADBEEFu;
constexpr int kIters = 64, kWaves = 8, kBlocks = 1024;

// fixed registers: v[10:13] take the data (settled by an s_nop), the store issues, the overwrite follows
#define LOADD "v_mov_b32 v10, %0\n\tv_mov_b32 v11, %1\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\ts_nop 7\n\t"
#define OVERWRITE "v_mov_b32 v10, %6\n\tv_mov_b32 v11, %6\n\tv_mov_b32 v12, %6\n\tv_mov_b32 v13, %6\n\ts_nop 7\n\t"
#define OPS : : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(voff), "s"(rsv), "v"(kPoison), "s"(soff) : "memory", "v10", "v11", "v12", "v13"

template <int MODE> __global__ void __launch_bounds__(64 * kWaves) k(unsigned* out) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned tile = blockIdx.x * kWaves + wave;                 // one 1 KB row per (tile, iteration)
  const unsigned long long pa = (unsigned long long)out;               // raw buffer descriptor: base, stride 0, records, flags
  const u32x4 rsv = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pa),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pa >> 32)) & 0xffffu, 0x7fffffffu, 0x00020000u};
  const unsigned voff = lane * 16;
  for (int it = 0; it < kIters; ++it) {
    unsigned d0 = (tile << 12) | (it << 6) | lane, d1 = d0 ^ 0x11111111u, d2 = d0 ^ 0x22222222u, d3 = d0 ^ 0x33333333u;
    const unsigned soff = __builtin_amdgcn_readfirstlane((tile * kIters + it) * 1024u);
    if (MODE == 0)        // x4, SGPR soffset, no wait state
      asm volatile(LOADD "buffer_store_dwordx4 v[10:13], %4, %5, %7 offen\n\t" OVERWRITE OPS);
    else if (MODE == 1)   // x4, SGPR soffset, s_nop 0
      asm volatile(LOADD "buffer_store_dwordx4 v[10:13], %4, %5, %7 offen\n\ts_nop 0\n\t" OVERWRITE OPS);
    else if (MODE == 2)   // x4, SGPR soffset, s_nop 1
      asm volatile(LOADD "buffer_store_dwordx4 v[10:13], %4, %5, %7 offen\n\ts_nop 1\n\t" OVERWRITE OPS);
    else if (MODE == 3) { // x4, soffset = 0 (the documented hazard; hipcc would insert the nop, this asm does not)
      const unsigned vo2 = voff + soff;
      asm volatile(LOADD "buffer_store_dwordx4 v[10:13], %4, %5, 0 offen\n\t" OVERWRITE
                   : : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(vo2), "s"(rsv), "v"(kPoison), "s"(soff) : "memory", "v10", "v11", "v12", "v13");
    } else                // two x2 stores, SGPR soffset, no wait state (what dwconv_roll.h ships)
      asm volatile(LOADD "buffer_store_dwordx2 v[10:11], %4, %5, %7 offen\n\tbuffer_store_dwordx2 v[12:13], %4, %5, %7 offen offset:8\n\t" OVERWRITE OPS);
  }
}

template <int MODE> void run(const char* name) {
  const size_t n = (size_t)kBlocks * kWaves * kIters * 256;           // dwords
  unsigned* out;
  (void)hipMalloc(&out, n * 4);
  (void)hipMemset(out, 0, n * 4);
  std::vector<unsigned> h(n);
  long long bad = 0, bad_lane[4] = {0, 0, 0, 0};
  for (int rep = 0; rep < 4; ++rep) {
    hipLaunchKernelGGL((k<MODE>), dim3(kBlocks), dim3(64 * kWaves), 0, 0, out);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
    for (size_t row = 0; row < (size_t)kBlocks * kWaves * kIters; ++row) {
      const unsigned tile = (unsigned)(row / kIters), it = (unsigned)(row % kIters);
      for (unsigned lane = 0; lane < 64; ++lane) {
        const unsigned d0 = (tile << 12) | (it << 6) | lane;
        const unsigned want[4] = {d0, d0 ^ 0x11111111u, d0 ^ 0x22222222u, d0 ^ 0x33333333u};
        for (int e = 0; e < 4; ++e)
          if (h[row * 256 + lane * 4 + e] != want[e]) { ++bad; ++bad_lane[(lane & 15) >> 2]; }
      }
    }
  }
  printf("%-58s wrong dwords: %lld of %zu  (by lane&15 quarter 0-3 / 4-7 / 8-11 / 12-15: %lld %lld %lld %lld)\n", name, bad,
         n * 4, bad_lane[0], bad_lane[1], bad_lane[2], bad_lane[3]);
  (void)hipFree(out);
}

int main() {
  run<0>("x4 store, SGPR soffset, VALU overwrite next");
  run<1>("x4 store, SGPR soffset, s_nop 0, overwrite");
  run<2>("x4 store, SGPR soffset, s_nop 1, overwrite");
  run<3>("x4 store, soffset 0 (documented hazard), overwrite next");
  run<4>("2 x x2 stores, SGPR soffset, overwrite next");
  return 0;
}
