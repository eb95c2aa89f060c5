// Where do the waves of a workgroup land?  Prints (workgroup, wave) -> XCC, SE, CU, SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o hwid hwid.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__global__ void probe(uint32_t* out, int spin) {
  extern __shared__ uint32_t sh[];
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
  const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
  uint32_t v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u;    // keep waves resident for a while
  sh[threadIdx.x] = v;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    out[2 * w] = hw; out[2 * w + 1] = (xcc & 0xF) | (sh[threadIdx.x] & 0x80000000u ? 0 : 0);
  }
}
int main(int argc, char** argv) {
  const int wg = argc > 1 ? atoi(argv[1]) : 512, thr = argc > 2 ? atoi(argv[2]) : 256, lds = argc > 3 ? atoi(argv[3]) : 55000;
  const int nw = wg * thr / 64;
  uint32_t* d; hipMalloc(&d, nw * 8);
  probe<<<wg, thr, lds>>>(d, 20000);
  uint32_t* h = (uint32_t*)malloc(nw * 8);
  hipMemcpy(h, d, nw * 8, hipMemcpyDeviceToHost);
  int hist[8][4] = {};  // wave-in-WG x simd
  for (int w = 0; w < nw; ++w) {
    const uint32_t hw = h[2 * w];
    const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, se = (hw >> 13) & 7, wave = hw & 15;
    if (w < 48) printf("wg %3d wave %d: xcc %u se %d cu %2d simd %d slot %d\n", w / (thr / 64), w % (thr / 64), h[2 * w + 1], se, cu, simd, wave);
    hist[w % (thr / 64) % 8][simd]++;
  }
  for (int i = 0; i < thr / 64 && i < 8; ++i) printf("wave-in-WG %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", i, hist[i][0], hist[i][1], hist[i][2], hist[i][3]);
  return 0;
}
