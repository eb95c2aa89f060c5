// Probe of the two primitives the persistent shapes rest on: LDS-DMA rows (global_load_lds_dword with M0) and the
// un-waited ticket atomic.  hipcc --offload-arch=gfx950 -O3 ldsdma_probe.hip -o ldsdma_probe && ./ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma_row(const uint32_t* base, uint32_t voff, uint32_t lds_addr) {
  uint32_t keep; uint64_t own;
  asm volatile("s_mov_b64 %1, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "=&s"(own) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
__global__ void k(const uint32_t* src, uint32_t* dst, int rows, int64_t pitch, int at) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x;
  uint32_t* inbox = lds + at;  // (round 4, second use: the generic kernel's columns may lie above 64 KB -- M0 must carry the whole address)
  const uint32_t ib = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)inbox;
  for (int i = lane; i < 64 * rows; i += 64) inbox[i] = 0xDEAD0000u + 300 + i;
  __syncthreads();
  const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)blockIdx.x);
  if (lane < 48)
    for (int w = 0; w < rows; ++w) dma_row(src + (int64_t)w * pitch + (int64_t)u * 64, 4u * lane, ib + w * 256u);
  // staging-like traffic between the DMA and its wait: global loads, ds_writes elsewhere, a barrier
  for (int i = lane; i < 300; i += 64) lds[i] = src[(i * 7 + blockIdx.x) % (rows * pitch)];
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int w = 0; w < rows; ++w) dst[((int64_t)blockIdx.x * rows + w) * 64 + lane] = inbox[w * 64 + lane];
  if (blockIdx.x == 0 && lane == 0) dst[(int64_t)gridDim.x * rows * 64] = ib;
}
int main() {
  const int rows = 15, blocks = 4096; const int64_t pitch = 64 * blocks;
  std::vector<uint32_t> h(rows * pitch);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 1000u + (uint32_t)i;
  uint32_t *src, *dst;
  hipMalloc(&src, h.size() * 4); hipMalloc(&dst, (blocks * rows * 64 + 1) * 4);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  int total = 0;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int at : {300, 16384 + 8, 32768 + 12, 39000 - 64 * rows}) {
  for (int rep = 0; rep < 5; ++rep)
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), (at + 64 * rows) * 4 + 512, 0, src, dst, rows, pitch, at);
  std::vector<uint32_t> o(blocks * rows * 64 + 1);
  hipMemcpy(o.data(), dst, o.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < rows; ++w) for (int l = 0; l < 64; ++l) {
    uint32_t got = o[(b * rows + w) * 64 + l], want = l < 48 ? 1000u + (uint32_t)(w * pitch + b * 64 + l) : 0xDEAD0000u + 300 + w * 64 + l;
    if (got != want && bad++ < 10) printf("block %d row %d lane %d: got %u (0x%x) want %u\n", b, w, l, got, got, want);
  }
  printf("inbox LDS byte address %u; %d mismatches\n", o.back(), bad);
  total += bad;
  }
  return total != 0;
}
