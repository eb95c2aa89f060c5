// Does gfx950 execute scalar atomics (s_atomic_add ... glc), and are they coherent across XCDs?  Every workgroup draws
// 64 tickets from one counter with s_atomic_add; the host checks that the tickets are 0..N-1, each exactly once.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(uint32_t* ctr, uint32_t* out, int per) {
  for (int i = 0; i < per; ++i) {
    uint32_t t = 1u;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(ctr) : "memory");
    if (threadIdx.x == 0) out[(size_t)blockIdx.x * per + i] = t;
  }
}
int main() {
  const int blocks = 2048, per = 64;
  uint32_t *ctr, *out;
  hipMalloc(&ctr, 256); hipMemset(ctr, 0, 256); hipMalloc(&out, (size_t)blocks * per * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, ctr, out, per);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint32_t> h((size_t)blocks * per);
  hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  uint32_t c; hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  size_t bad = 0;
  for (size_t i = 0; i < h.size(); ++i) bad += h[i] != i;
  printf("counter %u (want %d), %zu tickets out of place, %.3f ms = %.1f ns per ticket chip-wide\n", c, blocks * per, bad, ms, ms * 1e6 / (blocks * per));
  return bad != 0;
}
