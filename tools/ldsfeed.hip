// Why does an LDS-fed store loop stream slower than the same stores alone?  (profiles/r02_tuning.md)
// Same addresses as the headline kernel's render loop: per wave 75 iterations x 9 dword stores (256 B each),
// 1,048,576 environments x 2,700 B.  Every variant runs with the same dynamic LDS size (occupancy).
// Build: hipcc --offload-arch=gfx950 -O3 -o ldsfeed tools/ldsfeed.hip ; run: ./ldsfeed [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k(uint8_t* planes, uint32_t* sink) {
  extern __shared__ uint32_t sh[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4800; i += 64) sh[i] = MODE == 9 ? (uint32_t)i : i * 2654435761u;
  __syncthreads();
  uint8_t* const blk = planes + (size_t)blockIdx.x * 64 * 2700;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)blk), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)blk >> 32));
  uint8_t* const sblk = reinterpret_cast<uint8_t*>(((uint64_t)hi << 32) | lo);
  uint32_t acc = 0, pf = sh[lane];
  uint4 quad = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < 75; ++it) {
    const uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
    const uint32_t voff = e * 2700 + q * 4;
    uint32_t v;
    if (MODE == 0) v = f;
    else if (MODE == 1 || MODE == 6 || MODE == 9) v = sh[f];
    else if (MODE == 8) v = f * 2654435761u;
    else if (MODE == 2) { acc ^= sh[f]; v = f; }
    else if (MODE == 3) { v = pf; pf = sh[f + 64 < 4800 ? f + 64 : 0]; }
    else if (MODE == 4) v = __builtin_amdgcn_ds_bpermute(((lane + it) & 63) << 2, (int)f);
    else if (MODE == 5) v = __builtin_amdgcn_readlane((int)f, it & 63) + lane;
    else if (MODE == 7) { if ((it & 3) == 0) quad = *reinterpret_cast<const uint4*>(&sh[(it >> 2) * 256 + lane * 4]); v = (it & 3) == 0 ? quad.x : (it & 3) == 1 ? quad.y : (it & 3) == 2 ? quad.z : quad.w; }
    else v = f;
    if (MODE == 6) {
#pragma unroll
      for (int p = 0; p < 9; ++p) {
        const uint32_t d = v + p;
        uint8_t* base = sblk + p * 300;
        asm volatile("global_store_dword %0, %1, %2" : : "v"(voff), "v"(d), "s"(base));
      }
    } else {
#pragma unroll
      for (int p = 0; p < 9; ++p) *reinterpret_cast<uint32_t*>(blk + voff + p * 300) = v + p;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// The LDS reads of CH iterations are issued together, one wait, then CH iterations of stores from registers.
template <int CH>
__global__ __launch_bounds__(64) void kc(uint8_t* planes, uint32_t* sink) {
  extern __shared__ uint32_t sh[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4800; i += 64) sh[i] = i * 2654435761u;
  __syncthreads();
  uint8_t* const blk = planes + (size_t)blockIdx.x * 64 * 2700;
#pragma unroll 1
  for (int it0 = 0; it0 < 75; it0 += CH) {
    uint32_t c[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) c[j] = sh[(it0 + j) * 64 + lane];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const uint32_t f = (it0 + j) * 64 + lane, e = f / 75, q = f - e * 75;
      const uint32_t voff = e * 2700 + q * 4;
#pragma unroll
      for (int p = 0; p < 9; ++p) *reinterpret_cast<uint32_t*>(blk + voff + p * 300) = c[j] + p;
    }
  }
}

int main(int argc, char** argv) {
  const size_t envs = 1 << 20, bytes = envs * 2700;
  const int lds = argc > 1 ? atoi(argv[1]) : 20480;
  uint8_t* a; uint32_t* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMemset(a, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"stores only (data = index)", "LDS read feeds the stores", "LDS read, result unused by the stores", "LDS read prefetched one iteration ahead",
                         "ds_bpermute feeds the stores (LDS crossbar, no LDS memory)", "v_readlane feeds the stores (no LDS at all)",
                         "LDS read feeds the stores, scalar-base stores", "one ds_read_b128 per four iterations feeds the stores",
                         "stores only, data = index x 2654435761 (VALU, high entropy)", "LDS read feeds the stores, LDS holds the index (low entropy)",
                         "LDS reads of 5 iterations together, then their stores", "LDS reads of 15 iterations together, then their stores",
                         "LDS reads of 25 iterations together, then their stores", "LDS reads of all 75 iterations, then the stores"};
  for (int round = 0; round < 2; ++round)
    for (int mode = 0; mode < 14; ++mode) {
      float best = 1e9;
      for (int rep = 0; rep < 10; ++rep) {
        CK(hipEventRecord(e0));
        switch (mode) {
          case 0: k<0><<<envs / 64, 64, lds>>>(a, sink); break;
          case 1: k<1><<<envs / 64, 64, lds>>>(a, sink); break;
          case 2: k<2><<<envs / 64, 64, lds>>>(a, sink); break;
          case 3: k<3><<<envs / 64, 64, lds>>>(a, sink); break;
          case 4: k<4><<<envs / 64, 64, lds>>>(a, sink); break;
          case 5: k<5><<<envs / 64, 64, lds>>>(a, sink); break;
          case 6: k<6><<<envs / 64, 64, lds>>>(a, sink); break;
          case 7: k<7><<<envs / 64, 64, lds>>>(a, sink); break;
          case 8: k<8><<<envs / 64, 64, lds>>>(a, sink); break;
          case 9: k<9><<<envs / 64, 64, lds>>>(a, sink); break;
          case 10: kc<5><<<envs / 64, 64, lds>>>(a, sink); break;
          case 11: kc<15><<<envs / 64, 64, lds>>>(a, sink); break;
          case 12: kc<25><<<envs / 64, 64, lds>>>(a, sink); break;
          case 13: kc<75><<<envs / 64, 64, lds>>>(a, sink); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2 && ms < best) best = ms;
      }
      printf("lds %5d  %-62s %7.3f ms  %7.1f GB/s\n", lds, names[mode], best, bytes / best / 1e6);
    }
  return 0;
}
