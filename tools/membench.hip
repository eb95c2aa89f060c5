// membench.hip -- store-bandwidth ceilings on the box the step kernel runs on.
// Build: hipcc --offload-arch=gfx950 -O3 -o membench tools/membench.hip
// Modes: 0 dwordx4 contiguous fill, 1 dword contiguous fill,
//        2 the step kernel's store pattern (per 64-env block: 75 iterations x
//          9 dword stores at plane stride 300 B inside a 2700 B env record),
//        3 pattern 2 with dwordx4 stores (flat 16 B chunks of the block region),
//        4 float4 copy (read + write) for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void fill_x4(uint4* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  uint4 v = make_uint4(i, 1, 2, 3);
  for (; i < n; i += stride) p[i] = v;
}
__global__ void fill_x1(uint32_t* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = (uint32_t)i;
}
// one wave per 64 envs, same addressing as pcx_scrolly_maze_step phase B
__global__ __launch_bounds__(64) void fill_pattern(uint8_t* planes) {
  const int lane = threadIdx.x;
  uint8_t* blk = planes + (size_t)blockIdx.x * 64 * 2700;
  for (int it = 0; it < 75; ++it) {
    uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
    uint8_t* dst = blk + e * 2700 + q * 4;
#pragma unroll
    for (int p = 0; p < 9; ++p) *reinterpret_cast<uint32_t*>(dst + p * 300) = f + p;
  }
}
// pattern 2 under the step kernel's occupancy (dynamic LDS per wave)
__global__ __launch_bounds__(64) void fill_pattern_lds(uint8_t* planes) {
  extern __shared__ uint32_t sh[];
  const int lane = threadIdx.x;
  sh[lane] = lane;
  __syncthreads();
  uint8_t* blk = planes + (size_t)blockIdx.x * 64 * 2700;
  for (int it = 0; it < 75; ++it) {
    uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
    uint8_t* dst = blk + e * 2700 + q * 4;
    uint32_t v = sh[e];
#pragma unroll
    for (int p = 0; p < 9; ++p) *reinterpret_cast<uint32_t*>(dst + p * 300) = v + p;
  }
}
// LDS allocated but never touched
__global__ __launch_bounds__(64) void fill_pattern_lds_unused(uint8_t* planes) {
  extern __shared__ uint32_t sh[];
  const int lane = threadIdx.x;
  if (planes == nullptr) sh[lane] = lane;
  uint8_t* blk = planes + (size_t)blockIdx.x * 64 * 2700;
  for (int it = 0; it < 75; ++it) {
    uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
    uint8_t* dst = blk + e * 2700 + q * 4;
#pragma unroll
    for (int p = 0; p < 9; ++p) *reinterpret_cast<uint32_t*>(dst + p * 300) = f + p;
  }
}
// 256-thread blocks: 4 independent waves, each 64 envs, LDS read per iteration
__global__ __launch_bounds__(256) void fill_pattern_lds_256(uint8_t* planes) {
  extern __shared__ uint32_t sh[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  sh[threadIdx.x] = lane;
  __syncthreads();
  uint8_t* blk = planes + ((size_t)blockIdx.x * 4 + wave) * 64 * 2700;
  for (int it = 0; it < 75; ++it) {
    uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
    uint8_t* dst = blk + e * 2700 + q * 4;
    uint32_t v = sh[wave * 64 + e];
#pragma unroll
    for (int p = 0; p < 9; ++p) *reinterpret_cast<uint32_t*>(dst + p * 300) = v + p;
  }
}
// plane-major 16 B chunks: iteration = (plane p uniform, lanes over (env, chunk)); LDS read per iteration
struct __attribute__((packed, aligned(4))) u4u { uint32_t x, y, z, w; };
__global__ __launch_bounds__(64) void fill_chunks_x4(uint8_t* planes) {
  extern __shared__ uint32_t sh[];
  const int lane = threadIdx.x;
  sh[lane] = lane;
  __syncthreads();
  uint8_t* blk = planes + (size_t)blockIdx.x * 64 * 2700;
  for (int p = 0; p < 9; ++p)
    for (int it = 0; it < 19; ++it) {
      uint32_t f = it * 64 + lane, e = f / 19, c = f - e * 19;
      uint32_t v = sh[e] + p;
      uint8_t* dst = blk + e * 2700 + p * 300 + c * 16;
      if (c < 18) { u4u val = {v, v + 1, v + 2, v + 3}; *reinterpret_cast<u4u*>(dst) = val; }
      else { reinterpret_cast<uint32_t*>(dst)[0] = v; reinterpret_cast<uint32_t*>(dst)[1] = v; reinterpret_cast<uint32_t*>(dst)[2] = v; }
    }
}
// dword pattern + ~64 dependent VALU ops per iteration, no LDS
__global__ __launch_bounds__(64) void fill_pattern_valu(uint8_t* planes) {
  const int lane = threadIdx.x;
  uint8_t* blk = planes + (size_t)blockIdx.x * 64 * 2700;
  for (int it = 0; it < 75; ++it) {
    uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
    uint8_t* dst = blk + e * 2700 + q * 4;
    uint32_t v = f;
#pragma unroll
    for (int j = 0; j < 32; ++j) v = (v * 0x9E3779B1u) ^ (v >> 7);
#pragma unroll
    for (int p = 0; p < 9; ++p) *reinterpret_cast<uint32_t*>(dst + p * 300) = v + p;
  }
}
// same bytes, but the block's 172,800 B region written as flat 16 B chunks
__global__ __launch_bounds__(64) void fill_pattern_x4(uint4* planes) {
  const int lane = threadIdx.x;
  uint4* blk = planes + (size_t)blockIdx.x * (64 * 2700 / 16);
  for (int i = lane; i < 64 * 2700 / 16; i += 64) blk[i] = make_uint4(i, 1, 2, 3);
}
__global__ void copy_x4(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) b[i] = a[i];
}

int main() {
  const size_t envs = 1 << 20, bytes = envs * 2700;
  uint8_t *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"fill dwordx4", "fill dword", "step-kernel store pattern (dword)", "block-flat dwordx4", "copy dwordx4 (r+w)",
                         "store pattern, 4 KB LDS/wave", "store pattern, 8 KB LDS/wave", "store pattern, 10.5 KB LDS/wave", "store pattern, 16 KB LDS/wave",
                         "store pattern, 4 KB LDS unused", "store pattern, 256-thr blocks + LDS", "plane-chunk dwordx4 + LDS", "store pattern + 64 VALU/iter"};
  const int lds_bytes[] = {0, 0, 0, 0, 0, 4096, 8192, 10752, 16384, 4096, 4096, 4096, 0};
  for (int mode = 0; mode < 13; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 12; ++rep) {
      CK(hipEventRecord(e0));
      switch (mode) {
        case 0: fill_x4<<<2048 * 4, 256>>>((uint4*)a, bytes / 16); break;
        case 1: fill_x1<<<2048 * 4, 256>>>((uint32_t*)a, bytes / 4); break;
        case 2: fill_pattern<<<envs / 64, 64>>>(a); break;
        case 3: fill_pattern_x4<<<envs / 64, 64>>>((uint4*)a); break;
        case 4: copy_x4<<<2048 * 4, 256>>>((const uint4*)a, (uint4*)b, bytes / 16); break;
        case 9: fill_pattern_lds_unused<<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 10: fill_pattern_lds_256<<<envs / 256, 256, lds_bytes[mode]>>>(a); break;
        case 11: fill_chunks_x4<<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 12: fill_pattern_valu<<<envs / 64, 64>>>(a); break;
        default: fill_pattern_lds<<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2 && ms < best) best = ms;
    }
    double moved = mode == 4 ? 2.0 * bytes : (double)bytes;
    printf("%-36s %8.3f ms  %8.1f GB/s\n", names[mode], best, moved / best / 1e6);
  }
  return 0;
}
