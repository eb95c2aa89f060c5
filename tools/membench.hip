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
// Padded record layout: plane pitch 304 B (16-byte multiple), env stride 2736 B.
// 64 envs x 19 chunks = 19 iterations x 64 lanes, each lane stores one aligned
// 16-byte chunk to each of the 9 planes; VALU = dependent ops per iteration.
template <int VALU>
__global__ __launch_bounds__(64) void fill_padded_x4(uint8_t* planes) {
  extern __shared__ uint32_t sh[];
  const int lane = threadIdx.x;
  sh[lane] = lane;
  __syncthreads();
  uint8_t* blk = planes + (size_t)blockIdx.x * 64 * 2736;
  for (int it = 0; it < 19; ++it) {
    uint32_t f = it * 64 + lane, e = f / 19, c = f - e * 19;
    uint32_t v = sh[e] + c;
#pragma unroll
    for (int j = 0; j < VALU; ++j) v = (v << 3) ^ (v >> 5) ^ 0x9E3779B1u;
    uint4* dst = reinterpret_cast<uint4*>(blk + e * 2736 + c * 16);
#pragma unroll
    for (int p = 0; p < 9; ++p) dst[p * 19] = make_uint4(v + p, v, v ^ p, v);
  }
}
// same VALU/LDS load with the dword pattern of the step kernel
template <int VALU>
__global__ __launch_bounds__(64) void fill_pattern_work(uint8_t* planes) {
  extern __shared__ uint32_t sh[];
  const int lane = threadIdx.x;
  sh[lane] = lane;
  __syncthreads();
  uint8_t* blk = planes + (size_t)blockIdx.x * 64 * 2700;
  for (int it = 0; it < 75; ++it) {
    uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
    uint32_t v = sh[e] + q;
#pragma unroll
    for (int j = 0; j < VALU; ++j) v = (v << 3) ^ (v >> 5) ^ 0x9E3779B1u;
    uint8_t* dst = blk + e * 2700 + q * 4;
#pragma unroll
    for (int p = 0; p < 9; ++p) *reinterpret_cast<uint32_t*>(dst + p * 300) = v + p;
  }
}
// Hypothesis test: does a store-stalled wave block VALU issue of the other waves
// on its SIMD?  256-thread workgroups, 4 groups of 64 envs per workgroup.
//  mixed:      each of the 4 waves does stores + VALU for its own group;
//  segregated: wave 0 does ALL the stores (4 groups), waves 1-3 do ALL the VALU
//              work (same total instruction counts, results kept alive).
template <bool SEG, int VALU>
__global__ __launch_bounds__(256) void fill_roles(uint8_t* planes, uint32_t* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* blk4 = planes + (size_t)blockIdx.x * 4 * 64 * 2700;
  uint32_t acc = 0;
  if (!SEG) {
    uint8_t* blk = blk4 + (size_t)wave * 64 * 2700;
    for (int it = 0; it < 75; ++it) {
      uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
      uint32_t v = f;
#pragma unroll
      for (int j = 0; j < VALU; ++j) v = (v << 3) ^ (v >> 5) ^ 0x9E3779B1u;
      uint8_t* dst = blk + e * 2700 + q * 4;
#pragma unroll
      for (int p = 0; p < 9; ++p) *reinterpret_cast<uint32_t*>(dst + p * 300) = v + p;
    }
  } else if (wave == 0) {
    for (int g = 0; g < 4; ++g) {
      uint8_t* blk = blk4 + (size_t)g * 64 * 2700;
      for (int it = 0; it < 75; ++it) {
        uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
        uint8_t* dst = blk + e * 2700 + q * 4;
#pragma unroll
        for (int p = 0; p < 9; ++p) *reinterpret_cast<uint32_t*>(dst + p * 300) = f + p;
      }
    }
  } else {
    // 4 groups x 75 iterations of VALU work spread over 3 waves: 100 iterations each
    for (int it = 0; it < 100; ++it) {
      uint32_t v = it * 64 + lane;
#pragma unroll
      for (int j = 0; j < VALU; ++j) v = (v << 3) ^ (v >> 5) ^ 0x9E3779B1u;
      acc ^= v;
    }
  }
  if (acc == 0x12345678u) sink[threadIdx.x] = acc;  // keep the VALU work alive
}
// Role-specialised pipeline prototype: 256-thread workgroup = 3 worker waves +
// 1 streamer wave, padded records (stride 2736 B).  Each worker owns a group of
// 64 envs and fills an LDS slot with the records of CH envs per step (VALU work
// + 9 ds_write per lane task); the streamer drains the three slots of the
// previous step to HBM with aligned 16-byte stores.  Lockstep: one barrier per
// step, slots double-buffered.
template <int CH, int VALU>
__global__ __launch_bounds__(256) void fill_pipeline(uint8_t* planes, int groups) {
  extern __shared__ uint32_t sh[];
  constexpr int REC = 2736 / 4, SLOT = CH * REC;            // words
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nsteps = (64 + CH - 1) / CH;
  for (int g0 = blockIdx.x * 3; g0 < groups; g0 += gridDim.x * 3) {
    for (int st = 0; st <= nsteps; ++st) {
      if (wave < 3 && st < nsteps && g0 + wave < groups) {
        uint32_t* slot = sh + (wave * 2 + (st & 1)) * SLOT;
        const int envs = (64 - st * CH) < CH ? (64 - st * CH) : CH;
        for (int t = lane; t < envs * 76; t += 64) {
          uint32_t el = t / 76, q = t - el * 76;
          uint32_t v = t + st;
#pragma unroll
          for (int j = 0; j < VALU; ++j) v = (v << 3) ^ (v >> 5) ^ 0x9E3779B1u;
          uint32_t* rec = slot + el * REC + q;
#pragma unroll
          for (int p = 0; p < 9; ++p) rec[p * 76] = v + p;
        }
      }
      if (wave == 3 && st > 0) {
        for (int w = 0; w < 3; ++w) {
          if (g0 + w >= groups) break;
          const int pst = st - 1;
          const int envs = (64 - pst * CH) < CH ? (64 - pst * CH) : CH;
          const uint4* src = reinterpret_cast<const uint4*>(sh + (w * 2 + (pst & 1)) * SLOT);
          uint4* dst = reinterpret_cast<uint4*>(planes + ((size_t)(g0 + w) * 64 + pst * CH) * 2736);
          for (int i = lane; i < envs * 171; i += 64) dst[i] = src[i];
        }
      }
      __syncthreads();
    }
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
  const size_t envs = 1 << 20, bytes = envs * 2736;
  uint8_t *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"fill dwordx4", "fill dword", "step-kernel store pattern (dword)", "block-flat dwordx4", "copy dwordx4 (r+w)",
                         "store pattern, 4 KB LDS/wave", "store pattern, 8 KB LDS/wave", "store pattern, 10.5 KB LDS/wave", "store pattern, 16 KB LDS/wave",
                         "store pattern, 4 KB LDS unused", "store pattern, 256-thr blocks + LDS", "plane-chunk dwordx4 + LDS", "store pattern + 64 VALU/iter",
                         "dword pattern + LDS + 45 full-rate VALU/iter (75 iters)", "padded aligned x4, LDS, no VALU (19 iters)",
                         "padded aligned x4 + LDS + 150 VALU/iter (19 iters)",
                         "256-thr WG, every wave: stores + 45 VALU/iter", "256-thr WG, wave 0 all stores, waves 1-3 all VALU",
                         "256-thr WG, every wave: stores + 90 VALU/iter", "256-thr WG, segregated, 90 VALU/iter",
                         "pipeline 3 workers + 1 streamer, CH=3, 45 VALU, 2 WG/CU", "pipeline CH=4, 45 VALU, 1 WG/CU",
                         "pipeline CH=3, 0 VALU", "pipeline CH=2, 45 VALU, 3 WG/CU"};
  const int lds_bytes[] = {0, 0, 0, 0, 0, 4096, 8192, 10752, 16384, 4096, 4096, 4096, 0, 20480, 20480, 20480, 0, 0, 0, 0, 6 * 3 * 2736, 6 * 4 * 2736, 6 * 3 * 2736, 6 * 2 * 2736};
  uint32_t* sink; CK(hipMalloc(&sink, 4096));
  for (int mode = 0; mode < 24; ++mode) {
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
        case 13: fill_pattern_work<45><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 14: fill_padded_x4<0><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 15: fill_padded_x4<150><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 16: fill_roles<false, 45><<<envs / 256, 256>>>(a, sink); break;
        case 17: fill_roles<true, 45><<<envs / 256, 256>>>(a, sink); break;
        case 18: fill_roles<false, 90><<<envs / 256, 256>>>(a, sink); break;
        case 19: fill_roles<true, 90><<<envs / 256, 256>>>(a, sink); break;
        case 20: fill_pipeline<3, 45><<<512, 256, lds_bytes[mode]>>>(a, (int)(envs / 64)); break;
        case 21: fill_pipeline<4, 45><<<256, 256, lds_bytes[mode]>>>(a, (int)(envs / 64)); break;
        case 22: fill_pipeline<3, 0><<<512, 256, lds_bytes[mode]>>>(a, (int)(envs / 64)); break;
        case 23: fill_pipeline<2, 45><<<768, 256, lds_bytes[mode]>>>(a, (int)(envs / 64)); break;
        default: fill_pattern_lds<<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2 && ms < best) best = ms;
    }
    double moved = mode == 4 ? 2.0 * bytes : (mode == 14 || mode == 15 || mode >= 20) ? (double)envs * 2736 : mode >= 2 && mode != 4 ? (double)envs * 2700 : (double)bytes;
    printf("%-36s %8.3f ms  %8.1f GB/s\n", names[mode], best, moved / best / 1e6);
  }
  return 0;
}
