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
// Mixed roles, but each wave bounds its own outstanding stores with
// s_waitcnt vmcnt(K) before every iteration's burst: turns issue-stage stalls
// (which block the SIMD) into per-wave waits (which do not).
template <int K, int VALU, int LB>
__global__ __launch_bounds__(LB) void fill_throttle(uint8_t* planes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* blk = planes + ((size_t)blockIdx.x * (LB / 64) + wave) * 64 * 2700;
  constexpr int ENC = (K & 15) | 0x0F70 | ((K >> 4) << 14);
  for (int it = 0; it < 75; ++it) {
    uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
    uint32_t v = f;
#pragma unroll
    for (int j = 0; j < VALU; ++j) v = (v << 3) ^ (v >> 5) ^ 0x9E3779B1u;
    uint8_t* dst = blk + e * 2700 + q * 4;
    __builtin_amdgcn_s_waitcnt(ENC);
#pragma unroll
    for (int p = 0; p < 9; ++p) *reinterpret_cast<uint32_t*>(dst + p * 300) = v + p;
  }
}
// VALU sweep: same loop, V dependent ops per iteration; INDEP=1 keeps the
// stored data independent of the chain (result folded into a sink instead).
template <int VALU, int INDEP>
__global__ __launch_bounds__(64) void fill_sweep(uint8_t* planes, uint32_t* sink) {
  const int lane = threadIdx.x & 63;
  uint8_t* blk = planes + (size_t)blockIdx.x * 64 * 2700;
  uint32_t acc = 0;
  for (int it = 0; it < 75; ++it) {
    uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
    uint32_t v = f;
#pragma unroll
    for (int j = 0; j < VALU; ++j) v = (v << 3) ^ (v >> 5) ^ 0x9E3779B1u;
    uint8_t* dst = blk + e * 2700 + q * 4;
    if (INDEP) { acc ^= v; v = f; }
#pragma unroll
    for (int p = 0; p < 9; ++p) *reinterpret_cast<uint32_t*>(dst + p * 300) = v + p;
  }
  if (INDEP && acc == 0x12345678u) sink[lane] = acc;
}
// Data-content sweep: identical instruction stream (the hash is always
// computed); only the stored VALUES differ.
__device__ inline uint32_t mb_hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
template <int DATA>
__global__ __launch_bounds__(64) void fill_data(uint8_t* planes, uint32_t* sink) {
  const int lane = threadIdx.x & 63;
  uint8_t* blk = planes + (size_t)blockIdx.x * 64 * 2700;
  uint32_t acc = 0;
  for (int it = 0; it < 75; ++it) {
    uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
    uint8_t* dst = blk + e * 2700 + q * 4;
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      uint32_t h = mb_hash((blockIdx.x * 4800 + f) * 9 + p);
      uint32_t v;
      switch (DATA) {
        case 0: v = 0; break;
        case 1: v = f + p; break;
        case 2: v = h; break;
        case 3: v = h & 0x01010101u; break;
        case 4: v = p == 0 ? (0x20202020u | (h & 0x43434343u)) : (h & (h >> 1) & (h >> 2) & 0x01010101u); break;
        case 5: v = 0xFFFFFFFFu; break;
        case 6: v = (lane & 1) ? 0xFFFFFFFFu : 0u; break;
        case 7: v = p == 0 ? 0x23232323u : (p == 1 ? 0x01010101u : 0u); break;
        default: v = h & (h >> 1) & (h >> 2) & (h >> 3) & (h >> 4) & 0x01010101u; break;
      }
      acc ^= h;
      *reinterpret_cast<uint32_t*>(dst + p * 300) = v;
    }
  }
  if (acc == 0x12345678u) sink[lane] = acc;
}
// Same as fill_sweep but the nine stores leave as one asm burst from nine
// distinct data registers (no VALU between them, no register reuse).
template <int VALU>
__global__ __launch_bounds__(64) void fill_burst(uint8_t* planes) {
  const int lane = threadIdx.x & 63;
  uint8_t* blk = planes + (size_t)blockIdx.x * 64 * 2700;
#pragma unroll 1
  for (int it = 0; it < 75; ++it) {
    uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
    uint32_t v = f;
#pragma unroll
    for (int j = 0; j < VALU; ++j) v = (v << 3) ^ (v >> 5) ^ 0x9E3779B1u;
    uint8_t* dst = blk + e * 2700 + q * 4;
    uint32_t d[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) d[p] = v + p;
    asm volatile(
        "global_store_dword %0, %1, off\n\t"
        "global_store_dword %0, %2, off offset:300\n\t"
        "global_store_dword %0, %3, off offset:600\n\t"
        "global_store_dword %0, %4, off offset:900\n\t"
        "global_store_dword %0, %5, off offset:1200\n\t"
        "global_store_dword %0, %6, off offset:1500\n\t"
        "global_store_dword %0, %7, off offset:1800\n\t"
        "global_store_dword %0, %8, off offset:2100\n\t"
        "global_store_dword %0, %9, off offset:2400"
        : : "v"(dst), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(d[8]) : "memory");
  }
}
// fill_burst + per-wave throttle: s_waitcnt vmcnt(K) before each burst; LDS
// bytes (dynamic) bound the occupancy.
template <int VALU, int K>
__global__ __launch_bounds__(64) void fill_burst_thr(uint8_t* planes) {
  extern __shared__ uint32_t sh[];
  const int lane = threadIdx.x & 63;
  uint8_t* blk = planes + (size_t)blockIdx.x * 64 * 2700;
  constexpr int ENC = (K & 15) | 0x0F70 | ((K >> 4) << 14);
  sh[lane] = lane;
#pragma unroll 1
  for (int it = 0; it < 75; ++it) {
    uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
    uint32_t v = f + sh[(lane + it) & 63];
#pragma unroll
    for (int j = 0; j < VALU; ++j) v = (v << 3) ^ (v >> 5) ^ 0x9E3779B1u;
    uint8_t* dst = blk + e * 2700 + q * 4;
    uint32_t d[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) d[p] = v + p;
    __builtin_amdgcn_s_waitcnt(ENC);
    asm volatile(
        "global_store_dword %0, %1, off\n\t"
        "global_store_dword %0, %2, off offset:300\n\t"
        "global_store_dword %0, %3, off offset:600\n\t"
        "global_store_dword %0, %4, off offset:900\n\t"
        "global_store_dword %0, %5, off offset:1200\n\t"
        "global_store_dword %0, %6, off offset:1500\n\t"
        "global_store_dword %0, %7, off offset:1800\n\t"
        "global_store_dword %0, %8, off offset:2100\n\t"
        "global_store_dword %0, %9, off offset:2400"
        : : "v"(dst), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(d[8]) : "memory");
  }
}
// Calibration: the fill_burst loop with (MODE 0) stores predicated off, i.e.
// VALU only, or (MODE 1) every workgroup writing one of 16 blocks, i.e. the
// same store instructions absorbed by L2 instead of HBM.
template <int VALU, int MODE>
__global__ __launch_bounds__(64) void fill_calib(uint8_t* planes) {
  const int lane = threadIdx.x & 63;
  uint8_t* blk = planes + (size_t)(MODE == 1 ? (blockIdx.x & 15) : blockIdx.x) * 64 * 2700;
#pragma unroll 1
  for (int it = 0; it < 75; ++it) {
    uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
    uint32_t v = f;
#pragma unroll
    for (int j = 0; j < VALU; ++j) v = (v << 3) ^ (v >> 5) ^ 0x9E3779B1u;
    uint8_t* dst = blk + e * 2700 + q * 4;
    uint32_t d[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) d[p] = v + p;
    if (MODE == 0 && v != 0x12345678u) continue;
    asm volatile(
        "global_store_dword %0, %1, off\n\t"
        "global_store_dword %0, %2, off offset:300\n\t"
        "global_store_dword %0, %3, off offset:600\n\t"
        "global_store_dword %0, %4, off offset:900\n\t"
        "global_store_dword %0, %5, off offset:1200\n\t"
        "global_store_dword %0, %6, off offset:1500\n\t"
        "global_store_dword %0, %7, off offset:1800\n\t"
        "global_store_dword %0, %8, off offset:2100\n\t"
        "global_store_dword %0, %9, off offset:2400"
        : : "v"(dst), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(d[8]) : "memory");
  }
}
// Ring pipeline: W worker waves + 1 streamer wave per workgroup, no barriers.
// A worker owns one group of 64 envs: per iteration VALU ops, then its nine
// finished dwords go into slot it%R of its LDS ring and it publishes it+1 in
// ready[w].  The streamer polls the rings round-robin, moves a slot to HBM with
// the usual nine dword stores and publishes done[w].  Spins are bounded.
template <int W, int R, int VALU>
__global__ __launch_bounds__((W + 1) * 64) void fill_ring(uint8_t* planes, int groups, uint32_t* err) {
  extern __shared__ uint32_t sh[];
  uint32_t* ready = sh;            // [W]
  uint32_t* done = sh + W;         // [W]
  uint32_t* ring = sh + 64;        // [W][R][9][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 2 * W) sh[threadIdx.x] = 0;
  __syncthreads();
  const int g0 = blockIdx.x * W;
  if (wave < W) {
    if (g0 + wave >= groups) return;
    uint32_t* my = ring + wave * R * 576;
#pragma unroll 1
    for (int it = 0; it < 75; ++it) {
      uint32_t v = it * 64 + lane;
#pragma unroll
      for (int j = 0; j < VALU; ++j) v = (v << 3) ^ (v >> 5) ^ 0x9E3779B1u;
      int spins = 0;
      while ((int)(it - __hip_atomic_load(&done[wave], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >= R) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 18)) { if (lane == 0) err[0] = 1; return; }
      }
      uint32_t* slot = my + (it % R) * 576 + lane;
#pragma unroll
      for (int p = 0; p < 9; ++p) slot[p * 64] = v + p;
      __hip_atomic_store(&ready[wave], (uint32_t)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  } else {
    const int nw = (groups - g0) < W ? (groups - g0) : W;
#pragma unroll 1
    for (int it = 0; it < 75; ++it) {
      const uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
      const uint32_t off = e * 2700 + q * 4;
      // wait until every worker has published this iteration (one poll covers all)
      int spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int w = 0; w < W; ++w)
          ok = ok && (w >= nw || (int)__hip_atomic_load(&ready[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) > it);
        if (ok) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 18)) { if (lane == 0) err[0] = 2; return; }
      }
      uint32_t d[W][9];
#pragma unroll
      for (int w = 0; w < W; ++w) {
        const uint32_t* slot = ring + w * R * 576 + (it % R) * 576 + lane;
#pragma unroll
        for (int p = 0; p < 9; ++p) d[w][p] = slot[p * 64];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int w = 0; w < W; ++w)
        __hip_atomic_store(&done[w], (uint32_t)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int w = 0; w < W; ++w) {
        if (w >= nw) break;
        uint8_t* dst = planes + (size_t)(g0 + w) * 64 * 2700 + off;
        asm volatile(
            "global_store_dword %0, %1, off\n\t"
            "global_store_dword %0, %2, off offset:300\n\t"
            "global_store_dword %0, %3, off offset:600\n\t"
            "global_store_dword %0, %4, off offset:900\n\t"
            "global_store_dword %0, %5, off offset:1200\n\t"
            "global_store_dword %0, %6, off offset:1500\n\t"
            "global_store_dword %0, %7, off offset:1800\n\t"
            "global_store_dword %0, %8, off offset:2100\n\t"
            "global_store_dword %0, %9, off offset:2400"
            : : "v"(dst), "v"(d[w][0]), "v"(d[w][1]), "v"(d[w][2]), "v"(d[w][3]), "v"(d[w][4]), "v"(d[w][5]), "v"(d[w][6]), "v"(d[w][7]), "v"(d[w][8]) : "memory");
      }
    }
  }
}
// fill_burst with the loop unrolled U times over DISTINCT register sets: all
// VALU work of U iterations first, then U bursts, so a data register is not
// rewritten until U-1 later bursts have been issued.
template <int VALU, int U>
__global__ __launch_bounds__(64) void fill_burst_unroll(uint8_t* planes) {
  const int lane = threadIdx.x & 63;
  uint8_t* blk = planes + (size_t)blockIdx.x * 64 * 2700;
#pragma unroll 1
  for (int it0 = 0; it0 < 75; it0 += U) {
    uint32_t d[U][9];
    uint8_t* dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = it0 + u;
      uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
      uint32_t v = f;
#pragma unroll
      for (int j = 0; j < VALU; ++j) v = (v << 3) ^ (v >> 5) ^ 0x9E3779B1u;
      dst[u] = blk + e * 2700 + q * 4;
#pragma unroll
      for (int p = 0; p < 9; ++p) d[u][p] = v + p;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (it0 + u >= 75) break;
      asm volatile(
          "global_store_dword %0, %1, off\n\t"
          "global_store_dword %0, %2, off offset:300\n\t"
          "global_store_dword %0, %3, off offset:600\n\t"
          "global_store_dword %0, %4, off offset:900\n\t"
          "global_store_dword %0, %5, off offset:1200\n\t"
          "global_store_dword %0, %6, off offset:1500\n\t"
          "global_store_dword %0, %7, off offset:1800\n\t"
          "global_store_dword %0, %8, off offset:2100\n\t"
          "global_store_dword %0, %9, off offset:2400"
          : : "v"(dst[u]), "v"(d[u][0]), "v"(d[u][1]), "v"(d[u][2]), "v"(d[u][3]), "v"(d[u][4]), "v"(d[u][5]), "v"(d[u][6]), "v"(d[u][7]), "v"(d[u][8]) : "memory");
    }
  }
}
// Rotating register sets without batching: iteration it computes into set it%U
// and stores it immediately (U-way unrolled loop body).
template <int VALU, int U>
__global__ __launch_bounds__(64) void fill_burst_rotate(uint8_t* planes) {
  const int lane = threadIdx.x & 63;
  uint8_t* blk = planes + (size_t)blockIdx.x * 64 * 2700;
#pragma unroll 1
  for (int it0 = 0; it0 < 75; it0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = it0 + u;
      if (it >= 75) break;
      uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;
      uint32_t v = f;
#pragma unroll
      for (int j = 0; j < VALU; ++j) v = (v << 3) ^ (v >> 5) ^ 0x9E3779B1u;
      uint8_t* dst = blk + e * 2700 + q * 4;
      uint32_t d[9];
#pragma unroll
      for (int p = 0; p < 9; ++p) d[p] = v + p;
      asm volatile(
          "global_store_dword %0, %1, off\n\t"
          "global_store_dword %0, %2, off offset:300\n\t"
          "global_store_dword %0, %3, off offset:600\n\t"
          "global_store_dword %0, %4, off offset:900\n\t"
          "global_store_dword %0, %5, off offset:1200\n\t"
          "global_store_dword %0, %6, off offset:1500\n\t"
          "global_store_dword %0, %7, off offset:1800\n\t"
          "global_store_dword %0, %8, off offset:2100\n\t"
          "global_store_dword %0, %9, off offset:2400"
          : : "v"(dst), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(d[8]) : "memory");
    }
  }
}
// Store pattern with cache-policy bits on the stores (gfx94x/gfx950: sc0, sc1, nt).
#define MB_POLICY_KERNEL(NAME, BITS)                                                              \
  template <int VALU>                                                                             \
  __global__ __launch_bounds__(64) void NAME(uint8_t* planes) {                                   \
    const int lane = threadIdx.x & 63;                                                            \
    uint8_t* blk = planes + (size_t)blockIdx.x * 64 * 2700;                                       \
    _Pragma("unroll 1") for (int it = 0; it < 75; ++it) {                                         \
      uint32_t f = it * 64 + lane, e = f / 75, q = f - e * 75;                                    \
      uint32_t v = f;                                                                             \
      _Pragma("unroll") for (int j = 0; j < VALU; ++j) v = (v << 3) ^ (v >> 5) ^ 0x9E3779B1u;     \
      uint8_t* dst = blk + e * 2700 + q * 4;                                                      \
      uint32_t d[9];                                                                              \
      _Pragma("unroll") for (int p = 0; p < 9; ++p) d[p] = v + p;                                 \
      asm volatile("global_store_dword %0, %1, off " BITS "\n\t"                                   \
                   "global_store_dword %0, %2, off offset:300 " BITS "\n\t"                        \
                   "global_store_dword %0, %3, off offset:600 " BITS "\n\t"                        \
                   "global_store_dword %0, %4, off offset:900 " BITS "\n\t"                        \
                   "global_store_dword %0, %5, off offset:1200 " BITS "\n\t"                       \
                   "global_store_dword %0, %6, off offset:1500 " BITS "\n\t"                       \
                   "global_store_dword %0, %7, off offset:1800 " BITS "\n\t"                       \
                   "global_store_dword %0, %8, off offset:2100 " BITS "\n\t"                       \
                   "global_store_dword %0, %9, off offset:2400 " BITS                              \
                   : : "v"(dst), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), \
                       "v"(d[6]), "v"(d[7]), "v"(d[8]) : "memory");                                \
    }                                                                                             \
  }
MB_POLICY_KERNEL(fill_pol_nt, "nt")
MB_POLICY_KERNEL(fill_pol_sc1, "sc1")
MB_POLICY_KERNEL(fill_pol_sc0sc1, "sc0 sc1")
MB_POLICY_KERNEL(fill_pol_sc0, "sc0")
MB_POLICY_KERNEL(fill_pol_ntsc1, "sc1 nt")
MB_POLICY_KERNEL(fill_pol_all, "sc0 sc1 nt")

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
                         "pipeline CH=3, 0 VALU", "pipeline CH=2, 45 VALU, 3 WG/CU",
                         "throttle vmcnt(9), 45 VALU, 64-thr", "throttle vmcnt(18)", "throttle vmcnt(27)", "throttle vmcnt(36)", "throttle vmcnt(54)", "throttle vmcnt(63) (none)",
                         "throttle vmcnt(18), 256-thr", "throttle vmcnt(36), 256-thr", "throttle vmcnt(9), 256-thr", "throttle vmcnt(4), 256-thr",
                         "sweep 0 VALU", "sweep 4", "sweep 8", "sweep 16", "sweep 24", "sweep 32", "sweep 45", "sweep 90", "sweep 180", "sweep 360", "sweep 45 indep", "sweep 90 indep", "sweep 180 indep",
                         "data: zeros", "data: f+p", "data: random dwords", "data: random 0/1 bytes", "data: ascii board + 12% 0/1 layers", "data: all ones", "data: lanes alternate 0/~0", "data: constant wall board + full layer + empties", "data: 3% 0/1 bytes",
 "burst 0", "burst 4", "burst 16", "burst 45", "burst 90",
 "burst45 vmcnt(63) lds 0", "burst45 vmcnt(0) lds 0", "burst45 vmcnt(9) lds 0", "burst45 vmcnt(18) lds 0", "burst45 vmcnt(36) lds 0", "burst45 vmcnt(63) lds 20480", "burst45 vmcnt(0) lds 20480", "burst45 vmcnt(9) lds 20480", "burst45 vmcnt(18) lds 20480", "burst45 vmcnt(36) lds 20480", "burst45 vmcnt(63) lds 10240", "burst45 vmcnt(9) lds 10240", "burst45 vmcnt(18) lds 10240", "burst45 vmcnt(36) lds 10240",
 "calib 16 ops, VALU only", "calib 45 ops, VALU only", "calib 90 ops, VALU only", "calib 0 ops, stores to L2", "calib 16 ops, stores to L2", "calib 45 ops, stores to L2", "calib 90 ops, stores to L2",
 "ring W=3 R=2, 20 ops", "ring W=3 R=4, 20 ops", "ring W=3 R=2, 45 ops", "ring W=3 R=4, 45 ops", "ring W=3 R=4, 0 ops", "ring W=4 R=2, 20 ops", "ring W=2 R=4, 20 ops", "ring W=7 R=2, 20 ops", "ring W=3 R=8, 20 ops",
 "ring W=3 R=4, 20 ops, lds 56000", "ring W=3 R=8, 45 ops, lds 55552", "ring W=3 R=8, 0 ops, lds 55552", "ring W=3 R=16, 20 ops, lds 110848", "ring W=4 R=8, 20 ops, lds 73984", "ring W=7 R=4, 20 ops, lds 64768", "ring W=7 R=8, 20 ops, lds 129280", "ring W=3 R=8, 20 ops, lds 80000", "ring W=5 R=8, 20 ops, lds 92416", "ring W=3 R=12, 20 ops, lds 83200",
 "burst unroll V=16 U=1", "burst unroll V=16 U=2", "burst unroll V=16 U=3", "burst unroll V=16 U=5", "burst unroll V=45 U=3", "burst unroll V=45 U=5", "burst rotate V=16 U=3", "burst rotate V=16 U=5", "burst rotate V=45 U=5",
 "stores nt, 0 ops", "stores nt, 16 ops", "stores sc1, 0 ops", "stores sc1, 16 ops", "stores sc0 sc1, 0 ops", "stores sc0 sc1, 16 ops", "stores sc0, 0 ops", "stores sc0, 16 ops", "stores sc1 nt, 0 ops", "stores sc1 nt, 16 ops", "stores sc0 sc1 nt, 0 ops", "stores sc0 sc1 nt, 16 ops"};
  const int lds_bytes[] = {0, 0, 0, 0, 0, 4096, 8192, 10752, 16384, 4096, 4096, 4096, 0, 20480, 20480, 20480, 0, 0, 0, 0, 6 * 3 * 2736, 6 * 4 * 2736, 6 * 3 * 2736, 6 * 2 * 2736, 0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0, 0,0,0,0,0, 0,0,0,0,0,20480,20480,20480,20480,20480,10240,10240,10240,10240, 0,0,0,0,0,0,0, 14080,27904,14080,27904,27904,18688,18688,32512,55552, 56000,55552,55552,110848,73984,64768,129280,80000,92416,83200, 0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,0,0};
  uint32_t* sink; CK(hipMalloc(&sink, 4096)); CK(hipMemset(sink, 0, 4096));
  const int first = getenv("MB_FIRST") ? atoi(getenv("MB_FIRST")) : 0;
  for (int mode = first; mode < 122; ++mode) {
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
        case 24: fill_throttle<9, 45, 64><<<envs / 64, 64>>>(a); break;
        case 25: fill_throttle<18, 45, 64><<<envs / 64, 64>>>(a); break;
        case 26: fill_throttle<27, 45, 64><<<envs / 64, 64>>>(a); break;
        case 27: fill_throttle<36, 45, 64><<<envs / 64, 64>>>(a); break;
        case 28: fill_throttle<54, 45, 64><<<envs / 64, 64>>>(a); break;
        case 29: fill_throttle<63, 45, 64><<<envs / 64, 64>>>(a); break;
        case 30: fill_throttle<18, 45, 256><<<envs / 256, 256>>>(a); break;
        case 31: fill_throttle<36, 45, 256><<<envs / 256, 256>>>(a); break;
        case 32: fill_throttle<9, 45, 256><<<envs / 256, 256>>>(a); break;
        case 33: fill_throttle<4, 45, 256><<<envs / 256, 256>>>(a); break;
        case 34: fill_sweep<0, 0><<<envs / 64, 64>>>(a, sink); break;
        case 35: fill_sweep<4, 0><<<envs / 64, 64>>>(a, sink); break;
        case 36: fill_sweep<8, 0><<<envs / 64, 64>>>(a, sink); break;
        case 37: fill_sweep<16, 0><<<envs / 64, 64>>>(a, sink); break;
        case 38: fill_sweep<24, 0><<<envs / 64, 64>>>(a, sink); break;
        case 39: fill_sweep<32, 0><<<envs / 64, 64>>>(a, sink); break;
        case 40: fill_sweep<45, 0><<<envs / 64, 64>>>(a, sink); break;
        case 41: fill_sweep<90, 0><<<envs / 64, 64>>>(a, sink); break;
        case 42: fill_sweep<180, 0><<<envs / 64, 64>>>(a, sink); break;
        case 43: fill_sweep<360, 0><<<envs / 64, 64>>>(a, sink); break;
        case 44: fill_sweep<45, 1><<<envs / 64, 64>>>(a, sink); break;
        case 45: fill_sweep<90, 1><<<envs / 64, 64>>>(a, sink); break;
        case 46: fill_sweep<180, 1><<<envs / 64, 64>>>(a, sink); break;
        case 47: fill_data<0><<<envs / 64, 64>>>(a, sink); break;
        case 48: fill_data<1><<<envs / 64, 64>>>(a, sink); break;
        case 49: fill_data<2><<<envs / 64, 64>>>(a, sink); break;
        case 50: fill_data<3><<<envs / 64, 64>>>(a, sink); break;
        case 51: fill_data<4><<<envs / 64, 64>>>(a, sink); break;
        case 52: fill_data<5><<<envs / 64, 64>>>(a, sink); break;
        case 53: fill_data<6><<<envs / 64, 64>>>(a, sink); break;
        case 54: fill_data<7><<<envs / 64, 64>>>(a, sink); break;
        case 55: fill_data<8><<<envs / 64, 64>>>(a, sink); break;
        case 56: fill_burst<0><<<envs / 64, 64>>>(a); break;
        case 57: fill_burst<4><<<envs / 64, 64>>>(a); break;
        case 58: fill_burst<16><<<envs / 64, 64>>>(a); break;
        case 59: fill_burst<45><<<envs / 64, 64>>>(a); break;
        case 60: fill_burst<90><<<envs / 64, 64>>>(a); break;
        case 61: fill_burst_thr<45, 63><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 62: fill_burst_thr<45, 0><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 63: fill_burst_thr<45, 9><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 64: fill_burst_thr<45, 18><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 65: fill_burst_thr<45, 36><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 66: fill_burst_thr<45, 63><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 67: fill_burst_thr<45, 0><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 68: fill_burst_thr<45, 9><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 69: fill_burst_thr<45, 18><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 70: fill_burst_thr<45, 36><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 71: fill_burst_thr<45, 63><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 72: fill_burst_thr<45, 9><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 73: fill_burst_thr<45, 18><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 74: fill_burst_thr<45, 36><<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
        case 75: fill_calib<16, 0><<<envs / 64, 64>>>(a); break;
        case 76: fill_calib<45, 0><<<envs / 64, 64>>>(a); break;
        case 77: fill_calib<90, 0><<<envs / 64, 64>>>(a); break;
        case 78: fill_calib<0, 1><<<envs / 64, 64>>>(a); break;
        case 79: fill_calib<16, 1><<<envs / 64, 64>>>(a); break;
        case 80: fill_calib<45, 1><<<envs / 64, 64>>>(a); break;
        case 81: fill_calib<90, 1><<<envs / 64, 64>>>(a); break;
        case 82: fill_ring<3, 2, 20><<<(envs / 64 + 2) / 3, 256, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 83: fill_ring<3, 4, 20><<<(envs / 64 + 2) / 3, 256, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 84: fill_ring<3, 2, 45><<<(envs / 64 + 2) / 3, 256, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 85: fill_ring<3, 4, 45><<<(envs / 64 + 2) / 3, 256, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 86: fill_ring<3, 4, 0><<<(envs / 64 + 2) / 3, 256, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 87: fill_ring<4, 2, 20><<<(envs / 64 + 3) / 4, 320, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 88: fill_ring<2, 4, 20><<<(envs / 64 + 1) / 2, 192, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 89: fill_ring<7, 2, 20><<<(envs / 64 + 6) / 7, 512, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 90: fill_ring<3, 8, 20><<<(envs / 64 + 2) / 3, 256, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 91: fill_ring<3, 4, 20><<<(envs / 64 + 2) / 3, 256, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 92: fill_ring<3, 8, 45><<<(envs / 64 + 2) / 3, 256, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 93: fill_ring<3, 8, 0><<<(envs / 64 + 2) / 3, 256, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 94: fill_ring<3, 16, 20><<<(envs / 64 + 2) / 3, 256, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 95: fill_ring<4, 8, 20><<<(envs / 64 + 3) / 4, 320, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 96: fill_ring<7, 4, 20><<<(envs / 64 + 6) / 7, 512, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 97: fill_ring<7, 8, 20><<<(envs / 64 + 6) / 7, 512, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 98: fill_ring<3, 8, 20><<<(envs / 64 + 2) / 3, 256, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 99: fill_ring<5, 8, 20><<<(envs / 64 + 4) / 5, 384, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 100: fill_ring<3, 12, 20><<<(envs / 64 + 2) / 3, 256, lds_bytes[mode]>>>(a, (int)(envs / 64), sink); break;
        case 101: fill_burst_unroll<16, 1><<<envs / 64, 64>>>(a); break;
        case 102: fill_burst_unroll<16, 2><<<envs / 64, 64>>>(a); break;
        case 103: fill_burst_unroll<16, 3><<<envs / 64, 64>>>(a); break;
        case 104: fill_burst_unroll<16, 5><<<envs / 64, 64>>>(a); break;
        case 105: fill_burst_unroll<45, 3><<<envs / 64, 64>>>(a); break;
        case 106: fill_burst_unroll<45, 5><<<envs / 64, 64>>>(a); break;
        case 107: fill_burst_rotate<16, 3><<<envs / 64, 64>>>(a); break;
        case 108: fill_burst_rotate<16, 5><<<envs / 64, 64>>>(a); break;
        case 109: fill_burst_rotate<45, 5><<<envs / 64, 64>>>(a); break;
        case 110: fill_pol_nt<0><<<envs / 64, 64>>>(a); break;
        case 111: fill_pol_nt<16><<<envs / 64, 64>>>(a); break;
        case 112: fill_pol_sc1<0><<<envs / 64, 64>>>(a); break;
        case 113: fill_pol_sc1<16><<<envs / 64, 64>>>(a); break;
        case 114: fill_pol_sc0sc1<0><<<envs / 64, 64>>>(a); break;
        case 115: fill_pol_sc0sc1<16><<<envs / 64, 64>>>(a); break;
        case 116: fill_pol_sc0<0><<<envs / 64, 64>>>(a); break;
        case 117: fill_pol_sc0<16><<<envs / 64, 64>>>(a); break;
        case 118: fill_pol_ntsc1<0><<<envs / 64, 64>>>(a); break;
        case 119: fill_pol_ntsc1<16><<<envs / 64, 64>>>(a); break;
        case 120: fill_pol_all<0><<<envs / 64, 64>>>(a); break;
        case 121: fill_pol_all<16><<<envs / 64, 64>>>(a); break;
        default: fill_pattern_lds<<<envs / 64, 64, lds_bytes[mode]>>>(a); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2 && ms < best) best = ms;
    }
    double moved = mode == 4 ? 2.0 * bytes : (mode == 14 || mode == 15 || mode >= 20) ? (double)envs * 2736 : mode >= 2 && mode != 4 ? (double)envs * 2700 : (double)bytes;
    { uint32_t herr = 0; CK(hipMemcpy(&herr, sink, 4, hipMemcpyDeviceToHost)); if (herr) { printf("  [spin timeout %u]\n", herr); CK(hipMemset(sink, 0, 4)); } }
    printf("%-36s %8.3f ms  %8.1f GB/s\n", names[mode], best, moved / best / 1e6);
  }
  return 0;
}
