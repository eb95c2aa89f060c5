// How many streaming waves per CU does it take to saturate the write path, by store width?
// Persistent single-wave workgroups (W per CU) walk 172,800-byte chunks (one "unit" of the step kernel: 64
// environment records) of a 2.83 GB buffer, each chunk written sequentially by one wave:
//   mode 0: dword stores (256 B per instruction), mode 1: dwordx2 (512 B), mode 2: dwordx4 (1 KiB),
//   mode 3: the step kernel's own pattern (9 planes x 300 B per record, dword stores, 75 x 9 per chunk).
// hipcc --offload-arch=gfx950 -O3 store_width.hip -o store_width && ./store_width
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int CHUNK = 172800;  // bytes
// MODE 12: who pays for a load among the stores -- the wave or the CU?  Two-wave workgroups: wave 1 streams the chunks
// (no loads at all), wave 0 issues the 15 DMA row loads per chunk and nothing else, paced by s_sleep to about the same rate
__global__ __launch_bounds__(128) void fill_split(uint8_t* dst, int n_chunks, const uint32_t* state, int sleep_reps) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 64 * 77; i += 128) lds[i] = (uint32_t)(i * 2654435761u) & 0x07070707u;
  __syncthreads();
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    if (wave == 0) {
      const uint32_t* st = state + (size_t)c * 64 * 15;
      const uint32_t ib = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)(lds + 64 * 78);
#pragma unroll
      for (int w = 0; w < 15; ++w) {
        uint32_t keep; uint64_t own; const uint32_t* base = st + w * 64; const uint32_t vo = 4u * lane, la = ib + w * 256u;
        asm volatile("s_mov_b64 %1, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(own) : "v"(vo), "s"(base), "s"(la) : "memory");
      }
      for (int r = 0; r < sleep_reps; ++r) __builtin_amdgcn_s_sleep(64);
    } else {
      uint8_t* base = dst + (size_t)c * CHUNK;
      const uint64_t b64 = (uint64_t)base;
      const uint32_t lo32 = __builtin_amdgcn_readfirstlane((uint32_t)b64), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(b64 >> 32));
      uint8_t* sb = (uint8_t*)(((uint64_t)hi32 << 32) | lo32);
      asm volatile("s_mov_b64 %0, %0\n\ts_nop 4" : "+s"(sb));
      uint32_t q = lane, voff = 4u * lane, eF = 0;
      uint32_t code_pf = lds[q];
      const uint32_t chars_lo = 0x23402e20u, chars_hi = 0x63626150u;
#pragma unroll 1
      for (int i = 0; i < 75; ++i) {
        const uint32_t code = code_pf, vo = voff;
        q += 64; voff += 256u;
        const bool wrap = q >= 75u;
        q = wrap ? q - 75u : q;
        voff = wrap ? voff + 2400u : voff;
        eF = wrap ? eF + 77u : eF;
        code_pf = lds[i + 1 < 75 ? eF + q : 0u];
        uint32_t v = __builtin_amdgcn_perm(chars_hi, chars_lo, code);
        asm volatile("global_store_dword %0, %1, %2" : : "v"(vo), "v"(v), "s"(sb));
#pragma unroll
        for (int p = 1; p < 9; ++p) {
          uint8_t* pb = sb + 300 * p;
          v = __builtin_amdgcn_perm(p > 4 ? 1u << (8 * ((p - 1) & 3)) : 0u, p <= 4 ? 1u << (8 * ((p - 1) & 3)) : 0u, code);
          asm volatile("global_store_dword %0, %1, %2" : : "v"(vo), "v"(v), "s"(pb));
        }
      }
    }
  }
}
template <int MODE>
__global__ __launch_bounds__(64) void fill(uint8_t* dst, int n_chunks, uint32_t* ctr, int lds_pad) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 77; i += 64) lds[i] = (uint32_t)(i * 2654435761u) & 0x07070707u;
  __syncthreads();
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    uint8_t* base = dst + (size_t)c * CHUNK;
    const uint64_t b64 = (uint64_t)base;
    const uint32_t lo32 = __builtin_amdgcn_readfirstlane((uint32_t)b64), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(b64 >> 32));
    uint8_t* sb = (uint8_t*)(((uint64_t)hi32 << 32) | lo32);
    asm volatile("s_mov_b64 %0, %0\n\ts_nop 4" : "+s"(sb));  // (an SGPR pair fresh from v_readfirstlane must age 5 wait states before VMEM reads it)
    if (MODE == 0) {
      uint32_t voff = 4u * lane;
      for (int i = 0; i < CHUNK / 256; ++i, voff += 256u) {
        uint32_t v = voff ^ (uint32_t)c;
        asm volatile("global_store_dword %0, %1, %2" : : "v"(voff), "v"(v), "s"(sb));
      }
    } else if (MODE == 1) {
      uint32_t voff = 8u * lane;
      for (int i = 0; i < CHUNK / 512; ++i, voff += 512u) {
        uint64_t v = ((uint64_t)voff << 32) | (uint32_t)c;
        asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(voff), "v"(v), "s"(sb));
      }
      if (lane < (CHUNK % 512) / 8) { uint64_t v = c; asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(voff), "v"(v), "s"(sb)); }
    } else if (MODE == 2) {
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      uint32_t voff = 16u * lane;
      for (int i = 0; i < CHUNK / 1024; ++i, voff += 1024u) {
        u4 v = {voff, (uint32_t)c, voff + 1, voff + 2};
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(sb));
      }
      if (lane < (CHUNK % 1024) / 16) { u4 v = {1, 2, 3, 4}; asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(sb)); }
    } else if (MODE >= 4) {
      if (MODE >= 7) {
        // the same traffic with a unit's rows next to each other (unit-major state): MODE 7 loads + stores, 8 loads only, 9 stores only
        uint32_t* st = ctr + (size_t)c * 64 * 15 + lane;
        uint32_t acc = c;
        if (MODE != 9) {
#pragma unroll
          for (int w = 0; w < 15; ++w) acc += st[w * 64];
        }
        lds[64 * 77 + lane] = acc;
        if (MODE != 8) {
#pragma unroll
          for (int w = 0; w < 21; ++w) st[(w % 15) * 64] = acc + w;
        }
      }
      if (MODE == 12 || MODE == 13) {
        // the same 3,840 bytes as FOUR global_load_lds_dwordx4 (1 KiB each; the last one 768 B): is a row load's cost per
        // instruction or per byte?  MODE 13: one more dword DMA (the tape actions)
        const uint32_t* st = ctr + (size_t)c * 64 * 15;
        const uint32_t ib = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)(lds + 64 * 78);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          if (w == 3 && lane >= 48) break;
          uint32_t keep; uint64_t own; const uint32_t* base = st + w * 256; const uint32_t vo = 16u * lane, la = ib + w * 1024u;
          asm volatile("s_mov_b64 %1, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep), "=&s"(own) : "v"(vo), "s"(base), "s"(la) : "memory");
        }
        if (MODE == 13) {
          uint32_t keep; uint64_t own; const uint32_t* base = ctr + (size_t)((c * 7) & 16383) * 64; const uint32_t vo = 4u * lane, la = ib + 15 * 256u;
          asm volatile("s_mov_b64 %1, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, %1\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep), "=&s"(own) : "v"(vo), "s"(base), "s"(la) : "memory");
        }
      }
      if (MODE == 14) {
        const uint32_t* st = ctr + (size_t)c * 64 * 15;
        const uint32_t ib = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)(lds + 64 * 78);
#pragma unroll
        for (int w = 0; w < 15; ++w) {
          uint32_t keep; uint64_t own; const uint32_t* base = st + w * 64; const uint32_t vo = 4u * lane, la = ib + w * 256u;
          asm volatile("s_mov_b64 %1, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, %1 sc1\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep), "=&s"(own) : "v"(vo), "s"(base), "s"(la) : "memory");
        }
      }
      if (MODE == 15) {
        const uint32_t* st = ctr + (size_t)c * 64 * 15;
        const uint32_t ib = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)(lds + 64 * 78);
#pragma unroll
        for (int w = 0; w < 15; ++w) {
          uint32_t keep; uint64_t own; const uint32_t* base = st + w * 64; const uint32_t vo = 4u * lane, la = ib + w * 256u;
          asm volatile("s_mov_b64 %1, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, %1 sc0 sc1\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep), "=&s"(own) : "v"(vo), "s"(base), "s"(la) : "memory");
        }
      }
      if (MODE == 16) {
        const uint32_t* st = ctr + (size_t)c * 64 * 15;
        const uint32_t ib = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)(lds + 64 * 78);
#pragma unroll
        for (int w = 0; w < 15; ++w) {
          uint32_t keep; uint64_t own; const uint32_t* base = st + w * 64; const uint32_t vo = 4u * lane, la = ib + w * 256u;
          asm volatile("s_mov_b64 %1, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, %1 nt\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep), "=&s"(own) : "v"(vo), "s"(base), "s"(la) : "memory");
        }
      }
      if (MODE == 17) {
        const uint32_t* st = ctr + (size_t)c * 64 * 15;
        uint32_t acc = 0;
#pragma unroll
        for (int w = 0; w < 60; ++w) {
          typedef uint32_t u16 __attribute__((ext_vector_type(16)));
          u16 r;
          asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(st + w * 16) : "memory");
          acc ^= r[0] ^ r[15];
        }
        if (acc == 0x12345u) lds[64 * 78 + lane] = acc;
      }
      if (MODE == 18) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // one drain of the wave's own stores per chunk, no load
      if (MODE == 19) {  // one DMA row per chunk from ONE address (an L1 hit after the first)
        const uint32_t ib = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)(lds + 64 * 78);
        uint32_t keep; uint64_t own; const uint32_t* base = ctr; const uint32_t vo = 4u * lane, la = ib;
        asm volatile("s_mov_b64 %1, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(own) : "v"(vo), "s"(base), "s"(la) : "memory");
      }
      if (MODE >= 20) {
        constexpr int R = MODE - 20;
        const uint32_t* st = ctr + (size_t)c * 64 * 15;
        const uint32_t ib = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)(lds + 64 * 78);
#pragma unroll
        for (int w = 0; w < R; ++w) {
          uint32_t keep; uint64_t own; const uint32_t* base = st + (w % 15) * 64; const uint32_t vo = 4u * lane, la = ib + (w % 15) * 256u;
          asm volatile("s_mov_b64 %1, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, %1\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep), "=&s"(own) : "v"(vo), "s"(base), "s"(la) : "memory");
        }
      }
      if (MODE == 10 || MODE == 11) {
        // the 15 row loads as LDS-DMA, never waited for (the persistent shape's prefetch): does the READ TRAFFIC
        // itself cost the write stream anything?  MODE 11: the same 15 rows come out of a 61 KB table (L2 hits)
        const uint32_t* st = ctr + (MODE == 11 ? (size_t)(c & 15) * 64 * 15 : (size_t)c * 64 * 15);
        const uint32_t ib = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)(lds + 64 * 78);
#pragma unroll
        for (int w = 0; w < 15; ++w) {
          uint32_t keep; uint64_t own; const uint32_t* base = st + w * 64; const uint32_t vo = 4u * lane, la = ib + w * 256u;
          asm volatile("s_mov_b64 %1, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, %1\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep), "=&s"(own) : "v"(vo), "s"(base), "s"(la) : "memory");
        }
      }
      if (MODE == 6) {
        // the logic phase's memory traffic, nothing else: 15 coalesced 256-byte row loads of this unit's state words,
        // used (so that they are waited for), then 21 row stores
        uint32_t* st = ctr + (size_t)c * 64 + lane;
        const size_t pitch = (size_t)n_chunks * 64;
        uint32_t acc = 0;
#pragma unroll
        for (int w = 0; w < 15; ++w) acc += st[w * pitch];
        lds[64 * 77 + lane] = acc;
#pragma unroll
        for (int w = 0; w < 21; ++w) st[(w % 15) * pitch] = acc + w;
      }
      // the step kernel's render loop: one owner-code dword per (environment, board dword) from LDS (requested one
      // iteration ahead), nine v_perm_b32, nine dword stores; MODE 4: one data register reused (what the compiler
      // makes of the kernel's loop), MODE 5: nine data registers, the stores in one burst
      uint32_t q = lane, voff = 4u * lane, eF = 0;
      uint32_t code_pf = lds[q];
      const uint32_t chars_lo = 0x23402e20u, chars_hi = 0x63626150u;
#pragma unroll 1
      for (int i = 0; i < 75; ++i) {
        const uint32_t code = code_pf, vo = voff;
        q += 64; voff += 256u;
        const bool wrap = q >= 75u;
        q = wrap ? q - 75u : q;
        voff = wrap ? voff + 2400u : voff;
        eF = wrap ? eF + 77u : eF;
        code_pf = lds[i + 1 < 75 ? eF + q : 0u];
        if (MODE != 5) {
          uint32_t v = __builtin_amdgcn_perm(chars_hi, chars_lo, code);
          asm volatile("global_store_dword %0, %1, %2" : : "v"(vo), "v"(v), "s"(sb));
#pragma unroll
          for (int p = 1; p < 9; ++p) {
            uint8_t* pb = sb + 300 * p;
            v = __builtin_amdgcn_perm(p > 4 ? 1u << (8 * ((p - 1) & 3)) : 0u, p <= 4 ? 1u << (8 * ((p - 1) & 3)) : 0u, code);
            asm volatile("global_store_dword %0, %1, %2" : : "v"(vo), "v"(v), "s"(pb));
          }
        } else {
          uint32_t v[9];
          v[0] = __builtin_amdgcn_perm(chars_hi, chars_lo, code);
#pragma unroll
          for (int p = 1; p < 9; ++p) v[p] = __builtin_amdgcn_perm(p > 4 ? 1u << (8 * ((p - 1) & 3)) : 0u, p <= 4 ? 1u << (8 * ((p - 1) & 3)) : 0u, code);
          asm volatile(
              "global_store_dword %0, %1, %10\n\tglobal_store_dword %0, %2, %10 offset:300\n\tglobal_store_dword %0, %3, %10 offset:600\n\t"
              "global_store_dword %0, %4, %10 offset:900\n\tglobal_store_dword %0, %5, %10 offset:1200\n\tglobal_store_dword %0, %6, %10 offset:1500\n\t"
              "global_store_dword %0, %7, %10 offset:1800\n\tglobal_store_dword %0, %8, %10 offset:2100\n\tglobal_store_dword %0, %9, %10 offset:2400"
              : : "v"(vo), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "s"(sb));
        }
      }
    } else {
      // 75 iterations x 9 planes: lane -> (e, q) incrementally, nine dword stores 300 B apart
      uint32_t q = lane, voff = 4u * lane;
      for (int i = 0; i < 75; ++i) {
        const uint32_t v = voff ^ (uint32_t)c;
#pragma unroll
        for (int p = 0; p < 9; ++p) {
          uint8_t* pb = sb + 300 * p;
          asm volatile("global_store_dword %0, %1, %2" : : "v"(voff), "v"(v), "s"(pb));
        }
        q += 64; voff += 256u;
        const bool wrap = q >= 75u;
        q = wrap ? q - 75u : q;
        voff = wrap ? voff + 2400u : voff;
      }
    }
  }
}
int main() {
  const int n_chunks = 16384;
  uint8_t* dst; uint32_t* ctr;
  CHECK(hipMalloc(&dst, (size_t)n_chunks * CHUNK)); CHECK(hipMalloc(&ctr, (size_t)n_chunks * 64 * 15 * 4));
  CHECK(hipMemset(ctr, 0, (size_t)n_chunks * 64 * 15 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const char* names[18] = {"dword seq", "dwordx2 seq", "dwordx4 seq", "9-plane dword", "render loop", "render burst", "loop+state io",
                           "loop+unit-major io", "loop+unit-major ld", "loop+unit-major st", "loop+dma ld", "loop+dma ld (L2)", "loop+dma x4", "loop+dma x4+1",
                           "loop+dma sc1", "loop+dma sc0sc1", "loop+dma nt", "loop+s_load"};
  for (int mode : {0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 18, 19, 21, 35})
    for (int w : {1, 2, 3, 6, 8}) {
      size_t lds = (160 * 1024 / w) & ~255; if (lds < 64 * 94 * 4) lds = 64 * 94 * 4; const size_t l = lds > 65536 ? 65536 : lds;
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0));
        const dim3 g(256 * w), b(64);
        if (mode == 0) hipLaunchKernelGGL(fill<0>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 2) hipLaunchKernelGGL(fill<2>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 3) hipLaunchKernelGGL(fill<3>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 4) hipLaunchKernelGGL(fill<4>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 5) hipLaunchKernelGGL(fill<5>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 6) hipLaunchKernelGGL(fill<6>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 7) hipLaunchKernelGGL(fill<7>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 8) hipLaunchKernelGGL(fill<8>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 9) hipLaunchKernelGGL(fill<9>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 10) hipLaunchKernelGGL(fill<10>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 11) hipLaunchKernelGGL(fill<11>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 12) hipLaunchKernelGGL(fill<12>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 18) hipLaunchKernelGGL(fill<18>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 19) hipLaunchKernelGGL(fill<19>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 21) hipLaunchKernelGGL(fill<21>, g, b, l, 0, dst, n_chunks, ctr, 0);
        if (mode == 35) hipLaunchKernelGGL(fill<35>, g, b, l, 0, dst, n_chunks, ctr, 0);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      printf("%-14s (mode %d) %2d waves/CU: %.4f ms  %.0f GB/s\n", mode < 18 ? names[mode] : mode == 18 ? "loop+vmcnt(0)" : mode == 19 ? "loop+1 L1-hit row" : "loop+N dma rows", mode, w, best, (double)n_chunks * CHUNK / best / 1e6);
    }
  return 0;
}
