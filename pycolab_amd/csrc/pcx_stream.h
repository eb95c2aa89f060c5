// pcx_stream.h -- the render phase shared by the hand-written game kernels
// (pcx_warehouse.hip, pcx_marauders.hip): occlusion resolved once per
// environment, then the wavefront streams board + layer planes.
// Reference: engine.py:737-759 _render, rendering.py:85-184
// BaseObservationRenderer (paint back to front; layers[c] = board == c).
// gfx950 only.
//
// Contract with the logic phase (lane == environment).  For the group's 64
// environments it leaves in LDS
//   flat  [ND][64][FWP]  every drape's curtain as a flat cell-bit vector (bit i =
//                        cell i), environment-major with an odd pitch so that the
//                        logic phase (same word, 64 environments) and the
//                        streaming phase (same environment, consecutive words)
//                        both spread over the banks;
//   sdesc [NS][64]       per sprite {board dword it is painted in, byte mask},
//                        dword 0xFFFFFFFF when the sprite is not painted;
//   skip  [64]           environments this launch leaves untouched.
// After resolve_sprites() every board cell belongs to exactly one painter (a
// sprite, one curtain, or the backdrop), so painting is order-free and every
// layer is a mask already at hand: nothing is read back from HBM.
#pragma once

#ifdef __HIPCC_RTC__  // (a run-time build of a kernel: the device side only)
#include "pcx_device.h"
#else
#include "pcx_internal.h"
#endif
#include "pcx_crop_window.h"

namespace pcx {
namespace stream {

constexpr int WAVE = 64;

__device__ __forceinline__ uint8_t* uniform_ptr(uint8_t* p) {  // pin a wave-uniform pointer to an SGPR pair
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<uint8_t*>(((uint64_t)hi << 32) | lo);
}

// LDS-DMA (global_load_lds_dword): lane i's dword base[i] lands at LDS byte address lds_addr + 4 i -- no VGPR, nothing
// the compiler waits for; the issuer waits on vmcnt itself before reading the row.  M0 is written in the statement
// that uses it and restored (the compiler owns it); the base is copied by an SALU instruction so that an SGPR pair
// fresh from v_readfirstlane is never read by the VMEM instruction within the hazard window.  `base` and `lds_addr`
// must be wave-uniform (readfirstlane them where the compiler cannot prove it).
__device__ __forceinline__ void lds_dma_row(const uint32_t* base, uint32_t voff, uint32_t lds_addr) {
  uint32_t keep;
  uint64_t own;
  asm volatile(
      "s_mov_b64 %1, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, %1\n\ts_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(own)
      : "v"(voff), "s"(base), "s"(lds_addr)
      : "memory");
}
__device__ __forceinline__ uint32_t lds_byte_address(const uint32_t* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)p;
}
// ---- persistent workers (round 5: the launch shape of pcx_scrolly_maze_step, for the kernels built on this header) ----------
// A workgroup stays on its CU; each of its waves is a WORKER that draws work units (64 consecutive environments), steps a
// unit (lane == environment) and streams it, with the next unit's state words travelling into its LDS inbox by LDS-DMA in
// front of the current unit's plane stores (vmcnt counts in order and holds 63: after 64 plane stores they have landed),
// and at most `lock` workers of a workgroup in the streaming loop at a time.  pcx_scrolly_maze.hip has the measurements.

// The next ticket of a work counter: a SCALAR atomic (s_atomic_add; coherent across the XCDs: tools/experiments/
// satomic_probe.hip) -- through the scalar cache, not behind the CU's queue of plane stores; waited for on lgkmcnt.
__device__ __forceinline__ uint32_t scalar_ticket(uint32_t* ctr_any) {
  const uint64_t v = reinterpret_cast<uint64_t>(ctr_any);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  uint32_t* const ctr = reinterpret_cast<uint32_t*>(((uint64_t)hi << 32) | lo);
  uint32_t t = 1u;
  uint64_t own;
  asm volatile("s_mov_b64 %1, %2\n\ts_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t), "=&s"(own) : "s"(ctr) : "memory");
  return t;
}
// A counting semaphore in LDS around the streaming loop: lane 0 alone adds; a wave that finds the count at the limit takes
// its increment back and tries again a little later.  Giving up after SLOT_SPINS tries (seconds) costs nothing but speed:
// the wave then streams without a slot -- results never depend on the semaphore.
constexpr uint32_t SLOT_SPINS = 1u << 22;
__device__ __forceinline__ void slot_acquire(uint32_t lds_addr, int limit) {
  uint32_t one = 1u;
  uint32_t spins = 0;
  for (;;) {
    uint32_t old;
    uint64_t save;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %2, %3\n\ts_mov_b64 exec, %1\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(old), "=&s"(save) : "v"(lds_addr), "v"(one) : "memory");
    if (__builtin_amdgcn_readfirstlane((int)old) < limit || ++spins >= SLOT_SPINS) break;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_sub_u32 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(save) : "v"(lds_addr), "v"(one) : "memory");
    __builtin_amdgcn_s_sleep(8);
  }
}
__device__ __forceinline__ void slot_release(uint32_t lds_addr) {
  uint32_t one = 1u;
  uint64_t save;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_sub_u32 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(save) : "v"(lds_addr), "v"(one) : "memory");
}
// What a kernel's arguments carry for the scheduler (host: WorkerLaunch below fills it).
struct WorkArgs {
  uint32_t* ctr = nullptr;  // nine words 64 bytes apart: eight ticket shards (one per XCD as the hardware places workgroups: block b on XCD b % 8) and the workers-done count
  uint32_t n_units = 0;
  int32_t dynamic = 0;      // tickets (with stealing across the shards) or static round-robin
  int32_t lock = 0;         // streaming slots per workgroup (0: no limit)
};
// One worker's view of the units: first() its first unit, next(u) the one after u (>= n_units: none left).  Static:
// round-robin over all workers.  Dynamic: the first unit by position, every further one by ticket from the worker's own
// shard of the counter, then -- a shard found dry stays dry -- from the other shards (the XCDs do not finish together).
struct WorkQueue {
  uint32_t* ctr;
  uint32_t n, shards, x, wpw, nwk, wid, local, stolen;
  bool dynamic;
  // (workers_per_wg > 0: only the first so many waves of a workgroup are workers -- pcx_generic_step_pw's logic workers)
  __device__ __forceinline__ void init(const WorkArgs& w, int wave, int workers_per_wg = 0) {
    ctr = w.ctr; n = w.n_units; dynamic = w.dynamic != 0;
    wpw = workers_per_wg > 0 ? (uint32_t)workers_per_wg : blockDim.x >> 6;
    shards = gridDim.x < 8u ? gridDim.x : 8u;
    x = blockIdx.x % shards;
    nwk = gridDim.x * wpw;
    wid = blockIdx.x * wpw + (uint32_t)wave;
    local = (blockIdx.x / shards) * wpw + (uint32_t)wave;
    stolen = 0;
  }
  __device__ __forceinline__ uint32_t first() const { return dynamic ? x + shards * local : wid; }
  __device__ __forceinline__ uint32_t next(uint32_t u) {
    if (!dynamic) return u + nwk;
    while (stolen < shards) {
      const uint32_t y = x + stolen >= shards ? x + stolen - shards : x + stolen;
      const uint32_t nwk_y = ((gridDim.x - y + shards - 1u) / shards) * wpw;
      const uint32_t cand = y + shards * (nwk_y + scalar_ticket(ctr + 16u * y));
      if (cand < n) return cand;
      ++stolen;
    }
    return n;
  }
  // the last worker out rewinds the counters for the next launch (every ticket of this launch was drawn before its worker got here)
  __device__ __forceinline__ void finish(int lane) const {
    if (dynamic && lane == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (atomicAdd(ctr + 8 * 16, 1u) == nwk - 1u)
        for (int i = 0; i <= 8; ++i) atomicExch(ctr + 16 * i, 0u);
    }
  }
};

__device__ __forceinline__ const uint32_t* uniform_words(const uint32_t* p) {
  return reinterpret_cast<const uint32_t*>(uniform_ptr(reinterpret_cast<uint8_t*>(const_cast<uint32_t*>(p))));
}

// engine.py:751-757 for the last repaint of a step: a sprite is painted iff it
// is visible and nothing in front of it covers its cell; a painted sprite takes
// its cell away from every curtain (those in front do not have it, those behind
// lose it).  cell[s] = the cell sprite s is painted at, -1 when invisible.
// above[s]: bit j < NS = sprite j is in front of sprite s, bit NS + d = drape d is.
// Curtain-over-curtain occlusion must already be applied to `flat`.
template <int NS, int ND>
__device__ __forceinline__ void resolve_sprites(const int (&cell)[NS], const uint32_t (&above)[NS], uint32_t* flat,
                                                int FWP, int lane, uint2* sdesc) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int c = cell[s];
    bool shown = c >= 0;
    if (shown) {
      const uint32_t ab = above[s];
#pragma unroll
      for (int j = 0; j < NS; ++j)
        if (j != s && ((ab >> j) & 1) && cell[j] == c) shown = false;
      const int wi = c >> 5, sh = c & 31;
#pragma unroll
      for (int d = 0; d < ND; ++d)
        if (((ab >> (NS + d)) & 1) && ((flat[(d * WAVE + lane) * FWP + wi] >> sh) & 1)) shown = false;
      if (shown) {
#pragma unroll
        for (int d = 0; d < ND; ++d) flat[(d * WAVE + lane) * FWP + wi] &= ~(1u << sh);
      }
    }
    sdesc[s * WAVE + lane] = make_uint2(shown ? (uint32_t)(c >> 2) : 0xFFFFFFFFu, 0xFFu << ((c & 3) * 8));
  }
}

// resolve_sprites for the owner-code loop (stream_codes below): the same test, and a sprite that is painted writes its
// code byte over whatever the curtains left at its cell (`cb`: this environment's code bytes).  `flat` is read only.
template <int NS, int ND>
__device__ __forceinline__ void paint_sprites(const int (&cell)[NS], const uint32_t (&above)[NS], const uint32_t* flat, int FWP, int lane,
                                              uint8_t* cb, const uint32_t (&code)[NS]) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int c = cell[s];
    bool shown = c >= 0;
    const uint32_t ab = above[s];
#pragma unroll
    for (int j = 0; j < NS; ++j)
      if (j != s && ((ab >> j) & 1) && cell[j] == c) shown = false;
    const int cc = c >= 0 ? c : 0, wi = cc >> 5, sh = cc & 31;
#pragma unroll
    for (int d = 0; d < ND; ++d)
      if (((ab >> (NS + d)) & 1) && ((flat[(d * WAVE + lane) * FWP + wi] >> sh) & 1)) shown = false;
    if (shown) cb[c] = (uint8_t)code[s];
  }
}

// ... into nibble codes (stream_codes<..., NIB>): cell c of an environment is nibble (c >> 2) & 1 of byte (c >> 3) * 4 + (c & 3)
template <int NS, int ND>
__device__ __forceinline__ void paint_sprites_nib(const int (&cell)[NS], const uint32_t (&above)[NS], const uint32_t* flat, int FWP, int lane,
                                                  uint8_t* cb, const uint32_t (&code)[NS]) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int c = cell[s];
    bool shown = c >= 0;
    const uint32_t ab = above[s];
#pragma unroll
    for (int j = 0; j < NS; ++j)
      if (j != s && ((ab >> j) & 1) && cell[j] == c) shown = false;
    const int cc = c >= 0 ? c : 0, wi = cc >> 5, sh = cc & 31;
#pragma unroll
    for (int d = 0; d < ND; ++d)
      if (((ab >> (NS + d)) & 1) && ((flat[(d * WAVE + lane) * FWP + wi] >> sh) & 1)) shown = false;
    if (shown) {
      uint8_t* const b = cb + ((cc >> 3) << 2) + (cc & 3);
      const int nsh = ((cc >> 2) & 1) << 2;
      *b = (uint8_t)((*b & ~(0xFu << nsh)) | (code[s] << nsh));
    }
  }
}

// Engine(..., occlusion_in_layers=False) (rendering.py:187-301 BaseUnoccludedObservationRenderer):
// the board is painted as ever, the layers are the things' RAW masks -- a drape's whole curtain, a
// visible sprite's own cell, the backdrop character wherever the backdrop has it.  Call before
// resolve_sprites(): it keeps a copy of the curtains (FW words each) and of the sprites' cells
// that the occlusion pass will not touch; stream_planes<..., UNOCC = true> takes the layers from it.
template <int NS, int ND>
__device__ __forceinline__ void snapshot_raw(const int (&cell)[NS], const uint32_t* flat, int FW, int FWP, int lane,
                                             uint32_t* flatraw, uint2* sdescraw) {
#pragma unroll
  for (int d = 0; d < ND; ++d)
    for (int w = 0; w < FW; ++w) flatraw[(d * WAVE + lane) * FWP + w] = flat[(d * WAVE + lane) * FWP + w];
#pragma unroll
  for (int s = 0; s < NS; ++s)
    sdescraw[s * WAVE + lane] = make_uint2(cell[s] >= 0 ? (uint32_t)(cell[s] >> 2) : 0xFFFFFFFFu, 0xFFu << ((cell[s] & 3) * 8));
}

// What the streaming loop needs besides LDS: all wave-uniform.
template <int NS, int ND, int NB>
struct PlaneMap {
  uint32_t sprite_off[NS], drape_off[ND], bchar_off[NB];  // byte offset of the thing's layer plane in an environment record
  uint32_t sprite_ch4[NS], drape_ch4[ND];                 // character replicated into four bytes
};

// Optional epilogue of the streaming loop: rendering.ObservationToFeatureArray
// (rendering.py:545-661; default axis order, or channels last: `hwc` below) or
// rendering.ObservationToArray (rendering.py:409-542: `to_array` below) written by
// the step kernel itself.  Feature array: the loop holds every layer's mask dword in
// registers; a selected layer is also stored as four float32 (one 16-byte store per
// lane: 1 KiB contiguous per wave) into a caller-owned array [batch][depth][cells] --
// the consumer's tensor is ready when the step is, without a second pass over the
// planes.  slot: place of the layer in the feature stack, -1 = not selected.  With
// skip_layers the uint8 layer planes are not written at all (the board is).
struct EpilogueArgs {
  float* out = nullptr;      // the caller's array (float32 features, or elements of esize bytes: to_array); null = no epilogue
  uint32_t env_stride = 0;   // bytes per environment = depth * cells * 4 (to_array: * esize)
  uint32_t plane_bytes = 0;  // cells * 4 (to_array: * esize)
  int32_t skip_layers = 0;
  int32_t skip_board = 0;    // ... nor the board plane: the consumer ingests the epilogue's array only
  // the loop runs twice, first for the uint8 planes, then for the float32 planes: a wave then feeds half
  // as many write streams at a time (it composes every dword twice; the loop is store-bound)
  int32_t two_pass = 0;
  // channels last (ObservationToFeatureArray(permute=(1, 2, 0)), rendering.py:545-661): out is [batch][cells][depth].
  // A wave's 256 cells x depth floats are one contiguous piece of it; the lanes exchange their layer dwords through
  // 256 x depth bytes of LDS per wave (hwc_lds_off: word offset in the workgroup's dynamic LDS) so that every
  // store instruction still covers 1 KiB of consecutive bytes.
  int32_t hwc = 0, depth = 0;
  uint32_t hwc_lds_off = 0, magic_depth = 0;  // magic_depth: floor(2^32 / depth) + 1
  // ObservationToArray as the epilogue (rendering.py:409-542, default axis order): out is [batch][depth][cells]
  // elements of esize bytes, element (d, cell) = lut[d][board character]; the table ([depth][128] elements, device
  // memory) is copied to LDS by every wave for itself (lut_lds_off: word offset in the dynamic LDS).
  int32_t to_array = 0, esize = 4;
  uint32_t dword_bytes = 16;  // output bytes per board dword and plane: 4 * esize (16 for the float32 feature planes)
  uint32_t lut_lds_off = 0;
  const void* lut = nullptr;
  // ... as ObservationCharacterRepainter (rendering.py:304-406): a one-row uint8 table repaints the board (plane 0 of
  // out, a planes array [batch][1 + repaint][cells]) and plane 1 + k is the layer of repaint_ch[k]: board' == that character
  int32_t repaint = 0;
  uint8_t repaint_ch[PCX_POST_MAX_DEPTH] = {};
  int32_t sprite_slot[PCX_MAX_SPRITES], drape_slot[PCX_MAX_DRAPES], bchar_slot[PCX_MAX_CHARS];
};
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Channels-last epilogue.  A wave's 64 board dwords (first one: f_first, counted from the group's first dword) are
// 256 consecutive cells = 256 x depth consecutive floats of the output (boards of whole dwords: no padding
// between environments).  The lanes exchange through the wave's area of LDS, laid out like that piece of the
// output with a byte per float: hwc_put() drops the four cell bytes of one layer dword at [cell][layer], and
// hwc_emit() has lane l, trip j convert the dword at index j * 64 + l -- four consecutive output floats -- and store
// them: 1 KiB of consecutive bytes per store instruction.
__device__ __forceinline__ void hwc_put(uint32_t* hw, uint32_t depth, int lane, int32_t slot, uint32_t m01) {
  uint8_t* const b = reinterpret_cast<uint8_t*>(hw) + (uint32_t)(4 * lane) * depth + (uint32_t)slot;
  b[0] = (uint8_t)m01;
  b[depth] = (uint8_t)(m01 >> 8);
  b[2 * depth] = (uint8_t)(m01 >> 16);
  b[3 * depth] = (uint8_t)(m01 >> 24);
}
template <bool GUARD>
__device__ __forceinline__ void hwc_emit(const uint32_t* hw, const EpilogueArgs& epi, uint32_t f_first, int lane, bool any_skip,
                                         const uint32_t* skip, uint32_t qw, uint8_t* fbase, uint32_t f_limit) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint32_t depth = (uint32_t)epi.depth;
  const uint32_t piece = 16u * depth * f_first;  // byte offset of the piece from the group's base
  const bool check = any_skip || f_first + WAVE > f_limit;  // (uniform)
  for (uint32_t j = 0; j < depth; ++j) {
    const uint32_t g = j * WAVE + (uint32_t)lane;
    const uint32_t w = hw[g];
    f32x4 f;
    f.x = (float)(w & 0xFFu); f.y = (float)((w >> 8) & 0xFFu); f.z = (float)((w >> 16) & 0xFFu); f.w = (float)(w >> 24);
    // the four floats lie in one environment (cells * depth is a multiple of four): dropped with it
    bool dropped = false;
    if (check) {
      const uint32_t fsrc = f_first + (__umulhi(4u * g, epi.magic_depth) >> 2);  // the board dword of the floats' cell
      dropped = fsrc >= f_limit;
      if (!dropped && any_skip) dropped = skip[fsrc / qw] != 0;
    }
    if (!dropped) saddr_store_dwordx4<GUARD>(piece + 16u * g, f, fbase);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_wave_barrier();  // (the next iteration overwrites the exchange area)
}

// ObservationToArray epilogue.  to_array_stage: every wave copies the value table into LDS for itself (all waves
// write the same words, so none has to wait for another); to_array_emit: the four characters of a board dword
// through the table, one store per component plane (4, 16 or 2 x 16 bytes per lane).
__device__ __forceinline__ void to_array_stage(const EpilogueArgs& epi, uint32_t* lut_lds, int lane) {
  const uint32_t n = ((uint32_t)epi.depth * 128u * (uint32_t)epi.esize + 3u) >> 2;
  const uint32_t* const src = static_cast<const uint32_t*>(epi.lut);
  for (uint32_t i = (uint32_t)lane; i < n; i += WAVE) lut_lds[i] = src[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
template <bool GUARD>
__device__ __forceinline__ void to_array_emit(const EpilogueArgs& epi, const uint32_t* lut_lds, uint32_t board4, uint32_t aoff,
                                              uint8_t* fbase) {
  const uint32_t c0 = board4 & 127u, c1 = (board4 >> 8) & 127u, c2 = (board4 >> 16) & 127u, c3 = (board4 >> 24) & 127u;
  const uint32_t depth = (uint32_t)epi.depth;
  if (epi.esize == 1) {
    const uint8_t* const t = reinterpret_cast<const uint8_t*>(lut_lds);
    for (uint32_t d = 0; d < depth; ++d) {
      const uint32_t v = (uint32_t)t[d * 128u + c0] | ((uint32_t)t[d * 128u + c1] << 8) | ((uint32_t)t[d * 128u + c2] << 16) |
                         ((uint32_t)t[d * 128u + c3] << 24);
      saddr_store_dword<GUARD>(aoff + d * epi.plane_bytes, v, fbase);
      if (epi.repaint) {  // (depth == 1) the repainted observation's layers: bytes of v equal to the character, as 0x01
        for (int kk = 0; kk < epi.repaint; ++kk) {
          const uint32_t y = v ^ ((uint32_t)epi.repaint_ch[kk] * 0x01010101u);
          const uint32_t m = (~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) >> 7) & 0x01010101u;
          saddr_store_dword<GUARD>(aoff + (uint32_t)(1 + kk) * epi.plane_bytes, m, fbase);
        }
      }
    }
  } else if (epi.esize == 4) {
    for (uint32_t d = 0; d < depth; ++d) {
      f32x4 f;
      f.x = __uint_as_float(lut_lds[d * 128u + c0]); f.y = __uint_as_float(lut_lds[d * 128u + c1]);
      f.z = __uint_as_float(lut_lds[d * 128u + c2]); f.w = __uint_as_float(lut_lds[d * 128u + c3]);
      saddr_store_dwordx4<GUARD>(aoff + d * epi.plane_bytes, f, fbase);
    }
  } else {
    const uint2* const t = reinterpret_cast<const uint2*>(lut_lds);
    for (uint32_t d = 0; d < depth; ++d) {
      const uint2 a = t[d * 128u + c0], b = t[d * 128u + c1], c = t[d * 128u + c2], e = t[d * 128u + c3];
      f32x4 lo, hi;
      lo.x = __uint_as_float(a.x); lo.y = __uint_as_float(a.y); lo.z = __uint_as_float(b.x); lo.w = __uint_as_float(b.y);
      hi.x = __uint_as_float(c.x); hi.y = __uint_as_float(c.y); hi.z = __uint_as_float(e.x); hi.w = __uint_as_float(e.y);
      saddr_store_dwordx4<GUARD>(aoff + d * epi.plane_bytes, lo, fbase);
      saddr_store_dwordx4<GUARD>(aoff + d * epi.plane_bytes + 16u, hi, fbase);
    }
  }
}

// The wavefront streams board + layers of the group's 64 environments.
// One (environment e, board dword q) task per lane and iteration; consecutive
// lanes take consecutive dwords, so every plane store of a wave covers 256
// contiguous bytes of that plane (split over two or three environment records
// when a plane is shorter than 64 dwords).  Every store is `scalar plane base +
// one shared 32-bit lane offset`; indices advance incrementally (no multiplies
// or divisions in the loop).  NWAVES waves of a workgroup share the loop,
// iterations round-robin.
//   QW: dwords per plane (plane pitch / 4); record = (1 + L) planes.  QW == 0: the board's shape is
//     not known at compile time, qw_rt is the number of dwords (the run-time-shape instances that
//     step unshipped levels; the per-iteration work is the same, the strides live in registers).
//   EPI: also write the float32 feature-array epilogue (EpilogueArgs).
//   cell_ids (optional): a drape whose cells only ever disappear (coins) kept as a
//     per-environment bit mask over the template's list of its cells instead of a
//     flat curtain: cell_ids[q] = the list indices of board dword q's four cells
//     (0xFF = not a cell of the drape), `flat` = [64][FWP] alive masks.  Needs ND == 1.
//   UNOCC: occlusion_in_layers=False -- the layers come from the raw copies snapshot_raw() kept.
//   MODE (epilogue instances): 0 the float32 feature planes, 1 channels last, 2 ObservationToArray -- a compile-time
//     choice per loop (stream_planes() below picks one at run time): as run-time flags inside ONE loop body the two
//     later kinds cost the first a fifth of its speed (marauders 262,144: 1.61 -> 1.95 ms; profiles/r03_post_kernels.md).
template <int NS, int ND, int NB, int QW, int NWAVES, bool EPI, bool UNOCC, int MODE, bool DRAIN = true>
__device__ __forceinline__ void stream_planes_mode(const PlaneMap<NS, ND, NB>& pm, uint8_t* group_base, uint32_t env_stride,
                                                   const uint32_t* backdrop4, const uint32_t* bdmask, const uint32_t* flat,
                                                   const uint2* sdesc, const uint32_t* skip, int FWP, int lane, int wave,
                                                   const EpilogueArgs& epi, int64_t env0, const uint32_t* cell_ids,
                                                   int qw_rt, const uint32_t* flatraw, const uint2* sdescraw,
                                                   uint32_t* lds_base) {
  const uint32_t QWv = QW ? (uint32_t)QW : (uint32_t)qw_rt;
  uint8_t* const pb_board = uniform_ptr(group_base);
  uint8_t* pb_s[NS];
  uint8_t* pb_d[ND];
  uint8_t* pb_b[NB];
#pragma unroll
  for (int s = 0; s < NS; ++s) pb_s[s] = uniform_ptr(pb_board + pm.sprite_off[s]);
#pragma unroll
  for (int d = 0; d < ND; ++d) pb_d[d] = uniform_ptr(pb_board + pm.drape_off[d]);
#pragma unroll
  for (int b = 0; b < NB; ++b) pb_b[b] = uniform_ptr(pb_board + pm.bchar_off[b]);

  const bool any_skip = __ballot(skip[lane] != 0) != 0ull;
  // Drain the logic phase's own loads/stores once, here: the loop's stores are
  // inline asm the compiler cannot count, and without this it would protect a
  // register of an older store with a vmcnt(0) inside the loop.
  // (DRAIN = false: the persistent workers -- their next unit's state words are on the way into LDS and must not be
  // waited for here; the logic phase's own stores drain under the plane stores)
  if constexpr (DRAIN) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt/lgkmcnt untouched

  const bool two_pass = EPI && epi.two_pass;  // (with skip_layers the first sweep writes the board plane only)
  constexpr int lwaves = NWAVES;
  const int lwave = wave;
  constexpr uint32_t ADV = (uint32_t)lwaves * WAVE;   // tasks between a wave's consecutive iterations
  const uint32_t DE = ADV / QWv, DQ = ADV - DE * QWv;  // ... as whole environments + dwords (constants when QW != 0)
  const uint32_t f0 = (uint32_t)(lwave * WAVE + lane);
  uint32_t e = f0 / QWv, q = f0 - e * QWv;  // once
  uint32_t voff = e * env_stride + 4u * q, eF = e * (uint32_t)FWP;
  const uint32_t dvoff = DE * env_stride + 4u * DQ, dF = DE * (uint32_t)FWP;
  const uint32_t wrap_voff = env_stride - 4u * QWv;
  // epilogue addressing: one scalar base for the group, a lane offset that
  // advances like voff (16 bytes per board dword), the layer's slot added per store
  // EPI is a compile-time switch: the plain loop carries none of the epilogue's
  // branches (measured: they cost it 1.5-4 %, profiles/r02_post_kernels.md)
  constexpr bool epi_on = EPI;
  const bool layers_on = !(epi_on && epi.skip_layers);
  const uint32_t epi_rem = epi_on ? (epi.plane_bytes >> 2) & 3u : 0u;  // cells in the last dword of a plane, 0 = four
  uint8_t* const fbase = uniform_ptr(reinterpret_cast<uint8_t*>(epi.out) + (size_t)env0 * epi.env_stride);
  const uint32_t bpd = epi_on ? epi.dword_bytes : 16u;  // epilogue bytes per board dword and plane
  uint32_t foff = e * epi.env_stride + bpd * q;
  const uint32_t dfoff = DE * epi.env_stride + bpd * DQ, wrap_foff = epi.env_stride - bpd * QWv;
  constexpr bool to_array = EPI && MODE == 2;
  uint32_t* const lut_lds = to_array ? lds_base + epi.lut_lds_off : nullptr;
  if (to_array) to_array_stage(epi, lut_lds, lane);
  const uint32_t e_0 = e, q_0 = q, voff_0 = voff, eF_0 = eF, foff_0 = foff;
  // channels last: this wave's exchange area, rows of the layers nobody paints stay zero
  constexpr bool hwc = EPI && MODE == 1;
  // (two areas per wave, used in turn: an iteration drops its bytes into one and stores the floats of the
  // iteration before from the other, so that no wave waits for its own LDS writes)
  const uint32_t hw_words = hwc ? (uint32_t)epi.depth * WAVE : 0u;
  uint32_t* const hw = hwc ? lds_base + epi.hwc_lds_off + (uint32_t)wave * 2u * hw_words : nullptr;
  if (hwc)
    for (uint32_t sl = 0; sl < 2u * (uint32_t)epi.depth; ++sl) hw[sl * WAVE + lane] = 0u;
#pragma unroll 1
  for (int pass = 0; pass < (two_pass ? 2 : 1); ++pass) {
  const int role = two_pass ? pass : -1;  // 0: the uint8 planes, 1: the float32 planes, -1: both
  e = e_0; q = q_0; voff = voff_0; eF = eF_0; foff = foff_0;
  uint32_t ids_pf = cell_ids != nullptr ? cell_ids[q_0] : 0u;  // (the first iteration's coin ids; later ones are fetched an iteration ahead)
  int hw_it = -1;  // channels last: the iteration whose floats wait in the other exchange area
  uint32_t hw_sel = 0;
#pragma unroll 1
  for (int it = lwave; it < (int)QWv; it += lwaves) {
    const uint32_t e_now = e, q_now = q, voff_now = voff, eF_now = eF, foff_now = foff;
    q += DQ; e += DE; voff += dvoff; eF += dF; foff += dfoff;
    {
      const bool wrap = q >= QWv;
      q = wrap ? q - QWv : q;
      e = wrap ? e + 1 : e;
      voff = wrap ? voff + wrap_voff : voff;
      eF = wrap ? eF + FWP : eF;
      foff = wrap ? foff + wrap_foff : foff;
    }
    const uint32_t ids_now = ids_pf;  // (rotated BEFORE a skipped lane leaves the iteration: its next one must not see stale ids)
    if (cell_ids != nullptr) ids_pf = cell_ids[q];  // (q: the NEXT iteration's dword already, always inside the table)
    // (channels last: a skipped lane still takes part in the exchange below -- its stores are predicated instead)
    const bool skipped = any_skip && skip[e_now] != 0;
    if (skipped && !hwc) continue;
    // (only the single-wave shape without epilogue -- the store-issue-bound one, few enough plane
    // bases to stay in SGPRs -- takes the bare store; see pcx_internal.h saddr_store_dword)
    constexpr bool GUARD = NWAVES > 1 || EPI;
    auto put = [&](uint8_t* base, uint32_t v) { if (role != 1 && !skipped) saddr_store_dword<GUARD>(voff_now, v, base); };
    // a layer: its uint8 plane and, when selected, its float32 feature plane
    auto put_layer = [&](uint8_t* base, uint32_t m01, int32_t slot) {
      if (layers_on) put(base, m01);
      if (epi_on && slot >= 0 && role != 0 && hwc) {
        hwc_put(hw + hw_sel * hw_words, (uint32_t)epi.depth, lane, slot, m01);
      } else if (epi_on && slot >= 0 && role != 0) {
        f32x4 f;
        f.x = (float)(m01 & 0xFFu); f.y = (float)((m01 >> 8) & 0xFFu); f.z = (float)((m01 >> 16) & 0xFFu); f.w = (float)(m01 >> 24);
        const uint32_t fo = foff_now + (uint32_t)slot * epi.plane_bytes;
        // a board that is not a whole number of dwords: the plane's last dword holds 1-3 cells, and the
        // floats behind them belong to the next layer's plane
        if (epi_rem && q_now + 1u == QWv) {
          saddr_store_dword<GUARD>(fo, __float_as_uint(f.x), fbase);
          if (epi_rem > 1) saddr_store_dword<GUARD>(fo + 4u, __float_as_uint(f.y), fbase);
          if (epi_rem > 2) saddr_store_dword<GUARD>(fo + 8u, __float_as_uint(f.z), fbase);
        } else {
          saddr_store_dwordx4<GUARD>(fo, f, fbase);
        }
      }
    };
    // every LDS read of the iteration is issued up front -- the sprites' descriptors and the backdrop-only masks BEFORE the drapes'
    // bits (round 6): where `cell_ids` is a run-time pointer the drape section is a basic block of its own, and reads that follow
    // it in the source were a second LDS round trip per iteration
    uint32_t d = backdrop4[q_now];
    uint32_t md[ND > 0 ? ND : 1], ms[NS > 0 ? NS : 1], mb[NB > 0 ? NB : 1];
    uint2 sdv[NS > 0 ? NS : 1];
#pragma unroll
    for (int s = 0; s < NS; ++s) sdv[s] = sdesc[s * WAVE + e_now];
#pragma unroll
    for (int b = 0; b < NB; ++b) mb[b] = bdmask[b * QWv + q_now];
#pragma unroll
    for (int dd = 0; dd < ND; ++dd) {
      uint32_t bits;
      if (cell_ids != nullptr) {  // (uniform; resolved at compile time where the caller passes a constant)
        // Round 6: BRANCH-FREE, all four mask words requested at once, the dword's ids fetched an iteration ahead (ids_now).  The
        // four `if (id != 0xFF)` of rounds 2-5 compiled to four divergent branches, each with its own ds_read + s_waitcnt: with 92
        // coins on 4,005 cells nearly every wave iteration (256 cells) holds a coin, so a wave paid five to six DEPENDENT LDS round
        // trips per iteration -- ~1,200 cycles for eight stores, one wave per SIMD: the latency bound of
        // pcx_better_scrolly_step (profiles/r06_tuning.md section 6).
        uint32_t w4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t id = (ids_now >> (8 * j)) & 0xFFu;
          w4[j] = flat[eF_now + (id == 0xFFu ? 0u : id >> 5)];
        }
        bits = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t id = (ids_now >> (8 * j)) & 0xFFu;
          bits |= (id == 0xFFu ? 0u : (w4[j] >> (id & 31)) & 1u) << j;
        }
      } else {
        bits = (flat[dd * WAVE * FWP + eF_now + (q_now >> 3)] >> ((q_now & 7) * 4)) & 0xFu;
      }
      const uint32_t m01 = (bits * 0x00204081u) & 0x01010101u;  // bit i -> byte i
      uint32_t hi8 = m01 << 8;
      asm("" : "+v"(hi8));  // keep LLVM from folding (x << 8) - x into a quarter-rate x * 255
      md[dd] = hi8 - m01;   // 0x01 -> 0xFF per byte
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) ms[s] = sdv[s].x == q_now ? sdv[s].y : 0u;
    uint32_t uni = 0;
#pragma unroll
    for (int dd = 0; dd < ND; ++dd) {
      uni |= md[dd];
      d = (d & ~md[dd]) | (pm.drape_ch4[dd] & md[dd]);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      uni |= ms[s];
      d = (d & ~ms[s]) | (pm.sprite_ch4[s] & ms[s]);
    }
    if (!(epi_on && epi.skip_board)) put(pb_board, d);
    if (to_array && role != 0 && !skipped) to_array_emit<GUARD>(epi, lut_lds, d, foff_now, fbase);
    // rendering.py:177-179 layers[c] = (board == c): by construction the thing's
    // own mask, or the backdrop's precomputed mask where no thing paints
    if constexpr (UNOCC) {  // rendering.py:236-278: raw masks, the backdrop's included
#pragma unroll
      for (int dd = 0; dd < ND; ++dd) {
        const uint32_t bits = (flatraw[dd * WAVE * FWP + eF_now + (q_now >> 3)] >> ((q_now & 7) * 4)) & 0xFu;
        md[dd] = (bits * 0x00204081u) & 0x01010101u;
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const uint2 sr = sdescraw[s * WAVE + e_now];
        ms[s] = sr.x == q_now ? sr.y : 0u;
      }
      uni = 0;
    }
#pragma unroll
    for (int dd = 0; dd < ND; ++dd) put_layer(pb_d[dd], md[dd] & 0x01010101u, epi.drape_slot[dd]);
#pragma unroll
    for (int s = 0; s < NS; ++s) put_layer(pb_s[s], ms[s] & 0x01010101u, epi.sprite_slot[s]);
#pragma unroll
    for (int b = 0; b < NB; ++b) put_layer(pb_b[b], mb[b] & ~uni, epi.bchar_slot[b]);
    if (hwc && role != 0) {
      if (hw_it >= 0)
        hwc_emit<GUARD>(hw + (hw_sel ^ 1u) * hw_words, epi, (uint32_t)hw_it * WAVE, lane, any_skip, skip, QWv, fbase, (uint32_t)WAVE * QWv);
      hw_it = it;
      hw_sel ^= 1u;
    }
  }
  if (hwc && hw_it >= 0 && role != 0)  // the last iteration's floats
    hwc_emit<true>(hw + (hw_sel ^ 1u) * hw_words, epi, (uint32_t)hw_it * WAVE, lane, any_skip, skip, QWv, fbase, (uint32_t)WAVE * QWv);
  }  // passes
}

template <int NS, int ND, int NB, int QW, int NWAVES, bool EPI, bool UNOCC = false, bool DRAIN = true>
__device__ __forceinline__ void stream_planes(const PlaneMap<NS, ND, NB>& pm, uint8_t* group_base, uint32_t env_stride,
                                              const uint32_t* backdrop4, const uint32_t* bdmask, const uint32_t* flat,
                                              const uint2* sdesc, const uint32_t* skip, int FWP, int lane, int wave,
                                              const EpilogueArgs& epi, int64_t env0, const uint32_t* cell_ids = nullptr,
                                              int qw_rt = 0, const uint32_t* flatraw = nullptr, const uint2* sdescraw = nullptr,
                                              uint32_t* lds_base = nullptr) {
#define PCX_STREAM_MODE(m)                                                                                                       \
  stream_planes_mode<NS, ND, NB, QW, NWAVES, EPI, UNOCC, m, DRAIN>(pm, group_base, env_stride, backdrop4, bdmask, flat, sdesc, skip, FWP, lane, \
                                                            wave, epi, env0, cell_ids, qw_rt, flatraw, sdescraw, lds_base)
  if constexpr (!EPI) {
    PCX_STREAM_MODE(0);
  } else {  // (uniform: one of the three loops runs)
    if (epi.hwc) PCX_STREAM_MODE(1);
    else if (epi.to_array) PCX_STREAM_MODE(2);
    else PCX_STREAM_MODE(0);
  }
#undef PCX_STREAM_MODE
}

// The plain loop of stream_planes for ONE wave with its stores regrouped (round 6): KB consecutive iterations' dwords are
// composed first, then stored PLANE BY PLANE -- every plane receives KB x 256 contiguous bytes from the wave at a time instead
// of 256.  For boards whose planes are kilobytes long (better_scrolly_maze: 4,005 bytes) the streams of a wave are that far
// apart, and what the memory side sees from a CU is (planes x waves) interleaved streams of 256-byte pieces; boards of a few
// hundred cells write a whole record densely either way.  No epilogue, occluded layers, single-wave workgroups.
template <int NS, int ND, int NB, int QW, int KB, bool DRAIN = true>
__device__ __forceinline__ void stream_planes_burst(const PlaneMap<NS, ND, NB>& pm, uint8_t* group_base, uint32_t env_stride,
                                                    const uint32_t* backdrop4, const uint32_t* bdmask, const uint32_t* flat,
                                                    const uint2* sdesc, const uint32_t* skip, int FWP, int lane,
                                                    const uint32_t* cell_ids = nullptr, int qw_rt = 0) {
  const uint32_t QWv = QW ? (uint32_t)QW : (uint32_t)qw_rt;
  uint8_t* const pb_board = uniform_ptr(group_base);
  uint8_t* pb_s[NS > 0 ? NS : 1];
  uint8_t* pb_d[ND > 0 ? ND : 1];
  uint8_t* pb_b[NB > 0 ? NB : 1];
#pragma unroll
  for (int s = 0; s < NS; ++s) pb_s[s] = uniform_ptr(pb_board + pm.sprite_off[s]);
#pragma unroll
  for (int d = 0; d < ND; ++d) pb_d[d] = uniform_ptr(pb_board + pm.drape_off[d]);
#pragma unroll
  for (int b = 0; b < NB; ++b) pb_b[b] = uniform_ptr(pb_board + pm.bchar_off[b]);
  const bool any_skip = __ballot(skip[lane] != 0) != 0ull;
  if constexpr (DRAIN) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see stream_planes_mode
  const uint32_t DE = (uint32_t)WAVE / QWv, DQ = (uint32_t)WAVE - DE * QWv;
  uint32_t e = (uint32_t)lane / QWv, q = (uint32_t)lane - e * QWv;
  uint32_t voff = e * env_stride + 4u * q, eF = e * (uint32_t)FWP;
  const uint32_t dvoff = DE * env_stride + 4u * DQ, dF = DE * (uint32_t)FWP;
  const uint32_t wrap_voff = env_stride - 4u * QWv;
#pragma unroll 1
  for (int it = 0; it < (int)QWv; it += KB) {
    uint32_t dv[KB], vo[KB], mdv[KB][ND > 0 ? ND : 1], msv[KB][NS > 0 ? NS : 1], mbv[KB][NB > 0 ? NB : 1];
    bool live[KB];
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
      const bool in = it + kk < (int)QWv;  // (uniform)
      const uint32_t e_now = in ? e : 0u, q_now = in ? q : 0u, eF_now = in ? eF : 0u;
      vo[kk] = voff;
      live[kk] = in && !(any_skip && skip[e_now] != 0);
      if (in) {
        q += DQ; e += DE; voff += dvoff; eF += dF;
        const bool wrap = q >= QWv;
        q = wrap ? q - QWv : q;
        e = wrap ? e + 1 : e;
        voff = wrap ? voff + wrap_voff : voff;
        eF = wrap ? eF + FWP : eF;
      }
      uint32_t d = backdrop4[q_now];
#pragma unroll
      for (int dd = 0; dd < ND; ++dd) {
        uint32_t bits;
        if (cell_ids != nullptr) {
          const uint32_t ids = cell_ids[q_now];
          bits = 0;
          if (ids != 0xFFFFFFFFu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t id = (ids >> (8 * j)) & 0xFFu;
              if (id != 0xFFu) bits |= ((flat[eF_now + (id >> 5)] >> (id & 31)) & 1u) << j;
            }
          }
        } else {
          bits = (flat[dd * WAVE * FWP + eF_now + (q_now >> 3)] >> ((q_now & 7) * 4)) & 0xFu;
        }
        const uint32_t m01 = (bits * 0x00204081u) & 0x01010101u;
        uint32_t hi8 = m01 << 8;
        asm("" : "+v"(hi8));
        mdv[kk][dd] = hi8 - m01;
      }
      uint32_t uni = 0;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const uint2 sd = sdesc[s * WAVE + e_now];
        msv[kk][s] = sd.x == q_now ? sd.y : 0u;
      }
#pragma unroll
      for (int dd = 0; dd < ND; ++dd) {
        uni |= mdv[kk][dd];
        d = (d & ~mdv[kk][dd]) | (pm.drape_ch4[dd] & mdv[kk][dd]);
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        uni |= msv[kk][s];
        d = (d & ~msv[kk][s]) | (pm.sprite_ch4[s] & msv[kk][s]);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) mbv[kk][b] = bdmask[b * QWv + q_now] & ~uni;
      dv[kk] = d;
    }
    // plane by plane: KB stores to consecutive 256-byte pieces of one plane
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) if (live[kk]) saddr_store_dword<true>(vo[kk], dv[kk], pb_board);
#pragma unroll
    for (int dd = 0; dd < ND; ++dd) {
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) if (live[kk]) saddr_store_dword<true>(vo[kk], mdv[kk][dd] & 0x01010101u, pb_d[dd]);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) if (live[kk]) saddr_store_dword<true>(vo[kk], msv[kk][s] & 0x01010101u, pb_s[s]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) if (live[kk]) saddr_store_dword<true>(vo[kk], mbv[kk][b], pb_b[b]);
    }
  }
}

// ---------------------------------------------------------------------------
// Owner codes (round 6; pcx_scrolly_maze_step has rendered this way since round 2).  Instead of masks per thing the
// logic phase leaves, per environment, ONE BYTE PER BOARD CELL naming the character that shows there -- `codes`
// [64][CP] dwords, CP odd -- painted the way the reference paints the board (rendering.py:98-179: backdrop first, then
// the things back to front; here the backdrop's code dwords are a table staged once per workgroup and the things are
// byte writes).  The streaming loop then needs ONE LDS read per iteration and one v_perm_b32 per plane: the board
// dword picks each cell's character out of a table in SGPRs, layer i picks byte i of a one-hot table (layers[c] =
// board == c, rendering.py:177-179).  A v_perm_b32 selects among eight bytes, so:
//   L <= 8   code byte = i, the character's place in the template's sorted list (= its layer plane);
//   L <= 16  code byte = 0xC0 | i for i < 8, 0x0C | (i - 8) << 4 otherwise: the low nibbles select among the first
//            eight characters, the high nibbles among the others, and selector 12 is the constant 0x00 -- three more
//            VALU for the two selector dwords, two more for the board.
// Against the mask loop above (warehouse_manager: 12 LDS reads and ~60 VALU per 12 stores) this is 1 read and ~20 VALU.
// ---------------------------------------------------------------------------
template <int L>
struct CodeMap {
  uint32_t chars[4];  // the characters of codes 0-3, 4-7, 8-11, 12-15, a byte each
};
template <int L>
__host__ __device__ constexpr uint32_t code_byte(int i) { return L <= 8 ? (uint32_t)i : i < 8 ? 0xC0u | (uint32_t)i : 0x0Cu | ((uint32_t)(i - 8) << 4); }

// NIB (L <= 8 only): the codes are nibbles -- LDS dword j of an environment holds board dwords 2 j (low nibbles) and 2 j + 1
// (high nibbles), CP counts those -- half the LDS per unit for one shift and one mask per iteration (pcx_scrolly_maze_step's layout).
template <int L, int QW, int NWAVES, bool DRAIN = true, bool NIB = false>
__device__ __forceinline__ void stream_codes(const CodeMap<L>& cm, uint8_t* group_base, uint32_t env_stride, const uint32_t* codes,
                                             int CP, const uint32_t* skip, int lane, int wave, int qw_rt = 0) {
  static_assert(L >= 1 && L <= 16, "owner codes: at most sixteen characters");
  static_assert(!NIB || L <= 8, "nibble codes: at most eight characters");
  const uint32_t QWv = QW ? (uint32_t)QW : (uint32_t)qw_rt;
  uint8_t* pb[1 + L];
  pb[0] = uniform_ptr(group_base);
#pragma unroll
  for (int i = 0; i < L; ++i) pb[1 + i] = uniform_ptr(pb[0] + (uint32_t)(1 + i) * 4u * QWv);
  const uint32_t chA_lo = __builtin_amdgcn_readfirstlane(cm.chars[0]), chA_hi = __builtin_amdgcn_readfirstlane(cm.chars[1]);
  const uint32_t chB_lo = __builtin_amdgcn_readfirstlane(cm.chars[2]), chB_hi = __builtin_amdgcn_readfirstlane(cm.chars[3]);
  const bool any_skip = __ballot(skip[lane] != 0) != 0ull;
  if constexpr (DRAIN) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see stream_planes_mode
  constexpr uint32_t ADV = (uint32_t)NWAVES * WAVE;
  const uint32_t DE = ADV / QWv, DQ = ADV - DE * QWv;
  const uint32_t f0 = (uint32_t)(wave * WAVE + lane);
  uint32_t e = f0 / QWv, q = f0 - e * QWv;  // once
  // ci: the LDS word of (e, q) -- e * CP + q, or counted in board dwords with the environment's pitch doubled (NIB: word ci >> 1)
  constexpr uint32_t CS = NIB ? 2u : 1u;
  uint32_t voff = e * env_stride + 4u * q, ci = e * (uint32_t)CP * CS + q;
  const uint32_t dvoff = DE * env_stride + 4u * DQ, dci = DE * (uint32_t)CP * CS + DQ;
  const uint32_t wrap_voff = env_stride - 4u * QWv, wrap_ci = (uint32_t)CP * CS - QWv;
  const uint32_t ci_last = (uint32_t)(WAVE - 1) * (uint32_t)CP * CS + QWv - 1u;  // (the prefetch of the iteration past the end reads here)
  uint32_t code_pf = codes[NIB ? ci >> 1 : ci];
  uint32_t nib_pf = NIB ? (ci & 1u) << 2 : 0u;
  constexpr bool GUARD = NWAVES > 1 || L > 11;  // (the bare store only where every plane base stays in SGPRs)
#pragma unroll 1
  for (int it = wave; it < (int)QWv; it += NWAVES) {
    const uint32_t e_now = e, voff_now = voff, code = NIB ? (code_pf >> nib_pf) & 0x0F0F0F0Fu : code_pf;
    q += DQ; e += DE; voff += dvoff; ci += dci;
    {
      const bool wrap = q >= QWv;
      q = wrap ? q - QWv : q;
      e = wrap ? e + 1 : e;
      voff = wrap ? voff + wrap_voff : voff;
      ci = wrap ? ci + wrap_ci : ci;
    }
    {
      const uint32_t cn = ci < ci_last ? ci : ci_last;
      code_pf = codes[NIB ? cn >> 1 : cn];
      if constexpr (NIB) nib_pf = (cn & 1u) << 2;
    }
    if (any_skip && skip[e_now] != 0) continue;
    if constexpr (L <= 8) {
      saddr_store_dword<GUARD>(voff_now, __builtin_amdgcn_perm(chA_hi, chA_lo, code), pb[0]);
#pragma unroll
      for (int i = 0; i < L; ++i)
        saddr_store_dword<GUARD>(voff_now, __builtin_amdgcn_perm(i >= 4 ? 1u << (8 * (i & 3)) : 0u, i < 4 ? 1u << (8 * (i & 3)) : 0u, code), pb[1 + i]);
    } else {
      const uint32_t sa = code & 0x0F0F0F0Fu, sb = (code >> 4) & 0x0F0F0F0Fu;
      saddr_store_dword<GUARD>(voff_now, __builtin_amdgcn_perm(chA_hi, chA_lo, sa) | __builtin_amdgcn_perm(chB_hi, chB_lo, sb), pb[0]);
#pragma unroll
      for (int i = 0; i < L; ++i) {
        const int j = i & 7;
        saddr_store_dword<GUARD>(voff_now, __builtin_amdgcn_perm(j >= 4 ? 1u << (8 * (j & 3)) : 0u, j < 4 ? 1u << (8 * (j & 3)) : 0u, i < 8 ? sa : sb), pb[1 + i]);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Fused croppers (pcx_crop_window.h FusedCrops; include/pcx.h
// pcx_engine_fuse_croppers).  The frame is in LDS when the step kernel streams
// it, so the croppers' windows are cut from the same descriptors: the logic
// wave moves every window (cropping.py:393-598, one lane per environment) and
// all waves of the workgroup stream the windows' planes (cropping.py:118-227)
// -- no second pass over the observation in HBM, no extra launches.
// ---------------------------------------------------------------------------

constexpr int WCORNER_WORDS = crop::MAX_FUSED_CROPPERS * WAVE;  // LDS words a kernel sets aside for the corners
constexpr uint32_t WCORNER_NONE = 0x00008000u;                  // "this environment's window is not written"

// Logic phase, lane == environment.  track_of(i) = the packed track word
// (row | col << 8 | visible << 16) of template sprite i after this step.
// The raw curtains as the step kernels export them for drape-tracking croppers (StepArgs::export_curtains): word w
// of the template's drape d of environment e at bits[(d * FW + w) * bpad + e], cell bit r * C + c.
struct CurtainSrc { const uint32_t* bits; int64_t bpad; int FW, R, C; };

// ScrollingCropper._centroid of a drape (cropping.py:590-598): int(np.median(.)) of the set cells' row and column
// indices, one environment per lane, from the words this lane exported a moment ago.  A row is read as one or two
// 64-bit column vectors (C <= 128, R <= 63: checked on the host -- better_scrolly_maze's 45 x 89 board takes two);
// the per-column counts are kept bit-sliced (six 64-bit planes per half, a ripple-carry add per row), so the whole
// thing is a few loads and ~50 operations per row and half.
constexpr int CENTROID_MAX_ROWS = 63, CENTROID_MAX_COLS = 128;
__device__ __forceinline__ bool curtain_centroid(const CurtainSrc& cs, int d, int64_t env, int& crow, int& ccol) {
  const uint32_t* const base = cs.bits + (size_t)d * cs.FW * cs.bpad + env;
  const int R = cs.R, C = cs.C, FW = cs.FW;
  const int halves = C > 64 ? 2 : 1;
  auto word = [&](int w) { return w < FW ? base[(size_t)w * cs.bpad] : 0u; };
  auto row_bits = [&](int r, int half) {  // columns [64 half, 64 half + 64) of row r
    const int width = C - 64 * half;
    const uint32_t bit0 = (uint32_t)(r * C + 64 * half), w0 = bit0 >> 5, sh = bit0 & 31u;
    const uint64_t lo = (uint64_t)word((int)w0) | ((uint64_t)word((int)w0 + 1) << 32);
    const uint64_t hi = word((int)w0 + 2);
    uint64_t v = lo >> sh;
    v |= sh ? hi << (64u - sh) : 0ull;
    return width >= 64 ? v : v & ((1ull << width) - 1ull);
  };
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // (this lane's own export stores, read back below)
  int n = 0;
  uint64_t plane[2][6] = {};  // bit c of plane[h][k]: bit k of the number of set cells in column 64 h + c
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h >= halves) break;
      uint64_t carry = row_bits(r, h);
      n += __popcll(carry);
#pragma unroll
      for (int kk = 0; kk < 6; ++kk) { const uint64_t t = plane[h][kk] & carry; plane[h][kk] ^= carry; carry = t; }
    }
  // the two middle order statistics (0-based) of the sorted index list; their mean, truncated
  const int lo_rank = (n - 1) / 2, hi_rank = n / 2;
  int seen = 0, lo = -1, hi = -1;
  for (int r = 0; r < R; ++r) {
    const int cnt = __popcll(row_bits(r, 0)) + (halves > 1 ? __popcll(row_bits(r, 1)) : 0);
    lo = lo < 0 && seen + cnt > lo_rank ? r : lo;
    hi = hi < 0 && seen + cnt > hi_rank ? r : hi;
    seen += cnt;
  }
  crow = (lo + hi) >> 1;
  seen = 0; lo = -1; hi = -1;
  for (int c = 0; c < C; ++c) {
    int cnt = 0;
    const int h = c >> 6, b = c & 63;
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) cnt |= (int)(((h ? plane[1][kk] : plane[0][kk]) >> b) & 1ull) << kk;
    lo = lo < 0 && seen + cnt > lo_rank ? c : lo;
    hi = hi < 0 && seen + cnt > hi_rank ? c : hi;
    seen += cnt;
  }
  ccol = (lo + hi) >> 1;
  return n > 0;
}

template <typename TrackOf>
__device__ __forceinline__ void move_fused_windows(const crop::FusedCrops* fc, TrackOf track_of, bool new_episode,
                                                   int64_t env, int lane, uint32_t* wcorner, const CurtainSrc* curtains = nullptr) {
  const int n = fc->n;
  for (int w = 0; w < n; ++w) {
    const crop::FusedWindow& fw = fc->w[w];
    int top = fw.top, left = fw.left;
    if (fw.scrolling) {
      bool has = !new_episode && fw.has_corner[env] != 0;  // a new episode is a new Engine (cropping.py:378-391)
      int wrow = fw.corner[2 * env], wcol = fw.corner[2 * env + 1];
      bool have = false;
      int crow = 0, ccol = 0;
      for (int i = 0; i < fw.n_track; ++i) {  // :544-558 the first entity of to_track that has a centroid
        if (fw.track_kind[i] == 0) {         // a sprite: its position while it is visible
          const int32_t t = track_of(fw.track_sprite[i]);
          if (!have && ((t >> 16) & 1)) { crow = t & 0xFF; ccol = (t >> 8) & 0xFF; have = true; }
        } else if (curtains != nullptr) {    // a drape: the median of its raw curtain
          int r = 0, c = 0;
          const bool ok = curtain_centroid(*curtains, fw.track_sprite[i], env, r, c);
          if (!have && ok) { crow = r; ccol = c; have = true; }
        }
      }
      crop::move_window(fw.rule, have, crow, ccol, has, wrow, wcol);
      fw.has_corner[env] = 1;
      top = wrow;
      left = wcol;
    }
    fw.corner[2 * env] = top;
    fw.corner[2 * env + 1] = left;
    const bool err = crop::window_leaves_observation(fw.rule, top, left);
    fw.error[env] = (uint8_t)err;
    wcorner[w * WAVE + lane] = err ? WCORNER_NONE : (((uint32_t)top & 0xFFFFu) | ((uint32_t)left << 16));
  }
}

// All waves of the workgroup.  One (environment, output dword) task per lane:
// every output cell has exactly one source cell, so a lane asks the LDS
// descriptors who paints its four cells (backdrop character, a curtain's bit,
// the pad character outside the board: cropping.py:186-191), lays the painted
// sprites that fall into the window over the dword, and derives every layer from
// the finished board dword by a byte-wise compare -- rendering.py:177-179
// layers[c] = (board == c), which is how the reference builds the layers it
// crops.  Consecutive lanes write consecutive dwords of one output plane.
//   bchar_ch4: the backdrop-only characters, replicated into four bytes (plane order of pm.bchar_off).
//   IDS: the one drape is a mask over its cell list (cell_ids, as in stream_planes).
//   R == 0: run-time board shape `rt` (rows, cols, dwords per plane), as QW == 0 in stream_planes.
struct BoardShape { int rows, cols, qw; };
template <int NS, int ND, int NB, int QW, int NWAVES, int R, int C, bool IDS = false>
__device__ __forceinline__ void stream_windows(const crop::FusedCrops* fc, const PlaneMap<NS, ND, NB>& pm,
                                               const uint32_t (&bchar_ch4)[NB > 0 ? NB : 1], int64_t env0,
                                               const uint32_t* backdrop4, const uint32_t* flat, const uint2* sdesc,
                                               const uint32_t* skip, int FWP, int lane, int wave, const uint32_t* wcorner,
                                               const uint32_t* cell_ids = nullptr, BoardShape rt = BoardShape{0, 0, 0},
                                               int nwaves_rt = 0, int nb_rt = -1) {
  // NWAVES == 0: the number of waves sharing the loop is nwaves_rt; nb_rt >= 0: only the first nb_rt of
  // the NB backdrop-only characters exist (kernels whose character set is a run-time value)
  const int nbv = nb_rt >= 0 ? nb_rt : NB;
  const int L = NS + ND + nbv;
  const uint32_t wave_step = (uint32_t)(NWAVES ? NWAVES : nwaves_rt) * WAVE;
  const int Rv = R ? R : rt.rows, Cv = C ? C : rt.cols;
  const uint32_t pitch = 4u * (QW ? (uint32_t)QW : (uint32_t)rt.qw);
  uint32_t lay_s[NS > 0 ? NS : 1], lay_d[ND > 0 ? ND : 1], lay_b[NB > 0 ? NB : 1];
#pragma unroll
  for (int s = 0; s < NS; ++s) lay_s[s] = pm.sprite_off[s] / pitch;
#pragma unroll
  for (int d = 0; d < ND; ++d) lay_d[d] = pm.drape_off[d] / pitch;
#pragma unroll
  for (int b = 0; b < NB; ++b) lay_b[b] = pm.bchar_off[b] / pitch;
  // bytes of x equal to the bytes of c4, as 0x01 per byte (exact: no borrow between bytes)
  auto eq01 = [](uint32_t x, uint32_t c4) {
    const uint32_t y = x ^ c4;
    return (~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) >> 7) & 0x01010101u;
  };
  const int n = fc->n;
  for (int w = 0; w < n; ++w) {
    const crop::FusedWindow& fw = fc->w[w];
    const int rows = fw.rule.rows, cols = fw.rule.cols;
    const uint32_t opitch = (uint32_t)fw.out_pitch, qw = opitch >> 2, total = (uint32_t)WAVE * qw;
    const uint32_t ostride = (uint32_t)(1 + L) * opitch;
    uint8_t* const obase = uniform_ptr(fw.out + (size_t)env0 * ostride);
    const uint32_t pad = (uint32_t)(fw.rule.pad_char & 0xFF);
    const uint32_t magic_qw = 0xFFFFFFFFu / qw, magic_cols = 0xFFFFFFFFu / (uint32_t)cols;  // floor(2^32 / d) or one less
    // (round 6) what the loop needs of the window's feature stack, read ONCE: `fw` lives in global memory that the loop's own stores
    // may alias as far as the compiler knows, so `fw.feat` / `fw.feat_skip` inside the loop were a global load + s_waitcnt vmcnt(0)
    // per iteration -- and vmcnt counts the plane stores too: every iteration waited for all of its predecessor's stores to land
    float* const feat_out = fw.feat;
    const int feat_skip_w = feat_out ? fw.feat_skip : 0, feat_depth_w = feat_out ? fw.feat_depth : 0, feat_hwc_w = feat_out ? fw.feat_hwc : 0;
    // ... and the stack's characters as packed dwords (the loops over the layers read them per iteration: the same wait)
    constexpr int FCW = (crop::MAX_FUSED_FEATURES + 3) / 4;
    uint32_t fchw[FCW];
#pragma unroll
    for (int i = 0; i < FCW; ++i) {
      fchw[i] = 0;
      if (feat_out)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * i + j < crop::MAX_FUSED_FEATURES) fchw[i] |= (uint32_t)fw.feat_ch[4 * i + j] << (8 * j);
      fchw[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)fchw[i]);
    }
    auto feat_char = [&](uint32_t kk) {  // (kk is wave-uniform: a scalar select chain)
      uint32_t w = fchw[0];
#pragma unroll
      for (int i = 1; i < FCW; ++i) w = (kk >> 2) == (uint32_t)i ? fchw[i] : w;
      return (w >> (8u * (kk & 3u))) & 0xFFu;
    };
    // plane bases in SGPRs, one shared 32-bit lane offset (as in the board loop)
    uint8_t* pb_s[NS > 0 ? NS : 1];
    uint8_t* pb_d[ND > 0 ? ND : 1];
    uint8_t* pb_b[NB > 0 ? NB : 1];
#pragma unroll
    for (int s = 0; s < NS; ++s) pb_s[s] = uniform_ptr(obase + lay_s[s] * opitch);
#pragma unroll
    for (int d = 0; d < ND; ++d) pb_d[d] = uniform_ptr(obase + lay_d[d] * opitch);
#pragma unroll
    for (int b = 0; b < NB; ++b) pb_b[b] = uniform_ptr(obase + lay_b[b] * opitch);
    for (uint32_t f0 = (uint32_t)wave * WAVE; f0 < total; f0 += wave_step) {
      // straight-line code (selects, clamped indices); one predicated region for the stores at the end
      const uint32_t f = f0 + (uint32_t)lane;
      const bool in_range = f < total;
      uint32_t e = __umulhi(f, magic_qw), q = f - e * qw;  // f / qw: the estimate is at most one short
      const bool carry = q >= qw;
      q = carry ? q - qw : q;
      e = carry ? e + 1 : e;
      e = in_range ? e : 0u;
      const uint32_t cw = wcorner[w * WAVE + e], sk = skip[e];
      const bool active = (uint32_t)in_range & (uint32_t)(sk == 0u) & (uint32_t)(cw != WCORNER_NONE);
      const int top = (int)(int16_t)(cw & 0xFFFFu), left = (int)(int16_t)(cw >> 16);
      const uint32_t cell0 = q * 4u;
      uint32_t orow = __umulhi(cell0, magic_cols), ocol = cell0 - orow * (uint32_t)cols;
      const bool carry2 = ocol >= (uint32_t)cols;
      ocol = carry2 ? ocol - (uint32_t)cols : ocol;
      orow = carry2 ? orow + 1 : orow;
      const uint32_t eF = e * (uint32_t)FWP;
      // the cell every painted sprite shows at (resolve_sprites: at most one sprite per cell), none: no cell
      uint32_t scell[NS > 0 ? NS : 1];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        // one 8-byte read; dword 0xFFFFFFFF (not painted) gives a cell no window cell has
        const uint64_t sd = reinterpret_cast<const uint64_t*>(sdesc)[s * WAVE + e];
        scell[s] = ((uint32_t)sd << 2) | ((uint32_t)__builtin_ctz((uint32_t)(sd >> 32)) >> 3);
      }
      // The output dword four cells at a time, the way the board loop composes a board dword: a run of
      // window cells that lies in one window row is a run of consecutive board cells, so its backdrop
      // bytes are two aligned LDS dwords funnelled by the byte phase, a curtain's four bits one shifted
      // pair of flat words, a painted sprite a byte mask by its distance from the run's first cell.  A
      // dword that straddles window rows (cols % 4 != 0) is two or more such runs.
      const int QWsrc = (int)(pitch >> 2);
      uint32_t od = 0;
      int done = 0, wr = (int)orow, wc = (int)ocol;
      while (done < 4) {
        const int n = cols - wc < 4 - done ? cols - wc : 4 - done;  // cells of this run
        const bool real_row = wr < rows;                              // rows past the window are plane padding: zeros
        const int sr = top + wr, sc = left + wc;
        const bool row_in = real_row && (unsigned)sr < (unsigned)Rv;
        // the run's cells that lie inside the board: columns [lo, hi) of the run
        const int lo = sc < 0 ? -sc : 0, hi = Cv - sc < n ? Cv - sc : n;
        const bool any_in = row_in && lo < hi;
        const uint32_t a = any_in ? (uint32_t)(sr * Cv + sc + lo) : 0u;  // first board cell read
        const uint32_t qa = a >> 2, qb = (int)qa + 1 < QWsrc ? qa + 1 : qa, ph = a & 3u;
        uint32_t d = __builtin_amdgcn_alignbyte(backdrop4[qb], backdrop4[qa], ph);
        if constexpr (IDS) {
          const uint32_t ids = __builtin_amdgcn_alignbyte(cell_ids[qb], cell_ids[qa], ph);
          if (ids != 0xFFFFFFFFu) {
            uint32_t bits = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t id = (ids >> (8 * j)) & 0xFFu;
              const uint32_t word = flat[eF + ((id >> 5) & 7u)];  // (id 0xFF reads word 7 of the mask: in range, ignored)
              bits |= ((word >> (id & 31u)) & (uint32_t)(id != 0xFFu)) << j;
            }
            const uint32_t m01 = (bits * 0x00204081u) & 0x01010101u;
            uint32_t hi8 = m01 << 8;
            asm("" : "+v"(hi8));
            const uint32_t m = hi8 - m01;
            d = (d & ~m) | (pm.drape_ch4[0] & m);
          }
        } else {
#pragma unroll
          for (int dd = 0; dd < ND; ++dd) {
            const uint32_t w0 = a >> 5, w1 = (int)w0 + 1 < FWP ? w0 + 1 : w0;
            const uint64_t pair = (uint64_t)flat[dd * WAVE * FWP + eF + w0] | ((uint64_t)flat[dd * WAVE * FWP + eF + w1] << 32);
            const uint32_t bits = (uint32_t)(pair >> (a & 31u)) & 0xFu;
            const uint32_t m01 = (bits * 0x00204081u) & 0x01010101u;
            uint32_t hi8 = m01 << 8;
            asm("" : "+v"(hi8));
            const uint32_t m = hi8 - m01;
            d = (d & ~m) | (pm.drape_ch4[dd] & m);
          }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const uint32_t delta = scell[s] - a;
          const uint32_t m = delta < 4u ? 0xFFu << (8u * delta) : 0u;
          d = (d & ~m) | (pm.sprite_ch4[s] & m);
        }
        // bytes [0, hi - lo) of d are the board cells of the run's columns [lo, hi); the rest of the
        // run is the pad character, what is not a window cell at all stays zero
        const int nin = any_in ? hi - lo : 0;
        const uint32_t keep = nin >= 4 ? 0xFFFFFFFFu : (1u << (8 * nin)) - 1u;
        uint32_t run = any_in ? (d & keep) << (8 * lo) : 0u;
        const uint32_t in_mask = any_in ? keep << (8 * lo) : 0u;
        const uint32_t run_mask = n >= 4 ? 0xFFFFFFFFu : (1u << (8 * n)) - 1u;
        run |= real_row ? (pad * 0x01010101u) & run_mask & ~in_mask : 0u;
        od |= run << (8 * done);
        done += n;
        wc += n;
        if (wc >= cols) { wc = 0; ++wr; }
      }
      if (active) {
        const uint32_t voff = e * ostride + 4u * q;
        auto put = [&](uint8_t* base, uint32_t v) { saddr_store_dword<true>(voff, v, base); };  // (compute-bound loop)
        const int fskip = feat_skip_w;
        if (fskip < 2) put(obase, od);
        if (fskip < 1) {
#pragma unroll
          for (int d = 0; d < ND; ++d) put(pb_d[d], eq01(od, pm.drape_ch4[d]));
#pragma unroll
          for (int s = 0; s < NS; ++s) put(pb_s[s], eq01(od, pm.sprite_ch4[s]));
#pragma unroll
          for (int b = 0; b < NB; ++b)
            if (b < nbv) put(pb_b[b], eq01(od, bchar_ch4[b]));
        }
        if (feat_out) {
          // crop -> post-process in this launch: the window's feature stack (rendering.py:610-661 on the cropped
          // observation): layer k of the window is (window board == feat_ch[k]), as float32.  The window's cells
          // are rows x cols exactly (no plane padding in the array): its last dword may hold fewer than four.
          const uint32_t wcells = (uint32_t)(rows * cols), depth = (uint32_t)feat_depth_w;
          const uint32_t valid = wcells - 4u * q >= 4u ? 4u : wcells - 4u * q;
          uint8_t* const fbase = uniform_ptr(reinterpret_cast<uint8_t*>(feat_out) + (size_t)env0 * depth * wcells * 4u);
          const uint32_t fenv = e * depth * wcells * 4u;
          if (!feat_hwc_w) {
            for (uint32_t kk = 0; kk < depth; ++kk) {
              const uint32_t m = eq01(od, feat_char(kk) * 0x01010101u);
              const uint32_t fo = fenv + (kk * wcells + 4u * q) * 4u;
              if (valid == 4u) {
                f32x4 f;
                f.x = (float)(m & 0xFFu); f.y = (float)((m >> 8) & 0xFFu); f.z = (float)((m >> 16) & 0xFFu); f.w = (float)(m >> 24);
                saddr_store_dwordx4<true>(fo, f, fbase);
              } else {
                for (uint32_t j = 0; j < valid; ++j) *reinterpret_cast<float*>(fbase + fo + 4u * j) = (float)((m >> (8u * j)) & 0xFFu);
              }
            }
          } else {
            for (uint32_t j = 0; j < valid; ++j) {
              const uint32_t ch = (od >> (8u * j)) & 0xFFu;
              float* const cellp = reinterpret_cast<float*>(fbase + fenv + (4u * q + j) * depth * 4u);
              for (uint32_t kk = 0; kk < depth; ++kk) cellp[kk] = ch == feat_char(kk) ? 1.0f : 0.0f;
            }
          }
        }
      }
    }
  }
}

#ifndef __HIPCC_RTC__  // ---- host side from here to the matching #endif (a run-time build of a kernel has no use for it) ----
// Host side: an epilogue descriptor (include/pcx.h) as EpilogueArgs for a backend
// whose sprites / drape slots / backdrop-only characters paint the given
// characters.  (Boards that are not a whole number of dwords are fine: stream_planes writes the
// last dword of a feature plane cell by cell.)
// hwc_lds_room: bytes of LDS the channels-last exchange areas may take (64 KB minus the kernel's own LDS), for
// hwc_waves waves per workgroup at most -- a deeper stack is refused here, not at launch
inline bool fill_epilogue(EpilogueArgs& result, const pcx_epilogue_desc* d, int cells, const int* sprite_ch, int ns,
                          const int* drape_ch, int nd, const int* bchar_ch, int nb, size_t hwc_lds_room = 0, int hwc_waves = 1) {
  EpilogueArgs a;  // (committed to `result` on success only: a refused descriptor changes nothing)
  for (int i = 0; i < PCX_MAX_SPRITES; ++i) a.sprite_slot[i] = -1;
  for (int i = 0; i < PCX_MAX_DRAPES; ++i) a.drape_slot[i] = -1;
  for (int i = 0; i < PCX_MAX_CHARS; ++i) a.bchar_slot[i] = -1;
  if (!d) { result = a; return true; }
  a.out = d->out_dev;
  a.env_stride = (uint32_t)d->depth * (uint32_t)cells * 4u;
  a.plane_bytes = (uint32_t)cells * 4u;
  a.skip_layers = d->skip_layers != 0;
  a.skip_board = d->skip_layers >= 2;
  // more than sixteen write streams per wave (board + layers + float planes) go faster as two passes
  // (marauders 32,768: 0.275 -> 0.198 ms, step + separate kernel: 0.280; hello_world's fifteen do not:
  // profiles/r03_post_kernels.md)
  a.two_pass = 1 + ns + nd + nb + d->depth > 16;
  a.depth = d->depth;
  if (d->to_array) {  // the value table (uploaded by the engine: Backend::epilogue_args()->lut) goes to LDS next to the kernel's own
    const int esize = d->dtype == PCX_U8 ? 1 : (d->dtype == PCX_I32 || d->dtype == PCX_F32) ? 4 : 8;
    if (cells % 4 != 0 || d->channels_last) return false;  // (whole dwords of cells per plane; default axis order)
    if ((size_t)d->depth * 128 * esize + 16 > hwc_lds_room) return false;
    a.to_array = 1;
    a.esize = esize;
    a.dword_bytes = 4u * (uint32_t)esize;
    a.env_stride = (uint32_t)d->depth * (uint32_t)cells * (uint32_t)esize;
    a.plane_bytes = (uint32_t)cells * (uint32_t)esize;
    if (d->to_array == 2) {  // the repainter: one table row, 1 + depth output planes
      if (esize != 1) return false;
      a.repaint = d->depth;
      for (int i = 0; i < d->depth; ++i) a.repaint_ch[i] = d->chars[i];
      a.depth = 1;
      a.magic_depth = 0;
      a.env_stride = (uint32_t)(1 + d->depth) * (uint32_t)cells;
      a.two_pass = 1 + ns + nd + nb + 1 + d->depth > 16;
    }
  }
  a.magic_depth = 0xFFFFFFFFu / (uint32_t)d->depth + 1u;
  if (d->channels_last) {  // one float32 stream instead of `depth`; needs boards of whole dwords (no padding between environments)
    if (cells % 4 != 0) return false;
    if ((size_t)hwc_waves * 2 * (size_t)d->depth * WAVE * 4 > hwc_lds_room) return false;
    a.hwc = 1;
    a.two_pass = 1;  // measured: two sweeps win on every kernel (scrolly_maze 1M: 2.66 vs 3.42 ms; profiles/r03_post_kernels.md)
  }
  if (d->skip_layers) a.two_pass = 0;  // board plane + float planes only: one sweep (measured: tools/skip_layers_bench.py)
  if (const char* e = getenv("PCX_EPI_TWO_PASS")) a.two_pass = atoi(e) != 0;
  for (int f = 0; f < d->depth && !d->to_array; ++f) {
    for (int i = 0; i < ns; ++i) if (sprite_ch[i] == d->chars[f]) a.sprite_slot[i] = f;
    for (int i = 0; i < nd; ++i) if (drape_ch[i] == d->chars[f]) a.drape_slot[i] = f;
    for (int i = 0; i < nb; ++i) if (bchar_ch[i] == d->chars[f]) a.bchar_slot[i] = f;
  }
  result = a;
  return true;
}

// Host side, at launch: the channels-last epilogue's exchange area goes behind the kernel's own dynamic LDS.
inline EpilogueArgs with_hwc_scratch(EpilogueArgs a, size_t& lds_bytes, int waves_per_workgroup) {
  if (a.out && a.to_array) {  // the value table of the ObservationToArray epilogue
    a.lut_lds_off = (uint32_t)((lds_bytes + 3) / 4);
    lds_bytes = 4 * (size_t)a.lut_lds_off + (((size_t)a.depth * 128 * (size_t)a.esize + 15) & ~(size_t)15);
  }
  if (a.out && a.hwc) {
    a.hwc_lds_off = (uint32_t)((lds_bytes + 3) / 4);
    lds_bytes = 4 * (size_t)a.hwc_lds_off + (size_t)waves_per_workgroup * 2 * (size_t)a.depth * WAVE * 4;  // two areas per wave
  }
  return a;
}

// Host side of the fused croppers: their description lives in device memory (the
// kernels take a pointer, null = none), rewritten only when croppers are fused
// or released -- never on the step path.
struct FusedCropsHolder {
  DevArray<crop::FusedCrops> dev;
  bool on = false;
  bool only = false;  // windows only: the full-board planes are not written (the launch is a few KB per environment:
                      // the backends then share a group among four waves up to many more groups per CU)
  // rows, cols: the board, for kernels that take a drape's median from the curtains they export (curtain_centroid:
  // rows of at most 128 cells, at most 63 of them); drapes_ok: the kernel has its own way (pcx_generic.hip)
  int set(const crop::FusedCrops* fc, bool drapes_ok = false, int rows = 0, int cols = 0) {
    if (!drapes_ok && crop::tracks_drapes(fc) && !(rows > 0 && rows <= CENTROID_MAX_ROWS && cols > 0 && cols <= CENTROID_MAX_COLS))
      return set_error(PCX_E_UNSUPPORTED, "fused croppers: a cropper that tracks a drape is fused on boards of at most %d x %d cells (this one: %d x %d)",
                       CENTROID_MAX_ROWS, CENTROID_MAX_COLS, rows, cols);
    PCX_HIP(hipDeviceSynchronize());  // no launch in flight may still read the old description
    if (!fc || fc->n <= 0) { on = false; only = false; return 0; }
    if (!dev.ptr) { if (int rc = dev.alloc(1)) return rc; }
    PCX_HIP(hipMemcpy(dev.ptr, fc, sizeof *fc, hipMemcpyHostToDevice));
    on = true;
    only = fc->only != 0;
    return 0;
  }
  const crop::FusedCrops* ptr() const { return on ? dev.ptr : nullptr; }
};
#endif

// Constants of the streaming phase every backend derives the same way.
struct Layout {
  int cells = 0, pitch = 0, QW = 0, FW = 0, FWP = 0, CP = 0, CPN = 0;
  void set(int rows, int cols) {
    cells = rows * cols;
    pitch = (cells + 3) & ~3;
    QW = pitch / 4;
    FW = (cells + 31) / 32;
    FWP = FW | 1;
    CP = QW | 1;  // owner codes: dwords per environment, odd (the logic phase reads one word of 64 environments, the streaming loop consecutive words)
    CPN = ((QW + 1) / 2) | 1;  // ... as nibbles: two board dwords per LDS dword
  }
};

}  // namespace stream
}  // namespace pcx
