// pcx_device.h -- what the DEVICE side of every kernel file needs and nothing else: the fixed-width types, the
// C ABI's plain structs (include/pcx.h), the launch arguments and the store helpers.  It is the root of the headers
// that are also compiled at run time (pcx_generic.hip builds a copy of pcx_generic_step specialised for one template
// with hiprtc, which has no standard headers), so nothing here or in the headers above it may pull a host header in
// when __HIPCC_RTC__ is defined.  gfx950 only.
#pragma once

#ifdef __HIPCC_RTC__
// (hiprtc keeps its own fixed-width types in a namespace)
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long uintptr_t;
#define INT32_MIN (-2147483647 - 1)
#else
#include <hip/hip_runtime.h>

#include <cstdint>
#endif

#include "pcx.h"

namespace pcx {

// Error bits reported per environment (the reference would have raised).
enum : uint8_t { ERR_INDEX = 1, ERR_SCROLL = 2 };  // IndexError (numpy index rules, np.random.choice([])), scrolling.Error

struct StepArgs {
  const int32_t* actions = nullptr;  // device int32[batch] or null when hashed
  const uint8_t* reset_mask = nullptr;  // reset mode: device uint8[batch] or null
  int mode = 0;                      // 0 step, 1 reset
  int auto_reset = 0;
  int hashed = 0;
  uint64_t seed = 0;
  int64_t env_offset = 0;
  int64_t t = 0;
  // several consecutive steps in one launch (backends that can: max_fused_steps):
  // step i takes actions + i * action_stride, or hash step t + i
  int n_steps = 1;
  int64_t action_stride = 0;
  int envs_per_group = 64;  // cooperative launch shapes that split a wave's worth of environments further (pcx_scrolly_maze_step)
  int export_curtains = 0;  // write every drape's raw curtain bits to curtain_bits() (drape-tracking croppers)
  int debug = 0;  // ablation bits for profiling (PCX_DEBUG env): 1 skip entity updates, 2 skip phase B, 4 skip render descriptors
};

// Plane stores of the render loops: `global_store_dword voffset, data, sbase` -- a wave-uniform
// 64-bit base in an SGPR pair plus one 32-bit lane offset shared by every plane -- written as
// inline asm because the compiler does not pick this form by itself.  The price: the statement is
// opaque to the compiler's hazard recogniser.  The ISA wants 5 wait states between a VALU write
// of an SGPR (v_readlane_b32 / v_readfirstlane_b32: how the register allocator fetches an SGPR
// it had parked in a VGPR lane) and a VMEM instruction that reads that SGPR as its base; the
// compiler pads its own VMEM instructions, not these.  Two defences:
//  * GUARD: the statement copies the base with s_mov_b64 first -- an SALU read of a VALU-written
//    SGPR is interlocked, and an SALU write needs no wait before a VMEM read -- for the instances
//    that are not store-issue-bound;
//  * the build scans every kernel's assembly for the pattern (tools/sgpr_hazard_scan.py, run by
//    csrc/Makefile) and fails if an unguarded store sits within 5 wait states of such a write.
typedef float pcx_f32x4 __attribute__((ext_vector_type(4)));
template <bool GUARD>
__device__ __forceinline__ void saddr_store_dword(uint32_t voff, uint32_t v, uint8_t* base) {
  if constexpr (GUARD) {
    uint64_t own;
    asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dword %1, %2, %0" : "=&s"(own) : "v"(voff), "v"(v), "s"(base));
  } else {
    asm volatile("global_store_dword %0, %1, %2" : : "v"(voff), "v"(v), "s"(base));
  }
}
// (s_nop 1: a VMEM store of more than 64 bits must not be followed at once by a VALU write of its
// data registers -- again a wait state the compiler cannot insert into inline asm)
template <bool GUARD>
__device__ __forceinline__ void saddr_store_dwordx4(uint32_t voff, pcx_f32x4 v, uint8_t* base) {
  if constexpr (GUARD) {
    uint64_t own;
    asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dwordx4 %1, %2, %0\n\ts_nop 1" : "=&s"(own) : "v"(voff), "v"(v), "s"(base));
  } else {
    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(base));
  }
}

}  // namespace pcx
