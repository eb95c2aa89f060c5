// pcx_hello_world.hip -- hand-written fused step kernel for hello_world
// (reference: pycolab/examples/hello_world.py:72-123 driven by engine.py:583-847).
// gfx950 only.  Same shape as the other hand-written kernels: 64
// consecutive environments per workgroup, logic phase lane == environment,
// render phase = pcx_stream.h.
//
// One update group; nothing looks at the board.  RollingDrape's curtain is a flat
// cell-bit vector (bit r * C + c) held in registers with compile-time indices:
// np.roll(curtain, +-1, axis=0) is a rotate by C bits of the R*C-bit vector,
// np.roll(+-1, axis=1) two shifts and two constant column masks.  SlidingSprites
// are plain Sprites that wrap around the board.
// Other shapes or casts: the table-driven kernel (the engine falls back to it).

#include "pcx_internal.h"
#include "pcx_stream.h"

#include <cstdlib>
#include <cstring>

namespace pcx {
namespace hw {

using stream::WAVE;
constexpr int NS = 4, ND = 1, NB = 2;

enum : int { W_FRAME = 0, W_FLAGS, W_POS, W_D = W_POS + NS };
constexpr uint32_t F_OVER = 1u, F_ERR_SHIFT = 1;

struct Consts {
  int32_t n_actions;
  uint32_t visible;        // bit s: Sprite._visible (never changes)
  uint32_t dx[NS], dy[NS];  // per action a: ((d >> 2a) & 3) - 1   (hello_world.py:98-99)
  uint32_t above[NS];
  uint32_t init[W_D];
  uint32_t sprite_off[NS], sprite_ch4[NS], drape_off, drape_ch4, bchar_off[NB], bchar_ch4[NB];
  // owner codes (pcx_stream.h stream_codes; the CODES instances): a character's code is its layer plane (7 characters: one v_perm_b32 per plane)
  uint32_t sprite_code[NS], drape_code4, code_chars[4];
};

struct Ptrs {
  const uint32_t* tables;        // staged into LDS: backdrop4 [QW], bdmask [NB][QW], backdrop codes [QW]
  const uint32_t* init_curtain;  // [FW]
  uint32_t* state;               // [NW][bpad]
  int32_t* track;                // [NS][bpad]
  uint32_t* curtains;            // [1][FW][bpad] (export_curtains)
  int64_t batch, bpad;
  stream::WorkArgs work;         // PW instances: the persistent workers' scheduler (pcx_stream.h)
};

__device__ __forceinline__ uint32_t action_hash(uint64_t seed, uint64_t env, uint64_t t) {
  uint64_t x = seed ^ (env * 0x9E3779B97F4A7C15ull) ^ (t * 0xBF58476D1CE4E5B9ull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}

// ---- the R*C-bit curtain in FW registers (compile-time indices only) ----------
template <int R, int C>
struct Bits {
  static constexpr int cells = R * C, FW = (cells + 31) / 32;
  static constexpr uint32_t col_mask(int i, int col) {
    uint32_t m = 0;
    for (int b = 0; b < 32; ++b) { const int bit = 32 * i + b; if (bit < cells && bit % C == col) m |= 1u << b; }
    return m;
  }
  static constexpr uint32_t valid(int i) {
    return 32 * i + 32 <= cells ? 0xFFFFFFFFu : 32 * i >= cells ? 0u : ((1u << (cells - 32 * i)) - 1u);
  }
  template <int S>
  static __device__ __forceinline__ void shl(const uint32_t (&x)[FW], uint32_t (&out)[FW]) {
    constexpr int ws = S / 32, bs = S % 32;
#pragma unroll
    for (int i = 0; i < FW; ++i) {
      const uint32_t hi = i - ws >= 0 ? x[i - ws >= 0 ? i - ws : 0] : 0u;
      const uint32_t lo = i - ws - 1 >= 0 ? x[i - ws - 1 >= 0 ? i - ws - 1 : 0] : 0u;
      out[i] = bs ? (hi << bs) | (lo >> ((32 - bs) & 31)) : hi;
    }
  }
  template <int S>
  static __device__ __forceinline__ void shr(const uint32_t (&x)[FW], uint32_t (&out)[FW]) {
    constexpr int ws = S / 32, bs = S % 32;
#pragma unroll
    for (int i = 0; i < FW; ++i) {
      const uint32_t lo = i + ws < FW ? x[i + ws < FW ? i + ws : 0] : 0u;
      const uint32_t hi = i + ws + 1 < FW ? x[i + ws + 1 < FW ? i + ws + 1 : 0] : 0u;
      out[i] = bs ? (lo >> bs) | (hi << ((32 - bs) & 31)) : lo;
    }
  }
};

// PW (round 5): persistent workers, as in pcx_warehouse_step -- the workgroup stays on its CU, each of its waves draws units
// of 64 environments, steps one and streams it alone, the next unit's state rows prefetched into its LDS inbox by LDS-DMA,
// at most `work.lock` workers of a workgroup streaming at a time (pcx_stream.h).  Plain steps only.
// CODES (round 6): the render phase is pcx_stream.h's owner-code loop -- the logic lane leaves a code byte per board cell (the
// backdrop's code dwords with the rolling drape's bits merged in four at a time, the painted sprites as byte writes).  Plain steps.
template <int R, int C, int NWAVES, bool EPI = false, bool UNOCC = false, bool PW = false, bool CODES = false>
__global__ __launch_bounds__(PW ? 8 * WAVE : NWAVES* WAVE) void pcx_hello_world_step(const Consts k, const Ptrs P, const StepArgs a,
                                                                      const pcx_buffers out, const stream::EpilogueArgs epi,
                                                                      const crop::FusedCrops* fc) {
  extern __shared__ uint32_t lds[];
  using B = Bits<R, C>;
  constexpr int cells = R * C, pitch = (cells + 3) & ~3, QW = pitch / 4, FW = B::FW, FWP = FW | 1;
  constexpr int L = NS + ND + NB;
  constexpr int CP = ((QW + 1) / 2) | 1;  // nibble codes: two board dwords per LDS dword (7 characters)
  constexpr int O_BD = 0, O_BDM = O_BD + QW, O_BDC = O_BDM + NB * QW, O_TAB_END = O_BDC + QW;
  // (CODES: the per-environment code dwords take the place of the sprite descriptors)
  constexpr int O_FLAT = O_TAB_END, O_SDESC = (O_FLAT + WAVE * FWP + 1) & ~1, O_SKIP = O_SDESC + (CODES ? WAVE * CP : 2 * NS * WAVE);
  static_assert(!CODES || (!EPI && !UNOCC), "owner codes: plain steps");
  constexpr int O_WCORNER = O_SKIP + WAVE;  // fused croppers' window corners
  constexpr int O_FLATRAW = O_WCORNER + stream::WCORNER_WORDS, O_SDESCRAW = (O_FLATRAW + WAVE * FWP + 1) & ~1;  // UNOCC only
  static_assert(!PW || (NWAVES == 1 && !EPI && !UNOCC), "persistent workers: plain steps");
  // PW: a worker's own LDS region {flat, sdesc, skip, inbox}; the inbox holds the unit's state rows and its tape actions
  constexpr int NWORDS = W_D + FW, IB_ROWS = NWORDS + 1;
  constexpr int O_SEM = O_FLAT, O_W0 = O_FLAT + 2, W_WORDS = ((O_WCORNER - O_FLAT) + IB_ROWS * WAVE + 1) & ~1;
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
  const int mine = PW ? O_W0 - O_FLAT + __builtin_amdgcn_readfirstlane(wave) * W_WORDS : 0;  // (word offset of this worker's region)
  for (int i = threadIdx.x; i < O_TAB_END; i += (int)blockDim.x) lds[i] = P.tables[i];
  uint32_t* const flat = lds + O_FLAT + mine;
  uint2* const sdesc = reinterpret_cast<uint2*>(lds + O_SDESC + mine);
  uint32_t* const codes = lds + O_SDESC + mine;  // (CODES)
  uint32_t* const skipv = lds + O_SKIP + mine;
  uint32_t* const wcorner = lds + O_WCORNER;
  uint32_t* const inbox = lds + O_WCORNER + mine;  // (PW only: behind the worker's skip flags)
  if (PW && threadIdx.x == 0) lds[O_SEM] = 0;      // the streaming semaphore
  stream::WorkQueue wq;
  uint32_t unit = blockIdx.x;
  bool need_wait = true;
  auto prefetch = [&](uint32_t u_any) {  // the state rows of unit `u` (and its tape actions) into the inbox
    const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u_any);
    const int64_t e0 = (int64_t)u * WAVE;
    const uint32_t ib = (uint32_t)__builtin_amdgcn_readfirstlane((int)stream::lds_byte_address(inbox));
#pragma unroll
    for (int w = 0; w < NWORDS; ++w) stream::lds_dma_row(P.state + (int64_t)w * P.bpad + e0, 4u * lane, ib + (uint32_t)w * (4u * WAVE));
    if (!a.hashed && e0 + lane < P.batch) stream::lds_dma_row(reinterpret_cast<const uint32_t*>(a.actions) + e0, 4u * lane, ib + (uint32_t)NWORDS * (4u * WAVE));
  };
  if constexpr (PW) {
    wq.init(P.work, wave);
    unit = wq.first();
    if (unit < wq.n) prefetch(unit);  // (under the staging of the tables)
  }
  __syncthreads();

  for (;;) {  // (PW: this worker's units; else one round)
  if constexpr (PW) { if (unit >= wq.n) break; }
  const int64_t env0 = (int64_t)unit * WAVE;
  if (wave == 0 || PW) {
    const int64_t env = env0 + lane, bp = P.bpad;
    const bool live = env < P.batch;
    uint32_t* const st = P.state + env;
    uint32_t flags = 0, ld_frame = 0, ld_pos[NS] = {}, x[FW];
    int ld_action = PCX_ACTION_NONE;
    bool skip = !live, do_reset = false;
    int action = PCX_ACTION_NONE;
#pragma unroll
    for (int i = 0; i < FW; ++i) x[i] = 0;
    if constexpr (PW) {  // the unit's state rows are in the inbox (the first unit's must be waited for)
      if (need_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t* const ib = inbox + lane;
      flags = ib[W_FLAGS * WAVE]; ld_frame = ib[W_FRAME * WAVE];
#pragma unroll
      for (int s = 0; s < NS; ++s) ld_pos[s] = ib[(W_POS + s) * WAVE];
#pragma unroll
      for (int i = 0; i < FW; ++i) x[i] = ib[(W_D + i) * WAVE];
      if (!a.hashed && live) ld_action = (int)ib[NWORDS * WAVE];
    }
    if (live) {  // every state word is requested up front: one memory round trip
      if constexpr (!PW) flags = st[W_FLAGS * bp];
      if (!PW && a.mode != 1) {
        ld_frame = st[W_FRAME * bp];
#pragma unroll
        for (int s = 0; s < NS; ++s) ld_pos[s] = st[(W_POS + s) * bp];
#pragma unroll
        for (int i = 0; i < FW; ++i) x[i] = st[(W_D + i) * bp];
        if (!a.hashed) ld_action = a.actions[env];
      }
      if (a.mode == 1) {
        do_reset = a.reset_mask ? a.reset_mask[env] != 0 : true;
        skip = !do_reset;
      } else if (flags & F_OVER) {
        do_reset = a.auto_reset != 0;
        skip = !do_reset;
        if (skip) {  // a finished environment left alone reports an empty step (pcx.h)
          out.reward[env] = 0; out.reward_set[env] = 0; out.discount[env] = 0.0f;
        }
      } else {
        action = a.hashed ? (int)(action_hash(a.seed, (uint64_t)(a.env_offset + env), (uint64_t)a.t) % (uint32_t)k.n_actions)
                          : ld_action;
        if (action < 0) action = PCX_ACTION_NONE;
      }
    }
    if (!skip) {
      int frame, row[NS], col[NS];
      uint32_t err;
      if (do_reset) {  // engine.py:520-581 its_showtime: fresh template state, frame 0 = play(None)
        frame = (int)k.init[W_FRAME];
        err = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) { row[s] = (int)(k.init[W_POS + s] & 0xFFFFu); col[s] = (int)(k.init[W_POS + s] >> 16); }
#pragma unroll
        for (int i = 0; i < FW; ++i) x[i] = P.init_curtain[i];
        action = PCX_ACTION_NONE;
      } else {
        frame = (int)ld_frame;
        err = (flags >> F_ERR_SHIFT) & 7u;
#pragma unroll
        for (int s = 0; s < NS; ++s) { row[s] = (int)(ld_pos[s] & 0xFFFFu); col[s] = (int)(ld_pos[s] >> 16); }
      }
      int reward = 0, reward_set = 0, over = 0;
      float discount = 1.0f;
      frame += 1;  // engine.py:698-735

      // ---- SlidingSprite.update (hello_world.py:117-123): wrap around the board
      if ((unsigned)action <= 3u) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          col[s] += (int)((k.dx[s] >> (2 * action)) & 3u) - 1;
          row[s] += (int)((k.dy[s] >> (2 * action)) & 3u) - 1;
          col[s] = col[s] < 0 ? col[s] + C : col[s] >= C ? col[s] - C : col[s];
          row[s] = row[s] < 0 ? row[s] + R : row[s] >= R ? row[s] - R : row[s];
        }
      }
      // ---- RollingDrape.update (:79-91) ----------------------------------------------
      if (action == 4) { over = 1; discount = 0.0f; }
      if ((unsigned)action <= 3u) {
        uint32_t t0[FW], t1[FW];
        if (action == 0) {         // np.roll(curtain, -1, axis=0): every row moves up, the top one to the bottom
          B::template shr<C>(x, t0); B::template shl<cells - C>(x, t1);
#pragma unroll
          for (int i = 0; i < FW; ++i) x[i] = (t0[i] | t1[i]) & B::valid(i);
        } else if (action == 1) {  // +1, axis=0
          B::template shl<C>(x, t0); B::template shr<cells - C>(x, t1);
#pragma unroll
          for (int i = 0; i < FW; ++i) x[i] = (t0[i] | t1[i]) & B::valid(i);
        } else if (action == 2) {  // -1, axis=1: every column moves left, the first one to the end of its row
          B::template shr<1>(x, t0); B::template shl<C - 1>(x, t1);
#pragma unroll
          for (int i = 0; i < FW; ++i) x[i] = ((t0[i] & ~B::col_mask(i, C - 1)) | (t1[i] & B::col_mask(i, C - 1))) & B::valid(i);
        } else {                   // +1, axis=1
          B::template shl<1>(x, t0); B::template shr<C - 1>(x, t1);
#pragma unroll
          for (int i = 0; i < FW; ++i) x[i] = ((t0[i] & ~B::col_mask(i, 0)) | (t1[i] & B::col_mask(i, 0))) & B::valid(i);
        }
        reward += 1; reward_set = 1;  // a point for moving
      }

      // ---- _apply_and_clear_plot + state write-back -----------------------------------
      st[W_FRAME * bp] = (uint32_t)frame;
      st[W_FLAGS * bp] = (over ? F_OVER : 0u) | ((err & 7u) << F_ERR_SHIFT);
      int32_t tw[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        st[(W_POS + s) * bp] = (uint32_t)row[s] | ((uint32_t)col[s] << 16);
        tw[s] = row[s] | (col[s] << 8) | ((int)((k.visible >> s) & 1) << 16) | ((int)do_reset << 24);
        P.track[(size_t)s * bp + env] = tw[s];
      }
#pragma unroll
      for (int i = 0; i < FW; ++i) st[(W_D + i) * bp] = x[i];
      if (a.export_curtains)
#pragma unroll
        for (int i = 0; i < FW; ++i) P.curtains[(size_t)i * bp + env] = x[i];
      const stream::CurtainSrc csrc{P.curtains, bp, FW, R, C};
      if (fc)  // fused croppers (after the export: a cropper may follow the '@' drape): the windows follow this step's positions (cropping.py:393-426)
        stream::move_fused_windows(fc, [&](int ti) {
          int32_t t = 0;
#pragma unroll
          for (int s = 0; s < NS; ++s) t = ti == s ? tw[s] : t;
          return t;
        }, frame == 0, env, lane, wcorner, &csrc);
      out.reward[env] = reward;
      out.reward_set[env] = (uint8_t)reward_set;
      out.discount[env] = discount;
      out.done[env] = (uint8_t)over;
      out.frame[env] = frame;
      out.error[env] = (uint8_t)err;

      // ---- render descriptors ------------------------------------------------------------
#pragma unroll
      for (int i = 0; i < FW; ++i) flat[lane * FWP + i] = x[i];
      int cellv[NS];
      uint32_t above[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) { cellv[s] = ((k.visible >> s) & 1) ? row[s] * C + col[s] : -1; above[s] = k.above[s]; }
      if constexpr (UNOCC)  // occlusion_in_layers=False: the layers are the raw masks (rendering.py:236-278)
        stream::snapshot_raw<NS, ND>(cellv, flat, FW, FWP, lane, lds + O_FLATRAW, reinterpret_cast<uint2*>(lds + O_SDESCRAW));
      if constexpr (CODES) {
        // rendering.py:98-179 as byte writes: the backdrop's codes with the drape's cells merged in, then every sprite nothing in front of it covers
        uint32_t* const cd = codes + lane * CP;
        const uint32_t* const bdc = lds + O_BDC;
        auto code_dword = [&](int q) {  // board dword q: the backdrop's codes, the drape's where its bits are set
          const uint32_t bits = (x[q >> 3] >> ((q & 7) * 4)) & 0xFu;
          const uint32_t m01 = (bits * 0x00204081u) & 0x01010101u;  // bit i -> byte i
          uint32_t hi8 = m01 << 8;
          asm("" : "+v"(hi8));  // (keeps (x << 8) - x from becoming a quarter-rate multiply)
          const uint32_t m = hi8 - m01;
          return (bdc[q] & ~m) | (k.drape_code4 & m);
        };
#pragma unroll
        for (int j = 0; j < (QW + 1) / 2; ++j) cd[j] = code_dword(2 * j) | (2 * j + 1 < QW ? code_dword(2 * j + 1) << 4 : 0u);
        uint32_t scode[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) scode[s] = k.sprite_code[s];
        stream::paint_sprites_nib<NS, ND>(cellv, above, flat, FWP, lane, reinterpret_cast<uint8_t*>(cd), scode);
      } else {
        stream::resolve_sprites<NS, ND>(cellv, above, flat, FWP, lane, sdesc);
      }
    }
    skipv[lane] = skip;
  }
  if constexpr (!PW) {
    __syncthreads();
    if (a.debug & 2) return;
  }

  stream::PlaneMap<NS, ND, NB> pm;
#pragma unroll
  for (int s = 0; s < NS; ++s) { pm.sprite_off[s] = k.sprite_off[s]; pm.sprite_ch4[s] = k.sprite_ch4[s]; }
  pm.drape_off[0] = k.drape_off; pm.drape_ch4[0] = k.drape_ch4;
  uint32_t bch4[NB > 0 ? NB : 1] = {};
#pragma unroll
  for (int b = 0; b < NB; ++b) { pm.bchar_off[b] = k.bchar_off[b]; bch4[b] = k.bchar_ch4[b]; }
  constexpr uint32_t env_stride = (uint32_t)(1 + L) * (uint32_t)pitch;
  stream::CodeMap<L> cmap;
#pragma unroll
  for (int i = 0; i < 4; ++i) cmap.chars[i] = k.code_chars[i];
  if constexpr (PW) {
    // the next unit is drawn and its state rows start travelling now, in front of this unit's plane stores
    const uint32_t next = wq.next(unit);
    if (next < wq.n) prefetch(next);
    const bool any_skip = __ballot(skipv[lane] != 0) != 0ull;
    if (!(a.debug & 2)) {
      const uint32_t sem = stream::lds_byte_address(lds + O_SEM);
      if (P.work.lock) stream::slot_acquire(sem, P.work.lock);
      if constexpr (CODES)
        stream::stream_codes<L, QW, 1, false, true>(cmap, out.planes + (size_t)env0 * env_stride, env_stride, codes, CP, skipv, lane, 0);
      else
        stream::stream_planes<NS, ND, NB, QW, 1, false, false, false>(pm, out.planes + (size_t)env0 * env_stride, env_stride, lds + O_BD, lds + O_BDM,
                                                                      flat, sdesc, skipv, FWP, lane, 0, epi, env0, nullptr, 0, nullptr, nullptr, lds);
      if (P.work.lock) stream::slot_release(sem);
    }
    need_wait = any_skip || (a.debug & ~16) != 0 || QW * (1 + L) < 64;  // fewer than 64 plane stores behind the prefetch: wait for it
    if (need_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unit = next;
  } else {
    if constexpr (CODES) {
      stream::stream_codes<L, QW, NWAVES, true, true>(cmap, out.planes + (size_t)env0 * env_stride, env_stride, codes, CP, skipv, lane, wave);
      break;
    }
    if (!(fc && fc->only))
      stream::stream_planes<NS, ND, NB, QW, NWAVES, EPI, UNOCC>(pm, out.planes + (size_t)env0 * env_stride, env_stride, lds + O_BD, lds + O_BDM,
                                                                 flat, sdesc, skipv, FWP, lane, wave, epi, env0, nullptr, 0, lds + O_FLATRAW,
                                                                 reinterpret_cast<const uint2*>(lds + O_SDESCRAW), lds);
    if (fc)
      stream::stream_windows<NS, ND, NB, QW, NWAVES, R, C>(fc, pm, bch4, env0, lds + O_BD, flat, sdesc, skipv, FWP, lane, wave, wcorner);
    break;
  }
  }  // units
  if constexpr (PW) wq.finish(lane);
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------

#define PCX_HW_SHAPES(X) X(13, 36) X(8, 33)

class HelloWorldBackend : public Backend {
 public:
  int init(const pcx_template& t, int64_t batch) override;
  int launch(const StepArgs& a, const pcx_buffers& out, hipStream_t s) override;
  int read_things(int64_t env0, int64_t n, pcx_sprite_state* sprites, uint8_t* curtains) override;
  int64_t bytes_per_step() const override { return 4 + 8 * (int64_t)NW_ + (int64_t)(1 + L_) * lay_.cells + 15; }
  int tuner_done() const override { return tuner_.done(); }
  const char* kernel_name() const override { return "pcx_hello_world_step"; }
  int launch_shape() const override { return last_shape_; }  // 0 a workgroup per group, 10 cooperative, 3 persistent workers (include/pcx.h)
  const int32_t* sprite_track() const override { return track_.ptr; }
  const uint32_t* curtain_bits() const override { return curtains_.ptr; }
  int ensure_curtains() override { return curtains_.ptr ? 0 : curtains_.alloc((size_t)lay_.FW * bpad_); }
  int curtain_words() const override { return lay_.FW; }
  int64_t batch_pad() const override { return bpad_; }
  void persistent_arrays(std::vector<std::pair<void*, size_t>>& out) override {  // pcx_engine_export_state
    out.push_back({state_.ptr, state_.count * sizeof(uint32_t)});
    out.push_back({track_.ptr, track_.count * sizeof(int32_t)});
  }
  int plane_pitch() const override { return lay_.pitch; }
  bool fused_window_features() const override { return true; }
  int set_fused_croppers(const crop::FusedCrops* fc) override {
    if (fc && fc->n > 0 && unoccluded_)  // (the windows derive their layers from the board they cut)
      return set_error(PCX_E_UNSUPPORTED, "hello_world backend: fused croppers need occluded layers");
    return fused_.set(fc, false, R_, C_);
  }
  size_t base_lds_bytes(bool codes = false) const {  // the kernel's own dynamic LDS (before padding / the channels-last exchange areas)
    return ((size_t)lay_.QW * (2 + NB) + WAVE * lay_.FWP + 2 + (codes ? WAVE * lay_.CPN : 2 * NS * WAVE) + WAVE + stream::WCORNER_WORDS +
            (unoccluded_ ? WAVE * lay_.FWP + 2 + 2 * NS * WAVE : 0)) * 4;
  }
  stream::EpilogueArgs* epilogue_args() override { return &epi_; }
  int set_epilogue(const pcx_epilogue_desc* d) override {  // include/pcx.h pcx_engine_set_epilogue
    if (d && unoccluded_) return set_error(PCX_E_UNSUPPORTED, "hello_world backend: the feature-array epilogue needs occluded layers");
    int sc[NS], dc = k_.drape_ch4 & 0xFF, bc[NB > 0 ? NB : 1] = {};
    for (int s = 0; s < NS; ++s) sc[s] = k_.sprite_ch4[s] & 0xFF;
    for (int b = 0; b < NB; ++b) bc[b] = k_.bchar_ch4[b] & 0xFF;
    if (!stream::fill_epilogue(epi_, d, lay_.cells, sc, NS, &dc, 1, bc, NB, 64 * 1024 - base_lds_bytes(), 4))
      return set_error(PCX_E_UNSUPPORTED, "hello_world backend: the channels-last epilogue needs rows*cols %% 4 == 0 and a stack whose exchange areas fit the LDS left");
    return 0;
  }

 private:
  stream::FusedCropsHolder fused_;
  Consts k_{};
  stream::EpilogueArgs epi_{};
  stream::Layout lay_;
  int R_ = 0, C_ = 0, L_ = 0, NW_ = 0;
  bool unoccluded_ = false;  // Engine(..., occlusion_in_layers=False)
  int64_t batch_ = 0, bpad_ = 0;
  int num_cus_ = 256;
  DevArray<uint32_t> tables_, initc_, state_, curtains_, work_ctr_;
  ShapeTuner tuner_;
  int last_shape_ = -1;
  DevArray<int32_t> track_;
};

int HelloWorldBackend::init(const pcx_template& t, int64_t batch) {
  Consts& k = k_;
  batch_ = batch;
  bpad_ = (batch + WAVE - 1) / WAVE * WAVE;
  if (const char* e = getenv("PCX_FORCE_GENERIC")) if (atoi(e)) return set_error(PCX_E_UNSUPPORTED, "hello_world backend: PCX_FORCE_GENERIC");
  if (t.n_directives) return set_error(PCX_E_UNSUPPORTED, "hello_world backend: no directives");
  unoccluded_ = !t.occlusion_in_layers;  // (no entity of this game reads a layer: only the layer planes differ)
  R_ = t.rows; C_ = t.cols; L_ = t.n_chars;
  bool shape_ok = false;
#define X(r, c) shape_ok |= R_ == r && C_ == c;
  PCX_HW_SHAPES(X)
#undef X
  if (!shape_ok || t.n_sprites != NS || t.n_drapes != ND || L_ != NS + ND + NB || t.n_groups != 1)
    return set_error(PCX_E_UNSUPPORTED, "hello_world backend: the shipped board and cast only");
  lay_.set(R_, C_);
  NW_ = W_D + lay_.FW;
  k.visible = 0;
  for (int s = 0; s < NS; ++s) {
    const pcx_sprite_desc& sd = t.sprites[s];
    if (sd.program != PCX_PROG_HW_SLIDING || sd.is_walker) return set_error(PCX_E_UNSUPPORTED, "hello_world backend: unexpected cast");
    k.dx[s] = (uint32_t)sd.param[0]; k.dy[s] = (uint32_t)sd.param[1];
    if (sd.visible) k.visible |= 1u << s;
  }
  const pcx_drape_desc& dd = t.drapes[0];
  if (dd.program != PCX_PROG_HW_ROLLING || dd.is_scrolly) return set_error(PCX_E_UNSUPPORTED, "hello_world backend: the drape must be the RollingDrape");
  int zpos[NS + ND];
  for (int z = 0; z < t.n_things; ++z) {
    int idx = -1;
    for (int s = 0; s < NS; ++s) if (t.sprites[s].ch == t.z_order[z]) idx = s;
    if (t.z_order[z] == dd.ch) idx = NS;
    if (idx < 0) return set_error(PCX_E_INVALID, "hello_world backend: z_order names an unknown character");
    zpos[idx] = z;
  }
  for (int s = 0; s < NS; ++s) {
    k.above[s] = 0;
    for (int j = 0; j < NS + ND; ++j) if (zpos[j] > zpos[s]) k.above[s] |= 1u << j;
  }
  k.n_actions = t.n_actions;
  auto layer_of = [&](int ch) { for (int i = 0; i < L_; ++i) if (t.chars[i] == ch) return i; return -1; };
  for (int s = 0; s < NS; ++s) {
    k.sprite_off[s] = (uint32_t)(1 + layer_of(t.sprites[s].ch)) * lay_.pitch;
    k.sprite_ch4[s] = t.sprites[s].ch * 0x01010101u;
  }
  k.drape_off = (uint32_t)(1 + layer_of(dd.ch)) * lay_.pitch;
  k.drape_ch4 = dd.ch * 0x01010101u;
  std::vector<uint32_t> tab((size_t)lay_.QW * (2 + NB), 0);
  memcpy(tab.data(), t.backdrop, lay_.cells);
  int nb = 0;
  for (int i = 0; i < L_; ++i) {
    const int ch = t.chars[i];
    bool thing = ch == dd.ch;
    for (int s = 0; s < NS; ++s) thing |= t.sprites[s].ch == ch;
    if (thing) continue;
    if (nb >= NB) return set_error(PCX_E_INVALID, "hello_world backend: inconsistent character set");
    k.bchar_off[nb] = (uint32_t)(1 + i) * lay_.pitch;
    k.bchar_ch4[nb] = (uint32_t)ch * 0x01010101u;
    uint8_t* m = reinterpret_cast<uint8_t*>(tab.data() + (size_t)lay_.QW * (1 + nb));
    for (int c = 0; c < lay_.cells; ++c) m[c] = t.backdrop[c] == ch;
    ++nb;
  }
  if (nb != NB) return set_error(PCX_E_INVALID, "hello_world backend: inconsistent character set");
  {  // owner codes (pcx_stream.h): L = 7 characters, a character's code is its layer plane; plane padding is selector 12 (zero everywhere)
    static_assert(NS + ND + NB <= 8, "one v_perm_b32 per plane");
    for (int s = 0; s < NS; ++s) k.sprite_code[s] = (uint32_t)layer_of(t.sprites[s].ch);
    k.drape_code4 = (uint32_t)layer_of(dd.ch) * 0x01010101u;
    memset(k.code_chars, 0, sizeof k.code_chars);
    for (int i = 0; i < L_; ++i) k.code_chars[i >> 2] |= (uint32_t)t.chars[i] << (8 * (i & 3));
    uint8_t* bdc = reinterpret_cast<uint8_t*>(tab.data() + (size_t)lay_.QW * (1 + NB));
    for (int c = 0; c < lay_.pitch; ++c) bdc[c] = c < lay_.cells ? (uint8_t)layer_of(t.backdrop[c]) : (uint8_t)12;
  }
  std::vector<uint32_t> initc(lay_.FW, 0);
  for (int c = 0; c < lay_.cells; ++c) if (dd.curtain[c]) initc[c >> 5] |= 1u << (c & 31);
  memset(k.init, 0, sizeof k.init);
  k.init[W_FRAME] = (uint32_t)-1;
  for (int s = 0; s < NS; ++s) k.init[W_POS + s] = (uint32_t)t.sprites[s].row | ((uint32_t)t.sprites[s].col << 16);
  {
    int sc[NS], dc = dd.ch, bc[NB] = {0, 0};
    for (int s = 0; s < NS; ++s) sc[s] = t.sprites[s].ch;
    stream::fill_epilogue(epi_, nullptr, lay_.cells, sc, NS, &dc, 1, bc, NB);
  }
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      num_cus_ = prop.multiProcessorCount;
  }
  int rc;
  if ((rc = tables_.upload(tab))) return rc;
  if ((rc = initc_.upload(initc))) return rc;
  if ((rc = state_.alloc((size_t)NW_ * bpad_))) return rc;
  if ((rc = track_.alloc((size_t)NS * bpad_))) return rc;
  // the persistent workers' ticket shards and done-count (pcx_stream.h WorkQueue), 64 bytes apart: allocated and zeroed HERE,
  // with the device synchronisation that follows engine creation -- a first launch on a non-blocking stream (or under
  // stream capture) must not race a hipMemset on the null stream
  if ((rc = work_ctr_.alloc(16 * 9))) return rc;
  return 0;
}

int HelloWorldBackend::launch(const StepArgs& a, const pcx_buffers& out, hipStream_t s) {
  if (a.n_steps != 1) return set_error(PCX_E_INVALID, "hello_world backend: one step per launch");
  if (a.export_curtains) { int rc = ensure_curtains(); if (rc) return rc; }
  Ptrs P{tables_.ptr, initc_.ptr, state_.ptr, track_.ptr, curtains_.ptr, batch_, bpad_, {}};
  const int64_t groups = bpad_ / WAVE;
  int coop_below = 5, waves_per_cu = 6;
  if (fused_.only) coop_below = 17;  // windows only: little to stream per group, latency-bound (see pcx_better_scrolly.hip)
  if (const char* e = getenv("PCX_COOP_BELOW")) coop_below = atoi(e);
  if (const char* e = getenv("PCX_WAVES_PER_CU")) waves_per_cu = atoi(e);
  const bool coop = groups < (int64_t)num_cus_ * coop_below;
  size_t lds = base_lds_bytes();
  const stream::EpilogueArgs epi_ = stream::with_hwc_scratch(this->epi_, lds, coop ? 4 : 1);  // (channels-last epilogue: its exchange area behind the kernel's own LDS)
  if (!coop && waves_per_cu > 0) {
    size_t want = ((size_t)(160 * 1024) / (size_t)waves_per_cu) & ~(size_t)255;
    if (want > 64 * 1024) want = 64 * 1024;
    if (want > lds) lds = want;
  }
  bool launched = false;
  // (round 6) owner codes: plain steps -- no epilogue, no fused croppers, occluded layers; PCX_HW_CODES=0: the mask loop
  bool codes = !epi_.out && !fused_.on && !unoccluded_;
  if (const char* e = getenv("PCX_HW_CODES")) codes = codes && atoi(e) != 0;
  if (codes) {
    lds = base_lds_bytes(true);
    if (!coop && waves_per_cu > 0) {
      size_t want = ((size_t)(160 * 1024) / (size_t)waves_per_cu) & ~(size_t)255;
      if (want > 64 * 1024) want = 64 * 1024;
      if (want > lds) lds = want;
    }
  }
  // (round 5) persistent workers for plain steps, as in pcx_warehouse.hip (PCX_HW_PW=0: the round-2 shape)
  bool pw = !coop && !epi_.out && !fused_.on && !unoccluded_ && a.mode == 0 && !a.export_curtains && (a.debug & ~16) == 0;
  if (const char* e = getenv("PCX_HW_PW")) pw = pw && atoi(e) != 0;
  if (pw) {
    // Measured (profiles/r05_hello_world_workers_sweep.txt; round-2 shape 0.2256 / 0.8428 ms at 262,144 / 1,048,576 environments):
    // four single-worker workgroups per CU -- the round-2 residency made persistent, state prefetched -- 0.2128 / 0.7900;
    // workgroups of several workers with streaming slots lose here (five workers, four slots: 0.2639 / 0.8121).
    // The alternatives are measured on the engine's own first launches (ShapeTuner, pcx_internal.h), this one first.
    struct Cand { int workers, per_cu, lock; };
    static const Cand cands_m[ShapeTuner::NC] = {{1, 4, 0}, {2, 4, 1}, {2, 3, 1}, {5, 1, 4}};
    // (round 6, the owner-code loop: one streaming wave per workgroup is enough -- profiles/r06_hello_world_codes_sweep.txt)
    static const Cand cands_k[ShapeTuner::NC] = {{1, 4, 0}, {2, 2, 1}, {1, 3, 0}, {2, 3, 1}};
    const Cand* const cands = codes ? cands_k : cands_m;
    const bool knobs = getenv("PCX_HW_WORKERS") || getenv("PCX_HW_PER_CU") || getenv("PCX_HW_LOCK") || getenv("PCX_HW_GRID") ||
                       (getenv("PCX_HW_TUNE") && atoi(getenv("PCX_HW_TUNE")) == 0);
    if (knobs) tuner_.off = true;
    const Cand& cand = cands[tuner_.pick(a, s)];
    int workers = cand.workers, per_cu = cand.per_cu, lock = cand.lock;
    if (const char* e = getenv("PCX_HW_WORKERS")) { const int v = atoi(e); if (v >= 1 && v <= 8) workers = v; }
    if (const char* e = getenv("PCX_HW_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 8) per_cu = v; }
    if (const char* e = getenv("PCX_HW_LOCK")) lock = atoi(e);
    int dynamic = groups >= (int64_t)num_cus_ * 24;
    if (const char* e = getenv("PCX_HW_DYNAMIC")) dynamic = atoi(e) != 0;
    const size_t tab_words = (size_t)lay_.QW * (2 + NB);
    const size_t o_sdesc = (tab_words + (size_t)WAVE * lay_.FWP + 1) & ~(size_t)1;
    const size_t region = o_sdesc + (codes ? WAVE * lay_.CPN : 2 * NS * WAVE) + WAVE - tab_words;  // flat, sdesc / codes, skip: the kernel's O_WCORNER - O_FLAT
    const size_t w_words = (region + (size_t)(NW_ + 1) * WAVE + 1) & ~(size_t)1;
    size_t lds_pw = (tab_words + 2 + (size_t)workers * w_words) * 4;
    while (workers > 1 && lds_pw > 64 * 1024) { --workers; lds_pw = (tab_words + 2 + (size_t)workers * w_words) * 4; }
    int64_t wgs = (int64_t)num_cus_ * per_cu;
    if (const char* e = getenv("PCX_HW_GRID")) { const int v = atoi(e); if (v >= 1) wgs = v; }  // (tests: few workgroups, many units each)
    const int64_t want = (groups + workers - 1) / workers;
    if (wgs > want) wgs = want;
    if (wgs * workers >= groups) dynamic = 0;  // every unit is some worker's first
    P.work.ctr = work_ctr_.ptr;
    P.work.n_units = (uint32_t)groups;
    P.work.dynamic = dynamic;
    P.work.lock = lock;
#define X(r, c)                                                                                                  \
  if (!launched && R_ == r && C_ == c) {                                                                         \
    if (codes) hipLaunchKernelGGL((pcx_hello_world_step<r, c, 1, false, false, true, true>), dim3((unsigned)wgs), dim3(workers * WAVE), lds_pw, s, k_, P, a, out, epi_, fused_.ptr()); \
    else hipLaunchKernelGGL((pcx_hello_world_step<r, c, 1, false, false, true>), dim3((unsigned)wgs), dim3(workers * WAVE), lds_pw, s, k_, P, a, out, epi_, fused_.ptr()); \
    launched = true;                                                                                             \
  }
    PCX_HW_SHAPES(X)
#undef X
    if (launched) tuner_.launched(s);
  }
  last_shape_ = launched ? 3 : coop ? 10 : 0;
#define X(r, c)                                                                                                  \
  if (!launched && R_ == r && C_ == c) {                                                                         \
    if (unoccluded_ && coop) hipLaunchKernelGGL((pcx_hello_world_step<r, c, 4, false, true>), dim3((unsigned)groups), dim3(4 * WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr()); \
    else if (unoccluded_) hipLaunchKernelGGL((pcx_hello_world_step<r, c, 1, false, true>), dim3((unsigned)groups), dim3(WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr());       \
    else if (epi_.out && coop) hipLaunchKernelGGL((pcx_hello_world_step<r, c, 4, true>), dim3((unsigned)groups), dim3(4 * WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr()); \
    else if (epi_.out) hipLaunchKernelGGL((pcx_hello_world_step<r, c, 1, true>), dim3((unsigned)groups), dim3(WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr());       \
    else if (coop && codes) hipLaunchKernelGGL((pcx_hello_world_step<r, c, 4, false, false, false, true>), dim3((unsigned)groups), dim3(4 * WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr()); \
    else if (codes) hipLaunchKernelGGL((pcx_hello_world_step<r, c, 1, false, false, false, true>), dim3((unsigned)groups), dim3(WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr()); \
    else if (coop) hipLaunchKernelGGL((pcx_hello_world_step<r, c, 4>), dim3((unsigned)groups), dim3(4 * WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr()); \
    else hipLaunchKernelGGL((pcx_hello_world_step<r, c, 1>), dim3((unsigned)groups), dim3(WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr());          \
    launched = true;                                                                                             \
  }
  PCX_HW_SHAPES(X)
#undef X
  if (!launched) return set_error(PCX_E_UNSUPPORTED, "hello_world backend: no instance");
  PCX_HIP(hipGetLastError());
  return 0;
}

int HelloWorldBackend::read_things(int64_t env0, int64_t n, pcx_sprite_state* sprites, uint8_t* curtains) {
  std::vector<uint32_t> st((size_t)NW_ * n);
  PCX_HIP(hipDeviceSynchronize());
  for (int w = 0; w < NW_; ++w)
    PCX_HIP(hipMemcpy(st.data() + (size_t)w * n, state_.ptr + (size_t)w * bpad_ + env0, n * 4, hipMemcpyDeviceToHost));
  auto word = [&](int w, int64_t i) { return st[(size_t)w * n + i]; };
  for (int64_t i = 0; i < n; ++i) {
    if (sprites)
      for (int s = 0; s < NS; ++s) {
        pcx_sprite_state& o = sprites[i * NS + s];
        memset(&o, 0, sizeof o);
        o.row = o.vrow = (int)(word(W_POS + s, i) & 0xFFFFu);
        o.col = o.vcol = (int)(word(W_POS + s, i) >> 16);
        o.visible = (k_.visible >> s) & 1;
      }
    if (curtains)
      for (int c = 0; c < lay_.cells; ++c)
        curtains[(size_t)i * lay_.cells + c] = (word(W_D + (c >> 5), i) >> (c & 31)) & 1;
  }
  return 0;
}

}  // namespace hw

Backend* make_hello_world_backend() { return new hw::HelloWorldBackend(); }

}  // namespace pcx
