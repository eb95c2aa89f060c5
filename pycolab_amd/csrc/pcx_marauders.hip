// pcx_marauders.hip -- hand-written fused step kernel for
// extraterrestrial_marauders (reference:
// pycolab/examples/extraterrestrial_marauders.py:104-256 driven by
// engine.py:583-847 and prefab_parts/sprites.py MazeWalker).  gfx950 only.
//
// One launch = one Engine.play() of every environment of the batch.  Same shape
// as pcx_scrolly_maze.hip: a group of 64 consecutive
// environments per workgroup; logic phase lane == environment; render phase =
// the shared streaming loop of pcx_stream.h.
//
// The game has ONE update group, so every entity sees the repaint the step
// started with (engine.py:735 runs once, after the group), and the two
// curtains -- the only bulky state -- are kept as flat 624-bit vectors (bit
// r * 39 + c), the same words in HBM (SoA over the batch), in registers while
// the marauders march, and in LDS for the bolts' hit tests and the streaming
// phase:
//   * np.roll(curtain, 1, axis=0) is a 39-bit rotate of the 624-bit vector and
//     np.roll(curtain, +-1, axis=1) two shifts and two constant masks, all on
//     twenty registers with compile-time indices;
//   * a bolt's hit (`bolts & self.curtain`) is one LDS bit test at its cell;
//   * the marauders' return fire (`np.random.choice` over the columns that show
//     a marauder, :246-248) folds the occluded layer's rows with funnel shifts
//     and picks the n-th set bit by bisection; it reads the curtain the step
//     started with, so it is evaluated first;
//   * 'bunker_hitters' / 'marauder_hitters' / 'last_player_shot' /
//     'last_marauder_shot' are written and read inside one frame: registers.
// Other boards, occlusion_in_layers=False or a different cast are stepped by
// the table-driven kernel (pcx_generic.hip); the engine falls back to it.

#include "pcx_internal.h"
#include "pcx_stream.h"

#include <cstdlib>
#include <cstring>

namespace pcx {
namespace em {

using stream::WAVE;
constexpr int NS = 7;   // P, four upward bolts, two downward bolts (template order)
constexpr int NUP = 4, NDOWN = 2;
constexpr int ND = 2;   // slot 0 = bunkers 'B', slot 1 = marauders 'X'
constexpr int NB = 1;   // backdrop-only characters
constexpr int R = 16, C = 39, cells = R * C, pitch = cells, QW = cells / 4, FW = (cells + 31) / 32, FWP = FW | 1;
constexpr int L = NS + ND + NB;
static_assert(cells % 4 == 0, "board planes are whole dwords");

// State words (uint32 [NW][batch_padded]).
enum : int { W_FRAME = 0, W_FLAGS, W_RNG, W_POS, W_B = W_POS + NS, W_X = W_B + FW, NW = W_X + FW };
constexpr uint32_t F_OVER = 1u, F_ERR_SHIFT = 1;
constexpr int F_DX_SHIFT = 8;   // MarauderDrape._dx + 1, 2 bits
constexpr int F_SF_SHIFT = 12;  // per sprite: visible, prior_visible

struct Consts {
  int32_t n_actions;
  uint32_t confined;        // bit s
  uint32_t above[NS];       // bit j: sprite j in front of sprite s; bit NS + d: drape slot d in front
  uint32_t init[W_B];       // initial scalar words
  uint32_t sprite_off[NS], sprite_ch4[NS], drape_off[ND], drape_ch4[ND], bchar_off[NB], bchar_ch4[NB];
  uint32_t seed_lo, seed_hi, envoff_lo, envoff_hi;  // np.random.choice stand-in (shared with the oracle)
  int32_t drape_slot_tmpl[ND];  // template drape index of slot d
};

struct Ptrs {
  const uint32_t* tables;   // staged into LDS: backdrop4 [QW], bdmask [NB][QW]
  const uint32_t* init_curtains;  // [ND][FW] initial curtains
  uint32_t* state;          // [NW][bpad]
  int32_t* track;           // [NS][bpad]
  uint32_t* curtains;       // [ND][FW][bpad] raw curtains, template drape order (export_curtains)
  int64_t batch, bpad;
};

__device__ __forceinline__ uint32_t action_hash(uint64_t seed, uint64_t env, uint64_t t) {
  uint64_t x = seed ^ (env * 0x9E3779B97F4A7C15ull) ^ (t * 0xBF58476D1CE4E5B9ull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}
constexpr uint64_t EM_RNG_SALT = 0x4D415241554445ull;
__device__ __forceinline__ uint32_t pack_pos(int r, int c) { return ((uint32_t)r & 0xFFFFu) | ((uint32_t)c << 16); }
__device__ __forceinline__ int pos_r(uint32_t w) { return (int)(int16_t)(w & 0xFFFFu); }
__device__ __forceinline__ int pos_c(uint32_t w) { return (int)(int16_t)(w >> 16); }

// ---- the 624-bit curtain in twenty registers (compile-time indices only) -------
constexpr uint32_t col_mask_word(int i, int col) {  // bits of word i that are column `col` of some row
  uint32_t m = 0;
  for (int b = 0; b < 32; ++b) {
    const int bit = 32 * i + b;
    if (bit < cells && bit % C == col) m |= 1u << b;
  }
  return m;
}
constexpr uint32_t row_mask_word(int i, int row) {  // bits of word i that belong to row `row`
  uint32_t m = 0;
  for (int b = 0; b < 32; ++b) {
    const int bit = 32 * i + b;
    if (bit < cells && bit / C == row) m |= 1u << b;
  }
  return m;
}
constexpr uint32_t valid_mask_word(int i) {
  return 32 * i + 32 <= cells ? 0xFFFFFFFFu : 32 * i >= cells ? 0u : ((1u << (cells - 32 * i)) - 1u);
}
// out = x << S (towards higher cell indices), zero fill
template <int S>
__device__ __forceinline__ void shl_bits(const uint32_t (&x)[FW], uint32_t (&out)[FW]) {
  constexpr int ws = S / 32, bs = S % 32;
#pragma unroll
  for (int i = 0; i < FW; ++i) {
    const uint32_t hi = i - ws >= 0 ? x[i - ws >= 0 ? i - ws : 0] : 0u;
    const uint32_t lo = i - ws - 1 >= 0 ? x[i - ws - 1 >= 0 ? i - ws - 1 : 0] : 0u;
    out[i] = bs ? (hi << bs) | (lo >> ((32 - bs) & 31)) : hi;
  }
}
// out = x >> S, zero fill
template <int S>
__device__ __forceinline__ void shr_bits(const uint32_t (&x)[FW], uint32_t (&out)[FW]) {
  constexpr int ws = S / 32, bs = S % 32;
#pragma unroll
  for (int i = 0; i < FW; ++i) {
    const uint32_t lo = i + ws < FW ? x[i + ws < FW ? i + ws : 0] : 0u;
    const uint32_t hi = i + ws + 1 < FW ? x[i + ws + 1 < FW ? i + ws + 1 : 0] : 0u;
    out[i] = bs ? (lo >> bs) | (hi << ((32 - bs) & 31)) : lo;
  }
}
// n-th (0-based) set bit of w; n < popcount(w)
__device__ __forceinline__ int select_bit(uint32_t w, int n) {
  int pos = 0, c;
  c = __popc(w & 0xFFFFu); if (n >= c) { n -= c; pos += 16; w >>= 16; }
  c = __popc(w & 0xFFu);   if (n >= c) { n -= c; pos += 8;  w >>= 8; }
  c = __popc(w & 0xFu);    if (n >= c) { n -= c; pos += 4;  w >>= 4; }
  c = __popc(w & 0x3u);    if (n >= c) { n -= c; pos += 2;  w >>= 2; }
  c = (int)(w & 1u);       if (n >= c) pos += 1;
  return pos;
}

// NWAVES waves per workgroup: wave 0 steps the group, all share the render loop.
template <int NWAVES, bool EPI = false>
__global__ __launch_bounds__(NWAVES* WAVE) void pcx_marauders_step(const Consts k, const Ptrs P, const StepArgs a,
                                                                    const pcx_buffers out, const stream::EpilogueArgs epi,
                                                                    const crop::FusedCrops* fc) {
  extern __shared__ uint32_t lds[];
  constexpr int O_BD = 0, O_BDM = O_BD + QW, O_TAB_END = O_BDM + NB * QW;
  constexpr int O_FLAT = O_TAB_END, O_XS = O_FLAT + ND * WAVE * FWP, O_SDESC = (O_XS + WAVE * FWP + 1) & ~1,
                O_SKIP = O_SDESC + 2 * NS * WAVE;
  constexpr int O_WCORNER = O_SKIP + WAVE;  // fused croppers' window corners
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < O_TAB_END; i += NWAVES * WAVE) lds[i] = P.tables[i];
  uint32_t* const flat = lds + O_FLAT;  // [ND][64][FWP]
  uint32_t* const fb = flat + lane * FWP;               // this lane's bunker words
  uint32_t* const fx = flat + (WAVE + lane) * FWP;      // this lane's marauder words
  uint32_t* const xs = lds + O_XS + lane * FWP;         // scratch: the marauders' layer of the last repaint
  uint2* const sdesc = reinterpret_cast<uint2*>(lds + O_SDESC);
  uint32_t* const skipv = lds + O_SKIP;
  uint32_t* const wcorner = lds + O_WCORNER;
  __syncthreads();

  const int64_t env0 = (int64_t)blockIdx.x * WAVE;
  if (wave == 0) {
    // ---- logic phase: lane == environment -------------------------------------
    const int64_t env = env0 + lane, bp = P.bpad;
    const bool live = env < P.batch;
    uint32_t* const st = P.state + env;
    uint32_t flags = 0, ld_frame = 0, ld_rng = 0, ld_pos[NS] = {}, xb[FW], xx[FW];
    int ld_action = PCX_ACTION_NONE;
    bool skip = !live, do_reset = false;
    int action = PCX_ACTION_NONE;
#pragma unroll
    for (int i = 0; i < FW; ++i) xb[i] = xx[i] = 0;
    if (live) {  // every state word is requested up front: one memory round trip
      flags = st[W_FLAGS * bp];
      if (a.mode != 1) {
        ld_frame = st[W_FRAME * bp];
        ld_rng = st[W_RNG * bp];
#pragma unroll
        for (int s = 0; s < NS; ++s) ld_pos[s] = st[(W_POS + s) * bp];
#pragma unroll
        for (int i = 0; i < FW; ++i) { xb[i] = st[(W_B + i) * bp]; xx[i] = st[(W_X + i) * bp]; }
        if (!a.hashed) ld_action = a.actions[env];
      } else {
        ld_rng = st[W_RNG * bp];  // the draw counter survives resets
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
      int frame, dx;
      uint32_t err, sflags, draws = ld_rng;
      int vr[NS], vc[NS], vis[NS], prior[NS];
      if (do_reset) {  // engine.py:520-581 its_showtime: fresh template state, frame 0 = play(None)
        frame = (int)k.init[W_FRAME];
        dx = (int)((k.init[W_FLAGS] >> F_DX_SHIFT) & 3u) - 1;
        sflags = k.init[W_FLAGS] >> F_SF_SHIFT;
        err = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) { vr[s] = pos_r(k.init[W_POS + s]); vc[s] = pos_c(k.init[W_POS + s]); }
#pragma unroll
        for (int i = 0; i < FW; ++i) { xb[i] = P.init_curtains[i]; xx[i] = P.init_curtains[FW + i]; }
        action = PCX_ACTION_NONE;
      } else {
        frame = (int)ld_frame;
        dx = (int)((flags >> F_DX_SHIFT) & 3u) - 1;
        sflags = flags >> F_SF_SHIFT;
        err = (flags >> F_ERR_SHIFT) & 7u;
#pragma unroll
        for (int s = 0; s < NS; ++s) { vr[s] = pos_r(ld_pos[s]); vc[s] = pos_c(ld_pos[s]); }
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) { vis[s] = (sflags >> (2 * s)) & 1; prior[s] = (sflags >> (2 * s + 1)) & 1; }
      int reward = 0, reward_set = 0, over = 0;
      float discount = 1.0f;
      frame += 1;  // engine.py:698-735

      auto on_board = [](int r, int c) { return (unsigned)r < (unsigned)R && (unsigned)c < (unsigned)C; };
      auto true_cell = [&](int r, int c) { return on_board(r, c) ? r * C + c : 0; };  // Sprite.position
      auto teleport = [&](int s, int nr, int nc) {  // sprites.py:315-352
        const bool old_on = on_board(vr[s], vc[s]), new_on = on_board(nr, nc);
        if (old_on && !new_on) { prior[s] = vis[s]; vis[s] = 0; }
        if (!old_on && new_on) vis[s] = prior[s];
        vr[s] = nr; vc[s] = nc;
      };
      // sprites.py:356-389 _move with impassable == '': only the board's edge can block
      auto move = [&](int s, int dr, int dc) {
        const int nr = vr[s] + dr, nc = vc[s] + dc;
        if (!on_board(nr, nc) && ((k.confined >> s) & 1)) return;
        teleport(s, nr, nc);
      };

      // what the last repaint showed: every sprite's cell; a bolt's layer is its
      // cell unless a bolt in front of it shares it (rendering.py:177-179)
      int cell0[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) cell0[s] = vis[s] ? true_cell(vr[s], vc[s]) : -1;
      bool ontop[NS];
#pragma unroll
      for (int s = 1; s < NS; ++s) {
        ontop[s] = cell0[s] >= 0;
#pragma unroll
        for (int j = 1; j < NS; ++j)
          if (j != s && ((k.above[s] >> j) & 1)) ontop[s] = ontop[s] && cell0[j] != cell0[s];
      }
      ontop[0] = false;
#pragma unroll
      for (int i = 0; i < FW; ++i) { fb[i] = xb[i]; fx[i] = xx[i]; }

      // ---- DownwardLaserBoltSprite._fire target (:244-250), from the curtain and
      // the layers the step started with.  At most one bolt fires per frame
      // ('last_marauder_shot'): the first of y, z that is hidden.
      bool fire_empty = false;
      int fire_row = 0, fire_col = 0;
      if (!vis[NS - 2] || !vis[NS - 1]) {
#pragma unroll
        for (int i = 0; i < FW; ++i) xs[i] = xx[i];
#pragma unroll
        for (int s = 1; s < NS; ++s)  // layers['X'] loses the cells a bolt covers (all bolts are in front of it)
          if (cell0[s] >= 0) xs[cell0[s] >> 5] &= ~(1u << (cell0[s] & 31));
        uint32_t ly[FW];
#pragma unroll
        for (int i = 0; i < FW; ++i) ly[i] = xs[i];
        uint32_t cols_lo = 0, cols_hi = 0;  // layers['X'].sum(axis=0) != 0, 39 columns
#pragma unroll
        for (int r = 0; r < R; ++r) {
          constexpr int dummy = 0; (void)dummy;
          const int w = (r * C) / 32, sh = (r * C) % 32;
          const uint32_t w0 = ly[w], w1 = w + 1 < FW ? ly[w + 1 < FW ? w + 1 : 0] : 0u, w2 = w + 2 < FW ? ly[w + 2 < FW ? w + 2 : 0] : 0u;
          cols_lo |= sh ? (w0 >> sh) | (w1 << ((32 - sh) & 31)) : w0;
          cols_hi |= (sh ? (w1 >> sh) | (w2 << ((32 - sh) & 31)) : w1) & ((1u << (C - 32)) - 1u);
        }
        const int n_lo = __popc(cols_lo), n = n_lo + __popc(cols_hi);
        if (n == 0) {
          fire_empty = true;  // np.random.choice([]) raises
        } else {
          const uint64_t seed = ((uint64_t)k.seed_lo | ((uint64_t)k.seed_hi << 32)) ^ EM_RNG_SALT;
          const uint64_t genv = ((uint64_t)k.envoff_lo | ((uint64_t)k.envoff_hi << 32)) + (uint64_t)env;
          const int pick = (int)(action_hash(seed, genv, (uint64_t)draws) % (uint32_t)n);
          fire_col = pick < n_lo ? select_bit(cols_lo, pick) : 32 + select_bit(cols_hi, pick - n_lo);
#pragma unroll
          for (int r = 0; r < R; ++r) {  // the lowest marauder of that column (:248)
            const int b = r * C + fire_col;
            if ((xs[b >> 5] >> (b & 31)) & 1) fire_row = r;
          }
        }
      }

      // ---- PlayerSprite.update (:178-186) ------------------------------------------
      if (action == 0) move(0, 0, -1);
      else if (action == 1) move(0, 0, 1);
      else if (action == 4) { over = 1; discount = 0.0f; }

      // ---- BunkerDrape.update (:113-120) and the erosion half of MarauderDrape.update (:141-147)
      uint32_t hit_b = 0, hit_x = 0;  // 'bunker_hitters' / 'marauder_hitters': bit per sprite
      {
        int hits = 0;
#pragma unroll
        for (int s = 1; s < NS; ++s)
          if (ontop[s]) {
            const uint32_t w = fb[cell0[s] >> 5], bit = 1u << (cell0[s] & 31);
            if (w & bit) { fb[cell0[s] >> 5] = w & ~bit; ++hits; hit_b |= 1u << s; }
          }
        reward -= hits;
        reward_set = 1;  // add_reward is called every frame: the reward is never None
        hits = 0;
#pragma unroll
        for (int s = 1; s <= NUP; ++s)
          if (ontop[s]) {
            const uint32_t w = fx[cell0[s] >> 5], bit = 1u << (cell0[s] & 31);
            if (w & bit) { fx[cell0[s] >> 5] = w & ~bit; ++hits; hit_x |= 1u << s; }
          }
        reward += 10 * hits;
      }
#pragma unroll
      for (int i = 0; i < FW; ++i) { xb[i] = fb[i]; xx[i] = fx[i]; }

      // ---- the marching half of MarauderDrape.update (:149-163) ---------------------
      {
        int total = 0;
        uint32_t row10 = 0, edge = 0;
#pragma unroll
        for (int i = 0; i < FW; ++i) {
          total += __popc(xx[i]);
          row10 |= xx[i] & row_mask_word(i, 10);
          edge |= xx[i] & (col_mask_word(i, 0) | col_mask_word(i, C - 1));
        }
        if (total == 0 || row10) {
          over = 1; discount = 0.0f;
        } else {
          int period = (total - 1) / 8;  // total // 8.0000001
          if (period < 1) period = 1;
          if ((uint32_t)frame % (uint32_t)period == 0) {
            uint32_t t0[FW], t1[FW];
            if (edge) {  // reverse and descend one row: np.roll(curtain, 1, axis=0)
              dx = -dx;
              shl_bits<C>(xx, t0);
              shr_bits<cells - C>(xx, t1);
#pragma unroll
              for (int i = 0; i < FW; ++i) xx[i] = (t0[i] | t1[i]) & valid_mask_word(i);
            }
            if (dx > 0) {  // np.roll(curtain, +1, axis=1)
              shl_bits<1>(xx, t0);
              shr_bits<C - 1>(xx, t1);
#pragma unroll
              for (int i = 0; i < FW; ++i) xx[i] = ((t0[i] & ~col_mask_word(i, 0)) | (t1[i] & col_mask_word(i, 0))) & valid_mask_word(i);
            } else {       // np.roll(curtain, -1, axis=1)
              shr_bits<1>(xx, t0);
              shl_bits<C - 1>(xx, t1);
#pragma unroll
              for (int i = 0; i < FW; ++i) xx[i] = ((t0[i] & ~col_mask_word(i, C - 1)) | (t1[i] & col_mask_word(i, C - 1))) & valid_mask_word(i);
            }
          }
        }
      }

      // ---- UpwardLaserBoltSprite.update (:198-220) ------------------------------------
      bool player_fired = false, marauder_fired = false;  // 'last_player_shot' / 'last_marauder_shot' == frame
#pragma unroll
      for (int s = 1; s <= NUP; ++s) {
        if (vis[s]) {
          if (((hit_b | hit_x) >> s) & 1) teleport(s, -1, -1);
          else move(s, -1, 0);
        } else if (action == 2 && !player_fired) {
          player_fired = true;
          const bool on = on_board(vr[0], vc[0]);
          teleport(s, (on ? vr[0] : 0) - 1, on ? vc[0] : 0);
        }
      }
      // ---- DownwardLaserBoltSprite.update (:232-256) ----------------------------------
#pragma unroll
      for (int s = NUP + 1; s < NS; ++s) {
        if (vis[s]) {
          if ((hit_b >> s) & 1) { teleport(s, -1, -1); continue; }
          if (true_cell(vr[s], vc[s]) == true_cell(vr[0], vc[0])) { over = 1; discount = 0.0f; }
          move(s, 1, 0);
        } else if (!marauder_fired) {
          marauder_fired = true;
          if (fire_empty) { err |= ERR_INDEX; continue; }
          ++draws;
          teleport(s, fire_row + 1, fire_col);
        }
      }

      // ---- _apply_and_clear_plot (engine.py:761-847) + state write-back ---------------
      st[W_FRAME * bp] = (uint32_t)frame;
      uint32_t sf = 0;
      int32_t tw[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        st[(W_POS + s) * bp] = pack_pos(vr[s], vc[s]);
        sf |= ((uint32_t)vis[s] | ((uint32_t)prior[s] << 1)) << (2 * s);
        const bool on = on_board(vr[s], vc[s]);
        tw[s] = (on ? vr[s] : 0) | ((on ? vc[s] : 0) << 8) | (vis[s] << 16) | ((int)do_reset << 24);
        P.track[(size_t)s * bp + env] = tw[s];
      }
      st[W_FLAGS * bp] = (over ? F_OVER : 0u) | ((err & 7u) << F_ERR_SHIFT) | ((uint32_t)(dx + 1) << F_DX_SHIFT) | (sf << F_SF_SHIFT);
      st[W_RNG * bp] = draws;
#pragma unroll
      for (int i = 0; i < FW; ++i) { st[(W_B + i) * bp] = xb[i]; st[(W_X + i) * bp] = xx[i]; }
      if (a.export_curtains) {
#pragma unroll
        for (int i = 0; i < FW; ++i) {
          P.curtains[((size_t)k.drape_slot_tmpl[0] * FW + i) * bp + env] = xb[i];
          P.curtains[((size_t)k.drape_slot_tmpl[1] * FW + i) * bp + env] = xx[i];
        }
      }
      const stream::CurtainSrc csrc{P.curtains, bp, FW, R, C};
      if (fc)  // fused croppers (after the export: a cropper may follow the marauders or the bunkers): the windows follow this step's positions (cropping.py:393-426)
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

      // ---- render descriptors: marauders over bunkers, then the sprites ----------------
#pragma unroll
      for (int i = 0; i < FW; ++i) { fb[i] = xb[i] & ~xx[i]; fx[i] = xx[i]; }
      int cellv[NS];
      uint32_t above[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) { cellv[s] = vis[s] ? true_cell(vr[s], vc[s]) : -1; above[s] = k.above[s]; }
      stream::resolve_sprites<NS, ND>(cellv, above, flat, FWP, lane, sdesc);
    }
    skipv[lane] = skip;
  }
  __syncthreads();
  if (a.debug & 2) return;

  // ---- render phase --------------------------------------------------------------
  stream::PlaneMap<NS, ND, NB> pm;
#pragma unroll
  for (int s = 0; s < NS; ++s) { pm.sprite_off[s] = k.sprite_off[s]; pm.sprite_ch4[s] = k.sprite_ch4[s]; }
#pragma unroll
  for (int d = 0; d < ND; ++d) { pm.drape_off[d] = k.drape_off[d]; pm.drape_ch4[d] = k.drape_ch4[d]; }
  uint32_t bch4[NB > 0 ? NB : 1] = {};
#pragma unroll
  for (int b = 0; b < NB; ++b) { pm.bchar_off[b] = k.bchar_off[b]; bch4[b] = k.bchar_ch4[b]; }
  constexpr uint32_t env_stride = (uint32_t)(1 + L) * (uint32_t)pitch;
  if (!(fc && fc->only))
    stream::stream_planes<NS, ND, NB, QW, NWAVES, EPI>(pm, out.planes + (size_t)env0 * env_stride, env_stride, lds + O_BD, lds + O_BDM,
                                                 flat, sdesc, skipv, FWP, lane, wave, epi, env0, nullptr, 0, nullptr, nullptr, lds);
  if (fc)
    stream::stream_windows<NS, ND, NB, QW, NWAVES, R, C>(fc, pm, bch4, env0, lds + O_BD, flat, sdesc, skipv, FWP, lane, wave, wcorner);
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------

class MaraudersBackend : public Backend {
 public:
  int init(const pcx_template& t, int64_t batch) override;
  int launch(const StepArgs& a, const pcx_buffers& out, hipStream_t s) override;
  int read_things(int64_t env0, int64_t n, pcx_sprite_state* sprites, uint8_t* curtains) override;
  int64_t bytes_per_step() const override {
    // read: action 4 + state 4 NW; write: state 4 NW + planes (1 + L) cells + results 15
    return 4 + 8 * (int64_t)NW + (int64_t)(1 + L) * cells + 15;
  }
  const char* kernel_name() const override { return "pcx_marauders_step"; }
  const int32_t* sprite_track() const override { return track_.ptr; }
  const uint32_t* curtain_bits() const override { return curtains_.ptr; }
  int ensure_curtains() override { return curtains_.ptr ? 0 : curtains_.alloc((size_t)ND * FW * bpad_); }
  int curtain_words() const override { return FW; }
  int64_t batch_pad() const override { return bpad_; }
  void persistent_arrays(std::vector<std::pair<void*, size_t>>& out) override {  // pcx_engine_export_state
    out.push_back({state_.ptr, state_.count * sizeof(uint32_t)});
    out.push_back({track_.ptr, track_.count * sizeof(int32_t)});
  }
  int plane_pitch() const override { return pitch; }
  int set_fused_croppers(const crop::FusedCrops* fc) override { return fused_.set(fc, false, R, C); }
  bool fused_window_features() const override { return true; }
  static size_t base_lds_bytes() {  // the kernel's own dynamic LDS (before padding / the channels-last exchange areas)
    return ((size_t)QW * (1 + NB) + (ND + 1) * WAVE * FWP + 2 + 2 * NS * WAVE + WAVE + stream::WCORNER_WORDS) * 4;
  }
  stream::EpilogueArgs* epilogue_args() override { return &epi_; }
  int set_epilogue(const pcx_epilogue_desc* d) override {
    if (!stream::fill_epilogue(epi_, d, cells, sprite_ch_, NS, drape_ch_, ND, bchar_ch_, NB, 64 * 1024 - base_lds_bytes(), 8))
      return set_error(PCX_E_UNSUPPORTED, "marauders backend: the channels-last epilogue needs rows*cols %% 4 == 0 and a stack of at most %d layers",
                       (int)((64 * 1024 - base_lds_bytes()) / (8 * 2 * WAVE * 4)));
    return 0;
  }

 private:
  stream::FusedCropsHolder fused_;
  Consts k_{};
  stream::EpilogueArgs epi_{};
  int sprite_ch_[NS] = {}, drape_ch_[ND] = {}, bchar_ch_[NB] = {};
  int64_t batch_ = 0, bpad_ = 0;
  int num_cus_ = 256;
  DevArray<uint32_t> tables_, initc_, state_, curtains_;
  DevArray<int32_t> track_;
};

int MaraudersBackend::init(const pcx_template& t, int64_t batch) {
  Consts& k = k_;
  batch_ = batch;
  bpad_ = (batch + WAVE - 1) / WAVE * WAVE;
  if (const char* e = getenv("PCX_FORCE_GENERIC")) if (atoi(e)) return set_error(PCX_E_UNSUPPORTED, "marauders backend: PCX_FORCE_GENERIC");
  if (!t.occlusion_in_layers) return set_error(PCX_E_UNSUPPORTED, "marauders backend: occlusion_in_layers=False");
  if (t.n_directives) return set_error(PCX_E_UNSUPPORTED, "marauders backend: plot directives");
  if (t.rows != R || t.cols != C || t.n_sprites != NS || t.n_drapes != ND || t.n_chars != L || t.n_groups != 1)
    return set_error(PCX_E_UNSUPPORTED, "marauders backend: the shipped 16x39 board and cast only");
  // sprites in template order: P, four upward bolts, two downward bolts; no impassable characters
  for (int s = 0; s < NS; ++s) {
    const pcx_sprite_desc& sd = t.sprites[s];
    const int want = s == 0 ? PCX_PROG_EM_PLAYER : s <= NUP ? PCX_PROG_EM_UPBOLT : PCX_PROG_EM_DOWNBOLT;
    if (sd.program != want || !sd.is_walker || sd.egocentric) return set_error(PCX_E_UNSUPPORTED, "marauders backend: unexpected cast");
    for (int i = 0; i < 16; ++i) if (sd.impassable[i]) return set_error(PCX_E_UNSUPPORTED, "marauders backend: impassable sets must be empty");
  }
  int ib = -1, ix = -1;
  for (int d = 0; d < ND; ++d) {
    if (t.drapes[d].program == PCX_PROG_EM_BUNKER) ib = d;
    if (t.drapes[d].program == PCX_PROG_EM_MARAUDER) ix = d;
    if (t.drapes[d].is_scrolly) return set_error(PCX_E_UNSUPPORTED, "marauders backend: plain drapes only");
  }
  if (ib < 0 || ix < 0) return set_error(PCX_E_UNSUPPORTED, "marauders backend: needs the bunker and the marauder drape");
  k.drape_slot_tmpl[0] = ib; k.drape_slot_tmpl[1] = ix;
  // update schedule: P, B, X, the upward bolts, the downward bolts (one group)
  {
    const int want[NS + ND] = {t.sprites[0].ch, t.drapes[ib].ch, t.drapes[ix].ch, t.sprites[1].ch, t.sprites[2].ch,
                               t.sprites[3].ch, t.sprites[4].ch, t.sprites[5].ch, t.sprites[6].ch};
    if (t.n_things != NS + ND) return set_error(PCX_E_UNSUPPORTED, "marauders backend: unexpected cast");
    for (int i = 0; i < NS + ND; ++i)
      if (t.schedule[i] != want[i] || t.group_of[i] != 0) return set_error(PCX_E_UNSUPPORTED, "marauders backend: unexpected update schedule");
  }
  // z-order: player and bunkers behind the marauders, every bolt in front of them
  int zpos[NS + ND];
  for (int z = 0; z < t.n_things; ++z) {
    int idx = -1;
    for (int s = 0; s < NS; ++s) if (t.sprites[s].ch == t.z_order[z]) idx = s;
    if (t.z_order[z] == t.drapes[ib].ch) idx = NS;
    if (t.z_order[z] == t.drapes[ix].ch) idx = NS + 1;
    if (idx < 0) return set_error(PCX_E_INVALID, "marauders backend: z_order names an unknown character");
    zpos[idx] = z;
  }
  if (zpos[0] > zpos[NS + 1] || zpos[NS] > zpos[NS + 1]) return set_error(PCX_E_UNSUPPORTED, "marauders backend: unexpected z-order");
  for (int s = 1; s < NS; ++s)
    if (zpos[s] < zpos[NS + 1]) return set_error(PCX_E_UNSUPPORTED, "marauders backend: bolts must be in front of the marauders");
  for (int s = 0; s < NS; ++s) {
    k.above[s] = 0;
    for (int j = 0; j < NS + ND; ++j) if (zpos[j] > zpos[s]) k.above[s] |= 1u << j;
  }
  k.n_actions = t.n_actions;
  k.confined = 0;
  for (int s = 0; s < NS; ++s) if (t.sprites[s].confined) k.confined |= 1u << s;
  k.seed_lo = (uint32_t)t.param[0]; k.seed_hi = (uint32_t)t.param[1];
  k.envoff_lo = (uint32_t)t.param[2]; k.envoff_hi = (uint32_t)t.param[3];
  auto layer_of = [&](int ch) { for (int i = 0; i < L; ++i) if (t.chars[i] == ch) return i; return -1; };
  for (int s = 0; s < NS; ++s) {
    k.sprite_off[s] = (uint32_t)(1 + layer_of(t.sprites[s].ch)) * pitch;
    k.sprite_ch4[s] = t.sprites[s].ch * 0x01010101u;
    sprite_ch_[s] = t.sprites[s].ch;
  }
  for (int d = 0; d < ND; ++d) {
    const pcx_drape_desc& dd = t.drapes[k.drape_slot_tmpl[d]];
    k.drape_off[d] = (uint32_t)(1 + layer_of(dd.ch)) * pitch;
    k.drape_ch4[d] = dd.ch * 0x01010101u;
    drape_ch_[d] = dd.ch;
  }
  stream::fill_epilogue(epi_, nullptr, cells, sprite_ch_, NS, drape_ch_, ND, bchar_ch_, NB);
  std::vector<uint32_t> tab((size_t)QW * (1 + NB), 0);
  memcpy(tab.data(), t.backdrop, cells);
  int nb = 0;
  for (int i = 0; i < L; ++i) {
    const int ch = t.chars[i];
    bool thing = false;
    for (int s = 0; s < NS; ++s) thing |= t.sprites[s].ch == ch;
    for (int d = 0; d < ND; ++d) thing |= t.drapes[d].ch == ch;
    if (thing) continue;
    if (nb >= NB) return set_error(PCX_E_INVALID, "marauders backend: inconsistent character set");
    k.bchar_off[nb] = (uint32_t)(1 + i) * pitch;
    k.bchar_ch4[nb] = (uint32_t)ch * 0x01010101u;
    bchar_ch_[nb] = ch;
    uint8_t* m = reinterpret_cast<uint8_t*>(tab.data() + (size_t)QW * (1 + nb));
    for (int c = 0; c < cells; ++c) m[c] = t.backdrop[c] == ch;
    ++nb;
  }
  if (nb != NB) return set_error(PCX_E_INVALID, "marauders backend: inconsistent character set");
  std::vector<uint32_t> initc((size_t)ND * FW, 0);
  for (int d = 0; d < ND; ++d)
    for (int c = 0; c < cells; ++c)
      if (t.drapes[k.drape_slot_tmpl[d]].curtain[c]) initc[(size_t)d * FW + (c >> 5)] |= 1u << (c & 31);
  const int dx0 = t.drapes[ix].param[0];
  if (dx0 != 1 && dx0 != -1) return set_error(PCX_E_UNSUPPORTED, "marauders backend: MarauderDrape._dx must be +-1");
  memset(k.init, 0, sizeof k.init);
  k.init[W_FRAME] = (uint32_t)-1;
  uint32_t sf = 0;
  for (int s = 0; s < NS; ++s) {
    const pcx_sprite_desc& sd = t.sprites[s];
    sf |= ((uint32_t)(sd.visible != 0) | ((uint32_t)(sd.prior_visible != 0) << 1)) << (2 * s);
    k.init[W_POS + s] = ((uint32_t)sd.vrow & 0xFFFFu) | ((uint32_t)sd.vcol << 16);
  }
  k.init[W_FLAGS] = ((uint32_t)(dx0 + 1) << F_DX_SHIFT) | (sf << F_SF_SHIFT);
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      num_cus_ = prop.multiProcessorCount;
  }
  int rc;
  if ((rc = tables_.upload(tab))) return rc;
  if ((rc = initc_.upload(initc))) return rc;
  if ((rc = state_.alloc((size_t)NW * bpad_))) return rc;
  if ((rc = track_.alloc((size_t)NS * bpad_))) return rc;
  return 0;
}

int MaraudersBackend::launch(const StepArgs& a, const pcx_buffers& out, hipStream_t s) {
  if (a.n_steps != 1) return set_error(PCX_E_INVALID, "marauders backend: one step per launch");
  if (a.export_curtains && !curtains_.ptr) {
    int rc = curtains_.alloc((size_t)ND * FW * bpad_);
    if (rc) return rc;
  }
  Ptrs P{tables_.ptr, initc_.ptr, state_.ptr, track_.ptr, curtains_.ptr, batch_, bpad_};
  const int64_t groups = bpad_ / WAVE;
  // Launch shape: single-wave workgroups with LDS padded so that about four
  // share a CU; when the batch leaves the chip underfilled (BASELINE config 3:
  // 32,768 environments = two groups per CU) four waves share a group's render loop.
  // Measured (tools/knob_sweep_r02.sh, profiles/r02_tuning.md): 32,768 envs 4 waves 0.0413 ms vs 8 waves 0.0435;
  // 262,144 envs single-wave workgroups at 4 per CU 0.412 ms vs 0.490 at 8 (eleven 624-byte planes per
  // environment: fewer concurrent write streams per CU are faster, as for scrolly_maze).
  int waves_per_cu = 4, nwaves = groups < (int64_t)num_cus_ * 5 ? 4 : 1;
  if (const char* e = getenv("PCX_WAVES_PER_CU")) waves_per_cu = atoi(e);
  if (const char* e = getenv("PCX_EM_WAVES")) { const int v = atoi(e); if (v == 1 || v == 4 || v == 8) nwaves = v; }
  size_t lds = base_lds_bytes();
  const stream::EpilogueArgs epi_ = stream::with_hwc_scratch(this->epi_, lds, nwaves);  // (channels-last epilogue: its exchange area behind the kernel's own LDS)
  if (nwaves == 1 && waves_per_cu > 0) {
    size_t want = ((size_t)(160 * 1024) / (size_t)waves_per_cu) & ~(size_t)255;
    if (want > 64 * 1024) want = 64 * 1024;
    if (want > lds) lds = want;
  }
#define PCX_EM_LAUNCH(nw, epi)                                                                                 \
  hipLaunchKernelGGL((pcx_marauders_step<nw, epi>), dim3((unsigned)groups), dim3(nw * WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr())
  const bool epi = epi_.out != nullptr;  // the feature-array epilogue has its own instances
  if (nwaves == 8) { if (epi) PCX_EM_LAUNCH(8, true); else PCX_EM_LAUNCH(8, false); }
  else if (nwaves == 4) { if (epi) PCX_EM_LAUNCH(4, true); else PCX_EM_LAUNCH(4, false); }
  else { if (epi) PCX_EM_LAUNCH(1, true); else PCX_EM_LAUNCH(1, false); }
#undef PCX_EM_LAUNCH
  PCX_HIP(hipGetLastError());
  return 0;
}

int MaraudersBackend::read_things(int64_t env0, int64_t n, pcx_sprite_state* sprites, uint8_t* curtains) {
  std::vector<uint32_t> st((size_t)NW * n);
  PCX_HIP(hipDeviceSynchronize());
  for (int w = 0; w < NW; ++w)
    PCX_HIP(hipMemcpy(st.data() + (size_t)w * n, state_.ptr + (size_t)w * bpad_ + env0, n * 4, hipMemcpyDeviceToHost));
  auto word = [&](int w, int64_t i) { return st[(size_t)w * n + i]; };
  for (int64_t i = 0; i < n; ++i) {
    if (sprites)
      for (int s = 0; s < NS; ++s) {
        pcx_sprite_state& o = sprites[i * NS + s];
        memset(&o, 0, sizeof o);
        const uint32_t pw = word(W_POS + s, i);
        o.vrow = (int16_t)(pw & 0xFFFF); o.vcol = (int16_t)(pw >> 16);
        const bool on = o.vrow >= 0 && o.vrow < R && o.vcol >= 0 && o.vcol < C;
        o.row = on ? o.vrow : 0; o.col = on ? o.vcol : 0;
        o.visible = (word(W_FLAGS, i) >> (F_SF_SHIFT + 2 * s)) & 1;
      }
    if (curtains)
      for (int d = 0; d < ND; ++d)
        for (int c = 0; c < cells; ++c)
          curtains[((size_t)i * ND + k_.drape_slot_tmpl[d]) * cells + c] = (word((d == 0 ? W_B : W_X) + (c >> 5), i) >> (c & 31)) & 1;
  }
  return 0;
}

}  // namespace em

Backend* make_marauders_backend() { return new em::MaraudersBackend(); }

}  // namespace pcx
