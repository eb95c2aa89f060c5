// pcx_scrolly_maze_kernel.h -- the device side of pcx_scrolly_maze.hip: pcx_scrolly_maze_step and everything it calls.
// A header of its own (round 6) because it is compiled twice: into libpcx.so with every instance the launch code picks from,
// and at run time by hiprtc with PCX_SM_SPEC naming the constants of ONE template (see PCX_SM_SPEC below and
// ScrollyMazeBackend::specialise in pcx_scrolly_maze.hip) -- the instances with a level's constants compiled in, for levels
// the library was not built with.  Device code only: no host header may be reached from here when __HIPCC_RTC__ is defined
// (pcx_device.h).  gfx950 only.
#pragma once

#include "pcx_stream.h"  // EpilogueArgs: the feature-array epilogue shared with the other hand-written kernels

namespace pcx {
namespace sm {

constexpr int WAVE = 64;
constexpr int MAX_NS = 6;   // sprites (patrollers + player); their flag nibbles share one state word
static_assert(MAX_NS * 4 <= 32, "sprite flag nibbles must fit the W_SFLAGS word");
constexpr int MAX_Z = 8;
constexpr int MAX_L = 16;

// State words (uint32 [NW][batch_padded]).
enum : int { W_FRAME = 0, W_FLAGS, W_PERMIT_FRAME, W_MAZE, W_CASH, W_STALE, W_SFLAGS, W_SPOS };
// W_FLAGS bits
constexpr uint32_t F_OVER = 1u, F_ERR_SHIFT = 1, F_ERR_MASK = 7u << 1, F_REGISTERED = 1u << 4,
                   F_PERMIT_VALID = 1u << 5;
constexpr int F_PERMIT_SHIFT = 6;  // 9 bits
constexpr uint32_t STALE_NONE = 0xFFFFu;

// Everything that is the same for every environment.  Lives in device global
// memory; all accesses are wave-uniform so they compile to scalar loads.
struct Consts {
  int32_t R, C, cells, QW, L, n_things, NS;
  uint32_t magic_q, magic_c;
  int32_t PR, PC, WPR;
  int32_t lim_r, lim_c;
  int32_t have_margins, margin_n, margin_s, margin_w, margin_e;
  int32_t n_coins, CW, NW;
  int32_t ip;  // sprite index of 'P' (PatrollerSprite / CashDrape look it up)
  int32_t ie;  // index of the (single) egocentric sprite, -1 if none
  int32_t maze_ch, cash_ch;
  int32_t n_actions;
  uint32_t chars[MAX_L];
  // sprites, in engine insertion order (== update order inside group 1)
  int32_t prog[MAX_NS], confined[MAX_NS], egocentric[MAX_NS], sprite_ch[MAX_NS];
  int32_t tmpl_index[MAX_NS];  // sprite s here (update order) is sprite tmpl_index[s] of the template (insertion order)
  uint32_t imp[MAX_NS][4];
  // Probes number the things by z-order position (bit z = the z-th thing from the
  // back), so the character on top of a cell is the highest set presence bit.
  uint32_t relevant[MAX_NS];  // z-bits that can change a probe's verdict for walker s
  uint32_t imp_z[MAX_NS];     // z-bits of the things whose character is impassable to walker s
  int32_t relevant_backdrop[MAX_NS];
  int32_t zpos_sprite[MAX_NS], zpos_maze, zpos_cash;
  uint32_t init[24];          // initial state words (coin words are derived)
  // z-order, back to front
  int32_t z_kind[MAX_Z], z_idx[MAX_Z], z_ch[MAX_Z];
  // LDS layout (word offsets)
  // occlusion, resolved once per environment in phase A (so that phase B paints
  // in a fixed order): thing bits are sprites 0..NS-1, drapes NS (maze), NS+1 (cash)
  uint32_t above[MAX_NS + 2];   // things strictly in front of thing t
  int32_t lay_sprite[MAX_NS], lay_drape[2];  // layer plane index of each thing's character
  int32_t n_bchars, bchar[MAX_L], lay_bchar[MAX_L];  // characters only the backdrop paints
  int32_t FW;  // words of one flat curtain bit-vector (cells bits + 1 spill word)
  int32_t lds_walls, lds_backdrop, lds_rowstart, lds_coincol, lds_flat, lds_sdesc, lds_cmask,
      lds_skip, lds_bdmask, lds_buf_words, lds_flatraw, lds_sdescraw, lds_wcorner, lds_fparams, lds_words;
  // CODES instance (at most eight characters): every cell's painter as the index
  // of its character among the sorted characters ("owner code", one byte per
  // cell), from which v_perm_b32 makes the board dword and every layer dword
  int32_t lds_bdcode, lds_codes, lds_cmask_c, lds_skip_c, lds_words_codes;  // its own, compact LDS layout
  // persistent shapes of the CODES instance (PS): the state inbox (LDS-DMA target), the hand-over ring of the
  // logic/render wave pair, and the owner-code buffers (code table + skip flags each)
  int32_t lds_ps_inbox, lds_ps_ring, lds_ps_cmask, lds_ps_buf0, lds_ps_buf_words, lds_ps_words1;
  int32_t lds_ps_wave_words;  // PS == 3: from one wave's {inbox, coin masks, buffer} to the next wave's
  int32_t lds_ps_lut;         // persistent owner-code shapes: 256 words "eight cell bits -> eight nibble masks", then the backdrop's owner codes as nibbles
  uint32_t chars_lo, chars_hi;  // characters 0..3 / 4..7 as bytes
  int32_t sprite_by_z[MAX_NS];  // sprites back to front
};

// The shipped levels with their Consts compiled in (template parameter LV of the kernel).  pcx_sm_shipped.h is GENERATED
// (tools/gen_sm_shipped.py: the words pcx_debug_scrolly_consts() answers for tests/golden/templates/scrolly_maze_L0 / _L1 /
// _L2.npz -- the reference's examples/scrolly_maze.py MAZES_ART[0..2] as pycolab_amd/compiler.py compiles them) and
// committed; tests/test_host_api.py requires it to be what this library plans today, and launch() takes a compiled-in
// instance only when the engine's own Consts are the same words -- any other level keeps the instance that reads them from
// the kernel arguments, or (round 6) gets instances of its own at run time: a build of this header by hiprtc with
// PCX_SM_SPEC naming a header that holds ITS words (ScrollyMazeBackend::spec_header writes it: PCX_SM_SPEC_N,
// PCX_SM_SPEC_WORDS, PCX_SM_SPEC_PLAIN_WORDS), LV 7 and 8.  A header of another size (Consts changed, header not yet
// regenerated) compiles to instances nobody launches.  LV: 0 none; odd = a level's constants as the persistent shape launches
// them (1, 3: levels 0, 1; level 2 has six coin words and does not take that shape; 7: PCX_SM_SPEC's), even = as init()
// leaves them, for the cooperative shape (2, 4, 6: levels 0, 1, 2; 8: PCX_SM_SPEC's).
constexpr int CONSTS_WORDS = (int)(sizeof(Consts) / 4);
static_assert(sizeof(Consts) % 4 == 0, "Consts is a struct of 32-bit fields");
template <typename T>
constexpr Consts consts_from_words(const T& raw) {
  if constexpr (sizeof(T) == sizeof(Consts)) return __builtin_bit_cast(Consts, raw);
  else return Consts{};
}
#ifdef PCX_SM_SPEC
#include PCX_SM_SPEC
struct SpecWords { uint32_t w[PCX_SM_SPEC_N]; };
static_assert(sizeof(SpecWords) == sizeof(Consts), "PCX_SM_SPEC was written for another sm::Consts");
static constexpr SpecWords SPEC_WORDS[2] = {{{PCX_SM_SPEC_WORDS}}, {{PCX_SM_SPEC_PLAIN_WORDS}}};
static constexpr Consts SPEC_7 = consts_from_words(SPEC_WORDS[0]), SPEC_8 = consts_from_words(SPEC_WORDS[1]);
template <int LV>
__device__ __forceinline__ const Consts& baked_consts(const Consts& from_args) {
  if constexpr (LV == 7) return SPEC_7;
  else if constexpr (LV == 8 || LV == 9) return SPEC_8;
  else return from_args;
}
#else
#include "pcx_sm_shipped.h"
struct ShippedWords { uint32_t w[PCX_SM_SHIPPED_N > 0 ? PCX_SM_SHIPPED_N : 1]; };
constexpr bool SHIPPED_VALID = sizeof(ShippedWords) == sizeof(Consts);
constexpr int N_BAKED = 7;  // LV 1..6
static constexpr ShippedWords SHIPPED_WORDS[N_BAKED] = {{{0}}, {{PCX_SM_SHIPPED_L0_WORDS}}, {{PCX_SM_SHIPPED_L0_PLAIN_WORDS}}, {{PCX_SM_SHIPPED_L1_WORDS}},
                                                        {{PCX_SM_SHIPPED_L1_PLAIN_WORDS}}, {{0}}, {{PCX_SM_SHIPPED_L2_PLAIN_WORDS}}};
static constexpr Consts SHIPPED_1 = consts_from_words(SHIPPED_WORDS[1]), SHIPPED_2 = consts_from_words(SHIPPED_WORDS[2]),
                        SHIPPED_3 = consts_from_words(SHIPPED_WORDS[3]), SHIPPED_4 = consts_from_words(SHIPPED_WORDS[4]),
                        SHIPPED_6 = consts_from_words(SHIPPED_WORDS[6]);
template <int LV>
__device__ __forceinline__ const Consts& baked_consts(const Consts& from_args) {
  if constexpr (LV == 1) return SHIPPED_1;
  else if constexpr (LV == 2) return SHIPPED_2;
  else if constexpr (LV == 3) return SHIPPED_3;
  else if constexpr (LV == 4) return SHIPPED_4;
  else if constexpr (LV == 6) return SHIPPED_6;
  else return from_args;
}
#endif

struct Ptrs {
  const uint32_t* walls_bits;   // [PR][WPR]
  const uint32_t* backdrop4;    // [QW] backdrop as dwords, then [n_bchars][QW] (backdrop == bchar) 0/1 bytes
  const uint32_t* coin_bits;      // [PR][CWPR] the CashDrape's whole pattern as bit-rows (no spare word)
  const uint16_t* coin_rowbase;   // [PR+1] coins in the rows before row r (row-major coin ids)
  uint32_t* state;                // [NW][bpad]
  int32_t* track;                 // [NS][bpad] packed true row | col<<8 | visible<<16
  uint32_t* curtains;             // [2][FW][bpad] raw curtain bits, template drape order (export_curtains)
  int32_t maze_slot;              // template drape index of the maze drape (0 or 1)
  int64_t batch, bpad;
  // persistent shapes (PS): environments per work unit (64, 32 or 16), the ticket counter {next ticket, workgroups
  // done} the units beyond every workgroup's first two are drawn from, or static round-robin when `ps_dynamic` is 0
  uint32_t* ps_ctr;
  int32_t ps_unit, ps_dynamic;
  int32_t ps_lock;  // PS == 3: at most this many waves of a workgroup in their render loop at a time (0: no limit)
  int32_t ps_steal; // tickets: a worker whose shard of the work counter is dry draws from the other shards (PCX_SM_STEAL=0: goes home)
  // the batch's last environments go in SMALL units (ps_tail_unit environments each; units ps_n1 and up), so that what the
  // workers hold when the tickets run out -- the launch's tail -- is short; ps_n1 = all units when there is no such region
  int32_t ps_tail_unit;
  uint32_t ps_n1, ps_n;
  // phase timers (tools/ps_sweep.py --prof): 16 words per workgroup, 10 ns ticks of s_memrealtime summed over its units --
  // logic wave: [0] units, [1] wait for the inbox, [2] wait for a free buffer, [3] stepping, [4] wait for the ticket,
  // [5] lifetime; render wave: [8] units, [9] wait for a full buffer, [10] streaming, [11] lifetime (shape 1: [3], [10] and [5])
  uint32_t* ps_prof;
};

struct Walker {
  int vr, vc, vis, prior, var;
};
struct Scrolly {
  int r, c, pre_r, pre_c, moved;
};
struct Plot {
  int frame;
  uint32_t flags;      // registered / permit bits / error
  int permit_frame;
  int order_valid, o0, o1;
  int reward_set, reward, game_over;
  float discount;
};

__device__ __forceinline__ bool on_board(const Consts& k, int r, int c) {
  return (unsigned)r < (unsigned)k.R && (unsigned)c < (unsigned)k.C;
}
__device__ __forceinline__ int motion_bit(int dr, int dc) { return (dr + 1) * 3 + (dc + 1); }

// sprites.py:315-352 _teleport (+ _on_board_exit/_enter :223-275)
__device__ __forceinline__ void teleport(const Consts& k, Walker& w, int nr, int nc) {
  bool old_on = on_board(k, w.vr, w.vc), new_on = on_board(k, nr, nc);
  if (old_on && !new_on) { w.prior = w.vis; w.vis = 0; }
  w.vr = nr;
  w.vc = nc;
  if (!old_on && new_on) w.vis = w.prior;
}
// cell index the sprite is painted at (engine.py:752-753), -1 when invisible
__device__ __forceinline__ int paint_cell(const Consts& k, const Walker& w) {
  if (!w.vis) return -1;
  return on_board(k, w.vr, w.vc) ? w.vr * k.C + w.vc : 0;
}

struct Lds {
  const uint32_t* walls;
  const uint32_t* backdrop4;
  const uint32_t* coinbits;   // [PR][CWPR]
  const uint16_t* rowbase;    // [PR+1]
  uint32_t* flat;   // [2][FW][64] curtains as flat cell-bit vectors
  uint2* sdesc;     // [NS][64] sprite paint descriptors {dword index q, byte mask}
  uint32_t* cmask;  // [CW][64]
  uint32_t* skip;   // [64]
  const uint32_t* bdmask;  // [n_bchars][QW]
};

__device__ __forceinline__ int wall_at(const Consts& k, const Lds& l, int pr, int pc, uint32_t& err) {
  if ((unsigned)pr >= (unsigned)k.PR || (unsigned)pc >= (unsigned)k.PC) { err |= ERR_INDEX; return 0; }
  return (l.walls[pr * k.WPR + (pc >> 5)] >> (pc & 31)) & 1;
}
// Coins are numbered in row-major order of the pattern.  Number of template
// coins in row pr before column pc (pr in range, 0 <= pc <= PC): every read is
// independent of the others (one LDS round trip).
__device__ __forceinline__ int coins_before(const Consts& k, const Lds& l, int pr, int pc) {
  const int CWPR = k.WPR - 1, wi = pc >> 5;
  const uint32_t* row = l.coinbits + pr * CWPR;
  int n = l.rowbase[pr];
  for (int j = 0; j < CWPR; ++j) {  // uniform trip count, predicated by j <= wi
    const uint32_t w = row[j];
    n += j < wi ? __popc(w) : j == wi ? __popc(w & ((1u << (pc & 31)) - 1u)) : 0;
  }
  return n;
}
// id of the template coin at pattern cell (pr, pc), or -1.  Branch-free: cells
// outside the pattern read row 0 and answer -1.
__device__ __forceinline__ int coin_id_at(const Consts& k, const Lds& l, int pr, int pc) {
  const bool in = (unsigned)pr < (unsigned)k.PR && (unsigned)pc < (unsigned)k.PC;
  const int prc = in ? pr : 0, pcc = in ? pc : 0;
  const uint32_t w = l.coinbits[prc * (k.WPR - 1) + (pcc >> 5)];
  const int id = coins_before(k, l, prc, pcc);
  return (in && ((w >> (pcc & 31)) & 1u)) ? id : -1;
}
__device__ __forceinline__ bool coin_alive(const Lds& l, int lane, int id) {
  return (l.cmask[(id >> 5) * WAVE + lane] >> (id & 31)) & 1;
}

// Row r of both curtains of the environment in LDS column `e`: the walls' pattern window, and the
// pattern's coin bits in the window, each kept iff its coin is alive (or is the stale one).  The
// window's coins have consecutive ids, so their alive bits are one bit range of the coin mask.
// Returns false where the walls' window leaves the pattern (the reference raises IndexError).
__device__ __forceinline__ bool curtain_row_bits(const Consts& k, const Lds& l, int e, int maze_r, int maze_c, int cash_r,
                                                 int cash_c, uint32_t stale, int r, int C, uint32_t cmaskC, uint32_t& wbits,
                                                 uint32_t& cbits) {
  bool ok = true;
  const int pr = maze_r + r;
  wbits = 0;
  if ((unsigned)pr < (unsigned)k.PR && maze_c >= 0 && maze_c + C <= k.PC) {
    const uint32_t* row = l.walls + pr * k.WPR;
    const int wi = maze_c >> 5, sh = maze_c & 31;
    const uint64_t pair = (uint64_t)row[wi] | ((uint64_t)row[wi + 1] << 32);  // WPR has a spare word
    wbits = (uint32_t)(pair >> sh) & cmaskC;
  } else {
    ok = false;
  }
  const int cr = cash_r + r;
  cbits = 0;
  if ((unsigned)cr < (unsigned)k.PR) {
    const int CWPR = k.WPR - 1;
    uint32_t sbits;
    int id0;
    if (cash_c >= 0 && cash_c + C <= k.PC) {
      const uint32_t* row = l.coinbits + cr * CWPR;
      const int wi = cash_c >> 5, sh = cash_c & 31;
      const uint64_t pair = (uint64_t)row[wi] | ((uint64_t)row[wi + 1 < CWPR ? wi + 1 : wi] << 32);
      sbits = (uint32_t)(pair >> sh) & cmaskC;
      id0 = coins_before(k, l, cr, cash_c);
    } else {  // a window that leaves the pattern sideways (never, while the drape obeys its limits): cell by cell
      sbits = 0;
      id0 = -1;
      for (int col = 0; col < C; ++col) {
        const int id = coin_id_at(k, l, cr, cash_c + col);
        if (id >= 0) { sbits |= 1u << col; if (id0 < 0) id0 = id; }
      }
      if (id0 < 0) id0 = 0;
    }
    const int w0 = id0 >> 5;
    const uint32_t lo = l.cmask[w0 * WAVE + e];
    const uint32_t hi = l.cmask[(w0 + 1 < k.CW ? w0 + 1 : w0) * WAVE + e];
    uint32_t alive = (uint32_t)((((uint64_t)hi << 32) | lo) >> (id0 & 31));
    const uint32_t st_rel = stale - (uint32_t)id0;  // the coin picked up last frame is still drawn
    alive |= st_rel < 32u ? 1u << st_rel : 0u;
    for (uint32_t sb = sbits; sb; sb &= sb - 1u, alive >>= 1)  // deposit alive bit j at the j-th coin of the row window
      cbits |= (alive & 1u) ? sb & (0u - sb) : 0u;
  }
  return ok;
}

// What the last repaint showed, for lazy board probes.
template <int NS>
struct Snap {
  int cell[NS];
  int maze_r, maze_c, cash_r, cash_c;
  uint32_t stale;
};

// Is the character on top of board cell (r, c) impassable to walker s?
// (sprites.py:496-511 at()/is_impassable() over engine.py:737-759 _render.)
template <int NS>
__device__ __forceinline__ bool blocked_at(const Consts& k, const Lds& l, const Snap<NS>& sn, int s,
                                           const Walker& w, int dr, int dc, int lane, uint32_t& err) {
  // Straight-line code: the only branches are on kernel arguments (uniform), so
  // the probes of one _move -- up to three for the motion, eight for the scroll
  // permits -- have all their LDS reads in flight together.  A cell off the
  // board is probed at (0, 0) and the verdict replaced by EDGE's.
  const int r0 = w.vr + dr, c0 = w.vc + dc;
  const bool onb = on_board(k, r0, c0);
  const int r = onb ? r0 : 0, c = onb ? c0 : 0;
  const uint32_t rel = k.relevant[s];
  uint32_t present = 0;
  const int cell = r * k.C + c;
#pragma unroll
  for (int j = 0; j < NS; ++j)
    if ((rel >> k.zpos_sprite[j]) & 1) present |= (uint32_t)(sn.cell[j] == cell) << k.zpos_sprite[j];
  if ((rel >> k.zpos_maze) & 1) {
    const int pr = sn.maze_r + r, pc = sn.maze_c + c;
    const bool in = (unsigned)pr < (unsigned)k.PR && (unsigned)pc < (unsigned)k.PC;
    const uint32_t wbits = l.walls[in ? pr * k.WPR + (pc >> 5) : 0];
    if (onb && !in) err |= ERR_INDEX;
    present |= (uint32_t)(in && ((wbits >> (pc & 31)) & 1u)) << k.zpos_maze;
  }
  if ((rel >> k.zpos_cash) & 1) {
    const int id = coin_id_at(k, l, sn.cash_r + r, sn.cash_c + c);
    const bool there = id >= 0 && (coin_alive(l, lane, id < 0 ? 0 : id) || (uint32_t)id == sn.stale);
    present |= (uint32_t)there << k.zpos_cash;
  }
  bool blocked = (k.imp_z[s] >> (present ? 31 - __clz((int)present) : 0)) & 1;  // the thing in front decides
  if (k.relevant_backdrop[s]) {
    const int top = (l.backdrop4[cell >> 2] >> ((cell & 3) * 8)) & 0xFF;
    // the 128-bit impassable set as two 64-bit halves and one select: anything that looks like a
    // runtime index into the kernarg struct makes the compiler copy the table to scratch
    const uint64_t lo64 = (uint64_t)k.imp[s][0] | ((uint64_t)k.imp[s][1] << 32), hi64 = (uint64_t)k.imp[s][2] | ((uint64_t)k.imp[s][3] << 32);
    const uint64_t half = top < 64 ? lo64 : hi64;
    blocked = present ? blocked : (top < 128 && ((half >> (top & 63)) & 1ull) != 0);
  } else {
    blocked = present ? blocked : false;
  }
  return onb ? blocked : k.confined[s] != 0;  // EDGE
}

// The same probe for a walker that is the LANE's own (cooperative shape, four lanes per environment:
// lane j steps sprite j): what blocked_at() reads from Consts by compile-time index arrives as values.
struct WalkerConsts {
  uint32_t relevant, imp_z;
  bool relevant_backdrop, confined;
  uint64_t imp_lo, imp_hi;
};
template <int NS>
__device__ __forceinline__ WalkerConsts walker_consts(const Consts& k, int j) {
  WalkerConsts c{k.relevant[0], k.imp_z[0], k.relevant_backdrop[0] != 0, k.confined[0] != 0,
                 (uint64_t)k.imp[0][0] | ((uint64_t)k.imp[0][1] << 32), (uint64_t)k.imp[0][2] | ((uint64_t)k.imp[0][3] << 32)};
#pragma unroll
  for (int s = 1; s < NS; ++s) {
    const bool hit = j == s;
    c.relevant = hit ? k.relevant[s] : c.relevant;
    c.imp_z = hit ? k.imp_z[s] : c.imp_z;
    c.relevant_backdrop = hit ? k.relevant_backdrop[s] != 0 : c.relevant_backdrop;
    c.confined = hit ? k.confined[s] != 0 : c.confined;
    c.imp_lo = hit ? ((uint64_t)k.imp[s][0] | ((uint64_t)k.imp[s][1] << 32)) : c.imp_lo;
    c.imp_hi = hit ? ((uint64_t)k.imp[s][2] | ((uint64_t)k.imp[s][3] << 32)) : c.imp_hi;
  }
  return c;
}
template <int NS>
__device__ __forceinline__ bool blocked_at_lane(const Consts& k, const Lds& l, const Snap<NS>& sn, const WalkerConsts& wc,
                                                const Walker& w, int dr, int dc, int col, uint32_t& err) {
  const int r0 = w.vr + dr, c0 = w.vc + dc;
  const bool onb = on_board(k, r0, c0);
  const int r = onb ? r0 : 0, c = onb ? c0 : 0;
  uint32_t present = 0;
  const int cell = r * k.C + c;
#pragma unroll
  for (int j = 0; j < NS; ++j) present |= (uint32_t)(sn.cell[j] == cell) << k.zpos_sprite[j];
  {
    const int pr = sn.maze_r + r, pc = sn.maze_c + c;
    const bool in = (unsigned)pr < (unsigned)k.PR && (unsigned)pc < (unsigned)k.PC;
    const uint32_t wbits = l.walls[in ? pr * k.WPR + (pc >> 5) : 0];
    if (onb && !in && ((wc.relevant >> k.zpos_maze) & 1)) err |= ERR_INDEX;
    present |= (uint32_t)(in && ((wbits >> (pc & 31)) & 1u)) << k.zpos_maze;
  }
  {
    const int id = coin_id_at(k, l, sn.cash_r + r, sn.cash_c + c);
    const bool there = id >= 0 && (coin_alive(l, col, id < 0 ? 0 : id) || (uint32_t)id == sn.stale);
    present |= (uint32_t)there << k.zpos_cash;
  }
  present &= wc.relevant;  // (things that cannot change this walker's verdict were never looked at in blocked_at())
  bool blocked = (wc.imp_z >> (present ? 31 - __clz((int)present) : 0)) & 1;
  const int top = (l.backdrop4[cell >> 2] >> ((cell & 3) * 8)) & 0xFF;
  const uint64_t half = top < 64 ? wc.imp_lo : wc.imp_hi;
  const bool back = wc.relevant_backdrop && top < 128 && ((half >> (top & 63)) & 1ull) != 0;
  blocked = present ? blocked : back;
  return onb ? blocked : wc.confined;  // EDGE
}
template <int NS>
__device__ __forceinline__ bool check_motion_lane(const Consts& k, const Lds& l, const Snap<NS>& sn, const WalkerConsts& wc,
                                                  const Walker& w, int dr, int dc, int col, uint32_t& err) {
  if (dr == 0 && dc == 0) return false;
  if (dr != 0 && dc != 0) {
    if (blocked_at_lane<NS>(k, l, sn, wc, w, dr, dc, col, err)) return true;
    return blocked_at_lane<NS>(k, l, sn, wc, w, dr, 0, col, err) && blocked_at_lane<NS>(k, l, sn, wc, w, 0, dc, col, err);
  }
  return blocked_at_lane<NS>(k, l, sn, wc, w, dr, dc, col, err);
}

// sprites.py:479-546 _check_motion
template <int NS>
__device__ __forceinline__ bool check_motion(const Consts& k, const Lds& l, const Snap<NS>& sn, int s,
                                             const Walker& w, int dr, int dc, int lane, uint32_t& err) {
  if (dr == 0 && dc == 0) return false;
  if (dr != 0 && dc != 0) {
    if (blocked_at<NS>(k, l, sn, s, w, dr, dc, lane, err)) return true;
    return blocked_at<NS>(k, l, sn, s, w, dr, 0, lane, err) && blocked_at<NS>(k, l, sn, s, w, 0, dc, lane, err);
  }
  return blocked_at<NS>(k, l, sn, s, w, dr, dc, lane, err);
}

// sprites.py:356-389 _move (with :413-477 scrolling hooks)
// EGO: whether the walker is the egocentric one, when the instance knows (1 / 0), else -1 (ask Consts)
template <int NS, int EGO = -1>
__device__ __forceinline__ bool mw_move(const Consts& k, const Lds& l, const Snap<NS>& sn, int s, Walker& w,
                                        Plot& p, int dr, int dc, int lane, uint32_t& err, int quad_j = -1) {
  const bool ego = EGO >= 0 ? EGO != 0 : k.egocentric[s] != 0;
  if (ego) p.flags |= F_REGISTERED;  // scrolling.py:287-312
  if (p.order_valid) {               // sprites.py:446-454
    teleport(k, w, w.vr - p.o0, w.vc - p.o1);
    if (ego && p.o0 != dr && p.o1 != dc) err |= ERR_SCROLL;
  }
  bool blocked = check_motion<NS>(k, l, sn, s, w, dr, dc, lane, err);
  if (!blocked) teleport(k, w, w.vr + dr, w.vc + dc);
  if (ego) {  // sprites.py:456-477 + scrolling.py:372-434 permit()
    // the eight neighbours, each probed once
    bool n, so, we, ea, nw, ne, sw, se;
    if (quad_j >= 0) {
      // four lanes step this environment in lock step: lane j probes neighbours 2j and 2j + 1 (in the order N, S,
      // W, E, NW, NE, SW, SE), then the quad ORs its verdicts (and whatever the probes found wrong) together
      const int a0 = quad_j == 0 ? -1 : quad_j == 1 ? 0 : quad_j == 2 ? -1 : 1, b0 = quad_j == 0 ? 0 : -1;
      const int a1 = quad_j == 0 ? 1 : quad_j == 1 ? 0 : quad_j == 2 ? -1 : 1, b1 = quad_j == 0 ? 0 : 1;
      uint32_t nb = (uint32_t)blocked_at<NS>(k, l, sn, s, w, a0, b0, lane, err) << (2 * quad_j);
      nb |= (uint32_t)blocked_at<NS>(k, l, sn, s, w, a1, b1, lane, err) << (2 * quad_j + 1);
      nb |= err << 8;
      nb |= (uint32_t)__shfl_xor((int)nb, 1);
      nb |= (uint32_t)__shfl_xor((int)nb, 2);
      err |= (nb >> 8) & 7u;
      n = nb & 1u; so = (nb >> 1) & 1u; we = (nb >> 2) & 1u; ea = (nb >> 3) & 1u;
      nw = (nb >> 4) & 1u; ne = (nb >> 5) & 1u; sw = (nb >> 6) & 1u; se = (nb >> 7) & 1u;
    } else {
      n = blocked_at<NS>(k, l, sn, s, w, -1, 0, lane, err); so = blocked_at<NS>(k, l, sn, s, w, 1, 0, lane, err);
      we = blocked_at<NS>(k, l, sn, s, w, 0, -1, lane, err); ea = blocked_at<NS>(k, l, sn, s, w, 0, 1, lane, err);
      nw = blocked_at<NS>(k, l, sn, s, w, -1, -1, lane, err); ne = blocked_at<NS>(k, l, sn, s, w, -1, 1, lane, err);
      sw = blocked_at<NS>(k, l, sn, s, w, 1, -1, lane, err); se = blocked_at<NS>(k, l, sn, s, w, 1, 1, lane, err);
    }
    uint32_t legal = 1u << motion_bit(0, 0);
    legal |= (uint32_t)!n << motion_bit(-1, 0);
    legal |= (uint32_t)!so << motion_bit(1, 0);
    legal |= (uint32_t)!we << motion_bit(0, -1);
    legal |= (uint32_t)!ea << motion_bit(0, 1);
    legal |= (uint32_t)!(nw || (n && we)) << motion_bit(-1, -1);
    legal |= (uint32_t)!(ne || (n && ea)) << motion_bit(-1, 1);
    legal |= (uint32_t)!(sw || (so && we)) << motion_bit(1, -1);
    legal |= (uint32_t)!(se || (so && ea)) << motion_bit(1, 1);
    int my_frame = p.frame + 1;
    uint32_t mask = (p.flags >> F_PERMIT_SHIFT) & 0x1FF;
    if (!(p.flags & F_PERMIT_VALID) || p.permit_frame != my_frame) mask = 0;
    mask |= legal;
    p.flags = (p.flags & ~(0x1FFu << F_PERMIT_SHIFT)) | (mask << F_PERMIT_SHIFT) | F_PERMIT_VALID;
    p.permit_frame = my_frame;
  }
  return blocked;
}

// scrolling.py:437-485 is_possible (one egocentric participant at most)
__device__ __forceinline__ bool is_possible(const Plot& p, int dr, int dc) {
  if (!(p.flags & F_REGISTERED)) return true;
  if (!(p.flags & F_PERMIT_VALID) || p.permit_frame != p.frame) return false;
  return (p.flags >> (F_PERMIT_SHIFT + motion_bit(dr, dc))) & 1;
}

// drapes.py:487-659 _maybe_move.  `ego` is the egocentric sprite (if any).
__device__ __forceinline__ void maybe_move(const Consts& k, Scrolly& d, Plot& p, const Walker& ego, int dr,
                                           int dc, uint32_t& err) {
  if (!d.moved) { d.moved = 1; d.pre_r = d.r; d.pre_c = d.c; }  // :515-517
  if (p.order_valid) {                                         // :523-535
    if (dr != p.o0 && dc != p.o1) { err |= ERR_SCROLL; return; }
    d.r += p.o0;
    d.c += p.o1;
    return;
  }
  if (dr == 0 && dc == 0) return;  // :539-541
  int o0, o1;
  bool go;
  if (!k.have_margins) {  // :551-585
    go = is_possible(p, dr, dc);
    int north = d.r + dr, west = d.c + dc;
    o0 = (0 <= north && north <= k.lim_r) ? dr : 0;
    o1 = (0 <= west && west <= k.lim_c) ? dc : 0;
  } else {  // :592-659
    bool vert = false, horiz = false;
    if (k.ie >= 0 && (p.flags & F_REGISTERED)) {  // :661-687, Sprite.position is the true position
      bool on = on_board(k, ego.vr, ego.vc);
      int old_r = on ? ego.vr : 0, old_c = on ? ego.vc : 0;
      int new_r = old_r + dr, new_c = old_c + dc;
      vert = (old_r > new_r && new_r <= k.margin_n) || (old_r < new_r && new_r >= k.margin_s);
      horiz = (old_c > new_c && new_c <= k.margin_w) || (old_c < new_c && new_c >= k.margin_e);
    }
    if (!(vert || horiz)) return;
    o0 = vert ? dr : 0;
    o1 = horiz ? dc : 0;
    int pr = d.r + o0, pc = d.c + o1;
    go = 0 <= pr && pr <= k.lim_r && 0 <= pc && pc <= k.lim_c && is_possible(p, dr, dc);
  }
  if (go) {
    d.r += o0;
    d.c += o1;
    p.order_valid = 1;  // scrolling.py:530-531 (we are the first to order this frame)
    p.o0 = o0;
    p.o1 = o1;
  }
}

// examples/scrolly_maze.py: 0 N, 1 S, 2 W, 3 E, 4 stay
__device__ __forceinline__ bool sm_motion(int a, int& dr, int& dc) {
  dr = (a == 0) ? -1 : (a == 1) ? 1 : 0;
  dc = (a == 2) ? -1 : (a == 3) ? 1 : 0;
  return (unsigned)a <= 4u;
}

__device__ __forceinline__ uint32_t action_hash(uint64_t seed, uint64_t env, uint64_t t) {
  uint64_t x = seed ^ (env * 0x9E3779B97F4A7C15ull) ^ (t * 0xBF58476D1CE4E5B9ull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}

__device__ __forceinline__ uint32_t pack_pos(int r, int c) { return ((uint32_t)r & 0xFFFFu) | ((uint32_t)c << 16); }
__device__ __forceinline__ int pos_r(uint32_t w) { return (int)(int16_t)(w & 0xFFFFu); }
__device__ __forceinline__ int pos_c(uint32_t w) { return (int)(int16_t)(w >> 16); }

// Compile-time loop: f(IntC<I>) for I in [0, N).  (IntC: an integral constant of our own -- a run-time build has no <type_traits>)
template <int I>
struct IntC {
  static constexpr int value = I;
  constexpr operator int() const { return I; }
};
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    static_for<I + 1, N>(f);
  }
}

// Walker `FIXED` when it is known at compile time, else a select chain.
template <int NS, int FIXED>
__device__ __forceinline__ Walker pick(const Walker (&w)[NS], int dyn) {
  if constexpr (FIXED >= 0) {
    return w[FIXED];
  } else {
    // field-by-field value selects (a conditional struct copy can end up as a select between
    // two addresses of w[], which would pin the whole array in scratch)
    Walker r = w[0];
#pragma unroll
    for (int j = 1; j < NS; ++j) {
      const bool hit = j == dyn;
      r.vr = hit ? w[j].vr : r.vr;
      r.vc = hit ? w[j].vc : r.vc;
      r.vis = hit ? w[j].vis : r.vis;
      r.prior = hit ? w[j].prior : r.prior;
      r.var = hit ? w[j].var : r.var;
    }
    return r;
  }
}

// ---- the persistent launch shape (PS == 3) of the owner-code instance -----------------------------------
// A workgroup stays on its CU; each of its waves is a WORKER that draws work units (64, 32 or 16 consecutive
// environments) until none are left, steps a unit (lane == environment) and streams it.  The state words of a
// worker's NEXT unit travel from HBM straight into its LDS inbox (LDS-DMA, global_load_lds_dword: no VGPR in
// between, nothing for the compiler to wait for) issued in front of the current unit's plane stores: vmcnt counts
// in order and holds at most 63, so after 64 or more plane stores the DMA has landed -- the next logic phase starts
// without a wait instead of queueing fourteen loads behind the chip's plane stores.  Workers have their own inbox,
// coin masks and owner-code buffer and share the staged level; at most `ps_lock` of a workgroup's workers stream at
// a time (a counting semaphore in LDS around the render loop): a CU's write path is saturated by one or two streaming
// waves and loses efficiency with every further concurrent stream (tools/experiments/store_width.hip,
// profiles/r04_tuning.md), while the logic phase wants several waves in flight -- the semaphore decouples the counts.
// (Rounds 4's other persistent shapes -- single-wave workgroups, logic/render wave pairs over a ring of buffers, the
// mask-composing render loop as the persistent body -- were measured slower and are gone: profiles/r04_tuning.md.)
constexpr int PS_IB_ACTION = 15;  // inbox rows: the state words (at most 15), then the tape action
constexpr int PS_IB_ROWS = 16;
constexpr uint32_t PS_SPIN_LIMIT = 1u << 22;  // (x s_sleep 2: seconds) a broken hand-over gives up instead of hanging the GPU

// One row of the inbox: lane i's dword base[i] lands at LDS byte address lds_addr + 4 i.  M0 is written in the
// statement that uses it and restored (the compiler owns it); the base is copied by an SALU instruction so that an
// SGPR pair fresh from v_readfirstlane is never read by the VMEM instruction within the hazard window.
__device__ __forceinline__ void ps_dma_row(const uint32_t* base, uint32_t voff, uint32_t lds_addr) {
  uint32_t keep;
  uint64_t own;
  asm volatile(
      "s_mov_b64 %1, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, %1\n\ts_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(own)
      : "v"(voff), "s"(base), "s"(lds_addr)
      : "memory");
}
// The next ticket of the work counter: a SCALAR atomic (s_atomic_add; gfx950 has them, coherent across the XCDs:
// tools/experiments/satomic_probe.hip).  It travels through the scalar cache path, not behind the CU's queue of plane
// stores, and is waited for on lgkmcnt -- about a microsecond -- so a worker knows its next unit just before it
// needs it and reserves nothing further ahead (what a worker holds in reserve when the tickets run out is the tail).
__device__ __forceinline__ uint32_t ps_ticket(uint32_t* ctr_any) {
  // (uniform by construction; readfirstlane makes it provably so, and the SALU copy inside the statement keeps the scalar
  // memory instruction from reading an SGPR pair a VALU instruction has just written)
  const uint64_t v = reinterpret_cast<uint64_t>(ctr_any);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  uint32_t* const ctr = reinterpret_cast<uint32_t*>(((uint64_t)hi << 32) | lo);
  uint32_t t = 1u;
  uint64_t own;
  asm volatile("s_mov_b64 %1, %2\n\ts_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t), "=&s"(own) : "s"(ctr) : "memory");
  return t;
}
__device__ __forceinline__ uint32_t lds_byte_address(const uint32_t* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)p;
}

// NS sprites.  SR/SC/SL: board rows, cols and layer count when known at
// compile time (0 = take them from Consts); IP/IE: index of the player and of
// the egocentric sprite when known at compile time (-1 = from Consts).
// COOP: small batches.  A workgroup is four or eight waves around one group:
// wave 0 steps it, then all of them share the render loop (iterations round-robin).
// EPI: the render loop also writes the float32 feature-array epilogue (pcx_stream.h).
// CODES: the logic phase paints an owner-code byte per cell (LDS), the render loop
// is one LDS read and one v_perm_b32 per plane (static shape, <= 8 characters).
// PS: 3 = the persistent launch shape of the owner-code instance (above), 0 = a workgroup per group.
// LV: the instance of ONE level whose Consts are compile-time constants (pcx_sm_shipped.h; 0: none, Consts from the
// kernel arguments).  Round 5: with every table entry, stride and z-order bit known, the probes' index arithmetic folds,
// irrelevant probes disappear, and nothing of Consts competes for SGPRs (the run-time instance parked ~2,000 of its
// 15,000 instructions' operands in VGPR lanes: v_readlane / v_writelane).
template <int NS, int SR, int SC, int SL, int IP, int IE, bool UNOCC, bool COOP = false, bool EPI = false,
          bool CODES = false, int PS = 0, int LV = 0>
__global__ __launch_bounds__(COOP ? 8 * WAVE : PS == 3 ? 12 * WAVE : WAVE) void pcx_scrolly_maze_step(const Consts k_arg, const Ptrs P, const StepArgs a,
                                                                  const pcx_buffers out, const stream::EpilogueArgs epi,
                                                                  const crop::FusedCrops* fc_arg) {
  static_assert(LV == 0 || (LV == 9 && PS == 0 && !COOP && !EPI && !CODES && SR != 0) || (LV != 9 && (LV & 1) && PS == 3 && CODES) || (!(LV & 1) && COOP && !EPI),
                "compiled-in constants: odd LV = the persistent owner-code instance, even LV = the cooperative one, 9 = one workgroup per group of any static shape");
  const Consts& k = baked_consts<LV>(k_arg);
  // Fused croppers (include/pcx.h pcx_engine_fuse_croppers): the instances that keep the frame as curtain
  // bit vectors + sprite descriptors (pcx_stream.h's contract) and render a group in the round they step it
  constexpr bool FUSABLE = !CODES && !UNOCC && !EPI;
  static_assert(PS == 0 || (PS == 3 && !COOP && !EPI && !UNOCC && SR != 0 && CODES), "the persistent shape: the static-shape owner-code instance");
  const crop::FusedCrops* const fc = FUSABLE ? fc_arg : nullptr;
  extern __shared__ uint32_t lds_raw[];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x >> 6;
  // Environments per workgroup: a whole wave's worth, except that the cooperative shape may take 32 or
  // 16 (lanes beyond stay idle in the logic phase) so that a batch of a few thousand environments
  // still puts a workgroup on every CU -- its steps are latency-bound, and what is left to shorten is
  // each workgroup's share of the descriptor tasks and of the streaming (StepArgs::envs_per_group).
  const int EPW = COOP ? a.envs_per_group : WAVE;
  const int ngroups = (int)(P.bpad / EPW);
  // Cooperative shape (one group per workgroup, latency-bound): the logic wave asks for its
  // environments' state words before anybody stages the template constants into LDS, so that the two
  // memory round trips run side by side instead of one after the other.
  uint32_t pre_flags = 0, pre_frame = 0, pre_permit = 0, pre_mz = 0, pre_cs = 0, pre_stale = 0, pre_sflags = 0, pre_spos[NS] = {},
           pre_cm[4] = {0, 0, 0, 0};
  int pre_action = PCX_ACTION_NONE;
  if constexpr (COOP) {
    if (threadIdx.x < WAVE) {
      const int col0 = EPW <= 16 ? (int)threadIdx.x >> 2 : (int)threadIdx.x;
      const int64_t env_p = (int64_t)blockIdx.x * EPW + col0;
      if (col0 < EPW && env_p < P.batch) {
        const uint32_t* stp = P.state + env_p;
        const int64_t bpp = P.bpad;
        pre_flags = stp[W_FLAGS * bpp];
        if (a.mode != 1) {
          pre_frame = stp[W_FRAME * bpp]; pre_permit = stp[W_PERMIT_FRAME * bpp];
          pre_mz = stp[W_MAZE * bpp]; pre_cs = stp[W_CASH * bpp];
          pre_stale = stp[W_STALE * bpp]; pre_sflags = stp[W_SFLAGS * bpp];
#pragma unroll
          for (int s = 0; s < NS; ++s) pre_spos[s] = stp[(W_SPOS + s) * bpp];
#pragma unroll
          for (int i = 0; i < 4; ++i) if (i < k.CW) pre_cm[i] = stp[(W_SPOS + NS + i) * bpp];
          if (!a.hashed) pre_action = a.actions[env_p];
        }
      }
    }
  }
  const int R = SR ? SR : k.R, C = SC ? SC : k.C, L = SL ? SL : k.L;
  const int cells = R * C, pitch = (cells + 3) & ~3, QW = pitch >> 2;  // planes start dword-aligned (pad bytes are 0)
  const int FW = SR ? (SR * SC + 31) / 32 + 1 : k.FW;
  // curtain word w of drape d of environment e.  Environment-major with an odd
  // pitch: the logic phase (lane == e, same w) and the render phase (same e,
  // consecutive w) both touch 32 different banks.
  const int FWP = FW | 1;
#define FLAT(d, w, e) (((d) * WAVE + (e)) * FWP + (w))

  Lds l;
  uint32_t* lw = lds_raw + k.lds_walls;
  uint32_t* lb = lds_raw + k.lds_backdrop;
  uint32_t* lr = lds_raw + k.lds_rowstart;
  uint32_t* lc = lds_raw + k.lds_coincol;
  l.walls = lw;
  l.backdrop4 = lb;
  l.coinbits = lr;
  l.rowbase = reinterpret_cast<const uint16_t*>(lc);
  l.cmask = lds_raw + (CODES ? k.lds_cmask_c : k.lds_cmask);
  uint32_t* lm = lds_raw + k.lds_bdmask;
  l.bdmask = lm;

  // ---- stage the shared template constants into LDS (from L2) -------------
  for (int i = threadIdx.x; i < k.PR * k.WPR; i += blockDim.x) lw[i] = P.walls_bits[i];
  for (int i = threadIdx.x; i < QW; i += blockDim.x) lb[i] = P.backdrop4[i];
  if constexpr (!CODES)  // (the CODES instance has no use for the backdrop-character masks)
    for (int i = threadIdx.x; i < k.n_bchars * QW; i += blockDim.x) lm[i] = P.backdrop4[QW + i];
  {
    const uint32_t* rb = reinterpret_cast<const uint32_t*>(P.coin_rowbase);
    for (int i = threadIdx.x; i < k.PR * (k.WPR - 1); i += blockDim.x) lr[i] = P.coin_bits[i];
    for (int i = threadIdx.x; i < (k.PR + 2) / 2; i += blockDim.x) lc[i] = rb[i];
  }
  // NIB (the persistent shapes): owner codes as NIBBLES, two board dwords per LDS dword -- byte b of dword m holds the
  // code of cell 8 m + b in its low and of cell 8 m + 4 + b in its high nibble, so the render loop gets the four byte
  // codes of board dword q = 2 m (+ 1) with one shift and one mask.  Half the LDS per worker: more workers per CU.
  constexpr bool NIB = CODES && PS != 0;
  constexpr int CODE_PITCH = !SR ? 1 : NIB ? (((SR * SC / 4 + 1) / 2) | 1) : ((SR * SC / 4) | 1) + 2;  // dwords per environment, odd: logic (same q, 64
                                                               // environments) and render (same environment,
                                                               // consecutive q) both spread over the banks
  uint32_t* codes = lds_raw + k.lds_codes;
  // persistent shapes: this wave's unit, the one after it (whose state words are on their way into the inbox) and,
  // in the single-wave shape, the ticket in flight for the one after that
  // (the three layout words that depend on the work unit's size are read from the kernel arguments even by the instance with
  // the level's constants compiled in: units of 16 / 32 / 64 environments share it)
  const int ps_mine = PS == 3 ? __builtin_amdgcn_readfirstlane(wave) * k_arg.lds_ps_wave_words : 0;  // this worker's own LDS region
  uint32_t* const ps_inbox = lds_raw + k.lds_ps_inbox + ps_mine;
  // workers: the waves that draw units (PS == 3: every wave; else one per workgroup)
  const uint32_t ps_wid = PS == 3 ? blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave : blockIdx.x;
  const uint32_t ps_nwk = PS == 3 ? gridDim.x * (blockDim.x >> 6) : gridDim.x;
  // (an LDS-address-space pointer: as a generic one its volatile accesses become FLAT instructions, which count on vmcnt)
  typedef __attribute__((address_space(3))) volatile uint32_t lds_volatile_u32;
  lds_volatile_u32* const ps_ring = (lds_volatile_u32*)(lds_raw + k.lds_ps_ring);  // [3]: the streaming semaphore (how many of the workgroup's workers stream)
  const uint32_t ps_n = PS ? P.ps_n : 0u;
  // unit -> its first environment and how many it has
  auto ps_span = [&](uint32_t u, int64_t& e0, int& cnt) {
    const bool small = u >= P.ps_n1;
    const int size = small ? P.ps_tail_unit : P.ps_unit;
    e0 = small ? (int64_t)P.ps_n1 * P.ps_unit + (int64_t)(u - P.ps_n1) * P.ps_tail_unit : (int64_t)u * P.ps_unit;
    const int64_t left = P.bpad - e0;
    cnt = left < size ? (int)left : size;
  };
  // The work counter is sharded (one word per shard, 64 bytes apart): shard x = blockIdx.x % S owns the units
  // congruent to x mod S, its workgroups draw from its own word.  With S = 8 a shard is one XCD as the hardware places
  // workgroups today (block b on XCD b % 8: MI355X_MICROARCH.md) -- an eighth of the contention on each word and the
  // atomic served by the XCD's own L2 slice; nothing depends on the placement but the speed.
  const uint32_t ps_shards = gridDim.x < 8u ? gridDim.x : 8u;
  const uint32_t ps_x = blockIdx.x % ps_shards;
  const uint32_t ps_wpw = PS == 3 ? (blockDim.x >> 6) : 1u;                                  // workers per workgroup
  const uint32_t ps_local = (blockIdx.x / ps_shards) * ps_wpw + (PS == 3 ? (uint32_t)wave : 0u);
  uint32_t ps_u = P.ps_dynamic ? ps_x + ps_shards * ps_local : ps_wid, ps_un = 0;
  uint32_t ps_steal = 0;  // shards this worker has found dry (it draws from shard ps_x + ps_steal)
  // A launch of several steps (StepArgs::n_steps; round 5): every worker keeps ITS units (static round-robin) and walks them
  // step after step -- no ramp and no tail between the steps, the only wait is for the worker's own state stores of the step
  // before (same wave, same CU: coherent through the CU's vL1D) when it wraps around to its first unit.
  int ps_step = 0, ps_step_next = 0;
  bool ps_need_wait = true;
  const bool ps_prof = PS != 0 && P.ps_prof != nullptr;
  uint32_t pt_units = 0, pt_a = 0, pt_b = 0, pt_c = 0, pt_d = 0, pt_mark = 0;
  auto ps_now = [&]() { return ps_prof ? (uint32_t)__builtin_amdgcn_s_memrealtime() : 0u; };
  const uint32_t pt_start = ps_now();
  // the state words of unit `u` (and its tape actions) into the inbox; lanes past the unit's environments stay out
  auto ps_prefetch = [&](uint32_t u_any, int step_any = 0) {
    const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u_any);  // (uniform by construction; now provably)
    const int64_t step = (int64_t)__builtin_amdgcn_readfirstlane(step_any);
    int64_t e0_any;
    int cnt;
    ps_span(u, e0_any, cnt);
    const uint32_t e0_lo = __builtin_amdgcn_readfirstlane((uint32_t)e0_any), e0_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)e0_any >> 32));
    const int64_t e0 = (int64_t)(((uint64_t)e0_hi << 32) | e0_lo);
    cnt = __builtin_amdgcn_readfirstlane(cnt);
    const uint32_t ib = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_byte_address(ps_inbox));
    if (lane < cnt) {
#pragma unroll
      for (int w = 0; w < W_SPOS + NS + 4; ++w)
        if (w < k.NW) ps_dma_row(P.state + (int64_t)w * P.bpad + e0, 4u * lane, ib + (uint32_t)w * (4u * WAVE));
    }
    if (!a.hashed && lane < cnt && e0 + lane < P.batch)
      ps_dma_row(reinterpret_cast<const uint32_t*>(a.actions) + step * a.action_stride + e0, 4u * lane, ib + (uint32_t)PS_IB_ACTION * (4u * WAVE));
  };
  if constexpr (PS != 0) {
    if (threadIdx.x == 0) ps_ring[3] = 0;  // (the streaming semaphore)
    if (ps_u < ps_n) ps_prefetch(ps_u);  // (under the staging of the level below)
  }
  if constexpr (CODES) {
    uint32_t* lbc = lds_raw + k.lds_bdcode;
    for (int i = threadIdx.x; i < QW; i += blockDim.x) lbc[i] = P.backdrop4[QW * (1 + k.n_bchars) + i];
  }
  if constexpr (CODES && PS != 0) {
    // (round 5) the owner-code table is built eight cells at a time: entry x of the first table turns eight cell bits
    // into eight nibble masks in the code buffer's layout (bit b -> low nibble of byte b, bit 4 + b -> its high nibble);
    // the second is the backdrop's owner codes in that layout
    uint32_t* const lut = lds_raw + k.lds_ps_lut;
    for (int x = threadIdx.x; x < 256; x += blockDim.x) {
      const uint32_t lo = (((uint32_t)x & 0xFu) * 0x00204081u) & 0x01010101u, hi = (((uint32_t)x >> 4) * 0x00204081u) & 0x01010101u;
      lut[x] = (lo | (hi << 4)) * 15u;
    }
    const uint32_t* const bd = P.backdrop4 + QW * (1 + k.n_bchars);
    for (int m = threadIdx.x; m < (QW + 1) / 2; m += blockDim.x) lut[256 + m] = bd[2 * m] | (2 * m + 1 < QW ? bd[2 * m + 1] << 4 : 0u);
  }
  __syncthreads();  // LDS constants visible

  // One wave steps a group / unit and then streams it (cooperative shape: wave 0 steps, all waves of the workgroup
  // stream).  Rounds: the groups of a workgroup one after the other; the cooperative shape owns ONE group, and a launch
  // of several steps (StepArgs::n_steps: pcx_engine_step_n / _step_hashed at small batches) walks them here, the state
  // words staying in registers from one step to the next (no state-in / state-out chain, no kernel boundary per step);
  // the persistent shape: a worker's units.  (Rounds 1-4 also carried logic/render wave PAIRS in three variants -- a
  // barrier pipeline, a multi-step pipeline, a ring of LDS counters; all measured slower than what is here and removed in
  // round 5: profiles/r01_tuning.md, r04_tuning.md.)
  for (int round = 0;; ++round) {
  const int64_t g_render = COOP ? (int64_t)blockIdx.x : (int64_t)blockIdx.x + (int64_t)round * gridDim.x;
  const int64_t g_logic = g_render;
  const int tstep = COOP ? round : PS == 3 ? ps_step : 0;  // which of the launch's steps the logic wave is on
  const int coop_steps = a.n_steps > 1 ? a.n_steps : 1;
  bool have_render = g_render < ngroups && (!COOP || round < coop_steps);
  bool have_logic = have_render;
  // the environments this round's logic phase steps / its render phase streams: a group of EPW, or (persistent
  // shape) a work unit
  int64_t env0_logic = g_logic * EPW, env0_render = g_render * EPW;
  int cnt_logic = EPW, cnt_render = EPW;
  if constexpr (PS == 3) {
    if (ps_u >= ps_n) break;
    ps_span(ps_u, env0_logic, cnt_logic);
    env0_render = env0_logic;
    cnt_render = cnt_logic;
    have_logic = have_render = true;
    pt_mark = ps_now();
  } else {
    if (!have_render) break;
  }
  if constexpr (PS != 0) {
    codes = lds_raw + k.lds_ps_buf0 + ps_mine;
    // a buffer: the code table of the unit's environments, then 64 skip flags
    l.skip = codes + k_arg.lds_ps_buf_words - WAVE;
    l.cmask = lds_raw + k.lds_ps_cmask + ps_mine;
  } else {
    l.flat = lds_raw + k.lds_flat;
    l.sdesc = reinterpret_cast<uint2*>(lds_raw + k.lds_sdesc);
    l.skip = CODES ? lds_raw + k.lds_skip_c : lds_raw + k.lds_skip;
  }
  if (wave == 0 || PS == 3) {
  if (have_logic) {
  // ---- phase A (logic wave): lane == environment ---------------------------
  // (cooperative shape with 16 environments per workgroup: FOUR lanes per environment.  All four step it
  // identically -- same loads, same stores -- except that the egocentric sprite's eight scroll-permit probes,
  // half of a step's probes, are shared out two per lane and their verdicts exchanged with two quad shuffles.)
  const bool quad = COOP && EPW <= 16;
  const int col = quad ? lane >> 2 : lane;   // the environment's column in the per-environment LDS arrays
  const int quad_j = quad ? lane & 3 : -1;
  const int64_t env0 = env0_logic;
  const int64_t env = env0 + col;
  const bool live = col < cnt_logic && env < P.batch;
  uint32_t* st = P.state + env;  // word w at st[w * bpad]
  const int64_t bp = P.bpad;
  uint32_t flags = 0;
  bool skip = !live;
  bool do_reset = false;
  int action = PCX_ACTION_NONE;
  // every state word (and the tape action) is requested up front, next to the
  // flags word that decides what happens to the environment: one memory round
  // trip for the whole logic phase instead of two or three in a row
  uint32_t ld_frame = 0, ld_permit = 0, ld_mz = 0, ld_cs = 0, ld_stale = 0, ld_sflags = 0, ld_spos[NS] = {};
  int ld_action = PCX_ACTION_NONE;
  if constexpr (PS != 0) {
    // the unit's state words are in the inbox (ps_prefetch): the first unit's, and whatever was asked for with
    // fewer than 64 plane stores behind it, must be waited for; the pair's logic wave always waits (it has the time)
    if (ps_need_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ps_prof) { const uint32_t t = ps_now(); pt_a += t - pt_mark; pt_mark = t; }
    const uint32_t* const ib = ps_inbox + lane;
    pre_flags = ib[W_FLAGS * WAVE]; pre_frame = ib[W_FRAME * WAVE]; pre_permit = ib[W_PERMIT_FRAME * WAVE];
    pre_mz = ib[W_MAZE * WAVE]; pre_cs = ib[W_CASH * WAVE]; pre_stale = ib[W_STALE * WAVE]; pre_sflags = ib[W_SFLAGS * WAVE];
#pragma unroll
    for (int s = 0; s < NS; ++s) pre_spos[s] = ib[(W_SPOS + s) * WAVE];
#pragma unroll
    for (int i = 0; i < 4; ++i) pre_cm[i] = i < k.CW ? ib[(W_SPOS + NS + i) * WAVE] : 0u;
    pre_action = a.hashed ? PCX_ACTION_NONE : (int)ib[PS_IB_ACTION * WAVE];
  }
  if constexpr (COOP) {
    // (steps after the launch's first: their tape action; the state words are in the registers the step before left)
    if (round > 0 && live && !a.hashed) pre_action = a.actions[(int64_t)tstep * a.action_stride + env];
  }
  if constexpr (COOP || PS != 0) {  // (asked for at the top of the kernel / taken from the inbox)
    flags = pre_flags; ld_frame = pre_frame; ld_permit = pre_permit; ld_mz = pre_mz; ld_cs = pre_cs;
    ld_stale = pre_stale; ld_sflags = pre_sflags; ld_action = pre_action;
#pragma unroll
    for (int s = 0; s < NS; ++s) ld_spos[s] = pre_spos[s];
  }
  if (live) {
    if constexpr (!COOP && PS == 0) flags = st[W_FLAGS * bp];
    if (!COOP && PS == 0 && a.mode != 1) {
      ld_frame = st[W_FRAME * bp]; ld_permit = st[W_PERMIT_FRAME * bp];
      ld_mz = st[W_MAZE * bp]; ld_cs = st[W_CASH * bp];
      ld_stale = st[W_STALE * bp]; ld_sflags = st[W_SFLAGS * bp];
#pragma unroll
      for (int s = 0; s < NS; ++s) ld_spos[s] = st[(W_SPOS + s) * bp];
      if (!a.hashed) ld_action = a.actions[(int64_t)tstep * a.action_stride + env];
    }
    if (a.mode == 1) {
      do_reset = a.reset_mask ? a.reset_mask[env] != 0 : true;
      skip = !do_reset;
    } else if (flags & F_OVER) {
      do_reset = a.auto_reset != 0;
      skip = !do_reset;
      if (skip) { out.reward[env] = 0; out.reward_set[env] = 0; out.discount[env] = 0.0f; }  // a finished environment left alone reports an empty step (pcx.h)
    } else {
      action = a.hashed ? (int)(action_hash(a.seed, (uint64_t)(a.env_offset + env), (uint64_t)(a.t + tstep)) %
                               (uint32_t)k.n_actions)
                        : ld_action;
    }
  }
  if (!skip) {
    Walker w[NS];
    Scrolly maze, cash;
    Plot p;
    uint32_t stale;
    uint32_t err = do_reset ? 0u : (flags >> F_ERR_SHIFT) & 7u;  // sticky within an episode
    bool coins_dirty = false;
    // load (or rebuild) the state words
    uint32_t sflags, spos[NS], mz, cs;
    if (do_reset) {  // engine.py:520-581 its_showtime: fresh template state
      p.frame = (int)k.init[W_FRAME];
      flags = k.init[W_FLAGS];
      p.permit_frame = (int)k.init[W_PERMIT_FRAME];
      mz = k.init[W_MAZE];
      cs = k.init[W_CASH];
      stale = k.init[W_STALE];
      sflags = k.init[W_SFLAGS];
#pragma unroll
      for (int s = 0; s < NS; ++s) spos[s] = k.init[W_SPOS + s];
      for (int i = 0; i < k.CW; ++i) {
        int left = k.n_coins - 32 * i;
        l.cmask[i * WAVE + col] = left >= 32 ? 0xFFFFFFFFu : ((1u << left) - 1u);
      }
      coins_dirty = true;
      action = PCX_ACTION_NONE;
    } else {
      p.frame = (int)ld_frame;
      p.permit_frame = (int)ld_permit;
      mz = ld_mz;
      cs = ld_cs;
      stale = ld_stale;
      sflags = ld_sflags;
#pragma unroll
      for (int s = 0; s < NS; ++s) spos[s] = ld_spos[s];
      // (a launch's later steps in the cooperative shape: the coin masks are still in LDS, as the step before left them)
      if (!COOP || round == 0)
        for (int i = 0; i < k.CW; ++i) l.cmask[i * WAVE + col] = ((COOP || PS != 0) && i < 4) ? (i == 0 ? pre_cm[0] : i == 1 ? pre_cm[1] : i == 2 ? pre_cm[2] : pre_cm[3]) : st[(W_SPOS + NS + i) * bp];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      w[s].vr = pos_r(spos[s]);
      w[s].vc = pos_c(spos[s]);
      uint32_t f = sflags >> (4 * s);  // four bits per sprite (MAX_NS * 4 <= 32)
      w[s].vis = f & 1;
      w[s].prior = (f >> 1) & 1;
      w[s].var = (f >> 2) & 1;
    }
    maze = {pos_r(mz), pos_c(mz), 0, 0, 0};
    cash = {pos_r(cs), pos_c(cs), 0, 0, 0};
    p.flags = flags & ~(F_OVER | F_ERR_MASK);
    p.order_valid = 0; p.o0 = 0; p.o1 = 0;
    p.reward_set = 0; p.reward = 0; p.game_over = 0; p.discount = 1.0f;  // plot.py:98-104

    // ---- Engine.play(): engine.py:698-735 --------------------------------
    p.frame += 1;
    int dr, dc;
    const bool moves = sm_motion(action, dr, dc);
    Snap<NS> sn;  // the repaint every entity of this frame's groups 0/1 sees
#pragma unroll
    for (int s = 0; s < NS; ++s) sn.cell[s] = paint_cell(k, w[s]);
    sn.cash_r = cash.r; sn.cash_c = cash.c; sn.stale = stale;

    if (!(a.debug & 1)) {
    // group 0: MazeDrape.update (scrolly_maze.py:317-329)
    if (moves) maybe_move(k, maze, p, pick<NS, IE>(w, k.ie), dr, dc, err);
    sn.maze_r = maze.r; sn.maze_c = maze.c;  // repaint #1: walls already scrolled

    // group 1: sprites in insertion order, all reading repaint #1
    // One copy of the body per sprite index (a macro, not a loop or a lambda:
    // the compiler must see compile-time indices into w[] to keep it in VGPRs).
    // (instances that know which sprite is the player and which the egocentric one compile only
    // the one body each sprite runs, and the scroll-permit probes for the egocentric sprite alone)
#define PCX_SM_SPRITE(s)                                                                         \
  if constexpr ((s) < NS) {                                                                      \
    constexpr int ego_s = IE >= 0 ? (int)((s) == IE) : -1;                                       \
    if (IP >= 0 ? (s) != IP : k.prog[s] == PCX_PROG_SM_PATROLLER) { /* scrolly_maze.py:284-305 */ \
      const bool walks = !(p.frame & 1); /* odd frames: _stay */                                 \
      int mdc = 0;                                                                               \
      if (walks) {                                                                               \
        /* drapes.py:405-411 pattern_position_prescroll on the walls drape */                    \
        if (!maze.moved) { maze.pre_r = maze.r; maze.pre_c = maze.c; }                           \
        int pr = w[s].vr + maze.pre_r, pc = w[s].vc + maze.pre_c + (w[s].var ? 1 : -1);          \
        if (pr < 0) pr += k.PR; /* numpy negative-index wrap */                                  \
        if (pc < 0) pc += k.PC;                                                                  \
        if (wall_at(k, l, pr, pc, err)) w[s].var ^= 1;                                           \
        mdc = w[s].var ? 1 : -1;                                                                 \
      }                                                                                          \
      mw_move<NS, ego_s>(k, l, sn, s, w[s], p, 0, mdc, col, err, quad_j); /* one call site for both */  \
      if (walks) {                                                                               \
        const Walker pl = pick<NS, IP>(w, k.ip);                                                 \
        if (w[s].vr == pl.vr && w[s].vc == pl.vc) { p.game_over = 1; p.discount = 0.0f; }        \
      }                                                                                          \
    } else { /* PlayerSprite.update (scrolly_maze.py:259-271) */                                 \
      if (moves) mw_move<NS, ego_s>(k, l, sn, s, w[s], p, dr, dc, col, err, quad_j);                    \
    }                                                                                            \
  }
    bool group1_done = false;
    if constexpr (COOP && IP >= 0 && IE == IP && NS == 4) {
      if (quad) {
        // Four lanes per environment: lane j steps sprite j -- the three patrollers and the player move at
        // once, each against the same repaint (they are one update group) -- then all four probe two of the
        // player's eight neighbours for its scroll permits, and the quad exchanges what changed.
        group1_done = true;
        const int qbase = lane & ~3;
        const bool is_player = quad_j == IP;
        const WalkerConsts wc = walker_consts<NS>(k, quad_j);
        Walker me = pick<NS, -1>(w, quad_j);
        const Walker pl_before = w[IP];  // (the player updates last: the patrollers' catch test sees it where it was)
        const bool walks = !(p.frame & 1);
        int mdr = is_player ? dr : 0, mdc = is_player ? dc : 0;
        if (!is_player && walks) {  // scrolly_maze.py:284-305
          if (!maze.moved) { maze.pre_r = maze.r; maze.pre_c = maze.c; }
          int pr = me.vr + maze.pre_r, pc = me.vc + maze.pre_c + (me.var ? 1 : -1);
          if (pr < 0) pr += k.PR;
          if (pc < 0) pc += k.PC;
          if (wall_at(k, l, pr, pc, err)) me.var ^= 1;
          mdc = me.var ? 1 : -1;
        }
        if (!maze.moved) { maze.pre_r = maze.r; maze.pre_c = maze.c; }  // (what the patroller lanes noted, on every lane)
        const bool calls_move = !is_player || moves;
        if (calls_move) {  // sprites.py:356-389 _move
          if (is_player) p.flags |= F_REGISTERED;
          if (p.order_valid) {
            teleport(k, me, me.vr - p.o0, me.vc - p.o1);
            if (is_player && p.o0 != mdr && p.o1 != mdc) err |= ERR_SCROLL;
          }
          if (!check_motion_lane<NS>(k, l, sn, wc, me, mdr, mdc, col, err)) teleport(k, me, me.vr + mdr, me.vc + mdc);
        }
        if (!is_player && walks && me.vr == pl_before.vr && me.vc == pl_before.vc) { p.game_over = 1; p.discount = 0.0f; }
        // everybody learns everybody's new state
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
          w[s2].vr = __shfl(me.vr, qbase | s2); w[s2].vc = __shfl(me.vc, qbase | s2);
          w[s2].vis = __shfl(me.vis, qbase | s2); w[s2].prior = __shfl(me.prior, qbase | s2); w[s2].var = __shfl(me.var, qbase | s2);
        }
        // sprites.py:456-477: the player's permits, two neighbours per lane (N S | W E | NW NE | SW SE)
        uint32_t nb = 0;
        if (moves) {
          const int a0 = quad_j == 0 ? -1 : quad_j == 1 ? 0 : quad_j == 2 ? -1 : 1, b0 = quad_j == 0 ? 0 : -1;
          const int a1 = quad_j == 0 ? 1 : quad_j == 1 ? 0 : quad_j == 2 ? -1 : 1, b1 = quad_j == 0 ? 0 : 1;
          nb = (uint32_t)blocked_at<NS>(k, l, sn, IP, w[IP], a0, b0, col, err) << (2 * quad_j);
          nb |= (uint32_t)blocked_at<NS>(k, l, sn, IP, w[IP], a1, b1, col, err) << (2 * quad_j + 1);
        }
        nb |= err << 8 | (uint32_t)p.game_over << 11;
        nb |= (uint32_t)__shfl_xor((int)nb, 1);
        nb |= (uint32_t)__shfl_xor((int)nb, 2);
        err |= (nb >> 8) & 7u;
        if ((nb >> 11) & 1u) { p.game_over = 1; p.discount = 0.0f; }
        p.flags = (uint32_t)__shfl((int)p.flags, qbase | IP);  // (only the player's lane registered)
        if (moves) {
          const bool n = nb & 1u, so = (nb >> 1) & 1u, we = (nb >> 2) & 1u, ea = (nb >> 3) & 1u;
          const bool nw = (nb >> 4) & 1u, ne = (nb >> 5) & 1u, sw = (nb >> 6) & 1u, se = (nb >> 7) & 1u;
          uint32_t legal = 1u << motion_bit(0, 0);
          legal |= (uint32_t)!n << motion_bit(-1, 0) | (uint32_t)!so << motion_bit(1, 0);
          legal |= (uint32_t)!we << motion_bit(0, -1) | (uint32_t)!ea << motion_bit(0, 1);
          legal |= (uint32_t)!(nw || (n && we)) << motion_bit(-1, -1) | (uint32_t)!(ne || (n && ea)) << motion_bit(-1, 1);
          legal |= (uint32_t)!(sw || (so && we)) << motion_bit(1, -1) | (uint32_t)!(se || (so && ea)) << motion_bit(1, 1);
          const int my_frame = p.frame + 1;
          uint32_t mask = (p.flags >> F_PERMIT_SHIFT) & 0x1FF;
          if (!(p.flags & F_PERMIT_VALID) || p.permit_frame != my_frame) mask = 0;
          mask |= legal;
          p.flags = (p.flags & ~(0x1FFu << F_PERMIT_SHIFT)) | (mask << F_PERMIT_SHIFT) | F_PERMIT_VALID;
          p.permit_frame = my_frame;
        }
      }
    }
    if (!group1_done) {
    PCX_SM_SPRITE(0) PCX_SM_SPRITE(1) PCX_SM_SPRITE(2) PCX_SM_SPRITE(3) PCX_SM_SPRITE(4) PCX_SM_SPRITE(5)
    }
#undef PCX_SM_SPRITE

    // group 2: CashDrape.update (scrolly_maze.py:341-364)
    {
      const Walker pl = pick<NS, IP>(w, k.ip);
      bool on = on_board(k, pl.vr, pl.vc);
      // pattern_position_prescroll: this drape has not scrolled yet this frame
      int pr = (on ? pl.vr : 0) + cash.r, pc = (on ? pl.vc : 0) + cash.c;
      if (pr < 0) pr += k.PR;
      if (pc < 0) pc += k.PC;
      if ((unsigned)pr >= (unsigned)k.PR || (unsigned)pc >= (unsigned)k.PC) {
        err |= ERR_INDEX;
      } else {
        int id = coin_id_at(k, l, pr, pc);
        if (id >= 0 && coin_alive(l, col, id)) {
          p.reward_set = 1;
          p.reward += 100;
          l.cmask[(id >> 5) * WAVE + col] &= ~(1u << (id & 31));
          coins_dirty = true;
          stale = (uint32_t)id;  // still drawn until the curtain is refreshed
          uint32_t any = 0;
          for (int i = 0; i < k.CW; ++i) any |= l.cmask[i * WAVE + col];
          if (!any) { p.game_over = 1; p.discount = 0.0f; }
        }
      }
      if (moves) {
        maybe_move(k, cash, p, pick<NS, IE>(w, k.ie), dr, dc, err);
        stale = STALE_NONE;  // every _maybe_move path ends in _update_curtain
      } else if (action == 5) {
        p.game_over = 1; p.discount = 0.0f;
      }
    }

    }  // debug & 1
    if constexpr (COOP) {
      // Cooperative shape: the render descriptors are built by ALL waves of the workgroup after
      // this phase (one (environment, row) task per col -- the step is latency-bound at these batch
      // sizes and this wave has done its share); they need the two corners, the stale coin and the
      // sprites' cells.  What the row loop would have found wrong is known from the corner alone.
      uint32_t* const fp = lds_raw + k.lds_fparams;
      fp[0 * WAVE + col] = pack_pos(maze.r, maze.c);
      fp[1 * WAVE + col] = pack_pos(cash.r, cash.c);
      fp[2 * WAVE + col] = stale;
#pragma unroll
      for (int s = 0; s < NS; ++s) fp[(3 + s) * WAVE + col] = (uint32_t)paint_cell(k, w[s]);
      if (!(maze.r >= 0 && maze.r + R <= k.PR && maze.c >= 0 && maze.c + C <= k.PC)) err |= ERR_INDEX;
    }
    if (!COOP && !(a.debug & 4)) {
    // ---- render descriptors for phase B ------------------------------------
    // Both curtains as flat cell-bit vectors (bit i = cell i), so that phase B
    // finds the 4 bits of a board dword with one aligned LDS read.
    {
      constexpr int ACC = SR ? (SR * SC + 31) / 32 + 1 : 1;
      uint32_t accw[ACC], accc[ACC];
      if constexpr (SR != 0) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) accw[i] = accc[i] = 0;
      } else {
        for (int i = 0; i < FW; ++i) l.flat[FLAT(0, i, col)] = l.flat[FLAT(1, i, col)] = 0;
      }
      const uint32_t cmaskC = C >= 32 ? 0xFFFFFFFFu : ((1u << C) - 1u);
#pragma unroll
      for (int r = 0; r < (SR ? SR : R); ++r) {
        uint32_t wbits, cbits;
        if (!curtain_row_bits(k, l, col, maze.r, maze.c, cash.r, cash.c, stale, r, C, cmaskC, wbits, cbits)) err |= ERR_INDEX;
        const int off = r * C, wi = off >> 5, sh = off & 31;
        if constexpr (SR != 0) {
          accw[wi] |= wbits << sh;
          accc[wi] |= cbits << sh;
          if (sh + SC > 32) {
            accw[wi + 1] |= wbits >> (32 - sh);
            accc[wi + 1] |= cbits >> (32 - sh);
          }
        } else {
          l.flat[FLAT(0, wi, col)] |= wbits << sh;
          l.flat[FLAT(1, wi, col)] |= cbits << sh;
          if (sh + C > 32) {
            l.flat[FLAT(0, wi + 1, col)] |= wbits >> (32 - sh);
            l.flat[FLAT(1, wi + 1, col)] |= cbits >> (32 - sh);
          }
        }
      }
      // Resolve occlusion between the two curtains now (engine.py:751-757
      // paints back to front, so the one in front wins where both are set).
      const bool cash_in_front = (k.above[NS] >> (NS + 1)) & 1;
      if (a.export_curtains) {  // raw curtains for drape-tracking croppers
        const int ms = P.maze_slot, cs2 = 1 - P.maze_slot;
        if constexpr (SR != 0) {
#pragma unroll
          for (int i = 0; i < ACC; ++i) {
            P.curtains[((size_t)ms * ACC + i) * bp + env] = accw[i];
            P.curtains[((size_t)cs2 * ACC + i) * bp + env] = accc[i];
          }
        } else {
          for (int i = 0; i < FW; ++i) {
            P.curtains[((size_t)ms * FW + i) * bp + env] = l.flat[FLAT(0, i, col)];
            P.curtains[((size_t)cs2 * FW + i) * bp + env] = l.flat[FLAT(1, i, col)];
          }
        }
      }
      if constexpr (CODES && PS != 0 && SR != 0) {
        // (round 5) owner codes eight cells at a time: the cells' wall bits and coin bits are one byte each of the flat
        // vectors; a 256-entry table turns a byte into the eight nibble masks of one code dword, and the curtains are laid
        // over the backdrop's nibbles back to front with one v_bfi_b32 each (~9 instructions per eight cells; the
        // arithmetic expansion below costs ~40)
        const uint32_t* const lut = lds_raw + k.lds_ps_lut;
        const uint32_t wcode8 = (uint32_t)k.lay_drape[0] * 0x11111111u, ccode8 = (uint32_t)k.lay_drape[1] * 0x11111111u;
        constexpr int NM = (SR * SC / 4 + 1) / 2;
        // (in chunks: a chunk's table reads are all in flight before its first code is composed -- the compiler cannot tell
        // that the code buffer and the tables never overlap, so read / write pairs written one after the other stay that
        // way, one LDS round trip each)
        constexpr int CH = 13;
#pragma unroll
        for (int m0 = 0; m0 < NM; m0 += CH) {
          uint32_t mw[CH], mc[CH], bd[CH];
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            const int m = m0 + j < NM ? m0 + j : NM - 1;
            mw[j] = lut[(accw[m >> 2] >> (8 * (m & 3))) & 0xFFu];
            mc[j] = lut[(accc[m >> 2] >> (8 * (m & 3))) & 0xFFu];
            bd[j] = lut[256 + m];
          }
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            if (m0 + j >= NM) break;
            uint32_t code = bd[j];
            if (cash_in_front) { code = (code & ~mw[j]) | (wcode8 & mw[j]); code = (code & ~mc[j]) | (ccode8 & mc[j]); }
            else               { code = (code & ~mc[j]) | (ccode8 & mc[j]); code = (code & ~mw[j]) | (wcode8 & mw[j]); }
            codes[col * CODE_PITCH + m0 + j] = code;
          }
        }
      } else if constexpr (CODES) {
        // owner codes of the backdrop, then the two curtains painted over them
        // (one in front of the other where both are set): four cells per dword
        const uint32_t* const bdcode = lds_raw + k.lds_bdcode;
        const uint32_t wcode4 = (uint32_t)k.lay_drape[0] * 0x01010101u, ccode4 = (uint32_t)k.lay_drape[1] * 0x01010101u;
        uint32_t code_even = 0;  // (NIB: the codes of board dword 2 m wait for those of 2 m + 1)
#pragma unroll
        for (int q = 0; q < SR * SC / 4; ++q) {
          const int i = (4 * q) >> 5, sh = (4 * q) & 31;
          const uint32_t ww = cash_in_front ? accw[i] & ~accc[i] : accw[i];
          const uint32_t cc = cash_in_front ? accc[i] : accc[i] & ~accw[i];
          uint32_t mw = (((ww >> sh) & 0xFu) * 0x00204081u) & 0x01010101u, mc = (((cc >> sh) & 0xFu) * 0x00204081u) & 0x01010101u;
          uint32_t hw = mw << 8, hc = mc << 8;
          asm("" : "+v"(hw), "+v"(hc));  // keep (x << 8) - x from becoming a quarter-rate multiply
          mw = hw - mw; mc = hc - mc;
          uint32_t code = bdcode[q];
          code = (code & ~mw) | (wcode4 & mw);
          code = (code & ~mc) | (ccode4 & mc);
          if constexpr (NIB) {
            if (q & 1) codes[col * CODE_PITCH + (q >> 1)] = code_even | (code << 4);
            else if (q == SR * SC / 4 - 1) codes[col * CODE_PITCH + (q >> 1)] = code;
            else code_even = code;
          } else {
            codes[col * CODE_PITCH + q] = code;
          }
        }
      } else if constexpr (SR != 0) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) {
          const uint32_t ww = cash_in_front ? accw[i] & ~accc[i] : accw[i];
          const uint32_t cc = cash_in_front ? accc[i] : accc[i] & ~accw[i];
          l.flat[FLAT(0, i, col)] = ww;
          l.flat[FLAT(1, i, col)] = cc;
        }
      } else {
        for (int i = 0; i < FW; ++i) {
          const uint32_t ww = l.flat[FLAT(0, i, col)], cc = l.flat[FLAT(1, i, col)];
          if constexpr (UNOCC) {  // unoccluded layers are the raw curtains (rendering.py:236-278)
            (lds_raw + k.lds_flatraw)[FLAT(0, i, col)] = ww;
            (lds_raw + k.lds_flatraw)[FLAT(1, i, col)] = cc;
          }
          l.flat[FLAT(0, i, col)] = cash_in_front ? ww & ~cc : ww;
          l.flat[FLAT(1, i, col)] = cash_in_front ? cc : cc & ~ww;
        }
      }
    }
    // A sprite is painted iff nothing in front of it covers its cell; a sprite
    // that is painted takes its cell away from both curtains.  After this,
    // every board cell belongs to exactly one of {a sprite, a curtain, the
    // backdrop} and phase B needs no z-order.
    if constexpr (CODES) {
      // the sprites, back to front (engine.py:751-757): a sprite takes its cell
      // unless a curtain in front of it holds it; one byte each
      uint8_t* const mine = reinterpret_cast<uint8_t*>(codes + col * CODE_PITCH);
#pragma unroll
      for (int i = 0; i < NS; ++i) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          if (k.sprite_by_z[i] != s) continue;
          const int cell = paint_cell(k, w[s]);
          if (cell < 0) continue;
          const uint32_t ab = k.above[s];
          if constexpr (NIB) {  // the cell's nibble: dword cell / 8, byte cell % 4, high half for cells 4..7 of the eight
            uint32_t* const pw = codes + col * CODE_PITCH + (cell >> 3);
            const int sh = 8 * (cell & 3) + ((cell & 4) ? 4 : 0);
            const uint32_t word = *pw, top = (word >> sh) & 0xFu;
            const bool covered = (((ab >> NS) & 1) && top == (uint32_t)k.lay_drape[0]) ||
                                 (((ab >> (NS + 1)) & 1) && top == (uint32_t)k.lay_drape[1]);
            if (!covered) *pw = (word & ~(0xFu << sh)) | ((uint32_t)k.lay_sprite[s] << sh);
          } else {
            const uint32_t top = mine[cell];
            const bool covered = (((ab >> NS) & 1) && top == (uint32_t)k.lay_drape[0]) ||
                                 (((ab >> (NS + 1)) & 1) && top == (uint32_t)k.lay_drape[1]);
            if (!covered) mine[cell] = (uint8_t)k.lay_sprite[s];
          }
        }
      }
    } else {
      int cellv[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) cellv[s] = paint_cell(k, w[s]);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int cell = cellv[s];
        bool shown = cell >= 0;
        if (shown) {
          const uint32_t ab = k.above[s];
#pragma unroll
          for (int j = 0; j < NS; ++j)
            if (j != s && ((ab >> j) & 1) && cellv[j] == cell) shown = false;
          const int wi = cell >> 5, sh = cell & 31;
#pragma unroll
          for (int dd = 0; dd < 2; ++dd)
            if (((ab >> (NS + dd)) & 1) && ((l.flat[FLAT(dd, wi, col)] >> sh) & 1)) shown = false;
          if (shown) {
            l.flat[FLAT(0, wi, col)] &= ~(1u << sh);
            l.flat[FLAT(1, wi, col)] &= ~(1u << sh);
          }
        }
        l.sdesc[s * WAVE + col] = make_uint2(shown ? (uint32_t)(cell >> 2) : 0xFFFFFFFFu, 0xFFu << ((cell & 3) * 8));
        if constexpr (UNOCC)
          reinterpret_cast<uint2*>(lds_raw + k.lds_sdescraw)[s * WAVE + col] =
              make_uint2(cell >= 0 ? (uint32_t)(cell >> 2) : 0xFFFFFFFFu, 0xFFu << ((cell & 3) * 8));
      }
    }

    }  // debug & 4
    // ---- _apply_and_clear_plot (engine.py:761-847) + state write-back ------
    flags = p.flags | (p.game_over ? F_OVER : 0u) | ((err & 7u) << F_ERR_SHIFT);
    st[W_FRAME * bp] = (uint32_t)p.frame;
    st[W_FLAGS * bp] = flags;
    st[W_PERMIT_FRAME * bp] = (uint32_t)p.permit_frame;
    st[W_MAZE * bp] = pack_pos(maze.r, maze.c);
    st[W_CASH * bp] = pack_pos(cash.r, cash.c);
    st[W_STALE * bp] = stale;
    uint32_t sf = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      st[(W_SPOS + s) * bp] = pack_pos(w[s].vr, w[s].vc);
      sf |= ((uint32_t)w[s].vis | ((uint32_t)w[s].prior << 1) | ((uint32_t)w[s].var << 2)) << (4 * s);
      const bool on = on_board(k, w[s].vr, w[s].vc);
      P.track[k.tmpl_index[s] * bp + env] = (on ? w[s].vr : 0) | ((on ? w[s].vc : 0) << 8) | (w[s].vis << 16) |
                              ((int)do_reset << 24);
    }
    st[W_SFLAGS * bp] = sf;
    if constexpr (COOP) {
      // the next step of this launch (if any) starts from these registers, not from memory
      pre_flags = flags; pre_frame = (uint32_t)p.frame; pre_permit = (uint32_t)p.permit_frame;
      pre_mz = pack_pos(maze.r, maze.c); pre_cs = pack_pos(cash.r, cash.c); pre_stale = stale; pre_sflags = sf;
#pragma unroll
      for (int s = 0; s < NS; ++s) pre_spos[s] = pack_pos(w[s].vr, w[s].vc);
#pragma unroll
      for (int i = 0; i < 4; ++i) if (i < k.CW) pre_cm[i] = l.cmask[i * WAVE + col];
    }
    if constexpr (FUSABLE) {
      // (a cropper may follow the maze or the cash drape: in the single-wave shapes this lane has exported the raw
      // curtains above and takes the median here; the cooperative shape exports them later, from all waves, and moves
      // windows that follow a drape after that export -- below, "late windows")
      const stream::CurtainSrc csrc{P.curtains, bp, FW, R, C};
      if (fc && !(COOP && fc->drapes))  // fused croppers: the windows follow this step's positions (cropping.py:393-426)
        stream::move_fused_windows(fc, [&](int ti) {  // ti: the TEMPLATE's sprite index
          int32_t t = 0;
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            const bool on = on_board(k, w[s].vr, w[s].vc);
            const int32_t tw = (on ? w[s].vr : 0) | ((on ? w[s].vc : 0) << 8) | (w[s].vis << 16);
            t = k.tmpl_index[s] == ti ? tw : t;
          }
          return t;
        }, p.frame == 0, env, col, lds_raw + k.lds_wcorner, COOP ? nullptr : &csrc);
    }
    if (coins_dirty)
      for (int i = 0; i < k.CW; ++i) st[(W_SPOS + NS + i) * bp] = l.cmask[i * WAVE + col];
    out.reward[env] = p.reward;
    out.reward_set[env] = (uint8_t)p.reward_set;
    out.discount[env] = p.discount;
    out.done[env] = (uint8_t)p.game_over;
    out.frame[env] = p.frame;
    out.error[env] = (uint8_t)err;
  }
  if (quad) l.skip[lane] = 1;  // (columns past the group's environments: nobody's)
  l.skip[col] = skip;
  }  // have_logic
  } else if constexpr (COOP) {
    // (the other waves, while wave 0 steps the group: an empty slate for the curtains)
    for (int i = (int)threadIdx.x - WAVE; i < 2 * WAVE * FWP; i += (int)blockDim.x - WAVE) l.flat[i] = 0;
  }
  if constexpr (PS == 0) {
    __syncthreads();
  } else {
    asm volatile("" ::: "memory");  // one wave: its LDS instructions execute in order
  }
  if constexpr (COOP) {
    if (have_render && !(a.debug & 4)) {
      const uint32_t* const fp = lds_raw + k.lds_fparams;
      const uint32_t cmaskC = C >= 32 ? 0xFFFFFFFFu : ((1u << C) - 1u);
      // the curtains' rows: one (environment, row) task per lane, OR-ed into the flat bit vectors
      for (int task = (int)threadIdx.x; task < EPW * R; task += (int)blockDim.x) {
        const int e = task / R, r = task - e * R;
        if (l.skip[e]) continue;
        const uint32_t mz = fp[0 * WAVE + e], cs = fp[1 * WAVE + e];
        uint32_t wbits, cbits;
        curtain_row_bits(k, l, e, pos_r(mz), pos_c(mz), pos_r(cs), pos_c(cs), fp[2 * WAVE + e], r, C, cmaskC, wbits, cbits);
        const int off = r * C, wi = off >> 5, sh = off & 31;
        if (wbits) atomicOr(&l.flat[FLAT(0, wi, e)], wbits << sh);
        if (cbits) atomicOr(&l.flat[FLAT(1, wi, e)], cbits << sh);
        if (sh + C > 32) {
          if (wbits >> (32 - sh)) atomicOr(&l.flat[FLAT(0, wi + 1, e)], wbits >> (32 - sh));
          if (cbits >> (32 - sh)) atomicOr(&l.flat[FLAT(1, wi + 1, e)], cbits >> (32 - sh));
        }
      }
      __syncthreads();
      const bool cash_front = (k.above[NS] >> (NS + 1)) & 1;
      if (a.export_curtains) {  // raw curtains for drape-tracking croppers
        const int ms = P.maze_slot, cs2 = 1 - P.maze_slot;
        const int64_t env_base = g_render * EPW;
        for (int task = (int)threadIdx.x; task < WAVE * FW; task += (int)blockDim.x) {
          const int i = task / WAVE, e = task - i * WAVE;  // consecutive lanes = consecutive environments: coalesced
          if (e >= EPW || env_base + e >= P.batch || l.skip[e]) continue;
          P.curtains[((size_t)ms * FW + i) * P.bpad + env_base + e] = l.flat[FLAT(0, i, e)];
          P.curtains[((size_t)cs2 * FW + i) * P.bpad + env_base + e] = l.flat[FLAT(1, i, e)];
        }
        __syncthreads();
      }
      if constexpr (FUSABLE) {
        if (fc && fc->drapes) {
          // late windows (round 4): the raw curtains of the workgroup's environments are in memory now; one lane per
          // environment takes the medians (pcx_stream.h curtain_centroid) and moves every window exactly as the logic
          // phase of the single-wave shapes does -- the sprites' track words and the frame counter it needs were
          // written there by this workgroup's logic wave
          const int64_t envw = g_render * EPW + (int64_t)threadIdx.x;
          if ((int)threadIdx.x < EPW && envw < P.batch && !l.skip[threadIdx.x]) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            const stream::CurtainSrc csrc{P.curtains, P.bpad, FW, R, C};
            stream::move_fused_windows(fc, [&](int ti) { return P.track[(int64_t)ti * P.bpad + envw] & 0x1FFFF; },
                                       out.frame[envw] == 0, envw, (int)threadIdx.x, lds_raw + k.lds_wcorner, &csrc);
          }
          __syncthreads();
        }
      }
      if (fc) {  // the fused croppers' windows paint the curtains in index order: resolve them first
        for (int task = (int)threadIdx.x; task < EPW * FW; task += (int)blockDim.x) {
          const int e = task / FW, i = task - e * FW;
          const uint32_t ww = l.flat[FLAT(0, i, e)], cc = l.flat[FLAT(1, i, e)];
          l.flat[FLAT(0, i, e)] = cash_front ? ww & ~cc : ww;
          l.flat[FLAT(1, i, e)] = cash_front ? cc : cc & ~ww;
        }
        __syncthreads();
      }
      // the sprites, one (environment, sprite) task per lane: painted iff nothing in front covers the
      // cell (a curtain occluded by the other curtain still stands for "something covers it"); a
      // painted sprite takes its cell from both curtains.  Order-free, as in pcx_stream.h.
      for (int task = (int)threadIdx.x; task < EPW * NS; task += (int)blockDim.x) {
        const int e = task / NS, s = task - e * NS;
        if (l.skip[e]) continue;
        int cellv[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) cellv[j] = (int)fp[(3 + j) * WAVE + e];
        int cell = cellv[0];
        uint32_t ab = k.above[0];
#pragma unroll
        for (int j = 1; j < NS; ++j) { cell = s == j ? cellv[j] : cell; ab = s == j ? k.above[j] : ab; }
        bool shown = cell >= 0;
#pragma unroll
        for (int j = 0; j < NS; ++j) shown = shown && !(j != s && ((ab >> j) & 1) && cellv[j] == cell);
        const int cc = cell >= 0 ? cell : 0, wi = cc >> 5, sh = cc & 31;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
          shown = shown && !(((ab >> (NS + dd)) & 1) && ((l.flat[FLAT(dd, wi, e)] >> sh) & 1));
        if (shown) {
          atomicAnd(&l.flat[FLAT(0, wi, e)], ~(1u << sh));
          atomicAnd(&l.flat[FLAT(1, wi, e)], ~(1u << sh));
        }
        l.sdesc[s * WAVE + e] = make_uint2(shown ? (uint32_t)(cc >> 2) : 0xFFFFFFFFu, 0xFFu << ((cc & 3) * 8));
      }
      __syncthreads();
    }
  }
  if (have_render && !(a.debug & 2)) {


  // ---- phase B: the wavefront streams the observation planes ---------------
  // Occlusion was resolved in phase A, so painting is order-free and every
  // layer is a mask we already hold: nothing here depends on a memory load
  // other than LDS, and every LDS read of an iteration is issued up front.
  constexpr int NBS = SL ? SL - NS - 2 : MAX_L;  // backdrop-only characters
  const int NB = SL ? NBS : k.n_bchars;
  uint32_t sch4[NS], dch4[2];
  const uint32_t env_stride = (uint32_t)(1 + L) * (uint32_t)pitch;
  const int64_t env0 = env0_render;
  if constexpr (PS == 3) {
    // the next unit is drawn (scalar atomic, ~1 us) and its state words start travelling now, in front of this unit's
    // plane stores: they have landed when the loop below is through (vmcnt is in order, 63 at most)
    if (ps_prof) { const uint32_t t = ps_now(); pt_c += t - pt_mark; pt_mark = t; ++pt_units; }
    if (P.ps_dynamic) {
      // (round 5) ... from its own shard's counter while that has units, then from the other shards' in turn: the XCDs do
      // not finish together (phase timers: worker lifetimes 435-558 us around a mean of 493 at 1,048,576 environments --
      // the launch ended 65 us after its average worker), and a worker whose shard is dry takes what a slower one has left
      // instead of going home.  A shard found dry stays dry, so a worker asks every counter at most once too often.
      ps_un = ps_n;
      while (ps_steal < ps_shards) {
        const uint32_t y = ps_x + ps_steal >= ps_shards ? ps_x + ps_steal - ps_shards : ps_x + ps_steal;
        const uint32_t nwk_y = ((gridDim.x - y + ps_shards - 1u) / ps_shards) * ps_wpw;
        const uint32_t cand = y + ps_shards * (nwk_y + ps_ticket(P.ps_ctr + 16u * y));
        if (cand < ps_n) { ps_un = cand; break; }
        if (!P.ps_steal) { ps_steal = ps_shards; break; }
        ++ps_steal;
      }
    } else {
      ps_un = ps_u + ps_nwk;
      ps_step_next = ps_step;
      if (ps_un >= ps_n && ps_step + 1 < a.n_steps) {  // this step's last unit of mine: on to the next step, from my first unit
        ps_un = ps_wid;
        ps_step_next = ps_step + 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (its state words may be the ones this wave has just stored)
      }
    }
    if (ps_un < ps_n) ps_prefetch(ps_un, ps_step_next);
    if (P.ps_lock) {
      // at most ps_lock streaming waves per workgroup: a counting semaphore in LDS (lane 0 alone adds; a wave that
      // finds the count at the limit takes its increment back and tries again a little later)
      const uint32_t la = lds_byte_address(lds_raw + k.lds_ps_ring + 3);
      uint32_t spins = 0;
      for (;;) {
        uint32_t old, one = 1u;
        uint64_t save;
        asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %2, %3\n\ts_mov_b64 exec, %1\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(old), "=&s"(save) : "v"(la), "v"(one) : "memory");
        if (__builtin_amdgcn_readfirstlane((int)old) < P.ps_lock || ++spins >= PS_SPIN_LIMIT) break;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_sub_u32 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(save) : "v"(la), "v"(one) : "memory");
        __builtin_amdgcn_s_sleep(8);
      }
      if (ps_prof) { const uint32_t t = ps_now(); pt_b += t - pt_mark; pt_mark = t; }
    }
  }
  // Uniform per-plane base pointers: every store below is `scalar base +
  // 32-bit lane offset`, and the lane offset is the same for all nine planes.
  auto uniform_ptr = [](uint8_t* p) {  // pin a wave-uniform pointer to an SGPR pair
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<uint8_t*>(((uint64_t)hi << 32) | lo);
  };
  uint8_t* const pb_board = uniform_ptr(out.planes + (size_t)env0 * env_stride);
  uint8_t* pb_s[NS];
  uint8_t* pb_d[2];
  uint8_t* pb_b[NBS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    sch4[s] = (uint32_t)k.sprite_ch[s] * 0x01010101u;
    pb_s[s] = uniform_ptr(pb_board + (uint32_t)(1 + k.lay_sprite[s]) * (uint32_t)pitch);
  }
  dch4[0] = (uint32_t)k.maze_ch * 0x01010101u;
  dch4[1] = (uint32_t)k.cash_ch * 0x01010101u;
  pb_d[0] = uniform_ptr(pb_board + (uint32_t)(1 + k.lay_drape[0]) * (uint32_t)pitch);
  pb_d[1] = uniform_ptr(pb_board + (uint32_t)(1 + k.lay_drape[1]) * (uint32_t)pitch);
#pragma unroll
  for (int i = 0; i < NBS; ++i) pb_b[i] = uniform_ptr(pb_board + (uint32_t)(1 + k.lay_bchar[i]) * (uint32_t)pitch);
  const uint32_t magic_q = k.magic_q;
  const uint32_t e_skew = env_stride - 4u * (uint32_t)QW;  // voff = 4 f + e * e_skew
  const uint32_t* const flat_raw = lds_raw + k.lds_flatraw;
  const uint2* const sdesc_raw = reinterpret_cast<const uint2*>(lds_raw + k.lds_sdescraw);

  // Plane tags for compose(): 0 board, 1..2 curtains, 3..3+NS-1 sprites, then
  // the backdrop-only characters.  pb[] = uniform global base, po[] = uniform
  // byte offset of the plane inside an environment record.
  constexpr int NPL = 3 + NS + NBS;
  uint8_t* pb[NPL];
  uint32_t po[NPL];
  pb[0] = pb_board; po[0] = 0;
#pragma unroll
  for (int dd = 0; dd < 2; ++dd) { pb[1 + dd] = pb_d[dd]; po[1 + dd] = (uint32_t)(1 + k.lay_drape[dd]) * (uint32_t)pitch; }
#pragma unroll
  for (int s = 0; s < NS; ++s) { pb[3 + s] = pb_s[s]; po[3 + s] = (uint32_t)(1 + k.lay_sprite[s]) * (uint32_t)pitch; }
#pragma unroll
  for (int i = 0; i < NBS; ++i) { pb[3 + NS + i] = pb_b[i]; po[3 + NS + i] = (uint32_t)(1 + k.lay_bchar[i]) * (uint32_t)pitch; }

  // One (environment e, board dword q) task: compose the board dword and hand
  // it and the layer dwords to put(plane tag, value).
  auto compose = [&](uint32_t e, uint32_t q, uint32_t eF, auto&& put) {
    uint32_t d = l.backdrop4[q];
    uint32_t md[2], ms[NS], mb[NBS];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
      const uint32_t bits = (l.flat[dd * WAVE * FWP + eF + (q >> 3)] >> ((q & 7) * 4)) & 0xFu;
      const uint32_t m01 = (bits * 0x00204081u) & 0x01010101u;
      uint32_t hi8 = m01 << 8;
      asm("" : "+v"(hi8));  // keep LLVM from folding (x << 8) - x back into a quarter-rate x * 255
      md[dd] = hi8 - m01;   // 0x01 -> 0xFF per byte
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const uint2 sd = l.sdesc[s * WAVE + e];
      ms[s] = sd.x == q ? sd.y : 0u;
    }
#pragma unroll
    for (int i = 0; i < NBS; ++i) mb[i] = (SL || i < NB) ? l.bdmask[i * QW + q] : 0u;
    if constexpr (COOP) {  // (the cooperative shape leaves the two curtains unresolved against each other)
      const bool cash_front = (k.above[NS] >> (NS + 1)) & 1;
      const uint32_t m0 = md[0], m1 = md[1];
      md[0] = cash_front ? m0 & ~m1 : m0;
      md[1] = cash_front ? m1 : m1 & ~m0;
    }
    uint32_t uni = md[0] | md[1];
    d = (d & ~md[0]) | (dch4[0] & md[0]);
    d = (d & ~md[1]) | (dch4[1] & md[1]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      uni |= ms[s];
      d = (d & ~ms[s]) | (sch4[s] & ms[s]);
    }
    put(0, d);
    if constexpr (UNOCC) {  // layers are the raw masks, the backdrop's included
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) {
        const uint32_t bits = (flat_raw[dd * WAVE * FWP + eF + (q >> 3)] >> ((q & 7) * 4)) & 0xFu;
        const uint32_t m01 = (bits * 0x00204081u) & 0x01010101u;
        md[dd] = (m01 << 8) - m01;
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const uint2 sd = sdesc_raw[s * WAVE + e];
        ms[s] = sd.x == q ? sd.y : 0u;
      }
      uni = 0;
    }
    // rendering.py:177-179 layer[c] = (board == c): by construction that is
    // the thing's own mask, or the backdrop's where no thing paints.
    put(1, md[0] & 0x01010101u);
    put(2, md[1] & 0x01010101u);
#pragma unroll
    for (int s = 0; s < NS; ++s) put(3 + s, ms[s] & 0x01010101u);
#pragma unroll
    for (int i = 0; i < NBS; ++i) {
      if (!SL && i >= NB) break;
      put(3 + NS + i, mb[i] & ~uni);
    }
  };

  // (persistent shapes: a unit of fewer than 64 environments ends at cnt_render; its skip flags beyond are unset)
  const bool any_skip = __ballot(l.skip[lane] != 0 && (PS == 0 || lane < cnt_render)) != 0ull;  // same in both waves of a workgroup
  // The multi-wave instances are under SGPR pressure (the register allocator parks plane bases in
  // VGPR lanes and fetches them with v_readlane right in front of a store) and are latency-bound,
  // not store-issue-bound: their stores take the hazard-proof form (pcx_internal.h).
  constexpr bool GUARD_SADDR = COOP;
  {
  // Direct path.  Each wave store covers 256 contiguous bytes of one plane of
  // one or two environment records; all nine planes of a 64-dword span leave
  // together.
  // Drain the logic phase's own loads/stores once, here: the loop's stores are
  // inline asm the compiler cannot count, and without this it protects a
  // register of an older store with a vmcnt(0) *inside* the loop, which would
  // serialise every iteration behind all outstanding plane stores.
  if constexpr (PS == 0) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched
  // (persistent shapes: no wait here -- the write-back of the logic phase drains under the plane stores; the
  // build checks that the compiler has put no vmcnt wait into the loop: tools/sgpr_hazard_scan.py --no-loop-vmcnt)
  // (e, q) = the environment and the board dword this lane composes; both and
  // every address derived from them advance incrementally -- no multiplies or
  // divisions in the loop (v_mul_lo/_hi are quarter rate).
  // Static shape with at least 64 dwords per board: a lane wraps into the next
  // environment at most once per iteration, so the update is four selects.
  constexpr bool INCR = !COOP && SR != 0 && (SR * SC / 4) >= WAVE;
  uint32_t e = 0, q = lane, voff = 4u * lane, eF = 0;
  // epilogue (EPI): float32 planes of the selected layers, 16 bytes per board dword
  const bool layers_on = !(EPI && epi.skip_layers);
  uint8_t* const fbase = uniform_ptr(reinterpret_cast<uint8_t*>(epi.out) + (size_t)env0 * epi.env_stride);
  const uint32_t bpd = EPI ? epi.dword_bytes : 16u;  // epilogue bytes per board dword and plane (16: float32 feature planes)
  const uint32_t f_skew = epi.env_stride - bpd * (uint32_t)QW;  // foff = bpd f + e * f_skew
  uint32_t foff = bpd * lane;

  // CODES: planes in their natural order (plane 1 + k = layer of character k)
  constexpr int NPK = CODES ? 1 + SL : 1;
  uint8_t* pbk[NPK];
  int32_t lslot[NPK];  // epilogue slot of layer k
  if constexpr (CODES) {
#pragma unroll
    for (int kk = 0; kk < NPK; ++kk) pbk[kk] = uniform_ptr(pb_board + (uint32_t)kk * (uint32_t)pitch);
#pragma unroll
    for (int kk = 0; kk < SL; ++kk) {
      int32_t slot = -1;
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) if (k.lay_sprite[s2] == kk) slot = epi.sprite_slot[s2];
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) if (k.lay_drape[dd] == kk) slot = epi.drape_slot[dd];
#pragma unroll
      for (int b2 = 0; b2 < NBS; ++b2) if (k.lay_bchar[b2] == kk) slot = epi.bchar_slot[b2];
      lslot[kk] = EPI ? slot : -1;
    }
  }
  // CODES + INCR: the code dword of the NEXT iteration is requested before this iteration's stores are
  // issued, so its LDS latency runs under them (software pipelining by hand: the loop is not unrolled)
  constexpr bool PREFETCH = CODES && INCR;
  uint32_t code_pf = 0;
  const bool planes_on = !(fc && fc->only);  // fused croppers, windows only: the full-board planes are not written
  const int n_iter = COOP ? (EPW * QW + WAVE - 1) / WAVE : PS != 0 ? (cnt_render * QW + WAVE - 1) / WAVE : QW;  // 64 tasks per iteration
  // epilogue with more than 16 write streams per wave (pcx_stream.h fill_epilogue): the uint8 planes of the whole
  // group first, the float32 planes in a second sweep over the same codes
  constexpr bool TWO_PASS = EPI && CODES && INCR;
  const int n_pass = TWO_PASS && epi.two_pass ? 2 : 1;  // (with skip_layers the first sweep writes the board plane only)
  // The sweeps, compiled once per kind of epilogue (MODE 0: none / float32 feature planes, 1: channels last, 2:
  // ObservationToArray) and picked at run time below: as run-time flags inside one loop body the later kinds cost
  // the first a fifth of its speed (1,048,576 environments, feature planes: 2.88 -> 3.50 ms; profiles/r03_post_kernels.md).
  auto sweeps = [&](auto mode_c) {
  constexpr int MODE = decltype(mode_c)::value;
  // ObservationToArray as the epilogue (pcx_stream.h to_array_emit): the value table, copied to LDS by every wave for itself
  constexpr bool to_array = EPI && MODE == 2;
  uint32_t* const lut_lds = to_array ? lds_raw + epi.lut_lds_off : nullptr;
  if (to_array) stream::to_array_stage(epi, lut_lds, lane);
  // channels-last epilogue (pcx_stream.h hwc_emit): this wave's exchange area, rows of unselected layers stay zero
  constexpr bool hwc = EPI && MODE == 1;
  // (two areas per wave, used in turn: an iteration drops its bytes into one and stores the floats of the iteration
  // before from the other)
  const uint32_t hw_words = hwc ? (uint32_t)epi.depth * WAVE : 0u;
  uint32_t* const hw = hwc ? lds_raw + epi.hwc_lds_off + (uint32_t)(COOP ? wave : 0) * 2u * hw_words : nullptr;
  if (hwc)
    for (uint32_t sl = 0; sl < 2u * (uint32_t)epi.depth; ++sl) hw[sl * WAVE + lane] = 0u;
  const uint32_t hw_limit = (uint32_t)((COOP ? EPW : WAVE) * QW);
  // (a sweep that has nothing to store is left out: array-only consumers, EpilogueArgs::skip_board)
  const bool u8_needed = !EPI || !epi.skip_board || layers_on;
#pragma unroll 1
  for (int pass = n_pass == 2 && !u8_needed ? 1 : 0; pass < n_pass; ++pass) {
  const bool do_u8 = n_pass == 1 || pass == 0, do_f32 = n_pass == 1 || pass == 1;
  if constexpr (TWO_PASS) { e = 0; q = lane; voff = 4u * lane; eF = 0; foff = bpd * lane; }
  if constexpr (PREFETCH) code_pf = codes[eF + (NIB ? q >> 1 : q)];
  int hw_it = -1;
  uint32_t hw_sel = 0;
  auto hw_turn = [&](int it_now) {  // store the previous iteration's floats, then this iteration's area becomes "previous"
    if (hw_it >= 0)
      stream::hwc_emit<true>(hw + (hw_sel ^ 1u) * hw_words, epi, (uint32_t)hw_it * WAVE, lane, any_skip, l.skip, (uint32_t)QW, fbase, hw_limit);
    hw_it = it_now;
    hw_sel ^= 1u;
  };
#pragma unroll 1
  for (int it = !planes_on ? n_iter : COOP ? wave : 0; it < n_iter; it += COOP ? (int)(blockDim.x >> 6) : 1) {
    uint32_t e_now, q_now, voff_now, eF_now, foff_now = 0;
    uint32_t code_cur = 0;
    if constexpr (INCR) {
      e_now = e; q_now = q; voff_now = voff; eF_now = eF;
      q += WAVE; voff += 4u * WAVE;
      const bool wrap = q >= (uint32_t)QW;
      q = wrap ? q - QW : q;
      e = wrap ? e + 1 : e;
      voff = wrap ? voff + e_skew : voff;
      eF = wrap ? eF + (CODES ? CODE_PITCH : FWP) : eF;
      if constexpr (EPI) { foff_now = foff; foff += bpd * WAVE; foff = wrap ? foff + f_skew : foff; }
      if constexpr (PREFETCH) {
        code_cur = code_pf;
        code_pf = codes[it + 1 < n_iter ? eF + (NIB ? q >> 1 : q) : 0u];
      }
    } else {
      const uint32_t f = (uint32_t)it * WAVE + lane;
      e_now = (f * magic_q) >> 20;
      q_now = f - e_now * QW;
      voff_now = 4u * f + e_now * e_skew;
      eF_now = e_now * (CODES ? CODE_PITCH : FWP);
      if constexpr (EPI) foff_now = bpd * f + e_now * f_skew;
    }
    // (past the group's last environment, or an environment this launch leaves alone; with the channels-last
    // epilogue such a lane still takes part in the wave's exchange and only its stores are predicated)
    bool dead = false;
    if constexpr (COOP) dead = (int)e_now >= EPW;
    if constexpr (PS != 0) dead = (int)e_now >= cnt_render;
    if (!dead && any_skip) dead = l.skip[e_now] != 0;
    if (dead && !hwc) continue;
    if constexpr (CODES) {
      // one LDS read, then one v_perm_b32 per plane: the board dword picks each
      // cell's character out of the eight, layer k picks byte k of a one-hot table
      const uint32_t code_raw = PREFETCH ? code_cur : codes[eF_now + (NIB ? q_now >> 1 : q_now)];
      const uint32_t code = NIB ? (code_raw >> ((q_now & 1u) << 2)) & 0x0F0F0F0Fu : code_raw;
      if constexpr (TWO_PASS) {
        // ObservationToArray's own sweep: code -> board dword -> value table -> component planes, none of the
        // per-plane selection below (as part of the general body it cost ~120 scalar and ~50 vector
        // instructions per iteration: profiles/r03_post_kernels.md)
        if (to_array && n_pass == 2 && pass == 1) {
          if (!dead) stream::to_array_emit<true>(epi, lut_lds, __builtin_amdgcn_perm(k.chars_hi, k.chars_lo, code), foff_now, fbase);
          continue;
        }
      }
      auto put_plane = [&](uint8_t* base, uint32_t v, int32_t slot) {
        if constexpr (!EPI) {  // (the plain instance keeps every plane base in SGPRs: the bare store; pcx_internal.h)
          asm volatile("global_store_dword %0, %1, %2" : : "v"(voff_now), "v"(v), "s"(base));
        } else if ((slot == -2 ? !epi.skip_board : layers_on) && do_u8 && !dead) {
          saddr_store_dword<true>(voff_now, v, base);  // (the epilogue instances spill SGPRs: the base is copied inside the asm block)
        }
        if constexpr (EPI) {
          if (slot == -2 && to_array && do_f32 && !dead) stream::to_array_emit<true>(epi, lut_lds, v, foff_now, fbase);
          if (slot >= 0 && do_f32 && hwc) {
            stream::hwc_put(hw + hw_sel * hw_words, (uint32_t)epi.depth, lane, slot, v);
          } else if (slot >= 0 && do_f32) {
            stream::f32x4 f;
            f.x = (float)(v & 0xFFu); f.y = (float)((v >> 8) & 0xFFu); f.z = (float)((v >> 16) & 0xFFu); f.w = (float)(v >> 24);
            const uint32_t fo = foff_now + (uint32_t)slot * epi.plane_bytes;
            saddr_store_dwordx4<true>(fo, f, fbase);
          }
        }
      };
      put_plane(pbk[0], __builtin_amdgcn_perm(k.chars_hi, k.chars_lo, code), -2);
#pragma unroll
      for (int kk = 0; kk < SL; ++kk)
        put_plane(pbk[1 + kk], __builtin_amdgcn_perm(kk >= 4 ? 1u << (8 * (kk & 3)) : 0u, kk < 4 ? 1u << (8 * (kk & 3)) : 0u, code),
                  lslot[kk]);
      if constexpr (EPI) {
        if (hwc && do_f32) hw_turn(it);
      }
      continue;
    }
    // scalar base (pinned above) + 32-bit lane offset: one `global_store_dword
    // voffset, data, sbase` per plane, no per-store address arithmetic
    compose(e_now, q_now, eF_now, [&](int plane, uint32_t v) {
      if ((!EPI || (plane == 0 ? !epi.skip_board : layers_on)) && !dead) {
        if constexpr (SL != 0)  // the static-shape instance keeps all nine bases in SGPRs
          saddr_store_dword<GUARD_SADDR>(voff_now, v, pb[plane]);
        else
          *reinterpret_cast<uint32_t*>(pb[plane] + voff_now) = v;
      }
      if constexpr (EPI) {  // a selected layer also leaves as four float32 (rendering.py:545-661)
        if (plane == 0 && to_array && !dead) stream::to_array_emit<true>(epi, lut_lds, v, foff_now, fbase);
        const int32_t slot = plane == 0 ? -1 : plane < 3 ? epi.drape_slot[plane - 1]
                             : plane < 3 + NS ? epi.sprite_slot[plane - 3 < NS ? plane - 3 : 0] : epi.bchar_slot[plane - 3 - NS];
        if (slot >= 0 && hwc) {
          stream::hwc_put(hw + hw_sel * hw_words, (uint32_t)epi.depth, lane, slot, v & 0x01010101u);
        } else if (slot >= 0) {
          stream::f32x4 f;
          f.x = (float)(v & 0xFFu); f.y = (float)((v >> 8) & 0xFFu); f.z = (float)((v >> 16) & 0xFFu); f.w = (float)(v >> 24);
          const uint32_t fo = foff_now + (uint32_t)slot * epi.plane_bytes;
          saddr_store_dwordx4<GUARD_SADDR>(fo, f, fbase);
        }
      }
    });
    if constexpr (EPI) {
      if (hwc) hw_turn(it);
    }
  }
  if constexpr (EPI) {
    if (hwc && hw_it >= 0) hw_turn(-1);  // the last iteration's floats
  }
  }  // passes
  };  // sweeps
  if constexpr (PS == 3) {
    // (round 5, measured and dropped: composing board-dword PAIRS and storing them with global_store_dwordx2 -- half the
    // iterations and store instructions -- runs at 0.97-1.04 ms per 1,048,576 environments against 0.58-0.61 for this
    // dword loop, 8-byte aligned or not: the CU's write path moves a unit in ~13 us with dwordx2 stores whatever the
    // number of streaming waves, in ~8.3 us with dword stores; profiles/r05_tuning.md)
    const int stores_behind = n_iter * (1 + SL);
    sweeps(IntC<0>{});
    if (P.ps_lock) {  // (the last plane store is issued: the next wave may stream)
      const uint32_t la = lds_byte_address(lds_raw + k.lds_ps_ring + 3);
      uint32_t one = 1u;
      uint64_t save;
      asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_sub_u32 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(save) : "v"(la), "v"(one) : "memory");
    }
    // fewer than 64 plane stores behind the DMA and the ticket (units with environments left alone, ablation
    // runs): wait for them; otherwise the next unit's logic phase starts at once
    ps_need_wait = any_skip || (a.debug & ~16) != 0 || stores_behind < 64 || !planes_on;
    if (ps_need_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ps_u = ps_un;
    ps_step = ps_step_next;
    if (ps_prof) pt_d += ps_now() - pt_mark;
  } else if constexpr (!EPI) {
    sweeps(IntC<0>{});
  } else {  // (uniform: one of the three runs)
    if (epi.hwc) sweeps(IntC<1>{});
    else if (epi.to_array) sweeps(IntC<2>{});
    else sweeps(IntC<0>{});
  }
  if constexpr (FUSABLE) {
    if (fc) {  // the croppers' windows, cut from the same descriptors (pcx_stream.h stream_windows)
      stream::PlaneMap<NS, 2, NBS> pm;
      uint32_t bch4[NBS > 0 ? NBS : 1] = {};
#pragma unroll
      for (int s = 0; s < NS; ++s) { pm.sprite_off[s] = (uint32_t)(1 + k.lay_sprite[s]) * (uint32_t)pitch; pm.sprite_ch4[s] = sch4[s]; }
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) { pm.drape_off[dd] = (uint32_t)(1 + k.lay_drape[dd]) * (uint32_t)pitch; pm.drape_ch4[dd] = dch4[dd]; }
#pragma unroll
      for (int i = 0; i < NBS; ++i) { pm.bchar_off[i] = (uint32_t)(1 + k.lay_bchar[i]) * (uint32_t)pitch; bch4[i] = (uint32_t)k.bchar[i] * 0x01010101u; }
      stream::stream_windows<NS, 2, NBS, SR ? (SR * SC + 3) / 4 : 0, 0, SR, SC>(
          fc, pm, bch4, env0, l.backdrop4, l.flat, l.sdesc, l.skip, FWP, lane, COOP ? wave : 0, lds_raw + k.lds_wcorner, nullptr,
          stream::BoardShape{R, C, QW}, COOP ? (int)(blockDim.x >> 6) : 1, NB);
    }
  }
  }
  }  // render wave
  if constexpr (PS == 0) __syncthreads();  // swap buffers
  }  // rounds
  if constexpr (PS != 0) {
    if (ps_prof && lane == 0) {
      uint32_t* const pp = P.ps_prof + (size_t)ps_wid * 16;
      const uint32_t life = ps_now() - pt_start;
      pp[0] = pt_units; pp[1] = pt_a; pp[2] = pt_b; pp[3] = pt_c; pp[10] = pt_d; pp[5] = life;
    }
    // the last workgroup out rewinds the work counter for the next launch (every ticket of this launch was drawn
    // before its workgroup got here)
    if (P.ps_dynamic && lane == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (atomicAdd(P.ps_ctr + 8 * 16, 1u) == ps_nwk - 1u) {
        for (int x = 0; x <= 8; ++x) atomicExch(P.ps_ctr + 16 * x, 0u);
      }
    }
  }
#undef FLAT
}

#ifdef PCX_SM_SPEC
// A run-time build (hiprtc): the persistent owner-code instance and the cooperative one of the shipped 10x30 / 'abcP' shape
// with the constants of PCX_SM_SPEC compiled in.  ScrollyMazeBackend (jit::load) looks them up by their mangled names, which
// follow from this signature -- the kernel itself stays exactly the template libpcx.so is built from (a wrapper around a
// device function with the body in it was tried: the by-reference arguments cost the other instances registers and scratch).
// PCX_SM_SPEC_SHAPE (a level of ANOTHER board or cast: `NS, R, C, L, IP, IE`, PCX_SM_SPEC_UNOCC true / false): one instance -- a
// workgroup per group of 64 environments, the mask-composing render loop -- with the shape as template arguments AND the
// constants compiled in (LV 9), where libpcx.so only has the shape-generic instances (36-60 k instructions for four to six
// sprites, a quarter of them lane moves, because every stride and index is a run-time value).
#ifdef PCX_SM_SPEC_SHAPE
template __global__ void pcx_scrolly_maze_step<PCX_SM_SPEC_SHAPE, PCX_SM_SPEC_UNOCC, false, false, false, 0, 9>(const Consts, const Ptrs, const StepArgs, const pcx_buffers,
                                                                                                              const stream::EpilogueArgs, const crop::FusedCrops*);
#else
#ifndef PCX_SM_SPEC_NO_PS
template __global__ void pcx_scrolly_maze_step<4, 10, 30, 8, 3, 3, false, false, false, true, 3, 7>(const Consts, const Ptrs, const StepArgs, const pcx_buffers,
                                                                                                  const stream::EpilogueArgs, const crop::FusedCrops*);
#endif
template __global__ void pcx_scrolly_maze_step<4, 10, 30, 8, 3, 3, false, true, false, false, 0, 8>(const Consts, const Ptrs, const StepArgs, const pcx_buffers,
                                                                                                  const stream::EpilogueArgs, const crop::FusedCrops*);
#endif
#endif

}  // namespace sm
}  // namespace pcx
