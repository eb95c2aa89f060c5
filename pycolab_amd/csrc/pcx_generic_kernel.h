// pcx_generic_kernel.h -- the device side of pcx_generic.hip: pcx_generic_step, the table-driven step kernel, and
// everything it calls.  A header of its own because it is compiled twice: into libpcx.so as the interpreter every
// template can run on, and at run time by hiprtc with PCX_GENERIC_SPEC naming the constants of ONE template (see the
// comment at PCX_GENERIC_SPEC below and GenericBackend::specialise in pcx_generic.hip).  Device code only: no host
// header may be reached from here when __HIPCC_RTC__ is defined (pcx_device.h).  gfx950 only.
#pragma once

#include "pcx_device.h"
#include "pcx_crop_window.h"
#include "pcx_stream.h"

namespace pcx {
namespace gen {

constexpr int WAVE = 64;
constexpr int MAX_L = 20;

// per-thing table entry (uint32 words), staged into LDS
enum : int { T_CH = 0, T_KIND, T_IDX, T_PROG, T_LAYER, T_ABOVE, T_FLAGS, T_P0, T_P1, T_IMP0, T_IMP1, T_IMP2, T_IMP3,
             T_ABOVE_S, T_ABOVE_D,  // the things in front, as a sprite-index mask and a drape-index mask
             T_P2, T_P3,            // tabled entities: directive field of the action (shift, mask)
             T_GROUP,               // scrolling group (protocols/scrolling.py), 0..PCX_MAX_SCROLL_GROUPS-1
             T_IMPT,                // walkers: mask of the things whose character is in the impassable set
             T_WORDS };
// plot directives (include/pcx.h pcx_directive), four words each, staged into LDS
enum : int { D_WHO = 0 /* thing | kind << 8 | move_this thing << 16 | in_front_of thing << 24 (0xFF = None) */, D_SEL, D_REWARD,
             D_DISCOUNT, D_WORDS };
constexpr int MAX_ZQ = 8;  // z-order changes one environment may queue in one frame
constexpr uint32_t TF_WALKER = 1, TF_CONFINED = 2, TF_EGO = 4, TF_SCROLLY = 8;
// Scrolly drapes reuse the impassable words: pattern table offset, PR | PC << 16,
// have_margins | margin_rows << 8 | margin_cols << 16, words per pattern row
enum : int { T_PAT = T_IMP0, T_PDIM = T_IMP1, T_MARG = T_IMP2, T_PRW = T_IMP3 };

// state words
enum : int { W_FRAME = 0, W_FLAGS, W_V0, W_V1, W_V2, W_V3, W_RNG, W_SPRITES };
constexpr uint32_t F_OVER = 1u, F_ERR_SHIFT = 1;
// rows of the inbox: state words W_FRAME..W_RNG at their own index, then the protocol / z-order / chapter words, the
// action, and the (NS + 3) / 4 sprite flag words
enum : int { IB_SCROLL = W_SPRITES, IB_Z0, IB_Z1, IB_NEXT, IB_ACTION, IB_SFLAGS, IB_ROWS = IB_SFLAGS };
constexpr int64_t NEVER = INT32_MIN;

struct Consts {
  int32_t game, R, C, cells, pitch, QW, L, NS, ND, NT, n_groups, RW, FW, NW, n_actions, n_bchars;
  int32_t occl;  // Engine(..., occlusion_in_layers)
  int32_t n_dir, zdyn, w_z;  // plot directives; any change_z_order among them; state offset of the z-order words
  int32_t w_next;            // state offset of the_plot.next_chapter (-1: no entity assigns it)
  int32_t reward_float;      // pcx_template::reward_is_float: the reward lane and the directives' rewards are float32 bits
  int32_t l_dir, l_zord, l_zabove, l_zabove_s, l_zabove_d, l_zq, l_ztmp;
  int32_t has_scroll, w_scroll;  // any Scrolly drape / egocentric walker; state offset of the protocol words
  int32_t n_sgroups;             // distinct scrolling groups among the things
  uint32_t group_sprites[PCX_MAX_SCROLL_GROUPS];  // sprite-index mask of each group's members
  uint32_t magic_q, magic_c;  // 32-bit reciprocals of QW and C (exhaustively checked on the host)
  int32_t w_sflags, w_drapes;           // state word offsets
  int32_t ip;                           // thing index of 'P' (-1 if none)
  int32_t ix, ib, tx;                   // drape index of 'X' / 'B', thing index of 'X' (-1 if none)
  int32_t bolt_mask_all, bolt_mask_up;  // marauders: sprite-index masks of 'abcdyz' / 'abcd'
  int32_t box_mask;                     // warehouse: sprite-index mask of the box sprites
  // LDS layout (word offsets); per-lane arrays are [i][lane]
  int32_t l_things, l_z, l_sched, l_backdrop, l_bdmask, l_aux, l_init, l_initd, l_laybc, l_s2t, l_d2t;
  int32_t l_pos, l_flg, l_snap, l_cur, l_snapd, l_flat, l_sdesc, l_skip, l_flatraw, l_sdescraw, l_corner, l_pmask, l_pframe, l_words;
  int32_t l_wcorner;              // fused croppers: [MAX_FUSED_CROPPERS][lane] window corners (pcx_stream.h WCORNER_NONE)
  int32_t l_inbox;                // where the group's scalar state words land by LDS-DMA: [IB_ROWS + (NS + 3) / 4][lane], over what the render phase reads later
  int32_t l_bdcode;               // owner codes (pcx_stream.h stream_codes): the backdrop's code dwords [QW], a staged table (-1: more than 16 characters)
  int32_t l_codes;                // pcx_generic_step rendering from owner codes (round 6): the group's code dwords [64][QW | 1], laid over the per-lane
                                  // arrays that are dead by the time the logic phase writes them (-1: not laid out -- unoccluded layers, more than 16 characters)
  uint8_t chars[PCX_MAX_CHARS];   // character of layer plane 1 + i
};

struct Ptrs {
  const uint32_t* tables;  // everything staged into LDS, in l_* order
  int32_t n_table_words;
  uint32_t* state;         // word w of environment e at state[(e / 64) * state_unit + w * state_row + e % 64]:
  int64_t state_unit, state_row;  // unit-major (round 6: state_unit = NW * 64, state_row = 64 -- a unit's NW rows are ONE contiguous piece) or [NW][bpad] (64, bpad)
  int32_t* track;          // [NS][bpad]
  uint32_t* curtains;      // [ND][FW][bpad] raw curtain bits (export_curtains)
  int64_t batch, bpad;
  unsigned long long* stats;  // PCX_DEBUG & 8: cycles per program id (64 slots) and per section (64..)
  uint64_t seed;       // the engine's seed and the global index of its environment 0 (pcx_template::param[0..3]):
  int64_t env_offset;  // per engine, not per template -- a specialised build of the kernel serves every engine of a template
  // include/pcx.h pcx_engine_set_epilogue, the planar feature array (rendering.py:545-661, default axis order): float32
  // [batch][feat_depth][rows * cols], plane f = (board == feat_ch[f]); null: none.  feat_skip: 1 the uint8 layer planes
  // are not written, 2 nor is the board plane.
  float* feat;
  int32_t feat_depth, feat_skip;
  uint8_t feat_ch[PCX_POST_MAX_DEPTH];
  int32_t use_codes;  // pcx_generic_step: this launch renders from owner codes (Consts::l_codes) instead of masks -- plain steps without fused croppers / epilogue
  // include/pcx.h pcx_engine_set_plot_words: int32 [PCX_PLOT_WORDS][bpad] the plot words (state words W_V0..3) a new episode starts
  // with -- what a Story's `new_plot.update(old_plot)` left there (storytelling.py:449-450); null: the template's
  const int32_t* plot_in;
};

// A build of this file for ONE template (round 4): PCX_GENERIC_SPEC names a header that defines `static constexpr
// Consts K` and `static constexpr uint32_t TAB[]` -- the very values GenericBackend::init computes for the template
// (GenericBackend::spec_header writes them) -- and every table look-up with a uniform index, every loop bound and the
// program switch of the update schedule become compile-time constants: the interpreter is partially evaluated by the
// compiler.  Per-lane look-ups (the thing on top of a cell, the backdrop under it) keep reading the staged LDS copy.
#ifdef PCX_GENERIC_SPEC
namespace spec {
#include PCX_GENERIC_SPEC
}
#define PCX_SPEC_UNROLL _Pragma("unroll")
#else
#define PCX_SPEC_UNROLL
#endif
// Round 6, the specialised build only.  Its logic phase is a chain of LDS round trips (profiles/r06_generic.md), so:
//  * PCX_SREGS: every sprite's position word, flag byte and snapshot cell live in REGISTERS (arrays indexed by compile-time
//    constants once the loops over the sprites are unrolled; an index that is a lane's own goes through a select chain) --
//    the LDS columns l.pos only receive the state rows (LDS-DMA) and l.flg / l.snap are not touched at all;
//  * PCX_PROBE_UNROLL: the probes of one MazeWalker move (target + two flanks; the eight neighbours of an egocentric walker)
//    are unrolled, so that their LDS reads are in flight together instead of one probe after the other (the table-driven
//    build keeps ONE copy of the probe walked by a loop: it is bound by instruction issue, not by latency).
// Measured (profiles/r06_generic.md): registers -5 % on the launch, -12 % on its logic phase with up to eight sprites, a loss with
// the marauders' cast (their programs index sprites by a lane's own values: select chains) -- GenericBackend::plan decides and the
// constants' header says so (PCX_SPEC_SREGS; the LDS layout in the same header has no flag / snapshot columns then); the probes
// side by side LOSE 5-10 % everywhere (three probes' worth of work for a cardinal move), so that one is opt-in
// (PCX_GENERIC_SPEC_DEFS=-DPCX_X_PROBE_UNROLL).
#if defined(PCX_GENERIC_SPEC) && defined(PCX_SPEC_SREGS)
#define PCX_SREGS 1
#endif
// (the egocentric walker's eight neighbour probes ARE unrolled in the specialised build -- all eight are needed, their LDS reads
// overlap: walkers_scroll_groups 0.0768 -> 0.0737 ms, the other scrolling fixtures unchanged; -DPCX_X_NO_EGO_UNROLL: the loop)
#if defined(PCX_GENERIC_SPEC) && !defined(PCX_X_NO_EGO_UNROLL)
#define PCX_PROBE_UNROLL _Pragma("unroll")
#else
#define PCX_PROBE_UNROLL _Pragma("unroll 1")
#endif
#ifdef PCX_GENERIC_SPEC
// N values that must end up in registers: a recursive struct of scalars, read and written through select chains -- no array,
// no loop, nothing but constant member offsets from the first optimisation pass on.  (An array indexed in `#pragma unroll`
// loops was tried first: the loops are unrolled after the pass that splits aggregates has given up on the whole Ctx, which
// then lives in scratch -- 470 bytes of it.)
template <typename T, int N>
struct RegList {
  T head;
  RegList<T, N - 1> tail;
};
template <typename T>
struct RegList<T, 1> {
  T head;
};
template <typename T, int N>
__device__ __forceinline__ T reg_get(const RegList<T, N>& r, int s) {
  if constexpr (N == 1) return r.head;
  else return s == 0 ? r.head : reg_get(r.tail, s - 1);
}
template <typename T, int N>
__device__ __forceinline__ void reg_put(RegList<T, N>& r, int s, T v) {
  r.head = s == 0 ? v : r.head;
  if constexpr (N > 1) reg_put(r.tail, s - 1, v);
}
#endif
#ifdef PCX_GENERIC_SPEC
constexpr int SREG_N = spec::K.NS > 0 ? spec::K.NS : 1;
#endif

struct L {
  const uint32_t *things, *z, *sched, *backdrop4, *bdmask, *aux, *init, *initd, *laybc, *s2t, *d2t;
  uint32_t *pos, *flg, *cur, *snapd, *flat, *skip, *flatraw, *corner, *pmask, *pframe;
  const uint32_t* dir;
  uint32_t *zord, *zabove, *zabove_s, *zabove_d, *zq, *ztmp;  // per-lane z-order (only when a game changes it)
  int32_t* snap;
  uint2 *sdesc, *sdescraw;
  uint32_t* wcorner;  // fused croppers: window corners [MAX_FUSED_CROPPERS][lane]
  // owner codes (pcx_generic_step_pw): the staged backdrop code dwords, and where the logic phase leaves this group's code
  // dwords [64][QW | 1] for the render worker (null: it leaves flat / sdesc for the mask-composing loop)
  const uint32_t* bdcode;
  uint32_t* codes;
};
// the code byte of layer i of L characters (pcx_stream.h: one v_perm_b32 selects among eight)
__device__ __forceinline__ uint32_t code_of_layer(int L, uint32_t i) { return L <= 8 ? i : i < 8u ? 0xC0u | i : 0x0Cu | ((i - 8u) << 4); }

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

struct Ctx {
  const Consts& k;
  const L& l;
  int lane;
  int frame, action;
  uint32_t err;
  int reward_set, reward, game_over;
  float discount;
  int32_t v[4];  // program variables (state words W_V0..3)
  // protocols/scrolling.py, scrolling group '': the order lives one frame only
  // protocols/scrolling.py: the order of the running entity's scrolling group
  // (it lives one frame only); with several groups the others' orders wait in
  // `orders`, eight bits each (valid, o0 + 1, o1 + 1)
  int order_valid, o0, o1;
  uint32_t orders, gsprites;  // gsprites: sprite-index mask of that group's members
  uint32_t registered;  // bit per sprite index: 'scrolling_X_egocentrists' of every group X
  int nzq;              // queued change_z_order directives (plot.py:173-174)
  // A program's MazeWalker._move is not called where the program stands: it is left here and made
  // by the ONE mw_move the kernel contains, right after the program switch (the probes behind a move
  // are the bulk of the code; a copy per program tripled the kernel and its register pressure).
  int mv_dr, mv_dc, mv_post;  // mv_post: 0 no move, 1 move, 2 move + BS patroller's catch check
  int next;                   // the_plot.next_chapter as the episode's entities left it (PCX_CHAPTER_UNSET: untouched)
  uint64_t rng_seed;          // Ptrs::seed
#ifdef PCX_SREGS
  mutable RegList<uint32_t, SREG_N> rpos, rflg;  // sprite s: packed virtual position; flag byte (visible, prior, program bits)
  mutable RegList<int32_t, SREG_N> rsnap;        // the cell the last repaint painted it at (-1: not painted)
#endif
};
// sprite state: per-lane LDS columns, or (PCX_SREGS) registers
__device__ __forceinline__ uint32_t spos(const Ctx& x, int s) {
#ifdef PCX_SREGS
  return reg_get(x.rpos, s);
#else
  return x.l.pos[s * WAVE + x.lane];
#endif
}
__device__ __forceinline__ void spos_set(const Ctx& x, int s, uint32_t v) {
#ifdef PCX_SREGS
  reg_put(x.rpos, s, v);
#else
  x.l.pos[s * WAVE + x.lane] = v;
#endif
}
__device__ __forceinline__ uint32_t sflg(const Ctx& x, int s) {
#ifdef PCX_SREGS
  return reg_get(x.rflg, s);
#else
  return x.l.flg[s * WAVE + x.lane];
#endif
}
__device__ __forceinline__ void sflg_set(const Ctx& x, int s, uint32_t v) {
#ifdef PCX_SREGS
  reg_put(x.rflg, s, v);
#else
  x.l.flg[s * WAVE + x.lane] = v;
#endif
}
__device__ __forceinline__ int ssnap(const Ctx& x, int s) {
#ifdef PCX_SREGS
  return reg_get(x.rsnap, s);
#else
  return x.l.snap[s * WAVE + x.lane];
#endif
}
__device__ __forceinline__ void ssnap_set(const Ctx& x, int s, int v) {
#ifdef PCX_SREGS
  reg_put(x.rsnap, s, (int32_t)v);
#else
  x.l.snap[s * WAVE + x.lane] = v;
#endif
}

__device__ __forceinline__ bool on_board(const Consts& k, int r, int c) {
  return (unsigned)r < (unsigned)k.R && (unsigned)c < (unsigned)k.C;
}
// (every lane reads the same word: pinning the value to an SGPR keeps the interpreter's control flow
// scalar and its table values out of the vector registers)
__device__ __forceinline__ uint32_t tfield(const Ctx& x, int thing, int f) {
#ifdef PCX_GENERIC_SPEC
  return spec::TAB[spec::K.l_things + thing * T_WORDS + f];
#else
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)x.l.things[thing * T_WORDS + f]);
#endif
}
// ... and where the thing (or the field) is the lane's own -- the thing on top of a cell, a place in an
// environment's own z-order, the impassable word of the lane's backdrop character
__device__ __forceinline__ uint32_t tfield_v(const Ctx& x, int thing, int f) { return x.l.things[thing * T_WORDS + f]; }
// The z-order (engine.py:751-757 paints `_sprites_and_drapes` in order).  Things
// are numbered by their place in the template's order; a game whose entities
// issue change_z_order keeps every environment's current order, and the
// "who is in front of whom" masks derived from it, in per-lane LDS columns.
__device__ __forceinline__ int thing_at_z(const Ctx& x, int z) { return x.k.zdyn ? (int)x.l.zord[z * WAVE + x.lane] : (int)x.l.z[z]; }
__device__ __forceinline__ uint32_t above_things(const Ctx& x, int t) { return x.k.zdyn ? x.l.zabove[t * WAVE + x.lane] : tfield(x, t, T_ABOVE); }
__device__ __forceinline__ uint32_t above_sprites(const Ctx& x, int t) { return x.k.zdyn ? x.l.zabove_s[t * WAVE + x.lane] : tfield(x, t, T_ABOVE_S); }
__device__ __forceinline__ uint32_t above_drapes(const Ctx& x, int t) { return x.k.zdyn ? x.l.zabove_d[t * WAVE + x.lane] : tfield(x, t, T_ABOVE_D); }
__device__ __forceinline__ void derive_above(const Ctx& x) {  // masks from the order, front to back
  uint32_t all = 0, sp = 0, dr = 0;
  for (int z = x.k.NT - 1; z >= 0; --z) {
    const int t = (int)x.l.zord[z * WAVE + x.lane];
    x.l.zabove[t * WAVE + x.lane] = all;
    x.l.zabove_s[t * WAVE + x.lane] = sp;
    x.l.zabove_d[t * WAVE + x.lane] = dr;
    all |= 1u << t;
    if (tfield_v(x, t, T_KIND) == 0) sp |= 1u << tfield_v(x, t, T_IDX); else dr |= 1u << tfield_v(x, t, T_IDX);
  }
}

// ---- sprite state in per-lane LDS columns ---------------------------------
__device__ __forceinline__ void sprite_get(const Ctx& x, int s, int& vr, int& vc, int& vis, int& prior) {
  uint32_t p = spos(x, s), f = sflg(x, s);
  vr = pos_r(p); vc = pos_c(p); vis = f & 1; prior = (f >> 1) & 1;
}
__device__ __forceinline__ void sprite_put(const Ctx& x, int s, int vr, int vc, int vis, int prior) {
  spos_set(x, s, pack_pos(vr, vc));
  sflg_set(x, s, (sflg(x, s) & ~3u) | (uint32_t)vis | ((uint32_t)prior << 1));
}
// Sprite.position (true position): virtual if on board, else (0, 0) for walkers
__device__ __forceinline__ void sprite_true(const Ctx& x, int s, int& r, int& c) {
  uint32_t p = spos(x, s);
  r = pos_r(p); c = pos_c(p);
  if (!on_board(x.k, r, c)) { r = 0; c = 0; }
}
__device__ __forceinline__ int sprite_cell(const Ctx& x, int s) {  // engine.py:752-753
  int vr, vc, vis, prior;
  sprite_get(x, s, vr, vc, vis, prior);
  if (!vis) return -1;
  return on_board(x.k, vr, vc) ? vr * x.k.C + vc : 0;
}
// sprites.py:315-352 _teleport
__device__ __forceinline__ void teleport(const Ctx& x, int s, int nr, int nc) {
  int vr, vc, vis, prior;
  sprite_get(x, s, vr, vc, vis, prior);
  bool old_on = on_board(x.k, vr, vc), new_on = on_board(x.k, nr, nc);
  if (old_on && !new_on) { prior = vis; vis = 0; }
  if (!old_on && new_on) vis = prior;
  sprite_put(x, s, nr, nc, vis, prior);
}

// ---- curtains: bit-rows in per-lane LDS columns ------------------------------
__device__ __forceinline__ uint32_t* drape_rows(const Ctx& x, uint32_t* base, int d) {
  return base + (size_t)d * x.k.R * x.k.RW * WAVE + x.lane;  // word i at [i * WAVE]
}
__device__ __forceinline__ bool bit_at(const Ctx& x, uint32_t* base, int d, int r, int c) {
  return (drape_rows(x, base, d)[(r * x.k.RW + (c >> 5)) * WAVE] >> (c & 31)) & 1;
}
__device__ __forceinline__ uint64_t row_get(const Ctx& x, uint32_t* base, int d, int r) {
  const uint32_t* p = drape_rows(x, base, d) + (size_t)(r * x.k.RW) * WAVE;
  uint64_t v = p[0];
  if (x.k.RW > 1) v |= (uint64_t)p[WAVE] << 32;
  return v;
}
__device__ __forceinline__ void row_put(const Ctx& x, uint32_t* base, int d, int r, uint64_t v) {
  uint32_t* p = drape_rows(x, base, d) + (size_t)(r * x.k.RW) * WAVE;
  p[0] = (uint32_t)v;
  if (x.k.RW > 1) p[WAVE] = (uint32_t)(v >> 32);
}

// engine.py:735 repaint == remember what every probe of the next group sees
__device__ __forceinline__ void snapshot(const Ctx& x) {
  PCX_SPEC_UNROLL
  for (int s = 0; s < x.k.NS; ++s) ssnap_set(x, s, sprite_cell(x, s));
  const int n = x.k.ND * x.k.R * x.k.RW;
  for (int i0 = 0; i0 < n; i0 += 4) {  // four LDS reads in flight
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = x.l.cur[(i0 + j < n ? i0 + j : i0) * WAVE + x.lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) if (i0 + j < n) x.l.snapd[(i0 + j) * WAVE + x.lane] = v[j];
  }
}
// Which things the last repaint painted at board cell (r, c), as a mask over
// thing ids (= places in the template's z-order): every sprite's snapshot cell
// and every curtain's snapshot bit -- one per-lane LDS read per thing and no
// read that depends on another, instead of a walk over the z-order with three
// table look-ups and a dependent read per thing.
__device__ __forceinline__ uint32_t present_at(const Ctx& x, int r, int c) {
  const int cell = r * x.k.C + c, NS = x.k.NS, ND = x.k.ND;
  uint32_t m = 0;
#ifdef PCX_SREGS
#pragma unroll
#else
#pragma unroll 2
#endif
  for (int s = 0; s < NS; ++s) {
    const int at = ssnap(x, s);
    const uint32_t bit = 1u << __builtin_amdgcn_readfirstlane((int)x.l.s2t[s]);
    m |= at == cell ? bit : 0u;
  }
  const int wo = r * x.k.RW + (c >> 5), sh = c & 31, dstride = x.k.R * x.k.RW;
  for (int d = 0; d < ND; ++d) {
    const uint32_t w = x.l.snapd[(d * dstride + wo) * WAVE + x.lane];
    const uint32_t bit = 1u << __builtin_amdgcn_readfirstlane((int)x.l.d2t[d]);
    m |= ((w >> sh) & 1u) ? bit : 0u;
  }
  return m;
}
// The thing in front among those of a non-empty mask (engine.py:751-757 paints
// back to front): the highest id in the template's order, or -- where entities
// change the z-order -- the one none of the others is in front of.
__device__ __forceinline__ int top_thing(const Ctx& x, uint32_t m) {
  if (!x.k.zdyn) return 31 - __clz((int)m);
  int top = 31 - __clz((int)m);
  for (uint32_t rest = m; rest; rest &= rest - 1u) {
    const int t = __ffs((int)rest) - 1;
    if (!(m & x.l.zabove[t * WAVE + x.lane])) top = t;
  }
  return top;
}
// character on top of board cell (r, c) in the last repaint (rendering.py:85-184)
__device__ __forceinline__ int top_char(const Ctx& x, int r, int c) {
  const int cell = r * x.k.C + c;
  const int back = (x.l.backdrop4[cell >> 2] >> ((cell & 3) * 8)) & 0xFF;
  const uint32_t m = present_at(x, r, c);
  return m ? (int)tfield_v(x, top_thing(x, m), T_CH) : back;
}
// layers[thing's char][r, c] in the last repaint: with occlusion the thing must
// be the one on top (rendering.py:177-179); without, its own mask counts
// (rendering.py:236-278).
__device__ __forceinline__ bool thing_layer(const Ctx& x, int thing, int r, int c) {
  const uint32_t m = present_at(x, r, c);
  const bool raw = (m >> thing) & 1u;
  if (!x.k.occl || !raw) return raw;
  return top_thing(x, m) == thing;
}
// Row r of a drape's layer in the last repaint, 64 columns at once: its
// snapshot row, and with occlusion minus every cell a thing in front of it
// covers (the same rule as thing_layer, without a per-cell z-order walk).
__device__ __forceinline__ uint64_t drape_layer_row(const Ctx& x, int thing, int r) {
  uint64_t bits = row_get(x, x.l.snapd, tfield(x, thing, T_IDX), r);
  if (!x.k.occl || !bits) return bits;
  const uint32_t above = above_things(x, thing);
  const int lo = r * x.k.C, hi = lo + x.k.C;
  for (int u = 0; u < x.k.NT; ++u) {
    if (!((above >> u) & 1)) continue;
    const uint32_t idx = tfield(x, u, T_IDX);
    if (tfield(x, u, T_KIND) == 1) {
      bits &= ~row_get(x, x.l.snapd, idx, r);
    } else {
      const int cell = ssnap(x, (int)idx);
      if (cell >= lo && cell < hi) bits &= ~(1ull << (cell - lo));
    }
  }
  return bits;
}
// numpy `layers[c][r, col]` with Python index rules (negative wraps once)
__device__ __forceinline__ bool layer_at(Ctx& x, int thing, int r, int c) {
  if (r < 0) r += x.k.R;
  if (c < 0) c += x.k.C;
  if (!on_board(x.k, r, c)) { x.err |= ERR_INDEX; return false; }
  return thing_layer(x, thing, r, c);
}

// sprites.py:496-511 at()/is_impassable(), :479-546 _check_motion, :356-389 _move
// (no scrolling group exists in these games, so the protocol hooks are no-ops)
__device__ __forceinline__ bool blocked_at(Ctx& x, int thing, int vr, int vc, int dr, int dc) {
  const int r = vr + dr, c = vc + dc;
  if (!on_board(x.k, r, c)) return (tfield(x, thing, T_FLAGS) & TF_CONFINED) != 0;  // EDGE
  const int cell = r * x.k.C + c;
  const int back = (x.l.backdrop4[cell >> 2] >> ((cell & 3) * 8)) & 0xFF;
  const uint32_t impt = tfield(x, thing, T_IMPT);
#ifdef PCX_GENERIC_SPEC
  // (the walker's four impassable words are constants here: the word of the lane's backdrop character is a select among them, not
  // a second LDS read that depends on the first -- a probe is ONE LDS round trip)
  const uint32_t i0 = tfield(x, thing, T_IMP0), i1 = tfield(x, thing, T_IMP1), i2 = tfield(x, thing, T_IMP2), i3 = tfield(x, thing, T_IMP3);
  const int iw = (back >> 5) & 3;
  const uint32_t imp_back = iw == 0 ? i0 : iw == 1 ? i1 : iw == 2 ? i2 : i3;
#else
  const uint32_t imp_back = tfield_v(x, thing, T_IMP0 + ((back >> 5) & 3));
#endif
  const uint32_t m = present_at(x, r, c);
  if (m) return (impt >> top_thing(x, m)) & 1u;  // the thing in front decides
  if (back >= 128) return false;  // impassable sets are ASCII (compiler.py); anything else is passable
  return (imp_back >> (back & 31)) & 1u;
}
// One copy of the probe in the code, walked by a loop: the target cell, then (diagonals only) the
// two flanks (sprites.py:539-543: blocked by its own cell, or by both flanks).
__device__ __forceinline__ bool check_motion(Ctx& x, int thing, int vr, int vc, int dr, int dc) {
  if (dr == 0 && dc == 0) return false;
#if defined(PCX_GENERIC_SPEC) && defined(PCX_X_PROBE_UNROLL)
  // (opt-in, measured slower) the target cell and both flanks side by side, their LDS reads in flight together --
  // a probe writes nothing, and the flanks only count for a diagonal
  const bool b0 = blocked_at(x, thing, vr, vc, dr, dc), b1 = blocked_at(x, thing, vr, vc, dr, 0), b2 = blocked_at(x, thing, vr, vc, 0, dc);
  return b0 || (dr != 0 && dc != 0 && b1 && b2);
#else
  const int n = (dr != 0 && dc != 0) ? 3 : 1;
  bool hit[3] = {false, false, false};
#pragma unroll 1
  for (int i = 0; i < n; ++i) {
    const bool b = blocked_at(x, thing, vr, vc, i == 2 ? 0 : dr, i == 1 ? 0 : dc);
    hit[0] = i == 0 ? b : hit[0]; hit[1] = i == 1 ? b : hit[1]; hit[2] = i == 2 ? b : hit[2];
  }
  return hit[0] || (hit[1] && hit[2]);
#endif
}
__device__ __forceinline__ int motion_bit(int dr, int dc) { return (dr + 1) * 3 + (dc + 1); }
__device__ __forceinline__ void request_move(Ctx& x, int dr, int dc, int post = 1) { x.mv_dr = dr; x.mv_dc = dc; x.mv_post = post; }
__device__ __forceinline__ bool mw_move(Ctx& x, int thing, int dr, int dc) {
  const int s = tfield(x, thing, T_IDX);
  const bool ego = x.k.has_scroll && (tfield(x, thing, T_FLAGS) & TF_EGO);
  int vr, vc, vis, prior;
  if (x.k.has_scroll) {  // sprites.py:413-454 _obey_scrolling_order
    if (ego) x.registered |= 1u << s;
    if (x.order_valid) {
      sprite_get(x, s, vr, vc, vis, prior);
      teleport(x, s, vr - x.o0, vc - x.o1);
      if (ego && x.o0 != dr && x.o1 != dc) x.err |= ERR_SCROLL;
    }
  }
  sprite_get(x, s, vr, vc, vis, prior);
  const bool blocked = check_motion(x, thing, vr, vc, dr, dc);
  if (!blocked) { teleport(x, s, vr + dr, vc + dc); vr += dr; vc += dc; }
  if (ego) {  // sprites.py:456-477 + scrolling.py:372-434 permit()
    // the eight neighbours, each probed once (sprites.py:539-543: a diagonal is blocked by its
    // own cell or by both of its flanks)
    uint32_t nb = 0;  // bit motion_bit(a, b): the neighbour at (a, b) is impassable
    PCX_PROBE_UNROLL
    for (int i = 0; i < 9; ++i) {
      const int a = i / 3 - 1, b = i - 3 * (i / 3) - 1;
      if (i != 4 && blocked_at(x, thing, vr, vc, a, b)) nb |= 1u << i;
    }
    const bool n = (nb >> motion_bit(-1, 0)) & 1, so = (nb >> motion_bit(1, 0)) & 1;
    const bool we = (nb >> motion_bit(0, -1)) & 1, ea = (nb >> motion_bit(0, 1)) & 1;
    const bool nw = (nb >> motion_bit(-1, -1)) & 1, ne = (nb >> motion_bit(-1, 1)) & 1;
    const bool sw = (nb >> motion_bit(1, -1)) & 1, se = (nb >> motion_bit(1, 1)) & 1;
    uint32_t legal = 1u << motion_bit(0, 0);
    legal |= (uint32_t)!n << motion_bit(-1, 0) | (uint32_t)!so << motion_bit(1, 0);
    legal |= (uint32_t)!we << motion_bit(0, -1) | (uint32_t)!ea << motion_bit(0, 1);
    legal |= (uint32_t)!(nw || (n && we)) << motion_bit(-1, -1) | (uint32_t)!(ne || (n && ea)) << motion_bit(-1, 1);
    legal |= (uint32_t)!(sw || (so && we)) << motion_bit(1, -1) | (uint32_t)!(se || (so && ea)) << motion_bit(1, 1);
    const uint32_t my_frame = (uint32_t)(x.frame + 1);
    uint32_t mask = x.l.pmask[s * WAVE + x.lane];
    if (!(mask & 0x80000000u) || x.l.pframe[s * WAVE + x.lane] != my_frame) mask = 0;
    x.l.pmask[s * WAVE + x.lane] = mask | legal | 0x80000000u;  // bit 31: a permit frame exists
    x.l.pframe[s * WAVE + x.lane] = my_frame;
  }
  return blocked;
}

// ---- prefab_parts/drapes.py: Scrolly ----------------------------------------------
// scrolling.py:437-485 is_possible
__device__ __forceinline__ bool is_possible(const Ctx& x, int dr, int dc) {
  for (int s = 0; s < x.k.NS; ++s) {
    if (!(((x.registered & x.gsprites) >> s) & 1)) continue;
    const uint32_t mask = x.l.pmask[s * WAVE + x.lane];
    if (!(mask & 0x80000000u) || x.l.pframe[s * WAVE + x.lane] != (uint32_t)x.frame) return false;
    if (!((mask >> motion_bit(dr, dc)) & 1)) return false;
  }
  return true;
}
// drapes.py:689-695 _update_curtain: cur rows <- pattern window at the corner
__device__ __forceinline__ void update_curtain(Ctx& x, int thing) {
  const int d = tfield(x, thing, T_IDX), C = x.k.C;
  const uint32_t* pat = x.l.things - x.k.l_things + tfield(x, thing, T_PAT);  // tables base + offset
  const int PR = tfield(x, thing, T_PDIM) & 0xFFFF, PC = tfield(x, thing, T_PDIM) >> 16, PRW = tfield(x, thing, T_PRW);
  const uint32_t cw = x.l.corner[d * WAVE + x.lane];
  const int cr = pos_r(cw), cc = pos_c(cw);
  if (cr < 0 || cc < 0 || cr + x.k.R > PR || cc + C > PC) { x.err |= ERR_INDEX; return; }
  const uint64_t m = C >= 64 ? ~0ull : ((1ull << C) - 1ull);
#ifndef PCX_X_CURTAIN_ROW_BY_ROW
  // four rows' pattern words in flight (twelve LDS reads), then their four row stores: row by row every row waited for its own
  // reads -- the tables and the per-lane arrays are one LDS allocation, the compiler keeps reads and writes in program order
  const int wi = cc >> 5, sh = cc & 31, R = x.k.R;
  for (int r0 = 0; r0 < R; r0 += 4) {
    uint32_t w0[4], w1[4], w2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t* row = pat + (cr + (r0 + j < R ? r0 + j : R - 1)) * PRW;
      w0[j] = row[wi]; w1[j] = row[wi + 1]; w2[j] = row[sh ? wi + 2 : wi];  // (the third word only counts where the window is not word-aligned)
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (r0 + j >= R) break;
      uint64_t bits = ((uint64_t)w0[j] | ((uint64_t)w1[j] << 32)) >> sh;
      if (sh) bits |= (uint64_t)w2[j] << (64 - sh);
      row_put(x, x.l.cur, d, r0 + j, bits & m);
    }
  }
#else
  for (int r = 0; r < x.k.R; ++r) {
    const uint32_t* row = pat + (cr + r) * PRW;
    const int wi = cc >> 5, sh = cc & 31;
    const uint64_t lo = (uint64_t)row[wi] | ((uint64_t)row[wi + 1] << 32);
    uint64_t bits = lo >> sh;
    if (sh) bits |= (uint64_t)row[wi + 2] << (64 - sh);
    row_put(x, x.l.cur, d, r, bits & m);
  }
#endif
}
// drapes.py:487-659 _maybe_move
__device__ __forceinline__ void maybe_move(Ctx& x, int thing, int dr, int dc) {
  const int d = tfield(x, thing, T_IDX);
  const int PR = tfield(x, thing, T_PDIM) & 0xFFFF, PC = tfield(x, thing, T_PDIM) >> 16;
  const int lim_r = PR - x.k.R, lim_c = PC - x.k.C;
  const uint32_t marg = tfield(x, thing, T_MARG);
  uint32_t cw = x.l.corner[d * WAVE + x.lane];
  int cr = pos_r(cw), cc = pos_c(cw);
  if (x.order_valid) {  // :523-535
    if (dr != x.o0 && dc != x.o1) { x.err |= ERR_SCROLL; return; }
    x.l.corner[d * WAVE + x.lane] = pack_pos(cr + x.o0, cc + x.o1);
    update_curtain(x, thing);
    return;
  }
  if (dr == 0 && dc == 0) { update_curtain(x, thing); return; }  // :539-541
  int o0, o1;
  bool go;
  if (!(marg & 1)) {  // scroll_margins=None :551-585
    go = is_possible(x, dr, dc);
    const int north = cr + dr, west = cc + dc;
    o0 = (0 <= north && north <= lim_r) ? dr : 0;
    o1 = (0 <= west && west <= lim_c) ? dc : 0;
  } else {  // :592-659
    const int mrows = (marg >> 8) & 0xFF, mcols = (marg >> 16) & 0xFF;
    const int margin_n = mrows - 1, margin_s = x.k.R - mrows, margin_w = mcols - 1, margin_e = x.k.C - mcols;
    bool vert = false, horiz = false;
    for (int s = 0; s < x.k.NS; ++s) {  // registered egocentric *sprites* (:611)
      if (!(((x.registered & x.gsprites) >> s) & 1)) continue;
      int old_r, old_c;
      sprite_true(x, s, old_r, old_c);
      const int new_r = old_r + dr, new_c = old_c + dc;
      vert |= (old_r > new_r && new_r <= margin_n) || (old_r < new_r && new_r >= margin_s);
      horiz |= (old_c > new_c && new_c <= margin_w) || (old_c < new_c && new_c >= margin_e);
    }
    if (!(vert || horiz)) { update_curtain(x, thing); return; }
    o0 = vert ? dr : 0;
    o1 = horiz ? dc : 0;
    const int pr = cr + o0, pc = cc + o1;
    go = 0 <= pr && pr <= lim_r && 0 <= pc && pc <= lim_c && is_possible(x, dr, dc);
  }
  if (go) {
    x.l.corner[d * WAVE + x.lane] = pack_pos(cr + o0, cc + o1);
    x.order_valid = 1; x.o0 = o0; x.o1 = o1;  // scrolling.py:530-531
  }
  update_curtain(x, thing);
}

// Prefab-only entities (tests/test_things.py:203-295 restated with integer
// actions): per-entity action = (action >> P0) & P1 when P1 != 0, else the
// action; 0..7 = N NE E SE S SW W NW; anything else, None included, is
// `_stay` (which still takes part in the scrolling protocol).
__device__ __forceinline__ int entity_action(const Ctx& x, int thing) {
  if (x.action < 0) return 8;
  const uint32_t sh = tfield(x, thing, T_P0), mk = tfield(x, thing, T_P1);
  const uint32_t a = mk ? ((uint32_t)x.action >> sh) & mk : (uint32_t)x.action;
  return a > 8u ? 8 : (int)a;
}
__device__ __forceinline__ void motion9(int a, int& dr, int& dc) {
  dr = (a == 0 || a == 1 || a == 7) ? -1 : (a == 3 || a == 4 || a == 5) ? 1 : 0;
  dc = (a == 1 || a == 2 || a == 3) ? 1 : (a == 5 || a == 6 || a == 7) ? -1 : 0;
}
__device__ __forceinline__ void prog_walker(Ctx& x, int thing) {
  int dr, dc;
  motion9(entity_action(x, thing), dr, dc);
  request_move(x, dr, dc);
}
__device__ __forceinline__ void prog_scrolly(Ctx& x, int thing) {
  int dr, dc;
  motion9(entity_action(x, thing), dr, dc);
  maybe_move(x, thing, dr, dc);
}

__device__ __forceinline__ void terminate(Ctx& x, float discount = 0.0f) { x.game_over = 1; x.discount = discount; }  // plot.py:176-198
// plot.py:200-226.  x.reward holds the sum in the template's reward type: an int32, or the bits of a float32 (reward_float)
__device__ __forceinline__ void add_reward_bits(Ctx& x, uint32_t bits) {  // ... a pcx_directive's reward word
  x.reward_set = 1;
  if (x.k.reward_float) x.reward = __float_as_int(__int_as_float(x.reward) + __uint_as_float(bits));
  else x.reward += (int)bits;
}
__device__ __forceinline__ void add_reward(Ctx& x, int r) { add_reward_bits(x, x.k.reward_float ? __float_as_uint((float)r) : (uint32_t)r); }

// Plot directives of a tabled entity (include/pcx.h pcx_directive): issued
// before it moves, in table order, when its directive field of the action
// selects them (tests/engine_test.py:169-295 injects the same calls).
__device__ __forceinline__ void issue_directives(Ctx& x, int thing) {
  const uint32_t mask = tfield(x, thing, T_P3);
  if (x.action < 0 || !mask) return;
  const uint32_t sel = ((uint32_t)x.action >> tfield(x, thing, T_P2)) & mask;
  if (!sel) return;
  for (int i = 0; i < x.k.n_dir; ++i) {
    const uint32_t* d = x.l.dir + i * D_WORDS;
    if ((int)(d[D_WHO] & 0xFF) != thing || d[D_SEL] != sel) continue;
    switch ((d[D_WHO] >> 8) & 0xFF) {
      case PCX_DIR_ADD_REWARD: add_reward_bits(x, d[D_REWARD]); break;
      case PCX_DIR_TERMINATE: terminate(x, __uint_as_float(d[D_DISCOUNT])); break;
      case PCX_DIR_NEXT_CHAPTER: x.next = (int)d[D_REWARD]; break;  // plot.py:299-324: the last assignment stands
      case PCX_DIR_Z_ORDER:
        if (x.nzq < MAX_ZQ) x.l.zq[x.nzq++ * WAVE + x.lane] = d[D_WHO] >> 16;  // move_this | in_front_of << 8
        else x.err |= ERR_INDEX;
        break;
      default: break;
    }
  }
}
// engine.py:796-835: one directive at a time, the moving thing is taken out and
// put back behind everything (in_front_of None) or right in front of another
__device__ __forceinline__ void apply_z_updates(Ctx& x) {
  for (int u = 0; u < x.nzq; ++u) {
    const uint32_t w = x.l.zq[u * WAVE + x.lane];
    const int move = (int)(w & 0xFF), front = (int)((w >> 8) & 0xFF);
    int n = 0;
    if (front == 0xFF) x.l.ztmp[n++ * WAVE + x.lane] = (uint32_t)move;
    for (int z = 0; z < x.k.NT; ++z) {
      const int id = (int)x.l.zord[z * WAVE + x.lane];
      if (id == move) continue;
      x.l.ztmp[n++ * WAVE + x.lane] = (uint32_t)id;
      if (id == front) x.l.ztmp[n++ * WAVE + x.lane] = (uint32_t)move;
    }
    for (int z = 0; z < x.k.NT; ++z) x.l.zord[z * WAVE + x.lane] = x.l.ztmp[z * WAVE + x.lane];
  }
  if (x.nzq) derive_above(x);
}

// layers[ch][r, c] for any character, backdrop ones included
__device__ __forceinline__ bool char_layer_at(Ctx& x, int ch, int r, int c) {
  if (r < 0) r += x.k.R;
  if (c < 0) c += x.k.C;
  if (!on_board(x.k, r, c)) { x.err |= ERR_INDEX; return false; }
  for (int t = 0; t < x.k.NT; ++t)
    if ((int)tfield(x, t, T_CH) == ch) return thing_layer(x, t, r, c);
  if (x.k.occl) return top_char(x, r, c) == ch;
  const int cell = r * x.k.C + c;
  return (int)((x.l.backdrop4[cell >> 2] >> ((cell & 3) * 8)) & 0xFF) == ch;
}

// ---- examples/better_scrolly_maze.py -------------------------------------------
__device__ __forceinline__ void prog_bs_player(Ctx& x, int thing) {  // :258-272
  const int a = x.action;
  if ((unsigned)a <= 4u) request_move(x, a == 0 ? -1 : a == 1 ? 1 : 0, a == 2 ? -1 : a == 3 ? 1 : 0);
  if (a == 5) terminate(x);
}
__device__ __forceinline__ void prog_bs_patroller(Ctx& x, int thing) {  // :284-301
  const int s = tfield(x, thing, T_IDX);
  if (x.frame & 1) { request_move(x, 0, 0); return; }
  int r, c;
  sprite_true(x, s, r, c);
  uint32_t f = sflg(x, s);  // bit 2: _moving_east
  if (char_layer_at(x, '#', r, c - 1)) f |= 4u;
  if (char_layer_at(x, '#', r, c + 1)) f &= ~4u;
  sflg_set(x, s, f);
  request_move(x, 0, (f & 4u) ? 1 : -1, 2);  // ... then :298-301, in after_move()
}
// what follows the move in a program's update()
// ---- examples/ordeal.py ---------------------------------------------------------
// The Plot entries of its three classes are the plot words x.v[] (include/pcx.h PCX_PLOT_OD_*); chapter codes: the keys sorted.
enum : int { OD_CASTLE = 0, OD_CAVERN = 1, OD_KANSAS = 2 };
__device__ __forceinline__ void queue_z(Ctx& x, int move, int front) {  // the_plot.change_z_order (plot.py:136-174)
  if (x.nzq < MAX_ZQ) x.l.zq[x.nzq++ * WAVE + x.lane] = (uint32_t)move | ((uint32_t)front << 8);
  else x.err |= ERR_INDEX;
}
__device__ __forceinline__ void od_battle(Ctx& x, int thing) {  // ordeal.py:177-187, where the dragonduck stands after its move
  int r, c;
  sprite_true(x, tfield(x, thing, T_IDX), r, c);
  if (!layer_at(x, x.k.ip, r, c)) return;  // layers['P'] of the last repaint
  x.next = PCX_CHAPTER_NONE;
  terminate(x);
  if (x.v[PCX_PLOT_OD_HAS_SWORD]) { add_reward(x, 1); queue_z(x, thing, x.k.ip); }
  else { add_reward(x, -1); queue_z(x, x.k.ip, thing); }
}
__device__ __forceinline__ void od_save_position(Ctx& x, int thing) {  // ordeal.py:269
  int r, c;
  sprite_true(x, tfield(x, thing, T_IDX), r, c);
  x.v[PCX_PLOT_OD_LAST_POSITION] = (int32_t)pack_pos(r, c);
}
__device__ __forceinline__ void prog_od_sword(Ctx& x, int thing) {  // :121-126
  const int d = tfield(x, thing, T_IDX);
  int pr, pc;
  sprite_true(x, tfield(x, x.k.ip, T_IDX), pr, pc);
  const uint32_t w = drape_rows(x, x.l.cur, d)[(size_t)(pr * x.k.RW + (pc >> 5)) * WAVE];
  if ((w >> (pc & 31)) & 1) { x.v[PCX_PLOT_OD_HAS_SWORD] = 1; add_reward(x, 1); }
  if (x.v[PCX_PLOT_OD_HAS_SWORD])
    for (int i = 0; i < x.k.R * x.k.RW; ++i) drape_rows(x, x.l.cur, d)[(size_t)i * WAVE] = 0;
}
__device__ __forceinline__ void prog_od_dragonduck(Ctx& x, int thing) {  // :143-192
  if (x.frame == 0) return;
  int r, c, pr, pc;
  sprite_true(x, tfield(x, thing, T_IDX), r, c);
  sprite_true(x, tfield(x, x.k.ip, T_IDX), pr, pc);
  const int dr = r > pr ? -1 : r < pr ? 1 : 0, dc = c < pc ? 1 : c > pc ? -1 : 0;  // :147-170: one of the eight motions, or none
  if (dr | dc) request_move(x, dr, dc, 3);  // ... then the battle, in after_move()
  else od_battle(x, thing);
}
__device__ __forceinline__ void prog_od_player(Ctx& x, int thing) {  // :210-269
  const int a = x.action, chap = (int)tfield(x, thing, T_P0), lim_r = x.k.R - 1, lim_c = x.k.C - 1;
  int r, c;
  sprite_true(x, tfield(x, thing, T_IDX), r, c);
  if ((unsigned)a <= 3u) {
    const bool leave = a == 0 ? (chap == OD_KANSAS && r <= 0) : a == 1 ? (chap == OD_CASTLE && r >= lim_r)
                     : a == 2 ? (chap == OD_CAVERN && c <= 0) : (chap == OD_KANSAS && c >= lim_c);
    if (leave) { x.next = a == 0 ? OD_CASTLE : a == 3 ? OD_CAVERN : OD_KANSAS; terminate(x); }
    else { request_move(x, a == 0 ? -1 : a == 1 ? 1 : 0, a == 2 ? -1 : a == 3 ? 1 : 0, 4); return; }  // ... then :269, in after_move()
  } else if (a == 4) {
    x.next = PCX_CHAPTER_NONE;
    terminate(x);
  } else if (x.frame == 0) {  // :252-266: line up with where the last game was left
    const int prior = x.v[PCX_PLOT_OD_PRIOR_CHAPTER], lp = x.v[PCX_PLOT_OD_LAST_POSITION];
    const int lr = pos_r((uint32_t)lp), lc = pos_c((uint32_t)lp);
    int tr = 0, tc = 0;
    bool go = true;
    if (prior == OD_KANSAS && chap == OD_CASTLE) { tr = lim_r; tc = lc; }
    else if (prior == OD_CASTLE && chap == OD_KANSAS) { tr = 0; tc = lc; }
    else if (prior == OD_KANSAS && chap == OD_CAVERN) { tr = lr; tc = 0; }
    else if (prior == OD_CAVERN && chap == OD_KANSAS) { tr = lr; tc = lim_c; }
    else go = false;
    if (go) {
      if (lp == -1) x.err |= ERR_INDEX;  // the_plot['last_position']: KeyError
      else teleport(x, tfield(x, thing, T_IDX), tr, tc);
    }
  }
  od_save_position(x, thing);
}
__device__ __forceinline__ void after_move(Ctx& x, int thing) {
  if (x.mv_post == 3) { od_battle(x, thing); return; }
  if (x.mv_post == 4) { od_save_position(x, thing); return; }
  if (x.mv_post == 2) {  // better_scrolly_maze.py:298-301: the patroller catches the player
    int r, c, pr, pc;
    sprite_true(x, tfield(x, thing, T_IDX), r, c);
    sprite_true(x, tfield(x, x.k.ip, T_IDX), pr, pc);
    if (r == pr && c == pc) terminate(x);
  }
}
__device__ __forceinline__ void prog_bs_cash(Ctx& x, int thing) {  // :311-320
  const int d = tfield(x, thing, T_IDX);
  int pr, pc;
  sprite_true(x, tfield(x, x.k.ip, T_IDX), pr, pc);
  uint32_t* w = drape_rows(x, x.l.cur, d) + (size_t)(pr * x.k.RW + (pc >> 5)) * WAVE;
  if ((*w >> (pc & 31)) & 1) {
    add_reward(x, 100);
    *w &= ~(1u << (pc & 31));
    uint32_t any = 0;
    for (int i = 0; i < x.k.R * x.k.RW; ++i) any |= drape_rows(x, x.l.cur, d)[(size_t)i * WAVE];
    if (!any) terminate(x);
  }
}

// ---- examples/warehouse_manager.py ------------------------------------------
__device__ __forceinline__ void prog_wm_box(Ctx& x, int thing) {  // :214-226
  const int s = tfield(x, thing, T_IDX);
  int r, c;
  sprite_true(x, s, r, c);
  // actions 0..3 = N, S, W, E: the box moves that way iff the player stands on
  // its other side (one probe and one move for all four, not four code paths
  // that a wave with mixed actions would walk one after the other)
  const int a = x.action;
  if ((unsigned)a > 3u) return;
  const int dr = a == 0 ? -1 : a == 1 ? 1 : 0, dc = a == 2 ? -1 : a == 3 ? 1 : 0;
  if (layer_at(x, x.k.ip, r - dr, c - dc)) request_move(x, dr, dc);
}
__device__ __forceinline__ void prog_wm_judge(Ctx& x, int thing) {  // :245-266
  const int d = tfield(x, thing, T_IDX);
  for (int r = 0; r < x.k.R; ++r) row_put(x, x.l.cur, d, r, 0);
  for (int s = 0; s < x.k.NS; ++s) {
    if (!((x.k.box_mask >> s) & 1)) continue;
    int r, c;
    sprite_true(x, s, r, c);
    row_put(x, x.l.cur, d, r, row_get(x, x.l.cur, d, r) | (1ull << c));
  }
  int boxes = 0, on_goals = 0;
  for (int r = 0; r < x.k.R; ++r) {
    uint64_t bits = row_get(x, x.l.cur, d, r);
    const uint64_t goals = (uint64_t)x.l.aux[r * x.k.RW] | (x.k.RW > 1 ? (uint64_t)x.l.aux[r * x.k.RW + 1] << 32 : 0);
    boxes += __popcll(bits);
    bits &= goals;  // backdrop.curtain == backdrop.palette._
    on_goals += __popcll(bits);
    row_put(x, x.l.cur, d, r, bits);
  }
  add_reward(x, on_goals - x.v[0]);
  x.v[0] = on_goals;
  if (x.action == 5 || on_goals == boxes) terminate(x);
}
__device__ __forceinline__ void prog_wm_player(Ctx& x, int thing) {  // :285-295
  const int a = x.action;
  if ((unsigned)a <= 3u) request_move(x, a == 0 ? -1 : a == 1 ? 1 : 0, a == 2 ? -1 : a == 3 ? 1 : 0);
}

// ---- examples/hello_world.py ---------------------------------------------------
__device__ __forceinline__ uint64_t rot_cols(uint64_t bits, int shift, int C) {  // np.roll along axis 1
  const uint64_t m = C >= 64 ? ~0ull : ((1ull << C) - 1ull);
  return shift > 0 ? ((bits << 1) | (bits >> (C - 1))) & m : ((bits >> 1) | (bits << (C - 1))) & m;
}
__device__ __forceinline__ void roll_rows(Ctx& x, int d, int shift) {  // np.roll along axis 0
  const int R = x.k.R;
  if (shift > 0) {
    uint64_t carry = row_get(x, x.l.cur, d, R - 1);
    for (int r = 0; r < R; ++r) { uint64_t t = row_get(x, x.l.cur, d, r); row_put(x, x.l.cur, d, r, carry); carry = t; }
  } else {
    uint64_t carry = row_get(x, x.l.cur, d, 0);
    for (int r = R - 1; r >= 0; --r) { uint64_t t = row_get(x, x.l.cur, d, r); row_put(x, x.l.cur, d, r, carry); carry = t; }
  }
}
__device__ __forceinline__ void prog_hw_rolling(Ctx& x, int thing) {  // :79-91
  const int d = tfield(x, thing, T_IDX), a = x.action;
  if (a < 0) return;
  if (a == 4) terminate(x);
  if (a < 4) {
    const int shift = (a & 1) ? 1 : -1;
    if (a < 2) roll_rows(x, d, shift);
    else for (int r = 0; r < x.k.R; ++r) row_put(x, x.l.cur, d, r, rot_cols(row_get(x, x.l.cur, d, r), shift, x.k.C));
    add_reward(x, 1);
  }
}
__device__ __forceinline__ void prog_hw_sliding(Ctx& x, int thing) {  // :117-123
  const int s = tfield(x, thing, T_IDX), a = x.action;
  if ((unsigned)a > 3u) return;
  const int dx = (int)((tfield(x, thing, T_P0) >> (2 * a)) & 3) - 1, dy = (int)((tfield(x, thing, T_P1) >> (2 * a)) & 3) - 1;
  int vr, vc, vis, prior;
  sprite_get(x, s, vr, vc, vis, prior);
  vc = (vc + dx + x.k.C) % x.k.C;
  vr = (vr + dy + x.k.R) % x.k.R;
  sprite_put(x, s, vr, vc, vis, prior);
}

// ---- examples/extraterrestrial_marauders.py --------------------------------------
// v[0] bunker_hitters, v[1] marauder_hitters (bit per sprite index),
// v[2] last_player_shot, v[3] last_marauder_shot (frame, NEVER if unset);
// MarauderDrape._dx and the RNG draw counter live in the flags word.
__device__ __forceinline__ int em_erode(Ctx& x, int d, int bolt_mask, int& hitters) {
  int hits = 0;
  hitters = 0;
  for (int s = 0; s < x.k.NS; ++s) {
    if (!((bolt_mask >> s) & 1)) continue;
    const int cell = ssnap(x, s);
    if (cell < 0) continue;
    const int r = cell / x.k.C, c = cell - r * x.k.C;
    if (!bit_at(x, x.l.cur, d, r, c)) continue;
    // layers[bolt][r, c]: the bolt is at this cell of the last repaint; with occlusion it must
    // also be the thing in front there
    const uint32_t m = present_at(x, r, c);
    const int top = top_thing(x, m);  // (m holds the bolt itself: never empty)
    if (x.k.occl && top != (int)x.l.s2t[s]) continue;
    row_put(x, x.l.cur, d, r, row_get(x, x.l.cur, d, r) & ~(1ull << c));
    ++hits;
    // board[hits]: the character drawn on top of the hit cell (with occlusion
    // that is this bolt; without, a bolt in front of it may be the one named)
    if (tfield_v(x, top, T_KIND) == 0) hitters |= 1 << tfield_v(x, top, T_IDX);
  }
  return hits;
}
__device__ __forceinline__ void prog_em_bunker(Ctx& x, int thing) {  // :113-120
  int hitters;
  const int hits = em_erode(x, tfield(x, thing, T_IDX), x.k.bolt_mask_all, hitters);
  add_reward(x, -hits);
  x.v[0] = hitters;
}
__device__ __forceinline__ void prog_em_marauder(Ctx& x, int thing, int& dxv) {  // :141-163
  const int d = tfield(x, thing, T_IDX), R = x.k.R, C = x.k.C;
  int hitters;
  const int hits = em_erode(x, d, x.k.bolt_mask_up, hitters);
  add_reward(x, 10 * hits);
  x.v[1] = hitters;
  int total = 0;
  uint64_t any_edge = 0;
  for (int r = 0; r < R; ++r) { uint64_t b = row_get(x, x.l.cur, d, r); total += __popcll(b); any_edge |= b; }
  const bool row10 = R > 10 ? row_get(x, x.l.cur, d, 10) != 0 : false;
  if (R <= 10) x.err |= ERR_INDEX;
  if (total == 0 || row10) { terminate(x); return; }
  int period = (total - 1) / 8;  // total // 8.0000001
  if (period < 1) period = 1;
  if (x.frame % period) return;
  if ((any_edge & 1ull) || ((any_edge >> (C - 1)) & 1ull)) {
    dxv = -dxv;
    roll_rows(x, d, 1);
  }
  for (int r = 0; r < R; ++r) row_put(x, x.l.cur, d, r, rot_cols(row_get(x, x.l.cur, d, r), dxv, C));
}
__device__ __forceinline__ void prog_em_player(Ctx& x, int thing) {  // :178-186
  if (x.action == 0 || x.action == 1) request_move(x, 0, x.action == 0 ? -1 : 1);
  else if (x.action == 4) terminate(x);
}
__device__ __forceinline__ void prog_em_upbolt(Ctx& x, int thing) {  // :198-220
  const int s = tfield(x, thing, T_IDX);
  int vr, vc, vis, prior;
  sprite_get(x, s, vr, vc, vis, prior);
  if (vis) {
    if (((x.v[0] | x.v[1]) >> s) & 1) { teleport(x, s, -1, -1); return; }
    request_move(x, -1, 0);
  } else if (x.action == 2) {
    if (x.v[2] == x.frame) return;
    x.v[2] = x.frame;
    int pr, pc;
    sprite_true(x, tfield(x, x.k.ip, T_IDX), pr, pc);
    teleport(x, s, pr - 1, pc);
  }
}
constexpr uint64_t EM_RNG_SALT = 0x4D415241554445ull;
__device__ __forceinline__ void prog_em_downbolt(Ctx& x, int thing, uint32_t& draws, int64_t genv) {  // :232-256
  const int s = tfield(x, thing, T_IDX), R = x.k.R;
  int vr, vc, vis, prior;
  sprite_get(x, s, vr, vc, vis, prior);
  if (vis) {
    if ((x.v[0] >> s) & 1) { teleport(x, s, -1, -1); return; }
    int r, c, pr, pc;
    sprite_true(x, s, r, c);
    sprite_true(x, tfield(x, x.k.ip, T_IDX), pr, pc);
    if (r == pr && c == pc) terminate(x);
    request_move(x, 1, 0);
  } else {
    if (x.v[3] == x.frame) return;
    x.v[3] = x.frame;
    // columns of the layer of 'X' in the last repaint that hold any X
    // (np.flatnonzero(layers['X'].any(axis=0)), :246).  With occlusion the layer is X's snapshot minus the
    // cells of what stands in front of it: usually a handful of sprites (the bolts), whose rows and column bits
    // fit eight register pairs -- a row of the layer is then its snapshot row and eight selects, not a walk
    // over the things per row.
    const uint32_t dx = tfield(x, x.k.tx, T_IDX);
    constexpr int MAXC = 8;
    const uint32_t ab_s = x.k.occl ? above_sprites(x, x.k.tx) : 0u;
    const bool slow = x.k.occl && (x.k.zdyn || above_drapes(x, x.k.tx) != 0 || __popc(ab_s) > MAXC);  // (uniform)
    uint32_t cpos[MAXC];  // row << 8 | column of a covering sprite, 0xFFFFFFFF: none
    {
      uint32_t m = slow ? 0u : ab_s;
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const bool has = m != 0u;
        const int cell = has ? ssnap(x, __ffs((int)m) - 1) : -1;
        const uint32_t r = __umulhi((uint32_t)(cell >= 0 ? cell : 0), x.k.magic_c);
        cpos[i] = cell >= 0 ? (r << 8) | ((uint32_t)cell - r * (uint32_t)x.k.C) : 0xFFFFFFFFu;
        m &= m - 1u;
      }
    }
    auto layer_row = [&](int r) {
      if (slow) return drape_layer_row(x, x.k.tx, r);
      uint64_t cover = 0;
#pragma unroll
      for (int i = 0; i < MAXC; ++i) cover |= (cpos[i] >> 8) == (uint32_t)r ? 1ull << (cpos[i] & 63u) : 0ull;
      return row_get(x, x.l.snapd, dx, r) & ~cover;
    };
    uint64_t cols = 0;
    for (int r = 0; r < R; ++r) cols |= layer_row(r);
    const int n = __popcll(cols);
    if (n == 0) { x.err |= ERR_INDEX; return; }  // np.random.choice([]) raises
    const uint64_t seed = x.rng_seed ^ EM_RNG_SALT;
    int pick = (int)(action_hash(seed, (uint64_t)genv, (uint64_t)draws) % (uint32_t)n);
    ++draws;
    // the pick-th set column (0-based): bisection on popcounts, six steps
    int col = 0;
    {
      uint64_t v = cols;
#pragma unroll
      for (int w = 32; w >= 1; w >>= 1) {
        const int below = __popcll(v & ((1ull << w) - 1ull));
        const bool up = pick >= below;
        pick = up ? pick - below : pick;
        col = up ? col + w : col;
        v = up ? v >> w : v;
      }
    }
    int row = 0;  // the lowest X of that column (np.max(np.flatnonzero(layers['X'][:, col])), :248)
    for (int r = 0; r < R; ++r)
      if ((layer_row(r) >> col) & 1) row = r;
    teleport(x, s, row + 1, col);
  }
}

// Render phase (rendering.py:85-184, :187-301).  Occlusion was resolved per
// environment in the logic phase, so every thing's layer is a mask already in
// LDS and painting is order-free.  One iteration = one board dword (4 cells)
// per lane, consecutive lanes = consecutive dwords of one or two environments,
// so every plane store of a wave covers 256 contiguous bytes; the workgroup's
// waves take the iterations round-robin.  NTC = the thing count rounded up:
// the per-thing loop is unrolled and what describes a thing (where its mask
// lives, its character, its layer plane) is read once, into scalar registers.
template <int NTC>
__device__ __forceinline__ void render_planes(const Consts& k, const L& l, const Ptrs& P, const pcx_buffers& out, int64_t env0, int lane,
                                              int wave, int nwaves, bool any_skip) {
  const int QW = k.QW, FWP = k.FW | 1, pitch = k.pitch, NT = k.NT;
  const uint32_t env_stride = (uint32_t)(1 + k.L) * (uint32_t)pitch;
  uint8_t* const blk = out.planes + (size_t)env0 * env_stride;
  auto uni32 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  uint32_t src[NTC], ch4[NTC], plane[NTC];  // src: LDS word offset of the mask source, bit 31 = sprite
#pragma unroll
  for (int t = 0; t < NTC; ++t) {
    src[t] = ch4[t] = plane[t] = 0;
    if (t >= NT) continue;
    // (the specialised build: from the constants -- whether thing t is a sprite or a drape is then known where the loop below
    // branches on it, its LDS reads are all requested at the top of an iteration instead of one thing after the other, each behind
    // its own s_waitcnt: the "LDS-latency-bound per wave" of profiles/r06_generic.md section 2)
#ifdef PCX_GENERIC_SPEC
    auto tw = [&](int f) { return spec::TAB[spec::K.l_things + t * T_WORDS + f]; };
#else
    auto tw = [&](int f) { return l.things[t * T_WORDS + f]; };
#endif
    const uint32_t kind = tw(T_KIND), idx = tw(T_IDX);
    src[t] = uni32(kind == 0 ? (0x80000000u | (idx * WAVE)) : idx * WAVE * (uint32_t)FWP);
    ch4[t] = uni32(tw(T_CH) * 0x01010101u);
    plane[t] = uni32((1 + tw(T_LAYER)) * (uint32_t)pitch);
  }
  constexpr int MAXB = 8;
  const int NB = k.n_bchars;
  uint32_t bplane[MAXB];
#pragma unroll
  for (int b = 0; b < MAXB; ++b) bplane[b] = b < NB ? uni32((1 + l.laybc[b]) * (uint32_t)pitch) : 0u;
  const bool occl = k.occl != 0;
  // the feature-array epilogue: the board dword is in a register, plane f of the array is (board == feat_ch[f]) as
  // four float32 -- one 16-byte store per plane where boards are whole dwords, cell by cell otherwise (a plane of
  // rows * cols floats then starts at any multiple of 4 bytes)
  const bool feat = P.feat != nullptr, put_layers = !(feat && P.feat_skip >= 1), put_board = !(feat && P.feat_skip >= 2);
  const int cells = k.cells;
  const bool feat_x4 = (cells & 3) == 0;
#pragma unroll 1
  for (int it = wave; it < QW; it += nwaves) {
    const uint32_t f = (uint32_t)it * WAVE + lane;
    const uint32_t e = __umulhi(f, k.magic_q), q = f - e * QW;  // f / QW by 32-bit reciprocal
    if (any_skip && l.skip[e]) continue;
    uint8_t* const dst = blk + e * env_stride + q * 4;
    const uint32_t flat_at = e * FWP + (q >> 3), sh = (q & 7) * 4;
    uint32_t d = l.backdrop4[q], uni = 0;
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
      if (t >= NT) break;
      uint32_t m, lay;
      if (src[t] & 0x80000000u) {
        const uint32_t at = (src[t] & 0x7FFFFFFFu) + e;
        const uint2 sd = l.sdesc[at];
        m = sd.x == q ? sd.y : 0u;
        lay = m;
        if (!occl) {  // rendering.py:236-278: the raw mask
          const uint2 sr = l.sdescraw[at];
          lay = sr.x == q ? sr.y : 0u;
        }
      } else {
        const uint32_t bits = (l.flat[src[t] + flat_at] >> sh) & 0xFu;
        const uint32_t m01 = (bits * 0x00204081u) & 0x01010101u;
        m = (m01 << 8) - m01;
        lay = m;
        if (!occl) {
          const uint32_t rb = (l.flatraw[src[t] + flat_at] >> sh) & 0xFu;
          const uint32_t r01 = (rb * 0x00204081u) & 0x01010101u;
          lay = (r01 << 8) - r01;
        }
      }
      uni |= m;
      d = (d & ~m) | (ch4[t] & m);
      // rendering.py:177-179: after occlusion a thing's layer is its own mask
      if (put_layers) *reinterpret_cast<uint32_t*>(dst + plane[t]) = lay & 0x01010101u;
    }
    if (put_board) *reinterpret_cast<uint32_t*>(dst) = d;
    if (feat) {
      float* const fe = P.feat + ((size_t)(env0 + e) * (size_t)P.feat_depth) * (size_t)cells + (size_t)q * 4;
      for (int f = 0; f < P.feat_depth; ++f) {
        const uint32_t x = d ^ ((uint32_t)P.feat_ch[f] * 0x01010101u);  // a zero byte where the cell shows the character
        const pcx_f32x4 v = {(x & 0xFFu) ? 0.0f : 1.0f, (x & 0xFF00u) ? 0.0f : 1.0f, (x & 0xFF0000u) ? 0.0f : 1.0f, (x & 0xFF000000u) ? 0.0f : 1.0f};
        float* const fp = fe + (size_t)f * (size_t)cells;
        const int left = cells - (int)q * 4;  // (< 4: the last dword of a plane, its cells beyond the board are padding)
        if (feat_x4) {
          *reinterpret_cast<pcx_f32x4*>(fp) = v;
        } else if (left >= 4) {  // boards that are no whole number of dwords: planes start at any multiple of 4 bytes --
          struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };  // still one global_store_dwordx4 (dword-aligned is enough)
          *reinterpret_cast<F4*>(fp) = F4{v.x, v.y, v.z, v.w};
        } else {
          fp[0] = v.x;
          if (left > 1) fp[1] = v.y;
          if (left > 2) fp[2] = v.z;
        }
      }
    }
    // characters only the backdrop paints: their layer is the precomputed mask,
    // minus (with occlusion) whatever a thing covers
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
      if (b >= NB) break;
      const uint32_t m = l.bdmask[b * QW + q];
      if (put_layers) *reinterpret_cast<uint32_t*>(dst + bplane[b]) = occl ? m & ~uni : m;
    }
    for (int b = MAXB; b < NB && put_layers; ++b)
      *reinterpret_cast<uint32_t*>(dst + (1 + l.laybc[b]) * pitch) = occl ? l.bdmask[b * QW + q] & ~uni : l.bdmask[b * QW + q];
  }
}

// ---- fused croppers (include/pcx.h pcx_engine_fuse_croppers) -----------------
// ScrollingCropper._centroid of a drape (cropping.py:590-598): int(np.median(.)) of the set cells' row and
// column indices, taken from the raw curtain's bit rows (lane == environment).  False: the curtain is empty.
__device__ __forceinline__ bool drape_centroid(const Ctx& x, int d, int& crow, int& ccol) {
  const int R = x.k.R, C = x.k.C, RW = x.k.RW;
  const uint32_t* rows = drape_rows(x, x.l.cur, d);
  int n = 0;
  for (int r = 0; r < R; ++r)
    for (int w = 0; w < RW; ++w) {
      const int nb = C - 32 * w < 32 ? C - 32 * w : 32;
      n += __popc(rows[(r * RW + w) * WAVE] & (nb < 32 ? (1u << nb) - 1u : 0xFFFFFFFFu));
    }
  // the two middle order statistics (0-based) of the sorted index list; their mean, truncated
  const int lo_rank = (n - 1) / 2, hi_rank = n / 2;
  int seen = 0, lo = -1, hi = -1;
  for (int r = 0; r < R; ++r) {
    int cnt = 0;
    for (int w = 0; w < RW; ++w) {
      const int nb = C - 32 * w < 32 ? C - 32 * w : 32;
      cnt += __popc(rows[(r * RW + w) * WAVE] & (nb < 32 ? (1u << nb) - 1u : 0xFFFFFFFFu));
    }
    lo = lo < 0 && seen + cnt > lo_rank ? r : lo;
    hi = hi < 0 && seen + cnt > hi_rank ? r : hi;
    seen += cnt;
  }
  crow = (lo + hi) >> 1;
  seen = 0; lo = -1; hi = -1;
  for (int c = 0; c < C; ++c) {
    int cnt = 0;
    for (int r = 0; r < R; ++r) cnt += (rows[(r * RW + (c >> 5)) * WAVE] >> (c & 31)) & 1u;
    lo = lo < 0 && seen + cnt > lo_rank ? c : lo;
    hi = hi < 0 && seen + cnt > hi_rank ? c : hi;
    seen += cnt;
  }
  ccol = (lo + hi) >> 1;
  return n > 0;
}

// The logic wave moves every fused window for its environment (ScrollingCropper.crop, cropping.py:393-426)
// and leaves the corners in LDS for render_windows.
__device__ __forceinline__ void move_windows(const Ctx& x, const crop::FusedCrops* fc, int64_t env, uint32_t* wcorner) {
  const int n = fc->n;
  for (int w = 0; w < n; ++w) {
    const crop::FusedWindow& fw = fc->w[w];
    int top = fw.top, left = fw.left;
    if (fw.scrolling) {
      bool has = x.frame != 0 && fw.has_corner[env] != 0;  // a new episode is a new Engine (cropping.py:378-391)
      int wrow = fw.corner[2 * env], wcol = fw.corner[2 * env + 1];
      bool have = false;
      int crow = 0, ccol = 0;
      for (int i = 0; i < fw.n_track; ++i) {  // :544-558 the first entity of to_track that has a centroid
        int r = 0, c = 0;
        bool ok;
        if (fw.track_kind[i] == 0) {
          sprite_true(x, fw.track_sprite[i], r, c);
          ok = (sflg(x, fw.track_sprite[i]) & 1u) != 0;
        } else {
          ok = drape_centroid(x, fw.track_sprite[i], r, c);
        }
        if (!have && ok) { crow = r; ccol = c; have = true; }
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
    wcorner[w * WAVE + x.lane] = err ? stream::WCORNER_NONE : (((uint32_t)top & 0xFFFFu) | ((uint32_t)left << 16));
  }
}

// _do_crop (cropping.py:118-227) out of LDS: one (environment, output dword) task per lane; a run of window
// cells that lies in one window row is a run of consecutive board cells, composed like a board dword
// (backdrop bytes, each curtain's bits, the painted sprites' cells -- occlusion is resolved already), the
// pad character outside the board; every layer is `board == c` of the finished dword (rendering.py:177-179).
__device__ __forceinline__ void render_windows(const Consts& k, const L& l, const crop::FusedCrops* fc, const uint32_t* wcorner,
                                               int64_t env0, int lane, int wave, int nwaves) {
  auto eq01 = [](uint32_t v, uint32_t c4) {
    const uint32_t y = v ^ c4;
    return (~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) >> 7) & 0x01010101u;
  };
  auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  const int FWP = k.FW | 1, Rv = k.R, Cv = k.C, QWsrc = k.QW, NS = k.NS, ND = k.ND, Lc = k.L;
  const int n = fc->n;
  for (int w = 0; w < n; ++w) {
    const crop::FusedWindow& fw = fc->w[w];
    const int rows = fw.rule.rows, cols = fw.rule.cols;
    const uint32_t opitch = (uint32_t)fw.out_pitch, qw = opitch >> 2, total = (uint32_t)WAVE * qw;
    const uint32_t ostride = (uint32_t)(1 + Lc) * opitch;
    uint8_t* const obase = fw.out + (size_t)env0 * ostride;
    const uint32_t pad = (uint32_t)(fw.rule.pad_char & 0xFF);
    const uint32_t magic_qw = 0xFFFFFFFFu / qw, magic_cols = 0xFFFFFFFFu / (uint32_t)cols;
    for (uint32_t f = (uint32_t)wave * WAVE + (uint32_t)lane; f < total; f += (uint32_t)nwaves * WAVE) {
      uint32_t e = __umulhi(f, magic_qw), q = f - e * qw;  // f / qw: the estimate is at most one short
      if (q >= qw) { q -= qw; ++e; }
      const uint32_t cw = wcorner[w * WAVE + e];
      if (l.skip[e] || cw == stream::WCORNER_NONE) continue;
      const int top = (int)(int16_t)(cw & 0xFFFFu), left = (int)(int16_t)(cw >> 16);
      const uint32_t cell0 = q * 4u;
      uint32_t orow = __umulhi(cell0, magic_cols), ocol = cell0 - orow * (uint32_t)cols;
      if (ocol >= (uint32_t)cols) { ocol -= (uint32_t)cols; ++orow; }
      const uint32_t eF = e * (uint32_t)FWP;
      uint32_t od = 0;
      int done = 0, wr = (int)orow, wc = (int)ocol;
      while (done < 4) {
        const int nrun = cols - wc < 4 - done ? cols - wc : 4 - done;
        const bool real_row = wr < rows;  // rows past the window are plane padding: zeros
        const int sr = top + wr, sc = left + wc;
        const bool row_in = real_row && (unsigned)sr < (unsigned)Rv;
        const int lo = sc < 0 ? -sc : 0, hi = Cv - sc < nrun ? Cv - sc : nrun;  // the run's columns inside the board
        const bool any_in = row_in && lo < hi;
        const uint32_t a = any_in ? (uint32_t)(sr * Cv + sc + lo) : 0u;
        const uint32_t qa = a >> 2, qb = (int)qa + 1 < QWsrc ? qa + 1 : qa, ph = a & 3u;
        uint32_t d = __builtin_amdgcn_alignbyte(l.backdrop4[qb], l.backdrop4[qa], ph);
        const uint32_t w0 = a >> 5, w1 = (int)w0 + 1 < FWP ? w0 + 1 : w0;
        for (int dd = 0; dd < ND; ++dd) {
          const uint32_t base = (uint32_t)dd * WAVE * (uint32_t)FWP + eF;
          const uint64_t pair = (uint64_t)l.flat[base + w0] | ((uint64_t)l.flat[base + w1] << 32);
          const uint32_t bits = (uint32_t)(pair >> (a & 31u)) & 0xFu;
          const uint32_t m01 = (bits * 0x00204081u) & 0x01010101u;
          const uint32_t m = (m01 << 8) - m01;
          const uint32_t ch4 = uni(l.things[l.d2t[dd] * T_WORDS + T_CH]) * 0x01010101u;
          d = (d & ~m) | (ch4 & m);
        }
        for (int s2 = 0; s2 < NS; ++s2) {
          const uint2 sd = l.sdesc[s2 * WAVE + e];  // {board dword, byte mask}; dword 0xFFFFFFFF: not painted
          const uint32_t scell = (sd.x << 2) | ((uint32_t)__builtin_ctz(sd.y) >> 3);
          const uint32_t delta = scell - a;
          const uint32_t m = sd.x != 0xFFFFFFFFu && delta < 4u ? 0xFFu << (8u * delta) : 0u;
          const uint32_t ch4 = uni(l.things[l.s2t[s2] * T_WORDS + T_CH]) * 0x01010101u;
          d = (d & ~m) | (ch4 & m);
        }
        const int nin = any_in ? hi - lo : 0;
        const uint32_t keep = nin >= 4 ? 0xFFFFFFFFu : (1u << (8 * nin)) - 1u;
        uint32_t run = any_in ? (d & keep) << (8 * lo) : 0u;
        const uint32_t in_mask = any_in ? keep << (8 * lo) : 0u;
        const uint32_t run_mask = nrun >= 4 ? 0xFFFFFFFFu : (1u << (8 * nrun)) - 1u;
        run |= real_row ? (pad * 0x01010101u) & run_mask & ~in_mask : 0u;
        od |= run << (8 * done);
        done += nrun;
        wc += nrun;
        if (wc >= cols) { wc = 0; ++wr; }
      }
      uint8_t* const dst = obase + (size_t)e * ostride + 4u * q;
      *reinterpret_cast<uint32_t*>(dst) = od;
      for (int kk = 0; kk < Lc; ++kk)
        *reinterpret_cast<uint32_t*>(dst + (size_t)(1 + kk) * opitch) = eq01(od, (uint32_t)k.chars[kk] * 0x01010101u);
    }
  }
}

// The state words of work unit `unit` (64 consecutive environments; SoA over the batch: word w of the unit is one 256-byte row)
// straight into the per-lane LDS columns the logic phase keeps them in -- sprite positions, curtain rows, scrolling protocol words --
// and the scalars into an inbox, all by LDS-DMA issued back to back: ONE memory round trip instead of a dozen (a copy loop through
// registers has four loads in flight).  `off`: word offset of the worker's copy of the per-lane arrays (0: the one-group layout).
// The issuing wave waits on vmcnt itself.
__device__ __forceinline__ void dma_state_rows(const Consts& k, const Ptrs& P, const StepArgs& a, uint32_t* lds, int off, int64_t unit, int lane) {
    const uint32_t* const sb = stream::uniform_words(P.state + unit * P.state_unit);
    const uint32_t l0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)stream::lds_byte_address(lds));
    const uint32_t vo = 4u * (uint32_t)lane;
    const int64_t bpw = P.state_row;
    auto row = [&](int word, int lds_word) { stream::lds_dma_row(sb + (int64_t)word * bpw, vo, l0 + 4u * (uint32_t)(lds_word + off)); };
    for (int w = 0; w < W_SPRITES; ++w) row(w, k.l_inbox + w * WAVE);
    for (int s = 0; s < k.NS; ++s) row(W_SPRITES + s, k.l_pos + s * WAVE);
    for (int w = 0; w < (k.NS + 3) / 4; ++w) row(k.w_sflags + w, k.l_inbox + (IB_SFLAGS + w) * WAVE);
    const int ndw0 = k.ND * k.R * k.RW;
    for (int i = 0; i < ndw0; ++i) row(k.w_drapes + i, k.l_cur + i * WAVE);
    if (k.has_scroll) {
      row(k.w_scroll, k.l_inbox + IB_SCROLL * WAVE);
      for (int d = 0; d < k.ND; ++d) row(k.w_scroll + 1 + d, k.l_corner + d * WAVE);
      for (int s = 0; s < k.NS; ++s) {
        row(k.w_scroll + 1 + k.ND + 2 * s, k.l_pmask + s * WAVE);
        row(k.w_scroll + 2 + k.ND + 2 * s, k.l_pframe + s * WAVE);
      }
    }
    if (k.zdyn) { row(k.w_z, k.l_inbox + IB_Z0 * WAVE); row(k.w_z + 1, k.l_inbox + IB_Z1 * WAVE); }
    if (k.w_next >= 0) row(k.w_next, k.l_inbox + IB_NEXT * WAVE);
    if (a.mode == 0 && !a.hashed && unit * WAVE + lane < P.batch)  // (the tape has `batch` entries, not the padded count)
      stream::lds_dma_row(stream::uniform_words(reinterpret_cast<const uint32_t*>(a.actions) + unit * WAVE), vo,
                          l0 + 4u * (uint32_t)(k.l_inbox + IB_ACTION * WAVE + off));
}

// The LDS pointers of one group: the staged tables (shared), the per-lane arrays at word offset `off` (0: the one-group layout).
__device__ __forceinline__ L make_L(uint32_t* lds, const Consts& k, int off) {
  L l;
  l.things = lds + k.l_things; l.z = lds + k.l_z; l.sched = lds + k.l_sched;
  l.backdrop4 = lds + k.l_backdrop; l.bdmask = lds + k.l_bdmask; l.aux = lds + k.l_aux;
  l.init = lds + k.l_init; l.initd = lds + k.l_initd; l.laybc = lds + k.l_laybc; l.s2t = lds + k.l_s2t; l.d2t = lds + k.l_d2t;
  l.pos = lds + off + k.l_pos; l.flg = lds + off + k.l_flg; l.snap = reinterpret_cast<int32_t*>(lds + off + k.l_snap);
  l.cur = lds + off + k.l_cur; l.snapd = lds + off + k.l_snapd;
  l.corner = lds + off + k.l_corner; l.pmask = lds + off + k.l_pmask; l.pframe = lds + off + k.l_pframe;
  l.dir = lds + k.l_dir; l.zord = lds + off + k.l_zord; l.zabove = lds + off + k.l_zabove; l.zabove_s = lds + off + k.l_zabove_s;
  l.zabove_d = lds + off + k.l_zabove_d; l.zq = lds + off + k.l_zq; l.ztmp = lds + off + k.l_ztmp;
  l.flat = lds + off + k.l_flat; l.sdesc = reinterpret_cast<uint2*>(lds + off + k.l_sdesc); l.skip = lds + off + k.l_skip;
  l.flatraw = lds + off + k.l_flatraw; l.sdescraw = reinterpret_cast<uint2*>(lds + off + k.l_sdescraw); l.wcorner = lds + off + k.l_wcorner;
#ifdef PCX_GENERIC_SPEC  // tables that are only ever read at a uniform index: from the constants
  l.z = spec::TAB + k.l_z; l.sched = spec::TAB + k.l_sched; l.init = spec::TAB + k.l_init; l.initd = spec::TAB + k.l_initd;
  l.laybc = spec::TAB + k.l_laybc; l.s2t = spec::TAB + k.l_s2t; l.d2t = spec::TAB + k.l_d2t; l.dir = spec::TAB + k.l_dir;
#endif
  l.bdcode = lds + k.l_bdcode; l.codes = nullptr;
  return l;
}

// The logic phase of one group of 64 environments, lane == environment (engine.py:583-847 for every lane's environment): the update
// groups in order, _apply_and_clear_plot, the state write-back, and the occlusion of the final repaint left as render descriptors
// (l.flat / l.sdesc / l.skip: what render_planes reads).  `ib_rows`: the inbox dma_state_rows() filled (row r at [r * WAVE]);
// `logic_wave`: this wave steps the group (the others of a workgroup only share the render loop).
// `slot_free()`: called (by the lanes that step, and once more by all) before anything is written where the render side reads --
// l.codes and l.skip -- so that a persistent logic worker can wait for its hand-over slot as late as possible.
struct NoWait { __device__ __forceinline__ void operator()() const {} };
template <typename SlotFree = NoWait>
__device__ __forceinline__ void logic_phase(const Consts& k, const L& l, const Ptrs& P, const StepArgs& a, const pcx_buffers& out,
                                            const crop::FusedCrops* fc, const uint32_t* ib_rows, int64_t env0, int lane, bool logic_wave, bool timing,
                                            SlotFree slot_free = SlotFree()) {
  const int64_t env = env0 + lane;
  const unsigned long long t_start = timing ? __builtin_readcyclecounter() : 0ull;
  unsigned long long c_sec[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, c_prog[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool live = logic_wave && env < P.batch;
  const int64_t bp = P.bpad, srow = P.state_row;
  uint32_t* st = P.state + (env0 >> 6) * P.state_unit + lane;  // word w at st[w * srow]
  const uint32_t* const ib = ib_rows + lane;  // row r at [r * WAVE]
  uint32_t flags = 0;
  bool skip = !live, do_reset = false;
  int action = PCX_ACTION_NONE;
  if (live) {
    flags = ib[W_FLAGS * WAVE];
    if (a.mode == 1) { do_reset = a.reset_mask ? a.reset_mask[env] != 0 : true; skip = !do_reset; }
    else if (flags & F_OVER) {
      do_reset = a.auto_reset != 0; skip = !do_reset;
      if (skip) { out.reward[env] = 0; out.reward_set[env] = 0; out.discount[env] = 0.0f; }  // a finished environment left alone reports an empty step (pcx.h)
    }
    else action = a.hashed ? (int)(action_hash(a.seed, (uint64_t)(a.env_offset + env), (uint64_t)a.t) % (uint32_t)k.n_actions)
                           : (int)ib[IB_ACTION * WAVE];
    if (action < 0) action = PCX_ACTION_NONE;
  }
  const int ndw = k.ND * k.R * k.RW;
  if (!skip) {
    Ctx x{k, l, lane, 0, action, 0, 0, 0, 0, 1.0f, {0, 0, 0, 0}, 0, 0, 0, 0, 0xFFFFFFFFu, 0, 0, 0, 0, 0, PCX_CHAPTER_UNSET, P.seed};
    if (k.w_next >= 0 && !do_reset) x.next = (int)ib[IB_NEXT * WAVE];  // (a new episode starts with the Story's own)
    // bits 8..15 of the flags word: MarauderDrape._dx + 1; W_RNG: RNG draws so far (survive resets)
    uint32_t draws = ib[W_RNG * WAVE];
    int dxv;
    if (do_reset) {  // engine.py:520-581: fresh template state, pre-showtime render
      x.frame = (int)l.init[W_FRAME];
      for (int j = 0; j < 4; ++j) x.v[j] = P.plot_in ? P.plot_in[(size_t)j * bp + env] : (int32_t)l.init[W_V0 + j];
      dxv = (int)((l.init[W_FLAGS] >> 8) & 0xFF) - 1;
      PCX_SPEC_UNROLL
      for (int s = 0; s < k.NS; ++s) {
        spos_set(x, s, l.init[W_SPRITES + s]);
        sflg_set(x, s, (l.init[k.w_sflags + (s >> 2)] >> (8 * (s & 3))) & 0xFF);
      }
      for (int i = 0; i < ndw; ++i) l.cur[i * WAVE + lane] = l.initd[i];
      if (k.has_scroll) {
        x.registered = 0;
        for (int d = 0; d < k.ND; ++d) l.corner[d * WAVE + lane] = l.init[k.w_scroll + 1 + d];
        for (int s = 0; s < k.NS; ++s) { l.pmask[s * WAVE + lane] = 0; l.pframe[s * WAVE + lane] = 0; }
      }
      if (k.zdyn) for (int z = 0; z < k.NT; ++z) l.zord[z * WAVE + lane] = (uint32_t)z;  // the template's order
      x.action = PCX_ACTION_NONE;
    } else {
      x.frame = (int)ib[W_FRAME * WAVE];
      x.err = (flags >> F_ERR_SHIFT) & 7u;
      for (int j = 0; j < 4; ++j) x.v[j] = (int32_t)ib[(W_V0 + j) * WAVE];
      dxv = (int)((flags >> 8) & 0xFF) - 1;
      // (positions, curtain rows, window corners and the protocol's per-sprite words are in their columns already)
#ifdef PCX_SREGS
      PCX_SPEC_UNROLL
      for (int s = 0; s < k.NS; ++s) spos_set(x, s, l.pos[s * WAVE + lane]);  // ... the positions on their way to the registers
#endif
      PCX_SPEC_UNROLL
      for (int w = 0; w < (k.NS + 3) / 4; ++w) {
        const uint32_t f = ib[(IB_SFLAGS + w) * WAVE];
        PCX_SPEC_UNROLL
        for (int j = 0; j < 4; ++j) if (4 * w + j < k.NS) sflg_set(x, 4 * w + j, (f >> (8 * j)) & 0xFF);
      }
      if (k.has_scroll) x.registered = ib[IB_SCROLL * WAVE];
    }
    if (k.zdyn) {  // four bits per place: the thing at place z
      if (!do_reset) {
        const uint32_t z0 = ib[IB_Z0 * WAVE], z1 = ib[IB_Z1 * WAVE];
        for (int z = 0; z < k.NT; ++z) l.zord[z * WAVE + lane] = ((z < 8 ? z0 : z1) >> (4 * (z & 7))) & 0xFu;
      }
      derive_above(x);
    }
    snapshot(x);  // what the previous frame's last repaint showed
    if (timing) c_sec[0] = __builtin_readcyclecounter() - t_start;  // state load
    const unsigned long long t_play = timing ? __builtin_readcyclecounter() : 0ull;
    // ---- Engine.play(): engine.py:698-735 --------------------------------
    x.frame += 1;
    const int64_t genv = P.env_offset + env;
    // the schedule: thing | update group << 8 per entry, groups ascending (engine.py:706-735)
#ifdef PCX_GENERIC_SPEC
    auto sched_at = [&](int i2) { return spec::TAB[spec::K.l_sched + i2]; };
#else
    auto sched_at = [&](int i2) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)l.sched[i2]); };
#endif
    PCX_SPEC_UNROLL
    for (int i = 0; i < k.NT; ++i) {
      {
        const uint32_t sc = sched_at(i);
        const int thing = (int)(sc & 0xFF);
        const unsigned long long tp0 = timing ? __builtin_readcyclecounter() : 0ull;
        const int sgroup = k.n_sgroups > 1 ? (int)tfield(x, thing, T_GROUP) : 0;
        if (k.n_sgroups > 1) {  // this entity's scrolling group comes into view
          const uint32_t ov = (x.orders >> (8 * sgroup)) & 0xFFu;
          x.order_valid = ov & 1; x.o0 = (int)((ov >> 1) & 3u) - 1; x.o1 = (int)((ov >> 3) & 3u) - 1;
          x.gsprites = sgroup == 0 ? k.group_sprites[0] : sgroup == 1 ? k.group_sprites[1] : sgroup == 2 ? k.group_sprites[2] : k.group_sprites[3];
        }
        if (k.n_dir) {
          const uint32_t prog = tfield(x, thing, T_PROG);
          if (prog == PCX_PROG_WALKER || prog == PCX_PROG_SCROLLY || prog == PCX_PROG_STATIC) issue_directives(x, thing);
        }
        switch (tfield(x, thing, T_PROG)) {
          case PCX_PROG_WM_BOX: prog_wm_box(x, thing); break;
          case PCX_PROG_WM_JUDGE: prog_wm_judge(x, thing); break;
          case PCX_PROG_WM_PLAYER: prog_wm_player(x, thing); break;
          case PCX_PROG_HW_ROLLING: prog_hw_rolling(x, thing); break;
          case PCX_PROG_HW_SLIDING: prog_hw_sliding(x, thing); break;
          case PCX_PROG_EM_PLAYER: prog_em_player(x, thing); break;
          case PCX_PROG_EM_BUNKER: prog_em_bunker(x, thing); break;
          case PCX_PROG_EM_MARAUDER: prog_em_marauder(x, thing, dxv); break;
          case PCX_PROG_EM_UPBOLT: prog_em_upbolt(x, thing); break;
          case PCX_PROG_EM_DOWNBOLT: prog_em_downbolt(x, thing, draws, genv); break;
          case PCX_PROG_BS_PLAYER: prog_bs_player(x, thing); break;
          case PCX_PROG_BS_PATROLLER: prog_bs_patroller(x, thing); break;
          case PCX_PROG_BS_CASH: prog_bs_cash(x, thing); break;
          case PCX_PROG_OD_PLAYER: prog_od_player(x, thing); break;
          case PCX_PROG_OD_DRAGONDUCK: prog_od_dragonduck(x, thing); break;
          case PCX_PROG_OD_SWORD: prog_od_sword(x, thing); break;
          case PCX_PROG_WALKER: prog_walker(x, thing); break;
          case PCX_PROG_SCROLLY: prog_scrolly(x, thing); break;
          default: break;  // PCX_PROG_STATIC
        }
        if (x.mv_post) {  // the program's MazeWalker._move (sprites.py:356-389), the kernel's one copy
          mw_move(x, thing, x.mv_dr, x.mv_dc);
          after_move(x, thing);
          x.mv_post = 0;
        }
        if (k.n_sgroups > 1) {
          const uint32_t ov = (uint32_t)(x.order_valid & 1) | ((uint32_t)(x.o0 + 1) & 3u) << 1 | ((uint32_t)(x.o1 + 1) & 3u) << 3;
          x.orders = (x.orders & ~(0xFFu << (8 * sgroup))) | (ov << (8 * sgroup));
        }
        if (timing) {
          const unsigned long long dt = __builtin_readcyclecounter() - tp0;
          const int slot = (int)(tfield(x, thing, T_PROG) & 7);
#pragma unroll
          for (int b2 = 0; b2 < 8; ++b2) c_prog[b2] += slot == b2 ? dt : 0ull;
        }
      }
      // engine.py:735: a repaint after every update group (the last one is the render phase)
      if (i + 1 < k.NT && (sched_at(i + 1) >> 8) != (sched_at(i) >> 8)) snapshot(x);
    }
    if (timing) c_sec[1] = __builtin_readcyclecounter() - t_play;  // update groups
    const unsigned long long t_wb = timing ? __builtin_readcyclecounter() : 0ull;
    // ---- _apply_and_clear_plot + state write-back ---------------------------
    if (k.zdyn) {  // engine.py:796-835; the repaint it asks for (:632-637) is the render phase below
      apply_z_updates(x);
      uint32_t z0 = 0, z1 = 0;
      for (int z = 0; z < k.NT; ++z) {
        const uint32_t t = l.zord[z * WAVE + lane] & 0xFu;
        if (z < 8) z0 |= t << (4 * z); else z1 |= t << (4 * (z - 8));
      }
      st[k.w_z * srow] = z0;  // (before WB is declared: two words of the games that change the z-order)
      st[(k.w_z + 1) * srow] = z1;
    }
    flags = (x.game_over ? F_OVER : 0u) | ((x.err & 7u) << F_ERR_SHIFT) | ((uint32_t)((dxv + 1) & 0xFF) << 8);
#if defined(PCX_GENERIC_SPEC) && defined(PCX_X_WB4)
    // TIMING EXPERIMENT ONLY (results are wrong: the state never advances): the state words leave in 16-byte stores, four
    // words per lane and instruction, into the second half of a doubled allocation -- what a quad-interleaved state layout would cost
    RegList<uint32_t, 4> wbq{};
    int wbi = 0;
    uint32_t* const wb_base = P.state + (size_t)k.NW * bp + ((size_t)(env0 >> 6) * (size_t)((k.NW + 3) / 4 + 1)) * 256 + lane * 4;
    auto WB = [&](int, uint32_t v) {
      reg_put(wbq, wbi & 3, v);
      if ((wbi & 3) == 3) *reinterpret_cast<uint4*>(wb_base + (size_t)(wbi >> 2) * 256) = make_uint4(wbq.head, wbq.tail.head, wbq.tail.tail.head, wbq.tail.tail.tail.head);
      ++wbi;
    };
#else
    auto WB = [&](int w, uint32_t v) { st[w * srow] = v; };
#endif
#ifdef PCX_X_WB_PRIO  // (experiment: the logic wave's row stores at raised issue priority against the streaming waves of its SIMD)
    __builtin_amdgcn_s_setprio(3);
#endif
    WB(W_RNG, draws);
    if (k.w_next >= 0) WB(k.w_next, (uint32_t)x.next);
    WB(W_FRAME, (uint32_t)x.frame);
    WB(W_FLAGS, flags);
    for (int j = 0; j < 4; ++j) WB((W_V0 + j), (uint32_t)x.v[j]);
    PCX_SPEC_UNROLL
    for (int s = 0; s < k.NS; ++s) WB((W_SPRITES + s), spos(x, s));
    PCX_SPEC_UNROLL
    for (int w = 0; w < (k.NS + 3) / 4; ++w) {
      uint32_t f = 0;
      PCX_SPEC_UNROLL
      for (int j = 0; j < 4; ++j) if (4 * w + j < k.NS) f |= (sflg(x, 4 * w + j) & 0xFF) << (8 * j);
      WB((k.w_sflags + w), f);
    }
    for (int i2 = 0; i2 < ndw; ++i2) WB((k.w_drapes + i2), l.cur[i2 * WAVE + lane]);
    if (k.has_scroll) {
      WB(k.w_scroll, x.registered);
      for (int d = 0; d < k.ND; ++d) WB((k.w_scroll + 1 + d), l.corner[d * WAVE + lane]);
      for (int s = 0; s < k.NS; ++s) {
        WB((k.w_scroll + 1 + k.ND + 2 * s), l.pmask[s * WAVE + lane]);
        WB((k.w_scroll + 2 + k.ND + 2 * s), l.pframe[s * WAVE + lane]);
      }
    }
    out.reward[env] = x.reward;
    out.reward_set[env] = (uint8_t)x.reward_set;
    out.discount[env] = x.discount;
    out.done[env] = (uint8_t)x.game_over;
    out.frame[env] = x.frame;
    out.error[env] = (uint8_t)x.err;
#ifdef PCX_X_WB_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif

    if (timing) c_sec[2] = __builtin_readcyclecounter() - t_wb;  // write-back
    const unsigned long long t_occ = timing ? __builtin_readcyclecounter() : 0ull;
    // ---- occlusion for the final repaint (engine.py:751-757) ----------------
    // curtains -> flat cell-bit vectors; a curtain loses the cells a curtain in
    // front of it also covers; a sprite is shown iff nothing in front covers
    // its cell, and a shown sprite takes its cell from every curtain.
    // flat word w of drape d of environment e: environment-major with an odd
    // pitch, so that this phase (lane == e, same w) and the render phase (same
    // e, consecutive w) both touch 32 different banks
    const int FW = k.FW, FWP = k.FW | 1, C = k.C;
#define GFLAT(d, w, e) ((((d) * WAVE) + (e)) * FWP + (w))
    for (int d = 0; d < k.ND; ++d) {
      // the rows' bits, concatenated: stream them through a 64-bit register so
      // that every flat word is written once (no read-modify-write of LDS)
      uint64_t acc = 0;
      int have = 0, wi = 0;
#ifndef PCX_X_OCCL_WORD_BY_WORD
      // (eight row words in flight, then their flat words: word by word every LDS read waited for itself behind the store before it)
      const uint32_t* const src = drape_rows(x, l.cur, d);
      const int nw = k.R * k.RW;
      int w = 0;  // which 32 columns of its row the word holds
      for (int i0 = 0; i0 < nw; i0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(i0 + j < nw ? i0 + j : i0) * WAVE];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (i0 + j >= nw) break;
          const int nb = C - 32 * w < 32 ? C - 32 * w : 32;
          uint32_t bits = v[j];
          if (nb < 32) bits &= (1u << nb) - 1u;
          acc |= (uint64_t)bits << have;
          have += nb;
          if (have >= 32) { l.flat[GFLAT(d, wi, lane)] = (uint32_t)acc; ++wi; acc >>= 32; have -= 32; }
          w = w + 1 == k.RW ? 0 : w + 1;
        }
      }
#else
      for (int r = 0; r < k.R; ++r)
        for (int w = 0; w < k.RW; ++w) {  // up to 32 columns of row r at a time
          const int nb = C - 32 * w < 32 ? C - 32 * w : 32;
          uint32_t bits = drape_rows(x, l.cur, d)[(size_t)(r * k.RW + w) * WAVE];
          if (nb < 32) bits &= (1u << nb) - 1u;
          acc |= (uint64_t)bits << have;
          have += nb;
          if (have >= 32) { l.flat[GFLAT(d, wi, lane)] = (uint32_t)acc; ++wi; acc >>= 32; have -= 32; }
        }
#endif
      for (; wi < FW; ++wi) { l.flat[GFLAT(d, wi, lane)] = (uint32_t)acc; acc = 0; }
    }
    if (timing) c_sec[4] = __builtin_readcyclecounter() - t_occ;  // flat vectors built
    if (a.export_curtains)
      for (int d = 0; d < k.ND; ++d)
        for (int w = 0; w < FW; ++w) P.curtains[(size_t)(d * FW + w) * bp + env] = l.flat[GFLAT(d, w, lane)];
    if (!k.occl)  // unoccluded layers are the raw masks (rendering.py:236-278)
      for (int d = 0; d < k.ND; ++d)
        for (int w = 0; w < FW; ++w) l.flatraw[GFLAT(d, w, lane)] = l.flat[GFLAT(d, w, lane)];
    for (int t = 0; t < k.NT; ++t) {
      if (tfield(x, t, T_KIND) != 1) continue;
      const uint32_t d = tfield(x, t, T_IDX), above = above_things(x, t);
      for (int u = 0; u < k.NT; ++u) {
        if (!((above >> u) & 1) || tfield(x, u, T_KIND) != 1) continue;
        const uint32_t du = tfield(x, u, T_IDX);
#ifndef PCX_X_OCCL_WORD_BY_WORD
        for (int w0 = 0; w0 < FW; w0 += 4) {  // (four words of both curtains in flight)
          uint32_t mine[4], theirs[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int w = w0 + j < FW ? w0 + j : w0;
            mine[j] = l.flat[GFLAT(d, w, lane)]; theirs[j] = l.flat[GFLAT(du, w, lane)];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) if (w0 + j < FW) l.flat[GFLAT(d, w0 + j, lane)] = mine[j] & ~theirs[j];
        }
#else
        for (int w = 0; w < FW; ++w) l.flat[GFLAT(d, w, lane)] &= ~l.flat[GFLAT(du, w, lane)];
#endif
      }
    }
    if (timing) c_sec[5] = __builtin_readcyclecounter() - t_occ;  // + curtains over curtains
#if defined(PCX_GENERIC_SPEC) && !defined(PCX_X_RESOLVE_FROM_LDS)
    // (every specialised build, wherever the sprites' STATE lives: casts whose programs index sprites by a lane's own values -- the
    // marauders' -- keep that in LDS, but this loop indexes by constants only: marauders_custom_A 32,768 environments 0.0493 -> 0.0468 ms)
    // every sprite's cell once, in registers; who else is at the sprite's cell as masks over sprite and drape indices against
    // the masks of what is in front of it (in any order: a shown sprite only takes its cell from curtains that are behind it or
    // do not hold it); only the curtains' words come from LDS
    RegList<int, SREG_N> rcell{};
    PCX_SPEC_UNROLL
    for (int s = 0; s < k.NS; ++s) reg_put(rcell, s, sprite_cell(x, s));  // (never indexed by anything but a constant: see reg_get)
    if (timing) c_sec[7] = __builtin_readcyclecounter() - t_occ;  // + sprite cells
    PCX_SPEC_UNROLL
    for (int s = 0; s < k.NS; ++s) {
      const int t = (int)l.s2t[s];
      const int cell = reg_get(rcell, s);
      const uint32_t ab_s = above_sprites(x, t), ab_d = above_drapes(x, t);
      const bool vis = cell >= 0;
      const int cc = vis ? cell : 0, wi = cc >> 5, sh = cc & 31;
      uint32_t here_s = 0, here_d = 0;
      PCX_SPEC_UNROLL
      for (int j = 0; j < k.NS; ++j) here_s |= reg_get(rcell, j) == cell ? 1u << j : 0u;
      PCX_SPEC_UNROLL
      for (int d = 0; d < k.ND; ++d) here_d |= ((l.flat[GFLAT(d, wi, lane)] >> sh) & 1u) ? 1u << d : 0u;
      const bool shown = vis && !(here_s & ab_s) && !(here_d & ab_d);
      if (shown && here_d)
        for (int d = 0; d < k.ND; ++d) l.flat[GFLAT(d, wi, lane)] &= ~(1u << sh);
      l.sdesc[s * WAVE + lane] = make_uint2((uint32_t)cell, shown ? 1u : 0u);  // (what the owner codes and the descriptors below read)
    }
    if (timing) c_sec[8] = __builtin_readcyclecounter() - t_occ;  // + sprites resolved
#else
    // every sprite's cell once (sdesc doubles as the scratch: x = cell, y = shown)
    for (int s = 0; s < k.NS; ++s) l.sdesc[s * WAVE + lane] = make_uint2((uint32_t)sprite_cell(x, s), 0u);
    if (timing) c_sec[7] = __builtin_readcyclecounter() - t_occ;  // + sprite cells
    // In any order (a shown sprite only takes its cell from curtains that are behind it or do not
    // hold it): who else is at the sprite's cell, as masks over sprite and drape indices -- four
    // independent LDS reads at a time -- against the masks of what is in front of it.
    for (int s = 0; s < k.NS; ++s) {
      const int t = (int)l.s2t[s];
      const int cell = (int)l.sdesc[s * WAVE + lane].x;
      const uint32_t ab_s = above_sprites(x, t), ab_d = above_drapes(x, t);
      const bool vis = cell >= 0;
      const int cc = vis ? cell : 0, wi = cc >> 5, sh = cc & 31;
      uint32_t here_s = 0, here_d = 0;
#pragma unroll 2
      for (int j = 0; j < k.NS; ++j) here_s |= (int)l.sdesc[j * WAVE + lane].x == cell ? 1u << j : 0u;
      for (int d = 0; d < k.ND; ++d) here_d |= ((l.flat[GFLAT(d, wi, lane)] >> sh) & 1u) ? 1u << d : 0u;
      const bool shown = vis && !(here_s & ab_s) && !(here_d & ab_d);
      if (shown && here_d)
        for (int d = 0; d < k.ND; ++d) l.flat[GFLAT(d, wi, lane)] &= ~(1u << sh);
      l.sdesc[s * WAVE + lane].y = shown ? 1u : 0u;
    }
    if (timing) c_sec[8] = __builtin_readcyclecounter() - t_occ;  // + sprites resolved
#endif
    // the tracked positions (croppers) and, for the mask-composing render loop, the sprites' paint descriptors -- BEFORE the owner
    // codes are written: in pcx_generic_step they lie over the per-lane arrays this still reads (Consts::l_codes)
    const bool to_codes = l.codes != nullptr;
    for (int s = 0; s < k.NS; ++s) {
      if (!to_codes) {
        const uint2 cs = l.sdesc[s * WAVE + lane];
        const int cell = (int)cs.x;
        l.sdesc[s * WAVE + lane] = make_uint2(cs.y ? (uint32_t)(cell >> 2) : 0xFFFFFFFFu, 0xFFu << ((cell & 3) * 8));
        if (!k.occl) l.sdescraw[s * WAVE + lane] = make_uint2(cell >= 0 ? (uint32_t)(cell >> 2) : 0xFFFFFFFFu, 0xFFu << ((cell & 3) * 8));
      }
      int tr, tc;
      sprite_true(x, s, tr, tc);
      P.track[s * bp + env] = tr | (tc << 8) | ((int)(sflg(x, s) & 1) << 16) | ((int)do_reset << 24);
    }
    if (fc) move_windows(x, fc, env, l.wcorner);  // fused croppers follow this step's things
    if (l.codes != nullptr) {
      // Owner codes (rendering.py:98-179 as one byte per cell): the backdrop's code dwords, every curtain's cells (disjoint by
      // now: a curtain lost what a curtain in front covers) merged in four at a time, then the painted sprites as byte writes.
      slot_free();
      const int CP = k.QW | 1;
      uint32_t* const cd = l.codes + lane * CP;
#ifndef PCX_X_OCCL_WORD_BY_WORD
      // eight code dwords -- the 32 cells of ONE flat word per curtain -- at a time: their backdrop dwords in flight together, every
      // curtain's word read once (dword by dword it was read eight times, each read waiting behind the store of the dword before)
      for (int q0 = 0; q0 < k.QW; q0 += 8) {
        uint32_t dw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dw[j] = l.bdcode[q0 + j < k.QW ? q0 + j : q0];
        PCX_SPEC_UNROLL
        for (int t = 0; t < k.NT; ++t) {
          if (tfield(x, t, T_KIND) != 1) continue;
          const uint32_t word = l.flat[GFLAT(tfield(x, t, T_IDX), q0 >> 3, lane)];
          const uint32_t cbytes = code_of_layer(k.L, tfield(x, t, T_LAYER)) * 0x01010101u;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const uint32_t bits = (word >> (j * 4)) & 0xFu;
            const uint32_t m01 = (bits * 0x00204081u) & 0x01010101u;
            uint32_t hi8 = m01 << 8;
            asm("" : "+v"(hi8));  // (keeps (x << 8) - x from becoming a quarter-rate multiply)
            const uint32_t m = hi8 - m01;
            dw[j] = (dw[j] & ~m) | (cbytes & m);
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (q0 + j < k.QW) cd[q0 + j] = dw[j];
      }
#else
      for (int q = 0; q < k.QW; ++q) {
        uint32_t d = l.bdcode[q];
        PCX_SPEC_UNROLL
        for (int t = 0; t < k.NT; ++t) {
          if (tfield(x, t, T_KIND) != 1) continue;
          const uint32_t bits = (l.flat[GFLAT(tfield(x, t, T_IDX), q >> 3, lane)] >> ((q & 7) * 4)) & 0xFu;
          const uint32_t m01 = (bits * 0x00204081u) & 0x01010101u;
          uint32_t hi8 = m01 << 8;
          asm("" : "+v"(hi8));  // (keeps (x << 8) - x from becoming a quarter-rate multiply)
          const uint32_t m = hi8 - m01;
          d = (d & ~m) | ((code_of_layer(k.L, tfield(x, t, T_LAYER)) * 0x01010101u) & m);
        }
        cd[q] = d;
      }
#endif
      uint8_t* const cb = reinterpret_cast<uint8_t*>(cd);
      for (int s = 0; s < k.NS; ++s) {
        const uint2 cs = l.sdesc[s * WAVE + lane];
        if (cs.y) cb[cs.x] = (uint8_t)code_of_layer(k.L, tfield(x, (int)l.s2t[s], T_LAYER));
      }
    }
    if (timing) c_sec[3] = __builtin_readcyclecounter() - t_occ;  // occlusion + descriptors
  }
  if (timing && logic_wave && lane == 0) {
    atomicAdd(P.stats + 67, __builtin_readcyclecounter() - t_start);  // whole logic phase
    atomicAdd(P.stats + 69, 1ull);
    atomicAdd(P.stats + 64, c_sec[0]); atomicAdd(P.stats + 65, c_sec[1]);
    atomicAdd(P.stats + 66, c_sec[2]); atomicAdd(P.stats + 68, c_sec[3]);
    atomicAdd(P.stats + 70, c_sec[4]); atomicAdd(P.stats + 71, c_sec[5]);
    atomicAdd(P.stats + 72, c_sec[7]); atomicAdd(P.stats + 73, c_sec[8]);
    for (int b2 = 0; b2 < 8; ++b2) atomicAdd(P.stats + b2, c_prog[b2]);
  }
  slot_free();
  if (logic_wave) l.skip[lane] = skip;
#undef GFLAT
}

// The render worker's loop: pcx_stream.h stream_codes for one wave, with the number of characters a run-time value (a
// constant in the build specialised for the template).  Every store is `scalar plane base + one shared 32-bit lane offset`.
// `wave` / `nwaves`: the waves of a pcx_generic_step workgroup take the iterations round-robin (a render worker: 0 / 1).
__device__ __forceinline__ void render_codes(const Consts& k, const uint32_t* codes, const uint32_t* skip, const pcx_buffers& out, int64_t env0,
                                             int lane, int wave = 0, int nwaves = 1) {
  const uint32_t QW = (uint32_t)k.QW, CP = QW | 1u, pitch = (uint32_t)k.pitch;
  const int Lc = k.L;
  const uint32_t env_stride = (uint32_t)(1 + Lc) * pitch;
  uint8_t* const blk = stream::uniform_ptr(out.planes + (size_t)env0 * env_stride);
  uint32_t ch[4] = {0, 0, 0, 0};
  PCX_SPEC_UNROLL
  for (int i = 0; i < Lc && i < 16; ++i) ch[i >> 2] |= (uint32_t)k.chars[i] << (8 * (i & 3));
#pragma unroll
  for (int i = 0; i < 4; ++i) ch[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch[i]);
  const bool any_skip = __ballot(skip[lane] != 0) != 0ull;
  const uint32_t DE = (uint32_t)WAVE / QW, DQ = (uint32_t)WAVE - DE * QW;
  uint32_t e = (uint32_t)lane / QW, q = (uint32_t)lane - e * QW;
  uint32_t voff = e * env_stride + 4u * q, ci = e * CP + q;
  const uint32_t dvoff = DE * env_stride + 4u * DQ, dci = DE * CP + DQ;
  const uint32_t wrap_voff = env_stride - 4u * QW, wrap_ci = CP - QW;
  const uint32_t ci_last = (uint32_t)(WAVE - 1) * CP + QW - 1u;
  uint32_t code_pf = codes[ci];
#pragma unroll 1
  for (uint32_t it = 0; it < QW; ++it) {
    const uint32_t e_now = e, voff_now = voff, code = code_pf;
    q += DQ; e += DE; voff += dvoff; ci += dci;
    {
      const bool wrap = q >= QW;
      q = wrap ? q - QW : q;
      e = wrap ? e + 1 : e;
      voff = wrap ? voff + wrap_voff : voff;
      ci = wrap ? ci + wrap_ci : ci;
    }
    code_pf = codes[ci < ci_last ? ci : ci_last];
    if (nwaves > 1 && (int)(it % (uint32_t)nwaves) != wave) continue;
    if (any_skip && skip[e_now] != 0) continue;
    if (Lc <= 8) {
      saddr_store_dword<true>(voff_now, __builtin_amdgcn_perm(ch[1], ch[0], code), blk);
      PCX_SPEC_UNROLL
      for (int i = 0; i < Lc; ++i) {
        const uint32_t one = 1u << (8 * (i & 3));
        saddr_store_dword<true>(voff_now, __builtin_amdgcn_perm(i >= 4 ? one : 0u, i < 4 ? one : 0u, code), blk + (size_t)(1 + i) * pitch);
      }
    } else {
      const uint32_t sa = code & 0x0F0F0F0Fu, sb = (code >> 4) & 0x0F0F0F0Fu;
      saddr_store_dword<true>(voff_now, __builtin_amdgcn_perm(ch[1], ch[0], sa) | __builtin_amdgcn_perm(ch[3], ch[2], sb), blk);
      PCX_SPEC_UNROLL
      for (int i = 0; i < Lc; ++i) {
        const int j = i & 7;
        const uint32_t one = 1u << (8 * (j & 3));
        saddr_store_dword<true>(voff_now, __builtin_amdgcn_perm(j >= 4 ? one : 0u, j < 4 ? one : 0u, i < 8 ? sa : sb), blk + (size_t)(1 + i) * pitch);
      }
    }
  }
}

#ifdef __HIPCC_RTC__
extern "C"  // (the run-time build is looked up by this plain name)
#endif
#ifndef PCX_X_WPE
#define PCX_X_WPE 5  // (at least five waves per SIMD: at most 96 VGPRs)
#endif
__global__ __launch_bounds__(8 * WAVE) __attribute__((amdgpu_waves_per_eu(PCX_X_WPE, 8))) void pcx_generic_step(const Consts k_arg, const Ptrs P, const StepArgs a,
                                                         const pcx_buffers out, const crop::FusedCrops* fc) {
  extern __shared__ uint32_t lds[];
#ifdef PCX_GENERIC_SPEC
  const Consts& k = spec::K;
#else
  const Consts& k = k_arg;
#endif
  // A workgroup is 1, 2, 4 or 8 waves around one group of 64 environments: wave 0
  // steps them (lane == environment), then all waves share the render loop.
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  // The group's state words (SoA over the batch: word w of 64 consecutive environments is one 256-byte row) travel
  // straight into the per-lane LDS columns the logic phase keeps them in -- sprite positions, curtain rows, scrolling
  // protocol words -- and the scalars into an inbox, all by LDS-DMA, issued back to back in front of the staging of
  // the tables: ONE memory round trip instead of a dozen (a copy loop through registers has four loads in flight).
  if (wave == 0) dma_state_rows(k, P, a, lds, 0, (int64_t)blockIdx.x, lane);
  for (int i = threadIdx.x; i < P.n_table_words; i += blockDim.x) lds[i] = P.tables[i];
  if (wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows have landed
  __syncthreads();

  const bool timing = (a.debug & 8) != 0 && P.stats != nullptr;
  auto render_all = [&](const L& lr, int64_t env0r, int w, int nw) {
    const bool any_skip = __ballot(lr.skip[lane] != 0) != 0ull;
    if (fc) {
      render_windows(k, lr, fc, lr.wcorner, env0r, lane, w, nw);
      if (fc->only) return;  // the consumer ingests the windows only: no full-board planes
    }
    switch ((k.NT + 3) / 4) {
      case 0: case 1: render_planes<4>(k, lr, P, out, env0r, lane, w, nw, any_skip); break;
      case 2: render_planes<8>(k, lr, P, out, env0r, lane, w, nw, any_skip); break;
      case 3: render_planes<12>(k, lr, P, out, env0r, lane, w, nw, any_skip); break;
      default: render_planes<16>(k, lr, P, out, env0r, lane, w, nw, any_skip); break;
    }
  };
  const int64_t env0 = (int64_t)blockIdx.x * WAVE;
  // Round 6: plain steps render from OWNER CODES (P.use_codes; pcx_stream.h stream_codes: a byte per cell, one v_perm_b32 per plane)
  // -- a wave streams a group in 3-4 us where the mask-composing render_planes, bound by LDS latency, takes ten times that, and
  // a group holds its LDS for that much less time.  The code dwords lie over the per-lane arrays that are dead by then.
  L lm = make_L(lds, k, 0);
  if (P.use_codes && k.l_codes >= 0) lm.codes = lds + k.l_codes;
  const L l = lm;
  logic_phase(k, l, P, a, out, fc, lds + k.l_inbox, env0, lane, wave == 0, timing);
  __syncthreads();
  if (!(a.debug & 2)) {  // every wave of the workgroup streams board + layers
    if (l.codes != nullptr) render_codes(k, l.codes, l.skip, out, env0, lane, wave, nwaves);
    else render_all(l, env0, wave, nwaves);
  }
}

// (a run-time build holds this second kernel only when the engine asks for it -- PCX_GENERIC_PW=1 while it is created adds
// -DPCX_GENERIC_WITH_PW: the shape is opt-in, and leaving it out halves hiprtc's time per template)
#if !defined(PCX_GENERIC_SPEC) || defined(PCX_GENERIC_WITH_PW)
// ---------------------------------------------------------------------------
// Persistent workers with the logic phase DECOUPLED from the render phase (round 6).
//
// What bounds pcx_generic_step (profiles/r06_generic.md): a launch takes  units_per_CU x T / n_resident + the store floor,
// where T is one unit's logic phase -- a serial chain of LDS round trips, 20-35 us -- and n_resident is how many groups'
// per-lane arrays fit the CU's LDS (9-16).  A group holds its 16 KB of columns while it renders, and its render waves hold
// their wave slots while it steps.  Here a workgroup stays on its CU: its first `n_logic` waves are LOGIC WORKERS -- each
// owns one copy of the per-lane arrays, draws work units of 64 environments (pcx_stream.h WorkQueue) and steps one after
// the other without ever streaming -- and the next `n_render` waves are RENDER WORKERS that do nothing else.
// The hand-over is OWNER CODES (pcx_stream.h): the logic worker ends a unit by leaving one code byte per board cell in its
// hand-over slot ([64][QW | 1] dwords + the skip flags), and the render worker streams the slot with one LDS read and one
// v_perm_b32 per plane -- a loop light enough for TWO render waves to saturate the CU's store path (the mask-composing
// render_planes needs ~18).  One slot per logic worker is enough: it is written in the last microseconds of a logic phase
// of tens, so the worker waits for the slot -- one LDS flag word, 0 empty / unit + 1 full / DONE, one writer and one reader
// at a time, polled -- only then, long after the render worker has streamed the unit before.  Results never depend on the
// schedule: a unit is stepped by exactly one logic worker and streamed exactly once.
// Plain step launches of occluded games with at most 16 characters (no fused croppers, no feature epilogue, no reset
// launches: pcx_generic_step keeps those).
struct PwArgs {
  stream::WorkArgs work;      // the logic workers' scheduler
  int32_t n_logic, n_render;  // waves [0, n_logic) step, waves [n_logic, n_logic + n_render) stream
  int32_t lw_words;           // logic worker i's per-lane arrays lie i * lw_words behind the one-group layout's
  int32_t slot0, slot_words;  // hand-over slot i starts at word slot0 + i * slot_words: codes [64][QW | 1], then skip [64]
  int32_t flags;              // word offset of the n_logic flag words
};
constexpr uint32_t PW_DONE = 0xFFFFFFFFu;

#ifdef __HIPCC_RTC__
extern "C"
#endif
__global__ __launch_bounds__(16 * WAVE) void pcx_generic_step_pw(const Consts k_arg, const Ptrs P, const StepArgs a, const pcx_buffers out,
                                                                const PwArgs w) {
  extern __shared__ uint32_t lds[];
#ifdef PCX_GENERIC_SPEC
  const Consts& k = spec::K;
#else
  const Consts& k = k_arg;
#endif
  const int lane = threadIdx.x & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int NLW = w.n_logic, NRW = w.n_render;
  for (int i = threadIdx.x; i < P.n_table_words; i += blockDim.x) lds[i] = P.tables[i];
  if ((int)threadIdx.x < NLW) lds[w.flags + threadIdx.x] = 0u;
  __syncthreads();
  const bool timing = (a.debug & 8) != 0 && P.stats != nullptr;
  auto flag_load = [&](int i) {
    const uint32_t v = __hip_atomic_load(lds + w.flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
  };
  // Everything handed over lives in LDS, and a wave's LDS operations complete in order: `s_waitcnt lgkmcnt(0)` is all the
  // ordering the hand-over needs.  (A workgroup-scope release FENCE also waits for vmcnt(0) -- the render worker would drain
  // its plane stores after every unit, the logic worker its write-back: measured, 40 us of waiting per unit.)
  auto lds_order = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  auto flag_store = [&](int i, uint32_t v) {  // (after everything this wave wrote to / read from the slot)
    lds_order();
    __hip_atomic_store(lds + w.flags + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (every lane the same word)
  };
  const int CP = k.QW | 1;
  if (wave < NLW) {
    // ---- a logic worker ----------------------------------------------------------
    const int off = wave * w.lw_words;
    L l = make_L(lds, k, off);
    uint32_t* const slot = lds + w.slot0 + wave * w.slot_words;
    l.codes = slot;
    l.skip = slot + WAVE * CP;
    stream::WorkQueue wq;
    wq.init(w.work, wave, NLW);
    auto slot_free = [&]() {  // the render worker has streamed the unit before
      if (a.debug & 32) return;  // (timing experiment: no waiting -- the observation is then anybody's guess)
      uint32_t spins = 0;
      while (flag_load(wave) != 0u && ++spins < stream::SLOT_SPINS) __builtin_amdgcn_s_sleep(2);
      lds_order();
    };
    for (uint32_t unit = wq.first(); unit < wq.n; unit = wq.next(unit)) {
      dma_state_rows(k, P, a, lds, off, (int64_t)unit, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows have landed (and this worker's write-back of the unit before)
      logic_phase(k, l, P, a, out, nullptr, lds + k.l_inbox + off, (int64_t)unit * WAVE, lane, true, timing, slot_free);
      flag_store(wave, unit + 1u);
    }
    slot_free();
    flag_store(wave, PW_DONE);
    wq.finish(lane);
  } else if (wave < NLW + NRW) {
    // ---- a render worker: the slots of logic workers j, j + n_render, ... as they fill them ----------
    const int j = wave - NLW;
    uint32_t done = 0;
    int remaining = 0;
    for (int i = j; i < NLW; i += NRW) ++remaining;
    uint32_t idle = 0;
    while (remaining > 0 && idle < stream::SLOT_SPINS) {
      bool any = false;
      for (int i = j; i < NLW; i += NRW) {
        if ((done >> i) & 1u) continue;
        const uint32_t v = flag_load(i);
        if (v == 0u) continue;
        if (v == PW_DONE) { done |= 1u << i; --remaining; continue; }
        any = true;
        lds_order();
        const uint32_t* const slot = lds + w.slot0 + i * w.slot_words;
        if (!(a.debug & 2)) render_codes(k, slot, slot + WAVE * CP, out, (int64_t)(v - 1u) * WAVE, lane);
        flag_store(i, 0u);
      }
      if (any) idle = 0; else { ++idle; __builtin_amdgcn_s_sleep(2); }
    }
  }
}

#endif  // pcx_generic_step_pw

}  // namespace gen
}  // namespace pcx
