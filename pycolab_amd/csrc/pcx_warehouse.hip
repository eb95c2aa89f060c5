// pcx_warehouse.hip -- hand-written fused step kernel for the warehouse_manager
// game family (reference: pycolab/examples/warehouse_manager.py:181-295 driven
// by engine.py:583-847 and prefab_parts/sprites.py MazeWalker).  gfx950 only.
//
// One launch = one Engine.play() of every environment of the batch.  Same shape
// as pcx_scrolly_maze.hip: a group of 64 consecutive
// environments per workgroup; logic phase lane == environment with the whole
// per-environment state in registers (a dozen SoA words: no table is
// interpreted, the thing tables are compile-time); render phase = the shared
// streaming loop of pcx_stream.h.
//
// What makes the game cheap to step by hand:
//   * the update schedule is [boxes] [X] [P] and every entity of a group sees
//     the repaint that preceded the group (engine.py:735), so the three
//     repaints never have to exist: a MazeWalker probe (sprites.py:479-546) is
//     "is any thing painted at the target cell, or is the backdrop character
//     there impassable" -- the init code checks that every thing's character
//     is in every walker's impassable set, as in the shipped game -- evaluated
//     from the register snapshot of the sprites' cells and two bit-row tables;
//   * JudgeDrape's curtain (warehouse_manager.py:245-266) is recomputed from
//     the boxes' positions every frame: it is no state at all, and the curtain
//     of the previous frame (what group 0 sees) is the same function of the
//     positions the step starts with.
// Templates this kernel does not cover (other shapes than the instances below,
// occlusion_in_layers=False, unusual impassable sets or z-orders) are stepped
// by the table-driven kernel (pcx_generic.hip); the engine falls back to it.

#include "pcx_internal.h"
#include "pcx_stream.h"

#include <cstdlib>
#include <cstring>

namespace pcx {
namespace wm {

using stream::WAVE;
constexpr int MAX_NS = 11;  // ten boxes and the player
constexpr int MAX_NB = 8;   // characters only the backdrop paints

// State words (uint32 [NW][batch_padded]).
enum : int { W_FRAME = 0, W_FLAGS, W_SFLAGS, W_POS };
constexpr uint32_t F_OVER = 1u, F_ERR_SHIFT = 1;
constexpr int F_LAST_SHIFT = 8;  // JudgeDrape._last_num_boxes_on_goals, 8 bits

struct Consts {
  int32_t n_actions;
  int32_t rows, cols;        // the board (read by the run-time-shape instances)
  uint32_t confined;         // bit s: sprite s is confined to the board
  uint32_t above[MAX_NS];    // bit j: sprite j is in front of sprite s; bit NS: the judge drape is
  uint32_t init[W_POS + MAX_NS];
  uint32_t sprite_off[MAX_NS], sprite_ch4[MAX_NS], drape_off, drape_ch4, bchar_off[MAX_NB], bchar_ch4[MAX_NB];
  // owner codes (pcx_stream.h stream_codes; the CODES instances): the code byte of every thing's character, the characters by code
  uint32_t sprite_code[MAX_NS], drape_code, code_chars[4];
  int32_t tmpl_index[MAX_NS];  // sprite s here is sprite tmpl_index[s] of the template
};

struct Ptrs {
  const uint32_t* tables;  // staged into LDS: backdrop4 [QW], bdmask [NB][QW], goal rows [R], box-blocked rows [R], player-blocked rows [R], backdrop codes [QW]
  uint32_t* state;         // [NW][bpad]
  int32_t* track;          // [NS][bpad], template sprite order
  uint32_t* curtains;      // [1][FW][bpad] raw judge curtain (export_curtains)
  int64_t batch, bpad;
  stream::WorkArgs work;   // PW instances: the persistent workers' scheduler (pcx_stream.h)
};

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

// NS sprites: boxes 0..NS-2 in template order, the player is sprite NS-1 and
// the front-most thing.  R x C board, NB backdrop-only characters, NWAVES waves
// per workgroup (wave 0 steps the group, all of them share the render loop).
// SR x SC: the board's shape when the instance is compiled for it; 0 x 0: read from k.rows / k.cols
// (levels that are neither shipped nor among the fixtures' compiled shapes).
// PW (round 5): persistent workers -- the workgroup stays on its CU and each of its (up to eight) waves draws units of 64
// environments, steps one and streams it ALONE (NWAVES == 1), the next unit's state words prefetched into its LDS inbox, at
// most `work.lock` workers of the workgroup streaming at a time (pcx_stream.h; the launch shape pcx_scrolly_maze_step took
// in round 4).  Plain steps of the compiled shapes only: no epilogue, no fused croppers, occluded layers.
// CODES (round 6): the render phase is pcx_stream.h's owner-code loop -- the logic lane leaves a code byte per board cell
// (the backdrop's code dwords copied from a staged table, the judge's marks and the painted sprites as byte writes) instead
// of a curtain and sprite descriptors for the mask-composing loop.  Plain steps (no epilogue, no fused croppers, occluded layers).
template <int NS, int SR, int SC, int NB, int NWAVES, bool EPI = false, bool UNOCC = false, bool PW = false, bool CODES = false>
__global__ __launch_bounds__(PW ? 8 * WAVE : NWAVES* WAVE) void pcx_warehouse_step(const Consts k, const Ptrs P, const StepArgs a,
                                                                    const pcx_buffers out, const stream::EpilogueArgs epi,
                                                                    const crop::FusedCrops* fc) {
  extern __shared__ uint32_t lds[];
  const int R = SR ? SR : k.rows, C = SC ? SC : k.cols;  // (constants in the compiled-shape instances)
  constexpr int SQW = ((SR * SC + 3) & ~3) / 4;           // dwords per plane, 0: run-time shape
  const int cells = R * C, pitch = (cells + 3) & ~3, QW = pitch / 4, FW = (cells + 31) / 32, FWP = FW | 1;
  constexpr int L = NS + 1 + NB, IP = NS - 1, NBOX = NS - 1;
  const int CP = QW | 1;
  const int O_BD = 0, O_BDM = O_BD + QW, O_GOAL = O_BDM + NB * QW, O_BBLK = O_GOAL + R, O_PBLK = O_BBLK + R, O_BDC = O_PBLK + R,
            O_TAB_END = O_BDC + QW;
  // (CODES: the per-environment code dwords take the place of the sprite descriptors)
  const int O_FLAT = O_TAB_END, O_SDESC = (O_FLAT + WAVE * FWP + 1) & ~1, O_SKIP = O_SDESC + (CODES ? WAVE * CP : 2 * NS * WAVE);
  const int O_WCORNER = O_SKIP + WAVE;  // fused croppers' window corners
  const int O_FLATRAW = O_WCORNER + stream::WCORNER_WORDS, O_SDESCRAW = (O_FLATRAW + WAVE * FWP + 1) & ~1;  // UNOCC only
  static_assert(!PW || (NWAVES == 1 && !EPI && !UNOCC && SR != 0), "persistent workers: plain steps of the compiled shapes");
  static_assert(!CODES || (!EPI && !UNOCC), "owner codes: plain steps");
  // PW: a worker's own LDS region {flat, sdesc, skip, inbox}; the inbox holds the unit's state rows and its tape actions
  constexpr int NW = W_POS + NS, IB_ROWS = NW + 1;
  const int O_SEM = O_FLAT, O_W0 = O_FLAT + 2, W_WORDS = ((O_WCORNER - O_FLAT) + IB_ROWS * WAVE + 1) & ~1;
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
  const int mine = PW ? O_W0 - O_FLAT + __builtin_amdgcn_readfirstlane(wave) * W_WORDS : 0;  // (word offset of this worker's region)
  for (int i = threadIdx.x; i < O_TAB_END; i += (int)blockDim.x) lds[i] = P.tables[i];
  const uint32_t* const goal_rows = lds + O_GOAL;
  const uint32_t* const box_blocked = lds + O_BBLK;
  const uint32_t* const player_blocked = lds + O_PBLK;
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
  // the state rows of unit `u` (and its tape actions) into the inbox
  auto prefetch = [&](uint32_t u_any) {
    const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u_any);
    const int64_t e0 = (int64_t)u * WAVE;
    const uint32_t ib = (uint32_t)__builtin_amdgcn_readfirstlane((int)stream::lds_byte_address(inbox));
#pragma unroll
    for (int w = 0; w < NW; ++w) stream::lds_dma_row(P.state + (int64_t)w * P.bpad + e0, 4u * lane, ib + (uint32_t)w * (4u * WAVE));
    if (!a.hashed && e0 + lane < P.batch) stream::lds_dma_row(reinterpret_cast<const uint32_t*>(a.actions) + e0, 4u * lane, ib + (uint32_t)NW * (4u * WAVE));
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
    // ---- logic phase: lane == environment -------------------------------------
    const int64_t env = env0 + lane, bp = P.bpad;
    const bool live = env < P.batch;
    uint32_t* const st = P.state + env;  // word w at st[w * bpad]
    uint32_t flags = 0, ld_frame = 0, ld_sflags = 0, ld_pos[NS] = {};
    int ld_action = PCX_ACTION_NONE;
    bool skip = !live, do_reset = false;
    int action = PCX_ACTION_NONE;
    if constexpr (PW) {  // the unit's state rows are in the inbox (the first unit's must be waited for)
      if (need_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t* const ib = inbox + lane;
      flags = ib[W_FLAGS * WAVE]; ld_frame = ib[W_FRAME * WAVE]; ld_sflags = ib[W_SFLAGS * WAVE];
#pragma unroll
      for (int s = 0; s < NS; ++s) ld_pos[s] = ib[(W_POS + s) * WAVE];
      if (!a.hashed && live) ld_action = (int)ib[NW * WAVE];
    }
    if (live) {  // every state word is requested up front: one memory round trip
      if constexpr (!PW) flags = st[W_FLAGS * bp];
      if (!PW && a.mode != 1) {
        ld_frame = st[W_FRAME * bp];
        ld_sflags = st[W_SFLAGS * bp];
#pragma unroll
        for (int s = 0; s < NS; ++s) ld_pos[s] = st[(W_POS + s) * bp];
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
      int frame, last;
      uint32_t sflags, err;
      int vr[NS], vc[NS], vis[NS], prior[NS];
      if (do_reset) {  // engine.py:520-581 its_showtime: fresh template state, frame 0 = play(None)
        frame = (int)k.init[W_FRAME];
        last = (int)((k.init[W_FLAGS] >> F_LAST_SHIFT) & 0xFF);
        sflags = k.init[W_SFLAGS];
        err = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) { vr[s] = pos_r(k.init[W_POS + s]); vc[s] = pos_c(k.init[W_POS + s]); }
        action = PCX_ACTION_NONE;
      } else {
        frame = (int)ld_frame;
        last = (int)((flags >> F_LAST_SHIFT) & 0xFF);
        sflags = ld_sflags;
        err = (flags >> F_ERR_SHIFT) & 7u;
#pragma unroll
        for (int s = 0; s < NS; ++s) { vr[s] = pos_r(ld_pos[s]); vc[s] = pos_c(ld_pos[s]); }
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) { vis[s] = (sflags >> (2 * s)) & 1; prior[s] = (sflags >> (2 * s + 1)) & 1; }
      int reward = 0, reward_set = 0, over = 0;
      float discount = 1.0f;
      frame += 1;  // engine.py:698-735

      auto on_board = [&](int r, int c) { return (unsigned)r < (unsigned)R && (unsigned)c < (unsigned)C; };
      // Sprite.position: the virtual position while on the board, else (0, 0) (sprites.py:223-275)
      auto true_cell = [&](int r, int c) { return on_board(r, c) ? r * C + c : 0; };
      // sprites.py:315-352 _teleport with the exit/enter visibility bookkeeping
      auto teleport = [&](int& r, int& c, int& v, int& pv, int nr, int nc) {
        const bool old_on = on_board(r, c), new_on = on_board(nr, nc);
        if (old_on && !new_on) { pv = v; v = 0; }
        if (!old_on && new_on) v = pv;
        r = nr; c = nc;
      };
      const int dr = action == 0 ? -1 : action == 1 ? 1 : 0, dc = action == 2 ? -1 : action == 3 ? 1 : 0;
      const bool moving = (unsigned)action <= 3u;

      // ---- group 0: BoxSprite.update (warehouse_manager.py:214-226), every box
      // reading the repaint the step started with
      if (moving) {
        int cell0[NS], tcell0[NBOX];
#pragma unroll
        for (int s = 0; s < NS; ++s) cell0[s] = vis[s] ? true_cell(vr[s], vc[s]) : -1;  // engine.py:752-753
#pragma unroll
        for (int s = 0; s < NBOX; ++s) tcell0[s] = true_cell(vr[s], vc[s]);
#pragma unroll
        for (int s = 0; s < NBOX; ++s) {
          const bool on = on_board(vr[s], vc[s]);
          // layers['P'][rows - dr, cols - dc] with numpy's index rules
          int pr = (on ? vr[s] : 0) - dr, pc = (on ? vc[s] : 0) - dc;
          if (pr < 0) pr += R;
          if (pc < 0) pc += C;
          if (pr >= R || pc >= C) { err |= ERR_INDEX; continue; }
          if (cell0[IP] != pr * C + pc) continue;  // the player is the front-most thing: its layer is its cell
          // sprites.py:356-389 _move one step in a cardinal direction
          const int tr = vr[s] + dr, tc = vc[s] + dc;
          bool blocked;
          if (!on_board(tr, tc)) {
            blocked = (k.confined >> s) & 1;  // EDGE
          } else {
            const int tcell = tr * C + tc;
            bool thing = false, box_here = false;
#pragma unroll
            for (int j = 0; j < NS; ++j) if (j != s) thing |= cell0[j] == tcell;
#pragma unroll
            for (int j = 0; j < NBOX; ++j) box_here |= tcell0[j] == tcell;
            // the judge's curtain of the previous frame: boxes standing on goals
            thing |= box_here && ((goal_rows[tr] >> tc) & 1);
            blocked = thing || ((box_blocked[tr] >> tc) & 1);
          }
          if (!blocked) teleport(vr[s], vc[s], vis[s], prior[s], tr, tc);
        }
      }

      // ---- group 1: JudgeDrape.update (warehouse_manager.py:245-266) ----------
      int tcell1[NBOX];
      uint32_t on_goal_mask = 0;  // bit j: box j stands on a goal cell
      {
        int boxes = 0, on_goals = 0;
#pragma unroll
        for (int j = 0; j < NBOX; ++j) {
          const bool on = on_board(vr[j], vc[j]);
          const int r = on ? vr[j] : 0, c = on ? vc[j] : 0;
          tcell1[j] = r * C + c;
          bool dup = false;  // the curtain is a set of cells: np.sum counts a shared cell once
#pragma unroll
          for (int i = 0; i < j; ++i) dup |= tcell1[i] == tcell1[j];
          const bool g = (goal_rows[r] >> c) & 1;
          on_goal_mask |= (uint32_t)g << j;
          boxes += !dup;
          on_goals += !dup && g;
        }
        reward += on_goals - last;  // plot.py:200-226 add_reward (always called: reward is never None)
        reward_set = 1;
        last = on_goals;
        if (action == 5 || on_goals == boxes) { over = 1; discount = 0.0f; }  // plot.py:176-198
      }

      // ---- group 2: PlayerSprite.update (warehouse_manager.py:285-295) ---------
      if (moving) {
        const int tr = vr[IP] + dr, tc = vc[IP] + dc;
        bool blocked;
        if (!on_board(tr, tc)) {
          blocked = (k.confined >> IP) & 1;
        } else {
          const int tcell = tr * C + tc;
          bool thing = false;
#pragma unroll
          for (int j = 0; j < NBOX; ++j) {
            thing |= vis[j] && tcell1[j] == tcell;                     // the box itself
            thing |= ((on_goal_mask >> j) & 1) && tcell1[j] == tcell;  // the judge's mark on it
          }
          blocked = thing || ((player_blocked[tr] >> tc) & 1);
        }
        if (!blocked) teleport(vr[IP], vc[IP], vis[IP], prior[IP], tr, tc);
      }

      // ---- render descriptors: the judge's curtain, then occlusion ---------------
#pragma unroll
      for (int w = 0; w < FW; ++w) flat[lane * FWP + w] = 0;
#pragma unroll
      for (int j = 0; j < NBOX; ++j)
        if ((on_goal_mask >> j) & 1) flat[lane * FWP + (tcell1[j] >> 5)] |= 1u << (tcell1[j] & 31);
      if (a.export_curtains)
        for (int w = 0; w < FW; ++w) P.curtains[(size_t)w * bp + env] = flat[lane * FWP + w];
      int cellv[NS];
      uint32_t above[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) { cellv[s] = vis[s] ? true_cell(vr[s], vc[s]) : -1; above[s] = k.above[s]; }
      if constexpr (UNOCC)  // occlusion_in_layers=False: the layers are the raw masks (rendering.py:236-278)
        stream::snapshot_raw<NS, 1>(cellv, flat, FW, FWP, lane, lds + O_FLATRAW, reinterpret_cast<uint2*>(lds + O_SDESCRAW));
      if constexpr (CODES) {
        // rendering.py:98-179 as byte writes: the backdrop's codes, the judge's marks, then every sprite nothing in front of it covers
        uint32_t* const cd = codes + lane * CP;
        const uint32_t* const bdc = lds + O_BDC;
        if constexpr (SQW != 0) {
#pragma unroll
          for (int w = 0; w < SQW; ++w) cd[w] = bdc[w];
        } else {
          for (int w = 0; w < QW; ++w) cd[w] = bdc[w];
        }
        uint8_t* const cb = reinterpret_cast<uint8_t*>(cd);
#pragma unroll
        for (int j = 0; j < NBOX; ++j)
          if ((on_goal_mask >> j) & 1) cb[tcell1[j]] = (uint8_t)k.drape_code;
        uint32_t scode[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) scode[s] = k.sprite_code[s];
        stream::paint_sprites<NS, 1>(cellv, above, flat, FWP, lane, cb, scode);
      } else {
        stream::resolve_sprites<NS, 1>(cellv, above, flat, FWP, lane, sdesc);
      }

      // ---- _apply_and_clear_plot (engine.py:761-847) + state write-back ---------
      st[W_FRAME * bp] = (uint32_t)frame;
      st[W_FLAGS * bp] = (over ? F_OVER : 0u) | ((err & 7u) << F_ERR_SHIFT) | ((uint32_t)(last & 0xFF) << F_LAST_SHIFT);
      uint32_t sf = 0;
      int32_t tw[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        st[(W_POS + s) * bp] = pack_pos(vr[s], vc[s]);
        sf |= ((uint32_t)vis[s] | ((uint32_t)prior[s] << 1)) << (2 * s);
        const bool on = on_board(vr[s], vc[s]);
        tw[s] = (on ? vr[s] : 0) | ((on ? vc[s] : 0) << 8) | (vis[s] << 16) | ((int)do_reset << 24);
        P.track[(size_t)k.tmpl_index[s] * bp + env] = tw[s];
      }
      const stream::CurtainSrc csrc{P.curtains, bp, FW, R, C};
      if (fc)  // fused croppers: the windows follow this step's positions (cropping.py:393-426); a cropper may follow the judge's drape
        stream::move_fused_windows(fc, [&](int ti) {
          int32_t t = 0;
#pragma unroll
          for (int s = 0; s < NS; ++s) t = ti == (int)k.tmpl_index[s] ? tw[s] : t;
          return t;
        }, frame == 0, env, lane, wcorner, &csrc);
      st[W_SFLAGS * bp] = sf;
      out.reward[env] = reward;
      out.reward_set[env] = (uint8_t)reward_set;
      out.discount[env] = discount;
      out.done[env] = (uint8_t)over;
      out.frame[env] = frame;
      out.error[env] = (uint8_t)err;
    }
    skipv[lane] = skip;
  }
  if constexpr (!PW) {
    __syncthreads();
    if (a.debug & 2) return;
  }

  // ---- render phase --------------------------------------------------------------
  stream::PlaneMap<NS, 1, NB> pm;
#pragma unroll
  for (int s = 0; s < NS; ++s) { pm.sprite_off[s] = k.sprite_off[s]; pm.sprite_ch4[s] = k.sprite_ch4[s]; }
  pm.drape_off[0] = k.drape_off; pm.drape_ch4[0] = k.drape_ch4;
  uint32_t bch4[NB > 0 ? NB : 1] = {};
#pragma unroll
  for (int b = 0; b < NB; ++b) { pm.bchar_off[b] = k.bchar_off[b]; bch4[b] = k.bchar_ch4[b]; }
  const uint32_t env_stride = (uint32_t)(1 + L) * (uint32_t)pitch;
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
        stream::stream_codes<L, SQW, 1, false>(cmap, out.planes + (size_t)env0 * env_stride, env_stride, codes, CP, skipv, lane, 0, QW);
      else
        stream::stream_planes<NS, 1, NB, SQW, 1, false, false, false>(pm, out.planes + (size_t)env0 * env_stride, env_stride, lds + O_BD, lds + O_BDM,
                                                                      flat, sdesc, skipv, FWP, lane, 0, epi, env0, nullptr, QW, nullptr, nullptr, lds);
      if (P.work.lock) stream::slot_release(sem);
    }
    // fewer than 64 plane stores behind the prefetch (environments left alone, ablation runs): wait for it
    need_wait = any_skip || (a.debug & ~16) != 0 || QW * (1 + L) < 64;
    if (need_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unit = next;
  } else {
    if constexpr (CODES) {
      stream::stream_codes<L, SQW, NWAVES>(cmap, out.planes + (size_t)env0 * env_stride, env_stride, codes, CP, skipv, lane, wave, QW);
      break;
    }
    if (!(fc && fc->only))
      stream::stream_planes<NS, 1, NB, SQW, NWAVES, EPI, UNOCC>(pm, out.planes + (size_t)env0 * env_stride, env_stride, lds + O_BD, lds + O_BDM,
                                                          flat, sdesc, skipv, FWP, lane, wave, epi, env0, nullptr, QW, lds + O_FLATRAW,
                                                          reinterpret_cast<const uint2*>(lds + O_SDESCRAW), lds);
    if (fc)
      stream::stream_windows<NS, 1, NB, SQW, NWAVES, SR, SC>(fc, pm, bch4, env0, lds + O_BD, flat, sdesc, skipv, FWP, lane, wave, wcorner,
                                                           nullptr, stream::BoardShape{R, C, QW});
    break;
  }
  }  // units
  if constexpr (PW) wq.finish(lane);
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------

// The shapes with a compiled instance: sprites (boxes + player), rows, cols,
// backdrop-only characters.  The three shipped levels, and the two unshipped
// levels of the golden fixtures (oracle/custom_levels.py).
#define PCX_WM_SHAPES(X) X(6, 11, 10, 4) X(8, 11, 13, 4) X(10, 11, 13, 4) X(3, 7, 9, 4) X(11, 12, 18, 4)

class WarehouseBackend : public Backend {
 public:
  int init(const pcx_template& t, int64_t batch) override;
  int launch(const StepArgs& a, const pcx_buffers& out, hipStream_t s) override;
  int read_things(int64_t env0, int64_t n, pcx_sprite_state* sprites, uint8_t* curtains) override;
  int64_t bytes_per_step() const override {
    // read: action 4 + state 4 NW; write: state 4 NW + planes (1 + L) cells + results 15
    return 4 + 8 * (int64_t)NW_ + (int64_t)(1 + L_) * lay_.cells + 15;
  }
  int tuner_done() const override { return tuner_.done(); }
  const char* kernel_name() const override { return "pcx_warehouse_step"; }
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
      return set_error(PCX_E_UNSUPPORTED, "warehouse backend: fused croppers need occluded layers");
    return fused_.set(fc, false, R_, C_);
  }
  size_t base_lds_bytes(bool codes = false) const {  // the kernel's own dynamic LDS (before padding / the channels-last exchange areas)
    return ((size_t)lay_.QW * (2 + NB_) + 3 * R_ + WAVE * lay_.FWP + 2 + (codes ? WAVE * lay_.CP : 2 * NS_ * WAVE) + WAVE + stream::WCORNER_WORDS +
            (unoccluded_ ? WAVE * lay_.FWP + 2 + 2 * NS_ * WAVE : 0)) * 4;
  }
  stream::EpilogueArgs* epilogue_args() override { return &epi_; }
  int set_epilogue(const pcx_epilogue_desc* d) override {
    if (d && (!static_shape_ || unoccluded_)) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: the feature-array epilogue exists for the compiled shapes, occluded layers");
    if (!stream::fill_epilogue(epi_, d, lay_.cells, sprite_ch_, NS_, &drape_ch_, 1, bchar_ch_, NB_, 64 * 1024 - base_lds_bytes(), 4))
      return set_error(PCX_E_UNSUPPORTED, "warehouse backend: the channels-last epilogue needs rows*cols %% 4 == 0 and a stack whose exchange areas fit the LDS left");
    return 0;
  }

 private:
  stream::FusedCropsHolder fused_;
  Consts k_{};
  stream::EpilogueArgs epi_{};
  bool unoccluded_ = false;  // Engine(..., occlusion_in_layers=False): the run-time-shape instances' UNOCC variant
  int sprite_ch_[MAX_NS] = {}, drape_ch_ = 0, bchar_ch_[MAX_NB] = {};
  stream::Layout lay_;
  int NS_ = 0, R_ = 0, C_ = 0, NB_ = 0, L_ = 0, NW_ = 0;
  int64_t batch_ = 0, bpad_ = 0;
  int num_cus_ = 256;
  bool static_shape_ = false;  // a compiled instance for exactly this shape exists
  std::vector<uint8_t> goal_;  // host copy for read_things
  DevArray<uint32_t> tables_, state_, curtains_, work_ctr_;
  ShapeTuner tuner_;
  int last_shape_ = -1;
  DevArray<int32_t> track_;
};

int WarehouseBackend::init(const pcx_template& t, int64_t batch) {
  Consts& k = k_;
  batch_ = batch;
  bpad_ = (batch + WAVE - 1) / WAVE * WAVE;
  if (const char* e = getenv("PCX_FORCE_GENERIC")) if (atoi(e)) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: PCX_FORCE_GENERIC");
  // occlusion_in_layers=False: the game's rules read nothing an unoccluded layer changes (the boxes look at
  // layers['P'], and nothing is ever in front of P: checked below); only the layer planes differ
  unoccluded_ = !t.occlusion_in_layers;
  if (t.n_directives) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: plot directives");
  NS_ = t.n_sprites; R_ = t.rows; C_ = t.cols; L_ = t.n_chars;
  NB_ = L_ - NS_ - 1;
  static_shape_ = false;
#define X(ns, r, c, nb) static_shape_ |= NS_ == ns && R_ == r && C_ == c && NB_ == nb;
  PCX_WM_SHAPES(X)
#undef X
  // any other level takes a run-time-shape instance (one per number of sprites): rows of at most 32 cells
  // (goal / blocked tables are a word per row), the usual four backdrop-only characters
  const bool dynamic_ok = NS_ >= 2 && NS_ <= MAX_NS && NB_ == 4 && C_ <= 32 && R_ <= 255;
  if (t.n_drapes != 1 || !(static_shape_ || dynamic_ok) || NS_ > MAX_NS || NB_ > MAX_NB || (unoccluded_ && !dynamic_ok))
    return set_error(PCX_E_UNSUPPORTED, "warehouse backend: no instance for this shape");
  lay_.set(R_, C_);
  k.rows = R_; k.cols = C_;
  // sprites: boxes first, the player last (template order is kept: ascii_art.py:278-283)
  const int ip = NS_ - 1;
  for (int s = 0; s < NS_; ++s) {
    const pcx_sprite_desc& sd = t.sprites[s];
    if (!sd.is_walker || sd.egocentric || sd.program != (s == ip ? PCX_PROG_WM_PLAYER : PCX_PROG_WM_BOX))
      return set_error(PCX_E_UNSUPPORTED, "warehouse backend: sprites must be boxes followed by the player");
    k.tmpl_index[s] = s;
  }
  const pcx_drape_desc& dd = t.drapes[0];
  if (dd.program != PCX_PROG_WM_JUDGE || dd.is_scrolly) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: the drape must be the judge");
  for (int i = 0; i < R_ * C_; ++i)
    if (dd.curtain[i]) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: the judge's curtain must start empty");
  // update schedule [boxes] [judge] [player]
  if (t.n_groups != 3 || t.n_things != NS_ + 1) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: schedule must be [boxes][X][P]");
  for (int i = 0; i < t.n_things; ++i) {
    const int want_group = i < NS_ - 1 ? 0 : i == NS_ - 1 ? 1 : 2;
    const int want_ch = i < NS_ - 1 ? -1 : i == NS_ - 1 ? dd.ch : t.sprites[ip].ch;
    bool is_box = false;
    for (int s = 0; s < ip; ++s) is_box |= t.sprites[s].ch == t.schedule[i];
    if (t.group_of[i] != want_group || (want_ch >= 0 ? t.schedule[i] != want_ch : !is_box))
      return set_error(PCX_E_UNSUPPORTED, "warehouse backend: schedule must be [boxes][X][P]");
  }
  // every thing's character must be impassable to every other walker (then a probe is "any thing there?")
  auto imp_has = [&](int s, int ch) { return (t.sprites[s].impassable[ch >> 3] >> (ch & 7)) & 1; };
  for (int s = 0; s < NS_; ++s) {
    for (int j = 0; j < NS_; ++j)
      if (j != s && !imp_has(s, t.sprites[j].ch)) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: unusual impassable set");
    if (!imp_has(s, dd.ch)) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: unusual impassable set");
    for (int i = 0; i < R_ * C_; ++i)
      if (t.backdrop[i] >= 128) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: non-ASCII backdrop");
  }
  // z-order -> who is in front of whom; the player must be the front-most thing
  int zpos[MAX_NS + 1];
  for (int z = 0; z < t.n_things; ++z) {
    int idx = -1;
    for (int s = 0; s < NS_; ++s) if (t.sprites[s].ch == t.z_order[z]) idx = s;
    if (t.z_order[z] == dd.ch) idx = NS_;
    if (idx < 0) return set_error(PCX_E_INVALID, "warehouse backend: z_order names an unknown character");
    zpos[idx] = z;
  }
  if (zpos[ip] != t.n_things - 1) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: the player must be the front-most thing");
  for (int s = 0; s < NS_; ++s) {
    k.above[s] = 0;
    for (int j = 0; j <= NS_; ++j) if (zpos[j] > zpos[s]) k.above[s] |= 1u << j;
  }
  k.n_actions = t.n_actions;
  k.confined = 0;
  for (int s = 0; s < NS_; ++s) if (t.sprites[s].confined) k.confined |= 1u << s;
  auto layer_of = [&](int ch) { for (int i = 0; i < L_; ++i) if (t.chars[i] == ch) return i; return -1; };
  for (int s = 0; s < NS_; ++s) {
    k.sprite_off[s] = (uint32_t)(1 + layer_of(t.sprites[s].ch)) * lay_.pitch;
    k.sprite_ch4[s] = t.sprites[s].ch * 0x01010101u;
    sprite_ch_[s] = t.sprites[s].ch;
  }
  k.drape_off = (uint32_t)(1 + layer_of(dd.ch)) * lay_.pitch;
  k.drape_ch4 = dd.ch * 0x01010101u;
  drape_ch_ = dd.ch;
  stream::fill_epilogue(epi_, nullptr, lay_.cells, sprite_ch_, NS_, &drape_ch_, 1, bchar_ch_, NB_);

  // tables staged into LDS
  std::vector<uint32_t> tab((size_t)lay_.QW * (2 + NB_) + 3 * R_, 0);
  memcpy(tab.data(), t.backdrop, lay_.cells);
  int nb = 0;
  for (int i = 0; i < L_; ++i) {
    const int ch = t.chars[i];
    bool thing = ch == dd.ch;
    for (int s = 0; s < NS_; ++s) thing |= t.sprites[s].ch == ch;
    if (thing) continue;
    if (nb >= NB_) return set_error(PCX_E_INVALID, "warehouse backend: inconsistent character set");
    k.bchar_off[nb] = (uint32_t)(1 + i) * lay_.pitch;
    k.bchar_ch4[nb] = (uint32_t)ch * 0x01010101u;
    bchar_ch_[nb] = ch;
    uint8_t* m = reinterpret_cast<uint8_t*>(tab.data() + (size_t)lay_.QW * (1 + nb));
    for (int c = 0; c < lay_.cells; ++c) m[c] = t.backdrop[c] == ch;
    ++nb;
  }
  if (nb != NB_) return set_error(PCX_E_INVALID, "warehouse backend: inconsistent character set");
  uint32_t* goal = tab.data() + (size_t)lay_.QW * (1 + NB_);
  uint32_t* bblk = goal + R_;
  uint32_t* pblk = bblk + R_;
  goal_.assign(lay_.cells, 0);
  for (int r = 0; r < R_; ++r)
    for (int c = 0; c < C_; ++c) {
      const int ch = t.backdrop[r * C_ + c];
      if (ch == '_') { goal[r] |= 1u << c; goal_[r * C_ + c] = 1; }  // backdrop.palette._ (warehouse_manager.py:255)
      if (imp_has(0, ch)) bblk[r] |= 1u << c;
      if (imp_has(ip, ch)) pblk[r] |= 1u << c;
      for (int s = 1; s < ip; ++s)
        if (imp_has(s, ch) != imp_has(0, ch)) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: boxes must share their backdrop rules");
    }

  // owner codes (pcx_stream.h): a character's code is its place in the template's sorted list = its layer plane
  {
    auto code_of = [&](int ch) { const int i = layer_of(ch); return L_ <= 8 ? (uint32_t)i : i < 8 ? 0xC0u | (uint32_t)i : 0x0Cu | ((uint32_t)(i - 8) << 4); };
    if (L_ > 16) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: more than sixteen characters");
    for (int s = 0; s < NS_; ++s) k.sprite_code[s] = code_of(t.sprites[s].ch);
    k.drape_code = code_of(dd.ch);
    memset(k.code_chars, 0, sizeof k.code_chars);
    for (int i = 0; i < L_; ++i) k.code_chars[i >> 2] |= (uint32_t)t.chars[i] << (8 * (i & 3));
    uint8_t* bdc = reinterpret_cast<uint8_t*>(pblk + R_);
    for (int c = 0; c < lay_.cells; ++c) bdc[c] = (uint8_t)code_of(t.backdrop[c]);
    for (int c = lay_.cells; c < lay_.pitch; ++c) bdc[c] = (uint8_t)(L_ <= 8 ? 8 + 4 : 0xCC);  // plane padding: selector 12, zero bytes in every plane
  }

  // initial state words
  NW_ = W_POS + NS_;
  memset(k.init, 0, sizeof k.init);
  k.init[W_FRAME] = (uint32_t)-1;
  if (dd.param[0] < 0 || dd.param[0] > 255) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: _last_num_boxes_on_goals out of range");
  k.init[W_FLAGS] = (uint32_t)dd.param[0] << F_LAST_SHIFT;
  for (int s = 0; s < NS_; ++s) {
    const pcx_sprite_desc& sd = t.sprites[s];
    k.init[W_SFLAGS] |= ((uint32_t)(sd.visible != 0) | ((uint32_t)(sd.prior_visible != 0) << 1)) << (2 * s);
    k.init[W_POS + s] = ((uint32_t)sd.vrow & 0xFFFFu) | ((uint32_t)sd.vcol << 16);
  }
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      num_cus_ = prop.multiProcessorCount;
  }
  int rc;
  if ((rc = tables_.upload(tab))) return rc;
  if ((rc = state_.alloc((size_t)NW_ * bpad_))) return rc;
  if ((rc = track_.alloc((size_t)NS_ * bpad_))) return rc;
  // the persistent workers' ticket shards and done-count (pcx_stream.h WorkQueue), 64 bytes apart: allocated and zeroed HERE,
  // with the device synchronisation that follows engine creation -- a first launch on a non-blocking stream (or under
  // stream capture) must not race a hipMemset on the null stream
  if ((rc = work_ctr_.alloc(16 * 9))) return rc;
  return 0;
}

int WarehouseBackend::launch(const StepArgs& a, const pcx_buffers& out, hipStream_t s) {
  if (a.n_steps != 1) return set_error(PCX_E_INVALID, "warehouse backend: one step per launch");
  if (a.export_curtains && !curtains_.ptr) {
    int rc = curtains_.alloc((size_t)lay_.FW * bpad_);
    if (rc) return rc;
  }
  Ptrs P{tables_.ptr, state_.ptr, track_.ptr, curtains_.ptr, batch_, bpad_, {}};
  const int64_t groups = bpad_ / WAVE;
  // Launch shape as for scrolly_maze (profiles/r01_tuning.md): single-wave
  // workgroups with LDS padded so that about four of them share a CU; four
  // waves per group when the batch leaves most of the chip idle.
  int coop_below = 5, waves_per_cu = 4;  // measured: 1,048,576 envs 0.320 ms at 4 per CU vs 0.345 at 8 (tools/knob_sweep_r02.sh)
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
  const bool epi = epi_.out != nullptr;  // the feature-array epilogue has its own instances (whole-dword boards only)
  // (round 6) owner codes: plain steps -- no epilogue, no fused croppers, occluded layers; PCX_WM_CODES=0: the mask loop
  bool codes = !epi && !fused_.on && !unoccluded_;
  if (const char* e = getenv("PCX_WM_CODES")) codes = codes && atoi(e) != 0;
  if (codes) {
    lds = base_lds_bytes(true);
    if (!coop && waves_per_cu > 0) {
      size_t want = ((size_t)(160 * 1024) / (size_t)waves_per_cu) & ~(size_t)255;
      if (want > 64 * 1024) want = 64 * 1024;
      if (want > lds) lds = want;
    }
  }
  // (round 5) persistent workers: plain steps of the compiled shapes from four units per CU up.  One workgroup of W workers per
  // CU, `lock` of them streaming at a time, tickets (with stealing) from 24 units per CU up; PCX_WM_PW=0: the round-2 shape.
  bool pw = !coop && !epi && !fused_.on && !unoccluded_ && static_shape_ && a.mode == 0 && !a.export_curtains && (a.debug & ~16) == 0;
  if (const char* e = getenv("PCX_WM_PW")) pw = pw && atoi(e) != 0;
  if (pw) {
    // Measured (profiles/r05_warehouse_workers_sweep.txt; the round-2 shape: 0.0824 / 0.3203 ms at 262,144 / 1,048,576
    // environments, same box): this kernel's render loop composes every dword from masks -- one streaming wave moves a unit at
    // half the CU's rate (one slot: 0.167 ms), so four stream at a time: ONE workgroup of eight workers with four slots
    // 0.0712 / 0.3128; from 32 units per CU up four single-worker workgroups per CU (the round-2 residency, persistent,
    // state prefetched) 0.0775 / 0.2981.
    // ... and which of the two it is differs from box to box (262,144 environments: 0.0712 / 0.0775 on one, 0.0850 / 0.0806 on
    // another), so the engine measures on its own first launches (ShapeTuner, pcx_internal.h): the rule's choice first.
    const bool many = groups >= (int64_t)num_cus_ * 32;
    struct Cand { int workers, per_cu, lock; };
    static const Cand few_m[ShapeTuner::NC] = {{8, 1, 4}, {1, 4, 0}, {2, 4, 1}, {6, 1, 4}};
    static const Cand many_m[ShapeTuner::NC] = {{1, 4, 0}, {6, 1, 4}, {8, 1, 4}, {2, 3, 1}};
    // (round 6) the owner-code loop is a third of the instructions per store: one streaming wave moves a unit at 60 % of the CU's
    // rate (one slot: 0.111 ms at 262,144 environments, the mask loop 0.167), TWO saturate it, and fewer concurrent streams
    // leave shorter tails -- four workers with two shared slots 0.0771 / 0.3002 ms at 262,144 / 1,048,576 environments where
    // the mask loop's best shapes measure 0.0811 / 0.2984 on the same box (profiles/r06_warehouse_codes_sweep.txt)
    static const Cand few_k[ShapeTuner::NC] = {{4, 1, 2}, {1, 4, 0}, {2, 2, 2}, {6, 1, 3}};
    static const Cand many_k[ShapeTuner::NC] = {{4, 1, 2}, {2, 2, 2}, {1, 4, 0}, {8, 1, 3}};
    const Cand* const few_c = codes ? few_k : few_m;
    const Cand* const many_c = codes ? many_k : many_m;
    const bool knobs = getenv("PCX_WM_WORKERS") || getenv("PCX_WM_PER_CU") || getenv("PCX_WM_LOCK") || getenv("PCX_WM_GRID") ||
                       (getenv("PCX_WM_TUNE") && atoi(getenv("PCX_WM_TUNE")) == 0);
    if (knobs) tuner_.off = true;
    const Cand& cand = (many ? many_c : few_c)[tuner_.pick(a, s)];
    int workers = cand.workers, per_cu = cand.per_cu, lock = cand.lock;
    if (const char* e = getenv("PCX_WM_WORKERS")) { const int v = atoi(e); if (v >= 1 && v <= 8) workers = v; }
    if (const char* e = getenv("PCX_WM_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 8) per_cu = v; }
    if (const char* e = getenv("PCX_WM_LOCK")) lock = atoi(e);
    int dynamic = groups >= (int64_t)num_cus_ * 24;
    if (const char* e = getenv("PCX_WM_DYNAMIC")) dynamic = atoi(e) != 0;
    const size_t tab_words = (size_t)lay_.QW * (2 + NB_) + 3 * R_;
    const size_t o_sdesc = (tab_words + (size_t)WAVE * lay_.FWP + 1) & ~(size_t)1;
    const size_t region = o_sdesc + (codes ? WAVE * lay_.CP : 2 * NS_ * WAVE) + WAVE - tab_words;  // flat, sdesc / codes, skip: the kernel's O_WCORNER - O_FLAT
    const size_t w_words = (region + (size_t)(W_POS + NS_ + 1) * WAVE + 1) & ~(size_t)1;
    size_t lds_pw = (tab_words + 2 + (size_t)workers * w_words) * 4;
    while (workers > 1 && lds_pw > 64 * 1024) { --workers; lds_pw = (tab_words + 2 + (size_t)workers * w_words) * 4; }
    int64_t wgs = (int64_t)num_cus_ * per_cu;
    if (const char* e = getenv("PCX_WM_GRID")) { const int v = atoi(e); if (v >= 1) wgs = v; }  // (tests: few workgroups, many units each)
    const int64_t want = (groups + workers - 1) / workers;
    if (wgs > want) wgs = want;
    if (wgs * workers >= groups) dynamic = 0;  // every unit is some worker's first
    P.work.ctr = work_ctr_.ptr;
    P.work.n_units = (uint32_t)groups;
    P.work.dynamic = dynamic;
    P.work.lock = lock;
#define X(ns, r, c, nb)                                                                                     \
  if (!launched && NS_ == ns && R_ == r && C_ == c && NB_ == nb) {                                          \
    if (codes) hipLaunchKernelGGL((pcx_warehouse_step<ns, r, c, nb, 1, false, false, true, true>), dim3((unsigned)wgs), dim3(workers * WAVE), lds_pw, s, k_, P, a, out, epi_, fused_.ptr()); \
    else hipLaunchKernelGGL((pcx_warehouse_step<ns, r, c, nb, 1, false, false, true>), dim3((unsigned)wgs), dim3(workers * WAVE), lds_pw, s, k_, P, a, out, epi_, fused_.ptr()); \
    launched = true;                                                                                        \
  }
    PCX_WM_SHAPES(X)
#undef X
    if (launched) tuner_.launched(s);
    last_shape_ = launched ? 3 : last_shape_;
  }
  if (!launched) last_shape_ = coop ? 10 : 0;
#define PCX_WM_LAUNCH(ns, r, c, nb, nw, ep)                                                                  \
  do {                                                                                                      \
    if (!ep && codes) hipLaunchKernelGGL((pcx_warehouse_step<ns, r, c, nb, nw, false, false, false, true>), dim3((unsigned)groups), dim3(nw * WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr()); \
    else hipLaunchKernelGGL((pcx_warehouse_step<ns, r, c, nb, nw, ep>), dim3((unsigned)groups), dim3(nw * WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr()); \
  } while (0)
#define X(ns, r, c, nb)                                                                                     \
  if (!launched && !unoccluded_ && NS_ == ns && R_ == r && C_ == c && NB_ == nb) {                          \
    if (epi && coop) PCX_WM_LAUNCH(ns, r, c, nb, 4, true);                                                  \
    else if (epi) PCX_WM_LAUNCH(ns, r, c, nb, 1, true);                                                     \
    if (!epi && coop) PCX_WM_LAUNCH(ns, r, c, nb, 4, false);                                                \
    else if (!epi) PCX_WM_LAUNCH(ns, r, c, nb, 1, false);                                                   \
    launched = true;                                                                                        \
  }
  PCX_WM_SHAPES(X)
#undef X
  if (!launched && (!static_shape_ || unoccluded_) && !epi) {  // run-time-shape instances, one per number of sprites
    if (lds > 64 * 1024) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: board too large for the step kernel's LDS tables");
    switch (NS_) {
#define PCX_WM_DYN(ns)                                                                                      \
  case ns:                                                                                                  \
    if (unoccluded_) {                                                                                      \
      if (coop) hipLaunchKernelGGL((pcx_warehouse_step<ns, 0, 0, 4, 4, false, true>), dim3((unsigned)groups), dim3(4 * WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr()); \
      else hipLaunchKernelGGL((pcx_warehouse_step<ns, 0, 0, 4, 1, false, true>), dim3((unsigned)groups), dim3(WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr());          \
    } else if (coop) { PCX_WM_LAUNCH(ns, 0, 0, 4, 4, false); } else { PCX_WM_LAUNCH(ns, 0, 0, 4, 1, false); } \
    launched = true;                                                                                        \
    break;
      PCX_WM_DYN(2) PCX_WM_DYN(3) PCX_WM_DYN(4) PCX_WM_DYN(5) PCX_WM_DYN(6) PCX_WM_DYN(7) PCX_WM_DYN(8) PCX_WM_DYN(9)
      PCX_WM_DYN(10) PCX_WM_DYN(11)
#undef PCX_WM_DYN
      default: break;
    }
  }
#undef PCX_WM_LAUNCH
  if (!launched) return set_error(PCX_E_UNSUPPORTED, "warehouse backend: no instance");
  PCX_HIP(hipGetLastError());
  return 0;
}

int WarehouseBackend::read_things(int64_t env0, int64_t n, pcx_sprite_state* sprites, uint8_t* curtains) {
  std::vector<uint32_t> st((size_t)NW_ * n);
  PCX_HIP(hipDeviceSynchronize());
  for (int w = 0; w < NW_; ++w)
    PCX_HIP(hipMemcpy(st.data() + (size_t)w * n, state_.ptr + (size_t)w * bpad_ + env0, n * 4, hipMemcpyDeviceToHost));
  auto word = [&](int w, int64_t i) { return st[(size_t)w * n + i]; };
  for (int64_t i = 0; i < n; ++i) {
    if (curtains) memset(curtains + (size_t)i * lay_.cells, 0, lay_.cells);
    for (int s = 0; s < NS_; ++s) {
      const uint32_t pw = word(W_POS + s, i);
      const int vr = (int16_t)(pw & 0xFFFF), vc = (int16_t)(pw >> 16);
      const bool on = vr >= 0 && vr < R_ && vc >= 0 && vc < C_;
      if (sprites) {
        pcx_sprite_state& o = sprites[i * NS_ + s];
        memset(&o, 0, sizeof o);
        o.vrow = vr; o.vcol = vc;
        o.row = on ? vr : 0; o.col = on ? vc : 0;
        o.visible = (word(W_SFLAGS, i) >> (2 * s)) & 1;
      }
      // the judge's curtain: boxes standing on goal cells (recomputed every frame)
      if (curtains && s < NS_ - 1) {
        const int cell = on ? vr * C_ + vc : 0;
        if (goal_[cell]) curtains[(size_t)i * lay_.cells + cell] = 1;
      }
    }
  }
  return 0;
}

}  // namespace wm

Backend* make_warehouse_backend() { return new wm::WarehouseBackend(); }

}  // namespace pcx
