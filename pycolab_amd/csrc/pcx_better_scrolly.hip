// pcx_better_scrolly.hip -- hand-written fused step kernel for
// better_scrolly_maze (reference: pycolab/examples/better_scrolly_maze.py:250-320
// driven by engine.py:583-847 and prefab_parts/sprites.py MazeWalker): the game
// that pairs MazeWalkers with real ObservationCroppers on boards of up to 45x89
// cells.  gfx950 only.
//
// One launch = one Engine.play() of every environment of the batch; same shape as
// the other hand-written kernels: 64 consecutive environments per
// workgroup, logic phase lane == environment, render phase = pcx_stream.h.
//
// ONE update group (`a b c P @`), so every entity sees the repaint the step
// started with.  The maze walls are the (static) backdrop; CashDrape's
// whole-board curtain only ever loses cells, so per environment it is a bit mask
// over the template's coin list (92-186 coins: 3-6 words instead of a 4005-bit
// curtain), in HBM (SoA over the batch) and in LDS, where a static table gives
// every board dword the list indices of its four cells -- for the logic phase's
// "is there a coin here" and for the streaming phase alike (pcx_stream.h,
// cell_ids).  That keeps a group's LDS near 20 KB, i.e. eight workgroups per CU,
// where a flat curtain per environment allowed three (0.64 -> see
// profiles/r02_tuning.md).  The number of coins left rides in the flags word,
// so "no coins left" needs no reduction.  A MazeWalker probe with impassable ==
// '#' is "wall cell with nothing painted over it" from the register snapshot of
// the sprites' cells, the coin bit and a static wall bit vector.
// Other casts, z-orders, impassable sets or shapes: the table-driven kernel.

#include "pcx_internal.h"
#include "pcx_stream.h"

#include <cstdlib>
#include <cstring>

namespace pcx {
namespace bs {

using stream::WAVE;
constexpr int NS = 4;  // patrollers a, b, c and the player P (template order)
constexpr int IP = 3;
constexpr int ND = 1;  // the coins '@'
constexpr int NB = 2;  // ' ' and '#'

// State words (uint32 [NW][batch_padded]): frame, flags, positions, coin curtain.
enum : int { W_FRAME = 0, W_FLAGS, W_POS, W_COINS = W_POS + NS };
constexpr uint32_t F_OVER = 1u, F_ERR_SHIFT = 1;
constexpr int F_SF_SHIFT = 4;     // per sprite: visible, prior_visible, moving_east (3 bits)
constexpr int F_LEFT_SHIFT = 16;  // coins left (CashDrape.curtain.any() without a reduction)

constexpr int MAX_CW = 8;  // coin-mask words (at most 255 coins: a list index is a byte)

struct Consts {
  int32_t rows, cols;  // the board (read by the run-time-shape instance)
  int32_t n_actions, n_coins, CW;
  uint32_t confined;
  uint32_t above[NS];
  uint32_t init[W_COINS];
  uint32_t sprite_off[NS], sprite_ch4[NS], drape_off, drape_ch4, bchar_off[NB], bchar_ch4[NB];
};

struct Ptrs {
  const uint32_t* tables;        // staged into LDS: backdrop4 [QW], bdmask [NB][QW], coin ids [QW], wall bits [FW]
  const uint32_t* coin_cell;     // [n_coins] cell of every coin of the list (export_curtains only)
  uint32_t* state;               // [NW][bpad]
  int32_t* track;                // [NS][bpad]
  uint32_t* curtains;            // [1][FW][bpad] (export_curtains)
  int64_t batch, bpad;
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

// SR x SC: the board's shape when the instance is compiled for it (the three shipped boards); 0 x 0: read
// from k.rows / k.cols (boards the example does not ship).
template <int SR, int SC, int NWAVES, bool EPI = false>
__global__ __launch_bounds__(NWAVES* WAVE) void pcx_better_scrolly_step(const Consts k, const Ptrs P, const StepArgs a,
                                                                         const pcx_buffers out, const stream::EpilogueArgs epi,
                                                                         const crop::FusedCrops* fc) {
  extern __shared__ uint32_t lds[];
  const int R = SR ? SR : k.rows, C = SC ? SC : k.cols;  // (constants in the compiled-shape instances)
  constexpr int SQW = ((SR * SC + 3) & ~3) / 4;           // dwords per plane, 0: run-time shape
  const int cells = R * C, pitch = (cells + 3) & ~3, QW = pitch / 4, FW = (cells + 31) / 32;
  constexpr int L = NS + ND + NB, CWP = MAX_CW | 1;
  const int O_BD = 0, O_BDM = O_BD + QW, O_CID = O_BDM + NB * QW, O_WALL = O_CID + QW, O_TAB_END = O_WALL + FW;
  const int O_CM = O_TAB_END, O_SDESC = (O_CM + WAVE * CWP + 1) & ~1, O_SKIP = O_SDESC + 2 * NS * WAVE;
  const int O_WCORNER = O_SKIP + WAVE;  // fused croppers' window corners
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < O_TAB_END; i += NWAVES * WAVE) lds[i] = P.tables[i];
  const uint32_t* const cid = lds + O_CID;    // list indices of every board dword's four cells (0xFF: no coin there)
  const uint32_t* const wall = lds + O_WALL;
  uint32_t* const cm = lds + O_CM;            // [64][CWP]: which coins of the list every environment still has
  uint32_t* const mine = cm + lane * CWP;
  uint2* const sdesc = reinterpret_cast<uint2*>(lds + O_SDESC);
  uint32_t* const skipv = lds + O_SKIP;
  uint32_t* const wcorner = lds + O_WCORNER;
  const int CW = k.CW;
  __syncthreads();

  const int64_t env0 = (int64_t)blockIdx.x * WAVE;
  if (wave == 0) {
    // ---- logic phase: lane == environment -------------------------------------
    const int64_t env = env0 + lane, bp = P.bpad;
    const bool live = env < P.batch;
    uint32_t* const st = P.state + env;
    uint32_t flags = 0, ld_frame = 0, ld_pos[NS] = {};
    int ld_action = PCX_ACTION_NONE;
    bool skip = !live, do_reset = false;
    int action = PCX_ACTION_NONE;
    if (live) {
      flags = st[W_FLAGS * bp];
      if (a.mode != 1) {
        ld_frame = st[W_FRAME * bp];
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
      // the coin mask: from HBM (requested with the other words below) or all ones after a reset
      if (do_reset) {
        for (int i = 0; i < CW; ++i) {
          const int rest = k.n_coins - 32 * i;
          mine[i] = rest >= 32 ? 0xFFFFFFFFu : ((1u << rest) - 1u);
        }
      } else {
        uint32_t v[MAX_CW];
#pragma unroll
        for (int j = 0; j < MAX_CW; ++j) v[j] = j < CW ? st[(W_COINS + j) * bp] : 0u;
#pragma unroll
        for (int j = 0; j < MAX_CW; ++j) if (j < CW) mine[j] = v[j];
      }
      int frame, left;
      uint32_t sflags, err;
      int vr[NS], vc[NS], vis[NS], prior[NS], east[NS];
      if (do_reset) {  // engine.py:520-581 its_showtime: fresh template state, frame 0 = play(None)
        frame = (int)k.init[W_FRAME];
        sflags = k.init[W_FLAGS] >> F_SF_SHIFT;
        left = (int)(k.init[W_FLAGS] >> F_LEFT_SHIFT);
        err = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) { vr[s] = pos_r(k.init[W_POS + s]); vc[s] = pos_c(k.init[W_POS + s]); }
        action = PCX_ACTION_NONE;
      } else {
        frame = (int)ld_frame;
        sflags = flags >> F_SF_SHIFT;
        left = (int)(flags >> F_LEFT_SHIFT);
        err = (flags >> F_ERR_SHIFT) & 7u;
#pragma unroll
        for (int s = 0; s < NS; ++s) { vr[s] = pos_r(ld_pos[s]); vc[s] = pos_c(ld_pos[s]); }
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        vis[s] = (sflags >> (3 * s)) & 1; prior[s] = (sflags >> (3 * s + 1)) & 1; east[s] = (sflags >> (3 * s + 2)) & 1;
      }
      int reward = 0, reward_set = 0, over = 0;
      float discount = 1.0f;
      frame += 1;  // engine.py:698-735

      auto on_board = [&](int r, int c) { return (unsigned)r < (unsigned)R && (unsigned)c < (unsigned)C; };
      auto true_cell = [&](int r, int c) { return on_board(r, c) ? r * C + c : 0; };  // Sprite.position
      auto teleport = [&](int s, int nr, int nc) {  // sprites.py:315-352
        const bool old_on = on_board(vr[s], vc[s]), new_on = on_board(nr, nc);
        if (old_on && !new_on) { prior[s] = vis[s]; vis[s] = 0; }
        if (!old_on && new_on) vis[s] = prior[s];
        vr[s] = nr; vc[s] = nc;
      };
      // what the last repaint showed at a cell: layers['#'] (rendering.py:177-179) is a wall
      // cell of the backdrop with no sprite and no coin painted over it
      int cell0[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) cell0[s] = vis[s] ? true_cell(vr[s], vc[s]) : -1;
      auto coin_id = [&](int cell) { return (int)((cid[cell >> 2] >> (8 * (cell & 3))) & 0xFFu); };
      auto coin_alive = [&](int id) { return id != 0xFF && ((mine[id >> 5] >> (id & 31)) & 1); };
      auto shows_wall = [&](int cell) {
        bool painted = coin_alive(coin_id(cell));
#pragma unroll
        for (int j = 0; j < NS; ++j) painted |= cell0[j] == cell;
        return !painted && ((wall[cell >> 5] >> (cell & 31)) & 1);
      };
      // sprites.py:356-389 _move one column / row with impassable == '#'
      auto move = [&](int s, int dr, int dc) {
        const int nr = vr[s] + dr, nc = vc[s] + dc;
        if (!on_board(nr, nc)) { if (!((k.confined >> s) & 1)) teleport(s, nr, nc); return; }
        if (!shows_wall(nr * C + nc)) teleport(s, nr, nc);
      };

      // ---- PatrollerSprite.update (better_scrolly_maze.py:284-301), a, b, c in turn
      const int p_cell_start = true_cell(vr[IP], vc[IP]);  // things['P'].position: the player moves after them
#pragma unroll
      for (int s = 0; s < IP; ++s) {
        if (frame & 1) continue;  // _stay on odd frames
        const bool on = on_board(vr[s], vc[s]);
        const int row = on ? vr[s] : 0, col = on ? vc[s] : 0;
        // layers['#'][row, col - 1] / [row, col + 1] with numpy's index rules
        const int cw = col - 1 < 0 ? col - 1 + C : col - 1;
        if (shows_wall(row * C + cw)) east[s] = 1;
        if (col + 1 >= C) err |= ERR_INDEX;
        else if (shows_wall(row * C + col + 1)) east[s] = 0;
        move(s, 0, east[s] ? 1 : -1);
        if (true_cell(vr[s], vc[s]) == p_cell_start) { over = 1; discount = 0.0f; }
      }
      // ---- PlayerSprite.update (:258-272) --------------------------------------------
      if ((unsigned)action <= 3u) move(IP, action == 0 ? -1 : action == 1 ? 1 : 0, action == 2 ? -1 : action == 3 ? 1 : 0);
      if (action == 5) { over = 1; discount = 0.0f; }
      // ---- CashDrape.update (:311-320) -------------------------------------------------
      int changed_word = -1;
      {
        const int id = coin_id(true_cell(vr[IP], vc[IP]));
        if (coin_alive(id)) {
          reward += 100; reward_set = 1;
          mine[id >> 5] &= ~(1u << (id & 31));
          changed_word = id >> 5;
          if (--left == 0) { over = 1; discount = 0.0f; }
        }
      }

      // ---- _apply_and_clear_plot (engine.py:761-847) + state write-back -----------------
      st[W_FRAME * bp] = (uint32_t)frame;
      uint32_t sf = 0;
      int32_t tw[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        st[(W_POS + s) * bp] = pack_pos(vr[s], vc[s]);
        sf |= ((uint32_t)vis[s] | ((uint32_t)prior[s] << 1) | ((uint32_t)east[s] << 2)) << (3 * s);
        const bool on = on_board(vr[s], vc[s]);
        tw[s] = (on ? vr[s] : 0) | ((on ? vc[s] : 0) << 8) | (vis[s] << 16) | ((int)do_reset << 24);
        P.track[(size_t)s * bp + env] = tw[s];
      }
      st[W_FLAGS * bp] = (over ? F_OVER : 0u) | ((err & 7u) << F_ERR_SHIFT) | (sf << F_SF_SHIFT) | ((uint32_t)left << F_LEFT_SHIFT);
      if (do_reset) {
        for (int i = 0; i < CW; ++i) st[(W_COINS + i) * bp] = mine[i];
      } else if (changed_word >= 0) {
        st[(W_COINS + changed_word) * bp] = mine[changed_word];  // a pickup rewrites one word
      }
      if (a.export_curtains) {  // a cropper tracks the coins: the raw curtain as flat cell bits (rare path)
        for (int i = 0; i < FW; ++i) P.curtains[(size_t)i * bp + env] = 0;
        for (int id = 0; id < k.n_coins; ++id)
          if ((mine[id >> 5] >> (id & 31)) & 1) {
            const uint32_t cell = P.coin_cell[id];
            P.curtains[(size_t)(cell >> 5) * bp + env] |= 1u << (cell & 31);
          }
      }
      const stream::CurtainSrc csrc{P.curtains, bp, FW, R, C};
      if (fc)  // fused croppers (after the export: a cropper may follow the coins): the windows follow this step's positions (cropping.py:393-426)
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

      // ---- render descriptors ---------------------------------------------------------------
      // engine.py:751-757: a sprite is painted iff it is visible and neither a sprite in front of it
      // nor (if the coins are in front of it) a coin holds its cell; a painted sprite hides the coin
      // under it (in this LDS copy of the mask: the state words are already on their way to HBM)
      int cellv[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) cellv[s] = vis[s] ? true_cell(vr[s], vc[s]) : -1;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int c = cellv[s];
        bool shown = c >= 0;
        if (shown) {
          const uint32_t ab = k.above[s];
#pragma unroll
          for (int j = 0; j < NS; ++j)
            if (j != s && ((ab >> j) & 1) && cellv[j] == c) shown = false;
          const int id = coin_id(c);
          if (coin_alive(id)) {
            if ((ab >> NS) & 1) shown = false;
            else if (shown) mine[id >> 5] &= ~(1u << (id & 31));
          }
        }
        sdesc[s * WAVE + lane] = make_uint2(shown ? (uint32_t)(c >> 2) : 0xFFFFFFFFu, 0xFFu << ((c & 3) * 8));
      }
    }
    skipv[lane] = skip;
  }
  __syncthreads();
  if (a.debug & 2) return;

  stream::PlaneMap<NS, ND, NB> pm;
#pragma unroll
  for (int s = 0; s < NS; ++s) { pm.sprite_off[s] = k.sprite_off[s]; pm.sprite_ch4[s] = k.sprite_ch4[s]; }
  pm.drape_off[0] = k.drape_off; pm.drape_ch4[0] = k.drape_ch4;
  uint32_t bch4[NB > 0 ? NB : 1] = {};
#pragma unroll
  for (int b = 0; b < NB; ++b) { pm.bchar_off[b] = k.bchar_off[b]; bch4[b] = k.bchar_ch4[b]; }
  const uint32_t env_stride = (uint32_t)(1 + L) * (uint32_t)pitch;
  if (!(fc && fc->only)) {
    // (round 6) single-wave workgroups without an epilogue: the stores of KB iterations regrouped plane by plane
    // (pcx_stream.h stream_planes_burst; PCX_DEBUG bits 64 / 128 / 256: KB = 4 / 8 / 2 -- A/B)
    if constexpr (!EPI && NWAVES == 1) if (!fc) {
      if (a.debug & 64) { stream::stream_planes_burst<NS, ND, NB, SQW, 4>(pm, out.planes + (size_t)env0 * env_stride, env_stride, lds + O_BD, lds + O_BDM, cm, sdesc, skipv, CWP, lane, cid, QW); return; }
      if (a.debug & 128) { stream::stream_planes_burst<NS, ND, NB, SQW, 8>(pm, out.planes + (size_t)env0 * env_stride, env_stride, lds + O_BD, lds + O_BDM, cm, sdesc, skipv, CWP, lane, cid, QW); return; }
      if (a.debug & 256) { stream::stream_planes_burst<NS, ND, NB, SQW, 2>(pm, out.planes + (size_t)env0 * env_stride, env_stride, lds + O_BD, lds + O_BDM, cm, sdesc, skipv, CWP, lane, cid, QW); return; }
    }
    stream::stream_planes<NS, ND, NB, SQW, NWAVES, EPI>(pm, out.planes + (size_t)env0 * env_stride, env_stride, lds + O_BD, lds + O_BDM,
                                                         cm, sdesc, skipv, CWP, lane, wave, epi, env0, cid, QW, nullptr, nullptr, lds);
  }
  if (fc)
    stream::stream_windows<NS, ND, NB, SQW, NWAVES, SR, SC, true>(fc, pm, bch4, env0, lds + O_BD, cm, sdesc, skipv, CWP, lane, wave, wcorner,
                                                                 cid, stream::BoardShape{R, C, QW});
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------

#define PCX_BS_SHAPES(X) X(45, 89) X(29, 30) X(29, 89)

class BetterScrollyBackend : public Backend {
 public:
  int init(const pcx_template& t, int64_t batch) override;
  int launch(const StepArgs& a, const pcx_buffers& out, hipStream_t s) override;
  int read_things(int64_t env0, int64_t n, pcx_sprite_state* sprites, uint8_t* curtains) override;
  int64_t bytes_per_step() const override {
    // read: action 4 + state 4 NW; write: the scalar state words (a pickup adds one coin-mask word)
    // + planes (1 + L) cells + results 15
    return 4 + 4 * (int64_t)NW_ + 4 * (int64_t)W_COINS + (int64_t)(1 + L_) * lay_.cells + 15;
  }
  const char* kernel_name() const override { return "pcx_better_scrolly_step"; }
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
  int set_fused_croppers(const crop::FusedCrops* fc) override { return fused_.set(fc, false, R_, C_); }
  bool fused_window_features() const override { return true; }
  size_t base_lds_bytes() const {  // the kernel's own dynamic LDS (before padding / the channels-last exchange areas)
    return ((size_t)lay_.QW * (2 + NB) + lay_.FW + WAVE * (MAX_CW | 1) + 2 + 2 * NS * WAVE + WAVE + stream::WCORNER_WORDS) * 4;
  }
  stream::EpilogueArgs* epilogue_args() override { return &epi_; }
  int set_epilogue(const pcx_epilogue_desc* d) override {  // include/pcx.h pcx_engine_set_epilogue
    if (d && !static_shape_) return set_error(PCX_E_UNSUPPORTED, "better_scrolly backend: the feature-array epilogue exists for the compiled boards");
    int sc[NS], dc = k_.drape_ch4 & 0xFF, bc[NB > 0 ? NB : 1] = {};
    for (int s = 0; s < NS; ++s) sc[s] = k_.sprite_ch4[s] & 0xFF;
    for (int b = 0; b < NB; ++b) bc[b] = k_.bchar_ch4[b] & 0xFF;
    if (!stream::fill_epilogue(epi_, d, lay_.cells, sc, NS, &dc, 1, bc, NB, base_lds_bytes() < 64 * 1024 ? 64 * 1024 - base_lds_bytes() : 0, 4))
      return set_error(PCX_E_UNSUPPORTED, "better_scrolly backend: the channels-last epilogue needs rows*cols %% 4 == 0 and a stack whose exchange areas fit the LDS left");
    return 0;
  }

 private:
  stream::FusedCropsHolder fused_;
  Consts k_{};
  stream::EpilogueArgs epi_{};
  stream::Layout lay_;
  int R_ = 0, C_ = 0, L_ = 0, NW_ = 0;
  int64_t batch_ = 0, bpad_ = 0;
  int num_cus_ = 256;
  bool static_shape_ = false;  // a compiled instance for exactly this board exists
  DevArray<uint32_t> tables_, coin_cell_, state_, curtains_;
  DevArray<int32_t> track_;
  std::vector<uint32_t> h_coin_cell_;  // host copy for read_things
};

int BetterScrollyBackend::init(const pcx_template& t, int64_t batch) {
  Consts& k = k_;
  batch_ = batch;
  bpad_ = (batch + WAVE - 1) / WAVE * WAVE;
  if (const char* e = getenv("PCX_FORCE_GENERIC")) if (atoi(e)) return set_error(PCX_E_UNSUPPORTED, "better_scrolly backend: PCX_FORCE_GENERIC");
  if (!t.occlusion_in_layers || t.n_directives) return set_error(PCX_E_UNSUPPORTED, "better_scrolly backend: occluded layers, no directives");
  R_ = t.rows; C_ = t.cols; L_ = t.n_chars;
  static_shape_ = false;
#define X(r, c) static_shape_ |= R_ == r && C_ == c;
  PCX_BS_SHAPES(X)
#undef X
  // any other board takes the run-time-shape instance, as long as its tables fit the 64 KB of LDS a workgroup may have
  const bool shape_ok = static_shape_ || (R_ <= 255 && C_ <= 255 &&
      ((size_t)((R_ * C_ + 3) / 4) * (2 + NB) + (R_ * C_ + 31) / 32 + WAVE * (MAX_CW | 1) + 2 + 2 * NS * WAVE + WAVE + stream::WCORNER_WORDS) * 4 <= 64 * 1024);
  if (!shape_ok || t.n_sprites != NS || t.n_drapes != ND || L_ != NS + ND + NB || t.n_groups != 1 || t.n_things != NS + ND)
    return set_error(PCX_E_UNSUPPORTED, "better_scrolly backend: the shipped boards and cast only");
  lay_.set(R_, C_);
  k.rows = R_; k.cols = C_;
  // sprites: three patrollers then the player; impassable == {'#'}; schedule a b c P @; z-order: patrollers, coins, player
  for (int s = 0; s < NS; ++s) {
    const pcx_sprite_desc& sd = t.sprites[s];
    if (sd.program != (s == IP ? PCX_PROG_BS_PLAYER : PCX_PROG_BS_PATROLLER) || !sd.is_walker || sd.egocentric)
      return set_error(PCX_E_UNSUPPORTED, "better_scrolly backend: unexpected cast");
    for (int i = 0; i < 16; ++i)
      if (sd.impassable[i] != (i == ('#' >> 3) ? (1u << ('#' & 7)) : 0u))
        return set_error(PCX_E_UNSUPPORTED, "better_scrolly backend: impassable must be '#'");
    if (t.schedule[s] != sd.ch || t.group_of[s] != 0) return set_error(PCX_E_UNSUPPORTED, "better_scrolly backend: unexpected update schedule");
  }
  const pcx_drape_desc& dd = t.drapes[0];
  if (dd.program != PCX_PROG_BS_CASH || dd.is_scrolly || t.schedule[NS] != dd.ch)
    return set_error(PCX_E_UNSUPPORTED, "better_scrolly backend: the drape must be the CashDrape, updated last");
  int zpos[NS + ND];
  for (int z = 0; z < t.n_things; ++z) {
    int idx = -1;
    for (int s = 0; s < NS; ++s) if (t.sprites[s].ch == t.z_order[z]) idx = s;
    if (t.z_order[z] == dd.ch) idx = NS;
    if (idx < 0) return set_error(PCX_E_INVALID, "better_scrolly backend: z_order names an unknown character");
    zpos[idx] = z;
  }
  for (int s = 0; s < NS; ++s) {
    k.above[s] = 0;
    for (int j = 0; j < NS + ND; ++j) if (zpos[j] > zpos[s]) k.above[s] |= 1u << j;
  }
  k.n_actions = t.n_actions;
  k.confined = 0;
  for (int s = 0; s < NS; ++s) if (t.sprites[s].confined) k.confined |= 1u << s;
  auto layer_of = [&](int ch) { for (int i = 0; i < L_; ++i) if (t.chars[i] == ch) return i; return -1; };
  for (int s = 0; s < NS; ++s) {
    k.sprite_off[s] = (uint32_t)(1 + layer_of(t.sprites[s].ch)) * lay_.pitch;
    k.sprite_ch4[s] = t.sprites[s].ch * 0x01010101u;
  }
  k.drape_off = (uint32_t)(1 + layer_of(dd.ch)) * lay_.pitch;
  k.drape_ch4 = dd.ch * 0x01010101u;
  if (layer_of('#') < 0) return set_error(PCX_E_UNSUPPORTED, "better_scrolly backend: no '#' in the backdrop");

  std::vector<uint32_t> tab((size_t)lay_.QW * (2 + NB) + lay_.FW, 0);
  memcpy(tab.data(), t.backdrop, lay_.cells);
  int nb = 0;
  for (int i = 0; i < L_; ++i) {
    const int ch = t.chars[i];
    bool thing = ch == dd.ch;
    for (int s = 0; s < NS; ++s) thing |= t.sprites[s].ch == ch;
    if (thing) continue;
    if (nb >= NB) return set_error(PCX_E_INVALID, "better_scrolly backend: inconsistent character set");
    k.bchar_off[nb] = (uint32_t)(1 + i) * lay_.pitch;
    k.bchar_ch4[nb] = (uint32_t)ch * 0x01010101u;
    uint8_t* m = reinterpret_cast<uint8_t*>(tab.data() + (size_t)lay_.QW * (1 + nb));
    for (int c = 0; c < lay_.cells; ++c) m[c] = t.backdrop[c] == ch;
    ++nb;
  }
  if (nb != NB) return set_error(PCX_E_INVALID, "better_scrolly backend: inconsistent character set");
  // the coin list (row-major) and, per board dword, the list indices of its four cells
  uint8_t* ids = reinterpret_cast<uint8_t*>(tab.data() + (size_t)lay_.QW * (1 + NB));
  memset(ids, 0xFF, (size_t)lay_.QW * 4);
  h_coin_cell_.clear();
  for (int c = 0; c < lay_.cells; ++c)
    if (dd.curtain[c]) {
      if (h_coin_cell_.size() >= 255) return set_error(PCX_E_UNSUPPORTED, "better_scrolly backend: more than 255 coins");
      ids[c] = (uint8_t)h_coin_cell_.size();
      h_coin_cell_.push_back((uint32_t)c);
    }
  const int coins = (int)h_coin_cell_.size();
  k.n_coins = coins;
  k.CW = (coins + 31) / 32;
  if (k.CW > MAX_CW) return set_error(PCX_E_UNSUPPORTED, "better_scrolly backend: too many coins");
  NW_ = W_COINS + k.CW;
  uint32_t* wall = tab.data() + (size_t)lay_.QW * (2 + NB);
  for (int c = 0; c < lay_.cells; ++c) if (t.backdrop[c] == '#') wall[c >> 5] |= 1u << (c & 31);

  memset(k.init, 0, sizeof k.init);
  k.init[W_FRAME] = (uint32_t)-1;
  uint32_t sf = 0;
  for (int s = 0; s < NS; ++s) {
    const pcx_sprite_desc& sd = t.sprites[s];
    sf |= ((uint32_t)(sd.visible != 0) | ((uint32_t)(sd.prior_visible != 0) << 1) |
           ((uint32_t)(s != IP && sd.param[0] != 0) << 2)) << (3 * s);  // _moving_east, better_scrolly_maze.py:282
    k.init[W_POS + s] = ((uint32_t)sd.vrow & 0xFFFFu) | ((uint32_t)sd.vcol << 16);
  }
  k.init[W_FLAGS] = (sf << F_SF_SHIFT) | ((uint32_t)coins << F_LEFT_SHIFT);
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
  { std::vector<uint32_t> cc = h_coin_cell_; if (cc.empty()) cc.push_back(0); if ((rc = coin_cell_.upload(cc))) return rc; }
  if ((rc = state_.alloc((size_t)NW_ * bpad_))) return rc;
  if ((rc = track_.alloc((size_t)NS * bpad_))) return rc;
  return 0;
}

int BetterScrollyBackend::launch(const StepArgs& a, const pcx_buffers& out, hipStream_t s) {
  if (a.n_steps != 1) return set_error(PCX_E_INVALID, "better_scrolly backend: one step per launch");
  if (a.export_curtains) { int rc = ensure_curtains(); if (rc) return rc; }
  Ptrs P{tables_.ptr, coin_cell_.ptr, state_.ptr, track_.ptr, curtains_.ptr, batch_, bpad_};
  const int64_t groups = bpad_ / WAVE;
  // four waves share a group's render loop below this many groups per CU (measured crossover: the 45x89
  // board 2 vs 3 groups per CU, the 29x30 board 4 vs 8; profiles/r02_tuning.md)
  int coop_below = lay_.QW >= 512 ? 3 : 5;
  // windows only (fused croppers, only_crops): a launch writes ~5 KB per environment instead of 32 KB, and one wave
  // per SIMD cannot hide the window loop's latencies (SQ counters at 65,536 environments: 38 k VALU instructions per
  // wave, 1,024 waves on 1,024 SIMDs): four waves per group up to 16 groups per CU -- 0.134 -> 0.085 ms (49 % of
  // 8 TB/s; eight waves: the same), 262,144 environments 0.484 -> 0.420 ms; profiles/r03_post_kernels.md
  if (fused_.only) coop_below = 17;
  if (const char* e = getenv("PCX_COOP_BELOW")) coop_below = atoi(e);
  const bool coop = groups < (int64_t)num_cus_ * coop_below;
  size_t lds = base_lds_bytes();
  // single-wave workgroups: pad LDS so that about eight share a CU (as for scrolly_maze) -- four on the 45x89 board, whose waves
  // stream 32 KB per environment each (round 6, same box: 262,144 environments 2.18 -> 1.98 ms; 65,536: 0.509 -> 0.504; the 29x30 and
  // 29x89 boards want the eight: 0.4925 / 0.5027 and 0.6235 / 0.6390 ms; profiles/r06_tuning.md)
  int waves_per_cu = lay_.QW >= 768 ? 4 : 8;
  if (const char* e = getenv("PCX_WAVES_PER_CU")) waves_per_cu = atoi(e);
  const stream::EpilogueArgs epi_ = stream::with_hwc_scratch(this->epi_, lds, coop ? 4 : 1);  // (channels-last epilogue: its exchange area behind the kernel's own LDS)
  if (!coop && waves_per_cu > 0) {
    size_t want = ((size_t)(160 * 1024) / (size_t)waves_per_cu) & ~(size_t)255;
    if (want > 64 * 1024) want = 64 * 1024;
    if (want > lds) lds = want;
  }
  // (round 6) PCX_BS_WAVES=2: two waves share a group's render loop.  An experiment kept as a knob: it HALVES the speed (65,536
  // environments 0.509 -> 0.922 ms, 262,144: 2.18 -> 4.07) -- each wave then writes every other 256-byte piece of a plane, and what
  // the memory side is fast at is a wave that writes its planes contiguously; the four-wave cooperative shape below pays the same
  // price and is for batches that would otherwise leave CUs empty.
  bool pair = false;
  if (const char* e = getenv("PCX_BS_WAVES")) pair = atoi(e) == 2 && !coop && !epi_.out;
  bool launched = false;
#define X(r, c)                                                                                                   \
  if (!launched && R_ == r && C_ == c) {                                                                          \
    if (lds > 64 * 1024) {                                                                                        \
      PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pcx_better_scrolly_step<r, c, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pcx_better_scrolly_step<r, c, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    }                                                                                                             \
    if (lds > 64 * 1024 && epi_.out) {                                                                            \
      PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pcx_better_scrolly_step<r, c, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pcx_better_scrolly_step<r, c, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    }                                                                                                             \
    if (epi_.out && coop) hipLaunchKernelGGL((pcx_better_scrolly_step<r, c, 4, true>), dim3((unsigned)groups), dim3(4 * WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr()); \
    else if (epi_.out) hipLaunchKernelGGL((pcx_better_scrolly_step<r, c, 1, true>), dim3((unsigned)groups), dim3(WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr());       \
    else if (coop) hipLaunchKernelGGL((pcx_better_scrolly_step<r, c, 4>), dim3((unsigned)groups), dim3(4 * WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr()); \
    else if (pair) hipLaunchKernelGGL((pcx_better_scrolly_step<r, c, 2>), dim3((unsigned)groups), dim3(2 * WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr()); \
    else hipLaunchKernelGGL((pcx_better_scrolly_step<r, c, 1>), dim3((unsigned)groups), dim3(WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr());          \
    launched = true;                                                                                              \
  }
  PCX_BS_SHAPES(X)
#undef X
  if (!launched && !static_shape_) {  // the run-time-shape instance
    if (lds > 64 * 1024) return set_error(PCX_E_UNSUPPORTED, "better_scrolly backend: board too large for the step kernel's LDS tables");
    if (coop) hipLaunchKernelGGL((pcx_better_scrolly_step<0, 0, 4>), dim3((unsigned)groups), dim3(4 * WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr());
    else hipLaunchKernelGGL((pcx_better_scrolly_step<0, 0, 1>), dim3((unsigned)groups), dim3(WAVE), lds, s, k_, P, a, out, epi_, fused_.ptr());
    launched = true;
  }
  if (!launched) return set_error(PCX_E_UNSUPPORTED, "better_scrolly backend: no instance");
  PCX_HIP(hipGetLastError());
  return 0;
}

int BetterScrollyBackend::read_things(int64_t env0, int64_t n, pcx_sprite_state* sprites, uint8_t* curtains) {
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
        const uint32_t pw = word(W_POS + s, i);
        o.vrow = (int16_t)(pw & 0xFFFF); o.vcol = (int16_t)(pw >> 16);
        const bool on = o.vrow >= 0 && o.vrow < R_ && o.vcol >= 0 && o.vcol < C_;
        o.row = on ? o.vrow : 0; o.col = on ? o.vcol : 0;
        o.visible = (word(W_FLAGS, i) >> (F_SF_SHIFT + 3 * s)) & 1;
      }
    if (curtains) {
      memset(curtains + (size_t)i * lay_.cells, 0, lay_.cells);
      for (size_t id = 0; id < h_coin_cell_.size(); ++id)
        if ((word(W_COINS + (int)(id >> 5), i) >> (id & 31)) & 1) curtains[(size_t)i * lay_.cells + h_coin_cell_[id]] = 1;
    }
  }
  return 0;
}

}  // namespace bs

Backend* make_better_scrolly_backend() { return new bs::BetterScrollyBackend(); }

}  // namespace pcx
