/*
 * pcx_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT (see pcx_oracle.h).
 *
 * A literal, scalar, one-environment-at-a-time restatement of the reference
 * algorithm.  It deliberately keeps the reference's own data structures
 * (a full uint8 board, bool layers, a curtain per drape, a private copy of
 * every Scrolly pattern, a repaint after every update group) and none of the
 * tricks of the HIP path, so that the two implementations are independent.
 *
 * Each function cites the reference lines it follows (paths under pycolab/).
 * "One environment at a time" is about the algorithm: the loops over the (independent) environments of a batch are split
 * over short-lived host threads (for_envs below) so that the test suites do not wait for a single core.
 */
#ifndef _POSIX_C_SOURCE
#define _POSIX_C_SOURCE 200809L /* sysconf, pthreads under -std=c11 */
#endif
#include "pcx_oracle.h"

#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[512];
const char* pcxo_last_error(void) { return g_err; }
static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}
/* Environments are independent (engine.py:102-104: one Engine = one environment; the only random draw is counter-based), so
 * the loops over them are split over host threads: the GPU suite waits for this oracle, not for the GPU.  The threads live for
 * ONE call (created and joined inside it) -- a pool that outlives the call, OpenMP's included, does not survive fork(), and the
 * test harness forks worker processes after the oracle has been used (tests/test_gate_digests.py, oracle/ref_live.py).
 * PCX_ORACLE_THREADS caps their number (1: everything on the calling thread); batches below OX_PAR_MIN stay there anyway.
 * An error inside a worker: its message lives in that thread's g_err -- the first code is kept and its message copied over. */
#include <pthread.h>
#include <unistd.h>
#define OX_PAR_MIN 256
#define OX_MAX_THREADS 32
static pthread_mutex_t g_err_mu = PTHREAD_MUTEX_INITIALIZER;
static char g_err_shared[512];
static void keep_error(int* first, int rc) {
  if (!rc) return;
  pthread_mutex_lock(&g_err_mu);
  if (!*first) { *first = rc; snprintf(g_err_shared, sizeof g_err_shared, "%s", g_err); }
  pthread_mutex_unlock(&g_err_mu);
}
static int finish_error(int first) {
  if (first) snprintf(g_err, sizeof g_err, "%s", g_err_shared);
  return first;
}
typedef void (*ox_range_fn)(int64_t lo, int64_t hi, void* ctx);
typedef struct { ox_range_fn fn; int64_t lo, hi; void* ctx; } ox_job;
static void* ox_job_main(void* p) { ox_job* j = (ox_job*)p; j->fn(j->lo, j->hi, j->ctx); return NULL; }
static void for_envs(int64_t n, ox_range_fn fn, void* ctx) {
  int nt = 1;
  if (n >= OX_PAR_MIN) {
    long cores = sysconf(_SC_NPROCESSORS_ONLN);
    nt = cores > OX_MAX_THREADS ? OX_MAX_THREADS : cores < 1 ? 1 : (int)cores;
    const char* cap = getenv("PCX_ORACLE_THREADS");
    if (cap && atoi(cap) >= 1 && atoi(cap) < nt) nt = atoi(cap);
    if ((int64_t)nt > n / 64) nt = (int)(n / 64);
    if (nt < 1) nt = 1;
  }
  if (nt == 1) { fn(0, n, ctx); return; }
  ox_job jobs[OX_MAX_THREADS];
  pthread_t th[OX_MAX_THREADS];
  int started[OX_MAX_THREADS];
  for (int i = 0; i < nt; ++i) {
    jobs[i].fn = fn; jobs[i].ctx = ctx; jobs[i].lo = n * i / nt; jobs[i].hi = n * (i + 1) / nt;
    started[i] = i > 0 && pthread_create(&th[i], NULL, ox_job_main, &jobs[i]) == 0;
  }
  fn(jobs[0].lo, jobs[0].hi, ctx);
  for (int i = 1; i < nt; ++i) {
    if (started[i]) pthread_join(th[i], NULL);
    else fn(jobs[i].lo, jobs[i].hi, ctx);  /* (no thread to be had: the range on the calling thread) */
  }
}

/* Error bits reported per environment (the reference would have raised). */
#define OX_ERR_INDEX 1       /* numpy IndexError                        */
#define OX_ERR_SCROLL 2      /* scrolling.Error / egocentric RuntimeError */
#define OX_ERR_AFTER_OVER 4  /* play() after game_over (engine.py:622)   */

/* ---- counter-based action generator (shared definition, see pcx.h) ------ */
uint32_t pcxo_action_hash(uint64_t seed, uint64_t env, uint64_t t) {
  uint64_t x = seed ^ (env * 0x9E3779B97F4A7C15ull) ^ (t * 0xBF58476D1CE4E5B9ull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}

/* ---- per-environment state ---------------------------------------------- */

typedef struct {
  /* things.Sprite (things.py:309-319) */
  int row, col, visible;
  /* sprites.MazeWalker (sprites.py:196-204) */
  int vrow, vcol, prior_visible;
  int var[4]; /* program variables (e.g. PatrollerSprite._moving_east) */
} ox_sprite;

typedef struct {
  uint8_t* curtain; /* things.Drape.curtain, rows*cols */
  /* drapes.Scrolly (drapes.py:327-376) */
  uint8_t* pattern; /* private whole_pattern copy */
  int corner[2];
  int prescroll[2];
  int64_t last_maybe_move_frame; /* INT64_MIN = -inf */
  int var[4];
} ox_drape;

/* Entity ids inside one env: sprites 0..15, drapes 16..23. */
#define OX_DRAPE_ID(i) (PCX_MAX_SPRITES + (i))

typedef struct {
  int frame; /* plot.py:274 (starts at -1: engine.py:716 makes frame 0) */
  /* Plot._EngineDirectives (plot.py:69-104) */
  int reward_set;
  int64_t reward;
  double rewardf; /* ... where the game's rewards are Python floats (pcx_template::reward_is_float; ordeal.py:123, 187-190) */
  float discount;
  int game_over;
  /* protocols/scrolling.py state (:198-241), one set per scrolling group X;
   * every prefab entity belongs to the group its constructor named
   * (sprites.py:194, drapes.py:337) and `cur` is the group of the entity
   * whose update() is running */
  struct {
    int order_frame_valid, order_frame; /* 'scrolling_X_order_frame' */
    int order[2];                       /* 'scrolling_X_order'       */
    uint32_t egocentrists;              /* 'scrolling_X_egocentrists', bit per entity id */
  } sg[PCX_MAX_SCROLL_GROUPS];
  int cur;
  int permit_frame_valid[PCX_MAX_THINGS];
  int permit_frame[PCX_MAX_THINGS];
  uint16_t permit_mask[PCX_MAX_THINGS]; /* bit per motion index */
  /* free-form plot entries used by the shipped games */
  int64_t kv[8];
  /* Plot._EngineDirectives.z_updates (plot.py:136-174): entity ids, -1 = None */
  int n_z_updates;
  int z_move[PCX_MAX_DIRECTIVES], z_front[PCX_MAX_DIRECTIVES];
  /* Plot._next_chapter as the game's entities left it (plot.py:299-324): a chapter index,
   * PCX_CHAPTER_NONE for None, PCX_CHAPTER_UNSET while only the Story has written it */
  int32_t next_chapter;
  /* the Plot's dict entries that device programs use and a Story hands on (include/pcx.h PCX_PLOT_WORDS;
   * storytelling.py:449-450, examples/ordeal.py:121-126, 236-264) */
  int32_t pw[PCX_PLOT_WORDS];
} ox_plot;

typedef struct {
  ox_sprite sprites[PCX_MAX_SPRITES];
  ox_drape drapes[PCX_MAX_DRAPES];
  ox_plot plot;
  int game_over; /* Engine._game_over */
  int z_id[PCX_MAX_THINGS]; /* Engine._sprites_and_drapes order (engine.py:796-835 edits it) */
  int error;
  uint64_t rng_draws; /* marauders: draws so far in this env (survives resets) */
} ox_env;

struct pcxo_engine {
  pcx_template t;    /* deep copy */
  uint8_t* backdrop; /* rows*cols */
  uint8_t* init_curtain[PCX_MAX_DRAPES];
  uint8_t* init_pattern[PCX_MAX_DRAPES];
  int64_t batch;
  ox_env* envs;
  int showtime;
  /* z-order and schedule resolved to entity ids */
  int z_id[PCX_MAX_THINGS];
  int sched_id[PCX_MAX_THINGS];
  /* outputs */
  uint8_t* planes;
  int32_t* reward;
  uint8_t* reward_set;
  float* discount;
  uint8_t* done;
  int32_t* frame;
  uint8_t* error;
  int32_t* plot_in; /* [PCX_PLOT_WORDS][batch]: what an environment's next episode starts with (pcxo_engine_set_plot_words); NULL: defaults */
};

static inline int cells(const pcxo_engine* e) { return e->t.rows * e->t.cols; }
static inline uint8_t* env_board(pcxo_engine* e, int64_t b) {
  return e->planes + (size_t)b * (1 + e->t.n_chars) * cells(e);
}
static inline uint8_t* env_layer(pcxo_engine* e, int64_t b, int k) {
  return env_board(e, b) + (size_t)(1 + k) * cells(e);
}
static int char_index(const pcxo_engine* e, int ch) {
  for (int k = 0; k < e->t.n_chars; ++k)
    if (e->t.chars[k] == ch) return k;
  return -1;
}
/* entity id of the sprite/drape that paints `ch`, or -1 */
static int thing_id(const pcxo_engine* e, int ch) {
  for (int i = 0; i < e->t.n_sprites; ++i)
    if (e->t.sprites[i].ch == ch) return i;
  for (int i = 0; i < e->t.n_drapes; ++i)
    if (e->t.drapes[i].ch == ch) return OX_DRAPE_ID(i);
  return -1;
}

/* ---- Plot --------------------------------------------------------------- */

/* plot.py:343-353 _clear_engine_directives */
static void plot_clear_directives(ox_plot* p) {
  p->reward_set = 0;
  p->reward = 0;
  p->rewardf = 0.0;
  p->discount = 1.0f;
  p->game_over = 0;
  p->n_z_updates = 0;
}
/* plot.py:176-198 */
static void plot_terminate(ox_plot* p, float discount) {
  p->game_over = 1;
  p->discount = discount;
}
/* plot.py:200-226 */
static void plot_add_reward(ox_plot* p, int64_t r) {
  if (!p->reward_set) { p->reward_set = 1; p->reward = r; p->rewardf = (double)r; }
  else { p->reward += r; p->rewardf += (double)r; }
}
/* ... of a float (the sum is a Python float: a double) */
static void plot_add_rewardf(ox_plot* p, double r) {
  if (!p->reward_set) { p->reward_set = 1; p->rewardf = r; }
  else p->rewardf += r;
}

/* ---- protocols/scrolling.py ---------------------------------------------- */

static inline int motion_index(int dr, int dc) { return (dr + 1) * 3 + (dc + 1); }

/* scrolling.py:287-312 participate_as_egocentric */
static void scroll_participate(ox_plot* p, int id) { p->sg[p->cur].egocentrists |= 1u << id; }

/* scrolling.py:339-369 get_order: returns 1 and fills `order` if an order was
 * issued during the current frame. */
static int scroll_get_order(const ox_plot* p, int order[2]) {
  if (!p->sg[p->cur].order_frame_valid || p->sg[p->cur].order_frame != p->frame) return 0;
  order[0] = p->sg[p->cur].order[0];
  order[1] = p->sg[p->cur].order[1];
  return 1;
}

/* scrolling.py:372-434 permit */
static int scroll_permit(ox_plot* p, int id, uint16_t motions) {
  if (!(p->sg[p->cur].egocentrists & (1u << id))) return OX_ERR_SCROLL; /* :406-410 */
  int my_frame = p->frame + 1;                               /* :418 */
  if (!p->permit_frame_valid[id]) {                          /* setdefault :427 */
    p->permit_frame_valid[id] = 1;
    p->permit_frame[id] = my_frame;
  } else if (p->permit_frame[id] != my_frame) {
    p->permit_frame[id] = my_frame;
    p->permit_mask[id] = 0;
  }
  p->permit_mask[id] |= motions; /* :431 */
  return 0;
}

/* scrolling.py:437-485 is_possible */
static int scroll_is_possible(const ox_plot* p, int dr, int dc) {
  for (int id = 0; id < PCX_MAX_THINGS; ++id) {
    if (!(p->sg[p->cur].egocentrists & (1u << id))) continue;
    if (!p->permit_frame_valid[id] || p->permit_frame[id] != p->frame) return 0;
    if (!(p->permit_mask[id] & (1u << motion_index(dr, dc)))) return 0;
  }
  return 1;
}

/* scrolling.py:488-531 order (check_possible handled by callers) */
static int scroll_order(ox_plot* p, int dr, int dc) {
  if (p->sg[p->cur].order_frame_valid && p->sg[p->cur].order_frame == p->frame) return OX_ERR_SCROLL;
  p->sg[p->cur].order_frame_valid = 1;
  p->sg[p->cur].order_frame = p->frame;
  p->sg[p->cur].order[0] = dr;
  p->sg[p->cur].order[1] = dc;
  return 0;
}

/* ---- prefab_parts/sprites.py: MazeWalker --------------------------------- */

/* sprites.py:548-550 */
static inline int mw_on_board(const pcxo_engine* e, int r, int c) {
  return 0 <= r && r < e->t.rows && 0 <= c && c < e->t.cols;
}

/* sprites.py:315-352 _teleport (+ _on_board_exit/_enter :223-275) */
static void mw_teleport(const pcxo_engine* e, ox_sprite* s, int nr, int nc) {
  int old_on = mw_on_board(e, s->vrow, s->vcol);
  int new_on = mw_on_board(e, nr, nc);
  if (old_on && !new_on) { s->prior_visible = s->visible; s->visible = 0; }
  s->vrow = nr;
  s->vcol = nc;
  if (new_on) { s->row = nr; s->col = nc; }
  else { s->row = 0; s->col = 0; }
  if (!old_on && new_on) s->visible = s->prior_visible;
}

static inline int impassable_has(const pcx_sprite_desc* d, int ch) {
  return (d->impassable[ch >> 3] >> (ch & 7)) & 1;
}

/* sprites.py:496-511: at() + is_impassable() for one neighbour */
static int mw_blocked_at(const pcxo_engine* e, const pcx_sprite_desc* d,
                         const ox_sprite* s, const uint8_t* board, int dr, int dc) {
  int r = s->vrow + dr, c = s->vcol + dc;
  if (!mw_on_board(e, r, c)) return d->confined; /* EDGE */
  return impassable_has(d, board[r * e->t.cols + c]);
}

/* sprites.py:479-546 _check_motion: nonzero iff the motion is obstructed */
static int mw_check_motion(const pcxo_engine* e, const pcx_sprite_desc* d,
                           const ox_sprite* s, const uint8_t* board, int dr, int dc) {
  if (dr == 0 && dc == 0) return 0;
  if (dr != 0 && dc != 0) { /* diagonal :539-541 */
    /* neighbs = (left of motion vector, ahead, right of it); only whether
     * the two flanks are both impassable matters for the verdict. */
    if (mw_blocked_at(e, d, s, board, dr, dc)) return 1;
    if (mw_blocked_at(e, d, s, board, dr, 0) && mw_blocked_at(e, d, s, board, 0, dc)) return 1;
    return 0;
  }
  return mw_blocked_at(e, d, s, board, dr, dc); /* cardinal :542-543 */
}

/* sprites.py:413-454 _obey_scrolling_order */
static void mw_obey_order(const pcxo_engine* e, ox_env* env, int id, int dr, int dc) {
  const pcx_sprite_desc* d = &e->t.sprites[id];
  ox_sprite* s = &env->sprites[id];
  if (d->egocentric) scroll_participate(&env->plot, id);
  int order[2];
  if (scroll_get_order(&env->plot, order)) {
    mw_teleport(e, s, s->vrow - order[0], s->vcol - order[1]); /* _raw_move */
    if (d->egocentric && order[0] != dr && order[1] != dc) env->error |= OX_ERR_SCROLL;
  }
}

/* sprites.py:456-477 _update_scroll_permissions */
static void mw_update_permits(const pcxo_engine* e, ox_env* env, int id, const uint8_t* board) {
  const pcx_sprite_desc* d = &e->t.sprites[id];
  if (!d->egocentric) return;
  const ox_sprite* s = &env->sprites[id];
  uint16_t legal = 1u << motion_index(0, 0);
  for (int dr = -1; dr <= 1; ++dr)
    for (int dc = -1; dc <= 1; ++dc) {
      if (dr == 0 && dc == 0) continue;
      if (!mw_check_motion(e, d, s, board, dr, dc)) legal |= 1u << motion_index(dr, dc);
    }
  env->error |= scroll_permit(&env->plot, id, legal);
}

/* sprites.py:356-389 _move; returns nonzero iff obstructed */
static int mw_move(const pcxo_engine* e, ox_env* env, int id, const uint8_t* board, int dr, int dc) {
  const pcx_sprite_desc* d = &e->t.sprites[id];
  ox_sprite* s = &env->sprites[id];
  mw_obey_order(e, env, id, dr, dc);
  int blocked = mw_check_motion(e, d, s, board, dr, dc);
  if (!blocked) mw_teleport(e, s, s->vrow + dr, s->vcol + dc); /* _raw_move :391-411 */
  mw_update_permits(e, env, id, board);
  return blocked;
}

/* ---- prefab_parts/drapes.py: Scrolly -------------------------------------- */

/* drapes.py:689-695 _update_curtain.  numpy slicing clamps the slice to the
 * pattern; a window that leaves the pattern would make np.copyto raise. */
static void sc_update_curtain(const pcxo_engine* e, ox_env* env, int di) {
  const pcx_drape_desc* d = &e->t.drapes[di];
  ox_drape* s = &env->drapes[di];
  int R = e->t.rows, C = e->t.cols;
  if (s->corner[0] < 0 || s->corner[1] < 0 || s->corner[0] + R > d->pattern_rows ||
      s->corner[1] + C > d->pattern_cols) { env->error |= OX_ERR_INDEX; return; }
  for (int r = 0; r < R; ++r)
    memcpy(s->curtain + r * C,
           s->pattern + (size_t)(s->corner[0] + r) * d->pattern_cols + s->corner[1], C);
}

/* drapes.py:378-411 pattern_position_prescroll */
static void sc_pattern_position_prescroll(ox_env* env, int di, int vr, int vc, int out[2]) {
  ox_drape* s = &env->drapes[di];
  if (s->last_maybe_move_frame < env->plot.frame) { /* :407-408 */
    s->prescroll[0] = s->corner[0];
    s->prescroll[1] = s->corner[1];
  }
  out[0] = vr + s->prescroll[0];
  out[1] = vc + s->prescroll[1];
}

/* drapes.py:661-687 _sprite_burrows_into_a_margin */
static void sc_burrows(const pcxo_engine* e, const pcx_drape_desc* d, const ox_sprite* sp,
                       int dr, int dc, int* vert, int* horiz) {
  int margin_north = d->margin_rows - 1;            /* :355 */
  int margin_south = e->t.rows - d->margin_rows;    /* :356 */
  int margin_west = d->margin_cols - 1;             /* :357 */
  int margin_east = e->t.cols - d->margin_cols;     /* :358 */
  int old_r = sp->row, old_c = sp->col, new_r = old_r + dr, new_c = old_c + dc;
  *vert = (old_r > new_r && new_r <= margin_north) || (old_r < new_r && new_r >= margin_south);
  *horiz = (old_c > new_c && new_c <= margin_west) || (old_c < new_c && new_c >= margin_east);
}

/* drapes.py:487-659 _maybe_move */
static void sc_maybe_move(const pcxo_engine* e, ox_env* env, int di, int dr, int dc) {
  const pcx_drape_desc* d = &e->t.drapes[di];
  ox_drape* s = &env->drapes[di];
  ox_plot* p = &env->plot;
  int limit[2] = {d->pattern_rows - e->t.rows, d->pattern_cols - e->t.cols}; /* :342-343 */

  if (s->last_maybe_move_frame < p->frame) { /* :515-517 */
    s->last_maybe_move_frame = p->frame;
    s->prescroll[0] = s->corner[0];
    s->prescroll[1] = s->corner[1];
  }
  int order[2];
  if (scroll_get_order(p, order)) { /* :523-535 (a non-None tuple is truthy) */
    if (dr != order[0] && dc != order[1]) { env->error |= OX_ERR_SCROLL; return; }
    s->corner[0] += order[0];
    s->corner[1] += order[1];
    sc_update_curtain(e, env, di);
    return;
  }
  if (dr == 0 && dc == 0) { sc_update_curtain(e, env, di); return; } /* :539-541 */

  if (!d->have_margins) { /* case 1 :551-585 */
    if (scroll_is_possible(p, dr, dc)) {
      int north = s->corner[0] + dr, west = s->corner[1] + dc;
      int can_v = 0 <= north && north <= limit[0];
      int can_h = 0 <= west && west <= limit[1];
      int o0 = can_v ? dr : 0, o1 = can_h ? dc : 0;
      s->corner[0] += o0;
      s->corner[1] += o1;
      env->error |= scroll_order(p, o0, o1);
    }
    sc_update_curtain(e, env, di);
    return;
  }
  /* case 2 :592-659 */
  int vert = 0, horiz = 0;
  for (int id = 0; id < PCX_MAX_SPRITES; ++id) { /* only Sprites count :611 */
    if (!(p->sg[p->cur].egocentrists & (1u << id))) continue;
    int v, h;
    sc_burrows(e, d, &env->sprites[id], dr, dc, &v, &h);
    vert |= v;
    horiz |= h;
  }
  if (!(vert || horiz)) { sc_update_curtain(e, env, di); return; } /* :620-623 */
  int o0 = vert ? dr : 0, o1 = horiz ? dc : 0;
  int pr = s->corner[0] + o0, pc = s->corner[1] + o1;
  int can = 0 <= pr && pr <= limit[0];
  can &= 0 <= pc && pc <= limit[1];
  can &= scroll_is_possible(p, dr, dc); /* the *motion*, not the order :650-651 */
  if (can) {
    s->corner[0] = pr;
    s->corner[1] = pc;
    env->error |= scroll_order(p, o0, o1);
  }
  sc_update_curtain(e, env, di);
}

/* numpy 2-D indexing arr[r, c]: negative indices wrap once, else IndexError */
static int np_index(int i, int n, int* err) {
  if (i < 0) i += n;
  if (i < 0 || i >= n) { *err |= OX_ERR_INDEX; return 0; }
  return i;
}

/* ---- the nine motions ----------------------------------------------------- */
/* Order used by PCX_PROG_WALKER / PCX_PROG_SCROLLY action tables:
 * 0 N, 1 NE, 2 E, 3 SE, 4 S, 5 SW, 6 W, 7 NW, 8 STAY */
static const int MOTION9[9][2] = {{-1, 0}, {-1, 1}, {0, 1}, {1, 1}, {1, 0},
                                  {1, -1}, {0, -1}, {-1, -1}, {0, 0}};

/* ---- entity programs ------------------------------------------------------- */

typedef struct {
  pcxo_engine* e;
  ox_env* env;
  int64_t b;
  int action;
  const uint8_t* board;  /* last repaint */
} ox_ctx;

/* examples/scrolly_maze.py:259-271 PlayerSprite.update and :317-329
 * MazeDrape.update share this action table (0 N, 1 S, 2 W, 3 E, 4 stay). */
static int sm_motion(int action, int m[2]) {
  static const int T[5][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}, {0, 0}};
  if (action < 0 || action > 4) return 0;
  m[0] = T[action][0];
  m[1] = T[action][1];
  return 1;
}

static void prog_sm_player(ox_ctx* x, int id) {
  int m[2];
  if (sm_motion(x->action, m)) mw_move(x->e, x->env, id, x->board, m[0], m[1]);
}

/* examples/scrolly_maze.py:284-305 PatrollerSprite.update */
static void prog_sm_patroller(ox_ctx* x, int id) {
  pcxo_engine* e = x->e;
  ox_env* env = x->env;
  ox_sprite* s = &env->sprites[id];
  if (env->plot.frame % 2) { /* :288-290 (Python %, frame >= 0) */
    mw_move(e, env, id, x->board, 0, 0);
    return;
  }
  int walls = thing_id(e, '#') - PCX_MAX_SPRITES;
  int pp[2];
  sc_pattern_position_prescroll(env, walls, s->vrow, s->vcol, pp); /* :295-296 */
  const pcx_drape_desc* wd = &e->t.drapes[walls];
  int r = np_index(pp[0], wd->pattern_rows, &env->error);
  int c = np_index(pp[1] + (s->var[0] ? 1 : -1), wd->pattern_cols, &env->error);
  if (env->drapes[walls].pattern[(size_t)r * wd->pattern_cols + c]) s->var[0] = !s->var[0]; /* :297-299 */
  mw_move(e, env, id, x->board, 0, s->var[0] ? 1 : -1); /* :303 */
  const ox_sprite* P = &env->sprites[thing_id(e, 'P')];
  if (s->vrow == P->vrow && s->vcol == P->vcol) plot_terminate(&env->plot, 0.0f); /* :304-305 */
}

static void prog_sm_maze(ox_ctx* x, int di) {
  int m[2];
  if (sm_motion(x->action, m)) sc_maybe_move(x->e, x->env, di, m[0], m[1]);
}

/* examples/scrolly_maze.py:341-364 CashDrape.update */
static void prog_sm_cash(ox_ctx* x, int di) {
  pcxo_engine* e = x->e;
  ox_env* env = x->env;
  const pcx_drape_desc* d = &e->t.drapes[di];
  ox_drape* s = &env->drapes[di];
  const ox_sprite* P = &env->sprites[thing_id(e, 'P')];
  int pp[2];
  sc_pattern_position_prescroll(env, di, P->row, P->col, pp); /* :344-345 */
  int r = np_index(pp[0], d->pattern_rows, &env->error);
  int c = np_index(pp[1], d->pattern_cols, &env->error);
  uint8_t* cell = &s->pattern[(size_t)r * d->pattern_cols + c];
  if (*cell) { /* :347-351 */
    plot_add_reward(&env->plot, 100);
    *cell = 0;
    int any = 0;
    for (size_t i = 0; i < (size_t)d->pattern_rows * d->pattern_cols; ++i) any |= s->pattern[i];
    if (!any) plot_terminate(&env->plot, 0.0f);
  }
  int m[2];
  if (sm_motion(x->action, m)) sc_maybe_move(e, env, di, m[0], m[1]);
  else if (x->action == 5) plot_terminate(&env->plot, 0.0f); /* :363-364 */
}

/* ---- examples/better_scrolly_maze.py ----------------------------------------- */

/* better_scrolly_maze.py:258-272 PlayerSprite.update */
static void prog_bs_player(ox_ctx* x, int id) {
  int m[2];
  if (sm_motion(x->action, m)) mw_move(x->e, x->env, id, x->board, m[0], m[1]);
  if (x->action == 5) plot_terminate(&x->env->plot, 0.0f);
}

/* ---- examples/warehouse_manager.py ------------------------------------------ */

/* numpy layers[c][r, col] with Python index rules */
static int layer_at(ox_ctx* x, int ch, int r, int c) {
  pcxo_engine* e = x->e;
  int k = char_index(e, ch);
  if (k < 0) { x->env->error |= OX_ERR_INDEX; return 0; }
  int err = 0;
  r = np_index(r, e->t.rows, &err);
  c = np_index(c, e->t.cols, &err);
  if (err) { x->env->error |= err; return 0; }
  return env_layer(e, x->b, k)[r * e->t.cols + c];
}

/* layers[ch][r, c] for any character of the game (backdrop ones included) */
static int layer_char_at(ox_ctx* x, int ch, int r, int c);

/* better_scrolly_maze.py:284-301 PatrollerSprite.update */
static void prog_bs_patroller(ox_ctx* x, int id) {
  pcxo_engine* e = x->e;
  ox_env* env = x->env;
  ox_sprite* s = &env->sprites[id];
  if (env->plot.frame % 2) { mw_move(e, env, id, x->board, 0, 0); return; }
  int row = s->row, col = s->col;
  if (layer_char_at(x, '#', row, col - 1)) s->var[0] = 1;
  if (layer_char_at(x, '#', row, col + 1)) s->var[0] = 0;
  mw_move(e, env, id, x->board, 0, s->var[0] ? 1 : -1);
  const ox_sprite* P = &env->sprites[thing_id(e, 'P')];
  if (s->row == P->row && s->col == P->col) plot_terminate(&env->plot, 0.0f);
}

/* better_scrolly_maze.py:311-320 CashDrape.update */
static void prog_bs_cash(ox_ctx* x, int di) {
  pcxo_engine* e = x->e;
  ox_env* env = x->env;
  ox_drape* d = &env->drapes[di];
  const ox_sprite* P = &env->sprites[thing_id(e, 'P')];
  uint8_t* cell = &d->curtain[P->row * e->t.cols + P->col];
  if (*cell) {
    plot_add_reward(&env->plot, 100);
    *cell = 0;
    int any = 0;
    for (int i = 0; i < cells(e); ++i) any |= d->curtain[i];
    if (!any) plot_terminate(&env->plot, 0.0f);
  }
}

static int layer_char_at(ox_ctx* x, int ch, int r, int c) { return layer_at(x, ch, r, c); }

/* warehouse_manager.py:214-226 BoxSprite.update */
static void prog_wm_box(ox_ctx* x, int id) {
  const ox_sprite* s = &x->env->sprites[id];
  int r = s->row, c = s->col;
  switch (x->action) {
    case 0: if (layer_at(x, 'P', r + 1, c)) mw_move(x->e, x->env, id, x->board, -1, 0); break;
    case 1: if (layer_at(x, 'P', r - 1, c)) mw_move(x->e, x->env, id, x->board, 1, 0); break;
    case 2: if (layer_at(x, 'P', r, c + 1)) mw_move(x->e, x->env, id, x->board, 0, -1); break;
    case 3: if (layer_at(x, 'P', r, c - 1)) mw_move(x->e, x->env, id, x->board, 0, 1); break;
    default: break;
  }
}

/* warehouse_manager.py:245-266 JudgeDrape.update */
static void prog_wm_judge(ox_ctx* x, int di) {
  pcxo_engine* e = x->e;
  ox_env* env = x->env;
  ox_drape* d = &env->drapes[di];
  int n = cells(e);
  memset(d->curtain, 0, n);
  for (int ch = '0'; ch <= '9'; ++ch) { /* box chars known to the engine (:248) */
    if (char_index(e, ch) < 0) continue;
    int id = thing_id(e, ch);
    if (id < 0 || id >= PCX_MAX_SPRITES) { env->error |= OX_ERR_INDEX; continue; } /* things[c].position */
    d->curtain[env->sprites[id].row * e->t.cols + env->sprites[id].col] = 1;
  }
  int num_boxes = 0, on_goals = 0;
  for (int i = 0; i < n; ++i) num_boxes += d->curtain[i];
  for (int i = 0; i < n; ++i) { d->curtain[i] &= e->backdrop[i] == '_'; on_goals += d->curtain[i]; }
  plot_add_reward(&env->plot, on_goals - d->var[0]); /* :260 */
  d->var[0] = on_goals;
  if (x->action == 5 || on_goals == num_boxes) plot_terminate(&env->plot, 0.0f);
}

/* warehouse_manager.py:285-295 PlayerSprite.update (0 N, 1 S, 2 W, 3 E) */
static void prog_wm_player(ox_ctx* x, int id) {
  static const int T[4][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}};
  if (x->action >= 0 && x->action <= 3) mw_move(x->e, x->env, id, x->board, T[x->action][0], T[x->action][1]);
}

/* ---- examples/hello_world.py ------------------------------------------------- */

/* hello_world.py:79-91 RollingDrape.update */
static void prog_hw_rolling(ox_ctx* x, int di) {
  pcxo_engine* e = x->e;
  ox_drape* d = &x->env->drapes[di];
  int a = x->action, R = e->t.rows, C = e->t.cols;
  if (a < 0) return;                                  /* None */
  if (a == 4) plot_terminate(&x->env->plot, 0.0f);
  if (a < 4) {
    static const int AX[4] = {0, 0, 1, 1}, SH[4] = {-1, 1, -1, 1};
    uint8_t* tmp = (uint8_t*)malloc((size_t)R * C);
    for (int r = 0; r < R; ++r)
      for (int c = 0; c < C; ++c) { /* np.roll: out[(i + shift) % n] = in[i] */
        int rr = AX[a] == 0 ? ((r + SH[a]) % R + R) % R : r;
        int cc = AX[a] == 1 ? ((c + SH[a]) % C + C) % C : c;
        tmp[rr * C + cc] = d->curtain[r * C + c];
      }
    memcpy(d->curtain, tmp, (size_t)R * C);
    free(tmp);
    plot_add_reward(&x->env->plot, 1);
  }
}

/* hello_world.py:117-123 SlidingSprite.update; param[0]/[1] hold _dx/_dy as
 * four 2-bit fields (value + 1). */
static void prog_hw_sliding(ox_ctx* x, int id) {
  pcxo_engine* e = x->e;
  ox_sprite* s = &x->env->sprites[id];
  int a = x->action;
  if (a < 0 || a > 3) return;
  int dx = ((e->t.sprites[id].param[0] >> (2 * a)) & 3) - 1;
  int dy = ((e->t.sprites[id].param[1] >> (2 * a)) & 3) - 1;
  s->col = ((s->col + dx) % e->t.cols + e->t.cols) % e->t.cols;
  s->row = ((s->row + dy) % e->t.rows + e->t.rows) % e->t.rows;
  s->vrow = s->row; s->vcol = s->col;
}

/* ---- examples/extraterrestrial_marauders.py ----------------------------------- */

#define EM_BUNKER_HITTERS 0
#define EM_MARAUDER_HITTERS 1
#define EM_LAST_PLAYER_SHOT 2
#define EM_LAST_MARAUDER_SHOT 3
#define EM_NEVER INT64_MIN

/* hits = OR(layers[c] for c in chars) & curtain; curtain ^= hits; returns the
 * number of hits and the set (bit per sprite id) of characters drawn there. */
static int em_erode(ox_ctx* x, int di, const char* bolt_chars, int64_t* hitters) {
  pcxo_engine* e = x->e;
  ox_drape* d = &x->env->drapes[di];
  int n = cells(e), count = 0;
  *hitters = 0;
  for (int i = 0; i < n; ++i) {
    int bolt = 0;
    for (const char* c = bolt_chars; *c; ++c) {
      int k = char_index(e, *c);
      if (k >= 0) bolt |= env_layer(e, x->b, k)[i];
    }
    if (bolt && d->curtain[i]) {
      d->curtain[i] = 0;
      ++count;
      int id = thing_id(e, x->board[i]); /* board[hits] */
      if (id >= 0) *hitters |= (int64_t)1 << id;
    }
  }
  return count;
}

/* extraterrestrial_marauders.py:113-120 BunkerDrape.update */
static void prog_em_bunker(ox_ctx* x, int di) {
  int64_t hitters;
  int hits = em_erode(x, di, "abcdyz", &hitters);
  plot_add_reward(&x->env->plot, -hits);
  x->env->plot.kv[EM_BUNKER_HITTERS] = hitters;
}

/* extraterrestrial_marauders.py:141-163 MarauderDrape.update */
static void prog_em_marauder(ox_ctx* x, int di) {
  pcxo_engine* e = x->e;
  ox_env* env = x->env;
  ox_drape* d = &env->drapes[di];
  int R = e->t.rows, C = e->t.cols, n = R * C;
  int64_t hitters;
  int hits = em_erode(x, di, "abcd", &hitters);
  plot_add_reward(&env->plot, 10 * hits);
  env->plot.kv[EM_MARAUDER_HITTERS] = hitters;
  int total = 0, row10 = 0, edge = 0;
  for (int i = 0; i < n; ++i) total += d->curtain[i];
  if (R > 10) for (int c = 0; c < C; ++c) row10 |= d->curtain[10 * C + c];
  else env->error |= OX_ERR_INDEX;
  if (total == 0 || row10) { plot_terminate(&env->plot, 0.0f); return; } /* :151-152 */
  int period = (total - 1) / 8; /* == total // 8.0000001 for 1 <= total (:157) */
  if (period < 1) period = 1;
  if (env->plot.frame % period) return;
  for (int r = 0; r < R; ++r) edge |= d->curtain[r * C] | d->curtain[r * C + C - 1];
  uint8_t* tmp = (uint8_t*)malloc((size_t)n);
  if (edge) { /* :160-162 */
    d->var[0] = -d->var[0];
    for (int r = 0; r < R; ++r) memcpy(tmp + ((r + 1) % R) * C, d->curtain + r * C, C);
    memcpy(d->curtain, tmp, n);
  }
  int dx = d->var[0];
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) tmp[r * C + ((c + dx) % C + C) % C] = d->curtain[r * C + c];
  memcpy(d->curtain, tmp, n);
  free(tmp);
}

/* extraterrestrial_marauders.py:178-186 PlayerSprite.update */
static void prog_em_player(ox_ctx* x, int id) {
  if (x->action == 0) mw_move(x->e, x->env, id, x->board, 0, -1);
  else if (x->action == 1) mw_move(x->e, x->env, id, x->board, 0, 1);
  else if (x->action == 4) plot_terminate(&x->env->plot, 0.0f);
}

/* extraterrestrial_marauders.py:198-220 UpwardLaserBoltSprite.update */
static void prog_em_upbolt(ox_ctx* x, int id) {
  pcxo_engine* e = x->e;
  ox_env* env = x->env;
  ox_sprite* s = &env->sprites[id];
  if (s->visible) { /* _fly */
    if (((env->plot.kv[EM_BUNKER_HITTERS] | env->plot.kv[EM_MARAUDER_HITTERS]) >> id) & 1) {
      mw_teleport(e, s, -1, -1);
      return;
    }
    mw_move(e, env, id, x->board, -1, 0);
  } else if (x->action == 2) { /* _fire */
    if (env->plot.kv[EM_LAST_PLAYER_SHOT] == env->plot.frame) return;
    env->plot.kv[EM_LAST_PLAYER_SHOT] = env->plot.frame;
    const ox_sprite* P = &env->sprites[thing_id(e, 'P')];
    mw_teleport(e, s, P->row - 1, P->col);
  }
}

/* extraterrestrial_marauders.py:232-256 DownwardLaserBoltSprite.update.  The
 * reference draws from numpy's global RNG (:253); oracle, golden generator
 * and kernel all use the same counter-based draw instead:
 * choice(cols) = cols[pcx_action_hash(seed ^ EM_RNG_SALT, global_env, draw#) % len]. */
#define EM_RNG_SALT 0x4D415241554445ull
static void prog_em_downbolt(ox_ctx* x, int id) {
  pcxo_engine* e = x->e;
  ox_env* env = x->env;
  ox_sprite* s = &env->sprites[id];
  int R = e->t.rows, C = e->t.cols;
  if (s->visible) { /* _fly */
    if ((env->plot.kv[EM_BUNKER_HITTERS] >> id) & 1) { mw_teleport(e, s, -1, -1); return; }
    const ox_sprite* P = &env->sprites[thing_id(e, 'P')];
    if (s->row == P->row && s->col == P->col) plot_terminate(&env->plot, 0.0f);
    mw_move(e, env, id, x->board, 1, 0);
  } else { /* _fire */
    if (env->plot.kv[EM_LAST_MARAUDER_SHOT] == env->plot.frame) return;
    env->plot.kv[EM_LAST_MARAUDER_SHOT] = env->plot.frame;
    const uint8_t* lx = env_layer(e, x->b, char_index(e, 'X'));
    int cols[256], ncols = 0;
    for (int c = 0; c < C; ++c) {
      int any = 0;
      for (int r = 0; r < R; ++r) any |= lx[r * C + c];
      if (any) cols[ncols++] = c;
    }
    if (ncols == 0) { env->error |= OX_ERR_INDEX; return; } /* np.random.choice([]) raises */
    uint64_t seed = ((uint64_t)(uint32_t)e->t.param[0] | ((uint64_t)(uint32_t)e->t.param[1] << 32)) ^ EM_RNG_SALT;
    uint64_t genv = ((uint64_t)(uint32_t)e->t.param[2] | ((uint64_t)(uint32_t)e->t.param[3] << 32)) + (uint64_t)x->b;
    int col = cols[pcxo_action_hash(seed, genv, env->rng_draws++) % (uint32_t)ncols];
    int row = 0;
    for (int r = 0; r < R; ++r) if (lx[r * C + col]) row = r;
    mw_teleport(e, s, row + 1, col);
  }
}

/* prefab-only entities: per-entity action = (action >> param[0]) & param[1]
 * when param[1] != 0 (packed multi-agent actions), else the action itself;
 * values 0..7 index MOTION9, anything else -- None included -- is `_stay`.
 * Restates the test entities of tests/test_things.py:203-295 with integer
 * actions. */
/* ---- examples/ordeal.py ------------------------------------------------------- */
/* The Plot entries these three classes share live in plot.pw[] (include/pcx.h PCX_PLOT_OD_*); chapter codes are the
 * Story's keys sorted: 0 'castle', 1 'cavern', 2 'kansas'. */
enum { OD_CASTLE = 0, OD_CAVERN = 1, OD_KANSAS = 2 };

/* ordeal.py:121-126 SwordDrape.update */
static void prog_od_sword(ox_ctx* x, int di) {
  pcxo_engine* e = x->e;
  ox_env* env = x->env;
  ox_drape* d = &env->drapes[di];
  int pid = thing_id(e, 'P');
  if (pid < 0 || pid >= PCX_MAX_SPRITES) { env->error |= OX_ERR_INDEX; return; } /* things['P'] */
  const ox_sprite* P = &env->sprites[pid];
  if (d->curtain[P->row * e->t.cols + P->col]) { /* :122 (a confined MazeWalker's position is on the board) */
    env->plot.pw[PCX_PLOT_OD_HAS_SWORD] = 1;     /* :123 */
    plot_add_rewardf(&env->plot, 1.0);           /* :124 */
  }
  if (env->plot.pw[PCX_PLOT_OD_HAS_SWORD]) memset(d->curtain, 0, cells(e)); /* :126 */
}

/* ordeal.py:143-192 DragonduckSprite.update */
static void prog_od_dragonduck(ox_ctx* x, int id) {
  pcxo_engine* e = x->e;
  ox_env* env = x->env;
  ox_sprite* s = &env->sprites[id];
  if (env->plot.frame == 0) return; /* :144 */
  int pid = thing_id(e, 'P');
  if (pid < 0 || pid >= PCX_MAX_SPRITES) { env->error |= OX_ERR_INDEX; return; }
  const ox_sprite* P = &env->sprites[pid];
  /* :147-170: the four comparisons name one of the eight motions, or none when the two stand on the same cell */
  int dr = s->row > P->row ? -1 : s->row < P->row ? 1 : 0;
  int dc = s->col < P->col ? 1 : s->col > P->col ? -1 : 0;
  if (dr || dc) mw_move(e, env, id, x->board, dr, dc);
  if (layer_at(x, 'P', s->row, s->col)) { /* :177: the layer of the last repaint */
    env->plot.next_chapter = PCX_CHAPTER_NONE; /* :179 */
    plot_terminate(&env->plot, 0.0f);          /* :180 */
    ox_plot* p = &env->plot;
    if (p->pw[PCX_PLOT_OD_HAS_SWORD]) {        /* :182-184 */
      plot_add_rewardf(p, 1.0);
      p->z_move[p->n_z_updates] = id; p->z_front[p->n_z_updates] = pid; p->n_z_updates++;
    } else {                                   /* :186-187 */
      plot_add_rewardf(p, -1.0);
      p->z_move[p->n_z_updates] = pid; p->z_front[p->n_z_updates] = id; p->n_z_updates++;
    }
  }
}

/* ordeal.py:210-264 PlayerSprite.update; param[0]: the_plot.this_chapter as a chapter code */
static void prog_od_player(ox_ctx* x, int id) {
  pcxo_engine* e = x->e;
  ox_env* env = x->env;
  ox_sprite* s = &env->sprites[id];
  ox_plot* p = &env->plot;
  const int chap = e->t.sprites[id].param[0];
  const int lim_r = e->t.rows - 1, lim_c = e->t.cols - 1; /* :207: corner - 1 */
  const int a = x->action;
  if (a == 0) {        /* :216-221 */
    if (chap == OD_KANSAS && s->row <= 0) { p->next_chapter = OD_CASTLE; plot_terminate(p, 0.0f); }
    else mw_move(e, env, id, x->board, -1, 0);
  } else if (a == 1) { /* :223-228 */
    if (chap == OD_CASTLE && s->row >= lim_r) { p->next_chapter = OD_KANSAS; plot_terminate(p, 0.0f); }
    else mw_move(e, env, id, x->board, 1, 0);
  } else if (a == 2) { /* :230-235 */
    if (chap == OD_CAVERN && s->col <= 0) { p->next_chapter = OD_KANSAS; plot_terminate(p, 0.0f); }
    else mw_move(e, env, id, x->board, 0, -1);
  } else if (a == 3) { /* :237-242 */
    if (chap == OD_KANSAS && s->col >= lim_c) { p->next_chapter = OD_CAVERN; plot_terminate(p, 0.0f); }
    else mw_move(e, env, id, x->board, 0, 1);
  } else if (a == 4) { /* :244-246 */
    p->next_chapter = PCX_CHAPTER_NONE;
    plot_terminate(p, 0.0f);
  } else if (p->frame == 0) { /* :252-266: line up with where the last game was left */
    const int prior = p->pw[PCX_PLOT_OD_PRIOR_CHAPTER], lp = p->pw[PCX_PLOT_OD_LAST_POSITION];
    const int lr = (int16_t)(lp & 0xFFFF), lc = (int16_t)((uint32_t)lp >> 16);
    int tr = 0, tc = 0, go = 1;
    if (prior == OD_KANSAS && chap == OD_CASTLE) { tr = lim_r; tc = lc; }
    else if (prior == OD_CASTLE && chap == OD_KANSAS) { tr = 0; tc = lc; }
    else if (prior == OD_KANSAS && chap == OD_CAVERN) { tr = lr; tc = 0; }
    else if (prior == OD_CAVERN && chap == OD_KANSAS) { tr = lr; tc = lim_c; }
    else go = 0;
    if (go) {
      if (lp == -1) env->error |= OX_ERR_INDEX; /* the_plot['last_position']: KeyError */
      else mw_teleport(e, s, tr, tc);
    }
  }
  p->pw[PCX_PLOT_OD_LAST_POSITION] = (int32_t)(((uint32_t)s->row & 0xFFFFu) | ((uint32_t)s->col << 16)); /* :269 */
}

static int walker_action(int action, const int32_t* param) {
  if (action < 0) return 8;
  unsigned a = param[1] ? ((unsigned)action >> param[0]) & (unsigned)param[1] : (unsigned)action;
  return a > 8u ? 8 : (int)a;
}
static void prog_walker(ox_ctx* x, int id) {
  int a = walker_action(x->action, x->e->t.sprites[id].param);
  int blocked = mw_move(x->e, x->env, id, x->board, MOTION9[a][0], MOTION9[a][1]);
  x->env->sprites[id].var[0] = blocked; /* the_plot['walk_result_X'] truthiness */
}
static void prog_scrolly(ox_ctx* x, int di) {
  int a = walker_action(x->action, x->e->t.drapes[di].param);
  sc_maybe_move(x->e, x->env, di, MOTION9[a][0], MOTION9[a][1]);
}

/* include/pcx.h pcx_directive: a tabled entity issues, before it moves, the
 * plot directives its directive field of the action selects (what the
 * reference's tests inject with tt.pre_update, tests/engine_test.py:169-295). */
static void issue_directives(ox_ctx* x, int id, int ch, const int32_t* param) {
  pcxo_engine* e = x->e;
  if (!e->t.n_directives || x->action < 0 || !param[3]) return;
  int sel = (int)(((unsigned)x->action >> param[2]) & (unsigned)param[3]);
  if (!sel) return;
  (void)id;
  for (int i = 0; i < e->t.n_directives; ++i) {
    const pcx_directive* d = &e->t.directives[i];
    if (d->ch != ch || d->selector != sel) continue;
    ox_plot* p = &x->env->plot;
    switch (d->kind) {
      case PCX_DIR_ADD_REWARD:
        if (e->t.reward_is_float) { float f; memcpy(&f, &d->reward, 4); plot_add_rewardf(p, (double)f); }
        else plot_add_reward(p, d->reward);
        break;
      case PCX_DIR_TERMINATE: plot_terminate(p, d->discount); break;
      case PCX_DIR_NEXT_CHAPTER: p->next_chapter = d->reward; break; /* plot.py:299-324: the last call stands */
      case PCX_DIR_Z_ORDER: /* plot.py:173-174: appended, applied after the last group */
        p->z_move[p->n_z_updates] = thing_id(e, d->move_this);
        p->z_front[p->n_z_updates] = d->in_front_of ? thing_id(e, d->in_front_of) : -1;
        p->n_z_updates++;
        break;
      default: break;
    }
  }
}

static int run_program(ox_ctx* x, int id) {
  pcxo_engine* e = x->e;
  int prog = id < PCX_MAX_SPRITES ? e->t.sprites[id].program
                                  : e->t.drapes[id - PCX_MAX_SPRITES].program;
  int di = id - PCX_MAX_SPRITES;
  x->env->plot.cur = id < PCX_MAX_SPRITES ? e->t.sprites[id].scrolling_group : e->t.drapes[di].scrolling_group;
  if (x->env->plot.cur >= PCX_MAX_SCROLL_GROUPS) return -1;
  if (prog == PCX_PROG_WALKER || prog == PCX_PROG_SCROLLY || prog == PCX_PROG_STATIC) {
    if (id < PCX_MAX_SPRITES) issue_directives(x, id, e->t.sprites[id].ch, e->t.sprites[id].param);
    else issue_directives(x, id, e->t.drapes[di].ch, e->t.drapes[di].param);
  }
  switch (prog) {
    case PCX_PROG_SM_PLAYER: prog_sm_player(x, id); break;
    case PCX_PROG_SM_PATROLLER: prog_sm_patroller(x, id); break;
    case PCX_PROG_SM_MAZE: prog_sm_maze(x, di); break;
    case PCX_PROG_SM_CASH: prog_sm_cash(x, di); break;
    case PCX_PROG_BS_PLAYER: prog_bs_player(x, id); break;
    case PCX_PROG_BS_PATROLLER: prog_bs_patroller(x, id); break;
    case PCX_PROG_BS_CASH: prog_bs_cash(x, di); break;
    case PCX_PROG_WM_BOX: prog_wm_box(x, id); break;
    case PCX_PROG_WM_JUDGE: prog_wm_judge(x, di); break;
    case PCX_PROG_WM_PLAYER: prog_wm_player(x, id); break;
    case PCX_PROG_HW_ROLLING: prog_hw_rolling(x, di); break;
    case PCX_PROG_HW_SLIDING: prog_hw_sliding(x, id); break;
    case PCX_PROG_EM_PLAYER: prog_em_player(x, id); break;
    case PCX_PROG_EM_BUNKER: prog_em_bunker(x, di); break;
    case PCX_PROG_EM_MARAUDER: prog_em_marauder(x, di); break;
    case PCX_PROG_EM_UPBOLT: prog_em_upbolt(x, id); break;
    case PCX_PROG_EM_DOWNBOLT: prog_em_downbolt(x, id); break;
    case PCX_PROG_WALKER: prog_walker(x, id); break;
    case PCX_PROG_SCROLLY: prog_scrolly(x, di); break;
    case PCX_PROG_OD_PLAYER: prog_od_player(x, id); break;
    case PCX_PROG_OD_DRAGONDUCK: prog_od_dragonduck(x, id); break;
    case PCX_PROG_OD_SWORD: prog_od_sword(x, di); break;
    case PCX_PROG_STATIC: break;
    default: return -1;
  }
  return 0;
}

/* ---- rendering.py ---------------------------------------------------------- */

/* engine.py:737-759 _render with rendering.py:85-184 (occluded) or :187-301
 * (unoccluded) */
static void render(pcxo_engine* e, int64_t b) {
  ox_env* env = &e->envs[b];
  int n = cells(e), C = e->t.cols;
  uint8_t* board = env_board(e, b);
  memset(board, 0, n);           /* clear() */
  memcpy(board, e->backdrop, n); /* paint_all_of */
  int occl = e->t.occlusion_in_layers;
  if (!occl) {
    for (int k = 0; k < e->t.n_chars; ++k) { /* rendering.py:220-233 */
      uint8_t* layer = env_layer(e, b, k);
      for (int i = 0; i < n; ++i) layer[i] = e->backdrop[i] == e->t.chars[k];
    }
  }
  for (int z = 0; z < e->t.n_things; ++z) { /* engine.py:751-757 */
    int id = env->z_id[z];
    if (id < PCX_MAX_SPRITES) {
      const ox_sprite* s = &env->sprites[id];
      if (!s->visible) continue;
      board[s->row * C + s->col] = e->t.sprites[id].ch;
      if (!occl) env_layer(e, b, char_index(e, e->t.sprites[id].ch))[s->row * C + s->col] = 1;
    } else {
      const ox_drape* d = &env->drapes[id - PCX_MAX_SPRITES];
      uint8_t ch = e->t.drapes[id - PCX_MAX_SPRITES].ch;
      for (int i = 0; i < n; ++i)
        if (d->curtain[i]) board[i] = ch;
      if (!occl) memcpy(env_layer(e, b, char_index(e, ch)), d->curtain, n);
    }
  }
  if (occl) { /* rendering.py:177-179 */
    for (int k = 0; k < e->t.n_chars; ++k) {
      uint8_t* layer = env_layer(e, b, k);
      for (int i = 0; i < n; ++i) layer[i] = board[i] == e->t.chars[k];
    }
  }
}

/* ---- engine.py -------------------------------------------------------------- */

static void env_init(pcxo_engine* e, int64_t b) {
  ox_env* env = &e->envs[b];
  int n = cells(e);
  for (int i = 0; i < e->t.n_sprites; ++i) {
    const pcx_sprite_desc* d = &e->t.sprites[i];
    ox_sprite* s = &env->sprites[i];
    memset(s, 0, sizeof *s);
    s->row = d->row; s->col = d->col; s->visible = d->visible;
    s->vrow = d->vrow; s->vcol = d->vcol; s->prior_visible = d->prior_visible;
    if (d->program == PCX_PROG_SM_PATROLLER || d->program == PCX_PROG_BS_PATROLLER)
      s->var[0] = d->param[0]; /* _moving_east, scrolly_maze.py:282 / better_scrolly_maze.py:282 */
  }
  for (int i = 0; i < e->t.n_drapes; ++i) {
    const pcx_drape_desc* d = &e->t.drapes[i];
    ox_drape* s = &env->drapes[i];
    memcpy(s->curtain, e->init_curtain[i], n);
    if (d->is_scrolly)
      memcpy(s->pattern, e->init_pattern[i], (size_t)d->pattern_rows * d->pattern_cols);
    s->corner[0] = s->prescroll[0] = d->corner_row;
    s->corner[1] = s->prescroll[1] = d->corner_col;
    s->last_maybe_move_frame = INT64_MIN; /* drapes.py:373 */
    memset(s->var, 0, sizeof s->var);
    if (d->program == PCX_PROG_EM_MARAUDER) s->var[0] = d->param[0]; /* _dx, marauders.py:139 */
    if (d->program == PCX_PROG_WM_JUDGE) s->var[0] = d->param[0];    /* _last_num_boxes_on_goals :243 */
  }
  memset(&env->plot, 0, sizeof env->plot);
  env->plot.kv[EM_LAST_PLAYER_SHOT] = env->plot.kv[EM_LAST_MARAUDER_SHOT] = EM_NEVER;
  env->plot.frame = -1;
  env->plot.next_chapter = PCX_CHAPTER_UNSET; /* a new Engine has a new Plot */
  /* ... into which a Story copies the old one's entries before its_showtime() (storytelling.py:449-450) */
  env->plot.pw[PCX_PLOT_OD_LAST_POSITION] = env->plot.pw[PCX_PLOT_OD_PRIOR_CHAPTER] = -1;
  if (e->plot_in)
    for (int w = 0; w < PCX_PLOT_WORDS; ++w) env->plot.pw[w] = e->plot_in[(size_t)w * e->batch + b];
  plot_clear_directives(&env->plot);
  memcpy(env->z_id, e->z_id, sizeof env->z_id);
  env->game_over = 0;
  env->error = 0;
}

static void publish(pcxo_engine* e, int64_t b) {
  ox_env* env = &e->envs[b];
  e->reward[b] = (int32_t)env->plot.reward;
  if (e->t.reward_is_float) { float f = (float)env->plot.rewardf; memcpy(&e->reward[b], &f, 4); } /* the lane is a float32 */
  e->reward_set[b] = (uint8_t)env->plot.reward_set;
  e->discount[b] = env->plot.discount;
  e->done[b] = (uint8_t)env->game_over;
  e->frame[b] = env->plot.frame;
  e->error[b] = (uint8_t)env->error;
}

/* Batch rule (include/pcx.h, pcx_engine_step): a finished environment that is
 * stepped without auto_reset is left untouched -- the reference would raise
 * (engine.py:622-624) -- and reports an empty step: no reward, discount 0. */
static void frozen_step(pcxo_engine* e, int64_t b) {
  e->reward[b] = 0;
  e->reward_set[b] = 0;
  e->discount[b] = 0.0f;
}

/* engine.py:583-639 play (one environment) */
static int env_play(pcxo_engine* e, int64_t b, int action) {
  ox_env* env = &e->envs[b];
  env->plot.frame += 1; /* :716 */
  /* backdrop.update: base class no-op (things.py:146-147) */
  ox_ctx x = {e, env, b, action, env_board(e, b)};
  int i = 0;
  for (int g = 0; g < e->t.n_groups; ++g) { /* :726 */
    for (; i < e->t.n_things && e->t.group_of[i] == g; ++i)
      if (run_program(&x, e->sched_id[i])) return fail(PCX_E_UNSUPPORTED, "oracle: entity program not implemented");
    render(e, b); /* :735 */
  }
  /* _apply_and_clear_plot :761-847 */
  for (int u = 0; u < env->plot.n_z_updates; ++u) { /* :796-835, one directive at a time */
    int move = env->plot.z_move[u], front = env->plot.z_front[u];
    int order[PCX_MAX_THINGS], n = 0;
    if (front < 0) order[n++] = move; /* all the way to the back */
    for (int z = 0; z < e->t.n_things; ++z) {
      int id = env->z_id[z];
      if (id == move) continue;
      order[n++] = id;
      if (id == front) order[n++] = move;
    }
    memcpy(env->z_id, order, sizeof(int) * e->t.n_things);
  }
  if (env->plot.n_z_updates) render(e, b); /* engine.py:632-637 should_rerender */
  env->game_over = env->plot.game_over;
  publish(e, b);
  plot_clear_directives(&env->plot);
  return 0;
}

/* engine.py:520-581 its_showtime (one environment) */
static int env_showtime(pcxo_engine* e, int64_t b) {
  env_init(e, b);
  render(e, b);                          /* :578 */
  return env_play(e, b, PCX_ACTION_NONE); /* :581 */
}

int pcxo_engine_create(const pcx_template* t, int64_t batch, pcxo_engine** out) {
  if (!t || !out || batch <= 0) return fail(PCX_E_INVALID, "oracle: bad arguments");
  if (t->abi_version != PCX_ABI_VERSION) return fail(PCX_E_INVALID, "oracle: ABI version mismatch");
  pcxo_engine* e = (pcxo_engine*)calloc(1, sizeof *e);
  e->t = *t;
  e->batch = batch;
  int n = t->rows * t->cols;
  e->backdrop = (uint8_t*)malloc(n);
  memcpy(e->backdrop, t->backdrop, n);
  e->t.backdrop = e->backdrop;
  for (int i = 0; i < t->n_drapes; ++i) {
    const pcx_drape_desc* d = &t->drapes[i];
    e->init_curtain[i] = (uint8_t*)malloc(n);
    memcpy(e->init_curtain[i], d->curtain, n);
    e->t.drapes[i].curtain = e->init_curtain[i];
    if (d->is_scrolly) {
      size_t pn = (size_t)d->pattern_rows * d->pattern_cols;
      e->init_pattern[i] = (uint8_t*)malloc(pn);
      memcpy(e->init_pattern[i], d->pattern, pn);
      e->t.drapes[i].pattern = e->init_pattern[i];
    }
  }
  for (int z = 0; z < t->n_things; ++z) {
    e->z_id[z] = thing_id(e, t->z_order[z]);
    e->sched_id[z] = thing_id(e, t->schedule[z]);
    if (e->z_id[z] < 0 || e->sched_id[z] < 0) { pcxo_engine_destroy(e); return fail(PCX_E_INVALID, "oracle: z_order/schedule names an unknown character"); }
  }
  e->envs = (ox_env*)calloc((size_t)batch, sizeof(ox_env));
  for (int64_t b = 0; b < batch; ++b)
    for (int i = 0; i < t->n_drapes; ++i) {
      e->envs[b].drapes[i].curtain = (uint8_t*)malloc(n);
      if (t->drapes[i].is_scrolly)
        e->envs[b].drapes[i].pattern = (uint8_t*)malloc((size_t)t->drapes[i].pattern_rows * t->drapes[i].pattern_cols);
    }
  e->planes = (uint8_t*)calloc((size_t)batch * (1 + t->n_chars), n);
  e->reward = (int32_t*)calloc((size_t)batch, sizeof(int32_t));
  e->reward_set = (uint8_t*)calloc((size_t)batch, 1);
  e->discount = (float*)calloc((size_t)batch, sizeof(float));
  e->done = (uint8_t*)calloc((size_t)batch, 1);
  e->frame = (int32_t*)calloc((size_t)batch, sizeof(int32_t));
  e->error = (uint8_t*)calloc((size_t)batch, 1);
  *out = e;
  return 0;
}

void pcxo_engine_destroy(pcxo_engine* e) {
  if (!e) return;
  if (e->envs) {
    for (int64_t b = 0; b < e->batch; ++b)
      for (int i = 0; i < PCX_MAX_DRAPES; ++i) {
        free(e->envs[b].drapes[i].curtain);
        free(e->envs[b].drapes[i].pattern);
      }
    free(e->envs);
  }
  for (int i = 0; i < PCX_MAX_DRAPES; ++i) { free(e->init_curtain[i]); free(e->init_pattern[i]); }
  free(e->backdrop); free(e->planes); free(e->reward); free(e->reward_set);
  free(e->discount); free(e->done); free(e->frame); free(e->error); free(e->plot_in);
  free(e);
}

typedef struct { pcxo_engine* e; const uint8_t* mask; int first; } ox_reset_ctx;
static void ox_reset_range(int64_t lo, int64_t hi, void* p) {
  ox_reset_ctx* c = (ox_reset_ctx*)p;
  for (int64_t b = lo; b < hi; ++b)
    if (!c->mask || c->mask[b]) keep_error(&c->first, env_showtime(c->e, b));
}
int pcxo_engine_reset(pcxo_engine* e, const uint8_t* env_mask) {
  ox_reset_ctx c = {e, env_mask, 0};
  for_envs(e->batch, ox_reset_range, &c);
  if (c.first) return finish_error(c.first);
  e->showtime = 1;
  return 0;
}

/* environments [lo, hi): T steps each (an environment's steps are ordered, the environments are not); actions from the tape
 * (T == 1) or from the counter hash */
typedef struct { pcxo_engine* e; const int32_t* actions; int auto_reset; uint64_t seed; int64_t env_offset, t0; int T; int first; } ox_step_ctx;
static void ox_step_range(int64_t lo, int64_t hi, void* p) {
  ox_step_ctx* c = (ox_step_ctx*)p;
  pcxo_engine* e = c->e;
  const int n = e->t.n_actions;
  for (int64_t b = lo; b < hi; ++b)
    for (int t = 0; t < c->T && !c->first; ++t) {
      ox_env* env = &e->envs[b];
      int rc = 0;
      if (env->game_over) { if (c->auto_reset) rc = env_showtime(e, b); else frozen_step(e, b); }
      else rc = env_play(e, b, c->actions ? c->actions[b]
                                          : (int)(pcxo_action_hash(c->seed, (uint64_t)(c->env_offset + b), (uint64_t)(c->t0 + t)) % (uint32_t)n));
      keep_error(&c->first, rc);
    }
}
int pcxo_engine_step(pcxo_engine* e, const int32_t* actions, int auto_reset) {
  if (!e->showtime) return fail(PCX_E_STATE, "oracle: step before reset");
  ox_step_ctx c = {e, actions, auto_reset, 0, 0, 0, 1, 0};
  for_envs(e->batch, ox_step_range, &c);
  return finish_error(c.first);
}

int pcxo_engine_step_hashed(pcxo_engine* e, uint64_t seed, int64_t env_offset,
                            int64_t t0, int T, int auto_reset) {
  if (!e->showtime) return fail(PCX_E_STATE, "oracle: step before reset");
  ox_step_ctx c = {e, NULL, auto_reset, seed, env_offset, t0, T, 0};
  for_envs(e->batch, ox_step_range, &c);
  return finish_error(c.first);
}

int pcxo_engine_buffers(pcxo_engine* e, pcx_buffers* out) {
  out->batch = e->batch;
  out->rows = e->t.rows; out->cols = e->t.cols; out->n_chars = e->t.n_chars;
  out->planes = e->planes; out->reward = e->reward; out->reward_set = e->reward_set;
  out->discount = e->discount; out->done = e->done; out->frame = e->frame; out->error = e->error;
  return 0;
}

int pcxo_engine_plot_words(pcxo_engine* e, int32_t* out) {
  for (int w = 0; w < PCX_PLOT_WORDS; ++w)
    for (int64_t b = 0; b < e->batch; ++b) out[(size_t)w * e->batch + b] = e->envs[b].plot.pw[w];
  return 0;
}
int pcxo_engine_set_plot_words(pcxo_engine* e, const int32_t* words, const uint8_t* mask) {
  if (!e->plot_in) {
    e->plot_in = (int32_t*)calloc((size_t)e->batch * PCX_PLOT_WORDS, sizeof(int32_t));
    for (int64_t b = 0; b < e->batch; ++b)
      e->plot_in[(size_t)PCX_PLOT_OD_LAST_POSITION * e->batch + b] = e->plot_in[(size_t)PCX_PLOT_OD_PRIOR_CHAPTER * e->batch + b] = -1;
  }
  for (int w = 0; w < PCX_PLOT_WORDS; ++w)
    for (int64_t b = 0; b < e->batch; ++b)
      if (!mask || mask[b]) e->plot_in[(size_t)w * e->batch + b] = words[(size_t)w * e->batch + b];
  return 0;
}

int pcxo_engine_next_chapter(pcxo_engine* e, int32_t* out) {
  for (int64_t b = 0; b < e->batch; ++b) out[b] = e->envs[b].plot.next_chapter;
  return 0;
}

int pcxo_engine_read_things(pcxo_engine* e, int64_t env0, int64_t n,
                            pcx_sprite_state* sprites, uint8_t* curtains) {
  if (env0 < 0 || env0 + n > e->batch) return fail(PCX_E_INVALID, "oracle: env range");
  int nc = cells(e);
  for (int64_t i = 0; i < n; ++i) {
    const ox_env* env = &e->envs[env0 + i];
    if (sprites)
      for (int s = 0; s < e->t.n_sprites; ++s) {
        pcx_sprite_state* o = &sprites[i * e->t.n_sprites + s];
        memset(o, 0, sizeof *o);
        o->row = env->sprites[s].row; o->col = env->sprites[s].col;
        o->vrow = env->sprites[s].vrow; o->vcol = env->sprites[s].vcol;
        o->visible = (uint8_t)env->sprites[s].visible;
      }
    if (curtains)
      for (int d = 0; d < e->t.n_drapes; ++d)
        memcpy(curtains + ((size_t)i * e->t.n_drapes + d) * nc, env->drapes[d].curtain, nc);
  }
  return 0;
}

/* ---- croppers: see pcx_oracle_crop.c --------------------------------------- */

/* ---- accessors for pcx_oracle_crop.c ----------------------------------------- */
int pcxo__rows(const pcxo_engine* e) { return e->t.rows; }
int pcxo__cols(const pcxo_engine* e) { return e->t.cols; }
int pcxo__n_chars(const pcxo_engine* e) { return e->t.n_chars; }
int pcxo__char(const pcxo_engine* e, int k) { return e->t.chars[k]; }
int64_t pcxo__batch(const pcxo_engine* e) { return e->batch; }
const uint8_t* pcxo__planes(const pcxo_engine* e, int64_t b) { return env_board((pcxo_engine*)e, b); }
int pcxo__frame(const pcxo_engine* e, int64_t b) { return e->frame[b]; }
int pcxo__valid_char(const pcxo_engine* e, int ch) { return char_index(e, ch) >= 0; }

static int cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }
/* int(np.median(v)): mean of the two middle values for even counts, truncated */
static int median_int(int* v, int n) {
  qsort(v, n, sizeof(int), cmp_int);
  if (n & 1) return v[n / 2];
  return (int)((v[n / 2 - 1] + v[n / 2]) / 2.0);
}
/* cropping.py:551-598 _centroid */
int pcxo__centroid(const pcxo_engine* e, int64_t b, int ch, int* row, int* col) {
  int id = thing_id(e, ch);
  if (id < 0) return 0; /* the reference raises RuntimeError; callers validate */
  const ox_env* env = &e->envs[b];
  if (id < PCX_MAX_SPRITES) {
    const ox_sprite* s = &env->sprites[id];
    if (!s->visible) return 0;
    *row = s->row; *col = s->col;
    return 1;
  }
  const uint8_t* cur = env->drapes[id - PCX_MAX_SPRITES].curtain;
  int n = cells(e), m = 0;
  int* rs = (int*)malloc(sizeof(int) * n * 2);
  int* cs = rs + n;
  for (int i = 0; i < n; ++i)
    if (cur[i]) { rs[m] = i / e->t.cols; cs[m] = i % e->t.cols; ++m; }
  if (m == 0) { free(rs); return 0; }
  *row = median_int(rs, m);
  *col = median_int(cs, m);
  free(rs);
  return 1;
}
