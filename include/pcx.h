/*
 * pcx.h -- C ABI of the MI355X-native batched gridworld step engine.
 *
 * This is the drop-in boundary for pycolab's step path.  The reference has no
 * FFI of its own (it is pure Python); what a maintainer would bind is its
 * public stepping API, and every entry point below names the reference
 * interface it replaces (paths relative to the pycolab checkout):
 *
 *   pcx_engine_create   <- ascii_art.ascii_art_to_game()  pycolab/ascii_art.py:31-291
 *                          + Engine.__init__, add_X, set_X  pycolab/engine.py:98-518
 *                          (the host has already run the constructors; the
 *                          template is their result as plain data)
 *   pcx_engine_reset    <- Engine.its_showtime()          pycolab/engine.py:520-581
 *   pcx_engine_step     <- Engine.play(actions)           pycolab/engine.py:583-639
 *                          (= _update_and_render :698-735, _render :737-759,
 *                          _apply_and_clear_plot :761-847, and the entity
 *                          update() bodies of the shipped games)
 *   pcx_engine_buffers  <- the (Observation, reward, discount) triple that
 *                          play() returns              pycolab/engine.py:639,
 *                          rendering.Observation       pycolab/rendering.py:28-63,
 *                          Engine.game_over / the_plot.frame  engine.py:660, plot.py:274
 *   pcx_cropper_*       <- cropping.ObservationCropper._do_crop  pycolab/cropping.py:118-227,
 *                          FixedCropper.crop :255-268, ScrollingCropper.crop :393-426
 *   pcx_post_*          <- rendering.ObservationToArray / ObservationToFeatureArray /
 *                          ObservationCharacterRepainter  pycolab/rendering.py:304-661
 *   pcx_gather_*        <- (nothing in the reference: one Engine = one environment,
 *                          engine.py:102-104) the reward/discount/done gather of a
 *                          batch sharded over the GPUs of a node
 *
 * Conventions: every function returns 0 on success or a negative PCX_E_*
 * code and leaves a thread-local message for pcx_last_error().  Pointers in
 * signatures are plain host or device pointers plus sizes; no framework types.
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).
 * An engine is bound to one device and is not re-entrant.
 */
#ifndef PCX_H_
#define PCX_H_

#ifndef __HIPCC_RTC__ /* (a run-time build of a kernel gets the fixed-width types from csrc/pcx_device.h) */
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define PCX_ABI_VERSION 4u  /* 3: pcx_epilogue_desc grew (round 3), PCX_DIR_NEXT_CHAPTER, checkpoints carry a template hash;
                               4: pcx_template::reward_is_float / n_plot_words, pcx_engine_plot_words / pcx_engine_set_plot_words,
                                  the programs of examples/ordeal.py */

#define PCX_MAX_CHARS 32    /* distinct characters (layers) in one game   */
#define PCX_MAX_SPRITES 16
#define PCX_MAX_DRAPES 8
#define PCX_MAX_THINGS (PCX_MAX_SPRITES + PCX_MAX_DRAPES)
#define PCX_MAX_SCROLL_GROUPS 4 /* distinct scrolling_group names in one game */

/* Action value that stands for Python's `None` (frame 0, engine.py:581). */
#define PCX_ACTION_NONE (-1)

/* Error codes. */
#define PCX_OK 0
#define PCX_E_INVALID (-1)      /* bad argument / malformed template          */
#define PCX_E_UNSUPPORTED (-2)  /* template has no device program             */
#define PCX_E_HIP (-3)          /* a HIP runtime call failed                  */
#define PCX_E_STATE (-4)        /* call order violation (e.g. step before reset) */

/* Which hand-written device program steps this template's entities. */
enum pcx_game {
  PCX_GAME_SCROLLY_MAZE = 1, /* examples/scrolly_maze.py                    */
  PCX_GAME_MARAUDERS = 2,    /* examples/extraterrestrial_marauders.py      */
  PCX_GAME_WAREHOUSE = 3,    /* examples/warehouse_manager.py               */
  PCX_GAME_HELLO_WORLD = 4,  /* examples/hello_world.py                     */
  PCX_GAME_WALKERS = 5,      /* prefab-only games: MazeWalker/Scrolly with
                                action->motion tables (tests/test_things.py) */
  PCX_GAME_BETTER_SCROLLY = 6 /* examples/better_scrolly_maze.py             */
};

/* Per-entity device program ids (entity `update()` bodies). */
enum pcx_program {
  PCX_PROG_NONE = 0,
  /* examples/scrolly_maze.py */
  PCX_PROG_SM_PLAYER = 10,    /* PlayerSprite.update     :259-271 */
  PCX_PROG_SM_PATROLLER = 11, /* PatrollerSprite.update  :284-305 */
  PCX_PROG_SM_MAZE = 12,      /* MazeDrape.update        :317-329 */
  PCX_PROG_SM_CASH = 13,      /* CashDrape.update        :341-364 */
  /* examples/extraterrestrial_marauders.py */
  PCX_PROG_EM_PLAYER = 20,
  PCX_PROG_EM_BUNKER = 21,
  PCX_PROG_EM_MARAUDER = 22,
  PCX_PROG_EM_UPBOLT = 23,
  PCX_PROG_EM_DOWNBOLT = 24,
  /* examples/warehouse_manager.py */
  PCX_PROG_WM_BOX = 30,
  PCX_PROG_WM_JUDGE = 31,
  PCX_PROG_WM_PLAYER = 32,
  /* examples/hello_world.py */
  PCX_PROG_HW_ROLLING = 40,
  PCX_PROG_HW_SLIDING = 41,
  /* examples/better_scrolly_maze.py */
  PCX_PROG_BS_PLAYER = 60,    /* PlayerSprite.update    :258-272 */
  PCX_PROG_BS_PATROLLER = 61, /* PatrollerSprite.update :284-301 */
  PCX_PROG_BS_CASH = 62,      /* CashDrape.update       :311-320 */
  /* examples/ordeal.py (a Story of three games; the entities keep 'has_sword' / 'last_position' in the Plot) */
  PCX_PROG_OD_PLAYER = 70,     /* PlayerSprite.update     :210-264; param[0] = the_plot.this_chapter as a chapter code:
                                  0 'castle', 1 'cavern', 2 'kansas' (the keys sorted), -1 anything else */
  PCX_PROG_OD_DRAGONDUCK = 71, /* DragonduckSprite.update :143-192 */
  PCX_PROG_OD_SWORD = 72,      /* SwordDrape.update       :121-126 */
  /* prefab-only entities driven by an action->motion table */
  PCX_PROG_WALKER = 50,  /* MazeWalker subclass: action a -> motion table  */
  PCX_PROG_SCROLLY = 51, /* Scrolly subclass: action a -> motion table     */
  PCX_PROG_STATIC = 52   /* entity whose update() does nothing             */
};

/* A Sprite as left by its constructor (things.py:273-319; MazeWalker
 * sprites.py:153-204 when is_walker != 0). */
typedef struct pcx_sprite_desc {
  uint8_t ch;           /* character painted                                 */
  uint8_t is_walker;    /* derives from prefab_parts.sprites.MazeWalker      */
  uint8_t visible;      /* Sprite._visible after construction                */
  uint8_t prior_visible;/* MazeWalker._prior_visible (0 when None)           */
  uint8_t confined;     /* confined_to_board                                 */
  uint8_t egocentric;   /* egocentric_scroller                               */
  uint8_t scrolling_group; /* index of its scrolling_group among the game's distinct
                              group names, sorted; '' alone gives 0 (sprites.py:194) */
  uint8_t pad0;
  int32_t program;      /* enum pcx_program                                  */
  int32_t row, col;     /* true position (Sprite.position)                   */
  int32_t vrow, vcol;   /* MazeWalker virtual position                       */
  uint8_t impassable[16]; /* 128-bit set of impassable characters            */
  int32_t param[4];     /* program-specific constants                        */
} pcx_sprite_desc;

/* A Drape as left by its constructor (things.py:161-247; Scrolly
 * drapes.py:293-376 when is_scrolly != 0). */
typedef struct pcx_drape_desc {
  uint8_t ch;
  uint8_t is_scrolly;
  uint8_t have_margins;  /* scroll_margins is not None                       */
  uint8_t scrolling_group; /* same numbering as the sprites' field; drapes.py:337 */
  int32_t program;
  const uint8_t* curtain;   /* rows*cols bytes, 0/1: initial curtain          */
  /* Scrolly only: */
  const uint8_t* pattern;   /* pattern_rows*pattern_cols bytes, 0/1           */
  int32_t pattern_rows, pattern_cols;
  int32_t corner_row, corner_col;   /* board_northwest_corner                 */
  int32_t margin_rows, margin_cols; /* scroll_margins (if have_margins)       */
  int32_t param[4];
} pcx_drape_desc;

/* Plot directives a tabled entity issues from inside its update()
 * (plot.py:136-226): an entity whose program is WALKER, SCROLLY or STATIC reads
 * its "directive field" of the action, (action >> param[2]) & param[3], and
 * issues -- before it moves -- every directive of the table below that names
 * it and that value, in table order.  (The reference's tests inject the same
 * calls as Python callables, tests/engine_test.py:169-295; on this path an
 * entity's update() is a device program, so the calls are data.) */
#define PCX_MAX_DIRECTIVES 32
enum pcx_directive_kind {
  PCX_DIR_ADD_REWARD = 1, /* the_plot.add_reward(reward)              plot.py:200-226 */
  PCX_DIR_TERMINATE = 2,  /* the_plot.terminate_episode(discount)     plot.py:176-198 */
  PCX_DIR_Z_ORDER = 3,    /* the_plot.change_z_order(move_this, in_front_of)  plot.py:136-174,
                             applied by engine.py:796-835 after the last update group */
  PCX_DIR_NEXT_CHAPTER = 4 /* the_plot.next_chapter = reward  (plot.py:299-324; examples/ordeal.py:177-235:
                             a game entity names the Story's next game): `reward` is the chapter's index
                             or integer key, PCX_CHAPTER_NONE for None ("the story ends here") */
};
#define PCX_CHAPTER_NONE (-1)              /* the_plot.next_chapter = None                         */
#define PCX_CHAPTER_UNSET (-2147483647 - 1) /* no entity has set it in this episode: the Story's own */
typedef struct pcx_directive {
  uint8_t ch;          /* the entity that issues it                          */
  uint8_t kind;        /* enum pcx_directive_kind                            */
  uint8_t move_this;   /* Z_ORDER                                            */
  uint8_t in_front_of; /* Z_ORDER: a thing's character, 0 = None (to the back) */
  int32_t selector;    /* value of the entity's directive field that triggers it (> 0) */
  int32_t reward;      /* ADD_REWARD (the bits of a float32 where pcx_template::reward_is_float) */
  float discount;      /* TERMINATE, in [0, 1]                               */
} pcx_directive;

/* Plot words.  The reference's Plot is a dict that entities use as a blackboard (plot.py:27-60) and that a Story hands
 * from one game to the next (storytelling.py:449-450: `new_plot.update(old_plot)`); examples/ordeal.py:113-264 keeps
 * 'has_sword' and 'last_position' there and reads the_plot.prior_chapter.  On this path the entries device programs use
 * are PCX_PLOT_WORDS int32 words per environment.  They start every episode from what pcx_engine_set_plot_words staged
 * (zeros / "nothing there" values until it is called) and pcx_engine_plot_words reads them back -- that pair is how a
 * host-side Story carries them across a change of chapter.  The ordeal programs' words: */
#define PCX_PLOT_WORDS 4
#define PCX_PLOT_OD_HAS_SWORD 0      /* the_plot.get('has_sword'): 0 / 1                                   */
#define PCX_PLOT_OD_LAST_POSITION 1  /* the_plot['last_position']: row | col << 16, -1 while there is none */
#define PCX_PLOT_OD_PRIOR_CHAPTER 2  /* the_plot.prior_chapter as a chapter code (PCX_PROG_OD_PLAYER), -1 None */

/* A whole game as built by ascii_art_to_game(), before its_showtime(). */
typedef struct pcx_template {
  uint32_t abi_version;   /* PCX_ABI_VERSION                                  */
  int32_t game;           /* enum pcx_game                                    */
  int32_t rows, cols;
  int32_t occlusion_in_layers; /* Engine(..., occlusion_in_layers)            */
  int32_t n_chars;
  uint8_t chars[PCX_MAX_CHARS]; /* sorted; order of the layer planes          */
  const uint8_t* backdrop;      /* rows*cols bytes: Backdrop.curtain          */
  int32_t n_sprites;
  pcx_sprite_desc sprites[PCX_MAX_SPRITES];
  int32_t n_drapes;
  pcx_drape_desc drapes[PCX_MAX_DRAPES];
  int32_t n_things;
  uint8_t z_order[PCX_MAX_THINGS];   /* characters, back to front             */
  uint8_t schedule[PCX_MAX_THINGS];  /* characters in update order            */
  uint8_t group_of[PCX_MAX_THINGS];  /* update-group index of schedule[i]     */
  int32_t n_groups;
  int32_t n_actions;      /* actions 0..n_actions-1 are "ordinary" (bench/tests) */
  int32_t param[8];       /* game-specific constants                          */
  int32_t n_directives;
  pcx_directive directives[PCX_MAX_DIRECTIVES];
  int32_t reward_is_float; /* the rewards are Python floats (plot.py:200-226 sums anything `+=`-able; examples/ordeal.py:123,
                              187-190 adds +-1.0): pcx_buffers::reward then holds the bits of a float32 per environment
                              and so does pcx_directive::reward */
  int32_t n_plot_words;    /* how many plot words the template's programs use (0: none; informational)          */
} pcx_template;

/* Device (or host, for the oracle) pointers to what play() returns, batched.
 * Contents are valid until the next reset/step on the same engine -- the same
 * aliasing rule as the reference (rendering.py:55-63). */
typedef struct pcx_buffers {
  int64_t batch;
  int32_t rows, cols, n_chars;
  /* planes[b][0] = board (uint8 chars); planes[b][1+k] = layer of chars[k]
   * (uint8 0/1).  Shape [batch][1+n_chars][pitch], pitch = pcx_engine_plane_pitch()
   * (== rows*cols whenever rows*cols is a multiple of 4). */
  uint8_t* planes;
  int32_t* reward;      /* [batch] summed reward (0 when reward_set == 0); float32 bits where pcx_template::reward_is_float */
  uint8_t* reward_set;  /* [batch] 0 => the reference would return None       */
  float* discount;      /* [batch]                                             */
  uint8_t* done;        /* [batch] Engine.game_over                            */
  int32_t* frame;       /* [batch] the_plot.frame                              */
  uint8_t* error;       /* [batch] 0 ok; else the reference would have raised  */
} pcx_buffers;

/* Per-entity state readback (Engine.things[...] on the host facade). */
typedef struct pcx_sprite_state {
  int32_t row, col, vrow, vcol;
  uint8_t visible;
  uint8_t pad[3];
} pcx_sprite_state;

typedef struct pcx_engine pcx_engine;

/* engine.py:98-518 + ascii_art.py:31-291: upload shared constants, allocate
 * the SoA state of `batch` environments on `device_id`.  Template pointers
 * need only stay valid for the duration of the call. */
int pcx_engine_create(const pcx_template* t, int64_t batch, int device_id,
                      pcx_engine** out);
void pcx_engine_destroy(pcx_engine* e);

/* engine.py:520-581 its_showtime(): (re)build every environment selected by
 * env_mask_dev (device uint8[batch], NULL = all) from the template and run
 * frame 0 (play(None)). */
int pcx_engine_reset(pcx_engine* e, const uint8_t* env_mask_dev, void* stream);

/* engine.py:583-639 play(): one step of every environment.  actions_dev is a
 * device int32[batch] (PCX_ACTION_NONE = None).  With auto_reset != 0 an
 * environment whose episode is over is rebuilt and runs frame 0 instead
 * (counted as one env-step); with auto_reset == 0 its state and observation
 * are left untouched (the reference raises RuntimeError, engine.py:622-624)
 * and the step reports reward 0 / reward_set 0 / discount 0 for it, so that
 * a consumer summing rewards over the batch does not count the terminal
 * reward again; done stays 1. */
int pcx_engine_step(pcx_engine* e, const int32_t* actions_dev, int auto_reset,
                    void* stream);

/* T consecutive steps from a device action tape int32[T][batch]. Observations
 * of intermediate steps are overwritten, exactly as T calls to step would.
 * The engine may take several of the steps in one kernel launch (small
 * batches); the results are the same either way. */
int pcx_engine_step_n(pcx_engine* e, const int32_t* action_tape_dev, int T,
                      int auto_reset, void* stream);

/* T steps whose actions are generated on device by pcx_action_hash(seed,
 * global_env, t0 + t) % n_actions (global_env = env_offset + local index). */
int pcx_engine_step_hashed(pcx_engine* e, uint64_t seed, int64_t env_offset,
                           int64_t t0, int T, int auto_reset, void* stream);

int pcx_engine_buffers(pcx_engine* e, pcx_buffers* out);

/* Optional, before the first reset: make the engine write its outputs into
 * caller-owned device arrays (e.g. tensors of the host framework) instead of
 * allocating its own.  Every pointer in `ext` must be non-NULL and sized for
 * the engine's batch; batch/rows/cols/n_chars must match. */
int pcx_engine_bind_buffers(pcx_engine* e, const pcx_buffers* ext);

/* Host readback of entity state for environments [env0, env0+n): sprites as
 * pcx_sprite_state[n][n_sprites] (template order), drape curtains as
 * uint8[n][n_drapes][rows*cols].  Either pointer may be NULL. Synchronous. */
int pcx_engine_read_things(pcx_engine* e, int64_t env0, int64_t n,
                           pcx_sprite_state* sprites_host,
                           uint8_t* curtains_host);

/* "Did any environment raise?" without waiting for the device: enqueues, on
 * `stream`, a reduction of the error array and its copy into pinned host
 * memory, and stores in *seen what an EARLIER poll found (0 until one has
 * completed; nonzero = take the synchronous path and look at `error`).  The
 * cropper and post-processor polls below work the same way. */
int pcx_engine_error_poll(pcx_engine* e, void* stream, int32_t* seen);
/* Host copy of uint8[batch]: the `error` array ORed with every error bit an
 * earlier pcx_engine_error_poll saw.  An environment's error bits last until
 * its next reset -- with auto_reset that is a step or two -- so a host that
 * polls asynchronously reads the errors HERE: they stay until read with
 * clear != 0.  Synchronous. */
int pcx_engine_errors_seen(pcx_engine* e, uint8_t* errors_host, int32_t clear);

/* Checkpoint / resume.  The reference keeps an episode alive as a Python object
 * graph (an Engine is picklable: engine.py:98-246 holds plain attributes); here an
 * episode is the engine's device arrays.  export writes them into caller-owned
 * host memory of pcx_engine_state_size() bytes: the per-environment state words
 * (entity state, Plot scalars, RNG draw counters), the sprite track the
 * croppers read, and what the last play() returned besides the observation;
 * with_observation != 0 adds the observation planes (environments whose episode
 * is over and are left alone keep showing their last frame).  import restores a
 * checkpoint into an engine created from the same template with the same batch
 * (pcx_engine_reset need not have run): the steps that follow are exactly the
 * steps the exporting engine would have taken.  Synchronous. */
int pcx_engine_state_size(pcx_engine* e, int32_t with_observation, uint64_t* bytes);
int pcx_engine_export_state(pcx_engine* e, void* state_host, uint64_t bytes, int32_t with_observation);
int pcx_engine_import_state(pcx_engine* e, const void* state_host, uint64_t bytes);

/* Host copy of int32[batch]: what the environments' entities last assigned to
 * the_plot.next_chapter in their current episode (PCX_DIR_NEXT_CHAPTER; plot.py:299-324),
 * PCX_CHAPTER_NONE for None, PCX_CHAPTER_UNSET where no entity has -- a host-side Story
 * (storytelling.py:425-470) reads it when an environment's game ends.  Synchronous. */
int pcx_engine_next_chapter(pcx_engine* e, int32_t* next_host);

/* Plot words (see PCX_PLOT_WORDS above).  pcx_engine_plot_words: host copy of int32[PCX_PLOT_WORDS][batch], the words as
 * the last step left them.  pcx_engine_set_plot_words: the words environments START their next episode with, from host
 * int32[PCX_PLOT_WORDS][batch] -- only where mask_host[b] != 0 (NULL: everywhere); it takes effect at the environment's
 * next reset (pcx_engine_reset or an auto-reset) and stays until set again.  What the reference does with
 * `new_plot.update(old_plot)` before the new game's its_showtime() (storytelling.py:449-466).  Backends whose programs
 * keep nothing in the Plot answer PCX_E_UNSUPPORTED.  Synchronous. */
int pcx_engine_plot_words(pcx_engine* e, int32_t* words_host);
int pcx_engine_set_plot_words(pcx_engine* e, const int32_t* words_host, const uint8_t* mask_host);

/* Convenience synchronous copies (host <-> device) for thin FFI hosts. */
int pcx_memcpy_d2h(void* dst_host, const void* src_dev, uint64_t bytes);
int pcx_memcpy_h2d(void* dst_dev, const void* src_host, uint64_t bytes);
int pcx_device_malloc(void** out_dev, uint64_t bytes);
int pcx_device_free(void* dev);
int pcx_stream_synchronize(void* stream);
/* Measurement aid ("confirm the peak with a device microbench on the
 * box and report that measured peak too"): one launch that does nothing but
 * store `bytes` (a multiple of 4, dword-aligned) to dst_dev, a wave writing 256
 * contiguous bytes per instruction.  The caller times it on `stream`; bench.py
 * reports the step kernel against it as roofline.frac_of_achievable. */
int pcx_device_fill_probe(void* dst_dev, uint64_t bytes, void* stream);

/* The counter-based action generator shared by host, oracle and device. */
uint32_t pcx_action_hash(uint64_t seed, uint64_t env, uint64_t t);

/* Bytes between consecutive planes of one environment in `planes`: rows*cols
 * rounded up to a multiple of 4 (planes start dword-aligned; pad bytes are 0).
 * planes is [batch][1 + n_chars][pitch]; plane p of env b, cell (r, c) is at
 * ((b * (1 + n_chars) + p) * pitch + r * cols + c). */
int32_t pcx_engine_plane_pitch(const pcx_engine* e);

/* Algorithmic HBM bytes one env-step of this engine must move. */
int64_t pcx_engine_bytes_per_step(const pcx_engine* e);
/* Name of the dominant kernel, as rocprofv3 prints it. */
const char* pcx_engine_kernel_name(const pcx_engine* e);
/* Which launch shape the engine's LAST step / reset launch took (tests and benchmarks assert that the shape they
 * mean to measure is the one that ran; no reference counterpart).  pcx_scrolly_maze_step: 0 one single-wave
 * workgroup per group of 64 environments, 3 persistent workgroups of W workers (waves) that draw work units, with the
 * next unit's state words prefetched into LDS and a streaming semaphore (the default from 65,536 environments up), 5 the
 * same with the shipped level's constants compiled in (pcx_debug_scrolly_consts), 7 the same on the instance compiled for
 * the engine's own level at run time (pcx_scrolly_maze_specialise_check), 10 cooperative (several waves per
 * group), 12 the cooperative shape walking several steps per launch, 13 the persistent workers walking several steps per
 * launch (every worker keeps its units from step to step), 20 shape-generic instance, 21 the instance compiled for the engine's
 * own board shape and level at run time (1, 2, 4 and 11 were
 * launch shapes of rounds 1-4, measured slower and removed in round 5); pcx_generic_step: 30 the
 * table-driven build, 31 the build specialised for the engine's template at run time; -1: the backend does not say. */
int32_t pcx_engine_launch_shape(const pcx_engine* e);
/* The kernels with persistent workers (and pcx_generic_step's waves per workgroup) try a few equivalent launch
 * configurations on the engine's own first step launches and keep the fastest on this box (csrc/pcx_internal.h ShapeTuner:
 * 8 + 24 launches; results never depend on it).  1 once nothing is being measured any more -- settled, switched off by a
 * knob, or a launch shape without candidates; 0 while step launches still take turns.  A benchmark steps until this
 * answers 1 before it times anything (bench.py); call it after at least one step.  No reference counterpart. */
int32_t pcx_engine_tuner_done(const pcx_engine* e);
/* The table-driven kernel pcx_generic_step -- what every Engine the hand-written kernels do not cover runs on:
 * engine.py:583-847 around arbitrary Sprites / Drapes of the supported programs -- is also built per template at run time, with the
 * template's tables and schedule as compile-time constants (hiprtc; engines of PCX_GENERIC_JIT_MIN = 4,096
 * environments and more, PCX_GENERIC_JIT=0 / 1 never / always; code objects cached under $PCX_JIT_CACHE, default
 * <directory of libpcx.so>/jit_cache).  This entry plans the template and compiles that build -- or finds it in the
 * cache -- without creating an engine and WITHOUT a device: what `build` checks and the CPU tests call.  code_bytes: the
 * size of the code object; log: the compiler's words when it fails (PCX_E_UNSUPPORTED).  No reference counterpart. */
int pcx_generic_specialise_check(const pcx_template* t, char* log, int64_t log_bytes, int64_t* code_bytes);
/* The same for the kernel pcx_scrolly_maze_step, round 6: a scrolly_maze level of one's own on the example's 10x30 board with its
 * 'abcP' cast (a new entry of examples/scrolly_maze.py MAZES_ART, scrolly_maze.py:212-242) gets, at pcx_engine_create, the two
 * instances the shipped levels have in the library -- the persistent owner-code one and the cooperative small-batch one with
 * the level's constants compiled in (launch shape 7) -- and a level of ANOTHER board or cast (planes of whole dwords, one to six
 * sprites) one instance with its shape as template arguments and its constants compiled in (launch shape 21), where the library
 * has the shape-generic instances only.  Engines of PCX_SM_JIT_MIN = 4,096 environments and more; PCX_SM_JIT=0 / 1 never /
 * always; the same cache.  This entry plans the template and compiles that build without an engine and WITHOUT a device.
 * PCX_E_UNSUPPORTED with an empty log: a shipped level (its instances are part of the library), a template of another game,
 * a board the static-shape code does not take.  No reference counterpart. */
int pcx_scrolly_maze_specialise_check(const pcx_template* t, char* log, int64_t log_bytes, int64_t* code_bytes);
/* Build-time aid (no reference counterpart): pcx_scrolly_maze_step exists once more per shipped level with the constants of
 * that level (examples/scrolly_maze.py, MAZES_ART[0..2]) compiled in -- csrc/pcx_sm_shipped.h, generated by
 * tools/gen_sm_shipped.py from what this entry answers.  It plans `t` for that kernel WITHOUT a device and copies the
 * kernel's constants for work units of `unit` environments (64, 32, 16; 0: as the cooperative shape takes them) as 32-bit words; returns their number (words ==
 * NULL: only that), or a negative PCX_E_* when the kernel does not take the template.  An engine runs the baked
 * instance only while its own constants equal the header's, word for word (pcx_engine_launch_shape 5). */
int64_t pcx_debug_scrolly_consts(const pcx_template* t, int32_t unit, uint32_t* words, int64_t cap);
/* Profiling aid (no reference counterpart): the phase timers the last launch left when the backend was asked to keep
 * them (pcx_scrolly_maze_step's persistent shapes under PCX_SM_PROF=1: 16 words per workgroup, 10 ns ticks; layout in
 * pcx_scrolly_maze.hip Ptrs::ps_prof).  out_host == NULL with words == -1 clears them.  Synchronous.  PCX_E_UNSUPPORTED from
 * backends that keep none. */
int pcx_engine_debug_counters(pcx_engine* e, uint32_t* out_host, int64_t words);

const char* pcx_last_error(void);
uint32_t pcx_abi_version(void);

/* ------------------------------------------------------------------------ */
/* Croppers: cropping.py.  A cropper owns per-environment window state.      */

enum pcx_cropper_kind {
  PCX_CROP_FIXED = 1,     /* cropping.FixedCropper      :230-268 */
  PCX_CROP_SCROLLING = 2  /* cropping.ScrollingCropper  :271-598 */
};

typedef struct pcx_cropper_desc {
  int32_t kind;
  int32_t rows, cols;          /* window size                               */
  int32_t top, left;           /* FixedCropper top_left_corner              */
  int32_t pad_char;            /* -1 = None                                 */
  int32_t n_track;             /* ScrollingCropper to_track                 */
  uint8_t to_track[PCX_MAX_THINGS];
  int32_t margin_rows, margin_cols; /* resolved scroll_margins (never None) */
  int32_t initial_offset_rows, initial_offset_cols;
  int32_t saccade;
} pcx_cropper_desc;

typedef struct pcx_cropper pcx_cropper;

int pcx_cropper_create(pcx_engine* e, const pcx_cropper_desc* d,
                       pcx_cropper** out);
void pcx_cropper_destroy(pcx_cropper* c);
/* cropping.py:393-426 (or :255-268): crop the engine's current observation.
 * Environments whose engine was reset this step restart their window. */
int pcx_cropper_crop(pcx_cropper* c, void* stream);
/* planes: uint8 [batch][1+n_chars][pitch] of the cropped window, pitch =
 * pcx_cropper_plane_pitch() = rows*cols rounded up to a multiple of 4 (pad
 * bytes are 0) -- the cropper's own array, or the one the caller bound;
 * corner: int32 [batch][2] window corner (scrolling croppers). */
int pcx_cropper_buffers(pcx_cropper* c, uint8_t** planes_dev,
                        int32_t** corner_dev);
int32_t pcx_cropper_plane_pitch(const pcx_cropper* c);
/* Checkpoint / resume of a cropper (companion of pcx_engine_export_state; no reference counterpart: the reference's
 * croppers are pickled with the game).  The window state -- every environment's corner and whether it has one yet,
 * cropping.py:393-426 -- and, with_planes != 0, the cropped planes the last crop() (or fused step) produced.  Import
 * AFTER pcx_engine_import_state, into a cropper of the same window on an engine of the same game and batch; with the
 * planes the next crop() hands them out as they are, without them it cuts them from the engine's restored
 * observation (PCX_E_STATE if the engine writes no full-board planes).  Synchronous. */
int pcx_cropper_state_size(pcx_cropper* c, int32_t with_planes, uint64_t* bytes);
int pcx_cropper_export_state(pcx_cropper* c, void* host, uint64_t bytes, int32_t with_planes);
int pcx_cropper_import_state(pcx_cropper* c, const void* host, uint64_t bytes);
/* Optional: make crop() write into a caller-owned device array (e.g. a tensor
 * of the host framework) of batch * (1+n_chars) * pitch bytes, dword-aligned;
 * the reference's croppers likewise write a pre-allocated output
 * (cropping.py:131-134).  No copy, no synchronisation on the way out. */
int pcx_cropper_bind_output(pcx_cropper* c, uint8_t* planes_dev);
/* Fused croppers.  The step kernel holds the frame it paints in LDS, so it can
 * cut the croppers' windows from it directly: after this call every
 * pcx_engine_reset / pcx_engine_step of `e` also moves the windows of these
 * croppers (ScrollingCropper.crop, cropping.py:393-426, once per step -- i.e.
 * the caller crops every observation, as human_ui.py:269-293 does) and writes
 * their output planes, in the same launch; pcx_cropper_crop() on a fused
 * cropper then launches nothing.  `croppers`: n <= 4 croppers of this engine,
 * fixed or scrolling after at most four entities each: sprites, drapes (the
 * median of the raw curtain, cropping.py:590-598) and priority lists that
 * mix the two.  Drape trackers: the table-driven kernel on any board, the
 * hand-written kernels on boards of at most 63 x 128 cells (PCX_E_UNSUPPORTED
 * beyond).  only_crops != 0: the full-board
 * planes are no longer written (pcx_buffers.planes goes stale; the consumer
 * ingests the windows only).  n == 0 releases the croppers again.  When the
 * engine is already in play the croppers are brought up to date once, on
 * `stream`.  PCX_E_UNSUPPORTED (nothing changed) where the engine's kernel
 * cannot do it (occlusion_in_layers=False, an installed feature-array
 * epilogue, a drape tracker where the line above says so): the croppers then
 * run as their own kernels. */
int pcx_engine_fuse_croppers(pcx_engine* e, pcx_cropper* const* croppers,
                             int32_t n, int32_t only_crops, void* stream);
/* Device uint8[batch] behind pcx_cropper_errors (read it on the caller's stream). */
int pcx_cropper_error_buffer(pcx_cropper* c, const uint8_t** errors_dev);
int pcx_cropper_error_poll(pcx_cropper* c, void* stream, int32_t* seen);
/* Host copy of uint8[batch]: 1 where the reference would raise RuntimeError
 * (window leaves the observation and there is no pad character,
 * cropping.py:175-183).  Synchronous. */
int pcx_cropper_errors(pcx_cropper* c, uint8_t* errors_host);

/* ------------------------------------------------------------------------ */
/* Observation post-processors: rendering.py:304-661.  They read a planes array
 * (an engine's or a cropper's) and write their own device output.           */

typedef struct pcx_planes_view {
  const uint8_t* planes;  /* [batch][1 + n_chars][pitch], plane 0 = board    */
  int64_t batch;
  int32_t rows, cols, pitch, n_chars;
  uint8_t chars[PCX_MAX_CHARS];
} pcx_planes_view;

int pcx_engine_planes_view(pcx_engine* e, pcx_planes_view* out);
int pcx_cropper_planes_view(pcx_cropper* c, pcx_planes_view* out);

enum pcx_post_kind {
  PCX_POST_TO_ARRAY = 1,       /* rendering.ObservationToArray            :409-542 */
  PCX_POST_FEATURE_ARRAY = 2,  /* rendering.ObservationToFeatureArray     :545-661 */
  PCX_POST_REPAINT = 3         /* rendering.ObservationCharacterRepainter :304-406 */
};
enum pcx_post_dtype { PCX_U8 = 1, PCX_I32 = 2, PCX_F32 = 3, PCX_I64 = 4, PCX_F64 = 5 };

#define PCX_POST_MAX_DEPTH 32

typedef struct pcx_post_desc {
  int32_t kind;
  int32_t dtype;   /* TO_ARRAY: element type of the output (FEATURE_ARRAY is f32,
                      REPAINT is u8)                                          */
  int32_t depth;   /* TO_ARRAY: length of the value vectors (1 for scalars);
                      FEATURE_ARRAY: number of layers; REPAINT: output chars  */
  /* TO_ARRAY: lut[d][ch] = d-th component of value_mapping[chr(ch)], as the
   * raw little-endian bytes of `dtype` in a 64-bit cell; mapped[ch] != 0 iff
   * chr(ch) is a key.  REPAINT: lut[0][ch] = repainted character.            */
  uint64_t lut[PCX_POST_MAX_DEPTH][128];
  uint8_t mapped[128];
  /* FEATURE_ARRAY: the characters whose layers are stacked (a character the
   * observation does not have yields zeros).  REPAINT: the output layer
   * characters, in plane order.                                              */
  uint8_t chars[PCX_POST_MAX_DEPTH];
  /* output element (d, r, c) of environment b lives at index
   * b * depth * rows * cols + d * stride[0] + r * stride[1] + c * stride[2]
   * (np.transpose(..., permute) of the reference, made contiguous).          */
  int64_t stride[3];
} pcx_post_desc;

typedef struct pcx_post pcx_post;

int pcx_post_create(const pcx_planes_view* src, const pcx_post_desc* d, int device_id, pcx_post** out);
void pcx_post_destroy(pcx_post* p);
int pcx_post_run(pcx_post* p, void* stream);
/* Fused epilogue: rendering.ObservationToFeatureArray in its
 * default axis order written by the step kernel's own render loop -- the layer
 * masks are in registers there -- into a caller-owned float32 array
 * [batch][depth][rows*cols] (16-byte aligned), every step from the next one on.
 * chars: the stacked layers' characters (distinct; a character the game does
 * not have leaves its plane as the caller initialised it: zeros).
 * skip_layers != 0: the uint8 layer planes of `planes` are no longer written
 * (the board plane is) -- for consumers that only ingest the feature array;
 * skip_layers == 2: nor is the board plane (pcx_buffers.planes goes stale
 * altogether: the consumer ingests the epilogue's array and nothing else).
 * Answers PCX_E_UNSUPPORTED where the backend's render loop cannot do it
 * (occlusion_in_layers=False, fused croppers in the same kernel; the
 * table-driven kernel writes this planar float32 array and none of the other
 * kinds below): run pcx_post_* then.  Boards of any size (a plane's last dword is
 * stored cell by cell when rows*cols is no multiple of 4).
 * A NULL desc clears the epilogue. */
typedef struct pcx_epilogue_desc {
  int32_t depth;
  uint8_t chars[PCX_POST_MAX_DEPTH];
  float* out_dev;
  int32_t skip_layers;
  /* != 0: channels last -- out_dev is [batch][rows*cols][depth], what
   * ObservationToFeatureArray(permute=(1, 2, 0)) returns (rendering.py:545-661).
   * Needs rows*cols % 4 == 0 (PCX_E_UNSUPPORTED otherwise). */
  int32_t channels_last;
  /* != 0: rendering.ObservationToArray (rendering.py:409-542, default axis
   * order) instead of the feature array: out_dev is [batch][depth][rows*cols]
   * elements of `dtype` (enum pcx_post_dtype), element (d, cell) =
   * lut[d][board character]; `lut` and `mapped` are laid out like the fields
   * of the same names in struct pcx_post_desc and live in HOST memory (the
   * call copies them).  Every character of the game must be mapped
   * (the reference would raise on the first one that is not,
   * rendering.py:503-507) and rows*cols % 4 == 0; PCX_E_UNSUPPORTED otherwise.
   * With skip_layers the step writes the board and this array only.
   * to_array == 2: rendering.ObservationCharacterRepainter (rendering.py:304-406):
   * out_dev is a planes array uint8 [batch][1 + depth][rows*cols] -- plane 0 the
   * board repainted through lut[0], plane 1 + k the layer of chars[k] (the
   * repainted observation's characters) -- i.e. what pcx_post_* writes for
   * PCX_POST_REPAINT; dtype is PCX_U8.                                         */
  int32_t to_array;
  int32_t dtype;
  const uint64_t* lut;    /* [depth][128] */
  const uint8_t* mapped;  /* [128] */
} pcx_epilogue_desc;
int pcx_engine_set_epilogue(pcx_engine* e, const pcx_epilogue_desc* d);
/* Crop, THEN post-process, in the step kernel's launch -- the order the reference's own pipeline has
 * (human_ui.py:252-265 crop_and_repaint; better_scrolly_maze.py:237-247 followed by rendering.py:545-661): a cropper
 * that is fused into its engine's step kernel (pcx_engine_fuse_croppers) also writes the float32 feature stack of ITS
 * WINDOW -- out_dev [batch][depth][rows*cols] of the window (channels_last: [batch][rows*cols][depth]), layer k = window
 * board == chars[k] -- from the window's board dword while it is in a register.  skip_layers 1: the window's uint8
 * layer planes are no longer written, 2: nor its board plane (with only_crops the launch then writes
 * rows*cols*depth*4 bytes per environment and nothing else of the observation).  depth <= 16; to_array / lut unused.
 * d == NULL clears.  PCX_E_STATE: the cropper is not fused; PCX_E_UNSUPPORTED: the table-driven kernel (its window
 * loop is its own), a layer stacked twice.  The stack stays attached while the cropper is fused. */
int pcx_cropper_set_features(pcx_cropper* c, const pcx_epilogue_desc* d);

/* out_dev: TO_ARRAY/FEATURE_ARRAY [batch][depth*rows*cols] elements; REPAINT a
 * planes array [batch][1 + depth][pitch], pitch = pcx_post_plane_pitch() =
 * rows*cols rounded up to a multiple of 4 (pad bytes are 0). */
int pcx_post_output(pcx_post* p, void** out_dev, uint64_t* bytes);
int32_t pcx_post_plane_pitch(const pcx_post* p);
/* Optional: write into a caller-owned, 16-byte aligned device array of exactly
 * the size pcx_post_output reports (a tensor of the host framework: the
 * observation-to-tensor hand-off is then zero-copy). */
int pcx_post_bind_output(pcx_post* p, void* out_dev, uint64_t bytes);
/* Device uint8[batch] behind pcx_post_errors (read it on the caller's stream). */
int pcx_post_error_buffer(pcx_post* p, const uint8_t** errors_dev);
int pcx_post_error_poll(pcx_post* p, void* stream, int32_t* seen);
/* Host copy of uint8[batch]: 1 where a board character had no mapping
 * (rendering.py:503-507 RuntimeError) or was not ASCII.  Synchronous. */
int pcx_post_errors(pcx_post* p, uint8_t* errors_host);

/* ------------------------------------------------------------------------ */
/* Node-level gather of the step results: environments never
 * interact (one Engine per environment, pycolab/engine.py:102-104), so a batch
 * sharded over the GPUs of a node needs no data-path collective; the one
 * optional exchange is this gather of what play() returns besides the
 * observation -- reward i32, discount f32, reward_set u8, done u8 = 10 bytes
 * per environment (engine.py:639; plot.py:69-104).  For a host that drives
 * every GPU from ONE process: `engines` = n engines on n distinct devices;
 * create() builds one RCCL communicator per device (ncclCommInitAll; RCCL is
 * bound with dlopen at this call, so a process that never gathers never loads
 * it).  (One-process-per-GPU hosts gather the same packed record with their
 * framework's collective: pycolab_amd.distributed.ScalarGather.)              */
typedef struct pcx_gather pcx_gather;
int pcx_gather_create(pcx_engine* const* engines, int32_t n, pcx_gather** out);
void pcx_gather_destroy(pcx_gather* g);
/* One ncclAllGather per device, grouped, enqueued on streams[i] (hipStream_t of
 * engine i's device; NULL array = default streams) after whatever the engine
 * last launched there; asynchronous.  Engine i contributes the record
 * [reward i32[B_i] | discount f32[B_i] | reward_set u8[B_i] | done u8[B_i]]
 * zero-padded to `slot` = 10 * max_i(B_i) rounded up to 16 bytes (sent in place
 * when the engine's outputs were bound as exactly that one allocation).       */
int pcx_gather_scalars(pcx_gather* g, void* const* streams);
/* Where engine i's device received everybody's records: uint8 [n][slot]; the
 * record of engine r starts at r * slot and is laid out with ITS B_r.          */
int pcx_gather_buffers(pcx_gather* g, int32_t i, uint8_t** recv_dev, int64_t* slot_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PCX_H_ */
