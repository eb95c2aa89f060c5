// pcx_internal.h -- host-side plumbing shared by the C ABI and the game
// backends.  One backend = one shipped game family = one fused step kernel.
#pragma once

#include "pcx_device.h"

#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "pcx_crop_window.h"

namespace pcx {

int set_error(int code, const char* fmt, ...);

#define PCX_HIP(call)                                                          \
  do {                                                                         \
    hipError_t err__ = (call);                                                 \
    if (err__ != hipSuccess)                                                   \
      return ::pcx::set_error(PCX_E_HIP, "%s failed: %s (%s:%d)", #call,       \
                              hipGetErrorString(err__), __FILE__, __LINE__);   \
  } while (0)

// Device-resident copy of a byte array.
template <typename T>
struct DevArray {
  T* ptr = nullptr;
  size_t count = 0;
  ~DevArray() { if (ptr) (void)hipFree(ptr); }
  int alloc(size_t n) {
    count = n;
    PCX_HIP(hipMalloc(reinterpret_cast<void**>(&ptr), (n ? n : 1) * sizeof(T)));
    PCX_HIP(hipMemset(ptr, 0, (n ? n : 1) * sizeof(T)));
    return 0;
  }
  int upload(const std::vector<T>& host) {
    int rc = alloc(host.size());
    if (rc) return rc;
    if (!host.empty())
      PCX_HIP(hipMemcpy(ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
  }
};

// "Did any environment raise?" without a host synchronisation: poll() enqueues
// a reduction of a device uint8[n] error array into one word and its copy into
// pinned host memory, and reports what an EARLIER poll found (0 until one has
// completed).  The caller that sees nonzero then takes the synchronous path.
struct ErrorPoll {
  uint32_t* dev = nullptr;
  uint32_t* host = nullptr;
  // every poll also ORs the error array into this one: an error bit lives only
  // until its environment's next reset (auto-reset clears it within a step or
  // two), the copy here stays until the host has read it (errors_seen)
  uint8_t* sticky = nullptr;
  int64_t sticky_n = 0;
  ~ErrorPoll();
  int poll(const uint8_t* errors_dev, int64_t n, hipStream_t s, int32_t* seen);
  // host copy of (sticky | live); clear != 0 forgets the sticky part.  Synchronous.
  int errors_seen(const uint8_t* errors_dev, int64_t n, uint8_t* out_host, int clear);
};

// Which of a few equivalent launch configurations is fastest ON THIS BOX, measured on the engine's own launches (as
// GenericBackend::Tuner does for pcx_generic_step's waves per workgroup).  The persistent shape's best worker / slot counts
// differ from box to box by more than they differ from each other (profiles/r05_tuning.md: 2 x 3 workers with private
// slots 0.560 ms on one box where 4 x 1 with two shared slots gives 0.574, 0.569 / 0.562 on another, 0.617 / 0.592 (2 x 2)
// on a third), and the result does not depend on the choice.  After WARM step launches on the default, the candidates take
// turns in BLOCKS of three consecutive launches, twice round (a clock still ramping up after the engine's creation must
// not favour whoever is measured last); a block is timed over its second and third launch only -- single launches timed
// between neighbours of another shape overlap with those neighbours' tails and measured up to 20 % off the steady state
// (r05_ps_sweep_call7_pruned_tuner.txt: 0.068 against 0.086 ms).  Once the last block has completed (polled, never waited
// for) the candidate with the smallest time stays -- the default unless another beats it by 1.5 %.  Launches under stream
// capture and reset launches leave the tuner alone; the backends turn it off (`off = true`) when a knob fixes the shape
// (PCX_SM_WAVES / _PER_CU / _LOCK ..., PCX_SM_TUNE=0 / PCX_WM_TUNE=0 / PCX_HW_TUNE=0).  Shared by pcx_scrolly_maze_step's
// persistent shape and the persistent workers of pcx_warehouse_step / pcx_hello_world_step.
struct ShapeTuner {
  static constexpr int WARM = 8, NC = 4, ROUNDS = 2, BLOCK = 3, NB = NC * ROUNDS;
  int phase = 0, chosen = -1;
  bool off = false, measuring_begin = false, measuring_end = false, have_events = false;
  hipEvent_t ev[NB][2] = {};
  float ms[NC] = {};
  ~ShapeTuner() { drop(); }
  void drop() {
    if (have_events) for (auto& e : ev) { (void)hipEventDestroy(e[0]); (void)hipEventDestroy(e[1]); }
    have_events = false;
  }
  // the candidate this launch takes (0 = the default)
  int pick(const StepArgs& a, hipStream_t s) {
    measuring_begin = measuring_end = false;
    if (chosen >= 0) return chosen;
    if (off || a.mode != 0 || a.n_steps > 1) {  // (launches of several steps are not comparable with single steps)
      // ... and a launch the tuner does not own in the MIDDLE of a block would be charged to the block's candidate:
      // the block starts over with its first launch
      const int i = phase - WARM;
      if (i >= 0 && i < NB * BLOCK) phase = WARM + (i / BLOCK) * BLOCK;
      return 0;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return 0; }
    const int i = phase - WARM;
    if (i < 0) { ++phase; return 0; }
    if (i < NB * BLOCK) {
      if (!have_events) {
        for (auto& e : ev)
          if (hipEventCreate(&e[0]) != hipSuccess || hipEventCreate(&e[1]) != hipSuccess) { (void)hipGetLastError(); off = true; return 0; }
        have_events = true;
      }
      const int block = i / BLOCK, pos = i % BLOCK;
      if (pos == 1 && hipEventRecord(ev[block][0], s) != hipSuccess) { (void)hipGetLastError(); off = true; return 0; }
      measuring_begin = true;
      measuring_end = pos == BLOCK - 1;
      return block % NC;
    }
    if (hipEventQuery(ev[NB - 1][1]) != hipSuccess) { (void)hipGetLastError(); return 0; }  // (not through yet: the default meanwhile)
    float best = 0.0f;
    for (int c = 0; c < NC; ++c) {
      ms[c] = 0.0f;
      for (int r = 0; r < ROUNDS; ++r) {
        float t = 0.0f;
        if (hipEventElapsedTime(&t, ev[r * NC + c][0], ev[r * NC + c][1]) != hipSuccess) { (void)hipGetLastError(); off = true; drop(); return 0; }
        ms[c] += t / ((BLOCK - 1) * ROUNDS);
      }
      if (chosen < 0 || ms[c] < best * 0.985f) { best = ms[c]; chosen = c; }
    }
    drop();
    const char* dbg = getenv("PCX_DEBUG");
    if (dbg && (atoi(dbg) & 16))
      fprintf(stderr, "[pcx] launch shape candidate %d of %d (%.4f %.4f %.4f %.4f ms per launch)\n", chosen, NC, ms[0], ms[1], ms[2], ms[3]);
    return chosen;
  }
  // nothing left to measure: settled, switched off, or never consulted (the launch shapes that have no candidates)
  bool done() const { return off || chosen >= 0 || phase == 0; }
  void launched(hipStream_t s) {
    if (!measuring_begin) return;
    if (measuring_end && hipEventRecord(ev[(phase - WARM) / BLOCK][1], s) != hipSuccess) { (void)hipGetLastError(); off = true; }
    ++phase;
    measuring_begin = measuring_end = false;
  }
};

namespace stream { struct EpilogueArgs; }
class Backend {
 public:
  virtual ~Backend() {}
  // include/pcx.h pcx_engine_tuner_done: no launch of this engine measures launch shapes any more
  virtual int tuner_done() const { return 1; }
  // Validate the template and upload constants / allocate state.
  virtual int init(const pcx_template& t, int64_t batch) = 0;
  virtual int launch(const StepArgs& a, const pcx_buffers& out, hipStream_t s) = 0;
  // how many consecutive steps one launch() may take (StepArgs::n_steps)
  virtual int max_fused_steps() const { return 1; }
  virtual int read_things(int64_t env0, int64_t n, pcx_sprite_state* sprites,
                          uint8_t* curtains) = 0;
  virtual int64_t bytes_per_step() const = 0;
  virtual const char* kernel_name() const = 0;
  virtual int launch_shape() const { return -1; }  // include/pcx.h pcx_engine_launch_shape
  virtual int read_debug_counters(uint32_t* out_host, int64_t words);  // include/pcx.h pcx_engine_debug_counters
  // Sprite state for croppers: device int32 [n_sprites][batch] packed
  // (row | col << 8 | visible << 16), refreshed by every launch.
  virtual const int32_t* sprite_track() const { return nullptr; }
  // Raw (pre-occlusion) curtains as flat cell-bit vectors, uint32 [n_drapes][curtain_words()][bpad],
  // template drape order; refreshed by launches with StepArgs::export_curtains.
  virtual const uint32_t* curtain_bits() const { return nullptr; }
  // allocates the curtain export buffer if this backend has not yet (a
  // drape-tracking cropper attached after its_showtime(): pcx_crop.hip)
  virtual int ensure_curtains() { return 0; }
  virtual int curtain_words() const { return 0; }
  virtual int64_t batch_pad() const = 0;
  // include/pcx.h pcx_engine_next_chapter: device int32[bpad] the entities' the_plot.next_chapter
  // assignments live in, or null where the backend's entities never assign it
  virtual const int32_t* next_chapter_words() const { return nullptr; }
  // include/pcx.h pcx_engine_plot_words: device int32 [PCX_PLOT_WORDS][batch_pad()], null where the programs keep nothing in the Plot
  virtual const int32_t* plot_words() const { return nullptr; }
  virtual int set_plot_words(const int32_t* words_host, const uint8_t* mask_host);  // include/pcx.h pcx_engine_set_plot_words
  // include/pcx.h pcx_engine_export_state: every device array of the backend that carries an
  // episode from one launch to the next (state words incl. RNG counters, the croppers' sprite track)
  virtual void persistent_arrays(std::vector<std::pair<void*, size_t>>& out) = 0;
  // bytes between consecutive planes of one environment (>= rows*cols, multiple of 4)
  virtual int plane_pitch() const = 0;
  // include/pcx.h pcx_engine_set_epilogue: null clears.  Backends whose render
  // loop cannot produce it answer PCX_E_UNSUPPORTED (the caller then runs the
  // post-processor as its own kernel).
  virtual int set_epilogue(const pcx_epilogue_desc* d);
  // include/pcx.h pcx_engine_fuse_croppers: the step kernel moves these windows
  // and writes their planes itself; null (or n == 0) clears.  Backends whose
  // kernel cannot answer PCX_E_UNSUPPORTED (the croppers then run as their own
  // kernels, pcx_crop.hip).
  virtual int set_fused_croppers(const crop::FusedCrops* fc);
  // include/pcx.h pcx_cropper_set_features: does the kernel's window loop also write a window's float32 feature stack?
  virtual bool fused_window_features() const { return false; }
  // the installed epilogue's kernel arguments (the engine hangs the ObservationToArray value table in), null: none
  virtual stream::EpilogueArgs* epilogue_args() { return nullptr; }
};

Backend* make_scrolly_maze_backend();
Backend* make_generic_backend();
int generic_specialise_check(const pcx_template& t, char* log, int64_t log_bytes, int64_t* code_bytes);  // include/pcx.h pcx_generic_specialise_check
int scrolly_maze_specialise_check(const pcx_template& t, char* log, int64_t log_bytes, int64_t* code_bytes);  // include/pcx.h pcx_scrolly_maze_specialise_check
int64_t scrolly_maze_consts(const pcx_template& t, int32_t unit, uint32_t* words, int64_t cap);  // include/pcx.h pcx_debug_scrolly_consts
Backend* make_warehouse_backend();  // hand-written; init() answers PCX_E_UNSUPPORTED for templates it leaves to the table-driven kernel
Backend* make_marauders_backend();
Backend* make_better_scrolly_backend();
Backend* make_hello_world_backend();

}  // namespace pcx

struct pcx_engine {
  pcx_template t;  // shallow copy; pointer members are NOT valid after create
  uint64_t template_hash = 0;  // FNV-1a over everything of the template, the arrays behind its pointers included (checkpoints)
  int64_t batch = 0;
  int device = 0;
  bool showtime = false;
  pcx::Backend* backend = nullptr;
  pcx_buffers out{};       // where the kernels write (own or bound)
  bool own_out = false;
  uint64_t epoch = 0;      // bumped by every reset/step (croppers)
  pcx::ErrorPoll error_poll;
  bool want_curtains = false;   // a drape-tracking cropper exists
  bool curtains_fresh = false;  // the last launch exported curtains
  std::vector<struct pcx_cropper*> fused;  // croppers the step kernel runs itself (pcx_engine_fuse_croppers)
  bool fused_only = false;                 // ... and the full-board planes are no longer written
  void* epilogue_lut = nullptr;            // device copy of the ObservationToArray epilogue's value table (pcx_engine_set_epilogue)
};
