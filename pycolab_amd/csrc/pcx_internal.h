// pcx_internal.h -- host-side plumbing shared by the C ABI and the game
// backends.  One backend = one shipped game family = one fused step kernel.
#pragma once

#include "pcx_device.h"

#include <cstdio>
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

namespace stream { struct EpilogueArgs; }
class Backend {
 public:
  virtual ~Backend() {}
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
