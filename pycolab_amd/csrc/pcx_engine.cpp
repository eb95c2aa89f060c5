// pcx_engine.cpp -- the C ABI of include/pcx.h over the game backends.
#include "pcx_internal.h"
#include "pcx_stream.h"

#include <cstdarg>
#include <cstdlib>
#include <cstring>

namespace pcx {

static thread_local char g_error[1024] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof g_error, fmt, ap);
  va_end(ap);
  return code;
}

static uint32_t action_hash_host(uint64_t seed, uint64_t env, uint64_t t) {
  uint64_t x = seed ^ (env * 0x9E3779B97F4A7C15ull) ^ (t * 0xBF58476D1CE4E5B9ull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}

static int debug_flags() {  // (read at every launch: tools/ps_sweep.py alternates ablations on one engine)
  const char* e = getenv("PCX_DEBUG");
  return e ? atoi(e) : 0;
}

static void free_own_outputs(pcx_engine* e) {
  if (!e->own_out) return;
  (void)hipFree(e->out.planes); (void)hipFree(e->out.reward); (void)hipFree(e->out.reward_set);
  (void)hipFree(e->out.discount); (void)hipFree(e->out.done); (void)hipFree(e->out.frame);
  (void)hipFree(e->out.error);
  e->own_out = false;
  memset(&e->out, 0, sizeof e->out);
}

static int ensure_outputs(pcx_engine* e) {
  if (e->out.planes) return 0;
  size_t B = (size_t)e->batch;
  size_t plane_bytes = B * (size_t)(1 + e->t.n_chars) * (size_t)e->backend->plane_pitch();
  pcx_buffers& o = e->out;
  o.batch = e->batch; o.rows = e->t.rows; o.cols = e->t.cols; o.n_chars = e->t.n_chars;
  PCX_HIP(hipMalloc((void**)&o.planes, plane_bytes));
  PCX_HIP(hipMalloc((void**)&o.reward, B * 4));
  PCX_HIP(hipMalloc((void**)&o.reward_set, B));
  PCX_HIP(hipMalloc((void**)&o.discount, B * 4));
  PCX_HIP(hipMalloc((void**)&o.done, B));
  PCX_HIP(hipMalloc((void**)&o.frame, B * 4));
  PCX_HIP(hipMalloc((void**)&o.error, B));
  PCX_HIP(hipMemset(o.planes, 0, plane_bytes));
  PCX_HIP(hipMemset(o.reward, 0, B * 4));
  PCX_HIP(hipMemset(o.reward_set, 0, B));
  PCX_HIP(hipMemset(o.discount, 0, B * 4));
  PCX_HIP(hipMemset(o.done, 0, B));
  PCX_HIP(hipMemset(o.frame, 0, B * 4));
  PCX_HIP(hipMemset(o.error, 0, B));
  e->own_out = true;
  return 0;
}

__global__ void pcx_any_nonzero(const uint8_t* v, int64_t n, uint32_t* flag, uint8_t* sticky) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i >= n) return;
  uint32_t any = 0;
  if (i + 16 <= n && (reinterpret_cast<uintptr_t>(v) & 15u) == 0) {
    const uint4 w = *reinterpret_cast<const uint4*>(v + i);
    any = w.x | w.y | w.z | w.w;
  } else {
    for (int64_t j = i; j < n && j < i + 16; ++j) any |= v[j];
  }
  if (any) {
    atomicOr(flag, 1u);
    for (int64_t j = i; j < n && j < i + 16; ++j)  // rare: keep which environments, past their auto-reset
      if (v[j]) sticky[j] |= v[j];
  }
}

// include/pcx.h pcx_device_fill_probe: the plainest full-chip store stream (a wave
// writes 256 contiguous bytes per instruction, grid-stride), i.e. what this box's
// HBM takes from a kernel that does nothing but store
__global__ __launch_bounds__(256) void pcx_fill_probe(uint32_t* p, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = (uint32_t)i;
}

int Backend::set_epilogue(const pcx_epilogue_desc* d) {
  if (!d) return 0;
  return set_error(PCX_E_UNSUPPORTED, "%s has no fused feature-array epilogue", kernel_name());
}

int Backend::set_plot_words(const int32_t*, const uint8_t*) {
  return set_error(PCX_E_UNSUPPORTED, "pcx_engine_set_plot_words: this template's programs keep nothing in the Plot");
}
int Backend::set_fused_croppers(const crop::FusedCrops* fc) {
  if (!fc || fc->n <= 0) return 0;
  return set_error(PCX_E_UNSUPPORTED, "%s cannot run croppers itself", kernel_name());
}
int Backend::read_debug_counters(uint32_t*, int64_t) {
  return set_error(PCX_E_UNSUPPORTED, "%s keeps no debug counters", kernel_name());
}

ErrorPoll::~ErrorPoll() {
  if (dev) (void)hipFree(dev);
  if (host) (void)hipHostFree(host);
  if (sticky) (void)hipFree(sticky);
}

int ErrorPoll::errors_seen(const uint8_t* errors_dev, int64_t n, uint8_t* out_host, int clear) {
  PCX_HIP(hipDeviceSynchronize());
  PCX_HIP(hipMemcpy(out_host, errors_dev, (size_t)n, hipMemcpyDeviceToHost));
  if (sticky && sticky_n == n) {
    std::vector<uint8_t> st((size_t)n);
    PCX_HIP(hipMemcpy(st.data(), sticky, (size_t)n, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) out_host[i] |= st[i];
    if (clear) PCX_HIP(hipMemset(sticky, 0, (size_t)n));
  }
  return 0;
}

int ErrorPoll::poll(const uint8_t* errors_dev, int64_t n, hipStream_t s, int32_t* seen) {
  if (!dev) {
    PCX_HIP(hipMalloc(reinterpret_cast<void**>(&dev), 4));
    PCX_HIP(hipHostMalloc(reinterpret_cast<void**>(&host), 4, hipHostMallocDefault));
    *host = 0;
  }
  if (!sticky || sticky_n != n) {
    if (sticky) (void)hipFree(sticky);
    sticky = nullptr;
    PCX_HIP(hipMalloc(reinterpret_cast<void**>(&sticky), (size_t)(n ? n : 1)));
    PCX_HIP(hipMemset(sticky, 0, (size_t)(n ? n : 1)));
    sticky_n = n;
  }
  if (seen) *seen = (int32_t)*reinterpret_cast<volatile uint32_t*>(host);
  PCX_HIP(hipMemsetAsync(dev, 0, 4, s));
  const int64_t threads = (n + 15) / 16;
  hipLaunchKernelGGL(pcx_any_nonzero, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, errors_dev, n, dev, sticky);
  PCX_HIP(hipGetLastError());
  PCX_HIP(hipMemcpyAsync(host, dev, 4, hipMemcpyDeviceToHost, s));
  return 0;
}

}  // namespace pcx

using pcx::set_error;

extern "C" {

uint32_t pcx_abi_version(void) { return PCX_ABI_VERSION; }
const char* pcx_last_error(void) { return pcx::g_error; }
uint32_t pcx_action_hash(uint64_t seed, uint64_t env, uint64_t t) { return pcx::action_hash_host(seed, env, t); }

int pcx_engine_create(const pcx_template* t, int64_t batch, int device_id, pcx_engine** out) {
  if (!t || !out || batch <= 0) return set_error(PCX_E_INVALID, "pcx_engine_create: bad arguments");
  if (t->abi_version != PCX_ABI_VERSION) return set_error(PCX_E_INVALID, "pcx_engine_create: ABI version mismatch");
  if (t->n_chars <= 0 || t->n_chars > PCX_MAX_CHARS || t->n_sprites < 0 || t->n_sprites > PCX_MAX_SPRITES ||
      t->n_drapes < 0 || t->n_drapes > PCX_MAX_DRAPES || t->n_things != t->n_sprites + t->n_drapes ||
      t->rows <= 0 || t->cols <= 0 || !t->backdrop)
    return set_error(PCX_E_INVALID, "pcx_engine_create: malformed template");
  PCX_HIP(hipSetDevice(device_id));
  pcx::Backend* b = nullptr;
  // a float32 reward lane (pcx_template::reward_is_float) is the table-driven kernel's: the hand-written kernels add integers
  if (t->reward_is_float && t->game == PCX_GAME_SCROLLY_MAZE)
    return set_error(PCX_E_UNSUPPORTED, "pcx_engine_create: pcx_scrolly_maze_step has no float32 reward lane (reward_is_float)");
  switch (t->reward_is_float ? PCX_GAME_WALKERS : t->game) {
    case PCX_GAME_SCROLLY_MAZE: b = pcx::make_scrolly_maze_backend(); break;
    // hand-written kernels for the shipped shapes; anything else of the same
    // game (other boards, occlusion_in_layers=False) takes the table-driven one
    case PCX_GAME_WAREHOUSE: b = pcx::make_warehouse_backend(); break;
    case PCX_GAME_MARAUDERS: b = pcx::make_marauders_backend(); break;
    case PCX_GAME_BETTER_SCROLLY: b = pcx::make_better_scrolly_backend(); break;
    case PCX_GAME_HELLO_WORLD: b = pcx::make_hello_world_backend(); break;
    case PCX_GAME_WALKERS: b = pcx::make_generic_backend(); break;
    default:
      return set_error(PCX_E_UNSUPPORTED, "pcx_engine_create: no device program for game id %d", t->game);
  }
  int rc = b->init(*t, batch);
  if (rc == PCX_E_UNSUPPORTED && (t->game == PCX_GAME_WAREHOUSE || t->game == PCX_GAME_MARAUDERS || t->game == PCX_GAME_BETTER_SCROLLY ||
                                 t->game == PCX_GAME_HELLO_WORLD)) {
    delete b;
    b = pcx::make_generic_backend();
    rc = b->init(*t, batch);
  }
  if (rc) { delete b; return rc; }
  pcx_engine* e = new pcx_engine();
  e->t = *t;
  {
    // what makes two engines interchangeable for a checkpoint: the whole template, pointer targets by content
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { const uint8_t* b = static_cast<const uint8_t*>(p); for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
    pcx_template flat = *t;
    flat.backdrop = nullptr;
    for (int i = 0; i < PCX_MAX_DRAPES; ++i) flat.drapes[i].curtain = flat.drapes[i].pattern = nullptr;
    mix(&flat, sizeof flat);
    mix(t->backdrop, (size_t)t->rows * t->cols);
    for (int i = 0; i < t->n_drapes; ++i) {
      if (t->drapes[i].curtain) mix(t->drapes[i].curtain, (size_t)t->rows * t->cols);
      if (t->drapes[i].pattern) mix(t->drapes[i].pattern, (size_t)t->drapes[i].pattern_rows * t->drapes[i].pattern_cols);
    }
    e->template_hash = h;
  }
  e->t.backdrop = nullptr;
  for (int i = 0; i < PCX_MAX_DRAPES; ++i) e->t.drapes[i].curtain = e->t.drapes[i].pattern = nullptr;
  e->batch = batch;
  e->device = device_id;
  e->backend = b;
  *out = e;
  return 0;
}

void pcx_engine_destroy(pcx_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  delete e->backend;
  if (e->epilogue_lut) (void)hipFree(e->epilogue_lut);
  pcx::free_own_outputs(e);
  delete e;
}

int pcx_engine_bind_buffers(pcx_engine* e, const pcx_buffers* ext) {
  if (!e || !ext) return set_error(PCX_E_INVALID, "pcx_engine_bind_buffers: bad arguments");
  if (e->showtime) return set_error(PCX_E_STATE, "pcx_engine_bind_buffers: must precede the first reset");
  if (ext->batch != e->batch || ext->rows != e->t.rows || ext->cols != e->t.cols || ext->n_chars != e->t.n_chars)
    return set_error(PCX_E_INVALID, "pcx_engine_bind_buffers: shape mismatch");
  if (!ext->planes || !ext->reward || !ext->reward_set || !ext->discount || !ext->done || !ext->frame || !ext->error)
    return set_error(PCX_E_INVALID, "pcx_engine_bind_buffers: every output pointer must be non-NULL");
  pcx::free_own_outputs(e);
  e->out = *ext;
  return 0;
}

int pcx_engine_buffers(pcx_engine* e, pcx_buffers* out) {
  if (!e || !out) return set_error(PCX_E_INVALID, "pcx_engine_buffers: bad arguments");
  PCX_HIP(hipSetDevice(e->device));
  int rc = pcx::ensure_outputs(e);
  if (rc) return rc;
  *out = e->out;
  return 0;
}

int pcx_engine_reset(pcx_engine* e, const uint8_t* env_mask_dev, void* stream) {
  if (!e) return set_error(PCX_E_INVALID, "pcx_engine_reset: null engine");
  PCX_HIP(hipSetDevice(e->device));
  int rc = pcx::ensure_outputs(e);
  if (rc) return rc;
  if (!e->showtime && env_mask_dev)
    return set_error(PCX_E_STATE, "pcx_engine_reset: the first reset must cover every environment");
  pcx::StepArgs a;
  a.mode = 1;
  a.reset_mask = env_mask_dev;
  a.export_curtains = e->want_curtains;
  rc = e->backend->launch(a, e->out, (hipStream_t)stream);
  if (rc) return rc;
  e->showtime = true;
  e->epoch++;
  e->curtains_fresh = e->want_curtains && !env_mask_dev;
  return 0;
}

int pcx_engine_step(pcx_engine* e, const int32_t* actions_dev, int auto_reset, void* stream) {
  if (!e || !actions_dev) return set_error(PCX_E_INVALID, "pcx_engine_step: bad arguments");
  if (!e->showtime) return set_error(PCX_E_STATE, "pcx_engine_step: call pcx_engine_reset first (its_showtime)");
  PCX_HIP(hipSetDevice(e->device));
  pcx::StepArgs a;
  a.actions = actions_dev;
  a.auto_reset = auto_reset;
  a.export_curtains = e->want_curtains;
  a.debug = pcx::debug_flags();
  int rc = e->backend->launch(a, e->out, (hipStream_t)stream);
  if (rc) return rc;
  e->epoch++;
  e->curtains_fresh = e->want_curtains;
  return 0;
}

int pcx_engine_step_n(pcx_engine* e, const int32_t* action_tape_dev, int T, int auto_reset, void* stream) {
  if (!e || !action_tape_dev || T < 0) return set_error(PCX_E_INVALID, "pcx_engine_step_n: bad arguments");
  if (!e->showtime) return set_error(PCX_E_STATE, "pcx_engine_step_n: call pcx_engine_reset first (its_showtime)");
  PCX_HIP(hipSetDevice(e->device));
  const int fuse = e->backend->max_fused_steps();
  for (int t = 0; t < T;) {
    const int n = T - t < fuse ? T - t : fuse;
    pcx::StepArgs a;
    a.actions = action_tape_dev + (size_t)t * e->batch;
    a.n_steps = n; a.action_stride = e->batch;
    a.auto_reset = auto_reset;
    a.export_curtains = e->want_curtains;
    a.debug = pcx::debug_flags();
    int rc = e->backend->launch(a, e->out, (hipStream_t)stream);
    if (rc) return rc;
    e->epoch += n;
    e->curtains_fresh = e->want_curtains;
    t += n;
  }
  return 0;
}

int pcx_engine_step_hashed(pcx_engine* e, uint64_t seed, int64_t env_offset, int64_t t0, int T, int auto_reset,
                           void* stream) {
  if (!e || T < 0) return set_error(PCX_E_INVALID, "pcx_engine_step_hashed: bad arguments");
  if (!e->showtime) return set_error(PCX_E_STATE, "pcx_engine_step_hashed: call pcx_engine_reset first");
  PCX_HIP(hipSetDevice(e->device));
  const int fuse = e->backend->max_fused_steps();
  for (int t = 0; t < T;) {
    const int n = T - t < fuse ? T - t : fuse;
    pcx::StepArgs a;
    a.hashed = 1; a.seed = seed; a.env_offset = env_offset; a.t = t0 + t; a.auto_reset = auto_reset;
    a.n_steps = n;
    a.export_curtains = e->want_curtains;
    a.debug = pcx::debug_flags();
    int rc = e->backend->launch(a, e->out, (hipStream_t)stream);
    if (rc) return rc;
    e->epoch += n;
    e->curtains_fresh = e->want_curtains;
    t += n;
  }
  return 0;
}

int pcx_engine_error_poll(pcx_engine* e, void* stream, int32_t* seen) {
  if (!e || !e->out.error) return set_error(PCX_E_INVALID, "pcx_engine_error_poll: bad arguments");
  PCX_HIP(hipSetDevice(e->device));
  return e->error_poll.poll(e->out.error, e->batch, (hipStream_t)stream, seen);
}

int pcx_engine_next_chapter(pcx_engine* e, int32_t* next_host) {
  if (!e || !next_host) return set_error(PCX_E_INVALID, "pcx_engine_next_chapter: bad arguments");
  PCX_HIP(hipSetDevice(e->device));
  const int32_t* words = e->backend->next_chapter_words();
  if (!words) {
    for (int64_t i = 0; i < e->batch; ++i) next_host[i] = PCX_CHAPTER_UNSET;
    return 0;
  }
  PCX_HIP(hipDeviceSynchronize());
  PCX_HIP(hipMemcpy(next_host, words, (size_t)e->batch * 4, hipMemcpyDeviceToHost));
  return 0;
}

int pcx_engine_plot_words(pcx_engine* e, int32_t* words_host) {
  if (!e || !words_host) return set_error(PCX_E_INVALID, "pcx_engine_plot_words: bad arguments");
  PCX_HIP(hipSetDevice(e->device));
  const int32_t* words = e->backend->plot_words();
  if (!words) return set_error(PCX_E_UNSUPPORTED, "pcx_engine_plot_words: this template's programs keep nothing in the Plot");
  PCX_HIP(hipDeviceSynchronize());
  PCX_HIP(hipMemcpy2D(words_host, (size_t)e->batch * 4, words, (size_t)e->backend->batch_pad() * 4, (size_t)e->batch * 4, PCX_PLOT_WORDS,
                      hipMemcpyDeviceToHost));
  return 0;
}

int pcx_engine_set_plot_words(pcx_engine* e, const int32_t* words_host, const uint8_t* mask_host) {
  if (!e || !words_host) return set_error(PCX_E_INVALID, "pcx_engine_set_plot_words: bad arguments");
  PCX_HIP(hipSetDevice(e->device));
  return e->backend->set_plot_words(words_host, mask_host);
}

int pcx_engine_errors_seen(pcx_engine* e, uint8_t* errors_host, int32_t clear) {
  if (!e || !e->out.error || !errors_host) return set_error(PCX_E_INVALID, "pcx_engine_errors_seen: bad arguments");
  PCX_HIP(hipSetDevice(e->device));
  return e->error_poll.errors_seen(e->out.error, e->batch, errors_host, clear);
}

// ---- checkpoint / resume -------------------------------------------------------------
namespace {
struct StateHeader {
  uint32_t magic, abi;
  int32_t game, rows, cols, n_chars, n_sprites, n_drapes;
  int64_t batch;
  uint64_t epoch;
  int32_t with_observation, n_arrays;
  uint64_t template_hash;  // pcx_engine::template_hash: level, tables, directives and parameters, not only the dimensions
};
constexpr uint32_t STATE_MAGIC = 0x53584350u;  // "PCXS"

// the arrays of a checkpoint, in order: backend arrays, then what play() last returned
int state_arrays(pcx_engine* e, int with_obs, std::vector<std::pair<void*, size_t>>& arr) {
  int rc = pcx::ensure_outputs(e);
  if (rc) return rc;
  e->backend->persistent_arrays(arr);
  const size_t B = (size_t)e->batch;
  arr.push_back({e->out.reward, B * 4}); arr.push_back({e->out.reward_set, B}); arr.push_back({e->out.discount, B * 4});
  arr.push_back({e->out.done, B}); arr.push_back({e->out.frame, B * 4}); arr.push_back({e->out.error, B});
  if (with_obs) arr.push_back({e->out.planes, B * (size_t)(1 + e->t.n_chars) * (size_t)e->backend->plane_pitch()});
  return 0;
}
}  // namespace

int pcx_engine_state_size(pcx_engine* e, int32_t with_observation, uint64_t* bytes) {
  if (!e || !bytes) return set_error(PCX_E_INVALID, "pcx_engine_state_size: bad arguments");
  PCX_HIP(hipSetDevice(e->device));
  std::vector<std::pair<void*, size_t>> arr;
  int rc = state_arrays(e, with_observation, arr);
  if (rc) return rc;
  uint64_t n = sizeof(StateHeader);
  for (auto& a : arr) n += 8 + ((a.second + 7) & ~(size_t)7);
  *bytes = n;
  return 0;
}

int pcx_engine_export_state(pcx_engine* e, void* host, uint64_t bytes, int32_t with_observation) {
  if (!e || !host) return set_error(PCX_E_INVALID, "pcx_engine_export_state: bad arguments");
  if (!e->showtime) return set_error(PCX_E_STATE, "pcx_engine_export_state: the engine is not in play (its_showtime)");
  uint64_t need = 0;
  int rc = pcx_engine_state_size(e, with_observation, &need);
  if (rc) return rc;
  if (bytes < need) return set_error(PCX_E_INVALID, "pcx_engine_export_state: %llu bytes given, %llu needed",
                                     (unsigned long long)bytes, (unsigned long long)need);
  std::vector<std::pair<void*, size_t>> arr;
  if ((rc = state_arrays(e, with_observation, arr))) return rc;
  PCX_HIP(hipDeviceSynchronize());
  StateHeader h{STATE_MAGIC, PCX_ABI_VERSION, e->t.game, e->t.rows, e->t.cols, e->t.n_chars, e->t.n_sprites, e->t.n_drapes,
                e->batch, e->epoch, with_observation != 0, (int32_t)arr.size(), e->template_hash};
  uint8_t* p = static_cast<uint8_t*>(host);
  memcpy(p, &h, sizeof h); p += sizeof h;
  for (auto& a : arr) {
    const uint64_t n = a.second;
    memcpy(p, &n, 8); p += 8;
    PCX_HIP(hipMemcpy(p, a.first, a.second, hipMemcpyDeviceToHost));
    p += (a.second + 7) & ~(size_t)7;
  }
  return 0;
}

int pcx_engine_import_state(pcx_engine* e, const void* host, uint64_t bytes) {
  if (!e || !host || bytes < sizeof(StateHeader)) return set_error(PCX_E_INVALID, "pcx_engine_import_state: bad arguments");
  PCX_HIP(hipSetDevice(e->device));
  StateHeader h;
  memcpy(&h, host, sizeof h);
  if (h.magic != STATE_MAGIC || h.abi != PCX_ABI_VERSION) return set_error(PCX_E_INVALID, "pcx_engine_import_state: not a checkpoint of this ABI");
  if (h.game != e->t.game || h.rows != e->t.rows || h.cols != e->t.cols || h.n_chars != e->t.n_chars ||
      h.n_sprites != e->t.n_sprites || h.n_drapes != e->t.n_drapes || h.batch != e->batch)
    return set_error(PCX_E_INVALID, "pcx_engine_import_state: the checkpoint is of another game, board or batch");
  if (h.template_hash != e->template_hash)
    return set_error(PCX_E_INVALID, "pcx_engine_import_state: the checkpoint is of another level or parameter set of this game (template hash)");
  std::vector<std::pair<void*, size_t>> arr;
  int rc = state_arrays(e, h.with_observation, arr);
  if (rc) return rc;
  if ((int32_t)arr.size() != h.n_arrays) return set_error(PCX_E_INVALID, "pcx_engine_import_state: array count mismatch");
  const uint8_t* p = static_cast<const uint8_t*>(host) + sizeof h;
  const uint8_t* const end = static_cast<const uint8_t*>(host) + bytes;
  PCX_HIP(hipDeviceSynchronize());
  for (auto& a : arr) {
    uint64_t n = 0;
    if (p + 8 > end) return set_error(PCX_E_INVALID, "pcx_engine_import_state: truncated checkpoint");
    memcpy(&n, p, 8); p += 8;
    if (n != a.second || p + n > end)
      return set_error(PCX_E_INVALID, "pcx_engine_import_state: array size mismatch (another launch shape or kernel?)");
    PCX_HIP(hipMemcpy(a.first, p, a.second, hipMemcpyHostToDevice));
    p += (a.second + 7) & ~(size_t)7;
  }
  e->showtime = true;
  e->epoch = h.epoch + 1;  // croppers: the observation changed under them
  e->curtains_fresh = false;
  return 0;
}

int pcx_engine_set_epilogue(pcx_engine* e, const pcx_epilogue_desc* d) {
  if (!e) return set_error(PCX_E_INVALID, "pcx_engine_set_epilogue: null engine");
  if (d) {
    if (d->depth < 1 || d->depth > PCX_POST_MAX_DEPTH || !d->out_dev || (reinterpret_cast<uintptr_t>(d->out_dev) & 15u))
      return set_error(PCX_E_INVALID, "pcx_engine_set_epilogue: bad descriptor");
    for (int i = 0; i < d->depth && !d->to_array; ++i)
      for (int j = 0; j < i; ++j)
        if (d->chars[i] == d->chars[j]) return set_error(PCX_E_UNSUPPORTED, "pcx_engine_set_epilogue: a layer is stacked twice");
  }
  PCX_HIP(hipSetDevice(e->device));
  if (!d || !d->to_array) {
    const int rc = e->backend->set_epilogue(d);
    if (rc == 0 && e->epilogue_lut) {  // (no launch may still read the old table)
      PCX_HIP(hipDeviceSynchronize());
      (void)hipFree(e->epilogue_lut);
      e->epilogue_lut = nullptr;
    }
    return rc;
  }
  // ObservationToArray as the epilogue: the board can only show the game's own characters, so "every character
  // has a value" is decided here, once, instead of per cell and step (rendering.py:503-507 raises at run time)
  if (d->channels_last || !d->lut || !d->mapped || d->dtype < PCX_U8 || d->dtype > PCX_F64)
    return set_error(PCX_E_INVALID, "pcx_engine_set_epilogue: bad ObservationToArray descriptor");
  for (int i = 0; i < e->t.n_chars; ++i)
    if (e->t.chars[i] > 127 || !d->mapped[e->t.chars[i]])
      return set_error(PCX_E_UNSUPPORTED, "pcx_engine_set_epilogue: character %d of the game has no value in the mapping "
                                          "(the stand-alone post-processor reports it per environment)", (int)e->t.chars[i]);
  const int esize = d->dtype == PCX_U8 ? 1 : (d->dtype == PCX_I32 || d->dtype == PCX_F32) ? 4 : 8;
  const int rows = d->to_array == 2 ? 1 : d->depth;  // (the repainter: one row; depth counts its output layers)
  std::vector<uint8_t> packed((((size_t)rows * 128 * esize) + 15) & ~(size_t)15, 0);
  for (int k = 0; k < rows; ++k)
    for (int c = 0; c < 128; ++c) memcpy(packed.data() + ((size_t)k * 128 + c) * esize, &d->lut[(size_t)k * 128 + c], (size_t)esize);
  void* dev = nullptr;
  PCX_HIP(hipMalloc(&dev, packed.size()));
  if (hipMemcpy(dev, packed.data(), packed.size(), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(dev);
    return set_error(PCX_E_HIP, "pcx_engine_set_epilogue: table upload failed");
  }
  const int rc = e->backend->set_epilogue(d);
  pcx::stream::EpilogueArgs* args = rc == 0 ? e->backend->epilogue_args() : nullptr;
  if (!args) {
    (void)hipFree(dev);
    return rc ? rc : set_error(PCX_E_UNSUPPORTED, "%s has no fused epilogue", e->backend->kernel_name());
  }
  args->lut = dev;
  if (e->epilogue_lut) {
    PCX_HIP(hipDeviceSynchronize());
    (void)hipFree(e->epilogue_lut);
  }
  e->epilogue_lut = dev;
  return 0;
}

int pcx_engine_read_things(pcx_engine* e, int64_t env0, int64_t n, pcx_sprite_state* sprites_host,
                           uint8_t* curtains_host) {
  if (!e || env0 < 0 || n < 0 || env0 + n > e->batch) return set_error(PCX_E_INVALID, "pcx_engine_read_things: bad range");
  PCX_HIP(hipSetDevice(e->device));
  return e->backend->read_things(env0, n, sprites_host, curtains_host);
}

int32_t pcx_engine_plane_pitch(const pcx_engine* e) { return e ? e->backend->plane_pitch() : 0; }
int64_t pcx_engine_bytes_per_step(const pcx_engine* e) { return e ? e->backend->bytes_per_step() : 0; }
const char* pcx_engine_kernel_name(const pcx_engine* e) { return e ? e->backend->kernel_name() : ""; }
int32_t pcx_engine_launch_shape(const pcx_engine* e) { return e ? e->backend->launch_shape() : -1; }
int32_t pcx_engine_tuner_done(const pcx_engine* e) { return e ? e->backend->tuner_done() : 1; }
int pcx_generic_specialise_check(const pcx_template* t, char* log, int64_t log_bytes, int64_t* code_bytes) {
  if (!t) return set_error(PCX_E_INVALID, "pcx_generic_specialise_check: null template");
  return pcx::generic_specialise_check(*t, log, log_bytes, code_bytes);
}
int pcx_scrolly_maze_specialise_check(const pcx_template* t, char* log, int64_t log_bytes, int64_t* code_bytes) {
  if (!t) return set_error(PCX_E_INVALID, "pcx_scrolly_maze_specialise_check: null template");
  return pcx::scrolly_maze_specialise_check(*t, log, log_bytes, code_bytes);
}
int64_t pcx_debug_scrolly_consts(const pcx_template* t, int32_t unit, uint32_t* words, int64_t cap) {
  if (!t || cap < 0) return set_error(PCX_E_INVALID, "pcx_debug_scrolly_consts: bad arguments");
  return pcx::scrolly_maze_consts(*t, unit, words, cap);
}
int pcx_engine_debug_counters(pcx_engine* e, uint32_t* out_host, int64_t words) {
  if (!e || (!out_host && words != -1) || words < -1) return set_error(PCX_E_INVALID, "pcx_engine_debug_counters: bad arguments");
  PCX_HIP(hipSetDevice(e->device));
  return e->backend->read_debug_counters(out_host, words);
}

int pcx_memcpy_d2h(void* dst_host, const void* src_dev, uint64_t bytes) {
  PCX_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  return 0;
}
int pcx_memcpy_h2d(void* dst_dev, const void* src_host, uint64_t bytes) {
  PCX_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
  return 0;
}
int pcx_device_malloc(void** out_dev, uint64_t bytes) {
  PCX_HIP(hipMalloc(out_dev, bytes));
  return 0;
}
int pcx_device_free(void* dev) {
  PCX_HIP(hipFree(dev));
  return 0;
}
int pcx_device_fill_probe(void* dst_dev, uint64_t bytes, void* stream) {
  if (!dst_dev || bytes < 4 || (reinterpret_cast<uintptr_t>(dst_dev) & 3u))
    return set_error(PCX_E_INVALID, "pcx_device_fill_probe: bad arguments");
  hipLaunchKernelGGL(pcx::pcx_fill_probe, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<uint32_t*>(dst_dev), bytes / 4);
  PCX_HIP(hipGetLastError());
  return 0;
}
int pcx_stream_synchronize(void* stream) {
  PCX_HIP(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

// Croppers: see pcx_crop.hip.

}  // extern "C"
