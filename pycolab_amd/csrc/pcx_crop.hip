// pcx_crop.hip -- device croppers (reference: pycolab/cropping.py).
//   pcx_crop_update: one thread per environment moves the window
//     (ScrollingCropper.crop :393-426, _initialise :438-458, _can_pan_to
//     :460-506, _pan_to :508-534, _rectify :536-542, _centroid :551-598);
//   pcx_crop_copy: one lane per (environment, output dword) gathers the window
//     from the engine's observation planes (_do_crop :118-227), padding
//     included, plane after plane, with aligned dword loads and 256-byte
//     wave stores.
// Sprite positions come from the step kernel's per-step `track` words, drape
// curtains (only for drape-tracking croppers) from its raw curtain export.
#include "pcx_internal.h"
#include "pcx_crop_window.h"

#include <cstring>
#include <vector>

using pcx::set_error;

namespace {

struct CropParams {
  int32_t kind, rows, cols, top, left, pad_char;
  int32_t n_track;
  int32_t track_kind[PCX_MAX_THINGS];  // 0 sprite, 1 drape
  int32_t track_idx[PCX_MAX_THINGS];
  int32_t margin_rows, margin_cols, off_rows, off_cols, saccade;
  int32_t R, C, L, in_pitch, out_pitch, FW;
  uint32_t chars[PCX_MAX_CHARS];
  int64_t batch, bpad;
};

__host__ __device__ inline pcx::crop::WindowRule window_rule(const CropParams& p) {
  return pcx::crop::WindowRule{p.rows, p.cols, p.R, p.C, p.margin_rows, p.margin_cols, p.off_rows, p.off_cols, p.saccade, p.pad_char};
}

// int(np.median(indices)) over the set cells of one curtain, along one axis
__device__ int median_axis(const uint32_t* bits, int64_t stride, int R, int C, int n, bool rows_axis) {
  // the two middle order statistics (0-based) of the sorted index list
  const int lo_rank = (n - 1) / 2, hi_rank = n / 2;
  int seen = 0, lo = -1, hi = -1;
  const int outer = rows_axis ? R : C, inner = rows_axis ? C : R;
  for (int o = 0; o < outer && hi < 0; ++o) {
    int cnt = 0;
    for (int i = 0; i < inner; ++i) {
      const int cell = rows_axis ? o * C + i : i * C + o;
      cnt += (bits[(size_t)(cell >> 5) * stride] >> (cell & 31)) & 1;
    }
    if (lo < 0 && seen + cnt > lo_rank) lo = o;
    if (seen + cnt > hi_rank) hi = o;
    seen += cnt;
  }
  return (int)((lo + hi) / 2.0);
}

__global__ void pcx_crop_update(CropParams p, const int32_t* track, const uint32_t* curtains, const int32_t* frame,
                                int32_t* corner, uint8_t* has_corner, uint8_t* error) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.batch) return;
  int top = p.top, left = p.left;
  if (p.kind == PCX_CROP_SCROLLING) {
    if (frame[b] == 0) has_corner[b] = 0;  // a new episode is a new Engine (cropping.py:378-391)
    int crow = 0, ccol = 0;
    bool have = false;
    for (int i = 0; i < p.n_track && !have; ++i) {  // :544-549
      if (p.track_kind[i] == 0) {
        const int32_t w = track[(size_t)p.track_idx[i] * p.bpad + b];
        if ((w >> 16) & 1) { crow = w & 0xFF; ccol = (w >> 8) & 0xFF; have = true; }
      } else {
        const uint32_t* bits = curtains + (size_t)p.track_idx[i] * p.FW * p.bpad + b;
        int n = 0;
        for (int wd = 0; wd < p.FW; ++wd) n += __popc(bits[(size_t)wd * p.bpad]);
        if (n) {
          crow = median_axis(bits, p.bpad, p.R, p.C, n, true);
          ccol = median_axis(bits, p.bpad, p.R, p.C, n, false);
          have = true;
        }
      }
    }
    int wrow = corner[2 * b], wcol = corner[2 * b + 1];
    bool has = has_corner[b] != 0;
    pcx::crop::move_window(window_rule(p), have, crow, ccol, has, wrow, wcol);
    has_corner[b] = 1;
    corner[2 * b] = wrow;
    corner[2 * b + 1] = wcol;
    top = wrow;
    left = wcol;
  } else {
    corner[2 * b] = top;
    corner[2 * b + 1] = left;
  }
  error[b] = pcx::crop::window_leaves_observation(window_rule(p), top, left);
}

// _do_crop (cropping.py:118-227).  One lane per (environment, output dword);
// the window's place in the observation is the same for every plane, so a lane
// works out its source once and then walks the 1 + L planes: consecutive lanes
// write consecutive dwords of one output plane (256-byte wave stores), and read
// their four cells as two aligned source dwords funnelled by the window's byte
// phase -- no per-byte loads on the common path.  Cells outside the observation
// take the pad character (or its layer bit); a dword that straddles two window
// rows, or touches the observation's edge, takes the byte path.
__global__ void pcx_crop_copy(CropParams p, const uint8_t* in, const int32_t* corner, const uint8_t* error,
                              uint8_t* out) {
  const int planes = 1 + p.L, qw = p.out_pitch / 4;
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;  // < batch * qw (checked on the host)
  const uint32_t b = f / (uint32_t)qw, q = f - b * (uint32_t)qw;
  if ((int64_t)b >= p.batch || error[b]) return;
  const int top = corner[2 * b], left = corner[2 * b + 1];
  const int cell0 = (int)q * 4, orow = cell0 / p.cols, ocol = cell0 - orow * p.cols;
  const int sr = orow + top, sc = left + ocol;
  const bool fast = ocol + 3 < p.cols && (unsigned)sr < (unsigned)p.R && sc >= 0 && sc + 3 < p.C;
  const uint8_t* src = in + (size_t)b * planes * p.in_pitch;
  uint32_t* dst = reinterpret_cast<uint32_t*>(out + (size_t)b * planes * p.out_pitch) + q;
  // Planes four at a time: every load of a batch is issued before its first store, so a lane waits
  // for ONE memory round trip per four planes (the plain plane-after-plane loop made this kernel
  // latency-bound: 1 + L dependent round trips per lane; profiles/r03_post_kernels.md).
  constexpr int PB = 4;
  const int ipq = p.in_pitch / 4;
  if (fast) {
    const uint32_t a = (uint32_t)(sr * p.C + sc), phase = a & 3u;
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src) + (a >> 2);
    for (int pl0 = 0; pl0 < planes; pl0 += PB) {
      uint32_t lo[PB], hi[PB];
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const uint32_t* s = s32 + (size_t)(pl0 + j < planes ? pl0 + j : pl0) * ipq;
        lo[j] = s[0];
        hi[j] = s[phase ? 1 : 0];
      }
#pragma unroll
      for (int j = 0; j < PB; ++j)
        if (pl0 + j < planes) dst[(size_t)(pl0 + j) * qw] = __builtin_amdgcn_alignbyte(hi[j], lo[j], phase);
    }
    return;
  }
  int off[4];         // the four cells' byte offsets inside a source plane (0 where the cell is not read)
  uint32_t take = 0;  // bit j: cell j comes from the observation; bit 4 + j: cell j is a real window cell
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cell = cell0 + j;
    const bool real = cell < p.rows * p.cols;  // cells past the window are plane padding (zeros)
    const int r = cell / p.cols;
    const int rr = r + top, cc = cell - r * p.cols + left;
    const bool inside = real && (unsigned)rr < (unsigned)p.R && (unsigned)cc < (unsigned)p.C;
    off[j] = inside ? rr * p.C + cc : 0;
    take |= (uint32_t)inside << j | (uint32_t)real << (4 + j);
  }
  for (int pl0 = 0; pl0 < planes; pl0 += PB) {
    uint32_t byte[PB][4];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const uint8_t* s = src + (size_t)(pl0 + i < planes ? pl0 + i : pl0) * p.in_pitch;
#pragma unroll
      for (int j = 0; j < 4; ++j) byte[i][j] = s[off[j]];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int pl = pl0 + i;
      if (pl >= planes) break;
      const uint32_t pad = pl == 0 ? (uint32_t)p.pad_char : (uint32_t)((uint32_t)p.pad_char == p.chars[pl - 1]);
      uint32_t v = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t b8 = ((take >> j) & 1u) ? byte[i][j] : ((take >> (4 + j)) & 1u) ? pad : 0u;
        v |= (b8 & 0xFFu) << (8 * j);
      }
      dst[(size_t)pl * qw] = v;
    }
  }
}

}  // namespace

// Raw curtains [n_drapes][FW][bpad] as cell-bit vectors, rebuilt on the host from
// pcx_engine_read_things (synchronous; only when a drape-tracking cropper joins
// an engine that is already in play).
static int refresh_curtains(pcx_engine* e) {
  pcx::Backend* b = e->backend;
  const int nd = e->t.n_drapes, cells = e->t.rows * e->t.cols, FW = b->curtain_words();
  if (int rc = b->ensure_curtains()) return rc;
  uint32_t* dev = const_cast<uint32_t*>(b->curtain_bits());
  if (!dev || nd <= 0 || FW <= 0) return set_error(PCX_E_UNSUPPORTED, "croppers: this engine cannot export curtains");
  const int64_t bpad = b->batch_pad(), chunk = 1 << 14;
  std::vector<uint8_t> host((size_t)chunk * nd * cells);
  std::vector<uint32_t> bits((size_t)chunk);
  for (int64_t b0 = 0; b0 < e->batch; b0 += chunk) {
    const int64_t n = e->batch - b0 < chunk ? e->batch - b0 : chunk;
    if (int rc = b->read_things(b0, n, nullptr, host.data())) return rc;
    for (int d = 0; d < nd; ++d)
      for (int w = 0; w < FW; ++w) {
        for (int64_t i = 0; i < n; ++i) {
          uint32_t v = 0;
          const uint8_t* cur = host.data() + ((size_t)i * nd + d) * cells;
          for (int bit = 0; bit < 32 && 32 * w + bit < cells; ++bit) v |= (uint32_t)(cur[32 * w + bit] != 0) << bit;
          bits[i] = v;
        }
        PCX_HIP(hipMemcpy(dev + ((size_t)d * FW + w) * bpad + b0, bits.data(), (size_t)n * 4, hipMemcpyHostToDevice));
      }
  }
  e->curtains_fresh = true;
  return 0;
}

struct pcx_cropper {
  pcx_engine* e = nullptr;
  CropParams p{};
  pcx::DevArray<uint8_t> planes, has_corner, error;
  pcx::DevArray<int32_t> corner;
  pcx::ErrorPoll error_poll;
  uint8_t* bound = nullptr;  // caller-owned output planes (pcx_cropper_bind_output)
  bool tracks_drape = false;
  bool fused = false;  // the engine's step kernel moves this window and writes its planes (pcx_engine_fuse_croppers)
  // released from a windows-only fusion: the engine's planes are stale until its next launch, the cropper's own
  // output (written by that fusion) is what crop() must keep handing out until then
  bool hold = false;
  uint64_t hold_epoch = 0;
  // a checkpoint restored the window but not the output planes (pcx_cropper_import_state): the next crop() cuts them
  // from the engine's restored observation with the stand-alone kernels, also while the step kernel runs this cropper
  bool refresh = false;
  // the window's fused float32 feature stack (pcx_cropper_set_features), written by the step kernel while fused
  float* feat = nullptr;
  int feat_depth = 0, feat_hwc = 0, feat_skip = 0;
  uint8_t feat_ch[pcx::crop::MAX_FUSED_FEATURES] = {};
  uint8_t* out_planes() const { return bound ? bound : planes.ptr; }
  int ensure_planes() {  // own output planes only when the caller bound none
    if (bound || planes.ptr) return 0;
    return planes.alloc((size_t)e->batch * (1 + p.L) * p.out_pitch);
  }
};

extern "C" {

int pcx_cropper_create(pcx_engine* e, const pcx_cropper_desc* d, pcx_cropper** out) {
  if (!e || !d || !out || d->rows <= 0 || d->cols <= 0 || (d->kind != PCX_CROP_FIXED && d->kind != PCX_CROP_SCROLLING))
    return set_error(PCX_E_INVALID, "pcx_cropper_create: bad arguments");
  const pcx_template& t = e->t;
  auto known = [&](int ch) { for (int i = 0; i < t.n_chars; ++i) if (t.chars[i] == ch) return true; return false; };
  if (d->pad_char >= 0 && !known(d->pad_char))
    return set_error(PCX_E_INVALID, "An `ObservationCropper` tried to fill empty space with a character that isn't "
                                    "used by the current game engine.");
  if (d->kind == PCX_CROP_SCROLLING && d->pad_char < 0 && (t.rows < d->rows || t.cols < d->cols))
    return set_error(PCX_E_INVALID, "A ScrollingCropper with no pad character can't be larger than the board");
  if (t.rows > 255 || t.cols > 255) return set_error(PCX_E_UNSUPPORTED, "croppers: board larger than 255x255");
  PCX_HIP(hipSetDevice(e->device));
  pcx_cropper* c = new pcx_cropper();
  c->e = e;
  CropParams& p = c->p;
  p.kind = d->kind; p.rows = d->rows; p.cols = d->cols; p.top = d->top; p.left = d->left; p.pad_char = d->pad_char;
  p.margin_rows = d->margin_rows; p.margin_cols = d->margin_cols;
  p.off_rows = d->initial_offset_rows; p.off_cols = d->initial_offset_cols; p.saccade = d->saccade;
  p.R = t.rows; p.C = t.cols; p.L = t.n_chars; p.in_pitch = e->backend->plane_pitch();
  p.out_pitch = (d->rows * d->cols + 3) & ~3;
  p.FW = e->backend->curtain_words();
  p.batch = e->batch; p.bpad = e->backend->batch_pad();
  for (int i = 0; i < t.n_chars; ++i) p.chars[i] = t.chars[i];
  p.n_track = d->kind == PCX_CROP_SCROLLING ? d->n_track : 0;
  for (int i = 0; i < p.n_track; ++i) {
    int kind = -1, idx = -1;
    for (int s = 0; s < t.n_sprites; ++s) if (t.sprites[s].ch == d->to_track[i]) { kind = 0; idx = s; }
    for (int dd = 0; dd < t.n_drapes; ++dd) if (t.drapes[dd].ch == d->to_track[i]) { kind = 1; idx = dd; }
    if (kind < 0) { delete c; return set_error(PCX_E_INVALID, "ScrollingCropper was told to track a nonexistent game entity"); }
    p.track_kind[i] = kind; p.track_idx[i] = idx;
    c->tracks_drape |= kind == 1;
  }
  if (c->tracks_drape) {
    if (e->showtime && !e->curtains_fresh) {
      // attached after its_showtime() (as tests/cropping_test.py does): the step
      // kernels have not exported raw curtains so far; rebuild them once from
      // the entity state, from now on every launch exports them
      int rc = refresh_curtains(e);
      if (rc) { delete c; return rc; }
    }
    e->want_curtains = true;
  }
  int rc;
  if ((rc = c->has_corner.alloc(e->batch)) ||
      (rc = c->error.alloc(e->batch)) || (rc = c->corner.alloc((size_t)e->batch * 2))) { delete c; return rc; }
  if ((uint64_t)e->batch * (uint64_t)(p.out_pitch / 4) >= (1ull << 32)) {
    delete c;
    return set_error(PCX_E_UNSUPPORTED, "croppers: batch x window too large for 32-bit task indices");
  }
  *out = c;
  return 0;
}

// (Re)describe the engine's fused croppers to its backend.
static int push_fused(pcx_engine* e) {
  pcx::crop::FusedCrops fc{};
  fc.n = (int32_t)e->fused.size();
  fc.only = e->fused_only && fc.n > 0;
  for (int i = 0; i < fc.n; ++i) {
    pcx_cropper* c = e->fused[i];
    const CropParams& p = c->p;
    pcx::crop::FusedWindow& w = fc.w[i];
    w.out = c->out_planes(); w.corner = c->corner.ptr; w.has_corner = c->has_corner.ptr; w.error = c->error.ptr;
    w.rule = window_rule(p);
    w.scrolling = p.kind == PCX_CROP_SCROLLING; w.top = p.top; w.left = p.left;
    w.n_track = p.n_track;
    for (int j = 0; j < p.n_track; ++j) { w.track_sprite[j] = p.track_idx[j]; w.track_kind[j] = p.track_kind[j]; }
    w.out_pitch = p.out_pitch;
    w.pad_planes = 0;
    for (int k = 0; k < p.L; ++k) if (p.pad_char >= 0 && (uint32_t)p.pad_char == p.chars[k]) w.pad_planes |= 1u << k;
    w.feat = c->feat; w.feat_depth = c->feat_depth; w.feat_hwc = c->feat_hwc; w.feat_skip = c->feat ? c->feat_skip : 0;
    for (int k = 0; k < pcx::crop::MAX_FUSED_FEATURES; ++k) w.feat_ch[k] = c->feat_ch[k];
  }
  fc.drapes = pcx::crop::tracks_drapes(&fc);
  return e->backend->set_fused_croppers(&fc);
}

void pcx_cropper_destroy(pcx_cropper* c) {
  if (!c) return;
  (void)hipSetDevice(c->e->device);
  if (c->fused) {  // the step kernel must stop writing into this cropper's arrays first
    pcx_engine* e = c->e;
    for (size_t i = 0; i < e->fused.size(); ++i)
      if (e->fused[i] == c) { e->fused.erase(e->fused.begin() + i); break; }
    if (e->fused.empty()) e->fused_only = false;
    (void)push_fused(e);
  }
  delete c;
}

int pcx_engine_fuse_croppers(pcx_engine* e, pcx_cropper* const* croppers, int32_t n, int32_t only_crops, void* stream) {
  if (!e || n < 0 || (n > 0 && !croppers)) return set_error(PCX_E_INVALID, "pcx_engine_fuse_croppers: bad arguments");
  if (n > pcx::crop::MAX_FUSED_CROPPERS)
    return set_error(PCX_E_UNSUPPORTED, "pcx_engine_fuse_croppers: at most %d croppers", pcx::crop::MAX_FUSED_CROPPERS);
  PCX_HIP(hipSetDevice(e->device));
  for (int i = 0; i < n; ++i) {
    pcx_cropper* c = croppers[i];
    if (!c || c->e != e) return set_error(PCX_E_INVALID, "pcx_engine_fuse_croppers: a cropper of another engine");
    for (int j = 0; j < i; ++j) if (croppers[j] == c) return set_error(PCX_E_INVALID, "pcx_engine_fuse_croppers: a cropper twice");
    if (c->p.n_track > pcx::crop::MAX_FUSED_TRACK)
      return set_error(PCX_E_UNSUPPORTED, "pcx_engine_fuse_croppers: croppers that track more than %d entities "
                                          "run as their own kernels", pcx::crop::MAX_FUSED_TRACK);
    if (c->p.rows > 255 || c->p.cols > 255) return set_error(PCX_E_UNSUPPORTED, "pcx_engine_fuse_croppers: window larger than 255x255");
    if (int rc = c->ensure_planes()) return rc;
  }
  const std::vector<pcx_cropper*> before = e->fused;
  const bool before_only = e->fused_only;
  auto was_fused = [&](pcx_cropper* c) { for (pcx_cropper* b : before) if (b == c) return true; return false; };
  if (e->showtime && before_only)
    for (int i = 0; i < n; ++i)
      if (!was_fused(croppers[i]))
        return set_error(PCX_E_STATE, "pcx_engine_fuse_croppers: the full observation is not being written (only_crops); "
                                      "a cropper that joins now has nothing to start from");
  e->fused.assign(croppers, croppers + n);
  e->fused_only = n > 0 && only_crops != 0;
  if (int rc = push_fused(e)) {  // the backend cannot: nothing changed
    e->fused = before;
    e->fused_only = before_only;
    return rc;
  }
  for (pcx_cropper* c : before) {
    c->fused = false;
    c->hold = before_only;
    c->hold_epoch = e->epoch;
  }
  for (int i = 0; i < n; ++i) {
    pcx_cropper* c = croppers[i];
    // in play already: crop the current observation once the stand-alone way, so that the window
    // and the output planes are those a crop() at this point would have produced
    if (e->showtime && !was_fused(c))
      if (int rc = pcx_cropper_crop(c, stream)) return rc;
    c->fused = true;
  }
  return 0;
}

int pcx_cropper_set_features(pcx_cropper* c, const pcx_epilogue_desc* d) {
  if (!c) return set_error(PCX_E_INVALID, "pcx_cropper_set_features: null cropper");
  pcx_engine* e = c->e;
  PCX_HIP(hipSetDevice(e->device));
  if (d) {
    if (!c->fused) return set_error(PCX_E_STATE, "pcx_cropper_set_features: the cropper is not fused into its engine's step kernel (pcx_engine_fuse_croppers)");
    if (!e->backend->fused_window_features())
      return set_error(PCX_E_UNSUPPORTED, "%s does not write a fused window's feature stack", e->backend->kernel_name());
    if (d->depth < 1 || d->depth > pcx::crop::MAX_FUSED_FEATURES || !d->out_dev || (reinterpret_cast<uintptr_t>(d->out_dev) & 15u) || d->to_array)
      return set_error(PCX_E_INVALID, "pcx_cropper_set_features: bad descriptor (1..%d layers, a 16-byte aligned float32 array, no value table)", pcx::crop::MAX_FUSED_FEATURES);
    if ((uint64_t)64 * d->depth * c->p.rows * c->p.cols * 4 >= (1ull << 32))
      return set_error(PCX_E_UNSUPPORTED, "pcx_cropper_set_features: window x depth too large for 32-bit lane offsets");
    for (int i = 0; i < d->depth; ++i)
      for (int j = 0; j < i; ++j)
        if (d->chars[i] == d->chars[j]) return set_error(PCX_E_UNSUPPORTED, "pcx_cropper_set_features: a layer is stacked twice");
  }
  const auto before_feat = c->feat;
  const int before_depth = c->feat_depth, before_hwc = c->feat_hwc, before_skip = c->feat_skip;
  uint8_t before_ch[pcx::crop::MAX_FUSED_FEATURES];
  memcpy(before_ch, c->feat_ch, sizeof before_ch);
  c->feat = d ? d->out_dev : nullptr;
  c->feat_depth = d ? d->depth : 0;
  c->feat_hwc = d ? d->channels_last != 0 : 0;
  c->feat_skip = d ? d->skip_layers : 0;
  memset(c->feat_ch, 0, sizeof c->feat_ch);
  for (int i = 0; d && i < d->depth; ++i) c->feat_ch[i] = d->chars[i];
  if (!c->fused) return 0;  // (cleared on a cropper that is on its own: nothing to tell the kernel)
  if (int rc = push_fused(e)) {  // refused: nothing changed
    c->feat = before_feat; c->feat_depth = before_depth; c->feat_hwc = before_hwc; c->feat_skip = before_skip;
    memcpy(c->feat_ch, before_ch, sizeof before_ch);
    return rc;
  }
  return 0;
}

int pcx_cropper_crop(pcx_cropper* c, void* stream) {
  if (!c) return set_error(PCX_E_INVALID, "pcx_cropper_crop: null cropper");
  pcx_engine* e = c->e;
  if (!e->showtime) return set_error(PCX_E_STATE, "pcx_cropper_crop: the engine is not in play");
  if (c->fused && c->refresh && e->fused_only)
    return set_error(PCX_E_STATE, "pcx_cropper_crop: the checkpoint held no cropped planes and the engine writes no full-board "
                                  "planes to cut them from (export with the observation, or step once)");
  if (c->fused && !c->refresh) return 0;  // the step kernel moved the window and wrote the planes already
  c->refresh = false;
  if (c->hold) {
    if (c->hold_epoch == e->epoch) return 0;  // no launch since the windows-only fusion ended: its output stands
    c->hold = false;
  }
  if (c->tracks_drape && !e->curtains_fresh)
    return set_error(PCX_E_STATE, "pcx_cropper_crop: curtains were not exported by the last step");
  PCX_HIP(hipSetDevice(e->device));
  if (int rc = c->ensure_planes()) return rc;
  hipStream_t s = (hipStream_t)stream;
  const CropParams& p = c->p;
  hipLaunchKernelGGL(pcx_crop_update, dim3((unsigned)((p.batch + 255) / 256)), dim3(256), 0, s, p,
                     e->backend->sprite_track(), e->backend->curtain_bits(), e->out.frame, c->corner.ptr,
                     c->has_corner.ptr, c->error.ptr);
  const int64_t total = p.batch * (p.out_pitch / 4);
  hipLaunchKernelGGL(pcx_crop_copy, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, e->out.planes,
                     c->corner.ptr, c->error.ptr, c->out_planes());
  PCX_HIP(hipGetLastError());
  return 0;
}

// ---- checkpoint / resume (pcx_engine_export_state's companion: the window state lives here, not in the engine) ----
namespace {
struct CropStateHeader {
  uint32_t magic, abi;
  int64_t batch;
  int32_t rows, cols, n_chars, out_pitch, with_planes, pad;
};
constexpr uint32_t CROP_STATE_MAGIC = 0x57584350u;  // "PCXW"
}  // namespace

int pcx_cropper_state_size(pcx_cropper* c, int32_t with_planes, uint64_t* bytes) {
  if (!c || !bytes) return set_error(PCX_E_INVALID, "pcx_cropper_state_size: bad arguments");
  const uint64_t B = (uint64_t)c->e->batch;
  *bytes = sizeof(CropStateHeader) + B * 8 + ((B + 7) & ~7ull) + (with_planes ? B * (1 + c->p.L) * c->p.out_pitch : 0);
  return 0;
}

int pcx_cropper_export_state(pcx_cropper* c, void* host, uint64_t bytes, int32_t with_planes) {
  uint64_t need = 0;
  if (!c || !host || pcx_cropper_state_size(c, with_planes, &need)) return set_error(PCX_E_INVALID, "pcx_cropper_export_state: bad arguments");
  if (bytes < need) return set_error(PCX_E_INVALID, "pcx_cropper_export_state: %llu bytes given, %llu needed",
                                     (unsigned long long)bytes, (unsigned long long)need);
  PCX_HIP(hipSetDevice(c->e->device));
  if (with_planes) { if (int rc = c->ensure_planes()) return rc; }
  PCX_HIP(hipDeviceSynchronize());
  const uint64_t B = (uint64_t)c->e->batch;
  CropStateHeader h{CROP_STATE_MAGIC, PCX_ABI_VERSION, c->e->batch, c->p.rows, c->p.cols, c->p.L, c->p.out_pitch, with_planes != 0, 0};
  uint8_t* p = static_cast<uint8_t*>(host);
  memcpy(p, &h, sizeof h); p += sizeof h;
  PCX_HIP(hipMemcpy(p, c->corner.ptr, B * 8, hipMemcpyDeviceToHost)); p += B * 8;
  PCX_HIP(hipMemcpy(p, c->has_corner.ptr, B, hipMemcpyDeviceToHost)); p += (B + 7) & ~7ull;
  if (with_planes) PCX_HIP(hipMemcpy(p, c->out_planes(), B * (1 + c->p.L) * c->p.out_pitch, hipMemcpyDeviceToHost));
  return 0;
}

int pcx_cropper_import_state(pcx_cropper* c, const void* host, uint64_t bytes) {
  if (!c || !host || bytes < sizeof(CropStateHeader)) return set_error(PCX_E_INVALID, "pcx_cropper_import_state: bad arguments");
  CropStateHeader h;
  memcpy(&h, host, sizeof h);
  if (h.magic != CROP_STATE_MAGIC || h.abi != PCX_ABI_VERSION) return set_error(PCX_E_INVALID, "pcx_cropper_import_state: not a cropper checkpoint of this ABI");
  if (h.batch != c->e->batch || h.rows != c->p.rows || h.cols != c->p.cols || h.n_chars != c->p.L || h.out_pitch != c->p.out_pitch)
    return set_error(PCX_E_INVALID, "pcx_cropper_import_state: the checkpoint is of another window, game or batch");
  uint64_t need = 0;
  pcx_cropper_state_size(c, h.with_planes, &need);
  if (bytes < need) return set_error(PCX_E_INVALID, "pcx_cropper_import_state: truncated checkpoint");
  PCX_HIP(hipSetDevice(c->e->device));
  if (h.with_planes) { if (int rc = c->ensure_planes()) return rc; }
  PCX_HIP(hipDeviceSynchronize());
  const uint64_t B = (uint64_t)c->e->batch;
  const uint8_t* p = static_cast<const uint8_t*>(host) + sizeof h;
  PCX_HIP(hipMemcpy(c->corner.ptr, p, B * 8, hipMemcpyHostToDevice)); p += B * 8;
  PCX_HIP(hipMemcpy(c->has_corner.ptr, p, B, hipMemcpyHostToDevice)); p += (B + 7) & ~7ull;
  if (h.with_planes) PCX_HIP(hipMemcpy(c->out_planes(), p, B * (1 + c->p.L) * c->p.out_pitch, hipMemcpyHostToDevice));
  // the restored planes stand until the engine's next launch; without them the next crop() cuts them afresh
  c->hold = h.with_planes != 0;
  c->hold_epoch = c->e->epoch;
  c->refresh = !h.with_planes;
  return 0;
}

int pcx_cropper_buffers(pcx_cropper* c, uint8_t** planes_dev, int32_t** corner_dev) {
  if (!c) return set_error(PCX_E_INVALID, "pcx_cropper_buffers: null cropper");
  if (int rc = c->ensure_planes()) return rc;
  if (planes_dev) *planes_dev = c->out_planes();  // [batch][1+n_chars][pitch]
  if (corner_dev) *corner_dev = c->corner.ptr;
  return 0;
}

int pcx_cropper_planes_view(pcx_cropper* c, pcx_planes_view* out) {
  if (!c || !out) return set_error(PCX_E_INVALID, "pcx_cropper_planes_view: bad arguments");
  if (int rc = c->ensure_planes()) return rc;
  memset(out, 0, sizeof *out);
  out->planes = c->out_planes(); out->batch = c->e->batch; out->rows = c->p.rows; out->cols = c->p.cols;
  out->pitch = c->p.out_pitch; out->n_chars = c->p.L;
  memcpy(out->chars, c->e->t.chars, PCX_MAX_CHARS);
  return 0;
}

int32_t pcx_cropper_plane_pitch(const pcx_cropper* c) { return c ? c->p.out_pitch : 0; }

int pcx_cropper_bind_output(pcx_cropper* c, uint8_t* planes_dev) {
  if (!c || !planes_dev) return set_error(PCX_E_INVALID, "pcx_cropper_bind_output: bad arguments");
  c->bound = planes_dev;
  if (c->fused) return push_fused(c->e);  // the step kernel writes there from now on
  return 0;
}

int pcx_cropper_error_buffer(pcx_cropper* c, const uint8_t** errors_dev) {
  if (!c || !errors_dev) return set_error(PCX_E_INVALID, "pcx_cropper_error_buffer: bad arguments");
  *errors_dev = c->error.ptr;
  return 0;
}

int pcx_cropper_error_poll(pcx_cropper* c, void* stream, int32_t* seen) {
  if (!c) return set_error(PCX_E_INVALID, "pcx_cropper_error_poll: null cropper");
  PCX_HIP(hipSetDevice(c->e->device));
  return c->error_poll.poll(c->error.ptr, c->e->batch, (hipStream_t)stream, seen);
}

int pcx_cropper_errors(pcx_cropper* c, uint8_t* errors_host) {
  if (!c || !errors_host) return set_error(PCX_E_INVALID, "pcx_cropper_errors: bad arguments");
  PCX_HIP(hipSetDevice(c->e->device));
  PCX_HIP(hipDeviceSynchronize());
  PCX_HIP(hipMemcpy(errors_host, c->error.ptr, (size_t)c->e->batch, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
