// pcx_post.hip -- observation post-processors as streaming epilogue kernels
// (reference: pycolab/rendering.py:304-661).  One thread per output element
// group; inputs are the planes an engine or a cropper just wrote (L2-hot).
#include "pcx_internal.h"

#include <cstring>

using pcx::set_error;

namespace {

struct PostParams {
  int32_t kind, dtype, esize, depth, R, C, cells, pitch, n_planes;
  int64_t batch, stride[3];
  int32_t layer_plane[PCX_POST_MAX_DEPTH];  // FEATURE_ARRAY: source plane per output layer, -1 = absent
  uint32_t out_char[PCX_POST_MAX_DEPTH];    // REPAINT: output layer characters
};

// ObservationToArray: out[b][perm(d, r, c)] = lut[d][board[b][r][c]]
__global__ void pcx_post_to_array(PostParams p, const uint8_t* planes, const uint64_t* lut, const uint8_t* mapped,
                                  uint8_t* out, uint8_t* error) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.batch * p.cells) return;
  const int64_t b = i / p.cells;
  const int cell = (int)(i - b * p.cells), r = cell / p.C, c = cell - r * p.C;
  const uint32_t ch = planes[(size_t)b * p.n_planes * p.pitch + cell] & 127u;
  if (!mapped[ch]) { error[b] = 1; return; }
  for (int d = 0; d < p.depth; ++d) {
    const uint64_t v = lut[d * 128 + ch];
    const int64_t o = b * (int64_t)p.depth * p.cells + d * p.stride[0] + r * p.stride[1] + c * p.stride[2];
    switch (p.esize) {
      case 1: out[o] = (uint8_t)v; break;
      case 4: reinterpret_cast<uint32_t*>(out)[o] = (uint32_t)v; break;
      default: reinterpret_cast<uint64_t*>(out)[o] = v; break;
    }
  }
}

// ObservationToFeatureArray: float32 stack of chosen layer planes
__global__ void pcx_post_features(PostParams p, const uint8_t* planes, float* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.batch * p.depth * p.cells) return;
  const int cell = (int)(i % p.cells);
  const int d = (int)((i / p.cells) % p.depth);
  const int64_t b = i / ((int64_t)p.cells * p.depth);
  const int r = cell / p.C, c = cell - r * p.C;
  const int plane = p.layer_plane[d];
  const float v = plane < 0 ? 0.0f : (float)planes[((size_t)b * p.n_planes + plane) * p.pitch + cell];
  out[b * (int64_t)p.depth * p.cells + d * p.stride[0] + r * p.stride[1] + c * p.stride[2]] = v;
}

// ObservationCharacterRepainter: board through a 128-entry table, layers = board == c
__global__ void pcx_post_repaint(PostParams p, const uint8_t* planes, const uint64_t* lut, uint8_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.batch * p.cells) return;
  const int64_t b = i / p.cells;
  const int cell = (int)(i - b * p.cells);
  const uint32_t ch = (uint32_t)lut[planes[(size_t)b * p.n_planes * p.pitch + cell] & 127u] & 0xFFu;
  uint8_t* o = out + (size_t)b * (1 + p.depth) * p.cells + cell;
  o[0] = (uint8_t)ch;
  for (int d = 0; d < p.depth; ++d) o[(size_t)(1 + d) * p.cells] = ch == p.out_char[d];
}

}  // namespace

struct pcx_post {
  int device = 0;
  PostParams p{};
  const uint8_t* planes = nullptr;
  pcx::DevArray<uint64_t> lut;
  pcx::DevArray<uint8_t> mapped, out, error;
  uint64_t out_bytes = 0;
};

extern "C" {

int pcx_engine_planes_view(pcx_engine* e, pcx_planes_view* out) {
  if (!e || !out) return set_error(PCX_E_INVALID, "pcx_engine_planes_view: bad arguments");
  if (!e->out.planes) return set_error(PCX_E_STATE, "pcx_engine_planes_view: the engine has no output buffers yet");
  memset(out, 0, sizeof *out);
  out->planes = e->out.planes; out->batch = e->batch; out->rows = e->t.rows; out->cols = e->t.cols;
  out->pitch = e->backend->plane_pitch(); out->n_chars = e->t.n_chars;
  memcpy(out->chars, e->t.chars, PCX_MAX_CHARS);
  return 0;
}

int pcx_post_create(const pcx_planes_view* src, const pcx_post_desc* d, int device_id, pcx_post** out) {
  if (!src || !d || !out || !src->planes || src->batch <= 0 || d->depth < 1 || d->depth > PCX_POST_MAX_DEPTH)
    return set_error(PCX_E_INVALID, "pcx_post_create: bad arguments");
  PCX_HIP(hipSetDevice(device_id));
  pcx_post* q = new pcx_post();
  q->device = device_id;
  q->planes = src->planes;
  PostParams& p = q->p;
  p.kind = d->kind; p.dtype = d->dtype; p.depth = d->depth; p.R = src->rows; p.C = src->cols;
  p.cells = src->rows * src->cols; p.pitch = src->pitch; p.n_planes = 1 + src->n_chars; p.batch = src->batch;
  for (int i = 0; i < 3; ++i) p.stride[i] = d->stride[i];
  int rc = 0;
  switch (d->kind) {
    case PCX_POST_TO_ARRAY:
      p.esize = d->dtype == PCX_U8 ? 1 : (d->dtype == PCX_I32 || d->dtype == PCX_F32) ? 4 : 8;
      q->out_bytes = (uint64_t)p.batch * p.depth * p.cells * p.esize;
      break;
    case PCX_POST_FEATURE_ARRAY:
      p.esize = 4;
      q->out_bytes = (uint64_t)p.batch * p.depth * p.cells * 4;
      for (int i = 0; i < p.depth; ++i) {
        p.layer_plane[i] = -1;
        for (int k = 0; k < src->n_chars; ++k) if (src->chars[k] == d->chars[i]) p.layer_plane[i] = 1 + k;
      }
      break;
    case PCX_POST_REPAINT:
      p.esize = 1;
      q->out_bytes = (uint64_t)p.batch * (1 + p.depth) * p.cells;
      for (int i = 0; i < p.depth; ++i) p.out_char[i] = d->chars[i];
      break;
    default: delete q; return set_error(PCX_E_INVALID, "pcx_post_create: unknown kind");
  }
  std::vector<uint64_t> lut((size_t)PCX_POST_MAX_DEPTH * 128);
  memcpy(lut.data(), d->lut, sizeof d->lut);
  std::vector<uint8_t> mapped(d->mapped, d->mapped + 128);
  if ((rc = q->lut.upload(lut)) || (rc = q->mapped.upload(mapped)) || (rc = q->out.alloc(q->out_bytes)) ||
      (rc = q->error.alloc(p.batch))) { delete q; return rc; }
  *out = q;
  return 0;
}

void pcx_post_destroy(pcx_post* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  delete p;
}

int pcx_post_run(pcx_post* q, void* stream) {
  if (!q) return set_error(PCX_E_INVALID, "pcx_post_run: null");
  PCX_HIP(hipSetDevice(q->device));
  hipStream_t s = (hipStream_t)stream;
  const PostParams& p = q->p;
  PCX_HIP(hipMemsetAsync(q->error.ptr, 0, p.batch, s));
  if (p.kind == PCX_POST_TO_ARRAY) {
    const int64_t n = p.batch * p.cells;
    hipLaunchKernelGGL(pcx_post_to_array, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, q->planes, q->lut.ptr,
                       q->mapped.ptr, q->out.ptr, q->error.ptr);
  } else if (p.kind == PCX_POST_FEATURE_ARRAY) {
    const int64_t n = p.batch * p.depth * p.cells;
    hipLaunchKernelGGL(pcx_post_features, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, q->planes,
                       reinterpret_cast<float*>(q->out.ptr));
  } else {
    const int64_t n = p.batch * p.cells;
    hipLaunchKernelGGL(pcx_post_repaint, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, q->planes, q->lut.ptr,
                       q->out.ptr);
  }
  PCX_HIP(hipGetLastError());
  return 0;
}

int pcx_post_output(pcx_post* p, void** out_dev, uint64_t* bytes) {
  if (!p) return set_error(PCX_E_INVALID, "pcx_post_output: null");
  if (out_dev) *out_dev = p->out.ptr;
  if (bytes) *bytes = p->out_bytes;
  return 0;
}

int pcx_post_errors(pcx_post* p, uint8_t* errors_host) {
  if (!p || !errors_host) return set_error(PCX_E_INVALID, "pcx_post_errors: bad arguments");
  PCX_HIP(hipSetDevice(p->device));
  PCX_HIP(hipDeviceSynchronize());
  PCX_HIP(hipMemcpy(errors_host, p->error.ptr, (size_t)p->p.batch, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
