// pcx_post.hip -- observation post-processors as streaming epilogue kernels
// (reference: pycolab/rendering.py:304-661).  Inputs are the planes an engine
// or a cropper just wrote (L2 / Infinity-Cache hot); one lane per board dword
// (four cells), so plane reads are coalesced dwords and -- in the reference's
// default axis order -- every store is a 4-, 16- or 32-byte vector per lane
// (256 B to 2 KiB contiguous per wave).  The channels-last feature array
// (`permute=(1, 2, 0)`) divides its OUTPUT among the lanes instead (1 KiB
// contiguous per wave store, layer bytes gathered); other permuted outputs fall
// back to strided element stores.  The output array belongs to the caller
// when one was bound (pcx_post_bind_output): the host hands it on as a device
// tensor, no copy and no synchronisation.
#include "pcx_internal.h"

#include <cstring>

using pcx::set_error;

namespace {

struct PostParams {
  int32_t kind, dtype, esize, depth, R, C, cells, pitch, n_planes, out_pitch, linear;
  int64_t batch, stride[3];
  int32_t layer_plane[PCX_POST_MAX_DEPTH];  // FEATURE_ARRAY: source plane per output layer, -1 = absent
  uint32_t out_char[PCX_POST_MAX_DEPTH];    // REPAINT: output layer characters
};

template <typename T>
__device__ __forceinline__ void store4(T* dst, T a, T b, T c, T d);
template <>
__device__ __forceinline__ void store4<uint8_t>(uint8_t* dst, uint8_t a, uint8_t b, uint8_t c, uint8_t d) {
  *reinterpret_cast<uint32_t*>(dst) = (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
}
template <>
__device__ __forceinline__ void store4<uint32_t>(uint32_t* dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  *reinterpret_cast<uint4*>(dst) = make_uint4(a, b, c, d);
}
template <>
__device__ __forceinline__ void store4<uint64_t>(uint64_t* dst, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
  reinterpret_cast<ulonglong2*>(dst)[0] = make_ulonglong2(a, b);
  reinterpret_cast<ulonglong2*>(dst)[1] = make_ulonglong2(c, d);
}

// ObservationToArray (rendering.py:409-542): out[b][perm(d, r, c)] = lut[d][board[b][r][c]]
template <typename T>
__global__ void pcx_post_to_array(PostParams p, const uint8_t* planes, const uint64_t* lut, const uint8_t* mapped,
                                  T* out, uint8_t* error) {
  extern __shared__ uint64_t lds_raw64[];  // [depth][128] values of type T, then 128 mapped bytes
  T* const lds_lut = reinterpret_cast<T*>(lds_raw64);
  uint8_t* const lds_mapped = reinterpret_cast<uint8_t*>(lds_lut + p.depth * 128);
  for (int i = threadIdx.x; i < p.depth * 128; i += blockDim.x) lds_lut[i] = (T)lut[i];
  for (int i = threadIdx.x; i < 128; i += blockDim.x) lds_mapped[i] = mapped[i];
  __syncthreads();
  // the table is staged once per workgroup, so a workgroup takes many dwords: a contiguous chunk of the
  // batch's dwords, walked front to back (sequential read and write streams per workgroup), one lane per
  // board dword and trip; (environment, dword) advance incrementally -- one division per lane, not per trip
  const uint32_t qw = (uint32_t)p.pitch / 4u, total = (uint32_t)p.batch * qw;
  const uint32_t chunk = (total + gridDim.x - 1) / gridDim.x, stride = blockDim.x;
  const uint32_t db = stride / qw, dq = stride - db * qw;
  uint32_t f = blockIdx.x * chunk + threadIdx.x;
  const uint32_t end = (blockIdx.x + 1) * chunk < total ? (blockIdx.x + 1) * chunk : total;
  uint32_t b = f / qw, q = f - b * qw;
  for (; f < end; f += stride) {
    const uint32_t b_now = b, q_now = q;
    q += dq; b += db;
    if (q >= qw) { q -= qw; ++b; }
    const uint32_t d4 = reinterpret_cast<const uint32_t*>(planes + (size_t)b_now * p.n_planes * p.pitch)[q_now];
    const int cell0 = (int)q_now * 4, n = p.cells - cell0 < 4 ? p.cells - cell0 : 4;
    uint32_t ch[4], bad = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t c8 = (d4 >> (8 * j)) & 0xFFu;
      ch[j] = c8 & 127u;
      // rendering.py:503-507 a character the mapping lacks (arithmetic: no branch around the table read)
      bad |= (uint32_t)(j < n) & ((uint32_t)(c8 >= 128u) | (uint32_t)(lds_mapped[ch[j]] == 0));
    }
    if (bad) { error[b_now] = 1; continue; }
    T* const o = out + (size_t)b_now * p.depth * p.cells;
    if (p.linear && n == 4) {  // default axis order: (d, r, c) row-major, four consecutive elements per layer
      for (int d = 0; d < p.depth; ++d)
        store4<T>(o + (size_t)d * p.cells + cell0, lds_lut[d * 128 + ch[0]], lds_lut[d * 128 + ch[1]],
                  lds_lut[d * 128 + ch[2]], lds_lut[d * 128 + ch[3]]);
      continue;
    }
    for (int j = 0; j < n; ++j) {
      const int cell = cell0 + j, r = cell / p.C, c = cell - r * p.C;
      for (int d = 0; d < p.depth; ++d)
        o[d * p.stride[0] + r * p.stride[1] + c * p.stride[2]] = lds_lut[d * 128 + ch[j]];
    }
  }
}

// ObservationToFeatureArray (rendering.py:545-661): float32 stack of chosen layer planes
__global__ void pcx_post_features(PostParams p, const uint8_t* planes, float* out) {
  const int qw = p.pitch / 4;
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t b = f / (uint32_t)qw, q = f - b * (uint32_t)qw;
  if ((int64_t)b >= p.batch) return;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(planes + (size_t)b * p.n_planes * p.pitch) + q;
  const int cell0 = (int)q * 4, n = p.cells - cell0 < 4 ? p.cells - cell0 : 4;
  float* const o = out + (size_t)b * p.depth * p.cells;
  for (int d = 0; d < p.depth; ++d) {
    const int plane = p.layer_plane[d];
    const uint32_t m = plane < 0 ? 0u : src[(size_t)plane * qw];
    const float v0 = (float)(m & 0xFFu), v1 = (float)((m >> 8) & 0xFFu), v2 = (float)((m >> 16) & 0xFFu), v3 = (float)(m >> 24);
    if (p.linear == 1 && n == 4) {
      *reinterpret_cast<float4*>(o + (size_t)d * p.cells + cell0) = make_float4(v0, v1, v2, v3);
    } else {
      const float v[4] = {v0, v1, v2, v3};
      for (int j = 0; j < n; ++j) {
        const int cell = cell0 + j, r = cell / p.C, c = cell - r * p.C;
        o[d * p.stride[0] + r * p.stride[1] + c * p.stride[2]] = v[j];
      }
    }
  }
}

// ObservationToFeatureArray with permute=(1, 2, 0) ("channels last", [R][C][depth] per environment) on
// boards of whole dwords, where the output of the whole batch is one contiguous run of
// batch * cells * depth floats in cell order.  The OUTPUT is what the lanes divide: a wave owns 64 * depth
// consecutive float4s (the features of 256 consecutive cells), lane l writes float4 number i * 64 + l in
// trip i -- every store instruction is 1 KiB contiguous -- and takes the four layer bytes behind it from
// the wave's LDS copy of its cells' layer dwords (fetched with coalesced dword loads).  Lanes dividing the INPUT instead (a cell dword
// each, 16-byte stores 16 * depth bytes apart) measured 0.74 ms on marauders at 32,768 environments
// against 0.20 ms for the default axis order; profiles/r02_post_kernels.md.
__global__ void pcx_post_features_hwc(PostParams p, const uint8_t* planes, float* out, uint32_t depth_magic) {
  extern __shared__ uint32_t stage_all[];  // [waves per block][depth][64] layer dwords of the wave's 256 cells
  const uint32_t lane = threadIdx.x & 63u, depth = (uint32_t)p.depth, cells = (uint32_t)p.cells;
  uint32_t* const stage = stage_all + (threadIdx.x >> 6) * depth * 64u;
  const uint64_t wave_id = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t cell_first = wave_id * 256u, total_cells = (uint64_t)p.batch * (uint64_t)p.cells;
  if (cell_first >= total_cells) return;
  // stage: lane l fetches, per layer, the dword of cells 4l .. 4l+3 of the wave's run (coalesced; a dword never
  // straddles two environments because cells % 4 == 0)
  {
    const uint64_t cell = cell_first + 4u * lane;
    const bool live = cell < total_cells;
    const uint64_t b = live ? cell / cells : 0;
    const uint32_t c = live ? (uint32_t)(cell - b * cells) : 0u;
    const uint8_t* const src = planes + b * ((size_t)p.n_planes * p.pitch) + c;
    for (uint32_t d = 0; d < depth; ++d) {
      const int plane = p.layer_plane[d];  // (uniform)
      stage[d * 64u + lane] = (live && plane >= 0) ? *reinterpret_cast<const uint32_t*>(src + (size_t)plane * p.pitch) : 0u;
    }
  }
  // (a wave reads back only what it wrote itself: no barrier, the LDS accesses of one wave complete in order)
  const uint8_t* const bytes = reinterpret_cast<const uint8_t*>(stage);
  float4* const o = reinterpret_cast<float4*>(out + cell_first * depth);
  const uint32_t cells_here = total_cells - cell_first < 256u ? (uint32_t)(total_cells - cell_first) : 256u;
  for (uint32_t i = 0; i < depth; ++i) {
    const uint32_t g = i * 64u + lane;                 // float4 of the wave's run
    float v[4];
    uint32_t j0 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t e = 4u * g + (uint32_t)k;         // element of the run: cell e / depth, layer e % depth
      uint32_t j = __umulhi(e, depth_magic), d = e - j * depth;
      if (d >= depth) { d -= depth; ++j; }             // (the estimate is at most one short)
      if (k == 0) j0 = j;
      v[k] = (float)bytes[d * 256u + j];
    }
    if (j0 < cells_here) o[g] = make_float4(v[0], v[1], v[2], v[3]);  // (cells_here is a multiple of 4, depth elements per cell)
  }
}

// ObservationCharacterRepainter (rendering.py:304-406): board through a
// 128-entry table, then layers[c] = (board == c) by a byte-wise SWAR compare.
__global__ void pcx_post_repaint(PostParams p, const uint8_t* planes, const uint64_t* lut, uint8_t* out, uint8_t* error) {
  __shared__ uint8_t table[128];
  for (int i = threadIdx.x; i < 128; i += blockDim.x) table[i] = (uint8_t)lut[i];
  __syncthreads();
  const int qw = p.pitch / 4, oqw = p.out_pitch / 4;
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t b = f / (uint32_t)qw, q = f - b * (uint32_t)qw;
  if ((int64_t)b >= p.batch) return;
  const uint32_t d4 = reinterpret_cast<const uint32_t*>(planes + (size_t)b * p.n_planes * p.pitch)[q];
  const int cell0 = (int)q * 4, n = p.cells - cell0 < 4 ? p.cells - cell0 : 4;
  uint32_t v = 0, valid = 0;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t ch = (d4 >> (8 * j)) & 0xFFu;
    if (j < n) {
      bad |= ch >= 128u;
      v |= (uint32_t)table[ch & 127u] << (8 * j);
      valid |= 0x01u << (8 * j);
    }
  }
  if (bad) error[b] = 1;
  uint32_t* o = reinterpret_cast<uint32_t*>(out + (size_t)b * (1 + p.depth) * p.out_pitch) + q;
  *o = v;
  for (int d = 0; d < p.depth; ++d) {
    o += oqw;
    const uint32_t x = v ^ (p.out_char[d] * 0x01010101u);  // a zero byte where the board shows this character
    const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    *o = ((~(t | x | 0x7F7F7F7Fu)) >> 7) & valid;
  }
}

}  // namespace

struct pcx_post {
  int device = 0;
  PostParams p{};
  const uint8_t* planes = nullptr;
  pcx::DevArray<uint64_t> lut;
  pcx::DevArray<uint8_t> mapped, out, error;
  pcx::ErrorPoll error_poll;
  void* bound = nullptr;  // caller-owned output (pcx_post_bind_output)
  uint64_t out_bytes = 0;
  int ensure_out() {
    if (bound || out.ptr) return 0;
    return out.alloc(out_bytes);
  }
  void* out_ptr() const { return bound ? bound : (void*)out.ptr; }
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
  if (src->pitch % 4 != 0 || src->pitch < src->rows * src->cols)
    return set_error(PCX_E_INVALID, "pcx_post_create: plane pitch must be a multiple of 4 that covers rows*cols");
  if ((uint64_t)src->batch * (uint64_t)(src->pitch / 4) >= (1ull << 32))
    return set_error(PCX_E_UNSUPPORTED, "pcx_post_create: batch x board too large for 32-bit task indices");
  PCX_HIP(hipSetDevice(device_id));
  pcx_post* q = new pcx_post();
  q->device = device_id;
  q->planes = src->planes;
  PostParams& p = q->p;
  p.kind = d->kind; p.dtype = d->dtype; p.depth = d->depth; p.R = src->rows; p.C = src->cols;
  p.cells = src->rows * src->cols; p.pitch = src->pitch; p.n_planes = 1 + src->n_chars; p.batch = src->batch;
  p.out_pitch = (p.cells + 3) & ~3;
  for (int i = 0; i < 3; ++i) p.stride[i] = d->stride[i];
  // the reference's default axis order: element (d, r, c) at d * cells + r * C + c
  p.linear = d->stride[0] == p.cells && d->stride[1] == p.C && d->stride[2] == 1;
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
      q->out_bytes = (uint64_t)p.batch * (1 + p.depth) * p.out_pitch;
      for (int i = 0; i < p.depth; ++i) p.out_char[i] = d->chars[i];
      break;
    default: delete q; return set_error(PCX_E_INVALID, "pcx_post_create: unknown kind");
  }
  // vector stores need the per-environment block to keep the vector's alignment
  if (p.linear && ((uint64_t)p.depth * p.cells * p.esize) % (4u * p.esize) != 0) p.linear = 0;
  if (p.linear && p.cells % 4 != 0) p.linear = 0;
  // "channels last" (permute=(1, 2, 0)): element (d, r, c) at (r * C + c) * depth + d
  if (d->kind == PCX_POST_FEATURE_ARRAY && d->stride[0] == 1 && d->stride[2] == p.depth &&
      d->stride[1] == (int64_t)p.C * p.depth && p.cells % 4 == 0)
    p.linear = 2;
  std::vector<uint64_t> lut((size_t)PCX_POST_MAX_DEPTH * 128);
  memcpy(lut.data(), d->lut, sizeof d->lut);
  std::vector<uint8_t> mapped(d->mapped, d->mapped + 128);
  if ((rc = q->lut.upload(lut)) || (rc = q->mapped.upload(mapped)) || (rc = q->error.alloc(p.batch))) { delete q; return rc; }
  *out = q;
  return 0;
}

void pcx_post_destroy(pcx_post* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  delete p;
}

int pcx_post_bind_output(pcx_post* p, void* out_dev, uint64_t bytes) {
  if (!p || !out_dev) return set_error(PCX_E_INVALID, "pcx_post_bind_output: bad arguments");
  if (bytes != p->out_bytes) return set_error(PCX_E_INVALID, "pcx_post_bind_output: the output needs %llu bytes", (unsigned long long)p->out_bytes);
  if ((reinterpret_cast<uintptr_t>(out_dev) & 15u) != 0) return set_error(PCX_E_INVALID, "pcx_post_bind_output: the output must be 16-byte aligned");
  p->bound = out_dev;
  return 0;
}

int pcx_post_run(pcx_post* q, void* stream) {
  if (!q) return set_error(PCX_E_INVALID, "pcx_post_run: null");
  PCX_HIP(hipSetDevice(q->device));
  if (int rc = q->ensure_out()) return rc;
  hipStream_t s = (hipStream_t)stream;
  const PostParams& p = q->p;
  PCX_HIP(hipMemsetAsync(q->error.ptr, 0, p.batch, s));
  const int64_t n = p.batch * (p.pitch / 4);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (p.kind == PCX_POST_TO_ARRAY) {
    const size_t lds = (((size_t)p.depth * 128 * p.esize + 7) & ~(size_t)7) + 128;
    const dim3 grid((unsigned)((n + 255) / 256 < 256 * 16 ? (n + 255) / 256 : 256 * 16));  // 16 workgroups per CU stage the table once each
    if (p.esize == 1)
      hipLaunchKernelGGL(pcx_post_to_array<uint8_t>, grid, block, lds, s, p, q->planes, q->lut.ptr, q->mapped.ptr,
                         reinterpret_cast<uint8_t*>(q->out_ptr()), q->error.ptr);
    else if (p.esize == 4)
      hipLaunchKernelGGL(pcx_post_to_array<uint32_t>, grid, block, lds, s, p, q->planes, q->lut.ptr, q->mapped.ptr,
                         reinterpret_cast<uint32_t*>(q->out_ptr()), q->error.ptr);
    else
      hipLaunchKernelGGL(pcx_post_to_array<uint64_t>, grid, block, lds, s, p, q->planes, q->lut.ptr, q->mapped.ptr,
                         reinterpret_cast<uint64_t*>(q->out_ptr()), q->error.ptr);
  } else if (p.kind == PCX_POST_FEATURE_ARRAY) {
    if (p.linear == 2) {  // channels last on a whole-dword board: the lanes divide the output
      const int64_t waves = (p.batch * p.cells + 255) / 256;
      const uint32_t magic = 0xFFFFFFFFu / (uint32_t)p.depth;
      hipLaunchKernelGGL(pcx_post_features_hwc, dim3((unsigned)((waves + 3) / 4)), block, (size_t)p.depth * 256 * 4, s, p, q->planes,
                         reinterpret_cast<float*>(q->out_ptr()), magic);
    } else {
      hipLaunchKernelGGL(pcx_post_features, grid, block, 0, s, p, q->planes, reinterpret_cast<float*>(q->out_ptr()));
    }
  } else {
    hipLaunchKernelGGL(pcx_post_repaint, grid, block, 0, s, p, q->planes, q->lut.ptr, reinterpret_cast<uint8_t*>(q->out_ptr()),
                       q->error.ptr);
  }
  PCX_HIP(hipGetLastError());
  return 0;
}

int pcx_post_output(pcx_post* p, void** out_dev, uint64_t* bytes) {
  if (!p) return set_error(PCX_E_INVALID, "pcx_post_output: null");
  if (out_dev) {
    if (int rc = p->ensure_out()) return rc;
    *out_dev = p->out_ptr();
  }
  if (bytes) *bytes = p->out_bytes;
  return 0;
}

int32_t pcx_post_plane_pitch(const pcx_post* p) { return p ? p->p.out_pitch : 0; }

int pcx_post_error_buffer(pcx_post* p, const uint8_t** errors_dev) {
  if (!p || !errors_dev) return set_error(PCX_E_INVALID, "pcx_post_error_buffer: bad arguments");
  *errors_dev = p->error.ptr;
  return 0;
}

int pcx_post_error_poll(pcx_post* p, void* stream, int32_t* seen) {
  if (!p) return set_error(PCX_E_INVALID, "pcx_post_error_poll: null");
  PCX_HIP(hipSetDevice(p->device));
  return p->error_poll.poll(p->error.ptr, p->p.batch, (hipStream_t)stream, seen);
}

int pcx_post_errors(pcx_post* p, uint8_t* errors_host) {
  if (!p || !errors_host) return set_error(PCX_E_INVALID, "pcx_post_errors: bad arguments");
  PCX_HIP(hipSetDevice(p->device));
  PCX_HIP(hipDeviceSynchronize());
  PCX_HIP(hipMemcpy(errors_host, p->error.ptr, (size_t)p->p.batch, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
