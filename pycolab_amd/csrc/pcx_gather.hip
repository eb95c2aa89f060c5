// pcx_gather.hip -- include/pcx.h pcx_gather_*: the node-level gather of what
// play() returns besides the observation (reward, reward_set, discount, done:
// 10 bytes per environment), for a host that drives every GPU of the node from
// ONE process through the C ABI -- no torch, no launcher.
//
// One RCCL communicator per engine's device (ncclCommInitAll), one
// ncclAllGather per device and call, grouped, on the engines' own streams:
// over xGMI every GPU sends its 10 B/env block straight to its seven peers.
// RCCL is bound at first use with dlopen("librccl.so.1"): a process that has
// already loaded an RCCL (PyTorch-ROCm ships its own copy under the same
// SONAME) keeps exactly that one, and a host that never gathers never loads it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <set>

#include "pcx_internal.h"

using pcx::set_error;

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

int load_rccl(Rccl** out) {
  static Rccl r;
  if (!r.so) {
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names)
      if ((r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.so) return set_error(PCX_E_UNSUPPORTED, "pcx_gather: librccl.so.1 not found (%s)", dlerror());
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(r.so, name); ok = ok && p; return p; };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(r.so); r.so = nullptr; return set_error(PCX_E_UNSUPPORTED, "pcx_gather: librccl lacks a symbol"); }
  }
  *out = &r;
  return 0;
}

#define PCX_NCCL(r, call)                                                                      \
  do {                                                                                         \
    ncclResult_t res__ = (call);                                                               \
    if (res__ != ncclSuccess)                                                                  \
      return set_error(PCX_E_HIP, "%s failed: %s (%s:%d)", #call, (r)->GetErrorString(res__), __FILE__, __LINE__); \
  } while (0)

// [reward i32[n] | discount f32[n] | reward_set u8[n] | done u8[n]] -- the layout of
// pycolab_amd.distributed (unpack_scalars) -- from the engine's four arrays
__global__ __launch_bounds__(256) void pcx_pack_scalars(const int32_t* reward, const float* discount,
                                                        const uint8_t* reward_set, const uint8_t* done, int64_t n,
                                                        uint8_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  reinterpret_cast<int32_t*>(out)[i] = reward[i];
  reinterpret_cast<float*>(out + 4 * n)[i] = discount[i];
  out[8 * n + i] = reward_set[i];
  out[9 * n + i] = done[i];
}

}  // namespace

struct pcx_gather {
  Rccl* rccl = nullptr;
  int n = 0;
  int64_t slot = 0;  // bytes per engine in every receive buffer
  std::vector<pcx_engine*> engines;
  std::vector<ncclComm_t> comms;
  std::vector<uint8_t*> send, recv;  // per engine, on its device
};

extern "C" {

void pcx_gather_destroy(pcx_gather* g) {
  if (!g) return;
  for (int i = 0; i < g->n; ++i) {
    (void)hipSetDevice(g->engines[i]->device);
    if (i < (int)g->comms.size() && g->comms[i]) (void)g->rccl->CommDestroy(g->comms[i]);
    if (i < (int)g->send.size() && g->send[i]) (void)hipFree(g->send[i]);
    if (i < (int)g->recv.size() && g->recv[i]) (void)hipFree(g->recv[i]);
  }
  delete g;
}

int pcx_gather_create(pcx_engine* const* engines, int32_t n, pcx_gather** out) {
  if (!engines || n < 1 || !out) return set_error(PCX_E_INVALID, "pcx_gather_create: bad arguments");
  std::set<int> devices;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    if (!engines[i]) return set_error(PCX_E_INVALID, "pcx_gather_create: null engine");
    if (!devices.insert(engines[i]->device).second)
      return set_error(PCX_E_INVALID, "pcx_gather_create: two engines on device %d (RCCL wants one rank per GPU)",
                       engines[i]->device);
    most = engines[i]->batch > most ? engines[i]->batch : most;
  }
  Rccl* r = nullptr;
  int rc = load_rccl(&r);
  if (rc) return rc;
  pcx_gather* g = new pcx_gather();
  g->rccl = r;
  g->n = n;
  g->slot = (10 * most + 15) / 16 * 16;  // shards may differ in length; typed views of every slot stay aligned
  g->engines.assign(engines, engines + n);
  g->send.assign(n, nullptr);
  g->recv.assign(n, nullptr);
  g->comms.assign(n, nullptr);
  std::vector<int> devlist(n);
  for (int i = 0; i < n; ++i) devlist[i] = engines[i]->device;
  ncclResult_t res = r->CommInitAll(g->comms.data(), n, devlist.data());
  if (res != ncclSuccess) {
    g->comms.assign(n, nullptr);
    pcx_gather_destroy(g);
    return set_error(PCX_E_HIP, "ncclCommInitAll over %d device(s) failed: %s", n, r->GetErrorString(res));
  }
  for (int i = 0; i < n; ++i) {
    hipError_t e1 = hipSetDevice(engines[i]->device);
    hipError_t e2 = e1 == hipSuccess ? hipMalloc(reinterpret_cast<void**>(&g->send[i]), (size_t)g->slot) : e1;
    hipError_t e3 = e2 == hipSuccess ? hipMalloc(reinterpret_cast<void**>(&g->recv[i]), (size_t)g->slot * n) : e2;
    hipError_t e4 = e3 == hipSuccess ? hipMemset(g->send[i], 0, (size_t)g->slot) : e3;
    if (e4 != hipSuccess) {
      pcx_gather_destroy(g);
      return set_error(PCX_E_HIP, "pcx_gather_create: device buffers: %s", hipGetErrorString(e4));
    }
  }
  *out = g;
  return 0;
}

int pcx_gather_scalars(pcx_gather* g, void* const* streams) {
  if (!g) return set_error(PCX_E_INVALID, "pcx_gather_scalars: null");
  std::vector<const uint8_t*> src(g->n);
  for (int i = 0; i < g->n; ++i) {
    pcx_engine* e = g->engines[i];
    if (!e->showtime || !e->out.reward)
      return set_error(PCX_E_STATE, "pcx_gather_scalars: engine %d has not been reset (its_showtime)", i);
    PCX_HIP(hipSetDevice(e->device));
    hipStream_t s = streams ? (hipStream_t)streams[i] : nullptr;
    const int64_t B = e->batch;
    const uint8_t* base = reinterpret_cast<const uint8_t*>(e->out.reward);
    const bool packed = reinterpret_cast<const uint8_t*>(e->out.discount) == base + 4 * B && e->out.reward_set == base + 8 * B &&
                        e->out.done == base + 9 * B && 10 * B == g->slot;
    if (packed) {  // the host bound one packed allocation (pycolab_amd.Engine does): send it in place
      src[i] = base;
    } else {
      hipLaunchKernelGGL(pcx_pack_scalars, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, e->out.reward, e->out.discount,
                         e->out.reward_set, e->out.done, B, g->send[i]);
      PCX_HIP(hipGetLastError());
      src[i] = g->send[i];
    }
  }
  PCX_NCCL(g->rccl, g->rccl->GroupStart());
  for (int i = 0; i < g->n; ++i) {
    hipStream_t s = streams ? (hipStream_t)streams[i] : nullptr;
    ncclResult_t res = g->rccl->AllGather(src[i], g->recv[i], (size_t)g->slot, ncclUint8, g->comms[i], s);
    if (res != ncclSuccess) {
      (void)g->rccl->GroupEnd();
      return set_error(PCX_E_HIP, "ncclAllGather (engine %d) failed: %s", i, g->rccl->GetErrorString(res));
    }
  }
  PCX_NCCL(g->rccl, g->rccl->GroupEnd());
  return 0;
}

int pcx_gather_buffers(pcx_gather* g, int32_t i, uint8_t** recv_dev, int64_t* slot_bytes) {
  if (!g || i < 0 || i >= g->n) return set_error(PCX_E_INVALID, "pcx_gather_buffers: bad arguments");
  if (recv_dev) *recv_dev = g->recv[i];
  if (slot_bytes) *slot_bytes = g->slot;
  return 0;
}

}  // extern "C"
