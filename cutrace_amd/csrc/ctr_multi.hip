// ctr_multi.hip — one frame row-tiled over the GPUs of a node, ONE gather to device 0 (include/cutrace_amd.h).
//
// The reference is single-GPU (inc/kernel.hpp:86-130 launches one kernel on the current device); this is
// the multi-device form of the same boundary, SURVEY.md §8(b) item 3 / §8(e): the scene is tiny, so it is
// replicated on every device; pixels are independent, so device d renders the interleaved row blocks
// {b : b mod n == d} (contiguous bands are badly balanced) into a compact buffer; the n-1 compact buffers
// travel to device 0 in ONE grouped RCCL send/recv over xGMI (7 messages on 7 distinct links at n = 8), a
// HIP kernel re-interleaves the row blocks into the row-major frame, and one D2H delivers it.
//
// Single process, one host thread, one stream per device — everything is asynchronous until the final
// synchronisation.  Built on the PUBLIC per-device entry points (ctr_scene_create, ctr_render_device), so
// a device's part is rendered by exactly the code the single-GPU path runs: results are bitwise those of
// ctr_render (tested).  RCCL (librccl.so, 570 MB) is dlopen'ed only when a group of >= 2 distinct devices
// is created; a group that lists one device several times (rehearsal on a one-GPU box) or a box without
// RCCL moves the parts with hipMemcpyPeerAsync instead.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "cutrace_amd.h"
#include "scene_device.h"

namespace {

int mfail(int code, const std::string &msg) {
  ctr_internal_set_error(msg.c_str());
  fprintf(stderr, "cutrace_amd: %s\n", msg.c_str());  // print-and-continue, like cudaCheck (inc/cuda.hpp:12-22)
  return code;
}
#define MHIP(expr)                                                                                          \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess) return mfail(CTR_E_HIP_BASE + (int)e_, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// ---- RCCL, bound at run time ----
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool load() {
    if (lib) return true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    Send = (decltype(Send))dlsym(lib, "ncclSend");
    Recv = (decltype(Recv))dlsym(lib, "ncclRecv");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
  }
};
Rccl g_rccl;

constexpr int SLOTS = 2;  // frames in flight (ctr_multi_submit): frame k+1 renders while frame k is gathered and assembled

struct Part {         // one device's share of the frame
  int device = 0;
  ctr_scene *scene = nullptr;
  hipStream_t stream = nullptr;          // render kernels of this device
  hipStream_t cstream = nullptr;         // this device's side of the gather (ncclSend); on device 0: receive, assemble, D2H
  hipEvent_t ev0[SLOTS] = {nullptr, nullptr}, ev1[SLOTS] = {nullptr, nullptr};  // around the render kernel (timing; ev1 = "rendered")
  hipEvent_t ev_moved[SLOTS] = {nullptr, nullptr};  // the compact buffer of the slot has left / has been read: it may be rendered into again
  hipEvent_t ev_moved0[SLOTS] = {nullptr, nullptr}; // the same, recorded by DEVICE 0's transfer stream (peer copies; an event belongs to the device that records it)
  hipEvent_t ev_cnt[SLOTS] = {nullptr, nullptr};    // the slot's counters have landed on the host
  float *buf[SLOTS] = {nullptr, nullptr};           // compact [depth px | color 3 px | normal 3 px] on `device`
  unsigned long long *counters[SLOTS] = {nullptr, nullptr};
  float *gathered[SLOTS] = {nullptr, nullptr};      // the same bytes on device 0 (parts >= 1; part 0 only for "rccl-self")
  uint64_t rows = 0;                     // rows of the current frame size this part owns
  ncclComm_t comm = nullptr;
};

// sources of the re-interleave: where part p's three compact buffers live on the assembling device
struct Reint {
  const float *depth[CTR_MULTI_MAX_DEVICES];
  const float *color[CTR_MULTI_MAX_DEVICES];
  const float *normal[CTR_MULTI_MAX_DEVICES];
  uint32_t n_parts, block_rows, w, h;
};

// Row y of the frame is local row k of part p:  p = (y / B) % n,  k = (y / B / n) * B + y % B.
// One workgroup copies one row of all three buffers (7 w floats), 16 bytes per lane where alignment allows.
__global__ __launch_bounds__(256) void reinterleave_rows(Reint R, float *__restrict__ out_depth, float *__restrict__ out_color,
                                                         float *__restrict__ out_normal) {
  const uint32_t y = blockIdx.x;
  const uint32_t blk = y / R.block_rows, p = blk % R.n_parts;
  const uint32_t k = (blk / R.n_parts) * R.block_rows + (y % R.block_rows);
  const float *src3[3] = {R.depth[p] + (size_t)k * R.w, R.color[p] + (size_t)k * R.w * 3, R.normal[p] + (size_t)k * R.w * 3};
  float *dst3[3] = {out_depth + (size_t)y * R.w, out_color + (size_t)y * R.w * 3, out_normal + (size_t)y * R.w * 3};
  const uint32_t len3[3] = {R.w, 3 * R.w, 3 * R.w};
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const float *a = src3[q];
    float *b = dst3[q];
    const uint32_t n = len3[q];
    if ((((uintptr_t)a | (uintptr_t)b) & 15u) == 0) {
      const uint32_t n4 = n / 4;
      for (uint32_t i = threadIdx.x; i < n4; i += blockDim.x) ((float4 *)b)[i] = ((const float4 *)a)[i];
      for (uint32_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) b[i] = a[i];
    } else {
      for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) b[i] = a[i];
    }
  }
}

uint64_t part_rows(uint64_t h, uint64_t block_rows, uint32_t part, uint32_t n_parts) {
  uint64_t n = 0;
  for (uint64_t b = part; b * block_rows < h; b += n_parts) {
    const uint64_t lo = b * block_rows, hi = lo + block_rows < h ? lo + block_rows : h;
    n += hi - lo;
  }
  return n;
}

bool pinned(const void *p) {
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// the device's address of a page-locked host range (nullptr: not page-locked, or not mapped for the current device)
float *device_view(float *host, size_t n) {
  if (!host || !n || !pinned(host) || !pinned(host + n - 1)) return nullptr;
  void *d = nullptr;
  if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return (float *)d;
}

}  // namespace

struct Frame {         // one slot of the pipeline
  bool busy = false;
  bool sync_done = false;     // the frame went through ctr_render (one device, synchronous): nothing to wait for
  ctr_render_stats one{};     // ... and these are its statistics
  uint64_t w = 0, h = 0;
  hipEvent_t ev_done = nullptr;   // device 0: the frame is assembled and on its way / in the caller's buffers
  std::chrono::high_resolution_clock::time_point t0;
};

struct ctr_multi {
  uint32_t variant = 0;
  double single_ms = -1.0;    // kernel ms of the last waited frame when it went through ctr_render (one device), else < 0
  std::vector<Part> parts;
  uint64_t w = 0, h = 0;
  uint64_t cap_px = 0;        // pixels each compact buffer can hold
  float *frame[SLOTS] = {nullptr, nullptr};  // device 0: the re-interleaved frame [depth | color | normal]
  uint64_t frame_px = 0;
  unsigned long long *h_counters = nullptr;  // pinned, SLOTS x 16 words per part
  bool use_rccl = false;
  std::string transport = "single";
  Frame frames[SLOTS];
  int head = 0, inflight = 0;  // oldest frame in flight, number in flight
  int last = 0;                // slot of the frame waited for last (ctr_multi_kernel_ms)
};

namespace {

// after an error: nothing may still be running on buffers the caller could free or reuse
int fail_sync(ctr_multi *m, int st) {
  for (Part &P : m->parts) {
    if (hipSetDevice(P.device) != hipSuccess) continue;
    if (P.stream) (void)hipStreamSynchronize(P.stream);
    if (P.cstream) (void)hipStreamSynchronize(P.cstream);
  }
  for (Frame &f : m->frames) f.busy = false;
  m->inflight = 0;
  (void)hipGetLastError();
  return st;
}

int ensure_buffers(ctr_multi *m, uint64_t block_rows) {
  const uint32_t n = (uint32_t)m->parts.size();
  uint64_t cap = 0;
  for (uint32_t p = 0; p < n; p++) {
    m->parts[p].rows = part_rows(m->h, block_rows, p, n);
    cap = cap > m->parts[p].rows * m->w ? cap : m->parts[p].rows * m->w;
  }
  if (cap == 0) cap = 1;
  if (cap > m->cap_px) {
    if (m->inflight) return mfail(CTR_E_INVALID, "ctr_multi: the frame grew while frames are in flight (ctr_multi_wait first)");
    m->cap_px = 0;  // (an allocation that fails below leaves null buffers: the next call must come back here whatever its size)
    for (uint32_t p = 0; p < n; p++) {
      Part &P = m->parts[p];
      for (int q = 0; q < SLOTS; q++) {
        MHIP(hipSetDevice(P.device));
        if (P.buf[q]) (void)hipFree(P.buf[q]);
        P.buf[q] = nullptr;
        MHIP(hipMalloc((void **)&P.buf[q], sizeof(float) * 7 * cap));
        // (part 0 has a gathered copy only under "rccl-self"; it is sized like everybody's and regrown with them —
        //  round 2 grew only the others', and a larger frame then overran it)
        MHIP(hipSetDevice(m->parts[0].device));
        if (P.gathered[q]) (void)hipFree(P.gathered[q]);
        P.gathered[q] = nullptr;
        if (p > 0 || (n == 1 && m->use_rccl)) MHIP(hipMalloc((void **)&P.gathered[q], sizeof(float) * 7 * cap));
      }
    }
    m->cap_px = cap;
  }
  if (m->w * m->h > m->frame_px && n > 1) {
    if (m->inflight) return mfail(CTR_E_INVALID, "ctr_multi: the frame grew while frames are in flight (ctr_multi_wait first)");
    MHIP(hipSetDevice(m->parts[0].device));
    m->frame_px = 0;
    for (int q = 0; q < SLOTS; q++) {
      if (m->frame[q]) (void)hipFree(m->frame[q]);
      m->frame[q] = nullptr;
      MHIP(hipMalloc((void **)&m->frame[q], sizeof(float) * 7 * m->w * m->h));
    }
    m->frame_px = m->w * m->h;
  }
  return CTR_OK;
}

}  // namespace

extern "C" {

int ctr_multi_create(const ctr_scene_desc *desc, const int *devices, int n_devices, ctr_multi **out) {
  if (!desc || !devices || !out || n_devices < 1 || n_devices > CTR_MULTI_MAX_DEVICES)
    return mfail(CTR_E_INVALID, "ctr_multi_create: bad argument (1.." + std::to_string(CTR_MULTI_MAX_DEVICES) + " devices)");
  *out = nullptr;
  auto *m = new ctr_multi();
  m->w = desc->cam.w;
  m->h = desc->cam.h;
  m->parts.resize(n_devices);
  bool distinct = true;
  for (int i = 0; i < n_devices; i++)
    for (int j = 0; j < i; j++)
      if (devices[i] == devices[j]) distinct = false;
  for (int i = 0; i < n_devices; i++) {
    Part &P = m->parts[i];
    P.device = devices[i];
    int st = ctr_scene_create(desc, devices[i], &P.scene);  // the scene is replicated: it is < 1 MB
    if (st) { ctr_multi_destroy(m); return st; }
    hipError_t e = hipSetDevice(devices[i]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&P.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&P.cstream, hipStreamNonBlocking);
    for (int q = 0; q < SLOTS && e == hipSuccess; q++) {
      e = hipEventCreate(&P.ev0[q]);
      if (e == hipSuccess) e = hipEventCreate(&P.ev1[q]);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&P.ev_moved[q], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&P.ev_cnt[q], hipEventDisableTiming);
      if (e == hipSuccess) e = hipSetDevice(devices[0]);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&P.ev_moved0[q], hipEventDisableTiming);
      if (e == hipSuccess) e = hipSetDevice(devices[i]);
      if (e == hipSuccess) e = hipMalloc((void **)&P.counters[q], 16 * sizeof(unsigned long long));
    }
    if (e != hipSuccess) { ctr_multi_destroy(m); return mfail(CTR_E_HIP_BASE + (int)e, std::string("ctr_multi_create: ") + hipGetErrorString(e)); }
  }
  hipError_t e = hipHostMalloc((void **)&m->h_counters, sizeof(unsigned long long) * 16 * n_devices * SLOTS, hipHostMallocDefault);
  if (e == hipSuccess) e = hipSetDevice(devices[0]);
  for (int q = 0; q < SLOTS && e == hipSuccess; q++) e = hipEventCreateWithFlags(&m->frames[q].ev_done, hipEventDisableTiming);
  if (e != hipSuccess) { ctr_multi_destroy(m); return mfail(CTR_E_HIP_BASE + (int)e, "ctr_multi_create: hipHostMalloc / events"); }
  const char *force = getenv("CUTRACE_MULTI_TRANSPORT");  // "peer" forces hipMemcpyPeerAsync; "rccl-self" see below
  if (n_devices == 1 && force && !strcmp(force, "rccl-self") && g_rccl.load()) {
    // Self-test of the RCCL plumbing on a one-GPU box: a one-rank communicator, and the frame travels through one
    // grouped ncclSend / ncclRecv to and from rank 0 itself (library load, communicator, group call, stream order).
    ncclComm_t c = nullptr;
    ncclResult_t r = g_rccl.CommInitAll(&c, 1, devices);
    if (r == ncclSuccess) {
      m->parts[0].comm = c;
      m->use_rccl = true;
      m->transport = "rccl-self";
    } else {
      mfail(CTR_E_HIP_BASE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
    }
  }
  if (n_devices > 1) {
    if (distinct && !(force && !strcmp(force, "peer")) && g_rccl.load()) {
      std::vector<ncclComm_t> comms(n_devices);
      ncclResult_t r = g_rccl.CommInitAll(comms.data(), n_devices, devices);
      if (r == ncclSuccess) {
        for (int i = 0; i < n_devices; i++) m->parts[i].comm = comms[i];
        m->use_rccl = true;
        m->transport = "rccl";
      } else {
        mfail(CTR_E_HIP_BASE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r) + " — falling back to peer copies");
      }
    }
    if (!m->use_rccl) {
      m->transport = "peer-copy";
      if (distinct) {
        (void)hipSetDevice(devices[0]);
        for (int i = 1; i < n_devices; i++) {
          int can = 0;
          if (hipDeviceCanAccessPeer(&can, devices[0], devices[i]) == hipSuccess && can) {
            if (hipDeviceEnablePeerAccess(devices[i], 0) != hipSuccess) (void)hipGetLastError();  // already enabled is fine
          }
        }
      }
    }
  }
  *out = m;
  return CTR_OK;
}

void ctr_multi_destroy(ctr_multi *m) {
  if (!m) return;
  for (Part &P : m->parts) {  // nothing may be running when the buffers go
    if (hipSetDevice(P.device) != hipSuccess) continue;
    if (P.stream) (void)hipStreamSynchronize(P.stream);
    if (P.cstream) (void)hipStreamSynchronize(P.cstream);
  }
  for (Part &P : m->parts) {
    (void)hipSetDevice(P.device);
    if (P.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(P.comm);
    for (int q = 0; q < SLOTS; q++) {
      if (P.buf[q]) (void)hipFree(P.buf[q]);
      if (P.counters[q]) (void)hipFree(P.counters[q]);
      if (P.ev0[q]) (void)hipEventDestroy(P.ev0[q]);
      if (P.ev1[q]) (void)hipEventDestroy(P.ev1[q]);
      if (P.ev_moved[q]) (void)hipEventDestroy(P.ev_moved[q]);
      if (P.ev_moved0[q]) (void)hipEventDestroy(P.ev_moved0[q]);
      if (P.ev_cnt[q]) (void)hipEventDestroy(P.ev_cnt[q]);
    }
    if (P.stream) (void)hipStreamDestroy(P.stream);
    if (P.cstream) (void)hipStreamDestroy(P.cstream);
    if (P.scene) ctr_scene_destroy(P.scene);
  }
  if (!m->parts.empty()) {
    (void)hipSetDevice(m->parts[0].device);
    for (Part &P : m->parts)
      for (int q = 0; q < SLOTS; q++)
        if (P.gathered[q]) (void)hipFree(P.gathered[q]);
    for (int q = 0; q < SLOTS; q++) {
      if (m->frame[q]) (void)hipFree(m->frame[q]);
      if (m->frames[q].ev_done) (void)hipEventDestroy(m->frames[q].ev_done);
    }
  }
  if (m->h_counters) (void)hipHostFree(m->h_counters);
  delete m;
}

int ctr_multi_devices(const ctr_multi *m) { return m ? (int)m->parts.size() : 0; }
const char *ctr_multi_transport(const ctr_multi *m) { return m ? m->transport.c_str() : ""; }

int ctr_multi_size(const ctr_multi *m, uint64_t *w, uint64_t *h) {
  if (!m) return mfail(CTR_E_INVALID, "null group");
  if (w) *w = m->w;
  if (h) *h = m->h;
  return CTR_OK;
}

int ctr_multi_set_size(ctr_multi *m, uint64_t w, uint64_t h) {
  if (!m) return mfail(CTR_E_INVALID, "null group");
  if (m->inflight) return mfail(CTR_E_INVALID, "ctr_multi_set_size: frames in flight (ctr_multi_wait first)");
  for (Part &P : m->parts) {
    int st = ctr_scene_set_size(P.scene, w, h);
    if (st) return st;
  }
  m->w = w;
  m->h = h;
  return CTR_OK;
}

int ctr_multi_set_variant(ctr_multi *m, uint32_t bits) {
  if (!m) return mfail(CTR_E_INVALID, "null group");
  for (Part &P : m->parts) {
    int st = ctr_set_variant(P.scene, bits);
    if (st) return st;
  }
  m->variant = bits;
  return CTR_OK;
}

// ---- the pipeline ----
// ctr_multi_submit queues one frame and returns; ctr_multi_wait returns when the OLDEST queued frame is in the caller's
// buffers.  Two frames may be in flight, each in its own slot of every buffer.  Per device a render stream and a
// transfer stream: the kernel of frame k+1 starts as soon as that of frame k is done, while frame k's compact buffers
// travel to device 0 (one grouped ncclSend / ncclRecv), are re-interleaved there and leave for the host.  What orders them:
//   ev1[slot]       rendered      -> the device's send / device 0's assembly may read the compact buffer
//   ev_moved[slot]  sent / read   -> the render stream may write the compact buffer again (frame k+2)
//   ev_done[slot]   assembled, copied out -> ctr_multi_wait
// ctr_render_multi = submit + wait.  One device with a page-locked or pageable destination goes through ctr_render
// (host delivery / its own copies) and is synchronous; "page-locked" is what makes the final D2H asynchronous at all.
int ctr_multi_submit(ctr_multi *m, float fudge, int bounces, uint64_t block_rows, float *depth, float *color3, float *normal3) {
  if (!m) return mfail(CTR_E_INVALID, "null group");
  if (m->inflight >= SLOTS) return mfail(CTR_E_INVALID, "ctr_multi_submit: two frames are in flight already (ctr_multi_wait first)");
  if (block_rows == 0) block_rows = 8;
  const uint32_t n = (uint32_t)m->parts.size();
  int st = ensure_buffers(m, block_rows);
  if (st) return st;
  const int q = (m->head + m->inflight) % SLOTS;
  Frame &F = m->frames[q];
  F.t0 = std::chrono::high_resolution_clock::now();
  F.sync_done = false;
  F.w = m->w;
  F.h = m->h;
  const uint64_t w = m->w, h = m->h, fpx = w * h;
  Part &P0 = m->parts[0];
  // Page-locked destinations are written by device 0 itself: through ctr_render's host delivery when there is one
  // device, by the re-interleave kernel (whole rows, 16 bytes per lane) otherwise — no frame-sized D2H after it.
  MHIP(hipSetDevice(P0.device));
  float *zd = nullptr, *zc = nullptr, *zn = nullptr;
  const bool direct = fpx && !(m->variant & CTR_VAR_NO_DIRECT) && (zd = device_view(depth, fpx)) &&
                      (zc = device_view(color3, 3 * fpx)) && (zn = device_view(normal3, 3 * fpx));
  const bool dest_pinned = fpx && depth && color3 && normal3 && pinned(depth) && pinned(depth + fpx - 1) && pinned(color3) &&
                           pinned(color3 + 3 * fpx - 1) && pinned(normal3) && pinned(normal3 + 3 * fpx - 1);
  if (n == 1 && !m->use_rccl && (direct || !dest_pinned)) {
    // one device: ctr_render is the better path both for page-locked destinations (delivered by the kernel) and for
    // pageable ones (its three plain copies: 2.2 ms per 1080p frame where queuing them on a side stream took 4.8)
    // An earlier frame may have taken the asynchronous path (a page-locked destination under CTR_VAR_NO_DIRECT) and still
    // be running on P0's streams, which do not synchronise with the null stream ctr_render launches on: two launches on
    // ONE scene handle would share its counter shards, cost table and dispatch order.  Drain it first; the
    // frame stays queued (its events have fired by then, ctr_multi_wait returns it at once).
    if (m->inflight) {
      hipError_t e = hipStreamSynchronize(P0.stream);
      if (e == hipSuccess) e = hipStreamSynchronize(P0.cstream);
      if (e != hipSuccess) return fail_sync(m, mfail(CTR_E_HIP_BASE + (int)e, std::string("ctr_multi_submit: ") + hipGetErrorString(e)));
    }
    if ((st = ctr_render(P0.scene, fudge, bounces, nullptr, depth, color3, normal3, &F.one))) return fail_sync(m, st);
    F.sync_done = true;
    F.busy = true;
    m->inflight++;
    return CTR_OK;
  }
  unsigned long long *hc = m->h_counters + (size_t)16 * n * q;
  // ---- 1. every device renders its interleaved row blocks into its compact buffer of this slot ----
  for (uint32_t p = 0; p < n; p++) {
    Part &P = m->parts[p];
    if ((st = (int)hipSetDevice(P.device))) return fail_sync(m, mfail(CTR_E_HIP_BASE + st, "hipSetDevice"));
    hipError_t e = hipStreamWaitEvent(P.stream, P.ev_moved[q], 0);  // (never recorded yet: no wait)
    if (e == hipSuccess) e = hipStreamWaitEvent(P.stream, P.ev_moved0[q], 0);
    if (e == hipSuccess) e = hipMemsetAsync(P.counters[q], 0, 16 * sizeof(unsigned long long), P.stream);
    if (e == hipSuccess) e = hipEventRecord(P.ev0[q], P.stream);
    if (e != hipSuccess) return fail_sync(m, mfail(CTR_E_HIP_BASE + (int)e, std::string("ctr_multi_submit: ") + hipGetErrorString(e)));
    const uint64_t px = P.rows * w;
    if (px) {
      ctr_rows r{0, h, block_rows, p, n};
      st = ctr_render_device(P.scene, fudge, bounces, &r, P.buf[q], P.buf[q] + px, P.buf[q] + 4 * px, P.counters[q], P.stream);
      if (st) return fail_sync(m, st);
    }
    e = hipEventRecord(P.ev1[q], P.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hc + 16 * p, P.counters[q], 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, P.stream);
    if (e == hipSuccess) e = hipEventRecord(P.ev_cnt[q], P.stream);
    if (e != hipSuccess) return fail_sync(m, mfail(CTR_E_HIP_BASE + (int)e, std::string("ctr_multi_submit: ") + hipGetErrorString(e)));
  }
  // ---- 2. one gather to device 0, on the transfer streams ----
  hipError_t he = hipSuccess;
  if (n > 1) {
    if (m->use_rccl) {
      for (uint32_t p = 1; p < n && he == hipSuccess; p++) {  // a device sends once its own kernel is done
        he = hipSetDevice(m->parts[p].device);
        if (he == hipSuccess) he = hipStreamWaitEvent(m->parts[p].cstream, m->parts[p].ev1[q], 0);
      }
      if (he != hipSuccess) return fail_sync(m, mfail(CTR_E_HIP_BASE + (int)he, std::string("RCCL gather: ") + hipGetErrorString(he)));
      // (no early return between GroupStart and GroupEnd: an open group would poison every later call)
      ncclResult_t r = g_rccl.GroupStart();
      for (uint32_t p = 1; p < n && r == ncclSuccess && he == hipSuccess; p++) {
        Part &P = m->parts[p];
        const size_t cnt = (size_t)(7 * P.rows * w);
        if (!cnt) continue;
        he = hipSetDevice(P.device);
        if (he != hipSuccess) break;
        r = g_rccl.Send(P.buf[q], cnt, ncclFloat, 0, P.comm, P.cstream);
        if (r != ncclSuccess) break;
        he = hipSetDevice(P0.device);
        if (he != hipSuccess) break;
        r = g_rccl.Recv(P.gathered[q], cnt, ncclFloat, (int)p, P0.comm, P0.cstream);
      }
      const ncclResult_t r2 = g_rccl.GroupEnd();
      if (r == ncclSuccess) r = r2;
      if (he != hipSuccess) return fail_sync(m, mfail(CTR_E_HIP_BASE + (int)he, std::string("RCCL gather: ") + hipGetErrorString(he)));
      if (r != ncclSuccess) return fail_sync(m, mfail(CTR_E_HIP_BASE, std::string("RCCL gather: ") + g_rccl.GetErrorString(r)));
      for (uint32_t p = 1; p < n && he == hipSuccess; p++) {  // sent: the compact buffer may be rendered into again
        he = hipSetDevice(m->parts[p].device);
        if (he == hipSuccess) he = hipEventRecord(m->parts[p].ev_moved[q], m->parts[p].cstream);
      }
    } else {
      he = hipSetDevice(P0.device);
      for (uint32_t p = 1; p < n && he == hipSuccess; p++) {
        Part &P = m->parts[p];
        const size_t bytes = sizeof(float) * 7 * P.rows * w;
        he = hipStreamWaitEvent(P0.cstream, P.ev1[q], 0);
        if (bytes && he == hipSuccess) {
          if (P.device == P0.device) he = hipMemcpyAsync(P.gathered[q], P.buf[q], bytes, hipMemcpyDeviceToDevice, P0.cstream);
          else he = hipMemcpyPeerAsync(P.gathered[q], P0.device, P.buf[q], P.device, bytes, P0.cstream);
        }
        if (he == hipSuccess) he = hipEventRecord(P.ev_moved0[q], P0.cstream);
      }
    }
    if (he != hipSuccess) return fail_sync(m, mfail(CTR_E_HIP_BASE + (int)he, std::string("gather: ") + hipGetErrorString(he)));
  }
  // device 0's transfer stream also needs device 0's own part
  he = hipSetDevice(P0.device);
  if (he == hipSuccess) he = hipStreamWaitEvent(P0.cstream, P0.ev1[q], 0);
  if (he != hipSuccess) return fail_sync(m, mfail(CTR_E_HIP_BASE + (int)he, std::string("ctr_multi_submit: ") + hipGetErrorString(he)));
  const float *self_result = nullptr;
  if (n == 1 && m->use_rccl && fpx) {  // "rccl-self": the frame goes through RCCL once, rank 0 -> rank 0
    ncclResult_t r = g_rccl.GroupStart();
    if (r == ncclSuccess) r = g_rccl.Send(P0.buf[q], (size_t)(7 * fpx), ncclFloat, 0, P0.comm, P0.cstream);
    if (r == ncclSuccess) r = g_rccl.Recv(P0.gathered[q], (size_t)(7 * fpx), ncclFloat, 0, P0.comm, P0.cstream);
    const ncclResult_t r2 = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return fail_sync(m, mfail(CTR_E_HIP_BASE, std::string("RCCL self send/recv: ") + g_rccl.GetErrorString(r)));
    self_result = P0.gathered[q];
  }
  // ---- 3. re-interleave on device 0, 4. one D2H ----
  const float *result = self_result ? self_result : P0.buf[q];  // n == 1: the compact buffer IS the frame
  if (n > 1 && fpx) {
    Reint R{};
    for (uint32_t p = 0; p < n; p++) {
      const float *src = p == 0 ? P0.buf[q] : m->parts[p].gathered[q];
      const size_t px = (size_t)(m->parts[p].rows * w);
      R.depth[p] = src;
      R.color[p] = src + px;
      R.normal[p] = src + 4 * px;
    }
    R.n_parts = n;
    R.block_rows = (uint32_t)block_rows;
    R.w = (uint32_t)w;
    R.h = (uint32_t)h;
    float *fr = m->frame[q];
    if (direct) hipLaunchKernelGGL(reinterleave_rows, dim3((uint32_t)h), dim3(256), 0, P0.cstream, R, zd, zc, zn);
    else hipLaunchKernelGGL(reinterleave_rows, dim3((uint32_t)h), dim3(256), 0, P0.cstream, R, fr, fr + fpx, fr + 4 * fpx);
    he = hipGetLastError();
    if (he != hipSuccess) return fail_sync(m, mfail(CTR_E_HIP_BASE + (int)he, std::string("reinterleave_rows: ") + hipGetErrorString(he)));
    result = fr;
  }
  if (fpx && !(direct && n > 1)) {
    const bool packed = depth && color3 == depth + fpx && normal3 == color3 + 3 * fpx;
    if (packed && dest_pinned) {
      he = hipMemcpyAsync(depth, result, sizeof(float) * 7 * fpx, hipMemcpyDeviceToHost, P0.cstream);
    } else {
      // (hipMemcpyAsync into pageable memory is staged by the runtime and returns when the copy is done)
      if (depth) he = hipMemcpyAsync(depth, result, sizeof(float) * fpx, hipMemcpyDeviceToHost, P0.cstream);
      if (color3 && he == hipSuccess) he = hipMemcpyAsync(color3, result + fpx, sizeof(float) * 3 * fpx, hipMemcpyDeviceToHost, P0.cstream);
      if (normal3 && he == hipSuccess) he = hipMemcpyAsync(normal3, result + 4 * fpx, sizeof(float) * 3 * fpx, hipMemcpyDeviceToHost, P0.cstream);
    }
    if (he != hipSuccess) return fail_sync(m, mfail(CTR_E_HIP_BASE + (int)he, std::string("frame copy-out: ") + hipGetErrorString(he)));
  }
  // device 0's compact buffer has been read once the assembly (or, n == 1, the copy-out) is through
  he = hipEventRecord(P0.ev_moved[q], P0.cstream);
  if (he == hipSuccess) he = hipEventRecord(F.ev_done, P0.cstream);
  if (he != hipSuccess) return fail_sync(m, mfail(CTR_E_HIP_BASE + (int)he, std::string("ctr_multi_submit: ") + hipGetErrorString(he)));
  F.busy = true;
  m->inflight++;
  return CTR_OK;
}

int ctr_multi_wait(ctr_multi *m, ctr_render_stats *stats) {
  if (!m) return mfail(CTR_E_INVALID, "null group");
  if (m->inflight == 0) return mfail(CTR_E_INVALID, "ctr_multi_wait: no frame in flight");
  const int q = m->head;
  Frame &F = m->frames[q];
  const uint32_t n = (uint32_t)m->parts.size();
  m->single_ms = -1.0;
  if (F.sync_done) {
    m->single_ms = F.one.kernel_ms;
    if (stats) {
      *stats = F.one;
      stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - F.t0).count();
    }
  } else {
    hipError_t e = hipSetDevice(m->parts[0].device);
    if (e == hipSuccess) e = hipEventSynchronize(F.ev_done);
    for (uint32_t p = 0; p < n && e == hipSuccess; p++) {
      e = hipSetDevice(m->parts[p].device);
      if (e == hipSuccess) e = hipEventSynchronize(m->parts[p].ev_cnt[q]);
      if (e == hipSuccess) e = hipEventSynchronize(m->parts[p].ev_moved[q]);
      if (e == hipSuccess) e = hipEventSynchronize(m->parts[p].ev_moved0[q]);
    }
    if (e != hipSuccess) return fail_sync(m, mfail(CTR_E_HIP_BASE + (int)e, std::string("ctr_multi_wait: ") + hipGetErrorString(e)));
    auto t1 = std::chrono::high_resolution_clock::now();
    if (stats) {
      memset(stats, 0, sizeof(*stats));
      const unsigned long long *hc = m->h_counters + (size_t)16 * n * q;
      uint32_t bits = 0;
      for (uint32_t p = 0; p < n; p++) {
        float ms = 0.f;
        MHIP(hipSetDevice(m->parts[p].device));
        MHIP(hipEventElapsedTime(&ms, m->parts[p].ev0[q], m->parts[p].ev1[q]));
        stats->kernel_ms = stats->kernel_ms > ms ? stats->kernel_ms : ms;  // the slowest device's kernel
        stats->ray_count += hc[16 * p + 0];
        const uint32_t b = (uint32_t)hc[16 * p + 1];
        bits = b > bits ? b : bits;
      }
      memcpy(&stats->max_depth, &bits, 4);
      stats->rows = F.h;
      stats->total_ms = std::chrono::duration<double, std::milli>(t1 - F.t0).count();
    }
  }
  F.busy = false;
  m->last = q;
  m->head = (m->head + 1) % SLOTS;
  m->inflight--;
  return CTR_OK;
}

int ctr_render_multi(ctr_multi *m, float fudge, int bounces, uint64_t block_rows, float *depth, float *color3,
                     float *normal3, ctr_render_stats *stats) {
  if (!m) return mfail(CTR_E_INVALID, "null group");
  if (m->inflight) return mfail(CTR_E_INVALID, "ctr_render_multi: frames are in flight (ctr_multi_wait first)");
  int st = ctr_multi_submit(m, fudge, bounces, block_rows, depth, color3, normal3);
  if (st) return st;
  return ctr_multi_wait(m, stats);
}

int ctr_reinterleave_device(const ctr_reint_part *parts, uint32_t n_parts, uint64_t block_rows, uint64_t w, uint64_t h,
                            void *d_depth, void *d_color3, void *d_normal3, void *hip_stream) {
  if (!parts || n_parts == 0 || n_parts > CTR_MULTI_MAX_DEVICES || block_rows == 0 || !d_depth || !d_color3 || !d_normal3)
    return mfail(CTR_E_INVALID, "ctr_reinterleave_device: bad argument");
  if (w == 0 || h == 0) return CTR_OK;
  if (w > 0x7FFFFFFFull || h > 0x7FFFFFFFull) return mfail(CTR_E_INVALID, "ctr_reinterleave_device: image too large");
  Reint R{};
  for (uint32_t p = 0; p < n_parts; p++) {
    R.depth[p] = (const float *)parts[p].d_depth;
    R.color[p] = (const float *)parts[p].d_color3;
    R.normal[p] = (const float *)parts[p].d_normal3;
  }
  R.n_parts = n_parts;
  R.block_rows = (uint32_t)block_rows;
  R.w = (uint32_t)w;
  R.h = (uint32_t)h;
  hipLaunchKernelGGL(reinterleave_rows, dim3((uint32_t)h), dim3(256), 0, (hipStream_t)hip_stream, R, (float *)d_depth,
                     (float *)d_color3, (float *)d_normal3);
  MHIP(hipGetLastError());
  return CTR_OK;
}

int ctr_multi_kernel_ms(ctr_multi *m, double *ms_per_device, int capacity) {
  if (!m || !ms_per_device) return mfail(CTR_E_INVALID, "ctr_multi_kernel_ms: null argument");
  if (m->single_ms >= 0.0 && capacity > 0) {
    ms_per_device[0] = m->single_ms;
    return CTR_OK;
  }
  for (int p = 0; p < (int)m->parts.size() && p < capacity; p++) {
    float ms = 0.f;
    MHIP(hipSetDevice(m->parts[p].device));
    MHIP(hipEventElapsedTime(&ms, m->parts[p].ev0[m->last], m->parts[p].ev1[m->last]));
    ms_per_device[p] = ms;
  }
  return CTR_OK;
}

}  // extern "C"
