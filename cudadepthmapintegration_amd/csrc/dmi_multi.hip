// dmi_multi.hip -- one fusion over several MI355X of a node: the dmi_multi_* part of include/dmi.h.
//
// The reference drives one GPU on the default stream (Reconstruction/CudaReconstruction.cu:302-386); the seam this
// plugs into is the pair of driver calls at Reconstruction/vtkCudaReconstructionFilter.cxx:171-176.  The fusion is a
// sum over depth maps of independent per-voxel terms (cu:211), so it shards by views (every rank a private grid, one
// RCCL all-reduce of the grid over xGMI) or by z-slabs (no exchange at all).  Everything here is host code layered on
// the single-GPU C ABI (dmi_create / dmi_add_views / dmi_fuse_slab ...), HIP streams and events, and RCCL, which is
// loaded with dlopen on first use so that single-GPU users do not depend on it.
//
// A third exchange needs neither RCCL nor a compute unit for the transfers (DMI_EXCHANGE_PEER_COPY, ranks of one process):
// every slab is cut into `world` chunks, rank r collects the other ranks' chunk r in a staging buffer by peer-to-peer copies
// (hipMemcpyPeerAsync: the SDMA engines), adds them to its own in rank order with a small kernel queued behind its fusion,
// and copies the sum back into every other rank's grid -- a direct reduce-scatter + all-gather over the point-to-point
// xGMI links, deterministic (RCCL's ring order is not), and independent of whether a persistent fusion kernel leaves
// wave slots for a collective's kernels.
//
// Streams per rank: `compute` (handed to the rank's dmi_context: uploads' consumers, classification, fusion) and
// `comm` (RCCL, or the peer copies).  dmi_multi_fuse fuses the grid slab by slab on `compute`; after each slab an event lets `comm` start
// that slab's all-reduce while `compute` goes on with the next slab; at the end `compute` waits for `comm`.
#include "../../include/dmi.h"
#include "fusion_kernels.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_multi_create_error;

constexpr int kMaxSlabs = 64;
static_assert(DMI_Z_SLAB_ALIGNMENT % dmi::kMaxColumnHeight == 0, "z-slab boundaries must not cut a voxel column");
constexpr int kZSlabAlignment = DMI_Z_SLAB_ALIGNMENT;  // DMI_PARTITION_Z_SLABS: slab heights are multiples of the tallest voxel column (16)

// ---- RCCL through dlopen: the signatures come from <rccl/rccl.h>, the symbols from librccl.so.1 at run time --------
struct Rccl {
  void *handle = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};

// One table per process, filled once.  A process that already holds an RCCL (e.g. torch's own copy, same SONAME) gets
// that one back from dlopen, so there is never a second instance next to it.
Rccl *load_rccl() {
  static Rccl api;
  static bool tried = false;
  if (tried) return &api;
  tried = true;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (api.handle) break;
  }
  if (!api.handle) {
    const char *why = dlerror();
    api.error = std::string("librccl.so.1 cannot be loaded (") + (why ? why : "unknown reason") + ")";
    return &api;
  }
  bool ok = true;
  auto sym = [&](const char *name) {
    void *p = dlsym(api.handle, name);
    if (!p) {
      ok = false;
      api.error = std::string("librccl.so.1 lacks ") + name;
    }
    return p;
  };
  api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
  api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
  api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
  api.ReduceScatter = reinterpret_cast<decltype(api.ReduceScatter)>(sym("ncclReduceScatter"));
  api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
  api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  if (!ok) {
    dlclose(api.handle);
    api.handle = nullptr;
  }
  return &api;
}

// One rank = one GPU of the fusion.
struct Rank {
  int32_t device = 0;
  int32_t rank = 0;
  dmi_context *ctx = nullptr;  // nullptr: this rank owns no cell layer (Z_SLABS on a short grid)
  hipStream_t compute = nullptr, comm = nullptr;
  ncclComm_t nccl = nullptr;
  hipEvent_t slab_done[kMaxSlabs] = {};
  hipEvent_t exchanged = nullptr, step_start = nullptr, step_stop = nullptr;
  // DMI_EXCHANGE_PEER_COPY: the other ranks' parts of this rank's chunks ((world - 1) x the chunk size, every slab its own
  // place), and per slab: this rank's outgoing copies are done / its chunk is summed / the sum has reached every rank
  void *staging = nullptr;
  hipEvent_t copied[kMaxSlabs] = {}, summed[kMaxSlabs] = {}, gathered[kMaxSlabs] = {};
  int32_t z_first = 0, z_count = 0;  // cell layers of this rank's context (the whole grid under VIEWS)
  int64_t n_views = 0;               // views resident in ctx
};

}  // namespace

struct dmi_multi_context {
  dmi_grid_desc grid{};
  dmi_ray_potential ray{};
  dmi_multi_options opt{};
  int32_t world = 0;
  bool one_process = true;
  std::vector<Rank> ranks;  // the ranks this process drives, consecutive
  Rccl *rccl = nullptr;
  int64_t n_voxels = 0;
  int64_t n_views_total = 0;
  int32_t slab_z[kMaxSlabs] = {}, slab_n[kMaxSlabs] = {};
  int32_t n_slab_ranges = 0;
  bool step_pending = false;
  double fuse_ms_seen = 0.0;  // rank 0's dmi_timings.total_fuse_kernel_ms when the last step was accounted
  dmi_multi_timings timings{};
  std::string err;
};

namespace {

int mfail(dmi_multi_context *m, int code, const std::string &msg) {
  if (m)
    m->err = msg;
  else
    g_multi_create_error = msg;
  return code;
}

template <typename Body>
int guarded(dmi_multi_context *m, const char *entry, Body &&body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    try {
      return mfail(m, DMI_ERR_OUT_OF_MEMORY, std::string(entry) + ": host allocation failed");
    } catch (...) {
      return DMI_ERR_OUT_OF_MEMORY;
    }
  } catch (...) {
    try {
      return mfail(m, DMI_ERR_STATE, std::string(entry) + ": unexpected C++ exception");
    } catch (...) {
      return DMI_ERR_STATE;
    }
  }
}

#define DMI_M_HIP(m, call)                                                                     \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      (void)hipGetLastError();                                                                 \
      return mfail(m, e_ == hipErrorOutOfMemory ? DMI_ERR_OUT_OF_MEMORY : DMI_ERR_DEVICE,      \
                   std::string(#call) + ": " + hipGetErrorString(e_));                         \
    }                                                                                          \
  } while (0)

#define DMI_M_NCCL(m, call)                                                                                     \
  do {                                                                                                          \
    ncclResult_t r_ = (call);                                                                                   \
    if (r_ != ncclSuccess)                                                                                      \
      return mfail(m, DMI_ERR_DEVICE, std::string(#call) + ": " + (m)->rccl->GetErrorString(r_));               \
  } while (0)

// a failing call on a rank's single-GPU context: keep its message
#define DMI_M_CTX(m, r, call)                                                                                   \
  do {                                                                                                          \
    int rc_ = (call);                                                                                           \
    if (rc_ != DMI_OK)                                                                                          \
      return mfail(m, rc_, std::string(#call) + " (rank " + std::to_string((r).rank) + "): " + dmi_last_error((r).ctx)); \
  } while (0)

void shard(int64_t n, int32_t rank, int32_t world, int64_t *first, int64_t *count) {
  const int64_t base = n / world, extra = n % world;
  *first = rank * base + std::min<int64_t>(rank, extra);
  *count = base + (rank < extra ? 1 : 0);
}

size_t grid_elem(const dmi_multi_context *m) { return m->opt.grid_dtype == DMI_F64 ? 8 : 4; }

bool needs_comm(const dmi_multi_options &o) { return o.partition == DMI_PARTITION_VIEWS; }
bool needs_rccl(const dmi_multi_options &o) { return needs_comm(o) && o.exchange != DMI_EXCHANGE_PEER_COPY; }

// chunk c of the element range [e0, e1) cut into `world` pieces of q elements (q a multiple of 256, the last ones may be
// short or empty)
struct Chunk {
  int64_t first, count;
};
int64_t chunk_quantum(int64_t n, int32_t world) { return ((n + world - 1) / world + 255) / 256 * 256; }
Chunk chunk_of(int64_t e0, int64_t e1, int32_t world, int32_t c) {
  const int64_t q = chunk_quantum(e1 - e0, world);
  const int64_t lo = std::min(e1, e0 + (int64_t)c * q), hi = std::min(e1, e0 + (int64_t)(c + 1) * q);
  return Chunk{lo, hi - lo};
}

// own[i] = the sum over the ranks, IN RANK ORDER, of their values of element i: parts[j] is rank j's (the owner's entry
// points into its own grid); one rounding per addition in the grid's type, the same on every run
constexpr int kMaxPeerRanks = 16;
struct PeerParts {
  const void *p[kMaxPeerRanks];
};
template <typename T>
__global__ __launch_bounds__(256) void peer_sum_kernel(T *__restrict__ own, const PeerParts parts, int world, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  T v = static_cast<const T *>(parts.p[0])[i];
  for (int j = 1; j < world; ++j) v += static_cast<const T *>(parts.p[j])[i];
  own[i] = v;
}

int check_common(const dmi_grid_desc *grid, const dmi_ray_potential *ray, const dmi_multi_options &o, int32_t world) {
  if (!grid || !ray) return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create: null argument");
  if (world < 1 || world > 1024) return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create: world must be in [1, 1024]");
  if (o.grid_dtype != DMI_F32 && o.grid_dtype != DMI_F64)
    return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create: grid_dtype must be DMI_F32 or DMI_F64");
  if (o.partition != DMI_PARTITION_VIEWS && o.partition != DMI_PARTITION_Z_SLABS)
    return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create: unknown partition");
  if (o.exchange != DMI_EXCHANGE_ALL_REDUCE && o.exchange != DMI_EXCHANGE_REDUCE_SCATTER && o.exchange != DMI_EXCHANGE_PEER_COPY)
    return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create: unknown exchange");
  if (o.n_slabs < 0 || o.n_slabs > kMaxSlabs)
    return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create: n_slabs must be in [0, 64]");
  for (int a = 0; a < 3; ++a)
    if (grid->cell_dims[a] < 1) return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create: cell_dims must be >= 1");
  if (o.partition == DMI_PARTITION_VIEWS && o.exchange == DMI_EXCHANGE_REDUCE_SCATTER) {
    const int64_t nvox = (int64_t)grid->cell_dims[0] * grid->cell_dims[1] * grid->cell_dims[2];
    if (nvox % world != 0)
      return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT,
                   "dmi_multi_create: DMI_EXCHANGE_REDUCE_SCATTER needs a voxel count that is a multiple of the world size");
  }
  return DMI_OK;
}

// streams, events and the single-GPU context of one rank; the communicator is set up by the caller
int init_rank(dmi_multi_context *m, Rank &r) {
  DMI_M_HIP(m, hipSetDevice(r.device));
  DMI_M_HIP(m, hipStreamCreateWithFlags(&r.compute, hipStreamNonBlocking));
  DMI_M_HIP(m, hipStreamCreateWithFlags(&r.comm, hipStreamNonBlocking));
  for (int s = 0; s < kMaxSlabs; ++s) DMI_M_HIP(m, hipEventCreateWithFlags(&r.slab_done[s], hipEventDisableTiming));
  DMI_M_HIP(m, hipEventCreateWithFlags(&r.exchanged, hipEventDisableTiming));
  DMI_M_HIP(m, hipEventCreate(&r.step_start));
  DMI_M_HIP(m, hipEventCreate(&r.step_stop));
  dmi_grid_desc g = m->grid;
  dmi_options o;
  dmi_default_options(&o);
  o.device = r.device;
  o.grid_dtype = m->opt.grid_dtype;
  o.depth_storage = m->opt.depth_storage;
  o.kernel_variant = m->opt.kernel_variant;
  o.stream = r.compute;
  if (m->opt.partition == DMI_PARTITION_Z_SLABS) {
    dmi_multi_z_slab(m->grid.cell_dims[2], r.rank, m->world, &r.z_first, &r.z_count);
    if (r.z_count == 0) return DMI_OK;  // nothing to own: no context, nothing to fuse or download
    g.cell_dims[2] = r.z_count;
    o.z_first = r.z_first;
  } else {
    r.z_first = 0;
    r.z_count = m->grid.cell_dims[2];
  }
  int rc = dmi_create(&g, &m->ray, &o, &r.ctx);
  if (rc != DMI_OK) return mfail(m, rc, std::string("dmi_create (rank ") + std::to_string(r.rank) + "): " + dmi_last_error(nullptr));
  return DMI_OK;
}

// DMI_EXCHANGE_PEER_COPY: staging buffers, per-slab events, peer access between the devices (best effort: without it
// hipMemcpyPeerAsync still works, through host memory)
int init_peer_exchange(dmi_multi_context *m) {
  const int32_t world = m->world;
  const int64_t plane = (int64_t)m->grid.cell_dims[0] * m->grid.cell_dims[1];
  // every slab has its own place in the staging buffer: (world - 1) pieces of that slab's chunk size
  int64_t elems = 0;
  for (int s = 0; s < m->n_slab_ranges; ++s) elems += chunk_quantum((int64_t)m->slab_n[s] * plane, world) * (world - 1);
  for (Rank &r : m->ranks) {
    DMI_M_HIP(m, hipSetDevice(r.device));
    if (elems > 0) DMI_M_HIP(m, hipMalloc(&r.staging, (size_t)elems * grid_elem(m)));
    for (int s = 0; s < m->n_slab_ranges; ++s) {
      DMI_M_HIP(m, hipEventCreateWithFlags(&r.copied[s], hipEventDisableTiming));
      DMI_M_HIP(m, hipEventCreateWithFlags(&r.summed[s], hipEventDisableTiming));
      DMI_M_HIP(m, hipEventCreateWithFlags(&r.gathered[s], hipEventDisableTiming));
    }
    for (const Rank &o : m->ranks) {
      if (o.device == r.device) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, r.device, o.device) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess(o.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
      }
      (void)hipGetLastError();
    }
  }
  return DMI_OK;
}

dmi_multi_context *new_context(const dmi_grid_desc *grid, const dmi_ray_potential *ray, const dmi_multi_options &o,
                               int32_t world) {
  dmi_multi_context *m = new (std::nothrow) dmi_multi_context();
  if (!m) return nullptr;
  m->grid = *grid;
  m->ray = *ray;
  m->opt = o;
  if (m->opt.n_slabs == 0) m->opt.n_slabs = 4;
  m->world = world;
  m->n_voxels = (int64_t)grid->cell_dims[0] * grid->cell_dims[1] * grid->cell_dims[2];
  m->n_slab_ranges = dmi_multi_slab_ranges(grid->cell_dims[2], m->opt.n_slabs, m->slab_z, m->slab_n, kMaxSlabs);
  return m;
}

// only_local < 0: the batch is the whole fusion's, every local rank takes its part of it; otherwise the batch belongs to
// local rank `only_local` alone (the caller has partitioned)
int add_views_impl(dmi_multi_context *m, int32_t only_local, const double *depth64, const float *depth32,
                   const double *best_cost, double threshold, const double *K4, const double *RT4, int32_t n, int32_t W,
                   int32_t H) {
  if (!m) return DMI_ERR_INVALID_ARGUMENT;
  if ((!depth64 && !depth32) || !K4 || !RT4) return mfail(m, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_add_views: null pointer");
  if (n <= 0) return mfail(m, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_add_views: n must be positive");
  if (W < 1 || H < 1) return mfail(m, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_add_views: bad depth-map dimensions");
  if (only_local >= (int32_t)m->ranks.size()) return mfail(m, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_add_local_views: no such local rank");
  const size_t npix = (size_t)W * H;
  for (size_t i = 0; i < m->ranks.size(); ++i) {
    Rank &r = m->ranks[i];
    if (!r.ctx || (only_local >= 0 && (size_t)only_local != i)) continue;
    int64_t first = 0, count = n;
    if (only_local < 0 && m->opt.partition == DMI_PARTITION_VIEWS) shard(n, r.rank, m->world, &first, &count);
    if (count == 0) continue;
    if (depth32)
      DMI_M_CTX(m, r, dmi_add_views_f32(r.ctx, depth32 + (size_t)first * npix, K4 + 16 * first, RT4 + 16 * first,
                                        (int32_t)count, W, H));
    else
      DMI_M_CTX(m, r, dmi_add_views(r.ctx, depth64 + (size_t)first * npix, best_cost ? best_cost + (size_t)first * npix : nullptr,
                                    threshold, K4 + 16 * first, RT4 + 16 * first, (int32_t)count, W, H));
    r.n_views += count;
  }
  m->n_views_total += n;
  return DMI_OK;
}

// step_start .. step_stop of local rank 0 into the timings (waits for the step)
int drain_step(dmi_multi_context *m) {
  if (!m->step_pending) return DMI_OK;
  Rank &r0 = m->ranks[0];
  DMI_M_HIP(m, hipSetDevice(r0.device));
  DMI_M_HIP(m, hipEventSynchronize(r0.step_stop));
  float ms = 0.f;
  DMI_M_HIP(m, hipEventElapsedTime(&ms, r0.step_start, r0.step_stop));
  m->timings.last_step_ms = ms;
  m->timings.total_step_ms += ms;
  m->timings.steps += 1;
  m->step_pending = false;
  if (r0.ctx) {  // the step has finished: reading the rank's own timings waits for nothing
    dmi_timings t;
    if (dmi_get_timings(r0.ctx, &t) == DMI_OK) {
      m->timings.last_fuse_kernel_ms = t.total_fuse_kernel_ms - m->fuse_ms_seen;
      m->fuse_ms_seen = t.total_fuse_kernel_ms;
    }
  }
  return DMI_OK;
}

// DMI_EXCHANGE_PEER_COPY, one step: fuse slab by slab; behind slab s every rank sends the other ranks their chunk of it
// (peer copies on `comm`), and one slab LATER -- so that no compute stream ever waits for a copy that could still be in
// flight -- every rank adds what it has received to its own chunk (a small kernel on `compute`, rank order) and sends
// the sum back into everybody's grid.  grids[i]: rank i's grid pointer where the caller already asked for it.
int peer_exchange(dmi_multi_context *m, bool by_slabs, std::vector<void *> &grids) {
  const int32_t world = m->world;
  const size_t esz = grid_elem(m);
  const int64_t plane = (int64_t)m->grid.cell_dims[0] * m->grid.cell_dims[1];
  const int n_rounds = m->n_slab_ranges;
  std::vector<int64_t> stage_off((size_t)n_rounds + 1, 0);
  for (int s = 0; s < n_rounds; ++s)
    stage_off[(size_t)s + 1] = stage_off[(size_t)s] + chunk_quantum((int64_t)m->slab_n[s] * plane, world) * (world - 1);
  auto slot_of = [](int32_t sender, int32_t owner) { return sender < owner ? sender : sender - 1; };

  auto sum_and_gather = [&](int s) -> int {
    const int64_t e0 = (int64_t)m->slab_z[s] * plane, e1 = e0 + (int64_t)m->slab_n[s] * plane;
    const int64_t q = chunk_quantum(e1 - e0, world);
    for (Rank &r : m->ranks) {
      const Chunk c = chunk_of(e0, e1, world, r.rank);
      DMI_M_HIP(m, hipSetDevice(r.device));
      for (Rank &j : m->ranks)
        if (j.rank != r.rank) DMI_M_HIP(m, hipStreamWaitEvent(r.compute, j.copied[s], 0));
      if (c.count > 0) {
        char *own = static_cast<char *>(grids[(size_t)r.rank]) + c.first * (int64_t)esz;
        PeerParts parts{};
        for (int32_t j = 0; j < world; ++j)
          parts.p[j] = j == r.rank ? own : static_cast<char *>(r.staging) + (stage_off[(size_t)s] + (int64_t)slot_of(j, r.rank) * q) * (int64_t)esz;
        const dim3 blocks((unsigned)((c.count + 255) / 256));
        if (esz == 8)
          hipLaunchKernelGGL(peer_sum_kernel<double>, blocks, dim3(256), 0, r.compute, reinterpret_cast<double *>(own), parts, world, c.count);
        else
          hipLaunchKernelGGL(peer_sum_kernel<float>, blocks, dim3(256), 0, r.compute, reinterpret_cast<float *>(own), parts, world, c.count);
        DMI_M_HIP(m, hipGetLastError());
      }
      DMI_M_HIP(m, hipEventRecord(r.summed[s], r.compute));
      DMI_M_HIP(m, hipStreamWaitEvent(r.comm, r.summed[s], 0));
      if (c.count > 0) {
        for (Rank &j : m->ranks) {
          if (j.rank == r.rank) continue;
          DMI_M_HIP(m, hipMemcpyPeerAsync(static_cast<char *>(grids[(size_t)j.rank]) + c.first * (int64_t)esz, j.device,
                                          static_cast<char *>(grids[(size_t)r.rank]) + c.first * (int64_t)esz, r.device,
                                          (size_t)c.count * esz, r.comm));
        }
      }
      DMI_M_HIP(m, hipEventRecord(r.gathered[s], r.comm));
    }
    return DMI_OK;
  };

  for (int s = 0; s < n_rounds; ++s) {
    const int64_t e0 = (int64_t)m->slab_z[s] * plane, e1 = e0 + (int64_t)m->slab_n[s] * plane;
    const int64_t q = chunk_quantum(e1 - e0, world);
    for (Rank &r : m->ranks) {
      DMI_M_HIP(m, hipSetDevice(r.device));
      if (by_slabs && r.n_views > 0) DMI_M_CTX(m, r, dmi_fuse_slab(r.ctx, m->slab_z[s], m->slab_n[s]));
      if (!grids[(size_t)r.rank]) DMI_M_CTX(m, r, dmi::grid_pointer_for_sums(r.ctx, &grids[(size_t)r.rank]));
      DMI_M_HIP(m, hipEventRecord(r.slab_done[s], r.compute));
      DMI_M_HIP(m, hipStreamWaitEvent(r.comm, r.slab_done[s], 0));
    }
    for (Rank &j : m->ranks) {  // rank j's chunk r of the slab into rank r's staging
      DMI_M_HIP(m, hipSetDevice(j.device));
      for (Rank &r : m->ranks) {
        if (r.rank == j.rank) continue;
        const Chunk c = chunk_of(e0, e1, world, r.rank);
        if (c.count == 0) continue;
        char *dst = static_cast<char *>(r.staging) + (stage_off[(size_t)s] + (int64_t)slot_of(j.rank, r.rank) * q) * (int64_t)esz;
        DMI_M_HIP(m, hipMemcpyPeerAsync(dst, r.device, static_cast<char *>(grids[(size_t)j.rank]) + c.first * (int64_t)esz, j.device,
                                        (size_t)c.count * esz, j.comm));
      }
      DMI_M_HIP(m, hipEventRecord(j.copied[s], j.comm));
    }
    if (s > 0) {
      int rc = sum_and_gather(s - 1);
      if (rc != DMI_OK) return rc;
    }
  }
  int rc = sum_and_gather(n_rounds - 1);
  if (rc != DMI_OK) return rc;
  // Whoever touches a grid next on its compute stream sees every chunk's sum -- and finds the rank's OWN outgoing copies done:
  // rank r's comm stream is still reading r's summed chunk out of r's grid while it gathers it to the others, so r.compute waits
  // on r.gathered[s] too.  (Without that, the step's stop event preceded rank 0's outgoing copies -- the exposed exchange time
  // read short -- and the next writer of the grid on r.compute, a reset or an upload, could overwrite a chunk in flight.)
  for (Rank &r : m->ranks) {
    DMI_M_HIP(m, hipSetDevice(r.device));
    for (Rank &j : m->ranks)
      for (int s = 0; s < n_rounds; ++s) DMI_M_HIP(m, hipStreamWaitEvent(r.compute, j.gathered[s], 0));
  }
  return DMI_OK;
}

template <typename T>
int download_impl(dmi_multi_context *m, T *out, int64_t *owned_first, int64_t *owned_count) {
  if (!m || !out) return mfail(m, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_download_grid: null argument");
  constexpr bool want_f64 = sizeof(T) == 8;
  auto fetch = [&](Rank &r, T *dst) -> int {  // the whole grid of the rank's context
    if (want_f64) return dmi_download_grid_f64(r.ctx, reinterpret_cast<double *>(dst));
    return dmi_download_grid_f32(r.ctx, reinterpret_cast<float *>(dst));
  };
  int64_t lo = 0, n = 0;
  const int64_t plane = (int64_t)m->grid.cell_dims[0] * m->grid.cell_dims[1];
  if (m->opt.partition == DMI_PARTITION_Z_SLABS) {
    // every rank's context IS its slab: straight into its place
    lo = -1;
    for (Rank &r : m->ranks) {
      if (!r.ctx) continue;
      DMI_M_CTX(m, r, fetch(r, out + (int64_t)r.z_first * plane));
      if (lo < 0) lo = (int64_t)r.z_first * plane;
      n = ((int64_t)r.z_first + r.z_count) * plane - lo;
    }
    if (lo < 0) lo = 0;
  } else if (m->opt.exchange != DMI_EXCHANGE_REDUCE_SCATTER || m->world == 1) {  // all-reduce, by RCCL or by peer copies
    DMI_M_CTX(m, m->ranks[0], fetch(m->ranks[0], out));
    n = m->n_voxels;
  } else {
    // reduce-scatter: rank r holds the sum of elements [r * per, (r + 1) * per) inside its own full-size grid
    const int64_t per = m->n_voxels / m->world;
    const bool grid_f64 = m->opt.grid_dtype == DMI_F64;
    std::vector<unsigned char> staging;
    for (Rank &r : m->ranks) {
      void *dptr = nullptr;
      DMI_M_CTX(m, r, dmi_synchronize(r.ctx));
      DMI_M_CTX(m, r, dmi::grid_pointer_for_sums(r.ctx, &dptr));
      DMI_M_HIP(m, hipSetDevice(r.device));
      const int64_t first = (int64_t)r.rank * per;
      if (grid_f64 == want_f64) {
        DMI_M_HIP(m, hipMemcpy(out + first, static_cast<const char *>(dptr) + first * sizeof(T), (size_t)per * sizeof(T),
                               hipMemcpyDeviceToHost));
      } else {
        const size_t esz = grid_f64 ? 8 : 4;
        staging.resize((size_t)per * esz);
        DMI_M_HIP(m, hipMemcpy(staging.data(), static_cast<const char *>(dptr) + first * esz, (size_t)per * esz,
                               hipMemcpyDeviceToHost));
        if (grid_f64) {
          const double *s = reinterpret_cast<const double *>(staging.data());
          for (int64_t i = 0; i < per; ++i) out[first + i] = (T)s[i];
        } else {
          const float *s = reinterpret_cast<const float *>(staging.data());
          for (int64_t i = 0; i < per; ++i) out[first + i] = (T)s[i];
        }
      }
    }
    lo = (int64_t)m->ranks.front().rank * per;
    n = (int64_t)m->ranks.size() * per;
  }
  if (owned_first) *owned_first = lo;
  if (owned_count) *owned_count = n;
  return drain_step(m);
}

}  // namespace

extern "C" {

void dmi_multi_default_options(dmi_multi_options *opt) {
  if (!opt) return;
  std::memset(opt, 0, sizeof(*opt));
  opt->grid_dtype = DMI_F32;
  opt->depth_storage = DMI_DEPTH_AUTO;
  opt->partition = DMI_PARTITION_VIEWS;
  opt->exchange = DMI_EXCHANGE_ALL_REDUCE;
  opt->n_slabs = 4;
}

int dmi_multi_view_shard(int64_t n, int32_t rank, int32_t world, int64_t *first, int64_t *count) {
  if (!first || !count || n < 0 || world < 1 || rank < 0 || rank >= world) return DMI_ERR_INVALID_ARGUMENT;
  shard(n, rank, world, first, count);
  return DMI_OK;
}

int dmi_multi_z_slab(int32_t nz, int32_t rank, int32_t world, int32_t *z_first, int32_t *z_count) {
  if (!z_first || !z_count || nz < 1 || world < 1 || rank < 0 || rank >= world) return DMI_ERR_INVALID_ARGUMENT;
  const int64_t units = ((int64_t)nz + kZSlabAlignment - 1) / kZSlabAlignment;
  int64_t first = 0, count = 0;
  shard(units, rank, world, &first, &count);
  const int64_t z0 = std::min<int64_t>(first * kZSlabAlignment, nz), z1 = std::min<int64_t>((first + count) * kZSlabAlignment, nz);
  *z_first = (int32_t)z0;
  *z_count = (int32_t)(z1 - z0);
  return DMI_OK;
}

int dmi_multi_peer_chunk(int64_t n, int32_t world, int32_t c, int64_t *first, int64_t *count) {
  if (!first || !count || n < 0 || world < 1 || c < 0 || c >= world) return DMI_ERR_INVALID_ARGUMENT;
  const Chunk k = chunk_of(0, n, world, c);
  *first = k.first;
  *count = k.count;
  return DMI_OK;
}

int dmi_multi_slab_ranges(int32_t nz, int32_t n_slabs, int32_t *z_first, int32_t *z_count, int32_t max_slabs) {
  if (!z_first || !z_count || nz < 1 || max_slabs < 1) return 0;
  const int64_t align = DMI_SLAB_ALIGNMENT;
  const int64_t units = (nz + align - 1) / align;
  const int64_t n = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n_slabs, max_slabs), units));
  // The last slab's exchange is the one piece nothing hides: make it about half as thick as the others, which share
  // the difference (possible once every slab can have at least two units).
  int64_t sizes[kMaxSlabs];
  if (n > kMaxSlabs) return 0;
  if (n >= 2 && units >= 2 * n) {
    const int64_t last = std::max<int64_t>(1, units / (2 * n));
    const int64_t rest = (units - last) / (n - 1), extra = (units - last) % (n - 1);
    for (int64_t s = 0; s < n - 1; ++s) sizes[s] = rest + (s < extra ? 1 : 0);
    sizes[n - 1] = last;
  } else {
    for (int64_t s = 0; s < n; ++s) {
      int64_t first = 0, count = 0;
      shard(units, (int32_t)s, (int32_t)n, &first, &count);
      sizes[s] = count;
    }
  }
  int written = 0;
  int64_t lo = 0;
  for (int64_t s = 0; s < n; ++s) {
    const int64_t z0 = std::min<int64_t>(lo * align, nz), z1 = std::min<int64_t>((lo + sizes[s]) * align, nz);
    if (z1 > z0) {
      z_first[written] = (int32_t)z0;
      z_count[written] = (int32_t)(z1 - z0);
      ++written;
    }
    lo += sizes[s];
  }
  return written;
}

const char *dmi_multi_last_error(const dmi_multi_context *m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

int dmi_multi_get_unique_id(uint8_t id[DMI_UNIQUE_ID_BYTES]) {
  return guarded(nullptr, "dmi_multi_get_unique_id", [&]() -> int {
    static_assert(sizeof(ncclUniqueId) == DMI_UNIQUE_ID_BYTES, "unique id size");
    if (!id) return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_get_unique_id: null argument");
    Rccl *api = load_rccl();
    if (!api->handle) return mfail(nullptr, DMI_ERR_DEVICE, "dmi_multi_get_unique_id: " + api->error);
    ncclUniqueId u;
    const ncclResult_t r = api->GetUniqueId(&u);
    if (r != ncclSuccess) return mfail(nullptr, DMI_ERR_DEVICE, std::string("ncclGetUniqueId: ") + api->GetErrorString(r));
    std::memcpy(id, u.internal, DMI_UNIQUE_ID_BYTES);
    return DMI_OK;
  });
}

int dmi_multi_create(const dmi_grid_desc *grid, const dmi_ray_potential *ray, const dmi_multi_options *opt,
                     const int32_t *devices, int32_t n, dmi_multi_context **out) {
  return guarded(nullptr, "dmi_multi_create", [&]() -> int {
    if (!out || !devices) return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create: null argument");
    *out = nullptr;
    dmi_multi_options o;
    dmi_multi_default_options(&o);
    if (opt) o = *opt;
    int rc = check_common(grid, ray, o, n);
    if (rc != DMI_OK) return rc;
    if (o.partition == DMI_PARTITION_VIEWS && o.exchange == DMI_EXCHANGE_PEER_COPY && n > kMaxPeerRanks)
      return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create: DMI_EXCHANGE_PEER_COPY serves at most 16 ranks");
    const int ndev = dmi_device_count();
    if (ndev <= 0) return mfail(nullptr, DMI_ERR_DEVICE, "dmi_multi_create: no HIP device available");
    for (int32_t i = 0; i < n; ++i) {
      if (devices[i] < 0 || devices[i] >= ndev) return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create: device ordinal out of range");
      // (the peer-copy exchange has no communicator that would object: several of its ranks may share a device, which is
      // how a one-GPU box rehearses every step of it)
      for (int32_t j = 0; j < i; ++j)
        if (devices[j] == devices[i] && !(o.partition == DMI_PARTITION_VIEWS && o.exchange == DMI_EXCHANGE_PEER_COPY))
          return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create: a device is listed twice");
    }
    dmi_multi_context *m = new_context(grid, ray, o, n);
    if (!m) return mfail(nullptr, DMI_ERR_OUT_OF_MEMORY, "dmi_multi_create: host allocation failed");
    m->one_process = true;
    m->ranks.resize((size_t)n);
    auto give_up = [&](int code) {
      g_multi_create_error = m->err;
      dmi_multi_destroy(m);
      return code;
    };
    for (int32_t i = 0; i < n; ++i) {
      m->ranks[(size_t)i].device = devices[i];
      m->ranks[(size_t)i].rank = i;
      rc = init_rank(m, m->ranks[(size_t)i]);
      if (rc != DMI_OK) return give_up(rc);
    }
    if (needs_comm(m->opt) && !needs_rccl(m->opt)) {
      rc = init_peer_exchange(m);
      if (rc != DMI_OK) return give_up(rc);
    }
    if (needs_rccl(m->opt)) {
      m->rccl = load_rccl();
      if (!m->rccl->handle) {
        m->err = "dmi_multi_create: " + m->rccl->error;
        return give_up(DMI_ERR_DEVICE);
      }
      std::vector<ncclComm_t> comms((size_t)n);
      const ncclResult_t r = m->rccl->CommInitAll(comms.data(), n, devices);
      if (r != ncclSuccess) {
        m->err = std::string("ncclCommInitAll: ") + m->rccl->GetErrorString(r);
        return give_up(DMI_ERR_DEVICE);
      }
      for (int32_t i = 0; i < n; ++i) m->ranks[(size_t)i].nccl = comms[(size_t)i];
    }
    *out = m;
    return DMI_OK;
  });
}

int dmi_multi_create_rank(const dmi_grid_desc *grid, const dmi_ray_potential *ray, const dmi_multi_options *opt,
                          int32_t device, int32_t rank, int32_t world, const uint8_t id[DMI_UNIQUE_ID_BYTES],
                          dmi_multi_context **out) {
  return guarded(nullptr, "dmi_multi_create_rank", [&]() -> int {
    if (!out) return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create_rank: null argument");
    *out = nullptr;
    dmi_multi_options o;
    dmi_multi_default_options(&o);
    if (opt) o = *opt;
    int rc = check_common(grid, ray, o, world);
    if (rc != DMI_OK) return rc;
    if (rank < 0 || rank >= world) return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create_rank: rank out of range");
    if (o.partition == DMI_PARTITION_VIEWS && o.exchange == DMI_EXCHANGE_PEER_COPY)
      return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT,
                   "dmi_multi_create_rank: DMI_EXCHANGE_PEER_COPY needs every rank in one process (dmi_multi_create); ranks in "
                   "processes of their own would have to exchange IPC memory handles, which this library does not do");
    if (needs_comm(o) && !id) return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create_rank: the VIEWS partition needs a unique id");
    const int ndev = dmi_device_count();
    if (ndev <= 0) return mfail(nullptr, DMI_ERR_DEVICE, "dmi_multi_create_rank: no HIP device available");
    if (device < 0 || device >= ndev) return mfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_create_rank: device ordinal out of range");
    dmi_multi_context *m = new_context(grid, ray, o, world);
    if (!m) return mfail(nullptr, DMI_ERR_OUT_OF_MEMORY, "dmi_multi_create_rank: host allocation failed");
    m->one_process = false;
    m->ranks.resize(1);
    m->ranks[0].device = device;
    m->ranks[0].rank = rank;
    auto give_up = [&](int code) {
      g_multi_create_error = m->err;
      dmi_multi_destroy(m);
      return code;
    };
    rc = init_rank(m, m->ranks[0]);
    if (rc != DMI_OK) return give_up(rc);
    if (needs_comm(m->opt)) {
      m->rccl = load_rccl();
      if (!m->rccl->handle) {
        m->err = "dmi_multi_create_rank: " + m->rccl->error;
        return give_up(DMI_ERR_DEVICE);
      }
      ncclUniqueId u;
      std::memcpy(u.internal, id, DMI_UNIQUE_ID_BYTES);
      if (hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();
        m->err = "dmi_multi_create_rank: hipSetDevice failed";
        return give_up(DMI_ERR_DEVICE);
      }
      const ncclResult_t r = m->rccl->CommInitRank(&m->ranks[0].nccl, world, u, rank);
      if (r != ncclSuccess) {
        m->err = std::string("ncclCommInitRank: ") + m->rccl->GetErrorString(r);
        return give_up(DMI_ERR_DEVICE);
      }
    }
    *out = m;
    return DMI_OK;
  });
}

void dmi_multi_destroy(dmi_multi_context *m) {
  if (!m) return;
  for (Rank &r : m->ranks) {
    (void)hipSetDevice(r.device);
    if (r.compute) (void)hipStreamSynchronize(r.compute);
    if (r.comm) (void)hipStreamSynchronize(r.comm);
  }
  for (Rank &r : m->ranks) {
    (void)hipSetDevice(r.device);
    if (r.nccl && m->rccl && m->rccl->CommDestroy) (void)m->rccl->CommDestroy(r.nccl);
    if (r.ctx) dmi_destroy(r.ctx);  // before its stream
    for (int s = 0; s < kMaxSlabs; ++s)
      if (r.slab_done[s]) (void)hipEventDestroy(r.slab_done[s]);
    for (int s = 0; s < kMaxSlabs; ++s) {
      if (r.copied[s]) (void)hipEventDestroy(r.copied[s]);
      if (r.summed[s]) (void)hipEventDestroy(r.summed[s]);
      if (r.gathered[s]) (void)hipEventDestroy(r.gathered[s]);
    }
    if (r.staging) (void)hipFree(r.staging);
    if (r.exchanged) (void)hipEventDestroy(r.exchanged);
    if (r.step_start) (void)hipEventDestroy(r.step_start);
    if (r.step_stop) (void)hipEventDestroy(r.step_stop);
    if (r.comm) (void)hipStreamDestroy(r.comm);
    if (r.compute) (void)hipStreamDestroy(r.compute);
  }
  delete m;
}

int dmi_multi_add_views(dmi_multi_context *m, const double *depth, const double *best_cost, double threshold,
                        const double *K4, const double *RT4, int32_t n, int32_t width, int32_t height) {
  return guarded(m, "dmi_multi_add_views", [&]() -> int {
    return add_views_impl(m, -1, depth, nullptr, best_cost, threshold, K4, RT4, n, width, height);
  });
}

int dmi_multi_add_local_views(dmi_multi_context *m, int32_t local_index, const double *depth, const double *best_cost,
                              double threshold, const double *K4, const double *RT4, int32_t n, int32_t width, int32_t height) {
  return guarded(m, "dmi_multi_add_local_views", [&]() -> int {
    if (local_index < 0) return mfail(m, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_add_local_views: no such local rank");
    return add_views_impl(m, local_index, depth, nullptr, best_cost, threshold, K4, RT4, n, width, height);
  });
}

int dmi_multi_add_local_views_f32(dmi_multi_context *m, int32_t local_index, const float *depth, const double *K4,
                                  const double *RT4, int32_t n, int32_t width, int32_t height) {
  return guarded(m, "dmi_multi_add_local_views_f32", [&]() -> int {
    if (local_index < 0) return mfail(m, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_add_local_views_f32: no such local rank");
    return add_views_impl(m, local_index, nullptr, depth, nullptr, 0.0, K4, RT4, n, width, height);
  });
}

int dmi_multi_add_views_f32(dmi_multi_context *m, const float *depth, const double *K4, const double *RT4, int32_t n,
                            int32_t width, int32_t height) {
  return guarded(m, "dmi_multi_add_views_f32", [&]() -> int {
    return add_views_impl(m, -1, nullptr, depth, nullptr, 0.0, K4, RT4, n, width, height);
  });
}

int dmi_multi_clear_views(dmi_multi_context *m) {
  return guarded(m, "dmi_multi_clear_views", [&]() -> int {
    if (!m) return DMI_ERR_INVALID_ARGUMENT;
    for (Rank &r : m->ranks) {
      if (!r.ctx) continue;
      DMI_M_CTX(m, r, dmi_clear_views(r.ctx));
      r.n_views = 0;
    }
    m->n_views_total = 0;
    return DMI_OK;
  });
}

int dmi_multi_fuse(dmi_multi_context *m) {
  return guarded(m, "dmi_multi_fuse", [&]() -> int {
    if (!m) return DMI_ERR_INVALID_ARGUMENT;
    // (a rank of a multi-process fusion may legitimately hold no view: it contributes zeros to the exchange)
    if (m->one_process && m->n_views_total == 0)
      return mfail(m, DMI_ERR_STATE, "dmi_multi_fuse: no views (call dmi_multi_add_views first)");
    int rc = drain_step(m);
    if (rc != DMI_OK) return rc;
    const bool exchange = needs_comm(m->opt);
    const bool peer = exchange && m->opt.exchange == DMI_EXCHANGE_PEER_COPY;
    const bool by_slabs = exchange && (peer || m->opt.exchange == DMI_EXCHANGE_ALL_REDUCE) && m->n_slab_ranges > 1;
    const ncclDataType_t dtype = m->opt.grid_dtype == DMI_F64 ? ncclDouble : ncclFloat;
    const size_t esz = grid_elem(m);
    const int64_t plane = (int64_t)m->grid.cell_dims[0] * m->grid.cell_dims[1];
    std::vector<void *> grids(m->ranks.size(), nullptr);

    {  // the step's clock starts on local rank 0's compute stream whether or not that rank owns any cell layer (a z-slab
       // rank of a short grid may own none and has no context: its step is empty, its events must still pair up)
      Rank &r0 = m->ranks[0];
      DMI_M_HIP(m, hipSetDevice(r0.device));
      DMI_M_HIP(m, hipEventRecord(r0.step_start, r0.compute));
    }
    for (size_t i = 0; i < m->ranks.size(); ++i) {
      Rank &r = m->ranks[i];
      if (!r.ctx) continue;
      DMI_M_HIP(m, hipSetDevice(r.device));
      DMI_M_CTX(m, r, dmi_reset_grid(r.ctx));  // filt.cxx:133: every fusion starts from zeros
      if (exchange) {
        // the collective reads the grid whether or not this rank fused anything: the pointer call also settles a
        // deferred zero fill (a rank with no views contributes zeros)
        if (r.n_views == 0 || !by_slabs) DMI_M_CTX(m, r, dmi::grid_pointer_for_sums(r.ctx, &grids[i]));
      }
    }

    if (!by_slabs) {
      for (Rank &r : m->ranks)
        if (r.ctx && r.n_views > 0) DMI_M_CTX(m, r, dmi_fuse(r.ctx));
    }
    if (peer) {
      int rc_peer = peer_exchange(m, by_slabs, grids);
      if (rc_peer != DMI_OK) return rc_peer;
    } else if (exchange) {
      const int n_rounds = by_slabs ? m->n_slab_ranges : 1;
      for (int s = 0; s < n_rounds; ++s) {
        const int32_t z0 = by_slabs ? m->slab_z[s] : 0, zc = by_slabs ? m->slab_n[s] : m->grid.cell_dims[2];
        for (size_t i = 0; i < m->ranks.size(); ++i) {
          Rank &r = m->ranks[i];
          DMI_M_HIP(m, hipSetDevice(r.device));
          if (by_slabs && r.n_views > 0) DMI_M_CTX(m, r, dmi_fuse_slab(r.ctx, z0, zc));
          if (!grids[i]) DMI_M_CTX(m, r, dmi::grid_pointer_for_sums(r.ctx, &grids[i]));
          DMI_M_HIP(m, hipEventRecord(r.slab_done[s], r.compute));
          DMI_M_HIP(m, hipStreamWaitEvent(r.comm, r.slab_done[s], 0));
        }
        // one collective per round; the ranks of this process are issued as one group
        DMI_M_NCCL(m, m->rccl->GroupStart());
        ncclResult_t issued = ncclSuccess;
        for (size_t i = 0; i < m->ranks.size() && issued == ncclSuccess; ++i) {
          Rank &r = m->ranks[i];
          char *base = static_cast<char *>(grids[i]);
          if (m->opt.exchange == DMI_EXCHANGE_ALL_REDUCE) {
            char *p = base + (int64_t)z0 * plane * (int64_t)esz;
            issued = m->rccl->AllReduce(p, p, (size_t)((int64_t)zc * plane), dtype, ncclSum, r.nccl, r.comm);
          } else {
            const int64_t per = m->n_voxels / m->world;  // in place: the result lands in this rank's own part of its grid
            issued = m->rccl->ReduceScatter(base, base + (int64_t)r.rank * per * (int64_t)esz, (size_t)per, dtype, ncclSum, r.nccl,
                                            r.comm);
          }
        }
        const ncclResult_t closed = m->rccl->GroupEnd();
        if (issued != ncclSuccess) return mfail(m, DMI_ERR_DEVICE, std::string("RCCL collective: ") + m->rccl->GetErrorString(issued));
        if (closed != ncclSuccess) return mfail(m, DMI_ERR_DEVICE, std::string("ncclGroupEnd: ") + m->rccl->GetErrorString(closed));
      }
      // whoever touches the grid next on the compute stream (download, cell -> point, the next reset) sees the sums
      for (Rank &r : m->ranks) {
        DMI_M_HIP(m, hipSetDevice(r.device));
        DMI_M_HIP(m, hipEventRecord(r.exchanged, r.comm));
        DMI_M_HIP(m, hipStreamWaitEvent(r.compute, r.exchanged, 0));
      }
    }
    Rank &r0 = m->ranks[0];
    DMI_M_HIP(m, hipSetDevice(r0.device));
    DMI_M_HIP(m, hipEventRecord(r0.step_stop, r0.compute));
    m->step_pending = true;
    return DMI_OK;
  });
}

int dmi_multi_synchronize(dmi_multi_context *m) {
  return guarded(m, "dmi_multi_synchronize", [&]() -> int {
    if (!m) return DMI_ERR_INVALID_ARGUMENT;
    for (Rank &r : m->ranks) {
      DMI_M_HIP(m, hipSetDevice(r.device));
      DMI_M_HIP(m, hipStreamSynchronize(r.comm));
      if (r.ctx)
        DMI_M_CTX(m, r, dmi_synchronize(r.ctx));
      else
        DMI_M_HIP(m, hipStreamSynchronize(r.compute));
    }
    return drain_step(m);
  });
}

int dmi_multi_download_grid_f32(dmi_multi_context *m, float *out, int64_t *owned_first, int64_t *owned_count) {
  return guarded(m, "dmi_multi_download_grid_f32", [&]() -> int { return download_impl<float>(m, out, owned_first, owned_count); });
}

int dmi_multi_download_grid_f64(dmi_multi_context *m, double *out, int64_t *owned_first, int64_t *owned_count) {
  return guarded(m, "dmi_multi_download_grid_f64", [&]() -> int { return download_impl<double>(m, out, owned_first, owned_count); });
}

int dmi_multi_get_info(dmi_multi_context *m, dmi_multi_info *out) {
  return guarded(m, "dmi_multi_get_info", [&]() -> int {
    if (!m || !out) return mfail(m, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_get_info: null argument");
    std::memset(out, 0, sizeof(*out));
    out->world = m->world;
    out->n_local = (int32_t)m->ranks.size();
    out->first_rank = m->ranks.empty() ? 0 : m->ranks.front().rank;
    if (m->rccl && m->rccl->handle) {
      int v = 0;
      if (m->rccl->GetVersion(&v) == ncclSuccess) out->rccl_version = v;
      int cnt = 0;
      if (!m->ranks.empty() && m->ranks[0].nccl && m->rccl->CommCount(m->ranks[0].nccl, &cnt) == ncclSuccess) out->rccl_ranks = cnt;
    }
    out->partition = m->opt.partition;
    out->exchange = m->opt.exchange;
    out->n_slabs = m->n_slab_ranges;
    out->n_voxels = m->n_voxels;
    out->n_views_total = m->n_views_total;
    for (const Rank &r : m->ranks) out->n_views_local += r.n_views;
    return DMI_OK;
  });
}

int dmi_multi_get_timings(dmi_multi_context *m, dmi_multi_timings *out) {
  return guarded(m, "dmi_multi_get_timings", [&]() -> int {
    if (!m || !out) return mfail(m, DMI_ERR_INVALID_ARGUMENT, "dmi_multi_get_timings: null argument");
    int rc = drain_step(m);
    if (rc != DMI_OK) return rc;
    *out = m->timings;
    return DMI_OK;
  });
}

int dmi_multi_local_context(dmi_multi_context *m, int32_t local_index, dmi_context **out) {
  if (!m || !out || local_index < 0 || (size_t)local_index >= m->ranks.size()) return DMI_ERR_INVALID_ARGUMENT;
  *out = m->ranks[(size_t)local_index].ctx;
  return DMI_OK;
}

}  // extern "C"
