// Multi-GPU REG inside ONE process: voxgraph is a single process
// (voxgraph/src/voxgraph_mapping_node.cpp:6-26) whose Ceres solve evaluates every registration
// residual block per iteration (pose_graph.cpp:85-106).  Given the pose vector the constraints are
// independent, so the constraint list is pair-sharded over N contexts (one per GPU, submaps
// replicated); each context runs the fused pass on its share from its own host thread and its
// own stream, and the shares meet in ONE reduction per solver evaluation:
//
//     worker k : vgx_reg_batch_evaluate_normal -> vgx_reg_batch_assemble -> event
//     context 0: waits for the K events, sums the K fused buffers IN CONTEXT ORDER (peer-mapped
//                reads over xGMI, ~180 KB each for 200 submaps / 1176 constraints) -> host
//
// The sum is a fixed-order sum of deterministic partial buffers: bitwise reproducible, unlike an
// all-reduce whose ring order depends on the communicator.  The same sharding with one process
// per GPU and an RCCL all-reduce (what bench.py's torchrun launch does) uses the same entry
// points underneath (vgx_lpt_shards, vgx_reg_batch_*).
#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <numeric>
#include <thread>

// RCCL: types and prototypes only -- the library is opened at run time (RcclApi), so the build must not
// depend on its header being installed either
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclInt64 = 4, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
const char* ncclGetErrorString(ncclResult_t result);
}
#endif

#include "vgx_internal.h"

namespace vgx {

constexpr int kMaxShards = 16;

// all[g][.] = the block of global constraint g, read where its shard left it (src[g]: a pointer into that
// shard's [n_local][45] array -- on another GPU it is reached through the xGMI peer mapping).  A copy, not a
// sum: the complete array is the same whatever the placement was.
__global__ void multi_gather_kernel(const double* const* __restrict__ src, long long n, double* __restrict__ all) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * kNormalSize) return;
  // (pointers out of a table are generic to the compiler: say global, or the load goes out as a FLAT one)
  all[i] = ((const __attribute__((address_space(1))) double*)src[i / kNormalSize])[i % kNormalSize];
}

// RCCL, opened on demand (VGX_REDUCE_RCCL): the alternative reduction SURVEY.md 8(e) asks to compare --
// ONE ncclAllReduce(sum, f64) per solver evaluation over xGMI of the [n][45] array of per-constraint blocks,
// every context contributing its own rows in place (all other rows zero: the sum is exact in any order).  No link-time dependency: librccl.so.1 is whatever
// copy the process already has (PyTorch's, when the bench drives this) or the ROCm one.
struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok() const { return handle && CommInitAll && CommDestroy && AllReduce && GroupStart && GroupEnd && GetErrorString; }
};

static RcclApi& rccl_api() {
  static RcclApi api = [] {
    RcclApi a;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.handle) break;
    }
    if (a.handle) {
      a.CommInitAll = (decltype(a.CommInitAll))dlsym(a.handle, "ncclCommInitAll");
      a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
      a.AllReduce = (decltype(a.AllReduce))dlsym(a.handle, "ncclAllReduce");
      a.GroupStart = (decltype(a.GroupStart))dlsym(a.handle, "ncclGroupStart");
      a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.handle, "ncclGroupEnd");
      a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
    }
    return a;
  }();
  return api;
}

// The multi-context entry points select one device after another; the caller (voxgraph's own thread, or
// PyTorch in the bench) finds its current device unchanged afterwards.
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() {
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  }
  ~DeviceGuard() {
    if (dev >= 0) (void)hipSetDevice(dev);
  }
};

}  // namespace vgx

using namespace vgx;

struct vgx_reg_multi_s {
  struct Shard {
    vgx_ctx ctx = nullptr;
    vgx_reg_batch batch = nullptr;
    std::vector<int32_t> global;   // global constraint index of local constraint c
    std::vector<int32_t> status;   // per local constraint
    double* d_all = nullptr;       // VGX_REDUCE_RCCL: [n][45], this shard's rows filled, the rest zero
    double* h_normal = nullptr;    // pinned [n_local][45]
    hipEvent_t done = nullptr;
    int rc = VGX_OK;
    std::thread th;
  };
  std::vector<std::unique_ptr<Shard>> shards;
  int32_t n = 0;                   // global constraint count
  std::vector<int32_t> shard_of;   // [n]
  // job hand-off to the worker threads
  std::mutex mu;
  std::condition_variable go, finished;
  uint64_t generation = 0;
  int pending = 0;
  bool quit = false;
  int mode = 0;                    // 0 fused, 1 per-constraint normal blocks to the host, 2 per-constraint costs to the host
  const double* poses = nullptr;
  int32_t n_nodes = 0;
  // context 0: the complete [n][45] array (gathered from the shards), the list's node structure, the fused
  // buffer it assembles and its pinned mirror
  double* d_normal_all = nullptr;
  const double** d_src = nullptr;  // [n] where each constraint's block lies (peer pointers)
  vgx_reg_assembler assembler = nullptr;
  double* d_sum = nullptr;
  double* h_sum = nullptr;         // pinned
  int64_t sum_cap = 0;
  std::mutex call_mu;              // one evaluation at a time
  // VGX_REDUCE_RCCL: one communicator per shard (ncclCommInitAll over the shards' devices)
  int reduction = VGX_REDUCE_PEER_SUM;
  std::vector<ncclComm_t> comms;
};

static void run_shard(vgx_reg_multi_s* m, vgx_reg_multi_s::Shard& s) {
  s.rc = VGX_OK;
  const int32_t nl = (int32_t)s.global.size();
  if (m->mode == 0) {
    s.rc = vgx_reg_batch_evaluate_normal(s.batch, m->poses, m->n_nodes, nullptr, nullptr,
                                         nl ? s.status.data() : nullptr);
    if (s.rc == VGX_OK && m->reduction == VGX_REDUCE_RCCL)
      s.rc = vgx_reg_batch_scatter_normal(s.batch, nullptr, s.d_all, 1);
    if (s.rc == VGX_OK && (hipSetDevice(s.ctx->device) != hipSuccess ||
                           hipEventRecord(s.done, s.ctx->stream) != hipSuccess))
      s.rc = VGX_ERR_HIP;
  } else if (m->mode == 1) {
    s.rc = vgx_reg_batch_evaluate_normal(s.batch, m->poses, m->n_nodes, nullptr, nl ? s.h_normal : nullptr,
                                         nl ? s.status.data() : nullptr);
  } else {
    // cost only (vgx_reg_multi_evaluate_cost): the shard's costs into the first n_local doubles of its pinned block array
    s.rc = vgx_reg_batch_evaluate_cost(s.batch, m->poses, m->n_nodes, nullptr, nl ? s.h_normal : nullptr,
                                       nl ? s.status.data() : nullptr);
  }
}

static void worker_loop(vgx_reg_multi_s* m, int k) {
  uint64_t seen = 0;
  vgx_reg_multi_s::Shard& s = *m->shards[(size_t)k];
  (void)hipSetDevice(s.ctx->device);
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(m->mu);
      m->go.wait(lk, [&] { return m->quit || m->generation != seen; });
      if (m->quit) return;
      seen = m->generation;
    }
    run_shard(m, s);
    {
      std::lock_guard<std::mutex> lk(m->mu);
      if (--m->pending == 0) m->finished.notify_all();
    }
  }
}

static int dispatch(vgx_reg_multi_s* m, int mode, const double* poses, int32_t n_nodes) {
  {
    std::lock_guard<std::mutex> lk(m->mu);
    m->mode = mode;
    m->poses = poses;
    m->n_nodes = n_nodes;
    m->pending = (int)m->shards.size();
    ++m->generation;
  }
  m->go.notify_all();
  std::unique_lock<std::mutex> lk(m->mu);
  m->finished.wait(lk, [&] { return m->pending == 0; });
  for (auto& s : m->shards)
    if (s->rc != VGX_OK) return s->rc;
  return VGX_OK;
}

extern "C" {

int vgx_lpt_shards(int32_t n, const int64_t* weight, int32_t n_shards, int32_t* shard_of) {
  if (n < 0 || n_shards <= 0 || (n > 0 && (!weight || !shard_of))) return VGX_ERR_INVALID;
  // greedy longest-processing-time: heaviest constraint first onto the least loaded shard
  // (ties: lower constraint index first, lower shard index first)
  std::vector<int32_t> order((size_t)n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return weight[a] > weight[b]; });
  std::vector<int64_t> load((size_t)n_shards, 0);
  for (int32_t c : order) {
    int best = 0;
    for (int k = 1; k < n_shards; ++k)
      if (load[(size_t)k] < load[(size_t)best]) best = k;
    shard_of[c] = best;
    load[(size_t)best] += weight[c];
  }
  return VGX_OK;
}

int vgx_contiguous_shards(int32_t n, const int64_t* weight, int32_t n_shards, int32_t* shard_of) {
  if (n < 0 || n_shards <= 0 || (n > 0 && (!weight || !shard_of))) return VGX_ERR_INVALID;
  // the list cut into n_shards consecutive runs of (nearly) equal weight: constraint c goes to the shard its
  // weight MIDPOINT falls in, k = floor((prefix(c) + w(c) / 2) * n_shards / total).  voxgraph creates
  // constraints in submap order, i.e. along the trajectory, so a run touches the submaps of one stretch of
  // the map (+ what it overlaps): a shard needs only those resident.
  long double total = 0;
  for (int32_t c = 0; c < n; ++c) {
    if (weight[c] < 0) return VGX_ERR_INVALID;
    total += (long double)weight[c];
  }
  long double prefix = 0;
  for (int32_t c = 0; c < n; ++c) {
    int k = total > 0 ? (int)(((prefix + 0.5L * (long double)weight[c]) * n_shards) / total)
                      : (int)(((long long)c * n_shards) / std::max(n, 1));
    shard_of[c] = std::min(std::max(k, 0), n_shards - 1);
    prefix += (long double)weight[c];
  }
  return VGX_OK;
}

int vgx_reg_multi_destroy(vgx_reg_multi m) {
  if (!m) return VGX_ERR_INVALID;
  DeviceGuard guard;
  {
    std::lock_guard<std::mutex> lk(m->mu);
    m->quit = true;
  }
  m->go.notify_all();
  for (auto& s : m->shards)
    if (s->th.joinable()) s->th.join();
  for (ncclComm_t c : m->comms)
    if (c) (void)rccl_api().CommDestroy(c);
  for (auto& s : m->shards) {
    (void)hipSetDevice(s->ctx->device);
    if (s->batch) vgx_reg_batch_destroy(s->batch);
    if (s->d_all) (void)hipFree(s->d_all);
    if (s->h_normal) (void)hipHostFree(s->h_normal);
    if (s->done) (void)hipEventDestroy(s->done);
  }
  if (!m->shards.empty()) (void)hipSetDevice(m->shards[0]->ctx->device);
  if (m->assembler) vgx_reg_assembler_destroy(m->assembler);
  if (m->d_normal_all) (void)hipFree(m->d_normal_all);
  if (m->d_src) (void)hipFree((void*)m->d_src);
  if (m->d_sum) (void)hipFree(m->d_sum);
  if (m->h_sum) (void)hipHostFree(m->h_sum);
  delete m;
  return VGX_OK;
}

int vgx_reg_multi_create(int32_t n_ctx, const vgx_ctx* ctxs, int32_t n, const vgx_reg* regs,
                         const int32_t* node_pair, vgx_reg_multi* out) {
  if (!out) return VGX_ERR_INVALID;
  *out = nullptr;
  if (n_ctx <= 0 || n_ctx > kMaxShards || !ctxs || n < 0 || (n > 0 && (!regs || !node_pair))) return VGX_ERR_INVALID;
  for (int k = 0; k < n_ctx; ++k)
    if (!ctxs[k]) return VGX_ERR_INVALID;
  DeviceGuard guard;
  vgx_ctx ctx0 = ctxs[0];
  vgx_reg_multi m = new (std::nothrow) vgx_reg_multi_s();
  if (!m) return set_error(ctx0, VGX_ERR_NOMEM, "vgx_reg_multi_create: out of host memory");
  m->n = n;
  m->shard_of.assign((size_t)n, -1);
  for (int k = 0; k < n_ctx; ++k) {
    m->shards.emplace_back(new vgx_reg_multi_s::Shard());
    m->shards.back()->ctx = ctxs[k];
  }
  // a constraint runs on the context its cost function (and therefore its two submaps) lives on
  for (int c = 0; c < n; ++c) {
    int k = 0;
    while (k < n_ctx && (!regs[c] || regs[c]->ctx != ctxs[k])) ++k;
    if (k == n_ctx) {
      vgx_reg_multi_destroy(m);
      return set_error(ctx0, VGX_ERR_INVALID, "vgx_reg_multi_create: a constraint belongs to none of the contexts");
    }
    m->shard_of[(size_t)c] = k;
    m->shards[(size_t)k]->global.push_back(c);
  }
  int rc = VGX_OK;
  for (int k = 0; k < n_ctx && rc == VGX_OK; ++k) {
    vgx_reg_multi_s::Shard& s = *m->shards[(size_t)k];
    const int32_t nl = (int32_t)s.global.size();
    std::vector<vgx_reg> r((size_t)nl);
    std::vector<int32_t> np(2 * (size_t)nl);
    for (int32_t c = 0; c < nl; ++c) {
      r[(size_t)c] = regs[s.global[(size_t)c]];
      np[2 * (size_t)c] = node_pair[2 * (size_t)s.global[(size_t)c]];
      np[2 * (size_t)c + 1] = node_pair[2 * (size_t)s.global[(size_t)c] + 1];
    }
    rc = vgx_reg_batch_create(s.ctx, nl, r.data(), np.data(), s.global.data(), n, &s.batch);
    if (rc != VGX_OK) {
      set_error(ctx0, rc, std::string("vgx_reg_multi_create: shard batch: ") + vgx_last_error(s.ctx));
      break;
    }
    s.status.assign((size_t)nl, 0);
    if (hipSetDevice(s.ctx->device) != hipSuccess ||
        hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess ||
        (nl > 0 && hipHostMalloc((void**)&s.h_normal, (size_t)nl * kNormalSize * sizeof(double), hipHostMallocDefault) != hipSuccess))
      rc = set_error(ctx0, VGX_ERR_HIP, "vgx_reg_multi_create: event / pinned buffer creation failed");
    // shard 0 reads the other shards' buffers directly (xGMI peer mapping)
    if (rc == VGX_OK && k > 0 && s.ctx->device != ctx0->device) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, ctx0->device, s.ctx->device) != hipSuccess || !can) {
        rc = set_error(ctx0, VGX_ERR_UNSUPPORTED, "vgx_reg_multi_create: device 0 cannot peer-map another shard's device");
      } else {
        (void)hipSetDevice(ctx0->device);
        hipError_t e = hipDeviceEnablePeerAccess(s.ctx->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
          rc = set_error(ctx0, VGX_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
        (void)hipGetLastError();
      }
    }
  }
  // context 0: the list's node structure and the table of where every constraint's block will lie
  if (rc == VGX_OK) {
    rc = vgx_reg_assembler_create(ctx0, n, node_pair, &m->assembler);
    if (rc == VGX_OK && n > 0) {
      std::vector<const double*> src((size_t)n, nullptr);
      for (auto& sp : m->shards)
        for (size_t c = 0; c < sp->global.size(); ++c)
          src[(size_t)sp->global[c]] = sp->batch->d_normal + c * kNormalSize;
      if (hipSetDevice(ctx0->device) != hipSuccess ||
          hipMalloc((void**)&m->d_src, (size_t)n * sizeof(double*)) != hipSuccess ||
          hipMemcpy((void*)m->d_src, src.data(), (size_t)n * sizeof(double*), hipMemcpyHostToDevice) != hipSuccess ||
          hipMalloc(&m->d_normal_all, (size_t)n * kNormalSize * sizeof(double)) != hipSuccess)
        rc = set_error(ctx0, VGX_ERR_NOMEM, "vgx_reg_multi_create: context 0's gather buffers");
    }
  }
  if (rc != VGX_OK) {
    vgx_reg_multi_destroy(m);
    return rc;
  }
  for (int k = 0; k < n_ctx; ++k) m->shards[(size_t)k]->th = std::thread(worker_loop, m, k);
  *out = m;
  return VGX_OK;
}

int32_t vgx_reg_multi_num_shards(vgx_reg_multi m) { return m ? (int32_t)m->shards.size() : -1; }

int vgx_reg_multi_set_reduction(vgx_reg_multi m, int32_t reduction) {
  if (!m || (reduction != VGX_REDUCE_PEER_SUM && reduction != VGX_REDUCE_RCCL)) return VGX_ERR_INVALID;
  DeviceGuard guard;
  std::lock_guard<std::mutex> call(m->call_mu);
  vgx_ctx ctx0 = m->shards[0]->ctx;
  if (reduction == VGX_REDUCE_RCCL && m->comms.empty()) {
    RcclApi& api = rccl_api();
    if (!api.ok()) return set_error(ctx0, VGX_ERR_UNSUPPORTED, "vgx_reg_multi_set_reduction: librccl.so could not be opened");
    std::vector<int> devs;
    for (auto& s : m->shards) devs.push_back(s->ctx->device);
    for (size_t a = 0; a < devs.size(); ++a)
      for (size_t b = a + 1; b < devs.size(); ++b)
        if (devs[a] == devs[b])
          return set_error(ctx0, VGX_ERR_UNSUPPORTED,
                           "vgx_reg_multi_set_reduction: RCCL needs one device per context (two contexts share a device)");
    m->comms.assign(devs.size(), nullptr);
    ncclResult_t r = api.CommInitAll(m->comms.data(), (int)devs.size(), devs.data());
    if (r != ncclSuccess) {
      m->comms.clear();
      return set_error(ctx0, VGX_ERR_HIP, std::string("ncclCommInitAll: ") + api.GetErrorString(r));
    }
  }
  if (reduction == VGX_REDUCE_RCCL)
    for (auto& sp : m->shards)
      if (!sp->d_all) {
        VGX_HIP(ctx0, hipSetDevice(sp->ctx->device));
        VGX_HIP(ctx0, hipMalloc(&sp->d_all, std::max<size_t>((size_t)m->n * kNormalSize * sizeof(double), 8)));
      }
  m->reduction = reduction;
  return VGX_OK;
}

int vgx_reg_multi_shard_of(vgx_reg_multi m, int32_t* shard_of) {
  if (!m || !shard_of) return VGX_ERR_INVALID;
  std::copy(m->shard_of.begin(), m->shard_of.end(), shard_of);
  return VGX_OK;
}

int vgx_reg_multi_evaluate_fused(vgx_reg_multi m, const double* poses, int32_t n_nodes, double* fused_host,
                                 int32_t* status) {
  if (!m || !poses || !fused_host || n_nodes <= 0) return VGX_ERR_INVALID;
  DeviceGuard guard;
  std::lock_guard<std::mutex> call(m->call_mu);
  vgx_ctx ctx0 = m->shards[0]->ctx;
  const int64_t size = vgx_reg_fused_size(n_nodes, m->n);
  VGX_HIP(ctx0, hipSetDevice(ctx0->device));
  if (m->sum_cap < size) {
    VGX_HIP(ctx0, hipStreamSynchronize(ctx0->stream));
    if (m->d_sum) (void)hipFree(m->d_sum);
    if (m->h_sum) (void)hipHostFree(m->h_sum);
    m->d_sum = nullptr;
    m->h_sum = nullptr;
    m->sum_cap = 0;
    VGX_HIP(ctx0, hipMalloc(&m->d_sum, (size_t)size * sizeof(double)));
    VGX_HIP(ctx0, hipHostMalloc((void**)&m->h_sum, (size_t)size * sizeof(double), hipHostMallocDefault));
    m->sum_cap = size;
  }
  int rc = dispatch(m, 0, poses, n_nodes);
  if (rc != VGX_OK) {
    for (auto& s : m->shards)
      if (s->rc != VGX_OK) return set_error(ctx0, s->rc, std::string("vgx_reg_multi: shard failed: ") + vgx_last_error(s->ctx));
    return rc;
  }
  // The per-constraint blocks meet on context 0 -- copied, not summed -- and the fused buffer is assembled
  // there ONCE, in list order: bit for bit what one vgx_reg_batch over the whole list computes, whatever the
  // number of contexts and the placement (tests/test_multi_gpu.py: 8 contexts == 2 contexts == single batch).
  const double* d_all = m->d_normal_all;
  if (m->reduction == VGX_REDUCE_RCCL) {
    // one all-reduce per solver evaluation: every context's [n][45] array in place, each on its own stream
    // (behind that context's evaluation and scatter).  A row is non-zero on exactly one context, so the sum is
    // exact in whatever order RCCL adds: the same bits as the gather.
    RcclApi& api = rccl_api();
    ncclResult_t r = api.GroupStart();
    const bool group_open = r == ncclSuccess;
    bool device_failed = false;
    for (size_t k = 0; k < m->shards.size() && r == ncclSuccess && !device_failed; ++k) {
      vgx_reg_multi_s::Shard& s = *m->shards[k];
      // the stream belongs to the shard's context: nobody else enqueues on it while the collective goes in
      std::lock_guard<std::mutex> lk(s.ctx->mu);
      if (hipSetDevice(s.ctx->device) != hipSuccess) {
        device_failed = true;  // (no early return: an open RCCL group would stay open for the whole process)
        break;
      }
      // summed as 64-bit INTEGERS: every word is non-zero on at most one context, so the integer sum IS that
      // context's bit pattern whatever the order -- an f64 sum would also turn a -0.0 into +0.0
      r = api.AllReduce(s.d_all, s.d_all, (size_t)m->n * kNormalSize, ncclInt64, ncclSum, m->comms[k], s.ctx->stream);
    }
    if (group_open) {
      const ncclResult_t r2 = api.GroupEnd();
      if (r == ncclSuccess) r = r2;
    }
    if (device_failed) return set_error(ctx0, VGX_ERR_HIP, "vgx_reg_multi: hipSetDevice failed while enqueuing the all-reduce");
    if (r != ncclSuccess) return set_error(ctx0, VGX_ERR_HIP, std::string("ncclAllReduce: ") + api.GetErrorString(r));
    VGX_HIP(ctx0, hipSetDevice(ctx0->device));
    d_all = m->shards[0]->d_all;
  } else if (m->n > 0) {
    // one gather per solver evaluation, on context 0's stream, behind every context's evaluation
    VGX_HIP(ctx0, hipSetDevice(ctx0->device));
    for (size_t k = 1; k < m->shards.size(); ++k)
      VGX_HIP(ctx0, hipStreamWaitEvent(ctx0->stream, m->shards[k]->done, 0));
    const long long work = (long long)m->n * kNormalSize;
    hipLaunchKernelGGL(multi_gather_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, ctx0->stream, m->d_src,
                       (long long)m->n, m->d_normal_all);
    VGX_HIP(ctx0, hipGetLastError());
  }
  rc = vgx_reg_assembler_assemble(m->assembler, d_all, n_nodes, m->d_sum);
  if (rc != VGX_OK) return rc;
  const double* d_result = m->d_sum;
  VGX_HIP(ctx0, hipMemcpyAsync(m->h_sum, d_result, (size_t)size * sizeof(double), hipMemcpyDeviceToHost, ctx0->stream));
  VGX_HIP(ctx0, hipStreamSynchronize(ctx0->stream));
  std::memcpy(fused_host, m->h_sum, (size_t)size * sizeof(double));
  if (status)
    for (auto& s : m->shards)
      for (size_t c = 0; c < s->global.size(); ++c) status[s->global[c]] = s->status[c];
  return VGX_OK;
}

int vgx_reg_multi_evaluate_normal(vgx_reg_multi m, const double* poses, int32_t n_nodes, double* normal_host,
                                  int32_t* status) {
  if (!m || !poses || !normal_host || n_nodes <= 0) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> call(m->call_mu);
  vgx_ctx ctx0 = m->shards[0]->ctx;
  int rc = dispatch(m, 1, poses, n_nodes);
  if (rc != VGX_OK) {
    for (auto& s : m->shards)
      if (s->rc != VGX_OK) return set_error(ctx0, s->rc, std::string("vgx_reg_multi: shard failed: ") + vgx_last_error(s->ctx));
    return rc;
  }
  // no collective: every shard hands its own constraints' blocks back
  for (auto& s : m->shards)
    for (size_t c = 0; c < s->global.size(); ++c) {
      std::memcpy(normal_host + (size_t)s->global[c] * kNormalSize, s->h_normal + c * kNormalSize,
                  kNormalSize * sizeof(double));
      if (status) status[s->global[c]] = s->status[c];
    }
  return VGX_OK;
}

// The cost-only evaluation (vgx_reg_batch_evaluate_cost) of every context's share: cost_host[c] in the caller's
// constraint order, element 0 of vgx_reg_multi_evaluate_normal's blocks bit for bit; no reduction either.
int vgx_reg_multi_evaluate_cost(vgx_reg_multi m, const double* poses, int32_t n_nodes, double* cost_host, int32_t* status) {
  if (!m || !poses || !cost_host || n_nodes <= 0) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> call(m->call_mu);
  vgx_ctx ctx0 = m->shards[0]->ctx;
  int rc = dispatch(m, 2, poses, n_nodes);
  if (rc != VGX_OK) {
    for (auto& s : m->shards)
      if (s->rc != VGX_OK) return set_error(ctx0, s->rc, std::string("vgx_reg_multi: shard failed: ") + vgx_last_error(s->ctx));
    return rc;
  }
  for (auto& s : m->shards)
    for (size_t c = 0; c < s->global.size(); ++c) {
      cost_host[s->global[c]] = s->h_normal[c];
      if (status) status[s->global[c]] = s->status[c];
    }
  return VGX_OK;
}

}  // extern "C"
