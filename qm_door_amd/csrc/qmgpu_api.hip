// Device context and kernel orchestration behind the C ABI (include/qmgpu.h).
//
// One handle owns one HIP stream and all scratch in HBM, sized for (max_batch, max_nodes) at create time and laid out
// horizon-stacked (layout.h).  A call enqueues, on that stream:
//     mpc_init -> lq_node (batch*(N+1) wavefronts) -> riccati (batch wavefronts) -> linesearch (batch workgroups)
//     -> policy_eval -> wbc (batch workgroups)
// with no host synchronisation in between; the caller synchronises when it needs the results.
// There is no CPU fallback: without a HIP device qmgpu_create returns QMGPU_ERR_NO_DEVICE.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/qmgpu.h"
#include "host/host_error.h"
#include "kernels/mpc_pipeline.h"
#include "kernels/wbc_kernel.h"
#include "kernels/frontend_kernel.h"
#include "mpc32.h"

using namespace qmhost;
using namespace qmk;

#define HIP_CHECK(expr)                                                                                       \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess) throw HipFailure(std::string(#expr) + " failed: " + hipGetErrorString(e_));         \
  } while (0)

// Every entry point runs on the device its handle was created for and leaves the caller's current device as it found it (a
// process may hold handles on several GPUs).
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) HIP_CHECK(hipSetDevice(device));
    else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

struct qmgpu_context {
  int device = 0, maxBatch = 0, maxNodes = 0;
  hipStream_t ownStream = nullptr, stream = nullptr;
  // qmgpu_set_overlap: the WBC of qmgpu_cycle_batch runs on a stream of its own; the next cycle's node kernels fill the CUs its fast instances have left
  hipStream_t wbcStream = nullptr;
  hipEvent_t evPolicy = nullptr, evWbc = nullptr;
  bool overlap = false, wbcPending = false;
  qmgpu_problem hostProblem;
  int dtype = QMGPU_F64;
  // device buffers of the fp64 kernels (MPC scratch, model / settings, R'); the WBC and the front end always run in fp64
  MpcBuffers m;
  qmk32::Mpc32* m32 = nullptr;   // fp32 MPC path (dtype == QMGPU_F32): its own scratch, staging and launch chain (qmgpu_mpc32.hip)
  // policy evaluation outputs feeding the WBC inside qmgpu_cycle_batch
  double *dPolX = nullptr, *dPolU = nullptr;
  int* dPolMode = nullptr;
  qmgpu_gait* dGaits = nullptr;   // gait templates of the last qmgpu_gait_schedule_batch call
  std::vector<void*> allocations;
  bool timing = false, debugLq = false;
  // HIP-event ring: one set of 7 events per call while timing is enabled, read back without a per-call sync
  static constexpr int kRing = 256;
  hipEvent_t ring[kRing][7];
  int ringKind[kRing];  // bit0: mpc recorded, bit1: wbc recorded
  long callCount = 0;   // calls recorded since timing was enabled
  hipEvent_t* ev = nullptr;
  double lastMs[5] = {0, 0, 0, 0, 0};
  int lastBatch = 0, lastN = 0;

  std::vector<std::pair<void*, size_t>> scratch;   // buffers every call rewrites (qmgpu_debug_poison fills them with NaN)
  template <class T> T* alloc(size_t count, bool isScratch = true) {
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, count * sizeof(T)));
    allocations.push_back(p);
    if (isScratch) scratch.emplace_back(p, count * sizeof(T));
    return static_cast<T*>(p);
  }
};

static void checkTopology(const qmgpu_model& m) {
  bool ok = m.parent[0] == -1;
  for (int l = 0; l < 4; ++l) ok = ok && m.parent[1 + 3 * l] == 0 && m.parent[2 + 3 * l] == 1 + 3 * l && m.parent[3 + 3 * l] == 2 + 3 * l;
  ok = ok && m.parent[13] == 0;
  for (int a = 1; a < 6; ++a) ok = ok && m.parent[13 + a] == 12 + a;
  bool seen[4] = {false, false, false, false};
  for (int c = 0; c < 4; ++c) { const int b = m.foot_body[c]; ok = ok && (b == 3 || b == 6 || b == 9 || b == 12); if (ok) seen[b / 3 - 1] = true; }
  ok = ok && seen[0] && seen[1] && seen[2] && seen[3] && m.ee_body == 18;
  for (int l = 0; l < 4; ++l) for (int j = 0; j < 3; ++j) ok = ok && m.axis[1 + 3 * l + j] == LEG_AXIS[j];   // the tree sweep hard-codes the joint axes
  for (int a = 0; a < 6; ++a) ok = ok && m.axis[13 + a] == ARM_AXIS[a];
  if (!ok) throw UnsupportedModel("kernels are specialised to the AlienGo+Z1 topology (4 x 3-joint legs + 6-joint arm)");
}

extern "C" {

int qmgpu_create(const qmgpu_problem* problem, int device, int max_batch, int max_nodes, qmgpu_handle* out) {
  return qmgpu_create_ex(problem, device, max_batch, max_nodes, QMGPU_F64, out);
}

int qmgpu_create_ex(const qmgpu_problem* problem, int device, int max_batch, int max_nodes, int dtype, qmgpu_handle* out) {
  if (!problem || !out || max_batch < 1 || max_nodes < 1 || (dtype != QMGPU_F64 && dtype != QMGPU_F32)) return setError(QMGPU_ERR_INVALID_ARGUMENT, "bad arguments to qmgpu_create");
  *out = nullptr;
  qmgpu_context* ctx = nullptr;
  const int st = guarded([&]() {
    checkTopology(problem->model);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw NoDevice("no HIP device visible; qm_door_amd has no CPU execution path");
    if (device < 0 || device >= count) throw NoDevice("HIP device index out of range");
    DeviceGuard onDevice(device);
    ctx = new qmgpu_context();
    for (auto& set : ctx->ring) for (auto& e : set) e = nullptr;
    ctx->device = device; ctx->maxBatch = max_batch; ctx->maxNodes = max_nodes; ctx->hostProblem = *problem; ctx->dtype = dtype;
    HIP_CHECK(hipStreamCreate(&ctx->ownStream));
    ctx->stream = ctx->ownStream;
    const size_t B = size_t(max_batch), N1 = size_t(max_nodes) + 1, N = size_t(max_nodes);
    qmgpu_context* const owner = ctx;   // by value: the fp32 path keeps this allocator for buffers it creates on first use
    auto rawAlloc = [owner](size_t count, size_t elem, bool scratch) { return static_cast<void*>(owner->alloc<char>(count * elem, scratch)); };
    // the fp32 handle keeps the fp64 model / settings / R' (the WBC and the front end read them) but not the fp64 MPC scratch
    if (dtype == QMGPU_F64) allocateMpcBuffers(ctx->m, B, N, rawAlloc);
    else { ctx->m.dP = ctx->alloc<qmgpu_problem>(1, false); ctx->m.dRw = ctx->alloc<double>(qmk::QM_RW_DOUBLES, false); ctx->m.dZeros = ctx->alloc<double>(64, false); }
    ctx->dPolX = ctx->alloc<double>(B * 30);
    ctx->dPolU = ctx->alloc<double>(B * 30);
    ctx->dPolMode = ctx->alloc<int>(B);
    HIP_CHECK(hipMemcpy(ctx->m.dP, problem, sizeof(qmgpu_problem), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemsetAsync(ctx->m.dZeros, 0, 64 * sizeof(double), ctx->stream));
    for (auto& set : ctx->ring) for (auto& e : set) HIP_CHECK(hipEventCreate(&e));
    ctx->ev = ctx->ring[0];
    HIP_CHECK(QM_ALLOW_DYNAMIC_LDS(wbc_kernel, WBC_LDS_BYTES));
    HIP_CHECK(prepareMpcKernels());
    QM_LAUNCH(input_weight_kernel, 1, 64, ctx->stream, ctx->m.dP, ctx->m.dZeros, ctx->m.dRw);
    HIP_CHECK(hipGetLastError());
    if (dtype == QMGPU_F32) {
      ctx->m32 = qmk32::create(*problem, max_batch, max_nodes, ctx->stream, rawAlloc);
      if (!ctx->m32) throw HipFailure("fp32 MPC path could not be created");
    }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
  });
  if (st != QMGPU_OK) {
    if (ctx) {   // nothing of a failed create survives: device memory, events, stream
      int prev = -1;
      const bool sw = hipGetDevice(&prev) == hipSuccess && prev != ctx->device && hipSetDevice(ctx->device) == hipSuccess;
      for (void* p : ctx->allocations) hipFree(p);
      for (auto& set : ctx->ring) for (auto& e : set) if (e) hipEventDestroy(e);
      if (ctx->ownStream) hipStreamDestroy(ctx->ownStream);
      if (sw) hipSetDevice(prev);
      qmk32::destroy(ctx->m32);
      delete ctx;
    }
    return st;
  }
  *out = ctx;
  return QMGPU_OK;
}

int qmgpu_destroy(qmgpu_handle h) {
  if (!h) return QMGPU_OK;
  int prev = -1;
  const bool sw = hipGetDevice(&prev) == hipSuccess && prev != h->device && hipSetDevice(h->device) == hipSuccess;
  hipStreamSynchronize(h->stream);
  if (h->wbcStream) { hipStreamSynchronize(h->wbcStream); hipStreamDestroy(h->wbcStream); }
  if (h->evPolicy) hipEventDestroy(h->evPolicy);
  if (h->evWbc) hipEventDestroy(h->evWbc);
  for (void* p : h->allocations) hipFree(p);
  for (auto& set : h->ring) for (auto& e : set) if (e) hipEventDestroy(e);
  if (h->ownStream) hipStreamDestroy(h->ownStream);
  if (sw) hipSetDevice(prev);
  qmk32::destroy(h->m32);
  delete h;
  return QMGPU_OK;
}

// the handle's stream waits for a WBC still running on the overlap stream (no host wait); after it everything is ordered on h->stream again
static void joinWbc(qmgpu_handle h) {
  if (h->wbcPending) { HIP_CHECK(hipStreamWaitEvent(h->stream, h->evWbc, 0)); h->wbcPending = false; }
}

int qmgpu_set_stream(qmgpu_handle h, void* hip_stream) {
  if (!h) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null handle");
  // the NEW stream is the one that has to wait for a WBC still pending on the overlap stream: everything enqueued from here on goes there
  return guarded([&]() { DeviceGuard onDevice(h->device); h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->ownStream; joinWbc(h); });
}

int qmgpu_synchronize(qmgpu_handle h) {
  if (!h) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null handle");
  return guarded([&]() { if (!h) throw std::invalid_argument("null handle"); DeviceGuard onDevice(h->device); joinWbc(h); HIP_CHECK(hipStreamSynchronize(h->stream)); });
}

int qmgpu_set_overlap(qmgpu_handle h, int enable) {
  if (!h) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null handle");
  return guarded([&]() { DeviceGuard onDevice(h->device);
    joinWbc(h);
    if (enable) {   // (each resource on its own: a call that failed half way is repeated without leaking what it had created)
      // a HIGH-PRIORITY stream: the runtime spreads streams of one priority over a few hardware queues round robin, and two streams that land on the same queue run
      // their kernels one after the other (measured: the second handle of a process got no overlap at all); priority levels have hardware queues of their own.  On a
      // device without priority levels (range 0..0) the stream is an ordinary one: the results are the same, the overlap is whatever the runtime's queue assignment gives
      if (!h->wbcStream) {
        int least = 0, greatest = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_CHECK(hipStreamCreateWithPriority(&h->wbcStream, hipStreamNonBlocking, greatest));
      }
      if (!h->evPolicy) HIP_CHECK(hipEventCreateWithFlags(&h->evPolicy, hipEventDisableTiming));
      if (!h->evWbc) HIP_CHECK(hipEventCreateWithFlags(&h->evWbc, hipEventDisableTiming));
    }
    h->overlap = enable != 0;
  });
}

int qmgpu_join_wbc(qmgpu_handle h) {
  if (!h) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null handle");
  return guarded([&]() { DeviceGuard onDevice(h->device); joinWbc(h); });
}

int qmgpu_update_settings(qmgpu_handle h, const qmgpu_settings* settings) {
  if (!h || !settings) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null argument");
  return guarded([&]() {
    DeviceGuard onDevice(h->device);
    joinWbc(h);
    HIP_CHECK(hipStreamSynchronize(h->stream));   // kernels in flight (a WBC on the overlap stream included) still read the old values through dP
    h->hostProblem.settings = *settings;
    HIP_CHECK(hipMemcpy(&h->m.dP->settings, &h->hostProblem.settings, sizeof(qmgpu_settings), hipMemcpyHostToDevice));
    QM_LAUNCH(input_weight_kernel, 1, 64, h->stream, h->m.dP, h->m.dZeros, h->m.dRw);
    HIP_CHECK(hipGetLastError());
    if (h->m32 && !qmk32::updateProblem(h->m32, h->hostProblem, h->stream)) throw HipFailure("fp32 settings update failed");
  });
}

int qmgpu_get_input_weight(qmgpu_handle h, double* R_host) {
  if (!h || !R_host) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null argument");
  return guarded([&]() { if (!h) throw std::invalid_argument("null handle"); DeviceGuard onDevice(h->device); HIP_CHECK(hipStreamSynchronize(h->stream)); HIP_CHECK(hipMemcpy(R_host, h->m.dRw, 900 * sizeof(double), hipMemcpyDeviceToHost)); });
}

int qmgpu_enable_timing(qmgpu_handle h, int enable) {
  if (!h) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null handle");
  h->timing = enable != 0;
  h->callCount = 0;
  return QMGPU_OK;
}

// fills one CU's worth of LDS with NaN; launched with many more workgroups than CUs so that every CU gets some
__global__ void __launch_bounds__(256) lds_poison_kernel(int doubles, double* sink) {
  QM_DYNAMIC_LDS(lds);
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  for (int e = threadIdx.x; e < doubles; e += 256) lds[e] = nan;
  __syncthreads();
  if (sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) sink[0] = lds[1];   // keeps the stores alive
}

// X | U | WBC output | contact modes (as doubles: exact) of every instance into one row of `packed`: the record the ranks of a sharded batch all-gather
// (qm_door_amd/sharding.py: pack_len / unpack), written by one launch instead of four strided copies into a fresh allocation
__global__ void __launch_bounds__(256) pack_results_kernel(int batch, int N, const double* X, const double* U, const double* wbc, const int32_t* modes, double* packed) {
  const int inst = blockIdx.x;
  if (inst >= batch) return;
  const int nx = (N + 1) * 30, nu = N * 30, len = nx + nu + QMGPU_NWBC_OUT + (N + 1);
  double* row = packed + size_t(inst) * len;
  for (int e = threadIdx.x; e < len; e += 256) {
    double v;
    if (e < nx) v = X[size_t(inst) * nx + e];
    else if (e < nx + nu) v = U[size_t(inst) * nu + (e - nx)];
    else if (e < nx + nu + QMGPU_NWBC_OUT) v = wbc[size_t(inst) * QMGPU_NWBC_OUT + (e - nx - nu)];
    else v = double(modes[size_t(inst) * (N + 1) + (e - nx - nu - QMGPU_NWBC_OUT)]);
    row[e] = v;
  }
}

#ifdef QM_RICCATI_TIMING
// profiling build only (tools/riccati_phase_probe.py): the phase clocks of riccati_kernel's workgroup 0, [wavefront][16]
int qmgpu_debug_riccati_ticks(unsigned long long* out64, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return QMGPU_ERR_HIP;
  if (out64 && hipMemcpyFromSymbol(out64, HIP_SYMBOL(qmk::qmRiccatiTicks), sizeof(unsigned long long) * 2048) != hipSuccess) return QMGPU_ERR_HIP;
  if (reset) { static unsigned long long z[2048] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(qmk::qmRiccatiTicks), z, sizeof(z)) != hipSuccess) return QMGPU_ERR_HIP; }
  return QMGPU_OK;
}
// wall-clock (100 MHz) start / end of every ad_node_kernel workgroup of the last launch
int qmgpu_debug_ad_wg_clocks(unsigned long long* out64, int count) {
  if (hipDeviceSynchronize() != hipSuccess) return QMGPU_ERR_HIP;
  if (count > 2 * QM_AD_WG_CLOCKS) count = 2 * QM_AD_WG_CLOCKS;
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(qmk::qmAdWgClock), sizeof(unsigned long long) * size_t(count)) != hipSuccess) return QMGPU_ERR_HIP;
  return QMGPU_OK;
}
#endif

#ifdef QM_WBC_DUMP
// experiment build only (tools/wbc_variants.py): LDS images of instance 0 of the last wbc_kernel launch at the kernel's checkpoints
extern "C" int qmgpu_debug_wbc_dump(double* out, int doubles) {
  if (hipDeviceSynchronize() != hipSuccess) return QMGPU_ERR_HIP;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(qmk::qmWbcDump), sizeof(double) * size_t(doubles)) != hipSuccess) return QMGPU_ERR_HIP;
  return QMGPU_OK;
}
#endif

int qmgpu_debug_poison(qmgpu_handle h) {
  if (!h) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null handle");
  return guarded([&]() { if (!h) throw std::invalid_argument("null handle"); DeviceGuard onDevice(h->device);
    joinWbc(h);      // (a pending WBC still reads the policy buffers, which are scratch)
    for (auto& sc : h->scratch) HIP_CHECK(hipMemsetAsync(sc.first, 0xFF, sc.second, h->stream));
    constexpr int kDoubles = 160 * 1024 / 8;
    HIP_CHECK(QM_ALLOW_DYNAMIC_LDS(lds_poison_kernel, kDoubles * 8));
    QM_LAUNCH_DYN(lds_poison_kernel, 4096, 256, kDoubles * 8, h->stream, kDoubles, static_cast<double*>(nullptr));
    HIP_CHECK(hipGetLastError());
  });
}

int qmgpu_enable_debug(qmgpu_handle h, int enable) {
  if (!h) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null handle");
  return guarded([&]() { if (!h) throw std::invalid_argument("null handle"); DeviceGuard onDevice(h->device);
    if (enable && h->dtype != QMGPU_F64) throw std::invalid_argument("the per-node LQ dump exists in the fp64 path only");
    if (enable && !h->m.dDebug) h->m.dDebug = h->alloc<double>(size_t(h->maxBatch) * (h->maxNodes + 1) * DBG_DOUBLES);
    h->debugLq = enable != 0;
  });
}

static void checkMpcArgs(qmgpu_handle h, const qmgpu_mpc_args* a) {
  if (!h || !a) throw std::invalid_argument("null argument");
  if (a->batch < 1 || a->num_nodes < 1 || a->num_target_knots < 1) throw std::invalid_argument("batch, num_nodes and num_target_knots must be positive");
  if (a->batch > h->maxBatch || a->num_nodes > h->maxNodes) throw CapacityError("batch / num_nodes exceed the capacity given to qmgpu_create");
  if (h->dtype == QMGPU_F32 && a->num_target_knots > QMGPU_F32_MAX_TARGET_KNOTS) throw CapacityError("an fp32 handle stages at most QMGPU_F32_MAX_TARGET_KNOTS target knots per instance");
  if (!a->x0 || !a->target_times || !a->target_states || !a->sched_num_events || !a->sched_event_times || !a->sched_modes) throw std::invalid_argument("missing MPC input pointer");
  if (!a->time_grid && !a->t0) throw std::invalid_argument("either t0 or time_grid is required");
  if (!a->out_t || !a->out_x || !a->out_u || !a->out_mode) throw std::invalid_argument("missing MPC output pointer");
}

static void enqueueMpc(qmgpu_handle h, const qmgpu_mpc_args* a) {
  const int iterations = h->hostProblem.settings.sqp_iterations > 1 ? h->hostProblem.settings.sqp_iterations : 1;
  hipEvent_t* ev = h->timing ? h->ev : nullptr;
  if (a->algorithm != QMGPU_ALG_SQP && a->algorithm != QMGPU_ALG_DDP) throw std::invalid_argument("unknown qmgpu_mpc_args::algorithm");
  const int trials = ddpTrialCount(h->hostProblem.settings.ddp_min_step, h->hostProblem.settings.ddp_max_step);
  if (h->dtype == QMGPU_F32) {
    if (!qmk32::enqueue(h->m32, h->stream, a, h->hostProblem.settings.dt, iterations, trials, ev)) throw HipFailure("fp32 MPC launch failed");
  } else if (a->algorithm == QMGPU_ALG_DDP) {
    ensureDdpBuffers(h->m, size_t(h->maxBatch), size_t(h->maxNodes), [&](size_t count, size_t elem, bool scratch) { return static_cast<void*>(h->alloc<char>(count * elem, scratch)); });
    const MpcIo io{a->batch, a->num_nodes, a->num_target_knots, a->line_search, h->hostProblem.settings.dt, a->t0, a->time_grid, a->sched_event_times, a->x0, a->target_times,
                   a->target_states, a->sched_num_events, a->sched_event_times, a->sched_modes, a->warm_x, a->warm_u, a->out_t, a->out_x, a->out_u, a->out_mode, a->out_stats, a->ee_contact_ref,
                   a->algorithm};
    enqueueDdpKernels(h->stream, h->m, io, trials, ev);
  } else {
    const MpcIo io{a->batch, a->num_nodes, a->num_target_knots, a->line_search, h->hostProblem.settings.dt, a->t0, a->time_grid, a->sched_event_times, a->x0, a->target_times,
                   a->target_states, a->sched_num_events, a->sched_event_times, a->sched_modes, a->warm_x, a->warm_u, a->out_t, a->out_x, a->out_u, a->out_mode, a->out_stats, a->ee_contact_ref, a->algorithm};
    enqueueMpcKernels(h->stream, h->m, io, iterations, h->debugLq, ev);
  }
  HIP_CHECK(hipGetLastError());
  h->lastBatch = a->batch; h->lastN = a->num_nodes;
}

static void enqueueWbc(qmgpu_handle h, const qmgpu_wbc_args* w, hipStream_t stream) {
  if (!w || w->batch < 1) throw std::invalid_argument("bad WBC arguments");
  if (w->batch > h->maxBatch) throw CapacityError("WBC batch exceeds the capacity given to qmgpu_create");
  if (!w->state_desired || !w->input_desired || !w->rbd_measured || !w->mode || !w->period || !w->time || !w->input_last || !w->out) throw std::invalid_argument("missing WBC pointer");
  static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "working-set words");
  WbcArgs wa{h->m.dP, w->batch, w->variant, w->state_desired, w->input_desired, w->rbd_measured, w->mode, w->period, w->time, w->input_last, w->out, w->out_status, w->ee_force,
             reinterpret_cast<unsigned long long*>(w->working_set)};
  QM_LAUNCH_DYN(wbc_kernel, w->batch, WBC_THREADS, WBC_LDS_BYTES, stream, wa);
  HIP_CHECK(hipGetLastError());
}

static void beginTiming(qmgpu_handle h) {
  if (h->timing) h->ev = h->ring[h->callCount % qmgpu_context::kRing];
}
static void finishTiming(qmgpu_handle h, bool mpc, bool wbc) {
  if (!h->timing) return;
  h->ringKind[h->callCount % qmgpu_context::kRing] = (mpc ? 1 : 0) | (wbc ? 2 : 0);
  ++h->callCount;
}
// elapsed times of one recorded call: [ad_node, lq_node, riccati, linesearch, wbc, whole]  (events: 0 start, 6 after ad, 1 after lq, 2, 3, 4/5 wbc)
static void readTiming(qmgpu_handle h, long call, double* ms6) {
  hipEvent_t* ev = h->ring[call % qmgpu_context::kRing];
  const int kind = h->ringKind[call % qmgpu_context::kRing];
  const bool mpc = kind & 1, wbc = kind & 2;
  HIP_CHECK(hipEventSynchronize(ev[wbc ? 5 : 3]));
  float ms = 0.f;
  for (int i = 0; i < 6; ++i) ms6[i] = 0.0;
  if (mpc) {
    HIP_CHECK(hipEventElapsedTime(&ms, ev[0], ev[6])); ms6[0] = ms;
    HIP_CHECK(hipEventElapsedTime(&ms, ev[6], ev[1])); ms6[1] = ms;
    HIP_CHECK(hipEventElapsedTime(&ms, ev[1], ev[2])); ms6[2] = ms;
    HIP_CHECK(hipEventElapsedTime(&ms, ev[2], ev[3])); ms6[3] = ms;
  }
  if (wbc) { HIP_CHECK(hipEventElapsedTime(&ms, ev[4], ev[5])); ms6[4] = ms; }
  HIP_CHECK(hipEventElapsedTime(&ms, ev[mpc ? 0 : 4], ev[wbc ? 5 : 3]));
  ms6[5] = ms;
}

int qmgpu_mpc_solve_batch(qmgpu_handle h, const qmgpu_mpc_args* args) {
  return guarded([&]() { if (!h) throw std::invalid_argument("null handle"); DeviceGuard onDevice(h->device); checkMpcArgs(h, args); beginTiming(h); enqueueMpc(h, args); finishTiming(h, true, false); });
}

int qmgpu_policy_eval_batch(qmgpu_handle h, int batch, int num_nodes, const double* t_grid, const double* X, const double* U, const int32_t* modes, const double* t_eval,
                            double* x_out, double* u_out, int32_t* mode_out) {
  if (!h || !t_grid || !X || !U || !modes || !t_eval || !x_out || !u_out || !mode_out || batch < 1 || num_nodes < 1) return setError(QMGPU_ERR_INVALID_ARGUMENT, "bad arguments");
  return guarded([&]() { if (!h) throw std::invalid_argument("null handle"); DeviceGuard onDevice(h->device);
    QM_LAUNCH(policy_eval_kernel, batch, 64, h->stream, batch, num_nodes, t_grid, X, U, modes, t_eval, x_out, u_out, mode_out);
    HIP_CHECK(hipGetLastError());
  });
}

int qmgpu_frontend_batch(qmgpu_handle h, const qmgpu_frontend_args* a) {
  if (!h || !a || a->batch < 1 || !a->rbd_measured || !a->time || !a->command_kind || !a->command || !a->last_ee_target || !a->x0 || !a->target_times || !a->target_states)
    return setError(QMGPU_ERR_INVALID_ARGUMENT, "bad front-end arguments");
  return guarded([&]() { if (!h) throw std::invalid_argument("null handle"); DeviceGuard onDevice(h->device);
    FrontendArgs fa{h->m.dP, *a};
    QM_LAUNCH(frontend_kernel, (a->batch + 63) / 64, 64, h->stream, fa);
    HIP_CHECK(hipGetLastError());
  });
}

int qmgpu_gait_schedule_batch(qmgpu_handle h, int batch, const qmgpu_gait* templates, int num_templates, const int32_t* gait_index, const int32_t* prev_mode,
                              const double* t_phase0, const double* t_begin, const double* t_end, int32_t* sched_num_events, double* sched_event_times, int32_t* sched_modes, int32_t* status) {
  if (!h || batch < 1 || !templates || num_templates < 1 || num_templates > 64 || !gait_index || !t_phase0 || !t_begin || !t_end || !sched_num_events || !sched_event_times || !sched_modes)
    return setError(QMGPU_ERR_INVALID_ARGUMENT, "bad gait schedule arguments");
  return guarded([&]() {
    DeviceGuard onDevice(h->device);
    if (!h->dGaits) h->dGaits = h->alloc<qmgpu_gait>(64, false);
    HIP_CHECK(hipMemcpyAsync(h->dGaits, templates, sizeof(qmgpu_gait) * num_templates, hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));   // `templates` is the caller's (pageable) host memory: nothing is retained after return
    GaitArgs ga{batch, num_templates, h->dGaits, gait_index, prev_mode, h->hostProblem.settings.phase_transition_stance_time, t_phase0, t_begin, t_end, sched_num_events, sched_event_times, sched_modes, status};
    QM_LAUNCH(gait_schedule_kernel, (batch + 63) / 64, 64, h->stream, ga);
    HIP_CHECK(hipGetLastError());
  });
}

int qmgpu_wbc_solve_batch(qmgpu_handle h, const qmgpu_wbc_args* args) {
  if (!h) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null handle");
  return guarded([&]() { if (!h) throw std::invalid_argument("null handle"); DeviceGuard onDevice(h->device);
    joinWbc(h);
    beginTiming(h);
    if (h->timing) HIP_CHECK(hipEventRecord(h->ev[4], h->stream));
    enqueueWbc(h, args, h->stream);
    if (h->timing) HIP_CHECK(hipEventRecord(h->ev[5], h->stream));
    finishTiming(h, false, true);
  });
}

int qmgpu_warm_start_batch(qmgpu_handle h, int batch, int prev_nodes, const double* prev_grid, const double* prev_X, const double* prev_U, int new_nodes,
                           const double* new_grid, const double* x0, double* warm_x, double* warm_u) {
  if (!h || !prev_grid || !prev_X || !prev_U || !new_grid || !warm_x || !warm_u || batch < 1 || prev_nodes < 1 || new_nodes < 1)
    return setError(QMGPU_ERR_INVALID_ARGUMENT, "bad warm-start arguments");
  if (warm_x == prev_X || warm_u == prev_U) return setError(QMGPU_ERR_INVALID_ARGUMENT, "warm-start outputs must not alias the previous solution");
  return guarded([&]() { if (!h) throw std::invalid_argument("null handle"); DeviceGuard onDevice(h->device);
    QM_LAUNCH(warm_start_kernel, batch, 256, h->stream, batch, prev_nodes, prev_grid, prev_X, prev_U, new_nodes, new_grid, x0, warm_x, warm_u);
    HIP_CHECK(hipGetLastError());
  });
}

int qmgpu_cycle_batch(qmgpu_handle h, const qmgpu_mpc_args* mpc, const double* t_eval, qmgpu_wbc_args* wbc) {
  if (!h || !t_eval || !wbc) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null argument");
  return guarded([&]() { if (!h) throw std::invalid_argument("null handle"); DeviceGuard onDevice(h->device);
    checkMpcArgs(h, mpc);
    if (wbc->batch != mpc->batch) throw std::invalid_argument("MPC and WBC batch sizes differ");
    beginTiming(h);
    enqueueMpc(h, mpc);
    joinWbc(h);      // (overlap: the previous cycle's WBC still reads the policy buffers -- it ran next to the node kernels just enqueued)
    QM_LAUNCH(policy_eval_kernel, mpc->batch, 64, h->stream, mpc->batch, mpc->num_nodes, mpc->out_t, mpc->out_x, mpc->out_u, mpc->out_mode, t_eval, h->dPolX, h->dPolU,
              h->dPolMode);
    qmgpu_wbc_args w = *wbc;
    w.state_desired = h->dPolX; w.input_desired = h->dPolU; w.mode = h->dPolMode;
    hipStream_t ws = h->stream;
    if (h->overlap) {
      HIP_CHECK(hipEventRecord(h->evPolicy, h->stream));
      HIP_CHECK(hipStreamWaitEvent(h->wbcStream, h->evPolicy, 0));
      ws = h->wbcStream;
    }
    if (h->timing) HIP_CHECK(hipEventRecord(h->ev[4], ws));
    enqueueWbc(h, &w, ws);
    if (h->timing) HIP_CHECK(hipEventRecord(h->ev[5], ws));
    if (h->overlap) { HIP_CHECK(hipEventRecord(h->evWbc, ws)); h->wbcPending = true; }
    finishTiming(h, true, true);
  });
}

int qmgpu_debug_get_lq(qmgpu_handle h, int instance, int node, double* A, double* B, double* b, double* Q, double* R, double* q, double* r, double* C, double* D, double* e,
                       int32_t* nc) {
  if (!h) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null handle");
  return guarded([&]() { if (!h) throw std::invalid_argument("null handle"); DeviceGuard onDevice(h->device);
    if (!h->debugLq || !h->m.dDebug) throw std::invalid_argument("call qmgpu_enable_debug(h, 1) before the solve");
    if (instance < 0 || instance >= h->lastBatch || node < 0 || node > h->lastN) throw std::invalid_argument("instance / node out of range");
    HIP_CHECK(hipStreamSynchronize(h->stream));
    std::vector<double> rec(DBG_DOUBLES);
    const size_t idx = size_t(instance) * (h->lastN + 1) + node;
    HIP_CHECK(hipMemcpy(rec.data(), h->m.dDebug + idx * DBG_DOUBLES, DBG_DOUBLES * sizeof(double), hipMemcpyDeviceToHost));
    int ncv = 0;
    HIP_CHECK(hipMemcpy(&ncv, h->m.dStageNc + idx, sizeof(int), hipMemcpyDeviceToHost));
    auto cp = [&](double* dst, int off, int n) { if (dst) std::memcpy(dst, rec.data() + off, n * sizeof(double)); };
    cp(A, DBG_A, 900); cp(B, DBG_B, 900); cp(b, DBG_b, 30); cp(Q, DBG_Q, 900); cp(R, DBG_R, 900); cp(q, DBG_q, 30); cp(r, DBG_r, 30);
    cp(C, DBG_C, 16 * 30); cp(D, DBG_D, 16 * 30); cp(e, DBG_e, 16);
    if (nc) *nc = ncv;
  });
}

int qmgpu_last_kernel_ms(qmgpu_handle h, double* ms6) {
  if (!h || !ms6) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null argument");
  return guarded([&]() { if (!h) throw std::invalid_argument("null handle"); DeviceGuard onDevice(h->device);
    if (!h->timing || h->callCount == 0) throw std::invalid_argument("no timed call recorded (qmgpu_enable_timing)");
    readTiming(h, h->callCount - 1, ms6);
  });
}

int qmgpu_kernel_ms_mean(qmgpu_handle h, int last_calls, double* ms6) {
  if (!h || !ms6 || last_calls < 1) return setError(QMGPU_ERR_INVALID_ARGUMENT, "bad argument");
  return guarded([&]() { if (!h) throw std::invalid_argument("null handle"); DeviceGuard onDevice(h->device);
    if (!h->timing || h->callCount == 0) throw std::invalid_argument("no timed call recorded (qmgpu_enable_timing)");
    const long n = std::min<long>(std::min<long>(last_calls, h->callCount), qmgpu_context::kRing);
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (long c = h->callCount - n; c < h->callCount; ++c) { double m[6]; readTiming(h, c, m); for (int i = 0; i < 6; ++i) acc[i] += m[i]; }
    for (int i = 0; i < 6; ++i) ms6[i] = acc[i] / double(n);
  });
}

int qmgpu_pack_results(qmgpu_handle h, int batch, int num_nodes, const double* X, const double* U, const double* wbc_out, const int32_t* modes, double* packed) {
  if (!h || !X || !U || !wbc_out || !modes || !packed || batch < 1 || num_nodes < 1) return setError(QMGPU_ERR_INVALID_ARGUMENT, "bad arguments");
  return guarded([&]() { DeviceGuard onDevice(h->device);
    QM_LAUNCH(pack_results_kernel, batch, 256, h->stream, batch, num_nodes, X, U, wbc_out, modes, packed);
    HIP_CHECK(hipGetLastError());
  });
}

int qmgpu_kernel_ms_history(qmgpu_handle h, int last_calls, double* ms6_per_call) {
  if (!h || !ms6_per_call || last_calls < 1) return setError(QMGPU_ERR_INVALID_ARGUMENT, "bad argument");
  return guarded([&]() { if (!h) throw std::invalid_argument("null handle"); DeviceGuard onDevice(h->device);
    if (!h->timing || h->callCount == 0) throw std::invalid_argument("no timed call recorded (qmgpu_enable_timing)");
    if (last_calls > h->callCount || last_calls > qmgpu_context::kRing) throw std::invalid_argument("more calls asked for than the event ring holds");
    for (long c = h->callCount - last_calls, i = 0; c < h->callCount; ++c, ++i) readTiming(h, c, ms6_per_call + 6 * i);
  });
}

}  // extern "C"
