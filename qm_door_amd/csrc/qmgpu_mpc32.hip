// fp32 build of the MPC kernels (BASELINE.json configs[4]: "fp32 vs fp64 tolerance sweep").  This translation unit is compiled with
//     -DQM_REAL=float -Dqmk=qmk32
// so that the SAME kernel sources as the fp64 path (kernels/*.h, written in terms of `real`) are instantiated a second time in
// namespace qmk32 with v_mfma_f32_16x16x4_f32, fp32 LDS / HBM scratch (half the bytes) and fp32 vector arithmetic.  The boundary of
// the library stays fp64 (the reference's ocs2::scalar_t): the caller's arrays are converted on the device on the way in and out.
// The WBC is not part of this build: its interior point works at complementarity / pivot tolerances of 1e-9 .. 1e-13 that have no
// fp32 counterpart, and it always runs in fp64 on the (converted) policy of either MPC path.
#include "mpc32.h"

#include <new>

#include "kernels/mpc_pipeline.h"

static_assert(sizeof(qmk::real) == 4, "compile this file with -DQM_REAL=float -Dqmk=qmk32");

namespace qmk {   // = qmk32 in this translation unit

constexpr int kMaxKnots32 = QMGPU_F32_MAX_TARGET_KNOTS;   // include/qmgpu.h; checked by checkMpcArgs (QMGPU_ERR_CAPACITY) before this translation unit is reached

__global__ void __launch_bounds__(256) narrow_kernel(const double* src, float* dst, size_t n) {
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) dst[i] = float(src[i]);
}
__global__ void __launch_bounds__(256) widen_kernel(const float* src, double* dst, size_t n) {
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) dst[i] = double(src[i]);
}

struct Mpc32 {
  int maxBatch = 0, maxNodes = 0;
  RawAlloc alloc;   // the handle's allocator (device memory stays on the handle's list)
  MpcBuffers m;
  // fp32 staging of one call's arguments
  float *x0 = nullptr, *tgtT = nullptr, *tgtS = nullptr, *evT = nullptr, *warmX = nullptr, *warmU = nullptr;
  float *outT = nullptr, *outX = nullptr, *outU = nullptr, *outStats = nullptr, *contact = nullptr;
};

static bool uploadProblem(Mpc32* p, const qmgpu_problem& problem, hipStream_t stream) {
  ProblemR host;
  convert(problem, host);
  if (hipStreamSynchronize(stream) != hipSuccess) return false;
  if (hipMemcpy(p->m.dP, &host, sizeof(ProblemR), hipMemcpyHostToDevice) != hipSuccess) return false;
  QM_LAUNCH(input_weight_kernel, 1, 64, stream, p->m.dP, p->m.dZeros, p->m.dRw);
  return hipGetLastError() == hipSuccess;
}

Mpc32* create(const qmgpu_problem& problem, int maxBatch, int maxNodes, hipStream_t stream, const RawAlloc& alloc) {
  Mpc32* p = new (std::nothrow) Mpc32();
  if (!p) return nullptr;
  p->maxBatch = maxBatch; p->maxNodes = maxNodes; p->alloc = alloc;
  const size_t B = size_t(maxBatch), N = size_t(maxNodes), N1 = N + 1;
  allocateMpcBuffers(p->m, B, N, alloc);
  auto F = [&](size_t n) { return static_cast<float*>(alloc(n, sizeof(float), true)); };
  p->x0 = F(B * 30); p->tgtT = F(B * kMaxKnots32); p->tgtS = F(B * kMaxKnots32 * QMGPU_NTARGET); p->evT = F(B * QMGPU_MAX_EVENTS);
  p->warmX = F(B * N1 * 30); p->warmU = F(B * N * 30);
  p->outT = F(B * N1); p->outX = F(B * N1 * 30); p->outU = F(B * N * 30); p->outStats = F(B * QMGPU_NSTATS); p->contact = F(B * kMaxKnots32 * 6);
  if (hipMemsetAsync(p->m.dZeros, 0, 64 * sizeof(float), stream) != hipSuccess || prepareMpcKernels() != hipSuccess || !uploadProblem(p, problem, stream) ||
      hipStreamSynchronize(stream) != hipSuccess) {
    delete p;
    return nullptr;
  }
  return p;
}

void destroy(Mpc32* p) { delete p; }

bool updateProblem(Mpc32* p, const qmgpu_problem& problem, hipStream_t stream) { return p && uploadProblem(p, problem, stream); }

bool enqueue(Mpc32* p, hipStream_t s, const qmgpu_mpc_args* a, double dtD, int iterations, int ddpTrials, hipEvent_t* ev) {
  if (!p || a->num_target_knots > kMaxKnots32) return false;
  const size_t B = size_t(a->batch), N = size_t(a->num_nodes), N1 = N + 1, K = size_t(a->num_target_knots);
  auto narrow = [&](const double* src, float* dst, size_t n) {
    if (!src) return static_cast<const float*>(nullptr);
    QM_LAUNCH(narrow_kernel, unsigned((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), 256, s, src, dst, n);
    return static_cast<const float*>(dst);
  };
  MpcIo io{};
  io.batch = a->batch; io.N = a->num_nodes; io.K = a->num_target_knots; io.lineSearch = a->line_search;
  io.dtD = dtD; io.t0D = a->t0; io.timeGridD = a->time_grid; io.schedTimesD = a->sched_event_times;
  io.x0 = narrow(a->x0, p->x0, B * 30);
  io.targetTimes = narrow(a->target_times, p->tgtT, B * K);
  io.targetStates = narrow(a->target_states, p->tgtS, B * K * QMGPU_NTARGET);
  io.schedNum = a->sched_num_events;
  io.schedTimes = narrow(a->sched_event_times, p->evT, B * QMGPU_MAX_EVENTS);   // the 1e300 padding becomes +inf: still "never"
  io.schedModes = a->sched_modes;
  io.warmX = narrow(a->warm_x, p->warmX, B * N1 * 30);
  io.warmU = narrow(a->warm_u, p->warmU, B * N * 30);
  io.eeContact = narrow(a->ee_contact_ref, p->contact, B * K * 6);
  io.outT = p->outT; io.outX = p->outX; io.outU = p->outU; io.outMode = a->out_mode; io.outStats = a->out_stats ? p->outStats : nullptr;
  io.algorithm = a->algorithm;
  if (a->algorithm == QMGPU_ALG_DDP) {
    ensureDdpBuffers(p->m, size_t(p->maxBatch), size_t(p->maxNodes), p->alloc);
    enqueueDdpKernels(s, p->m, io, ddpTrials, ev);
  } else {
    enqueueMpcKernels(s, p->m, io, iterations, false, ev);
  }
  auto widen = [&](const float* src, double* dst, size_t n) { QM_LAUNCH(widen_kernel, unsigned((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), 256, s, src, dst, n); };
  widen(p->outT, a->out_t, B * N1);
  widen(p->outX, a->out_x, B * N1 * 30);
  widen(p->outU, a->out_u, B * N * 30);
  if (a->out_stats) widen(p->outStats, a->out_stats, B * QMGPU_NSTATS);
  return hipGetLastError() == hipSuccess;
}

}  // namespace qmk
