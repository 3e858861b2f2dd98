// The launch chain of one MPC solve and the HBM scratch it runs in, in the kernels' arithmetic type (real.h).  Shared by the fp64
// path (qmgpu_api.hip) and the fp32 path (qmgpu_mpc32.hip, where the caller's fp64 arrays are converted on the way in and out).
//     mpc_init -> ad_node (3 nodes per wavefront) -> lq_node (1 node per wavefront) -> riccati (1 instance per workgroup) -> linesearch
// repeated sqp.sqpIteration times on one stream, no host synchronisation in between.
#pragma once
#include <vector>

#include "aux_kernels.h"
#include "ddp_kernel.h"
#include "gpu_rt.h"
#include "layout.h"
#include "linesearch_kernel.h"
#include "lq_kernel.h"
#include "riccati_kernel.h"

namespace qmk {

struct MpcBuffers {
  ProblemR* dP = nullptr;
  real *dRw = nullptr, *dZeros = nullptr;
  real *dTgrid = nullptr, *dDtgrid = nullptr, *dX = nullptr, *dU = nullptr, *dStages = nullptr, *dAdRows = nullptr, *dMetrics = nullptr, *dGains = nullptr, *ddX = nullptr, *ddU = nullptr;
  real *dXt = nullptr, *dUt = nullptr, *dInstStats = nullptr, *dDebug = nullptr;
  int *dStageNc = nullptr, *dNodeMode = nullptr, *dNodePhase = nullptr, *dDone = nullptr;
  real *dDdpX = nullptr, *dDdpU = nullptr, *dDdpMerit = nullptr;   // DDP variant: trial trajectories / merits, allocated on first use
  int cus = 256;                                                   // compute units of the device the buffers live on (allocateMpcBuffers): launch shapes that depend on batch > CUs
};

// the arguments of one call (qmgpu_mpc_args) as `real` device arrays
struct MpcIo {
  int batch, N, K, lineSearch;
  double dtD;                                           // settings.dt in fp64
  const double *t0D, *timeGridD, *schedTimesD;          // the caller's fp64 times (mpc_init_kernel forms the grid and the node phases from them)
  const real *x0, *targetTimes, *targetStates;
  const int* schedNum; const real* schedTimes; const int* schedModes;
  const real *warmX, *warmU;
  real *outT, *outX, *outU; int* outMode; real* outStats;
  const real* eeContact;                                // [batch][K][6] or null (force tracking)
  int algorithm;                                        // QMGPU_ALG_SQP / QMGPU_ALG_DDP
};

// Alloc: callable (size_t count, size_t elemSize, bool scratch) -> void*
template <class Alloc> inline void allocateMpcBuffers(MpcBuffers& m, size_t B, size_t N, Alloc&& alloc) {
  const size_t N1 = N + 1;
  auto R = [&](size_t n, bool scratch = true) { return static_cast<real*>(alloc(n, sizeof(real), scratch)); };
  auto I = [&](size_t n) { return static_cast<int*>(alloc(n, sizeof(int), true)); };
  m.dP = static_cast<ProblemR*>(alloc(1, sizeof(ProblemR), false));
  m.dRw = R(QM_RW_DOUBLES, false);
  m.dZeros = R(64, false);
  m.dTgrid = R(B * N1);
  m.dDtgrid = R(B * N1);
  m.dNodePhase = I(B * N1);
  m.dX = R(B * N1 * 30);
  m.dU = R(B * N * 30);
  m.dStages = R(B * N1 * STAGE_DOUBLES);
  m.dAdRows = R(B * N1 * AD_DOUBLES);
  m.dMetrics = R(B * N1 * NODE_METRICS);
  m.dGains = R(B * N * GAIN_DOUBLES);
  m.ddX = R(B * N1 * 30);
  m.ddU = R(B * N * 30);
  m.dXt = R(2 * B * N1 * 30);   // two trial steps are evaluated side by side (linesearch_kernel)
  m.dUt = R(2 * B * N * 30);
  m.dInstStats = R(B * 4);
  m.dStageNc = I(B * N1);
  m.dNodeMode = I(B * N1);
  m.dDone = I(B);
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) m.cus = cus;
}

template <class Alloc> inline void ensureDdpBuffers(MpcBuffers& m, size_t B, size_t N, Alloc&& alloc) {
  if (m.dDdpX) return;
  m.dDdpX = static_cast<real*>(alloc(size_t(DDP_MAX_TRIALS) * B * (N + 1) * 30, sizeof(real), true));
  m.dDdpU = static_cast<real*>(alloc(size_t(DDP_MAX_TRIALS) * B * N * 30, sizeof(real), true));
  m.dDdpMerit = static_cast<real*>(alloc(size_t(DDP_MAX_TRIALS) * B * 2, sizeof(real), true));
}

inline hipError_t prepareMpcKernels() {
  const hipError_t e = QM_ALLOW_DYNAMIC_LDS(riccati_kernel<RICCATI_WAVES>, RICCATI_LDS_BYTES);
  return e != hipSuccess ? e : QM_ALLOW_DYNAMIC_LDS(linesearch_kernel, 160 * 1024 - LS_STATIC_LDS_BYTES);
}

// events (optional, 7 entries as in qmgpu_api.hip): [0] start, [6] after ad_node, [1] after lq_node, [2] after riccati, [3] after the line search
inline void enqueueMpcKernels(hipStream_t s, const MpcBuffers& m, const MpcIo& io, int iterations, bool debugLq, hipEvent_t* ev) {
  const int B = io.batch, N = io.N;
  if (ev) (void)hipEventRecord(ev[0], s);
  // sqp.sqpIteration iterations (task.info:77; 1 in the reference's configuration): later iterations warm-start from the iterate the
  // line search just wrote to the output buffers.  After every iteration the line-search kernel applies upstream's convergence
  // test per instance; the kernels of the following iterations return at once for the instances that have converged.
  for (int it = 0; it < iterations; ++it) {
    InitArgs ia{m.dP, B, N, io.dtD, io.t0D, io.timeGridD, io.schedTimesD, io.x0, it == 0 ? io.warmX : io.outX, it == 0 ? io.warmU : io.outU, io.schedNum, io.schedModes,
                m.dTgrid, m.dDtgrid, m.dNodePhase, m.dX, m.dU, it, m.dDone};
    QM_LAUNCH(mpc_init_kernel, B, 128, s, ia);
    LqArgs la{m.dP, m.dRw, B, N, io.K, m.dTgrid, m.dDtgrid, m.dNodePhase, m.dX, m.dU, io.targetTimes, io.targetStates, io.schedNum, io.schedTimes,
              io.schedModes, m.dZeros, m.dStages, m.dStageNc, m.dNodeMode, m.dMetrics, debugLq ? m.dDebug : nullptr, m.dAdRows, m.dDone, io.eeContact};
    QM_LAUNCH(ad_node_kernel, adGridFor(B * (N + 1)), 64, s, la);
    if (ev) (void)hipEventRecord(ev[6], s);
    QM_LAUNCH(lq_node_kernel, B * (N + 1), 64, s, la);
    if (ev) (void)hipEventRecord(ev[1], s);
    RiccatiArgs ra{B, N, m.dStages, m.dStageNc, m.dDtgrid, io.x0, m.dX, m.dGains, m.ddX, m.ddU, m.dInstStats, m.dDone};
    QM_LAUNCH_DYN(riccati_kernel<RICCATI_WAVES>, B, RICCATI_WAVES * 64, RICCATI_LDS_BYTES, s, ra);
    if (ev) (void)hipEventRecord(ev[2], s);
    LsArgs ls{m.dP, m.dRw, B, N, io.K, io.lineSearch, io.eeContact, m.dTgrid, m.dDtgrid, m.dNodePhase, m.dX, m.dU, m.ddX, m.ddU, io.targetTimes, io.targetStates, io.schedNum,
              io.schedTimes, io.schedModes, m.dMetrics, m.dInstStats, m.dNodeMode, m.dXt, m.dUt, io.outT, io.outX, io.outU, io.outMode, io.outStats, it, lsTrialLdsBytes(N, lsThreads(B, N, m.cus)) > 0, m.dDone};
    QM_LAUNCH_DYN(linesearch_kernel, B, lsThreads(B, N, m.cus), lsTrialLdsBytes(N, lsThreads(B, N, m.cus)), s, ls);
    if (ev) (void)hipEventRecord(ev[3], s);
  }
}

// One DDP iteration (ddp_kernel.h): rollout -> LQ approximation along it -> Riccati -> policy rollouts for every step length -> selection.
// `trials` = number of step lengths maxStep * 2^-i >= minStep (host side, <= DDP_MAX_TRIALS).
inline void enqueueDdpKernels(hipStream_t s, const MpcBuffers& m, const MpcIo& io, int trials, hipEvent_t* ev) {
  const int B = io.batch, N = io.N;
  if (ev) (void)hipEventRecord(ev[0], s);
  InitArgs ia{m.dP, B, N, io.dtD, io.t0D, io.timeGridD, io.schedTimesD, io.x0, io.warmX, io.warmU, io.schedNum, io.schedModes, m.dTgrid, m.dDtgrid, m.dNodePhase, m.dX, m.dU, 0, m.dDone};
  QM_LAUNCH(mpc_init_kernel, B, 128, s, ia);
  DdpArgs ra{m.dP, m.dRw, B, N, io.K, 0, io.eeContact, m.dTgrid, m.dDtgrid, m.dNodePhase, io.x0, m.dX, m.dU, io.targetTimes, io.targetStates, io.schedNum, io.schedTimes, io.schedModes,
             m.dStages, m.dStageNc, m.dGains, m.dX, m.dU, m.dDdpMerit};
  if (!io.warmX) QM_LAUNCH(ddp_rollout_kernel, (B + 63) / 64, 64, s, ra);   // no warm states: the nominal trajectory is the open-loop rollout of the inputs
  LqArgs la{m.dP, m.dRw, B, N, io.K, m.dTgrid, m.dDtgrid, m.dNodePhase, m.dX, m.dU, io.targetTimes, io.targetStates, io.schedNum, io.schedTimes,
            io.schedModes, m.dZeros, m.dStages, m.dStageNc, m.dNodeMode, m.dMetrics, nullptr, m.dAdRows, m.dDone, io.eeContact};
  QM_LAUNCH(ad_node_kernel, adGridFor(B * (N + 1)), 64, s, la);
  if (ev) (void)hipEventRecord(ev[6], s);
  QM_LAUNCH(lq_node_kernel, B * (N + 1), 64, s, la);
  if (ev) (void)hipEventRecord(ev[1], s);
  RiccatiArgs ri{B, N, m.dStages, m.dStageNc, m.dDtgrid, io.x0, m.dX, m.dGains, m.ddX, m.ddU, m.dInstStats, m.dDone};
  QM_LAUNCH_DYN(riccati_kernel<RICCATI_WAVES>, B, RICCATI_WAVES * 64, RICCATI_LDS_BYTES, s, ri);
  if (ev) (void)hipEventRecord(ev[2], s);
  ra.trials = trials; ra.Xout = m.dDdpX; ra.Uout = m.dDdpU;
  QM_LAUNCH(ddp_rollout_kernel, (B * trials + 63) / 64, 64, s, ra);
  DdpSelectArgs sa{m.dP, B, N, trials, m.dTgrid, m.dNodeMode, m.dX, m.dU, m.dMetrics, m.dInstStats, m.dDdpX, m.dDdpU, m.dDdpMerit, io.outT, io.outX, io.outU, io.outMode, io.outStats, m.dDone};
  QM_LAUNCH(ddp_select_kernel, B, 256, s, sa);
  if (ev) (void)hipEventRecord(ev[3], s);
}

inline int ddpTrialCount(double minStep, double maxStep) {
  int n = 0;
  for (double a = maxStep; a >= minStep && n < DDP_MAX_TRIALS; a *= 0.5) ++n;
  return n > 0 ? n : 1;
}

}  // namespace qmk
