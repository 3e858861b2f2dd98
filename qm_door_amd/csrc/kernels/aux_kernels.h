// Small set-up kernels: input-weight mapping, initial trajectories, policy evaluation.
#pragma once
#include "layout.h"
#include "lq_kernel.h"

namespace qmk {

// R' = R_task with the 12x12 leg-velocity block replaced by J^T R_task J, J = feet Jacobian w.r.t. the leg joints at the
// initial state (QMInterface::initializeInputCostWeight, QMInterface.cpp:274-299).  One wavefront; the Jacobian columns come
// from the same structured sweep the LQ kernel uses (lane 3 + j carries d/d(v_joint j) in its velocity slot).
__global__ void __launch_bounds__(64) input_weight_kernel(const ProblemR* P, const real* zeros, real* Rw) {
  __shared__ real J[12 * 12];
  __shared__ real RJ[12 * 12];
  const int lane = threadIdx.x;
  const ModelR& md = P->model;
  const int dd = lane < AD_DIRS ? lane : AD_DIRS - 1;
  const AdIn in{P->settings.initial_state, zeros, P->settings.initial_state, dd, 0.0_r};
  FlowOut<Du, Du3, Du3> f;
  BaseMotion2<Du, Du3> bm;
  centroidalSweep2<Du, Du3, Du3>(
      md, P->settings.gravity, in,
      [&](int c, Vec3<Du>, Vec3<Du3> v) {   // foot velocity caused by the joint rates: its velocity slot is the Jacobian column
        if (lane >= 3 && lane < 15) { const int j = lane - 3; J[(3 * c + 0) * 12 + j] = v.x.e; J[(3 * c + 1) * 12 + j] = v.y.e; J[(3 * c + 2) * 12 + j] = v.z.e; }
      },
      [&](Vec3<Du>, const Mat3<Du>&) { return Vec3<Du3>(); }, f, bm);
  __syncthreads();
  const real* Rt = P->settings.R_task;
  for (int e = lane; e < 144; e += 64) {
    const int i = e / 12, j = e % 12;
    real s = 0.0_r;
    for (int k = 0; k < 12; ++k) s += Rt[(12 + i) * 30 + 12 + k] * J[k * 12 + j];
    RJ[e] = s;
  }
  __syncthreads();
  for (int e = lane; e < 900; e += 64) {
    const int i = e / 30, j = e % 30;
    real v = Rt[e];
    if (i >= 12 && i < 24 && j >= 12 && j < 24) {
      v = 0.0_r;
      for (int k = 0; k < 12; ++k) v += J[k * 12 + (i - 12)] * RJ[k * 12 + (j - 12)];
    }
    Rw[e] = v;
  }
  // barrier constants (layout.h: QM_RW_DERIVED): formed with the same value() every node evaluation used to call
  real* bc = Rw + QM_RW_DERIVED;
  const SettingsR& st = P->settings;
  if (lane == 0) { bc[QM_BC_LOGD_POS] = log(st.joint_pos_barrier_delta); bc[QM_BC_LOGD_VEL] = log(st.joint_vel_barrier_delta); bc[QM_BC_LOGD_FRIC] = log(st.friction_barrier_delta); bc[3] = 0.0_r; }
  if (lane < 6) {
    const Barrier bp{st.joint_pos_barrier_mu, st.joint_pos_barrier_delta}, bv{st.joint_vel_barrier_mu, st.joint_vel_barrier_delta};
    bc[QM_BC_POS0 + lane] = bp.value(-md.q_lower[12 + lane]) + bp.value(md.q_upper[12 + lane]);
    bc[QM_BC_VEL0 + lane] = bv.value(-st.arm_vel_lower[lane]) + bv.value(st.arm_vel_upper[lane]);
  }
}

struct InitArgs {
  const ProblemR* P;
  int batch, N;
  // Times are decisions, not arithmetic: the node times, the steps between them and the phase of every node are formed from the caller's
  // fp64 values in BOTH builds (an fp32 node time a few ulps to the wrong side of an event time would change the contact mode).
  double dtD; const double* t0D; const double* timeGridD; const double* schedTimesD;
  const real* x0; const real* warmX; const real* warmU;
  const int* schedNum; const int* schedModes;
  real* tgrid; real* dtgrid; int* nodePhase; real* X; real* U;
  int iteration;   // SQP iteration of this call (0 = first)
  int* done;       // [batch] convergence flags: cleared by iteration 0, set by linesearch_kernel; later iterations skip converged instances
};

// Time grid + initial trajectories: previous solution when given, else QMInitializer::compute (QMInitializer.cpp:33-41):
// u = weight-compensating input for the contact flags at t_k, x_{k+1} = x_k.  x[0] is always the measured state.
__global__ void mpc_init_kernel(InitArgs a) {
  const int inst = blockIdx.x;
  if (a.iteration == 0) { if (threadIdx.x == 0) a.done[inst] = 0; }
  else if (a.done[inst]) return;
  const SettingsR& st = a.P->settings;
  const int numEvents = a.schedNum[inst];
  const double* evD = a.schedTimesD + size_t(inst) * QMGPU_MAX_EVENTS;
  const int* modes = a.schedModes + size_t(inst) * (QMGPU_MAX_EVENTS + 1);
  auto timeOf = [&](int k) { return a.timeGridD ? a.timeGridD[size_t(inst) * (a.N + 1) + k] : a.t0D[inst] + k * a.dtD; };
  for (int k = threadIdx.x; k <= a.N; k += blockDim.x) {
    const double tD = timeOf(k);
    int phase = 0;   // nodePhaseAt (schedule_dev.h) on the fp64 times
    while (phase < numEvents && evD[phase] <= tD) ++phase;
    a.tgrid[size_t(inst) * (a.N + 1) + k] = real(tD);
    a.dtgrid[size_t(inst) * (a.N + 1) + k] = k < a.N ? real(timeOf(k + 1) - tD) : 0.0_r;
    a.nodePhase[size_t(inst) * (a.N + 1) + k] = phase;
    if (k < a.N && !a.warmU) {   // QMInitializer: the weight shared by the feet in contact at the node's time, nothing else -- stores only, nothing to wait for
      real* u = a.U + (size_t(inst) * a.N + k) * 30;
      const int mode = modes[phase];
      int n = 0;
      for (int c = 0; c < 4; ++c) n += contactOf(mode, c) ? 1 : 0;
      const real w = n > 0 ? a.P->model.total_mass * st.gravity / n : 0.0_r;
      for (int i = 0; i < 30; ++i) u[i] = (i < 12 && i % 3 == 2 && contactOf(mode, i / 3)) ? w : 0.0_r;
    }
  }
  // The copied trajectories (x0 / warm states, warm inputs) as flat copies, consecutive threads consecutive entries, eight entries per thread and pass with all
  // loads in front of the first store (a thread per node copied its 30 + 30 entries one by one, 240 bytes apart from its neighbour's, each load behind the
  // previous store).
  constexpr int UN = 8;
  const int nthr = blockDim.x, tid = threadIdx.x;
  const int nX = (a.N + 1) * 30, nU = a.N * 30;
  real* Xi = a.X + size_t(inst) * nX;
  const real* wX = a.warmX ? a.warmX + size_t(inst) * nX : nullptr;
  const real* x0 = a.x0 + size_t(inst) * 30;
  for (int base = tid; base < nX; base += UN * nthr) {
    real v[UN];
#pragma unroll
    for (int q = 0; q < UN; ++q) { const int e = base + q * nthr, ec = e < nX ? e : tid; v[q] = (wX && ec >= 30) ? wX[ec] : x0[ec % 30]; }
#pragma unroll
    for (int q = 0; q < UN; ++q) QM_KEEP(v[q]);
#pragma unroll
    for (int q = 0; q < UN; ++q) { const int e = base + q * nthr; if (e < nX) Xi[e] = v[q]; }
  }
  if (a.warmU) {
    real* Ui = a.U + size_t(inst) * nU;
    const real* wU = a.warmU + size_t(inst) * nU;
    for (int base = tid; base < nU; base += UN * nthr) {
      real v[UN];
#pragma unroll
      for (int q = 0; q < UN; ++q) { const int e = base + q * nthr; v[q] = wU[e < nU ? e : tid]; }
#pragma unroll
      for (int q = 0; q < UN; ++q) QM_KEEP(v[q]);
#pragma unroll
      for (int q = 0; q < UN; ++q) { const int e = base + q * nthr; if (e < nU) Ui[e] = v[q]; }
    }
  }
}

// Warm start of the next solve: the previous solution resampled on the new grid.  One workgroup per instance; a wavefront takes FOUR new nodes per pass (lanes
// 0..29 the state, 32..61 the input; the input trajectory has one entry less than the grid and holds its last value) so that the passes' memory round trips
// are shared: the four times, then the grid (each lane its entries, the interval of every time = the number of entries below it, from ballots: timeSegmentWave's
// rule without its early exit), then the interval ends, then the eight trajectory rows, then the stores.  One node per pass cost four dependent round trips per
// node, 25 nodes per wavefront: 35 us per launch.  Index and alpha as timeSegment (schedule_dev.h), bit for bit.
__global__ void __launch_bounds__(256) warm_start_kernel(int batch, int Np, const real* gridP, const real* Xp, const real* Up, int Nn, const real* gridN,
                                                         const real* x0, real* warmX, real* warmU) {
  const int inst = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (inst >= batch) return;
  constexpr int NB = 4;
  const real* tg = gridP + size_t(inst) * (Np + 1);
  const int K = Np + 1, last = K - 1;
  for (int k0 = wave * NB; k0 <= Nn; k0 += nw * NB) {
    real t[NB]; int lb[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) { t[q] = gridN[size_t(inst) * (Nn + 1) + min(k0 + q, Nn)]; lb[q] = 0; }
    for (int base = 0; base < K; base += 64) {
      const int i = base + lane;
      const real g = tg[i < K ? i : 0];
#pragma unroll
      for (int q = 0; q < NB; ++q) lb[q] += qmPopCount(qmBallot(i < K && g < t[q]));
    }
    real ta[NB], tb[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) { const int ic = max(0, min(lb[q] - 1, K - 2)); ta[q] = tg[ic]; tb[q] = tg[min(ic + 1, last)]; }
    int idx[NB]; real alpha[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int interval = lb[q] - 1;
      const real len = tb[q] - ta[q];
      const real a = (len > 2.0_r * REAL_EPS) ? (tb[q] - t[q]) / len : 1.0_r;
      const bool before = interval < 0 || K <= 1, after = !before && interval >= last;
      idx[q] = before ? 0 : (after ? max(last - 1, 0) : interval);
      alpha[q] = before ? 1.0_r : (after ? 0.0_r : a);
    }
    const bool isX = lane < 30, isU = lane >= 32 && lane < 62;
    const int c = isX ? lane : (isU ? lane - 32 : 0);
    real l[NB], r[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int iu0 = min(idx[q], Np - 1), iu1 = min(idx[q] + 1, Np - 1);
      const real* pl = isU ? Up + (size_t(inst) * Np + iu0) * 30 : Xp + (size_t(inst) * (Np + 1) + idx[q]) * 30;
      const real* pr = isU ? Up + (size_t(inst) * Np + iu1) * 30 : Xp + (size_t(inst) * (Np + 1) + min(idx[q] + 1, last)) * 30;
      l[q] = pl[c]; r[q] = pr[c];
    }
    const real x0v = (x0 && isX) ? x0[size_t(inst) * 30 + c] : 0.0_r;
#pragma unroll
    for (int q = 0; q < NB; ++q) { QM_KEEP(l[q]); QM_KEEP(r[q]); }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int k = k0 + q;
      const real v = alpha[q] * l[q] + (1.0_r - alpha[q]) * r[q];
      if (k <= Nn && isX) warmX[(size_t(inst) * (Nn + 1) + k) * 30 + c] = (k == 0 && x0) ? x0v : v;
      else if (k < Nn && isU) warmU[(size_t(inst) * Nn + k) * 30 + c] = v;
    }
  }
}

// MPC_MRT_Interface::evaluatePolicy (call site QMController.cpp:134-142): linear interpolation of (X, U) at t_eval, planned mode
// of the interval.  One wavefront per instance: lanes 0..29 interpolate the state, lanes 32..61 the input (every lane locates the segment).
__global__ void __launch_bounds__(64) policy_eval_kernel(int batch, int N, const real* tgrid, const real* X, const real* U, const int* modes, const real* tEval,
                                                         real* xOut, real* uOut, int* modeOut) {
  const int inst = blockIdx.x, lane = threadIdx.x;
  if (inst >= batch) return;
  const real* tg = tgrid + size_t(inst) * (N + 1);
  const real t = tEval[inst];
  int idx; real alpha;
  timeSegmentWave(tg, N + 1, t, lane, idx, alpha);
  const int kMode = gridCountBelow(tg + 1, N, t, lane);   // node interval containing t (lower_bound convention of the grid): the leading entries of t_1 .. t_N below t
  if (lane < 30) {
    const real* xl = X + (size_t(inst) * (N + 1) + idx) * 30;
    xOut[size_t(inst) * 30 + lane] = alpha * xl[lane] + (1.0_r - alpha) * xl[30 + lane];
  } else if (lane >= 32 && lane < 62) {
    // the input trajectory has N entries; upstream pads it by repeating the last input at the final time
    const int i = lane - 32, iu0 = min(idx, N - 1), iu1 = min(idx + 1, N - 1);
    const real* ul = U + (size_t(inst) * N + iu0) * 30; const real* ur = U + (size_t(inst) * N + iu1) * 30;
    uOut[size_t(inst) * 30 + i] = alpha * ul[i] + (1.0_r - alpha) * ur[i];
  } else if (lane == 63) {
    // mode of the node interval containing t (lower_bound convention of the grid)
    modeOut[inst] = modes[size_t(inst) * (N + 1) + kMode];
  }
}

}  // namespace qmk
