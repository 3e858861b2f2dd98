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
    real* x = a.X + (size_t(inst) * (a.N + 1) + k) * 30;
    const real* src = (a.warmX && k > 0) ? a.warmX + (size_t(inst) * (a.N + 1) + k) * 30 : a.x0 + size_t(inst) * 30;
    for (int i = 0; i < 30; ++i) x[i] = src[i];
    if (k < a.N) {
      real* u = a.U + (size_t(inst) * a.N + k) * 30;
      if (a.warmU) {
        const real* su = a.warmU + (size_t(inst) * a.N + k) * 30;
        for (int i = 0; i < 30; ++i) u[i] = su[i];
      } else {
        const int mode = modes[phase];
        int n = 0;
        for (int c = 0; c < 4; ++c) n += contactOf(mode, c) ? 1 : 0;
        for (int i = 0; i < 30; ++i) u[i] = 0.0_r;
        if (n > 0) for (int c = 0; c < 4; ++c) if (contactOf(mode, c)) u[3 * c + 2] = a.P->model.total_mass * st.gravity / n;
      }
    }
  }
}

// Warm start of the next solve: the previous solution resampled on the new grid.  One workgroup per instance, one new node per wavefront
// pass (lanes 0..29 the state, 32..61 the input); the input trajectory has one entry less than the grid and holds its last value.
__global__ void __launch_bounds__(256) warm_start_kernel(int batch, int Np, const real* gridP, const real* Xp, const real* Up, int Nn, const real* gridN,
                                                         const real* x0, real* warmX, real* warmU) {
  const int inst = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (inst >= batch) return;
  const real* tg = gridP + size_t(inst) * (Np + 1);
  for (int k = wave; k <= Nn; k += nw) {
    const real t = gridN[size_t(inst) * (Nn + 1) + k];
    int idx; real alpha;
    timeSegmentWave(tg, Np + 1, t, lane, idx, alpha);
    if (lane < 30) {
      const real* xl = Xp + (size_t(inst) * (Np + 1) + idx) * 30;
      const real v = alpha * xl[lane] + (1.0_r - alpha) * xl[30 + lane];
      warmX[(size_t(inst) * (Nn + 1) + k) * 30 + lane] = (k == 0 && x0) ? x0[size_t(inst) * 30 + lane] : v;
    } else if (lane >= 32 && lane < 62 && k < Nn) {
      const int i = lane - 32, iu0 = min(idx, Np - 1), iu1 = min(idx + 1, Np - 1);
      const real* ul = Up + (size_t(inst) * Np + iu0) * 30; const real* ur = Up + (size_t(inst) * Np + iu1) * 30;
      warmU[(size_t(inst) * Nn + k) * 30 + i] = alpha * ul[i] + (1.0_r - alpha) * ur[i];
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
    int k = 0;
    while (k < N && tg[k + 1] < t) ++k;
    modeOut[inst] = modes[size_t(inst) * (N + 1) + k];
  }
}

}  // namespace qmk
