// DDP variant of the MPC solve (qmgpu_mpc_args::algorithm = QMGPU_ALG_DDP; SURVEY.md section 8(f) rank 3: the ddp{} block of the task file,
// qm_controllers/config/task.info:34-72, parsed at qm_interface/src/QMInterface.cpp:70 and never instantiated by the reference).
//
//   nominal trajectory           the warm start (warm_x, warm_u) as it is -- its defects are part of the linearisation, as in Gauss-Newton multiple
//                                shooting -- or, without warm_x, ddp_rollout_kernel (init): one lane per instance, x_{k+1} = RK2(x_k, u_k) from x0 with
//                                the warm-start / initializer inputs (an open-loop rollout over a long horizon drifts: the centroidal model is unstable)
//   ad_node -> lq_node -> riccati   the LQ approximation along that trajectory and the projected Riccati recursion, unchanged
//   ddp_rollout_kernel (trials)  one lane per (instance, step length): the feedback policy
//                                    u_k = u_nom,k + alpha (Pe_k + Pu_k k_k) + (Px_k + Pu_k K_k) (x_k - x_nom,k)
//                                rolled out through the nonlinear dynamics, merit = sum of dt-scaled costs + penalty * dt |eq|^2
//   ddp_select_kernel            first step length alpha = maxStep * 2^-i >= minStep whose merit passes the Armijo test; outputs, statistics
//
// Stated deviations from upstream's SLQ: RK2 steps of the shooting grid instead of the ODE45 rollout (task.info:129-137), discrete-time backward
// pass (upstream's ILQR form) instead of the continuous-time Riccati ODE.  A rollout is a scalar dependent chain over the horizon, so the lanes of
// a wavefront carry different (instance, step length) pairs rather than nodes.
#pragma once
#include "layout.h"
#include "linesearch_kernel.h"

namespace qmk {

constexpr int DDP_MAX_TRIALS = 8;

struct DdpArgs {
  const ProblemR* P;
  const real* Rw;
  int batch, N, K;
  int trials;                 // 0: initial open-loop rollout (one lane per instance); > 0: policy rollouts, lane = instance * trials + trial
  const real* eeContact;
  const real* tgrid; const real* dtgrid; const int* nodePhase;
  const real* x0;             // [batch][30]
  const real* X; const real* U;   // nominal trajectory (policy rollouts) / inputs of the initial rollout
  const real* targetTimes; const real* targetStates;
  const int* schedNum; const real* schedTimes; const int* schedModes;
  const real* stages; const int* stageNc; const real* gains;
  real* Xout; real* Uout;     // init: [batch] trajectories (the nominal X); trials: [batch * trials] trajectories
  real* merit;                // [batch * trials][2]: merit, dt |eq|^2
};

__device__ __forceinline__ real ddpStepLength(const SettingsR& st, int trial) {
  real a = st.ddp_max_step;
  for (int i = 0; i < trial; ++i) a *= 0.5_r;
  return a;
}

// as linesearch_kernel: only declared in the product build of qmgpu_api.hip, defined in qmgpu_ls.hip (interprocedural register allocation on: the called node
// evaluation no longer saves its callee-saved registers through scratch)
#if defined(QM_LS_EXTERN) && !defined(QM_RICCATI_TIMING)
__global__ void __launch_bounds__(64) ddp_rollout_kernel(DdpArgs a);
#else
__global__ void __launch_bounds__(64) ddp_rollout_kernel(DdpArgs a) {
  const int id = blockIdx.x * 64 + threadIdx.x;
  const bool init = a.trials == 0;
  const int per = init ? 1 : a.trials;
  if (id >= a.batch * per) return;
  const int inst = id / per, trial = id - inst * per;
  const SettingsR& st = a.P->settings;
  const int N = a.N;
  const real alpha = init ? 0.0_r : ddpStepLength(st, trial);
  const real* tg = a.tgrid + size_t(inst) * (N + 1);
  const Schedule sched{a.schedNum[inst], a.schedTimes + size_t(inst) * QMGPU_MAX_EVENTS, a.schedModes + size_t(inst) * (QMGPU_MAX_EVENTS + 1)};
  const real* tTimes = a.targetTimes + size_t(inst) * a.K;
  const real* tStates = a.targetStates + size_t(inst) * a.K * QMGPU_NTARGET;
  const real* contact = a.eeContact ? a.eeContact + size_t(inst) * a.K * 6 : nullptr;
  real* Xo = a.Xout + size_t(id) * (N + 1) * 30;
  real* Uo = a.Uout + size_t(id) * N * 30;
  // per-lane state / input / next state live in LDS (a private array indexed in rolled loops would sit in scratch memory); the block stride of
  // 129 reals keeps the lanes of a wavefront on different banks
  __shared__ real work[64 * 129];
  real* x = work + threadIdx.x * 129; real* u = x + 32; real* xn = x + 64; real* dx = x + 96;
  real dut[MT];
  for (int i = 0; i < 30; ++i) x[i] = a.x0[size_t(inst) * 30 + i];
  real merit = 0.0_r, eqSum = 0.0_r;
  const bool armed = !init && alpha < st.ddp_min_step;   // step lengths below the minimum are not evaluated (the selection never takes them)
  if (armed) { a.merit[size_t(id) * 2] = 1e30_r; a.merit[size_t(id) * 2 + 1] = 0.0_r; return; }
#pragma unroll 1
  for (int k = 0; k < N; ++k) {
    const size_t node = size_t(inst) * (N + 1) + k;
    const real* un = a.U + (size_t(inst) * N + k) * 30;
    if (init) {
      for (int i = 0; i < 30; ++i) u[i] = un[i];
    } else {
      const real* xnom = a.X + node * 30;
      const real* rec = a.stages + node * STAGE_DOUBLES;
      const real* gn = a.gains + (size_t(inst) * N + k) * GAIN_DOUBLES;
      const int nt = 30 - a.stageNc[node];
      for (int i = 0; i < 30; ++i) dx[i] = x[i] - xnom[i];
      for (int r = 0; r < MT; ++r) {
        real s = 0.0_r;
        if (r < nt) { s = alpha * gn[OFF_kff + r]; for (int c = 0; c < 30; ++c) s += gn[OFF_KFB + r * 30 + c] * dx[c]; }
        dut[r] = s;
      }
      for (int i = 0; i < 30; ++i) {
        real s = alpha * rec[OFF_PE + i];
        if (i >= 12) for (int c = 0; c < 30; ++c) s += rec[offPxRow(i) + c] * dx[c];   // rows 0..11 (force inputs) of Px are structurally zero and not stored (layout.h)
        if (i >= 12) { for (int r = 0; r < nt; ++r) s += rec[offPuRow(i) + r] * dut[r]; }
        else { const int pc = puColumnOfForce(int(rec[OFF_MODE]), i); if (pc >= 0) s += dut[pc]; }   // force rows of Pu: unit vectors, not stored (layout.h)
        u[i] = un[i] + s;
      }
    }
    for (int i = 0; i < 30; ++i) { Xo[k * 30 + i] = x[i]; Uo[k * 30 + i] = u[i]; }
    real c, d, e;
    nodePerformance<const real*>(a.P->model, a.P->settings, a.P->settings.Q, a.Rw, 0, a.Rw + QM_RW_DERIVED, sched, tTimes, tStates, contact, a.K, tg[k], a.dtgrid[node], a.nodePhase[node], false, x, u, nullptr, c, d, e, xn);
    merit += c; eqSum += e;
    for (int i = 0; i < 30; ++i) x[i] = xn[i];
  }
  for (int i = 0; i < 30; ++i) Xo[N * 30 + i] = x[i];
  if (!init) {
    real c, d, e;
    nodePerformance<const real*>(a.P->model, a.P->settings, a.P->settings.Q, a.Rw, 0, a.Rw + QM_RW_DERIVED, sched, tTimes, tStates, contact, a.K, tg[N], 0.0_r, a.nodePhase[size_t(inst) * (N + 1) + N], true, x, u, nullptr, c, d, e);
    merit += c;
    a.merit[size_t(id) * 2] = merit + st.ddp_constraint_penalty * eqSum;
    a.merit[size_t(id) * 2 + 1] = eqSum;
  }
}
#endif   // QM_LS_EXTERN

struct DdpSelectArgs {
  const ProblemR* P;
  int batch, N, trials;
  const real* tgrid; const int* nodeMode;
  const real* X; const real* U;          // nominal trajectory
  const real* metrics;                   // node metrics of the nominal trajectory (lq_node_kernel): dt cost, dt |defect|^2, dt |eq|^2
  const real* instStats;                 // armijo descent, Riccati status
  const real* Xt; const real* Ut; const real* merit;
  real* outT; real* outX; real* outU; int* outMode; real* outStats;
  int* done;
};

#ifndef QM_LS_UNIT   // (qmgpu_ls.hip includes this file for ddp_rollout_kernel only)
__global__ void __launch_bounds__(256) ddp_select_kernel(DdpSelectArgs a) {
  __shared__ real red[2 * 256];
  __shared__ int pick;
  const int inst = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x, N = a.N;
  const SettingsR& st = a.P->settings;
  real m0 = 0.0_r, e0 = 0.0_r;   // the nominal trajectory may carry defects (a warm start that is not a rollout): they count like the equalities
  for (int k = tid; k <= N; k += nthr) { const real* m = a.metrics + (size_t(inst) * (N + 1) + k) * NODE_METRICS; m0 += m[0]; e0 += m[2] + m[1]; }
  red[tid] = m0; red[256 + tid] = e0;
  __syncthreads();
  if (tid == 0) {
    real s0 = 0.0_r, s1 = 0.0_r;
    for (int i = 0; i < nthr; ++i) { s0 += red[i]; s1 += red[256 + i]; }
    const real merit0 = s0 + st.ddp_constraint_penalty * s1;
    const real armijo = a.instStats[size_t(inst) * 4 + 0], ricStatus = a.instStats[size_t(inst) * 4 + 1];
    int chosen = -1, evaluated = 0;
    if (ricStatus == 0.0_r) {
      for (int t = 0; t < a.trials && chosen < 0; ++t) {
        const real al = ddpStepLength(st, t);
        if (al < st.ddp_min_step) break;
        ++evaluated;
        const real m1 = a.merit[(size_t(inst) * a.trials + t) * 2];
        if (m1 == m1 && m1 <= merit0 - st.armijo_factor * al * fabs(armijo)) chosen = t;   // NaN-safe; upstream's Armijo descent test on the merit
      }
    }
    pick = chosen;
    if (a.outStats) {
      real* s = a.outStats + size_t(inst) * QMGPU_NSTATS;
      s[0] = merit0; s[1] = sqrt(s1);
      s[2] = chosen >= 0 ? a.merit[(size_t(inst) * a.trials + chosen) * 2] : merit0;
      s[3] = chosen >= 0 ? sqrt(a.merit[(size_t(inst) * a.trials + chosen) * 2 + 1]) : sqrt(s1);
      s[4] = chosen >= 0 ? ddpStepLength(st, chosen) : 0.0_r; s[5] = real(evaluated); s[6] = armijo; s[7] = ricStatus; s[8] = 1.0_r; s[9] = 1.0_r;
    }
    a.done[inst] = 1;
  }
  __syncthreads();
  const int chosen = pick;
  const real* Xs = chosen >= 0 ? a.Xt + (size_t(inst) * a.trials + chosen) * (N + 1) * 30 : a.X + size_t(inst) * (N + 1) * 30;
  const real* Us = chosen >= 0 ? a.Ut + (size_t(inst) * a.trials + chosen) * N * 30 : a.U + size_t(inst) * N * 30;
  real* oX = a.outX + size_t(inst) * (N + 1) * 30; real* oU = a.outU + size_t(inst) * N * 30;
  for (int e = tid; e < (N + 1) * 30; e += nthr) oX[e] = Xs[e];
  for (int e = tid; e < N * 30; e += nthr) oU[e] = Us[e];
  for (int k = tid; k <= N; k += nthr) { a.outT[size_t(inst) * (N + 1) + k] = a.tgrid[size_t(inst) * (N + 1) + k]; a.outMode[size_t(inst) * (N + 1) + k] = a.nodeMode[size_t(inst) * (N + 1) + k]; }
}
#endif

}  // namespace qmk
