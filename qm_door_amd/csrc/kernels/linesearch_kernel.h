// linesearch_kernel -- filter line search + trajectory update.  One workgroup per MPC instance, ONE SHOOTING NODE PER LANE
// (value-only evaluation, T = double instantiation of the same tree sweep the LQ kernel differentiates).
//
// Replaces upstream ocs2_sqp::SqpSolver::takeStep / computePerformance and FilterLinesearch::acceptStep (settings
// g_max / g_min: task.info:82-83; alpha_decay 0.5, alpha_min 1e-4, gamma_c 1e-6, armijo 1e-4 are the upstream defaults),
// evaluating the node terms assembled in qm_interface/src/QMInterface.cpp:99-131.
#pragma once
#include "layout.h"
#include "schedule_dev.h"
#include "sweep_dev.h"

namespace qmk {

struct LsArgs {
  const qmgpu_problem* P;
  const double* Rw;
  int batch, N, K, lineSearch;
  const double* tgrid; const double* X; const double* U; const double* dX; const double* dU;
  const double* targetTimes; const double* targetStates;
  const int* schedNum; const double* schedTimes; const int* schedModes;
  const double* metrics;    // baseline node metrics from lq_node_kernel
  const double* instStats;  // armijo, riccati status
  const int* nodeMode;
  double* Xt; double* Ut;   // trial trajectories (scratch) [batch][N+1][30], [batch][N][30]
  double* outT; double* outX; double* outU; int* outMode; double* outStats;
  int iteration;   // SQP iteration of this call
  int* done;       // [batch] convergence flags (see InitArgs)
};

struct DblIn {
  const double* x; const double* u; double dtS; const double* k1;
  __device__ __forceinline__ double hn(int i) const { return x[i] + dtS * k1[i]; }
  __device__ __forceinline__ double euler(int i) const { return x[9 + i] + dtS * k1[9 + i]; }
  __device__ __forceinline__ double q(int j) const { return x[12 + j] + dtS * u[12 + j]; }
  __device__ __forceinline__ double qd(int j) const { return u[12 + j]; }
  __device__ __forceinline__ Vec3<double> force(int c) const { return Vec3<double>(u[3 * c], u[3 * c + 1], u[3 * c + 2]); }
};

// dt-scaled cost, dt*|defect|^2, dt*|eq|^2 of one node at (x, u, xnext)
__device__ inline void nodePerformance(const qmgpu_problem& P, const double* Rw, const Schedule& sched, const double* tTimes, const double* tStates, int K, double t, double dt,
                                       bool terminal, const double* x, const double* u, const double* xnext, double& cost, double& dyn, double& eq) {
  const qmgpu_model& md = P.model;
  const qmgpu_settings& st = P.settings;
  const int phase = nodePhaseAt(sched, t);
  const int mode = sched.modes[phase];
  double eePosRef[3], eeQuatRef[4];
  eeReference(tTimes, tStates, K, t, eePosRef, eeQuatRef);
  double k1[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) k1[i] = 0.0;
  double c = 0.0;
  dyn = 0.0; eq = 0.0;
  double phi[12];
#pragma unroll 1
  for (int stage = 0; stage < (terminal ? 1 : 2); ++stage) {
    const DblIn in{x, u, stage ? dt : 0.0, k1};
    Feet<double> feet;
    double f[12];
    BaseMotion<double> bm;
    centroidalSweep<double>(
        md, st.gravity, in, [&](int cc, Vec3<double> r, Vec3<double> v) { feet.set(cc, r, v); },
        [&](Vec3<double> r, const Mat3<double>& R) {
          if (stage == 0) {
            double qee[4];
            matrixToQuaternion(R, qee);
            const Vec3<double> od = quaternionDistance(qee, eeQuatRef);
            const double muP = terminal ? st.ee_final_mu_position : st.ee_mu_position, muO = terminal ? st.ee_final_mu_orientation : st.ee_mu_orientation;
            const double hx = x[6] + r.x - eePosRef[0], hy = x[7] + r.y - eePosRef[1], hz = x[8] + r.z - eePosRef[2];
            c += 0.5 * muP * (hx * hx + hy * hy + hz * hz) + 0.5 * muO * (od.x * od.x + od.y * od.y + od.z * od.z);
          }
        },
        f, bm);
    if (stage == 0) {
      if (!terminal) {
        for (int cc = 0; cc < 4; ++cc) {
          const Vec3<double> r = feet.r(cc);
          const Vec3<double> vf = bm.dp + cross(bm.omega, r) + feet.v(cc);
          if (contactOf(mode, cc)) { const double hz = vf.z + st.position_error_gain * (x[8] + r.z); eq += vf.x * vf.x + vf.y * vf.y + hz * hz; }
          else {
            double zp, zv;
            swingReference(st, sched, cc, t, phase, zp, zv);
            const double h = vf.z - zv + st.position_error_gain * (x[8] + r.z - zp);
            eq += u[3 * cc] * u[3 * cc] + u[3 * cc + 1] * u[3 * cc + 1] + u[3 * cc + 2] * u[3 * cc + 2] + h * h;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) { k1[i] = f[i]; phi[i] = 0.5 * dt * f[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 12; ++i) phi[i] += 0.5 * dt * f[i];
    }
  }
  if (terminal) { cost = c; return; }
  for (int i = 0; i < 12; ++i) { const double d = x[i] + phi[i] - xnext[i]; dyn += d * d; }
  for (int j = 0; j < 18; ++j) { const double d = x[12 + j] + dt * u[12 + j] - xnext[12 + j]; dyn += d * d; }
  // tracking cost
  int tIdx; double tAlpha;
  timeSegment(tTimes, K, t, tIdx, tAlpha);
  int nStance = 0;
  for (int k = 0; k < 4; ++k) nStance += contactOf(mode, k) ? 1 : 0;
  const double fzNom = nStance > 0 ? md.total_mass * st.gravity / nStance : 0.0;
  for (int i = 0; i < 30; ++i) {
    const double dxi = x[i] - xReference(tStates, K, tIdx, tAlpha, i);
    const double dui = u[i] - ((i < 12 && (i % 3) == 2 && contactOf(mode, i / 3)) ? fzNom : 0.0);
    double qs = 0.0, rs = 0.0;
    for (int j = 0; j < 30; ++j) {
      const double dxj = x[j] - xReference(tStates, K, tIdx, tAlpha, j);
      const double duj = u[j] - ((j < 12 && (j % 3) == 2 && contactOf(mode, j / 3)) ? fzNom : 0.0);
      qs += st.Q[i * 30 + j] * dxj; rs += Rw[i * 30 + j] * duj;
    }
    c += 0.5 * dxi * qs + 0.5 * dui * rs;
  }
  const Barrier bp{st.joint_pos_barrier_mu, st.joint_pos_barrier_delta}, bv{st.joint_vel_barrier_mu, st.joint_vel_barrier_delta}, bf{st.friction_barrier_mu, st.friction_barrier_delta};
  for (int i = 0; i < 6; ++i) {
    const double lo = md.q_lower[12 + i], up = md.q_upper[12 + i];
    c += bp.value(x[24 + i] - lo) + bp.value(up - x[24 + i]) - (bp.value(-lo) + bp.value(up));
    c += bv.value(u[24 + i] - st.arm_vel_lower[i]) + bv.value(st.arm_vel_upper[i] - u[24 + i]) - (bv.value(-st.arm_vel_lower[i]) + bv.value(st.arm_vel_upper[i]));
  }
  for (int cc = 0; cc < 4; ++cc) if (contactOf(mode, cc)) {
    const double fx = u[3 * cc], fy = u[3 * cc + 1], fz = u[3 * cc + 2];
    c += bf.value(st.friction_coefficient * fz - sqrt(fx * fx + fy * fy + st.friction_regularization));
  }
  cost = dt * c; dyn *= dt; eq *= dt;
}

__global__ void __launch_bounds__(256) linesearch_kernel(LsArgs a) {
  __shared__ double red[3 * 256];
  __shared__ double ctl[8];
  const int inst = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  if (a.done[inst]) return;   // converged in an earlier iteration of this call: outputs and statistics stay as they are
  const int N = a.N;
  const qmgpu_settings& st = a.P->settings;
  const double* tg = a.tgrid + size_t(inst) * (N + 1);
  const Schedule sched{a.schedNum[inst], a.schedTimes + size_t(inst) * QMGPU_MAX_EVENTS, a.schedModes + size_t(inst) * (QMGPU_MAX_EVENTS + 1)};
  const double* tTimes = a.targetTimes + size_t(inst) * a.K;
  const double* tStates = a.targetStates + size_t(inst) * a.K * QMGPU_NTARGET;
  const double* X = a.X + size_t(inst) * (N + 1) * 30; const double* U = a.U + size_t(inst) * N * 30;
  const double* dX = a.dX + size_t(inst) * (N + 1) * 30; const double* dU = a.dU + size_t(inst) * N * 30;
  // Two consecutive trial steps (alpha, alpha * decay) are evaluated side by side when the horizon fits half the workgroup: the
  // second half of the threads would otherwise idle, and a launch is as slow as its slowest instance -- one instance of the batch
  // that rejects the full step no longer doubles the kernel time.  Acceptance is tested in the sequential order, so the result is
  // the one of FilterLinesearch's loop.
  const int half = (N + 1 <= nthr / 2) ? nthr / 2 : nthr;
  const int nTr = nthr / half, myTr = tid / half, ltid = tid - myTr * half;
  double* Xt = a.Xt + (size_t(inst) * 2 + myTr) * (N + 1) * 30; double* Ut = a.Ut + (size_t(inst) * 2 + myTr) * N * 30;

  // baseline performance (sum of the LQ kernel's node metrics)
  double m0 = 0.0, d0 = 0.0, e0 = 0.0;
  for (int k = tid; k <= N; k += nthr) { const double* m = a.metrics + (size_t(inst) * (N + 1) + k) * NODE_METRICS; m0 += m[0]; d0 += m[1]; e0 += m[2]; }
  red[tid] = m0; red[256 + tid] = d0; red[512 + tid] = e0;
  __syncthreads();
  if (tid == 0) {
    double s0 = 0, s1 = 0, s2 = 0;
    for (int i = 0; i < nthr; ++i) { s0 += red[i]; s1 += red[256 + i]; s2 += red[512 + i]; }
    ctl[0] = s0; ctl[1] = sqrt(s1 + s2);
  }
  __syncthreads();
  const double merit0 = ctl[0], viol0 = ctl[1];
  const double armijo = a.instStats[size_t(inst) * 4 + 0];
  const double ricStatus = a.instStats[size_t(inst) * 4 + 1];

  double alpha = 1.0, merit1 = merit0, viol1 = viol0;
  int stepType = 0;
  bool accepted = false;
#pragma unroll 1
  for (int trial = 0; trial < 64; ++trial) {
    const double alphaMine = myTr ? alpha * st.alpha_decay : alpha;
    for (int e = ltid; e < (N + 1) * 30; e += half) Xt[e] = X[e] + alphaMine * dX[e];
    for (int e = ltid; e < N * 30; e += half) Ut[e] = U[e] + alphaMine * dU[e];
    __syncthreads();
    double cs = 0.0, ds = 0.0, es = 0.0;
    for (int k = ltid; k <= N; k += half) {
      double c, d, e;
      const bool term = k == N;
      nodePerformance(*a.P, a.Rw, sched, tTimes, tStates, a.K, tg[k], term ? 0.0 : tg[k + 1] - tg[k], term, Xt + k * 30, term ? Ut : Ut + k * 30, term ? Xt + k * 30 : Xt + (k + 1) * 30, c, d, e);
      cs += c; ds += d; es += e;
    }
    red[tid] = cs; red[256 + tid] = ds; red[512 + tid] = es;
    __syncthreads();
    if (tid == 0) {
      double acc = 0.0, accAlpha = alpha, m1 = merit0, v1 = viol0; int type = 0;
      for (int tr = 0; tr < nTr && acc == 0.0; ++tr) {
        const double al = tr ? alpha * st.alpha_decay : alpha;
        if (tr && al < st.alpha_min) break;        // the sequential loop would have stopped before this trial
        double s0 = 0, s1 = 0, s2 = 0;
        for (int i = tr * half; i < (tr + 1) * half; ++i) { s0 += red[i]; s1 += red[256 + i]; s2 += red[512 + i]; }
        m1 = s0; v1 = sqrt(s1 + s2);
        bool ok;
        // upstream FilterLinesearch::acceptStep
        if (!a.lineSearch) { ok = true; type = 0; }
        else if (v1 > st.g_max) { ok = v1 < (1.0 - st.gamma_c) * viol0; type = 1; }
        else if (v1 < st.g_min && viol0 < st.g_min && al * armijo < 0.0) { ok = m1 < merit0 + st.armijo_factor * al * armijo; type = 3; }
        else { ok = m1 < merit0 - st.gamma_c * viol0 || v1 < (1.0 - st.gamma_c) * viol0; type = 2; }
        accAlpha = al;
        if (ok) acc = 1.0;
      }
      ctl[2] = m1; ctl[3] = v1; ctl[4] = acc; ctl[5] = double(type); ctl[6] = accAlpha;
    }
    __syncthreads();
    merit1 = ctl[2]; viol1 = ctl[3]; stepType = int(ctl[5]);
    accepted = ctl[4] != 0.0;
    const double lastAlpha = ctl[6];
    __syncthreads();
    if (accepted) { alpha = lastAlpha; break; }
    alpha = lastAlpha * st.alpha_decay;
    if (alpha < st.alpha_min) break;
  }
  if (!accepted) { alpha = 0.0; stepType = 4; merit1 = merit0; viol1 = viol0; }
  // ---- write the new iterate
  double* oX = a.outX + size_t(inst) * (N + 1) * 30; double* oU = a.outU + size_t(inst) * N * 30;
  for (int e = tid; e < (N + 1) * 30; e += nthr) oX[e] = X[e] + alpha * dX[e];
  for (int e = tid; e < N * 30; e += nthr) oU[e] = U[e] + alpha * dU[e];
  for (int k = tid; k <= N; k += nthr) { a.outT[size_t(inst) * (N + 1) + k] = tg[k]; a.outMode[size_t(inst) * (N + 1) + k] = a.nodeMode[size_t(inst) * (N + 1) + k]; }
  // ---- upstream SqpSolver::checkConvergence: iteration limit, step size, metrics, primal step (l2 norms over the whole horizon)
  double sx = 0.0, su = 0.0;
  for (int e = tid; e < (N + 1) * 30; e += nthr) sx += dX[e] * dX[e];
  for (int e = tid; e < N * 30; e += nthr) su += dU[e] * dU[e];
  __syncthreads();
  red[tid] = sx; red[256 + tid] = su;
  __syncthreads();
  if (tid == 0) {
    double s0 = 0, s1 = 0;
    for (int i = 0; i < nthr; ++i) { s0 += red[i]; s1 += red[256 + i]; }
    int conv = 0;
    if (a.iteration + 1 >= st.sqp_iterations) conv = 1;
    else if (alpha < st.alpha_min) conv = 2;
    else if (fabs(merit1 - merit0) < st.cost_tol && viol1 < st.g_min) conv = 3;
    else if (alpha * sqrt(s0) < st.delta_tol && alpha * sqrt(s1) < st.delta_tol) conv = 4;
    if (ricStatus != 0.0 && conv == 0) conv = 2;   // a failed factorisation leaves the iterate where it was: nothing more to do
    a.done[inst] = conv;
    if (a.outStats) {
      double* s = a.outStats + size_t(inst) * QMGPU_NSTATS;
      s[0] = merit0; s[1] = viol0; s[2] = merit1; s[3] = viol1; s[4] = alpha; s[5] = double(stepType); s[6] = armijo; s[7] = ricStatus;
      s[8] = double(a.iteration + 1); s[9] = double(conv);
    }
  }
}

}  // namespace qmk
