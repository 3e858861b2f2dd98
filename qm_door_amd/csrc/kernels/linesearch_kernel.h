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
  const ProblemR* P;
  const real* Rw;
  int batch, N, K, lineSearch;
  const real* eeContact;   // [batch][K][6] or null (force tracking)
  const real* tgrid; const real* dtgrid; const int* nodePhase;   // as in LqArgs
  const real* X; const real* U; const real* dX; const real* dU;
  const real* targetTimes; const real* targetStates;
  const int* schedNum; const real* schedTimes; const int* schedModes;
  const real* metrics;    // baseline node metrics from lq_node_kernel
  const real* instStats;  // armijo, riccati status
  const int* nodeMode;
  real* Xt; real* Ut;   // trial trajectories (scratch) [batch][N+1][30], [batch][N][30]
  real* outT; real* outX; real* outU; int* outMode; real* outStats;
  int iteration;   // SQP iteration of this call
  int trialInLds;  // the trial trajectories fit the dynamic LDS of the launch (lsTrialLdsBytes): they never touch HBM
  int* done;       // [batch] convergence flags (see InitArgs)
};

// dynamic LDS of a line-search launch that keeps the trial trajectories on chip: 0 if they do not fit beside the static arrays
constexpr int LS_STATIC_LDS_BYTES = (3 * 256 + 8 + 1800) * int(sizeof(real)) + int(sizeof(ModelR)) + 256 * 4 + 256;
inline int lsTrialLdsBytes(int N, int threads = 256) {
  const int trials = (N + 1 <= threads / 2) ? 2 : 1;
  const long long need = (long long)trials * (2 * N + 1) * 30 * (long long)sizeof(real);
  return need + LS_STATIC_LDS_BYTES <= 160 * 1024 ? int(need) : 0;
}
// Threads per workgroup of a line-search launch.  256 = two trial steps side by side (the kernel comment): right while every instance has a CU of its own, where
// the launch lasts as long as its slowest instance.  With more instances than CUs the launch is a queue, throughput counts, and the speculative second trial (wasted
// in every instance that accepts the full step: all of them in the bench sets) is better spent on a second INSTANCE: 128 threads evaluate one trial at a time with
// the same node-to-thread assignment and the same order of summation (64 < N + 1 <= 128: half = 128 either way; N + 1 <= 64: one node per thread either way --
// merit, violation, alpha, step type and the iterate are bit-identical),
// one trial's trajectories in LDS (48 KB at N = 100 instead of 96), so two workgroups share a CU.
inline int lsThreads(int B, int N, int cus) { return (B > cus && N + 1 <= 128) ? 128 : 256; }

struct DblIn {
  const real* x; const real* u; real dtS; const real* k1;
  __device__ __forceinline__ real hn(int i) const { return x[i] + dtS * k1[i]; }
  __device__ __forceinline__ real euler(int i) const { return x[9 + i] + dtS * k1[9 + i]; }
  __device__ __forceinline__ real q(int j) const { return x[12 + j] + dtS * u[12 + j]; }
  __device__ __forceinline__ real qd(int j) const { return u[12 + j]; }
  __device__ __forceinline__ Vec3<real> force(int c) const { return Vec3<real>(u[3 * c], u[3 * c + 1], u[3 * c + 2]); }
};

// dt-scaled cost, dt*|defect|^2, dt*|eq|^2 of one node at (x, u, xnext); xnOut (optional, 30): the RK2 image of (x, u), in which case xnext may be
// null and the defect is not formed (rollouts of the DDP variant)
// NOT inlined: with this body inlined into the 35 k-instruction line-search kernel of the fp32 build the compiler produced a binary whose
// equality-violation sum read stale registers (2-3x too large, different from run to run; tests/test_gpu_configs.py pins both symptoms).
// As a called function the body is compiled once, for linesearch_kernel and ddp_rollout_kernel alike.
// WP: pointer type of the two 30 x 30 weight matrices: LDS (line search: staged once per workgroup) or generic (DDP rollouts).
// weightStructure: 0 = Q and R' dense; 1 = Q diagonal and R' = diag (forces 0..11) + dense block (leg joint velocities 12..23, the block QMInterface.cpp:283-296 maps
// through the foot Jacobian) + diag (arm 24..29) -- what the reference's task.info produces; decided by the caller from the actual values.  The structured forms skip
// products with exact zeros in the same order of accumulation: bit-identical results.
template <class WP> __device__ __attribute__((noinline)) void nodePerformance(const ModelR& md, const SettingsR& st, WP Qw, WP Rw, int weightStructure, const real* bc, const Schedule& sched, const real* tTimes, const real* tStates, const real* contact, int K, real t, real dt, int phase,
                                       bool terminal, const real* x, const real* u, const real* xnext, real& cost, real& dyn, real& eq, real* xnOut = nullptr) {
  QM_TICK_DECL;
  const int mode = sched.modes[phase];
  real eePosRef[3], eeQuatRef[4];
  eeReference(tTimes, tStates, K, t, eePosRef, eeQuatRef);
  real Ke = 0.0_r, fRef[3] = {0.0_r, 0.0_r, 0.0_r}, pEnv[3] = {0.0_r, 0.0_r, 0.0_r};   // force tracking (own formulation), intermediate nodes only
  if (contact && !terminal) { Ke = st.ee_contact_stiffness; eeContactReference(tTimes, contact, K, t, fRef, pEnv); }
  real k1[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) k1[i] = 0.0_r;
  real c = 0.0_r;
  dyn = 0.0_r; eq = 0.0_r;
  real phi[12];
#pragma unroll 1
  for (int stage = 0; stage < (terminal ? 1 : 2); ++stage) {
    const DblIn in{x, u, stage ? dt : 0.0_r, k1};
    Feet<real> feet;
    real f[12];
    BaseMotion<real> bm;
    centroidalSweep<real>(
        md, st.gravity, in, [&](int cc, Vec3<real> r, Vec3<real> v) { feet.set(cc, r, v); },
        [&](Vec3<real> r, const Mat3<real>& R) {
          const real sdt = stage ? dt : 0.0_r;
          const Vec3<real> fe(-Ke * (x[6] + sdt * k1[6] + r.x - pEnv[0]), -Ke * (x[7] + sdt * k1[7] + r.y - pEnv[1]), -Ke * (x[8] + sdt * k1[8] + r.z - pEnv[2]));
          if (stage == 0) {
            const real muF = Ke != 0.0_r ? st.ee_force_mu : 0.0_r;
            c += 0.5_r * muF * ((fe.x - fRef[0]) * (fe.x - fRef[0]) + (fe.y - fRef[1]) * (fe.y - fRef[1]) + (fe.z - fRef[2]) * (fe.z - fRef[2]));
            real qee[4];
            matrixToQuaternion(R, qee);
            const Vec3<real> od = quaternionDistance(qee, eeQuatRef);
            const real muP = terminal ? st.ee_final_mu_position : st.ee_mu_position, muO = terminal ? st.ee_final_mu_orientation : st.ee_mu_orientation;
            const real hx = x[6] + r.x - eePosRef[0], hy = x[7] + r.y - eePosRef[1], hz = x[8] + r.z - eePosRef[2];
            c += 0.5_r * muP * (hx * hx + hy * hy + hz * hz) + 0.5_r * muO * (od.x * od.x + od.y * od.y + od.z * od.z);
          }
          return fe;
        },
        f, bm);
    if (stage == 0) {
      if (!terminal) {
        for (int cc = 0; cc < 4; ++cc) {
          const Vec3<real> r = feet.r(cc);
          const Vec3<real> vf = bm.dp + cross(bm.omega, r) + feet.v(cc);
          if (contactOf(mode, cc)) { const real hz = vf.z + st.position_error_gain * (x[8] + r.z); eq += vf.x * vf.x + vf.y * vf.y + hz * hz; }
          else {
            real zp, zv;
            swingReference(st, sched, cc, t, phase, zp, zv);
            const real h = vf.z - zv + st.position_error_gain * (x[8] + r.z - zp);
            eq += u[3 * cc] * u[3 * cc] + u[3 * cc + 1] * u[3 * cc + 1] + u[3 * cc + 2] * u[3 * cc + 2] + h * h;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) { k1[i] = f[i]; phi[i] = 0.5_r * dt * f[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 12; ++i) phi[i] += 0.5_r * dt * f[i];
    }
  }
  QM_TICK(0);
  if (terminal) { cost = c; return; }
  if (xnOut) {
    for (int i = 0; i < 12; ++i) xnOut[i] = x[i] + phi[i];
    for (int j = 0; j < 18; ++j) xnOut[12 + j] = x[12 + j] + dt * u[12 + j];
  } else {
    for (int i = 0; i < 12; ++i) { const real d = x[i] + phi[i] - xnext[i]; dyn += d * d; }
    for (int j = 0; j < 18; ++j) { const real d = x[12 + j] + dt * u[12 + j] - xnext[12 + j]; dyn += d * d; }
  }
  QM_TICK(1);
  // tracking cost
  int tIdx; real tAlpha;
  timeSegment(tTimes, K, t, tIdx, tAlpha);
  int nStance = 0;
  for (int k = 0; k < 4; ++k) nStance += contactOf(mode, k) ? 1 : 0;
  const real fzNom = nStance > 0 ? md.total_mass * st.gravity / nStance : 0.0_r;
  // deviations once, in registers; then the two quadratic forms fully unrolled: the weights are read with compile-time offsets from
  // wave-uniform addresses (scalar loads), x / u / the reference are not touched again.  (The rolled double loop re-read x[j], u[j] and the
  // reference states from HBM for each of the 900 (i, j) pairs: most of the kernel's time.)  Same order of operations as before.
  real dx[30], du[30];
#pragma unroll
  for (int j = 0; j < 30; ++j) {
    dx[j] = x[j] - xReference(tStates, K, tIdx, tAlpha, j);
    du[j] = u[j] - ((j < 12 && (j % 3) == 2 && contactOf(mode, j / 3)) ? fzNom : 0.0_r);
  }
  if (weightStructure == 1) {   // wave uniform
#pragma unroll
    for (int i = 0; i < 30; ++i) {
      real qs = 0.0_r, rs = 0.0_r;
      qs += Qw[i * 30 + i] * dx[i];
      if (i >= 12 && i < 24) {
#pragma unroll
        for (int j = 12; j < 24; ++j) rs += Rw[i * 30 + j] * du[j];
      } else rs += Rw[i * 30 + i] * du[i];
      c += 0.5_r * dx[i] * qs + 0.5_r * du[i] * rs;
    }
  } else
#pragma unroll
  for (int i = 0; i < 30; ++i) {
    real qs = 0.0_r, rs = 0.0_r;
#pragma unroll
    for (int j = 0; j < 30; ++j) { qs += Qw[i * 30 + j] * dx[j]; rs += Rw[i * 30 + j] * du[j]; }
    c += 0.5_r * dx[i] * qs + 0.5_r * du[i] * rs;
    // One row of weights at a time.  Without the fence the fp32 build's scheduler hoists all 450 ds_read_b128 of the 1800 weights in front of the
    // first multiply-add, runs out of registers and parks 5.9 KB per lane in scratch (740 spill instructions in this function): the fp32 line
    // search then takes 1.50 ms where the fp64 one takes 0.56 (1024 x N = 200, profiles/r02_fp32_sweep.json).
    if (sizeof(real) == 4) QM_SCHED_FENCE();
  }
  QM_TICK(2);
  const Barrier bp{st.joint_pos_barrier_mu, st.joint_pos_barrier_delta}, bv{st.joint_vel_barrier_mu, st.joint_vel_barrier_delta}, bf{st.friction_barrier_mu, st.friction_barrier_delta};
  for (int i = 0; i < 6; ++i) {
    const real lo = md.q_lower[12 + i], up = md.q_upper[12 + i];
    // (bc: log(delta) of the three barriers and the constant value(-lower) + value(upper) of every limit, formed once by input_weight_kernel, layout.h)
    c += bp.valueL(x[24 + i] - lo, bc[QM_BC_LOGD_POS]) + bp.valueL(up - x[24 + i], bc[QM_BC_LOGD_POS]) - bc[QM_BC_POS0 + i];
    c += bv.valueL(u[24 + i] - st.arm_vel_lower[i], bc[QM_BC_LOGD_VEL]) + bv.valueL(st.arm_vel_upper[i] - u[24 + i], bc[QM_BC_LOGD_VEL]) - bc[QM_BC_VEL0 + i];
  }
  for (int cc = 0; cc < 4; ++cc) if (contactOf(mode, cc)) {
    const real fx = u[3 * cc], fy = u[3 * cc + 1], fz = u[3 * cc + 2];
    c += bf.valueL(st.friction_coefficient * fz - sqrt(fx * fx + fy * fy + st.friction_regularization), bc[QM_BC_LOGD_FRIC]);
  }
  cost = dt * c; dyn *= dt; eq *= dt;
  QM_TICK(3);
  QM_TICK_FLUSH(224, blockIdx.x == 0 && threadIdx.x == 5);
}

// out[e] = x[e] + alpha dx[e] for e = first, first + stride, ... < n, EIGHT elements per pass with all sixteen loads issued before the first store: written as a
// plain loop every element waited for its own two loads (the compiler may not move a load across the store to a pointer it cannot tell apart), 24-47 memory round
// trips in a row per trajectory and pass -- two thirds of this kernel's time (round 3)
__device__ __forceinline__ void axpyStrided(real* out, const real* x, const real* dx, real alpha, int n, int first, int stride) {
  constexpr int UN = 8;
  for (int base = first; base < n; base += UN * stride) {
    real xv[UN], dv[UN];
#pragma unroll
    for (int q = 0; q < UN; ++q) { const int e = base + q * stride, ec = e < n ? e : first; xv[q] = x[ec]; dv[q] = dx[ec]; }
#pragma unroll
    for (int q = 0; q < UN; ++q) { QM_KEEP(xv[q]); QM_KEEP(dv[q]); }
#pragma unroll
    // alpha == 0 (no trial accepted, or a failed factorisation: the direction may hold Inf / NaN from pivots replaced by 1): the iterate itself, not x + 0 * dx
    for (int q = 0; q < UN; ++q) { const int e = base + q * stride; if (e < n) out[e] = alpha == 0.0_r ? xv[q] : xv[q] + alpha * dv[q]; }
  }
}
// sum of squares of v[first], v[first + stride], ..., eight loads in flight
__device__ __forceinline__ real sumSquaresStrided(const real* v, int n, int first, int stride) {
  constexpr int UN = 8;
  real s = 0.0_r;
  for (int base = first; base < n; base += UN * stride) {
    real t[UN];
#pragma unroll
    for (int q = 0; q < UN; ++q) { const int e = base + q * stride; t[q] = v[e < n ? e : first]; }
#pragma unroll
    for (int q = 0; q < UN; ++q) QM_KEEP(t[q]);
#pragma unroll
    for (int q = 0; q < UN; ++q) { const int e = base + q * stride; if (e < n) s += t[q] * t[q]; }
  }
  return s;
}

// section clocks of the kernel itself in the profiling build (tools/riccati_phase_probe.py; thread 0 of instance 0 -> slots 240..247); nothing in the product build
#ifdef QM_RICCATI_TIMING
#define QM_LS_CLOCK(slot) do { const unsigned long long n_ = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) qmk::qmRiccatiTicks[240 + (slot)] += n_ - lsClk; lsClk = n_; } while (0)
#else
#define QM_LS_CLOCK(slot)
#endif
// QM_LS_EXTERN (the product build of qmgpu_api.hip, qm_door_amd/build.py): the kernel is only declared here and DEFINED in qmgpu_ls.hip, a translation unit of its own
// that is compiled with LLVM's interprocedural register allocation ON.  The rest of the library needs it off (wbc_kernel), and without it
// the called node evaluation saves and restores 156 callee-saved registers per lane through scratch memory -- 1,248 B per lane, 0.16 GB of HBM traffic per step and
// 14 % of this kernel's time (profiles/r04h_variant_timing.txt: 0.109 -> 0.095 ms).  The profiling build keeps one translation unit (its clocks live in one device symbol).
#if defined(QM_LS_EXTERN) && !defined(QM_RICCATI_TIMING)
__global__ void __launch_bounds__(256) linesearch_kernel(LsArgs a);
#else
__global__ void __launch_bounds__(256) linesearch_kernel(LsArgs a) {
#ifdef QM_RICCATI_TIMING
  unsigned long long lsClk = clock64();
#endif
  __shared__ real red[3 * 256];
  __shared__ real ctl[8];
  __shared__ int structVotes[256];   // per thread: one of my weight entries lies outside the structured pattern
  __shared__ ModelR mdS;   // the model constants: the sweeps read them with wave-uniform indices, from LDS instead of through the scalar cache
  __shared__ __attribute__((aligned(16))) real wQ[900], wR[900];   // state / input weights of the tracking cost: every lane reads all 1800 of them per node
  const int inst = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  if (a.done[inst]) return;   // converged in an earlier iteration of this call: outputs and statistics stay as they are
  const int N = a.N;
  const SettingsR& st = a.P->settings;
  const real* tg = a.tgrid + size_t(inst) * (N + 1);
  const Schedule sched{a.schedNum[inst], a.schedTimes + size_t(inst) * QMGPU_MAX_EVENTS, a.schedModes + size_t(inst) * (QMGPU_MAX_EVENTS + 1)};
  const real* tTimes = a.targetTimes + size_t(inst) * a.K;
  const real* tStates = a.targetStates + size_t(inst) * a.K * QMGPU_NTARGET;
  const real* X = a.X + size_t(inst) * (N + 1) * 30; const real* U = a.U + size_t(inst) * N * 30;
  const real* dX = a.dX + size_t(inst) * (N + 1) * 30; const real* dU = a.dU + size_t(inst) * N * 30;
  // Two consecutive trial steps (alpha, alpha * decay) are evaluated side by side when the horizon fits half the workgroup: the
  // second half of the threads would otherwise idle, and a launch is as slow as its slowest instance -- one instance of the batch
  // that rejects the full step no longer doubles the kernel time.  Acceptance is tested in the sequential order, so the result is
  // the one of FilterLinesearch's loop.
  const int half = (N + 1 <= nthr / 2) ? nthr / 2 : nthr;
  const int nTr = nthr / half, myTr = tid / half, ltid = tid - myTr * half;
  // trial trajectories: in LDS when the launch reserved room for them (every lane then reads its node's x, u, x_next from LDS instead of
  // 90 scattered HBM loads per node); in the HBM scratch otherwise (long horizons)
  QM_DYNAMIC_LDS(trialLds);
  real* Xt = a.trialInLds ? trialLds + size_t(myTr) * (2 * N + 1) * 30 : a.Xt + (size_t(inst) * 2 + myTr) * (N + 1) * 30;
  real* Ut = a.trialInLds ? Xt + (N + 1) * 30 : a.Ut + (size_t(inst) * 2 + myTr) * N * 30;

  // the weights into LDS; while copying, every thread checks its entries against the structured pattern (nodePerformance: weightStructure) and records a vote
  // (structVotes[tid] = 1: an entry outside the pattern); thread 0 combines the votes below, after the barrier, into ctl[7] = 1 (structured) / 0 (dense forms)
  // the model constants and the baseline node metrics are requested together with the weights: ONE memory round trip for the prologue (were three in a row)
  constexpr int MDW = (int(sizeof(ModelR) / 4) + 255) / 256;   // 32-bit words of the model per thread of a 256-thread launch
  int mdw[2 * MDW];                                            // (a 128-thread launch takes twice as many)
  const int mdPer = nthr >= 256 ? MDW : 2 * MDW;
  real m0 = 0.0_r, d0 = 0.0_r, e0 = 0.0_r;
  {
    bool outside = false;
    real qv[4], rv4[4];   // (900 = 3.5 x 256: four entries per thread, all eight loads first)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) { const int e = tid + q4 * nthr, ec = e < 900 ? e : tid; qv[q4] = st.Q[ec]; rv4[q4] = a.Rw[ec]; }
    {
      const int* src = reinterpret_cast<const int*>(&a.P->model);
#pragma unroll
      for (int w = 0; w < 2 * MDW; ++w) { const int e = tid + w * nthr; mdw[w] = (w < mdPer && e < int(sizeof(ModelR) / 4)) ? src[e] : 0; }
    }
    real mk[3] = {0.0_r, 0.0_r, 0.0_r};
    if (tid <= N) { const real* m = a.metrics + (size_t(inst) * (N + 1) + tid) * NODE_METRICS; mk[0] = m[0]; mk[1] = m[1]; mk[2] = m[2]; }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) { QM_KEEP(qv[q4]); QM_KEEP(rv4[q4]); }
#pragma unroll
    for (int w = 0; w < 2 * MDW; ++w) QM_KEEP(mdw[w]);
    QM_KEEP(mk[0]); QM_KEEP(mk[1]); QM_KEEP(mk[2]);
    if (tid <= N) { m0 = mk[0]; d0 = mk[1]; e0 = mk[2]; }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int e = tid + q4 * nthr;
      if (e < 900) {
        const real q = qv[q4], r = rv4[q4];
        wQ[e] = q; wR[e] = r;
        const int i = e / 30, j = e - 30 * i;
        const bool legBlock = i >= 12 && i < 24 && j >= 12 && j < 24;
        outside = outside || (i != j && q != 0.0_r) || (i != j && !legBlock && r != 0.0_r);
      }
    }
    for (int e = tid + 4 * nthr; e < 900; e += nthr) {   // a 128-thread launch (lsThreads): the rest of the 900 entries; no iteration at 256 threads
      const real q = st.Q[e], r = a.Rw[e];
      wQ[e] = q; wR[e] = r;
      const int i = e / 30, j = e - 30 * i;
      const bool legBlock = i >= 12 && i < 24 && j >= 12 && j < 24;
      outside = outside || (i != j && q != 0.0_r) || (i != j && !legBlock && r != 0.0_r);
    }
    structVotes[tid] = outside ? 1 : 0;
  }
  {
    int* dst = reinterpret_cast<int*>(&mdS);
#pragma unroll
    for (int w = 0; w < 2 * MDW; ++w) { const int e = tid + w * nthr; if (w < mdPer && e < int(sizeof(ModelR) / 4)) dst[e] = mdw[w]; }
  }
  // baseline performance (sum of the LQ kernel's node metrics): the first node of every thread came with the prologue's loads, the others (horizons beyond the thread count) here
  for (int k = tid + nthr; k <= N; k += nthr) { const real* m = a.metrics + (size_t(inst) * (N + 1) + k) * NODE_METRICS; m0 += m[0]; d0 += m[1]; e0 += m[2]; }
  red[tid] = m0; red[256 + tid] = d0; red[512 + tid] = e0;
  __syncthreads();
  if (tid == 0) {
    real s0 = 0, s1 = 0, s2 = 0;
    int dense = 0;
    const int nSum = N + 1 < nthr ? N + 1 : nthr;   // threads beyond the last node hold exact zeros: the sum over the nodes, in node order, is the same number
    for (int i = 0; i < nSum; ++i) { s0 += red[i]; s1 += red[256 + i]; s2 += red[512 + i]; }
    for (int i = 0; i < nthr; ++i) dense |= structVotes[i];
    ctl[0] = s0; ctl[1] = sqrt(s1 + s2); ctl[7] = dense ? 0.0_r : 1.0_r;
  }
  __syncthreads();
  QM_LS_CLOCK(0);
  const real merit0 = ctl[0], viol0 = ctl[1];
  const int weightStructure = int(ctl[7]);
  const real armijo = a.instStats[size_t(inst) * 4 + 0];
  const real ricStatus = a.instStats[size_t(inst) * 4 + 1];

  real alpha = 1.0_r, merit1 = merit0, viol1 = viol0;
  int stepType = 0;
  bool accepted = false;
  // A failed Riccati factorisation (pivots replaced by 1: the direction is garbage) takes no trial at all: alpha = 0, the new iterate IS the
  // incoming one -- as the oracle's sqpIteration, which returns (X, U) unchanged with status 1 (workgroup-uniform: read from HBM by every thread).
#pragma unroll 1
  for (int trial = 0; trial < (ricStatus == 0.0_r ? 64 : 0); ++trial) {
    const real alphaMine = myTr ? alpha * st.alpha_decay : alpha;
    axpyStrided(Xt, X, dX, alphaMine, (N + 1) * 30, ltid, half);
    axpyStrided(Ut, U, dU, alphaMine, N * 30, ltid, half);
    __syncthreads();
    QM_LS_CLOCK(1);
    real cs = 0.0_r, ds = 0.0_r, es = 0.0_r;
    for (int k = ltid; k <= N; k += half) {
      real c, d, e;
      const bool term = k == N;
      nodePerformance(mdS, st, QM_TO_LDS_PTR(real, wQ), QM_TO_LDS_PTR(real, wR), weightStructure, a.Rw + QM_RW_DERIVED, sched, tTimes, tStates, a.eeContact ? a.eeContact + size_t(inst) * a.K * 6 : nullptr, a.K, tg[k], a.dtgrid[size_t(inst) * (N + 1) + k], a.nodePhase[size_t(inst) * (N + 1) + k], term, Xt + k * 30, term ? Ut : Ut + k * 30, term ? Xt + k * 30 : Xt + (k + 1) * 30, c, d, e);
      cs += c; ds += d; es += e;
    }
    QM_LS_CLOCK(2);
    red[tid] = cs; red[256 + tid] = ds; red[512 + tid] = es;
    __syncthreads();
    QM_LS_CLOCK(3);
    if (tid == 0) {
      real acc = 0.0_r, accAlpha = alpha, m1 = merit0, v1 = viol0; int type = 0;
      for (int tr = 0; tr < nTr && acc == 0.0_r; ++tr) {
        const real al = tr ? alpha * st.alpha_decay : alpha;
        if (tr && al < st.alpha_min) break;        // the sequential loop would have stopped before this trial
        real s0 = 0, s1 = 0, s2 = 0;
        const int nTrial = N + 1 < half ? N + 1 : half;   // as above: the threads of the trial beyond its last node hold zeros
        for (int i = tr * half; i < tr * half + nTrial; ++i) { s0 += red[i]; s1 += red[256 + i]; s2 += red[512 + i]; }
        m1 = s0; v1 = sqrt(s1 + s2);
        bool ok;
        // upstream FilterLinesearch::acceptStep
        if (!a.lineSearch) { ok = true; type = 0; }
        else if (v1 > st.g_max) { ok = v1 < (1.0_r - st.gamma_c) * viol0; type = 1; }
        else if (v1 < st.g_min && viol0 < st.g_min && al * armijo < 0.0_r) { ok = m1 < merit0 + st.armijo_factor * al * armijo; type = 3; }
        else { ok = m1 < merit0 - st.gamma_c * viol0 || v1 < (1.0_r - st.gamma_c) * viol0; type = 2; }
        accAlpha = al;
        if (ok) acc = 1.0_r;
      }
      ctl[2] = m1; ctl[3] = v1; ctl[4] = acc; ctl[5] = real(type); ctl[6] = accAlpha;
    }
    __syncthreads();
    QM_LS_CLOCK(4);
    merit1 = ctl[2]; viol1 = ctl[3]; stepType = int(ctl[5]);
    accepted = ctl[4] != 0.0_r;
    const real lastAlpha = ctl[6];
    __syncthreads();
    if (accepted) { alpha = lastAlpha; break; }
    alpha = lastAlpha * st.alpha_decay;
    if (alpha < st.alpha_min) break;
  }
  if (!accepted) { alpha = 0.0_r; stepType = 4; merit1 = merit0; viol1 = viol0; }
  // ---- write the new iterate
  real* oX = a.outX + size_t(inst) * (N + 1) * 30; real* oU = a.outU + size_t(inst) * N * 30;
  axpyStrided(oX, X, dX, alpha, (N + 1) * 30, tid, nthr);
  axpyStrided(oU, U, dU, alpha, N * 30, tid, nthr);
  for (int k = tid; k <= N; k += nthr) { a.outT[size_t(inst) * (N + 1) + k] = tg[k]; a.outMode[size_t(inst) * (N + 1) + k] = a.nodeMode[size_t(inst) * (N + 1) + k]; }
  QM_LS_CLOCK(5);
  // ---- upstream SqpSolver::checkConvergence: iteration limit, step size, metrics, primal step (l2 norms over the whole horizon)
  const real sx = sumSquaresStrided(dX, (N + 1) * 30, tid, nthr), su = sumSquaresStrided(dU, N * 30, tid, nthr);
  __syncthreads();
  red[tid] = sx; red[256 + tid] = su;
  __syncthreads();
  if (tid == 0) {
    real s0 = 0, s1 = 0;
    for (int i = 0; i < nthr; ++i) { s0 += red[i]; s1 += red[256 + i]; }
    int conv = 0;
    if (a.iteration + 1 >= st.sqp_iterations) conv = 1;
    else if (alpha < st.alpha_min) conv = 2;
    else if (fabs(merit1 - merit0) < st.cost_tol && viol1 < st.g_min) conv = 3;
    else if (alpha * sqrt(s0) < st.delta_tol && alpha * sqrt(s1) < st.delta_tol) conv = 4;
    if (ricStatus != 0.0_r && conv == 0) conv = 2;   // a failed factorisation leaves the iterate where it was: nothing more to do
    a.done[inst] = conv;
    if (a.outStats) {
      real* s = a.outStats + size_t(inst) * QMGPU_NSTATS;
      s[0] = merit0; s[1] = viol0; s[2] = merit1; s[3] = viol1; s[4] = alpha; s[5] = real(stepType); s[6] = armijo; s[7] = ricStatus;
      s[8] = real(a.iteration + 1); s[9] = real(conv);
    }
  }
  QM_LS_CLOCK(6);
}
#endif   // QM_LS_EXTERN

}  // namespace qmk
