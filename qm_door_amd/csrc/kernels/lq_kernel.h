// lq_node_kernel -- fused per-node LQ approximation + constraint projection.  One wavefront per shooting node.
//
// Replaces, per node: QMPreComputation::request (QMPreComputation.cpp:50-89),
// QMDynamicsAD::linearApproximation x2 for the RK2 stages (QMDynamicsAD.cpp:30-33), the quadratic approximation of the
// tracking cost / EE soft constraint / joint-limit and friction-cone barriers (QMInterface.cpp:99-121), the per-foot
// equality constraints (QMInterface.cpp:123-131) and upstream ocs2_sqp's discretisation + QR constraint projection.
//
// Wave layout
//   phase AD   lane l carries the tangent d/dx_l (l<30) or d/du_{l-30} (30<=l<60) through BOTH RK2 stages, so after the
//              sweep lane l owns column l of [A_d | B_d] and of every constraint / EE-error Jacobian (du.h).
//   phase LQ   lane c<30 owns column c of Q, R (symmetric) and of Px; lane 30 owns Pe; lane 31+j owns null-space column j
//              of Pu.  All small GEMMs are "matrix in LDS (broadcast reads) x my column in registers", no cross-lane
//              reductions; Householder vectors are applied column-wise the same way.
// LDS per wave: B (7.0 KiB) + R (7.0 KiB) + [C D e] aliased with [Px Pe Pu] (11.7 KiB) + reflectors/R1/vectors (~7 KiB).
#pragma once
#include "layout.h"
#include "schedule_dev.h"
#include "sweep_dev.h"

namespace qmk {

struct LqArgs {
  const qmgpu_problem* P;
  const double* Rw;          // R' [30][30]
  int batch, N, K;
  const double* tgrid;       // [batch][N+1]
  const double* X;           // [batch][N+1][30] current iterate
  const double* U;           // [batch][N][30]
  const double* targetTimes; // [batch][K]
  const double* targetStates;// [batch][K][37]
  const int* schedNum;       // [batch]
  const double* schedTimes;  // [batch][MAX_EVENTS]
  const int* schedModes;     // [batch][MAX_EVENTS+1]
  const double* zeros;       // >= 64 zeros
  double* stages;            // [batch][N+1][STAGE_DOUBLES]
  int* stageNc;              // [batch][N+1]
  int* nodeMode;             // [batch][N+1]
  double* metrics;           // [batch][N+1][NODE_METRICS]
  double* debug;             // [batch][N+1][DBG_DOUBLES] or null
};

struct DuIn {
  const double* x;
  const double* u;
  int lane;
  double dtS;
  const Du* k1;  // [12] first-stage slope (zero while dtS == 0)
  __device__ __forceinline__ Du sx(int i) const { return Du(x[i], lane == i ? 1.0 : 0.0); }
  __device__ __forceinline__ Du su(int i) const { return Du(u[i], lane == 30 + i ? 1.0 : 0.0); }
  __device__ __forceinline__ Du hn(int i) const { return sx(i) + dtS * k1[i]; }
  __device__ __forceinline__ Du euler(int i) const { return sx(9 + i) + dtS * k1[9 + i]; }
  __device__ __forceinline__ Du q(int j) const { return sx(12 + j) + dtS * su(12 + j); }
  __device__ __forceinline__ Du qd(int j) const { return su(12 + j); }
  __device__ __forceinline__ Vec3<Du> force(int c) const { return Vec3<Du>(su(3 * c), su(3 * c + 1), su(3 * c + 2)); }
};

constexpr int PAW = 52;                      // row stride of [Px | Pe | Pu] in LDS (49 used, read in groups of four)
constexpr int CDW = 62;                      // row stride of [C | D | e] (61 used), aliases the same region
constexpr int L_B = 0;                       // B      [30][30]
constexpr int L_R = L_B + 900;               // R      [30][30]   (dt-scaled)
constexpr int L_PA = L_R + 900;              // Pall   [30][PAW]  /  CD [16][CDW]
constexpr int L_V = L_PA + 30 * PAW;         // Householder vectors [16][32] (entry 30 = beta)
constexpr int L_RL = L_V + 16 * 32;          // R1 [16][16]
constexpr int L_EEJ = L_RL + 256;            // EE error Jacobian [6][32]
constexpr int L_VEC = L_EEJ + 192;           // small vectors: b[30] r[30] e[16] eeh[6] ...
constexpr int L_RED = L_VEC + 96;            // reduction scratch [64]
constexpr int LQ_LDS_DOUBLES = L_RED + 64;   // 4420 doubles = 34.5 KiB
static_assert(16 * CDW <= 30 * PAW, "CD must fit in the Pall region");

// dot product of a broadcast LDS row with a register vector, three independent FMA chains (one wavefront per SIMD: the fp64 FMA
// latency is hidden by instruction-level parallelism only)
__device__ __forceinline__ double dot30(const double* row, const double (&z)[30]) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int i = 0; i < 30; i += 3) { s0 += row[i] * z[i]; s1 += row[i + 1] * z[i + 1]; s2 += row[i + 2] * z[i + 2]; }
  return s0 + s1 + s2;
}

__device__ __forceinline__ double waveSum(double* red, int lane, double v) {
  red[lane] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < 64; ++i) s += red[i];
  __syncthreads();
  return s;
}

__global__ void __launch_bounds__(64) lq_node_kernel(LqArgs a) {
  __shared__ double lds[LQ_LDS_DOUBLES];
  const int lane = threadIdx.x;
  const int node = blockIdx.x % (a.N + 1);
  const int inst = blockIdx.x / (a.N + 1);
  const bool terminal = node == a.N;
  const qmgpu_model& md = a.P->model;
  const qmgpu_settings& st = a.P->settings;

  double* Bm = lds + L_B; double* Rm = lds + L_R; double* PA = lds + L_PA; double* CD = lds + L_PA; double* Vh = lds + L_V;
  double* RL = lds + L_RL; double* EEJ = lds + L_EEJ; double* bv = lds + L_VEC; double* rv = bv + 30; double* ev = rv + 30; double* eeh = ev + 16;
  double* red = lds + L_RED;

  const double* tg = a.tgrid + size_t(inst) * (a.N + 1);
  const double t = tg[node];
  const double dt = terminal ? 0.0 : tg[node + 1] - t;
  const double* x = a.X + (size_t(inst) * (a.N + 1) + node) * 30;
  const double* u = terminal ? a.zeros : a.U + (size_t(inst) * a.N + node) * 30;
  const double* xnext = terminal ? x : x + 30;
  const Schedule sched{a.schedNum[inst], a.schedTimes + size_t(inst) * QMGPU_MAX_EVENTS, a.schedModes + size_t(inst) * (QMGPU_MAX_EVENTS + 1)};
  const int phase = phaseAt(sched, t);
  const int mode = sched.modes[phase];
  const double* tTimes = a.targetTimes + size_t(inst) * a.K;
  const double* tStates = a.targetStates + size_t(inst) * a.K * QMGPU_NTARGET;
  double eePosRef[3], eeQuatRef[4];
  eeReference(tTimes, tStates, a.K, t, eePosRef, eeQuatRef);
  const double muP = terminal ? st.ee_final_mu_position : st.ee_mu_position, muO = terminal ? st.ee_final_mu_orientation : st.ee_mu_orientation;

  // ================================================================== phase AD: both RK2 stages with lane tangents
  Du k1[12];  // first-stage slope; after the second stage it holds phi = dt/2 (k1 + k2)
#pragma unroll
  for (int i = 0; i < 12; ++i) k1[i] = Du(0.0);
  int nc = 0;
#pragma unroll 1
  for (int stage = 0; stage < (terminal ? 1 : 2); ++stage) {
    const DuIn in{x, u, lane, stage ? dt : 0.0, k1};
    Feet<Du> feet;
    Du f[12];
    BaseMotion<Du> bm;
    const Du p0x = in.sx(6) + in.dtS * k1[6], p0y = in.sx(7) + in.dtS * k1[7], p0z = in.sx(8) + in.dtS * k1[8];
    centroidalSweep<Du>(
        md, st.gravity, in, [&](int c, Vec3<Du> r, Vec3<Du> v) { feet.set(c, r, v); },
        [&](Vec3<Du> r, const Mat3<Du>& R) {
          if (stage == 0) {  // end-effector pose error (EndEffectorConstraint.cpp:36-78), rows -> LDS
            Du qee[4];
            matrixToQuaternion(R, qee);
            const Vec3<Du> od = quaternionDistance(qee, eeQuatRef);
            const Du h[6] = {p0x + r.x - eePosRef[0], p0y + r.y - eePosRef[1], p0z + r.z - eePosRef[2], od.x, od.y, od.z};
#pragma unroll
            for (int q = 0; q < 6; ++q) { if (lane < 32) EEJ[q * 32 + lane] = h[q].d; eeh[q] = h[q].v; }
          }
        },
        f, bm);
    if (stage == 0) {
      // ---- equality constraints in the insertion order of QMInterface.cpp:116-131 -> rows [C | D | e] in LDS
      if (!terminal) {
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
          const bool contact = contactOf(mode, c);
          const Vec3<Du> r = feet.r(c);
          const Vec3<Du> vf = bm.dp + cross(bm.omega, r) + feet.v(c);
          auto putRow = [&](int row, Du h) { if (lane < 60) CD[row * CDW + lane] = h.d; ev[row] = h.v; };
          if (contact) {  // zeroVelocity (QMInterface.cpp:126, 324-339)
            putRow(nc, vf.x); putRow(nc + 1, vf.y); putRow(nc + 2, vf.z);
            nc += 3;
          } else {  // zeroForce (QMInterface.cpp:123-124) then normalVelocity (QMPreComputation.cpp:56-66)
#pragma unroll
            for (int q = 0; q < 3; ++q) putRow(nc + q, in.su(3 * c + q));
            double zp, zv;
            swingReference(st, sched, c, t, phase, zp, zv);
            putRow(nc + 3, vf.z - zv + st.position_error_gain * (p0z + r.z - zp));
            nc += 4;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) k1[i] = f[i];
    } else {
#pragma unroll
      for (int i = 0; i < 12; ++i) k1[i] = 0.5 * dt * (k1[i] + f[i]);
    }
  }
  __syncthreads();

  double* rec = a.stages + (size_t(inst) * (a.N + 1) + node) * STAGE_DOUBLES;
  double* dbg = a.debug ? a.debug + (size_t(inst) * (a.N + 1) + node) * DBG_DOUBLES : nullptr;
  if (lane == 0) { a.stageNc[size_t(inst) * (a.N + 1) + node] = nc; a.nodeMode[size_t(inst) * (a.N + 1) + node] = mode; }
  const Du* phi = k1;

  // ================================================================== cost (lanes < 30 own a column of Q / R)
  int tIdx; double tAlpha;
  timeSegment(tTimes, a.K, t, tIdx, tAlpha);
  const int c = lane;
  double qc = 0.0, costPart = 0.0;
  const double sc = terminal ? 1.0 : dt;  // intermediate costs are scaled by dt, the terminal cost is not
  {
    double Qcol[30];
#pragma unroll
    for (int i = 0; i < 30; ++i) Qcol[i] = 0.0;
    if (c < 30) {
      double ej[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) ej[q] = EEJ[q * 32 + c];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const double mu = q < 3 ? muP : muO;
        qc += mu * eeh[q] * ej[q];
#pragma unroll
        for (int i = 0; i < 30; ++i) Qcol[i] += mu * EEJ[q * 32 + i] * ej[q];
      }
      if (c < 6) costPart += 0.5 * (c < 3 ? muP : muO) * eeh[c] * eeh[c];
    }
    if (!terminal && c < 30) {
      double Qdx = 0.0;
#pragma unroll
      for (int i = 0; i < 30; ++i) {
        const double qw = st.Q[i * 30 + c];
        Qdx += qw * (x[i] - xReference(tStates, a.K, tIdx, tAlpha, i));
        Qcol[i] += qw;
      }
      qc += Qdx;
      costPart += 0.5 * (x[c] - xReference(tStates, a.K, tIdx, tAlpha, c)) * Qdx;
      if (c >= 24) {  // arm joint position soft box (QMInterface.cpp:177-219)
        const Barrier bp{st.joint_pos_barrier_mu, st.joint_pos_barrier_delta};
        const double lo = md.q_lower[c - 12], up = md.q_upper[c - 12];
        const double hl = x[c] - lo, hu = up - x[c];
        costPart += bp.value(hl) + bp.value(hu) - (bp.value(-lo) + bp.value(up));
        qc += bp.d1(hl) - bp.d1(hu);
        const double dd = bp.d2(hl) + bp.d2(hu);
#pragma unroll
        for (int k = 24; k < 30; ++k) if (k == c) Qcol[k] += dd;
      }
    }
    if (c < 30) {
#pragma unroll
      for (int i = 0; i < 30; ++i) rec[OFF_QT + i * 30 + c] = sc * Qcol[i];  // Q~ is completed in place after the projection
      if (dbg) { for (int i = 0; i < 30; ++i) dbg[DBG_Q + i * 30 + c] = sc * Qcol[i]; }
    }
    qc *= sc;
  }

  if (terminal) {
    const double nodeCost = waveSum(red, lane, costPart);
    if (c < 30) { rec[OFF_qt + c] = qc; if (dbg) dbg[DBG_q + c] = qc; }
    if (lane == 0) { double* m = a.metrics + (size_t(inst) * (a.N + 1) + node) * NODE_METRICS; m[0] = nodeCost; m[1] = 0.0; m[2] = 0.0; m[3] = 0.0; }
    return;
  }

  // ---- Jacobian columns of the RK2 map: Phi = x + dt/2 (k1 + k2); rows 12.. are x_j + dt v_j exactly.
  //      A columns go straight to the stage record (completed in place below), B columns to LDS.
  if (c < 60) {
    double col[30];
#pragma unroll
    for (int i = 0; i < 12; ++i) col[i] = phi[i].d + (c == i ? 1.0 : 0.0);
#pragma unroll
    for (int j = 0; j < 18; ++j) col[12 + j] = (c == 12 + j ? 1.0 : 0.0) + (c == 42 + j ? dt : 0.0);
    if (c < 30) {
#pragma unroll
      for (int i = 0; i < 30; ++i) rec[OFF_AT + i * 30 + c] = col[i];
      if (dbg) { for (int i = 0; i < 30; ++i) dbg[DBG_A + i * 30 + c] = col[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 30; ++i) Bm[i * 30 + (c - 30)] = col[i];
    }
  }
  if (lane == 0) {
    for (int i = 0; i < 12; ++i) bv[i] = x[i] + phi[i].v - xnext[i];
    for (int j = 0; j < 18; ++j) bv[12 + j] = x[12 + j] + dt * u[12 + j] - xnext[12 + j];
  }
  // ---- input cost: R' + friction-cone and arm-velocity barriers (column c of R into LDS)
  {
    int nStance = 0;
    for (int k = 0; k < 4; ++k) nStance += contactOf(mode, k) ? 1 : 0;
    const double fzNom = nStance > 0 ? md.total_mass * st.gravity / nStance : 0.0;
    if (c < 30) {
      double Rdu = 0.0;
      double Rcol[30];
#pragma unroll
      for (int i = 0; i < 30; ++i) {
        const double unom = (i < 12 && (i % 3) == 2 && contactOf(mode, i / 3)) ? fzNom : 0.0;
        const double rw = a.Rw[i * 30 + c];
        Rdu += rw * (u[i] - unom);
        Rcol[i] = rw;
      }
      const double unomc = (c < 12 && (c % 3) == 2 && contactOf(mode, c / 3)) ? fzNom : 0.0;
      double rc = Rdu;
      costPart += 0.5 * (u[c] - unomc) * Rdu;
      if (c >= 24) {  // arm joint velocity soft box (QMInterface.cpp:221-254)
        const Barrier bvel{st.joint_vel_barrier_mu, st.joint_vel_barrier_delta};
        const int i = c - 24;
        const double vl = u[c] - st.arm_vel_lower[i], vu = st.arm_vel_upper[i] - u[c];
        costPart += bvel.value(vl) + bvel.value(vu) - (bvel.value(-st.arm_vel_lower[i]) + bvel.value(st.arm_vel_upper[i]));
        rc += bvel.d1(vl) - bvel.d1(vu);
        const double dd = bvel.d2(vl) + bvel.d2(vu);
#pragma unroll
        for (int k = 24; k < 30; ++k) if (k == c) Rcol[k] += dd;
      }
      if (c < 12 && contactOf(mode, c / 3)) {  // friction cone barrier (QMInterface.cpp:344-358), column c of its 3x3 block
        const Barrier bf{st.friction_barrier_mu, st.friction_barrier_delta};
        const int fo = 3 * (c / 3), ac = c % 3;
        const double fx = u[fo], fy = u[fo + 1], fz = u[fo + 2];
        const double F = sqrt(fx * fx + fy * fy + st.friction_regularization), F3 = F * F * F;
        const double h = st.friction_coefficient * fz - F;
        const double gx = -fx / F, gy = -fy / F, gz = st.friction_coefficient;
        const double hxx = -(fy * fy + st.friction_regularization) / F3 - st.friction_hessian_shift, hxy = fx * fy / F3;
        const double hyy = -(fx * fx + st.friction_regularization) / F3 - st.friction_hessian_shift, hzz = -st.friction_hessian_shift;
        const double gac = ac == 0 ? gx : (ac == 1 ? gy : gz);
        const double h0 = ac == 0 ? hxx : (ac == 1 ? hxy : 0.0), h1 = ac == 0 ? hxy : (ac == 1 ? hyy : 0.0), h2 = ac == 2 ? hzz : 0.0;
        const double p1 = bf.d1(h), p2 = bf.d2(h);
        if (ac == 0) costPart += bf.value(h);
        rc += p1 * gac;
        const double e0 = p2 * gx * gac + p1 * h0, e1 = p2 * gy * gac + p1 * h1, e2 = p2 * gz * gac + p1 * h2;
#pragma unroll
        for (int k = 0; k < 12; ++k) { if (k == fo) Rcol[k] += e0; if (k == fo + 1) Rcol[k] += e1; if (k == fo + 2) Rcol[k] += e2; }
      }
#pragma unroll
      for (int i = 0; i < 30; ++i) Rm[i * 30 + c] = dt * Rcol[i];
      rv[c] = dt * rc;
    }
  }
  const double nodeCost = dt * waveSum(red, lane, costPart);  // (barriers inside: LDS writes above are visible below)

  if (lane == 0) {
    double dyn = 0.0, eq = 0.0;
    for (int i = 0; i < 30; ++i) dyn += bv[i] * bv[i];
    for (int i = 0; i < nc; ++i) eq += ev[i] * ev[i];
    double* m = a.metrics + (size_t(inst) * (a.N + 1) + node) * NODE_METRICS;
    m[0] = nodeCost; m[1] = dt * dyn; m[2] = dt * eq; m[3] = 0.0;
  }
  if (dbg && c < 30) {
    for (int i = 0; i < 30; ++i) { dbg[DBG_B + i * 30 + c] = Bm[i * 30 + c]; dbg[DBG_R + i * 30 + c] = Rm[i * 30 + c]; }
    dbg[DBG_b + c] = bv[c]; dbg[DBG_q + c] = qc; dbg[DBG_r + c] = rv[c];
    for (int r = 0; r < nc; ++r) { dbg[DBG_C + r * 30 + c] = CD[r * CDW + c]; dbg[DBG_D + r * 30 + c] = CD[r * CDW + 30 + c]; }
    if (c < nc) dbg[DBG_e + c] = ev[c];
  }

  // ================================================================== projection: Householder QR of D^T (30 x nc)
  // lane j < nc owns column j of D^T (= row j of D); lane c < 31 owns column c of [C | e]
  double z[30];
  {
    double dcol[30];
#pragma unroll
    for (int i = 0; i < 30; ++i) dcol[i] = (lane < nc) ? CD[lane * CDW + 30 + i] : 0.0;
    double ce[NCMAX];
#pragma unroll
    for (int r = 0; r < NCMAX; ++r) ce[r] = (lane < 30 && r < nc) ? CD[r * CDW + lane] : ((lane == 30 && r < nc) ? ev[r] : 0.0);
    __syncthreads();  // the [C D e] region is free from here on (it becomes W = R Pall)
#pragma unroll 1
    for (int k = 0; k < nc; ++k) {
      if (lane == k) {
        double n2 = 0.0, dk = 0.0;
#pragma unroll
        for (int i = 0; i < 30; ++i) { if (i >= k) n2 += dcol[i] * dcol[i]; if (i == k) dk = dcol[i]; }
        const double nrm = sqrt(n2);
        const double alpha = dk > 0.0 ? -nrm : nrm;
        double vn = 0.0;
#pragma unroll
        for (int i = 0; i < 30; ++i) {
          const double v = (i > k) ? dcol[i] : ((i == k) ? dk - alpha : 0.0);
          Vh[k * 32 + i] = v;
          vn += v * v;
          if (i == k) dcol[i] = alpha; else if (i > k) dcol[i] = 0.0;
        }
        Vh[k * 32 + 30] = vn > 0.0 ? 2.0 / vn : 0.0;
      }
      __syncthreads();
      if (lane > k && lane < nc) {
        double s = dot30(Vh + k * 32, dcol) * Vh[k * 32 + 30];
#pragma unroll
        for (int i = 0; i < 30; ++i) dcol[i] -= s * Vh[k * 32 + i];
      }
    }
    // R1 (upper triangular, nc x nc): lane j holds column j
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < NCMAX; ++i) RL[i * 16 + lane] = (lane < nc) ? dcol[i] : (i == lane ? 1.0 : 0.0);
    }
    __syncthreads();
    // Y = R1^-T [C | e]  (forward substitution, column per lane), then z = -Q [Y; 0]  /  z = Q e_{nc+j}
    double y[NCMAX];
#pragma unroll
    for (int i = 0; i < NCMAX; ++i) {
      double s = ce[i];
#pragma unroll
      for (int k = 0; k < NCMAX; ++k) if (k < i) s -= RL[k * 16 + i] * y[k];
      y[i] = (i < nc) ? s / RL[i * 16 + i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 30; ++i) {
      double v = 0.0;
      if (lane <= 30) { if (i < NCMAX) v = -y[i < NCMAX ? i : 0]; }
      else if (i == nc + (lane - 31)) v = 1.0;
      z[i] = v;
    }
  }
  const int nt = 30 - nc;  // projected input dimension m~
#pragma unroll 1
  for (int k = nc - 1; k >= 0; --k) {
    const double s = dot30(Vh + k * 32, z) * Vh[k * 32 + 30];
#pragma unroll
    for (int i = 0; i < 30; ++i) z[i] -= s * Vh[k * 32 + i];
  }
  const bool active = lane < 31 + nt;
  const bool isX = lane < 30, isE = lane == 30, isU = lane > 30 && active;

  // ================================================================== projected dynamics and cost
  // row by row: o_i = (B z)_i completes A~ / b~ / B~ in the stage record, w_i = (R z)_i goes to the shared matrix W = R [Px Pe Pu]
  double* WL = PA;
  double tz = 0.0;
#pragma unroll
  for (int i = 0; i < 30; ++i) tz += rv[i] * z[i];
  if (isX) {
#pragma unroll
    for (int i = 0; i < 30; ++i) rec[OFF_PX + i * 30 + lane] = z[i];
  } else if (isE) {
#pragma unroll
    for (int i = 0; i < 30; ++i) rec[OFF_PE + i] = z[i];
  } else if (isU) {
#pragma unroll
    for (int i = 0; i < 30; ++i) rec[OFF_PU + i * MT + (lane - 31)] = z[i];
  }
#pragma unroll 1
  for (int i0 = 0; i0 < 30; i0 += 3) {  // three rows per trip: six independent FMA chains
    double sb[3] = {0.0, 0.0, 0.0}, sr[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 30; ++k) {
#pragma unroll
      for (int r = 0; r < 3; ++r) { sb[r] += Bm[(i0 + r) * 30 + k] * z[k]; sr[r] += Rm[(i0 + r) * 30 + k] * z[k]; }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int i = i0 + r;
      if (isX) rec[OFF_AT + i * 30 + lane] += sb[r];
      else if (isE) rec[OFF_bt + i] = bv[i] + sb[r];
      else if (isU) rec[OFF_BT + i * MT + (lane - 31)] = sb[r];
      if (active) WL[i * PAW + lane] = sr[r];
    }
  }
  __syncthreads();
  // G[a][lane] = sum_k W[k][a] z[k]   (= Pall_a^T R Pall_lane)
  double g30 = 0.0;
#pragma unroll 1
  for (int a0 = 0; a0 < 31 + nt; a0 += 4) {  // four columns of W per trip: four independent FMA chains
    double sg[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 30; ++k) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sg[r] += WL[k * PAW + a0 + r] * z[k];  // columns beyond 30 + nt are padding (PAW = 50 >= 49 + 3)
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int aa = a0 + r;
      if (aa >= 31 + nt) continue;
      const double sv = sg[r];
      if (aa == 30) g30 = sv;
      if (isX) {
        if (aa < 30) rec[OFF_QT + aa * 30 + lane] += sv;
        else if (aa > 30) rec[OFF_PT + (aa - 31) * 30 + lane] = sv;
      } else if (isU && aa > 30) rec[OFF_RT + (aa - 31) * MT + (lane - 31)] = sv;
    }
  }
  if (isX) rec[OFF_qt + lane] = qc + tz + g30;
  else if (isU) rec[OFF_rt + (lane - 31)] = tz + g30;
}

}  // namespace qmk
