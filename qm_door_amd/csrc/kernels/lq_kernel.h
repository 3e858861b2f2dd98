// lq_node_kernel -- per-node cost, constraint projection and projected stage record.  One wavefront per shooting node.
// The derivative rows it consumes come from ad_node_kernel (ad_kernel.h): two launches so that each half gets the occupancy it can
// use -- the AD sweep is fp64-VALU bound and needs the whole register file, the projection is latency bound and runs two wavefronts
// per SIMD (19.7 KiB of LDS, 187 VGPRs).  The AD rows (17 KiB per node) cross HBM once in between.
//
// Replaces, per node: the quadratic approximation of the tracking cost / EE soft
// constraint / joint-limit and friction-cone barriers (QMInterface.cpp:99-121) and upstream ocs2_sqp's QR constraint projection.
//
// Wave layout: lane c<30 owns entry c of the cost gradients and column c of [C | e]; the Householder QR of D_v^T keeps one column per
// lane in registers and broadcasts reflectors with v_readlane; every dense product runs on v_mfma_f64_16x16x4_f64 with operands
// read from LDS (or assembled in registers) in the lane layout of gpu_rt.h; results leave in accumulator layout.
#pragma once
#include "ad_kernel.h"

namespace qmk {

// LDS of lq_node_kernel (doubles): 19.7 KiB per node, eight nodes per CU = TWO wavefronts per SIMD.  The kernel is latency bound
// (readlane chains, LDS round trips, dependent matrix-core accumulations): at one wavefront per SIMD it ran 1.10 ms per launch.
// What keeps it this small:
//   * Pall = [Px | Pe | 0 | Pu] is stored for its 18 dense joint-velocity rows only; the 12 force rows are unit vectors / pinned
//     values and are synthesised into the matrix-core operands from registers;
//   * R' and Q never enter LDS: the operand / accumulator entries are assembled where they are needed from the constant
//     matrices (global, L1 resident) plus the few barrier terms parked in LDS;
//   * W = R Pall is produced and consumed one 16-column tile at a time;
//   * region X is recycled four times.
constexpr int PAW = 50;                      // row stride of the dense rows of Pall (columns 0..29 Px, 30 Pe, 31 zero, 32..32+m~-1 Pu)
constexpr int CDW = 49;                      // row stride of [C | D_v] (48 used: the 30 state columns and the 18 joint-velocity columns)
constexpr int LDQ = 18, LDY = 34, LDT = 18, LDW = 17;
constexpr int X_DOUBLES = 1152;
constexpr int L_X = 0;                       // X: x u x_next x_ref [4][32] | Q_v [32][LDQ] + Y [16][LDY] | At [32][LDT] + Bt [32][LDT] | W tile [32][LDW]
constexpr int L_XU = L_X, L_QS = L_X, L_YM = L_X + 32 * LDQ, L_AT = L_X, L_BT = L_X + 32 * LDT, L_WT = L_X;
constexpr int L_PA = L_X + X_DOUBLES;        // rows 12..29 of Pall [18][PAW]  /  [C | D_v] [16][CDW]
constexpr int L_EEJ = L_PA + 18 * PAW;       // EE error Jacobian [6][32]
constexpr int L_VEC = L_EEJ + 192;           // b[30] r[30] e[16] eeh[6] (+2) | fin[64] | pe[12] fb[36] ddp[6] ddv[6] (+4)
constexpr int L_RED = L_VEC + 84 + 64 + 64;  // wavefront exchange scratch
constexpr int RED_DOUBLES = 64;
constexpr int LQ_LDS_DOUBLES = L_RED + RED_DOUBLES;
static_assert(16 * CDW <= 18 * PAW && 32 * LDQ + 16 * LDY <= X_DOUBLES && 2 * 32 * LDT <= X_DOUBLES && 32 * LDW <= X_DOUBLES, "aliases must fit");
static_assert(LQ_LDS_DOUBLES * sizeof(real) <= 20480, "eight nodes per CU");

// Both kernels of this file run one wavefront per workgroup: LDS hand-offs between lanes need no hardware barrier (a wavefront's
// LDS operations complete in issue order), only the compiler fence QM_WAVE_SYNC() -- and, unlike __syncthreads(), that does not
// wait for the global stores of the stage record that are still in flight.
// dot product of a broadcast LDS row with a register vector, three independent FMA chains (one wavefront per SIMD: the fp64 FMA
// latency is hidden by instruction-level parallelism only)
__device__ __forceinline__ real dot30(const real* row, const real (&z)[30]) {
  real s0 = 0.0_r, s1 = 0.0_r, s2 = 0.0_r;
#pragma unroll
  for (int i = 0; i < 30; i += 3) { s0 += row[i] * z[i]; s1 += row[i + 1] * z[i + 1]; s2 += row[i + 2] * z[i + 2]; }
  return s0 + s1 + s2;
}

__device__ __forceinline__ real waveSum(real* red, int lane, real v) {
  red[lane] = v;
  QM_WAVE_SYNC();
  real s = 0.0_r;
  for (int i = 0; i < 64; ++i) s += red[i];
  QM_WAVE_SYNC();
  return s;
}

// ---- kernel 2: cost, projection, projected stage record
// QM_LQ_EXTERN (the product build of qmgpu_api.hip): only declared here, defined in qmgpu_lq.hip, which is compiled at -O2 (measured 2.7 % faster; that file says how)
#if defined(QM_LQ_EXTERN) && !defined(QM_RICCATI_TIMING)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) lq_node_kernel(LqArgs a);
#else
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) lq_node_kernel(LqArgs a) {
  __shared__ real lds[LQ_LDS_DOUBLES];
  QM_POISON_LDS(lds, LQ_LDS_DOUBLES);
  const int lane = threadIdx.x;
  const int l16 = lane & 15, h = lane >> 4, la = qmARow(l16);   // la: the row of an A operand this lane supplies (gpu_rt.h)
  const int node = blockIdx.x % (a.N + 1);
  const int inst = blockIdx.x / (a.N + 1);
  if (a.done[inst]) return;
  const bool terminal = node == a.N;
  const ModelR& md = a.P->model;
  const SettingsR& st = a.P->settings;
  const real* bcs = a.Rw + QM_RW_DERIVED;   // barrier constants (layout.h)

  real* PA = lds + L_PA; real* CD = lds + L_PA;
  real* AT = lds + L_AT; real* BT = lds + L_BT; real* WT = lds + L_WT;
  real* EEJ = lds + L_EEJ; real* bv = lds + L_VEC; real* rv = bv + 30; real* ev = rv + 30; real* eeh = ev + 16;
  real* fin = bv + 84; real* pev = fin + 64; real* fb = pev + 12; real* ddp = fb + 36; real* ddv = ddp + 6;
  real* red = lds + L_RED;

  QM_TICK_DECL;
  const real* tg = a.tgrid + size_t(inst) * (a.N + 1);
  const real t = tg[node];
  const real dt = a.dtgrid[size_t(inst) * (a.N + 1) + node];
  const real* xG = a.X + (size_t(inst) * (a.N + 1) + node) * 30;
  const real* uG = terminal ? a.zeros : a.U + (size_t(inst) * a.N + node) * 30;
  // x, u, x_next and the reference state are read many times with wave-uniform indices: one vector load each into LDS instead of
  // chains of dependent scalar loads.  They live in region X and are dead before the QR publishes its factors there.
  real* x = lds + L_XU; real* u = x + 32; real* xnext = x + 64; real* xref = x + 96;
  const Schedule sched{a.schedNum[inst], a.schedTimes + size_t(inst) * QMGPU_MAX_EVENTS, a.schedModes + size_t(inst) * (QMGPU_MAX_EVENTS + 1)};
  const int phase = a.nodePhase[size_t(inst) * (a.N + 1) + node];
  const int mode = sched.modes[phase];
  const real* tTimes = a.targetTimes + size_t(inst) * a.K;
  const real* tStates = a.targetStates + size_t(inst) * a.K * QMGPU_NTARGET;
  const real muP = terminal ? st.ee_final_mu_position : st.ee_mu_position, muO = terminal ? st.ee_final_mu_orientation : st.ee_mu_orientation;
  // force tracking (own formulation): soft constraint 1/2 mu_f |f_e - f_ref|^2 with f_e = -K_e (p_ee - p_env): its Jacobian is -K_e times the
  // position rows of the end-effector error, so it only changes the weights of those rows (Gauss-Newton) and adds to the gradient
  const bool ftOn = a.eeContact && !terminal;
  const real Ke = ftOn ? st.ee_contact_stiffness : 0.0_r, muF = ftOn ? st.ee_force_mu : 0.0_r;

  QM_TICK(0);
  // ---- rows of the AD sweep (ad_node_kernel)
  const real* ad = a.adrows + (size_t(inst) * (a.N + 1) + node) * AD_DOUBLES;
  int nc = 0;
  int zfForce[NCMAX];   // force input a zero-force row pins (swing foot), or -1: those rows are not stored by ad_node_kernel (C = 0, D = unit vector, e = u)
#pragma unroll
  for (int r = 0; r < NCMAX; ++r) zfForce[r] = -1;
  if (!terminal) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (contactOf(mode, k)) nc += 3;
      else {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
          for (int r = 0; r < NCMAX; ++r) if (r == nc + q) zfForce[r] = 3 * k + q;
        }
        nc += 4;
      }
    }
  }
  const int firstStored = contactOf(mode, 0) ? 0 : 3;   // first row ad_node_kernel stored: a stance foot's first velocity row or a swing foot's normal-velocity row
  // force input pinned by row r of the constraint set, or -1 (recomputed from the mode: the debug dump below indexes it with a run-time row)
  auto zeroForceInputOf = [&](int r) {
    int row = 0, res = -1;
    for (int k = 0; k < 4; ++k) {
      if (contactOf(mode, k)) row += 3;
      else { if (r >= row && r < row + 3) res = 3 * k + (r - row); row += 4; }
    }
    return res;
  };
  {
    real cdv[NCMAX], eev[6];   // all global loads in flight before the first LDS store
#pragma unroll
    for (int r = 0; r < NCMAX; ++r) {
      if (zfForce[r] >= 0) cdv[r] = lane == 60 ? uG[zfForce[r]] : 0.0_r;   // wave uniform: no load for a zero-force row
      else cdv[r] = ad[AD_CD + (r < nc ? r : firstStored) * 64 + lane];     // (rows >= nc re-read a row that exists; their values are never used)
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) eev[q] = (lane < 30 || lane >= 60) ? ad[AD_EE + q * 64 + lane] : 0.0_r;   // columns 30..59 of a state-only row are zero and NOT stored (ad_kernel.h: putGlobal): those words of the buffer are undefined and are not loaded
    // [C | D_v]: state columns from lanes 0..29, joint-velocity columns (inputs 12..29) from lanes 42..59, e from lane 60.  The
    // force columns of D are not kept: a zero-force row is a unit vector there, a velocity row is zero (see the projection below).
#pragma unroll
    for (int r = 0; r < NCMAX; ++r) {
      if (r < nc) { if (lane < 30) CD[r * CDW + lane] = cdv[r]; else if (lane >= 42 && lane < 60) CD[r * CDW + lane - 12] = cdv[r]; else if (lane == 60) ev[r] = cdv[r]; }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) { if (lane < 32) EEJ[q * 32 + lane] = eev[q]; else if (lane == 60) eeh[q] = eev[q]; }
  }
  real hf[3];   // f_e - f_ref (column 61 of the position rows, ad_node_kernel)
#pragma unroll
  for (int q = 0; q < 3; ++q) hf[q] = ftOn ? ad[AD_EE + q * 64 + 61] : 0.0_r;
  real phid[12], phiv[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) { phid[i] = ad[AD_PHI + i * 64 + lane]; phiv[i] = ad[AD_PHI + i * 64 + 60]; }

  // (the record stores stay ordinary stores: as streaming stores -- QM_STREAM_STORE, which ad_node_kernel uses for its rows -- they made this kernel slower,
  //  0.656 -> 0.686 ms, and streaming loads of the AD rows as well, 0.656 -> 0.676 ms; measured in round 3)
#ifdef QM_LQ_SAMEREC   // timing experiment only: every node writes the same record (no HBM write traffic)
  real* rec = a.stages + size_t(blockIdx.x & 255) * STAGE_DOUBLES;
#else
  real* rec = a.stages + (size_t(inst) * (a.N + 1) + node) * STAGE_DOUBLES;
#endif
  real* dbg = a.debug ? a.debug + (size_t(inst) * (a.N + 1) + node) * DBG_DOUBLES : nullptr;
  int tIdx; real tAlpha;
  timeSegment(tTimes, a.K, t, tIdx, tAlpha);
  if (lane < 30) {
    x[lane] = xG[lane]; u[lane] = uG[lane]; xnext[lane] = terminal ? xG[lane] : xG[30 + lane];
    xref[lane] = xReference(tStates, a.K, tIdx, tAlpha, lane);
  }
  QM_WAVE_SYNC();
  const int c = lane;
  real qc = 0.0_r, costPart = 0.0_r;
  const real sc = terminal ? 1.0_r : dt;  // intermediate costs are scaled by dt, the terminal cost is not

  QM_TICK(1);
  // ---- state cost, gradient entry c (lanes < 30): tracking + EE soft constraint (Gauss-Newton) + arm joint position soft box.
  //      The Hessian is never formed column-wise: qEntry(i, j) below assembles single entries where they are needed.
  {
    real dd = 0.0_r;
    if (c < 30) {
#pragma unroll
      for (int q = 0; q < 6; ++q) qc += ((q < 3 ? muP : muO) * eeh[q] - (q < 3 ? muF * Ke * hf[q < 3 ? q : 0] : 0.0_r)) * EEJ[q * 32 + c];
      if (c < 6) costPart += 0.5_r * (c < 3 ? muP : muO) * eeh[c] * eeh[c];
      if (c < 3) costPart += 0.5_r * muF * hf[c] * hf[c];
    }
    if (!terminal && c < 30) {
      real Qdx0 = 0.0_r, Qdx1 = 0.0_r;
#pragma unroll
      for (int i = 0; i < 30; i += 2) {
        Qdx0 += st.Q[i * 30 + c] * (x[i] - xref[i]);
        Qdx1 += st.Q[(i + 1) * 30 + c] * (x[i + 1] - xref[i + 1]);
      }
      const real Qdx = Qdx0 + Qdx1;
      qc += Qdx;
      costPart += 0.5_r * (x[c] - xref[c]) * Qdx;
      if (c >= 24) {  // arm joint position soft box (QMInterface.cpp:177-219)
        const Barrier bp{st.joint_pos_barrier_mu, st.joint_pos_barrier_delta};
        const real lo = md.q_lower[c - 12], up = md.q_upper[c - 12];
        const real hl = x[c] - lo, hu = up - x[c];
        real vl_, gl_, hl2, vu_, gu_, hu2;
        bp.eval(hl, bcs[QM_BC_LOGD_POS], vl_, gl_, hl2); bp.eval(hu, bcs[QM_BC_LOGD_POS], vu_, gu_, hu2);
        costPart += vl_ + vu_ - bcs[QM_BC_POS0 + c - 24];
        qc += gl_ - gu_;
        dd = hl2 + hu2;
      }
    }
    if (c >= 24 && c < 30) ddp[c - 24] = dd;
    qc *= sc;
  }
  QM_WAVE_SYNC();
  // entry (i, j) of the state-cost Hessian (unscaled); i, j may be any lane-dependent indices < 32
  // the same without the end-effector Gauss-Newton term (tracking weight + joint-limit barrier): the matrix-core path adds J^T diag(mu) J itself
  auto qBase = [&](int i, int j) {
    const int ic = i < 30 ? i : 0, jc = j < 30 ? j : 0;
    real v = terminal ? 0.0_r : st.Q[ic * 30 + jc];
    const real dg = ddp[ic >= 24 ? ic - 24 : 0];
    if (ic == jc && ic >= 24) v += dg;
    return (i < 30 && j < 30) ? v : 0.0_r;
  };
  auto qEntry = [&](int i, int j) {
    const int ic = i < 30 ? i : 0, jc = j < 30 ? j : 0;
    real v = terminal ? 0.0_r : st.Q[ic * 30 + jc];
#pragma unroll
    for (int q = 0; q < 6; ++q) v += (q < 3 ? muP + muF * Ke * Ke : muO) * EEJ[q * 32 + ic] * EEJ[q * 32 + jc];
    const real dg = ddp[ic >= 24 ? ic - 24 : 0];
    if (ic == jc && ic >= 24) v += dg;
    return (i < 30 && j < 30) ? v : 0.0_r;
  };

  if (terminal) {
    if (c < 30) {
#pragma unroll 6
      for (int i = 0; i < 30; ++i) { const real v = qEntry(i, c); rec[OFF_QT + i * 30 + c] = v; if (dbg) dbg[DBG_Q + i * 30 + c] = v; }
    }
    const real nodeCost = waveSum(red, lane, costPart);
    if (c < 30) { rec[OFF_qt + c] = qc; if (dbg) dbg[DBG_q + c] = qc; }
    if (lane == 0) { real* m = a.metrics + (size_t(inst) * (a.N + 1) + node) * NODE_METRICS; m[0] = nodeCost; m[1] = 0.0_r; m[2] = 0.0_r; m[3] = 0.0_r; }
    return;
  }

  QM_TICK(2);
  // ---- Jacobian of the RK2 map Phi = x + dt/2 (k1 + k2): rows 12.. are x_j + dt v_j exactly, so only the twelve momentum /
  //      base-pose rows of [A | B] are dense (phid).
  if (dbg && c < 60) {
    for (int i = 0; i < 30; ++i) { const real v = i < 12 ? phid[i < 12 ? i : 0] + (c == i ? 1.0_r : 0.0_r) : (c == i ? 1.0_r : 0.0_r) + (c == 30 + i ? dt : 0.0_r); if (c < 30) dbg[DBG_A + i * 30 + c] = v; else dbg[DBG_B + i * 30 + (c - 30)] = v; }
  }
  if (lane == 0) {
    for (int i = 0; i < 12; ++i) bv[i] = x[i] + phiv[i] - xnext[i];
    for (int j = 0; j < 18; ++j) bv[12 + j] = x[12 + j] + dt * u[12 + j] - xnext[12 + j];
  }
  QM_TICK(3);
  // ---- input cost: R' + friction-cone and arm-velocity barriers.  Lane c < 30 forms the gradient entry c and the barrier terms
  //      of its column; the terms go to LDS (fb: four 3x3 friction blocks, ddv: arm-velocity diagonal), R' itself stays in HBM / L1.
  {
    int nStance = 0;
    for (int k = 0; k < 4; ++k) nStance += contactOf(mode, k) ? 1 : 0;
    const real fzNom = nStance > 0 ? md.total_mass * st.gravity / nStance : 0.0_r;
    if (c < 30) {
      real Rdu0 = 0.0_r, Rdu1 = 0.0_r;
#pragma unroll
      for (int i = 0; i < 30; i += 2) {
        const real un0 = (i < 12 && (i % 3) == 2 && contactOf(mode, i / 3)) ? fzNom : 0.0_r;
        const real un1 = (i + 1 < 12 && ((i + 1) % 3) == 2 && contactOf(mode, (i + 1) / 3)) ? fzNom : 0.0_r;
        Rdu0 += a.Rw[i * 30 + c] * (u[i] - un0);
        Rdu1 += a.Rw[(i + 1) * 30 + c] * (u[i + 1] - un1);
      }
      const real Rdu = Rdu0 + Rdu1;
      const real unomc = (c < 12 && (c % 3) == 2 && contactOf(mode, c / 3)) ? fzNom : 0.0_r;
      real rc = Rdu;
      costPart += 0.5_r * (u[c] - unomc) * Rdu;
      if (c >= 24) {  // arm joint velocity soft box (QMInterface.cpp:221-254)
        const Barrier bvel{st.joint_vel_barrier_mu, st.joint_vel_barrier_delta};
        const int i = c - 24;
        const real vl = u[c] - st.arm_vel_lower[i], vu = st.arm_vel_upper[i] - u[c];
        real va_, ga_, ha_, vb_, gb_, hb_;
        bvel.eval(vl, bcs[QM_BC_LOGD_VEL], va_, ga_, ha_); bvel.eval(vu, bcs[QM_BC_LOGD_VEL], vb_, gb_, hb_);
        costPart += va_ + vb_ - bcs[QM_BC_VEL0 + i];
        rc += ga_ - gb_;
        ddv[i] = ha_ + hb_;
      }
      if (c < 12) {
        real e0 = 0.0_r, e1 = 0.0_r, e2 = 0.0_r;
        if (contactOf(mode, c / 3)) {  // friction cone barrier (QMInterface.cpp:344-358), column c % 3 of its 3x3 block
          const Barrier bf{st.friction_barrier_mu, st.friction_barrier_delta};
          const int fo = 3 * (c / 3), ac = c % 3;
          const real fx = u[fo], fy = u[fo + 1], fz = u[fo + 2];
          // F = sqrt(fx^2 + fy^2 + reg) > 0: 1 / F from the reciprocal square root (gpu_rt.h), F = F^2 / F, 1 / F^3 = (1 / F)^3 -- a square root and five divisions before
          const real F2 = fx * fx + fy * fy + st.friction_regularization;
          const real rF = F2 > 0.0_r ? qmRsqrtPos(F2) : 0.0_r, F = F2 * rF, rF3 = rF * rF * rF;
          const real hh = st.friction_coefficient * fz - F;
          const real gx = -fx * rF, gy = -fy * rF, gz = st.friction_coefficient;
          const real hxx = -(fy * fy + st.friction_regularization) * rF3 - st.friction_hessian_shift, hxy = fx * fy * rF3;
          const real hyy = -(fx * fx + st.friction_regularization) * rF3 - st.friction_hessian_shift, hzz = -st.friction_hessian_shift;
          const real gac = ac == 0 ? gx : (ac == 1 ? gy : gz);
          const real h0 = ac == 0 ? hxx : (ac == 1 ? hxy : 0.0_r), h1 = ac == 0 ? hxy : (ac == 1 ? hyy : 0.0_r), h2 = ac == 2 ? hzz : 0.0_r;
          real pv, p1, p2;
          bf.eval(hh, bcs[QM_BC_LOGD_FRIC], pv, p1, p2);
          if (ac == 0) costPart += pv;
          rc += p1 * gac;
          e0 = p2 * gx * gac + p1 * h0; e1 = p2 * gy * gac + p1 * h1; e2 = p2 * gz * gac + p1 * h2;
        }
        const int fo = 3 * (c / 3), ac = c % 3;      // fb[input k][column of its foot's block]
        fb[(fo + 0) * 3 + ac] = e0; fb[(fo + 1) * 3 + ac] = e1; fb[(fo + 2) * 3 + ac] = e2;
      }
      rv[c] = dt * rc;
    }
  }
  QM_WAVE_SYNC();
  // entry (k, i) of the dt-scaled input-cost Hessian, extended by column 30 = r (so that row 30 of W = R Pall is r^T Pall) and a
  // zero column 31; rows >= 30 are zero
  auto rEntry = [&](int k, int i) {
    const int kc = k < 30 ? k : 0, ic = i < 30 ? i : 0;
    real v = a.Rw[kc * 30 + ic];
    const real fbv = fb[(kc < 12 ? kc : 0) * 3 + ic % 3], dv = ddv[kc >= 24 ? kc - 24 : 0];
    if (kc < 12 && ic < 12 && kc / 3 == ic / 3) v += fbv;
    if (kc == ic && kc >= 24) v += dv;
    v *= dt;
    const real rk = rv[kc];
    return k < 30 ? (i < 30 ? v : (i == 30 ? rk : 0.0_r)) : 0.0_r;
  };
  if (dbg && c < 30) {
    for (int i = 0; i < 30; ++i) dbg[DBG_R + i * 30 + c] = rEntry(i, c);
    dbg[DBG_b + c] = bv[c]; dbg[DBG_r + c] = rv[c];
    for (int r = 0; r < nc; ++r) {
      const int zf = zeroForceInputOf(r);
      dbg[DBG_C + r * 30 + c] = zf >= 0 ? 0.0_r : ad[AD_CD + r * 64 + c];
      dbg[DBG_D + r * 30 + c] = zf >= 0 ? (c == zf ? 1.0_r : 0.0_r) : ad[AD_CD + r * 64 + 30 + c];
    }
    if (c < nc) dbg[DBG_e + c] = ev[c];
  }
  real dynSq = 0.0_r, eqSq = 0.0_r;   // lane 0: node metrics (x, u, ... leave LDS below)
  if (lane == 0) {
    for (int i = 0; i < 30; ++i) dynSq += bv[i] * bv[i];
    for (int i = 0; i < nc; ++i) eqSq += ev[i] * ev[i];
  }

  QM_TICK(4);
  // ================================================================== projection: QR of the velocity block of D^T (18 x nv)
  // The constraint set of the reference (QMInterface.cpp:116-131) has a fixed structure: a zero-force row is a unit vector on one
  // force input (C = 0), a zero-velocity / normal-velocity row depends on the inputs only through the 18 joint velocities.  So
  //   D = [E_f 0; 0 D_v],  Q = blkdiag(permutation of the 12 force inputs, Q_v):
  // only D_v^T (18 x nv, nv = 3 n_st + n_sw <= 12) needs the Householder QR; swing-foot forces are pinned to -e_f, stance-foot forces
  // are free directions (unit vectors of Pu).  This is a different (equally orthonormal) basis of the same null space as the generic
  // 30 x nc factorisation -- the reduced stage differs by an orthogonal change of du~, the step du does not.
  // Lane j < 16 owns column j of D_v^T, lane 16 + c column c of the 18 x 18 identity: every reflector is applied to both, so the
  // identity lanes end up with Q_v^T.  The reflector of step k is built by EVERY lane from column k, fetched with v_readlane from
  // lane k: no LDS hand-off, no divergent branch, no barrier inside the factorisation.
  const int nt = 30 - nc;  // projected input dimension m~
  constexpr int NVMAX = 12;
  int nv = 0, nStF = 0;     // velocity rows; free (stance) force inputs
  int vrOf[NVMAX];          // CD row of velocity row r (wave uniform)
  int frcRowOf[12];         // CD row pinning force input i (swing foot), or -1 (stance foot: free)
  int puColOf[12];          // Pu column of a free force input, or -1
#pragma unroll
  for (int r = 0; r < NVMAX; ++r) vrOf[r] = 0;
  {
    int row = 0;
#pragma unroll
    for (int cft = 0; cft < 4; ++cft) {
      if (contactOf(mode, cft)) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
          for (int r = 0; r < NVMAX; ++r) if (r == nv + q) vrOf[r] = row + q;
          frcRowOf[3 * cft + q] = -1; puColOf[3 * cft + q] = nStF + q;
        }
        nv += 3; nStF += 3; row += 3;
      } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) { frcRowOf[3 * cft + q] = row + q; puColOf[3 * cft + q] = -1; }
#pragma unroll
        for (int r = 0; r < NVMAX; ++r) if (r == nv) vrOf[r] = row + 3;
        nv += 1; row += 4;
      }
    }
  }
  QM_TICK(5);
  // Pall operand of the matrix cores, element (k, j) with k = 4 ks + h (this lane's row of k step ks) and j = 16 tq + l16:
  // rows k < 12 (force inputs, k steps 0..2) are synthesised -- Pe = pinned swing force in column 30, a unit entry in the Pu column
  // of a free stance force --, rows 12..29 come from LDS, rows 30 and 31 are zero.
  real peK[3]; int puK[3];
  {
    real peForce = 0.0_r;   // pinned swing-foot forces: Pe = -e_f
#pragma unroll
    for (int i = 0; i < 12; ++i) if (lane == i && frcRowOf[i] >= 0) peForce = -ev[frcRowOf[i] >= 0 ? frcRowOf[i] : 0];
    if (lane < 12) pev[lane] = peForce;
    QM_WAVE_SYNC();
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int kk = 4 * ks + h;
      peK[ks] = pev[kk];
      int pu = -1;
#pragma unroll
      for (int i = 0; i < 12; ++i) if (kk == i) pu = puColOf[i];
      puK[ks] = pu;
    }
  }
  auto pallOpAt = [&](int ks, int j) {
    if (ks < 3) return j == 30 ? peK[ks] : ((j >= 32 && j - 32 == puK[ks]) ? 1.0_r : 0.0_r);
    const int kk = 4 * ks + h;
    const real raw = PA[((kk < 30 ? kk : 12) - 12) * PAW + (j < PAW ? j : 0)];
    return (kk < 30 && j < PAW) ? raw : 0.0_r;
  };
  auto pallOp = [&](int ks, int tq) { return pallOpAt(ks, tq * 16 + l16); };    // as a B operand: column of Pall
  auto pallOpT = [&](int ks, int tq) { return pallOpAt(ks, tq * 16 + la); };    // as an A operand (Pall^T): row of Pall^T
  // The eight operands of one tile column / row at once, held with QM_KEEP: left inside the multiply loops the compiler turned every select above into a
  // predicated LDS read of its own (s_and_saveexec; ds_read; s_waitcnt lgkmcnt(0)) directly in front of the matrix-core instruction that uses it -- one LDS
  // round trip per instruction (round 3: 112 of them in the products (2)(3) alone).
  auto pallOps = [&](real (&o)[8], int tq) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) o[ks] = pallOp(ks, tq);
#pragma unroll
    for (int ks = 3; ks < 8; ++ks) QM_KEEP(o[ks]);
  };
  auto pallOpsT = [&](real (&o)[8], int tq) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) o[ks] = pallOpT(ks, tq);
#pragma unroll
    for (int ks = 3; ks < 8; ++ks) QM_KEEP(o[ks]);
  };
  {
    int myVr = 0;
#pragma unroll
    for (int r = 0; r < NVMAX; ++r) if (lane == r) myVr = vrOf[r];
    real ce[NVMAX];   // my column of [C_v | e_v] (lanes <= 30)
#pragma unroll
    for (int r = 0; r < NVMAX; ++r) {
      const real cv = CD[vrOf[r] * CDW + (lane < 30 ? lane : 0)], evr = ev[vrOf[r]];
      ce[r] = (r < nv) ? (lane < 30 ? cv : (lane == 30 ? evr : 0.0_r)) : 0.0_r;
    }
    real qcol[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const real dv = CD[myVr * CDW + 30 + i];
      qcol[i] = lane < 16 ? (lane < nv ? dv : 0.0_r) : ((lane < 34 && i == lane - 16) ? 1.0_r : 0.0_r);
    }
    QM_WAVE_SYNC();  // the [C D_v] region is free from here on (it becomes Pall); so is x | u | x_next | x_ref in region X
#pragma unroll
    for (int k = 0; k < NVMAX; ++k) {
      if (k < nv) {
        real v[18];
        real n2a = 0.0_r, n2b = 0.0_r;
#pragma unroll
        for (int i = k; i < 18; ++i) { v[i] = qmReadLane(qcol[i], k, red); if ((i - k) & 1) n2b += v[i] * v[i]; else n2a += v[i] * v[i]; }
        // reflector v = column - alpha e_k, alpha = -sign(d_k) |column|; beta = 2 / v^T v = 1 / (|column| |v_k|): one reciprocal square root and one
        // reciprocal on the dependent chain of the step (a square root and an IEEE division before: ~28 dependent instructions)
        const real dk = v[k], n2 = n2a + n2b;
        const real rs = n2 > 0.0_r ? qmRsqrtPos(n2) : 0.0_r;
        const real nrm = n2 * rs;
        const real alpha = dk > 0.0_r ? -nrm : nrm;
        v[k] = dk - alpha;
        const real avk = fabs(v[k]);
        const real beta = avk > 0.0_r ? rs * qmRcpPos(avk) : 0.0_r;
        real sa = 0.0_r, sb = 0.0_r;
#pragma unroll
        for (int i = k; i < 18; ++i) { if ((i - k) & 1) sb += v[i] * qcol[i]; else sa += v[i] * qcol[i]; }
        const real sf = (sa + sb) * beta;
#pragma unroll
        for (int i = k; i < 18; ++i) qcol[i] -= sf * v[i];
        qcol[k] = lane == k ? alpha : qcol[k];   // R1[k][k] exactly; the entries below it in lane k (rounding residue instead of zeros) are never read:
                                                 // R1[r][c], r <= c, lives in lane c register r, later reflectors touch entries >= their own index only
      }
    }
    // Y = R1^-T [C_v | e_v]: forward substitution, R1[k][i] (k <= i) lives in lane i, register k
    real y[NVMAX];
    real rinv;   // 1 / R1[i][i] in lane i: ONE division for all pivots instead of one on the dependent chain of every row
    {
      real mine = 1.0_r;
#pragma unroll
      for (int i = 0; i < NVMAX; ++i) mine = (lane == i && i < nv) ? qcol[i] : mine;
      rinv = 1.0_r / mine;
    }
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) {
      y[i] = 0.0_r;
      if (i < nv) {   // (wave uniform: trot has 8 velocity rows, stance 12)
        real sacc = ce[i];
#pragma unroll
        for (int k = 0; k < i; ++k) sacc -= qmReadLane(qcol[k], i, red) * y[k];
        y[i] = sacc * qmReadLane(rinv, i, red);
      }
    }
    // publish Y (rows k < 16, my column) and Q_v (lane 16 + c holds row c; rows 18..31 cleared) in region X
    real* Ym = lds + L_YM;
    real* Qs = lds + L_QS;
    if (lane < 32) {
#pragma unroll
      for (int k = 0; k < NCMAX; ++k) Ym[k * LDY + lane] = (k < NVMAX && lane <= 30) ? y[k < NVMAX ? k : 0] : 0.0_r;
    }
    if (lane >= 16 && lane < 48) {
#pragma unroll
      for (int r = 0; r < 18; ++r) Qs[(lane - 16) * LDQ + r] = lane < 34 ? qcol[r] : 0.0_r;
    }
    QM_WAVE_SYNC();
    // [Px | Pe] rows 12..29 = -Q_v1 Y on the matrix cores (K = 16 >= nv; rows of Y beyond nv are zero)
    QmAcc pc[4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
      for (int r = 0; r < 4; ++r) pc[t4][r] = 0.0_r;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = 4 * ks + h;
      const real a0 = -Qs[la * LDQ + kk], a1 = -Qs[(16 + la) * LDQ + kk];
      const real b0 = Ym[kk * LDY + l16], b1 = Ym[kk * LDY + 16 + l16];
      qmMfma(pc[0], a0, b0, red); qmMfma(pc[1], a0, b1, red); qmMfma(pc[2], a1, b0, red); qmMfma(pc[3], a1, b1, red);
    }
    // Pall: rows 0..11 (forces): Px = 0, Pu = unit columns of the free stance forces -- neither is stored (layout.h) --, Pe = pinned swing forces;
    //       rows 12..29 (joint velocities): [Px | Pe] from the tiles, Pu = Q_v2 (record and LDS)
    if (lane < 12) rec[OFF_PE + lane] = pev[lane];
    else if (lane == 30 || lane == 31) rec[OFF_PE + lane] = lane == 30 ? real(mode) : dt;   // OFF_MODE: the consumers rebuild the force rows of Pu from it; OFF_DT
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = (t4 >> 1) * 16 + h + 4 * r, j = (t4 & 1) * 16 + l16;   // i: joint-velocity input 12 + i
        const int off = (i < 18 && j <= 30) ? (j < 30 ? offPxRow(12 + i) + j : OFF_PE + 12 + i) : -1;   // Px rows live in the A~ area (layout.h)
        if (i < 18) PA[i * PAW + j] = j <= 30 ? pc[t4][r] : 0.0_r;
        if (off >= 0) rec[off] = pc[t4][r];
      }
    if (lane >= 32 && lane < PAW) {
      const int jj = lane - 32;
      const bool isQ2 = jj >= nStF && jj < nt;
      real qv[18];   // all reads first (QM_KEEP: otherwise every read is predicated on isQ2 and waited for in front of its two stores)
#pragma unroll
      for (int i = 0; i < 18; ++i) qv[i] = Qs[i * LDQ + (isQ2 ? nv + (jj - nStF) : 0)];
#pragma unroll
      for (int i = 0; i < 18; ++i) QM_KEEP(qv[i]);
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        const real v = isQ2 ? qv[i] : 0.0_r;
        PA[i * PAW + lane] = v;
        rec[offPuRow(12 + i) + jj] = v;                                             // (the B~ area) jj < MT = PAW - 32: zero padding included
      }
    }
  }
  const bool isX = lane < 30, isU = lane >= 32 && lane < 32 + nt;
  QM_WAVE_SYNC();   // Pall complete; Q_v / Y (region X) are dead
  QM_TICK(6);
  // rows 12..29 of [A~ | b~ | B~] = [I | b | 0] + dt Pall (the joint rows of B are dt * identity): A~ and B~ rows are NOT stored -- riccati_kernel forms them
  // from the Px / Pu rows above (layout.h) --, only b~
  if (lane >= 12 && lane < 30) rec[OFF_bt + lane] = bv[lane] + dt * PA[(lane - 12) * PAW + 30];
  if (lane == 32 || lane == 33) {   // the neighbours' steps for riccati_kernel (layout.h)
    const int nb = lane == 32 ? node - 1 : node + 1;
    rec[lane == 32 ? OFF_DTPREV : OFF_DTNEXT] = (nb >= 0 && nb < a.N) ? a.dtgrid[size_t(inst) * (a.N + 1) + nb] : 0.0_r;
  }
  {  // transposed dense rows: At[j][i] = A[i][j], Bt[k][i] = B[i][k], i < 12 (columns 12..15 and rows 30,31 zero)
    real* dst = (lane < 30) ? AT + lane * LDT : (lane < 60 ? BT + (lane - 30) * LDT : AT + 30 * LDT + (lane - 60) * LDT);
    const bool pad = lane >= 60;  // lanes 60..61 clear rows 30,31 of At; lanes 62..63 rows 30,31 of Bt
    if (lane >= 62) dst = BT + (30 + lane - 62) * LDT;
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[i] = (!pad && i < 12) ? phid[i < 12 ? i : 0] + ((lane < 30 && lane == i) ? 1.0_r : 0.0_r) : 0.0_r;
  }
  QM_WAVE_SYNC();

  QM_TICK(7);
  // ================================================================== products on the fp64 matrix cores
  // (1) rows 0..11 of [A~ | b~ | B~] = [A | b | 0] + B Pall
  const int nTn = nt > 16 ? 4 : 3;   // 16-column tiles of Pall in use
  {
    QmAcc c1[4];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = h + 4 * r, j = tn * 16 + l16;
        const real av = AT[(j < 32 ? j : 0) * LDT + i], bb = bv[i < 30 ? i : 0];
        c1[tn][r] = (i < 12) ? (j < 30 ? av : (j == 30 ? bb : 0.0_r)) : 0.0_r;
      }
    real ab[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) ab[ks] = BT[(4 * ks + h) * LDT + la];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      if (tn < nTn) {
        real pb[8];
        pallOps(pb, tn);
        // columns 0..15 of the force rows of Pall (k steps 0..2) are zero: Px = 0 there, Pe is column 30, Pu starts at column 32
#pragma unroll
        for (int ks = (tn == 0 ? 3 : 0); ks < 8; ++ks) qmMfma(c1[tn], ab[ks], pb[ks], red);
      }
    }
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      if (tn < nTn) {
        const int j = tn * 16 + l16;
#pragma unroll
        for (int r = 0; r < 3; ++r) {   // (r = 3: rows 12..15 of the tile do not exist)
          const int i = h + 4 * r;
          // ONE store per accumulator register, its place chosen with selects (nested conditions became nested predicated regions, each with its own branch)
          const int off = j < 30 ? OFF_AT + i * 30 + j : (j == 30 ? OFF_bt + i : ((j >= 32 && j < 32 + nt) ? OFF_BT + i * MT + (j - 32) : -1));
          if (off >= 0) rec[off] = c1[tn][r];
        }
      }
    }
  }
  QM_WAVE_SYNC();   // every lane has consumed At / Bt: region X becomes the W tile
  const real nodeCost = dt * waveSum(red, lane, costPart);
  if (lane == 0) {
    real* m = a.metrics + (size_t(inst) * (a.N + 1) + node) * NODE_METRICS;
    m[0] = nodeCost; m[1] = dt * dynSq; m[2] = dt * eqSq; m[3] = 0.0_r;
  }
  if (dbg && c < 30) { for (int i = 0; i < 30; ++i) dbg[DBG_Q + i * 30 + c] = sc * qEntry(i, c); dbg[DBG_q + c] = qc; }

  QM_TICK(8);
  // (2) W = R Pall and (3) G = Pall^T W, [Q~ | P~^T; P~ | R~] = [Q | 0; 0 | 0] + G, one 16-column tile of W at a time: the tile goes
  // through LDS (accumulator layout -> operand layout) and is consumed by the tiles (tm, tn) of G that the record needs:
  //   tn = 0, 1: tm = 0, 1 (state block, row 30 carries Pe^T R Px), 2 and 3 (P~; tm = 3 only when m~ > 16)
  //   tn = 2, 3: tm = 2, 3 (R~; column 30 of tn = 1 carries Pu^T R Pe)
  // Row 30 of W is r^T Pall; together with row / column 30 of G it completes q~ and r~ (fin).
  {
    fin[lane] = 0.0_r;
    real r0[8], r1[8];   // R operand (symmetric: R[i][k] read as R[k][i]), shared by all tiles
    // the eight entries (4 ks + h, i) of one operand column at once: all loads (R' from L1, the barrier terms from LDS) before the first use -- written as
    // rEntry calls each entry waited for its own global load inside a predicated region
    auto rEntries = [&](real (&o)[8], int i) {
      const int ic = i < 30 ? i : 0;
      real raw[8], fbv[8], dvv[8], rkv[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int kk = 4 * ks + h, kc = kk < 30 ? kk : 0;
        raw[ks] = a.Rw[kc * 30 + ic]; fbv[ks] = fb[(kc < 12 ? kc : 0) * 3 + ic % 3]; dvv[ks] = ddv[kc >= 24 ? kc - 24 : 0]; rkv[ks] = rv[kc];
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) { QM_KEEP(raw[ks]); QM_KEEP(fbv[ks]); QM_KEEP(dvv[ks]); QM_KEEP(rkv[ks]); }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int kk = 4 * ks + h, kc = kk < 30 ? kk : 0;
        const real add = (kc < 12 && ic < 12 && kc / 3 == ic / 3) ? fbv[ks] : ((kc == ic && kc >= 24) ? dvv[ks] : 0.0_r);   // (the two cases exclude each other)
        const real v = (raw[ks] + add) * dt;
        o[ks] = kk < 30 ? (i < 30 ? v : (i == 30 ? rkv[ks] : 0.0_r)) : 0.0_r;
      }
    };
    rEntries(r0, la);
    rEntries(r1, 16 + la);
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      if (tn < nTn) {
        QmAcc wA, wB;
#pragma unroll
        for (int r = 0; r < 4; ++r) { wA[r] = 0.0_r; wB[r] = 0.0_r; }
        {
          real pb[8];
          pallOps(pb, tn);
#pragma unroll
          for (int ks = (tn == 0 ? 3 : 0); ks < 8; ++ks) { qmMfma(wA, r0[ks], pb[ks], red); qmMfma(wB, r1[ks], pb[ks], red); }   // (force rows of Pall: zero in columns 0..15)
        }
        QM_WAVE_SYNC();   // the previous tile's readers are done
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int i = h + 4 * r; WT[i * LDW + l16] = wA[r]; WT[(16 + i) * LDW + l16] = wB[r]; }
        if (h == 2) fin[tn * 16 + l16] += wB[3];  // row 30 of W
        QM_WAVE_SYNC();
        const bool top = tn < 2;         // tiles tm = 0, 1
        const bool low3 = nt > 16;       // tile tm = 3
        QmAcc g[4];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int r = 0; r < 4; ++r) g[tm][r] = 0.0_r;
        if (tn < 2) {   // + sc * (tracking weight + joint-limit barrier): the eight entries of this lane, loads first (as rEntries above)
          const int j = tn * 16 + l16, jc = j < 30 ? j : 0;
          real qraw[8], dgv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int i = (e >> 2) * 16 + h + 4 * (e & 3), ic = i < 30 ? i : 0;
            qraw[e] = st.Q[ic * 30 + jc]; dgv[e] = ddp[ic >= 24 ? ic - 24 : 0];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) { QM_KEEP(qraw[e]); QM_KEEP(dgv[e]); }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int i = (e >> 2) * 16 + h + 4 * (e & 3), ic = i < 30 ? i : 0;
            const real v = qraw[e] + ((ic == jc && ic >= 24) ? dgv[e] : 0.0_r);
            g[e >> 2][e & 3] = (i < 30 && j < 30) ? sc * v : 0.0_r;
          }
        }
        if (tn < 2) {   // + J_ee^T diag(mu) J_ee (Gauss-Newton term of the end-effector soft constraints) on the matrix cores: K = 6 error rows in two k steps
          real ej[2], e0[2], e1[2];
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int qq = 4 * ks + h, qc2 = qq < 6 ? qq : 0;     // error row this lane supplies
            ej[ks] = EEJ[qc2 * 32 + tn * 16 + l16]; e0[ks] = EEJ[qc2 * 32 + la]; e1[ks] = EEJ[qc2 * 32 + 16 + la];
          }
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) { QM_KEEP(ej[ks]); QM_KEEP(e0[ks]); QM_KEEP(e1[ks]); }
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int qq = 4 * ks + h;
            const bool on = qq < 6;
            const real wq = on ? sc * (qq < 3 ? muP + muF * Ke * Ke : muO) : 0.0_r;
            qmMfma(g[0], wq * e0[ks], on ? ej[ks] : 0.0_r, red); qmMfma(g[1], wq * e1[ks], on ? ej[ks] : 0.0_r, red);
          }
        }
        real wb[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wb[ks] = WT[(4 * ks + h) * LDW + l16];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
          if ((tm < 2 && top) || tm == 2 || (tm == 3 && low3)) {
            real pa[8];
            pallOpsT(pa, tm);
            // rows 0..15 of Pall^T (columns of Px) have no force-row entries: k steps 0..2 multiply zeros
#pragma unroll
            for (int ks = (tm == 0 ? 3 : 0); ks < 8; ++ks) qmMfma(g[tm], pa[ks], wb[ks], red);
          }
        }
        const int j = tn * 16 + l16;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
          if ((tm < 2 && top) || tm == 2 || (tm == 3 && low3)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = tm * 16 + h + 4 * r;
              const real v = g[tm][r];
              // ONE record store per accumulator register, its place chosen with selects (tm, tn, r are compile-time; nested conditions had become nested
              // predicated regions with a branch each).  Q~ is symmetric and the backward sweep reads the tile (0,1) and the upper triangles of (0,0), (1,1)
              // only: nothing else is stored; R~: the lower triangle (the backward sweep reads the mirror image for the rest).
              int off = -1;
              if (tm < 2) { if (tn < 2 && !(tm == 1 && tn == 0)) off = (i < 30 && j < 30 && !(tm == tn && j < i)) ? OFF_QT + i * 30 + j : -1; }
              else if (tn < 2) off = (i < 32 + nt && j < 30) ? OFF_PT + (i - 32) * 30 + j : -1;
              else off = (i < 32 + nt && j <= i) ? OFF_RT + (i - 32) * MT + (j - 32) : -1;
              if (off >= 0) rec[off] = v;
              if (tm == 1 && r == 3) { if (i == 30 && j < 30) fin[j] += v; }                        // row 30: Pe^T R Px
              if (tm >= 2 && tn == 1) { if (j == 30 && i < 32 + nt) fin[i] += v; }                  // column 30: Pu^T R Pe
            }
          }
        }
      }
    }
  }
  QM_WAVE_SYNC();
  if (isX) rec[OFF_qt + lane] = qc + fin[lane];
  else if (isU) rec[OFF_rt + (lane - 32)] = fin[lane];
  QM_TICK(9);
  // zero padding of B~, Pu (above) and r~ beyond m~ columns: the forward sweep of riccati_kernel multiplies whole MT-wide rows
  // (P~, R~ keep undefined padding: their consumer masks by m~)
  if (nt < MT) {
    if (lane >= 32 + nt && lane < 32 + MT) rec[OFF_rt + (lane - 32)] = 0.0_r;
    if (lane < 48) { const int i = lane >> 2, jj = nt + (lane & 3); if (jj < MT) rec[OFF_BT + i * MT + jj] = 0.0_r; }        // rows 0..11 of B~
  }
  QM_TICK(10);
  QM_TICK_FLUSH(128, blockIdx.x == 7);
}
#endif   // QM_LQ_EXTERN

}  // namespace qmk
