// ad_node_kernel -- derivatives of the RK2 shooting map and of the state-input constraints, THREE shooting nodes per wavefront.
//
// Replaces, per node: QMDynamicsAD::linearApproximation for both RK2 stages
// (qm_interface/src/dynamics/QMDynamicsAD.cpp:30-33, CppAD), QMPreComputation::request (QMPreComputation.cpp:50-89), the Jacobians of
// the per-foot equality constraints (QMInterface.cpp:123-131) and of the end-effector error (EndEffectorConstraint.cpp:36-78), and
// upstream ocs2_sqp's sensitivity discretisation  A_d = I + dt/2 (A1 + A2 (I + dt A1)),  B_d = dt/2 (B1 + A2 dt B1 + B2).
//
// Structure of the flow map f(x, u), x = [h/m (6); p (3); zyx (3); q_j (18)], u = [F (12); v_j (18)]:
//   * f does not depend on the base position p at all (translation invariance);
//   * f is LINEAR in the momentum h, the joint rates v_j and the contact forces F;
//   * only the 21 configuration coordinates (zyx, q_j) enter non-linearly.
// So a lane carries ONE configuration tangent through the tree sweep (Du, slot d) and, in the quantities that are linear in a
// velocity-like argument, a second tangent (Du3, slot e) that only ever multiplies configuration values:
//     lane direction dd in [0, 21):   slot d = d/d zyx_dd (dd < 3) or d/d q_{dd-3};
//                                     slot e = d/d h_ang_dd (dd < 3) or d/d v_{j, dd-3} in the kinematic rows,
//                                              d/d F_{dd-6} (6 <= dd < 18) in the momentum-rate rows;
//                                     the remaining columns (h_lin, p, the value itself) are closed forms written by lanes 0..5, 18.
// 21 lanes cover all 60 columns of one node: a wavefront differentiates three nodes side by side (lanes 0..20, 21..41, 42..62) with
// the instruction count one node used to take with sixty full tangents.  Both RK2 stages are differentiated at their own arguments
// (partials only); the chain rule that the sixty-lane version carried through the second sweep is the product
//     d k2 / d (x, u) = J2 + dt J2[:, 0:12] J1 + dt [0 | J2[:, q_j] -> v_j columns]
// whose dense part runs on the fp64 matrix cores (12 x 12 x 64 per node).
//
// Output: the AD rows of lq_kernel.h (one 64-double row per differentiated scalar: entries < 60 the derivative along (x, u), entry
// 60 the value) -- unchanged, lq_node_kernel does not know how they were produced.
#pragma once
#include "layout.h"
#include "schedule_dev.h"
#include "sweep_dev.h"

namespace qmk {

struct LqArgs {
  const ProblemR* P;
  const real* Rw;          // R' [30][30]
  int batch, N, K;
  const real* tgrid;       // [batch][N+1] node times, the step to the next node (0 at the terminal node) and the phase of the mode schedule
  const real* dtgrid;      //              the node lies in -- all three formed by mpc_init_kernel from the fp64 times
  const int* nodePhase;
  const real* X;           // [batch][N+1][30] current iterate
  const real* U;           // [batch][N][30]
  const real* targetTimes; // [batch][K]
  const real* targetStates;// [batch][K][37]
  const int* schedNum;       // [batch]
  const real* schedTimes;  // [batch][MAX_EVENTS]
  const int* schedModes;     // [batch][MAX_EVENTS+1]
  const real* zeros;       // >= 64 zeros
  real* stages;            // [batch][N+1][STAGE_DOUBLES]
  int* stageNc;              // [batch][N+1]
  int* nodeMode;             // [batch][N+1]
  real* metrics;           // [batch][N+1][NODE_METRICS]
  real* debug;             // [batch][N+1][DBG_DOUBLES] or null
  real* adrows;            // [batch][N+1][AD_DOUBLES]: ad_node_kernel -> lq_node_kernel
  const int* done;           // [batch] instances whose SQP iterations have converged are skipped
  const real* eeContact;     // [batch][K][6] or null: force tracking (qmgpu_mpc_args::ee_contact_ref)
};

// AD rows: one 64-double row per differentiated scalar; entry l < 60 = d/d(x,u)_l, entry 60 = the value itself
constexpr int AD_PHI = 0;                    // [12] RK2 increment phi = dt/2 (k1 + k2) of the momentum / base-pose states
constexpr int AD_CD = AD_PHI + 12 * 64;      // [16] equality constraint rows
constexpr int AD_EE = AD_CD + 16 * 64;       // [6]  end-effector pose error
constexpr int AD_DOUBLES = AD_EE + 6 * 64;   // 2176

constexpr int AD_DIRS = 21;                  // configuration directions = lanes per node
constexpr int AD_NODES = 3;                  // nodes per wavefront
__host__ __device__ constexpr int adGridFor(int nodes) { return (nodes + AD_NODES - 1) / AD_NODES; }

// Inputs of the sweep as seen by one lane: values from the node's x | u staged in LDS, seeds from the lane's direction.
struct AdIn {
  const real* x;    // this node's state (30) ...
  const real* u;    // ... and input (30) in LDS
  const real* xs;   // the twelve momentum / base-pose states the stage is evaluated at: x (first stage) or x + dt k1 (second)
  int dd;             // direction 0..20
  real dtS;         // 0 (first stage) or dt (second stage: q_j + dt v_j)
  __device__ __forceinline__ Du3 hn(int i) const { return Du3(xs[i], 0.0_r, (i >= 3 && dd == i - 3) ? 1.0_r : 0.0_r); }
  __device__ __forceinline__ Du euler(int i) const { return Du(xs[9 + i], dd == i ? 1.0_r : 0.0_r); }
  __device__ __forceinline__ Du q(int j) const { return Du(fma(dtS, u[12 + j], x[12 + j]), dd == 3 + j ? 1.0_r : 0.0_r); }
  __device__ __forceinline__ Du3 qd(int j) const { return Du3(u[12 + j], 0.0_r, dd == 3 + j ? 1.0_r : 0.0_r); }
  __device__ __forceinline__ Vec3<Du3> force(int c) const {
    return Vec3<Du3>(Du3(u[3 * c], 0.0_r, dd == 6 + 3 * c ? 1.0_r : 0.0_r), Du3(u[3 * c + 1], 0.0_r, dd == 7 + 3 * c ? 1.0_r : 0.0_r), Du3(u[3 * c + 2], 0.0_r, dd == 8 + 3 * c ? 1.0_r : 0.0_r));
  }
};

// LDS of one wavefront (doubles): 37 KiB, four wavefronts (one per SIMD: the sweep needs the whole register file) per CU
constexpr int ADL_XU = 0;                              // [3][64]  x (0..29) | u (32..61) per node
constexpr int ADL_X2 = ADL_XU + AD_NODES * 64;         // [3][12]  x + dt k1, momentum / base-pose part
constexpr int ADL_A2 = ADL_X2 + AD_NODES * 12;         // [3][12][16] J2[:, 0:12] (operand of the chain-rule product), columns 12..15 zero
constexpr int ADL_PARK = ADL_A2 + AD_NODES * 12 * 16;  // feet of the first stage [4][15][64]  /  J1 then J1 + J2 + ... [3][12][64]
constexpr int AD_PARK_DOUBLES = 4 * 15 * 64;
constexpr int ADL_PUB = ADL_PARK + AD_PARK_DOUBLES;    // [3][119] primal composites of the five kinematic chains of each node (sweep_dev.h: centroidalSweepOwnChain)
constexpr int AD_LDS_DOUBLES = ADL_PUB + AD_NODES * SWEEP_PUB_NODE;
static_assert(AD_NODES * 12 * 64 <= AD_PARK_DOUBLES, "the Jacobian rows reuse the parking area");
static_assert(AD_LDS_DOUBLES * sizeof(real) <= 40960, "four wavefronts per CU");

#ifndef QM_LQ_UNIT   // (qmgpu_lq.hip includes this file for LqArgs and the row layout only)
__global__ void __launch_bounds__(64) QM_ONE_WAVE_PER_SIMD ad_node_kernel(LqArgs a) {
  __shared__ real lds[AD_LDS_DOUBLES];
  QM_POISON_LDS(lds, AD_LDS_DOUBLES);
  const int lane = threadIdx.x;
  const int l16 = lane & 15, h = lane >> 4, la = qmARow(l16);   // la: the row of an A operand this lane supplies (gpu_rt.h)
  const int grp = lane / AD_DIRS < AD_NODES ? lane / AD_DIRS : AD_NODES - 1;   // lane 63 shadows the last lane of node 2 (stores nothing)
  const int dd = qmOpaqueLane(lane < AD_NODES * AD_DIRS ? lane - grp * AD_DIRS : AD_DIRS - 1);
  const int total = a.batch * (a.N + 1);
  const int gRaw = blockIdx.x * AD_NODES + grp;
  const int gnode = gRaw < total ? gRaw : total - 1;
  const int node = gnode % (a.N + 1), inst = gnode / (a.N + 1);
  const bool live = lane < AD_NODES * AD_DIRS && gRaw < total && !a.done[inst];   // this lane's results reach HBM
  const bool terminal = node == a.N;
  // (QM_CONSTANT_REF -- scalar loads for the model constants, gpu_rt.h -- was measured here and NOT kept: 57 s_load instead of 353 vector loads per
  //  wavefront, but the scalar results do not fit the SGPR file of a kernel that already sits at 512 VGPRs: 243 instead of 64 VGPR spills, 0.50 -> 0.72 ms)
  const ModelR& md = a.P->model;
  const SettingsR& st = a.P->settings;
  // column of the AD row each of this lane's three values goes to
  const int cD = 9 + dd;                                                                   // zyx / q_j
  const int cV = dd < 3 ? 3 + dd : 39 + dd;                                                // h_ang / v_j
  const int cC = dd < 3 ? dd : (dd < 6 ? 3 + dd : (dd < 18 ? 24 + dd : (dd == 18 ? 60 : 42 + dd)));   // h_lin | p | F | value | padding 61, 62
  const bool isF = dd >= 3 && dd < 18, isVal = dd == 18;   // lanes whose closed-form column is a force-type slot (p only matters with the EE contact)

  real* ad = a.adrows + size_t(gnode) * AD_DOUBLES;
  real* xu = lds + ADL_XU + grp * 64;
  real* x2 = lds + ADL_X2 + grp * 12;
  real* A2 = lds + ADL_A2 + grp * 192;
  real* park = lds + ADL_PARK + lane;                 // lane-private columns while the feet wait for the base twist
  real* L1 = lds + ADL_PARK + grp * 768;              // [12][64] Jacobian rows of this node

  const real* tg = a.tgrid + size_t(inst) * (a.N + 1);
  const real t = tg[node];
  const real dt = a.dtgrid[gnode];
  {  // x | u of the three nodes: lanes dd load 30 + 30 values of their node
    const real* xG = a.X + size_t(gnode) * 30;
    const real* uG = terminal ? a.zeros : a.U + (size_t(inst) * a.N + node) * 30;
    for (int i = dd; i < 30; i += AD_DIRS) { xu[i] = xG[i]; xu[32 + i] = uG[i]; }
  }
  QM_WAVE_SYNC();
  const real* x = xu; const real* u = xu + 32;
  const Schedule sched{a.schedNum[inst], a.schedTimes + size_t(inst) * QMGPU_MAX_EVENTS, a.schedModes + size_t(inst) * (QMGPU_MAX_EVENTS + 1)};
  const int phase = a.nodePhase[gnode];
  const int mode = sched.modes[phase];
  real eePosRef[3], eeQuatRef[4];
  eeReference(a.targetTimes + size_t(inst) * a.K, a.targetStates + size_t(inst) * a.K * QMGPU_NTARGET, a.K, t, eePosRef, eeQuatRef);
  // force tracking (own formulation): compliant environment at the end-effector, f_e = -K_e (p_ee - p_env), intermediate nodes only
  real Ke = 0.0_r, fRef[3] = {0.0_r, 0.0_r, 0.0_r}, pEnv[3] = {0.0_r, 0.0_r, 0.0_r};
  if (a.eeContact && !terminal) { Ke = st.ee_contact_stiffness; eeContactReference(a.targetTimes + size_t(inst) * a.K, a.eeContact + size_t(inst) * a.K * 6, a.K, t, fRef, pEnv); }

  // one row of the AD format: the lane's configuration slot, its velocity / force slot and its closed-form column
  const bool owner = lane < AD_NODES * AD_DIRS;   // lane 63 computes along with the others but owns no column
  auto putRow = [&](real* dst, int row, real dval, real vval, real cval) {
    real* r = dst + row * 64;
    if (owner) { r[cD] = dval; r[cV] = vval; r[cC] = cval; }
  };
  // rows that leave for HBM: streaming (non-temporal) stores -- 51 KB per workgroup that nothing on this CU reads again must not push the 2.9 KB of model
  // constants out of the vector L1 (the first sweep of a workgroup, right after the previous workgroup's rows went out, cost 1.9 x the second one)
  // stateOnly: a row that does not depend on the inputs (the end-effector pose error): its columns 30..59 are identically zero, lq_node_kernel does not read them
  // and they are not written (one of the four 128-byte lines of the row never exists)
  auto putGlobal = [&](int base, int row, real dval, real vval, real cval, bool stateOnly = false) {
    real* r = ad + base + row * 64;
    if (live && owner) {
      QM_STREAM_STORE(&r[cD], dval);
      if (!stateOnly || cV < 30) QM_STREAM_STORE(&r[cV], vval);
      if (!stateOnly || cC < 30 || cC >= 60) QM_STREAM_STORE(&r[cC], cval);
    }
  };

  // phase clocks of the profiling build (workgroup 1000; tools/riccati_phase_probe.py): 0 inputs | 1 first sweep | 2 constraint rows + J1 | 3 second sweep |
  // 4 J2 operand + chain rule on the matrix cores | 5 J1 += J2 in place | 6 phi rows out | 7 foot callbacks (parking) | 8 end-effector callback (pose error rows).  Plus, per workgroup, wall-clock start / end (100 MHz) for the
  // occupancy picture of the launch (rounds, tail).
  QM_TICK_DECL;
#ifdef QM_RICCATI_TIMING
  const unsigned long long qmWgStart = wall_clock64();
#endif
  int nc = 0;
  QM_TICK(0);
#pragma unroll 1
  for (int stage = 0; stage < 2; ++stage) {
    const AdIn in{x, u, stage ? x2 : x, dd, stage ? dt : 0.0_r};
    FlowOut<Du, Du3, Du3> f;
    BaseMotion2<Du, Du3> bm;
    QM_WAVE_SYNC();   // (the published composites of the previous stage have been read by every lane)
    auto footCb =
        [&](int c, Vec3<Du> r, Vec3<Du3> v) {
          QM_TICK(stage ? 3 : 1);
          if (stage == 0) {
            real* p = park + (c * 15) * 64;
            p[0] = r.x.v; p[64] = r.x.d; p[128] = r.y.v; p[192] = r.y.d; p[256] = r.z.v; p[320] = r.z.d;
            p[384] = v.x.v; p[448] = v.x.d; p[512] = v.x.e; p[576] = v.y.v; p[640] = v.y.d; p[704] = v.y.e; p[768] = v.z.v; p[832] = v.z.d; p[896] = v.z.e;
          }
          QM_TICK(7);
        };
    auto eeCb =
        [&](Vec3<Du> r, const Mat3<Du>& R) {
          // external force of the compliant contact: linear in the base position (force-type slot of lanes 3..5), configuration tangent -K dr
          QM_TICK(stage ? 3 : 1);
          const real* xs = stage ? x2 : x;
          const Vec3<Du3> fe(Du3(-Ke * (xs[6] + r.x.v - pEnv[0]), -Ke * r.x.d, dd == 3 ? -Ke : 0.0_r), Du3(-Ke * (xs[7] + r.y.v - pEnv[1]), -Ke * r.y.d, dd == 4 ? -Ke : 0.0_r),
                             Du3(-Ke * (xs[8] + r.z.v - pEnv[2]), -Ke * r.z.d, dd == 5 ? -Ke : 0.0_r));
          if (stage == 0) {  // end-effector pose error (EndEffectorConstraint.cpp:36-78): no velocity / force dependence, d/dp = identity
            // orientation error and its tangent in closed form: the quaternion of the PRIMAL rotation (one square root, one division, in plain arithmetic), and for
            // the tangent the world-frame rotation increment  [dtheta]x = dR R^T  of the lane's direction: dq = 1/2 (dtheta, 0) (x) q, and quaternionDistance is
            // linear in q.  (Until round 4 the whole chain matrix -> quaternion -> distance ran in dual arithmetic: a dual square root and two dual divisions per lane.)
            Mat3<real> Rp;
            Rp.c0 = Vec3<real>(R.c0.x.v, R.c0.y.v, R.c0.z.v); Rp.c1 = Vec3<real>(R.c1.x.v, R.c1.y.v, R.c1.z.v); Rp.c2 = Vec3<real>(R.c2.x.v, R.c2.y.v, R.c2.z.v);
            real qp[4];
            matrixToQuaternion(Rp, qp);
            const Vec3<real> odp = quaternionDistance(qp, eeQuatRef);
            const real thx = R.c0.z.d * R.c0.y.v + R.c1.z.d * R.c1.y.v + R.c2.z.d * R.c2.y.v;   // (dR R^T)[2][1]
            const real thy = R.c0.x.d * R.c0.z.v + R.c1.x.d * R.c1.z.v + R.c2.x.d * R.c2.z.v;   // (dR R^T)[0][2]
            const real thz = R.c0.y.d * R.c0.x.v + R.c1.y.d * R.c1.x.v + R.c2.y.d * R.c2.x.v;   // (dR R^T)[1][0]
            real dq[4];
            dq[0] = 0.5_r * (thx * qp[3] + thy * qp[2] - thz * qp[1]);
            dq[1] = 0.5_r * (thy * qp[3] + thz * qp[0] - thx * qp[2]);
            dq[2] = 0.5_r * (thz * qp[3] + thx * qp[1] - thy * qp[0]);
            dq[3] = -0.5_r * (thx * qp[0] + thy * qp[1] + thz * qp[2]);
            const Vec3<real> odd = quaternionDistance(dq, eeQuatRef);
            const Vec3<Du> od(Du(odp.x, odd.x), Du(odp.y, odd.y), Du(odp.z, odd.z));
            const Du hq[6] = {x[6] + r.x - eePosRef[0], x[7] + r.y - eePosRef[1], x[8] + r.z - eePosRef[2], od.x, od.y, od.z};
#pragma unroll
            for (int q = 0; q < 6; ++q) {
              // padding column 61 of the three position rows carries the force error f_e - f_ref for lq_node_kernel's soft constraint
              const real hf = q < 3 ? (q == 0 ? fe.x.v : (q == 1 ? fe.y.v : fe.z.v)) - fRef[q < 3 ? q : 0] : 0.0_r;
              putGlobal(AD_EE, q, hq[q].d, 0.0_r, isVal ? hq[q].v : ((q < 3 && dd == 3 + q) ? 1.0_r : ((q < 3 && dd == 19) ? hf : 0.0_r)), true);
            }
          }
          QM_TICK(8);
          return fe;
        };
    real* pubN = lds + ADL_PUB + grp * SWEEP_PUB_NODE;
    centroidalSweepOwnChain(md, st.gravity, in, dd, pubN, footCb, eeCb, f, bm);
    QM_TICK(stage ? 3 : 1);
    if (stage == 0) {
      // ---- equality constraints in the insertion order of QMInterface.cpp:116-131
      if (!terminal) {
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
          const bool contact = contactOf(mode, c);
          const real* p = park + (c * 15) * 64;
          const Vec3<Du> r(Du(p[0], p[64]), Du(p[128], p[192]), Du(p[256], p[320]));
          const Vec3<Du3> vj(Du3(p[384], p[448], p[512]), Du3(p[576], p[640], p[704]), Du3(p[768], p[832], p[896]));
          const Vec3<Du3> vf = bm.dp + cross(bm.omega, r) + vj;
          // velocity-type row: d/dh_lin = identity through dp; d/dp_z only through the position-error gain
          auto putVel = [&](int row, Du3 hv, int axisIdx, real gainZ) {
            putGlobal(AD_CD, row, hv.d, hv.e, isVal ? hv.v : (dd == axisIdx ? 1.0_r : (dd == 5 ? gainZ : 0.0_r)));
          };
          if (contact) {  // zeroVelocity (QMInterface.cpp:126, 324-339; Ax(2,2) = positionErrorGain)
            putVel(nc, vf.x, 0, 0.0_r); putVel(nc + 1, vf.y, 1, 0.0_r);
            putVel(nc + 2, vf.z + st.position_error_gain * (x[8] + r.z), 2, st.position_error_gain);
            nc += 3;
          } else {  // zeroForce (QMInterface.cpp:123-124) then normalVelocity (QMPreComputation.cpp:56-66)
            // rows nc .. nc + 2: C = 0, D = unit vector on force input 3 c + q, e = u[3 c + q] -- known from the mode alone, so they are NOT stored:
            // lq_node_kernel synthesises them (1.5 KB per swing foot and node each way)
            real zp, zv;
            swingReference(st, sched, c, t, phase, zp, zv);
            putVel(nc + 3, vf.z - zv + st.position_error_gain * (x[8] + r.z - zp), 2, st.position_error_gain);
            nc += 4;
          }
        }
      }
      QM_WAVE_SYNC();   // every lane has read its parked feet: the area becomes the Jacobian rows
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        putRow(L1, i, f.lin[i].d, 0.0_r, isF ? f.lin[i].e : (isVal ? f.lin[i].v : 0.0_r));
        putRow(L1, 3 + i, f.ang[i].d, 0.0_r, isF ? f.ang[i].e : (isVal ? f.ang[i].v : 0.0_r));
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) putRow(L1, 6 + i, f.kin[i].d, f.kin[i].e, isVal ? f.kin[i].v : ((i < 3 && dd == i) ? 1.0_r : 0.0_r));
      if (dd == 19) {   // the columns nobody owns: p (f does not depend on the base position) and padding 63
#pragma unroll
        for (int i = 0; i < 12; ++i) L1[i * 64 + 63] = 0.0_r;
      }
      if (dd == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { x2[i] = fma(dt, f.lin[i].v, x[i]); x2[3 + i] = fma(dt, f.ang[i].v, x[3 + i]); }
#pragma unroll
        for (int i = 0; i < 6; ++i) x2[6 + i] = fma(dt, f.kin[i].v, x[6 + i]);
      }
      QM_WAVE_SYNC();
      QM_TICK(2);
    } else {
      // ---- second stage: J2[:, 0:12] as a matrix-core operand, the rest of J2 stays in registers until J1 has been multiplied
      if (dd < 3) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const real dv = i < 3 ? f.lin[i].d : (i < 6 ? f.ang[i - 3].d : f.kin[i - 6].d);
          const real vv = i < 6 ? 0.0_r : f.kin[i - 6].e;
          A2[i * 16 + dd] = (i >= 6 && i < 9 && i - 6 == dd) ? 1.0_r : 0.0_r;   // d/dh_lin
          A2[i * 16 + 3 + dd] = vv;                                          // d/dh_ang
          A2[i * 16 + 9 + dd] = dv;                                          // d/dzyx
          A2[i * 16 + 12 + dd] = 0.0_r;
        }
      } else if (dd < 6) {   // d/dp: only the contact force of the force-tracking formulation depends on the base position
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          A2[i * 16 + 3 + dd] = i < 3 ? f.lin[i].e : (i < 6 ? f.ang[i - 3].e : 0.0_r);
          if (dd == 3) A2[i * 16 + 15] = 0.0_r;
        }
      }
      QM_WAVE_SYNC();
      // ---- chain rule on the matrix cores: acc[g][tn] = J2[:, 0:12] J1[:, 16 tn ..] for the three nodes
      QmAcc acc[AD_NODES][4];
#pragma unroll
      for (int g = 0; g < AD_NODES; ++g) {
        const real* A2g = lds + ADL_A2 + g * 192;
        const real* L1g = lds + ADL_PARK + g * 768;
        real av[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) { const real raw = A2g[(la < 12 ? la : 0) * 16 + 4 * ks + h]; av[ks] = la < 12 ? raw : 0.0_r; }
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[g][tn][r] = 0.0_r;
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) qmMfma(acc[g][tn], av[ks], L1g[(4 * ks + h) * 64 + tn * 16 + l16]);
        }
      }
      QM_WAVE_SYNC();   // J1 has been read by every lane: the slot owners add J2 (+ dt J2[:, q_j] into the v_j columns) in place
      QM_TICK(4);
      {
        const real sh = dd >= 3 ? dt : 0.0_r;
        // (36 read-modify-writes of distinct LDS words: all reads first, held, then the sums, then the stores -- as `r[c] += v` every one waited for its own read, the compiler
        //  cannot move a load across a store to the same array)
        real dv[12], vv[12], cv[12];
        auto addRow = [&](int row, real dval, real vval, real cval) { dv[row] = dval; vv[row] = vval; cv[row] = cval; };
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          addRow(i, f.lin[i].d, 0.0_r, isF ? f.lin[i].e : (isVal ? f.lin[i].v : 0.0_r));
          addRow(3 + i, f.ang[i].d, 0.0_r, isF ? f.ang[i].e : (isVal ? f.ang[i].v : 0.0_r));
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) addRow(6 + i, f.kin[i].d, f.kin[i].e, isVal ? f.kin[i].v : ((i < 3 && dd == i) ? 1.0_r : 0.0_r));
        if (owner) {
          real a0[12], a1[12], a2[12];
#pragma unroll
          for (int row = 0; row < 12; ++row) { const real* r = L1 + row * 64; a0[row] = r[cD]; a1[row] = r[cV]; a2[row] = r[cC]; }
#pragma unroll
          for (int row = 0; row < 12; ++row) { QM_KEEP(a0[row]); QM_KEEP(a1[row]); QM_KEEP(a2[row]); }
#pragma unroll
          for (int row = 0; row < 12; ++row) { real* r = L1 + row * 64; r[cD] = a0[row] + dv[row]; r[cV] = a1[row] + fma(sh, dv[row], vv[row]); r[cC] = a2[row] + cv[row]; }
        }
      }
      QM_WAVE_SYNC();
      QM_TICK(5);
      // ---- phi = dt/2 (k1 + k2): rows leave in accumulator layout (4 rows x 128-byte runs per store instruction)
#pragma unroll
      for (int g = 0; g < AD_NODES; ++g) {
        const int gR = blockIdx.x * AD_NODES + g;
        const int gN = gR < total ? gR : total - 1;
        const int nodeG = gN % (a.N + 1), instG = gN / (a.N + 1);
        const bool termG = nodeG == a.N;
        const bool liveG = gR < total && !a.done[instG];
        const real dtG = a.dtgrid[gN];
        const real* L1g = lds + ADL_PARK + g * 768;
        real* adG = a.adrows + size_t(gN) * AD_DOUBLES;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const int i = h + 4 * r, c = tn * 16 + l16;
            const real s = L1g[i * 64 + c];
            // terminal node: the slope itself (one stage); otherwise dt/2 (J1 + J2 + dt J2x J1), the value column without the product
            const real v = termG ? 0.5_r * s : 0.5_r * dtG * (c < 60 ? fma(dtG, acc[g][tn][r], s) : s);
            if (liveG) QM_STREAM_STORE(&adG[AD_PHI + i * 64 + c], v);
          }
        }
      }
    }
  }
  if (live && dd == 0) { a.stageNc[gnode] = nc; a.nodeMode[gnode] = mode; }
  QM_TICK(6);
  QM_TICK_FLUSH(320, blockIdx.x == 1000 && lane == 0);
#ifdef QM_RICCATI_TIMING
  if (lane == 0 && blockIdx.x < QM_AD_WG_CLOCKS) { qmk::qmAdWgClock[2 * blockIdx.x] = qmWgStart; qmk::qmAdWgClock[2 * blockIdx.x + 1] = wall_clock64(); }
#endif
}
#endif   // QM_LQ_UNIT

}  // namespace qmk
