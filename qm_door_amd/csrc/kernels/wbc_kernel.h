// wbc_kernel -- whole-body controller: rigid-body model update, task assembly and the three-level hierarchical QP.
// One wavefront per robot instance; every matrix lives in (dynamic) LDS.
//
// Replaces qm::WbcBase::update / updateMeasured / updateDesired (qm_wbc/src/WbcBase.cpp:123-238), the task builders
// (WbcBase.cpp:240-578), Task stacking (qm_wbc/include/qm_wbc/Task.h:28-65), the HoQp cascade (qm_wbc/src/HoQp.cpp:12-158)
// including its qpOASES solve (HoQp.cpp:135-150) and null-space update (HoQp.cpp:126-133), the task grouping of
// HierarchicalWbc::update / HierarchicalMpcWbc::update and WbcBase::updateCmd (WbcBase.cpp:580-595).
//
// Model update: one recursive velocity/bias-acceleration pass over the tree (RNEA forward sweep with zero generalized
// acceleration) gives every body's twist and classical bias acceleration, from which
//   nle_k  = sum_b  J_k,b^T [ m (a_c + g) ; I alpha + w x I w ]          (lane k = generalized velocity k)
//   M_ik   = sum_b  J_i,b^T diag(m, I_b) J_k,b                           (lane k owns column k)
//   dJ v   = bias acceleration of the frame point                        (no dJ matrix is ever formed)
// QP: the reference's (H, c, D, f) of HoQp::formulateProblem with the slack block eliminated analytically (it is diagonal), so the factorised system is n x n
// (n <= 36) instead of (n + 56)^2; every level ends with a primal active-set method (qp_dev.h), an interior point only hands it its starting point.
// Only the highest-priority task may carry inequality rows (true for both reference controllers).
#pragma once
#include <type_traits>
#include "../../../include/qmgpu.h"
#include "gpu_rt.h"
#ifndef QMGPU_DEBUG_INST
#define QMGPU_DEBUG_INST 0
#endif
#include "linesearch_kernel.h"  // DblIn
#include "sweep_dev.h"
#include "qp_dev.h"
#include "wave_gemm.h"

namespace qmk {

struct WbcArgs {
  const qmgpu_problem* P;
  int batch, variant;
  const double* xDes; const double* uDes; const double* rbd; const int* mode; const double* period; const double* time;
  double* inputLast; double* out; int* status;
  const double* eeForce;   // [batch][3] or null: external force on the arm end-effector (force tracking, own formulation)
  unsigned long long* workingSet;   // [batch][QMGPU_WBC_STATE_WORDS] in / out or null: the working sets of the previous tick (qmgpu_wbc_args::working_set)
};

constexpr int ND = 36, NVV = 24, MAXR = 22, MAXM = 56;
constexpr int LDZ = 37, LDK = 37;
// ---- LDS carve (doubles)
constexpr int W_IN = 0;                          // rbd[55] xDes[30] uDes[30] inputLast[30] -> 160
constexpr int W_Q = W_IN + 160;                  // qM vM qD vD [4][24]
constexpr int W_BODY = W_Q + 96;                 // per body: R9 p3 c3 I6 w3 al3 vo3 ao3 = 33  -> 19*33 = 627 (+pad)
constexpr int W_DOF = W_BODY + 640;              // dof axis[24][3], origin[24][3]
constexpr int W_WR = W_DOF + 144;                // body wrench force[19][3] torque[19][3]
constexpr int W_M = W_WR + 120;                  // M [24][24]
constexpr int W_NLE = W_M + 576;                 // nle[24]
constexpr int W_JF = W_NLE + 24;                 // feet J [12][24]
constexpr int W_JA = W_JF + 288;                 // arm J [6][24]
constexpr int W_MISC = W_JA + 144;               // see offsets below (144)
constexpr int W_A = W_MISC + 144;                // task A [MAXR][36], b[MAXR]
constexpr int W_B = W_A + MAXR * ND;
constexpr int W_D0 = W_B + 24;                   // D0 [MAXM][36]
constexpr int W_F0 = W_D0 + MAXM * ND;           // f0[56], slack solution v0[56]
constexpr int W_Z = W_F0 + 2 * MAXM;             // Z [36][LDZ]
constexpr int W_ZN = W_Z + ND * LDZ;             // Znew
constexpr int W_AZ = W_ZN + ND * LDZ;            // A Z [MAXR][LDZ]
constexpr int W_DZ = W_AZ + MAXR * LDZ;          // D0 Z [MAXM][LDZ]
constexpr int W_K = W_DZ + MAXM * LDZ;           // K / Cholesky [36][LDK]
constexpr int W_G = W_K + ND * LDK;              // G = AZ^T AZ + eps [36][LDK]
constexpr int W_VH = W_G + ND * LDK;             // Householder vectors [MAXR][40]
constexpr int W_VEC = W_VH + MAXR * 40;          // vectors: x[36] z[36] g[36] rd[36] rhs[36] dz[36] fhat[56] lam[56] wt[56] tz[56] red[64]
constexpr int W_BODY2 = W_VEC + 6 * 36 + 4 * 56 + 1024 + 8;   // (red[1024]: wavefront exchange scratch, only the host emulation uses more than 64) body / dof tables of the desired pass (wavefront 1)
constexpr int W_DOF2 = W_BODY2 + 640;
constexpr int W_TP = W_DOF2 + 144;               // T_P = L^-1 DZ_P' of the level solver's pinned rows [QP_KMAX][LDK] (round 6; until then over the K square, one row at a time)
constexpr int WBC_LDS_DOUBLES = W_TP + QP_KMAX * LDK;
constexpr int WBC_LDS_BYTES = WBC_LDS_DOUBLES * 8;
constexpr int WBC_THREADS = 256;   // the solving wavefront + three helpers (one per SIMD of the CU)
// misc block
constexpr int MI_FOOTPM = 0, MI_FOOTVM = 12, MI_FOOTDJV = 24, MI_FOOTPD = 36, MI_FOOTVD = 48, MI_EEPM = 60, MI_EEVM = 63, MI_EEWM = 66, MI_EEDJL = 69, MI_EEDJA = 72,
              MI_EERM = 75, MI_EEPD = 84, MI_EEVD = 87, MI_EERD = 90, MI_AL0 = 99, MI_BACC = 102, MI_JACC = 108 /*18*/, MI_BAX = 126 /*measured base Euler axes, 9*/;

// -DQM_WBC_DUMP (tools/wbc_variants.py, experiments only): instance 0 copies its whole LDS carve to a device symbol at a few checkpoints, so that two
// build variants of the kernel can be compared array by array (qmgpu_debug_wbc_dump).  The product build compiles every QM_WBC_CHECKPOINT to nothing.
#if defined(QM_WBC_DUMP) && !defined(QMGPU_HOST_EMULATION)
constexpr int WBC_DUMP_POINTS = 8;
__device__ double qmWbcDump[WBC_DUMP_POINTS * 17000];
#define QM_WBC_CHECKPOINT(cp) do { if (blockIdx.x == 0 && wave == 0) { QM_WAVE_SYNC(); for (int e_ = lane; e_ < WBC_LDS_DOUBLES; e_ += 64) qmk::qmWbcDump[(cp) * 17000 + e_] = lds[e_]; QM_WAVE_SYNC(); } } while (0)
#else
#define QM_WBC_CHECKPOINT(cp)
#endif
__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// Recursive pass: placements, twists and bias accelerations (generalized accelerations: 0 for the base, qddj for joints).
// The five kinematic chains hanging off the base (four legs of three joints, the arm of six: checked in qmgpu_create) are walked side by
// side, chain c in lane c; every lane publishes the bodies of its own chain to LDS, lane 0 the base as well.
#ifndef QM_WBC_EXP
#define QM_WBC_EXP 0     // experiments of tools/wbc_variants.py on the opaque-base failure; 0 in the product
#endif
#if QM_WBC_EXP == 2
__device__ __forceinline__ void bodyPass(
#else
__device__ inline void bodyPass(
#endif
const qmgpu_model& md, const double* q, const double* v, const double* qddj, double* body, double* dof, int lane) {
  double sz, cz, sy, cy, sx, cx;
  qmSinCos(q[3], sz, cz); qmSinCos(q[4], sy, cy); qmSinCos(q[5], sx, cx);
  double R0[9] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx, sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx, -sy, cy * sx, cy * cx};  // row major
  const double a3[3] = {0, 0, 1}, a4[3] = {-sz, cz, 0}, a5[3] = {cz * cy, sz * cy, -sy};
  double w0[3], al0[3], t1[3], t2[3], t3[3];
  for (int i = 0; i < 3; ++i) { t1[i] = a3[i] * v[3]; t2[i] = a4[i] * v[4]; t3[i] = a5[i] * v[5]; w0[i] = t1[i] + t2[i] + t3[i]; }
  { double c1[3], s12[3], c2[3]; cross3(t1, t2, c1); for (int i = 0; i < 3; ++i) s12[i] = t1[i] + t2[i]; cross3(s12, t3, c2); for (int i = 0; i < 3; ++i) al0[i] = c1[i] + c2[i]; }
  const double p0[3] = {q[0], q[1], q[2]}, vo0[3] = {v[0], v[1], v[2]}, ao0[3] = {0, 0, 0};
  if (lane == 0) {
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) { dof[k * 3 + i] = (i == k) ? 1.0 : 0.0; dof[72 + k * 3 + i] = 0.0; }
    for (int i = 0; i < 3; ++i) { dof[9 + i] = a3[i]; dof[12 + i] = a4[i]; dof[15 + i] = a5[i]; dof[72 + 9 + i] = p0[i]; dof[72 + 12 + i] = p0[i]; dof[72 + 15 + i] = p0[i]; }
  }
  auto publish = [&](int b, bool mine, const double* R, const double* p, const double* w, const double* al, const double* vo, const double* ao) {
    double c[3], RI[9], Iw[6];
    const double* cm = md.com[b];
    for (int i = 0; i < 3; ++i) c[i] = p[i] + R[i * 3] * cm[0] + R[i * 3 + 1] * cm[1] + R[i * 3 + 2] * cm[2];
    const double* in = md.inertia[b];
    const double I[9] = {in[0], in[1], in[2], in[1], in[3], in[4], in[2], in[4], in[5]};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) RI[i * 3 + j] = R[i * 3] * I[j] + R[i * 3 + 1] * I[3 + j] + R[i * 3 + 2] * I[6 + j];
    int e = 0;
    for (int i = 0; i < 3; ++i) for (int j = i; j < 3; ++j) Iw[e++] = RI[i * 3] * R[j * 3] + RI[i * 3 + 1] * R[j * 3 + 1] + RI[i * 3 + 2] * R[j * 3 + 2];
    if (mine) {
      double* o = body + b * 33;
      for (int i = 0; i < 9; ++i) o[i] = R[i];
      for (int i = 0; i < 3; ++i) { o[9 + i] = p[i]; o[12 + i] = c[i]; o[21 + i] = w[i]; o[24 + i] = al[i]; o[27 + i] = vo[i]; o[30 + i] = ao[i]; }
      for (int i = 0; i < 6; ++i) o[15 + i] = Iw[i];
    }
  };
  publish(0, lane == 0, R0, p0, w0, al0, vo0, ao0);
  const int chain = lane < 5 ? lane : 4;                 // idle lanes shadow the arm chain (their results are not published)
  const int first = chain < 4 ? 1 + 3 * chain : 13, len = chain < 4 ? 3 : 6;
  double R[9], p[3], w[3], al[3], vo[3], ao[3];
  for (int i = 0; i < 9; ++i) R[i] = R0[i];
  for (int i = 0; i < 3; ++i) { p[i] = p0[i]; w[i] = w0[i]; al[i] = al0[i]; vo[i] = vo0[i]; ao[i] = ao0[i]; }
#pragma unroll 1
  for (int sIdx = 0; sIdx < 6; ++sIdx) {
    const bool live = sIdx < len;
    const int b = first + (live ? sIdx : len - 1);       // finished chains repeat their last joint without publishing
    const double* off = md.joint_offset[b];
    double ow[3], tmp[3], tmp2[3];
    for (int i = 0; i < 3; ++i) ow[i] = R[i * 3] * off[0] + R[i * 3 + 1] * off[1] + R[i * 3 + 2] * off[2];
    // origin velocity / bias acceleration use the PARENT twist
    cross3(w, ow, tmp);
    for (int i = 0; i < 3; ++i) vo[i] += tmp[i];
    cross3(w, tmp, tmp2);
    cross3(al, ow, tmp);
    for (int i = 0; i < 3; ++i) { ao[i] += tmp[i] + tmp2[i]; p[i] += ow[i]; }
    const int ax = md.axis[b];
    const double aw[3] = {ax == 0 ? R[0] : (ax == 1 ? R[1] : R[2]), ax == 0 ? R[3] : (ax == 1 ? R[4] : R[5]), ax == 0 ? R[6] : (ax == 1 ? R[7] : R[8])};
    const double qd = v[5 + b], qdd = qddj ? qddj[b - 1] : 0.0;
    double wj[3] = {aw[0] * qd, aw[1] * qd, aw[2] * qd};
    cross3(w, wj, tmp);
    for (int i = 0; i < 3; ++i) { al[i] += tmp[i] + aw[i] * qdd; w[i] += wj[i]; }
    const bool mine = live && lane < 5;
    if (mine) for (int i = 0; i < 3; ++i) { dof[(5 + b) * 3 + i] = aw[i]; dof[72 + (5 + b) * 3 + i] = p[i]; }
    double sn, cs;
    qmSinCos(q[5 + b], sn, cs);
    for (int i = 0; i < 3; ++i) {   // rotation about the body axis ax: columns (ax + 1) % 3 and (ax + 2) % 3 mix
      const double r0 = R[i * 3], r1 = R[i * 3 + 1], r2 = R[i * 3 + 2];
      const double rb = ax == 0 ? r1 : (ax == 1 ? r2 : r0), rc = ax == 0 ? r2 : (ax == 1 ? r0 : r1);
      const double nb = cs * rb + sn * rc, nc = cs * rc - sn * rb;
      R[i * 3] = ax == 0 ? r0 : (ax == 1 ? nc : nb);
      R[i * 3 + 1] = ax == 0 ? nb : (ax == 1 ? r1 : nc);
      R[i * 3 + 2] = ax == 0 ? nc : (ax == 1 ? nb : r2);
    }
    publish(b, mine, R, p, w, al, vo, ao);
  }
}

// point kinematics on a body: position, velocity, classical bias acceleration
__device__ __forceinline__ void pointKin(const qmgpu_model& md, const double* body, int b, const double* off, double* pos, double* vel, double* acc) {
  const double* o = body + b * 33;
  double d[3], t[3], t2[3];
  for (int i = 0; i < 3; ++i) d[i] = o[i * 3] * off[0] + o[i * 3 + 1] * off[1] + o[i * 3 + 2] * off[2];
  for (int i = 0; i < 3; ++i) pos[i] = o[9 + i] + d[i];
  cross3(o + 21, d, t);
  for (int i = 0; i < 3; ++i) vel[i] = o[27 + i] + t[i];
  cross3(o + 21, t, t2);
  cross3(o + 24, d, t);
  for (int i = 0; i < 3; ++i) acc[i] = o[30 + i] + t[i] + t2[i];
}
__device__ __forceinline__ bool dofMoves(int k, int b) {
  if (k < 6) return true;
  const int jb = k - 5;
  const int hi = jb <= 12 ? 3 * ((jb + 2) / 3) : 18;
  return b >= jb && b <= hi;
}
// Jacobian column k of a point r on body b (world axes): lin, ang
__device__ __forceinline__ void jacCol(const double* dof, int k, int b, const double* r, double* lin, double* ang) {
  lin[0] = lin[1] = lin[2] = 0.0; ang[0] = ang[1] = ang[2] = 0.0;
  if (!dofMoves(k, b)) return;
  const double* a = dof + k * 3;
  if (k < 3) { lin[0] = a[0]; lin[1] = a[1]; lin[2] = a[2]; return; }
  const double* o = dof + 72 + k * 3;
  const double d[3] = {r[0] - o[0], r[1] - o[1], r[2] - o[2]};
  cross3(a, d, lin);
  ang[0] = a[0]; ang[1] = a[1]; ang[2] = a[2];
}
__device__ __forceinline__ void symMul(const double* I6, const double* v, double* o) {
  o[0] = I6[0] * v[0] + I6[1] * v[1] + I6[2] * v[2]; o[1] = I6[1] * v[0] + I6[3] * v[1] + I6[4] * v[2]; o[2] = I6[2] * v[0] + I6[4] * v[1] + I6[5] * v[2];
}

__device__ __forceinline__ double wbcSum(double* red, int lane, double v) { red[lane] = v; QM_WAVE_SYNC(); double s = 0; for (int i = 0; i < 64; ++i) s += red[i]; QM_WAVE_SYNC(); return s; }
__device__ __forceinline__ double wbcMax(double* red, int lane, double v) { red[lane] = v; QM_WAVE_SYNC(); double s = red[0]; for (int i = 1; i < 64; ++i) s = fmax(s, red[i]); QM_WAVE_SYNC(); return s; }
__device__ __forceinline__ double wbcMin(double* red, int lane, double v) { red[lane] = v; QM_WAVE_SYNC(); double s = red[0]; for (int i = 1; i < 64; ++i) s = fmin(s, red[i]); QM_WAVE_SYNC(); return s; }

// In-place Cholesky of the n x n matrix K (row stride LDK) in LDS, lane = row; pivots are floored at floorv
// (1e-13 x the largest diagonal entry of the level's cost Hessian G, as in the oracle's choleskyFloored).
__device__ inline void ldsCholesky(double* K, int n, int lane, double floorv) {
#pragma unroll 1
  for (int j = 0; j < n; ++j) {
    const double d = K[j * LDK + j];
    const double dj = sqrt(d > floorv ? d : floorv);
    QM_WAVE_SYNC();
    if (lane == j) K[j * LDK + j] = dj;
    else if (lane > j && lane < n) K[lane * LDK + j] = K[lane * LDK + j] / dj;
    QM_WAVE_SYNC();
    if (lane > j && lane < n) {
      const double lij = K[lane * LDK + j];
      for (int q = j + 1; q <= lane; ++q) K[lane * LDK + q] -= lij * K[q * LDK + j];
    }
    QM_WAVE_SYNC();
  }
}
// Solve L L^T x = y in place (y in LDS), lane = row.
__device__ inline void ldsCholSolve(const double* L, int n, double* y, int lane) {
#pragma unroll 1
  for (int j = 0; j < n; ++j) {
    const double xj = y[j] / L[j * LDK + j];
    QM_WAVE_SYNC();
    if (lane == j) y[j] = xj;
    else if (lane > j && lane < n) y[lane] -= L[lane * LDK + j] * xj;
    QM_WAVE_SYNC();
  }
#pragma unroll 1
  for (int j = n - 1; j >= 0; --j) {
    const double xj = y[j] / L[j * LDK + j];
    QM_WAVE_SYNC();
    if (lane == j) y[j] = xj;
    else if (lane < j) y[lane] -= L[j * LDK + lane] * xj;
    QM_WAVE_SYNC();
  }
}

// N = kernel(rows) (rows: r x n in LDS, row stride LDZ, destroyed): the reference takes Eigen's FullPivLU::kernel() (HoQp.cpp:129), i.e. the basis [-U11^-1 U12; I] in the
// column order full pivoting leaves -- NOT an orthonormal one -- with Eigen 3.3's pivot order: the largest entry of the remaining corner, ties to the smallest column
// position, then the smallest row position (its scalar visitor walks the column-major corner column by column and keeps the first strict maximum).  The level tasks carry
// unit rows: exact ties are the rule, and the basis -- the coordinates the minimum-norm representative of a level is taken in -- depends on the order.  The kernels and the
// CPU restatement of the tests take the same decisions with the same roundings.  Result: N (n x nNew, row stride LDK) in K; returns nNew.
// Also used for the implied equalities of a level (rows = the strongly active inequality rows, wbc_kernel).  A called function: three call sites, one copy; the arrays
// arrive as offsets into the dynamic LDS (qp_dev.h: qpSolve).
__device__ __attribute__((noinline)) int wbcNullSpace(int rowsOff, int r, int n, int kOff, int vhOff, int redOff, int lane) {
  QM_DYNAMIC_LDS(ldsBase);
  double* rows = ldsBase + rowsOff; double* K = ldsBase + kOff; double* Vh = ldsBase + vhOff; double* red = ldsBase + redOff;
  (void)red;
  QM_TICK_DECL;
  static_assert(MAXR <= 24 && ND <= 36 && 128 + 64 + 32 <= MAXR * 40, "index tables of the null-space step fit the region they are carved from");
  {
      int* ip = reinterpret_cast<int*>(Vh);        // colPerm[36] | rowOf[MAXR] | pivOk[MAXR] | freePos[36]  (the reflector table of round 3: free here)
      int* colPerm = ip; int* rowOf = ip + 40; int* pivOk = ip + 64; int* freePos = ip + 96;
      const int size = r < n ? r : n;
      // Lanes: row i of A Z is worked on by up to three lanes, i, r + i and 2 r + i ("chunks"), each eliminating a contiguous third of the column positions behind the pivot
      // (r <= 21: three chunks, 22: two) -- with one lane per row only r of the 64 lanes worked and a step was 35 dependent LDS round trips long.  Same arithmetic per
      // entry, same decisions: the chunks' column ranges are in increasing order, so "largest magnitude, first position on ties" of a row is the best of chunk 0, else 1,
      // else 2; the per-row state (rowPos, best, bj) is kept identical in the lanes of a row, the pivot's row is chosen among the chunk-0 lanes (lane = row index) as before.
      const int nChunk = 3 * r <= 64 ? 3 : (2 * r <= 64 ? 2 : 1);
      const int chunk = (lane >= r ? 1 : 0) + (lane >= 2 * r ? 1 : 0);
      const bool active = lane < nChunk * r;
      const int rowIdx = active ? lane - chunk * r : 0;
      double* xchgV = Vh + 128; int* xchgJ = reinterpret_cast<int*>(Vh + 192);   // exchange of the chunks' candidates (64 doubles, 64 ints; beyond the index tables)
      int rowPos = rowIdx;
      int colPermReg = lane;     // lane j: the original column at position j (swapped between lanes with v_readlane; LDS copy after the loop)
      int rowOfReg = 0;          // lane k: the row that gave pivot k
      double maxPivot = 0.0;
      int nonzero = 0;
      double* row = rows + rowIdx * LDZ;
      // largest entry of this lane's row over the column positions >= k, first one on ties (Eigen's visitor keeps the first strict maximum of its column-major walk: within a row that is the smallest column position).  Loads in
      // batches of eight before any store: a store to LDS between two loads of the same array serialises them (the compiler cannot tell the rows apart)
      double best = -1.0; int bj = 0;
      if (active) {
#pragma unroll 1
        for (int j0 = 0; j0 < n; j0 += 8) {
          double v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = fabs(row[j0 + q < n ? j0 + q : 0]);
#pragma unroll
          for (int q = 0; q < 8; ++q) if (j0 + q < n && v[q] > best) { best = v[q]; bj = j0 + q; }
        }
      }
      // One step = one dependent chain; what is on it besides the elimination itself is kept in registers: the pivot's lane from a ballot (the
      // smallest-row-position rule needs a second reduction only when two rows tie), its column and value by v_readlane, the permutations in lanes.
      QM_TICK(0);
#pragma unroll 1
      for (int k = 0; k < size; ++k) {
        const bool mine = lane < r && rowPos >= k;
        const double gmax = qmAllMax(mine ? best : -1.0, red);
        if (!(gmax > 0.0)) break;
        const bool cand = mine && best == gmax;
        const unsigned long long tied = qmBallot(cand);
        int Lp;
        if ((tied & (tied - 1)) == 0) Lp = qmFirstBit(tied);
        else Lp = int(qmAllMin(cand ? double((bj * 64 + rowPos) * 64 + lane) : 1e9, red)) & 63;   // ties between rows: the smallest COLUMN position, then the smallest row position (Eigen's column-major scan)
        const int pr = qmReadLaneInt(rowPos, Lp), pc = qmReadLaneInt(bj, Lp);
        QM_TICK(1);
        maxPivot = fmax(maxPivot, gmax);
        if (active) { if (rowIdx == Lp) rowPos = k; else if (rowPos == k) rowPos = pr; }
        { const int ck = qmReadLaneInt(colPermReg, k), cp = qmReadLaneInt(colPermReg, pc); if (lane == k) colPermReg = cp; else if (lane == pc) colPermReg = ck; }
        if (lane == k) rowOfReg = Lp;
        double vk = 0.0, vpc = 0.0;
        if (active) { vk = row[k]; vpc = row[pc]; }
        QM_WAVE_SYNC();
        if (pc != k && lane < r) { row[k] = vpc; row[pc] = vk; }
        const double pivot = qmReadLane(vpc, Lp);
        QM_WAVE_SYNC();
        QM_TICK(2);
        // elimination of the rows still below the pivot, this lane's share of the column positions; the largest entry of the updated row (positions > k) is found on
        // the way: the next step's candidate
        const double* prow = rows + Lp * LDZ;
        best = -1.0; bj = k + 1;
        const int len = n - k - 1, per = (len + nChunk - 1) / nChunk;
        const int jLo = k + 1 + chunk * per, jHi = (jLo + per < n) ? jLo + per : n;
        if (active && rowPos > k) {
          const double f = vpc / pivot;
#pragma unroll 1
          for (int j0 = jLo; j0 < jHi; j0 += 8) {
            double a[8], pv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int j = j0 + q < jHi ? j0 + q : jLo; a[q] = row[j]; pv[q] = prow[j]; }
            // product and difference rounded separately (no fused multiply-add): the pivot search compares these numbers for EQUALITY of magnitude with
            // entries of other rows (the level tasks carry unit rows, so exact ties are the rule, not the exception) and must take the decisions the
            // oracle's kernelFullPivLU takes on the host
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = qmSubNoFma(a[q], qmMulNoFma(f, pv[q]));
#pragma unroll
            for (int q = 0; q < 8; ++q) if (j0 + q < jHi) { row[j0 + q] = a[q]; const double v = fabs(a[q]); if (v > best) { best = v; bj = j0 + q; } }
          }
        }
        QM_TICK(3);
        if (nChunk > 1) {   // the row's candidate: the chunks' candidates in the order of their column ranges (strictly larger wins: first position on ties)
          QM_WAVE_SYNC();
          xchgV[lane] = best; xchgJ[lane] = bj;
          QM_WAVE_SYNC();
          if (active) {
            double b0 = xchgV[rowIdx]; int j0 = xchgJ[rowIdx];
            const double b1 = xchgV[rowIdx + r]; const int j1 = xchgJ[rowIdx + r];
            const double b2 = nChunk > 2 ? xchgV[rowIdx + 2 * r] : -1.0; const int j2 = nChunk > 2 ? xchgJ[rowIdx + 2 * r] : 0;
            if (b1 > b0) { b0 = b1; j0 = j1; }
            if (b2 > b0) { b0 = b2; j0 = j2; }
            best = b0; bj = j0;
          }
        }
        ++nonzero;
        QM_WAVE_SYNC();
        QM_TICK(4);
      }
      if (lane < n) colPerm[lane] = colPermReg;
      if (lane < nonzero) rowOf[lane] = rowOfReg;
      QM_WAVE_SYNC();
      // rank: pivots above Eigen's default threshold eps * size * max pivot; the others' columns count as free
      const double thresh = maxPivot * 2.220446049250313e-16 * double(size);
      // (rank and the list of free column positions from two ballots: as loops over the LDS table -- one of them on lane 0 alone -- they were 36 + 18 dependent LDS round trips)
      const bool okReg = lane < nonzero && fabs(rows[rowOfReg * LDZ + lane]) > thresh;
      if (lane < nonzero) pivOk[lane] = okReg ? 1 : 0;
      const int rank = qmPopCount(qmBallot(okReg));
      const int nNew = n - rank;
      const bool freeReg = lane < n && !okReg;
      const unsigned long long freeMask = qmBallot(freeReg);
      if (freeReg) freePos[qmPopCount(freeMask & ((1ull << lane) - 1ull))] = lane;
      for (int e = lane; e < ND * LDK; e += 64) K[e] = 0.0;      // N (n x nNew), one kernel vector per lane / column
      QM_WAVE_SYNC();
      QM_TICK(5);
      {
        // U11 X = -U12 for this lane's free column: back substitution over the accepted pivots, X(:, lane) in registers (fully unrolled: compile-time
        // indices), the U entries as wave-uniform LDS reads; the results leave for LDS after the loop (no store between the loads)
        const int fp = lane < nNew ? freePos[lane] : 0;
        // (which pivots count and where their rows are: read once, all loads in flight together, instead of one LDS round trip in front of every step)
        bool okk[MAXR]; int rb[MAXR];
#pragma unroll
        for (int k = 0; k < MAXR; ++k) { const int ok = pivOk[k < nonzero ? k : 0], ro = rowOf[k < nonzero ? k : 0]; okk[k] = k < nonzero && ok != 0; rb[k] = ro * LDZ; }
        double xk[MAXR];
#pragma unroll
        for (int k = MAXR - 1; k >= 0; --k) {
          xk[k] = 0.0;
          if (okk[k]) {   // wave-uniform
            const double* urow = rows + rb[k];
            double u[MAXR];
#pragma unroll
            for (int k2 = k + 1; k2 < MAXR; ++k2) u[k2] = urow[k2 < n ? k2 : 0];
            double sacc = fp >= k ? -urow[fp] : 0.0;
#pragma unroll
            for (int k2 = k + 1; k2 < MAXR; ++k2) sacc -= (k2 < nonzero ? u[k2] : 0.0) * xk[k2];
            xk[k] = sacc / urow[k];
          }
        }
        if (lane < nNew) {
#pragma unroll
          for (int k = 0; k < MAXR; ++k) if (okk[k]) K[colPerm[k] * LDK + lane] = xk[k];
          K[colPerm[fp] * LDK + lane] = 1.0;
        }
      }
      QM_WAVE_SYNC();
      QM_TICK(6);
      QM_TICK_FLUSH(352, blockIdx.x == 0 && lane == 0);
    return nNew;
  }
}

__global__ void __launch_bounds__(WBC_THREADS) QM_ONE_WAVE_PER_SIMD wbc_kernel(WbcArgs a) {
#if defined(QM_WBC_OPAQUE_MASK) && !defined(QMGPU_HOST_EMULATION)
  // Experiment (tools/wbc_variants.py): array group g of the LDS carve is addressed through one opaque
  // address-space-3 base register when bit g of the mask is set (round 2's "whole base opaque" = mask 31 returned wrong torques).
  QM_DYNAMIC_LDS(lds);
  QM_OPAQUE_LDS(double, ldsO, lds);
  double* ldsQ = (double*)ldsO;
#define QM_WBC_BASE(g) ((((QM_WBC_OPAQUE_MASK) >> (g)) & 1) ? ldsQ : lds)
#else
  QM_DYNAMIC_LDS(lds);
#define QM_WBC_BASE(g) lds
#endif
  QM_POISON_LDS(lds, WBC_LDS_DOUBLES);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, inst = blockIdx.x;
  // (the model constants reach this kernel through ~800 vector loads per instance -- after its first global store a kernel cannot use scalar loads for them,
  //  gpu_rt.h: QM_CONSTANT_REF; a copy of the struct in LDS was measured in round 3 and not kept: 0.4405 -> 0.4445 ms, the loads are batched well enough)
  const qmgpu_model& md = a.P->model;
  const qmgpu_settings& st = a.P->settings;
  double fe[3] = {0.0, 0.0, 0.0};   // external force on the arm end-effector (force tracking; zero otherwise)
  if (a.eeForce) for (int i = 0; i < 3; ++i) fe[i] = a.eeForce[size_t(inst) * 3 + i];
  // opaque-mask groups: 0 inputs | 1 coordinates | 2 model | 3 task equalities | 4 task inequalities | 5 QP matrices | 6 small vectors | 7 row vectors | 8 exchange scratch + control words
  double* in = QM_WBC_BASE(0) + W_IN; double* rbd = in; double* xDes = in + 55; double* uDes = in + 85; double* il = in + 115;
  double* qM = QM_WBC_BASE(1) + W_Q; double* vM = qM + 24; double* qD = qM + 48; double* vD = qM + 72;
  double* body = QM_WBC_BASE(2) + W_BODY; double* dof = QM_WBC_BASE(2) + W_DOF; double* wr = QM_WBC_BASE(2) + W_WR; double* M = QM_WBC_BASE(2) + W_M; double* nle = QM_WBC_BASE(2) + W_NLE;
  double* Jf = QM_WBC_BASE(2) + W_JF; double* Ja = QM_WBC_BASE(2) + W_JA; double* mi = QM_WBC_BASE(2) + W_MISC;
  double* A = QM_WBC_BASE(3) + W_A; double* bvec = QM_WBC_BASE(3) + W_B; double* D0 = QM_WBC_BASE(4) + W_D0; double* f0 = QM_WBC_BASE(4) + W_F0; double* v0 = f0 + MAXM;
  double* Z = QM_WBC_BASE(5) + W_Z; double* Zn = QM_WBC_BASE(5) + W_ZN; double* AZ = QM_WBC_BASE(5) + W_AZ; double* DZ = QM_WBC_BASE(5) + W_DZ; double* K = QM_WBC_BASE(5) + W_K; double* G = QM_WBC_BASE(5) + W_G; double* Vh = QM_WBC_BASE(5) + W_VH;
  double* xs = QM_WBC_BASE(6) + W_VEC; double* zs = xs + 36; double* gs = zs + 36; double* rds = gs + 36; double* rhs = rds + 36; double* dzs = rhs + 36;
  double* fhat = QM_WBC_BASE(7) + (W_VEC + 6 * 36); double* lam = fhat + 56; double* wt = lam + 56; double* tzv = wt + 56; double* red = QM_WBC_BASE(8) + (W_VEC + 6 * 36 + 4 * 56); double* ctl = red + 1024;

  // Wavefront 0 solves the instance; the other three sit on the CU's idle SIMDs and take their share of the matrix-core tiles of the
  // interior point between two workgroup barriers (ipm_dev.h: ipmKTiles).  Command word: ctl[4] (0 = leave).
  double* forkCmd = ctl + 4; double* forkJob = red + 512;   // (red[0..63] carries the interior point's broadcasts, red[128..383] its partial sums; nothing else of it is used on the GPU)
  // ---- S5: desired pass (WbcBase.cpp:205-237), on wavefront 1 while wavefront 0 runs the measured pass, M, nle and the Jacobians: it has its own
  //      body / dof tables and writes only v_des of the base and the desired entries of mi, none of which is read before the join after S4.
  //      v_des base from the centroidal map (WbcBase.cpp:217-219) with the MPC's own sweep.
  auto desiredPass = [&](double* body, double* dof) {
    {
      double k1z[12];
  #pragma unroll
      for (int i = 0; i < 12; ++i) k1z[i] = 0.0;
      const DblIn din{xDes, uDes, 0.0, k1z};
      double f[12];
      BaseMotion<double> bm;
      centroidalSweep<double>(md, st.gravity, din, [&](int, Vec3<double>, Vec3<double>) {}, [&](Vec3<double>, const Mat3<double>&) { return Vec3<double>(); }, f, bm);
      if (lane == 0) for (int i = 0; i < 6; ++i) vD[i] = f[6 + i];
    }
    QM_WAVE_SYNC();
    bodyPass(md, qD, vD, mi + MI_JACC, body, dof, lane);
    QM_WAVE_SYNC();
    if (lane == 0) {
      // momentum rate produced by (v_des, joint accelerations, zero base acceleration): Adot v + Aj qdd_j (WbcBase.cpp:231-234)
      double ct[3] = {0, 0, 0};
      for (int b = 0; b < QMGPU_NB; ++b) for (int i = 0; i < 3; ++i) ct[i] += md.mass[b] * body[b * 33 + 12 + i];
      for (int i = 0; i < 3; ++i) ct[i] /= md.total_mass;
      double hl[3] = {0, 0, 0}, ha[3] = {0, 0, 0}, Ic[6] = {0, 0, 0, 0, 0, 0};
      for (int b = 0; b < QMGPU_NB; ++b) {
        const double* o = body + b * 33;
        double pos[3], vel[3], acc[3], Iw_w[3], Iw_al[3], t[3], r[3], t2[3];
        pointKin(md, body, b, md.com[b], pos, vel, acc);
        symMul(o + 15, o + 21, Iw_w); symMul(o + 15, o + 24, Iw_al); cross3(o + 21, Iw_w, t);
        for (int i = 0; i < 3; ++i) r[i] = pos[i] - ct[i];
        double ma[3] = {md.mass[b] * acc[0], md.mass[b] * acc[1], md.mass[b] * acc[2]};
        cross3(r, ma, t2);
        const double rr = dot3(r, r), m = md.mass[b];
        for (int i = 0; i < 3; ++i) { hl[i] += ma[i]; ha[i] += Iw_al[i] + t[i] + t2[i]; }
        Ic[0] += o[15] + m * (rr - r[0] * r[0]); Ic[1] += o[16] - m * r[0] * r[1]; Ic[2] += o[17] - m * r[0] * r[2];
        Ic[3] += o[18] + m * (rr - r[1] * r[1]); Ic[4] += o[19] - m * r[1] * r[2]; Ic[5] += o[20] + m * (rr - r[2] * r[2]);
      }
      double rl[3] = {-hl[0], -hl[1], -md.total_mass * st.gravity - hl[2]}, ra[3] = {-ha[0], -ha[1], -ha[2]};
      for (int c = 0; c < 4; ++c) {
        double pos[3], vel[3], acc[3], t[3], r[3];
        pointKin(md, body, md.foot_body[c], md.foot_offset[c], pos, vel, acc);
        for (int i = 0; i < 3; ++i) { mi[MI_FOOTPD + 3 * c + i] = pos[i]; mi[MI_FOOTVD + 3 * c + i] = vel[i]; r[i] = pos[i] - ct[i]; rl[i] += uDes[3 * c + i]; }
        cross3(r, uDes + 3 * c, t);
        for (int i = 0; i < 3; ++i) ra[i] += t[i];
      }
      {  // external end-effector force in the desired momentum rate (zero without force tracking)
        double pos[3], vel[3], acc[3], t[3], r[3];
        pointKin(md, body, md.ee_body, md.ee_offset, pos, vel, acc);
        for (int i = 0; i < 3; ++i) { r[i] = pos[i] - ct[i]; rl[i] += fe[i]; }
        cross3(r, fe, t);
        for (int i = 0; i < 3; ++i) ra[i] += t[i];
      }
      // wdot = Ic^-1 ra ; euler acceleration = T^-1 wdot ; linear = rl/m - wdot x (c - p0)
      Sym3<double> S; S.xx = Ic[0]; S.xy = Ic[1]; S.xz = Ic[2]; S.yy = Ic[3]; S.yz = Ic[4]; S.zz = Ic[5];
      const Vec3<double> wd = solveSym3(S, Vec3<double>(ra[0], ra[1], ra[2]));
      const double wdv[3] = {wd.x, wd.y, wd.z}, rc[3] = {ct[0] - qD[0], ct[1] - qD[1], ct[2] - qD[2]};
      double t[3];
      cross3(wdv, rc, t);
      double sz, cz, sy, cy;
      sincos(qD[3], &sz, &cz); sincos(qD[4], &sy, &cy);
      const double tmp = (cz * wd.x + sz * wd.y) / cy;
      for (int i = 0; i < 3; ++i) mi[MI_BACC + i] = rl[i] / md.total_mass - t[i];
      mi[MI_BACC + 3] = sy * tmp + wd.z; mi[MI_BACC + 4] = cz * wd.y - sz * wd.x; mi[MI_BACC + 5] = tmp;
      double pos[3], vel[3], acc[3];
      pointKin(md, body, md.ee_body, md.ee_offset, pos, vel, acc);
      for (int i = 0; i < 3; ++i) { mi[MI_EEPD + i] = pos[i]; mi[MI_EEVD + i] = vel[i]; }
      for (int i = 0; i < 9; ++i) mi[MI_EERD + i] = body[md.ee_body * 33 + i];
    }
    QM_WAVE_SYNC();
  };
  if (wave != 0) {
    QM_LDS_BARRIER();                                     // inputs and coordinates (S1, S2) are in LDS
#if QM_WBC_EXP == 3
    if (wave == 2) desiredPass(lds + W_BODY2, lds + W_DOF2);     // experiment: the desired pass on helper wavefront 2 instead of 1
#else
    if (wave == 1) desiredPass(lds + W_BODY2, lds + W_DOF2);
#endif
#if QM_WBC_EXP == 4
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // experiment: drain every outstanding memory operation before the fork-join loop
#elif QM_WBC_EXP == 5
    __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);   // experiment: pure timing -- every helper arrives ~16 k cycles later
#endif
    const QpIo hio{G, nullptr, nullptr, DZ, fhat, K, wt, zs, red, forkCmd, nullptr, nullptr};
#if defined(QM_WBC_DUMP) && !defined(QMGPU_HOST_EMULATION)
    int dbgIt = 0;
#endif
    for (;;) {
      QM_LDS_BARRIER();
      const int op = int(forkCmd[0]);
#if defined(QM_WBC_DUMP) && !defined(QMGPU_HOST_EMULATION)
      // what each helper wavefront saw, in order: tail of checkpoint image (wave - 1): per iteration [op, exec lo, exec hi, wave, first active lane, job M, N, K]
      if (blockIdx.x == 0 && dbgIt < 45) {
        const unsigned long long ex = __builtin_amdgcn_read_exec();
        double* o = qmk::qmWbcDump + ((wave - 1) * 17000 + 16640 + dbgIt * 8);
        o[0] = double(op); o[1] = double(unsigned(ex)); o[2] = double(unsigned(ex >> 32)); o[3] = double(wave); o[4] = double(__builtin_amdgcn_readfirstlane(lane));
        o[5] = forkJob[4]; o[6] = forkJob[5]; o[7] = forkJob[6];
      }
      ++dbgIt;
#endif
      if (op == 0) break;
      if (op == 36) ipmKTiles<36, LDZ, LDK>(hio, wave, lane);
      else if (op == 20) ipmKTiles<20, LDZ, LDK>(hio, wave, lane);
      else if (op == 200) ipmColSumShare<LDZ>(hio, wave, lane);
      else if (op == 300) {}                                // join of the desired pass
      else {   // 100 / 101: C = A B / A^T B, described in forkJob (pointers as offsets from the LDS base)
        const double* jA = lds + int(forkJob[0]); const double* jB = lds + int(forkJob[2]); double* jD = lds + int(forkJob[7]);
        const int lda = int(forkJob[1]), ldb = int(forkJob[3]), jM = int(forkJob[4]), jN = int(forkJob[5]), jK = int(forkJob[6]), ldd = int(forkJob[8]);
        if (op == 101) waveGemmTiles<true>(jA, lda, jB, ldb, jM, jN, jK, jD, ldd, forkJob[9], wave, lane, red);
        else waveGemmTiles<false>(jA, lda, jB, ldb, jM, jN, jK, jD, ldd, forkJob[9], wave, lane, red);
      }
      QM_LDS_BARRIER();
    }
    return;
  }
  // C (M x N, LDS) = op(A) B with the four wavefronts sharing the tiles
  auto forkGemm = [&](bool ta, const double* jA, int lda, const double* jB, int ldb, int jM, int jN, int jK, double* jD, int ldd, double diagAdd) {
    if (lane == 0) {
      forkJob[0] = double(jA - lds); forkJob[1] = lda; forkJob[2] = double(jB - lds); forkJob[3] = ldb; forkJob[4] = jM; forkJob[5] = jN; forkJob[6] = jK;
      forkJob[7] = double(jD - lds); forkJob[8] = ldd; forkJob[9] = diagAdd; forkCmd[0] = ta ? 101.0 : 100.0;
    }
    QM_LDS_BARRIER();
    if (ta) waveGemmTiles<true>(jA, lda, jB, ldb, jM, jN, jK, jD, ldd, diagAdd, 0, lane, red);
    else waveGemmTiles<false>(jA, lda, jB, ldb, jM, jN, jK, jD, ldd, diagAdd, 0, lane, red);
    QM_LDS_BARRIER();
  };
  const int mode = a.mode[inst];
  const double period = a.period[inst], time = a.time[inst];
  bool contact[4]; int nst = 0;
  for (int c = 0; c < 4; ++c) { contact[c] = contactOf(mode, c); nst += contact[c] ? 1 : 0; }
  const int nsw = 4 - nst;

  for (int e = lane; e < MAXM * LDZ; e += 64) DZ[e] = 0.0;  // rows >= m0 are never written: the interior point only needs them finite
  QM_TICK_DECL;
#ifdef QM_RICCATI_TIMING
  const unsigned long long qmStart = clock64();
#endif
  // ---- S1: inputs
  if (lane < 55) rbd[lane] = a.rbd[size_t(inst) * 55 + lane];
  if (lane < 30) { xDes[lane] = a.xDes[size_t(inst) * 30 + lane]; uDes[lane] = a.uDes[size_t(inst) * 30 + lane]; il[lane] = a.inputLast[size_t(inst) * 30 + lane]; }
  QM_WAVE_SYNC();
  // ---- S2: Pinocchio coordinates of the measured state (WbcBase.cpp:150-156)
  if (lane == 0) {
    for (int i = 0; i < 3; ++i) { qM[i] = rbd[3 + i]; qM[3 + i] = rbd[i]; vM[i] = rbd[24 + 3 + i]; }
    for (int j = 0; j < 18; ++j) { qM[6 + j] = rbd[6 + j]; vM[6 + j] = rbd[24 + 6 + j]; }
    double sz, cz, sy, cy;
    sincos(qM[3], &sz, &cz); sincos(qM[4], &sy, &cy);
    const double wx = rbd[24], wy = rbd[25], wz = rbd[26];
    const double tmp = cz * wx / cy + sz * wy / cy;
    vM[3] = sy * tmp + wz; vM[4] = -sz * wx + cz * wy; vM[5] = tmp;
    for (int j = 0; j < 18; ++j) { qD[6 + j] = xDes[12 + j]; vD[6 + j] = uDes[12 + j]; mi[MI_JACC + j] = (uDes[12 + j] - il[12 + j]) / period; }
    for (int i = 0; i < 6; ++i) qD[i] = xDes[6 + i];
  }
  if (lane < 30) a.inputLast[size_t(inst) * 30 + lane] = uDes[lane];  // WbcBase.cpp:225
  QM_LDS_BARRIER();      // wavefront 1 starts the desired pass (S5) from here

  QM_TICK(0);
  // ---- S3: measured pass (zero generalized acceleration -> bias terms)
  bodyPass(md, qM, vM, nullptr, body, dof, lane);
  QM_WAVE_SYNC();
  // body wrenches for the nonlinear effects: f = m (a_c + g), n = I alpha + w x I w
  if (lane < QMGPU_NB) {
    const double* o = body + lane * 33;
    double pos[3], vel[3], acc[3], Iw_w[3], Iw_al[3], t[3];
    pointKin(md, body, lane, md.com[lane], pos, vel, acc);
    symMul(o + 15, o + 21, Iw_w); symMul(o + 15, o + 24, Iw_al); cross3(o + 21, Iw_w, t);
    for (int i = 0; i < 3; ++i) { wr[lane * 3 + i] = md.mass[lane] * (acc[i] + (i == 2 ? st.gravity : 0.0)); wr[57 + lane * 3 + i] = Iw_al[i] + t[i]; }
  }
  QM_WAVE_SYNC();
  QM_TICK(1);
  // ---- S4: lane k = generalized velocity k: nle_k, column k of M, Jacobian columns
  // The Jacobian column of every (body b, velocity i) pair at the body's centre of mass is formed ONCE (by lane i, into the LDS that Z / Z_new / A Z
  // take over later) instead of by every lane for every pair: M[i][k] = sum_b J_bi^T diag(m_b, I_b) J_bk is then six multiply-adds per term.
  double* JL = Z;   // [19][24][6] = 2736 doubles over Z, Z_new and the head of A Z (contiguous, all unused before the first level)
  static_assert(W_ZN == W_Z + ND * LDZ && W_AZ == W_ZN + ND * LDZ && QMGPU_NB * NVV * 6 <= 2 * ND * LDZ + MAXR * LDZ, "Jacobian columns fit the Z / Z_new / A Z regions");
  if (lane < NVV) {
#pragma unroll 1
    for (int b = 0; b < QMGPU_NB; ++b) {
      if (!dofMoves(lane, b)) continue;
      double lk[3], ak[3];
      jacCol(dof, lane, b, body + b * 33 + 12, lk, ak);
      double* d = JL + (b * NVV + lane) * 6;
      d[0] = lk[0]; d[1] = lk[1]; d[2] = lk[2]; d[3] = ak[0]; d[4] = ak[1]; d[5] = ak[2];
    }
  }
  QM_WAVE_SYNC();
  if (lane < NVV) {
    const int k = lane;
    double macc[NVV];
#pragma unroll
    for (int i = 0; i < NVV; ++i) macc[i] = 0.0;
    double h = 0.0;
#pragma unroll 1
    for (int b = 0; b < QMGPU_NB; ++b) {
      if (!dofMoves(k, b)) continue;
      const double* o = body + b * 33;
      const double* jk = JL + (b * NVV + k) * 6;
      const double lk[3] = {jk[0], jk[1], jk[2]}, ak[3] = {jk[3], jk[4], jk[5]};
      double Fk[3], Nk[3];
      h += dot3(lk, wr + b * 3) + dot3(ak, wr + 57 + b * 3);
      for (int i = 0; i < 3; ++i) Fk[i] = md.mass[b] * lk[i];
      symMul(o + 15, ak, Nk);
#pragma unroll
      for (int i = 0; i < NVV; ++i) {
        if (dofMoves(i, b)) {   // wave uniform for the lanes of this iteration (b is per lane, but every live lane of a given b agrees)
          const double* ji = JL + (b * NVV + i) * 6;
          macc[i] += (ji[0] * Fk[0] + ji[1] * Fk[1] + ji[2] * Fk[2]) + (ji[3] * Nk[0] + ji[4] * Nk[1] + ji[5] * Nk[2]);
        }
      }
    }
    const double nleRaw = h;
#pragma unroll
    for (int i = 0; i < NVV; ++i) M[i * NVV + k] = macc[i];
    for (int c = 0; c < 4; ++c) {
      const int b = md.foot_body[c];
      double pos[3], vel[3], acc[3], lin[3], ang[3];
      pointKin(md, body, b, md.foot_offset[c], pos, vel, acc);
      jacCol(dof, k, b, pos, lin, ang);
      for (int r = 0; r < 3; ++r) Jf[(3 * c + r) * NVV + k] = lin[r];
    }
    {
      double pos[3], vel[3], acc[3], lin[3], ang[3];
      pointKin(md, body, md.ee_body, md.ee_offset, pos, vel, acc);
      jacCol(dof, k, md.ee_body, pos, lin, ang);
      for (int r = 0; r < 3; ++r) { Ja[r * NVV + k] = lin[r]; Ja[(3 + r) * NVV + k] = ang[r]; }
      // M qdd + nle = S^T tau + Jc^T F + Jee^T f_e: the external end-effector force is folded into nle (equations of motion, torque limits, torque recovery)
      nle[k] = nleRaw - (lin[0] * fe[0] + lin[1] * fe[1] + lin[2] * fe[2]);
    }
  }
  if (lane >= 32 && lane < 36) {  // feet: position, velocity, dJ v
    const int c = lane - 32;
    double pos[3], vel[3], acc[3];
    pointKin(md, body, md.foot_body[c], md.foot_offset[c], pos, vel, acc);
    for (int r = 0; r < 3; ++r) { mi[MI_FOOTPM + 3 * c + r] = pos[r]; mi[MI_FOOTVM + 3 * c + r] = vel[r]; mi[MI_FOOTDJV + 3 * c + r] = acc[r]; }
  }
  if (lane == 40) {  // arm end-effector + base angular bias
    double pos[3], vel[3], acc[3];
    pointKin(md, body, md.ee_body, md.ee_offset, pos, vel, acc);
    const double* o = body + md.ee_body * 33;
    for (int r = 0; r < 3; ++r) {
      mi[MI_EEPM + r] = pos[r]; mi[MI_EEVM + r] = vel[r]; mi[MI_EEDJL + r] = acc[r]; mi[MI_EEWM + r] = o[21 + r];
      mi[MI_AL0 + r] = body[24 + r];
      for (int j = 0; j < 3; ++j) mi[MI_BAX + 3 * j + r] = dof[(3 + j) * 3 + r];
      mi[MI_EEDJA + r] = o[24 + r] - body[24 + r];  // (dJ_ang with columns 3..5 zeroed) v  (WbcBase.cpp:550-553)
    }
    for (int i = 0; i < 9; ++i) mi[MI_EERM + i] = o[i];
  }
  QM_WAVE_SYNC();

  QM_TICK(2);
  // ---- S5 runs on wavefront 1 (desiredPass above); join: its results are in LDS once everybody has passed this pair of barriers
  if (lane == 0) forkCmd[0] = 300.0;
  QM_LDS_BARRIER();
  QM_LDS_BARRIER();

  QM_WBC_CHECKPOINT(0);   // model, Jacobians, desired pass
  QM_TICK(3);
  // ================================================================== hierarchical QP
  int status = 0;
  // x = 0, Z = I
  for (int e = lane; e < ND * LDZ; e += 64) Z[e] = ((e / LDZ) == (e % LDZ)) ? 1.0 : 0.0;
  if (lane < ND) xs[lane] = 0.0;
  // ---- task 0 inequality rows (kept hard, with their slacks, by the lower levels): torque limits + friction pyramid (+ zero rows)
  const int m0 = 36 + 5 * nst + 3 * nsw;
  for (int e = lane; e < MAXM * ND; e += 64) D0[e] = 0.0;
  QM_WAVE_SYNC();
  // WbcBase.cpp:392-415; the LF leg limits are reused for every leg (WbcBase.cpp:599-600).  Rows i and 18 + i = +-[M_joint | -J_joint^T], element by element
  for (int e = lane; e < 18 * ND; e += 64) {
    const int i = e / ND, j = e - i * ND;
    const double v = j < NVV ? M[(6 + i) * NVV + j] : -Jf[(j - NVV) * NVV + 6 + i];
    D0[e] = v; D0[18 * ND + e] = -v;
  }
  if (lane < 18) {
    const int i = lane;
    const double lim = i < 12 ? md.effort_limit[i % 3] : md.effort_limit[i];
    f0[i] = lim - nle[6 + i]; f0[18 + i] = lim + nle[6 + i];
  }
  if (lane == 32) {  // WbcBase.cpp:439-469
    const double mu = st.wbc_friction_coefficient;
    int j = 0;
    for (int c = 0; c < 4; ++c) if (contact[c]) {
      double* r = D0 + (36 + 5 * j) * ND + 24 + 3 * c;
      r[2] = -1.0; r[ND] = 1.0; r[ND + 2] = -mu; r[2 * ND] = -1.0; r[2 * ND + 2] = -mu; r[3 * ND + 1] = 1.0; r[3 * ND + 2] = -mu; r[4 * ND + 1] = -1.0; r[4 * ND + 2] = -mu;
      ++j;
    }
    for (int r = 36; r < m0; ++r) f0[r] = 0.0;
  }
  QM_WAVE_SYNC();

  // scratch of the implied-equality step: the body / wrench tables of the model update are free by now (the desired pass has joined)
  double* scrA = lds + W_BODY;       // 904 doubles: W_BODY | W_DOF | W_WR
  double* scrB = lds + W_BODY2;      // 784 doubles: W_BODY2 | W_DOF2
  static_assert(W_DOF == W_BODY + 640 && W_WR == W_DOF + 144 && W_M == W_WR + 120 && 24 * LDZ <= 904 && MAXR * LDZ <= 904 && 18 * LDZ <= 784, "scratch regions of the implied-equality step");
  static_assert(QP_KMAX * QP_SLD <= MAXR * 40, "the small system of the pinned rows fits the table region");
  // rows of `cnt` x n  <-  rows N_E (N_E: n x nE in K), in place, 24 rows at a time through scrA; columns >= nE cleared
  auto rightMultiply = [&](double* rowsP, int cnt, int n, int nE) {
#pragma unroll 1
    for (int r0 = 0; r0 < cnt; r0 += 24) {
      const int c = cnt - r0 < 24 ? cnt - r0 : 24;
      forkGemm(false, rowsP + r0 * LDZ, LDZ, K, LDK, c, nE, n, scrA, LDZ, 0.0);
      QM_WAVE_SYNC();
      for (int e = lane; e < c * LDZ; e += 64) { const int j = e % LDZ; rowsP[r0 * LDZ + e] = j < nE ? scrA[e] : 0.0; }
      QM_WAVE_SYNC();
    }
  };
  // The QP of one level (or of its canonical representative) in nVars variables: task rows AZp (rRows x nVars) with residual rhatp at z = 0, inequality rows DZ / fhat
  // (own: the level's own, soft; else inherited, hard).  Rows a higher level left strongly active (eqIn) are equalities here: removed exactly by the change of variables
  // z = N_E w (N_E = kernel of those rows; DESIGN.md section 4.7 has the argument), the QP is solved in w.  Result: z in zs[0 .. nVars);
  // strongOut: this lane's row is strongly active at the solution; returns the solver's status.  AZp and DZ are overwritten when rows are eliminated.
  // warmIo (in / out): the word of this solve in the instance's working-set record (0: none / cold); passes gets bit 7 when the carried guess was refuted.
  // warmZ (global memory or null): where the solution of this solve travels with its rows (bit 62 of the word; not when implied equalities changed the variables).
  // regular: a level's own solve (HoQp's regulariser applies: kept literally where qp_dev.h's `lit` says), not the canonical representative's.
  auto levelQp = [&](double* AZp, int rRows, double* rhatp, int nVars, bool own, bool rowOnIn, bool eqIn, bool& strongOut, int& passes, unsigned long long& warmIo, double* warmZ, bool regular) -> int {
    int nQ = nVars;
    bool rowOn = rowOnIn, reduced = false;
    strongOut = false;
    const unsigned long long warmIn = warmIo;
    warmIo = 0ull;
    if (!own) {
      unsigned long long eqMask = qmBallot(rowOn && eqIn);
      if (eqMask != 0ull) {
        // (at most MAXR rows go into the elimination -- the rest, necessarily combinations of them in <= 18 variables, stay inequality rows)
        int slotE = qmPopCount(eqMask & ((1ull << lane) - 1ull));
        const bool mineE = rowOn && eqIn && slotE < MAXR;
        const int kE = qmPopCount(eqMask) < MAXR ? qmPopCount(eqMask) : MAXR;
        if (mineE) for (int j = 0; j < LDZ; ++j) scrA[slotE * LDZ + j] = j < nVars ? DZ[lane * LDZ + j] : 0.0;
        QM_WAVE_SYNC();
        nQ = wbcNullSpace(int(scrA - lds), kE, nVars, int(K - lds), int(Vh - lds), int(red - lds), lane);
        if (nQ == 0) { if (lane < ND) zs[lane] = 0.0; QM_WAVE_SYNC(); return 0; }     // the equalities leave nothing to decide
        for (int e = lane; e < nVars * LDZ; e += 64) { const int i = e / LDZ, j = e - i * LDZ; scrB[e] = j < nQ ? K[i * LDK + j] : 0.0; }     // N_E survives the solve in scrB
        double dnOld = 0.0, dnNew = 0.0;
        if (lane < m0) for (int j = 0; j < nVars; ++j) dnOld = fmax(dnOld, fabs(DZ[lane * LDZ + j]));
        QM_WAVE_SYNC();
        rightMultiply(AZp, rRows, nVars, nQ);
        rightMultiply(DZ, m0, nVars, nQ);
        if (lane < m0) for (int j = 0; j < nQ; ++j) dnNew = fmax(dnNew, fabs(DZ[lane * LDZ + j]));
        // rows that are combinations of the eliminated ones vanish up to rounding in the new variables: they stay tight, and carry no information
        rowOn = rowOn && !mineE && dnNew > 1e-12 * dnOld;
        if (lane < m0 && !rowOn) for (int j = 0; j < LDZ; ++j) DZ[lane * LDZ + j] = 0.0;
        QM_WAVE_SYNC();
        reduced = true;
      }
    }
    // ---- minimum-norm start of a level with own rows only (the first level; DESIGN.md section 4.7 has the argument): its task can be met exactly, away from the
    //      limits its minimisers are the solutions of A Z z = -rhat, and the one taken is the one of smallest weighted norm -- the variables the zero-bound rows act on
    //      (the contact forces under their cones) 1e4 times cheaper than the rest, so that forces carry the robot and accelerations stay small --
    //        z = W^-1 (A Z)' y,   (A Z) W^-1 (A Z)' y = -rhat        (a Cholesky of the size of the TASK, 18, in LDS: no 36 x 36 factorisation, no working set)
    //      kept if every own row is satisfied at it.
    if (own && rRows <= nQ && rRows <= MAXR) {
      double winv = 1.0;
      if (lane < nQ) for (int i = 0; i < m0; ++i) if (fhat[i] == 0.0 && DZ[i * LDZ + lane] != 0.0) winv = 1e4;
      // B = W^-1 (A Z)' (n x r) in Zn (free until the level's null space), then (A Z) B on the matrix cores into K
      for (int e = lane; e < nQ * rRows; e += 64) { const int c = e / rRows, j = e - c * rRows; Zn[c * LDZ + j] = AZp[j * LDZ + c]; }
      QM_WAVE_SYNC();
      if (lane < nQ) for (int j = 0; j < rRows; ++j) Zn[lane * LDZ + j] *= winv;
      for (int e = lane; e < ND * LDK; e += 64) K[e] = 0.0;
      QM_WAVE_SYNC();
      forkGemm(false, AZp, LDZ, Zn, LDZ, rRows, rRows, nQ, K, LDK, 0.0);
      QM_WAVE_SYNC();
      const double dmax = qmAllMax(lane < rRows ? K[lane * LDK + lane] : 0.0, red);
      // plain Cholesky in LDS (lane = row); a pivot lost against the diagonal (dependent task rows) ends the attempt
      bool ok = true;
#pragma unroll 1
      for (int j = 0; j < rRows; ++j) {
        const double d = K[j * LDK + j];
        if (!(d > 1e-10 * dmax)) { ok = false; break; }
        const double dj = sqrt(d);
        QM_WAVE_SYNC();
        if (lane == j) K[j * LDK + j] = dj;
        else if (lane > j && lane < rRows) K[lane * LDK + j] = K[lane * LDK + j] / dj;
        QM_WAVE_SYNC();
        if (lane > j && lane < rRows) {
          const double lij = K[lane * LDK + j];
          for (int q = j + 1; q <= lane; ++q) K[lane * LDK + q] -= lij * K[q * LDK + j];
        }
        QM_WAVE_SYNC();
      }
      if (ok) {
        if (lane < rRows) dzs[lane] = -rhatp[lane];
        QM_WAVE_SYNC();
        ldsCholSolve(K, rRows, dzs, lane);
        QM_WAVE_SYNC();
        double zc = 0.0;
        if (lane < nQ) { for (int q = 0; q < rRows; ++q) zc += AZp[q * LDZ + lane] * dzs[q]; zc *= winv; }
        QM_WAVE_SYNC();
        if (lane < ND) zs[lane] = lane < nQ ? zc : 0.0;
        QM_WAVE_SYNC();
        double res = 0.0, rsc = 1.0, dzr = 0.0;
        if (lane < rRows) { res = rhatp[lane]; for (int c = 0; c < nQ; ++c) res += AZp[lane * LDZ + c] * zs[c]; rsc = fabs(rhatp[lane]); }
        if (lane < m0) { for (int c = 0; c < nQ; ++c) dzr += DZ[lane * LDZ + c] * zs[c]; dzr -= fhat[lane]; }
        const double resmax = qmAllMax(fabs(res), red), rscale = fmax(1.0, qmAllMax(rsc, red));
        const bool viol = rowOn && !(dzr <= 0.0);
        const double nanProbe = qmAllSum(zc, red);
        ok = resmax <= 1e-9 * rscale && qmBallot(viol) == 0ull && nanProbe == nanProbe;
      }
      if (ok) { passes = 0; strongOut = false; return 0; }
    }
    // G = (A Z)'(A Z) (HoQp.cpp:60-76, without its 1e-12 I: qp_dev.h)
    for (int e = lane; e < ND * LDK; e += 64) G[e] = 0.0;
    QM_WAVE_SYNC();
    forkGemm(true, AZp, LDZ, AZp, LDZ, nQ, nQ, rRows, G, LDK, 0.0);
    QM_WAVE_SYNC();
    const QpOff io{int(G - lds), int(AZp - lds), int(rhatp - lds), int(DZ - lds), int(fhat - lds), int(K - lds), int(wt - lds), int(zs - lds), int(red - lds), int(forkCmd - lds), int(Vh - lds), W_TP};
    auto solve = [&](bool tryHeld, bool ownIpm) {
      QpResult rr;
      const double sigma0 = ownIpm ? 0.5 : ((own || nQ <= 8) ? -1.0 : 0.5);         // (small levels go without the interior point: cold, the active-set method is shorter there in mean and in the tail)
      const double* wz = reduced ? nullptr : warmZ;
      const bool lit = regular && !own && !reduced && nQ <= 12;      // (= LevelQp::lit of the CPU restatement: kLiteralRegMaxN)
      if (nQ <= 8) rr = qpSolve<8, LDZ, LDK>(io, nQ, rRows, m0, own, rowOn, sigma0, tryHeld, warmIn, wz, lit, lane);
      else if (nQ <= 20) rr = qpSolve<20, LDZ, LDK>(io, nQ, rRows, m0, own, rowOn, sigma0, tryHeld, warmIn, wz, lit, lane);
      else rr = qpSolve<36, LDZ, LDK>(io, nQ, rRows, m0, own, rowOn, sigma0, tryHeld, warmIn, wz, false, lane);
      QM_WAVE_SYNC();
      return rr;
    };
    QpResult res = solve(own, false);                // own rows: first with the variables of the zero-bound rows held (qp_dev.h) ...
    if (own && res.status != 0) res = solve(false, res.heldTried);  // ... and, if the cost wants them moved (or that form takes more than QP_HELD_CAP iterations), with those rows as rows behind the interior point
    passes = (res.ipmIterations + res.iterations < 127 ? res.ipmIterations + res.iterations : 127) | (res.warmRefuted ? 128 : 0);
    warmIo = res.pinMask;
    if (warmIo != 0ull && warmZ != nullptr && !reduced) { warmIo |= 1ull << 62; if (lane < nQ) warmZ[lane] = zs[lane]; }     // (the solution travels with the rows)
    strongOut = rowOn && res.strong;
    if (reduced) {   // z = N_E w
      double zf = 0.0;
      if (lane < nVars) for (int j = 0; j < nQ; ++j) zf += scrB[lane * LDZ + j] * zs[j];
      QM_WAVE_SYNC();
      if (lane < ND) zs[lane] = lane < nVars ? zf : 0.0;
      QM_WAVE_SYNC();
    }
    return res.status;
  };
  // margins of the inequality rows at the current x, with the slack the first level left them (>= 0 up to rounding: clamped, as in the oracle's HoQp)
  auto marginsAtX = [&]() {
    if (lane < m0) { double s = f0[lane]; for (int q = 0; q < ND; ++q) s -= D0[lane * ND + q] * xs[q]; s += v0[lane]; fhat[lane] = fmax(s, 0.0); }
  };
  auto rowNonZero = [&](int n) { bool nz = false; if (lane < m0) for (int j = 0; j < n; ++j) nz = nz || DZ[lane * LDZ + j] != 0.0; return nz; };
  // row-major copy (stride ND -> stride LDZ) of `rows` rows: what the products with Z = I of the first level are, entry for entry (x * 1 + 0 + ... is exact)
  auto copyRows = [&](const double* src, double* dst, int rows) {
#pragma unroll 1
    for (int e0 = 0; e0 < rows * ND; e0 += 8 * 64) {
      double t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int e = e0 + 64 * q + lane; t[q] = src[e < rows * ND ? e : 0]; }
#pragma unroll
      for (int q = 0; q < 8; ++q) QM_KEEP(t[q]);
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int e = e0 + 64 * q + lane; if (e < rows * ND) dst[(e / ND) * LDZ + (e % ND)] = t[q]; }
    }
  };

  // Directions no task sees are fixed in the reference by HoQp's 1e-12 I alone (HoQp.cpp:66): every level returns, among its minimisers, the one of smallest norm in its
  // own variables z.  Where the last level decides everything that is left (every gait of gait.info once the start-up branch is over) that choice is invisible -- the next
  // level re-decides the same directions -- and the cascade runs without it (pass 0).  Where directions are left over at the end, the cascade runs again with the
  // canonical representative taken at every level (pass 1; DESIGN.md section 4.7 has the argument).
  int n = ND;
  // (HierarchicalMpcWbc: the canonical representative at every level from the first pass on -- its last level sees one arm direction through a curvature below the regulariser's,
  //  so where the level above left that direction shows in the answer; the CPU restatement's wbcUpdate has the measurement)
  bool canonical = a.variant == 1;
  // The working sets of the previous tick (one word per solve: [1 + 6 pass + 2 level + completion]) are guesses for this one as long as the rows mean the same thing:
  // same contact mode, controller and task set (word 0); anything else starts cold.  Words 13 / 14: passes of every solve of this tick (a byte each, bit 7 = guess refuted).
  unsigned long long* wsRec = a.workingSet ? a.workingSet + size_t(inst) * QMGPU_WBC_STATE_WORDS : nullptr;
  const unsigned long long wsKey = (1ull << 63) | (unsigned long long)(mode & 15) | ((unsigned long long)(a.variant & 1) << 8) | ((a.variant == 0 && time < 10.0) ? (1ull << 9) : 0ull);
  const bool wsLive = wsRec != nullptr && wsRec[0] == wsKey;
  unsigned long long wsCount[2] = {0ull, 0ull};
  auto wsLoad = [&](int pass, int level, int completion) { return wsLive ? wsRec[1 + 6 * pass + 2 * level + completion] : 0ull; };
  auto wsStore = [&](int pass, int level, int completion, unsigned long long word, int passes) {
    wsCount[pass] |= (unsigned long long)(passes & 255) << (8 * (2 * level + completion));
    if (wsRec && lane == 0) wsRec[1 + 6 * pass + 2 * level + completion] = word;
  };
  if (wsRec && !wsLive && lane < QMGPU_WBC_STATE_WORDS) wsRec[lane] = lane == 0 ? wsKey : 0ull;     // (the loads above are wave uniform and precede this store in program order)
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
  status = 0; n = ND;
  // x = 0, Z = I
  for (int e = lane; e < ND * LDZ; e += 64) Z[e] = ((e / LDZ) == (e % LDZ)) ? 1.0 : 0.0;
  if (lane < ND) xs[lane] = 0.0;
  if (lane < MAXM) v0[lane] = 0.0;
  bool eqRow = false;     // this lane's inequality row is strongly active at some level so far: an equality for the levels below
  QM_WAVE_SYNC();
#pragma unroll 1
  for (int level = 0; level < 3; ++level) {
    if (n == 0) break;  // FLY: nothing left to decide
    QM_TICK(4);
    // ---- assemble this level's equality task A x = b  (rows r)
    for (int e = lane; e < MAXR * ND; e += 64) A[e] = 0.0;
    QM_WAVE_SYNC();
    int r = 0;
    if (level == 0) {
      r = 18;
      if (lane < 6) {  // floating-base equations of motion (WbcBase.cpp:370-388)
        for (int j = 0; j < NVV; ++j) A[lane * ND + j] = M[lane * NVV + j];
        for (int j = 0; j < 12; ++j) A[lane * ND + 24 + j] = -Jf[j * NVV + lane];
        bvec[lane] = -nle[lane];
      }
      {   // rows of Jacobians are copied by one lane per column (lanes 32..55), the right-hand sides by lane 56
        const int jc = lane - 32;
        int row = 6;
        for (int c = 0; c < 4; ++c) if (contact[c]) {  // no contact motion (WbcBase.cpp:418-433)
          for (int q = 0; q < 3; ++q) {
            if (jc >= 0 && jc < NVV) A[(row + q) * ND + jc] = Jf[(3 * c + q) * NVV + jc];
            if (lane == 56) bvec[row + q] = -mi[MI_FOOTDJV + 3 * c + q];
          }
          row += 3;
        }
        for (int c = 0; c < 4; ++c) if (!contact[c]) {  // swing feet carry no force (WbcBase.cpp:440-449)
          if (lane == 56) for (int q = 0; q < 3; ++q) { A[(row + q) * ND + 24 + 3 * c + q] = 1.0; bvec[row + q] = 0.0; }
          row += 3;
        }
      }
    } else if (level == 1) {
      const bool startup = a.variant == 0 && time < 10.0;  // HierarchicalWbc.cpp:32-37
      if (startup) {
        r = 6;
        if (lane < 6) {  // arm joint tracking (WbcBase.cpp:471-497)
          A[lane * ND + 18 + lane] = 1.0;
          bvec[lane] = st.kp_arm_joint[lane] * (qD[18 + lane] - qM[18 + lane]) + st.kd_arm_joint[lane] * (vD[18 + lane] - vM[18 + lane]);
        }
      } else {
        const int extra = a.variant == 0 ? 6 : 2;
        r = 4 + extra + 3 * nsw;
        if (lane == 0) {
          // base height (WbcBase.cpp:308-320)
          A[2] = 1.0;
          bvec[0] = mi[MI_BACC + 2] + st.kp_base_height * (qD[2] - qM[2]) + st.kd_base_height * (vD[2] - vM[2]);
          // base angular (WbcBase.cpp:270-305): Euler maps at the MEASURED angles
          for (int q = 0; q < 3; ++q) for (int j = 3; j < 6; ++j) A[(1 + q) * ND + j] = mi[MI_BAX + 3 * (j - 3) + q];
          double sz, cz, sy, cy;
          sincos(qM[3], &sz, &cz); sincos(qM[4], &sy, &cy);
          auto omegaOf = [&](const double* de, double* w) { w[0] = -sz * de[1] + cy * cz * de[2]; w[1] = cz * de[1] + cy * sz * de[2]; w[2] = de[0] - sy * de[2]; };
          double wM[3], wD[3], acc[3];
          omegaOf(vM + 3, wM); omegaOf(vD + 3, wD);
          {
            const double* de = vD + 3; const double* dde = mi + MI_BACC + 3;
            const double szt = cz * de[0], czt = -sz * de[0], syt = cy * de[1], cyt = -sy * de[1];
            acc[0] = -sz * dde[1] + cy * cz * dde[2] - szt * de[1] + (cyt * cz + cy * czt) * de[2];
            acc[1] = cz * dde[1] + cy * sz * dde[2] + czt * de[1] + (cyt * sz + cy * szt) * de[2];
            acc[2] = dde[0] - sy * dde[2] - syt * de[2];
          }
          // rotation error log(R_des R_meas^T)
          double Rd[9], Rm[9];
          {
            double s3, c3, s4, c4, s5, c5;
            sincos(qD[3], &s3, &c3); sincos(qD[4], &s4, &c4); sincos(qD[5], &s5, &c5);
            const double t[9] = {c3 * c4, c3 * s4 * s5 - s3 * c5, c3 * s4 * c5 + s3 * s5, s3 * c4, s3 * s4 * s5 + c3 * c5, s3 * s4 * c5 - c3 * s5, -s4, c4 * s5, c4 * c5};
            for (int i = 0; i < 9; ++i) Rd[i] = t[i];
            sincos(qM[5], &s5, &c5);
            const double u[9] = {cz * cy, cz * sy * s5 - sz * c5, cz * sy * c5 + sz * s5, sz * cy, sz * sy * s5 + cz * c5, sz * sy * c5 - cz * s5, -sy, cy * s5, cy * c5};
            for (int i = 0; i < 9; ++i) Rm[i] = u[i];
          }
          auto rotErr = [&](const double* L, const double* Rr, double* e) {
            double E[9];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) E[i * 3 + j] = L[i * 3] * Rr[j * 3] + L[i * 3 + 1] * Rr[j * 3 + 1] + L[i * 3 + 2] * Rr[j * 3 + 2];
            const double tr = E[0] + E[4] + E[8];
            const double cth = fmax(-1.0, fmin(1.0, 0.5 * (tr - 1.0)));
            const double th = acos(cth);
            const double kk = th < 1e-4 ? 0.5 + th * th / 12.0 : 0.5 * th / sin(th);
            e[0] = kk * (E[7] - E[5]); e[1] = kk * (E[2] - E[6]); e[2] = kk * (E[3] - E[1]);
          };
          double err[3];
          rotErr(Rd, Rm, err);
          for (int q = 0; q < 3; ++q) bvec[1 + q] = acc[q] + st.kp_base_angular * err[q] + st.kd_base_angular * (wD[q] - wM[q]) - mi[MI_AL0 + q];
          int row = 4;
          if (a.variant == 0) {
            // end-effector linear (WbcBase.cpp:499-524) and angular (WbcBase.cpp:526-563: columns 3..5 zeroed, desired angular velocity unused)
            double eerr[3];
            rotErr(mi + MI_EERD, mi + MI_EERM, eerr);
            for (int q = 0; q < 3; ++q) {
              bvec[row + q] = st.kp_ee_linear[q] * (mi[MI_EEPD + q] - mi[MI_EEPM + q]) + st.kd_ee_linear[q] * (mi[MI_EEVD + q] - mi[MI_EEVM + q]) - mi[MI_EEDJL + q];
              bvec[row + 3 + q] = st.kp_ee_angular[q] * eerr[q] + st.kd_ee_angular[q] * (-mi[MI_EEWM + q]) - mi[MI_EEDJA + q];
            }
            row += 6;
          } else {
            for (int q = 0; q < 2; ++q) {  // base linear (WbcBase.cpp:240-252)
              A[(row + q) * ND + q] = 1.0;
              bvec[row + q] = mi[MI_BACC + q] + st.kp_base_linear * (qD[q] - qM[q]) + st.kd_base_linear * (vD[q] - vM[q]);
            }
            row += 2;
          }
          for (int c = 0; c < 4; ++c) if (!contact[c]) {  // swing legs, weight 100 (WbcBase.cpp:323-346, HierarchicalWbc.cpp:29)
            for (int q = 0; q < 3; ++q) {
              const double acc2 = st.kp_swing * (mi[MI_FOOTPD + 3 * c + q] - mi[MI_FOOTPM + 3 * c + q]) + st.kd_swing * (mi[MI_FOOTVD + 3 * c + q] - mi[MI_FOOTVM + 3 * c + q]);
              bvec[row + q] = 100.0 * (acc2 - mi[MI_FOOTDJV + 3 * c + q]);
            }
            row += 3;
          }
        }
        if (lane >= 32 && lane < 32 + NVV) {   // the Jacobian rows of the same tasks, one lane per column (lane 0 is busy with the right-hand sides)
          const int j = lane - 32;
          int row = 4;
          if (a.variant == 0) {
            for (int q = 0; q < 3; ++q) { A[(row + q) * ND + j] = Ja[q * NVV + j]; A[(row + 3 + q) * ND + j] = (j >= 3 && j < 6) ? 0.0 : Ja[(3 + q) * NVV + j]; }
            row += 6;
          } else row += 2;
          for (int c = 0; c < 4; ++c) if (!contact[c]) {
            for (int q = 0; q < 3; ++q) A[(row + q) * ND + j] = 100.0 * Jf[(3 * c + q) * NVV + j];
            row += 3;
          }
        }
      }
    } else {
      r = a.variant == 0 ? 14 : 12;
      if (lane < 12) { A[lane * ND + 24 + lane] = 1.0; bvec[lane] = uDes[lane]; }  // contact forces (WbcBase.cpp:566-578)
      if (a.variant == 0 && lane >= 12 && lane < 14) {  // base linear (WbcBase.cpp:240-252)
        const int q = lane - 12;
        A[lane * ND + q] = 1.0;
        bvec[lane] = mi[MI_BACC + q] + st.kp_base_linear * (qD[q] - qM[q]) + st.kd_base_linear * (vD[q] - vM[q]);
      }
    }
    QM_WAVE_SYNC();

#ifdef QMGPU_EMU_DEBUG
    if (lane == 0 && inst == QMGPU_DEBUG_INST) { printf("EMU level %d r %d n %d b:", level, r, n); for (int i = 0; i < r; ++i) printf(" %.10g", bvec[i]); printf("\n"); }
#endif
    if (level == 0) QM_WBC_CHECKPOINT(1);   // task 0: A, b, D0, f0
    QM_TICK(5);
    // ---- reduced data: AZ = A Z (r x n), rhat = A x - b, DZ = D0 Z, fhat
    for (int e = lane; e < MAXR * LDZ; e += 64) AZ[e] = 0.0;
    for (int e = lane; e < MAXM * LDZ; e += 64) DZ[e] = 0.0;     // columns >= n stay zero padding (the solver always spans whole tiles); rows >= m0 finite
    QM_WAVE_SYNC();
    if (level == 0) { copyRows(A, AZ, r); copyRows(D0, DZ, m0); }   // Z = I (n = ND)
    else { forkGemm(false, A, ND, Z, LDZ, r, n, ND, AZ, LDZ, 0.0); forkGemm(false, D0, ND, Z, LDZ, m0, n, ND, DZ, LDZ, 0.0); }
    QM_WAVE_SYNC();
    QM_TICK(12);
    if (lane < r) { double s = -bvec[lane]; for (int q = 0; q < ND; ++q) s += A[lane * ND + q] * xs[q]; tzv[lane] = s; }   // A x_prev - b
    if (level == 0) { if (lane < m0) fhat[lane] = f0[lane]; } else marginsAtX();
    QM_WAVE_SYNC();
    QM_TICK(13);
    if (level == 0) QM_WBC_CHECKPOINT(2);   // reduced data of level 0: AZ, DZ, fhat
    QM_TICK(6);
    // ---- the level's QP (qp_dev.h)
    bool strong = false; int passes = 0;
    const bool hadEq = level > 0 && qmBallot(eqRow && rowNonZero(n)) != 0ull;
    unsigned long long wsWord = wsLoad(pass, level, 0);
    // (words 16..33 / 34..41 of the record: the solutions of the second and third level of pass 0)
    double* wsZ = (wsRec && pass == 0 && ((level == 1 && n <= 18) || (level == 2 && n <= 8))) ? reinterpret_cast<double*>(wsRec + (level == 1 ? 16 : 34)) : nullptr;
    const int st = levelQp(AZ, r, tzv, n, level == 0, rowNonZero(n), eqRow, strong, passes, wsWord, wsZ, true);
    wsStore(pass, level, 0, wsWord, passes);
    eqRow = eqRow || strong;
    if (st != 0) status |= (1 << level);
#ifdef QM_RICCATI_TIMING
    if (lane == 0 && inst < 256) qmk::qmRiccatiTicks[512 + inst * 4 + 1 + level] = (unsigned long long)passes;
#endif
    if (level == 0) QM_WBC_CHECKPOINT(3);   // z of level 0
    QM_TICK(7);
    // ---- x = x_prev + Z z (HoQp.h:31-34); the level's own variables stay in rds (the canonical representative is taken relative to them)
    double xn = 0.0;
    if (lane < ND) { xn = xs[lane]; for (int j = 0; j < n; ++j) xn += Z[lane * LDZ + j] * zs[j]; rds[lane] = zs[lane]; }
    QM_WAVE_SYNC();
    if (lane < ND) xs[lane] = xn;
    QM_WAVE_SYNC();
    // slack solution of task 0 (HoQp.cpp:152-158): v = max(0, D x - f), exactly
    if (level == 0 && lane < m0) { double s = -f0[lane]; for (int q = 0; q < ND; ++q) s += D0[lane * ND + q] * xs[q]; v0[lane] = fmax(0.0, s); }
    QM_WAVE_SYNC();
#ifdef QMGPU_EMU_DEBUG
    if (lane == 0 && inst == QMGPU_DEBUG_INST) { printf("EMU level %d passes %d status %d x:", level, passes, st); for (int i = 0; i < 36; ++i) printf(" %.10g", xs[i]); printf("\n"); }
#endif
    if (level == 0) QM_WBC_CHECKPOINT(4);   // x, v0 after level 0
    QM_TICK(8);
    // ---- Z <- Z kernel(A Z) (HoQp.cpp:126-133); A Z again where the implied equalities overwrote it
    if (hadEq) { for (int e = lane; e < MAXR * LDZ; e += 64) AZ[e] = 0.0; QM_WAVE_SYNC(); forkGemm(false, A, ND, Z, LDZ, r, n, ND, AZ, LDZ, 0.0); QM_WAVE_SYNC(); }
    const int nOld = n;
    const int nNew = wbcNullSpace(int(AZ - lds), r, n, int(K - lds), int(Vh - lds), int(red - lds), lane);
    QM_TICK(16);
    // Z N on the matrix cores (columns >= n of Z are zero, rows >= n of N too)
    forkGemm(false, Z, LDZ, K, LDK, ND, nNew, n, Zn, LDZ, 0.0);
    QM_WAVE_SYNC();
    if (canonical && nNew > 0 && level > 0) {   // N itself is the task matrix of the canonical representative (level 0: Z = I, so Z N = N stays in Zn)
      for (int e = lane; e < MAXR * LDZ; e += 64) { const int i = e / LDZ, j = e - i * LDZ; AZ[e] = (i < nOld && j < nNew) ? K[i * LDK + j] : 0.0; }
      QM_WAVE_SYNC();
    }
    {   // Z <- Z N with the columns >= nNew cleared: all loads first (a store between two loads of LDS serialises them), the column index carried along instead of e % LDZ
      constexpr int NIT = (ND * LDZ + 63) / 64;
      double t[NIT];
#pragma unroll
      for (int i = 0; i < NIT; ++i) { const int e = lane + 64 * i; t[i] = Zn[e < ND * LDZ ? e : 0]; }
#pragma unroll
      for (int i = 0; i < NIT; ++i) QM_KEEP(t[i]);
      int col = lane >= LDZ ? lane - LDZ : lane;
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int e = lane + 64 * i;
        if (e < ND * LDZ) Z[e] = col < nNew ? t[i] : 0.0;
        col += 64 - LDZ; if (col >= LDZ) col -= LDZ;
      }
    }
    n = nNew;
    QM_WAVE_SYNC();
    QM_TICK(17);
    if (level == 0) QM_WBC_CHECKPOINT(5);   // Z after the first null space
    if (level == 1) QM_WBC_CHECKPOINT(6);
    // ---- canonical representative of the level (pass 1): min |z* + N w|^2 inside the inequality rows, i.e. task rows N (nOld x n), residual z* (rds); x += Z w
    if (canonical && n > 0) {
      double* AZc = level == 0 ? Zn : AZ;
      if (level == 0) { for (int e = lane; e < ND * LDZ; e += 64) { const int j = e % LDZ; if (j >= n) Zn[e] = 0.0; } }
      for (int e = lane; e < MAXM * LDZ; e += 64) DZ[e] = 0.0;
      QM_WAVE_SYNC();
      forkGemm(false, D0, ND, Z, LDZ, m0, n, ND, DZ, LDZ, 0.0);
      marginsAtX();
      QM_WAVE_SYNC();
      bool strongC = false; int passesC = 0;
      unsigned long long wsWordC = wsLoad(pass, level, 1);
      const int stc = levelQp(AZc, nOld, rds, n, false, rowNonZero(n), eqRow, strongC, passesC, wsWordC, nullptr, false);
      wsStore(pass, level, 1, wsWordC, passesC);
#ifdef QMGPU_EMU_DEBUG
      if (lane == 0 && inst == QMGPU_DEBUG_INST) printf("EMU canonical representative of level %d: n %d rows %d passes %d status %d\n", level, n, nOld, passesC, stc);
#endif
      if (stc != 0) status |= 8;
      double xc = 0.0;
      if (lane < ND) { xc = xs[lane]; for (int j = 0; j < n; ++j) xc += Z[lane * LDZ + j] * zs[j]; }
      QM_WAVE_SYNC();
      if (lane < ND) xs[lane] = xc;
      QM_WAVE_SYNC();
    }
  }
  if (canonical || n == 0) break;
  canonical = true;
  }
  QM_WAVE_SYNC();
  QM_WBC_CHECKPOINT(7);
  QM_TICK(9);
  // ---- updateCmd (WbcBase.cpp:580-595): tau = [M_j, -J_j^T] x + h_j
  if (lane < ND) a.out[size_t(inst) * 54 + lane] = xs[lane];
  if (lane < 18) {
    double s = nle[6 + lane];
    for (int j = 0; j < NVV; ++j) s += M[(6 + lane) * NVV + j] * xs[j];
    for (int j = 0; j < 12; ++j) s -= Jf[j * NVV + 6 + lane] * xs[24 + j];
    a.out[size_t(inst) * 54 + 36 + lane] = s;
  }
  if (lane == 0 && a.status) a.status[inst] = status;
  if (wsRec && lane == 0) { wsRec[13] = wsCount[0]; wsRec[14] = wsCount[1]; }
  forkCmd[0] = 0.0;      // release the helper wavefronts
  QM_LDS_BARRIER();
  QM_TICK(10);
  QM_TICK_FLUSH(192, blockIdx.x == 0 && lane == 0);
#ifdef QM_RICCATI_TIMING
  if (lane == 0 && inst < 256) qmk::qmRiccatiTicks[512 + inst * 4] = clock64() - qmStart;
#endif
}

}  // namespace qmk

