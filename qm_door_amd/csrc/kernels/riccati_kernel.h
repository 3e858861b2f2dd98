// riccati_kernel -- backward Riccati factorisation + forward substitution of the projected, equality-free OCP-QP.
// One wavefront per MPC instance, sequential over the horizon, stage blocks streamed HBM -> LDS.
//
// Replaces upstream HPIPM's OCP-QP solve as called by ocs2_sqp::SqpSolver (the object built at
// qm_controllers/src/QMController.cpp:288-289, settings task.info:76-93): with the state-input equalities projected out
// (projectStateInputEqualityConstraints true) and all inequalities handled as soft costs the QP has no inequality rows, so
// HPIPM's interior point reduces to one Riccati factorisation and solve.
//
// Lane roles per stage (n = 30 states, m~ = 30 - nc <= 18 projected inputs):
//   lane c < 30   column c of A~      -> S A~ column, G column, K column, new S column
//   lane 30       b~                  -> s + S b~,    g,         k,        new s
//   lane 31 + j   column j of B~      -> S B~ column, H column
// Every product is "matrix in LDS (broadcast reads) x my column in registers"; the only cross-lane structure is the
// m~ x m~ Cholesky in LDS.
#pragma once
#include "layout.h"
#include "gpu_rt.h"

namespace qmk {

struct RiccatiArgs {
  int batch, N;
  const double* stages;   // [batch][N+1][STAGE_DOUBLES]
  const int* stageNc;     // [batch][N+1]
  const double* x0;       // [batch][30]
  const double* X;        // [batch][N+1][30]
  double* gains;          // [batch][N][GAIN_DOUBLES]
  double* dX;             // [batch][N+1][30]
  double* dU;             // [batch][N][30]
  double* instStats;      // [batch][4]: armijo descent metric, status, -, -
};

constexpr int R_STG = 0;                         // staged record (first OFF_PX doubles used backward, all of it forward)
constexpr int R_S = R_STG + STAGE_DOUBLES;       // S [30][30]
constexpr int R_SV = R_S + 900;                  // s [30] (+2 pad)
constexpr int R_Y = R_SV + 32;                   // y columns, lane private [30][64]
constexpr int R_GH = R_Y + 30 * 64;              // [G | g | H] columns, lane private [MT][64]; G[j][i] = GH[j*64 + i]
constexpr int R_H = R_GH + MT * 64;              // H / L [MT][MT+1]
constexpr int R_T = R_H + MT * (MT + 1);         // new value function [30][32]
constexpr int R_GAIN = R_T + 960;                // staged gains (forward)
constexpr int R_VEC = R_GAIN + GAIN_DOUBLES;     // dx[30] dut[18] ...
constexpr int RICCATI_LDS_DOUBLES = R_VEC + 64;
constexpr int RICCATI_LDS_BYTES = RICCATI_LDS_DOUBLES * 8;  // ~86 KiB (dynamic LDS)

__global__ void __launch_bounds__(64) riccati_kernel(RiccatiArgs a) {
  QM_DYNAMIC_LDS(lds);
  const int lane = threadIdx.x;
  const int inst = blockIdx.x;
  const int N = a.N;
  double* stg = lds + R_STG; double* S = lds + R_S; double* sv = lds + R_SV; double* YL = lds + R_Y; double* GH = lds + R_GH; double* HL = lds + R_H; double* Tm = lds + R_T;
  double* gn = lds + R_GAIN; double* dxv = lds + R_VEC; double* dut = dxv + 32;
  const double* stagesI = a.stages + size_t(inst) * (N + 1) * STAGE_DOUBLES;
  const int* ncI = a.stageNc + size_t(inst) * (N + 1);
  int status = 0;

  // ---- terminal value function: S_N = Q_N, s_N = q_N
  {
    const double* rec = stagesI + size_t(N) * STAGE_DOUBLES;
    for (int e = lane; e < 900; e += 64) S[e] = rec[OFF_QT + e];
    if (lane < 30) sv[lane] = rec[OFF_qt + lane];
  }
  __syncthreads();

#pragma unroll 1
  for (int k = N - 1; k >= 0; --k) {
    const double* rec = stagesI + size_t(k) * STAGE_DOUBLES;
    const int nt = 30 - ncI[k];
    for (int e = lane; e < OFF_PX; e += 64) stg[e] = rec[e];
    __syncthreads();
    const bool isA = lane < 30, isb = lane == 30, isB = lane > 30 && lane < 31 + nt;
    // ---- y = S col (+ s for the b~ lane); outer loop rolled, my column of [A~ | b~ | B~] in registers
    {
      double col[30];
#pragma unroll
      for (int i = 0; i < 30; ++i) col[i] = isA ? stg[OFF_AT + i * 30 + lane] : (isb ? stg[OFF_bt + i] : (isB ? stg[OFF_BT + i * MT + (lane - 31)] : 0.0));
#pragma unroll 1
      for (int i = 0; i < 30; ++i) {
        double s = isb ? sv[i] : 0.0;
#pragma unroll
        for (int q = 0; q < 30; ++q) s += S[i * 30 + q] * col[q];
        YL[i * 64 + lane] = s;
      }
    }
    double y[30];
#pragma unroll
    for (int i = 0; i < 30; ++i) y[i] = YL[i * 64 + lane];
    // ---- gh = B~^T y + [P~ | r~ | R~] column
#pragma unroll 1
    for (int j = 0; j < nt; ++j) {
      double s = isA ? stg[OFF_PT + j * 30 + lane] : (isb ? stg[OFF_rt + j] : (isB ? stg[OFF_RT + j * MT + (lane - 31)] : 0.0));
#pragma unroll
      for (int i = 0; i < 30; ++i) s += stg[OFF_BT + i * MT + j] * y[i];
      GH[j * 64 + lane] = s;
      if (isB) HL[j * (MT + 1) + (lane - 31)] = s;
    }
    __syncthreads();
    // ---- Cholesky H = L L^T in LDS (lane r owns row r)
#pragma unroll 1
    for (int j = 0; j < nt; ++j) {
      const double d = HL[j * (MT + 1) + j];
      if (!(d > 0.0)) status = 1;
      const double dj = sqrt(d > 0.0 ? d : 1.0);
      __syncthreads();
      if (lane == j) HL[j * (MT + 1) + j] = dj;
      else if (lane > j && lane < nt) HL[lane * (MT + 1) + j] = HL[lane * (MT + 1) + j] / dj;
      __syncthreads();
      if (lane > j && lane < nt) {
        const double lij = HL[lane * (MT + 1) + j];
        for (int q = j + 1; q <= lane; ++q) HL[lane * (MT + 1) + q] -= lij * HL[q * (MT + 1) + j];
      }
      __syncthreads();
    }
    // ---- solve L L^T x = gh for the G columns and g (lanes <= 30): K = -x
    double kx[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      double s = (j < nt) ? GH[j * 64 + lane] : 0.0;
#pragma unroll
      for (int q = 0; q < MT; ++q) if (q < j) s -= HL[j * (MT + 1) + q] * kx[q];
      kx[j] = (j < nt) ? s / HL[j * (MT + 1) + j] : 0.0;
    }
#pragma unroll
    for (int j = MT - 1; j >= 0; --j) {
      double s = kx[j];
#pragma unroll
      for (int q = 0; q < MT; ++q) if (q > j && q < nt) s -= HL[q * (MT + 1) + j] * kx[q];
      kx[j] = (j < nt) ? s / HL[j * (MT + 1) + j] : 0.0;
    }
    double* gain = a.gains + (size_t(inst) * N + k) * GAIN_DOUBLES;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      kx[j] = -kx[j];
      if (isA) gain[OFF_KFB + j * 30 + lane] = kx[j];
      else if (isb) gain[OFF_kff + j] = kx[j];
    }
    // ---- new value function column: base + A~^T y + G^T kx   (lanes <= 30; rolled over the output row)
    if (lane <= 30) {
#pragma unroll 1
      for (int i = 0; i < 30; ++i) {
        double s = isA ? stg[OFF_QT + i * 30 + lane] : stg[OFF_qt + i];
#pragma unroll
        for (int q = 0; q < 30; ++q) s += stg[OFF_AT + q * 30 + i] * y[q];
#pragma unroll
        for (int j = 0; j < MT; ++j) if (j < nt) s += GH[j * 64 + i] * kx[j];
        Tm[i * 32 + lane] = s;
      }
    }
    __syncthreads();
    if (isA) {
#pragma unroll 1
      for (int i = 0; i < 30; ++i) S[i * 30 + lane] = 0.5 * (Tm[i * 32 + lane] + Tm[lane * 32 + i]);
      sv[lane] = Tm[lane * 32 + 30];
    }
    __syncthreads();
  }

  // ================================================================== forward substitution
  if (lane < 30) dxv[lane] = a.x0[size_t(inst) * 30 + lane] - a.X[size_t(inst) * (N + 1) * 30 + lane];
  double armijo = 0.0;
  __syncthreads();
#pragma unroll 1
  for (int k = 0; k < N; ++k) {
    const double* rec = stagesI + size_t(k) * STAGE_DOUBLES;
    const double* gain = a.gains + (size_t(inst) * N + k) * GAIN_DOUBLES;
    const int nt = 30 - ncI[k];
    for (int e = lane; e < STAGE_DOUBLES; e += 64) stg[e] = rec[e];
    for (int e = lane; e < GAIN_DOUBLES; e += 64) gn[e] = gain[e];
    __syncthreads();
    if (lane < 30) a.dX[(size_t(inst) * (N + 1) + k) * 30 + lane] = dxv[lane];
    // du~ = K dx + k
    if (lane < nt) {
      double s = gn[OFF_kff + lane];
      for (int c = 0; c < 30; ++c) s += gn[OFF_KFB + lane * 30 + c] * dxv[c];
      dut[lane] = s;
    }
    __syncthreads();
    double nx = 0.0;
    if (lane < 30) {
      // du = Pe + Px dx + Pu du~ ; dx+ = A~ dx + B~ du~ + b~
      double du = stg[OFF_PE + lane];
      nx = stg[OFF_bt + lane];
      for (int c = 0; c < 30; ++c) { du += stg[OFF_PX + lane * 30 + c] * dxv[c]; nx += stg[OFF_AT + lane * 30 + c] * dxv[c]; }
      for (int j = 0; j < nt; ++j) { du += stg[OFF_PU + lane * MT + j] * dut[j]; nx += stg[OFF_BT + lane * MT + j] * dut[j]; }
      a.dU[(size_t(inst) * N + k) * 30 + lane] = du;
      armijo += stg[OFF_qt + lane] * dxv[lane];
    }
    if (lane < nt) armijo += stg[OFF_rt + lane] * dut[lane];
    __syncthreads();
    if (lane < 30) dxv[lane] = nx;
    __syncthreads();
  }
  if (lane < 30) {
    a.dX[(size_t(inst) * (N + 1) + N) * 30 + lane] = dxv[lane];
    armijo += stagesI[size_t(N) * STAGE_DOUBLES + OFF_qt + lane] * dxv[lane];
  }
  // reduce armijo over lanes through LDS
  Tm[lane] = armijo;
  __syncthreads();
  if (lane == 0) {
    double s = 0.0;
    for (int i = 0; i < 64; ++i) s += Tm[i];
    a.instStats[size_t(inst) * 4 + 0] = s;
    a.instStats[size_t(inst) * 4 + 1] = double(status);
  }
}

}  // namespace qmk
