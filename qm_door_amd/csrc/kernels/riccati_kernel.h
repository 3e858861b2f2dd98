// riccati_kernel -- backward Riccati factorisation + forward substitution of the projected, equality-free OCP-QP.
// One WORKGROUP of NW wavefronts per MPC instance (one wavefront per SIMD of the CU that owns the instance), sequential over
// the horizon, stage blocks streamed HBM -> registers -> LDS one stage ahead of their use.
//
// Replaces upstream HPIPM's OCP-QP solve as called by ocs2_sqp::SqpSolver (the object built at
// qm_controllers/src/QMController.cpp:288-289, settings task.info:76-93): with the state-input equalities projected out
// (projectStateInputEqualityConstraints true) and all inequalities handled as soft costs the QP has no inequality rows, so
// HPIPM's interior point reduces to one Riccati factorisation and solve.
//
// Work split per stage (n = 30 states, m~ = 30 - nc <= 18 projected inputs):
//   lane c (same in every wavefront):  c < 30 column c of A~ | c == 30 the vector b~ | c = 31 + j column j of B~
//   wavefront w:                       the output ROWS [w*RPW, (w+1)*RPW) of every "matrix in LDS x my column in registers" product
// so a product costs each wavefront 1/NW of the FMAs and of the (broadcast) LDS reads.  The m~ x m~ Cholesky runs in wavefront 0
// with one row per lane in registers; the column solves are repeated by every wavefront (cheaper than exchanging K).
// Idle lanes alias lane 0's data through per-lane (offset, stride) pairs -- no exec-mask branching in the hot loops.
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

constexpr int RICCATI_WAVES = 4;
constexpr int R_STG = 0;                         // staged record (first OFF_PX doubles used backward, all of it forward)
constexpr int R_GAIN = R_STG + STAGE_DOUBLES;    // staged gains (forward)
constexpr int R_S = R_GAIN + GAIN_DOUBLES;       // S [30][30]
constexpr int R_SV = R_S + 900;                  // s [30] (+2 pad)
constexpr int R_Y = R_SV + 32;                   // y columns, lane private [30][64]
constexpr int R_GH = R_Y + 30 * 64;              // [G | g | H] columns, lane private [MT][64]; G[j][i] = GH[j*64 + i]
constexpr int R_H = R_GH + MT * 64;              // H / L [MT][MT+1]
constexpr int R_T = R_H + MT * (MT + 1);         // new value function [30][32]
constexpr int R_LC = R_T + 960;                  // Cholesky exchange: pivots [MT] (+2) + two column buffers [2][64]
constexpr int R_VEC = R_LC + MT + 2 + 128 + MT + 2; // dx[32] dut[32]   (R_LC also holds the reciprocal diagonal of L)
constexpr int R_ZERO = R_VEC + 64;               // 32 zeros
constexpr int RICCATI_LDS_DOUBLES = R_ZERO + 32;
constexpr int RICCATI_LDS_BYTES = RICCATI_LDS_DOUBLES * 8;  // ~88 KiB (dynamic LDS)

// Register-staged HBM -> LDS copy for a whole workgroup: issue() puts PF 16-byte loads per thread in flight, commit() drains
// them into LDS.  Between the two the workgroup computes on the *current* stage, so the memory latency of the next stage is
// hidden without a second LDS buffer.
template <int PF, int NTHR> struct StagePrefetch {
  double2 v[PF];
  __device__ __forceinline__ void issue(const double* src, int n, int tid) {
    const double2* s2 = reinterpret_cast<const double2*>(src);
    const int n2 = n >> 1;
#pragma unroll
    for (int k = 0; k < PF; ++k) { const int idx = tid + k * NTHR; v[k] = s2[idx < n2 ? idx : tid]; }
  }
  __device__ __forceinline__ void commit(double* dst, int n, int tid) const {
    double2* d2 = reinterpret_cast<double2*>(dst);
    const int n2 = n >> 1;
#pragma unroll
    for (int k = 0; k < PF; ++k) { const int idx = tid + k * NTHR; if (idx < n2) d2[idx] = v[k]; }
  }
};

template <int NW> __global__ void __launch_bounds__(NW * 64) riccati_kernel(RiccatiArgs a) {
  QM_DYNAMIC_LDS(lds);
  constexpr int NTHR = NW * 64;
  constexpr int RPW = (30 + NW - 1) / NW;   // output rows per wavefront
  constexpr int JPW = (MT + NW - 1) / NW;   // projected-input rows per wavefront
  constexpr int PFB = (OFF_PX / 2 + NTHR - 1) / NTHR;
  constexpr int PFR = (STAGE_DOUBLES / 2 + NTHR - 1) / NTHR;
  constexpr int PFG = (GAIN_DOUBLES / 2 + NTHR - 1) / NTHR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int inst = blockIdx.x;
  const int N = a.N;
  double* stg = lds + R_STG; double* gn = lds + R_GAIN; double* S = lds + R_S; double* sv = lds + R_SV; double* YL = lds + R_Y; double* GH = lds + R_GH;
  double* HL = lds + R_H; double* Tm = lds + R_T; double* LCp = lds + R_LC; double* LCc = LCp + MT + 2; double* dxv = lds + R_VEC; double* dut = dxv + 32; const double* invD = LCp + MT + 2 + 128;
  double* zero32 = lds + R_ZERO;
  const double* stagesI = a.stages + size_t(inst) * (N + 1) * STAGE_DOUBLES;
  const double* gainsI = a.gains + size_t(inst) * N * GAIN_DOUBLES;
  const int* ncI = a.stageNc + size_t(inst) * (N + 1);
  int status = 0;
  if (tid < 32) zero32[tid] = 0.0;

  // ---- terminal value function S_N = Q_N, s_N = q_N, and the first stage to process
  {
    const double* rec = stagesI + size_t(N) * STAGE_DOUBLES;
    for (int e = tid; e < 900; e += NTHR) S[e] = rec[OFF_QT + e];
    if (tid < 30) sv[tid] = rec[OFF_qt + tid];
    StagePrefetch<PFB, NTHR> pf;
    pf.issue(stagesI + size_t(N - 1) * STAGE_DOUBLES, OFF_PX, tid);
    pf.commit(stg, OFF_PX, tid);
  }
  __syncthreads();

#pragma unroll 1
  for (int k = N - 1; k >= 0; --k) {
    const int nt = 30 - ncI[k];
    StagePrefetch<PFB, NTHR> pf;
    pf.issue(stagesI + size_t(k > 0 ? k - 1 : 0) * STAGE_DOUBLES, OFF_PX, tid);  // next stage's blocks, in flight during this stage
    const bool isA = lane < 30, isb = lane == 30, isB = lane > 30 && lane < 31 + nt;
    const int colOff = isA ? OFF_AT + lane : (isb ? OFF_bt : (isB ? OFF_BT + (lane - 31) : OFF_AT));
    const int colStr = isA ? 30 : (isb ? 1 : (isB ? MT : 30));
    const int ghOff = isA ? OFF_PT + lane : (isb ? OFF_rt : (isB ? OFF_RT + (lane - 31) : OFF_PT));
    const int ghStr = isA ? 30 : (isb ? 1 : (isB ? MT : 30));
    const double* accInit = isb ? sv : zero32;  // s enters only the b~ lane's product
    // ---- y = S col (+ s): my column of [A~ | b~ | B~] in registers, this wavefront's RPW output rows as independent FMA chains
    {
      double col[30];
#pragma unroll
      for (int i = 0; i < 30; ++i) col[i] = stg[colOff + i * colStr];
#pragma unroll 1
      for (int rr = 0; rr < RPW; rr += 4) {  // four independent FMA chains per trip
        const int i0 = wave * RPW + rr;
        int row[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) row[r] = (i0 + r < 30 && rr + r < RPW) ? i0 + r : 29;
        double acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = accInit[row[r]];
#pragma unroll
        for (int q = 0; q < 30; ++q) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] += S[row[r] * 30 + q] * col[q];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) if (i0 + r < 30 && rr + r < RPW) YL[(i0 + r) * 64 + lane] = acc[r];
      }
    }
    __syncthreads();
    // ---- gh = B~^T y + [P~ | r~ | R~] column; this wavefront's JPW rows (rows >= m~ are padding: computed, never used)
    {
      double y[30];
#pragma unroll
      for (int i = 0; i < 30; ++i) y[i] = YL[i * 64 + lane];
      const int j0 = wave * JPW;
      double acc[JPW];
#pragma unroll
      for (int r = 0; r < JPW; ++r) acc[r] = stg[ghOff + (j0 + r < MT ? j0 + r : MT - 1) * ghStr];
#pragma unroll
      for (int i = 0; i < 30; ++i) {
#pragma unroll
        for (int r = 0; r < JPW; ++r) acc[r] += stg[OFF_BT + i * MT + (j0 + r < MT ? j0 + r : MT - 1)] * y[i];
      }
#pragma unroll
      for (int r = 0; r < JPW; ++r) {
        const int j = j0 + r;
        if (j < MT) { GH[j * 64 + lane] = acc[r]; if (isB) HL[j * (MT + 1) + (lane - 31)] = acc[r]; }
      }
    }
    __syncthreads();
    // ---- Cholesky H = L L^T: wavefront 0, lane r holds row r in registers; pivots and columns pass through LDS
    if (wave == 0) {
      double hrow[MT];
#pragma unroll
      for (int q = 0; q < MT; ++q) hrow[q] = HL[(lane < MT ? lane : 0) * (MT + 1) + q];
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        if (j < nt) {
          if (lane == j) LCp[j] = hrow[j];
          QM_WAVE_SYNC();
          const double d = LCp[j];
          if (!(d > 0.0)) status = 1;
          const double dj = sqrt(d > 0.0 ? d : 1.0), idj = 1.0 / dj;
          if (lane == j) LCp[MT + 2 + 128 + j] = idj;  // reciprocal diagonal, read by the column solves
          const double l = (lane == j) ? dj : hrow[j] * idj;
          hrow[j] = l;
          LCc[(j & 1) * 64 + lane] = l;
          QM_WAVE_SYNC();
#pragma unroll
          for (int q = j + 1; q < MT; ++q) hrow[q] -= l * LCc[(j & 1) * 64 + q];
        }
      }
      if (lane < MT) {
#pragma unroll
        for (int q = 0; q < MT; ++q) HL[lane * (MT + 1) + q] = hrow[q];
      }
    }
    __syncthreads();
    // ---- solve L L^T x = gh for the G columns and g (every wavefront, for all of its lanes): K = -x
    double kx[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      double s = (j < nt) ? GH[j * 64 + lane] : 0.0;
#pragma unroll
      for (int q = 0; q < MT; ++q) if (q < j) s -= HL[j * (MT + 1) + q] * kx[q];
      kx[j] = (j < nt) ? s * invD[j] : 0.0;
    }
#pragma unroll
    for (int j = MT - 1; j >= 0; --j) {
      double s = kx[j];
#pragma unroll
      for (int q = 0; q < MT; ++q) if (q > j && q < nt) s -= HL[q * (MT + 1) + j] * kx[q];
      kx[j] = (j < nt) ? s * invD[j] : 0.0;
    }
    double* gain = a.gains + (size_t(inst) * N + k) * GAIN_DOUBLES;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      kx[j] = -kx[j];
      if ((j % NW) == wave) {  // the K rows are written once, spread over the wavefronts
        if (isA) gain[OFF_KFB + j * 30 + lane] = kx[j];
        else if (isb) gain[OFF_kff + j] = kx[j];
      }
    }
    // ---- new value function column: base + A~^T y + G^T kx   (lanes <= 30 matter; this wavefront's RPW rows)
    {
      const int qOff = isA ? OFF_QT + lane : (isb ? OFF_qt : OFF_QT), qStr = isb ? 1 : 30;
      const int tLane = lane <= 30 ? lane : 31;  // idle lanes dump into the spare column 31 of Tm
      double y[30];
#pragma unroll
      for (int i = 0; i < 30; ++i) y[i] = YL[i * 64 + lane];
#pragma unroll 1
      for (int rr = 0; rr < RPW; rr += 4) {
        const int i0 = wave * RPW + rr;
        int row[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) row[r] = (i0 + r < 30 && rr + r < RPW) ? i0 + r : 29;
        double acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = stg[qOff + row[r] * qStr];
#pragma unroll
        for (int q = 0; q < 30; ++q) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] += stg[OFF_AT + q * 30 + row[r]] * y[q];
        }
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          if (j < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += GH[j * 64 + row[r]] * kx[j];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) if (i0 + r < 30 && rr + r < RPW) Tm[(i0 + r) * 32 + tLane] = acc[r];
      }
    }
    __syncthreads();
    // symmetrise into S (rows of this wavefront), new s, and drain the prefetched next stage into the (now free) staging buffer
    if (isA) {
#pragma unroll
      for (int r = 0; r < RPW; ++r) { const int i = wave * RPW + r; if (i < 30) S[i * 30 + lane] = 0.5 * (Tm[i * 32 + lane] + Tm[lane * 32 + i]); }
      if (wave == 0) sv[lane] = Tm[lane * 32 + 30];
    }
    pf.commit(stg, OFF_PX, tid);
    __syncthreads();
  }

  // ================================================================== forward substitution
  // wavefront 0: du~ = K dx + k, du = Pe + Px dx + Pu du~ ; wavefront 1: dx+ = A~ dx + B~ du~ + b~ ; everybody prefetches
  constexpr int WX = 1 % NW;
  {
    StagePrefetch<PFR, NTHR> pr;
    StagePrefetch<PFG, NTHR> pg;
    pr.issue(stagesI, STAGE_DOUBLES, tid);
    pg.issue(gainsI, GAIN_DOUBLES, tid);
    pr.commit(stg, STAGE_DOUBLES, tid);
    pg.commit(gn, GAIN_DOUBLES, tid);
  }
  if (tid < 30) dxv[tid] = a.x0[size_t(inst) * 30 + tid] - a.X[size_t(inst) * (N + 1) * 30 + tid];
  double armijo = 0.0;
  __syncthreads();
#pragma unroll 1
  for (int k = 0; k < N; ++k) {
    const int nt = 30 - ncI[k];
    const int kn = k + 1 < N ? k + 1 : k;
    StagePrefetch<PFR, NTHR> pr;
    StagePrefetch<PFG, NTHR> pg;
    pr.issue(stagesI + size_t(kn) * STAGE_DOUBLES, STAGE_DOUBLES, tid);
    pg.issue(gainsI + size_t(kn) * GAIN_DOUBLES, GAIN_DOUBLES, tid);
    if (wave == 0 && lane < 30) a.dX[(size_t(inst) * (N + 1) + k) * 30 + lane] = dxv[lane];
    if (wave == 0 && lane < nt) {
      double s0 = gn[OFF_kff + lane], s1 = 0.0;
#pragma unroll
      for (int c = 0; c < 30; c += 2) { s0 += gn[OFF_KFB + lane * 30 + c] * dxv[c]; s1 += gn[OFF_KFB + lane * 30 + c + 1] * dxv[c + 1]; }
      dut[lane] = s0 + s1;
    }
    __syncthreads();
    double nx = 0.0;
    if (wave == 0 && lane < 30) {  // du = Pe + Px dx + Pu du~
      double s0 = stg[OFF_PE + lane], s1 = 0.0;
#pragma unroll
      for (int c = 0; c < 30; c += 2) { s0 += stg[OFF_PX + lane * 30 + c] * dxv[c]; s1 += stg[OFF_PX + lane * 30 + c + 1] * dxv[c + 1]; }
      for (int j = 0; j < nt; ++j) s0 += stg[OFF_PU + lane * MT + j] * dut[j];
      a.dU[(size_t(inst) * N + k) * 30 + lane] = s0 + s1;
    }
    if (wave == WX && lane < 30) {  // dx+ = A~ dx + B~ du~ + b~ ; armijo contribution q~ . dx
      double s0 = stg[OFF_bt + lane], s1 = 0.0;
#pragma unroll
      for (int c = 0; c < 30; c += 2) { s0 += stg[OFF_AT + lane * 30 + c] * dxv[c]; s1 += stg[OFF_AT + lane * 30 + c + 1] * dxv[c + 1]; }
      for (int j = 0; j < nt; ++j) s0 += stg[OFF_BT + lane * MT + j] * dut[j];
      nx = s0 + s1;
      armijo += stg[OFF_qt + lane] * dxv[lane];
    }
    if (wave == WX && lane >= 32 && lane < 32 + nt) armijo += stg[OFF_rt + (lane - 32)] * dut[lane - 32];
    __syncthreads();
    if (wave == WX && lane < 30) dxv[lane] = nx;
    pr.commit(stg, STAGE_DOUBLES, tid);
    pg.commit(gn, GAIN_DOUBLES, tid);
    __syncthreads();
  }
  if (wave == WX && lane < 30) {
    a.dX[(size_t(inst) * (N + 1) + N) * 30 + lane] = dxv[lane];
    armijo += stagesI[size_t(N) * STAGE_DOUBLES + OFF_qt + lane] * dxv[lane];
  }
  // reduce armijo over the lanes of wavefront WX through LDS
  if (wave == WX) Tm[lane] = armijo;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int i = 0; i < 64; ++i) s += Tm[i];
    a.instStats[size_t(inst) * 4 + 0] = s;
    a.instStats[size_t(inst) * 4 + 1] = double(status);
  }
}

}  // namespace qmk
