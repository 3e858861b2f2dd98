// riccati_kernel -- backward Riccati factorisation + forward substitution of the projected, equality-free OCP-QP.
// One WORKGROUP of 4 wavefronts per MPC instance (one wavefront per SIMD of the CU that owns the instance), sequential over the
// horizon, stage blocks streamed HBM -> registers -> LDS one stage ahead of their use.
//
// Replaces upstream HPIPM's OCP-QP solve as called by ocs2_sqp::SqpSolver (the object built at
// qm_controllers/src/QMController.cpp:288-289, settings task.info:76-93): with the state-input equalities projected out
// (projectStateInputEqualityConstraints true) and all inequalities handled as soft costs the QP has no inequality rows, so
// HPIPM's interior point reduces to one Riccati factorisation and solve.
//
// Backward stage (n = 30 states, m~ = 30 - nc <= 18 projected inputs), every product on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64, 16x16 output tiles, operands read from LDS in the lane layout of gpu_rt.h: qmMfma):
//   M  = [A~ | b~ | 0 | B~ | 0]           30 x 64 view of the staged record (columns 0..29, 30, 32..32+m~-1)
//   P1 Y  = S M  (+ s in column 30)        so Y = [S A~ | S b~ + s | . | S B~]
//   P2 T  = B~^T Y + [P~ | r~ | . | R~]    so T = [G | g | . | H]
//   P3 H  = L L^T and L^-1                 wavefront 0, one column of [H | I] per lane in registers, row operations with the
//                                          multipliers broadcast by v_readlane (no LDS round trip on the dependent chain)
//   P4 W  = L^-1 [G | g]
//   P5 [K | k] = -L^-T W                   -> gains (HBM)
//   P6 [S' | s'] = [Q~ | q~] + A~^T [S A~ | y] - W^T W
// Column 30 carries the affine terms (b~, y, g, k, s') through the same tiles as the matrices.  Zero padding: rows/columns
// 30,31 of S, rows >= m~ of T / W / L^-1 are kept at exactly zero so that partial tiles need no predication on the k loops.
#pragma once
#include "layout.h"
#include "gpu_rt.h"

namespace qmk {

struct RiccatiArgs {
  int batch, N;
  const real* stages;   // [batch][N+1][STAGE_DOUBLES]
  const int* stageNc;     // [batch][N+1]
  const real* x0;       // [batch][30]
  const real* X;        // [batch][N+1][30]
  real* gains;          // [batch][N][GAIN_DOUBLES]
  real* dX;             // [batch][N+1][30]
  real* dU;             // [batch][N][30]
  real* instStats;      // [batch][4]: armijo descent metric, status, -, -
  const int* done;        // [batch] converged instances are skipped
};

constexpr int RICCATI_WAVES = 4;
// LDS strides (doubles) = 16 mod 32: the four k-rows x sixteen consecutive columns one MFMA operand read touches hit distinct banks
constexpr int LDS_S = 48, LDS_Y = 80, LDS_W = 48, LDS_TS = 34, LDS_LT = 49;   // LDS_LT odd: lane c writes row c of L^-T without bank conflicts
constexpr int STG_B = OFF_PX + 4;                 // doubles of a record the backward sweep needs (padded)
constexpr int STG_F = STAGE_DOUBLES + GAIN_DOUBLES;  // record + gains of one stage for the forward sweep
constexpr int R_STG = 0;                          // two staging buffers: [2][STG_B] backward, [2][STG_F] forward (over Y / T, dead by then)
constexpr int R_Y = R_STG + 2 * STG_B;            // Y [32][LDS_Y]
constexpr int R_T = R_Y + 32 * LDS_Y;             // T [32][LDS_Y]
constexpr int R_S = R_T + 32 * LDS_Y;             // S [32][LDS_S]
constexpr int R_SV = R_S + 32 * LDS_S;            // s [32]
constexpr int R_W = R_SV + 32;                    // W [20][LDS_W]
constexpr int R_LI = R_W + 20 * LDS_W;            // L^-1 row major [20][LDS_W]
constexpr int R_LIT = R_LI + 20 * LDS_W;          // L^-T row major [20][LDS_LT]
constexpr int R_VEC = R_LIT + 20 * LDS_LT + 4;    // dx[32] dut[32]
constexpr int R_SCR = R_VEC + 64;                 // exchange scratch of the host emulation [4][256]; armijo reduction
constexpr int RICCATI_LDS_DOUBLES = R_SCR + 4 * 256;
constexpr int RICCATI_LDS_BYTES = RICCATI_LDS_DOUBLES * int(sizeof(real));  // ~105 KiB (dynamic LDS)
static_assert(2 * STG_F <= RICCATI_LDS_DOUBLES, "forward-sweep staging fits");

// Register-staged HBM -> LDS copy for a whole workgroup: issue() puts PF 16-byte loads per thread in flight, commit() drains
// them into LDS.  Between the two the workgroup computes on the *current* stage, so the memory latency of the next stage is
// hidden without a second LDS buffer.
template <int PF, int NTHR> struct StagePrefetch {
  // named members, not an array: an array indexed inside (even fully unrolled) loops was left in scratch memory by the compiler
  // in this kernel, which turned the prefetch into an HBM round trip (profiles/r01b_notes.md)
  static_assert(PF <= 13, "add members");
  QmD2 v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12;
#define QM_PF_FOR_EACH(X) X(0, v0) X(1, v1) X(2, v2) X(3, v3) X(4, v4) X(5, v5) X(6, v6) X(7, v7) X(8, v8) X(9, v9) X(10, v10) X(11, v11) X(12, v12)
  __device__ __forceinline__ void issue(const real* src, int n, int tid) {
    const QmD2* s2 = reinterpret_cast<const QmD2*>(src);
    const int n2 = n >> 1;
#define QM_PF_ISSUE(K, V) if constexpr (PF > K) { const int idx = tid + K * NTHR; V = s2[idx < n2 ? idx : tid]; }
    QM_PF_FOR_EACH(QM_PF_ISSUE)
#undef QM_PF_ISSUE
  }
  __device__ __forceinline__ void commit(real* dst, int n, int tid) const {
    QmD2* d2 = reinterpret_cast<QmD2*>(dst);
    const int n2 = n >> 1;
#define QM_PF_COMMIT(K, V) if constexpr (PF > K) { const int idx = tid + K * NTHR; if (idx < n2) d2[idx] = V; }
    QM_PF_FOR_EACH(QM_PF_COMMIT)
#undef QM_PF_COMMIT
  }
#undef QM_PF_FOR_EACH
};

template <int NW> __global__ void __launch_bounds__(NW * 64) QM_ONE_WAVE_PER_SIMD riccati_kernel(RiccatiArgs a) {
  static_assert(NW == 4, "tile ownership below is written for four wavefronts");
  QM_DYNAMIC_LDS(lds);
  QM_POISON_LDS(lds, RICCATI_LDS_DOUBLES);
  constexpr int NTHR = NW * 64;
  constexpr int PFB = (OFF_PX / 2 + NTHR - 1) / NTHR;
  // forward sweep: only A~ B~ (the head of the record) and b~ q~ r~ Px Pu Pe (its tail) are read; Q~ P~ R~ in between are not
  constexpr int FWD_HEAD = OFF_QT, FWD_TAIL0 = OFF_bt, FWD_TAIL = STAGE_DOUBLES - OFF_bt;
  static_assert(FWD_HEAD % 2 == 0 && FWD_TAIL0 % 2 == 0 && FWD_TAIL % 2 == 0, "16-byte units");
  constexpr int PFH = (FWD_HEAD / 2 + NTHR - 1) / NTHR, PFT = (FWD_TAIL / 2 + NTHR - 1) / NTHR;
  constexpr int PFG = (GAIN_DOUBLES / 2 + NTHR - 1) / NTHR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, h = lane >> 4, la = qmARow(l16);   // MFMA operand coordinates of this lane; la: row of an A operand (gpu_rt.h)
  const int inst = blockIdx.x;
  if (a.done[inst]) return;   // workgroup uniform
  const int N = a.N;
  real* S = lds + R_S; real* sv = lds + R_SV; real* Y = lds + R_Y; real* T = lds + R_T;
  real* TS = lds + R_T;   // [32][LDS_TS] raw S' of a stage (aliases T, dead after P4)
  real* W = lds + R_W; real* LI = lds + R_LI; real* LIT = lds + R_LIT; real* dxv = lds + R_VEC; real* dut = dxv + 32;
  real* scr = lds + R_SCR + wave * 256; real* red = lds + R_SCR;
  const real* stagesI = a.stages + size_t(inst) * (N + 1) * STAGE_DOUBLES;
  const real* gainsI = a.gains + size_t(inst) * N * GAIN_DOUBLES;
  const int* ncI = a.stageNc + size_t(inst) * (N + 1);
  int status = 0;

  // ---- terminal value function S_N = Q_N, s_N = q_N (zero padded), and the first stage to process
  {
    const real* rec = stagesI + size_t(N) * STAGE_DOUBLES;
    for (int e = tid; e < 32 * LDS_S; e += NTHR) { const int i = e / LDS_S, j = e % LDS_S; S[e] = (i < 30 && j < 30) ? rec[OFF_QT + i * 30 + j] : 0.0_r; }
    if (tid < 32) sv[tid] = tid < 30 ? rec[OFF_qt + tid] : 0.0_r;
    for (int e = tid; e < 2 * 20 * LDS_W + 20 * LDS_LT; e += NTHR) W[e] = 0.0_r;        // W, L^-1, L^-T (contiguous)
    for (int e = tid; e < 2 * 32 * LDS_Y; e += NTHR) Y[e] = 0.0_r;        // Y, T (contiguous)
    StagePrefetch<PFB, NTHR> pf;
    pf.issue(stagesI + size_t(N - 1) * STAGE_DOUBLES, OFF_PX, tid);
    pf.commit(lds + R_STG + ((N - 1) & 1) * STG_B, OFF_PX, tid);
  }
  __syncthreads();

#pragma unroll 1
  for (int k = N - 1; k >= 0; --k) {
    const real* stg = lds + R_STG + (k & 1) * STG_B;        // this stage (committed during the previous one)
    real* stgNext = lds + R_STG + ((k + 1) & 1) * STG_B;    // buffer of stage k - 1
    const int nt = 30 - ncI[k];
    const int mtTiles = nt > 16 ? 2 : 1;     // 16-row tiles covering the m~ projected inputs
    const int nTiles = nt > 16 ? 4 : 3;      // 16-column tiles covering [A~ | b~ | . | B~]
    StagePrefetch<PFB, NTHR> pf;
    pf.issue(stagesI + size_t(k > 0 ? k - 1 : 0) * STAGE_DOUBLES, OFF_PX, tid);  // next stage's blocks, in flight during this stage
    // ---- P1 + P2: wavefront w owns the 16 columns [16 w, 16 w + 16) of Y and of T
    const int jc = wave * 16 + l16;          // my column of M / Y / T
    const bool jA = jc < 30, jb = jc == 30, jB = jc >= 32 && jc < 32 + nt;
    if (wave < nTiles) {
      const int mOff = jA ? OFF_AT + jc : (jb ? OFF_bt : (jB ? OFF_BT + (jc - 32) : 0));
      const int mStr = jA ? 30 : (jb ? 1 : (jB ? MT : 0));
      const bool mValid = jA || jb || jB;
      QmAcc c0, c1;
#pragma unroll
      for (int r = 0; r < 4; ++r) { const real s0 = sv[h + 4 * r], s1 = sv[16 + h + 4 * r]; c0[r] = jb ? s0 : 0.0_r; c1[r] = jb ? s1 : 0.0_r; }
      real a0[8], a1[8], bv[8];   // all operands first (unconditional loads, selects afterwards): the LDS latency is paid once
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int kk = 4 * ks + h, kc = kk < 30 ? kk : 29;  // rows 30,31 of S^T are zero: the clamped b operand is multiplied by 0
        a0[ks] = S[kk * LDS_S + la]; a1[ks] = S[kk * LDS_S + 16 + la];  // S is symmetric: S[i][k] read as S[k][i]
        const real raw = stg[mOff + kc * mStr];
        bv[ks] = mValid ? raw : 0.0_r;
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) { qmMfma(c0, a0[ks], bv[ks], scr); qmMfma(c1, a1[ks], bv[ks], scr); }
#pragma unroll
      for (int r = 0; r < 4; ++r) { Y[(h + 4 * r) * LDS_Y + jc] = c0[r]; Y[(16 + h + 4 * r) * LDS_Y + jc] = c1[r]; }
    }
    QM_LDS_BARRIER();
    if (wave < nTiles) {
      QmAcc c0, c1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i0 = h + 4 * r, i1 = 16 + h + 4 * r;
        const int i0c = i0 < MT ? i0 : 0, i1c = i1 < MT ? i1 : 0;
        const int jcA = jA ? jc : 0, jcB = jB ? jc - 32 : 0;
        const real p0 = stg[OFF_PT + i0c * 30 + jcA], q0 = stg[OFF_rt + i0c], w0 = stg[OFF_RT + i0c * MT + jcB];
        const real p1 = stg[OFF_PT + i1c * 30 + jcA], q1 = stg[OFF_rt + i1c], w1 = stg[OFF_RT + i1c * MT + jcB];
        const real v0 = jA ? p0 : (jb ? q0 : (jB ? w0 : 0.0_r));
        const real v1 = jA ? p1 : (jb ? q1 : (jB ? w1 : 0.0_r));
        c0[r] = i0 < nt ? v0 : 0.0_r;
        c1[r] = i1 < nt ? v1 : 0.0_r;
      }
      const bool a0ok = la < nt, a1ok = 16 + la < nt;
      const int a1c = 16 + la < MT ? 16 + la : 0;
      real a0[8], a1[8], bv[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int kk = 4 * ks + h, kc = kk < 30 ? kk : 29;  // rows 30,31 of Y are zero
        bv[ks] = Y[kk * LDS_Y + jc];
        const real r0 = stg[OFF_BT + kc * MT + (la < MT ? la : 0)], r1 = stg[OFF_BT + kc * MT + a1c];   // B~^T[i][k] = B~[k][i]
        a0[ks] = a0ok ? r0 : 0.0_r; a1[ks] = a1ok ? r1 : 0.0_r;
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) { qmMfma(c0, a0[ks], bv[ks], scr); if (mtTiles == 2) qmMfma(c1, a1[ks], bv[ks], scr); }
#pragma unroll
      for (int r = 0; r < 4; ++r) { T[(h + 4 * r) * LDS_Y + jc] = c0[r]; if (mtTiles == 2) T[(16 + h + 4 * r) * LDS_Y + jc] = c1[r]; }
    }
    QM_LDS_BARRIER();
    // ---- P3: H = L L^T and L^-1 by row operations on [H | I]; lane c < 32 holds column c of H, lane 32 + c column c of I
    if (wave == 0) {
      const bool isH = lane < 32;
      const int c = isH ? lane : lane - 32;
      real col[MT];
#pragma unroll
      for (int r = 0; r < MT; ++r) {
        const real e = (r == c) ? 1.0_r : 0.0_r;
        const real hv = T[r * LDS_Y + 32 + (lane < MT ? lane : 0)];
        col[r] = (isH && lane < nt && r < nt) ? hv : e;
      }
      // steps j >= m~ meet identity columns (pivot 1, multipliers 0): no branch, one basic block.  The reciprocal square root of
      // pivot j + 1 is started right after row j + 1 has received its update, so its latency hides behind the remaining updates.
      real inv;
      {
        const real piv = qmReadLane(col[0], 0, scr);
        if (!(piv > 0.0_r)) status = 1;
        inv = qmRsqrt(piv > 0.0_r ? piv : 1.0_r);
      }
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        col[j] *= inv;                                     // row j of [L^T | .] / sqrt(pivot)
        const QmGather gj = qmGather(col[j], scr);
        if (j + 1 < MT) {
          col[j + 1] -= gj.get(j + 1) * col[j];
          const real piv = qmReadLane(col[j + 1], j + 1, scr);
          if (!(piv > 0.0_r)) status = 1;
          inv = qmRsqrt(piv > 0.0_r ? piv : 1.0_r);
        }
#pragma unroll
        for (int r = j + 2; r < MT; ++r) col[r] -= gj.get(r) * col[j];   // L[r][j] = gj.get(r) (zero for r >= m~: identity columns)
      }
      if (!isH && c < 20) {
#pragma unroll
        for (int r = 0; r < MT; ++r) { const real v = (c < nt && r < nt) ? col[r] : 0.0_r; LI[r * LDS_W + c] = v; LIT[c * LDS_LT + r] = v; }
      }
    }
    pf.commit(stgNext, OFF_PX, tid);   // stage k - 1 lands in the other buffer: wavefronts 1..3 do it while wavefront 0 factorises
    QM_LDS_BARRIER();
    // ---- P4: W = L^-1 [G | g]: wavefront w owns tile (w >> 1, w & 1)
    const int tm = wave >> 1, tn = wave & 1;
    {
      QmAcc c;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = 0.0_r;
      real av[5], bw[5];
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) { const int kk = 4 * ks + h; av[ks] = LIT[kk * LDS_LT + tm * 16 + la]; bw[ks] = T[kk * LDS_Y + tn * 16 + l16]; }   // L^-1[i][k] = L^-T[k][i]
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) qmMfma(c, av[ks], bw[ks], scr);
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int i = tm * 16 + h + 4 * r; if (i < 20) W[i * LDS_W + tn * 16 + l16] = c[r]; }
    }
    QM_LDS_BARRIER();
    // ---- P5: [K | k] = -L^-T W -> gains
    {
      QmAcc c;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = 0.0_r;
      real av[5], bw[5];
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) { const int kk = 4 * ks + h; av[ks] = -LI[kk * LDS_W + tm * 16 + la]; bw[ks] = W[kk * LDS_W + tn * 16 + l16]; }   // L^-T[i][k] = L^-1[k][i]
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) qmMfma(c, av[ks], bw[ks], scr);
      real* gain = a.gains + (size_t(inst) * N + k) * GAIN_DOUBLES;
      const int j = tn * 16 + l16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = tm * 16 + h + 4 * r;
        if (i < MT) {
          const real v = i < nt ? c[r] : 0.0_r;
          if (j < 30) gain[OFF_KFB + i * 30 + j] = v;
          else if (j == 30) gain[OFF_kff + i] = v;
        }
      }
    }
    // ---- P6: [S' | s'] = [Q~ | q~] + A~^T [S A~ | y] - W^T W: wavefront w owns tile (w >> 1, w & 1) of the 32 x 32 result
    {
      const int j = tn * 16 + l16;
      QmAcc c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = tm * 16 + h + 4 * r, ic = i < 30 ? i : 0;
        const real v = j < 30 ? stg[OFF_QT + ic * 30 + j] : (j == 30 ? stg[OFF_qt + ic] : 0.0_r);
        c[r] = i < 30 ? v : 0.0_r;
      }
      const int ai = tm * 16 + la < 30 ? tm * 16 + la : 29;   // rows 30,31 of the result are discarded
      real av[13], bw[13];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int kk = 4 * ks + h, kc = kk < 30 ? kk : 29;      // rows 30,31 of Y are zero
        av[ks] = stg[OFF_AT + kc * 30 + ai]; bw[ks] = Y[kk * LDS_Y + j];   // A~^T[i][k] = A~[k][i]
      }
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) { const int kk = 4 * ks + h; av[8 + ks] = -W[kk * LDS_W + tm * 16 + la]; bw[8 + ks] = W[kk * LDS_W + j]; }
#pragma unroll
      for (int ks = 0; ks < 13; ++ks) qmMfma(c, av[ks], bw[ks], scr);
      // The raw result goes to a scratch square (T is dead; stride 34 makes both the row and the column walk conflict free), s' in
      // place; after the barrier S = (C + C^T) / 2.  Without the symmetrisation the antisymmetric part of the rounding error is
      // propagated by the OPEN-loop dynamics (it sees A~^T . A~ but not the cancelling G^T H^-1 G) and grows ~1.13x per stage.
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = tm * 16 + h + 4 * r;
        TS[i * LDS_TS + j] = c[r];
        if (i < 30 && j == 30) sv[i] = c[r];
      }
      QM_LDS_BARRIER();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = tm * 16 + h + 4 * r;
        const real up = TS[i * LDS_TS + j], lo = TS[j * LDS_TS + i];
        if (i < 30 && j < 30) S[i * LDS_S + j] = 0.5_r * (up + lo);
      }
    }
    QM_LDS_BARRIER();
  }

  // ================================================================== forward substitution
  // wavefront 0: du~ = K dx + k, du = Pe + Px dx + Pu du~ ; wavefront 1: dx+ = A~ dx + B~ du~ + b~ ; everybody prefetches
  constexpr int WX = 1 % NW;
  __syncthreads();   // full barrier: the gains written to HBM by every wavefront are read back by all of them below
  {
    StagePrefetch<PFH, NTHR> ph;
    StagePrefetch<PFT, NTHR> pt;
    StagePrefetch<PFG, NTHR> pg;
    ph.issue(stagesI, FWD_HEAD, tid);
    pt.issue(stagesI + FWD_TAIL0, FWD_TAIL, tid);
    pg.issue(gainsI, GAIN_DOUBLES, tid);
    ph.commit(lds + R_STG, FWD_HEAD, tid);
    pt.commit(lds + R_STG + FWD_TAIL0, FWD_TAIL, tid);
    pg.commit(lds + R_STG + STAGE_DOUBLES, GAIN_DOUBLES, tid);
  }
  if (tid < 30) dxv[tid] = a.x0[size_t(inst) * 30 + tid] - a.X[size_t(inst) * (N + 1) * 30 + tid];
  real armijo = 0.0_r;
  __syncthreads();
#pragma unroll 1
  for (int k = 0; k < N; ++k) {
    const int nt = 30 - ncI[k];
    const int kn = k + 1 < N ? k + 1 : k;
    const real* stg = lds + R_STG + (k & 1) * STG_F; const real* gn = stg + STAGE_DOUBLES;
    real* stgNext = lds + R_STG + ((k + 1) & 1) * STG_F;
    StagePrefetch<PFH, NTHR> ph;
    StagePrefetch<PFT, NTHR> pt;
    StagePrefetch<PFG, NTHR> pg;
    ph.issue(stagesI + size_t(kn) * STAGE_DOUBLES, FWD_HEAD, tid);
    pt.issue(stagesI + size_t(kn) * STAGE_DOUBLES + FWD_TAIL0, FWD_TAIL, tid);
    pg.issue(gainsI + size_t(kn) * GAIN_DOUBLES, GAIN_DOUBLES, tid);
    if (wave == 0 && lane < 30) a.dX[(size_t(inst) * (N + 1) + k) * 30 + lane] = dxv[lane];
    if (wave == 0 && lane < nt) {
      real s0 = gn[OFF_kff + lane], s1 = 0.0_r;
#pragma unroll
      for (int c = 0; c < 30; c += 2) { s0 += gn[OFF_KFB + lane * 30 + c] * dxv[c]; s1 += gn[OFF_KFB + lane * 30 + c + 1] * dxv[c + 1]; }
      dut[lane] = s0 + s1;
    }
    QM_LDS_BARRIER();
    real nx = 0.0_r;
    if (wave == 0 && lane < 30) {  // du = Pe + Px dx + Pu du~
      real s0 = stg[OFF_PE + lane], s1 = 0.0_r;
#pragma unroll
      for (int c = 0; c < 30; c += 2) { s0 += stg[OFF_PX + lane * 30 + c] * dxv[c]; s1 += stg[OFF_PX + lane * 30 + c + 1] * dxv[c + 1]; }
      for (int j = 0; j < nt; ++j) s0 += stg[OFF_PU + lane * MT + j] * dut[j];
      a.dU[(size_t(inst) * N + k) * 30 + lane] = s0 + s1;
    }
    if (wave == WX && lane < 30) {  // dx+ = A~ dx + B~ du~ + b~ ; armijo contribution q~ . dx
      real s0 = stg[OFF_bt + lane], s1 = 0.0_r;
#pragma unroll
      for (int c = 0; c < 30; c += 2) { s0 += stg[OFF_AT + lane * 30 + c] * dxv[c]; s1 += stg[OFF_AT + lane * 30 + c + 1] * dxv[c + 1]; }
      for (int j = 0; j < nt; ++j) s0 += stg[OFF_BT + lane * MT + j] * dut[j];
      nx = s0 + s1;
      armijo += stg[OFF_qt + lane] * dxv[lane];
    }
    if (wave == WX && lane >= 32 && lane < 32 + nt) armijo += stg[OFF_rt + (lane - 32)] * dut[lane - 32];
    QM_LDS_BARRIER();
    if (wave == WX && lane < 30) dxv[lane] = nx;
    ph.commit(stgNext, FWD_HEAD, tid);
    pt.commit(stgNext + FWD_TAIL0, FWD_TAIL, tid);
    pg.commit(stgNext + STAGE_DOUBLES, GAIN_DOUBLES, tid);
    QM_LDS_BARRIER();
  }
  if (wave == WX && lane < 30) {
    a.dX[(size_t(inst) * (N + 1) + N) * 30 + lane] = dxv[lane];
    armijo += stagesI[size_t(N) * STAGE_DOUBLES + OFF_qt + lane] * dxv[lane];
  }
  // reduce armijo over the lanes of wavefront WX through LDS
  __syncthreads();
  if (wave == WX) red[lane] = armijo;
  __syncthreads();
  if (tid == 0) {
    real s = 0.0_r;
    for (int i = 0; i < 64; ++i) s += red[i];
    a.instStats[size_t(inst) * 4 + 0] = s;
    a.instStats[size_t(inst) * 4 + 1] = real(status);
  }
}

}  // namespace qmk
