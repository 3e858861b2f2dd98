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
//   P3 H  = L L^T,  W = L^-1 [G | g]       wavefront 0, one column of [H | G g] per lane in registers (49 lanes), row operations with the
//                                          multipliers broadcast by v_readlane / DPP row_newbcast (no LDS round trip on the dependent chain):
//                                          the H lanes end with L, the G lanes with W.  MEANWHILE wavefronts 1..3 -- which have nothing on the
//                                          critical path -- commit the next stage, form and symmetrise  [Q~ | q~] + A~^T [S A~ | y]  (P6a) and
//                                          the gains of the PREVIOUS stage  [K | k] = -L^-T W  by back-substitution (only the forward sweep
//                                          needs them)
//   P6b [S' | s'] = P6a - W^T W            written straight into S; s' is kept as row 30 of S (M carries a unit entry at (30, 30))
// Column 30 carries the affine terms (b~, y, g, k, s') through the same tiles as the matrices.  Zero padding: row / column 31 of S, rows >= m~
// of W and the columns >= m~ of B~ (padded by lq_node_kernel) are exactly zero, so partial tiles need no predication.
#pragma once
#include "layout.h"
#include "gpu_rt.h"

namespace qmk {

struct RiccatiArgs {
  int batch, N;
  const real* stages;   // [batch][N+1][STAGE_DOUBLES]
  const int* stageNc;     // [batch][N+1]
  const real* dtgrid;   // [batch][N+1] step of every node: the joint rows of A~ / B~ are formed from Px / Pu with it (layout.h)
  const real* x0;       // [batch][30]
  const real* X;        // [batch][N+1][30]
  real* gains;          // [batch][N][GAIN_DOUBLES]
  real* dX;             // [batch][N+1][30]
  real* dU;             // [batch][N][30]
  real* instStats;      // [batch][4]: armijo descent metric, status, -, -
  const int* done;        // [batch] converged instances are skipped
};

constexpr int RICCATI_WAVES = 4;
// Phase clocks of the profiling build: QM_TICK* (gpu_rt.h; tools/riccati_phase_probe.py, -DQM_RICCATI_TIMING).  Nothing in the product build.
// LDS strides (doubles) = 16 mod 32: the four k-rows x sixteen consecutive columns one MFMA operand read touches hit distinct banks
constexpr int LDS_S = 50, LDS_Y = 80, LDS_W = 48, LDS_TS = 34, LDS_LL = 18;   // LDS_LL / LDS_S: sixteen lanes one row apart (144 / 400 B) hit distinct banks: column walks are as conflict free as row walks
constexpr int STG_B = OFF_TAIL + 4;               // doubles of a record the backward sweep needs (padded)
constexpr int STG_F = STAGE_DOUBLES + GAIN_DOUBLES;  // record + gains of one stage for the forward sweep
constexpr int R_STG = 0;                          // two staging buffers: [2][STG_B] backward, [2][STG_F] forward (over Y / T, dead by then)
constexpr int R_Y = R_STG + 2 * STG_B;            // Y [32][LDS_Y]
constexpr int R_T = R_Y + 32 * LDS_Y;             // T [32][LDS_Y]
constexpr int R_S = R_T + 32 * LDS_Y;             // S [32][LDS_S]
constexpr int W_DOUBLES = 20 * LDS_W;             // W [20][LDS_W] of one stage
constexpr int LT_DOUBLES = 20 * LDS_LL;            // L [20][LDS_LL] row major, lower triangle; the diagonal slot holds L_cc out of the factorisation and 1 / L_cc from the stage's P6b on (riccatiInvertDiagonal)
constexpr int R_W = R_S + 32 * LDS_S;              // W of the stage in flight and of the previous one (by stage parity): the gains of stage k + 1 are
constexpr int R_LT = R_W + 2 * W_DOUBLES;         // formed while stage k factorises, from W / L^T of stage k + 1
constexpr int R_KST = R_LT + 2 * LT_DOUBLES;        // the gains record [GAIN_DOUBLES] of two stages: formed here by one wavefront, copied to HBM by two others a stage later
constexpr int R_SYM = R_KST + 2 * GAIN_DOUBLES;     // [32][LDS_TS] scratch of the wavefront-local symmetrisation (T is being read by the factorisation at that time)
constexpr int R_Y2 = R_SYM + 32 * LDS_TS;            // [16][LDS_Y] second half of the k sum of Y rows 16..31 when only three column tiles exist (P1 below)
constexpr int R_BWD_END = R_Y2 + 16 * LDS_Y;
static_assert(R_KST % 2 == 0 && GAIN_DOUBLES % 2 == 0, "16-byte copies");
// forward sweep (over everything above, dead by then): a ring of three staging buffers [3][STG_F], then the B-operand images of dx and du~
constexpr int FWD_ZV = 80;                             // z = [dx (30) | du~ (MT) | Px dx + Pu du~ of the joint rows (30, entries 12..29 used) | 2] of one stage
constexpr int F_ZV = 3 * STG_F, R_FWD_END = F_ZV + 3 * FWD_ZV;
constexpr int R_SCR = R_BWD_END > R_FWD_END ? R_BWD_END : R_FWD_END;   // armijo reduction [64]
constexpr int RICCATI_LDS_DOUBLES = R_SCR + 64;
constexpr int RICCATI_LDS_BYTES = RICCATI_LDS_DOUBLES * int(sizeof(real));  // ~145 KiB at fp64 (dynamic LDS)
static_assert(RICCATI_LDS_DOUBLES <= 20480, "one CU's LDS at fp64");

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
  // (only the last round of units can be partial: PF is the rounded-up quotient -- every other round copies without a test)
#define QM_PF_ISSUE(K, V) if constexpr (PF > K) { const int idx = tid + K * NTHR; V = s2[(K + 1 < PF || idx < n2) ? idx : tid]; }
    QM_PF_FOR_EACH(QM_PF_ISSUE)
#undef QM_PF_ISSUE
  }
  __device__ __forceinline__ void commit(real* dst, int n, int tid) const {
    QmD2* d2 = reinterpret_cast<QmD2*>(dst);
    const int n2 = n >> 1;
#define QM_PF_COMMIT(K, V) if constexpr (PF > K) { const int idx = tid + K * NTHR; if (K + 1 < PF || idx < n2) d2[idx] = V; }
    QM_PF_FOR_EACH(QM_PF_COMMIT)
#undef QM_PF_COMMIT
  }
  // the same for a copy that starts at the head of a record: rows 12..29 of the A~ / B~ areas arrive as Px / Pu (layout.h) and land as
  // A~ = e_i + dt Px, B~ = dt Pu -- one multiply-add per number, the operands exactly those lq_node_kernel used when it still wrote these rows.
  // jm: three bits per 16-byte unit of this thread (jointRowMask below): joint row | unit diagonal in .x | in .y
  __device__ __forceinline__ void commitDynamics(real* dst, int n, int tid, unsigned long long jm, real dt) const {
    QmD2* d2 = reinterpret_cast<QmD2*>(dst);
    const int n2 = n >> 1;
#define QM_PF_COMMITD(K, V) if constexpr (PF > K) { const int idx = tid + K * NTHR; if (K + 1 < PF || idx < n2) { const unsigned f = unsigned(jm >> (3 * K)); const real m = (f & 1u) ? dt : 1.0_r; \
      QmD2 w; w.x = fma(V.x, m, (f & 2u) ? 1.0_r : 0.0_r); w.y = fma(V.y, m, (f & 4u) ? 1.0_r : 0.0_r); d2[idx] = w; } }
    QM_PF_FOR_EACH(QM_PF_COMMITD)
#undef QM_PF_COMMITD
  }
#undef QM_PF_FOR_EACH
};

template <int PF, int NTHR> __device__ __forceinline__ unsigned long long jointRowMask(int tid) {
  static_assert(3 * PF <= 64 && OFF_AT == 0 && OFF_BT == 900 && MT % 2 == 0, "one register of flags; a 16-byte unit never straddles two rows");
  unsigned long long jm = 0;
#pragma unroll
  for (int K = 0; K < PF; ++K) {
    const int e = 2 * (tid + K * NTHR);
    const bool inA = e < OFF_BT, inB = e >= OFF_BT && e < OFF_QT;
    const int row = inA ? e / 30 : (inB ? (e - OFF_BT) / MT : 0), col = e - row * 30;
    const bool joint = (inA || inB) && row >= 12;
    if (joint) jm |= 1ull << (3 * K);
    if (joint && inA && col == row) jm |= 2ull << (3 * K);
    if (joint && inA && col + 1 == row) jm |= 4ull << (3 * K);
  }
  return jm;
}

// P3: rows 0..NT-1 of [H | G g] (T, one column per lane: lanes < MT the columns of H, lanes MT..MT+30 those of [G | g]) -> L (row c written by
// lane c, L_cc on the diagonal: inverted later, riccatiInvertDiagonal) and W = L^-1 [G | g].  nt <= NT is the number of real pivots; rows / columns nt..NT-1 are identity.
// One elimination step, written as a template recursion so that the DPP controls are immediates.  Multipliers L[r][J] = (scaled row J)
// at lane r: rows J + 1 (on the pivot chain) and J + 2 take them by v_readlane, rows J + 3 .. 15 by DPP row_newbcast from a copy of the
// row's lanes 0..15 replicated into the four rows of 16 lanes -- ONE v_fmac_f64_dpp per row update instead of two v_readlane, a wait
// state and the multiply-add -- applied one step late so that the replication's round trip through the LDS crossbar is off the chain;
// rows 16, 17 (m~ > 16) by v_readlane.  Row updates of one row commute, so the order does not matter.
template <int J, int R, int REND> struct RiccatiDppRows {
  static __device__ __forceinline__ void run(real* col, real bc, real nc, real* scr) {
    if constexpr (R < REND) { qmFmacRowBcast<R, R == J + 3>(col[R], bc, nc, scr); RiccatiDppRows<J, R + 1, REND>::run(col, bc, nc, scr); }
  }
};
template <int NT, int J> struct RiccatiStep {
  static constexpr int DEND = NT < 16 ? NT : 16;    // DPP rows end here
  static __device__ __forceinline__ void run(real* col, real& inv, real& bcP, real& ncP, int& status, real* scr, real* out, int ostr) {
    if constexpr (J < NT) {
      col[J] *= inv;                                     // row J of [L^T | W] / sqrt(pivot)
      // final: element (c, J) of L for an H lane, (J, c) of W for a G lane (asynchronous LDS write); lane J's own entry is L_JJ.  (Until round 6 every lane
      // carried the reciprocal of "its" diagonal entry through the steps for the gains' back-substitution; the compiler kept all the reciprocals and lane masks
      // alive to the end and selected there -- ~50 instructions behind the last pivot, on the critical wavefront, and spilled masks inside the loop.  The
      // diagonal is inverted after the factorisation now, while this wavefront has nothing to do: riccatiInvertDiagonal.)
      *out = col[J];
      out += ostr;
      const QmGather gj = qmGather(col[J], scr);
      if constexpr (J + 1 < NT) {
        col[J + 1] -= gj.get(J + 1) * col[J];
        const real piv = qmReadLane(col[J + 1], J + 1, scr);
        if (!(piv > REAL_PIVOT_MIN)) status = 1;         // beside the chain: a failed pivot (not a positive number) flags the instance, whose step is then discarded (linesearch_kernel) ...
        inv = qmRsqrtPos(fmax(piv, REAL_PIVOT_MIN));     // ... and the factorisation runs on with the pivot floored: ONE instruction on the chain (fmax drops a NaN) where the exact
                                                         // select "failed ? 1 : piv" was a compare, two scalar selects and their way back to the vector unit.  Started here: its latency hides behind the remaining updates
      }
      if constexpr (J + 2 < NT) col[J + 2] -= gj.get(J + 2) * col[J];
      if constexpr (J >= 1) RiccatiDppRows<J - 1, J + 2, DEND>::run(col, bcP, ncP, scr);     // the previous step's rows J + 2 .. 15
#pragma unroll
      for (int r = (J + 3 > 16 ? J + 3 : 16); r < NT; ++r) col[r] -= gj.get(r) * col[J];
      if constexpr (J + 3 < DEND) { bcP = qmReplicateRow0(col[J], scr); ncP = -col[J]; }
      RiccatiStep<NT, J + 1>::run(col, inv, bcP, ncP, status, scr, out, ostr);
    }
  }
};

template <int NT> __device__ __forceinline__ void riccatiFactorise(const real* T, real* W, real* LL, int nt, int lane, int& status, real* scr, unsigned long long* tk = nullptr) {
  const bool isH = lane < MT, isG = lane >= MT && lane < MT + 31;
  const int c = isH ? lane : (isG ? lane - MT : 0);
#ifdef QM_RICCATI_TIMING
  const unsigned long long tqA = clock64();
#endif
  real col[NT];
#pragma unroll
  for (int r = 0; r < NT; ++r) col[r] = T[r * LDS_Y + (isH ? 32 + c : c)];
#pragma unroll
  for (int r = 0; r < NT; ++r) QM_KEEP(col[r]);
  const bool live = isH ? c < nt : isG;
#pragma unroll
  for (int r = 0; r < NT; ++r) {
    const real e = (isH && r == c) ? 1.0_r : 0.0_r;
    col[r] = (live && r < nt) ? col[r] : e;
  }
#ifdef QM_RICCATI_TIMING
  const unsigned long long tq0 = clock64();
#endif
  // steps j >= nt meet identity columns (pivot 1, multipliers 0): no branch, one basic block
  real inv, bcP = 0.0_r, ncP = 0.0_r;
  {
    const real piv = qmReadLane(col[0], 0, scr);
    if (!(piv > REAL_PIVOT_MIN)) status = 1;
    inv = qmRsqrtPos(fmax(piv, REAL_PIVOT_MIN));
  }
  // every lane streams its finished rows out as the elimination goes: H lane c writes row c of L (entries right of the diagonal are
  // elimination residue and never read), G lane c column c of W, the idle lanes a scratch word (row 19 of L)
  real* out = isH ? LL + c * LDS_LL : (isG ? W + c : LL + 19 * LDS_LL);
  const int ostr = isH ? 1 : (isG ? LDS_W : 0);
  RiccatiStep<NT, 0>::run(col, inv, bcP, ncP, status, scr, out, ostr);
#ifdef QM_RICCATI_TIMING
  { real keep_ = col[NT - 1]; QM_KEEP(keep_); col[NT - 1] = keep_; }
  const unsigned long long tq1 = clock64();
  tk[0] += tq0 - tqA; tk[1] += tq1 - tq0;
#endif
#pragma unroll
  for (int r = NT; r < MT; ++r) out[r * ostr] = 0.0_r;   // identity rows / columns beyond the unrolled size; rows of W beyond m~ (the buffer may hold a stage with more inputs)
#ifdef QM_RICCATI_TIMING
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  tk[2] += clock64() - tq1;
#endif
}

// The factorisation leaves L_qq in the diagonal slots of L; the gains' back-substitution multiplies by 1 / L_qq.  Lane q inverts its own entry -- by the factorising
// wavefront itself behind the stage's third barrier, where it waits for the others' P6b and has nothing else to do.  A slot that does not hold a positive
// number (a failed factorisation: flagged, its step is discarded; or a row beyond the unrolled size: zero, never read) is taken as 1, as the factorisation took it.
__device__ __forceinline__ void riccatiInvertDiagonal(real* LL, int lane) {
  if (lane < MT) { const real dq = LL[lane * LDS_LL + lane]; LL[lane * LDS_LL + lane] = qmRcpPos(dq > REAL_PIVOT_MIN ? dq : 1.0_r); }
}

// [K | k] = -L^-T W of one stage by back-substitution, one column of [K | k] per lane (31 lanes of one wavefront), in the axpy order:
// the dependent chain is one multiply + one multiply-add per row, the other multiply-adds of a step are independent.
// Rows QLO .. QHI - 1 of L (each up to and including its diagonal slot = 1 / L_qq, the same address in every lane) are requested TOGETHER and the substitution
// steps of those rows follow: one LDS latency per batch.  (Row by row -- the loop as written until round 6 -- every step waited for its own row: sixteen
// latencies in a row, two thirds of this wavefront's phase.)
template <int QLO, int QHI> __device__ __forceinline__ void riccatiGainsRows(real* w, const real* LLp) {
  if constexpr (QHI > QLO) {
    constexpr int NB = QHI - QLO, NP = (QHI + 1) / 2;   // pairs of the longest row
    QmD2 lr[NB][NP];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) if (2 * pr <= QLO + b) lr[b][pr] = *reinterpret_cast<const QmD2*>(LLp + (QLO + b) * LDS_LL + 2 * pr);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) if (2 * pr <= QLO + b) QM_KEEP(lr[b][pr]);
    }
#pragma unroll
    for (int q = QHI - 1; q >= QLO; --q) {
      const int b = q - QLO;
      w[q] *= (q & 1) ? lr[b][q >> 1].y : lr[b][q >> 1].x;
#pragma unroll
      for (int r = 0; r < q; ++r) w[r] -= ((r & 1) ? lr[b][r >> 1].y : lr[b][r >> 1].x) * w[q];
    }
  }
}
template <int NTP> __device__ __forceinline__ void riccatiGainsN(const real* Wp, const real* LLp, int ntp, int lane, real* kst) {
  real w[NTP];   // rows >= NTP of W are zero and those rows of L identity: their gains are zero
#pragma unroll
  for (int r = 0; r < NTP; ++r) w[r] = Wp[r * LDS_W + lane];
  constexpr int B3 = NTP < 16 ? NTP : 16, B2 = NTP < 13 ? NTP : 13, B1 = NTP < 9 ? NTP : 9, B0 = NTP < 4 ? NTP : 4;   // <= 24 pairs (96 registers) per batch
  riccatiGainsRows<B3, NTP>(w, LLp);
  riccatiGainsRows<B2, B3>(w, LLp);
  riccatiGainsRows<B1, B2>(w, LLp);
  riccatiGainsRows<B0, B1>(w, LLp);
  riccatiGainsRows<0, B0>(w, LLp);
  // into the LDS image of the record (immediate-offset ds_write, consecutive lanes consecutive addresses)
  real* gp = kst + (lane < 30 ? OFF_KFB + lane : OFF_kff);
  if (lane < 30) {
#pragma unroll
    for (int r = 0; r < MT; ++r) gp[r * 30] = r < NTP ? (r < ntp ? -w[r < NTP ? r : 0] : 0.0_r) : 0.0_r;
  } else {
#pragma unroll
    for (int r = 0; r < MT; ++r) gp[r] = r < NTP ? (r < ntp ? -w[r < NTP ? r : 0] : 0.0_r) : 0.0_r;
  }
}
__device__ __forceinline__ void riccatiGains(const real* Wp, const real* LLp, int ntp, int lane, real* kst) {
  switch (ntp) {   // unrolled for the stage's number of projected inputs, as the factorisation
    case 16: riccatiGainsN<16>(Wp, LLp, ntp, lane, kst); break;
    case 14: riccatiGainsN<14>(Wp, LLp, ntp, lane, kst); break;
    case 17: riccatiGainsN<17>(Wp, LLp, ntp, lane, kst); break;
    default: riccatiGainsN<MT>(Wp, LLp, ntp, lane, kst); break;
  }
}
__device__ __forceinline__ void riccatiGainsOut(const real* kst, real* gain, int t) {
  const QmD2* src = reinterpret_cast<const QmD2*>(kst);
  QmD2* dst = reinterpret_cast<QmD2*>(gain);
#pragma unroll
  for (int i = 0; i < (GAIN_DOUBLES / 2 + 127) / 128; ++i) { const int idx = t + 128 * i; if (idx < GAIN_DOUBLES / 2) dst[idx] = src[idx]; }
}

#ifdef QM_RICCATI_TIMING
#define QM_TK , qmTs + 19
#else
#define QM_TK
#endif
template <int NW> __global__ void __launch_bounds__(NW * 64) QM_ONE_WAVE_PER_SIMD riccati_kernel(RiccatiArgs a) {
  static_assert(NW == 4, "tile ownership below is written for four wavefronts");
  QM_DYNAMIC_LDS(lds);
  QM_POISON_LDS(lds, RICCATI_LDS_DOUBLES);
  constexpr int NTHR = NW * 64;
  constexpr int PFB = (OFF_TAIL / 2 + NTHR - 1) / NTHR;
  constexpr int NCP = 128;                                      // the same copy during the factorisation: by wavefronts 1 and 3 (wavefront 2 forms a tile AND the deferred gains: it is as long as the factorisation itself)
  constexpr int PFW = (OFF_TAIL / 2 + NCP - 1) / NCP;
  // forward sweep: only the head of the record (A~ B~ rows 0..11, Px Pu rows 12..29) and b~ q~ r~, Pu rows 0..11, Pe (its tail) are read; Q~ P~ R~ in between are not
  constexpr int FWD_HEAD = OFF_QT, FWD_TAIL0 = OFF_bt;
  static_assert(FWD_HEAD % 2 == 0 && FWD_TAIL0 % 2 == 0, "16-byte units");
  constexpr int PFH = (FWD_HEAD / 2 + NTHR - 1) / NTHR;
  constexpr int PFG = (GAIN_DOUBLES / 2 + NTHR - 1) / NTHR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, h = lane >> 4, la = qmARow(l16);   // MFMA operand coordinates of this lane; la: row of an A operand (gpu_rt.h)
  const int inst = blockIdx.x;
  if (a.done[inst]) return;   // workgroup uniform
  const int N = a.N;
  real* S = lds + R_S; real* Y = lds + R_Y; real* T = lds + R_T;
  real* scr = nullptr; real* red = lds + R_SCR;
  const real* stagesI = a.stages + size_t(inst) * (N + 1) * STAGE_DOUBLES;
  const real* gainsI = a.gains + size_t(inst) * N * GAIN_DOUBLES;
  const int* ncI = a.stageNc + size_t(inst) * (N + 1);
  const real* dtI = a.dtgrid + size_t(inst) * (N + 1);
  int status = 0;

  // ---- terminal value function S_N = Q_N, s_N = q_N (zero padded), and the first stage to process
  {
    const real* rec = stagesI + size_t(N) * STAGE_DOUBLES;
    // ONE memory round trip for the whole prologue: the record of stage N - 1 is requested first, the terminal Q_N with s_N = q_N as its row 30 (the B operands
    // carry a unit entry at (30, 30)) behind it, and nothing is waited for before all of them are on their way (were three round trips and a barrier in a row: Q_N,
    // then q_N into the row the first pass had zeroed, then the record)
    StagePrefetch<PFB, NTHR> pf;
    pf.issue(stagesI + size_t(N - 1) * STAGE_DOUBLES, OFF_TAIL, tid);
    const real dtLast = dtI[N - 1];
    {   // all of a thread's loads before its first LDS store (a load may not pass the store in front of it: seven memory round trips in a row otherwise)
      constexpr int NS = (32 * LDS_S + NTHR - 1) / NTHR;
      real sv[NS];
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const int e = tid + q * NTHR, i = e / LDS_S, j = e % LDS_S;
        const bool inQ = e < 32 * LDS_S && i < 30 && j < 30, inq = e < 32 * LDS_S && i == 30 && j < 30;
        sv[q] = (inQ || inq) ? rec[inq ? OFF_qt + j : OFF_QT + i * 30 + j] : 0.0_r;
      }
#pragma unroll
      for (int q = 0; q < NS; ++q) QM_KEEP(sv[q]);
#pragma unroll
      for (int q = 0; q < NS; ++q) { const int e = tid + q * NTHR; if (e < 32 * LDS_S) S[e] = sv[q]; }
    }
    for (int e = tid; e < 2 * W_DOUBLES + 2 * LT_DOUBLES + 2 * GAIN_DOUBLES; e += NTHR) lds[R_W + e] = 0.0_r;   // W, L, gains images of both parities (contiguous)
    for (int e = tid; e < 2 * 32 * LDS_Y; e += NTHR) Y[e] = 0.0_r;        // Y, T (contiguous)
    for (int e = tid; e < 16 * LDS_Y; e += NTHR) lds[R_Y2 + e] = 0.0_r;
    pf.commitDynamics(lds + R_STG + ((N - 1) & 1) * STG_B, OFF_TAIL, tid, jointRowMask<PFB, NTHR>(tid), dtLast);
  }
  __syncthreads();
  QM_TICK_DECL;

  // ---- operand addressing of the products, per lane and independent of the stage: column jc of M = [A~ | b~ | . | B~ | .] (P1) and of
  //      [P~ | r~ | . | R~ | .] (P2), entry (i, j) of [Q~ | q~] (P6a).  The producers pad B~ with zero columns beyond m~ and the factorisation
  //      ignores rows / columns of T beyond m~, so nothing here depends on m~: the offsets are formed once, a lane without a source reads
  //      offset 0 (a finite number) and multiplies by zero.
  const int jc = wave * 16 + l16;          // my column of M / Y / T
  int mOffK[8]; real mOne, mAdd7;          // P1: offset of M[k][jc] for my eight k; mAdd7: the unit entry at (30, 30) that picks s out of row 30 of S
  int cOffR[8]; real cOne;                 // P2: offsets of the initial value of T[i][jc], i = h + 4 r and 16 + h + 4 r
  int bOffK[8]; real b1One;                // P2: offsets of B~[k][la] (second row tile: B~[k][16 + la], only 16 + la < MT exists)
  {
    const bool jA = jc < 30, jb = jc == 30, jB = jc >= 32 && jc < 32 + MT;
    const int mOff = jA ? OFF_AT + jc : (jb ? OFF_bt : (jB ? OFF_BT + (jc - 32) : 0));
    const int mStr = jA ? 30 : (jb ? 1 : (jB ? MT : 0));
    mOne = (jA || jb || jB) ? 1.0_r : 0.0_r;
    const int cOff = jA ? OFF_PT + jc : (jb ? OFF_rt : (jB ? OFF_RT + (jc - 32) : 0));
    const int cStr = jA ? 30 : (jb ? 1 : (jB ? MT : 0));
    cOne = mOne;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int kk = 4 * ks + h, kc = kk < 30 ? kk : 29;     // rows 30, 31: S[30] = s meets the unit entry below, S[31] = 0
      mOffK[ks] = mOff + kc * mStr;
      bOffK[ks] = OFF_BT + kc * MT + la;
      QM_KEEP(mOffK[ks]); QM_KEEP(bOffK[ks]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i0 = h + 4 * r, i1 = 16 + h + 4 * r;
      const int i1c = i1 < MT ? i1 : 0;
      cOffR[r] = cOff + i0 * cStr; cOffR[4 + r] = cOff + i1c * cStr;
      if (jB) {   // R~ is stored as its lower triangle (lq_node_kernel): an entry above the diagonal is read from its mirror image
        const int jr = jc - 32;
        if (jr > i0) cOffR[r] = OFF_RT + jr * MT + i0;
        if (jr > i1c) cOffR[4 + r] = OFF_RT + jr * MT + i1c;
      }
      QM_KEEP(cOffR[r]); QM_KEEP(cOffR[4 + r]);
    }
    mAdd7 = (h == 2 && jb) ? 1.0_r : 0.0_r;
    b1One = 16 + la < MT ? 1.0_r : 0.0_r;
    QM_KEEP(mOne); QM_KEEP(cOne); QM_KEEP(mAdd7); QM_KEEP(b1One);
  }
  const real m7One = h >= 2 ? 0.0_r : mOne;   // k step 7: rows 30, 31 of M carry no data
  // P1 with three column tiles (m~ <= 16: trot, flight): 6 output tiles x 8 k steps on four wavefronts = 12 matrix-core instructions each instead of 16 on three --
  // wavefronts 0..2 stop the lower row tile (rows 16..31) of their column tile after k steps 0..3, wavefront 3 forms k steps 4..7 of all three into Y2, and the
  // consumers of those rows (P2, P6a) add the two halves.  Its operand offsets, for the column tiles t = 0, 1, 2 and k steps 4..7:
  constexpr int KS3 = 5;   // first k step of wavefront 3: 8 + 5 = 13 instructions on wavefronts 0..2, 3 x 3 = 9 on wavefront 3 (whose operand set is three times as wide)
  int w3Off[3][8 - KS3]; real w3One[3], w3Add7[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int jt = t * 16 + l16;
    const bool jA = jt < 30, jb = jt == 30, jB = jt >= 32;
    const int off = jA ? OFF_AT + jt : (jb ? OFF_bt : (jB ? OFF_BT + (jt - 32) : 0)), str = jA ? 30 : (jb ? 1 : (jB ? MT : 0));
#pragma unroll
    for (int q = 0; q < 8 - KS3; ++q) { const int kk = 4 * (KS3 + q) + h, kc = kk < 30 ? kk : 29; w3Off[t][q] = off + kc * str; QM_KEEP(w3Off[t][q]); }
    w3One[t] = (jA || jb || jB) ? 1.0_r : 0.0_r; w3Add7[t] = (h == 2 && jb) ? 1.0_r : 0.0_r;
    QM_KEEP(w3One[t]); QM_KEEP(w3Add7[t]);
  }
  int ncCur = ncI[N - 1], ncPrev = 0;   // constraint rows of stage k and of stage k + 1; the next one is loaded a stage ahead
  // the staged copy of the next record is made by wavefronts 1 and 3 (13 units of 16 bytes per thread); wavefront 2 also runs the deferred gains
  const int ptid = wave == 3 ? 64 + lane : lane;
  const bool copier = wave == 1 || wave == 3;
  const unsigned long long jmW = copier ? jointRowMask<PFW, NCP>(ptid) : 0ull;   // which of my units of the staged copy are joint-row entries
#pragma unroll 1
  for (int k = N - 1; k >= 0; --k) {
    const real* stg = lds + R_STG + (k & 1) * STG_B;        // this stage (committed during the previous one)
    real* stgNext = lds + R_STG + ((k + 1) & 1) * STG_B;    // buffer of stage k - 1
    const int nt = 30 - ncCur;
    // constraint rows of the stage after this one: wanted at the END of this stage.  Requested here through an address the compiler cannot prove uniform, and made
    // a scalar down there: as a uniform load it was a global_load followed at once by s_waitcnt vmcnt(0) + v_readfirstlane -- a trip to the L2 (~300 cycles) on
    // every wavefront at the head of every stage (round 6: the "issue" slot of the phase clocks).
    const int ncLoadV = ncI[(k > 0 ? k - 1 : 0) + qmOpaqueLane(0)];
    const int mtTiles = nt > 16 ? 2 : 1;     // 16-row tiles covering the m~ projected inputs
    const int nTiles = nt > 16 ? 4 : 3;      // 16-column tiles covering [A~ | b~ | . | B~]
    QM_TICK(0);
    // ---- P1 + P2: wavefront w owns the 16 columns [16 w, 16 w + 16) of Y and of T
    const bool splitK = nTiles == 3;                 // (wave uniform, per stage)
    const real y2On = splitK ? 1.0_r : 0.0_r;        // consumers of Y rows 16..31 add Y2 times this (Y2 always holds finite numbers)
    real* Y2 = lds + R_Y2;
    // P6a of one 16 x 16 tile (tm, tn) of [Q~ | q~] + A~^T [S A~ | y] (needs all of Y: behind the stage's first barrier), in accumulator layout
    auto p6aTile = [&](int tm, int tn) -> QmAcc {
      const int j = tn * 16 + l16;
      real qv[4], qq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = tm * 16 + h + 4 * r, ic = i < 30 ? i : 0;
        const int jq = j < 30 ? j : 0;   // diagonal tiles: only the upper triangle of the symmetric Q~ is stored (lq_node_kernel)
        qv[r] = stg[OFF_QT + (tm == tn && jq < ic ? jq * 30 + ic : ic * 30 + jq)]; qq[r] = stg[OFF_qt + ic];
      }
      const int ai = tm * 16 + la < 30 ? tm * 16 + la : 29;   // rows 30,31 of the result are discarded
      real av[8], bw[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int kk = 4 * ks + h, kc = kk < 30 ? kk : 29;      // rows 30,31 of Y are zero
        av[ks] = stg[OFF_AT + kc * 30 + ai]; bw[ks] = Y[kk * LDS_Y + j];   // A~^T[i][k] = A~[k][i]
      }
#pragma unroll
      for (int ks = 4; ks < 8; ++ks) bw[ks] = fma(y2On, lds[R_Y2 + (4 * (ks - 4) + h) * LDS_Y + j], bw[ks]);   // rows 16..31 of Y: the other half of the k sum (P1)
      QmAcc c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        QM_KEEP(qv[r]); QM_KEEP(qq[r]);
        const int i = tm * 16 + h + 4 * r;
        const real v = j < 30 ? qv[r] : (j == 30 ? qq[r] : 0.0_r);
        c[r] = i < 30 ? v : 0.0_r;
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) qmMfma(c, av[ks], bw[ks], scr);
      return c;
    };
    // The off-diagonal tile (0,1) belongs to wavefront 2, which also forms the gains of the previous stage -- with both it was the longest wavefront of the
    // factorisation phase once the factorisation had lost its tail (round 6).  In stages with three column tiles (m~ <= 16) wavefront 3 has no part in P2:
    // it forms that tile THERE and parks it in the free square of the symmetrisation scratch (rows 0..15, columns 16..31); wavefront 2 picks it up behind its gains.
    const bool early01 = nTiles == 3;
    if (wave < nTiles) {
      QmAcc c0, c1;
      real a0[8], a1[8], bv[8];   // all operands first: the LDS latency is paid once
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int kk = 4 * ks + h;
        a0[ks] = S[kk * LDS_S + la]; a1[ks] = S[kk * LDS_S + 16 + la];  // S is symmetric: S[i][k] read as S[k][i]; row 30 is s
        bv[ks] = stg[mOffK[ks]];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { c0[r] = 0.0_r; c1[r] = 0.0_r; }
#pragma unroll
      for (int ks = 0; ks < 7; ++ks) bv[ks] *= mOne;
      bv[7] = bv[7] * m7One + mAdd7;
      if (splitK) {
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks) { qmMfma(c0, a0[ks], bv[ks], scr); qmMfma(c1, a1[ks], bv[ks], scr); }
#pragma unroll
        for (int ks = KS3; ks < 8; ++ks) qmMfma(c0, a0[ks], bv[ks], scr);
      } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { qmMfma(c0, a0[ks], bv[ks], scr); qmMfma(c1, a1[ks], bv[ks], scr); }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { Y[(h + 4 * r) * LDS_Y + jc] = c0[r]; Y[(16 + h + 4 * r) * LDS_Y + jc] = c1[r]; }
    } else if (splitK) {   // wavefront 3: k steps 4..7 of the lower row tile of the three column tiles
      constexpr int NQ = 8 - KS3;
      real a1[NQ], bt[3][NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int kk = 4 * (KS3 + q) + h;
        a1[q] = S[kk * LDS_S + 16 + la];
#pragma unroll
        for (int t = 0; t < 3; ++t) bt[t][q] = stg[w3Off[t][q]];
      }
      QmAcc d[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) d[t][r] = 0.0_r;
#pragma unroll
        for (int q = 0; q < NQ - 1; ++q) bt[t][q] *= w3One[t];
        bt[t][NQ - 1] = bt[t][NQ - 1] * (h >= 2 ? 0.0_r : w3One[t]) + w3Add7[t];
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int t = 0; t < 3; ++t) qmMfma(d[t], a1[q], bt[t][q], scr);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Y2[(h + 4 * r) * LDS_Y + t * 16 + l16] = d[t][r];
      }
    }
    QM_TICK(1);
    QM_LDS_BARRIER();
    QM_TICK(2);
    if (wave < nTiles) {
      QmAcc c0, c1;
      real ci[8], a0[8], a1[8], bv[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) ci[r] = stg[cOffR[r]];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int kk = 4 * ks + h;          // rows 30, 31 of Y are zero
        bv[ks] = Y[kk * LDS_Y + jc];
        a0[ks] = stg[bOffK[ks]];            // B~^T[i][k] = B~[k][i]; columns beyond m~ are zero in the record
      }
#pragma unroll
      for (int ks = 4; ks < 8; ++ks) bv[ks] = fma(y2On, Y2[(4 * (ks - 4) + h) * LDS_Y + jc], bv[ks]);   // rows 16..31 of Y: the other half of the k sum (P1)
#pragma unroll
      for (int r = 0; r < 4; ++r) c0[r] = ci[r] * cOne;
      if (mtTiles == 2) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) a1[ks] = stg[bOffK[ks] + (16 + la < MT ? 16 : 0)];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) QM_KEEP(a1[ks]);   // all eight reads before the first product (left alone, each read sat with its own wait in front of its instruction)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) a1[ks] *= b1One;
#pragma unroll
        for (int r = 0; r < 4; ++r) c1[r] = ci[4 + r] * cOne;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { qmMfma(c0, a0[ks], bv[ks], scr); qmMfma(c1, a1[ks], bv[ks], scr); }
#pragma unroll
        for (int r = 0; r < 4; ++r) { T[(h + 4 * r) * LDS_Y + jc] = c0[r]; T[(16 + h + 4 * r) * LDS_Y + jc] = c1[r]; }
      } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qmMfma(c0, a0[ks], bv[ks], scr);
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(h + 4 * r) * LDS_Y + jc] = c0[r];
      }
    }
    if (wave == 3 && early01) {
      const QmAcc ce = p6aTile(0, 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[R_SYM + (h + 4 * r) * LDS_TS + 16 + l16] = ce[r];
    }
    QM_TICK(3);
    QM_LDS_BARRIER();
    QM_TICK(4);
    // ---- P3 on wavefront 0; P6a + deferred gains on wavefronts 1..3
    real* W = lds + R_W + (k & 1) * W_DOUBLES;
    real* LL = lds + R_LT + (k & 1) * LT_DOUBLES;
    // S' is symmetric: only the tiles (0,0), (0,1) and (1,1) of the 32 x 32 update are formed, by wavefronts 1, 2, 3; the tile (1,0) is
    // the mirror of (0,1).  Tile t = (t >> 1, t & 1).
    const int myTile = wave == 1 ? 0 : (wave == 2 ? 1 : 3);
    QmAcc c6;
    if (wave == 0) {
      // the elimination is unrolled for the stage's number of projected inputs: 18 stance, 17 three-leg support, 16 trot, 14 flight
      __builtin_amdgcn_s_setprio(3);    // the wavefront on the critical path of the stage goes first at the shared units (LDS)
      switch (nt) {
        case 16: riccatiFactorise<16>(T, W, LL, nt, lane, status, scr QM_TK); break;
        case 14: riccatiFactorise<14>(T, W, LL, nt, lane, status, scr QM_TK); break;
        case 17: riccatiFactorise<17>(T, W, LL, nt, lane, status, scr QM_TK); break;
        default: riccatiFactorise<MT>(T, W, LL, nt, lane, status, scr QM_TK); break;
      }
      __builtin_amdgcn_s_setprio(0);
    } else {
      // the next stage's blocks HBM -> registers -> LDS by the three wavefronts that are off the critical path here; the other
      // staging buffer was last read before the final barrier of the previous stage
      StagePrefetch<PFW, NCP> pf;
      if (copier) pf.issue(stagesI + size_t(k > 0 ? k - 1 : 0) * STAGE_DOUBLES, OFF_TAIL, ptid);
      // ---- P6a: [Q~ | q~] + A~^T [S A~ | y] for my tile (independent of the factorisation), symmetrised here: (C + C^T) / 2 on the diagonal
      //      tiles through a scratch square inside the wavefront.  W^T W, subtracted after the factorisation, is symmetric bit for bit
      //      (the same products in the same order on both sides), so S' needs no second pass.  Without the symmetrisation the
      //      antisymmetric part of the rounding error is propagated by the OPEN-loop dynamics (it sees A~^T . A~ but not the cancelling
      //      G^T H^-1 G) and grows ~1.13x per stage.
      const int tm = myTile >> 1, tn = myTile & 1, j = tn * 16 + l16;
      if (wave != 2 || !early01) c6 = p6aTile(tm, tn);
      if (wave != 2) {    // diagonal tiles (0,0) and (1,1); column 30 (s') and the rows / columns beyond 29 stay as they are
        real* SYM = lds + R_SYM;
#pragma unroll
        for (int r = 0; r < 4; ++r) SYM[(tm * 16 + h + 4 * r) * LDS_TS + j] = c6[r];
        QM_WAVE_SYNC();   // the transposed read meets this wavefront's own writes (LDS operations of one wavefront complete in order)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = tm * 16 + h + 4 * r;
          const real m = SYM[j * LDS_TS + i];
          c6[r] = (i < 30 && j < 30) ? 0.5_r * (c6[r] + m) : c6[r];
        }
      }
      // ---- gains of stage k + 1 (its W and L sit in the other parity's buffers)
      //      into the LDS image of their record (wavefront 2: its tile needs no symmetrisation); wavefronts 1 and 3 send the image finished a
      //      stage ago (stage k + 2) to HBM
      if (wave == 2) {
        if (k + 1 < N && lane < 31)
          riccatiGains(lds + R_W + ((k + 1) & 1) * W_DOUBLES, lds + R_LT + ((k + 1) & 1) * LT_DOUBLES, 30 - ncPrev, lane, lds + R_KST + ((k + 1) & 1) * GAIN_DOUBLES);
        if (early01) {
#pragma unroll
          for (int r = 0; r < 4; ++r) c6[r] = lds[R_SYM + (h + 4 * r) * LDS_TS + 16 + l16];
        }
      } else if (k + 2 < N) {
        riccatiGainsOut(lds + R_KST + (k & 1) * GAIN_DOUBLES, a.gains + (size_t(inst) * N + k + 2) * GAIN_DOUBLES, wave == 1 ? lane : 64 + lane);
      }
      if (copier) pf.commitDynamics(stgNext, OFF_TAIL, ptid, jmW, stg[OFF_DTPREV]);   // stage k - 1 lands in the other buffer, its joint rows as A~ / B~ (its step came with stage k)
    }
    QM_TICK(5);
    QM_TICK(6);
    QM_LDS_BARRIER();
    QM_TICK(7);
    // ---- P6b: S' = P6a - W^T W on the tiles' owners, written straight into S (s' = column 30 -> row 30 of S); the owner of the tile
    //      (0,1) also writes its mirror image (1,0)
    if (wave == 0) riccatiInvertDiagonal(LL, lane);
    if (wave != 0) {
      const int tm = myTile >> 1, tn = myTile & 1, j = tn * 16 + l16;
      real av[5], bw[5];
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) { const int kk = 4 * ks + h; av[ks] = -W[kk * LDS_W + tm * 16 + la]; bw[ks] = W[kk * LDS_W + j]; }
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) qmMfma(c6, av[ks], bw[ks], scr);
      // one store per accumulator register, no branch: an entry of the padding (row / column 30, 31: they must stay zero) goes to a scratch word of the
      // symmetrisation square (free here), the s' entry (column 30) to row 30 of S; the mirror image of the off-diagonal tile likewise
      real* sink = lds + R_SYM + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = tm * 16 + h + 4 * r;
        const bool inS = i < 30 && j < 30;
        real* dst = inS ? S + i * LDS_S + j : ((i < 30 && j == 30) ? S + 30 * LDS_S + i : sink);
        *dst = c6[r];
        if (wave == 2) { real* mir = inS ? S + j * LDS_S + i : sink + 64; *mir = c6[r]; }
      }
    }
    QM_TICK(8);
    QM_TICK(9);
    QM_TICK(10);
    QM_LDS_BARRIER();
    QM_TICK(11);
    ncPrev = ncCur; ncCur = qmReadLaneInt(qmOpaqueLane(ncLoadV), 0);
  }
  // ---- gains of stage 0 (nobody factorises any more)
  if (wave == 2) { if (lane < 31) riccatiGains(lds + R_W, lds + R_LT, 30 - ncPrev, lane, lds + R_KST); }
  else if (wave != 0 && N > 1) riccatiGainsOut(lds + R_KST + GAIN_DOUBLES, a.gains + (size_t(inst) * N + 1) * GAIN_DOUBLES, wave == 1 ? lane : 64 + lane);
  QM_LDS_BARRIER();
  if (wave == 1 || wave == 3) riccatiGainsOut(lds + R_KST, a.gains + size_t(inst) * N * GAIN_DOUBLES, wave == 1 ? lane : 64 + lane);

  // ================================================================== forward substitution
  // The recursion  du~ = K dx + k,  dx+ = A~ dx + B~ du~ + b~  runs on wavefront 0 alone (a v_mfma_f64 holds a SIMD's matrix pipe for 64
  // cycles whatever its operands, so a matrix-vector product is cheaper as plain multiply-adds: each row's dot product is split over the
  // two halves of the wavefront and joined by one v_permlane32_swap); the hand-off du~ -> second product and dx+ -> next stage goes
  // through LDS inside the wavefront (z = [dx | du~], 48 numbers per stage), and the workgroup meets at ONE barrier per stage.
  // Everything off the chain runs one stage behind, from rings of three buffers: wavefront 1 forms du = Pe + [Px | Pu] z with its rows
  // of Px / Pu read straight from HBM into registers a stage ahead, wavefront 2 stores dx and accumulates the Armijo slope
  // q~.dx + r~.du~, and wavefronts 2 and 3 stream A~, B~, b~, q~, r~ and the gains of the next stage HBM -> registers -> LDS.
  // All MT columns of B~ / Pu / K are multiplied: the producers pad with zeros (lq_node_kernel; riccatiGains).
  constexpr int ZV = FWD_ZV;
  constexpr int FWD_SMALL = STAGE_DOUBLES - OFF_bt;   // b~ q~ r~ (neighbours' steps) Pe, mode, step (+ padding)
  static_assert(FWD_SMALL % 2 == 0 && (OFF_AT + 24) % 2 == 0 && OFF_BT % 2 == 0 && MT % 2 == 0, "16-byte units");
  constexpr int NPF = 64;           // wavefronts 2 and 3 stream the blocks, a whole stage each, taking turns
  constexpr int PFH3 = (FWD_HEAD / 2 + NPF - 1) / NPF, PFT3 = (FWD_SMALL / 2 + NPF - 1) / NPF, PFG3 = (GAIN_DOUBLES / 2 + NPF - 1) / NPF;
  constexpr int PFS = (FWD_SMALL / 2 + NTHR - 1) / NTHR;
  QM_TICK(12);
  __syncthreads();   // full barrier: the gains written to HBM above are read back below
  {
    StagePrefetch<PFH, NTHR> ph;
    StagePrefetch<PFS, NTHR> pt;
    StagePrefetch<PFG, NTHR> pg;
    ph.issue(stagesI, FWD_HEAD, tid);
    pt.issue(stagesI + FWD_TAIL0, FWD_SMALL, tid);
    pg.issue(gainsI, GAIN_DOUBLES, tid);
    for (int e = tid; e < 3 * ZV; e += NTHR) lds[F_ZV + e] = 0.0_r;
    ph.commit(lds + R_STG, FWD_HEAD, tid);   // the forward sweep works on the record as it is: rows 12..29 of the head are Px / Pu (see the chain below)
    pt.commit(lds + R_STG + FWD_TAIL0, FWD_SMALL, tid);
    pg.commit(lds + R_STG + STAGE_DOUBLES, GAIN_DOUBLES, tid);
  }
  __syncthreads();
  if (tid < 30) lds[F_ZV + tid] = a.x0[size_t(inst) * 30 + tid] - a.X[size_t(inst) * (N + 1) * 30 + tid];
  real armijo = 0.0_r;
  __syncthreads();
  const bool upper = lane >= 32;            // second half of the wavefront: the second half of every dot product
  const int rowl = lane & 31;               // row of a matrix-vector product handled by this lane
  const int rK = rowl < MT ? rowl : 0, rX = rowl < 30 ? rowl : 0;
  // The blocks of stage k + 1 land in LDS during iteration k and were requested during iteration k - 2: an iteration (~0.8 us) is shorter than the HBM latency
  // under load.  Wavefronts 2 and 3 take turns -- wavefront 2 lands a WHOLE stage in the even iterations, wavefront 3 in the odd ones --, so each has one
  // register set in flight for two iterations and, at its commit, nothing of its own that is newer: two sets per wavefront were built first and the compiler's
  // wait-count insertion waited for the newer set at every commit (behind a branch, in straight-line pairs of iterations, with unconditional requests alike:
  // vmcnt(9) .. vmcnt(0) where vmcnt(19) .. vmcnt(10) would do; 0.516 -> 0.521-0.553 ms).  The wavefront that is not landing a stage stores dx and
  // accumulates the Armijo slope.  The loads in flight live in registers across the loop back-edge.
  StagePrefetch<PFH3, NPF> ph;
  StagePrefetch<PFT3, NPF> pt;
  StagePrefetch<PFG3, NPF> pg;
  auto requestStage = [&](int stage) {
    ph.issue(stagesI + size_t(stage) * STAGE_DOUBLES, FWD_HEAD, lane);
    pt.issue(stagesI + size_t(stage) * STAGE_DOUBLES + FWD_TAIL0, FWD_SMALL, lane);
    pg.issue(gainsI + size_t(stage) * GAIN_DOUBLES, GAIN_DOUBLES, lane);
  };
  if (wave == 2 && N > 1) requestStage(1);
  if (wave == 3 && N > 2) requestStage(2);
#pragma unroll 1
  for (int k = 0; k <= N; ++k) {   // iteration k: the chain does stage k (k < N), the others finish stage k - 1 and stage the blocks of k + 1
    const int sl = k % 3, slPrev = (k + 2) % 3, slNext = (k + 1) % 3;
    QM_TICK(14);
    if (wave == 0) {
      if (k < N) {
        const real* stg = lds + R_STG + sl * STG_F; const real* gn = stg + STAGE_DOUBLES;
        real* zv = lds + F_ZV + sl * ZV; real* zvNext = lds + F_ZV + slNext * ZV;
        // Every product is split in the middle between the two halves of the wavefront (row r of the product on lanes r and 32 + r): the same instruction
        // stream on both halves, so all LDS operands are requested before the first multiply-add, and the part of dx+ that has to wait for du~ is 9
        // multiply-adds, not 18.  The head of the record is used as it is (layout.h): rows 0..11 are [A~ | B~], rows 12..29 [Px | Pu], so with
        // s = row . [dx; du~]:   dx+_i = b~_i + s (i < 12),   dx+_i = b~_i + dx_i + dt s (joint rows: x_j+ = x_j + dt v_j exactly),
        // and s of a joint row is also what wavefront 1 needs for du = Pe + Px dx + Pu du~: it goes out next to dx+ -- nobody reads Px / Pu a second time.
        // (Odd row strides for the image the chain walks -- 31 / 19 doubles instead of the record's 30 / 18, against bank conflicts of one row per lane -- were
        //  built and measured in round 3: 1.10 k -> 1.05 k ticks for this phase, 0.493 -> 0.495 ms for the launch: not kept.)
        // du~ = K dx + k: columns 0..14 on the lower half, 15..29 on the upper half
        const real* Krow = gn + OFF_KFB + rK * 30 + (upper ? 15 : 0);
        const real* xh = zv + (upper ? 15 : 0);
        const real* Arow = stg + OFF_AT + rX * 30 + (upper ? 15 : 0);
        real kv[15], av[15], xv[15];
#pragma unroll
        for (int c = 0; c < 15; ++c) { kv[c] = Krow[c]; xv[c] = xh[c]; av[c] = Arow[c]; }
        real s0 = upper ? 0.0_r : gn[OFF_kff + rK], s1 = 0.0_r;
        const bool jointRow = rX >= 12;
        const real tail = upper ? stg[OFF_bt + rX] : (jointRow ? zv[rX] : 0.0_r);   // b~_i on one half, dx_i of a joint row on the other
        const real mul = jointRow ? stg[OFF_DT] : 1.0_r;
        const real* Brow = stg + OFF_BT + rX * MT + (upper ? 9 : 0);
        real bw[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) bw[j] = Brow[j];
#pragma unroll
        for (int c = 0; c < 14; c += 2) { s0 += kv[c] * xv[c]; s1 += kv[c + 1] * xv[c + 1]; }
        s0 += kv[14] * xv[14];
        const real sd = s0 + s1;
        const real du = sd + qmHalfXor32(sd, upper);
        QM_TICK(15);
        if (lane < MT) zv[30 + lane] = du;
        // the [A~; Px] dx half (needs no du~) while du~ makes its way through LDS
        real t0 = 0.0_r, t1 = 0.0_r;
#pragma unroll
        for (int c = 0; c < 14; c += 2) { t0 += av[c] * xv[c]; t1 += av[c + 1] * xv[c + 1]; }
        t0 += av[14] * xv[14];
        QM_WAVE_SYNC();
        QM_TICK(16);
        {
          const real* uh = zv + 30 + (upper ? 9 : 0);
#pragma unroll
          for (int j = 0; j < 8; j += 2) { t0 += bw[j] * uh[j]; t1 += bw[j + 1] * uh[j + 1]; }
          t0 += bw[8] * uh[8];
        }
        const real th = t0 + t1;
        const real sRow = th + qmHalfXor32(th, upper);            // s, complete on both halves
        const real nx = (tail + qmHalfXor32(tail, upper)) + mul * sRow;
        if (lane >= 12 && lane < 30) zv[48 + lane] = sRow;
        if (lane < 30) zvNext[lane] = nx;
      }
    } else {
      QM_TICK(15);
      const bool landing = wave >= 2 && ((k ^ wave) & 1) == 0;   // wavefront 2 in the even iterations, wavefront 3 in the odd ones
      if (landing) {
        if (k + 1 < N) {
          real* dst = lds + R_STG + slNext * STG_F;
          ph.commit(dst, FWD_HEAD, lane);
          pt.commit(dst + FWD_TAIL0, FWD_SMALL, lane);
          pg.commit(dst + STAGE_DOUBLES, GAIN_DOUBLES, lane);
        }
        if (k + 3 < N) requestStage(k + 3);
      }
      QM_TICK(16);
      if (k > 0) {
        const int j = k - 1;
        const real* stg = lds + R_STG + slPrev * STG_F;
        const real* zv = lds + F_ZV + slPrev * ZV;
        if (wave == 1) {   // du = Pe + Px dx + Pu du~: joint rows from the chain's s, force rows from the contact mode (layout.h: puColumnOfForce)
          if (lane < 30) {
            const int puCol = puColumnOfForce(int(stg[OFF_MODE]), lane < 12 ? lane : 0);
            const real rowPart = lane >= 12 ? zv[48 + lane] : (puCol >= 0 ? zv[30 + (puCol >= 0 ? puCol : 0)] : 0.0_r);
            a.dU[(size_t(inst) * N + j) * 30 + lane] = stg[OFF_PE + lane] + rowPart;
          }
        } else if (!landing) {   // dx out; Armijo slope q~ . dx + r~ . du~
          if (lane < 30) {
            const real dxl = zv[lane];
            a.dX[(size_t(inst) * (N + 1) + j) * 30 + lane] = dxl;
            armijo += stg[OFF_qt + lane] * dxl;
          } else if (lane >= 32 && lane < 32 + MT) {
            armijo += stg[OFF_rt + (lane - 32)] * zv[30 + lane - 32];     // rows >= m~ of du~ and of r~ are zero
          }
        }
      }
    }
    QM_TICK(17);
    QM_LDS_BARRIER();
    QM_TICK(18);
  }
  QM_TICK(13);
  QM_TICK_FLUSH((threadIdx.x >> 6) * 32, blockIdx.x == 0 && (threadIdx.x & 63) == 0);
  if (wave == 2 && lane < 30) {   // terminal node: dx_N sits in slot N % 3
    const real dxl = lds[F_ZV + (N % 3) * ZV + lane];
    a.dX[(size_t(inst) * (N + 1) + N) * 30 + lane] = dxl;
    armijo += stagesI[size_t(N) * STAGE_DOUBLES + OFF_qt + lane] * dxl;
  }
  // reduce armijo over the lanes of wavefronts 2 and 3 through LDS
  __syncthreads();
  if (wave == 2) red[lane] = armijo;
  __syncthreads();
  if (wave == 3) red[lane] += armijo;
  __syncthreads();
  if (tid == 0) {
    real s = 0.0_r;
    for (int i = 0; i < 64; ++i) s += red[i];
    a.instStats[size_t(inst) * 4 + 0] = s;
    a.instStats[size_t(inst) * 4 + 1] = real(status);
  }
}

}  // namespace qmk
