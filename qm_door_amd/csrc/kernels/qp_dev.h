// qp_dev.h -- one wavefront solves the QP of one HoQp level (qm_wbc/src/HoQp.cpp:60-150) with the slack block eliminated,
//   min 1/2 |AZ z + rhat|^2 + 1/2 sum_{i own} max(0, DZ_i z - f_i)^2     s.t.   DZ_i z <= f_i  (i inherited; f_i >= 0: z = 0 is feasible).
// Replaces the qpOASES call of HoQp.cpp:136-149 (an online active-set method, cold start, nWSR = 100) by a method of the same class:
//   phase 2, ALWAYS: a primal active-set method -- a working set, one row changes per iteration, the iterate stays feasible, the point returned satisfies the KKT
//     conditions on its working set: THE minimiser.  The minimiser on a working set is computed exactly by the range-space method on an augmented Hessian of the size
//     of the cost's own (no penalty factor: a weakly curved direction keeps its digits):
//        K = G + sum_P w_j d_j d_j' + sum_V d d'  = L L'      (matrix cores + in-register Cholesky; directions without curvature left out)
//        T = L^-1 DZ'  (all rows, one forward substitution per lane),  S = T_P'T_P  (pinned rows; small Cholesky in LDS; dependent rows skipped)
//        u = L^-1 (-grad - DZ_P' W r_P),   S mu = T_P'u + r_P,   p = L^-T (u - T_P mu)      =>  DZ_P (z + p) = f_P,  mu = the multipliers,  DZ p = T'(u - T_P mu)
//   phase 1, levels with inherited rows only: Mehrotra's interior point from z = 0, until the working set can be read off its iterate -- a starting point and a guess
//     for phase 2, nothing more (one active-set iteration then usually ends the level).
// Same algorithm, constants and decisions as the CPU restatement (the tests' checker; DESIGN.md section 4.7 has the reasoning).
//
// NP = n padded (8 / 20 / 36) sizes every register array and loop.  Lane roles: lane i < m0 owns inequality ROW i; lane c < NP owns COLUMN c (z_c, column c of K and
// row c of L); lane q < r owns task row q of AZ (residual).
#pragma once
#include "gpu_rt.h"

namespace qmk {

#if defined(QM_QP_TRACE) && !defined(QMGPU_HOST_EMULATION)
#define QP_TRACE_ON (blockIdx.x == (QM_QP_TRACE) && lane == 0)      // experiments only: device printf of one instance's active-set iterations
#else
#define QP_TRACE_ON (lane == 0)
#endif
constexpr double QP_EPS = 2.220446049250313e-16;
constexpr double QP_REG = 1e-12;             // HoQp's regulariser (HoQp.cpp:66): a direction it alone would carry counts as having no curvature (x10)
constexpr double QP_LAM_TOL = 8.0;           // = kAsLamTol of the CPU restatement
constexpr int QP_MAX_CHANGES = 100;          // nWSR of HoQp.cpp:141
constexpr int QP_HELD_CAP = 4;               // the held-variable form of the first level is given up beyond this many iterations (status 6; = kHeldFormMaxIterations of the CPU restatement): the interior point takes over
constexpr int QP_KMAX = 28;                  // pinned rows the small system holds (S: QP_KMAX x (QP_KMAX + 1) doubles of LDS)
constexpr int QP_SLD = QP_KMAX + 1;
constexpr double QP_STAGNATION_MU = 1e-10;

struct QpIo {
  const double* G;      // [36][ldk] (A Z)'(A Z), zero outside n x n (no regulariser)
  const double* AZ;     // [r][ldz] task rows in the level's variables
  const double* rhat;   // [r]
  const double* DZ;     // [56][ldz]; columns >= n zero; rows >= m0 finite
  const double* fhat;   // [56]
  double* Kt;           // [36][ldk] scratch: K tiles, then the rows of L, then the rows of T of the pinned set
  double* wtL;          // [64] scratch: row weights
  double* zs;           // [36] out: solution (lanes >= n write 0)
  double* red;          // exchange scratch (>= 1024 doubles; the host emulation uses all of it)
  double* fork;         // [1] command word of the fork-join with the three helper wavefronts (wbc_kernel): 0 = leave, NP = K tiles of that size
  double* S;            // [QP_KMAX][QP_SLD] scratch: the small system of the pinned rows
  double* Tp;           // [QP_KMAX][ldk] scratch: T_P = L^-1 DZ_P', one row per pinned row (slot order)
};

// K = G + DZ' diag(w) DZ: the upper-triangle 16 x 16 tiles t with t % 4 == wave (wave < 0: all of them) on the matrix cores, written
// (and mirrored) into the LDS square io.Kt.  Called by the solving wavefront and, between two workgroup barriers, by the three helper
// wavefronts of wbc_kernel: a v_mfma_f64 holds one SIMD's matrix pipe for 64 cycles, the six tiles of NP = 36 are 84 of them.
template <int NP, int LDZ_, int LDK_> __device__ __forceinline__ void ipmKTiles(const QpIo& io, int wave, int lane) {
  constexpr int TP = (NP + 15) / 16, KS = 14;
  const int l16 = lane & 15, h = lane >> 4;
  int t = 0;
#pragma unroll
  for (int ti = 0; ti < TP; ++ti)
#pragma unroll
    for (int tj = ti; tj < TP; ++tj, ++t) {
      if (wave >= 0 && (t & 3) != wave) continue;
      QmAcc acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ti * 16 + h + 4 * r, j = tj * 16 + l16;
        const double gv = io.G[(i < 36 ? i : 0) * LDK_ + (j < 36 ? j : 0)];
        acc[r] = (i < 36 && j < 36) ? gv : 0.0;
      }
#pragma unroll 1
      for (int k0 = 0; k0 < KS; k0 += 7) {   // the operands of seven k steps are read from LDS before the first matrix-core instruction
        double av[7], bv[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
          const int kk = 4 * (k0 + q) + h, ja = ti * 16 + l16, jb = tj * 16 + l16;
          const double w = io.wtL[kk];
          const double ra = io.DZ[kk * LDZ_ + (ja < 36 ? ja : 0)], rb = io.DZ[kk * LDZ_ + (jb < 36 ? jb : 0)];
          av[q] = ja < NP ? w * ra : 0.0; bv[q] = jb < NP ? rb : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) qmMfma(acc, av[q], bv[q], io.red);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ti * 16 + h + 4 * r, j = tj * 16 + l16;
        if (i < NP && j < NP) { io.Kt[i * LDK_ + j] = acc[r]; if (ti != tj) io.Kt[j * LDK_ + i] = acc[r]; }
      }
    }
}

// sum over 14 of the 56 rows of DZ[i][column of this lane] * bc[i] (bc: io.red[0..63], published by the solving wavefront): the share
// of wavefront `wave` of a 56-row column sum; the partial sums meet in io.red[128 + 64 wave + lane]
template <int LDZ_> __device__ __forceinline__ void ipmColSumShare(const QpIo& io, int wave, int lane) {
  const int colL = lane < 36 ? lane : 0, i0 = 14 * wave;
  const double* bc = io.red;
  double t[14], g[14];
#pragma unroll
  for (int q = 0; q < 14; ++q) { t[q] = io.DZ[(i0 + q) * LDZ_ + colL]; g[q] = bc[i0 + q]; }
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int q = 0; q < 14; q += 2) { a0 += t[q] * g[q]; a1 += t[q + 1] * g[q + 1]; }
  io.red[128 + 64 * wave + lane] = a0 + a1;
}
// the whole sum on four wavefronts (fork-join as for the K tiles), or on this one
template <int LDZ_> __device__ __forceinline__ double ipmColSum(const QpIo& io, int lane) {
  if (io.fork) {
    io.fork[0] = 200.0;
    QM_LDS_BARRIER();
    ipmColSumShare<LDZ_>(io, 0, lane);
    QM_LDS_BARRIER();
    return (io.red[128 + lane] + io.red[192 + lane]) + (io.red[256 + lane] + io.red[320 + lane]);
  }
  double s = 0.0;
  for (int w = 0; w < 4; ++w) { ipmColSumShare<LDZ_>(io, w, lane); QM_WAVE_SYNC(); s += io.red[128 + 64 * w + lane]; }
  return s;
}

// One step of the factorisation K = L L^T by row operations (lane c holds column c of K in kc), as a template recursion so that the DPP
// controls are immediates.  Multipliers L[r][J] = (scaled row J) at lane r: rows J + 1 (on the pivot chain) and J + 2 take them by
// v_readlane; rows J + 3 .. NP - 1 by DPP row_newbcast -- ONE v_fmac_f64_dpp per row update -- from copies of the row's lanes
// 16 g .. 16 g + 15 replicated into the four rows of 16 lanes, applied one step late so that the replication (LDS crossbar) is off the
// pivot chain.  Row updates of one row commute.
// A pivot that does not stand clear of its own rounding -- it is a difference, K_jj - sum_k L_jk^2, rounded relative to K_jj: floorAbs + floorRel (j + 1) K_jj -- marks
// a direction without curvature: its column of L becomes the unit vector (its row is cleared after the recursion), its right-hand side entry zero (exMask).
template <int J, int R, int NP> struct IpmDppRows {
  static __device__ __forceinline__ void run(double* kc, const double* bcP, double ncP, double* red) {
    if constexpr (R < NP) { qmFmacRowBcast<R % 16, R == J + 3 || R % 16 == 0>(kc[R], bcP[R / 16], ncP, red); IpmDppRows<J, R + 1, NP>::run(kc, bcP, ncP, red); }
  }
};
template <int NP, int J> struct IpmFactorStep {
  static constexpr int NG = (NP + 15) / 16;
  static __device__ __forceinline__ void run(double* kc, double& myInv, double* bcP, double& ncP, double diag0, double floorAbs, double floorRel, unsigned long long forced, unsigned long long& exMask, int lane, double* red) {
    if constexpr (J < NP) {
      const double piv = qmReadLane(kc[J], J, red);
      const double d0 = qmReadLane(diag0, J, red);
      const bool ex = ((forced >> J) & 1ull) || !(piv > floorAbs + floorRel * double(J + 1) * d0);     // (wave uniform; NaN pivots count as excluded: the caller checks the result)
      if (ex) exMask |= 1ull << J;
      const double dfl = ex ? 1.0 : piv;
      const double inv = qmRsqrtPos(dfl);
      kc[J] = (lane == J) ? dfl * inv : (ex ? 0.0 : kc[J] * inv);
      if (lane == J) myInv = inv;                              // 1 / L_jj
      const QmGather gk = qmGather(kc[J], red);                // L[r][j] = gk.get(r)
      if constexpr (J + 1 < NP) kc[J + 1] -= gk.get(J + 1) * kc[J];
      if constexpr (J + 2 < NP) kc[J + 2] -= gk.get(J + 2) * kc[J];
      if constexpr (J >= 1) IpmDppRows<J - 1, J + 2, NP>::run(kc, bcP, ncP, red);     // the previous step's rows J + 2 .. NP - 1
      if constexpr (J + 3 < NP) {
        if constexpr (NG > 0 && (J + 3) / 16 <= 0) bcP[0] = qmReplicateRow<0>(kc[J], red);
        if constexpr (NG > 1 && (J + 3) / 16 <= 1) bcP[1] = qmReplicateRow<1>(kc[J], red);
        if constexpr (NG > 2 && (J + 3) / 16 <= 2) bcP[2] = qmReplicateRow<2>(kc[J], red);
        ncP = -kc[J];
      }
      IpmFactorStep<NP, J + 1>::run(kc, myInv, bcP, ncP, diag0, floorAbs, floorRel, forced, exMask, lane, red);
    }
  }
};

struct QpOff { int G, AZ, rhat, DZ, fhat, Kt, wtL, zs, red, fork, S, Tp; };
struct QpResult { int status; int ipmIterations, iterations; bool strong; unsigned long long pinMask; bool warmRefuted; bool heldTried; };   // pinMask: the rows pinned at the solution (status 0); status: 0 ok | 1 working-set changes exhausted | 2 numerical failure | 3 final check failed | 4 more pinned rows than the small system holds | 5 the cost wants held variables moved | 6 held-variable form given up (QP_HELD_CAP); heldTried: variables were held

// A called function, not inlined: the kernel around it sits at 512 VGPRs with scratch, and three inlined instantiations of this body add
// ~1400 scalar-register spills to it; as a function each instantiation gets its own allocation.  The arrays arrive as offsets into the
// workgroup's dynamic LDS and are re-based on that symbol here, so that every access stays a ds_ instruction (pointers passed through a
// call are generic: the same body ran 20 % slower on flat loads).
// n: variables; r: task rows of AZ; m0: inequality rows; own: the rows are the level's own (soft) -- otherwise inherited (hard); rowOn: this lane's row takes part;
// sigma0: starting slacks / multipliers of the interior point in units of sqrt(scale) (<= 0: no interior point -- the level's own rows, and the tests' cold runs); tryHeld (a level with own rows whose bound is
// zero -- the friction rows of the first level, each acting on the contact forces only): the variables those rows act on are HELD at zero and the rows left out, instead
// of the rows being pinned: no working set to carry.  status 5 = the cost wants a held variable moved -- the caller solves again with the rows as rows.
// warm (bit 63 = valid, bits 0..55 = rows; inherited rows only): the working set the previous tick of this robot ended this level with (qmgpu_wbc_args::working_set), taken
// as the guess under the rules of the interior point's guess; a first step that any row cuts short refutes it and the level starts over the cold way.  warmZ (global memory,
// with bit 62 of warm; else null): the level's solution of that tick, the starting point -- scaled back by t <= 1 until every row outside the carried set holds (z = 0 is
// feasible, the rows are convex; the CPU restatement's solveLevel has the reasoning: the minimiser reached from z = 0 through the directions the cost sees is usually outside
// the rows, the previous tick's is a minimiser inside them up to the tick's change).
// lit: HoQp's 1e-12 I is kept LITERALLY -- on the diagonal of the factorised matrix and in the gradient, no absolute exclusion floor -- instead of in the limit: the small last
// levels (at most 12 variables, still the reference's own z), where a direction the task sees through a singular value of 1e-7 has a gradient that counts and a curvature below
// any floor (the CPU restatement's LevelQp::lit has the case and the numbers).
template <int NP, int LDZ_, int LDK_>
__device__ __attribute__((noinline)) QpResult qpSolve(QpOff off, int n, int r, int m0, bool own, bool rowOnIn, double sigma0, bool tryHeld, unsigned long long warm, const double* warmZ, bool lit, int lane) {
  QM_DYNAMIC_LDS(ldsBase);
  const QpIo io{ldsBase + off.G, ldsBase + off.AZ, ldsBase + off.rhat, ldsBase + off.DZ, ldsBase + off.fhat, ldsBase + off.Kt, ldsBase + off.wtL, ldsBase + off.zs, ldsBase + off.red, ldsBase + off.fork, ldsBase + off.S, ldsBase + off.Tp};
  enum { ST_I = 0, ST_P = 1, ST_V = 2 };
  const double* G = io.G; const double* DZ = io.DZ; const double* AZ = io.AZ; double* red = io.red;
  double* bc = io.red;              // [0..63] broadcast line (z, multipliers, u, v ...)
  double* ms = io.red + 64;         // [64..127] the small system's right-hand side / solution, by slot
  double* resL = io.red + 384;      // [384..447] task residual, by task row; [448..511] |.| version for the rounding bound
  auto allSum = [&](double v) { return qmAllSum(v, red); };
  auto allMax = [&](double v) { return qmAllMax(v, red); };
  auto allMin = [&](double v) { return qmAllMin(v, red); };
  QM_TICK_DECL;
  const int colL = lane < NP ? lane : 0;       // idle lanes alias column 0 / row 0 (results unused)
  const int rowL = lane < 56 ? lane : 0;
  const int tskL = lane < r ? lane : 0;
  const bool colOn = lane < n;
  // ---- what the two phases share (prepareLevel of the CPU restatement)
  const double hmax = allMax(colOn ? G[colL * LDK_ + colL] : 0.0);
  bool rowOn = rowOnIn;
  const double fl = rowOn ? io.fhat[rowL] : 0.0;
  double dn = 0.0, d2 = 0.0;
  {
#pragma unroll 1
    for (int j = 0; j < NP; j += 4) {
      const double t0 = DZ[rowL * LDZ_ + j], t1 = DZ[rowL * LDZ_ + j + 1], t2 = DZ[rowL * LDZ_ + j + 2], t3 = DZ[rowL * LDZ_ + j + 3];
      dn = fmax(fmax(dn, fmax(fabs(t0), fabs(t1))), fmax(fabs(t2), fabs(t3)));
      d2 += t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3;
    }
  }
  const double wA = rowOn ? fmax(1.0, hmax) / d2 : 0.0;      // augmentation weight of the row when pinned
  double gC;                                                   // g = AZ' rhat (for the scale)
  {
    double a0 = 0.0;
    for (int q = 0; q < r; ++q) a0 += AZ[q * LDZ_ + colL] * io.rhat[q];
    gC = colOn ? a0 : 0.0;
  }
  const double scale = fmax(1.0, allMax(fmax(rowOn ? fabs(fl) : 0.0, fabs(gC))));
  const double tol = 1e-9 * scale;
  unsigned long long heldMask = 0ull;
  if (tryHeld && own) {
    const bool cand = rowOn && fabs(fl) <= tol;
    for (int j = 0; j < n; ++j) if (qmBallot(cand && DZ[rowL * LDZ_ + j] != 0.0) != 0ull) heldMask |= 1ull << j;
    bool inside = cand;       // a zero-bound row that acts on held variables only is satisfied with them: left out
    for (int j = 0; j < n; ++j) inside = inside && (DZ[rowL * LDZ_ + j] == 0.0 || ((heldMask >> j) & 1ull));
    rowOn = rowOn && !inside;
  }
  const double floorAbs = lit ? 0.0 : 10.0 * QP_REG, floorRel = 16.0 * QP_EPS;
  double zc = 0.0;
  double kc[NP], uc[NP], myInv = 1.0;   // row c of L, row c of L^T
  unsigned long long exMask = 0ull;
#pragma unroll
  for (int q = 0; q < NP; ++q) { kc[q] = 0.0; uc[q] = 0.0; }

  // D z of this lane's row for the vector in bc
  auto rowDot = [&]() {
    double d0 = 0.0, d1 = 0.0;
#pragma unroll 1
    for (int j = 0; j < NP; j += 4) {   // eight LDS reads in flight, then the multiply-adds (a lone wavefront has nothing else to hide them)
      const double t0 = DZ[rowL * LDZ_ + j], t1 = DZ[rowL * LDZ_ + j + 1], t2 = DZ[rowL * LDZ_ + j + 2], t3 = DZ[rowL * LDZ_ + j + 3];
      const double z0 = bc[j], z1 = bc[j + 1], z2 = bc[j + 2], z3 = bc[j + 3];
      d0 += t0 * z0; d1 += t1 * z1; d0 += t2 * z2; d1 += t3 * z3;
    }
    return d0 + d1;
  };
  // gradient of the smooth cost at the vector in bc, in residual form AZ'(AZ z + rhat) (the CPU restatement's costGradient); absForm: the rounding bound
  // |AZ|'(|AZ| |z| + |rhat|) instead
  auto costGradient = [&](bool absForm) {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll 1
    for (int j = 0; j < NP; j += 2) {
      const double t0 = AZ[tskL * LDZ_ + j], t1 = AZ[tskL * LDZ_ + j + 1];
      const double z0 = bc[j], z1 = bc[j + 1];
      if (absForm) { a0 += fabs(t0) * fabs(z0); a1 += fabs(t1) * fabs(z1); } else { a0 += t0 * z0; a1 += t1 * z1; }
    }
    const double rh = io.rhat[tskL];
    QM_WAVE_SYNC();
    if (lane < 64) resL[lane] = lane < r ? (absForm ? (a0 + a1) + fabs(rh) : (a0 + a1) + rh) : 0.0;
    QM_WAVE_SYNC();
    double g0 = 0.0, g1 = 0.0;
#pragma unroll 1
    for (int q = 0; q + 1 < r; q += 2) {
      const double t0 = AZ[q * LDZ_ + colL], t1 = AZ[(q + 1) * LDZ_ + colL];
      if (absForm) { g0 += fabs(t0) * resL[q]; g1 += fabs(t1) * resL[q + 1]; } else { g0 += t0 * resL[q]; g1 += t1 * resL[q + 1]; }
    }
    if (r & 1) { const double t0 = AZ[(r - 1) * LDZ_ + colL]; g0 += (absForm ? fabs(t0) : t0) * resL[r - 1]; }
    return colOn ? (g0 + g1) + ((lit && !absForm) ? QP_REG * bc[colL] : 0.0) : 0.0;
  };
  // K = G + DZ' diag(wt) DZ = L L^T: tiles on the matrix cores, factorisation in registers, rows of L (and 1 / L_cc) to LDS, rows of L^T back
  auto factorise = [&](double wt) {
    QM_TICK(7);
    if (lane < 56) io.wtL[lane] = wt;
    QM_WAVE_SYNC();
    if (NP > 16 && io.fork) {   // several tiles: the helper wavefronts take theirs between two workgroup barriers
      io.fork[0] = double(NP);
      QM_LDS_BARRIER();
      ipmKTiles<NP, LDZ_, LDK_>(io, 0, lane);
      QM_LDS_BARRIER();
    } else {
      ipmKTiles<NP, LDZ_, LDK_>(io, -1, lane);
    }
    QM_WAVE_SYNC();
    QM_TICK(9);
    // lane c holds column c of K in kc; after step j, kc[j] of lane c is L^T[j][c] = L[c][j], i.e. lane c ends up with ROW c of L (entries r <= c)
    myInv = 1.0;
    // (all NP loads first, held in registers, then the selects: a load sunk into its select is a predicated LDS read with a wait of its own -- NP round trips in a row)
#pragma unroll
    for (int q = 0; q < NP; ++q) kc[q] = io.Kt[q * LDK_ + colL];   // K symmetric: column c = row c, read conflict free
#pragma unroll
    for (int q = 0; q < NP; ++q) QM_KEEP(kc[q]);
#pragma unroll
    for (int q = 0; q < NP; ++q) kc[q] = (colOn && q < n) ? kc[q] + ((lit && q == lane) ? QP_REG : 0.0) : ((q == lane) ? 1.0 : 0.0);   // identity padding beyond n; lit: HoQp's regulariser on the diagonal
    double diag0 = 0.0;
#pragma unroll
    for (int q = 0; q < NP; ++q) diag0 = (q == lane) ? kc[q] : diag0;
    exMask = 0ull;
    QM_TICK(10);
    {
      double bcP[3] = {0.0, 0.0, 0.0}, ncP = 0.0;
      IpmFactorStep<NP, 0>::run(kc, myInv, bcP, ncP, diag0, floorAbs, floorRel, heldMask, exMask, lane, red);
    }
    QM_TICK(11);
    const bool myEx = lane < NP && ((exMask >> lane) & 1ull);
#pragma unroll
    for (int q = 0; q < NP; ++q) kc[q] = myEx ? ((q == lane) ? 1.0 : 0.0) : kc[q];     // an excluded direction: its row of L is the unit vector as well
    // the back substitution L^T x = t walks the COLUMNS of L^T: U[r][c] (c > r) sits in lane c, register r.  One transpose through LDS per factorisation puts it into
    // lane r, register c -- and leaves the rows of L in LDS for the forward substitutions of the inequality rows (T, below)
    QM_WAVE_SYNC();
    if (lane < NP) {
#pragma unroll
      for (int q = 0; q < NP; ++q) io.Kt[lane * LDK_ + q] = kc[q];
    }
    QM_WAVE_SYNC();
#pragma unroll
    for (int cc = 0; cc < NP; ++cc) uc[cc] = io.Kt[cc * LDK_ + colL];   // U[lane][cc] for cc > lane
    QM_WAVE_SYNC();
    QM_TICK(2);
  };
  // L t = rhs (forward substitution; lane c owns row c of L), then L^T x = t (back substitution; lane r owns row r of L^T in uc); excluded directions: zero
  auto forward = [&](double acc) {
    acc = (lane < NP && ((exMask >> lane) & 1ull)) ? 0.0 : acc;
    double tC = 0.0;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const double tr = qmReadLane(acc * myInv, q, red);
      if (lane == q) tC = tr;
      acc -= (lane > q) ? kc[q] * tr : 0.0;
    }
    return tC;
  };
  auto backward = [&](double tC) {
    double bacc = (lane < NP && ((exMask >> lane) & 1ull)) ? 0.0 : tC, x = 0.0;
#pragma unroll
    for (int cc = NP - 1; cc >= 0; --cc) {
      const double dc = qmReadLane(bacc * myInv, cc, red);
      if (lane == cc) x = dc;
      bacc -= (lane < cc) ? uc[cc] * dc : 0.0;
    }
    return colOn ? x : 0.0;
  };

  QM_TICK(0);
  // ================================================================== phase 1: interior point (inherited rows only): a starting point and a guess
  int ipmIt = 0;
  bool usable = false;
  double s1 = 1.0, l1 = 0.0;
  const double nRows = allSum(rowOn ? 1.0 : 0.0);
  // (round 6) own rows take part in the interior point when the caller asks for it (sigma0 > 0: the level's held-variable form was rejected -- limits are violated wherever the
  //  task is met, and from z = 0 the active-set method needed 40-46 changes on the diverged robots of the bench's steady-state leg): as what they are in the reference's QP,
  //  D z - v <= f with 1/2 v'v in the cost; v = lam at the optimum, the row reads D z + s - lam = f, its weight in the normal equations is lam / (s + lam).
  const bool ipmOn = sigma0 > 0.0 && nRows > 0.0;
  if (ipmOn) {
    const double sigma = sigma0 * sqrt(scale);      // (start in units of sqrt(scale): the CPU restatement has the numbers)
    if (!own) { s1 = rowOn ? fmax(sigma, fl) : 1.0; l1 = rowOn ? sigma : 0.0; }
    else if (fl >= 0.0) { s1 = rowOn ? fl + sigma : 1.0; l1 = rowOn ? sigma : 0.0; }      // s - lam = f at z = 0: the row's equation holds from the start
    else { s1 = rowOn ? sigma : 1.0; l1 = rowOn ? sigma - fl : 0.0; }
  }
  double muTarget = 1e-8;            // duality measure (x scale) at which the working set is read off the iterate (= kIpmHandOverMu of the CPU restatement)
  int resumed = 0, status = 0, it = 0;
  bool strong = false;
  bool warmTry = !own && (warm >> 63) != 0ull, warmRefuted = false;
  unsigned long long pinOut = 0ull;
  if (warmTry && warmZ != nullptr && ((warm >> 62) & 1ull)) {
    const double z0 = colOn ? warmZ[colL] : 0.0;
    QM_WAVE_SYNC();
    bc[lane] = z0;
    QM_WAVE_SYNC();
    const double Dz0 = rowDot();
    const bool carried = lane < 56 && ((warm >> lane) & 1ull);
    const double t = allMin((rowOn && !carried && Dz0 > fl) ? fl / Dz0 : 1.0);
    const double probe = allSum(z0) + t;
    zc = (probe == probe) ? z0 * t : 0.0;
  }
  // The interior point hands over; if the step that is to bring its guessed rows onto their bounds is cut short by another row, the guess is wrong -- nothing has moved
  // yet, the interior point goes on from its iterate (target x 1e-2) and the working set is read again, at most twice; after that the step is taken as far as it goes.
#pragma unroll 1
  for (;;) {
  if (ipmOn && !warmTry) {
    const int itStart = ipmIt;
    double zcPrev = zc, s1p = s1, l1p = l1, nrdPrev = 0.0, muPrev = 0.0;
    usable = false;
#pragma unroll 1
    for (; ipmIt < 40; ++ipmIt) {
      QM_WAVE_SYNC();
      bc[lane] = zc;
      QM_WAVE_SYNC();
      const double Dz = rowDot();
      const double rp1 = rowOn ? (own ? (Dz + s1 - fl) - l1 : (Dz + s1 - fl)) : 0.0;
      double rdz;
      {
        double a0 = gC, a1 = 0.0;
#pragma unroll 1
        for (int j = 0; j < NP; j += 4) {   // G symmetric
          const double t0 = G[j * LDK_ + colL], t1 = G[(j + 1) * LDK_ + colL], t2 = G[(j + 2) * LDK_ + colL], t3 = G[(j + 3) * LDK_ + colL];
          const double z0 = bc[j], z1 = bc[j + 1], z2 = bc[j + 2], z3 = bc[j + 3];
          a0 += t0 * z0; a1 += t1 * z1; a0 += t2 * z2; a1 += t3 * z3;
        }
        QM_WAVE_SYNC();
        bc[lane] = rowOn ? l1 : 0.0;
        QM_WAVE_SYNC();
        const double dtl = ipmColSum<LDZ_>(io, lane);     // D^T lambda: 56 rows, shared with the helper wavefronts
        rdz = colOn ? ((a0 + a1) + (lit ? QP_REG * zc : 0.0)) + dtl : 0.0;
      }
      const double mu = allSum(rowOn ? s1 * l1 : 0.0) / nRows;
      const double nrd = allMax(fabs(rdz));
      const double nrp = allMax(fabs(rp1));
      const double nanProbe = allSum(rdz + rp1);  // NaN anywhere -> NaN here (fmax drops NaNs)
      // a late Newton step of a degenerate problem (barrier weights ~1e18) can lose all accuracy: the previous iterate is what the active-set method starts from
      if (ipmIt > itStart && (!(nanProbe == nanProbe) || !(mu == mu) || nrd > 100.0 * fmax(nrdPrev, 1e-9 * scale))) { zc = zcPrev; s1 = s1p; l1 = l1p; usable = true; break; }
#if defined(QMGPU_EMU_DEBUG) || defined(QM_QP_TRACE)
      if (QP_TRACE_ON) printf("EMU   ipm it %d n %d mu/s %.3e nrd/s %.3e nrp/s %.3e\n", ipmIt, n, mu / scale, nrd / scale, nrp / scale);
#endif
      if (nrd <= 1e-4 * scale && nrp <= 1e-9 * scale && mu <= muTarget * scale) { usable = true; break; }               // the working set can be read: over to the active-set method, for good
      if (ipmIt > itStart && mu > 0.5 * muPrev && mu <= QP_STAGNATION_MU * scale) { usable = true; break; }               // stagnation at the rounding floor
      zcPrev = zc; s1p = s1; l1p = l1; nrdPrev = nrd; muPrev = mu;
      const double den1 = own ? s1 + l1 : s1;
      const double w1 = l1 / den1;
      QM_TICK(12);
      factorise(rowOn ? w1 : 0.0);
      double ds1 = 0.0, dl1 = 0.0, dzc = 0.0, alphaAff = 1.0, sigma = 0.0, cw = 1.0;
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        const double rc1 = pass == 0 ? s1 * l1 : s1 * l1 + cw * ds1 * dl1 - sigma * mu;
        const double t1 = rowOn ? (l1 * rp1 - rc1) / den1 : 0.0;
        QM_WAVE_SYNC();
        bc[lane] = t1;
        QM_WAVE_SYNC();
        const double dtt = ipmColSum<LDZ_>(io, lane);   // D^T t
        QM_TICK(13);
        dzc = backward(forward(colOn ? -rdz - dtt : 0.0));
        QM_TICK(14);
        QM_WAVE_SYNC();
        bc[lane] = dzc;
        QM_WAVE_SYNC();
        const double Ddz = rowDot();
        QM_TICK(15);
        if (rowOn) {
          if (own) { dl1 = (l1 * (Ddz + rp1) - rc1) / den1; ds1 = (-rp1 - Ddz) + dl1; }
          else { ds1 = -rp1 - Ddz; dl1 = (-rc1 - l1 * ds1) / s1; }
        }
        double amax = 1.0;
        if (rowOn) { if (ds1 < 0) amax = fmin(amax, -s1 / ds1); if (dl1 < 0) amax = fmin(amax, -l1 / dl1); }
        amax = allMin(amax);
        if (pass == 0) {
          alphaAff = amax;
          const double muAff = allSum(rowOn ? (s1 + alphaAff * ds1) * (l1 + alphaAff * dl1) : 0.0) / nRows;
          const double ratio = muAff / mu;
          sigma = ratio * ratio * ratio;
          cw = fmin(1.0, 4.0 * alphaAff);
        } else {
          const double tau = fmax(0.995, 1.0 - mu);
          const double al = fmin(1.0, tau * amax);
          zc += al * dzc;
          if (rowOn) { s1 += al * ds1; l1 += al * dl1; }
        }
      }
    }
    QM_TICK(16);
    usable = true;                       // every way out of the loop hands an iterate over (the iteration cap too: the last iterate, like any other)
    if (!(allSum(zc) == allSum(zc))) { usable = false; zc = 0.0; }
  }

  QM_TICK(1);
  // ================================================================== phase 2: primal active set
  int state = ST_I;
  bool guess = false, stuck = false;
  if (own && rowOn) state = fl < -tol ? ST_V : (fl <= tol ? ST_P : ST_I);
  if (usable) {
    QM_WAVE_SYNC();
    bc[lane] = zc;
    QM_WAVE_SYNC();
    const double Dz = rowDot();
    if (!own) { if (rowOn && (l1 > s1 || Dz - fl > 0.0)) { state = ST_P; guess = true; } }
    else if (rowOn) { const double rr = Dz - fl; state = rr > tol ? ST_V : (rr >= -tol ? ST_P : ST_I); }      // own rows: on the side of their bound the iterate has them on (no guess: a violated row is a penalty wherever it stands)
  }
  if (warmTry && rowOn && lane < 56 && ((warm >> lane) & 1ull)) { state = ST_P; guess = true; }
  double lam = 0.0;                  // multiplier of this lane's row (pinned rows)
  int lastReleased = -1, fullSteps = 0, changes = 0, guard = 0;
  bool refuted = false;
  status = 0; it = 0; strong = false;
#pragma unroll 1
  for (;; ++it) {
    if (tryHeld && heldMask != 0ull && it > QP_HELD_CAP) { status = 6; break; }      // (the held-variable form given up: the caller runs the interior point)
    if (it > QP_MAX_CHANGES || ++guard > 4 * QP_MAX_CHANGES) { status = 1; break; }      // (guard: every trip of the loop counts, also those that do not change the working set)
    bool pinned = rowOn && state == ST_P;
    unsigned long long pinMask = qmBallot(pinned);
    int k = qmPopCount(pinMask);
    // this lane's place among the pinned rows, those already on their bounds first: a dependency then shows on a row of the guess, never on a row the ratio test pinned
    const unsigned long long tightMask = qmBallot(pinned && !guess), below = (1ull << lane) - 1ull;
    int slot = qmPopCount(tightMask & below);
    if (pinMask != tightMask) {
      // the guessed rows by decreasing multiplier estimate of the interior point, lam |d| (ties: smaller index): of two guessed rows that depend on each other -- the two
      // sides of a friction pyramid at its apex -- the one with the smaller estimate then shows the vanishing pivot and leaves (the CPU restatement has the numbers)
      const double key = warmTry ? 0.0 : l1 * dn;      // (a carried working set has no estimates: index order)
      int rank = 0;
      unsigned long long gm = pinMask & ~tightMask;
#pragma unroll 1
      while (gm != 0ull) {
        const int b = qmFirstBit(gm); gm &= gm - 1ull;
        const double kb = qmReadLane(key, b, red);
        rank += (kb > key || (kb == key && b < lane)) ? 1 : 0;
      }
      if (pinned && guess) slot = qmPopCount(tightMask) + rank;
    }
    // the small system holds QP_KMAX rows (= kAsMaxPinned of the CPU restatement): of a guess that is larger -- the interior point of a degenerate level, dozens of zero-margin
    // rows with multiplier above slack -- the rows with the smallest estimates stay out; the ratio test meets them again if the step crosses them
    if (k > QP_KMAX) {
      if (pinned && guess && slot >= QP_KMAX) { state = ST_I; guess = false; }
      pinned = rowOn && state == ST_P; pinMask = qmBallot(pinned); k = qmPopCount(pinMask);
    }
    if (k > QP_KMAX) { status = 4; break; }
    factorise(rowOn ? (state == ST_P ? wA : (state == ST_V ? 1.0 : 0.0)) : 0.0);
    if (!(allSum(myInv) == allSum(myInv))) { status = 2; break; }
    QM_TICK(7);
    // ---- T_P = L^-1 DZ_P' by slot into LDS, S = T_P'T_P
    unsigned long long depMask = 0ull;
    if (k > 0) {
      // Round 6 (tools/wbc_tick_probe.py on the slowest ticks of the bench's steady-state leg: T, S and the small factorisation were 58 % of a 36-variable active-set iteration,
      // 110 k ticks of 190 k, as one substitution per pinned row through the register-resident factor, S in loops of single LDS round trips and a right-looking factorisation with
      // three barriers and a read-modify-write chain per step): ALL pinned rows at once, lane s = slot s.
      // T_P = L^-1 DZ_P': the rows of L and 1 / L_cc come from LDS as wave-uniform (broadcast) reads, the lane's own t from its row of Tp; a finished entry is read back by the
      // same lane only (LDS is in order within a wavefront).  Directions without curvature: their row of L is the unit vector, their right-hand side entry zero (exMask).
      QM_WAVE_SYNC();
      if (lane < NP) io.wtL[lane] = myInv;                       // (the row weights of the K tiles are no longer needed)
      int* rowOfSlot = reinterpret_cast<int*>(ms);               // [QP_KMAX] ints over the small system's right-hand side (rebuilt by every pass)
      if (pinned) rowOfSlot[slot] = lane;
      QM_WAVE_SYNC();
      const int sa = lane < k ? lane : 0;
      const int myRow = rowOfSlot[sa];
      QM_WAVE_SYNC();
      {
        // four rows of L at a time: the part of their dot products that lies left of the block shares the lane's loads of its own t (one load of t feeds four multiply-adds, four
        // independent accumulation chains), the 4 x 4 triangle on the diagonal is solved in registers
        double* Trow = io.Tp + sa * LDK_;
        const double* drow = DZ + myRow * LDZ_;
        static_assert(NP % 4 == 0, "blocks of four rows");
#pragma unroll 1
        for (int c0 = 0; c0 < NP; c0 += 4) {
          const double* L0 = io.Kt + c0 * LDK_; const double* L1 = L0 + LDK_; const double* L2 = L1 + LDK_; const double* L3 = L2 + LDK_;
          double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 1
          for (int q = 0; q < c0; q += 4) {      // (c0 is a multiple of four: twenty loads in flight in front of sixteen multiply-adds, same order of the sums as two columns per trip)
            const double t0 = Trow[q], t1 = Trow[q + 1], t2 = Trow[q + 2], t3 = Trow[q + 3];
            const double l00 = L0[q], l01 = L0[q + 1], l10 = L1[q], l11 = L1[q + 1], l20 = L2[q], l21 = L2[q + 1], l30 = L3[q], l31 = L3[q + 1];
            const double l02 = L0[q + 2], l03 = L0[q + 3], l12 = L1[q + 2], l13 = L1[q + 3], l22 = L2[q + 2], l23 = L2[q + 3], l32 = L3[q + 2], l33 = L3[q + 3];
            a0 += l00 * t0; a1 += l10 * t0; a2 += l20 * t0; a3 += l30 * t0;
            a0 += l01 * t1; a1 += l11 * t1; a2 += l21 * t1; a3 += l31 * t1;
            a0 += l02 * t2; a1 += l12 * t2; a2 += l22 * t2; a3 += l32 * t2;
            a0 += l03 * t3; a1 += l13 * t3; a2 += l23 * t3; a3 += l33 * t3;
          }
          const double l10 = L1[c0], l20 = L2[c0], l21 = L2[c0 + 1], l30 = L3[c0], l31 = L3[c0 + 1], l32 = L3[c0 + 2];
          const double i0 = io.wtL[c0], i1 = io.wtL[c0 + 1], i2 = io.wtL[c0 + 2], i3 = io.wtL[c0 + 3];
          const double r0 = drow[c0], r1 = drow[c0 + 1], r2 = drow[c0 + 2], r3 = drow[c0 + 3];
          const unsigned ex4 = unsigned(exMask >> c0) & 15u;
          const double d0 = (c0 < n && !(ex4 & 1u)) ? r0 : 0.0, d1 = (c0 + 1 < n && !(ex4 & 2u)) ? r1 : 0.0, d2 = (c0 + 2 < n && !(ex4 & 4u)) ? r2 : 0.0, d3 = (c0 + 3 < n && !(ex4 & 8u)) ? r3 : 0.0;
          const double t0 = (d0 - a0) * i0;
          const double t1 = ((d1 - a1) - l10 * t0) * i1;
          const double t2 = (((d2 - a2) - l20 * t0) - l21 * t1) * i2;
          const double t3 = ((((d3 - a3) - l30 * t0) - l31 * t1) - l32 * t2) * i3;
          if (lane < k) { Trow[c0] = t0; Trow[c0 + 1] = t1; Trow[c0 + 2] = t2; Trow[c0 + 3] = t3; }
        }
      }
      QM_WAVE_SYNC();
      QM_TICK(3);
      {   // S = T_P T_P': lane a its row; its own t in registers, the other row as broadcast reads
        double ta[NP];
#pragma unroll
        for (int c = 0; c < NP; ++c) ta[c] = io.Tp[sa * LDK_ + c];
#pragma unroll 1
        for (int sb = 0; sb < k; ++sb) {
          const double* Tb = io.Tp + sb * LDK_;
          double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
          for (int c = 0; c + 3 < NP; c += 4) { a0 += ta[c] * Tb[c]; a1 += ta[c + 1] * Tb[c + 1]; a2 += ta[c + 2] * Tb[c + 2]; a3 += ta[c + 3] * Tb[c + 3]; }
          if (lane < k) io.S[lane * QP_SLD + sb] = (a0 + a1) + (a2 + a3);
        }
        static_assert(NP % 4 == 0, "S sums four entries at a time");
      }
      QM_WAVE_SYNC();
      // small Cholesky, left-looking by columns: at step j lane i >= j forms S_ij - sum_{q < j} L_iq L_jq (its own row and row j of the factor, finished columns only), the
      // pivot is lane j's value; a pivot lost against the row's own diagonal entry marks a dependent row (its row / column of the factor cleared: skipped by the solves).
      // One barrier per step, no read-modify-write of the trailing matrix.
      const double sdiag = io.S[sa * QP_SLD + sa];
#pragma unroll 1
      for (int j = 0; j < k; ++j) {
        const double* Li = io.S + sa * QP_SLD;
        const double* Lj = io.S + j * QP_SLD;
        double a0 = 0.0, a1 = 0.0;
        int q = 0;
#pragma unroll 1
        for (; q + 3 < j; q += 4) {
          const double x0 = Li[q], x1 = Li[q + 1], x2 = Li[q + 2], x3 = Li[q + 3];
          const double y0 = Lj[q], y1 = Lj[q + 1], y2 = Lj[q + 2], y3 = Lj[q + 3];
          a0 += x0 * y0; a1 += x1 * y1; a0 += x2 * y2; a1 += x3 * y3;
        }
        for (; q < j; ++q) a0 += Li[q] * Lj[q];
        const double v = Li[j] - (a0 + a1);
        const double d = qmReadLane(v, j, red);
        const double sj = qmReadLane(sdiag, j, red);
        const bool dep = !(d > 1e-11 * sj);
        if (dep) depMask |= 1ull << j;
        const double dj = dep ? 1.0 : sqrt(d);
        QM_WAVE_SYNC();
        if (lane == j) io.S[j * QP_SLD + j] = dj;
        else if (lane > j && lane < k) io.S[lane * QP_SLD + j] = dep ? 0.0 : v / dj;
        if (dep && lane < j) io.S[j * QP_SLD + lane] = 0.0;
        QM_WAVE_SYNC();
      }
    }
    QM_TICK(4);
    // ---- the interior point's guess: its rows are still off their bounds; the first step is meant to bring them there and is only taken in full.  A guess with a
    //      dependent row, or whose step another row cuts short, is dropped
    const bool offBound = qmBallot(pinned && guess) != 0ull;
    const bool depGuess = qmBallot(pinned && guess && ((depMask >> slot) & 1ull)) != 0ull;     // (a dependency among rows already on their bounds is harmless: skipped by the solve)
    if (offBound && depGuess) {     // a guess with dependent rows: those leave first -- the ratio test meets them again if the step crosses them
#if defined(QMGPU_EMU_DEBUG) || defined(QM_QP_TRACE)
      if (QP_TRACE_ON) printf("EMU   AS it %d: dependent rows of the guess leave (k %d dep %llx)\n", it, k, depMask);
#endif
      if (pinned && guess && ((depMask >> slot) & 1ull)) { state = ST_I; guess = false; }
      fullSteps = 0; --it; continue;
    }
    // ---- passes on this working set
    bool rebuild = false, done = false;
#pragma unroll 1
    for (;;) {
      QM_WAVE_SYNC();
      bc[lane] = zc;
      QM_WAVE_SYNC();
      const double Dz = rowDot();
      const double rRow = Dz - fl;
      const double gradC = costGradient(false);
      QM_WAVE_SYNC();
      bc[lane] = rowOn ? (state == ST_P ? wA * rRow : (state == ST_V ? rRow : 0.0)) : 0.0;
      QM_WAVE_SYNC();
      const double dtt = ipmColSum<LDZ_>(io, lane);
      QM_TICK(5);
      const double uC = forward(colOn ? -(gradC + dtt) : 0.0);
      QM_WAVE_SYNC();
      bc[lane] = lane < NP ? uC : 0.0;
      QM_WAVE_SYNC();
      double muMine = 0.0;
      if (k > 0) {
        // right-hand side of the small system: T_P u + r_P -- lane s < k takes row s of T_P from LDS (u as broadcast reads), the pinned lanes add their residuals
        double mval;
        {
          const int sl2 = lane < k ? lane : 0;
          double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
          for (int c = 0; c + 3 < NP; c += 4) { a0 += io.Tp[sl2 * LDK_ + c] * bc[c]; a1 += io.Tp[sl2 * LDK_ + c + 1] * bc[c + 1]; a2 += io.Tp[sl2 * LDK_ + c + 2] * bc[c + 2]; a3 += io.Tp[sl2 * LDK_ + c + 3] * bc[c + 3]; }
          QM_WAVE_SYNC();
          if (lane < k) ms[lane] = (a0 + a1) + (a2 + a3);
          QM_WAVE_SYNC();
          if (pinned) ms[slot] += rRow;
          QM_WAVE_SYNC();
          mval = lane < k ? ms[lane] : 0.0;
        }
        // S mu = rhs through the small factor, the right-hand side in registers (lane = row): the value of step j travels by v_readlane, the lane's own entries of the factor
        // come from LDS four steps ahead -- no barrier and no LDS round trip on the chain (until round 6: two of each per step, 17 k ticks per pass at 24 pinned rows)
        {
          const int sl2 = lane < k ? lane : 0;
          const double dinv = 1.0 / io.S[sl2 * QP_SLD + sl2];
#pragma unroll 1
          for (int j0 = 0; j0 < k; j0 += 4) {
            double lj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) lj[u] = io.S[sl2 * QP_SLD + (j0 + u < k ? j0 + u : 0)];      // L[lane][j] (used for lane > j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int j = j0 + u;
              if (j < k) {
                const double xj = ((depMask >> j) & 1ull) ? 0.0 : qmReadLane(mval * dinv, j, red);
                mval = (lane == j) ? xj : ((lane > j && lane < k) ? mval - lj[u] * xj : mval);
              }
            }
          }
#pragma unroll 1
          for (int j0 = k - 1; j0 >= 0; j0 -= 4) {
            double lj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) lj[u] = io.S[(j0 - u >= 0 ? j0 - u : 0) * QP_SLD + sl2];      // L[j][lane] (used for lane < j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int j = j0 - u;
              if (j >= 0) {
                const double xj = ((depMask >> j) & 1ull) ? 0.0 : qmReadLane(mval * dinv, j, red);
                mval = (lane == j) ? xj : ((lane < j) ? mval - lj[u] * xj : mval);
              }
            }
          }
          QM_WAVE_SYNC();
          if (lane < k) ms[lane] = mval;
          QM_WAVE_SYNC();
        }
        muMine = pinned ? ms[slot] : 0.0;
      }
      double vC = uC;
      {
        double a0 = 0.0, a1 = 0.0;
        int sb = 0;
#pragma unroll 1
        for (; sb + 3 < k; sb += 4) {
          const double t0 = io.Tp[sb * LDK_ + colL], t1 = io.Tp[(sb + 1) * LDK_ + colL], t2 = io.Tp[(sb + 2) * LDK_ + colL], t3 = io.Tp[(sb + 3) * LDK_ + colL];
          const double m0 = ms[sb], m1 = ms[sb + 1], m2 = ms[sb + 2], m3 = ms[sb + 3];
          a0 += t0 * m0; a1 += t1 * m1; a0 += t2 * m2; a1 += t3 * m3;
        }
        for (; sb < k; ++sb) a0 += io.Tp[sb * LDK_ + colL] * ms[sb];
        vC -= a0 + a1;
      }
      vC = lane < NP ? vC : 0.0;
      const double pC = backward(vC);
      QM_WAVE_SYNC();
      bc[lane] = pC;
      QM_WAVE_SYNC();
      const double Dp = rowDot();
      QM_TICK(6);
#if defined(QMGPU_EMU_DEBUG) || defined(QM_QP_TRACE)
      { const double pm_ = allMax(fabs(pC)); if (QP_TRACE_ON) printf("EMU   AS it %d n %d k %d dep %llx ex %llx offBound %d fullSteps %d pmax %.3e\n", it, n, k, depMask, exMask, int(offBound), fullSteps, pm_); }
#endif
      const double nanProbe = allSum(pC);
      if (!(nanProbe == nanProbe)) { status = 2; done = true; break; }
      const double pmax = allMax(fabs(pC));
      const double zmax0 = fmax(1.0, allMax(fabs(zc)));
      // what the pinned rows let through: a row that is a combination of pinned rows shows a step component of that size and must not be taken for a blocking row
      const double leak = allMax(pinned ? fabs(Dp + rRow) / dn : 0.0);
      // first sign change along the step
      double a = 2.0;
      if (rowOn && state != ST_P) {
        const double epsP = fmax(1e-13 * fmax(1.0, pmax), 1e3 * leak) * dn;
        if (state == ST_I) { if (Dp > epsP) a = fmax(0.0, -rRow) / Dp; }
        else { if (Dp < -epsP) a = fmax(0.0, rRow) / -Dp; }
      }
      double amin = allMin(a);
      if (offBound && amin >= 1.0 - 1e-9) amin = 2.0;       // (a row the guessed step reaches at its very end is not in its way)
      if (fullSteps > 0 && pmax <= 1e-9 * zmax0) amin = 2.0;  // (a refinement correction at rounding size changes no row's side)
      if (amin < 1.0 && warmTry && changes == 0) { refuted = true; done = true; break; }                                 // the carried working set is refuted (also the empty one): the cold way
      if (amin < 1.0 && offBound && ipmOn && resumed < 2 && changes == 0) { refuted = true; done = true; break; }     // the guess is refuted before anything moved: back to the interior point
      if (amin < 1.0) {
        const int block = qmFirstBit(qmBallot(a == amin));          // ties keep the smallest row index
#if defined(QMGPU_EMU_DEBUG) || defined(QM_QP_TRACE)
        if (QP_TRACE_ON) printf("EMU     blocked by row %d at alpha %.3e\n", block, amin);
#endif
        const bool moved = amin * pmax > 1e-13 * zmax0;             // a step that does not move the point beyond its rounding counts as zero-length
        zc += amin * pC;
        if (moved) stuck = false; else if (lane == block && block == lastReleased) stuck = true;
        lastReleased = -1;
        if (lane == block) state = ST_P;
        ++changes;
        fullSteps = 0; rebuild = true;
        break;
      }
      zc += pC;
      guess = false;                      // a full step: every pinned row is on its bound now
      lam = muMine;
      // refinement: the same working set once more from the new point until the correction is rounding -- at most three full steps in a row
      const double zmax1 = fmax(1.0, allMax(fabs(zc)));
      // (only a step that moved the point by more than 1e-4 of its size: a smaller one -- from the interior point's iterate -- is exact up to a rounding that scales with it)
      if (pmax > 1e-13 * zmax1 && (pmax > 1e-4 * zmax1 || fullSteps > 0) && fullSteps < 3) { ++fullSteps; continue; }
      QM_TICK(7);
      // multipliers: one counts once lam |d| stands clear of the rounding of the gradient it balances
      QM_WAVE_SYNC();
      bc[lane] = zc;
      QM_WAVE_SYNC();
      const double bound = allMax(costGradient(true));
      const double gradNoise = QP_LAM_TOL * QP_EPS * fmax(bound, 1e-300);
      const double bad = (pinned && !stuck) ? (own ? fabs(lam) : -lam) * dn / gradNoise : 0.0;
      const double worst = allMax(bad);
      if (worst > 1.0) {
        const int rel = qmFirstBit(qmBallot(bad == worst));
#if defined(QMGPU_EMU_DEBUG) || defined(QM_QP_TRACE)
        if (QP_TRACE_ON) printf("EMU     release row %d (worst %.3e)\n", rel, worst);
#endif
        if (lane == rel) { state = (own && lam > 0.0) ? ST_V : ST_I; lam = 0.0; }
        lastReleased = rel; fullSteps = 0; rebuild = true; ++changes;
        break;
      }
      // the point satisfies the KKT conditions on its working set; the bounds themselves once more (partial steps accumulate rounding)
      QM_WAVE_SYNC();
      bc[lane] = zc;
      QM_WAVE_SYNC();
      const double Df = rowDot() - fl;
      const bool viol = rowOn && (!own || state != ST_V) && !(Df <= tol);
      if (qmBallot(viol) != 0ull) status = 3;
      if (heldMask != 0ull) {      // held variables: the cost must not want them moved
        QM_WAVE_SYNC();
        bc[lane] = zc;
        QM_WAVE_SYNC();
        const double gz = costGradient(false);
        QM_WAVE_SYNC();
        bc[lane] = rowOn ? (state == ST_P ? lam : (state == ST_V ? Df : 0.0)) : 0.0;
        QM_WAVE_SYNC();
        const double gd = ipmColSum<LDZ_>(io, lane);
        const bool wants = lane < n && ((heldMask >> lane) & 1ull) && fabs(gz + gd) > gradNoise;
        if (qmBallot(wants) != 0ull) status = 5;
      }
      strong = rowOn && ((state == ST_V && Df * dn > gradNoise) || (state == ST_P && lam * dn > gradNoise));       // (a violated own row's multiplier is its violation)
      pinOut = qmBallot(rowOn && state == ST_P);
      QM_TICK(8);
      done = true;
      break;
    }
    if (done) break;
    (void)rebuild;
  }
  if (!refuted) break;
  if (warmTry) { warmTry = false; warmRefuted = true; zc = 0.0; continue; }     // (the cold way: z = 0, the interior point starts where it always does)
  ++resumed; muTarget *= 1e-2;
  }
  QM_TICK(7);
  QM_TICK_FLUSH(NP == 36 ? 160 : (NP == 20 ? 256 : 288), blockIdx.x == 0 && lane == 0);
  if (status == 2) zc = 0.0;      // numerical failure: the level is skipped (x stays the higher priorities' solution) and flagged
  if (lane < 36) io.zs[lane] = colOn ? zc : 0.0;
  return QpResult{status, ipmIt, it, strong, (status == 0 && !own) ? (pinOut | (1ull << 63)) : 0ull, warmRefuted, heldMask != 0ull};
}

}  // namespace qmk
