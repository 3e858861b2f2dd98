// ipm_dev.h -- one wavefront solves   min 1/2 z'Gz + g'z (+ 1/2 |v|^2)   s.t.  DZ z (- v) <= fhat, (v >= 0)
// by Mehrotra's predictor-corrector interior point: the QP of one HoQp level (qm_wbc/src/HoQp.cpp:60-134) with the slack block
// eliminated analytically.  Replaces the qpOASES call of HoQp.cpp:136-149; same iteration and tolerances as the oracle's
// solveQpIpm.
//
// NP = n padded (8 / 20 / 36) sizes every register array and loop.  Lane roles: lane i < m0 owns inequality ROW i
// (slack, multiplier, residuals); lane c < NP owns COLUMN c (z_c, column c of K and of the identity during the factorisation).
//   K = G + DZ' diag(w) DZ    on the fp64 matrix cores (16x16 tiles, upper triangle, mirrored into LDS)
//   K = L L'                  row operations, one column per lane in registers, multipliers broadcast by v_readlane
//   L t = rhs, L' dz = t      substitutions with the pivots' results broadcast by v_readlane (backward stable: no explicit
//                             inverse -- the barrier weights reach 1e14 in the last iterations)
// Nothing on the dependent chain of an iteration goes through LDS except the K tiles and the DZ rows/columns themselves.
#pragma once
#include "gpu_rt.h"

namespace qmk {

// Stagnation exit of the interior point (complementarity no longer halves): only once mu <= this x scale.  Round 3 used 1e-6: with robots in motion
// 14-28 % of the level-1 problems took this exit after ONE slow iteration at mu ~ 1e-6 scale, an unconverged iterate whose active set cannot be read
// (polish rejected) and whose dual residual (<= 1e-7 scale, scale = |c|max ~ 6e4 with the x100 swing weight) leaves the weakly weighted task directions
// (singular values 0.02 .. 60 of A Z) off by O(1) -- the source of GPU / oracle torque deviations of 1e-5 .. 1e-1 (profiles/r04_notes.md section 1).
#ifndef QM_IPM_STAGNATION_MU
#define QM_IPM_STAGNATION_MU 1e-10
#endif
#ifndef QM_IPM_POLISH_ADD
#define QM_IPM_POLISH_ADD 0               // = kPolishAdd of the oracle
#endif
#ifndef QM_IPM_POLISH_CORRECTIONS
#define QM_IPM_POLISH_CORRECTIONS 4       // = kPolishCorrections of the oracle
#endif
// early polish attempts of a level with slack variables of its own (= kEarly*Own of the oracle)
#define QM_IPM_EARLY_TRIES_OWN 4
#define QM_IPM_EARLY_MU_OWN 1e-2
#define QM_IPM_EARLY_NRP_OWN 1e-2
#define QM_IPM_EARLY_NRD_OWN 1e-1
#define QM_IPM_EARLY_DROP_OWN 0.1
// A level with slack variables of its own (the first: equations of motion, torque limits, friction cones) first tries the active-set polish BEFORE any
// interior-point iteration with the guess "no limit binds" (= kZeroTryOwn of the CPU restatement's solveQpIpm): pinned are only the rows with a
// zero right-hand side (the friction rows of a swing leg, 0 <= 0), every slack variable is zero.  Away from the limits that IS the solution -- one
// factorisation instead of two interior-point iterations and a polish; a rejected try leaves the interior point's starting point untouched.
#ifndef QM_IPM_ZERO_TRY_OWN
#define QM_IPM_ZERO_TRY_OWN 1
#endif

struct IpmIo {
  const double* G;      // [36][ldk], zero outside n x n
  const double* g;      // [36]
  const double* DZ;     // [56][ldz]; columns >= n zero; rows >= m0 finite
  const double* fhat;   // [56]
  double* Kt;           // [36][ldk] scratch for the K tiles
  double* wtL;          // [64] scratch: row weights
  double* zs;           // [36] out: solution (lanes >= n write 0)
  double* red;          // wavefront exchange scratch of the host emulation (>= 1024 doubles)
  double* fork;         // [1] command word of the fork-join with the three helper wavefronts (wbc_kernel): 0 = leave, NP = K tiles of that size
};

// K = G + DZ' diag(w) DZ: the upper-triangle 16 x 16 tiles t with t % 4 == wave (wave < 0: all of them) on the matrix cores, written
// (and mirrored) into the LDS square io.Kt.  Called by the solving wavefront and, between two workgroup barriers, by the three helper
// wavefronts of wbc_kernel: a v_mfma_f64 holds one SIMD's matrix pipe for 64 cycles, the six tiles of NP = 36 are 84 of them.
template <int NP, int LDZ_, int LDK_> __device__ __forceinline__ void ipmKTiles(const IpmIo& io, int wave, int lane) {
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
template <int LDZ_> __device__ __forceinline__ void ipmColSumShare(const IpmIo& io, int wave, int lane) {
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
template <int LDZ_> __device__ __forceinline__ double ipmColSum(const IpmIo& io, int lane) {
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
template <int J, int R, int NP> struct IpmDppRows {
  static __device__ __forceinline__ void run(double* kc, const double* bcP, double ncP, double* red) {
    if constexpr (R < NP) { qmFmacRowBcast<R % 16, R == J + 3 || R % 16 == 0>(kc[R], bcP[R / 16], ncP, red); IpmDppRows<J, R + 1, NP>::run(kc, bcP, ncP, red); }
  }
};
template <int NP, int J> struct IpmFactorStep {
  static constexpr int NG = (NP + 15) / 16;
  static __device__ __forceinline__ void run(double* kc, double& myInv, double* bcP, double& ncP, double pivotFloor, int lane, double* red) {
    if constexpr (J < NP) {
      const double piv = qmReadLane(kc[J], J, red);
      const double dfl = piv > pivotFloor ? piv : pivotFloor;  // pivots floored as in the oracle's choleskyFloored
      const double inv = qmRsqrtPos(dfl);                      // dfl >= pivotFloor > 0
      kc[J] = (lane == J) ? dfl * inv : kc[J] * inv;
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
      IpmFactorStep<NP, J + 1>::run(kc, myInv, bcP, ncP, pivotFloor, lane, red);
    }
  }
};

// returns the iteration count (60 = not converged) and this lane's slack value
// A called function, not inlined: the kernel around it sits at 512 VGPRs with scratch, and three inlined instantiations of this body add
// ~1400 scalar-register spills to it; as a function each instantiation gets its own allocation.  The arrays arrive as offsets into the
// workgroup's dynamic LDS and are re-based on that symbol here, so that every access stays a ds_ instruction (pointers passed through a
// call are generic: the same body ran 20 % slower on flat loads).
struct IpmOff { int G, g, DZ, fhat, Kt, wtL, zs, red, fork; };
struct IpmResult { int iterations; double v; };
template <int NP, int LDZ_, int LDK_>
__device__ __attribute__((noinline)) IpmResult ipmSolve(IpmOff off, int n, int m0, bool own, bool rowActive, double sigma0, int lane) {
  QM_DYNAMIC_LDS(ldsBase);
  const IpmIo io{ldsBase + off.G, ldsBase + off.g, ldsBase + off.DZ, ldsBase + off.fhat, ldsBase + off.Kt, ldsBase + off.wtL, ldsBase + off.zs, ldsBase + off.red, ldsBase + off.fork};
  constexpr int TP = (NP + 15) / 16;           // 16-wide tiles per dimension
  constexpr int KS = 14;                       // k steps of 4 rows: 56 inequality rows
  const int l16 = lane & 15, h = lane >> 4;
  const double* G = io.G; const double* DZ = io.DZ; double* red = io.red;
  auto allSum = [&](double v) { return qmAllSum(v, red); };
  auto allMax = [&](double v) { return qmAllMax(v, red); };
  auto allMin = [&](double v) { return qmAllMin(v, red); };
  const int colL = lane < NP ? lane : 0;       // idle lanes alias column 0 / row 0 (results unused)
  const int rowL = lane < 56 ? lane : 0;
  const bool colOn = lane < n;
  const double gC = io.g[colL];
  const double pivotFloor = 1e-13 * allMax(colOn ? G[colL * LDK_ + colL] : 0.0);
  double fl = rowActive ? io.fhat[rowL] : 0.0;
  const double scale = fmax(1.0, allMax(fmax(rowActive ? fabs(fl) : 0.0, colOn ? fabs(gC) : 0.0)));
  const double nRowsTot = allSum(rowActive ? (own ? 2.0 : 1.0) : 0.0);
  double zc = 0.0, zcPrev = 0.0;
  // starting point: slacks max(sigma0, f), multipliers sigma0 -- 1 for the top level, 300 below it (the oracle's kLowerLevelStart: a unit start
  // spends up to fifteen iterations of the lower levels on tiny steps)
  double v = 0.0, s1 = rowActive ? fmax(sigma0, fl) : 1.0, l1 = sigma0, s2 = sigma0, l2 = sigma0;  // own rows: (s1,l1) constraint, (s2,l2) v >= 0
  double s1p = s1, l1p = l1, s2p = s2, l2p = l2, vp = v, nrdPrev = 0.0, muPrev = 0.0;
  // Active-set polish (same as the oracle's solveQpIpm): once the interior point has stopped, the active set is read off the final
  // iterate and three augmented-Lagrangian Newton steps on the equality-constrained QP run through the SAME loop body (K tiles,
  // factorisation, substitutions) with the barrier weights replaced by {rho: row pinned, 1: violated soft row, 0: inactive}.
  // Second attempt (as the oracle's HoQp): a degenerate low-priority level (more inherited rows active than free directions: no
  // interior) stalls the interior point.  Restart (twice at most) with every inherited row relaxed to a margin of at least 1e-5, then
  // 1e-3, and slacks / multipliers of O(sqrt(scale)).  If that fails too the level is skipped (z = 0: x stays the higher priorities' solution) and the
  // failure is reported (return value 60).
  int attempt = 0;
  bool early = false; int earlyTries = 0; double lastTryMu = 1e300;   // early polish attempt in flight / attempts so far / complementarity at the last one
  auto restartOrGiveUp = [&]() {
    if (attempt < 2) {
      ++attempt;
      const double sg = sqrt(scale);
      if (rowActive && !own) fl = fmax(fl, attempt == 1 ? 1e-5 : 1e-3);   // second attempt 1e-5, third 1e-3 (still < 1e-4 of the limits)
      zc = 0.0; zcPrev = 0.0; v = 0.0; vp = 0.0;
      s1 = rowActive ? fmax(sg, fl) : 1.0; l1 = sg; s2 = sg; l2 = sg;
      s1p = s1; l1p = l1; s2p = s2; l2p = l2; nrdPrev = 0.0; muPrev = 0.0;
      early = false; earlyTries = 0; lastTryMu = 1e300;
      return true;
    }
    zc = 0.0; v = 0.0;
    return false;
  };
  int polish = 0;                       // 0 interior point; 1..3 polish step; 4 final check
  int corrections = 0;                  // releases + additions of this polish attempt so far
  // the polish is first tried as soon as the active set can plausibly be read (mu <= 1e-6 scale, at most twice): an accepted
                                            // vertex is exact whatever iterate it started from, a rejected one resumes the interior point
  bool isE = false, isV = false;        // my row: pinned (equality) / violated soft row of this level (exact quadratic)
  double lamE = 0.0, zIpm = 0.0;
  const double rho = 1e6 * fmax(1.0, pivotFloor * 1e13);
  int it = 0, itOut = 0;
  bool zeroTry = false, zeroPinned = false;
  if (QM_IPM_ZERO_TRY_OWN && own) {
    zeroTry = true; early = true;
    isE = rowActive && fl <= 1e-9 * scale; isV = false; lamE = 0.0; zIpm = zc; polish = 1; corrections = 0;
    zeroPinned = allMax(isE ? 1.0 : 0.0) > 0.0;   // nothing pinned: the first Newton step is the minimiser of the quadratic, the check follows at once
  }
  double kc[NP], uc[NP], myInv = 1.0;   // factor of the current K (row c of L, row c of L^T): survives across the polish steps, whose K is constant
#pragma unroll
  for (int r = 0; r < NP; ++r) { kc[r] = 0.0; uc[r] = 0.0; }
  QM_TICK_DECL;
#pragma unroll 1
  for (; it < 70; ++it) {
    QM_TICK(0);
    // ---- residuals
    // vectors every lane needs element by element (z, the multipliers, the right-hand sides) go through one LDS line (io.wtL) and come
    // back as broadcast reads, two numbers per ds_read_b128: a v_readlane pair + wait state per element costs three issue slots more
    double* bc = io.red;     // (the exchange scratch of the host emulation: free on both builds; 64 entries used)
    QM_WAVE_SYNC();
    bc[lane] = zc;
    QM_WAVE_SYNC();
    double Dz;
    {
      double d0 = 0.0, d1 = 0.0;
#pragma unroll 1
      for (int j = 0; j < NP; j += 4) {   // eight LDS reads in flight, then the multiply-adds (a lone wavefront has nothing else to hide them)
        const double t0 = DZ[rowL * LDZ_ + j], t1 = DZ[rowL * LDZ_ + j + 1], t2 = DZ[rowL * LDZ_ + j + 2], t3 = DZ[rowL * LDZ_ + j + 3];
        const double z0 = bc[j], z1 = bc[j + 1], z2 = bc[j + 2], z3 = bc[j + 3];
        d0 += t0 * z0; d1 += t1 * z1; d0 += t2 * z2; d1 += t3 * z3;
      }
      Dz = d0 + d1;
    }
    const double rp1 = rowActive ? (Dz - (own ? v : 0.0) + s1 - fl) : 0.0;
    const double rp2 = (rowActive && own) ? (-v + s2) : 0.0;
    const double rdv = (rowActive && own) ? (v - l1 - l2) : 0.0;
    const double rRow = Dz - fl;
    if (polish > 1 && isE) lamE += rho * rRow;                       // multiplier update of the previous polish step
    if (zeroTry && polish == 2 && !zeroPinned) polish = 4;
    if (polish == 4) {                                               // keep the polished point only if it is a valid vertex
      bool bad = false;
      if (rowActive) {
        if (isE) bad = !(lamE >= -1e-9 * scale) || !(fabs(rRow) <= 1e-9 * scale) || (zeroTry && !(lamE <= 1e-9 * scale));   // (zero try: the row's slack variable is pinned at 0 as well, its multiplier is -lamE)
        else if (isV) bad = !(rRow >= -1e-9 * scale);
        else bad = !(rRow <= 1e-9 * scale);
      }
      const bool rejected = allMax((bad || !(zc == zc)) ? 1.0 : 0.0) > 0.0;
      if (rejected) zc = zIpm;
      if (rejected && zeroTry) { zeroTry = false; early = false; polish = 0; it = -1; continue; }   // the interior point starts as if nothing had happened
      if (rejected && early) { early = false; polish = 0; continue; }   // back to the interior point (its slacks / multipliers were not touched)
      break;
    }
    if (polish == 2 && !own) {
      // active-set correction loop (the oracle's solveQpIpm, activeSetCorrection): the estimates after the first step of an attempt already tell
      // whether the active set was read correctly.  Negative multipliers at a feasible point: release those rows; rows outside the guess that the step
      // violates (weakly active rows whose slack and multiplier both vanish -- the interior point cannot classify them): add them; either way the
      // polish starts again from the interior-point iterate, at most QM_IPM_POLISH_CORRECTIONS times per attempt.  Still wrong: abandon the attempt
      // now, not after two more steps and the check
      const double viol = allMax(rowActive ? rRow : -1e300);
      const double lmin = allMin((rowActive && isE) ? lamE : 0.0);
      const bool canFix = corrections < QM_IPM_POLISH_CORRECTIONS;
      const bool release = canFix && lmin < -1e-9 * scale && viol <= 1e-6 * scale;
      const bool add = QM_IPM_POLISH_ADD && canFix && !early && !release && viol > 1e-8 * scale && viol <= 0.1 * scale;   // (only once the interior point has converged: an early guess at mu ~ 1e-7 scale that needs rows added is abandoned instead)
      if (release || add || !(viol <= 1e-8 * scale) || !(lmin >= -1e-8 * scale)) {
        zc = zIpm;
        if (release) { ++corrections; isE = isE && !(lamE < 0.0); lamE = isE ? l1 : 0.0; polish = 1; continue; }
        if (add) { ++corrections; isE = isE || (rowActive && rRow > 1e-9 * scale); lamE = isE ? l1 : 0.0; polish = 1; continue; }
        if (early) { early = false; polish = 0; continue; }
        break;
      }
    }
    const double lamR = polish ? (isE ? lamE + rho * rRow : (isV ? rRow : 0.0)) : (rowActive ? l1 : 0.0);
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
      bc[lane] = lamR;
      QM_WAVE_SYNC();
      const double dtl = ipmColSum<LDZ_>(io, lane);     // D^T lambda: 56 rows, shared with the helper wavefronts
      rdz = colOn ? (a0 + a1) + dtl : 0.0;
    }
    QM_TICK(1);
    const double mu = allSum(rowActive ? (s1 * l1 + (own ? s2 * l2 : 0.0)) : 0.0) / nRowsTot;
    const double nrd = allMax(fmax(fabs(rdz), fabs(rdv)));
    const double nrp = allMax(fmax(fabs(rp1), fabs(rp2)));
    const double nanProbe = allSum(rdz + rdv + rp1 + rp2);  // NaN anywhere -> NaN here (fmax drops NaNs)
    // A late Newton step of a degenerate problem can lose all accuracy (barrier weights ~1e18).  As in the oracle's
    // solveQpIpm: a step that blows the dual residual up or yields NaN is rejected and the previous iterate returned --
    // as converged if its complementarity was already <= 1e-8 * scale, flagged (it = 60) otherwise.
#ifdef QMGPU_EMU_DEBUG
    if (lane == 0) printf("EMU ipm NP %d n %d it %d attempt %d polish %d mu/s %.3e nrd/s %.3e nrp/s %.3e scale %.3e\n", NP, n, it, attempt, polish, mu / scale, nrd / scale, nrp / scale, scale);
#endif
    if (!polish) {
      bool done = false;
      if (it > 0 && (!(nanProbe == nanProbe) || !(mu == mu) || nrd > 100.0 * fmax(nrdPrev, 1e-9 * scale))) {
        zc = zcPrev; s1 = s1p; l1 = l1p; s2 = s2p; l2 = l2p; v = vp;
        if (!(muPrev <= 1e-8 * scale)) { if (restartOrGiveUp()) { it = -1; continue; } itOut = 60; break; }
        done = true;
      } else if (nrd <= 1e-7 * scale && nrp <= 1e-9 * scale && mu <= 1e-12 * scale) done = true;  // same tolerances as the oracle's solveQpIpm
      // (a level with slack variables of its own -- the first -- is tried from mu <= 1e-2 scale on, up to four times: kEarly*Own of the CPU restatement)
      else if (earlyTries < (own ? QM_IPM_EARLY_TRIES_OWN : 2) && nrd <= (own ? QM_IPM_EARLY_NRD_OWN : 1e-4) * scale && nrp <= (own ? QM_IPM_EARLY_NRP_OWN : 1e-6) * scale &&
               mu <= (own ? QM_IPM_EARLY_MU_OWN : 1e-6) * scale && mu <= (own ? QM_IPM_EARLY_DROP_OWN : 0.01) * lastTryMu) { done = true; early = true; ++earlyTries; lastTryMu = mu; }
      // stagnation: complementarity no longer halves although it is already small (round-off floor of the normal equations) --
      // stop here instead of iterating into the divergence that follows; the polish finishes the job
      else if (it > 0 && mu > 0.5 * muPrev && mu <= QM_IPM_STAGNATION_MU * scale && nrp <= 1e-9 * scale && nrd <= 1e-7 * scale) done = true;
      if (it >= 39 && !done) { if (restartOrGiveUp()) { it = -1; continue; } itOut = 60; break; }
      if (done) {
        itOut = it;
        const bool c1 = rowActive && l1 > s1, c2 = rowActive && own && l2 > s2;
        isE = c1 && (!own || c2); isV = c1 && own && !c2;
        lamE = isE ? l1 : 0.0; zIpm = zc; polish = 1; corrections = 0;
        continue;                                                    // residuals again, now in polish form (zc may have been restored)
      }
      zcPrev = zc; s1p = s1; l1p = l1; s2p = s2; l2p = l2; vp = v; nrdPrev = nrd; muPrev = mu;
    }

    QM_TICK(2);
    const double w1 = l1 / s1, w2 = l2 / s2, kvv = 1.0 + w1 + w2;
    if (polish <= 1) {   // the polish steps share one matrix: build and factorise it once
      // ---- K = G + DZ' diag(w) DZ: upper-triangle tiles on the matrix cores, mirrored into LDS
      if (lane < 56) io.wtL[lane] = polish ? (isE ? rho : (isV ? 1.0 : 0.0)) : (rowActive ? (own ? w1 - w1 * w1 / kvv : w1) : 0.0);
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

      QM_TICK(3);
      // ---- factorisation K = L L^T by row operations: lane c holds column c of K in kc; after step j, kc[j] of lane c is
      //      L^T[j][c] = L[c][j], i.e. lane c ends up with ROW c of L (entries r <= c)
      myInv = 1.0;
  #pragma unroll
      for (int r = 0; r < NP; ++r) {
        const double kv = io.Kt[r * LDK_ + colL];   // K symmetric: column c = row c, read conflict free
        kc[r] = (colOn && r < n) ? kv : ((r == lane) ? 1.0 : 0.0);   // identity padding beyond n
      }
      {
        double bcP[3] = {0.0, 0.0, 0.0}, ncP = 0.0;
        IpmFactorStep<NP, 0>::run(kc, myInv, bcP, ncP, pivotFloor, lane, red);
      }
      // the back substitution L^T dz = t walks the COLUMNS of L^T: U[r][c] (c > r) sits in lane c, register r.  One transpose
      // through LDS per factorisation puts it into lane r, register c.
      QM_WAVE_SYNC();
      if (lane < NP) {
  #pragma unroll
        for (int r = 0; r < NP; ++r) io.Kt[lane * LDK_ + r] = kc[r];
      }
      QM_WAVE_SYNC();
  #pragma unroll
      for (int cc = 0; cc < NP; ++cc) uc[cc] = io.Kt[cc * LDK_ + colL];   // U[lane][cc] for cc > lane
      QM_WAVE_SYNC();
    }

    QM_TICK(4);
    double dv = 0.0, ds1 = 0.0, ds2 = 0.0, dl1 = 0.0, dl2 = 0.0, dzc = 0.0;
    double alphaAff = 1.0, sigma = 0.0, cw = 1.0;
#pragma unroll 1
    for (int pass = 0; pass < (polish ? 1 : 2); ++pass) {
      const double rc1 = pass == 0 ? s1 * l1 : s1 * l1 + cw * ds1 * dl1 - sigma * mu;
      const double rc2 = pass == 0 ? s2 * l2 : s2 * l2 + cw * ds2 * dl2 - sigma * mu;
      const double t1 = rowActive ? (l1 * rp1 - rc1) / s1 : 0.0;
      const double t2 = (rowActive && own) ? (l2 * rp2 - rc2) / s2 : 0.0;
      const double rhsv = -rdv + t1 + t2;
      const double tz = polish ? 0.0 : (rowActive ? (own ? t1 - (w1 / kvv) * rhsv : t1) : 0.0);   // polish: rhs = -gradient only
      // right-hand side of the reduced system
      double acc;
      {
        QM_WAVE_SYNC();
        bc[lane] = tz;
        QM_WAVE_SYNC();
        const double dtt = ipmColSum<LDZ_>(io, lane);   // D^T t: 56 rows, shared with the helper wavefronts (every lane takes part in the barriers)
        acc = colOn ? -rdz - dtt : 0.0;
      }
      QM_TICK(6);
      // L t = rhs (forward substitution; lane c owns row c of L)
      double tC = 0.0;
#pragma unroll
      for (int r = 0; r < NP; ++r) {
        const double tr = qmReadLane(acc * myInv, r, red);
        if (lane == r) tC = tr;
        acc -= (lane > r) ? kc[r] * tr : 0.0;
      }
      QM_TICK(7);
      // L^T dz = t (back substitution; lane r owns row r of L^T in uc)
      {
        double bacc = tC;
#pragma unroll
        for (int cc = NP - 1; cc >= 0; --cc) {
          const double dc = qmReadLane(bacc * myInv, cc, red);
          if (lane == cc) dzc = dc;
          bacc -= (lane < cc) ? uc[cc] * dc : 0.0;
        }
        dzc = colOn ? dzc : 0.0;
      }
      QM_TICK(8);
      if (polish) { zc += dzc; ++polish; break; }
      double Ddz;
      {
        QM_WAVE_SYNC();
        bc[lane] = dzc;
        QM_WAVE_SYNC();
        double d0 = 0.0, d1 = 0.0;
#pragma unroll 1
        for (int j = 0; j < NP; j += 4) {
          const double t0 = DZ[rowL * LDZ_ + j], t1 = DZ[rowL * LDZ_ + j + 1], t2 = DZ[rowL * LDZ_ + j + 2], t3 = DZ[rowL * LDZ_ + j + 3];
          const double z0 = bc[j], z1 = bc[j + 1], z2 = bc[j + 2], z3 = bc[j + 3];
          d0 += t0 * z0; d1 += t1 * z1; d0 += t2 * z2; d1 += t3 * z3;
        }
        Ddz = d0 + d1;
      }
      QM_TICK(9);
      if (rowActive) {
        if (own) {
          dv = (rhsv + w1 * Ddz) / kvv;
          ds1 = -rp1 - (Ddz - dv); ds2 = -rp2 + dv;
          dl1 = (-rc1 - l1 * ds1) / s1; dl2 = (-rc2 - l2 * ds2) / s2;
        } else { ds1 = -rp1 - Ddz; dl1 = (-rc1 - l1 * ds1) / s1; }
      }
      double amax = 1.0;
      if (rowActive) {
        if (ds1 < 0) amax = fmin(amax, -s1 / ds1);
        if (dl1 < 0) amax = fmin(amax, -l1 / dl1);
        if (own) { if (ds2 < 0) amax = fmin(amax, -s2 / ds2); if (dl2 < 0) amax = fmin(amax, -l2 / dl2); }
      }
      amax = allMin(amax);
      if (pass == 0) {
        alphaAff = amax;
        const double muAff = allSum(rowActive ? ((s1 + alphaAff * ds1) * (l1 + alphaAff * dl1) + (own ? (s2 + alphaAff * ds2) * (l2 + alphaAff * dl2) : 0.0)) : 0.0) / nRowsTot;
        const double ratio = muAff / mu;
        sigma = ratio * ratio * ratio;
        cw = fmin(1.0, 4.0 * alphaAff);
      } else {
        const double tau = fmax(0.995, 1.0 - mu);
        const double al = fmin(1.0, tau * amax);
        zc += al * dzc;
        if (rowActive) { s1 += al * ds1; l1 += al * dl1; if (own) { v += al * dv; s2 += al * ds2; l2 += al * dl2; } }
      }
      QM_TICK(10);
    }
  }
  QM_TICK(5);
  QM_TICK_FLUSH(NP == 36 ? 160 : (NP == 20 ? 256 : 288), blockIdx.x == 0 && lane == 0);
  if (lane < 36) io.zs[lane] = colOn ? zc : 0.0;
  return IpmResult{itOut, v};
}

}  // namespace qmk
