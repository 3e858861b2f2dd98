#!/usr/bin/env python3
"""What does the reference's OWN hierarchical QP return on a WBC tick, to 50 digits?  (VERDICT r05 task 1; TEST INFRASTRUCTURE, never imported by qm_door_amd.)

The cascade of qm_wbc/src/HoQp.cpp:12-158 is restated here LITERALLY, level by level, in mpmath arithmetic (50 digits):

    H = blkdiag((A Z)'(A Z) + 1e-12 I, I_slack)                                  HoQp.cpp:60-76  -- the regulariser is IN the matrix, nothing is taken in a limit
    c = [(A Z)'(A x_prev - b); 0]                                                HoQp.cpp:78-90
    D = [0 -I; D_prev Z 0; D Z -I],  f = [0; f_prev - D_prev x_prev + v_prev; f - D x_prev]      HoQp.cpp:92-124 (margins NOT clamped)
    z* = argmin 1/2 z'Hz + c'z  s.t.  D z <= f                                    HoQp.cpp:136-149 (what qpOASES is asked for; H is positive definite: z* is unique)
    x = x_prev + Z z*,   v = slack part of z*,   Z <- Z kernel(A Z)               HoQp.cpp:126-133, HoQp.h:31-34

Two things are taken from double precision because the reference takes them there: the task matrices (A, b, D, f) of the tick -- the oracle's task builders, the same
numbers the reference's WbcBase::formulate*Task would hand to HoQp -- and the CHOICE of free columns of Eigen's FullPivLU::kernel() (the pivot order of the double
computation, oracle/qmo_core.h kernelFullPivLU = Eigen 3.3's order); given the free columns the basis [-U11^-1 U12; I] is unique and is formed here exactly.  The basis
matters: the regulariser acts in the level's OWN coordinates z, so among the minimisers of a level's task it selects the one of smallest |z| -- another basis, another point.

The QP itself: the slack block is diagonal, v_i = max(0, d_i z - f_i) at the optimum, so the dense (n + s)-variable QP equals
    min 1/2 z'(G + 1e-12 I) z + g'z + 1/2 sum_own max(0, d_i z - f_i)^2   s.t.  d_i z <= f_i (inherited)
exactly; that form is solved by a primal-dual active-set iteration on (P = pinned inherited rows, V = violated own rows) in 50-digit arithmetic, started from the sets
the double-precision solution shows, and the result is VERIFIED against the literal dense QP: (z*, v*) with its multipliers satisfies the KKT conditions of
(H, c, D, f) as built above -- stationarity, primal and dual feasibility, complementarity -- to 1e-40.  A level that fails the check raises.

Output per tick: x* (36) and tau = [M_j, -J_j'] x* + h_j (WbcBase::updateCmd, WbcBase.cpp:580-595), in blocks: legs tau[0:12] (what QMController.cpp:428-431 commands in
the separated-system plugin), arm tau[12:18], accelerations x[0:24], contact forces x[24:36].

  python tools/hoqp_exact.py --make-fixture      # CPU: tests/golden/hoqp_exact_ticks.npz = inputs + exact answers of seeded ticks of both controllers
  python tools/hoqp_exact.py --make-gaits        # CPU: tests/golden/hoqp_exact_gaits.npz = the same for the start-up branch, flying trot and static walk
  python tools/hoqp_exact.py --make-offenders    # CPU, after a GPU run of the full-size closed loop / stress tests: tests/golden/hoqp_exact_offenders.npz
  python tools/hoqp_exact.py --report [--noise] [--fixture F]    # CPU: oracle vs exact on the fixture, per block  (the GPU side: tests/test_gpu_wbc.py::test_wbc_against_the_50_digit_solution_of_the_reference_qp)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import mpmath as mp  # noqa: E402

DIGITS = 50
REG = mp.mpf("1e-12")            # HoQp.cpp:66
BLOCKS = {"tau_legs": (36, 48), "tau_arm": (48, 54), "accelerations": (0, 24), "contact_forces": (24, 36)}


# ------------------------------------------------------------------------------------------------ small dense linear algebra on lists of mpf
def M(a):
    a = np.asarray(a, dtype=np.float64)
    return [[mp.mpf(float(v)) for v in row] for row in a] if a.ndim == 2 else [mp.mpf(float(v)) for v in a]


def matmul(A, B):
    Bt = list(zip(*B)) if B and B[0] else []
    return [[mp.fdot(row, col) for col in Bt] for row in A]


def matvec(A, x):
    return [mp.fdot(row, x) for row in A]


def tmatvec(A, y):      # A' y
    if not A:
        return []
    return [mp.fdot([A[i][j] for i in range(len(A))], y) for j in range(len(A[0]))]


def cholesky(K):
    n = len(K)
    L = [[mp.mpf(0)] * n for _ in range(n)]
    for j in range(n):
        d = K[j][j] - mp.fdot(L[j][:j], L[j][:j])
        if d <= 0:
            raise ArithmeticError("matrix not positive definite in 50-digit arithmetic")
        L[j][j] = mp.sqrt(d)
        for i in range(j + 1, n):
            L[i][j] = (K[i][j] - mp.fdot(L[i][:j], L[j][:j])) / L[j][j]
    return L


def chol_solve(L, b):
    n = len(L)
    y = [mp.mpf(0)] * n
    for i in range(n):
        y[i] = (b[i] - mp.fdot(L[i][:i], y[:i])) / L[i][i]
    x = [mp.mpf(0)] * n
    for i in range(n - 1, -1, -1):
        x[i] = (y[i] - mp.fdot([L[k][i] for k in range(i + 1, n)], x[i + 1:])) / L[i][i]
    return x


def lu_solve_square(A, B):
    """A X = B by Gaussian elimination with partial pivoting (A square, non-singular), B a list of right-hand-side columns as a matrix"""
    n = len(A)
    A = [row[:] for row in A]
    B = [row[:] for row in B]
    for k in range(n):
        p = max(range(k, n), key=lambda i: abs(A[i][k]))
        if A[p][k] == 0:
            raise ArithmeticError("singular pivot block")
        A[k], A[p] = A[p], A[k]
        B[k], B[p] = B[p], B[k]
        for i in range(k + 1, n):
            f = A[i][k] / A[k][k]
            if f != 0:
                A[i] = [a - f * b for a, b in zip(A[i], A[k])]
                B[i] = [a - f * b for a, b in zip(B[i], B[k])]
    X = [None] * n
    for i in range(n - 1, -1, -1):
        X[i] = [(B[i][j] - mp.fdot([A[i][k] for k in range(i + 1, n)], [X[k][j] for k in range(i + 1, n)])) / A[i][i] for j in range(len(B[0]))]
    return X


# ------------------------------------------------------------------------------------------------ Eigen's kernel basis, free columns from the double computation
def kernel_basis(orc, AZ):
    """N = P [-U11^-1 U12; I] of Eigen's FullPivLU::kernel() (HoQp.cpp:129): the free columns and the pivot rows are those of the DOUBLE computation (the reference's),
    the entries are exact.  Returns N (n x dim) as mpf lists and the free columns."""
    r, n = len(AZ), len(AZ[0])
    Ad = np.array([[float(v) for v in row] for row in AZ])
    _, free, seq = orc.kernel_full_piv_lu(Ad)
    # replay the pivot positions on index lists to get the ORIGINAL rows / columns of the pivots that count (rank = n - dim)
    rows, cols = list(range(r)), list(range(n))
    rank = n - len(free)
    prow, pcol = [], []
    for k, (pr, pc) in enumerate(seq):
        rows[k], rows[pr] = rows[pr], rows[k]
        cols[k], cols[pc] = cols[pc], cols[k]
        if k < rank:
            prow.append(rows[k]); pcol.append(cols[k])
    assert sorted(pcol + list(free)) == list(range(n)), "pivot replay does not partition the columns"
    if not free:
        return [[] for _ in range(n)], []
    A11 = [[AZ[i][j] for j in pcol] for i in prow]
    A12 = [[-AZ[i][j] for j in free] for i in prow]
    X = lu_solve_square(A11, A12) if rank else []
    N = [[mp.mpf(0)] * len(free) for _ in range(n)]
    for a, j in enumerate(pcol):
        N[j] = X[a]
    for b, j in enumerate(free):
        N[j] = [mp.mpf(1) if q == b else mp.mpf(0) for q in range(len(free))]
    # every row of A Z (not only the pivot rows) must vanish on the basis: the double computation's rank is the exact one up to its threshold
    res = max((abs(v) for row in matmul(AZ, N) for v in row), default=mp.mpf(0))
    scale = max(abs(v) for row in AZ for v in row)
    assert res <= mp.mpf("1e-9") * scale, f"kernel residual {mp.nstr(res, 5)}: the double computation dropped a pivot that counts"
    return N, list(free)


# ------------------------------------------------------------------------------------------------ one level, exactly
def kkt_point(G, g, Down, fown, Dinh, finh, V, Pl):
    """The minimiser of 1/2 z'(G + REG I) z + g'z + 1/2 sum_{i in V} (d_i z - f_i)^2 on {d_i z = f_i, i in Pl} and the multipliers of Pl (range-space method; pinned rows
    that are combinations of others get a zero multiplier: they are implied)."""
    n = len(g)
    K = [[G[i][j] + (REG if i == j else 0) for j in range(n)] for i in range(n)]
    rhs = [-v for v in g]
    for i in V:
        d = Down[i]
        for a in range(n):
            if d[a] != 0:
                rhs[a] += d[a] * fown[i]
                for b in range(n):
                    K[a][b] += d[a] * d[b]
    L = cholesky(K)
    lam = {}
    if not Pl:
        return chol_solve(L, rhs), lam
    Kinv_rhs = chol_solve(L, rhs)
    cols = [chol_solve(L, Dinh[i]) for i in Pl]                 # K^-1 d_i
    S = [[mp.fdot(Dinh[i], cols[b]) for b in range(len(Pl))] for i in Pl]
    r = [mp.fdot(Dinh[i], Kinv_rhs) - finh[i] for i in Pl]
    idx = list(range(len(Pl)))
    A = [S[a][:] + [r[a]] for a in range(len(Pl))]
    dmax = max(S[a][a] for a in range(len(Pl)))
    rk = 0
    for k in range(len(Pl)):
        p = max(range(k, len(Pl)), key=lambda a: abs(A[a][a]))
        A[k], A[p] = A[p], A[k]
        for row in A:
            row[k], row[p] = row[p], row[k]
        idx[k], idx[p] = idx[p], idx[k]
        if abs(A[k][k]) <= mp.mpf("1e-30") * dmax:
            break
        rk += 1
        for a in range(k + 1, len(Pl)):
            f_ = A[a][k] / A[k][k]
            if f_ != 0:
                A[a] = [x - f_ * y for x, y in zip(A[a], A[k])]
    sol = [mp.mpf(0)] * rk
    for k in range(rk - 1, -1, -1):
        sol[k] = (A[k][-1] - mp.fdot(A[k][k + 1:rk], sol[k + 1:rk])) / A[k][k]
    for k in range(len(Pl)):
        lam[Pl[idx[k]]] = sol[k] if k < rk else mp.mpf(0)
    z = [a - mp.fdot([cols[b][q] for b in range(len(Pl))], [lam[Pl[b]] for b in range(len(Pl))]) for q, a in enumerate(Kinv_rhs)]
    return z, lam


def solve_level(G, g, Down, fown, Dinh, finh, guess_z):
    """min 1/2 z'(G + REG I) z + g'z + 1/2 sum max(0, Down z - fown)^2  s.t.  Dinh z <= finh.  A primal-dual active-set iteration on (V, P) from the sets the double-
    precision solution shows (one to three solves when it settles); if it cycles, the textbook primal method from z = 0.  Returns z, v, multipliers of the inherited rows, solves."""
    mo, mi = len(Down), len(Dinh)
    gz = [mp.mpf(float(v)) for v in guess_z]
    rv = [a - b for a, b in zip(matvec(Down, gz), fown)] if mo else []
    ri = [a - b for a, b in zip(matvec(Dinh, gz), finh)] if mi else []
    fs = max([mp.mpf(1)] + [abs(v) for v in fown] + [abs(v) for v in finh])
    tol = mp.mpf("1e-40") * fs
    V = {i for i in range(mo) if rv[i] > 0}
    P = {i for i in range(mi) if ri[i] > -mp.mpf("1e-9") * fs}
    seen = set()
    for it in range(60):
        key = (frozenset(V), frozenset(P))
        if key in seen:
            break
        seen.add(key)
        z, lam = kkt_point(G, g, Down, fown, Dinh, finh, V, sorted(P))
        rv = [a - b for a, b in zip(matvec(Down, z), fown)] if mo else []
        ri = [a - b for a, b in zip(matvec(Dinh, z), finh)] if mi else []
        Vn = {i for i in range(mo) if rv[i] > 0}
        Pn = {i for i in range(mi) if (i in P and lam[i] > 0) or ri[i] > tol}
        if Vn == V and all(ri[i] <= tol for i in range(mi)) and all(lam[i] >= -tol for i in P):
            return z, [max(mp.mpf(0), x) for x in rv], {i: lam.get(i, mp.mpf(0)) for i in range(mi)}, it + 1
        V, P = Vn, Pn
    return solve_level_primal(G, g, Down, fown, Dinh, finh)


def solve_level_primal(G, g, Down, fown, Dinh, finh):
    """The textbook primal active-set method in 50-digit arithmetic, from z = 0 (feasible: the inherited margins are >= 0): the step to the minimiser on the working set is cut
    at the first inherited row it reaches (pinned) or the first own row that changes side (its quadratic penalty switches on or off: the objective is C^1, the curvature changes);
    at a stationary point the pinned row with the most negative multiplier leaves (ties: smallest index)."""
    n, mo, mi = len(g), len(Down), len(Dinh)
    fs = max([mp.mpf(1)] + [abs(v) for v in fown] + [abs(v) for v in finh])
    tol = mp.mpf("1e-40") * fs
    z = [mp.mpf(0)] * n
    assert all(f >= -tol for f in finh), "z = 0 is not feasible for the inherited rows"
    V = {i for i in range(mo) if -fown[i] > 0}
    P = set()
    for it in range(3000):
        zt, lam = kkt_point(G, g, Down, fown, Dinh, finh, V, sorted(P))
        p = [a - b for a, b in zip(zt, z)]
        pm = max(abs(x) for x in p)
        if pm <= mp.mpf("1e-45") * max([mp.mpf(1)] + [abs(x) for x in z]):
            neg = [i for i in sorted(P) if lam[i] < -tol]
            if not neg:
                rv = [a - b for a, b in zip(matvec(Down, z), fown)] if mo else []
                return z, [max(mp.mpf(0), x) for x in rv], {i: lam.get(i, mp.mpf(0)) for i in range(mi)}, it + 1
            worst = min(neg, key=lambda i: (lam[i], i))
            P.discard(worst)
            continue
        alpha, event = mp.mpf(1), None
        if mi:
            Dp, Dz = matvec(Dinh, p), matvec(Dinh, z)
            for i in range(mi):
                if i in P or Dp[i] <= 0:
                    continue
                a = max(mp.mpf(0), finh[i] - Dz[i]) / Dp[i]
                if a < alpha:
                    alpha, event = a, ("pin", i)
        if mo:
            Dp, Dz = matvec(Down, p), matvec(Down, z)
            for i in range(mo):
                r = Dz[i] - fown[i]
                if i in V and Dp[i] < 0:
                    a = max(mp.mpf(0), r) / -Dp[i]
                    if a < alpha:
                        alpha, event = a, ("off", i)
                elif i not in V and Dp[i] > 0:
                    a = max(mp.mpf(0), -r) / Dp[i]
                    if a < alpha:
                        alpha, event = a, ("on", i)
        z = [a + alpha * b for a, b in zip(z, p)]
        if event:
            kind, i = event
            if kind == "pin":
                P.add(i)
            elif kind == "off":
                V.discard(i)
            else:
                V.add(i)
    raise ArithmeticError("primal active-set method did not terminate")


def verify_literal(G, g, Down, fown, Dinh, finh, z, v, lam_inh):
    """KKT conditions of the dense QP (H, c, D, f) exactly as HoQp::formulateProblem builds it -- H = blkdiag(G + 1e-12 I, I), c = [g; 0], D = [0 -I; D_inh 0; D_own -I] --
    at (z, v): returns the largest violation (relative)."""
    n, s, mi = len(z), len(v), len(Dinh)
    Gz = matvec(G, z)
    c = g
    # multipliers: rows [0 -I] (v >= 0): nu_i; inherited rows: lam_inh; own rows [D Z -I]: mu_i.  Stationarity in v: v_i - nu_i - mu_i = 0; with mu_i = v_i (active own row), nu_i = 0.
    mu = v[:]
    stat = [Gz[a] + REG * z[a] + c[a] for a in range(n)]
    for i in range(mi):
        if lam_inh[i] != 0:
            stat = [x + lam_inh[i] * d for x, d in zip(stat, Dinh[i])]
    for i in range(s):
        if mu[i] != 0:
            stat = [x + mu[i] * d for x, d in zip(stat, Down[i])]
    scale = max([mp.mpf(1)] + [abs(x) for x in c] + [abs(x) for x in fown] + [abs(x) for x in finh])
    worst = max([abs(x) for x in stat] + [mp.mpf(0)])
    ri = [a - b for a, b in zip(matvec(Dinh, z), finh)] if mi else []
    ro = [a - b - w for a, b, w in zip(matvec(Down, z), fown, v)] if s else []
    worst = max([worst] + [max(mp.mpf(0), x) for x in ri] + [max(mp.mpf(0), x) for x in ro])                      # primal feasibility
    worst = max([worst] + [max(mp.mpf(0), -lam_inh[i]) for i in range(mi)] + [max(mp.mpf(0), -x) for x in v])     # dual feasibility, v >= 0
    worst = max([worst] + [abs(lam_inh[i] * ri[i]) for i in range(mi)] + [abs(mu[i] * ro[i]) for i in range(s)])  # complementarity
    return worst / scale


def exact_tick(orc, variant, xd, ud, rbd, mode, period, time, il, verbose=False, round_data=None, round_products=None):
    """The cascade on one tick.  Returns dict(out [54] = [x (36); tau (18)] as float64, levels=[...]).
    round_data (a numpy Generator): every non-zero entry of the task data (A, b, D, f of the three levels) is multiplied by 1 + 2^-53 u, u uniform in (-1, 1) -- ONE rounding of
    the numbers the reference hands to HoQp.  The exact solution of the problem so perturbed is what a backward-stable double-precision solver of the reference's problem returns at
    best.  The exact problem turns out to be insensitive to that (1e-16 .. 1e-14): what double precision loses is lost when HoQp FORMS its matrices.
    round_products (a numpy Generator): A Z, D Z and then H = (A Z)'(A Z), c = (A Z)'(A x - b) are rounded as a double-precision matrix product rounds them -- every entry of a
    product sum_k a_k b_k by 2^-53 u sum_k |a_k b_k|, u uniform in (-1, 1), H kept symmetric -- (HoQp.cpp:60-90 computes them in double with Eigen) and the QP with THOSE matrices is
    solved exactly: no solver error included, so how far the answer moves is a LOWER bound of the reference's own double-precision noise on this tick (--noise)."""
    mp.mp.dps = DIGITS
    U = mp.mpf(2) ** -53

    def rprod(Am, Bm, sym=False):
        """A B with the rounding of a computed product"""
        if not Am or not Bm or not Bm[0]:
            return matmul(Am, Bm)
        Bt = list(zip(*Bm))
        out = [[mp.fdot(row, col) + U * mp.mpf(float(round_products.uniform(-1, 1))) * mp.fdot([abs(x) for x in row], [abs(y) for y in col]) for col in Bt] for row in Am]
        if sym:
            for i in range(len(out)):
                for j in range(i):
                    out[i][j] = out[j][i]
        return out
    tasks = [orc.wbc_task(l, xd, ud, rbd, mode, period, time, il, variant) for l in range(3)]
    model = orc.wbc_model(xd, ud, rbd, period, il)
    nd = 36
    x = [mp.mpf(0)] * nd
    Z = [[mp.mpf(1) if i == j else mp.mpf(0) for j in range(nd)] for i in range(nd)]
    Dprev, fprev, vprev = [], [], []            # stacked inequality rows of the levels above, with the slack they ended with
    info = []
    for level, t in enumerate(tasks):
        n = len(Z[0]) if Z else 0
        if n == 0:
            break            # FLY: nothing left to decide
        A, b, D, f = M(t["A"]), M(t["b"]), M(t["D"]), M(t["f"])
        if round_data is not None:
            u = mp.mpf(2) ** -53
            A = [[v * (1 + u * mp.mpf(float(round_data.uniform(-1, 1)))) if v != 0 else v for v in row] for row in A]
            D = [[v * (1 + u * mp.mpf(float(round_data.uniform(-1, 1)))) if v != 0 else v for v in row] for row in D]
            b = [v * (1 + u * mp.mpf(float(round_data.uniform(-1, 1)))) for v in b]
            f = [v * (1 + u * mp.mpf(float(round_data.uniform(-1, 1)))) for v in f]
        mm = rprod if round_products is not None else (lambda Am, Bm, sym=False: matmul(Am, Bm))
        AZ = mm(A, Z) if A else []
        resid = [a - bb for a, bb in zip(matvec(A, x), b)] if A else []
        if round_products is not None and A:
            resid = [r_ + U * mp.mpf(float(round_products.uniform(-1, 1))) * (mp.fdot([abs(v) for v in row], [abs(v) for v in x]) + abs(bb)) for r_, row, bb in zip(resid, A, b)]
        G = mm(list(map(list, zip(*AZ))), AZ, True) if AZ else [[mp.mpf(0)] * n for _ in range(n)]
        g = [r_[0] for r_ in mm(list(map(list, zip(*AZ))), [[v] for v in resid])] if AZ else [mp.mpf(0)] * n
        Dinh = mm(Dprev, Z) if Dprev else []
        finh = [fp - dx + vp for fp, dx, vp in zip(fprev, matvec(Dprev, x), vprev)] if Dprev else []      # HoQp.cpp:104-111: not clamped
        if round_products is not None:
            finh = [max(mp.mpf(0), v_) for v_ in finh]      # (with rounded products the slack of the level above and D x no longer cancel exactly: a margin of -1e-16 is a zero margin)
        Down = mm(D, Z) if D else []
        fown = [fi - dx for fi, dx in zip(f, matvec(D, x))] if D else []
        lv = orc.wbc_level(level, xd, ud, rbd, mode, period, time, il, variant)
        guess = lv["sol"][:n] if lv["num_dec"] == n else np.zeros(n)
        z, v, lam, its = solve_level(G, g, Down, fown, Dinh, finh, guess)
        kkt = verify_literal(G, g, Down, fown, Dinh, finh, z, v, lam)
        assert kkt <= mp.mpf("1e-38"), f"level {level}: KKT residual of the literal QP {mp.nstr(kkt, 5)}"
        x = [xi + dz for xi, dz in zip(x, matvec(Z, z))]
        info.append(dict(level=level, n=n, rows_own=len(Down), rows_inherited=len(Dinh), pd_iterations=its, kkt=float(kkt),
                         pinned=[i for i in range(len(Dinh)) if lam[i] > 0], violated=[i for i in range(len(Down)) if v[i] > 0],
                         smallest_curvature=float(min((G[i][i] for i in range(n)), default=mp.mpf(0)))))
        if verbose:
            print(f"  level {level}: n {n} own {len(Down)} inherited {len(Dinh)} iterations {its} KKT {mp.nstr(kkt, 3)} pinned {info[-1]['pinned']} violated {info[-1]['violated']}", file=sys.stderr)
        # stack: [own; inherited] (Task::operator+), slacks alongside
        Dprev = D + Dprev
        fprev = f + fprev
        vprev = v + vprev
        if AZ:
            N, _ = kernel_basis(orc, AZ)
            Z = matmul(Z, N) if N and N[0] else [[] for _ in range(nd)]
    Mm, nle, J = model["M"], model["nle"], model["J"]
    tau = []
    for i in range(18):
        s = mp.mpf(float(nle[6 + i]))
        s += mp.fdot([mp.mpf(float(Mm[6 + i, j])) for j in range(24)], x[:24])
        s -= mp.fdot([mp.mpf(float(J[j, 6 + i])) for j in range(12)], x[24:36])
        tau.append(s)
    out = np.array([float(v) for v in x] + [float(v) for v in tau])
    return dict(out=out, levels=info)


def block_dev(got, ref):
    """rel-inf deviation per block of a [54] output (x | tau)"""
    return {k: float(np.abs(got[a:b] - ref[a:b]).max() / max(1.0, np.abs(ref[a:b]).max())) for k, (a, b) in BLOCKS.items()}


# ------------------------------------------------------------------------------------------------ fixture: seeded ticks of both controllers + the slow / ill-conditioned ones
def closed_loop_ticks(itf, orc, variant, batch, cycles, ticks, seed, pick, t_start=9.9, gait_start=0.15, gait="trot", first_cycle=None):
    """WBC inputs of ticks of the oracle's own closed loop (tests/closed_loop.py: plan-following robots in motion, inputLast_ carried): `pick` random (cycle, tick, instance)
    triples after the start-up branch, seeded."""
    import closed_loop as CL
    sc = CL.Scenario(itf, batch, cycles=cycles, t_start=t_start, gait_start=gait_start, seed=seed, gait=gait)
    be = CL.OracleBackend(orc, sc, variant)
    rng = np.random.default_rng(seed + 1000)
    wanted = set()
    while len(wanted) < pick:
        wanted.add((int(rng.integers(cycles // 3 if first_cycle is None else first_cycle, cycles)), int(rng.integers(0, ticks)), int(rng.integers(0, batch))))
    out = []
    rbd = sc.first_measurement()
    for k in range(cycles):
        t0 = sc.t_start + k * CL.MPC_PERIOD
        N, grid = sc.grid(t0)
        be.observe(rbd, t0)
        plan = be.mpc(t0, N, grid)
        for j in range(ticks):
            t = t0 + j * CL.WBC_PERIOD
            if not (k == 0 and j == 0):
                rbd = CL.measurement(sc, plan, t)
            w = be.tick(t, rbd, t)
            xd, ud, rb, md, tm, il = be.last
            for (kk, jj, i) in wanted:
                if kk == k and jj == j:
                    out.append(dict(variant=variant, xd=xd[i].copy(), ud=ud[i].copy(), rbd=rb[i].copy(), mode=int(md[i]), period=CL.WBC_PERIOD, time=float(tm), il=il[i].copy(),
                                    oracle=w["out"][i].copy(), source=(f"closed loop seed {seed} cycle {k} tick {j} instance {i}" if gait == "trot" and first_cycle is None else
                                                                       f"closed loop {gait} t_start {t_start} variant {variant} seed {seed} cycle {k} tick {j} instance {i}")))
        rbd = CL.measurement(sc, plan, t0 + CL.MPC_PERIOD)
    return out


def offender_ticks(itf, orc, path, variant=1, limit=40):
    """The ticks tests/test_closed_loop.py::test_closed_loop_256_instances_100_cycles[variant] recorded as offenders (GPU and oracle torques more than 1e-6 apart) in `path`
    (gpurun_out/closed_loop_v1.json of a GPU run): the oracle's side of that loop is deterministic and is run again here, the WBC inputs of those ticks are what it saw."""
    import closed_loop as CL
    rec = json.load(open(path))
    off = sorted(rec["offenders"], key=lambda o: -o["tau_dev"])[:limit]
    want = {(o["cycle"], o["tick"]): [] for o in off}
    for o in off:
        want[(o["cycle"], o["tick"])].append(o)
    B = rec["instances"]
    sc = CL.Scenario(itf, B, cycles=max(o["cycle"] for o in off) + 1, gait_start=0.55)        # the test's scenario
    be = CL.OracleBackend(orc, sc, variant)
    out = []
    rbd = sc.first_measurement()
    for k in range(sc.cycles):
        t0 = sc.t_start + k * CL.MPC_PERIOD
        N, grid = sc.grid(t0)
        be.observe(rbd, t0)
        plan = be.mpc(t0, N, grid)
        for j in range(rec["ticks_per_cycle"]):
            t = t0 + j * CL.WBC_PERIOD
            if not (k == 0 and j == 0):
                rbd = CL.measurement(sc, plan, t)
            w = be.tick(t, rbd, t)
            xd, ud, rb, md, tm, il = be.last
            for o in want.get((k, j), []):
                i = o["instance"]
                out.append(dict(variant=variant, xd=xd[i].copy(), ud=ud[i].copy(), rbd=rb[i].copy(), mode=int(md[i]), period=CL.WBC_PERIOD, time=float(tm), il=il[i].copy(), oracle=w["out"][i].copy(),
                                source=f"closed loop of test_closed_loop_256_instances_100_cycles[{variant}]: cycle {k} tick {j} instance {i}, GPU / oracle torques {o['tau_dev']:.1e} apart in that run"))
        rbd = CL.measurement(sc, plan, t0 + CL.MPC_PERIOD)
        print(f"  cycle {k}", end="\r", file=sys.stderr, flush=True)
    return out


def stress_ticks(itf, orc, variant, npz, extra=()):
    """Instances of the WBC stress batch (tests/support.py: wbc_stress_batch, seeded) whose GPU and oracle torques were more than 1e-6 apart in the GPU run that wrote `npz`
    (gpurun_out/wbc_stress_v<variant>.npz), plus `extra` instance numbers."""
    import support as S
    c = S.wbc_stress_batch(itf, variant)
    d = np.load(npz)
    e = np.maximum(S.rel_inf(d["gpu"][:, 36:], d["oracle"][:, 36:]), np.maximum(S.rel_inf(d["gpu"][:, 36:48], d["oracle"][:, 36:48]), S.rel_inf(d["gpu"][:, 48:], d["oracle"][:, 48:])))
    idx = sorted(set(np.nonzero(e > 1e-6)[0].tolist()) | set(extra))
    out = []
    for i in idx:
        st, ref, _ = orc.wbc_update(c["xd"][i], c["u"][i], c["rbd"][i], int(c["mode"][i]), 0.002, float(c["t"][i]), c["il"][i].copy(), variant=variant)
        out.append(dict(variant=variant, xd=c["xd"][i], ud=c["u"][i], rbd=c["rbd"][i], mode=int(c["mode"][i]), period=0.002, time=float(c["t"][i]), il=c["il"][i], oracle=ref,
                        source=f"stress batch variant {variant} instance {i}, GPU / oracle torques {e[i]:.1e} apart in that run"))
    return out


def solve_and_save(orc, ticks, path, note):
    for n_, t in enumerate(ticks):
        try:
            ex = exact_tick(orc, t["variant"], t["xd"], t["ud"], t["rbd"], t["mode"], t["period"], t["time"], t["il"])
        except (ArithmeticError, AssertionError) as e:       # said, not hidden: the tick stays out of the fixture
            print(n_, t["source"], "NOT SOLVED:", e, flush=True)
            t["exact"] = None
            continue
        t["exact"] = ex["out"]
        t["levels"] = json.dumps(ex["levels"])
        print(n_, t["source"], "variant", t["variant"], "mode", t["mode"], {k: f"{v:.1e}" for k, v in block_dev(t["oracle"], ex["out"]).items()}, flush=True)
    ticks = [t for t in ticks if t["exact"] is not None]
    keys = ("variant", "xd", "ud", "rbd", "mode", "period", "time", "il", "exact", "oracle", "levels", "source")
    np.savez_compressed(path, **{k: np.array([t[k] for t in ticks]) for k in keys}, note=np.array(note))
    print("wrote", path, len(ticks), "ticks")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--make-fixture", action="store_true")
    ap.add_argument("--make-gaits", action="store_true", help="tests/golden/hoqp_exact_gaits.npz: the start-up branch (t < 10 s), flying trot and static walk, both controllers")
    ap.add_argument("--make-offenders", action="store_true", help="tests/golden/hoqp_exact_offenders.npz from gpurun_out/closed_loop_v1.json and gpurun_out/wbc_stress_v{0,1}.npz of a GPU run")
    ap.add_argument("--report", action="store_true")
    ap.add_argument("--noise", action="store_true", help="with --report: also how far the EXACT solution moves when HoQp's matrices are formed with double-precision product rounding, three seeded draws per tick")
    ap.add_argument("--ticks-per-controller", type=int, default=50)
    ap.add_argument("--fixture", default=os.path.join(ROOT, "tests", "golden", "hoqp_exact_ticks.npz"))
    args = ap.parse_args()
    import support as S
    from qm_door_amd import abi, api
    itf = api.QMInterface(lib=abi.load_library(S.build_emu()))
    orc = S.Oracle(itf.problem)
    note = "exact: tools/hoqp_exact.py (50-digit solve of the reference's literal level QPs, 1e-12 I included, Eigen's kernel basis); oracle: oracle/ at generation time, for the record only"
    if args.make_fixture:
        ticks = []
        for variant in (0, 1):
            ticks += closed_loop_ticks(itf, orc, variant, batch=16, cycles=36, ticks=10, seed=61 + variant, pick=args.ticks_per_controller)
        solve_and_save(orc, ticks, args.fixture, note)
    if args.make_gaits:
        ticks = []
        for variant, gait, t_start, seed in ((0, "trot", 9.0, 71), (0, "flying_trot", 10.5, 72), (0, "static_walk", 10.5, 73), (1, "flying_trot", 10.5, 74), (1, "static_walk", 10.5, 75)):
            ticks += closed_loop_ticks(itf, orc, variant, batch=8, cycles=30, ticks=10, seed=seed, pick=16, t_start=t_start, gait_start=0.05, gait=gait, first_cycle=2)
        solve_and_save(orc, ticks, os.path.join(ROOT, "tests", "golden", "hoqp_exact_gaits.npz"), note)
    if args.make_offenders:
        fast = S.Oracle(itf.problem, fast=True)
        ticks = offender_ticks(itf, fast, os.path.join(ROOT, "gpurun_out", "closed_loop_v1.json"))
        ticks += stress_ticks(itf, orc, 1, os.path.join(ROOT, "gpurun_out", "wbc_stress_v1.npz"), extra=(925, 221, 917))       # (VERDICT r05 weak 2 names these three)
        ticks += stress_ticks(itf, orc, 0, os.path.join(ROOT, "gpurun_out", "wbc_stress_v0.npz"))
        solve_and_save(orc, ticks, os.path.join(ROOT, "tests", "golden", "hoqp_exact_offenders.npz"), note)
    if args.report:
        d = np.load(args.fixture, allow_pickle=False)
        rep = {"fixture": os.path.relpath(args.fixture, ROOT), "what": "rel-inf deviation per block (each block its own norm) of the CPU restatement (oracle/) from the 50-digit solution of the "
               "reference's literal level QPs (HoQp.cpp:60-134, 1e-12 I in the matrix, Eigen's kernel basis)" + ("; noise: how far that 50-digit solution itself moves when HoQp's "
               "matrices A Z, D Z, H = (A Z)'(A Z), c are rounded the way a double-precision product rounds them (HoQp.cpp:60-90 forms them in double) and the QP is still solved exactly, three seeded "
               "draws: a LOWER bound of the reference's own double-precision noise (its solver's rounding comes on top)" if args.noise else "")}
        for variant in (0, 1):
            idx = np.nonzero(d["variant"] == variant)[0]
            if len(idx) == 0:
                continue
            devs, noise = [], []
            for i in idx:
                st, out, _ = orc.wbc_update(d["xd"][i], d["ud"][i], d["rbd"][i], int(d["mode"][i]), float(d["period"][i]), float(d["time"][i]), d["il"][i].copy(), variant=variant)
                assert st == 0
                devs.append(block_dev(out, d["exact"][i]))
                if args.noise:
                    worst = {k: 0.0 for k in BLOCKS}
                    for draw in range(3):
                        ex = exact_tick(orc, variant, d["xd"][i], d["ud"][i], d["rbd"][i], int(d["mode"][i]), float(d["period"][i]), float(d["time"][i]), d["il"][i].copy(),
                                        round_products=np.random.default_rng(1000 * int(i) + draw))
                        for k, v in block_dev(ex["out"], d["exact"][i]).items():
                            worst[k] = max(worst[k], v)
                    noise.append(worst)
                    print(int(i), str(d["source"][i])[:60], "oracle", {k: f"{v:.1e}" for k, v in devs[-1].items()}, "noise", {k: f"{v:.1e}" for k, v in worst.items()}, file=sys.stderr, flush=True)
            name = "HierarchicalWbc" if variant == 0 else "HierarchicalMpcWbc"
            rep[name] = {"ticks": int(len(idx)), "oracle_vs_exact": {k: {"median": float(np.median([e[k] for e in devs])), "max": float(np.max([e[k] for e in devs]))} for k in BLOCKS}}
            if args.noise:
                rep[name]["exact_when_HoQp_forms_its_matrices_in_double"] = {k: {"median": float(np.median([e[k] for e in noise])), "max": float(np.max([e[k] for e in noise]))} for k in BLOCKS}
                ratio = [max(devs[n][k] / max(noise[n][k], 1e-300) for k in ("tau_legs", "tau_arm")) for n in range(len(idx))]
                rep[name]["oracle_deviation_over_that_noise_torques"] = {"median": float(np.median(ratio)), "p90": float(np.percentile(ratio, 90)), "max": float(np.max(ratio)),
                                                                           "ticks_where_the_oracle_is_within_10x_the_noise": int(sum(r <= 10 for r in ratio))}
        print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
