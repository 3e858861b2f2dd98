"""TEST INFRASTRUCTURE.  A textbook primal active-set method for  min 1/2 z^T H z + c^T z  s.t.  D z <= f  (H positive semi-definite), in plain numpy:
what qpOASES is at its core, written here independently of both the oracle's interior point and the kernels', to pin the full-size HoQP level
problems ((92,112), (18,56), (8,56) in stance) by a solver of the same class as the reference's.

Each iteration solves the equality-constrained QP on the working set through its KKT system (least squares: working sets of the WBC are
routinely degenerate -- the friction-cone task carries all-zero rows); the step is cut at the first blocking constraint (Bland's rule on ties);
at a stationary point the most negative multiplier leaves the working set."""
import numpy as np


def _solve_once(H, c, D, f, z0, max_iter=1500, tol=1e-9, perturb=1e-9, seed=0):
    """`perturb`: the bounds are relaxed by a random amount of that relative size (the classical remedy against cycling at degenerate vertices:
    the WBC's working sets are degenerate by construction); the minimiser moves by the same order."""
    n = H.shape[0]
    z = np.array(z0, dtype=float)
    scale = max(1.0, np.abs(f).max(), np.abs(c).max())
    f_exact = f
    f = f + perturb * scale * np.random.default_rng(seed).uniform(0.5, 1.0, f.shape)
    assert (D @ z - f).max() <= 1e-7 * scale, "infeasible start"
    work = []                                           # working set (row indices), kept linearly independent
    for it in range(max_iter):
        g = H @ z + c
        Dw = D[work] if work else np.zeros((0, n))
        k = len(work)
        K = np.block([[H, Dw.T], [Dw, np.zeros((k, k))]])
        sol = np.linalg.lstsq(K, np.r_[-g, np.zeros(k)], rcond=1e-13)[0]
        p, lam = sol[:n], sol[n:]
        if np.abs(p).max() <= 1e-7 * max(1.0, np.abs(z).max()):     # stationary on the working set (least-squares noise of a vertex with n active rows is ~1e-9)
            if k == 0 or lam.min() >= -tol * scale:
                if k:   # the optimal working set is known: land on the vertex of the UNPERTURBED bounds
                    Kx = np.block([[H, Dw.T], [Dw, np.zeros((k, k))]])
                    zx = np.linalg.lstsq(Kx, np.r_[-c, f_exact[work]], rcond=1e-13)[0][:n]
                    if (D @ zx - f_exact).max() <= 1e-7 * scale:
                        z = zx
                return z, it
            work.pop(int(np.argmin(lam)))               # most negative multiplier leaves
            continue
        Dp = D @ p
        slack = f - D @ z
        alpha, block = 1.0, -1
        for i in np.where(Dp > tol * max(1.0, np.abs(p).max()))[0]:
            if i in work:
                continue
            a = max(slack[i], 0.0) / Dp[i]
            if a < alpha - 1e-14:
                alpha, block = a, i
        z = z + alpha * p
        if block >= 0:
            work = work + [block]
    raise RuntimeError("active-set method did not terminate")


def feasible_start(D, f, num_dec):
    """z = 0 in the decision part; the slack part (last D.shape[1] - num_dec variables, rows [0 -I] and [D_own -I]) large enough for every row."""
    n = D.shape[1]
    z = np.zeros(n)
    ns = n - num_dec
    if ns > 0:
        viol = np.maximum(0.0, -f)                       # D z = 0 at z = 0: a row with f < 0 needs its slack
        # the own rows are the LAST ns rows, each with -1 on its slack column
        z[num_dec:] = viol[-ns:] + 1e-9
    return z


def solve(H, c, D, f, z0):
    """Retries with other / larger random bound perturbations when a degenerate vertex makes one run cycle."""
    last = None
    for attempt, (perturb, seed) in enumerate(((1e-9, 0), (1e-9, 1), (1e-8, 2), (1e-8, 3), (1e-7, 4), (1e-7, 5))):
        try:
            return _solve_once(H, c, D, f, z0, perturb=perturb, seed=seed)
        except RuntimeError as e:
            last = e
    raise last
