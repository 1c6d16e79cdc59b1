"""Per-grid constants of the slow-mode landing guard of conv_plan 3 (include/beacon_hip.h: bcn_set_slow_mode_bound).

The reference stops its Jacobi solve at the first sweep k with err_k = d_k' W d_k <= tol, d_k = phi_k - phi_{k-1}, W = I + G
(rayleigh.py:448-454, mixing.py:457-463; G counts a cell's mirrored ghost sides).  The kernels do not evaluate err after every
sweep, so an evaluation that directly follows SKIPPED sweeps (a "landing") has to prove that none of them passed.  The global
bound err_k <= 1.030 err_j (scripts/weighted_norm_bound.py) does that with a landing threshold of BCN_CONV_GUARD = 1.035 tol --
but the residual decays by ~0.3 % per sweep where solves end, so 3.5 % are a dozen sweeps that must be evaluated one by one.

The growth the global bound allows needs fast modes: J = cx (E + W) + cy (N + S) is symmetric, its eigenvectors are products of
1D cosine modes, and d_k = J^(k-1) d_1.  Split d_k = l_k + h_k along the eigenvalue cutoff |lambda| >= lc / < lc (both spans are
invariant under J and orthogonal in the plain inner product).  Then |h_k|_2 <= lc^(k-1) |d_1|_2, and within the slow span

    C_L(lc) = sup_{m >= 1} sup_{x in span{v: |lambda_v| >= lc}}  (J^m x)' W (J^m x) / x' W x

is what this module computes: 1.0000 (lc = 0.9) and 1.0055 (lc = 0.8) on the 128x64 grid, against 1.030 over all modes.  If a
skipped sweep j had passed (err_j <= tol), then with e_j = sqrt(3) lc^(j-1) sqrt(|d_1|^2 / tol)   (W <= 3 I)

    sqrt(err_k) <= |W^1/2 l_k| + |W^1/2 h_k| <= sqrt(C_L) (sqrt(tol) + sqrt(tol) e_j) + sqrt(tol) e_k <= sqrt(C_L tol) (1 + 2 e_j),

so a landing at sweep k behind the last evaluated sweep i that finds err_k > C_L (1 + 2 e_(i+1))^2 tol proves that no skipped sweep
passed (C_L := max(1, C_L); e_j falls with j).  The kernel takes the smaller of this and BCN_CONV_GUARD, over two cutoffs; |d_1|^2
is the plain sum of squares of the right-hand side, reduced once per solve (ns2d_fast_impl.h).

C_L is a property of (nx, ny, boundary kind, cx): bounds(...) computes it with dense linear algebra on the slow span (seconds; cached
on disk beside the JIT plugins); the library has the reference's default grids built in (capi.hip) and takes any other through
bcn_set_slow_mode_bound -- without it the guard stays at BCN_CONV_GUARD, which holds for every grid with no side below 48 cells.
The constant mode of the all-Neumann problem (lambda = 1) is left out, as in the global bound: the increments of that problem sum
to zero (the right-hand side is the divergence of a field with no flow through the walls, and J keeps the sum) -- up to rounding,
which the kernels' 0.1 % margin on the guard covers.
"""
import json
import os

import numpy as np

CUTOFFS = (0.9, 0.8)       # the two cutoffs the kernels evaluate (NS2DArgs::slow_l2lc)
MAX_MODES = 2400           # spans above this size are skipped (bound = inf: that cutoff is not used)
PAD = 2e-4                 # added to every computed bound (Lanczos tolerance, the m not sampled between the geometric steps)


def _modes_1d(n, neumann_hi):
    a = np.zeros((n, n))
    i = np.arange(n - 1)
    a[i, i + 1] = a[i + 1, i] = 1.0
    a[0, 0] = 1.0
    if neumann_hi:
        a[n - 1, n - 1] = 1.0
    return np.linalg.eigh(a)


def slow_span(nx, ny, kind, cx, lc, basis=False):
    """(lam, G0): eigenvalues of J above the cutoff and the Gram matrix of W on their eigenvectors (basis=True: and the
    eigenvectors as an [nx * ny, n] matrix -- tests)."""
    cy = 0.5 - cx
    ax, vx = _modes_1d(nx, True)
    ay, vy = _modes_1d(ny, kind == 0)          # kind 1: Dirichlet-zero ghost above the top row (mixing.py:450-451)
    lam = cx * ax[:, None] + cy * ay[None, :]
    keep = np.abs(lam) >= lc
    if kind == 0:
        keep[np.unravel_index(np.argmax(lam), lam.shape)] = False       # the constant mode (lambda = 1): see the module docstring
    p, q = np.nonzero(keep)
    ex = np.outer(vx[0], vx[0]) + np.outer(vx[-1], vx[-1])                      # boundary columns, in mode space
    ey = np.outer(vy[0], vy[0]) + (np.outer(vy[-1], vy[-1]) if kind == 0 else 0.0)
    g0 = ex[np.ix_(p, p)] * (q[:, None] == q[None, :]) + ey[np.ix_(q, q)] * (p[:, None] == p[None, :])
    g0[np.diag_indices_from(g0)] += 1.0
    if basis:
        return lam[p, q], g0, np.einsum("ik,jk->ijk", vx[:, p], vy[:, q]).reshape(nx * ny, len(p))
    return lam[p, q], g0


def growth_in_span(lam, g0, ms, vectors=False):
    """C(m) = sup_x (L^m x)' G0 (L^m x) / x' G0 x for every m in ms: the top eigenvalue of A'A, A = S L^m S^-1, S = G0^(1/2).
    vectors=True: [(C(m), x)] with the maximiser x in mode coordinates (x' G0 x = 1).
    Lanczos (ARPACK, matrix-free: four n x n products per step) to a residual of 1e-7 C(m) (PAD is 2e-4), warm-started from the previous m's
    maximiser (it moves slowly with m), the first from a fixed vector: deterministic constants.  (Until round 5 a fixed 400 steps of
    power iteration: that approaches the eigenvalue from BELOW, and with gaps of ~0.5 % and a cold start it was 1.5e-3 short at
    m = 1 -- 0.99913 against 1.00061 on 128x64 -- where the guard needs an UPPER bound)"""
    from scipy.sparse.linalg import LinearOperator, eigsh
    w, v = np.linalg.eigh(g0)
    si = (v / np.sqrt(w)) @ v.T                # G0^(-1/2)
    sh = (v * np.sqrt(w)) @ v.T                # G0^(1/2)
    n = len(lam)
    out = []
    x = np.ones(n) / np.sqrt(n)
    for m in ms:
        lm = lam ** m
        if n <= 48:
            a = sh * lm[None, :] @ si
            ev, evec = np.linalg.eigh(a.T @ a)
            val, x = float(ev[-1]), evec[:, -1]
        else:
            def ata(y, lm=lm):
                return si @ (lm * (sh @ (sh @ (lm * (si @ y)))))        # A'A y (S, S^-1 symmetric)
            ev, evec = eigsh(LinearOperator((n, n), matvec=ata, dtype=np.float64), k=1, which="LA", tol=1e-7, ncv=min(n - 1, 48),
                             v0=x, maxiter=50000)
            val, x = float(ev[0]), evec[:, 0]
        out.append((val, si @ x) if vectors else val)
    return out


def growth_exact(lam, g0, ms):
    """C(m) with a cold start for every m (tests: the scan's warm start must not matter)."""
    return [growth_in_span(lam, g0, [m])[0] for m in ms]


def sweep_counts(limit=4096):
    """Every m up to 32, then steps of 10 % (C(m) is smooth in m; the peak region is refined by bound())."""
    ms = list(range(1, 33))
    m = 32.0
    while m < limit:
        m *= 1.1
        if int(m) > ms[-1]:
            ms.append(int(m))
    return ms


def bound(nx, ny, kind, cx, lc):
    """max(1, sup_m C(m)) + PAD on the slow span of the cutoff lc."""
    cy = 0.5 - cx
    ax = 2.0 * np.cos(np.pi * np.arange(nx) / nx)
    n_modes = int(np.count_nonzero(np.abs(cx * ax[:, None] + cy * _modes_1d(ny, kind == 0)[0][None, :]) >= lc))
    if n_modes > MAX_MODES:
        return float("inf")                     # (the dense algebra below is cubic in the span: not worth a minute per grid)
    lam, g0 = slow_span(nx, ny, kind, cx, lc)
    if len(lam) == 0:
        return 1.0 + PAD
    ms = sweep_counts()
    c = growth_in_span(lam, g0, ms)
    k = int(np.argmax(c))
    if ms[k] > 32:                              # the peak lies where m was sampled: every m between its neighbours
        lo, hi = ms[max(k - 1, 0)], ms[min(k + 1, len(ms) - 1)]
        fine = list(range(lo, hi + 1))
        cf = growth_in_span(lam, g0, fine)
        ms, c = ms + fine, c + cf
        k = int(np.argmax(c))
    return max(1.0, max(c)) + PAD


def _cache_file(nx, ny, kind, cx):
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_jit")     # beside the kernel plugins (beacon_amd/jit.py: JIT_DIR)
    return os.path.join(root, "slow_mode_%dx%d_k%d_cx%.9f.json" % (nx, ny, kind, cx))


def bounds(nx, ny, kind, cx, cutoffs=CUTOFFS, cache=True):
    """[(cutoff, C_L)] for the grid; cached on disk (the computation takes seconds to a minute for the widest grids)."""
    path = _cache_file(nx, ny, kind, cx)
    if cache and os.path.exists(path):
        try:
            with open(path) as f:
                d = json.load(f)
            if d.get("asked") == list(cutoffs) and d.get("pad") == PAD and d.get("exact") == 2:
                return list(zip(d["cutoffs"], d["bounds"]))
        except (OSError, ValueError):
            pass
    out = [(lc, bound(nx, ny, kind, cx, lc)) for lc in cutoffs]
    out = [(lc, b) for lc, b in out if np.isfinite(b)]
    if cache:
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            tmp = path + ".%d.tmp" % os.getpid()
            with open(tmp, "w") as f:
                json.dump({"asked": list(cutoffs), "cutoffs": [c for c, _ in out], "bounds": [b for _, b in out], "pad": PAD, "exact": 2}, f)
            os.replace(tmp, path)
        except OSError:
            pass
    return out
