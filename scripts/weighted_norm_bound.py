"""How much can the reference's residual norm GROW from one Jacobi sweep to a later one?  

The reference stops its Jacobi solve at the first sweep k with err_k = sum over the WHOLE array (ghosts included) of
(phi_k - phi_{k-1})^2 <= tol (rayleigh.py:448-454, mixing.py:457-463).  The increments obey d_{k+1} = J d_k with
J = cx (E + W) + cy (N + S) on the interior cells (a Neumann ghost copies its neighbour: the cell's own value comes back;
mixing's Dirichlet top ghost is 0), and err = d' W d with W = I + G, G = number of mirrored ghost sides of a cell (the ghost
copies are counted on top of the interior; corner ghosts are never set).  J is symmetric with |lambda| <= 1, so the PLAIN norm
d'd never grows -- but W does not commute with J, and err can:

    C_m = sup_d  err_{k+m} / err_k = || W^(1/2) J^m W^(-1/2) ||_2^2          (over increments with zero sum for the
                                                                               all-Neumann problem: the rhs is a divergence)

This script computes C_m by Lanczos (scipy svds on the matrix-free operator) for m = 1 .. 64 and prints max_m C_m per grid.
What the kernels use it for (include/beacon_hip.h, "conv_plan" 3): an evaluation of the residual that directly follows SKIPPED
sweeps proves that none of them passed the test only if it finds err > C tol -- had a skipped sweep j passed, err_j <= tol, then
err_k <= C_{k-j} err_j <= C tol.  The kernels' constant BCN_CONV_GUARD = 1.035 (csrc/bcn_common.h) must stay above every
figure of the reference's constructor space (rayleigh.py:26-27: nx = 50 L, ny = 50 H; mixing.py:27-28: 100 L, 100 H; L, H >= 1, so
no side is shorter than 50 cells): 1.0301 at most, whatever the aspect ratio of grid and cells.  Grids with a side below 48 cells
-- the boundary layers of two walls interact -- reach 1.041 (20x40, Dirichlet top); the library runs those under the proven plan 1
(capi.hip).  tests/test_oracle.py::test_weighted_norm_growth_bound_is_below_the_kernels_guard checks three grids.

The same quantity restricted to the SLOW modes of J (|lambda| >= 0.9 / 0.8) is what the kernels' second, much lower landing
threshold rests on (1.0002 / 1.0057 on 128x64 against 1.030 here): beacon_amd/stoprule.py has the derivation and computes the
per-grid constants; `--slow` prints them.

    python scripts/weighted_norm_bound.py                 the table of DESIGN.md (about two minutes)
    python scripts/weighted_norm_bound.py NX NY KIND      one grid (KIND 0 rayleigh, 1 mixing); optional CX (default 0.25)
    python scripts/weighted_norm_bound.py --slow NX NY KIND [CX]    the slow-mode constants of that grid (beacon_amd/stoprule.py)"""
import sys

import numpy as np
from scipy.sparse.linalg import LinearOperator, svds

GUARD = 1.035          # == BCN_CONV_GUARD in beacon_amd/csrc/bcn_common.h


def operators(nx, ny, kind, cx=0.25):
    """(J, sqrt(W), projector) on [nx][ny] arrays; kind 1: Dirichlet-zero ghost above the top row (mixing.py:450-451)."""
    cy = 0.5 - cx
    w = np.ones((nx, ny))
    w[0, :] += 1
    w[-1, :] += 1
    w[:, 0] += 1
    if kind == 0:
        w[:, -1] += 1

    def J(x):
        e = np.vstack([x[1:], x[-1:]])
        wv = np.vstack([x[:1], x[:-1]])
        n = np.hstack([x[:, 1:], x[:, -1:] if kind == 0 else 0 * x[:, -1:]])
        s = np.hstack([x[:, :1], x[:, :-1]])
        return cx * (e + wv) + cy * (n + s)
    one = np.ones((nx, ny)) / np.sqrt(nx * ny)
    P = (lambda x: x - one * (one * x).sum()) if kind == 0 else (lambda x: x)
    return J, np.sqrt(w), P


def growth(nx, ny, kind, m, cx=0.25, vectors=False):
    """C_m = || W^(1/2) J^m P W^(-1/2) ||_2^2 (and, with vectors=True, the increment d_k that attains it)."""
    J, sw, P = operators(nx, ny, kind, cx)

    def mv(v):
        x = P(v.reshape(nx, ny) / sw)
        for _ in range(m):
            x = J(x)
        return (sw * x).ravel()

    def rmv(v):
        x = v.reshape(nx, ny) * sw
        for _ in range(m):
            x = J(x)
        return (P(x) / sw).ravel()
    op = LinearOperator((nx * ny, nx * ny), matvec=mv, rmatvec=rmv, dtype=np.float64)
    if not vectors:
        return float(svds(op, k=1, return_singular_vectors=False, tol=1e-10)[0]) ** 2
    u, s, vt = svds(op, k=1, tol=1e-10)
    d = P(vt[0].reshape(nx, ny) / sw)
    return float(s[0]) ** 2, d


def bound(nx, ny, kind, cx=0.25, ms=(1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 24, 32, 48, 64)):
    c = [(growth(nx, ny, kind, m, cx), m) for m in ms]
    return max(c)


def main(argv):
    if argv and argv[0] == "--slow":
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from beacon_amd import stoprule
        nx, ny, kind = int(argv[1]), int(argv[2]), int(argv[3])
        cx = float(argv[4]) if len(argv) > 4 else 0.25
        for lc, b in stoprule.bounds(nx, ny, kind, cx, cache=False):
            print("%dx%d kind %d cx %.3f: within the modes |lambda| >= %.2f the norm grows by at most %.5f (global guard %.3f)" % (nx, ny, kind, cx, lc, b, GUARD))
        return 0
    if len(argv) >= 3:
        nx, ny, kind = int(argv[0]), int(argv[1]), int(argv[2])
        cx = float(argv[3]) if len(argv) > 3 else 0.25
        c, m = bound(nx, ny, kind, cx)
        print("%dx%d kind %d cx %.3f: max_m C_m = %.5f at m = %d   (guard %.3f)" % (nx, ny, kind, cx, c, m, GUARD))
        return 0 if c < GUARD else 1
    worst = 0.0
    rows = [(128, 64, 0), (50, 50, 0), (100, 100, 1), (100, 100, 0), (300, 50, 0), (448, 50, 0), (50, 150, 0), (50, 250, 0),
            (200, 100, 1), (100, 200, 1), (100, 256, 1), (110, 64, 0), (75, 50, 0), (60, 53, 0)]
    print("grid, kind (0 rayleigh: Neumann on four sides, zero-sum increments; 1 mixing: Dirichlet top)   max_m C_m   at m")
    for nx, ny, kind in rows:
        c, m = bound(nx, ny, kind)
        worst = max(worst, c)
        print("%4dx%-4d %d   %.5f   %d" % (nx, ny, kind, c, m), flush=True)
    print("anisotropic cells (dx != dy: nx = int(50 L) with 50 L not an integer), 50x50:")
    for cx in (0.2, 0.225, 0.24, 0.26, 0.275, 0.3, 0.35, 0.4):
        for kind in (0, 1):
            c, m = bound(50, 50, kind, cx)
            worst = max(worst, c)
            print("  cx = %.3f kind %d   %.5f   %d" % (cx, kind, c, m), flush=True)
    print("worst: %.5f; kernels' guard BCN_CONV_GUARD = %.3f (margin %.4f)" % (worst, GUARD, GUARD - worst))
    print("outside the reference's constructor space (a side below 48 cells: proven plan only):")
    for nx, ny, kind in ((40, 40, 1), (30, 30, 1), (18, 18, 1), (20, 40, 1), (25, 75, 0), (40, 120, 0), (16, 16, 0), (8, 8, 0)):
        c, m = bound(nx, ny, kind)
        print("%4dx%-4d %d   %.5f   %d" % (nx, ny, kind, c, m), flush=True)
    return 0 if worst < GUARD else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
