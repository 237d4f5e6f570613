"""Closed-form known answers of the reference's HPCG generator + solveCG, at ANY size and rank count.

The reference generates the 27-point problem with diagonal 27, off-diagonals -1, neighbours clipped at the
grid boundary (src/matrix.c:30-121: P bricks of nx x ny x nz stacked in z form one nx x ny x (nz P) grid) and
b = 27 - (nnzrow - 1) = 28 - nnzrow, x0 = 0 (src/CGSolver.c:25-36).  With c(i) = cx cy cz the number of grid
points in the clipped 3x3x3 box around point i (c_d = 2 at a boundary, 3 inside):

  r.r of the prologue (src/CGSolver.c:97-98, r0 = b)      = sum_i (28 - c_i)^2
  p.Ap of the first loop body (:109, :123-125, p = b)      = b' A b = 28 sum_i b_i^2 - sum_i sum_{j in box(i)} b_i b_j

Both sums factor over the three dimensions, every term is an integer far below 2^53, so ANY summation order
gives the exact value in fp64: the GPU path must reproduce them exactly whatever its dot order, rank count or
format.  p.Ap of the first body involves one halo exchange of p = b, so it also checks that every neighbour's
values arrived (a stale or misplaced halo entry changes it: b differs between corner, edge and face rows).

BASELINE.md section 3 lists the values the reference itself prints / the survey probed (8^3 ... 128^3, 16^3 x 4
ranks); tests/test_known_answers.py pins these formulas on them and on the oracle's P-rank runs.
Used by bench.py's pre-flight check and the tests; plain integer arithmetic, no dependency.
"""


def _line(n):
    """1-D pieces: S = sum c, Q = sum c^2, T = sum_i sum_{j in nb(i)} c_i c_j (nb includes i) for a line of n points"""
    c = [1] * n if n == 1 else [2 if i in (0, n - 1) else 3 for i in range(n)]
    S = sum(c)
    Q = sum(v * v for v in c)
    T = sum(c[i] * c[j] for i in range(n) for j in range(max(0, i - 1), min(n, i + 2)))
    return S, Q, T


def hpcg_sums(nx, ny, nz_total):
    (Sx, Qx, Tx), (Sy, Qy, Ty), (Sz, Qz, Tz) = _line(nx), _line(ny), _line(nz_total)
    return nx * ny * nz_total, Sx * Sy * Sz, Qx * Qy * Qz, Tx * Ty * Tz


def hpcg_nnz(nx, ny, nz_total):
    """true number of nonzeros = sum_i c_i = (3nx-2)(3ny-2)(3nz-2)"""
    return hpcg_sums(nx, ny, nz_total)[1]


def hpcg_rr0(nx, ny, nz_total):
    """r.r of the prologue: sum (28 - c)^2 = 784 N - 56 S + Q; for a cube: (n-2)^3 + 600(n-2)^2 + 3072(n-2) + 3200"""
    N, S, Q, _ = hpcg_sums(nx, ny, nz_total)
    return 784 * N - 56 * S + Q


def hpcg_pAp1(nx, ny, nz_total):
    """p.Ap of the first loop body (p = r = b)"""
    N, S, Q, T = hpcg_sums(nx, ny, nz_total)
    sum_b2 = 784 * N - 56 * S + Q
    box = 784 * S - 56 * Q + T  # sum_i sum_{j in box(i)} (28 - c_i)(28 - c_j)
    return 28 * sum_b2 - box
