"""The fp64 MFMA tile Cholesky (chol_mfma64_kernel, csrc/rbd_kernels.hip) restated lane by lane in numpy — TEST INFRASTRUCTURE, no GPU.
v_mfma_f64_4x4x4f64 is emulated with the lane layout probed on gfx950 (scripts/ubench/mfma_f64_4x4x4.hip: lane = 16 x + 4 block + y;
D[i][j] at (x = i, y = j), B[k][j] at (x = k, y = j), A[i][k] at (x = k, y = i)), and the kernel's steps are followed one for one: transposed
tiles, W = (L_d^-1)' panel, trailing update with the negated panel tile, diagonal tile gathered to its 16 lanes, substitutions in the chained
X / Y vector forms.  What this pins without hardware is the ALGEBRA of that arrangement (which operand must be transposed where, which sum
runs over which lane coordinate) against numpy's Cholesky and solve; the GPU tests pin the kernel itself."""
import numpy as np
import pytest

LANE = np.arange(64)
X, BLK, Y = LANE >> 4, (LANE >> 2) & 3, LANE & 3


def mfma(a, b, c):
    """D = A B + C on 4 blocks; a, b, c, result: one value per lane."""
    d = np.array(c, float)
    for lane in range(64):
        i, blk, j = X[lane], BLK[lane], Y[lane]
        for k in range(4):
            d[lane] += a[16 * k + 4 * blk + i] * b[16 * k + 4 * blk + j]  # A[i][k] at (x = k, y = i); B[k][j] at (x = k, y = j)
    return d


def sum_over_y(v):
    return v.reshape(4, 4, 4).sum(axis=2, keepdims=True).repeat(4, axis=2).reshape(64)


def sum_over_x(v):
    return v.reshape(4, 4, 4).sum(axis=0, keepdims=True).repeat(4, axis=0).reshape(64)


def tile_cholesky(M, b, nv):
    """M: (4, nv, nv) lower triangles of 4 states, b: (4, nv) -> (L (4, nv, nv), x (4, nv))."""
    NT = (nv + 3) // 4
    t = {}
    for I in range(NT):
        for J in range(I + 1):
            row, col = 4 * I + Y, 4 * J + X
            if I == J:
                row, col = np.maximum(row, col), np.minimum(row, col)  # only the lower triangle is there to read
            val = np.where(row == col, 1.0, 0.0)
            ok = row < nv
            val[ok] = M[BLK[ok], row[ok], col[ok]]
            t[I, J] = val
    bY = [np.where(4 * I + Y < nv, b[BLK, np.minimum(4 * I + Y, nv - 1)], 0.0) for I in range(NT)]
    linv = {}
    for J in range(NT):
        D = np.zeros((4, 4, 4))
        D[BLK, X, Y] = t[J, J]                      # the gather: element (x, y) of every state's diagonal tile
        W = np.zeros(64)
        for blk in range(4):
            Ld = np.linalg.cholesky(np.tril(D[blk]) + np.tril(D[blk], -1).T)
            Li = np.linalg.inv(Ld)
            m = BLK == blk
            t[J, J][m] = Ld[Y[m], X[m]]             # Tt(J,J) = L_d'
            linv.setdefault(J, np.zeros(64))[m] = Li[X[m], Y[m]]
            W[m] = Li[Y[m], X[m]]                   # (L_d^-1)'
        for I in range(J + 1, NT):
            t[I, J] = mfma(W, t[I, J], np.zeros(64))               # Xt_I = L_d^-1 Tt(I,J)
        for Jp in range(J + 1, NT):
            for I in range(Jp, NT):
                t[I, Jp] = mfma(-t[Jp, J], t[I, J], t[I, Jp])      # Tt(I,J') -= X_J' X_I'
    vX = {}
    for I in range(NT):
        part = sum((t[I, J] * vX[J] for J in range(I)), np.zeros(64))
        rhs = bY[I] - sum_over_x(part)
        vX[I] = sum_over_y(linv[I] * rhs)
    xY = {}
    for J in range(NT - 1, -1, -1):
        part = sum((t[I, J] * xY[I] for I in range(J + 1, NT)), np.zeros(64))
        rhs = vX[J] - sum_over_y(part)
        xY[J] = sum_over_x(linv[J] * rhs)
    L = np.zeros((4, nv, nv)); x = np.zeros((4, nv))
    for I in range(NT):
        for J in range(I + 1):
            row, col = 4 * I + Y, 4 * J + X
            ok = (row < nv) & (col <= row)
            L[BLK[ok], row[ok], col[ok]] = t[I, J][ok]
        ok = (X == 0) & (4 * I + Y < nv)
        x[BLK[ok], 4 * I + Y[ok]] = xY[I][ok]
    return L, x


@pytest.mark.parametrize("nv", [2, 7, 12, 36])
def test_transposed_tile_cholesky_on_the_probed_mfma_layout(nv):
    rng = np.random.default_rng(nv)
    A = rng.standard_normal((4, nv, nv))
    A = A @ A.transpose(0, 2, 1) + nv * np.eye(nv)
    b = rng.standard_normal((4, nv))
    L, x = tile_cholesky(np.tril(A), b, nv)
    assert np.abs(L - np.linalg.cholesky(A)).max() <= 1e-12 * np.abs(A).max() ** 0.5
    xr = np.linalg.solve(A, b[..., None])[..., 0]
    assert np.abs(x - xr).max() <= 1e-11 * max(1.0, np.abs(xr).max())
