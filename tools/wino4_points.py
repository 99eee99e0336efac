"""Interpolation points for Winograd F(4x4,3x3) in fp32: error of the whole convolution (transforms in fp32, U = G g G^T rounded once from
fp64, fp32 accumulation over the channels) against an fp64 convolution, for symmetric point sets 0, +-a, +-b, infinity.  CPU only (numpy).
The kernels (vspbfr_amd/csrc/conv_wino4.hip) use a = 3/4, b = 3/2: every constant of B^T and A^T is a dyadic rational."""
import sys
from fractions import Fraction as Fr

import numpy as np


def matrices(points, m=4, r=3):
    """A^T (m x n), G (n x r), B^T (n x n) of the Toom-Cook construction for n - 1 finite points + infinity (n = m + r - 1)."""
    n = m + r - 1
    P = [Fr(p) for p in points]
    AT = [[(P[j] ** i if j < n - 1 else (Fr(1) if i == m - 1 else Fr(0))) for j in range(n)] for i in range(m)]
    G = []
    for j in range(n - 1):
        N = Fr(1)
        for k in range(n - 1):
            if k != j:
                N *= P[j] - P[k]
        G.append([P[j] ** k / N for k in range(r)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    A = np.array([[float(AT[i][j] * G[j][k]) for j in range(n)] for i in range(m) for k in range(r)])
    BT = np.zeros((n, n))
    for col in range(n):     # sum_j AT[i][j] G[j][k] BT[j][l] = [l == i + k]
        rhs = np.array([1.0 if col == i + k else 0.0 for i in range(m) for k in range(r)])
        sol = np.linalg.lstsq(A, rhs, rcond=None)[0]
        assert np.abs(A @ sol - rhs).max() < 1e-9
        BT[:, col] = sol
    BT = np.array([[float(Fr(x).limit_denominator(4096)) for x in row] for row in BT])
    f = lambda M: np.array([[float(x) for x in row] for row in M])
    return f(AT), f(G), BT


def conv_error(AT, G, BT, m, C=256, Co=8, T=64, seed=1):
    rng = np.random.default_rng(seed)
    n = BT.shape[0]
    d = rng.standard_normal((C, T, n, n))
    g = rng.standard_normal((Co, C, 3, 3)) / np.sqrt(C * 9)
    ref = np.zeros((Co, T, m, m))
    for i in range(m):
        for j in range(m):
            ref[:, :, i, j] = np.einsum("ocab,ctab->ot", g, d[:, :, i:i + 3, j:j + 3])
    f32 = np.float32
    U = np.einsum("ia,ocab,jb->ocij", G, g, G).astype(f32)
    V = np.einsum("ia,ctab->ctib", BT.astype(f32), d.astype(f32)).astype(f32)
    V = np.einsum("ctib,jb->ctij", V, BT.astype(f32)).astype(f32)
    M = np.zeros((Co, T, n, n), dtype=f32)
    for c in range(C):
        M += (U[:, c, None] * V[None, c]).astype(f32)
    Y = np.einsum("ia,otab->otib", AT.astype(f32), M).astype(f32)
    Y = np.einsum("otib,jb->otij", Y, AT.astype(f32)).astype(f32)
    e = Y - ref
    return np.abs(e).max(), np.sqrt((e ** 2).mean())


if __name__ == "__main__":
    sets = [(1, 2), (Fr(1, 2), Fr(3, 2)), (Fr(5, 8), Fr(3, 2)), (Fr(3, 4), Fr(3, 2)), (Fr(3, 4), Fr(7, 4)), (Fr(1, 2), 2)]
    for a, b in sets:
        AT, G, BT = matrices((0, a, -a, b, -b))
        mx, rms = conv_error(AT, G, BT, 4)
        print(f"points 0, +-{a}, +-{b}, inf: max {mx:.2e} rms {rms:.2e}")
    if "-v" in sys.argv:
        np.set_printoptions(linewidth=160, precision=8, suppress=True)
        AT, G, BT = matrices((0, Fr(3, 4), -Fr(3, 4), Fr(3, 2), -Fr(3, 2)))
        print("B^T\n", BT, "\nG\n", G, "\nA^T\n", AT)
