"""Winograd / Cook-Toom transform matrices for the conv stacks' 3x3 and 5x5 ResNet convolutions.

F(m, r) computes m outputs of an r-tap correlation from n = m + r - 1 inputs with n multiplications:
    y = A^T [ (G g) * (B^T d) ]
F(4, 3) and F(2, 5) have n = 6 and are built on the same interpolation points {0, 1, -1, 2, -2, inf}, so they
share B^T (the input transform of csrc/net_epilogue.hip::k_wino_in) and differ in G (weights, applied once at
Model.fuse()) and A^T (k_wino_out); F(4, 5) has n = 8 and adds the points +-1/2.  2-D: U = G g G^T,
V = B^T d B, Y = A^T (U * V) A; summed over input channels the element-wise product is a GEMM per
transform position.  The matrices are constructed in Vandermonde form and B^T is solved from the
bilinear identity (float64, exact to 1e-14); F(4, 3) comes out as the familiar Lavin-Gray matrices.
"""
import numpy as np
import torch

POINTS = (0.0, 1.0, -1.0, 2.0, -2.0)                    # 6-point transforms: F(4,3), F(2,5)
POINTS8 = (0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5)        # 8-point transform: F(4,5)


def cook_toom(m, r, pts=POINTS):
    """-> (AT [m, n], G [n, r], BT [n, n]) float64 for the correlation y_o = sum_k d_{o+k} g_k."""
    n = m + r - 1
    assert len(pts) == n - 1
    AT, G = np.zeros((m, n)), np.zeros((n, r))
    for j, a in enumerate(pts):
        norm = np.prod([a - b for l, b in enumerate(pts) if l != j])
        AT[:, j] = [a ** i for i in range(m)]
        G[j, :] = [a ** k / norm for k in range(r)]
    AT[m - 1, n - 1] = 1.0
    G[n - 1, r - 1] = 1.0
    BT = np.zeros((n, n))
    rows = np.array([AT[o, :] * G[:, k] for o in range(m) for k in range(r)])
    for i in range(n):
        rhs = np.array([1.0 if i == o + k else 0.0 for o in range(m) for k in range(r)])
        BT[:, i] = np.linalg.lstsq(rows, rhs, rcond=None)[0]
    BT = np.round(BT * 64) / 64    # dyadic rationals for these points; the identity is re-checked below
    d, g = np.arange(1.0, n + 1) ** 1.5, np.cos(np.arange(r) + 0.3)
    want = np.array([sum(d[o + k] * g[k] for k in range(r)) for o in range(m)])
    assert np.abs(AT @ ((G @ g) * (BT @ d)) - want).max() < 1e-9 * np.abs(want).max()
    return AT, G, BT


# kernel size -> (tile size, tile stride) of the transform the conv stacks use
#   3x3: F(4x4, 3x3), 6x6 tiles;  5x5: F(4x4, 5x5), 8x8 tiles (F(2x2, 5x5) = (6, 2) is the alternative)
CONFIG = {3: (6, 4), 5: (8, 4)}


def tile_config(kernel_size, small_tiles=False):
    return (6, 2) if (kernel_size == 5 and small_tiles) else CONFIG[kernel_size]


def transform_weights(w, cfg=None):
    """w [Cout, Cin, r, r] (r = 3 or 5) -> U [ts*ts, Cout, Cin] float32, U[ts*i + j] = (G w G^T)[i, j],
    computed in float64."""
    r = w.shape[-1]
    ts, ms = cfg or tile_config(r)
    assert ts == ms + r - 1
    _, G, _ = cook_toom(ms, r, POINTS if ts == 6 else POINTS8)
    G = torch.from_numpy(G).to(w.device)
    U = torch.einsum("ik,ockl,jl->ijoc", G, w.double(), G)
    return U.reshape(ts * ts, w.shape[0], w.shape[1]).float().contiguous()
