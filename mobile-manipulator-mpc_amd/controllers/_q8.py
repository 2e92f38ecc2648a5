"""The half-space rows of controllers/mpc_wholebody_qref.py AS WRITTEN, evaluated on the host.

`obsAvoidConvex` (mpc_wholebody_qref.py:57-89) calls `subject_to` inside its loop over the planes and keeps ONE 6 x L
matrix `constr` for all stages (:156), so for L >= 2 planes it emits L rows per (stage k, arm sample point i):

    row (k, i, j):   -max( c_{k,i,0..j} ,  stale_{i,j+1..L-1} )  <=  s_k ,      c_{k,i,j} = n_j . ((pi_j - 0.03 n_j) - P_i(x_k))

where a stale entry is the free decision variable `constr[i, j']` at k = 0 (such a row can always be met by that variable and
never restricts (X, U, s)) and the PREVIOUS stage's expression c_{k-1,i,j'} for k >= 1.  Row j = L-1 is the intended row
-max_j c_{k,i,j} <= s_k.  The rows j < L-1 of the stages k >= 1 couple x_k with x_{k-1}; the generic kernel carries them
(mmpc_config.as_written: a row whose previous-stage entry attains the max acts on x_{k-1}, and a slack s_k that reaches back
is eliminated one stage earlier through the linearised dynamics, csrc/mmpc_core.h).  This module is their plain evaluation
on the host: the controller re-checks every returned solution against it (post-condition), the tests use it to show how the
intended and the as-written NLP differ.
"""
import numpy as np

A2, A3, A5, A6, A7 = 0.316, 0.0825, 0.384, 0.088, 0.107   # manipulator_3DoF.py:18-22
BX, BZ = -0.007, 0.606 + 0.333                            # mobile_manipulator.py:14-15
EXPAND = 0.03                                             # mpc_wholebody_qref.py:44
# sample points [j2/2, j2, (j2+j3)/2, j3, (j3+e)/2, e] (:216-217) as coefficients of (j2, j3, e)
POINTS = np.array([[0.5, 0, 0], [1, 0, 0], [0.5, 0.5, 0], [0, 1, 0], [0, 0.5, 0.5], [0, 0, 1]], float)


def sample_points(X):
    """World positions of the six arm sample points for states X (..., 9) -> (..., 6, 3)  (mobile_manipulator.py:36-53)."""
    X = np.asarray(X, float)
    q1, q2, q3 = X[..., 6], X[..., 7], X[..., 8]
    a, b = q1 - q2, q1 - q2 - q3
    r2 = A2 * np.sin(q1) + A3 * np.cos(q1); z2 = A2 * np.cos(q1) - A3 * np.sin(q1)
    r3 = r2 - A3 * np.cos(a) + A5 * np.sin(a); z3 = z2 + A3 * np.sin(a) + A5 * np.cos(a)
    re = r3 + A6 * np.cos(b) - A7 * np.sin(b); ze = z3 - A6 * np.sin(b) - A7 * np.cos(b)
    c, s = np.cos(X[..., 2]), np.sin(X[..., 2])

    def world(r, z):
        return np.stack([X[..., 0] + (r + BX) * c, X[..., 1] + (r + BX) * s, z + BZ], axis=-1)
    pts = np.stack([world(r2, z2), world(r3, z3), world(re, ze)], axis=-2)        # (..., 3, 3): j2, j3, e
    return np.einsum("ip,...pc->...ic", POINTS, pts)


def plane_values(X, hs):
    """c[..., i, j] = n_j . ((pi_j - 0.03 n_j) - P_i(x)) for states X (..., 9) and planes hs (L, 6) = (point, normal)."""
    hs = np.asarray(hs, float).reshape(-1, 6)
    P = sample_points(X)                                                           # (..., 6, 3)
    shifted = hs[:, :3] - EXPAND * hs[:, 3:]                                       # (L, 3)
    return np.einsum("jc,jc->j", hs[:, 3:], shifted) - np.einsum("...ic,jc->...ij", P, hs[:, 3:])


def as_written_extra_rows(X, s, hs):
    """Values  -max(c_{k,i,0..j}, c_{k-1,i,j+1..L-1}) - s_k  of the extra rows (k = 1..N, i = 0..5, j = 0..L-2) for
    trajectories X (..., N+1, 9), slacks s (..., N+1): array (..., N, 6, L-1); <= 0 means satisfied."""
    c = plane_values(X, hs)                                                        # (..., N+1, 6, L)
    L = c.shape[-1]
    if L < 2:
        return np.zeros(c[..., 1:, :, :0].shape)
    out = []
    for j in range(L - 1):
        cur = c[..., 1:, :, :j + 1].max(axis=-1)
        stale = c[..., :-1, :, j + 1:].max(axis=-1)
        out.append(-np.maximum(cur, stale) - np.asarray(s, float)[..., 1:, None])
    return np.stack(out, axis=-1)
