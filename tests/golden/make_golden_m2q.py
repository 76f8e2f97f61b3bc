#!/usr/bin/env python
"""Fixture for the branch of `matrix_to_quaternion` that LBS actually exercises: NON-orthonormal 3x3 matrices (blended joint rotations,
network/avatar.py:87-89 -> pytorch3d 0.7.4 `transforms/rotation_conversions.py`).

pytorch3d is not installable here, so this is NOT pytorch3d's output: it is an independent, scalar, float64 derivation of the published
0.7.4 algorithm written without torch or numpy broadcasting (plain Python `math`), for a table of matrices that exercises every arg-max
branch, the clamped-negative square roots and exact ties (the 0.1 floor of the divisor can never bind on the SELECTED row: the four t_x
below sum to 4, so the largest is >= 1 and its root >= 1 -- the floor only guards the three discarded candidates):

    t_r = 1 + m00 + m11 + m22,  t_i = 1 + m00 - m11 - m22,  t_j = 1 - m00 + m11 - m22,  t_k = 1 - m00 - m11 + m22
    a_x = sqrt(t_x) if t_x > 0 else 0                                  (_sqrt_positive_part)
    row = FIRST index of the largest a_x                               (torch.argmax returns the first maximum)
    q   = candidate[row] / (2 * max(a_row, 0.1))                       (no sign standardisation in 0.7.4)
    candidates: r: (a_r^2, m21 - m12, m02 - m20, m10 - m01)   i: (m21 - m12, a_i^2, m10 + m01, m02 + m20)
                j: (m02 - m20, m10 + m01, a_j^2, m12 + m21)   k: (m10 - m01, m20 + m02, m21 + m12, a_k^2)

    python tests/golden/make_golden_m2q.py      -> tests/golden/m2q_nonorthonormal.npz (M [n,3,3] float64, q [n,4] float64, row [n])
"""
import math
import os

import numpy as np


def m2q_scalar(M):
    m00, m01, m02 = M[0]
    m10, m11, m12 = M[1]
    m20, m21, m22 = M[2]
    t = [1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22]
    a = [math.sqrt(x) if x > 0.0 else 0.0 for x in t]
    row = 0
    for c in range(1, 4):
        if a[c] > a[row]:
            row = c
    cand = [
        (a[0] * a[0], m21 - m12, m02 - m20, m10 - m01),
        (m21 - m12, a[1] * a[1], m10 + m01, m02 + m20),
        (m02 - m20, m10 + m01, a[2] * a[2], m12 + m21),
        (m10 - m01, m20 + m02, m21 + m12, a[3] * a[3]),
    ][row]
    d = 2.0 * max(a[row], 0.1)
    return [c / d for c in cand], row


def rot(axis, ang):
    x, y, z = axis / np.linalg.norm(axis)
    c, s = math.cos(ang), math.sin(ang)
    C = 1 - c
    return np.array([[c + x * x * C, x * y * C - z * s, x * z * C + y * s],
                     [y * x * C + z * s, c + y * y * C, y * z * C - x * s],
                     [z * x * C - y * s, z * y * C + x * s, c + z * z * C]])


def main():
    rs = np.random.RandomState(20240924)
    mats = []
    # convex blends of 2-4 joint rotations (what sum_j w_j A_j produces), angles up to 170 degrees so that all four branches occur
    for _ in range(1200):
        k = rs.randint(2, 5)
        w = rs.dirichlet(np.ones(k))
        mats.append(sum(wi * rot(rs.standard_normal(3), rs.uniform(0, math.pi * 170 / 180)) for wi in w))
    # blends of near-opposite rotations: small, far-from-orthonormal matrices -> clamped square roots
    for _ in range(300):
        ax = rs.standard_normal(3)
        a = rs.uniform(2.6, math.pi)
        mats.append(0.5 * rot(ax, a) + 0.5 * rot(ax, -a) * rs.uniform(0.8, 1.0) + 0.02 * rs.standard_normal((3, 3)))
    for _ in range(200):
        mats.append(0.05 * rs.standard_normal((3, 3)))                 # nearly zero matrices: all four roots close to 1, near-ties
    # exact ties of the arg-max (first maximum wins) and exact zeros
    mats += [np.zeros((3, 3)), np.eye(3), -np.eye(3), np.diag([1.0, -1.0, -1.0]), np.diag([-1.0, 1.0, -1.0]), np.diag([-1.0, -1.0, 1.0]),
             np.diag([0.0, 0.0, 0.0]) + 1e-3, np.diag([1.0, 1.0, -1.0]), 0.5 * np.eye(3)]
    M = np.stack(mats).astype(np.float64)
    out = [m2q_scalar(m.tolist()) for m in M]
    q = np.array([o[0] for o in out], np.float64)
    row = np.array([o[1] for o in out], np.int64)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "m2q_nonorthonormal.npz"), M=M, q=q, row=row)
    print("wrote", len(M), "matrices; rows used:", np.bincount(row, minlength=4).tolist())


if __name__ == "__main__":
    main()
