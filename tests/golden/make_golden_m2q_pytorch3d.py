#!/usr/bin/env python
"""The pytorch3d pin of the LBS quaternion chain -- runs the moment the wheel is importable.

`network/avatar.py:87,89` calls pytorch3d 0.7.4's `quaternion_to_matrix` / `matrix_to_quaternion` (requirements.txt:9).  pytorch3d is not in this
image and cannot be installed (no network), so `tests/golden/m2q_nonorthonormal.npz` holds an independent float64 derivation of the published
algorithm (make_golden_m2q.py) and the row stays "parity unpinned".  This script closes the gap on any machine that HAS pytorch3d:

    python tests/golden/make_golden_m2q_pytorch3d.py

re-evaluates the SAME 1 709 matrices (and the quaternions of the q2m leg) with pytorch3d itself, writes
`tests/golden/m2q_pytorch3d.npz` (M, q_m2q, Q, R_q2m, pytorch3d.__version__) and prints how far the committed derivation is from it.
`tests/test_avatar_oracle_cpu.py::test_which_source_pins_the_quaternion_helpers` reports which of the two sources pinned the oracle in a run: with
the new file present the oracle's two helpers are asserted against pytorch3d's own numbers.  Without pytorch3d the script exits 3 and changes nothing."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main() -> int:
    try:
        import torch
        import pytorch3d
        from pytorch3d.transforms import matrix_to_quaternion, quaternion_to_matrix
    except Exception as e:                                     # noqa: BLE001 -- any import failure means "not on this machine"
        print(f"pytorch3d is not importable here ({e!r}): nothing written; row a5 stays pinned to the independent derivation only")
        return 3
    g = np.load(os.path.join(HERE, "m2q_nonorthonormal.npz"))
    M = torch.from_numpy(g["M"])
    q = matrix_to_quaternion(M).numpy()
    rs = np.random.RandomState(7)
    Q = rs.standard_normal((512, 4))                           # NOT normalised: avatar.py:87 feeds raw network outputs
    R = quaternion_to_matrix(torch.from_numpy(Q)).numpy()
    np.savez_compressed(os.path.join(HERE, "m2q_pytorch3d.npz"), M=g["M"], q_m2q=q, Q=Q, R_q2m=R, version=np.array(pytorch3d.__version__))
    d = np.abs(q - g["q"])
    print(f"pytorch3d {pytorch3d.__version__}: matrix_to_quaternion on the 1709 fixture matrices, max |pytorch3d - independent derivation| = {d.max():.3e} "
          f"({int((d > 1e-12).sum())} elements beyond 1e-12); wrote tests/golden/m2q_pytorch3d.npz")
    return 0


if __name__ == "__main__":
    sys.exit(main())
