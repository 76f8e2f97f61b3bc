"""Posed-Gaussian PLY export / import -- SURVEY.md §8(f)-4; mirrors ``gaussians/obj_io.py:9-100`` (called per test frame at
``main_avatar.py:768``) without the ``plyfile`` package: the standard 3DGS vertex layout, binary little-endian, 62 floats per
Gaussian -- ``x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3`` -- with the reference's conventions: colours are
swapped to R, G, B and stored as the degree-0 SH coefficient ``(c - 0.5) / C0``, opacity as its logit, scales as logarithms, normals
and higher SH bands zero.  ``plyfile`` is not in this image: byte-level parity with its writer is unpinned (the header below is
what ``PlyData([PlyElement.describe(float32 structured array, 'vertex')]).write`` emits); values round-trip (tests/test_formats_cpu.py)."""
from __future__ import annotations

import os

import numpy as np
import torch

C0 = 0.28209479177387814                     # utils/sh_utils.py:26
_NAMES = (['x', 'y', 'z', 'nx', 'ny', 'nz'] + [f'f_dc_{i}' for i in range(3)] + [f'f_rest_{i}' for i in range(45)] + ['opacity']
          + [f'scale_{i}' for i in range(3)] + [f'rot_{i}' for i in range(4)])


def save_gaussians_as_ply(path: str, gaussian_vals: dict) -> None:
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    f32 = lambda t: t.detach().to('cpu', torch.float32).numpy()  # noqa: E731
    xyz = f32(gaussian_vals['positions'])
    n = xyz.shape[0]
    out = np.zeros((n, len(_NAMES)), '<f4')
    out[:, 0:3] = xyz
    out[:, 6:9] = (f32(gaussian_vals['colors'])[:, [2, 1, 0]] - 0.5) / C0                       # RGB2SH of the swapped colours
    op = f32(gaussian_vals['opacity']).reshape(n, 1)
    out[:, 54:55] = np.log(op / (1 - op))                                                          # inverse_sigmoid
    out[:, 55:58] = np.log(f32(gaussian_vals['scales']))
    out[:, 58:62] = f32(gaussian_vals['rotations'])
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join(f"property float {k}\n" for k in _NAMES) + "end_header\n"
    with open(path, 'wb') as f:
        f.write(header.encode('ascii'))
        f.write(out.tobytes())


def load_gaussians_from_ply(path: str, device="cuda") -> dict:
    with open(path, 'rb') as f:
        buf = f.read()
    end = buf.index(b'end_header\n') + len(b'end_header\n')
    lines = buf[:end].decode('ascii').split('\n')
    if lines[0] != 'ply' or not lines[1].startswith('format binary_little_endian'):
        raise ValueError(f"{path}: only binary little-endian PLY files are read")
    n, props, in_vertex = 0, [], False
    for ln in lines[2:]:
        tok = ln.split()
        if tok[:1] == ['element']:
            in_vertex = tok[1] == 'vertex'
            if in_vertex:
                n = int(tok[2])
        elif tok[:1] == ['property'] and in_vertex:
            if tok[1] not in ('float', 'float32'):
                raise ValueError(f"{path}: property {tok[2]} is {tok[1]}, expected float")
            props.append(tok[2])
    a = np.frombuffer(buf, '<f4', n * len(props), end).reshape(n, len(props))
    col = lambda k: a[:, props.index(k)]  # noqa: E731
    by_idx = lambda pre: sorted((p for p in props if p.startswith(pre)), key=lambda s: int(s.split('_')[-1]))  # noqa: E731
    xyz = np.stack([col('x'), col('y'), col('z')], 1)
    dc = np.stack([col('f_dc_0'), col('f_dc_1'), col('f_dc_2')], 1)
    extra = np.stack([col(k) for k in by_idx('f_rest_')], 1).reshape(n, 3, 15)
    scales = np.stack([col(k) for k in by_idx('scale_')], 1)
    rots = np.stack([col(k) for k in by_idx('rot')], 1)
    t = lambda x: torch.tensor(np.ascontiguousarray(x), dtype=torch.float, device=device)  # noqa: E731
    return {
        'positions': t(xyz),
        'colors': t((dc * C0 + 0.5)[:, [2, 1, 0]]),
        'opacity': torch.sigmoid(t(col('opacity')[:, None])),
        'scales': torch.exp(t(scales)),
        'rotations': torch.nn.functional.normalize(t(rots)),
        'features_extr': t(extra),
    }
