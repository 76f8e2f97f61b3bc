"""Seeded synthetic inputs for the render hot path (no dataset or checkpoint can be fetched here).

Two scene families, both from ``numpy.random.RandomState(31359)`` (the reference's seed,
``main_avatar.py:817``; numpy's legacy generator is bit-stable across versions and machines):

* ``random_gaussians``  - BASELINE.json configs[0]: P random Gaussians in a body-sized box, one 512x512 front
  camera (f = 550, c = 256: the dataset default, ``dataset/dataset_mv_rgb.py:216-218``).
* ``avatar_map_gaussians`` - configs[1]/[4]: Gaussians laid out like the reference's canonical front|back position
  map (``gen_data/gen_pos_maps.py:42,61,112-113``): one Gaussian per set pixel of a (S x 2S) silhouette mask,
  S = 1024 -> ~234 k Gaussians on a 2 mm grid; free-view cameras at f = 1100, 1024x1024, subject 2.5 m away
  (``main_avatar.py:601-615``).
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np

from . import camera as cam

f32 = np.float32
SEED = 31359


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def random_gaussians(P: int = 10000, seed: int = SEED, img: int = 512, focal: float = 550.0) -> Dict[str, object]:
    rs = np.random.RandomState(seed)
    means3D = np.stack([rs.uniform(-0.5, 0.5, P), rs.uniform(-0.9, 0.9, P), rs.uniform(-0.2, 0.2, P)], 1).astype(f32)
    scales = np.exp(rs.normal(math.log(0.01), 0.3, (P, 3))).astype(f32)
    q = rs.normal(0, 1, (P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q *= rs.uniform(0.8, 1.2, (P, 1))          # deliberately NOT unit: the rasterizer uses raw quaternions
    rotations = q.astype(f32)
    opacities = _sigmoid(rs.normal(0, 1.5, (P, 1))).astype(f32)
    colors = rs.uniform(0, 1, (P, 3)).astype(f32)
    bg = rs.uniform(0, 1, 3).astype(f32)
    extr = cam.calc_front_mv(np.zeros(3, f32), tar_pos=(0.0, 0.0, 2.5))
    intr = np.array([[focal, 0, img / 2], [0, focal, img / 2], [0, 0, 1]], f32)
    scene = {
        "means3D": means3D, "scales": scales, "rotations": rotations, "opacities": opacities, "colors": colors,
        "bg": bg, "extr": extr, "intr": intr, "img_w": img, "img_h": img,
    }
    scene.update(upstream_grads(img, img, seed + 1))
    return scene


def upstream_grads(W: int, H: int, seed: int) -> Dict[str, np.ndarray]:
    rs = np.random.RandomState(seed)
    return {
        "dL_dcolor": rs.normal(0, 1, (3, H, W)).astype(f32),
        "dL_ddepth": rs.normal(0, 1, (1, H, W)).astype(f32),
        "dL_dalpha": rs.normal(0, 1, (1, H, W)).astype(f32),
    }


def body_mask(S: int = 1024) -> np.ndarray:
    """(S, S) bool silhouette: union of ellipses (head, torso, two arms in A-pose, two legs)."""
    v, u = np.mgrid[0:S, 0:S].astype(np.float64)
    x = (u + 0.5) / S * 2.0 - 1.0       # [-1, 1] across the map
    y = (v + 0.5) / S * 2.0 - 1.0       # top (-1) -> bottom (+1)

    def ell(cx, cy, rx, ry, ang=0.0):
        c, s = math.cos(ang), math.sin(ang)
        xr = (x - cx) * c + (y - cy) * s
        yr = -(x - cx) * s + (y - cy) * c
        return (xr / rx) ** 2 + (yr / ry) ** 2 <= 1.0

    m = ell(0.0, -0.78, 0.085, 0.11)                     # head
    m |= ell(0.0, -0.36, 0.17, 0.30)                     # torso
    m |= ell(-0.33, -0.36, 0.056, 0.30, math.radians(-38))   # arms
    m |= ell(0.33, -0.36, 0.056, 0.30, math.radians(38))
    m |= ell(-0.10, 0.42, 0.075, 0.46, math.radians(4))   # legs
    m |= ell(0.10, 0.42, 0.075, 0.46, math.radians(-4))
    return m


def avatar_map_gaussians(S: int = 1024, seed: int = SEED) -> Dict[str, object]:
    """Canonical Gaussians on an (S x 2S) front|back map.  Pixel pitch = 2.048 m / S."""
    rs = np.random.RandomState(seed)
    m = body_mask(S)
    mask = np.concatenate([m, m[:, ::-1]], axis=1)            # back half is the mirrored silhouette
    pitch = 2.048 / S
    vv, uu = np.nonzero(mask)                                  # row-major == the reference's boolean-mask order
    front = uu < S
    ul = np.where(front, uu, 2 * S - 1 - uu).astype(np.float64)
    x = (ul + 0.5 - S / 2) * pitch
    y = (S / 2 - (vv + 0.5)) * pitch
    bump = 0.10 * np.cos(np.clip(x / 0.45, -1, 1) * math.pi / 2)       # thickness profile
    z = np.where(front, bump, -bump)
    means3D = np.stack([x, y, z], 1).astype(f32)
    P = means3D.shape[0]
    base = math.log(pitch)                                     # ~mean 3-NN distance (create_from_pcd)
    scales = np.exp(base + rs.normal(0, 0.2, (P, 3))).astype(f32)
    q = rs.normal(0, 1, (P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    rotations = q.astype(f32)
    opacities = _sigmoid(rs.normal(2.0, 1.0, (P, 1))).astype(f32)
    colors = rs.uniform(0, 1, (P, 3)).astype(f32)
    bg = np.array([1.0, 1.0, 1.0], f32)
    return {
        "mask": mask, "means3D": means3D, "scales": scales, "rotations": rotations,
        "opacities": opacities, "colors": colors, "bg": bg,
    }


def free_view_cameras(n_views: int = 8, img: int = 1024, focal: float = 1100.0, dist: float = 2.5) -> List[Dict[str, object]]:
    cams = []
    for i in range(n_views):
        rot_Y = i / float(n_views) * 2.0 * math.pi
        extr = cam.calc_free_mv(np.zeros(3, f32), tar_pos=(0.0, 0.0, dist), rot_Y=rot_Y)
        intr = np.array([[focal, 0, img / 2], [0, focal, img / 2], [0, 0, 1]], f32)
        cams.append({"extr": extr, "intr": intr, "img_w": img, "img_h": img})
    return cams


def named_fill(state: Dict[str, object], seed: int = SEED) -> Dict[str, object]:
    """Deterministic synthetic values for a network's tensors, a pure function of (name, shape, seed): the same
    numbers can be produced wherever the names are known (the reference module in the build container, this package
    on the GPU box), which is how the StyleUNet parity fixture pins a 74M-parameter network without shipping weights.

    Scales follow the reference's initialisers (randn weights, randn/lr_mul mapping weights come out of the shapes
    alone), with small non-zero biases / noise strengths so that every path contributes."""
    import zlib

    import torch

    out = {}
    for name, t in state.items():
        if name.endswith((".kernel", ".ll", ".lh", ".hl", ".hh")):
            continue
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
        v = torch.randn(tuple(t.shape), generator=g, dtype=torch.float32)
        if name.startswith("style.") and name.endswith(".weight"):
            v = v * 100.0                                  # randn / lr_mlp
        elif name.endswith("modulation.bias"):
            v = 1.0 + 0.1 * v
        elif name.endswith(".bias") or name.endswith("noise.weight"):
            v = 0.1 * v
        out[name] = v
    return out


def pose_map(S: int = 512, seed: int = SEED):
    """Smooth synthetic position map [1, 3, S, S] inside a body-shaped mask (stand-in for smpl_pos_map)."""
    import torch

    rng = np.random.default_rng(seed + 77)
    yy, xx = np.meshgrid(np.linspace(-1, 1, S, dtype=np.float32), np.linspace(-1, 1, S, dtype=np.float32), indexing="ij")
    chans = []
    for c in range(3):
        a = rng.uniform(0.5, 3.0, 4).astype(np.float32)
        chans.append(0.5 * np.sin(a[0] * xx + a[1]) * np.cos(a[2] * yy + a[3]) + 0.1 * xx * (c - 1))
    m = (np.abs(xx) < 0.8) & (np.abs(yy) < 0.9)
    return torch.from_numpy((np.stack(chans) * m[None]).astype(np.float32))[None]


# SMPL-X kinematic tree (55 joints: pelvis, 21 body, jaw, 2 eyes, 15 + 15 finger joints) as the model files' kintree_table[0]
# holds it; any tree with parents[j] < j exercises the same code.
SMPLX_PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
                 20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
                 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53)


def smplx_model_arrays(seed: int = SEED, V: int = 10475, shape_dims: int = 400, n_faces: int = 20908) -> Dict[str, np.ndarray]:
    """A synthetic body model with the ARRAY NAMES, SHAPES AND DTYPES of `SMPLX_NEUTRAL.npz` (the licensed model files are
    not redistributable and not in this image): what smplx/body_models.py:237-260,604-625,995-1073 read from the file.
    Random but body-sized: vertices in a 0.6 x 1.7 x 0.3 m box, 1-cm shape / 2-mm pose-corrective bases, each joint a convex
    combination of 32 vertices, 4 skinning weights per vertex."""
    rng = np.random.default_rng(seed)
    J = len(SMPLX_PARENTS)
    m: Dict[str, np.ndarray] = {}
    m['v_template'] = ((rng.random((V, 3)) - 0.5) * np.array([0.6, 1.7, 0.3])).astype(np.float64)
    m['shapedirs'] = (rng.standard_normal((V, 3, shape_dims)) * 0.01).astype(np.float64)
    m['posedirs'] = (rng.standard_normal((V, 3, 9 * (J - 1))) * 0.002).astype(np.float64)
    Jr = np.zeros((J, V))
    for j in range(J):
        idx = rng.choice(V, 32, replace=False)
        w = rng.random(32)
        Jr[j, idx] = w / w.sum()
    m['J_regressor'] = Jr
    kin = np.zeros((2, J), np.int64)
    kin[0] = SMPLX_PARENTS
    kin[0, 0] = 2 ** 32 - 1                      # the files store the root's parent as uint32(-1); smplx overwrites it
    kin[1] = np.arange(J)
    m['kintree_table'] = kin
    W = np.zeros((V, J))
    for k, idx in enumerate(rng.integers(0, J, (4, V))):
        W[np.arange(V), idx] += rng.random(V) + (1.0 if k == 0 else 0.0)
    m['weights'] = W / W.sum(1, keepdims=True)
    m['f'] = rng.integers(0, V, (n_faces, 3)).astype(np.uint32)
    m['hands_componentsl'] = rng.standard_normal((45, 45)) * 0.3
    m['hands_componentsr'] = rng.standard_normal((45, 45)) * 0.3
    m['hands_meanl'] = rng.standard_normal(45) * 0.2
    m['hands_meanr'] = rng.standard_normal(45) * 0.2
    m['lmk_faces_idx'] = rng.integers(0, n_faces, 51).astype(np.int64)
    b = rng.random((51, 3))
    m['lmk_bary_coords'] = b / b.sum(1, keepdims=True)
    return m


def smplx_pose_params(seed: int = SEED, n: int = 1) -> Dict[str, np.ndarray]:
    """`n` frames of the parameter arrays a dataset's smpl_params.npz holds (dataset_mv_rgb.py:44-45)."""
    rng = np.random.default_rng(seed + 17)
    f32 = np.float32
    return {
        'betas': (rng.standard_normal((1, 10)) * 0.8).astype(f32),
        'global_orient': (rng.standard_normal((n, 3)) * 0.6).astype(f32),
        'transl': (rng.standard_normal((n, 3)) * 0.5).astype(f32),
        'body_pose': (rng.standard_normal((n, 63)) * 0.35).astype(f32),
        'jaw_pose': (rng.standard_normal((n, 3)) * 0.1).astype(f32),
        'expression': (rng.standard_normal((n, 10)) * 0.7).astype(f32),
        'left_hand_pose': (rng.standard_normal((n, 45)) * 0.3).astype(f32),
        'right_hand_pose': (rng.standard_normal((n, 45)) * 0.3).astype(f32),
    }
