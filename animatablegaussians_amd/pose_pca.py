"""PCA projection of the pose condition -- SURVEY.md §8(f)-4; ``dataset/dataset_mv_rgb.py:287-321``.

For animation with out-of-distribution poses the reference projects the front position map (its valid pixels, flattened) onto the
first ``n_components`` principal directions of the training poses, clamps the coefficients to ``sigma_pca`` standard deviations and
reconstructs (``transform_pca``).  The reference uses scikit-learn's ``PCA`` on the CPU (``fit`` = centre + SVD, components with the
deterministic sign of ``svd_flip``); this class keeps the same attributes (``mean_``, ``components_``, ``explained_variance_``) as
torch tensors on the device of the maps and evaluates ``transform_pca`` there: two skinny library GEMVs over the ``[n_components,
3 * pixels]`` basis (rocBLAS through torch -- plain library GEMVs, not a hot-path kernel).  Checked against scikit-learn in
tests/test_formats_cpu.py."""
from __future__ import annotations

import torch


class PosePCA:
    def __init__(self, n_components: int = 10):
        self.n_components = n_components
        self.mean_ = self.components_ = self.explained_variance_ = None

    def fit(self, pose_conds: torch.Tensor) -> "PosePCA":
        """``pose_conds`` [n_poses, D]: sklearn.decomposition.PCA(n_components).fit (full SVD path, svd_flip(u_based_decision=False))."""
        X = pose_conds.to(torch.float64)
        self.mean_ = X.mean(0)
        U, S, Vt = torch.linalg.svd(X - self.mean_, full_matrices=False)
        # sign convention of sklearn.utils.extmath.svd_flip on V: the largest-magnitude entry of every component is positive
        idx = Vt.abs().argmax(1)
        sign = torch.sign(Vt[torch.arange(Vt.shape[0]), idx])
        Vt = Vt * sign[:, None]
        k = self.n_components
        self.components_ = Vt[:k].to(pose_conds.dtype)
        self.explained_variance_ = ((S ** 2) / (X.shape[0] - 1))[:k].to(pose_conds.dtype)
        self.mean_ = self.mean_.to(pose_conds.dtype)
        return self

    def to(self, device) -> "PosePCA":
        self.mean_, self.components_, self.explained_variance_ = (t.to(device) for t in (self.mean_, self.components_, self.explained_variance_))
        return self

    def transform(self, x: torch.Tensor) -> torch.Tensor:
        return (x - self.mean_) @ self.components_.T

    def inverse_transform(self, low: torch.Tensor) -> torch.Tensor:
        return low @ self.components_ + self.mean_

    def transform_pca(self, pose_conds: torch.Tensor, sigma_pca: float = 2.) -> torch.Tensor:
        """dataset_mv_rgb.py:312-321: [P, 3] valid-pixel positions -> their clamped low-rank reconstruction, [P, 3]."""
        low = self.transform(pose_conds.reshape(1, -1))
        std = torch.sqrt(self.explained_variance_)
        low = torch.minimum(torch.maximum(low, -sigma_pca * std), sigma_pca * std)
        return self.inverse_transform(low).reshape(-1, 3)

    def project_pose_map(self, smpl_pos_map: torch.Tensor, mask: torch.Tensor, sigma_pca: float = 2.) -> torch.Tensor:
        """main_avatar.py:722-733 without the device -> host -> device round trip: the FRONT half (channels 0..2) of
        ``items['smpl_pos_map']`` [6, S, S] is replaced at ``mask`` [S, S] (the dataset's ``pos_map_mask``) by its clamped PCA
        reconstruction; the back half is passed through.  Returns ``items['smpl_pos_map_pca']`` [6, S, S]."""
        out = smpl_pos_map.clone()
        front = out[:3].permute(1, 2, 0)                       # [S, S, 3] view of the clone
        front[mask] = self.transform_pca(front[mask], sigma_pca).to(out.dtype)
        return out
