"""Work figures of blend_backward_pixel_kernel (round 6) on bench views: list entries walked per 8 x 8 region, survivors of the region cull,
consume iterations, active (pixel, entry) lanes, coverage bits.  Uses the diagnostic build
    bash profiles/ub/build_variant.sh pxstats ag_blend_backward -DAG_BWD_STATS -DAG_BWD_PIXEL_KERNEL
python profiles/bwd_px_stats.py [view ...] -> one JSON line per view."""
import ctypes
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AG_LIB_PATH", os.path.join(ROOT, "profiles", "ub", "ko", "libag_pxstats.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import helpers as h  # noqa: E402
from animatablegaussians_amd import synth, _lib  # noqa: E402

NAMES = ["items", "walked", "survivors", "iters", "active_pairs", "mask_bits", "trips", "batches"]


def read(L):
    buf = (ctypes.c_ulonglong * 8)()
    assert L.ag_debug_bwd_px_stats(buf) == 0
    return np.array(list(buf), np.float64)


def main():
    views = [int(v) for v in sys.argv[1:]] or [0, 2, 5]
    L = ctypes.CDLL(_lib.LIB_PATH)
    for vi in views:
        scene = dict(synth.avatar_map_gaussians(), **synth.free_view_cameras()[vi])
        scene.update(synth.upstream_grads(1024, 1024, 11))
        cam = h.cam_of(scene)
        fw = h.gpu_native_forward(scene, cam)
        read(L)
        h.gpu_native_backward(fw, {k: scene[k] for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")})
        torch.cuda.synchronize()
        d = dict(zip(NAMES, read(L)))
        it = max(1.0, d["items"])
        print(json.dumps({"view": vi, "items": int(d["items"]), "walked_per_item": round(d["walked"] / it, 1),
                          "survivors_per_item": round(d["survivors"] / it, 1), "batches_per_item": round(d["batches"] / it, 2),
                          "iters_per_item": round(d["iters"] / it, 2), "trips_per_item": round(d["trips"] / it, 2),
                          "active_lane_share_of_iters": round(d["active_pairs"] / max(1.0, 64 * d["iters"]), 3),
                          "mask_bits_per_active_pair": round(d["mask_bits"] / max(1.0, d["active_pairs"]), 3),
                          "active_pairs": int(d["active_pairs"])}))


if __name__ == "__main__":
    main()
