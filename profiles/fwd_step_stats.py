"""Work figures of blend_forward_kernel on bench views (diagnostic build: bash profiles/ub/build_variant.sh fstats ag_blend_forward -DAG_FWD_STATS).
python profiles/fwd_step_stats.py [view ...] -> one JSON line per view."""
import ctypes
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AG_LIB_PATH", os.path.join(ROOT, "profiles", "ub", "ko", "libag_fstats.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import helpers as h  # noqa: E402
from animatablegaussians_amd import synth, _lib  # noqa: E402

NAMES = ["items", "chunks", "list", "region_survivors", "subcull_passes", "wave_survivors", "steps", "fast_steps", "active_pairs", "breaks"]


def read(L):
    buf = (ctypes.c_ulonglong * 12)()
    assert L.ag_debug_fwd_stats(buf) == 0
    return np.array(list(buf), np.float64)


def main():
    views = [int(v) for v in sys.argv[1:]] or [0, 2, 5]
    L = ctypes.CDLL(_lib.LIB_PATH)
    for vi in views:
        scene = dict(synth.avatar_map_gaussians(), **synth.free_view_cameras()[vi])
        cam = h.cam_of(scene)
        h.gpu_native_forward(scene, cam)
        torch.cuda.synchronize()
        read(L)
        h.gpu_native_forward(scene, cam)
        torch.cuda.synchronize()
        d = dict(zip(NAMES, read(L)[:10]))
        it = max(1.0, d["items"])
        out = {"view": vi, "items_regions": int(d["items"]), "list_per_item": round(d["list"] / it, 1), "chunks_walked_per_item": round(d["chunks"] / it, 2),
               "region_survivors_per_item": round(d["region_survivors"] / it, 1), "subcull_passes_per_wave_item": round(d["subcull_passes"] / it / 8, 2),
               "wave_survivors_per_wave_item": round(d["wave_survivors"] / it / 8, 1), "steps_per_wave_item": round(d["steps"] / it / 8, 2),
               "fast_step_share": round(d["fast_steps"] / max(1.0, d["steps"]), 3), "valid_lane_share_of_steps": round(d["active_pairs"] / max(1.0, 64 * d["steps"]), 3),
               "early_breaks_per_wave_item": round(d["breaks"] / it / 8, 2)}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
