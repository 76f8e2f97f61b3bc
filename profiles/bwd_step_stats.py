"""Work figures of blend_backward_wave_kernel on bench views: list entries walked, survivors of the block cull, steps executed / skipped,
active (pixel, entry) lanes, entries of a step that are active on at least one pixel.  Uses the diagnostic build
    bash profiles/ub/build_variant.sh stats ag_blend_backward -DAG_BWD_STATS
(global counters, not in the product).  python profiles/bwd_step_stats.py [view ...] -> one JSON line per view."""
import ctypes
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AG_LIB_PATH", os.path.join(ROOT, "profiles", "ub", "ko", "libag_stats.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import helpers as h  # noqa: E402
from animatablegaussians_amd import synth, _lib  # noqa: E402

NAMES = ["items", "walked", "survivors", "steps", "skipped_steps", "active_pairs", "active_entries"]


def read(L):
    buf = (ctypes.c_ulonglong * 12)()
    assert L.ag_debug_bwd_stats(buf) == 0
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
        c = read(L)
        d = dict(zip(NAMES, c[:7]))
        it = max(1.0, d["items"])
        out = {"view": vi, "items": int(d["items"]), "walked_per_item": round(d["walked"] / it, 1), "survivors_per_item": round(d["survivors"] / it, 1),
               "steps_per_item": round(d["steps"] / it, 2), "skipped_share_of_steps": round(d["skipped_steps"] / max(1.0, d["steps"]), 3),
               "active_lane_share_of_steps": round(d["active_pairs"] / max(1.0, 64 * d["steps"]), 3),
               "entries_active_on_some_pixel_share": round(d["active_entries"] / max(1.0, 4 * d["steps"]), 3),
               "steps_by_active_entries_0_to_4": [round(x / max(1.0, d["steps"]), 3) for x in c[7:12]]}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
