"""Per-item timeline of blend_backward_kernel on the bench workload (BASELINE configs[1]): where does an item's time go, how busy is
the chip over the launch, what does the tail look like?

Uses the diagnostic build profiles/ub/ko/libag_timeline.so (profiles/ub/build_timeline.sh: the product sources with -DAG_BWD_TIMELINE):
thread 0 of every workgroup stamps each (tile, region) item with the 100-MHz wall clock at four points --

    t0     item start (header of the item in registers)
    t_hdr  after the per-pixel loads (n_contrib, alpha, upstream gradients), the wave reduction of the largest n_contrib and the barrier
    t_rec  after the first point_list -> record round trips, the cull and its barrier  (t_rec - t0 = the item's start-up chain)
    t_end  after the last flush was issued

and the item's sizes: tile list length, walked length (largest n_contrib), survivors of the region cull, blend steps of wave 0
(executed / skipped by the all-lanes-inactive test) and its active (pixel, entry) pairs.

    python profiles/bwd_wg_times.py [view ...]        -> text + one JSON line (committed as profiles/r03_bwd_timeline.*)
"""
import ctypes
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AG_LIB_PATH", os.path.join(ROOT, "profiles", "ub", "ko", "libag_timeline.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import helpers as h  # noqa: E402
from animatablegaussians_amd import synth, _lib  # noqa: E402

ITEM = np.dtype([("wg", "<u4"), ("seq", "<u4"), ("tile", "<u4"), ("region", "<u4"), ("list_len", "<u4"), ("wmax", "<u4"),
                 ("survivors", "<u4"), ("steps", "<u4"), ("skipped", "<u4"), ("active_pairs", "<u4"), ("chunks", "<u4"), ("pad", "<u4"),
                 ("t0", "<u8"), ("t_hdr", "<u8"), ("t_rec", "<u8"), ("t_end", "<u8")])
CAP = 16384


def collect(L):
    buf = np.zeros(CAP, ITEM)
    n = ctypes.c_uint32(0)
    rc = L.ag_debug_bwd_timeline(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(CAP), ctypes.byref(n))
    assert rc == 0, rc
    return buf[: n.value]


def pct(a, qs=(5, 25, 50, 75, 95)):
    return [round(float(x), 2) for x in np.percentile(a, qs)] if len(a) else []


def analyse(it, label):
    us = lambda v: (v.astype(np.int64) - int(it["t0"].min())) / 100.0   # noqa: E731
    t0, th, tr, te = us(it["t0"]), us(it["t_hdr"]), us(it["t_rec"]), us(it["t_end"])
    span = float(te.max())
    dur, startup, hdr, body = te - t0, tr - t0, th - t0, te - tr
    out = {"label": label, "items": int(len(it)), "workgroups": int(len(np.unique(it["wg"]))), "kernel_span_us": round(span, 1),
           "item_us_pct_5_25_50_75_95": pct(dur), "startup_us_pct": pct(startup), "header_phase_us_pct": pct(hdr), "body_us_pct": pct(body),
           "sum_item_us": round(float(dur.sum()), 1), "sum_startup_us": round(float(startup.sum()), 1),
           "startup_share_of_item_time": round(float(startup.sum() / dur.sum()), 3),
           "list_len_mean": round(float(it["list_len"].mean()), 1), "walked_mean": round(float(it["wmax"].mean()), 1),
           "survivors_mean": round(float(it["survivors"].mean()), 1), "chunks_mean": round(float(it["chunks"].mean()), 2),
           "steps_wave0_mean": round(float(it["steps"].mean()), 2), "skipped_steps_wave0_mean": round(float(it["skipped"].mean()), 2),
           "active_lane_fraction_of_executed_steps": round(float(it["active_pairs"].sum() / max(1, 64 * it["steps"].sum())), 3)}
    # body time per executed step and per survivor: the slope of a least-squares fit body = a + b * sub-chunks
    subs = np.ceil(it["survivors"] / 32.0)
    A = np.stack([np.ones_like(subs), subs, it["chunks"].astype(np.float64)], 1)
    coef, *_ = np.linalg.lstsq(A, body, rcond=None)
    out["body_fit_us"] = {"const": round(float(coef[0]), 3), "per_32_entry_subchunk": round(float(coef[1]), 3), "per_512_chunk": round(float(coef[2]), 3)}
    # concurrency: items in flight over time (resident slots = 3 per CU x 256), in 5-us bins
    edges = np.arange(0, span + 5, 5.0)
    inflight = [(int(((t0 < b) & (te > a)).sum())) for a, b in zip(edges[:-1], edges[1:])]
    busy = [round(float(np.clip(np.minimum(te, b) - np.maximum(t0, a), 0, None).sum() / (b - a)), 1) for a, b in zip(edges[:-1], edges[1:])]
    out["mean_items_in_flight_per_5us_bin"] = busy
    out["time_when_90pct_of_item_time_done_us"] = round(float(np.sort(te)[int(0.9 * len(te))]), 1)
    # gaps between consecutive items of a workgroup (dispatch of a new workgroup is not visible here)
    per_wg_end = {}
    for w in np.unique(it["wg"]):
        m = it["wg"] == w
        per_wg_end[int(w)] = (float(t0[m].min()), float(te[m].max()), float(dur[m].sum()))
    starts = np.array([v[0] for v in per_wg_end.values()]); ends = np.array([v[1] for v in per_wg_end.values()])
    out["wg_start_us_pct"] = pct(starts); out["wg_end_us_pct"] = pct(ends)
    out["slot_time_model"] = {"resident_slots": 768, "sum_item_us_over_slots": round(float(dur.sum() / 768), 1)}
    # the longest items
    order = np.argsort(-dur)[:5]
    out["longest_items"] = [dict(us=round(float(dur[i]), 1), startup=round(float(startup[i]), 1), start_at=round(float(t0[i]), 1), list_len=int(it["list_len"][i]),
                                 survivors=int(it["survivors"][i]), chunks=int(it["chunks"][i])) for i in order]
    return out


def main():
    views = [int(v) for v in sys.argv[1:]] or [0, 2, 5]
    L = ctypes.CDLL(_lib.LIB_PATH)
    L.ag_debug_bwd_timeline.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    res = []
    for vi in views:
        av = synth.avatar_map_gaussians()
        camd = synth.free_view_cameras()[vi]
        scene = dict(av, **camd)
        scene.update(synth.upstream_grads(1024, 1024, 11))
        cam = h.cam_of(scene)
        for _ in range(3):
            fw = h.gpu_native_forward(scene, cam)
            collect(L)
            h.gpu_native_backward(fw, {k: scene[k] for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")})
        torch.cuda.synchronize()
        it = collect(L)
        r = analyse(it, f"view {vi}")
        res.append(r)
        for k, v in r.items():
            print(f"  {k}: {v}")
        print()
    print(json.dumps({"bwd_timeline": res}))


if __name__ == "__main__":
    main()
