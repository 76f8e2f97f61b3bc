"""Per-workgroup start / end times of blend_backward_kernel with a 512-workgroup grid: how unbalanced is the static deal?

Needs an instrumented build of the library selected with AG_LIB_PATH (a throw-away copy of csrc/, not committed):
    __device__ unsigned long long g_dbg[1024 * 2];                                   // before blend_backward_kernel
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_dbg[blockIdx.x * 2] = wall_clock64();       // first statement of the kernel
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_dbg[blockIdx.x * 2 + 1] = wall_clock64();   // last statement
    extern "C" int ag_debug_bwd_times(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(g_dbg)); }
Result on one box (512 workgroups): kernel span 154 / 207 / 178 us for views 0 / 2 / 5, workgroup end times 82-154, 41-207, 93-178 us,
busy fraction 0.75 / 0.63 / 0.76 -> kBlendGrid (ag_common.h)."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np, torch
import helpers as h
from animatablegaussians_amd import synth, _lib
L = ctypes.CDLL(_lib.LIB_PATH)
for vi in (0, 2, 5):
    av = synth.avatar_map_gaussians(); camd = synth.free_view_cameras()[vi]
    scene = dict(av, **camd); scene.update(synth.upstream_grads(1024, 1024, 11))
    cam = h.cam_of(scene)
    for it in range(3):
        fw = h.gpu_native_forward(scene, cam)
        g = h.gpu_native_backward(fw, {k: scene[k] for k in ('dL_dcolor', 'dL_ddepth', 'dL_dalpha')})
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 2048)()
    assert L.ag_debug_bwd_times(buf) == 0
    t = np.array(buf[:1024], dtype=np.int64).reshape(512, 2)
    t0 = t[:, 0].min()
    st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0      # us (100 MHz)
    dur = en - st
    print(f"view {vi}: kernel span {en.max():.1f} us; WG start max {st.max():.1f}; WG end min/median/mean/max {en.min():.1f} {np.median(en):.1f} {en.mean():.1f} {en.max():.1f}; "
          f"busy fraction {dur.sum() / (512 * en.max()):.3f}")
    q = np.percentile(en, [5, 25, 50, 75, 95])
    print("   end-time percentiles 5/25/50/75/95:", np.round(q, 1), " per-XCD mean end:", np.round([en[x::8].mean() for x in range(8)], 1),
          " per-XCD max end:", np.round([en[x::8].max() for x in range(8)], 1))
