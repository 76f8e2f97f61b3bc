// C-ABI entry points of libag_hip.so (declared in include/ag_raster.h): argument validation, scratch sizing,
// stage orchestration.  Host code only; the kernels live in the other translation units.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <cstring>
#include <vector>

#include "ag_common.h"

namespace ag {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what)
{
    if (e == hipSuccess) return AG_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return AG_ERR_HIP;
}

// ---- kernel timing log ---------------------------------------------------------------------------------------
// One record per bracketed launch: its own event pair (created on the device the launch runs on), the launch stream
// and the work the caller declared for it.  A ProfScope keeps the index of ITS record, so launches of the same kernel
// from several threads / streams / devices never pair each other's events.
static uint32_t g_prof_mask = 0;
struct ProfRec { int id, dev; hipEvent_t a, b; double work; bool ended; char tag[64]; };
static std::vector<ProfRec> g_prof_log;
static int g_prof_gen = 0;       // bumped by every ag_prof_collect: a handle taken before a collect must not touch the refilled log
static std::vector<std::pair<int, hipEvent_t>> g_prof_pool;   // (device, event)
static std::mutex g_prof_mu;

static hipEvent_t prof_event(int dev)
{
    for (size_t i = g_prof_pool.size(); i-- > 0;)
        if (g_prof_pool[i].first == dev) {
            hipEvent_t e = g_prof_pool[i].second;
            g_prof_pool.erase(g_prof_pool.begin() + (long)i);
            return e;
        }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

int prof_begin(int id, hipStream_t s, double work, const char* tag)
{
    if (!(g_prof_mask & (1u << id))) return -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_log.size() >= (1u << 20)) return -1;
    ProfRec r{ id, dev, prof_event(dev), prof_event(dev), work, false, { 0 } };
    if (tag) { strncpy(r.tag, tag, sizeof(r.tag) - 1); r.tag[sizeof(r.tag) - 1] = 0; }
    (void)hipEventRecord(r.a, s);
    g_prof_log.push_back(r);
    return ((g_prof_gen & 0x3ff) << 20) | (int)(g_prof_log.size() - 1);      // generation in the upper bits
}

void prof_end(int handle, hipStream_t s)
{
    if (handle < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const size_t idx = (size_t)(handle & 0xfffff);
    if ((handle >> 20) != (g_prof_gen & 0x3ff) || idx >= g_prof_log.size()) return;   // another thread collected in between: record dropped
    (void)hipEventRecord(g_prof_log[idx].b, s);
    g_prof_log[idx].ended = true;
}

// One pinned word per thread for the num_rendered read-back (the reference's blocking cudaMemcpy,
// rasterizer_impl.cu:282).
static int32_t* pinned_word()
{
    static thread_local int32_t* w = nullptr;
    if (!w) {
        if (hipHostMalloc(reinterpret_cast<void**>(&w), 64, hipHostMallocDefault) != hipSuccess) w = nullptr;
    }
    return w;
}

static hipEvent_t plan_event()
{
    static thread_local hipEvent_t ev = nullptr;
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
    return ev;
}

static int validate_forward(const AgRasterForwardArgs* a, bool need_bin, int R)
{
    if (!a) { set_error("null args"); return AG_ERR_INVALID_ARGUMENT; }
    if (a->P < 0 || a->W <= 0 || a->H <= 0) { set_error("bad sizes P=%d W=%d H=%d", a->P, a->W, a->H); return AG_ERR_INVALID_ARGUMENT; }
    if (a->P == 0) return AG_OK;
    if (!a->means3D || !a->opacities || !a->bg || !a->viewmatrix || !a->projmatrix) {
        set_error("means3D/opacities/bg/viewmatrix/projmatrix must be non-null");
        return AG_ERR_INVALID_ARGUMENT;
    }
    if (!a->colors_precomp) {
        if (!a->shs || !a->campos) { set_error("need colors_precomp, or shs together with campos"); return AG_ERR_INVALID_ARGUMENT; }
        if (a->sh_degree < 0 || a->sh_degree > 3 || a->sh_coeffs < (a->sh_degree + 1) * (a->sh_degree + 1)) {
            set_error("SH degree %d needs 0 <= degree <= 3 and at least %d coefficients per Gaussian (got %d)", a->sh_degree,
                      (a->sh_degree + 1) * (a->sh_degree + 1), a->sh_coeffs);
            return AG_ERR_INVALID_ARGUMENT;
        }
    }
    const bool sr = a->scales && a->rotations;
    if (!sr && !a->cov3D_precomp) { set_error("need scales+rotations or cov3D_precomp"); return AG_ERR_INVALID_ARGUMENT; }
    if (!a->out_color || !a->out_depth || !a->out_alpha || !a->radii) { set_error("null output"); return AG_ERR_INVALID_ARGUMENT; }
    if (!a->geom_buffer || a->geom_bytes < ag_raster_geom_bytes(a->P)) { set_error("geom_buffer too small"); return AG_ERR_SCRATCH_TOO_SMALL; }
    if (!a->image_buffer || a->image_bytes < ag_raster_image_bytes(a->W, a->H)) { set_error("image_buffer too small"); return AG_ERR_SCRATCH_TOO_SMALL; }
    if (need_bin && R > 0 && (!a->binning_buffer || a->binning_bytes < ag_raster_binning_bytes(R))) {
        set_error("binning_buffer too small for %d instances", R);
        return AG_ERR_SCRATCH_TOO_SMALL;
    }
    return AG_OK;
}

}  // namespace ag

using namespace ag;

extern "C" {

int ag_abi_version(void) { return AG_ABI_VERSION; }
const char* ag_last_error(void) { return g_err; }

size_t ag_raster_geom_bytes(int32_t P) { return GeomLayout((size_t)(P > 0 ? P : 0)).total; }
size_t ag_raster_image_bytes(int32_t W, int32_t H) { return ImageLayout((size_t)W, (size_t)H).total; }
size_t ag_raster_binning_bytes(int32_t R) { return BinLayout((size_t)(R > 0 ? R : 0)).total; }
size_t ag_raster_accum_bytes(int32_t P) { return (size_t)(P > 0 ? P : 0) * kAccumFloats * sizeof(float) + 512; }

int ag_raster_describe_scratch(int32_t P, int32_t W, int32_t H, int32_t R, AgRasterScratchLayout* o)
{
    if (!o) return AG_ERR_INVALID_ARGUMENT;
    GeomLayout gl((size_t)P);
    ImageLayout il((size_t)W, (size_t)H);
    BinLayout bl((size_t)(R > 0 ? R : 0));
    o->geom_rec_off = gl.rec; o->geom_rec_stride = sizeof(GaussRec);
    o->geom_cov3d_off = gl.cov3d; o->geom_tiles_touched_off = gl.tiles_touched;
    o->img_ranges_off = il.ranges; o->img_n_contrib_off = il.n_contrib; o->img_tile_count_off = il.tile_count;
    o->img_num_rendered_off = il.num_rendered;
    o->bin_point_list_off = bl.point_list; o->bin_keys_off = bl.keys;
    return AG_OK;
}

// Large-class sort launch (ag_binning.hip launch_bin_sort): enqueued behind a frame only once some frame of the process has had a tile of >= 2048
// instances.  The paths that enqueue everything before the host knows the counts (optimistic, enqueue + collect) consult this flag; a frame that
// needed the launch although it was skipped comes back as AG_ERR_SCRATCH_TOO_SMALL (untouched outputs, like an overflow) with the flag raised, and
// the caller's redo then gets the launch.  The host-synchronised path (plan + render) always launches it.
static std::atomic<int> g_large_tiles_seen{ 0 };

int ag_raster_forward_plan(const AgRasterForwardArgs* a, void* stream, int32_t* num_rendered_host)
{
    if (!num_rendered_host) { set_error("null num_rendered_host"); return AG_ERR_INVALID_ARGUMENT; }
    *num_rendered_host = 0;
    int rc = validate_forward(a, false, 0);
    if (rc) return rc;
    if (a->P == 0) return AG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if ((rc = launch_preprocess(*a, s))) return rc;
    if ((rc = launch_tile_scan(*a, s))) return rc;
    int32_t* w = pinned_word();
    if (!w) { set_error("hipHostMalloc failed"); return AG_ERR_HIP; }
    ImageLayout il((size_t)a->W, (size_t)a->H);
    const char* ib = aligned_base(a->image_buffer);
    if ((rc = check_hip(hipMemcpyAsync(w, ib + il.num_rendered, sizeof(int32_t), hipMemcpyDeviceToHost, s), "read num_rendered"))) return rc;
    if ((rc = check_hip(hipStreamSynchronize(s), "sync after plan"))) return rc;
    *num_rendered_host = *w;
    return AG_OK;
}

int ag_raster_forward_optimistic(const AgRasterForwardArgs* a, int32_t capacity, void* stream, int32_t* num_rendered_host)
{
    if (!num_rendered_host) { set_error("null num_rendered_host"); return AG_ERR_INVALID_ARGUMENT; }
    *num_rendered_host = 0;
    if (capacity <= 0) { set_error("capacity must be positive"); return AG_ERR_INVALID_ARGUMENT; }
    int rc = validate_forward(a, true, capacity);
    if (rc) return rc;
    if (a->P == 0) return ag_raster_forward_render(a, 0, stream);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int32_t* w = pinned_word();
    hipEvent_t ev = plan_event();
    if (!w || !ev) { set_error("hipHostMalloc / hipEventCreate failed"); return AG_ERR_HIP; }
    const bool skip_large = g_large_tiles_seen.load(std::memory_order_relaxed) == 0;
    if ((rc = launch_preprocess(*a, s))) return rc;
    if ((rc = launch_tile_scan(*a, s, (uint32_t)capacity, skip_large))) return rc;
    ImageLayout il((size_t)a->W, (size_t)a->H);
    const char* ib = aligned_base(a->image_buffer);
    if ((rc = check_hip(hipMemcpyAsync(w, ib + il.num_rendered, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s), "read num_rendered"))) return rc;
    if ((rc = check_hip(hipEventRecord(ev, s), "record plan event"))) return rc;
    // everything else is enqueued BEFORE the host learns the count: the GPU never waits for the host round trip
    if ((rc = launch_bin_sort(*a, capacity, s, skip_large))) return rc;
    if ((rc = launch_blend_forward(*a, capacity, s))) return rc;
    if ((rc = check_hip(hipEventSynchronize(ev), "wait for the instance count"))) return rc;
    *num_rendered_host = w[0];
    if (w[3] > 0) g_large_tiles_seen.store(1, std::memory_order_relaxed);
    if (w[2]) {
        if (w[0] <= capacity) set_error("optimistic forward: %d tiles of >= 2048 instances and the large-class sort was not enqueued for this frame; redo it", w[3]);
        else set_error("optimistic forward: %d instances exceed the capacity of %d; redo with ag_raster_forward_plan + _render", w[0], capacity);
        return AG_ERR_SCRATCH_TOO_SMALL;
    }
    return AG_OK;
}

int ag_raster_forward_render(const AgRasterForwardArgs* a, int32_t R, void* stream)
{
    int rc = validate_forward(a, true, R);
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (a->P == 0) {
        // rasterize_points.cu:68-83: the native call is skipped; colour/depth/alpha stay all-zero (not bg)
        const size_t HW = (size_t)a->W * a->H;
        if ((rc = check_hip(hipMemsetAsync(a->out_color, 0, 3 * HW * sizeof(float), s), "memset"))) return rc;
        if ((rc = check_hip(hipMemsetAsync(a->out_depth, 0, HW * sizeof(float), s), "memset"))) return rc;
        return check_hip(hipMemsetAsync(a->out_alpha, 0, HW * sizeof(float), s), "memset");
    }
    if ((rc = launch_bin_sort(*a, R, s))) return rc;
    return launch_blend_forward(*a, R, s);
}

static int validate_backward(const AgRasterBackwardArgs* a)
{
    if (!a) { set_error("null args"); return AG_ERR_INVALID_ARGUMENT; }
    if (a->P < 0 || a->W <= 0 || a->H <= 0 || a->num_rendered < 0) { set_error("bad sizes"); return AG_ERR_INVALID_ARGUMENT; }
    if (a->P == 0) return AG_OK;   // all outputs are empty
    if (!a->colors_precomp) {
        if (!a->shs || !a->campos || !a->dL_dsh) { set_error("SH backward needs shs, campos and dL_dsh"); return AG_ERR_INVALID_ARGUMENT; }
        if (a->sh_degree < 0 || a->sh_degree > 3 || a->sh_coeffs < (a->sh_degree + 1) * (a->sh_degree + 1)) {
            set_error("bad SH degree / coefficient count");
            return AG_ERR_INVALID_ARGUMENT;
        }
        if (a->accumulate) { set_error("accumulate: colours-precomp path only"); return AG_ERR_UNSUPPORTED; }
    }
    if (!a->means3D || !a->radii || !a->bg || !a->viewmatrix || !a->projmatrix || !a->alphas || !a->dL_dout_color ||
        !a->dL_dout_depth || !a->dL_dout_alpha || !a->geom_buffer || !a->image_buffer) {
        set_error("null input to backward");
        return AG_ERR_INVALID_ARGUMENT;
    }
    if (a->num_rendered > 0 && !a->binning_buffer) { set_error("null binning_buffer"); return AG_ERR_INVALID_ARGUMENT; }
    if (!(a->scales && a->rotations) && !a->cov3D_precomp) { set_error("need scales+rotations or cov3D_precomp"); return AG_ERR_INVALID_ARGUMENT; }
    if (!a->dL_dmeans2D || !a->dL_dcolors || !a->dL_dopacity || !a->dL_dmeans3D || !a->dL_dcov3D || !a->dL_dscales ||
        !a->dL_drotations) {
        set_error("null gradient output");
        return AG_ERR_INVALID_ARGUMENT;
    }
    if (!a->accum_buffer || a->accum_bytes < ag_raster_accum_bytes(a->P)) { set_error("accum_buffer too small"); return AG_ERR_SCRATCH_TOO_SMALL; }
    return AG_OK;
}

int ag_raster_backward(const AgRasterBackwardArgs* a, void* stream)
{
    int rc = validate_backward(a);
    if (rc || a->P == 0) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if ((rc = launch_blend_backward(*a, s))) return rc;
    return launch_preprocess_backward(*a, s);
}

// Tickets of views whose instance count has not been read yet (ag_raster_forward_backward_enqueue / ag_raster_collect): a pinned
// 3-word landing zone and an event each, per device, reused.
namespace {
struct Ticket { int32_t* w = nullptr; hipEvent_t ev = nullptr; int dev = -1; int32_t capacity = 0; bool busy = false; };
constexpr int kTickets = 64;
Ticket g_tk[kTickets];
std::mutex g_tk_mu;

int ticket_acquire(int32_t capacity)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_tk_mu);
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < kTickets; i++) {
            Ticket& t = g_tk[i];
            if (t.busy) continue;
            if (pass == 0 && t.dev != dev) continue;          // first a free ticket of this device, then an unused one
            if (pass == 1 && t.dev != -1) continue;
            if (t.dev == -1) {
                if (hipHostMalloc(reinterpret_cast<void**>(&t.w), 64, hipHostMallocDefault) != hipSuccess) { t.w = nullptr; return -1; }
                if (hipEventCreateWithFlags(&t.ev, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(t.w); t.w = nullptr; t.ev = nullptr; return -1; }
                t.dev = dev;
            }
            t.busy = true; t.capacity = capacity;
            return i;
        }
    return -1;
}
}  // namespace

int ag_raster_forward_backward_enqueue(const AgRasterForwardArgs* f, AgRasterBackwardArgs* b, int32_t capacity, void* stream, int32_t* ticket)
{
    if (!f || !b || !ticket) { set_error("null args"); return AG_ERR_INVALID_ARGUMENT; }
    *ticket = -1;
    if (capacity <= 0) { set_error("capacity must be positive"); return AG_ERR_INVALID_ARGUMENT; }
    if (f->shs && !f->colors_precomp) { set_error("ag_raster_forward_backward: colours-precomp path only"); return AG_ERR_UNSUPPORTED; }
    // complete the backward arguments from the forward's
    b->P = f->P; b->W = f->W; b->H = f->H; b->sh_degree = f->sh_degree; b->sh_coeffs = f->sh_coeffs; b->num_rendered = capacity;
    b->tan_fovx = f->tan_fovx; b->tan_fovy = f->tan_fovy; b->scale_modifier = f->scale_modifier;
    b->bg = f->bg; b->means3D = f->means3D; b->radii = f->radii; b->colors_precomp = f->colors_precomp; b->shs = f->shs;
    b->scales = f->scales; b->rotations = f->rotations; b->cov3D_precomp = f->cov3D_precomp;
    b->viewmatrix = f->viewmatrix; b->projmatrix = f->projmatrix; b->campos = f->campos; b->alphas = f->out_alpha;
    b->geom_buffer = f->geom_buffer; b->image_buffer = f->image_buffer; b->binning_buffer = f->binning_buffer;
    int rc = validate_forward(f, true, capacity);
    if (rc) return rc;
    if ((rc = validate_backward(b))) return rc;
    if (f->P == 0) return ag_raster_forward_render(f, 0, stream);           // nothing to count: no ticket (*ticket stays -1)
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int tk = ticket_acquire(capacity);
    if (tk < 0) { set_error("no free read-back ticket (64 views pending) or hipHostMalloc / hipEventCreate failed"); return AG_ERR_HIP; }
    Ticket& t = g_tk[tk];
    auto fail = [&](int code) { std::lock_guard<std::mutex> lk(g_tk_mu); t.busy = false; return code; };
    const bool skip_large = g_large_tiles_seen.load(std::memory_order_relaxed) == 0;
    if ((rc = launch_preprocess(*f, s))) return fail(rc);
    if ((rc = launch_tile_scan(*f, s, (uint32_t)capacity, skip_large))) return fail(rc);
    ImageLayout il((size_t)f->W, (size_t)f->H);
    const char* ib = aligned_base(f->image_buffer);
    if ((rc = check_hip(hipMemcpyAsync(t.w, ib + il.num_rendered, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s), "read num_rendered"))) return fail(rc);
    if ((rc = check_hip(hipEventRecord(t.ev, s), "record plan event"))) return fail(rc);
    if ((rc = launch_bin_sort(*f, capacity, s, skip_large))) return fail(rc);
    if ((rc = launch_blend_forward(*f, capacity, s))) return fail(rc);
    if ((rc = launch_blend_backward(*b, s))) return fail(rc);
    if ((rc = launch_preprocess_backward(*b, s))) return fail(rc);
    *ticket = tk;
    return AG_OK;
}

int ag_raster_collect(int32_t ticket, int32_t* num_rendered_host)
{
    if (!num_rendered_host) { set_error("null num_rendered_host"); return AG_ERR_INVALID_ARGUMENT; }
    *num_rendered_host = 0;
    if (ticket < 0 || ticket >= kTickets || !g_tk[ticket].busy) { set_error("ag_raster_collect: no such pending view (%d)", ticket); return AG_ERR_INVALID_ARGUMENT; }
    Ticket& t = g_tk[ticket];
    const int rc = check_hip(hipEventSynchronize(t.ev), "wait for the instance count");
    const int32_t n = t.w[0], over = t.w[2], n_large = t.w[3], cap = t.capacity;
    { std::lock_guard<std::mutex> lk(g_tk_mu); t.busy = false; }
    if (rc) return rc;
    *num_rendered_host = n;
    if (n_large > 0) g_large_tiles_seen.store(1, std::memory_order_relaxed);
    if (over) {
        if (n <= cap) set_error("forward+backward: %d tiles of >= 2048 instances and the large-class sort was not enqueued for this view; redo it", n_large);
        else set_error("forward+backward: %d instances exceed the capacity of %d; redo the view with a larger capacity", n, cap);
        return AG_ERR_SCRATCH_TOO_SMALL;
    }
    return AG_OK;
}

int ag_raster_large_tile_sort(int32_t mode)
{
    if (mode == 0 || mode == 1) g_large_tiles_seen.store(mode, std::memory_order_relaxed);
    else if (mode != -1) { set_error("ag_raster_large_tile_sort: mode must be -1 (query), 0 or 1"); return AG_ERR_INVALID_ARGUMENT; }
    return g_large_tiles_seen.load(std::memory_order_relaxed);
}

int ag_raster_forward_backward(const AgRasterForwardArgs* f, AgRasterBackwardArgs* b, int32_t capacity, void* stream,
                               int32_t* num_rendered_host)
{
    if (!num_rendered_host) { set_error("null args"); return AG_ERR_INVALID_ARGUMENT; }
    *num_rendered_host = 0;
    int32_t ticket = -1;
    const int rc = ag_raster_forward_backward_enqueue(f, b, capacity, stream, &ticket);
    if (rc || ticket < 0) return rc;
    return ag_raster_collect(ticket, num_rendered_host);
}

const char* ag_prof_kernel_name(int32_t id)
{
    static const char* names[AG_K_COUNT] = { "preprocess_kernel", "tile_scan_kernel", "scatter_kernel", "tile_sort_kernel",
                                             "blend_forward_kernel", "blend_backward_kernel", "preprocess_backward_kernel",
                                             "gather_conv_kernel", "wgrad_kernel" };
    return (id >= 0 && id < AG_K_COUNT) ? names[id] : "";
}

int ag_prof_enable(uint32_t mask)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_mask = mask;
    return AG_OK;
}

static FILE* g_prof_dump = nullptr;      // ag_prof_collect_to: one line per record (kernel, tag, work, ms) while collecting

int ag_prof_collect(int32_t* launches, float* total_ms, double* work)
{
    if (!launches || !total_ms) return AG_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < AG_K_COUNT; i++) { launches[i] = 0; total_ms[i] = 0.f; if (work) work[i] = 0.0; }
    int rc = AG_OK, cur = 0;
    (void)hipGetDevice(&cur);
    for (auto& r : g_prof_log) {
        float ms = 0.f;
        (void)hipSetDevice(r.dev);
        if (!r.ended) { /* begun on another thread, not ended yet: not measured */ }
        else if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) rc = AG_ERR_HIP;
        else {
            launches[r.id]++; total_ms[r.id] += ms; if (work) work[r.id] += r.work;
            if (g_prof_dump) fprintf(g_prof_dump, "%s,%s,%.0f,%.5f\n", ag_prof_kernel_name(r.id), r.tag, r.work, ms);
        }
        g_prof_pool.push_back({ r.dev, r.a });
        g_prof_pool.push_back({ r.dev, r.b });
    }
    (void)hipSetDevice(cur);
    g_prof_log.clear();
    g_prof_gen++;
    return rc;
}

/* ag_prof_collect that also writes one CSV line per bracketed launch, in launch order: kernel,tag,work,ms (the tag is whatever the launcher
 * declared: the convolutions put their shape there).  Diagnostic (profiles/conv_launch_table.py). */
int ag_prof_collect_to(const char* path, int32_t* launches, float* total_ms, double* work)
{
    if (!path) return AG_ERR_INVALID_ARGUMENT;
    FILE* f = fopen(path, "w");
    if (!f) { set_error("ag_prof_collect_to: cannot open %s", path); return AG_ERR_INVALID_ARGUMENT; }
    fprintf(f, "kernel,tag,work,ms\n");
    g_prof_dump = f;
    const int rc = ag_prof_collect(launches, total_ms, work);
    g_prof_dump = nullptr;
    fclose(f);
    return rc;
}

int ag_debug_atomic_rate(float* accum, int32_t lines, int32_t blocks, int32_t iters, int32_t comps, void* stream)
{
    if (!accum || lines <= 0 || blocks <= 0 || iters < 0 || comps < 1 || comps > 16) { set_error("bad atomic-rate arguments"); return AG_ERR_INVALID_ARGUMENT; }
    return launch_debug_atomic_rate(accum, lines, blocks, iters, comps, reinterpret_cast<hipStream_t>(stream));
}

int ag_raster_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                           uint8_t* present, void* stream)
{
    (void)projmatrix;
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) { set_error("bad mark_visible args"); return AG_ERR_INVALID_ARGUMENT; }
    if (P == 0) return AG_OK;
    return launch_mark_visible(P, means3D, viewmatrix, present, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
