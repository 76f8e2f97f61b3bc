// Block/fiber scheduler of the CUDA-on-CPU shim (see cuda_runtime.h).  TEST INFRASTRUCTURE ONLY.
#include <set>

#include "cuda_runtime.h"

namespace cuemu {

ThreadCtx g_ctx;

namespace {
constexpr size_t kStackBytes = 256 * 1024;

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool finished = false, waiting = false;
    ThreadCtx coords;
};

struct BlockRun {
    std::vector<Fiber> fibers;
    ucontext_t sched;
    int current = -1;
    int arrived = 0, pred_sum = 0, released_count = 0;
    bool saw_barrier = false;
    void (*thunk)(void*) = nullptr;
    void* closure = nullptr;
};

BlockRun* g_run = nullptr;       // non-null while a block executes in fiber mode
bool g_direct_mode = false;      // true while a block executes as a plain loop (barriers are illegal there)
std::set<const void*> g_barrier_free_kernels;
std::vector<char*> g_stack_pool;

void trampoline()
{
    BlockRun* r = g_run;
    Fiber& f = r->fibers[r->current];
    r->thunk(r->closure);
    f.finished = true;
    swapcontext(&f.ctx, &r->sched);
}

void run_block_fibers(BlockRun& r, dim3 grid, dim3 block, uint3 bidx)
{
    const unsigned n = block.x * block.y * block.z;
    r.fibers.assign(n, Fiber());
    while (g_stack_pool.size() < n) g_stack_pool.push_back(static_cast<char*>(malloc(kStackBytes)));
    for (unsigned t = 0; t < n; t++) {
        Fiber& f = r.fibers[t];
        f.stack = g_stack_pool[t];
        f.coords.tIdx = { t % block.x, (t / block.x) % block.y, t / (block.x * block.y) };
        f.coords.bIdx = bidx;
        f.coords.bDim = block;
        f.coords.gDim = grid;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStackBytes;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, trampoline, 0);
    }
    g_run = &r;
    unsigned alive = n;
    while (alive > 0) {
        // run every runnable fiber until it parks at a barrier or finishes
        for (unsigned t = 0; t < n; t++) {
            Fiber& f = r.fibers[t];
            if (f.finished || f.waiting) continue;
            r.current = (int)t;
            g_ctx = f.coords;
            swapcontext(&r.sched, &f.ctx);
            if (f.finished) alive--;
        }
        if (alive == 0) break;
        // everybody still alive is parked: release the barrier
        if (r.arrived != (int)alive) {
            fprintf(stderr, "cuda_cpu: barrier reached by %d of %u live threads (divergent barrier)\n", r.arrived, alive);
            abort();
        }
        r.released_count = r.pred_sum;
        r.arrived = 0;
        r.pred_sum = 0;
        for (auto& f : r.fibers) f.waiting = false;
    }
    g_run = nullptr;
}

}  // namespace

void barrier_wait(int pred, int* count_out)
{
    if (g_direct_mode) {
        fprintf(stderr, "cuda_cpu: barrier inside a kernel classified barrier-free\n");
        abort();
    }
    BlockRun* r = g_run;
    Fiber& f = r->fibers[r->current];
    r->saw_barrier = true;
    r->arrived++;
    r->pred_sum += pred;
    f.waiting = true;
    swapcontext(&f.ctx, &r->sched);
    g_ctx = f.coords;
    if (count_out) *count_out = r->released_count;
}

void run_grid(dim3 grid, dim3 block, void (*thunk)(void*), void* closure, const void* kernel_key)
{
    // Plain-loop execution is used only for 1-D-block kernels that went through one complete launch in fiber mode
    // without ever reaching a barrier (the reference's per-Gaussian kernels).  Kernels with 2-D blocks (the two
    // renderCUDA kernels, which synchronise) always run as fibers.
    const bool one_d = block.y == 1 && block.z == 1;
    const bool direct = one_d && g_barrier_free_kernels.count(kernel_key) != 0;
    bool saw_barrier = false;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                const uint3 bidx = { bx, by, bz };
                if (direct) {
                    g_direct_mode = true;
                    g_ctx.bIdx = bidx; g_ctx.bDim = block; g_ctx.gDim = grid;
                    for (unsigned tx = 0; tx < block.x; tx++) {
                        g_ctx.tIdx = { tx, 0, 0 };
                        thunk(closure);
                    }
                    g_direct_mode = false;
                } else {
                    BlockRun r;
                    r.thunk = thunk;
                    r.closure = closure;
                    run_block_fibers(r, grid, block, bidx);
                    saw_barrier |= r.saw_barrier;
                }
            }
    if (!direct && one_d && !saw_barrier) g_barrier_free_kernels.insert(kernel_key);
}

}  // namespace cuemu
