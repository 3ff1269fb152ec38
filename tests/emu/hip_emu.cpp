// Fiber scheduler behind tests/emu/hip_emu.h (TEST INFRASTRUCTURE ONLY).
#include "hip_emu.h"

#include <stdio.h>
#include <ucontext.h>
#include <vector>

uint3_emu threadIdx, blockIdx;
dim3 blockDim, gridDim;
alignas(64) static unsigned char g_lds_one[160 * 1024];
unsigned char* aae_emu_dyn_smem = g_lds_one;

namespace {

constexpr size_t kStackBytes = 256 * 1024;

struct WaveState {
    int arrived = 0;
    int live = 0;
    unsigned gen = 0;
    alignas(16) unsigned char buf[2][64][64];
};

// what a block's threads share: the barrier, the wave collectives, the LDS image, the block index
struct BlockState {
    int live = 0;
    int bar_arrived = 0;
    unsigned bar_gen = 0;
    std::vector<WaveState> waves;
    unsigned char* lds = nullptr;
    uint3_emu bid = {0, 0, 0};
};

struct Fiber {
    ucontext_t ctx;
    unsigned char* stack = nullptr;
    bool done = false;
    uint3_emu tid;
    BlockState* block = nullptr;
};

ucontext_t g_sched;
std::vector<Fiber> g_fibers;
Fiber* g_cur = nullptr;
BlockState* g_bs = nullptr;   // block of the running fiber
const std::function<void()>* g_body = nullptr;
unsigned long g_events = 0;   // arrivals / releases / exits, for deadlock detection
int g_block_order = 0;        // 0 ascending, 1 descending, 2 scrambled (aae_emu_set_block_order)

void enter(Fiber& f) {
    g_cur = &f;
    g_bs = f.block;
    threadIdx = f.tid;
    blockIdx = f.block->bid;
    aae_emu_dyn_smem = f.block->lds;
}

void yield_to_scheduler() {
    Fiber* me = g_cur;
    swapcontext(&me->ctx, &g_sched);
    enter(*me);
}

void fiber_entry() {
    (*g_body)();
    g_cur->done = true;
    swapcontext(&g_cur->ctx, &g_sched);
}

int flat_tid(const uint3_emu& t) { return t.x + blockDim.x * (t.y + blockDim.y * t.z); }

void make_fiber(Fiber& f, BlockState* bs, int t, dim3 block) {
    f.done = false;
    f.block = bs;
    f.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStackBytes;
    f.ctx.uc_link = &g_sched;
    makecontext(&f.ctx, fiber_entry, 0);
}

void ensure_fibers(size_t n) {
    if (g_fibers.size() < n) {
        const size_t old = g_fibers.size();
        g_fibers.resize(n);
        for (size_t i = old; i < g_fibers.size(); ++i) g_fibers[i].stack = (unsigned char*)malloc(kStackBytes);
    }
}

// run fibers [0, n) (n = threads per block x blocks alive) until all are done
void run_fibers(int n, int nthreads) {
    int remaining = n;
    while (remaining > 0) {
        const unsigned long events_before = g_events;
        for (int t = 0; t < n; ++t) {
            Fiber& f = g_fibers[t];
            if (f.done) continue;
            enter(f);
            swapcontext(&g_sched, &f.ctx);
            if (f.done) {
                BlockState& bs = *f.block;
                --remaining;
                --bs.live;
                WaveState& w = bs.waves[(t % nthreads) >> 6];
                --w.live;
                ++g_events;
                // an exiting thread may complete a pending barrier / collective
                if (bs.live > 0 && bs.bar_arrived >= bs.live) { bs.bar_arrived = 0; ++bs.bar_gen; }
                if (w.live > 0 && w.arrived >= w.live) { w.arrived = 0; ++w.gen; }
            }
        }
        if (g_events == events_before) {
            fprintf(stderr, "emu: deadlock (last block (%u,%u,%u)): a barrier / collective / grid-wide wait never completes\n",
                    blockIdx.x, blockIdx.y, blockIdx.z);
            abort();
        }
    }
}

void check_launch(dim3 block, size_t smem_bytes) {
    if (smem_bytes > sizeof(g_lds_one)) {
        fprintf(stderr, "emu: dynamic LDS request %zu exceeds 160 KiB\n", smem_bytes);
        abort();
    }
    if ((block.x * block.y * block.z) % 64 != 0) {
        fprintf(stderr, "emu: block size %u not a multiple of the 64-lane wave\n", block.x * block.y * block.z);
        abort();
    }
}

}  // namespace

void __syncthreads() {
    BlockState& bs = *g_bs;
    ++g_events;
    if (++bs.bar_arrived >= bs.live) {
        bs.bar_arrived = 0;
        ++bs.bar_gen;
        return;
    }
    const unsigned gen = bs.bar_gen;
    while (bs.bar_gen == gen) yield_to_scheduler();
}

namespace aae_emu {

// a thread that polls memory another block will write (grid-wide waits of a resident launch)
void spin_yield() { yield_to_scheduler(); }
void note_progress() { ++g_events; }

const lane_slot* wave_exchange(const void* mine, int nbytes) {
    const int ft = flat_tid(threadIdx);
    WaveState& w = g_bs->waves[ft >> 6];
    const int p = w.gen & 1;
    ++g_events;
    memcpy(w.buf[p][ft & 63], mine, nbytes);
    if (++w.arrived >= w.live) {
        w.arrived = 0;
        ++w.gen;
    } else {
        const unsigned gen = w.gen;
        while (w.gen == gen) yield_to_scheduler();
    }
    return w.buf[p];
}

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    check_launch(block, smem_bytes);
    const int nthreads = block.x * block.y * block.z;
    gridDim = grid;
    blockDim = block;
    ensure_fibers(nthreads);
    static BlockState one;
    one.lds = g_lds_one;
    g_body = &body;
    const unsigned long long nblocks = (unsigned long long)grid.x * grid.y * grid.z;
    for (unsigned long long seq = 0; seq < nblocks; ++seq) {
        // block execution order: ascending (default), descending, or a fixed pseudo-random permutation -- kernels
        // whose blocks hand work to "the last block to arrive" must give the same bits in every order
        unsigned long long id = seq;
        if (g_block_order == 1) id = nblocks - 1 - seq;
        else if (g_block_order == 2) {
            unsigned long long stride = 1;                 // a stride coprime to nblocks visits every block once
            for (unsigned long long c = nblocks / 2 + 1; c < nblocks; ++c) {
                unsigned long long a = c, b = nblocks;
                while (b) { const unsigned long long t = a % b; a = b; b = t; }
                if (a == 1) { stride = c; break; }
            }
            id = (seq * stride + 7) % nblocks;
        }
        one.bid = {(unsigned)(id % grid.x), (unsigned)((id / grid.x) % grid.y), (unsigned)(id / ((unsigned long long)grid.x * grid.y))};
        // poison LDS between blocks so stale reuse is visible (NaN pattern)
        memset(g_lds_one, 0xFF, smem_bytes);
        one.live = nthreads;
        one.bar_arrived = 0;
        one.waves.assign(nthreads / 64, WaveState());
        for (auto& w : one.waves) w.live = 64;
        for (int t = 0; t < nthreads; ++t) make_fiber(g_fibers[t], &one, t, block);
        run_fibers(nthreads, nthreads);
    }
    g_body = nullptr;
    aae_emu_dyn_smem = g_lds_one;
}

// Every block of the grid alive at once (own LDS image, own barrier state), threads of all blocks interleaved: what a
// persistent kernel with grid-wide waits needs.  Small 1-D grids only (tests pick a handful of blocks).  The interleaving
// follows the block order setting: ascending, descending or scrambled fiber order.
void launch_resident(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    check_launch(block, smem_bytes);
    const int nthreads = block.x * block.y * block.z;
    const int nblocks = (int)(grid.x * grid.y * grid.z);
    if (nblocks > 16) {
        fprintf(stderr, "emu: resident launch of %d blocks (at most 16 are emulated side by side)\n", nblocks);
        abort();
    }
    gridDim = grid;
    blockDim = block;
    ensure_fibers((size_t)nthreads * nblocks);
    std::vector<BlockState> blocks(nblocks);
    g_body = &body;
    for (int seq = 0; seq < nblocks; ++seq) {
        int id = seq;
        if (g_block_order == 1) id = nblocks - 1 - seq;
        else if (g_block_order == 2) id = nblocks <= 2 ? nblocks - 1 - seq : (seq * (nblocks % 2 == 0 ? nblocks - 1 : 2) + 1) % nblocks;   // (a permutation: the multiplier is coprime to nblocks)
        BlockState& bs = blocks[seq];
        bs.bid = {(unsigned)(id % grid.x), (unsigned)((id / grid.x) % grid.y), (unsigned)(id / (grid.x * grid.y))};
        bs.lds = (unsigned char*)malloc(sizeof(g_lds_one));
        memset(bs.lds, 0xFF, sizeof(g_lds_one));
        bs.live = nthreads;
        bs.waves.assign(nthreads / 64, WaveState());
        for (auto& w : bs.waves) w.live = 64;
        for (int t = 0; t < nthreads; ++t) make_fiber(g_fibers[(size_t)seq * nthreads + t], &bs, t, block);
    }
    run_fibers(nthreads * nblocks, nthreads);
    for (auto& bs : blocks) free(bs.lds);
    g_body = nullptr;
    aae_emu_dyn_smem = g_lds_one;
}

}  // namespace aae_emu

extern "C" void aae_emu_set_block_order(int order) { g_block_order = order; }
