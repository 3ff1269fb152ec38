// Fiber scheduler behind tests/emu/hip_emu.h (TEST INFRASTRUCTURE ONLY).
#include "hip_emu.h"

#include <stdio.h>
#include <ucontext.h>
#include <vector>

uint3_emu threadIdx, blockIdx;
dim3 blockDim, gridDim;
alignas(64) unsigned char aae_emu_dyn_smem[160 * 1024];

namespace {

constexpr size_t kStackBytes = 256 * 1024;

struct Fiber {
    ucontext_t ctx;
    unsigned char* stack = nullptr;
    bool done = false;
    uint3_emu tid;
};

struct WaveState {
    int arrived = 0;
    int live = 0;
    unsigned gen = 0;
    alignas(16) unsigned char buf[2][64][64];
};

ucontext_t g_sched;
std::vector<Fiber> g_fibers;
std::vector<WaveState> g_waves;
Fiber* g_cur = nullptr;
const std::function<void()>* g_body = nullptr;
int g_live = 0;
int g_bar_arrived = 0;
unsigned g_bar_gen = 0;
unsigned long g_events = 0;   // arrivals / releases / exits, for deadlock detection
int g_block_order = 0;        // 0 ascending, 1 descending, 2 scrambled (aae_emu_set_block_order)

void yield_to_scheduler() {
    Fiber* me = g_cur;
    swapcontext(&me->ctx, &g_sched);
    threadIdx = me->tid;
}

void fiber_entry() {
    (*g_body)();
    g_cur->done = true;
    swapcontext(&g_cur->ctx, &g_sched);
}

int flat_tid(const uint3_emu& t) { return t.x + blockDim.x * (t.y + blockDim.y * t.z); }

}  // namespace

void __syncthreads() {
    ++g_events;
    if (++g_bar_arrived >= g_live) {
        g_bar_arrived = 0;
        ++g_bar_gen;
        return;
    }
    const unsigned gen = g_bar_gen;
    while (g_bar_gen == gen) yield_to_scheduler();
}

namespace aae_emu {

const lane_slot* wave_exchange(const void* mine, int nbytes) {
    const int ft = flat_tid(threadIdx);
    WaveState& w = g_waves[ft >> 6];
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
    if (smem_bytes > sizeof(aae_emu_dyn_smem)) {
        fprintf(stderr, "emu: dynamic LDS request %zu exceeds 160 KiB\n", smem_bytes);
        abort();
    }
    const int nthreads = block.x * block.y * block.z;
    if (nthreads % 64 != 0) {
        fprintf(stderr, "emu: block size %d not a multiple of the 64-lane wave\n", nthreads);
        abort();
    }
    gridDim = grid;
    blockDim = block;
    if ((int)g_fibers.size() < nthreads) {
        const size_t old = g_fibers.size();
        g_fibers.resize(nthreads);
        for (size_t i = old; i < g_fibers.size(); ++i) g_fibers[i].stack = (unsigned char*)malloc(kStackBytes);
    }
    g_waves.assign(nthreads / 64, WaveState());
    g_body = &body;
    const unsigned long long nblocks = (unsigned long long)grid.x * grid.y * grid.z;
    for (unsigned long long seq = 0; seq < nblocks; ++seq) {
        {
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
            const unsigned bx = (unsigned)(id % grid.x), by = (unsigned)((id / grid.x) % grid.y), bz = (unsigned)(id / ((unsigned long long)grid.x * grid.y));
            blockIdx = {bx, by, bz};
        }
        {
            // poison LDS between blocks so stale reuse is visible (NaN pattern)
            memset(aae_emu_dyn_smem, 0xFF, smem_bytes);
            g_live = nthreads;
            g_bar_arrived = 0;
            for (auto& w : g_waves) { w.arrived = 0; w.live = 64; }
            for (int t = 0; t < nthreads; ++t) {
                Fiber& f = g_fibers[t];
                f.done = false;
                f.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
                getcontext(&f.ctx);
                f.ctx.uc_stack.ss_sp = f.stack;
                f.ctx.uc_stack.ss_size = kStackBytes;
                f.ctx.uc_link = &g_sched;
                makecontext(&f.ctx, fiber_entry, 0);
            }
            int remaining = nthreads;
            while (remaining > 0) {
                const unsigned long events_before = g_events;
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = g_fibers[t];
                    if (f.done) continue;
                    g_cur = &f;
                    threadIdx = f.tid;
                    swapcontext(&g_sched, &f.ctx);
                    if (f.done) {
                        --remaining;
                        --g_live;
                        --g_waves[t >> 6].live;
                        ++g_events;
                        // an exiting thread may complete a pending barrier / collective
                        if (g_live > 0 && g_bar_arrived >= g_live) { g_bar_arrived = 0; ++g_bar_gen; }
                        WaveState& w = g_waves[t >> 6];
                        if (w.live > 0 && w.arrived >= w.live) { w.arrived = 0; ++w.gen; }
                    }
                }
                if (g_events == events_before) {
                    fprintf(stderr, "emu: deadlock in block (%u,%u,%u): a barrier/collective never completes\n",
                            blockIdx.x, blockIdx.y, blockIdx.z);
                    abort();
                }
            }
        }
    }
    g_body = nullptr;
}

}  // namespace aae_emu

extern "C" void aae_emu_set_block_order(int order) { g_block_order = order; }
