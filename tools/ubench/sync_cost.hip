// Micro-benchmark (MI355X): what does a cross-block hand-off inside one launch cost?
//   hipcc --offload-arch=gfx950 -O3 -o sync_cost sync_cost.hip && ./sync_cost
// Variants: nothing / __threadfence only / fetch-add ticket / CAS-loop ticket / sc1 (agent-scope relaxed atomic) stores
// + workgroup barrier + fetch-add ticket / two-level tickets.  Each block writes 1 KiB of "partials" first.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_plain(float* part, unsigned long long* word, int* out) {
    part[blockIdx.x * 256 + threadIdx.x] = (float)threadIdx.x;
}
__global__ void k_fence(float* part, unsigned long long* word, int* out) {
    part[blockIdx.x * 256 + threadIdx.x] = (float)threadIdx.x;
    __threadfence();
}
__global__ void k_fence_add(float* part, unsigned long long* word, int* out) {
    part[blockIdx.x * 256 + threadIdx.x] = (float)threadIdx.x;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long old = atomicAdd(word, 1ull);
        if (old + 1 == gridDim.x) { *word = 0; out[0] = 1; }
    }
}
__global__ void k_fence_cas(float* part, unsigned long long* word, int* out) {
    part[blockIdx.x * 256 + threadIdx.x] = (float)threadIdx.x;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            const unsigned long long want = (seen + 1 == gridDim.x) ? 0ull : seen + 1;
            const unsigned long long prev = atomicCAS(word, seen, want);
            if (prev == seen) { if (want == 0) out[0] = 1; break; }
            seen = prev;
        }
    }
}
// sc1 stores (agent-scope relaxed atomic store = write-through, no L2 write-back instruction), workgroup barrier, fetch-add
__global__ void k_sc1_add(float* part, unsigned long long* word, int* out) {
    __hip_atomic_store(part + blockIdx.x * 256 + threadIdx.x, (float)threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long old = __hip_atomic_fetch_add(word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == gridDim.x) { __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); out[0] = 1; }
    }
}
// the same, and the last block reads every block's partials back with sc1 loads and sums them
__global__ void k_sc1_add_read(float* part, unsigned long long* word, int* out) {
    __shared__ int last;
    __hip_atomic_store(part + blockIdx.x * 256 + threadIdx.x, (float)threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long old = __hip_atomic_fetch_add(word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (old + 1 == gridDim.x);
        if (last) __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last) return;
    float s = 0.f;
    for (unsigned b = 0; b < gridDim.x; ++b) s += __hip_atomic_load(part + b * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[1 + threadIdx.x] = (int)s;
}
// two-level: 16 group words (blockIdx & 15), the last of a group arrives at the top word
__global__ void k_sc1_add2(float* part, unsigned long long* word, int* out) {
    __hip_atomic_store(part + blockIdx.x * 256 + threadIdx.x, (float)threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = blockIdx.x & 15, members = (gridDim.x - g + 15) / 16;
        unsigned long long old = __hip_atomic_fetch_add(word + 16 * (1 + g), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == members) {
            __hip_atomic_store(word + 16 * (1 + g), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned groups = gridDim.x < 16 ? gridDim.x : 16;
            unsigned long long o2 = __hip_atomic_fetch_add(word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (o2 + 1 == groups) { __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); out[0] = 1; }
        }
    }
}
// release/acquire fences at agent scope written as the builtin (what __threadfence() expands to) but only in thread 0
__global__ void k_fence_t0_add(float* part, unsigned long long* word, int* out) {
    part[blockIdx.x * 256 + threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        unsigned long long old = atomicAdd(word, 1ull);
        if (old + 1 == gridDim.x) { *word = 0; out[0] = 1; }
    }
}

typedef void (*kern_t)(float*, unsigned long long*, int*);

int main() {
    float* part; unsigned long long* word; int* out;
    CHECK(hipMalloc(&part, 4096 * 256 * sizeof(float)));
    CHECK(hipMalloc(&word, 4096));
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(word, 0, 4096));
    CHECK(hipMemset(out, 0, 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct { const char* name; kern_t k; } ks[] = {
        {"plain stores", k_plain}, {"+ threadfence (all threads)", k_fence}, {"+ threadfence + fetch-add", k_fence_add},
        {"+ threadfence + CAS loop", k_fence_cas}, {"barrier, thread-0 fence + fetch-add", k_fence_t0_add},
        {"sc1 stores + barrier + fetch-add", k_sc1_add}, {"sc1 ... + last block reads all back", k_sc1_add_read},
        {"sc1 stores + two-level fetch-add", k_sc1_add2}};
    const int grids[] = {64, 256, 721, 2048};
    for (auto& kk : ks)
        for (int g : grids) {
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kk.k, dim3(g), dim3(256), 0, 0, part, word, out);
            CHECK(hipDeviceSynchronize());
            const int reps = 50;
            CHECK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kk.k, dim3(g), dim3(256), 0, 0, part, word, out);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long w = 1;
            CHECK(hipMemcpy(&w, word, 8, hipMemcpyDeviceToHost));
            printf("{\"what\": \"sync_cost\", \"variant\": \"%s\", \"blocks\": %d, \"us_per_launch\": %.2f, \"word_after\": %llu}\n", kk.name, g, ms / reps * 1e3, w);
        }
    return 0;
}
