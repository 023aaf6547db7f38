// Micro-benchmark: what a device-wide barrier between dependent stages costs inside ONE launch, against the boundary between
// dependent launches.  G workgroups of 512 threads run NB rounds of: write a word, barrier, read the neighbour's word (checked).
// The barrier is count + generation in global memory (self-resetting: safe under graph replay), agent-scope fences either side
// (on gfx950 the release writes the XCD's dirty L2 lines back and the acquire invalidates L1 / non-local L2 lines: the workgroups
// of one launch sit on all eight XCDs).  A polling workgroup gives up after 2^22 polls (the run reports it) instead of hanging.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o tools/micro/grid_barrier && tools/micro/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ bool grid_barrier(unsigned* count, unsigned* gen, unsigned& my_gen, unsigned G) {
    bool ok = true;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = atomicAdd(count, 1u);
        if (old == G - 1) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int polls = 0;
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > (1 << 22)) { ok = false; break; }
            }
        }
    }
    ++my_gen;
    __syncthreads();
    __threadfence();
    return ok;
}

__global__ __launch_bounds__(512) void rounds(unsigned* count, unsigned* gen, unsigned* slots, int nb, unsigned* bad) {
    const unsigned G = gridDim.x;
    unsigned my_gen = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned wrong = 0;
    for (int k = 1; k <= nb; ++k) {
        if (threadIdx.x == 0) slots[blockIdx.x * 32] = (unsigned)k;            // 128 B apart
        if (!grid_barrier(count, gen, my_gen, G)) { wrong |= 2; break; }
        if (threadIdx.x == 64) wrong |= (slots[((blockIdx.x + 1) % G) * 32] != (unsigned)k);
        if (!grid_barrier(count, gen, my_gen, G)) { wrong |= 2; break; }       // the slot is rewritten next round
    }
    if (wrong) atomicOr(bad, wrong);
}

__global__ __launch_bounds__(512) void one_stage(unsigned* slots, int k) {
    if (threadIdx.x == 0) slots[blockIdx.x * 32] = (unsigned)k + slots[((blockIdx.x + 1) % gridDim.x) * 32];
}

int main() {
    unsigned *count, *gen, *slots, *bad;
    hipMalloc(&count, 4); hipMalloc(&gen, 4); hipMalloc(&bad, 4); hipMalloc(&slots, 512 * 128);
    hipMemset(count, 0, 4); hipMemset(gen, 0, 4); hipMemset(bad, 0, 4); hipMemset(slots, 0, 512 * 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nb = 500;
    for (int G : {8, 16, 32, 54, 64, 128, 256}) {
        hipLaunchKernelGGL(rounds, dim3(G), dim3(512), 0, 0, count, gen, slots, 10, bad);
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(rounds, dim3(G), dim3(512), 0, 0, count, gen, slots, nb, bad);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        unsigned hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        // dependent launches of the same grid
        float bestl = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(one_stage, dim3(G), dim3(512), 0, 0, slots, k);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < bestl) bestl = ms;
        }
        printf("G = %3d workgroups: %.2f us per in-kernel barrier (%d barriers, check %s)   %.2f us per dependent launch\n", G,
               best * 1e3f / (2 * nb), 2 * nb, hb == 0 ? "ok" : (hb & 2 ? "TIMEOUT" : "STALE DATA"), bestl * 1e3f / 200);
        hipMemset(bad, 0, 4); hipMemset(count, 0, 4);
    }
    return 0;
}
