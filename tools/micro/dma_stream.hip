// Micro-benchmark: what an LDS-DMA operand stream delivers per CU at a given depth, for the two layouts of an A operand.
// 256 workgroups (8 waves) each stream the 256-row panels p = wg, wg + 256, ... of a [M, K] 16-bit tensor through LDS in chunks of
// 64 columns (256 rows x 128 B = 32 KB: one 1-KiB global_load_lds_dwordx4 per wave x 4), DEPTH chunks in flight (counted vmcnt), nothing
// else: no MFMA, no LDS reads.  Layout 0 = row-major (a chunk = 256 segments of 128 B at a stride of 2 K bytes: what the encoder GEMMs
// read today); layout 1 = chunk-contiguous ([panel][chunk][row][64]: every DMA instruction reads 1 KiB of consecutive addresses, a
// chunk is 32 KB of consecutive addresses).  Reports GB/s and, by Little's law, the mean time a chunk is in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_stream.hip -o tools/micro/dma_stream && tools/micro/dma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int DEPTH, int LAYOUT>
__global__ __launch_bounds__(512) void stream(const unsigned short* __restrict__ A, int M, int K, int passes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npanel = M / 256, nchunk = K / 64;
    const int row = tid >> 3, seg = tid & 7;                    // slot tid of a 64-row quarter: row, 16-byte piece
    int issued = 0;
    for (int pass = 0; pass < passes; ++pass)
        for (int p = blockIdx.x; p < npanel; p += gridDim.x)
            for (int c = 0; c < nchunk; ++c) {
                char* dst = smem + (issued % DEPTH) * 32768 + wave * 1024;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                   // four quarters of 64 rows
                    const int r = q * 64 + row;
                    const unsigned short* src = LAYOUT == 0 ? A + ((size_t)p * 256 + r) * K + c * 64 + seg * 8
                                                            : A + (((size_t)p * nchunk + c) * 256 + r) * 64 + seg * 8;
                    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(dst + q * 8192), 16, 0, 0);
                }
                ++issued;
                if (issued >= DEPTH) {
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (DEPTH - 1)) : "memory");
                    __builtin_amdgcn_s_barrier();               // the consumer side of a GEMM tile: everyone's pieces have landed
                }
            }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) sink[blockIdx.x] = ((unsigned*)smem)[0];
}

template <int DEPTH, int LAYOUT>
void run(const unsigned short* A, int M, int K, unsigned* sink) {
    const int lds = DEPTH * 32768;
    hipFuncSetAttribute((const void*)stream<DEPTH, LAYOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int passes = 3;
    stream<DEPTH, LAYOUT><<<256, 512, lds>>>(A, M, K, 1, sink);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        stream<DEPTH, LAYOUT><<<256, 512, lds>>>(A, M, K, passes, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)M * K * 2 * passes, gbs = bytes / best / 1e6;
    const double inflight = 256.0 * DEPTH * 32768;               // bytes in flight on the chip
    printf("M=%d K=%d  %s  depth %d (%3d KB per CU in flight): %7.1f GB/s  (%.1f us per pass; mean time in flight of a chunk %.2f us)\n", M, K,
           LAYOUT == 0 ? "row-major (128-B segments, stride 2K bytes)" : "chunk-contiguous (32 KB per chunk)        ", DEPTH, DEPTH * 32,
           gbs, best * 1e3 / passes, inflight / (gbs * 1e3));
}

int main() {
    const int M = 32256;
    for (int K : {768, 3072}) {
        unsigned short* A; unsigned* sink;
        hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&sink, 1024);
        hipMemset(A, 0x3c, (size_t)M * K * 2);
        run<2, 0>(A, M, K, sink); run<2, 1>(A, M, K, sink);
        run<3, 0>(A, M, K, sink); run<3, 1>(A, M, K, sink);
        run<4, 0>(A, M, K, sink); run<4, 1>(A, M, K, sink);
        hipFree(A); hipFree(sink);
    }
    return 0;
}
