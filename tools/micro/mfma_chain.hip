// Micro-benchmark: issue interval of v_mfma_f32_32x32x16_bf16 / 16x16x32 on ONE wave per SIMD, as a function of how many
// independent accumulators the chain alternates over.  hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k32(float* out, unsigned long long* cyc, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 24 / NACC; ++rep)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k16(float* out, unsigned long long* cyc, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = 0.f;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 24 / NACC; ++rep)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[n], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <typename K> void run(const char* name, K kern, int waves, float* out, unsigned long long* cyc) {
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 24.0 * iters;
    printf("%-28s waves/WG %d: %6.1f ticks/MFMA (wave 0), %7.2f ns/MFMA per wave by events\n", name, waves, c / n, ms * 1e6 / n);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    run("32x32x16 1 acc", k32<1, 4>, 4, out, cyc);
    run("32x32x16 2 acc", k32<2, 4>, 4, out, cyc);
    run("32x32x16 3 acc", k32<3, 4>, 4, out, cyc);
    run("32x32x16 4 acc", k32<4, 4>, 4, out, cyc);
    run("32x32x16 1 acc 8 waves", k32<1, 8>, 8, out, cyc);
    run("32x32x16 2 acc 8 waves", k32<2, 8>, 8, out, cyc);
    run("16x16x32 1 acc", k16<1, 4>, 4, out, cyc);
    run("16x16x32 2 acc", k16<2, 4>, 4, out, cyc);
    run("16x16x32 3 acc", k16<3, 4>, 4, out, cyc);
    run("16x16x32 4 acc", k16<4, 4>, 4, out, cyc);
    run("16x16x32 8 acc", k16<8, 4>, 4, out, cyc);
    run("16x16x32 1 acc 8 waves", k16<1, 8>, 8, out, cyc);
    return 0;
}
