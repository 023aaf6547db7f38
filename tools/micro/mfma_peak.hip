// Micro-benchmark: what this chip sustains on the matrix pipe alone.  N workgroups (one per CU, 2 waves per SIMD), each wave
// issues independent back-to-back MFMAs on register operands with random contents (no LDS, no memory) for several
// milliseconds; reports TFLOP/s and the clock implied by the instruction's issue interval (16 cycles for 16x16x32 f16 / bf16,
// 32 for 16x16x128 f8).   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/micro/mfma_peak.hip -o tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

__device__ inline unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

template <int MODE>   // 0 f16, 1 bf16, 2 fp8
__global__ __launch_bounds__(512) void peak(float* out, int iters) {
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 12345u;
    f16x8 ah[4], bh[4]; bf16x8 ab[4], bb[4]; v8i a8[4], b8[4];
    for (int f = 0; f < 4; ++f)
        for (int i = 0; i < 8; ++i) {
            const float x = ((int)(rnd(s) >> 8) % 2001 - 1000) * 1e-3f, y = ((int)(rnd(s) >> 8) % 2001 - 1000) * 1e-3f;
            ah[f][i] = (_Float16)x; bh[f][i] = (_Float16)y; ab[f][i] = (__bf16)x; bb[f][i] = (__bf16)y;
            a8[f][i] = (int)(rnd(s) & 0x7e7e7e7e); b8[f][i] = (int)(rnd(s) & 0x7e7e7e7e);    // finite e4m3 bytes
        }
    f32x4 acc[16];
    for (int n = 0; n < 16; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            if (MODE == 0) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[n & 3], bh[(n >> 2) & 3], acc[n], 0, 0, 0);
            else if (MODE == 1) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[n & 3], bb[(n >> 2) & 3], acc[n], 0, 0, 0);
            else acc[n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[n & 3], b8[(n >> 2) & 3], acc[n], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
    }
    float r = 0;
    for (int n = 0; n < 16; ++n) r += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char* name, int wgs, float* out, double flop_per_mfma, int cyc_per_mfma) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;                                   // 640 k MFMAs per wave: ~6-13 ms
    hipLaunchKernelGGL(peak<MODE>, dim3(wgs), dim3(512), 0, 0, out, 2000);
    hipDeviceSynchronize();
    double best = 1e30, tot = 0; const int reps = 4;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(peak<MODE>, dim3(wgs), dim3(512), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        tot += ms; if (ms < best) best = ms;
    }
    const double ms = tot / reps, n_wave = 16.0 * iters;
    const double flops = n_wave * 8 * wgs * flop_per_mfma;     // 8 waves per workgroup
    // 2 waves share a SIMD: per SIMD 2 * n_wave MFMAs, cyc_per_mfma cycles each
    printf("%-5s %3d workgroups: %7.2f ms (best %7.2f)  %8.1f TFLOP/s   implied matrix-pipe clock %.2f GHz\n", name, wgs, ms, best,
           flops / (ms * 1e-3) / 1e12, 2 * n_wave * cyc_per_mfma / (ms * 1e-3) / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int grids[] = {32, 64, 128, 256};
    for (int g : grids) run<0>("f16", g, out, 2.0 * 16 * 16 * 32, 16);
    for (int g : grids) run<1>("bf16", g, out, 2.0 * 16 * 16 * 32, 16);
    for (int g : grids) run<2>("fp8", g, out, 2.0 * 16 * 16 * 128, 32);
    return 0;
}
