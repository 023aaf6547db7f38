// Micro-benchmark: is work issued in the MFMA shadow free on a chip whose clock is set by its power budget?
// All 256 CUs, 2 waves per SIMD, each wave: back-to-back 16x16x32 f16 MFMAs with F independent VALU FMAs (or F ds_read_b128)
// issued between consecutive MFMAs -- they fit the 16-cycle issue interval (3 free issue slots) for F <= 3.
// If the MFMA rate stays at the F = 0 value the shadow is free (overlap pays); if it falls, time is energy.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/micro/mfma_shadow.hip -o tools/micro/mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int F, int KIND>   // KIND 0: VALU fma fillers, 1: LDS read fillers
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[512 * 4 * 4];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 12345u;
    f16x8 a[4], b[4];
    for (int f = 0; f < 4; ++f)
        for (int i = 0; i < 8; ++i) {
            s = s * 1664525u + 1013904223u; a[f][i] = (_Float16)(((int)(s >> 8) % 2001 - 1000) * 1e-3f);
            s = s * 1664525u + 1013904223u; b[f][i] = (_Float16)(((int)(s >> 8) % 2001 - 1000) * 1e-3f);
        }
    for (int i = threadIdx.x; i < 512 * 16; i += 512) lds[i] = (float)i * 1e-3f;
    __syncthreads();
    f32x4 acc[16];
    for (int n = 0; n < 16; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = 1.0f + 1e-3f * (threadIdx.x + i);
    f32x4 l[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    const float c1 = 1.0000001f, c2 = 1e-7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[n & 3], b[(n >> 2) & 3], acc[n], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < F; ++f) {
                if (KIND == 0) { float& x = v[(n * F + f) % 12]; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2)); }
                else { f32x4& x = l[f]; const float* p = lds + ((threadIdx.x * 4 + ((it + n + f) & 3) * 2048) & (512 * 16 - 4));
                       asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"((unsigned)(size_t)p)); }
            }
        }
        if (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float r = 0;
    for (int n = 0; n < 16; ++n) r += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    for (int i = 0; i < 12; ++i) r += v[i];
    for (int i = 0; i < 3; ++i) r += l[i][0] + l[i][1] + l[i][2] + l[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int F, int KIND> void run(float* out, int wgs) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 30000;
    hipLaunchKernelGGL((k<F, KIND>), dim3(wgs), dim3(512), 0, 0, out, 2000);
    (void)hipDeviceSynchronize();
    double tot = 0; const int reps = 3;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<F, KIND>), dim3(wgs), dim3(512), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); tot += ms;
    }
    const double ms = tot / reps, n_wave = 16.0 * iters;
    printf("%s fillers per MFMA = %d, %3d workgroups: %7.2f ms  MFMA %7.1f TFLOP/s  (matrix-pipe clock if back-to-back %.2f GHz)\n",
           KIND ? "ds_read_b128" : "v_fma_f32   ", F, wgs, ms, n_wave * 8 * wgs * 16384.0 / (ms * 1e-3) / 1e12, 2 * n_wave * 16 / (ms * 1e-3) / 1e9);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    for (int wgs : {64, 256}) {
        run<0, 0>(out, wgs); run<1, 0>(out, wgs); run<2, 0>(out, wgs); run<3, 0>(out, wgs);
        run<1, 1>(out, wgs); run<2, 1>(out, wgs);
    }
    return 0;
}
