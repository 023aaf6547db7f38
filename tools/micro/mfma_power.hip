// Micro-benchmark: what sets the clock the chip sustains under a pure MFMA stream (the 3-term GEMM K loops are power-bound: DESIGN.md section 8).
// 256 workgroups, 8 waves (2 per SIMD) or 4 waves (1 per SIMD), back-to-back independent MFMAs on register operands, no LDS, no memory.
// Varied: the instruction shape (16x16x32 vs 32x32x16 f16: the same flops per cycle, half the operand and accumulator register traffic per flop),
// the operand CONTENTS (uniform random; zeros; a hi/lo pair like the 3-term product's planes), the operand reuse pattern.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/micro/mfma_power.hip -o tools/micro/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ inline float gauss(unsigned& s) {
    float a = 0; for (int i = 0; i < 12; ++i) a += (rnd(s) >> 8) * (1.f / 16777216.f); return a - 6.f;
}

// DATA: 0 uniform random in [-1,1]; 1 zeros; 2 gaussian hi planes (a = hi of N(0,1), b = hi of N(0,0.02)); 3 hi x lo (b = lo residual plane);
//       4 the 3-term mix (of every 3 MFMAs: hi*hi, hi*lo, lo*hi)
template <int SHAPE, int WAVES>   // SHAPE 0: 16x16x32, 1: 32x32x16, 2: 16x16x32 with the A operand held over 4 consecutive MFMAs
__global__ __launch_bounds__(WAVES * 64) void power(float* out, int iters, int data) {
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 12345u;
    f16x8 ahi[4], alo[4], bhi[4], blo[4];
    for (int f = 0; f < 4; ++f)
        for (int i = 0; i < 8; ++i) {
            float x, y;
            if (data == 0) { x = ((int)(rnd(s) >> 8) % 2001 - 1000) * 1e-3f; y = ((int)(rnd(s) >> 8) % 2001 - 1000) * 1e-3f; }
            else if (data == 1) { x = 0.f; y = 0.f; }
            else { x = gauss(s); y = 0.02f * gauss(s); }
            ahi[f][i] = (_Float16)x; alo[f][i] = (_Float16)(x - (float)ahi[f][i]);
            bhi[f][i] = (_Float16)y; blo[f][i] = (_Float16)(y - (float)bhi[f][i]);
        }
    f16x8 A[3][4], Bm[3][4];     // per term
    for (int f = 0; f < 4; ++f) {
        A[0][f] = ahi[f]; Bm[0][f] = bhi[f];
        A[1][f] = (data >= 3) ? ahi[f] : ahi[f]; Bm[1][f] = (data >= 3) ? blo[f] : bhi[f];
        A[2][f] = (data == 4) ? alo[f] : A[1][f]; Bm[2][f] = (data == 4) ? bhi[f] : Bm[1][f];
        if (data == 3) { A[0][f] = ahi[f]; Bm[0][f] = blo[f]; }
    }
    float r = 0;
    if (SHAPE == 1) {
        f32x16 acc[4];
        for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int n = 0; n < 4; ++n)      // 12 MFMAs of 32x32x16 = 24 of 16x16x32 in flops
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[t][n & 1], Bm[t][n >> 1], acc[n], 0, 0, 0);
        }
        for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) r += acc[n][i];
    } else {
        f32x4 acc[16];
        for (int n = 0; n < 16; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int h = 0; h < 8; ++h) {     // 24 MFMAs of 16x16x32
                    const int n = (t * 8 + h) & 15;
                    if (SHAPE == 0) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t][h & 3], Bm[t][(h >> 2) & 3], acc[n], 0, 0, 0);
                    else acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t][h >> 2], Bm[t][h & 3], acc[n], 0, 0, 0);
                }
        }
        for (int n = 0; n < 16; ++n) r += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int SHAPE, int WAVES> void run(const char* name, int data, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = 256, iters = 30000 * (WAVES == 4 ? 2 : 1);
    hipLaunchKernelGGL((power<SHAPE, WAVES>), dim3(wgs), dim3(WAVES * 64), 0, 0, out, 2000, data);
    hipDeviceSynchronize();
    double best = 1e30, tot = 0; const int reps = 4;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((power<SHAPE, WAVES>), dim3(wgs), dim3(WAVES * 64), 0, 0, out, iters, data);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        tot += ms; if (ms < best) best = ms;
    }
    const double ms = tot / reps;
    const double mfma16 = 24.0 * iters;                        // per wave, in 16x16x32 equivalents (16 cycles each)
    const double flops = mfma16 * WAVES * wgs * 2.0 * 16 * 16 * 32;
    static const char* dn[] = {"uniform random", "zeros", "gaussian hi x hi", "gaussian hi x lo", "3-term mix hi*hi, hi*lo, lo*hi"};
    printf("%-34s %d waves/SIMD  %-32s %7.2f ms (best %7.2f)  %7.1f TFLOP/s   implied clock %.3f GHz\n", name, WAVES / 4, dn[data], ms, best,
           flops / (ms * 1e-3) / 1e12, (WAVES / 4) * mfma16 * 16 / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        for (int d = 0; d < 5; ++d) {
            run<0, 8>("16x16x32 A changes every MFMA", d, out);
            run<2, 8>("16x16x32 A held over 4 MFMAs", d, out);
            run<1, 8>("32x32x16", d, out);
        }
        run<0, 4>("16x16x32 A changes every MFMA", 0, out);
        run<1, 4>("32x32x16", 0, out);
        run<0, 4>("16x16x32 A changes every MFMA", 4, out);
        run<1, 4>("32x32x16", 4, out);
    }
    return 0;
}
