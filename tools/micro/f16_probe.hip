// Probe (GPU box): does the f16 MFMA keep subnormal inputs, and does f32 -> f16 conversion produce them?
// Decides whether the fp16 hi/lo operand split needs power-of-two pre-scaling.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(float a_val, float b_val, float* out, unsigned short* bits) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    f32x16 d;
    for (int i = 0; i < 16; ++i) d[i] = 0.f;
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = c[0]; out[1] = d[0];
        _Float16 h = (_Float16)a_val;
        bits[0] = *(unsigned short*)&h;
        // hi/lo split of a small value: lo must survive
        float v = 0.0123456789f;
        _Float16 hi = (_Float16)v; _Float16 lo = (_Float16)(v - (float)hi);
        out[2] = (float)hi; out[3] = (float)lo; out[4] = v - ((float)hi + (float)lo);
    }
}

int main() {
    float* out; unsigned short* bits;
    hipMalloc(&out, 64); hipMalloc(&bits, 16);
    const float tests[][2] = {{9.5367431640625e-07f /*2^-20*/, 1024.f}, {5.9604644775390625e-08f /*2^-24, smallest subnormal*/, 1024.f},
                              {6.103515625e-05f /*2^-14, min normal*/, 1024.f}, {1.0f, 9.5367431640625e-07f}};
    for (auto& t : tests) {
        hipMemset(out, 0, 64);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, t[0], t[1], out, bits);
        float h[8]; unsigned short hb[2];
        hipMemcpy(h, out, 32, hipMemcpyDeviceToHost); hipMemcpy(hb, bits, 4, hipMemcpyDeviceToHost);
        const double want16 = 32.0 * (double)t[0] * (double)t[1], want32 = 16.0 * (double)t[0] * (double)t[1];
        printf("a=%.3e b=%.3e : mfma16x16x32 -> %.6e (exact %.6e)  mfma32x32x16 -> %.6e (exact %.6e)  f16 bits of a = 0x%04x  split: hi=%.9g lo=%.9g resid=%.3e\n",
               t[0], t[1], h[0], want16, h[1], want32, hb[0], h[2], h[3], h[4]);
    }
    return 0;
}
