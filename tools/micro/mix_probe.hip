// Probe (GPU box): split4_f16 (v_cvt_pk_f16_f32 + v_fma_mixlo/mixhi_f16, dev_common.h) against split2 (cast, subtract, cast) bit for bit:
// random values over the fp16 range, exact ties of the hi rounding, values whose lo part is an fp16 subnormal, zeros, negatives.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vtamiq_amd/csrc tools/micro/mix_probe.hip -o tools/micro/mix_probe && tools/micro/mix_probe
#include "dev_common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k(const float* in, uint32_t* a, uint32_t* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * i + 3 >= n) return;
    float v[4];
    for (int j = 0; j < 4; ++j) { v[j] = in[4 * i + j]; asm volatile("" : "+v"(v[j])); }
    f16x4 h, l;
    split4_f16(v, h, l);
    typedef __attribute__((ext_vector_type(2))) uint32_t u2;
    const u2 hb = __builtin_bit_cast(u2, h), lb = __builtin_bit_cast(u2, l);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f16 x, y;
        split2<f16>(v[j], x, y);
        const uint32_t hj = (hb[j >> 1] >> (16 * (j & 1))) & 0xffffu, lj = (lb[j >> 1] >> (16 * (j & 1))) & 0xffffu;
        a[4 * i + j] = (hj << 16) | lj;
        b[4 * i + j] = ((uint32_t)__builtin_bit_cast(unsigned short, x) << 16) | __builtin_bit_cast(unsigned short, y);
    }
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const int kind = i & 7;
        const float u = (float)rand() / RAND_MAX, w = (float)rand() / RAND_MAX;
        float v;
        if (kind == 0) v = (u - 0.5f) * 2.0f;                                   // [-1, 1]
        else if (kind == 1) v = (u - 0.5f) * 131000.0f;                          // the fp16 range
        else if (kind == 2) v = ldexpf(1.0f + floorf(u * 1024.0f) / 1024.0f + 1.0f / 2048.0f, (int)(w * 30) - 15);   // exact ties of the hi rounding
        else if (kind == 3) v = ldexpf(u, -14 - (int)(w * 12));                  // hi (and lo) in the fp16 subnormal range
        else if (kind == 4) v = (i & 8) ? 0.0f : -0.0f;
        else if (kind == 5) v = ldexpf(1.0f + u, (int)(w * 8) - 4) * ((i & 16) ? -1.f : 1.f);
        else if (kind == 6) v = expf(-20.0f * u);                                // softmax-like
        else v = (u - 0.5f) * 20.0f;
        h[i] = v;
    }
    float* d; uint32_t *da, *db;
    (void)hipMalloc(&d, n * 4); (void)hipMalloc(&da, n * 4); (void)hipMalloc(&db, n * 4);
    (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 4 / 256), dim3(256), 0, 0, d, da, db, n);
    std::vector<uint32_t> a(n), b(n);
    (void)hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i)
        if (a[i] != b[i]) { if (bad < 10) printf("MISMATCH v=%.9g  mix hi/lo %04x %04x  split2 %04x %04x\n", h[i], a[i] >> 16, a[i] & 0xffff, b[i] >> 16, b[i] & 0xffff); ++bad; }
    printf("split4_f16 vs split2 on %d values: %ld mismatches\n", n, bad);
    return bad != 0;
}
