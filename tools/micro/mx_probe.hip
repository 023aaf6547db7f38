// Probe (GPU box): v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit block scales.
//   D[i][j] = sum_k A[i][k] * B[j][k]  (both operands K-contiguous rows, lane l supplies row l&15, bytes 32*(l>>4) .. +31)
// Checks (1) the builtin's signature / format selectors, (2) that "same byte range in both operands" pairs the same k,
// (3) the C/D layout (col = lane&15, row = 4*(lane>>4) + reg, as for the 16-bit 16x16 forms), (4) the e4m3 conversion builtin.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const uint8_t* A, const uint8_t* B, float* D, const float* fin, uint8_t* fout) {
    const int lane = threadIdx.x;
    const int row = lane & 15, g = lane >> 4;
    v8i a, b;
    const int* pa = (const int*)(A + row * 128 + g * 32);
    const int* pb = (const int*)(B + row * 128 + g * 32);
    for (int i = 0; i < 8; ++i) { a[i] = pa[i]; b[i] = pb[i]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    // (a, b, c, cbsz = A format, blgp = B format, opsel_a, scale_a, opsel_b, scale_b); format 0 = fp8 e4m3; scale 127 = 2^0
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + row] = c[r];      // D[i = 4g + r][j = lane & 15]
    // conversion probe: fp32 -> e4m3 (two values per call), RNE, saturating?
    if (lane < 8) {
        const float x0 = fin[2 * lane], x1 = fin[2 * lane + 1];
        const int packed = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, 0, false);
        fout[2 * lane] = packed & 0xFF; fout[2 * lane + 1] = (packed >> 8) & 0xFF;
    }
}

static float e4m3_to_float(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 0) f = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) f = NAN;
    else f = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}

int main() {
    uint8_t hA[16 * 128], hB[16 * 128];
    // asymmetric small-integer data, exactly representable in e4m3: values 0..7 * 2^e
    unsigned seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 16; };
    for (int i = 0; i < 16 * 128; ++i) { hA[i] = (uint8_t)(((rnd() % 3 + 6) << 3) | (rnd() % 8) | ((rnd() & 1) << 7)); hB[i] = (uint8_t)(((rnd() % 3 + 5) << 3) | (rnd() % 8) | ((rnd() & 1) << 7)); }
    uint8_t *dA, *dB, *dfo; float *dD, *dfi;
    float hfi[16] = {1.0f, 0.3f, 448.0f, 500.0f, 1e-3f, 0.0019f, -0.07f, 17.3f, 3.14159f, 2.5e-4f, 240.1f, 0.0625f, 1.0625f, 1.1875f, -460.f, 0.001953125f};
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, 256 * 4); (void)hipMalloc(&dfi, 64); (void)hipMalloc(&dfo, 16);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    (void)hipMemcpy(dfi, hfi, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, dfi, dfo);
    float hD[256]; uint8_t hfo[16];
    (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); (void)hipMemcpy(hfo, dfo, 16, hipMemcpyDeviceToHost);
    double maxerr = 0, maxerr_t = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0;
            for (int kk = 0; kk < 128; ++kk) ref += (double)e4m3_to_float(hA[i * 128 + kk]) * e4m3_to_float(hB[j * 128 + kk]);
            maxerr = fmax(maxerr, fabs(hD[i * 16 + j] - ref) / fmax(1.0, fabs(ref)));
            maxerr_t = fmax(maxerr_t, fabs(hD[j * 16 + i] - ref) / fmax(1.0, fabs(ref)));
        }
    printf("mfma_scale 16x16x128 e4m3: max rel err with D[i][j] = A_i . B_j: %.3e   (transposed reading: %.3e)\n", maxerr, maxerr_t);
    for (int i = 0; i < 16; ++i) printf("cvt_pk_fp8_f32(%g) = 0x%02x = %g\n", hfi[i], hfo[i], e4m3_to_float(hfo[i]));
    return 0;
}
