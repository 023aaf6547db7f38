// Probe (GPU box): the f16 truncation split used on the attention probabilities: hi = v_cvt_pkrtz, lo = pkrtz(v_fma_mix(v - hi)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* hi_out, float* lo_out, float* r_out, int n) {
    const int i = threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = in[2 * i], b = in[2 * i + 1];
    const auto hp = __builtin_amdgcn_cvt_pkrtz(a, b);
    const unsigned hw = __builtin_bit_cast(unsigned, hp);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hw), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hw), "v"(b));
    const f16x2 h = __builtin_bit_cast(f16x2, hp);
    const f16x2 l = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(r0, r1));
    hi_out[2 * i] = (float)h[0]; hi_out[2 * i + 1] = (float)h[1];
    lo_out[2 * i] = (float)l[0]; lo_out[2 * i + 1] = (float)l[1];
    r_out[2 * i] = r0; r_out[2 * i + 1] = r1;
}
int main() {
    const int n = 16;
    float h[n] = {1.0f, 0.3333333f, 4095.77f, 1e-3f, 0.186f, 2.5e-5f, 1234.567f, 0.9999999f, 3.14159265f, 100.001f, 7e-5f, 6.2e-5f, 0.5f, 2049.5f, 1.0009766f, 65.4321f};
    float *d, *dh, *dl, *dr;
    (void)hipMalloc(&d, n * 4); (void)hipMalloc(&dh, n * 4); (void)hipMalloc(&dl, n * 4); (void)hipMalloc(&dr, n * 4);
    (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dh, dl, dr, n);
    float oh[n], ol[n], orr[n];
    (void)hipMemcpy(oh, dh, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(ol, dl, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(orr, dr, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
        printf("v=%.9g hi=%.9g r=v-hi=%.9g (fma_mix gave %.9g) lo=%.9g  rel.err of hi+lo = %.3e\n", h[i], oh[i], h[i] - oh[i], orr[i], ol[i],
               fabs((double)oh[i] + (double)ol[i] - (double)h[i]) / h[i]);
    return 0;
}
