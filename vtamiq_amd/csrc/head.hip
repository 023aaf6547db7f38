// DiffNet head: the one-time weight fold of an RCAB (the stages themselves run on the MFMA pipe, skinny.hip).
//
// RCAB (modules/RCAN/channel_attention.py:41-50) = x + CA(Conv1d(PReLU(x))); CALayer (:53-62, 77-86) squeezes the conv output
// c with Conv1d(H, H/r) -> ReLU.  On a length-1 sequence both convs are matrix products, so the squeeze is folded into the
// conv: [c ; t_pre] = [Wc ; Wd Wc] prelu(x) + [bc ; Wd bc + bd]  -- one stage instead of two (exact algebra, done in fp32).
#include "dev_common.h"
#include "kernels.h"

namespace vtq {
namespace {

// one-time weight fold for an RCAB: Wcat[H + hid][H] = [Wc ; Wd Wc], bcat = [bc ; Wd bc + bd]
// (CALayer squeeze conv applied to the RCAB conv output, channel_attention.py:45 -> :58; exact algebra, done in fp32)
__global__ __launch_bounds__(256) void fold_ca_kernel(const float* __restrict__ Wc, const float* __restrict__ bc,
                                                      const float* __restrict__ Wd, const float* __restrict__ bd,
                                                      float* __restrict__ Wcat, float* __restrict__ bcat, int H, int hid) {
    const int j = blockIdx.y;                       // 0 .. H + hid - 1
    const int k = blockIdx.x * 256 + threadIdx.x;   // column
    if (j < H) {
        if (k < H) Wcat[(int64_t)j * H + k] = Wc[(int64_t)j * H + k];
        if (k == 0) bcat[j] = bc[j];
        return;
    }
    // eight independent partial sums: the 768-deep dot product is otherwise one chain of dependent load + FMA latencies
    // (196 us per RCAB at weight load); H % 8 == 0 for every supported width
    const float* wd = Wd + (int64_t)(j - H) * H;
    if (k < H) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < H; m += 8)
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = fmaf(wd[m + u], Wc[(int64_t)(m + u) * H + k], a[u]);
        Wcat[(int64_t)j * H + k] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {           // bias: one wave, lanes stride the sum, butterfly reduce
        float a = 0.f;
        for (int m = threadIdx.x; m < H; m += 64) a = fmaf(wd[m], bc[m], a);
        a = wave_sum(a);
        if (threadIdx.x == 0) bcat[j] = a + bd[j - H];
    }
}

}  // namespace

hipError_t launch_fold_ca(const float* Wc, const float* bc, const float* Wd, const float* bd, float* Wcat, float* bcat, int H, int hid,
                          hipStream_t s) {
    hipLaunchKernelGGL(fold_ca_kernel, dim3((H + 255) / 256, H + hid), dim3(256), 0, s, Wc, bc, Wd, bd, Wcat, bcat, H, hid);
    return hipGetLastError();
}

}  // namespace vtq
