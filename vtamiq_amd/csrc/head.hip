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
    const float* wd = Wd + (int64_t)(j - H) * H;
    if (k < H) {
        float a = 0.f;
        for (int m = 0; m < H; ++m) a += wd[m] * Wc[(int64_t)m * H + k];
        Wcat[(int64_t)j * H + k] = a;
    }
    if (k == 0) {
        float a = bd[j - H];
        for (int m = 0; m < H; ++m) a += wd[m] * bc[m];
        bcat[j] = a;
    }
}

}  // namespace

hipError_t launch_fold_ca(const float* Wc, const float* bc, const float* Wd, const float* bd, float* Wcat, float* bcat, int H, int hid,
                          hipStream_t s) {
    hipLaunchKernelGGL(fold_ca_kernel, dim3((H + 255) / 256, H + hid), dim3(256), 0, s, Wc, bc, Wd, bd, Wcat, bcat, H, hid);
    return hipGetLastError();
}

}  // namespace vtq
