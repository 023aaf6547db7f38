// DiffNet (RCAN residual channel-attention) + quality predictor on the [B, H] CLS-difference, fp32 end to end.
//
// Replaces get_quality_decoder / ResidualGroup / RCAB / CALayer (modules/vtamiq/vtamiq.py:12-23,
// modules/RCAN/channel_attention.py:13-86) and q_predictor (vtamiq.py:71-77,116-117).  Every Conv1d(k=1) on a
// (B, C, 1) tensor is a [B, C] x [C_out, C]^T product.  The whole head is 2.7e7 flop per pair (1.4e-4 of the
// forward) and weight-streaming bound (fp32 weights: 50 MB for 21 HxH convs), so it stays in exact fp32 FMAs:
// one wave per output channel keeps its weight row in registers and sweeps the B rows with coalesced float4 loads.
#include "dev_common.h"
#include "kernels.h"

namespace vtq {
namespace {

__device__ __forceinline__ float prelu(float v, float a) { return v >= 0.f ? v : a * v; }

// y[b][n] = post(sum_k W[n][k] * pre(x[b][k]) + bias[n]) + res[b][n];  K % 4 == 0, K <= 1024
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ bias, const float* __restrict__ pre_slope,
                                                           const float* __restrict__ post_slope, const float* __restrict__ res,
                                                           float* __restrict__ y, int B, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int K4 = K >> 2;
    float4 w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        w[i] = idx < K4 ? ((const float4*)(W + (int64_t)n * K))[idx] : float4{0.f, 0.f, 0.f, 0.f};
    }
    const bool has_pre = pre_slope != nullptr;
    const float a_pre = has_pre ? *pre_slope : 0.f;
    const float bn = bias[n];
    const bool has_post = post_slope != nullptr;
    const float a_post = has_post ? *post_slope : 0.f;
    for (int b = 0; b < B; ++b) {
        const float4* xr = (const float4*)(x + (int64_t)b * K);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i;
            if (idx < K4) {
                float4 v = xr[idx];
                if (has_pre) { v.x = prelu(v.x, a_pre); v.y = prelu(v.y, a_pre); v.z = prelu(v.z, a_pre); v.w = prelu(v.w, a_pre); }
                acc += (w[i].x * v.x + w[i].y * v.y) + (w[i].z * v.z + w[i].w * v.w);
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            float v = acc + bn;
            if (has_post) v = prelu(v, a_post);
            if (res) v += res[(int64_t)b * N + n];
            y[(int64_t)b * N + n] = v;
        }
    }
}

// RCAB tail (channel_attention.py:49-50, 82-86): out = r + c * sigmoid(Wu relu(Wd c + bd) + bu); one workgroup per sample.
__global__ __launch_bounds__(256) void ca_residual_kernel(const float* __restrict__ c, const float* __restrict__ r,
                                                          const float* __restrict__ Wd, const float* __restrict__ bd,
                                                          const float* __restrict__ Wu, const float* __restrict__ bu,
                                                          float* __restrict__ out, int H, int hid) {
    __shared__ __attribute__((aligned(16))) float cs[1024];
    __shared__ __attribute__((aligned(16))) float ts[256];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* cb = c + (int64_t)b * H;
    for (int i = tid; i < H; i += 256) cs[i] = cb[i];
    __syncthreads();
    const int H4 = H >> 2;
    for (int j = wave; j < hid; j += 4) {
        const float4* wr = (const float4*)(Wd + (int64_t)j * H);
        float acc = 0.f;
        for (int idx = lane; idx < H4; idx += 64) {
            const float4 wv = wr[idx];
            const float4 xv = ((const float4*)cs)[idx];
            acc += (wv.x * xv.x + wv.y * xv.y) + (wv.z * xv.z + wv.w * xv.w);
        }
        acc = wave_sum(acc);
        if (lane == 0) ts[j] = fmaxf(acc + bd[j], 0.f);
    }
    __syncthreads();
    const int hid4 = hid >> 2;
    for (int n = tid; n < H; n += 256) {
        const float4* wr = (const float4*)(Wu + (int64_t)n * hid);
        float acc = 0.f;
        for (int idx = 0; idx < hid4; ++idx) {
            const float4 wv = wr[idx];
            const float4 tv = ((const float4*)ts)[idx];
            acc += (wv.x * tv.x + wv.y * tv.y) + (wv.z * tv.z + wv.w * tv.w);
        }
        const float wgt = 1.0f / (1.0f + expf(-(acc + bu[n])));
        out[(int64_t)b * H + n] = r[(int64_t)b * H + n] + cs[n] * wgt;
    }
}

}  // namespace

hipError_t launch_small_linear(const float* x, const float* W, const float* bias, const float* pre_slope,
                               const float* post_slope, const float* res, float* y, int B, int N, int K, hipStream_t s) {
    if (K % 4 || K > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(small_linear_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, W, bias, pre_slope, post_slope, res, y, B, N, K);
    return hipGetLastError();
}

hipError_t launch_ca_residual(const float* c, const float* r, const float* Wd, const float* bd, const float* Wu, const float* bu,
                              float* out, int B, int H, int hid, hipStream_t s) {
    if (H > 1024 || H % 4 || hid > 256 || hid % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ca_residual_kernel, dim3(B), dim3(256), 0, s, c, r, Wd, bd, Wu, bu, out, H, hid);
    return hipGetLastError();
}

}  // namespace vtq
