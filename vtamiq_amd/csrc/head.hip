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

enum { HEAD_LINEAR = 0, HEAD_PRELU = 1, HEAD_RELU = 2, HEAD_GATE = 3, HEAD_CONVCAT = 4 };

// Sum 8 per-lane partials across the 64 lanes with a transposing butterfly: 4 + 2 + 1 exchanges halve the value count while
// doubling the lanes summed, 3 plain steps finish; lane l ends with the total of v[l >> 3] (10 shuffles instead of 48).
__device__ __forceinline__ float reduce8(float (&v)[8], int lane) {
#pragma unroll
    for (int step = 0; step < 3; ++step) {
        const int off = 32 >> step, cnt = 4 >> step;
        const bool upper = (lane & off) != 0;
#pragma unroll
        for (int i = 0; i < cnt; ++i) {
            const float send = upper ? v[i] : v[i + cnt];
            const float keep = upper ? v[i + cnt] : v[i];
            v[i] = keep + __shfl_xor(send, off, 64);
        }
    }
    float r = v[0];
    r += __shfl_xor(r, 4, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 1, 64);
    return r;
}

// y[b][n] = epi(sum_k W[n][k] * pre(x[b][k]) + bias[n]);  K % 4 == 0, K <= 1024.
//   grid (ceil(N/4), ceil(B/8)): one wave = one output channel n x 8 batch rows; the weight row stays in registers, the
//   8 activation rows are independent coalesced float4 streams (8 loads in flight), one butterfly reduces all 8 sums.
//   epi: LINEAR v (+ res[b][n]) | PRELU prelu(v, *slope) | RELU max(v,0) | GATE res[b][n] + aux[b][n] * sigmoid(v)
//        CONVCAT (RCAB conv with the channel-attention squeeze folded in, N = H + hid): n < nsplit -> y[b][n] = v,
//                 n >= nsplit -> y2[b][n - nsplit] = relu(v)
template <int EPI>
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ bias, const float* __restrict__ pre_slope,
                                                           const float* __restrict__ post_slope, const float* __restrict__ res,
                                                           const float* __restrict__ aux, float* __restrict__ y, float* __restrict__ y2,
                                                           int B, int N, int K, int nsplit) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int b0 = blockIdx.y * 8;
    const int K4 = K >> 2;
    float4 w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        w[i] = idx < K4 ? ((const float4*)(W + (int64_t)n * K))[idx] : float4{0.f, 0.f, 0.f, 0.f};
    }
    const bool has_pre = pre_slope != nullptr;
    const float a_pre = has_pre ? *pre_slope : 0.f;
    float acc[8];
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) {
        const int b = (b0 + bb < B) ? b0 + bb : B - 1;
        const float4* xr = (const float4*)(x + (int64_t)b * K);
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i;
            if (idx < K4) {
                float4 v = xr[idx];
                if (has_pre) { v.x = prelu(v.x, a_pre); v.y = prelu(v.y, a_pre); v.z = prelu(v.z, a_pre); v.w = prelu(v.w, a_pre); }
                a += (w[i].x * v.x + w[i].y * v.y) + (w[i].z * v.z + w[i].w * v.w);
            }
        }
        acc[bb] = a;
    }
    const float mine = reduce8(acc, lane);
    const int b = b0 + (lane >> 3);
    if ((lane & 7) == 0 && b < B) {
        float v = mine + bias[n];
        if constexpr (EPI == HEAD_CONVCAT) {
            if (n < nsplit) y[(int64_t)b * nsplit + n] = v;
            else y2[(int64_t)b * (N - nsplit) + (n - nsplit)] = fmaxf(v, 0.f);
        } else {
            const int64_t o = (int64_t)b * N + n;
            if constexpr (EPI == HEAD_PRELU) v = prelu(v, *post_slope);
            else if constexpr (EPI == HEAD_RELU) v = fmaxf(v, 0.f);
            else if constexpr (EPI == HEAD_GATE) v = res[o] + aux[o] * (1.0f / (1.0f + expf(-v)));
            else if (res) v += res[o];
            y[o] = v;
        }
    }
}

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

hipError_t launch_small_linear(const float* x, const float* W, const float* bias, const float* pre_slope,
                               const float* post_slope, const float* res, float* y, int B, int N, int K, hipStream_t s) {
    if (K % 4 || K > 1024) return hipErrorInvalidValue;
    const dim3 g((N + 3) / 4, (B + 7) / 8), blk(256);
    if (post_slope) hipLaunchKernelGGL(small_linear_kernel<HEAD_PRELU>, g, blk, 0, s, x, W, bias, pre_slope, post_slope, res, nullptr, y, nullptr, B, N, K, 0);
    else hipLaunchKernelGGL(small_linear_kernel<HEAD_LINEAR>, g, blk, 0, s, x, W, bias, pre_slope, post_slope, res, nullptr, y, nullptr, B, N, K, 0);
    return hipGetLastError();
}

// RCAB (channel_attention.py:41-50, 77-86) in two launches with the folded weights of launch_fold_ca:
//   [c | t] = [Wc ; Wd Wc] prelu(r) + bcat, t = relu(.)        out = r + c * sigmoid(Wu t + bu)
hipError_t launch_rcab(const float* r, const float* slope, const float* Wcat, const float* bcat, const float* Wu, const float* bu,
                       float* c, float* t, float* out, int B, int H, int hid, hipStream_t s) {
    if (H > 1024 || H % 4 || hid > 1024 || hid % 4) return hipErrorInvalidValue;
    const dim3 blk(256);
    hipLaunchKernelGGL(small_linear_kernel<HEAD_CONVCAT>, dim3((H + hid + 3) / 4, (B + 7) / 8), blk, 0, s, r, Wcat, bcat, slope, nullptr,
                       nullptr, nullptr, c, t, B, H + hid, H, H);
    hipLaunchKernelGGL(small_linear_kernel<HEAD_GATE>, dim3((H + 3) / 4, (B + 7) / 8), blk, 0, s, t, Wu, bu, nullptr, nullptr, r, c, out,
                       nullptr, B, H, hid, 0);
    return hipGetLastError();
}

hipError_t launch_fold_ca(const float* Wc, const float* bc, const float* Wd, const float* bd, float* Wcat, float* bcat, int H, int hid,
                          hipStream_t s) {
    hipLaunchKernelGGL(fold_ca_kernel, dim3((H + 255) / 256, H + hid), dim3(256), 0, s, Wc, bc, Wd, bd, Wcat, bcat, H, hid);
    return hipGetLastError();
}

}  // namespace vtq
