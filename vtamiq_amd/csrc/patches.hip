// On-device image -> patch tensor: what the reference's CPU loader workers do per item, for GIVEN sample
// coordinates (the sampler's RNG stays on the host):
//   transform_img    data/utils.py:76-94          uint8 HWC -> f32 CHW / 255, hflip, vflip, (x - mean) / std
//   get_iqa_patches  data/patch_sampling.py:529-611  patch[c,i,j] = level_s[c, row+i, col+j]; pos = clamp((sample + P/2) / (dim - P/2), 0, 1-1e-6);
//                                                  level_{s+1} = AvgPool2d(2)(level_s)
// Shipping uint8 images (0.6 MB for 384x512) + coordinates instead of fp32 patches (3.08 MB / pair at N = 500) removes the
// 16 GB/s/GPU of host-side gather and H2D traffic that the patch tensor would need at engine speed.
// All three kernels are HBM-bound elementwise/gather work; arithmetic follows the torch op order bit for bit
// (true divisions, pooling sum in (0,0),(0,1),(1,0),(1,1) order then /4).
#include "dev_common.h"
#include "kernels.h"

namespace vtq {
namespace {

// out[ni][c][y][x] = ((float)in[ni][ys][xs][c] / 255 - mean[c]) / std[c], (ys, xs) = flipped source pixel
__global__ __launch_bounds__(256) void image_normalize_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int H, int W,
                                                              const int* __restrict__ flips, float m0, float m1, float m2, float s0,
                                                              float s1, float s2) {
    const int ni = blockIdx.z, c = blockIdx.y;
    const int hf = flips ? flips[ni * 2] : 0, vf = flips ? flips[ni * 2 + 1] : 0;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    const int64_t plane = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        const int ys = vf ? H - 1 - y : y, xs = hf ? W - 1 - x : x;
        const float v = (float)in[((int64_t)ni * plane + (int64_t)ys * W + xs) * 3 + c];
        out[((int64_t)ni * 3 + c) * plane + i] = (__fdiv_rn(v, 255.0f) - mean) / sd;
    }
}

// torch.nn.AvgPool2d(2) on [NC, H, W] -> [NC, H/2, W/2]
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t nc = blockIdx.y;
    const float* ip = in + nc * H * W;
    float* op = out + nc * Ho * Wo;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Ho * Wo; i += gridDim.x * blockDim.x) {
        const int y = i / Wo, x = i - y * Wo;
        const float* r0 = ip + (int64_t)(2 * y) * W + 2 * x;
        float sum = r0[0];
        sum += r0[1];
        sum += r0[W];
        sum += r0[W + 1];
        op[i] = sum / 4.0f;
    }
}

struct Levels { const float* p[4]; int h[4]; int w[4]; };

// one workgroup per (patch n, image ni): P*P threads = the pixels of a PxP patch (P = 16 | 8), loop over the 3 channels
__global__ __launch_bounds__(256) void gather_patches_kernel(Levels lv, const int* __restrict__ samples, const int* __restrict__ scale_ids,
                                                             float* __restrict__ patches, float* __restrict__ pos, float* __restrict__ scales,
                                                             int N, int write_scales, int P) {
    const int n = blockIdx.x, ni = blockIdx.y;
    const int64_t pn = (int64_t)ni * N + n;
    const int sc = scale_ids ? scale_ids[pn] : 0;
    const float* base = sc == 0 ? lv.p[0] : (sc == 1 ? lv.p[1] : (sc == 2 ? lv.p[2] : lv.p[3]));
    const int h = sc == 0 ? lv.h[0] : (sc == 1 ? lv.h[1] : (sc == 2 ? lv.h[2] : lv.h[3]));
    const int w = sc == 0 ? lv.w[0] : (sc == 1 ? lv.w[1] : (sc == 2 ? lv.w[2] : lv.w[3]));
    const int row = samples[pn * 2], col = samples[pn * 2 + 1];
    const int PP = P * P, i = threadIdx.x / P, j = threadIdx.x - i * P;
    const float* img = base + (int64_t)ni * 3 * h * w;
    float* dst = patches + pn * 3 * PP;
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[c * PP + threadIdx.x] = img[((int64_t)c * h + row + i) * w + col + j];
    if (threadIdx.x < 2) {
        const float half = (float)(P / 2);                       // data/patch_sampling.py:565-568
        const float smp = (float)(threadIdx.x == 0 ? row : col) + half;
        const float den = (float)(threadIdx.x == 0 ? h : w) - half;
        pos[pn * 2 + threadIdx.x] = fminf(fmaxf(smp / den, 0.0f), (float)(1.0 - 1e-6));
    }
    if (write_scales && threadIdx.x == 2) scales[pn] = (float)sc;
}

}  // namespace

hipError_t launch_image_normalize(const uint8_t* in, float* out, int NI, int H, int W, const int* flips, const float* mean, const float* sd,
                                  hipStream_t s) {
    const int64_t plane = (int64_t)H * W;
    const int gx = (int)((plane + 255) / 256 > 2048 ? 2048 : (plane + 255) / 256);
    hipLaunchKernelGGL(image_normalize_kernel, dim3(gx, 3, NI), dim3(256), 0, s, in, out, H, W, flips, mean[0], mean[1], mean[2], sd[0],
                       sd[1], sd[2]);
    return hipGetLastError();
}

hipError_t launch_avgpool2(const float* in, float* out, int NC, int H, int W, hipStream_t s) {
    const int n = (H / 2) * (W / 2);
    if (n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(avgpool2_kernel, dim3((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256, NC), dim3(256), 0, s, in, out, H, W);
    return hipGetLastError();
}

hipError_t launch_gather_patches(const float* const* levels, const int* hs, const int* ws, int nlevels, const int* samples,
                                 const int* scale_ids, float* patches, float* pos, float* scales, int NI, int N, hipStream_t s, int P) {
    if (nlevels < 1 || nlevels > 4 || (P != 16 && P != 8)) return hipErrorInvalidValue;
    Levels lv{};
    for (int i = 0; i < 4; ++i) {
        const int k = i < nlevels ? i : nlevels - 1;
        lv.p[i] = levels[k]; lv.h[i] = hs[k]; lv.w[i] = ws[k];
    }
    hipLaunchKernelGGL(gather_patches_kernel, dim3(N, NI), dim3(P * P), 0, s, lv, samples, scale_ids, patches, pos, scales, N, scales != nullptr, P);
    return hipGetLastError();
}

}  // namespace vtq
