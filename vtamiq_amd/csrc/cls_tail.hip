// Last encoder layer, CLS row only.
//
// VTAMIQ consumes token 0 of the encoder output and nothing else (modules/vtamiq/vtamiq.py:104-108), so in the LAST
// EncoderLayer (modules/VisionTransformer/transformer.py:275-285) only the K/V projections need every row; the query,
// the attention output, out-proj, LayerNorm_2 and the MLP are needed for the 2B CLS rows alone.  This file holds the row
// LayerNorm and the single-query attention of that tail; its four linear stages run on the MFMA pipe (skinny.hip).  Results are
// identical in exact arithmetic to running the full layer and reading row 0 (SURVEY.md 8d allows the pruning; bench reports
// executed flops).
#include <mutex>

#include "dev_common.h"
#include "kernels.h"

namespace vtq {
namespace {

// LayerNorm(eps 1e-6) of gathered fp32 rows: src row r at src + r*stride; writes ln[r][H] and optionally a copy of the row.
template <int V4>
__global__ __launch_bounds__(256) void rows_ln_kernel(const float* __restrict__ src, int64_t stride, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ ln, float* __restrict__ copy,
                                                      int rows, PlaneOut po) {
    constexpr int H = 256 * V4;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float4* xr = (const float4*)(src + r * stride);
    float4 v[V4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        v[i] = xr[i * 64 + lane];
        if (copy) ((float4*)(copy + (int64_t)r * H))[i * 64 + lane] = v[i];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / H) + 1e-6f);
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        const float4 w4 = ((const float4*)w)[i * 64 + lane], b4 = ((const float4*)b)[i * 64 + lane];
        float4 y = {v[i].x * rstd * w4.x + b4.x, v[i].y * rstd * w4.y + b4.y, v[i].z * rstd * w4.z + b4.z, v[i].w * rstd * w4.w + b4.w};
        ((float4*)(ln + (int64_t)r * H))[i * 64 + lane] = y;
        plane_store4(po, r, (i * 64 + lane) * 4, y.x, y.y, y.z, y.w);
    }
}

// attention of the single CLS query of each (sequence, head) over the S keys of the sequence; K, V from the packed 16-bit planes.
// One 4-wave workgroup per (sequence, head): thread t scores keys t, t+256, ... (one 128-byte K row per plane per key); for the
// PV sum thread t owns 8 output dims (t & 7) of key group t >> 3 and walks keys kg, kg+32, ... with 16-byte V loads, four keys
// in flight; partial maxima / sums / outputs meet in LDS.
template <typename T, int NPL>
__global__ __launch_bounds__(256) void cls_attention_kernel(const float* __restrict__ q, const T* __restrict__ qkv, int64_t plane,
                                                            float* __restrict__ out, int S, int S_pad, int H, PlaneOut po, int q_log2) {
    typedef typename Vec<T>::x8 tx8;
    extern __shared__ __attribute__((aligned(16))) float cls_smem[];     // [8] red | [32][64] part | [S] scores
    float* red = cls_smem;
    float (*part)[64] = (float (*)[64])(cls_smem + 8);
    float* ps = cls_smem + 8 + 32 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.x, seq = blockIdx.y;
    const int ld = 3 * H;
    const T* kb = qkv + ((int64_t)seq * S_pad) * ld + H + head * 64;
    const T* vb = kb + H;
    // scores: 8 threads per key, each one 16-byte chunk of the key row, so a wave instruction reads 8 whole 128-byte rows (one
    // thread per key made every load touch 64 different lines and thrashed the L1: 127 us at 64 x 501 x 12 heads)
    const int kg = tid >> 3, dc = (tid & 7) * 8;
    float qv[8];
    {
        const float4 t0 = *(const float4*)(q + (int64_t)seq * H + head * 64 + dc), t1 = *(const float4*)(q + (int64_t)seq * H + head * 64 + dc + 4);
        qv[0] = t0.x; qv[1] = t0.y; qv[2] = t0.z; qv[3] = t0.w; qv[4] = t1.x; qv[5] = t1.y; qv[6] = t1.z; qv[7] = t1.w;
    }
    float mx = -INFINITY;
    auto score_key = [&](int key) {
        const T* kr = kb + (int64_t)key * ld + dc;
        const tx8 kh = *(const tx8*)kr;
        tx8 kl;
        if constexpr (NPL == 2) kl = *(const tx8*)(kr + plane);
        float sc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float kv = (float)kh[e];
            if constexpr (NPL == 2) kv += (float)kl[e];
            sc += qv[e] * kv;
        }
        return sc;
    };
    auto finish = [&](int key, float sc) {
        sc += __shfl_xor(sc, 1, 64);
        sc += __shfl_xor(sc, 2, 64);
        sc += __shfl_xor(sc, 4, 64);
        if (!q_log2) sc *= 0.125f;                     // q_log2: the scale (and log2 e) came with q
        if ((tid & 7) == 0) ps[key] = sc;
        mx = fmaxf(mx, sc);
    };
    {
        int key = kg;
        for (; key + 96 < S; key += 128) {
            const float s0 = score_key(key), s1 = score_key(key + 32), s2 = score_key(key + 64), s3 = score_key(key + 96);
            finish(key, s0); finish(key + 32, s1); finish(key + 64, s2); finish(key + 96, s3);
        }
        for (; key < S; key += 32) finish(key, score_key(key));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int key = tid; key < S; key += 256) {
        const float p = q_log2 ? exp2f(ps[key] - mx) : expf(ps[key] - mx);
        ps[key] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    // out[d] = sum_key p[key] * V[key][d]
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto acc_key = [&](int key) {
        const T* vr = vb + (int64_t)key * ld + dc;
        const tx8 vh = *(const tx8*)vr;
        tx8 vl;
        if constexpr (NPL == 2) vl = *(const tx8*)(vr + plane);
        const float pk = ps[key];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float vv = (float)vh[e];
            if constexpr (NPL == 2) vv += (float)vl[e];
            o[e] += pk * vv;
        }
    };
    int key = kg;
    for (; key + 96 < S; key += 128) { acc_key(key); acc_key(key + 32); acc_key(key + 64); acc_key(key + 96); }
    for (; key < S; key += 32) acc_key(key);
#pragma unroll
    for (int e = 0; e < 8; ++e) part[kg][dc + e] = o[e];
    __syncthreads();
    if (tid < 16) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            const float4 pv = *(const float4*)&part[g][4 * tid];
            t[0] += pv.x; t[1] += pv.y; t[2] += pv.z; t[3] += pv.w;
        }
        const float inv = 1.0f / sum;
        const int col = head * 64 + 4 * tid;
        if (out) *(float4*)(out + (int64_t)seq * H + col) = float4{t[0] * inv, t[1] * inv, t[2] * inv, t[3] * inv};
        plane_store4(po, seq, col, t[0] * inv, t[1] * inv, t[2] * inv, t[3] * inv);
    }
}

}  // namespace

hipError_t launch_rows_ln(const float* src, int64_t stride, const float* w, const float* b, float* ln, float* copy, int rows, int H,
                          PlaneOut po, hipStream_t s) {
    const dim3 g((rows + 3) / 4), blk(256);
    if (H == 768) hipLaunchKernelGGL(rows_ln_kernel<3>, g, blk, 0, s, src, stride, w, b, ln, copy, rows, po);
    else if (H == 1024) hipLaunchKernelGGL(rows_ln_kernel<4>, g, blk, 0, s, src, stride, w, b, ln, copy, rows, po);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

namespace {
constexpr int kClsLdsMax = 160 * 1024 - 4096;                 // leave room for the runtime's own LDS use
constexpr int kClsFixed = (8 + 32 * 64) * 4;
template <typename T, int NPL>
hipError_t launch_cls_attention_t(const float* q, const void* qkv, int64_t plane, float* out, int nseq, int S, int S_pad, int H,
                                  PlaneOut po, hipStream_t s, bool q_log2) {
    const int lds = kClsFixed + ((S + 3) & ~3) * 4;
    if (lds > 48 * 1024) {                                    // beyond the default limit: raise it once per device
        static std::mutex mu;
        static bool configured[64] = {false};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!configured[dev]) {
            e = hipFuncSetAttribute((const void*)cls_attention_kernel<T, NPL>, hipFuncAttributeMaxDynamicSharedMemorySize, kClsLdsMax);
            if (e != hipSuccess) return e;
            configured[dev] = true;
        }
    }
    hipLaunchKernelGGL((cls_attention_kernel<T, NPL>), dim3(H / 64, nseq), dim3(256), lds, s, q, (const T*)qkv, plane, out, S, S_pad, H, po, q_log2 ? 1 : 0);
    return hipGetLastError();
}
}  // namespace

int cls_attention_max_seq() { return (kClsLdsMax - kClsFixed) / 4; }

hipError_t launch_cls_attention(const float* q, const void* qkv, int64_t plane, float* out, int nseq, int S, int S_pad, int H,
                                int f16_, int planes, PlaneOut po, hipStream_t s, bool q_log2) {
    if (S < 1 || S > cls_attention_max_seq() || (planes != 1 && planes != 2)) return hipErrorInvalidValue;
    if (!f16_) return planes == 1 ? launch_cls_attention_t<bf16, 1>(q, qkv, plane, out, nseq, S, S_pad, H, po, s, q_log2)
                                  : launch_cls_attention_t<bf16, 2>(q, qkv, plane, out, nseq, S, S_pad, H, po, s, q_log2);
    return planes == 1 ? launch_cls_attention_t<f16, 1>(q, qkv, plane, out, nseq, S, S_pad, H, po, s, q_log2)
                       : launch_cls_attention_t<f16, 2>(q, qkv, plane, out, nseq, S, S_pad, H, po, s, q_log2);
}

}  // namespace vtq
