// HBM-bound kernels of the ViT path: fp32 -> 16-bit (hi[, lo]) plane packing, LayerNorm, embedding index/token rows.
// T = bf16 | f16 element type of the planes, NPL = 1 (single plane) | 2 (hi + lo).
// One wave per row where a row reduction is needed (wave64 shuffles, no LDS); 16-byte vector accesses everywhere.
#include "dev_common.h"
#include "kernels.h"

namespace vtq {
namespace {

template <typename T, int NPL>
__device__ __forceinline__ void store4(T* dst, int64_t plane, float a, float b, float c, float d, float scale = 1.0f) {
    typedef typename Vec<T>::x4 tx4;
    if constexpr (std::is_same<T, f8>::value) {          // e4m3 bytes of value * scale (static per-tensor scale, fp8 mode)
        *(uint32_t*)dst = pack_fp8x4(a * scale, b * scale, c * scale, d * scale);
    } else if constexpr (NPL == 1) {
        tx4 h = {(T)a, (T)b, (T)c, (T)d};
        *(tx4*)dst = h;
    } else {
        tx4 h, l;
        T x, y;
        split2<T>(a, x, y); h[0] = x; l[0] = y;
        split2<T>(b, x, y); h[1] = x; l[1] = y;
        split2<T>(c, x, y); h[2] = x; l[2] = y;
        split2<T>(d, x, y); h[3] = x; l[3] = y;
        *(tx4*)dst = h;
        *(tx4*)(dst + plane) = l;
    }
}

template <typename T, int NPL>
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t plane,
                                                    int64_t n4, float scale, Fp8Obs obs) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = ((const float4*)src)[i];
        if constexpr (std::is_same<T, f8>::value) m = amax4(m, v.x, v.y, v.z, v.w);
        store4<T, NPL>(dst + i * 4, plane, v.x, v.y, v.z, v.w, scale);
    }
    if constexpr (std::is_same<T, f8>::value) fp8_report(obs, m, scale);
}

// rows of K floats -> rows of Kp >= K elements, zero beyond K (a weight whose K is padded to the GEMM's K-tile: ViT-B/8's 192-wide
// patch embedding)
template <typename T, int NPL>
__global__ __launch_bounds__(256) void split_rows_pad_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t plane, int rows,
                                                             int K4, int Kp4, float scale) {
    const int64_t total = (int64_t)rows * Kp4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / Kp4;
        const int c4 = (int)(i - r * Kp4);
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (c4 < K4) v = ((const float4*)src)[r * K4 + c4];
        store4<T, NPL>(dst + i * 4, plane, v.x, v.y, v.z, v.w, scale);
    }
}

// fp8 weights: one wave per row of W[N][K]: s = the largest power of two with max|row| * s <= 448 (1 for an all-zero row),
// dst = e4m3(W * s), inv_scale[n] = 1 / s   (per-output-channel scale; oracle/fp8_oracle.py quant_rows)
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst,
                                                             float* __restrict__ inv_scale, int N, int K, int Kp) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float4* row = (const float4*)(src + (int64_t)n * K);
    float mx = 0.f;
    for (int i = lane; i < K / 4; i += 64) {
        const float4 v = row[i];
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float s = 1.0f;
    if (mx > 0.f) s = exp2f(floorf(log2f(448.0f / mx)));
    if (mx * s > 448.0f) s *= 0.5f;                       // log2f rounding next to an exact power of two, either side
    else if (mx > 0.f && mx * s * 2.0f <= 448.0f) s *= 2.0f;
    for (int i = lane; i < Kp / 4; i += 64) {            // destination rows are Kp >= K bytes, zero beyond K
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (i < K / 4) v = row[i];
        ((uint32_t*)(dst + (int64_t)n * Kp))[i] = pack_fp8x4(v.x * s, v.y * s, v.z * s, v.w * s);
    }
    if (lane == 0) inv_scale[n] = 1.0f / s;
}

// rows [k*BN, (k+1)*BN) from image k (k < nimg: ref, dist[, dist2]); rows >= nimg*BN zero.  K = 768 floats per row.
struct ImgPtrs { const float* p[3]; };

template <typename T, int NPL>
__global__ __launch_bounds__(256) void pack_patches_kernel(ImgPtrs src, int nimg, T* __restrict__ dst, int64_t plane, int BN, int K4,
                                                           int Kp4, int64_t total4, float scale, Fp8Obs obs) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / Kp4;                    // destination rows are Kp4 >= K4 quads wide, zero beyond K4
        const int c4 = (int)(i - row * Kp4);
        const int img = (int)(row / BN);
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (img < nimg && c4 < K4) {
            const float* sp = img == 0 ? src.p[0] : (img == 1 ? src.p[1] : src.p[2]);
            v = ((const float4*)sp)[(row - (int64_t)img * BN) * K4 + c4];
        }
        if constexpr (std::is_same<T, f8>::value) m = amax4(m, v.x, v.y, v.z, v.w);
        store4<T, NPL>(dst + i * 4, plane, v.x, v.y, v.z, v.w, scale);
    }
    if constexpr (std::is_same<T, f8>::value) fp8_report(obs, m, scale);
}

// UvPosEmbedding.forward index (transformer.py:417-421): floor(pos*G) -> i0*G + i1 + 1, evaluated in fp32 like torch;
// ScaleEmbedding.forward index (transformer.py:396-398): clamp(scale, 0, num_scales-1) + 1.
// The reference indexes its table with whatever comes out (transformer.py:417-421) and raises IndexError / a device assert for
// pos outside [0, 1); here an out-of-range index (pos < 0, pos >= 1, NaN) is clamped into the table and reported through
// *err (bit 0), which vtq_input_errors() reads back: never an out-of-bounds gather.
__global__ void embed_index_kernel(ImgPtrs pos, ImgPtrs sc, int nimg, int* __restrict__ pidx, int* __restrict__ sidx,
                                   int* __restrict__ row_map, int B, int N, int rows_pad, SeqMap sm, int T, int grid, int num_scales,
                                   int* __restrict__ err) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows_pad) return;
    const int BN = B * N;
    const int img = r / BN;
    if (img >= nimg) { pidx[r] = 0; sidx[r] = 0; row_map[r] = -1; return; }
    const int rr = r - img * BN;
    const float* pb = img == 0 ? pos.p[0] : (img == 1 ? pos.p[1] : pos.p[2]);
    const float* pp = pb + (int64_t)rr * 2;
    const float g = (float)grid;
    const float f0 = floorf(pp[0] * g), f1 = floorf(pp[1] * g);
    const bool ok = f0 >= 0.0f && f0 < g && f1 >= 0.0f && f1 < g;            // false for NaN too
    if (!ok) atomicOr(err, 1);
    const float c0 = fminf(fmaxf(f0, 0.0f), g - 1.0f), c1 = fminf(fmaxf(f1, 0.0f), g - 1.0f);   // fmaxf(NaN, 0) = 0
    pidx[r] = (int)(c0 * g + c1 + 1.0f);
    int si = 0;
    if (sc.p[0]) {
        const float* sb = img == 0 ? sc.p[0] : (img == 1 ? sc.p[1] : sc.p[2]);
        float sv = sb[rr];
        sv = fminf(fmaxf(sv, 0.0f), (float)(num_scales - 1)) + 1.0f;
        si = (int)sv;
    }
    sidx[r] = si;
    const int b = rr / N, n = rr - b * N;
    row_map[r] = (int)seq_row(sm, img * B + b) + T + n;
}

// Embeddings.forward on PRE-EMBEDDED input (transformer.py:534-535: a (B, N, H) tensor skips the patch convolution): token row = feature row + positional
// row + scale row, in the reference's order of additions (:540-552); the destinations and table indices are embed_index_kernel's.
__global__ __launch_bounds__(256) void embed_rows_kernel(ImgPtrs feats, int nimg, int BN, const int* __restrict__ row_map, const int* __restrict__ pidx,
                                                         const int* __restrict__ sidx, const float* __restrict__ table1,
                                                         const float* __restrict__ table2, float* __restrict__ x, int H4) {
    const int r = blockIdx.x;
    const int img = r / BN, rr = r - img * BN;
    const float* fb = img == 0 ? feats.p[0] : (img == 1 ? feats.p[1] : feats.p[2]);
    const float4* src = (const float4*)(fb + (int64_t)rr * H4 * 4);
    const int orow = row_map[r];
    if (orow < 0) return;
    const float4* t1 = (const float4*)table1 + (int64_t)pidx[r] * H4;
    const float4* t2 = table2 ? (const float4*)table2 + (int64_t)sidx[r] * H4 : nullptr;
    float4* dst = (float4*)x + (int64_t)orow * H4;
    for (int c = threadIdx.x; c < H4; c += blockDim.x) {
        float4 v = src[c];
        const float4 a = t1[c];
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        if (t2) { const float4 b = t2[c]; v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        dst[c] = v;
    }
}

// Embeddings.forward_tokens (transformer.py:507-524): row 0 = cls + pos_table[0]; rows 1..T-1 = register tokens.
__global__ void tokens_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos_table,
                              const float* __restrict__ extra, SeqMap sm, int T, int H) {
    const int seq = blockIdx.x, t = blockIdx.y;
    float* dst = x + (seq_row(sm, seq) + t) * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) dst[c] = (t == 0) ? cls[c] + pos_table[c] : extra[(t - 1) * H + c];
}

__global__ void zero_pad_rows_kernel(float* __restrict__ x, int nseq, int S, SeqMap sm, int H4, int rows_total) {
    const int per_seq = sm.pitch - S;                                   // pad rows behind every sequence
    const int nparts = (nseq + sm.per - 1) / sm.per;
    const int64_t part_rows = (int64_t)sm.per * sm.pitch + sm.gap;
    const int n_seq = nseq * per_seq, n_gap = nparts * sm.gap;
    const int npad = n_seq + n_gap + (int)(rows_total - nparts * part_rows);
    for (int i = blockIdx.x; i < npad; i += gridDim.x) {
        int64_t row;
        if (i < n_seq) { const int sq = i / per_seq; row = seq_row(sm, sq) + S + (i - sq * per_seq); }
        else if (i < n_seq + n_gap) { const int j = i - n_seq, pt = j / sm.gap; row = pt * part_rows + (int64_t)sm.per * sm.pitch + (j - pt * sm.gap); }
        else row = nparts * part_rows + (i - n_seq - n_gap);
        float4* d = (float4*)x + row * H4;
        for (int c = threadIdx.x; c < H4; c += blockDim.x) d[c] = float4{0.f, 0.f, 0.f, 0.f};
    }
}

__global__ void copy_tokens_kernel(const float* __restrict__ x, float* __restrict__ dst, SeqMap sm, int T, int H) {
    const int seq = blockIdx.x, t = blockIdx.y;
    const float* src = x + (seq_row(sm, seq) + t) * H;
    float* d = dst + ((int64_t)seq * T + t) * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) d[c] = src[c];
}

// torch.nn.LayerNorm(H, eps=1e-6) (transformer.py:253-254): two-pass fp32 statistics held in registers.
// One wave per row, V4 float4 per lane (H = 256*V4).
template <int V4>
__device__ __forceinline__ void ln_row(const float* __restrict__ xr, const float* __restrict__ w, const float* __restrict__ b,
                                       int lane, float4 (&y)[V4]) {
    constexpr int H = 256 * V4;
    float4 v[V4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        v[i] = ((const float4*)xr)[i * 64 + lane];
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
        const float4 w4 = ((const float4*)w)[i * 64 + lane];
        const float4 b4 = ((const float4*)b)[i * 64 + lane];
        y[i].x = v[i].x * rstd * w4.x + b4.x;
        y[i].y = v[i].y * rstd * w4.y + b4.y;
        y[i].z = v[i].z * rstd * w4.z + b4.z;
        y[i].w = v[i].w * rstd * w4.w + b4.w;
    }
}

template <int V4, typename T, int NPL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, T* __restrict__ out, int64_t o_plane,
                                                        int rows, float scale, Fp8Obs obs) {
    constexpr int H = 256 * V4;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4 y[V4];
    ln_row<V4>(x + (int64_t)row * H, w, b, lane, y);
    T* o = out + (int64_t)row * H;
#pragma unroll
    for (int i = 0; i < V4; ++i) store4<T, NPL>(o + (i * 64 + lane) * 4, o_plane, y[i].x, y[i].y, y[i].z, y[i].w, scale);
    if constexpr (std::is_same<T, f8>::value) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < V4; ++i) m = amax4(m, y[i].x, y[i].y, y[i].z, y[i].w);
        fp8_report(obs, m, scale);
    }
}

// encoder_norm on the two CLS rows of pair b only (transformer.py:376 applies it to all rows; only token 0 is
// consumed, vtamiq.py:107-108), then diff and diff_scale (vtamiq.py:111).
template <int V4>
__global__ __launch_bounds__(64) void final_diff_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, const float* __restrict__ gamma,
                                                        float* __restrict__ d, int B, SeqMap sm, PlaneOut po, int* __restrict__ err) {
    constexpr int H = 256 * V4;
    const int lane = threadIdx.x;
    const int pb = blockIdx.x, j = blockIdx.y;           // j-th distorted image (0 for FR pairs; 0,1 for pairwise triplets)
    float4 yr[V4], yd[V4];
    ln_row<V4>(x + seq_row(sm, pb) * H, w, b, lane, yr);
    ln_row<V4>(x + seq_row(sm, (j + 1) * B + pb) * H, w, b, lane, yd);
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        float4 r = {yr[i].x - yd[i].x, yr[i].y - yd[i].y, yr[i].z - yd[i].z, yr[i].w - yd[i].w};
        if (gamma) {
            const float4 g4 = ((const float4*)gamma)[i * 64 + lane];
            r.x *= g4.x; r.y *= g4.y; r.z *= g4.z; r.w *= g4.w;
        }
        ((float4*)(d + (int64_t)(j * B + pb) * H))[i * 64 + lane] = r;
        plane_store4(po, (int64_t)(j * B + pb), (i * 64 + lane) * 4, r.x, r.y, r.z, r.w);
        // a non-finite CLS difference: some operand upstream left its format's range (fp16 planes: |v| > 65504) -- any inf / NaN in a
        // sequence reaches its CLS row through the softmax.  Reported through vtq_input_errors bit 1.
        if (err && !(isfinite(r.x) && isfinite(r.y) && isfinite(r.z) && isfinite(r.w))) atomicOr(err, 2);
    }
}

inline int grid_for(int64_t work, int block) {
    int64_t g = (work + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

}  // namespace

// dispatch on (f16, planes) -> K<T, NPL>; e4m3 outputs (f16 == 2) exist in builds of the fp8 experiment only (-DVTQ_WITH_FP8)
#ifdef VTQ_WITH_FP8
#define VTQ_FMT_F8(f16_, npl_, CALL) else if ((f16_) == 2 && (npl_) == 1) { CALL(f8, 1); }
#else
#define VTQ_FMT_F8(f16_, npl_, CALL) else if ((f16_) == 2) return hipErrorNotSupported;
#endif
#define VTQ_FMT_DISPATCH(f16_, npl_, CALL)                      \
    do {                                                        \
        if (!(f16_) && (npl_) == 1) { CALL(bf16, 1); }          \
        else if (!(f16_) && (npl_) == 2) { CALL(bf16, 2); }     \
        else if ((f16_) == 1 && (npl_) == 1) { CALL(f16, 1); }  \
        else if ((f16_) == 1 && (npl_) == 2) { CALL(f16, 2); }  \
        VTQ_FMT_F8(f16_, npl_, CALL)                            \
        else return hipErrorInvalidValue;                       \
    } while (0)

hipError_t launch_quant_rows_fp8(const float* src, void* dst, float* inv_scale, int N, int K, hipStream_t s, int Kp) {
#ifndef VTQ_WITH_FP8
    return hipErrorNotSupported;
#endif
    if (Kp == 0) Kp = K;
    if (K % 4 || Kp % 4 || Kp < K) return hipErrorInvalidValue;
    hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((N + 3) / 4), dim3(256), 0, s, src, (uint8_t*)dst, inv_scale, N, K, Kp);
    return hipGetLastError();
}

hipError_t launch_split_rows_pad(const float* src, void* dst, int64_t plane, int rows, int K, int Kp, int f16_, int planes, hipStream_t s) {
    if (K % 4 || Kp % 4 || Kp < K) return hipErrorInvalidValue;
    const int K4 = K / 4, Kp4 = Kp / 4;
    const int64_t total = (int64_t)rows * Kp4;
#define VTQ_CALL(TT, NP) \
    hipLaunchKernelGGL((split_rows_pad_kernel<TT, NP>), dim3(grid_for(total, 256)), dim3(256), 0, s, src, (TT*)dst, plane, rows, K4, Kp4, 1.0f)
    VTQ_FMT_DISPATCH(f16_, planes, VTQ_CALL);
#undef VTQ_CALL
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void scale_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n, float mul) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i] * mul;
}

hipError_t launch_scale_copy(const float* src, float* dst, int64_t n, float mul, hipStream_t s) {
    hipLaunchKernelGGL(scale_copy_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, src, dst, n, mul);
    return hipGetLastError();
}

hipError_t launch_split(const float* src, void* dst, int64_t plane, int64_t numel, int f16_, int planes, hipStream_t s, float scale, Fp8Obs obs) {
    if (numel % 4) return hipErrorInvalidValue;
    const int64_t n4 = numel / 4;
#define VTQ_CALL(TT, NP) hipLaunchKernelGGL((split_kernel<TT, NP>), dim3(grid_for(n4, 256)), dim3(256), 0, s, src, (TT*)dst, plane, n4, scale, obs)
    VTQ_FMT_DISPATCH(f16_, planes, VTQ_CALL);
#undef VTQ_CALL
    return hipGetLastError();
}

hipError_t launch_pack_patches(const float* const* imgs, int nimg, void* dst, int64_t plane, int BN, int K, int rows_pad, int f16_,
                               int planes, hipStream_t s, float scale, int Kp, Fp8Obs obs) {
    if (Kp == 0) Kp = K;
    if (K % 4 || Kp % 4 || Kp < K) return hipErrorInvalidValue;
    const int K4 = K / 4, Kp4 = Kp / 4;
    const int64_t total4 = (int64_t)rows_pad * Kp4;
    ImgPtrs ip{{imgs[0], imgs[1], nimg > 2 ? imgs[2] : nullptr}};
#define VTQ_CALL(TT, NP) \
    hipLaunchKernelGGL((pack_patches_kernel<TT, NP>), dim3(grid_for(total4, 256)), dim3(256), 0, s, ip, nimg, (TT*)dst, plane, BN, K4, Kp4, total4, scale, obs)
    VTQ_FMT_DISPATCH(f16_, planes, VTQ_CALL);
#undef VTQ_CALL
    return hipGetLastError();
}

hipError_t launch_embed_index(const float* const* pos, const float* const* sc, int nimg, int* pidx, int* sidx, int* row_map, int B, int N,
                              int rows_pad, SeqMap sm, int T, int grid, int num_scales, int* err, hipStream_t s) {
    ImgPtrs pp{{pos[0], pos[1], nimg > 2 ? pos[2] : nullptr}};
    ImgPtrs sp{{sc ? sc[0] : nullptr, sc ? sc[1] : nullptr, (sc && nimg > 2) ? sc[2] : nullptr}};
    hipLaunchKernelGGL(embed_index_kernel, dim3((rows_pad + 255) / 256), dim3(256), 0, s, pp, sp, nimg, pidx, sidx, row_map, B, N, rows_pad,
                       sm, T, grid, num_scales, err);
    return hipGetLastError();
}

hipError_t launch_embed_rows(const float* const* feats, int nimg, int BN, const int* row_map, const int* pidx, const int* sidx,
                             const float* table1, const float* table2, float* x, int H, hipStream_t s) {
    ImgPtrs fp{{feats[0], feats[1], nimg > 2 ? feats[2] : nullptr}};
    hipLaunchKernelGGL(embed_rows_kernel, dim3(nimg * BN), dim3(256), 0, s, fp, nimg, BN, row_map, pidx, sidx, table1, table2, x, H / 4);
    return hipGetLastError();
}

hipError_t launch_tokens(float* x, const float* cls, const float* pos_table, const float* extra, int nseq, SeqMap sm, int T,
                         int H, hipStream_t s) {
    hipLaunchKernelGGL(tokens_kernel, dim3(nseq, T), dim3(256), 0, s, x, cls, pos_table, extra, sm, T, H);
    return hipGetLastError();
}

hipError_t launch_zero_pad_rows(float* x, int nseq, int S, SeqMap sm, int H, int rows_total, hipStream_t s) {
    const int nparts = (nseq + sm.per - 1) / sm.per;
    const int64_t npad = (int64_t)nseq * (sm.pitch - S) + (int64_t)nparts * sm.gap + (rows_total - nparts * ((int64_t)sm.per * sm.pitch + sm.gap));
    if (npad <= 0) return hipSuccess;
    hipLaunchKernelGGL(zero_pad_rows_kernel, dim3(npad < 4096 ? (int)npad : 4096), dim3(256), 0, s, x, nseq, S, sm, H / 4, rows_total);
    return hipGetLastError();
}

hipError_t launch_copy_tokens(const float* x, float* dst, int nseq, SeqMap sm, int T, int H, hipStream_t s) {
    hipLaunchKernelGGL(copy_tokens_kernel, dim3(nseq, T), dim3(256), 0, s, x, dst, sm, T, H);
    return hipGetLastError();
}

hipError_t launch_layernorm(const float* x, const float* w, const float* b, void* out, int64_t o_plane, int rows, int H,
                            int f16_, int planes, hipStream_t s, float scale, Fp8Obs obs) {
    const dim3 g((rows + 3) / 4), blk(256);
    if (H == 768) {
#define VTQ_CALL(TT, NP) hipLaunchKernelGGL((layernorm_kernel<3, TT, NP>), g, blk, 0, s, x, w, b, (TT*)out, o_plane, rows, scale, obs)
        VTQ_FMT_DISPATCH(f16_, planes, VTQ_CALL);
#undef VTQ_CALL
    } else if (H == 1024) {
#define VTQ_CALL(TT, NP) hipLaunchKernelGGL((layernorm_kernel<4, TT, NP>), g, blk, 0, s, x, w, b, (TT*)out, o_plane, rows, scale, obs)
        VTQ_FMT_DISPATCH(f16_, planes, VTQ_CALL);
#undef VTQ_CALL
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_final_diff(const float* x, const float* ln_w, const float* ln_b, const float* gamma, float* d, int B, int ndist,
                             SeqMap sm, int H, PlaneOut po, hipStream_t s, int* err) {
    if (H == 768) hipLaunchKernelGGL(final_diff_kernel<3>, dim3(B, ndist), dim3(64), 0, s, x, ln_w, ln_b, gamma, d, B, sm, po, err);
    else if (H == 1024) hipLaunchKernelGGL(final_diff_kernel<4>, dim3(B, ndist), dim3(64), 0, s, x, ln_w, ln_b, gamma, d, B, sm, po, err);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace vtq
