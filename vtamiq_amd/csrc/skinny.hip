// Skinny linear layers on the MFMA pipe: out[R, N] = act[R, K] * W[N, K]^T for a FEW rows (R = the batch's CLS rows).
//
// Two users, both chains of small dependent stages:
//   * the CLS-only tail of the last encoder layer (query projection, out-proj, fc1, fc2 for the 2B CLS rows;
//     modules/VisionTransformer/transformer.py:275-285 restricted to token 0, see cls_tail.hip), and
//   * the DiffNet head + quality predictor on the [B, H] CLS difference (modules/vtamiq/vtamiq.py:12-23, 71-77, 111-117,
//     modules/RCAN/channel_attention.py:13-86: every Conv1d(k=1) on (B, C, 1) is such a product).
// Round 1 ran these as fp32 FMA kernels that re-read the activation rows once per output channel (14 us per 768x768 stage,
// 25-37 us per tail stage); here one workgroup computes 16 output channels for 64 rows on mfma_f32_16x16x32 with the
// operand formats of the big GEMM (gemm.hip: 1, 2 or 3 MFMAs per product on 16-bit hi[/lo] planes, fp32 accumulate):
//   * activations arrive as 16-bit planes (written by the producing stage's epilogue), weights are 16-bit planes [N_pad, K];
//     both are streamed straight from global memory into MFMA fragments (a stage reads each weight once: no LDS staging);
//   * the W fragment is the MFMA A-operand, so a lane ends up with 4 consecutive output channels of one row;
//   * the 8 waves split K (k-step = 32) round-robin, three k-steps' loads in flight per wave (a 768-deep product is one
//     exposed memory latency), and combine their partial sums through LDS; wave w < 4 then finishes row block w (16 rows):
//     bias, activation, residual / gate forms, and writes fp32 and / or the planes the next stage reads.
#include <mutex>

#include "dev_common.h"
#include "kernels.h"

namespace vtq {
namespace {

__device__ __forceinline__ float prelu1(float v, float a) { return v >= 0.f ? v : a * v; }

template <typename T, int TERMS, int RB>      // RB = row blocks of 16 per workgroup: 64, 32 or 16 rows (the activation rows are the
__global__ __launch_bounds__(512) void skinny_linear_kernel(SkinnyArgs p) {   // bulk of a workgroup's traffic: do not load more than R needs)
    typedef typename Vec<T>::x8 tx8;
    typedef typename Vec<T>::x4 tx4;
    constexpr int APL = (TERMS == 1) ? 1 : 2, WPL = (TERMS == 3) ? 2 : 1, NW = 8, UN = (RB == 4) ? 3 : 6;
    __shared__ __attribute__((aligned(16))) float red[NW][RB][64][4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16, r0 = blockIdx.y * (16 * RB);
    const T* wrow = (const T*)p.W + (int64_t)(n0 + fr) * p.K + 8 * fq;
    const T* xrow = (const T*)p.xa + (int64_t)(r0 + fr) * p.ldx + 8 * fq;
    f32x4 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Epilogue operands of the finishing waves (wave w < RB finishes row block w), requested here so that their latency passes under the
    // product's instead of after it: a stage is a chain of exposed latencies and little else.
    const int e_row = r0 + wave * 16 + fr, e_n = n0 + 4 * fq;
    const bool e_live = wave < RB && e_row < p.R && e_n < p.N;
    float e_bias[4] = {0.f, 0.f, 0.f, 0.f}, e_res[4] = {0.f, 0.f, 0.f, 0.f}, e_aux[4] = {0.f, 0.f, 0.f, 0.f}, e_gam[4] = {1.f, 1.f, 1.f, 1.f};
    float post = 0.f, nxt = 1.f;
    if (e_live) {
        post = p.post_slope ? *p.post_slope : 0.f;
        nxt = p.next_slope ? *p.next_slope : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = e_n + e;
            if (c < p.N) {
                e_bias[e] = p.bias[c];
                if (p.epi == SK_RESID || p.epi == SK_GATE) e_res[e] = p.res[(int64_t)e_row * p.ldr + c];
                if (p.epi == SK_GATE) e_aux[e] = p.aux[(int64_t)e_row * p.ldr + c];
                if (p.epi == SK_RESID && p.gamma) e_gam[e] = p.gamma[c];
            }
        }
    }
    const int nks = p.K >> 5;
    for (int base = wave; base < nks; base += NW * UN) {
        tx8 w[UN][WPL], x[UN][RB][APL];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int ks = base + NW * u;
            if (ks < nks) {                                  // wave-uniform
                const int k = ks << 5;
#pragma unroll
                for (int pl = 0; pl < WPL; ++pl) w[u][pl] = *(const tx8*)(wrow + pl * p.w_plane + k);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int pl = 0; pl < APL; ++pl) x[u][rb][pl] = *(const tx8*)(xrow + pl * p.xa_plane + (int64_t)rb * 16 * p.ldx + k);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (base + NW * u < nks) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    acc[rb] = mfma16<T>(w[u][0], x[u][rb][0], acc[rb]);
                    if constexpr (TERMS >= 2) acc[rb] = mfma16<T>(w[u][0], x[u][rb][1], acc[rb]);
                    if constexpr (TERMS == 3) acc[rb] = mfma16<T>(w[u][1], x[u][rb][0], acc[rb]);
                }
            }
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) *(f32x4*)red[wave][rb][lane] = acc[rb];
    __syncthreads();
    if (wave >= RB) return;
    // wave w finishes row block w: row = r0 + 16 w + fr, channels n0 + 4 fq .. + 3
    const int rb = wave;
    f32x4 v = *(const f32x4*)red[0][rb][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        const f32x4 o = *(const f32x4*)red[w][rb][lane];
        v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
    }
    const int row = e_row;
    const int n = e_n;
    if (!e_live) return;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = n + e;
        float val = v[e] + e_bias[e];
        if (c < p.N) {
            switch (p.epi) {
                case SK_GELU: val = gelu_erf(val); break;
                case SK_PRELU: val = prelu1(val, post); break;
                case SK_RESID: val = e_res[e] + e_gam[e] * val; break;
                case SK_GATE: val = e_res[e] + e_aux[e] * (1.0f / (1.0f + __expf(-val))); break;
                case SK_CONVCAT: if (c >= p.nsplit) val = fmaxf(val, 0.f); break;
                default: break;
            }
        }
        o[e] = val;
    }
    // fp32 destination: columns [0, ycols); 16-bit planes: columns [pcol0, N) of act(out), stored at column c - pcol0
    if (p.y) {
        if (n + 3 < p.ycols) *(float4*)(p.y + (int64_t)row * p.ldy + n) = float4{o[0], o[1], o[2], o[3]};
        else
            for (int e = 0; e < 4; ++e)
                if (n + e < p.ycols) p.y[(int64_t)row * p.ldy + n + e] = o[e];
    }
    if (p.ya && n + 3 >= p.pcol0) {
        T* dst = (T*)p.ya + (int64_t)row * p.ldya + (n - p.pcol0);
        float a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = (n + e < p.N) ? prelu1(o[e], nxt) : 0.f;      // columns >= N: the zero K-padding of the consumer
        if (n >= p.pcol0) {               // (pcol0 is a multiple of 4 in every use, so a lane's 4 columns are on one side)
            if (p.ya_planes == 1) {
                *(tx4*)dst = tx4{(T)a[0], (T)a[1], (T)a[2], (T)a[3]};
            } else {
                tx4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) { T x, y; split2<T>(a[e], x, y); h[e] = x; l[e] = y; }
                *(tx4*)dst = h;
                *(tx4*)(dst + p.ya_plane) = l;
            }
        }
    }
}

// fp32 rows -> 16-bit planes with an optional PReLU (slope pointer) first: the entry of a chain (CLS difference -> first RCAB)
template <typename T, int NPL>
__global__ __launch_bounds__(256) void rows_to_planes_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ slope, T* __restrict__ out,
                                                             int64_t plane, int ldo, int R, int K4) {
    typedef typename Vec<T>::x4 tx4;
    const float a = slope ? *slope : 1.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R * K4; i += gridDim.x * blockDim.x) {
        const int r = i / K4, c = (i - r * K4) * 4;
        const float4 v = *(const float4*)(x + (int64_t)r * ldx + c);
        const float f[4] = {prelu1(v.x, a), prelu1(v.y, a), prelu1(v.z, a), prelu1(v.w, a)};
        T* dst = out + (int64_t)r * ldo + c;
        if constexpr (NPL == 1) {
            *(tx4*)dst = tx4{(T)f[0], (T)f[1], (T)f[2], (T)f[3]};
        } else {
            tx4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { T hh, ll; split2<T>(f[e], hh, ll); h[e] = hh; l[e] = ll; }
            *(tx4*)dst = h;
            *(tx4*)(dst + plane) = l;
        }
    }
}

}  // namespace

hipError_t launch_skinny(const SkinnyArgs& a, Num num, hipStream_t s) {
    if (a.R < 1 || a.N < 1 || a.K < 32 || a.K % 32 || a.ldx % 8 || (a.ya && (a.pcol0 % 4 || a.ldya % 4)) || !num_valid(num)) return hipErrorInvalidValue;
    // Rows per workgroup: every workgroup pulls its rows' activations through its CU's L1 (64 B / clk), which is what a stage's time is made
    // of beyond the exposed latency, so split the rows over workgroups (16 each) while the grid still fits the 256 CUs in one round, and
    // take 32 / 64 rows per workgroup for the wide stages (the tail's fc1).  The result does not depend on the choice: a row's sums are
    // formed in the same order.
    const int nb = (a.N + 15) / 16;
    int rb = 4;
    for (int t = 1; t <= 4; t *= 2)
        if (a.R <= 16 * t || nb * ((a.R + 16 * t - 1) / (16 * t)) <= 256) { rb = t; break; }
    const dim3 g(nb, (a.R + 16 * rb - 1) / (16 * rb)), blk(512);
#define VTQ_SK(TT, TM)                                                                              \
    do {                                                                                            \
        if (rb == 1) hipLaunchKernelGGL((skinny_linear_kernel<TT, TM, 1>), g, blk, 0, s, a);        \
        else if (rb == 2) hipLaunchKernelGGL((skinny_linear_kernel<TT, TM, 2>), g, blk, 0, s, a);   \
        else hipLaunchKernelGGL((skinny_linear_kernel<TT, TM, 4>), g, blk, 0, s, a);                \
    } while (0)
    if (!num.f16) { if (num.terms == 1) VTQ_SK(bf16, 1); else if (num.terms == 3) VTQ_SK(bf16, 3); else return hipErrorInvalidValue; }
    else { if (num.terms == 1) VTQ_SK(f16, 1); else if (num.terms == 2) VTQ_SK(f16, 2); else VTQ_SK(f16, 3); }
#undef VTQ_SK
    return hipGetLastError();
}

hipError_t launch_rows_to_planes(const float* x, int ldx, const float* slope, void* out, int64_t plane, int ldo, int R, int K, int f16_,
                                 int planes, hipStream_t s) {
    if (K % 4 || ldx % 4 || ldo % 4) return hipErrorInvalidValue;
    const int K4 = K / 4, total = R * K4;
    const dim3 g((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256), blk(256);
#define VTQ_RP(TT, NP) hipLaunchKernelGGL((rows_to_planes_kernel<TT, NP>), g, blk, 0, s, x, ldx, slope, (TT*)out, plane, ldo, R, K4)
    if (!f16_) { if (planes == 1) VTQ_RP(bf16, 1); else VTQ_RP(bf16, 2); }
    else { if (planes == 1) VTQ_RP(f16, 1); else VTQ_RP(f16, 2); }
#undef VTQ_RP
    return hipGetLastError();
}

}  // namespace vtq
