// Fused scaled-dot-product attention for gfx950, head_dim 64, no S x S tensor in HBM.
//
// Replaces MultiHeadSelfAttention.forward lines 158-166 of modules/VisionTransformer/transformer.py
// (QK^T / sqrt(dh) -> softmax -> PV -> merge heads); the (B,h,S,S) probabilities the reference materialises and
// returns (:161) are never formed.
//
// Structure (cdna_hip_programming.md Appendix B "fused attention", section 3 "accumulator tile as next operand"):
//   * workgroup = 4 waves = 128 query rows of one (sequence, head); each wave owns 32 query rows;
//   * K/V tiles of 64 keys stream through LDS by 16-byte LDS-DMA, double buffered, one barrier per tile;
//   * S^T = K Q^T with mfma_f32_32x32x16_bf16 (K fragment = A operand): the query index lands on the lane, so the
//     online-softmax row statistics are lane-local (+ one lane^32 shuffle);
//   * O^T += V^T P^T: the exponentiated accumulator registers 8s..8s+7 are the B-operand fragment of k-step s with no
//     lane movement; V^T fragments come from the row-major V tile through ds_read_b64_tr_b16 (hardware transpose);
//   * K tile chunks are XOR-swizzled ((row>>1)&7) for conflict-free ds_read_b128; V tile 64-byte halves are swapped on
//     rows with bit 1 set so the four rows of a transposed read hit disjoint banks;
//   * NSPLIT == 3: Q,K,V,P are hi/lo 16-bit pairs and each product is hi*hi + hi*lo + lo*hi (fp32 accumulate).
//   * T = bf16 | f16 operand planes.  P (in (0, 1]) is split with a TRUNCATED hi part, so that lo = v - hi is exact in fp32:
//     f16: v_cvt_pkrtz for a pair (subnormal results are kept: tools/micro/split_probe.hip); bf16: mask + subtract.
#include <cstdlib>
#include <mutex>

#include "dev_common.h"
#include "kernels.h"

namespace vtq {
namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

// Split 8 probabilities into the hi / lo MFMA fragments (element j of the fragment = p[j]).
template <typename T>
__device__ __forceinline__ void split_p8(const float (&p)[8], typename Vec<T>::x8& hi, typename Vec<T>::x8& lo) {
    if constexpr (std::is_same<T, f16>::value) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const auto hp = __builtin_amdgcn_cvt_pkrtz(p[2 * j], p[2 * j + 1]);      // truncation: hi <= p, p - hi exact in fp32
            const h2 hh = __builtin_bit_cast(h2, hp);
            hw[j] = __builtin_bit_cast(uint32_t, hp);
            // plain C++ (v_cvt_f32_f16 + v_sub): an inline-asm v_fma_mix here read v_exp results inside the hardware's
            // trans -> VALU forwarding window, which hipcc does not pad for asm operands: rare wrong lo halves (measured)
            lw[j] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(p[2 * j] - (float)hh[0], p[2 * j + 1] - (float)hh[1]));
        }
        typedef __attribute__((ext_vector_type(4))) uint32_t u4;
        hi = __builtin_bit_cast(f16x8, u4{hw[0], hw[1], hw[2], hw[3]});
        lo = __builtin_bit_cast(f16x8, u4{lw[0], lw[1], lw[2], lw[3]});
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t hb = __builtin_bit_cast(uint32_t, p[j]) & 0xFFFF0000u;     // bf16 by truncation
            const float hf = __builtin_bit_cast(float, hb);
            hi[j] = __builtin_bit_cast(bf16, (unsigned short)(hb >> 16));
            lo[j] = (bf16)(p[j] - hf);
        }
    }
}

template <typename T, int NSPLIT>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ qkv, int64_t plane, T* __restrict__ out,
                                                        int64_t o_plane, int S, int S_pad, int H, float out8_scale, Fp8Obs obs) {
    typedef typename Vec<T>::x8 tx8;
    typedef typename Vec<T>::x4 tx4;
    constexpr int NPL = (NSPLIT == 1) ? 1 : 2;
    constexpr int KT = 64, KB = KT / 32;     // 64-key K/V tiles = two 32-key blocks; two LDS buffers (3-deep rings and 32-key tiles: no gain)
    constexpr int TB = KT * 128;             // one KT-key x 64-dim bf16 tile
    constexpr int STAGE = TB * NPL * 2;      // K planes then V planes
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hh = lane >> 5;
    // 1-D grid, XCD-aware: the q-blocks of one (sequence, head) re-read the same K/V, so they must share an L2.
    // Blocks are dealt round-robin over the 8 XCDs; remap so each XCD owns a contiguous range of work ids (bijective).
    const int nqb = (S_pad + 127) / 128, nh = H / 64;
    int wid = blockIdx.x;
    {
        const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = wid & 7, idx = wid >> 3;
        wid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int qb = wid % nqb;
    const int head = (wid / nqb) % nh;
    const int seq = wid / (nqb * nh);
    const int ld = 3 * H;
    const int64_t row0 = (int64_t)seq * S_pad;
    const int q_row = qb * 128 + wave * 32 + c;

    // ---- Q fragments: B operand of S^T = K Q^T, element j <-> d = 16t + 8hh + j ------------------------------
    tx8 qf[NPL][4];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            qf[pl][t] = *(const tx8*)(qkv + pl * plane + (row0 + q_row) * ld + head * 64 + 16 * t + 8 * hh);

    // ---- DMA source offsets (elements) of this thread for the two rounds of a 64-row tile --------------------
    uint32_t k_off[KB], v_off[KB];
#pragma unroll
    for (int r = 0; r < KB; ++r) {
        const int slot = r * 256 + tid;
        const int row = slot >> 3, s = slot & 7;
        k_off[r] = (uint32_t)(row * ld + H + head * 64 + ((s ^ ((row >> 1) & 7)) << 3));
        v_off[r] = (uint32_t)(row * ld + 2 * H + head * 64 + ((s ^ (((row >> 1) & 1) << 2)) << 3));
    }
    auto stage = [&](int t, int buf) {
        char* sb = smem + buf * STAGE + wave * 1024;
        const T* base = qkv + (row0 + (int64_t)t * KT) * ld;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int r = 0; r < KB; ++r) {
                glds16(base + pl * plane + k_off[r], sb + pl * TB + r * 4096);
                glds16(base + pl * plane + v_off[r], sb + (NPL + pl) * TB + r * 4096);
            }
    };

    // K fragment read offset: row = kb*32 + c, chunk = 2t + hh, swizzle (row>>1)&7 == (c>>1)&7
    const int k_rd = c * 128;
    const int k_sw = (c >> 1) & 7;
    // V transposed-read offsets: 16-lane group g, lane i = 4*qq + pp supplies row qq, columns 4pp..4pp+3
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int v_row = 4 * (g >> 1) + qq;                                        // + kb*32 + 16*s2 (+8)
    const int v_colb = ((16 * (g & 1) + 4 * pp) * 2) ^ (((qq >> 1) & 1) << 6);  // d-block toggles bit 6 too (XOR)

    f32x16 o_acc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float sc = 0.125f * 1.4426950408889634f;   // 1/sqrt(64) * log2(e)

    // two K/V buffers: tile t+1 is in flight while tile t is consumed, one barrier per tile
    const int nt = (S + KT - 1) / KT;          // the last tile may run into the next sequence's rows: keys >= S are masked
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        const int nxt = cur ^ 1;
        if (t + 1 < nt) stage(t + 1, nxt);
        const char* sk = smem + cur * STAGE;
        const char* sv = sk + NPL * TB;

        // ---- S^T[key][q] for the 64 keys of this tile ---------------------------------------------------------
        f32x16 sacc[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int off = kb * 32 * 128 + k_rd + (((2 * tt + hh) ^ k_sw) << 4);
                const tx8 kf = *(const tx8*)(sk + off);
                sacc[kb] = mfma32<T>(kf, qf[0][tt], tt == 0 ? zero16 : sacc[kb]);
                if constexpr (NSPLIT == 3) {
                    const tx8 kl = *(const tx8*)(sk + TB + off);
                    sacc[kb] = mfma32<T>(kf, qf[1][tt], sacc[kb]);
                    sacc[kb] = mfma32<T>(kl, qf[0][tt], sacc[kb]);
                }
            }
        }

        // ---- online softmax (base-2 domain; scale folded into one FMA per score) -----------------------------------
        if ((t + 1) * KT > S) {                 // wave-uniform: only the last tile(s) hold padded keys
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= S) sacc[kb][r] = -INFINITY;
                }
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sacc[kb][r]), sacc[kb][r + 1]);   // v_max3_f32
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);                                              // raw (unscaled) running max
        const float nm = -m_new * sc;
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], sc, nm));
                sacc[kb][r] = pv;
                rs += pv;
            }
        if (__builtin_amdgcn_ballot_w64(m_new > m_run)) {        // some row's max moved: rescale (exact; usually skipped)
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
        }
        m_run = m_new;
        l_run += rs;

        // ---- O^T[d][q] += V^T[d][key] P^T[key][q] -------------------------------------------------------------
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                tx8 ph, pl_;
                if constexpr (NSPLIT == 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ph[j] = (T)sacc[kb][8 * s2 + j];
                } else {
                    float pj[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) pj[j] = sacc[kb][8 * s2 + j];
                    split_p8<T>(pj, ph, pl_);
                }
                const int vrow = kb * 32 + 16 * s2 + v_row;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    // byte column of (d-block, 16-column half, 4-column piece); bit 6 carries the row swizzle
                    const char* a0 = sv + vrow * 128 + (v_colb ^ (d << 6));
                    const s16x4 v0 = lds_tr16(a0);
                    const s16x4 v1 = lds_tr16(a0 + 8 * 128);
                    const tx8 vf = __builtin_bit_cast(tx8, s16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]});
                    o_acc[d] = mfma32<T>(vf, ph, o_acc[d]);
                    if constexpr (NSPLIT == 3) {
                        const s16x4 w0 = lds_tr16(a0 + TB);
                        const s16x4 w1 = lds_tr16(a0 + TB + 8 * 128);
                        const tx8 vl = __builtin_bit_cast(tx8, s16x8{w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]});
                        o_acc[d] = mfma32<T>(vf, pl_, o_acc[d]);
                        o_acc[d] = mfma32<T>(vl, ph, o_acc[d]);
                    }
                }
            }

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // tile t+1 has landed
        __syncthreads();
        cur = nxt;
    }

    // ---- normalise and write merged heads: out[row][head*64 + d] -------------------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    float amax8 = 0.f;                      // fp8 mode: max |context value| of this lane's real rows (kernels.h Fp8Obs)
    if (q_row < S_pad) {
        T* o = out + (row0 + q_row) * H + head * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dcol = 32 * d + 8 * g4 + 4 * hh;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o_acc[d][4 * g4 + e] * inv;
                if (out8_scale > 0.f) {          // fp8 mode: the out-proj GEMM reads e4m3 bytes
                    if (q_row < S) amax8 = amax4(amax8, v[0], v[1], v[2], v[3]);
                    *(uint32_t*)((uint8_t*)out + (row0 + q_row) * H + head * 64 + dcol) =
                        pack_fp8x4(v[0] * out8_scale, v[1] * out8_scale, v[2] * out8_scale, v[3] * out8_scale);
                } else if constexpr (NSPLIT == 1) {
                    tx4 hv = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
                    *(tx4*)(o + dcol) = hv;
                } else {
                    tx4 hv, lv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { T a, b; split2<T>(v[e], a, b); hv[e] = a; lv[e] = b; }
                    *(tx4*)(o + dcol) = hv;
                    *(tx4*)(o + o_plane + dcol) = lv;
                }
            }
    }
    if (out8_scale > 0.f) fp8_report(obs, amax8, out8_scale);
}

}  // namespace

template <typename T, int NSPLIT>
hipError_t launch_attention_t(const void* qkv, int64_t plane, void* out, int64_t o_plane, int nseq, int S, int S_pad, int H, hipStream_t s,
                              float out8_scale, Fp8Obs obs) {
    constexpr int LDS = 2 * 2 * 64 * 128 * (NSPLIT == 1 ? 1 : 2);
    static std::mutex mu;
    static bool configured[64] = {false};          // hipFuncSetAttribute is per device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!configured[dev]) {
            e = hipFuncSetAttribute((const void*)attention_kernel<T, NSPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) return e;
            configured[dev] = true;
        }
    }
    const dim3 grid(((S_pad + 127) / 128) * (H / 64) * nseq), blk(256);
    hipLaunchKernelGGL((attention_kernel<T, NSPLIT>), grid, blk, LDS, s, (const T*)qkv, plane, (T*)out, o_plane, S, S_pad, H, out8_scale, obs);
    return hipGetLastError();
}

hipError_t launch_attention(const void* qkv, int64_t plane, void* out, int64_t o_plane, int nseq, int S, int S_pad, int H,
                            Num num, hipStream_t s, float out8_scale, Fp8Obs obs) {
    if (H % 64 || S < 1 || S > S_pad || S <= S_pad - 64 || (num.terms != 1 && num.terms != 3) || num.f16 > 1) return hipErrorInvalidValue;
    if (!num.f16) {
        if (num.terms == 1) return launch_attention_t<bf16, 1>(qkv, plane, out, o_plane, nseq, S, S_pad, H, s, out8_scale, obs);
        return launch_attention_t<bf16, 3>(qkv, plane, out, o_plane, nseq, S, S_pad, H, s, out8_scale, obs);
    }
    if (num.terms == 1) return launch_attention_t<f16, 1>(qkv, plane, out, o_plane, nseq, S, S_pad, H, s, out8_scale, obs);
    return launch_attention_t<f16, 3>(qkv, plane, out, o_plane, nseq, S, S_pad, H, s, out8_scale, obs);
}

}  // namespace vtq
