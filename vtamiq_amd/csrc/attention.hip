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
#include <atomic>
#include <cstdlib>
#include <mutex>

#include "dev_common.h"
#include "kernels.h"

namespace vtq {
namespace {

// Diagnostic builds (-DVTQ_ATTN_DIAG, tools/build_abl.sh) accumulate per-wave s_memtime spans of the phases of a tile into the
// buffer of gemm_set_diag (16 words per wave; tools/attn_probe.py); the shipped library executes no stamp.
#ifdef VTQ_ATTN_DIAG
#define VTQ_AT_STAMP(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define VTQ_AT_SPAN(acc) { unsigned long long dg_t1; VTQ_AT_STAMP(dg_t1); acc += dg_t1 - dg_t; dg_t = dg_t1; }
#else
#define VTQ_AT_SPAN(acc)
#endif

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

// Two scores at a time: the exponent's argument and the row sum as packed fp32 operations (v_pk_add_f32 / v_pk_fma_f32: two lanes'
// worth of work per vector issue slot; the exponential itself has no packed form).  The row sum therefore runs as TWO partial sums
// (even / odd accumulator registers), added at the end of a tile -- both kernels use this helper, so they stay bit-identical.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NSPLIT>
__device__ __forceinline__ void exp_pair(f32x16& v, int r, float m_new, float sc, float nm, f32x2& rs2) {      // registers r, r + 1 of v
    f32x2 t = {v[r], v[r + 1]};
    if constexpr (NSPLIT == 3) t = t - f32x2{m_new, m_new};
    else t = t * f32x2{sc, sc} + f32x2{nm, nm};
    f32x2 pv = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    v[r] = pv[0];
    v[r + 1] = pv[1];
    rs2 += pv;
}

// 3-term formats: the softmax scale (1/sqrt(64) * log2 e) is folded into Q once per query block -- q c = (hi + lo) c in fp32, split
// again -- so that scores arrive in log2 units and the exponent is exp2(s - m): one subtraction that is EXACT for the row maximum at
// any magnitude.  (The one-FMA form exp2(s c - m c) subtracts the rounded product m c: fine at ordinary logits, inf at the 1e13
// logits of a 1e7-gain model, where the fp32 reference is finite; the single-plane formats keep it behind a magnitude guard, since
// re-rounding q c to 11 bits would cost them accuracy.)
template <typename T>
__device__ __forceinline__ void prescale_q(typename Vec<T>::x8 (&qf)[2][4], float sc) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = ((float)qf[0][t][j] + (float)qf[1][t][j]) * sc;
            T a, b;
            split2<T>(v, a, b);
            qf[0][t][j] = a;
            qf[1][t][j] = b;
        }
}

template <typename T, int NSPLIT>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ qkv, int64_t plane, T* __restrict__ out,
                                                        int64_t o_plane, int S, int S_pad, int H, float out8_scale, Fp8Obs obs,
                                                        unsigned long long* diag, int q_log2, int q0, int Sq) {
    // q0, Sq: this launch covers the query rows [q0, Sq) of every sequence (0, S_pad: all of them; launch_attention's split form gives
    // the rows behind the last full 256-row block to this kernel)
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
#ifdef VTQ_ATTN_DIAG
    unsigned long long dg_k0, dg_r0, dg_t, dg_qk = 0, dg_sm = 0, dg_pv = 0, dg_bar = 0, dg_stage = 0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_k0), "=s"(dg_r0) :: "memory");
#endif
    // 1-D grid, XCD-aware: the q-blocks of one (sequence, head) re-read the same K/V, so they must share an L2.
    // Blocks are dealt round-robin over the 8 XCDs; remap so each XCD owns a contiguous range of work ids (bijective).
    const int nqb = (Sq - q0 + 127) / 128, nh = H / 64;
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
    const int q_row = q0 + qb * 128 + wave * 32 + c;
    // a wave whose 32 query rows all lie behind the sequence (ragged last block) stages and synchronises, nothing else
    const bool wave_active = (q0 + qb * 128 + wave * 32) < S_pad;

    // ---- Q fragments: B operand of S^T = K Q^T, element j <-> d = 16t + 8hh + j ------------------------------
    tx8 qf[2][4];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            qf[pl][t] = *(const tx8*)(qkv + pl * plane + (row0 + q_row) * ld + head * 64 + 16 * t + 8 * hh);
    if constexpr (NSPLIT == 3) { if (!q_log2) prescale_q<T>(qf, 0.125f * 1.4426950408889634f); }

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
    // The masked keys' probabilities are exactly 0, but their V rows belong to the NEXT sequence (sequences are packed back to back): a NaN /
    // inf there would reach this sequence through 0 x NaN.  Every thread therefore zeroes, in the LDS image of the LAST key tile, the V pieces
    // it staged itself whose row is a masked key -- after its DMA has landed (the vmcnt wait), before the barrier that publishes the tile.
    // (K needs nothing: a NaN score of a masked key is replaced by -inf with a select.)
    const int tail_valid = S - (nt - 1) * KT;   // real keys in the last tile (1 .. 64)
    auto zero_masked_v = [&](int buf) {
#pragma unroll
        for (int r = 0; r < KB; ++r) {
            const int row = (r * 256 + tid) >> 3;
            if (row >= tail_valid) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr(smem + buf * STAGE + (NPL + pl) * TB + r * 4096 + tid * 16)), "v"(z) : "memory");
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef VTQ_ATTN_NO_VMASK                          // measurement builds only: the cost of the masked-key V rows' zeroing (A/B)
    if (nt == 1) zero_masked_v(0);
#endif
    __syncthreads();
    int cur = 0;
#ifdef VTQ_ATTN_DIAG
    unsigned long long dg_loop0, dg_loop1;
    VTQ_AT_STAMP(dg_loop0);
#endif
    for (int t = 0; t < nt; ++t) {
        const int nxt = cur ^ 1;
#ifdef VTQ_ATTN_DIAG
        VTQ_AT_STAMP(dg_t);
#endif
        if (t + 1 < nt) stage(t + 1, nxt);
        VTQ_AT_SPAN(dg_stage);
        const char* sk = smem + cur * STAGE;
        const char* sv = sk + NPL * TB;

        if (wave_active) {
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

        VTQ_AT_SPAN(dg_qk);
        // ---- online softmax (base-2 domain; scale folded into one FMA per score) -----------------------------------
        if ((t + 1) * KT > S) {                 // wave-uniform: only the last tile(s) hold padded keys
            int hq = 4 * hh;
            asm volatile("" : "+v"(hq));         // keeps the 31 key offsets of this rare path out of loop-invariant registers
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + hq;
                    if (key >= S) sacc[kb][r] = -INFINITY;
                }
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sacc[kb][r]), sacc[kb][r + 1]);   // v_max3_f32
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);                                              // running max (3-term: log2 units; else raw)
        float nm = -m_new * sc;
        f32x2 rs2 = {0.f, 0.f};
        if constexpr (NSPLIT == 3) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) exp_pair<NSPLIT>(sacc[kb], r, m_new, sc, nm, rs2);
        } else {
            // single plane: exp2(s c - m c) as ONE FMA per score is exact enough only while |m c| is small: the FMA subtracts the ROUNDED
            // product m c, so the maximum's own exponent is the rounding error of m c (2^-24 |m c|) instead of 0 -- harmless at
            // |m c| <= 64 (3e-6), inf at the logits of a 1e7-gain model (the fp32 reference returns finite scores there).  Beyond 64
            // (rare; the branch is wave-uniform, the decision per query row, so a row's bits never depend on its wave's other rows):
            // subtract the maximum first, then the same FMA with a zero addend.
            const bool big = fabsf(nm) > 64.f;
            if (__builtin_amdgcn_ballot_w64(big)) {
                const float sub = big ? m_new : 0.f;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[kb][r] -= sub;
                nm = big ? 0.f : nm;
            }
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) exp_pair<NSPLIT>(sacc[kb], r, m_new, sc, nm, rs2);
        }
        const float rs = rs2[0] + rs2[1];
        if (__builtin_amdgcn_ballot_w64(m_new > m_run)) {        // some row's max moved: rescale (exact; usually skipped)
            const float alpha = __builtin_amdgcn_exp2f(NSPLIT == 3 ? m_run - m_new : (m_run - m_new) * sc);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
        }
        m_run = m_new;
        l_run += rs;
        VTQ_AT_SPAN(dg_sm);

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

        VTQ_AT_SPAN(dg_pv);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // tile t+1 has landed
#ifndef VTQ_ATTN_NO_VMASK
        if (t + 2 == nt) zero_masked_v(nxt);                // ... and it is the last one: its masked keys' V rows become zeros
#endif
        __syncthreads();
        VTQ_AT_SPAN(dg_bar);
        cur = nxt;
    }

#ifdef VTQ_ATTN_DIAG
    VTQ_AT_STAMP(dg_loop1);
#endif
    // ---- normalise and write merged heads: out[row][head*64 + d] -------------------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    float amax8 = 0.f;                      // fp8 mode: max |context value| of this lane's real rows (kernels.h Fp8Obs)
    if (out8_scale > 0.f) {                 // fp8 mode: the out-proj GEMM reads e4m3 bytes (direct stores, one 64-byte piece per row)
        if (q_row < S_pad) {
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int dcol = 32 * d + 8 * g4 + 4 * hh;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = o_acc[d][4 * g4 + e] * inv;
                    if (q_row < S) amax8 = amax4(amax8, v[0], v[1], v[2], v[3]);
                    *(uint32_t*)((uint8_t*)out + (row0 + q_row) * H + head * 64 + dcol) =
                        pack_fp8x4(v[0] * out8_scale, v[1] * out8_scale, v[2] * out8_scale, v[3] * out8_scale);
                }
        }
    } else if constexpr (NSPLIT == 1) {
        // Single plane: staged through LDS (the K / V buffers are free: the loop ended with a barrier), as the pipelined kernel does -- each wave transposes
        // its 32 rows x 128 bytes through its own 4 KB image (chunk index XORed with row & 7), so that every store instruction writes eight whole 128-byte
        // row segments, 16 B per lane, instead of 32 pieces of 8 B at the row stride.  Same values.  Inside a forward (fp16 mode, B = 32) 89.5 -> 83.3 us;
        // the two-plane formats keep the direct stores below (staged: S = 521 +2 %, S = 1025 +3 %; profiles/r06_attention_loop.txt section 11).
        if (wave_active) {
            char* const o_stage = smem + wave * 4096;
            const int r_row = lane >> 3, r_chunk = lane & 7;
#pragma unroll
            for (int k = 0; k < 8; ++k) {                         // chunk k = 4 d + g4 of row c, bytes 8 hh .. 8 hh + 7
                tx4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = o_acc[k >> 2][4 * (k & 3) + e] * inv;
                    asm volatile("" : "+v"(v));                   // rounded product, then converted (no fused form)
                    hv[e] = (T)v;
                }
                *(tx4*)(o_stage + c * 128 + ((k ^ (c & 7)) << 4) + 8 * hh) = hv;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = r_row + 8 * k;
                const uint4 w = *(const uint4*)(o_stage + row * 128 + ((r_chunk ^ (row & 7)) << 4));
                const int qr = q0 + qb * 128 + wave * 32 + row;
                if (qr < S_pad) *(uint4*)(out + (row0 + qr) * H + head * 64 + 8 * r_chunk) = w;
            }
        }
    } else if (q_row < S_pad) {
        T* o = out + (row0 + q_row) * H + head * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dcol = 32 * d + 8 * g4 + 4 * hh;
                tx4 hv, lv;
#pragma unroll
                for (int e = 0; e < 4; ++e) { T a, b; split2<T>(o_acc[d][4 * g4 + e] * inv, a, b); hv[e] = a; lv[e] = b; }
                *(tx4*)(o + dcol) = hv;
                *(tx4*)(o + o_plane + dcol) = lv;
            }
    }
    if (out8_scale > 0.f) fp8_report(obs, amax8, out8_scale);
#ifdef VTQ_ATTN_DIAG
    if (diag) {
        unsigned long long dg_k1, dg_r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_k1), "=s"(dg_r1) :: "memory");
        if (lane == 0) {          // one 16-word slot per wave, plain stores (same-address atomics of 12 k waves would take a millisecond)
            unsigned long long* dq = diag + ((size_t)blockIdx.x * 4 + wave) * 16;
            dq[0] = dg_k1 - dg_k0; dq[1] = dg_r1 - dg_r0; dq[2] = dg_qk; dq[3] = dg_sm; dq[4] = dg_pv; dq[5] = dg_bar; dq[6] = 1; dq[7] = nt; dq[8] = dg_stage;
            dq[9] = dg_loop0 - dg_k0; dq[10] = dg_k1 - dg_loop1;
        }
    }
#endif
}


// =====================================================================================================================
// Helpers of the software-pipelined kernel below.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// LDS fragment reads hidden from hipcc's waitcnt bookkeeping ("=v" outputs; the consumer is fenced by an s_waitcnt statement that names
// the destinations "+v": cdna_hip_programming.md 'What hipcc does not do', form (ii))
#define PP_DS_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(dst) : "v"(addr), "i"(off))
#define PP_DS_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%c2" : "=v"(dst) : "v"(addr), "i"(off))

template <typename T, int NSPLIT>
#ifndef VTQ_SW_DIST
#define VTQ_SW_DIST 2                 // LDS fragment groups read ahead of the MFMAs that consume them (2 or 3; 3 measured: see the profile)
#endif
struct PPFrags {                      // DIST + 1 rolling fragment buffers: group G (QK^T 0..7, PV 8..15 of a tile) uses slot G % (DIST + 1)
    u32x4 ka[VTQ_SW_DIST + 1], kl[VTQ_SW_DIST + 1];               // QK^T groups: K fragment hi / lo (ds_read_b128)
    u32x2 va0[VTQ_SW_DIST + 1], va1[VTQ_SW_DIST + 1], vl0[VTQ_SW_DIST + 1], vl1[VTQ_SW_DIST + 1];   // PV groups: V^T fragment halves (ds_read_b64_tr_b16)
};

// =====================================================================================================================
// Software-pipelined form (used for the 3-term formats when 256-row blocks fill the chip; launch_attention).
// Workgroup = 8 waves = 256 query rows, PERSISTENT over a contiguous list of (sequence, head, 256-row block) items -- the blocks of one
// (sequence, head) are consecutive, so their K/V come from L2 the second time -- with ONE continuous stream of 64-key K/V tiles through
// a 4-deep LDS ring (slot = stream index & 3) that does not stop at block seams.  Every wave runs one instruction stream per tile in
// which the vector work sits in the shadow of its own MFMAs (cdna_hip_programming.md 'one-wave-per-SIMD' rules: <= 5 issues and
// <= 1 transcendental per MFMA gap; sched_group_barrier):
//   phase 1: the 24 MFMAs of QK^T of tile t + 1  ||  hi / lo split of P(t), rescale of O when the running max moved
//   phase 2: the 24 MFMAs of PV of tile t          ||  softmax of tile t + 1 (max, exp2, row sum)
// LDS fragments are read two groups (6 MFMAs) ahead by asm reads with counted lgkmcnt; one s_barrier per tile.  In iteration t every
// wave issues its eighth of tile t + 3 (ring slot last read in iteration t - 1); at the end of the iteration vmcnt(NI) -- everything
// but that tile has landed -- precedes the barrier.  At a block seam: the next block's Q is loaded IN PLACE (asm, hidden from hipcc's
// vmcnt bookkeeping, which would otherwise drain the LDS-DMA in every iteration) right after the last QK^T that needs the old Q and
// completes behind the same counted wait; the finished block's O is normalised, split and staged through 4 KB of LDS per wave in the
// middle of the NEXT iteration (phase 1 does not touch O), so that its stores are 16 B per lane on whole 128-byte row segments, older
// than that iteration's LDS-DMA and covered by its phase 2.  Waves whose rows lie behind the sequence (ragged last block) only load.
// Same arithmetic in the same order per query row as attention_kernel: outputs are bit-identical (tools/attn_ab.py, tests).
// Measurements, the skeleton ablations behind the switches below and the ping-pong variant that lost: profiles/r03_attention_anatomy.txt.
#ifndef VTQ_SW_NOFILL
#define VTQ_SW_NOFILL 0
#endif
#ifndef VTQ_SW_QPF
#define VTQ_SW_QPF 1                  // the next block's Q rows are pulled into L2 two iterations before they are loaded (0: measurement builds)
#endif
#ifndef VTQ_SW_EARLY_WRITE
#define VTQ_SW_EARLY_WRITE 1          // a finished block's output is written at the top of the next iteration (0: in its middle, the round-3 place)
#endif
#ifndef VTQ_SW_PRIO
#define VTQ_SW_PRIO 0                 // measurement builds: issue priority alternating between the two waves of a SIMD (1: per phase, 2: per fragment group, 3: static for waves 4-7)
#endif
#ifndef VTQ_SW_NOSTORE
#define VTQ_SW_NOSTORE 0
#endif
#ifndef VTQ_SW_PAIRED
#define VTQ_SW_PAIRED 1
#endif
#ifndef VTQ_SW_NOQ
#define VTQ_SW_NOQ 0
#endif
#ifndef VTQ_SW_NODMA
#define VTQ_SW_NODMA 0
#endif
#ifndef VTQ_SW_NOMFMA
#define VTQ_SW_NOMFMA 0
#endif
template <typename T>
__device__ __forceinline__ f32x16 SW_MFMA(typename Vec<T>::x8 a, typename Vec<T>::x8 b, f32x16 c) {
#if VTQ_SW_NOMFMA
    asm volatile("" :: "v"(a), "v"(b));
    return c;
#else
    return mfma32<T>(a, b, c);
#endif
}
#define SW_MFMA_VALU(n)                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          \
    __builtin_amdgcn_sched_group_barrier(0x002, n, 0)

template <typename T>
__device__ __forceinline__ void split_p4(const float (&p)[4], uint32_t (&hw)[2], uint32_t (&lw)[2]) {
    if constexpr (std::is_same<T, f16>::value) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const auto hp = __builtin_amdgcn_cvt_pkrtz(p[2 * j], p[2 * j + 1]);
            const h2 hh = __builtin_bit_cast(h2, hp);
            hw[j] = __builtin_bit_cast(uint32_t, hp);
            lw[j] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(p[2 * j] - (float)hh[0], p[2 * j + 1] - (float)hh[1]));
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint32_t h[2], l[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t hb = __builtin_bit_cast(uint32_t, p[2 * j + e]) & 0xFFFF0000u;
                const float hf = __builtin_bit_cast(float, hb);
                h[e] = hb >> 16;
                l[e] = (uint32_t)__builtin_bit_cast(unsigned short, (bf16)(p[2 * j + e] - hf));
            }
            hw[j] = h[0] | (h[1] << 16);
            lw[j] = l[0] | (l[1] << 16);
        }
    }
}

// LDS reads of fragment group G of a tile (QK^T groups 0..7: K hi [+ lo] by ds_read_b128; PV groups 8..15: V^T hi [+ lo] by two
// ds_read_b64_tr_b16 each), and the reads still in flight when group G is consumed: those of the next VTQ_SW_DIST groups below `end`
#ifndef VTQ_SW_HALFREADS
#define VTQ_SW_HALFREADS 0            // measurement builds: every second fragment group is not read (its registers keep the previous group's): what
#endif                                //   a kernel with HALF the K / V fragment reads per MFMA -- 64 query rows per wave -- could save at most (results wrong)
template <int NSPLIT>
constexpr int sw_reads(int G) {
    if (VTQ_SW_HALFREADS && (G & 1)) return 0;
    return G < 8 ? (NSPLIT == 1 ? 1 : 2) : (G < 16 ? (NSPLIT == 1 ? 2 : 4) : 0);
}
template <int NSPLIT>
constexpr int sw_ahead(int G, int end) {
    int n = 0;
    for (int j = 1; j <= VTQ_SW_DIST; ++j)
        if (G + j < end) n += sw_reads<NSPLIT>(G + j);
    return n;
}

template <typename T, int NSPLIT>
__global__ __launch_bounds__(512) void attention_sw_kernel(const T* __restrict__ qkv, int64_t plane, T* __restrict__ out, int64_t o_plane,
                                                           int S, int S_pad, int H, int nblk, int per, float out8_scale, Fp8Obs obs,
                                                           unsigned long long* diag, int q_log2, int Sq) {
    // Sq: this launch covers the query rows [0, Sq) of every sequence -- S_pad (all of them), or the full 256-row blocks only when
    // launch_attention hands the short rest to the 4-wave kernel; S_pad stays the pitch of a sequence
    typedef typename Vec<T>::x8 tx8;
    typedef typename Vec<T>::x4 tx4;
    constexpr int NPL = (NSPLIT == 1) ? 1 : 2;
    constexpr int KT = 64, TB = KT * 128, STAGE = TB * NPL * 2, NI = 2 * NPL;
    // (LDS reads per fragment group: sw_reads above)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hh = lane >> 5;
    const int nqb = (Sq + 255) / 256, nh = H / 64;
    const int nt = (S + KT - 1) / KT;
    const int ld = 3 * H;
    // Blocks of this workgroup: b0, b0 + bstep, ... < b1.
    // per == 0 (the default): XCD-STRIDED.  XCD x (= blockIdx & 7 under the round-robin dispatch; a speed assumption only) owns a contiguous run of
    // (sequence, head) items, and its workgroups walk that run together: slot sl takes blocks r0 + sl, r0 + sl + n, ... (n = workgroups of the
    // XCD).  At any time the CUs of an XCD therefore work on the SAME few (sequence, head) pairs -- all nqb query blocks of a pair run side by
    // side and in step -- so every K / V tile comes from HBM once and from that XCD's L2 for the other readers, whatever nqb is (S = 501: the
    // two blocks of a pair, as the paired form before; S = 1025: four; S = 5001: twenty -- a pair's K / V there are 2.5 MB, which one CU walking
    // its blocks one after the other re-read from the Infinity Cache nqb times).
    // per > 0 (measurement: vtq_debug_attention_map(1)): the round-3 form -- `per` consecutive blocks per workgroup, or, with two query blocks per
    // pair and an evenly divided grid, workgroups w and w + 8 on the two blocks of the same pairs.
    int b0, b1, bstep;
    if (per == 0) {
        const int x = blockIdx.x & 7, sl = blockIdx.x >> 3;
        const int nx = ((int)gridDim.x - x + 7) >> 3;
        const int items = nblk / nqb;
        const int i0 = (int)((long long)items * x >> 3), i1 = (int)((long long)items * (x + 1) >> 3);
        b0 = i0 * nqb + sl;
        bstep = nx;
        b1 = i1 * nqb;
    } else if (VTQ_SW_PAIRED && nqb == 2 && (gridDim.x & 15) == 0 && (int)gridDim.x * per == nblk) {
        const int x = blockIdx.x & 7, sl = blockIdx.x >> 3;
        const int g = x * ((int)gridDim.x >> 4) + (sl >> 1);
        b0 = 2 * g * per + (sl & 1);
        bstep = 2;
        b1 = b0 + 2 * per;
    } else {
        b0 = blockIdx.x * per;
        b1 = (b0 + per < nblk) ? b0 + per : nblk;
        bstep = 1;
    }
    if (b0 >= b1) return;
#if defined(VTQ_SW_SEAM_STAGGER) && VTQ_SW_SEAM_STAGGER > 0
    // Measurement builds (tools/runs/r05g_attn_seams.sh): every second workgroup of an XCD starts VTQ_SW_SEAM_STAGGER microseconds late, so that the
    // block seams of the chip (Q loads, output stores) do not coincide; s_sleep 32 = ~2048 cycles = ~1 us.  Same results.
    if ((blockIdx.x >> 3) & 1)
        for (int i = 0; i < VTQ_SW_SEAM_STAGGER; ++i) __builtin_amdgcn_s_sleep(32);
#endif
    const int NT = ((b1 - b0 + bstep - 1) / bstep) * nt;
    const float sc = 0.125f * 1.4426950408889634f;
#ifdef VTQ_ATTN_DIAG
    unsigned long long dg_k0, dg_r0, dg_t, dg_p1 = 0, dg_p2 = 0, dg_bar = 0, dg_pro = 0, dg_rest = 0;
    unsigned long long dg_it0 = 0, dg_kind[3] = {0, 0, 0}, dg_nkind[3] = {0, 0, 0}, dg_wb = 0, dg_tail0 = 0;   // whole iterations by kind: plain / writes a block / loads Q
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_k0), "=s"(dg_r0) :: "memory");
    dg_t = dg_k0;
#endif

    auto block_base = [&](int b, int& qb) __attribute__((always_inline)) -> int64_t {
        qb = b % nqb;
        const int head = (b / nqb) % nh, seq = b / (nqb * nh);
        return (int64_t)seq * S_pad * ld + head * 64;
    };
    auto block_out = [&](int b) __attribute__((always_inline)) -> int64_t {
        const int head = (b / nqb) % nh, seq = b / (nqb * nh);
        return (int64_t)seq * S_pad * H + head * 64;
    };
    auto load_q = [&](int b, tx8 (&qf)[2][4]) __attribute__((always_inline)) {
        int qb;
        const int64_t base = block_base(b, qb);
        int qr = qb * 256 + wave * 32 + c;
        qr = qr < Sq ? qr : Sq - 1;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int t = 0; t < 4; ++t) qf[pl][t] = *(const tx8*)(qkv + pl * plane + base + (int64_t)qr * ld + 16 * t + 8 * hh);
    };

    // the same loads hidden from hipcc's vmcnt bookkeeping (in place: no copy of a register whose data has not landed); the caller
    // waits with a counted vmcnt that names qf
    auto load_q_async = [&](int b, tx8 (&qf)[2][4]) __attribute__((always_inline)) {
        int qb;
        const int64_t base = block_base(b, qb);
        int qr = qb * 256 + wave * 32 + c;
        qr = qr < Sq ? qr : Sq - 1;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const T* ptr = qkv + pl * plane + base + (int64_t)qr * ld + 16 * t + 8 * hh;
                asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(qf[pl][t]) : "v"(ptr) : "memory");
            }
    };
    auto pin_q = [&](tx8 (&qf)[2][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(qf[pl][t]));
    };

    const int d_row = tid >> 3, d_s = tid & 7;
    const uint32_t k_off = (uint32_t)(d_row * ld + H + ((d_s ^ ((d_row >> 1) & 7)) << 3));
    const uint32_t v_off = (uint32_t)(d_row * ld + 2 * H + ((d_s ^ (((d_row >> 1) & 1) << 2)) << 3));
    int ib = b0, it = 0, itau = 0, iqb;
    int64_t ibase = block_base(ib, iqb);
    auto issue_tile = [&]() __attribute__((always_inline)) -> bool {
        if (ib >= b1) return false;
        char* sb = smem + (itau & 3) * STAGE + wave * 1024;
        const T* base = qkv + ibase + (int64_t)it * KT * ld;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            if (VTQ_SW_NODMA) break;
            glds16(base + pl * plane + k_off, sb + pl * TB);
            glds16(base + pl * plane + v_off, sb + (NPL + pl) * TB);
        }
        ++itau;
        if (++it == nt) {
            it = 0;
            if ((ib += bstep) < b1) ibase = block_base(ib, iqb);
        }
        return true;
    };

    tx8 qf[2][4];
    load_q(b0, qf);
    if constexpr (NSPLIT == 3) { if (!q_log2) prescale_q<T>(qf, sc); }          // scores in log2 units (see prescale_q)
    issue_tile();
    issue_tile();
    issue_tile();

    const uint32_t lds0 = lds_addr(smem);
    const int k_sw = (c >> 1) & 7;
    uint32_t k_lane[4], v_lane[2];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) k_lane[tt] = lds0 + c * 128 + (((2 * tt + hh) ^ k_sw) << 4);
    {
        const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
        const int v_row = 4 * (g >> 1) + qq;
        const int v_colb = ((16 * (g & 1) + 4 * pp) * 2) ^ (((qq >> 1) & 1) << 6);
#pragma unroll
        for (int d = 0; d < 2; ++d) v_lane[d] = lds0 + NPL * TB + v_row * 128 + (v_colb ^ (d << 6));
    }

    f32x16 o_acc[2], sA[2], sB[2];               // sA: P of the current tile (fp32), sB: scores, then P, of the next one
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
    uint32_t q_pf = 0;                           // destination of the Q prefetch (never read; reserved while a prefetch may be in flight)
    uint32_t lw_keep[2] = {0, 0};                // lo words of a half step, carried to the group that has room for them (split_half)
    float m_run = -1e30f, l_run = 0.f, l_fin = 0.f;
    float amax8 = 0.f;
    float alpha = 1.f;
    bool rescale = false;

    int cb = b0, ct = 0, cqb;                    // block / tile-in-block of the tile whose PV runs in this iteration
    block_base(cb, cqb);
    bool wr_pending = false;                     // a finished block waits for its output (written in the middle of the next iteration)
    int wr_b = 0, wr_qb = 0;

    // Finished block: O / l, split into the output planes, staged through 4 KB of LDS per wave (above the ring) so that every global
    // store instruction writes eight whole 128-byte row segments (16 B per lane) instead of 32 pieces of 8 B at the row stride.
    // LDS image of one plane: [32 rows][8 chunks of 16 B], chunk index XORed with (row & 7); only this wave touches its region.
    char* const o_stage = smem + 4 * STAGE + wave * 4096;
    auto write_block = [&](float l_blk, int b, int qb) __attribute__((always_inline)) {
        const float l_tot = l_blk + __shfl_xor(l_blk, 32, 64);
        const float inv = 1.0f / l_tot;
        const int q_row = qb * 256 + wave * 32 + c;
        const int64_t obase = block_out(b);
        if (out8_scale > 0.f) {                                    // fp8 mode: bytes, direct stores (one 64-byte piece per row)
            if (q_row < Sq && !VTQ_SW_NOSTORE) {
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int dcol = 32 * d + 8 * g4 + 4 * hh;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = o_acc[d][4 * g4 + e] * inv;
                        if (q_row < S) amax8 = amax4(amax8, v[0], v[1], v[2], v[3]);
                        *(uint32_t*)((uint8_t*)out + obase + (int64_t)q_row * H + dcol) =
                            pack_fp8x4(v[0] * out8_scale, v[1] * out8_scale, v[2] * out8_scale, v[3] * out8_scale);
                    }
            }
        } else {
            const int r_row = lane >> 3, r_chunk = lane & 7;      // read-back: 8 lanes per row, 4 x 8 rows
            // O / l is formed and split ONCE: the hi halves go to the LDS image, the lo halves wait in 16 registers for the second plane.  (Formed per
            // plane with split2, as up to round 5, the block cost 5 200 cycles per wave, most of it the 450 vector instructions of two passes --
            // all 8 waves at once, nothing to overlap with; fp16: the 6-instruction split of four values, bit-identical to split2.)
            tx4 lo_keep[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {                         // chunk k = 4 d + g4 of row c, bytes 8 hh .. 8 hh + 7
                float v4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = o_acc[k >> 2][4 * (k & 3) + e] * inv;
                tx4 hv;
                if constexpr (NSPLIT == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { float u = v4[e]; asm volatile("" : "+v"(u)); hv[e] = (T)u; }   // rounded product, then converted
                } else if constexpr (std::is_same<T, f16>::value) {
                    split4_f16(v4, hv, lo_keep[k]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { T x, y; split2<T>(v4[e], x, y); hv[e] = x; lo_keep[k][e] = y; }
                }
                *(tx4*)(o_stage + c * 128 + ((k ^ (c & 7)) << 4) + 8 * hh) = hv;
            }
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                if (pl == 1) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) *(tx4*)(o_stage + c * 128 + ((k ^ (c & 7)) << 4) + 8 * hh) = lo_keep[k];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = r_row + 8 * k;
                    const uint4 w = *(const uint4*)(o_stage + row * 128 + ((r_chunk ^ (row & 7)) << 4));
                    const int qr = qb * 256 + wave * 32 + row;
                    if (qr < Sq && !VTQ_SW_NOSTORE) *(uint4*)(out + pl * o_plane + obase + (int64_t)qr * H + 8 * r_chunk) = w;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the reads are done before the next plane overwrites the image
            }
        }
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
    };

    // softmax pieces on sB (tile index in its block: tb).  finish: statistics after the last exp2.
    auto mask_tail = [&](int tb) __attribute__((always_inline)) {
        if ((tb + 1) * KT > S) {
            int hq = 4 * hh;
            asm volatile("" : "+v"(hq));         // keeps the 31 key offsets of this rare path out of loop-invariant registers
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = tb * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + hq;
                    if (key >= S) sB[kb][r] = -INFINITY;
                }
        }
    };
    float mx = 0.f, m_new = 0.f, nm = 0.f;
    f32x2 rs2 = {0.f, 0.f};
    auto max_part = [&](int half) __attribute__((always_inline)) {          // half 0 / 1: the eight v_max3 of sB[half]
        if (half == 0) mx = sB[0][0];
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sB[half][r]), sB[half][r + 1]);
        if (half == 1) {
            // v_permlane32_swap: lanes 32..63 of the first register <-> lanes 0..31 of the second.  The second operand is an opaque
            // copy: with the same SSA value for both, hipcc (ROCm 7.2) folded the two results into one and dropped the exchange.
            uint32_t ma = __builtin_bit_cast(uint32_t, mx), mb = ma;
            asm volatile("" : "+v"(mb));
            const auto sw = __builtin_amdgcn_permlane32_swap(ma, mb, false, false);
            uint32_t s0 = sw[0], s1 = sw[1];
            asm volatile("" : "+v"(s0), "+v"(s1));
            mx = fmaxf(__builtin_bit_cast(float, s0), __builtin_bit_cast(float, s1));
            m_new = fmaxf(m_run, mx);
            nm = NSPLIT == 3 ? -m_new : -m_new * sc;     // 3-term: log2 units already
            rs2 = f32x2{0.f, 0.f};
            if constexpr (NSPLIT != 3) {                 // huge logits (rare; decided per query row): subtract first, see attention_kernel
                const bool big = fabsf(nm) > 64.f;
                if (__builtin_amdgcn_ballot_w64(big)) {
                    const float sub = big ? m_new : 0.f;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sB[kb][r] -= sub;
                    nm = big ? 0.f : nm;
                }
            }
        }
    };
    auto exp_part = [&](int i0, int i1) __attribute__((always_inline)) {    // scores i0 .. i1 - 1 of the 32 (even bounds: pairs)
#pragma unroll
        for (int i = i0; i < i1; i += 2) exp_pair<NSPLIT>(sB[i >> 4], i & 15, m_new, sc, nm, rs2);
        asm volatile("" : "+v"(rs2));           // the slice is computed in the group it is written in
#pragma unroll
        for (int i = i0; i < i1; ++i) asm volatile("" : "+v"(sB[i >> 4][i & 15]));
    };
    auto finish_softmax = [&]() __attribute__((always_inline)) {
        rescale = __builtin_amdgcn_ballot_w64(m_new > m_run) != 0;
        if (rescale) {
            alpha = __builtin_amdgcn_exp2f(NSPLIT == 3 ? m_run - m_new : (m_run - m_new) * sc);
            l_run *= alpha;
        }
        m_run = m_new;
        l_run += rs2[0] + rs2[1];
    };

    PPFrags<T, NSPLIT> fr;
    constexpr int NS = VTQ_SW_DIST + 1, DIST = VTQ_SW_DIST;
    auto issue_k = [&](auto gc, const uint32_t (&kaddr)[4]) __attribute__((always_inline)) {       // QK^T group g = kb * 4 + tt
        constexpr int g = decltype(gc)::value, sl = g % NS, kb = g >> 2, tt = g & 3, off = kb * 4096;
        PP_DS_B128(fr.ka[sl], kaddr[tt], off);
        if constexpr (NSPLIT == 3) PP_DS_B128(fr.kl[sl], kaddr[tt], off + TB);
    };
    auto issue_v = [&](auto gc, const uint32_t (&vaddr)[2]) __attribute__((always_inline)) {       // PV group g = step * 2 + d
        constexpr int g = decltype(gc)::value, sl = (g + 8) % NS, step = g >> 1, d = g & 1, off = (step >> 1) * 4096 + (step & 1) * 2048;
        PP_DS_TR(fr.va0[sl], vaddr[d], off);
        PP_DS_TR(fr.va1[sl], vaddr[d], off + 1024);
        if constexpr (NSPLIT == 3) {
            PP_DS_TR(fr.vl0[sl], vaddr[d], off + TB);
            PP_DS_TR(fr.vl1[sl], vaddr[d], off + TB + 1024);
        }
    };

    auto issue_g = [&](auto Gc, const uint32_t (&kaddr)[4], const uint32_t (&vaddr)[2]) __attribute__((always_inline)) {   // group G of a tile
        constexpr int G = decltype(Gc)::value;
        if constexpr (VTQ_SW_HALFREADS && (G & 1)) return;
        if constexpr (G < 8) issue_k(std::integral_constant<int, G>{}, kaddr);
        else if constexpr (G < 16) issue_v(std::integral_constant<int, G - 8>{}, vaddr);
    };
    auto ahead_of = [](int G, int end) constexpr { return sw_ahead<NSPLIT>(G, end); };

    // Masked keys (see attention_kernel zero_masked_v): tile g of this workgroup's stream is the last tile of its block iff g % nt == nt - 1;
    // every thread zeroes its own V piece of such a tile (row d_row of the tile, one 16-byte piece per plane, at byte tid * 16 of the slot's V
    // planes) when that row is a masked key -- behind the wait that lands the piece, in front of the barrier that publishes the tile.
    const int tail_valid = S - (nt - 1) * KT;
    auto zero_masked_v = [&](int g) __attribute__((always_inline)) {
        if (d_row >= tail_valid) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                const u32x4 z = {0u, 0u, 0u, 0u};
                asm volatile("ds_write_b128 %0, %1" ::"v"(lds0 + (uint32_t)((g & 3) * STAGE + (NPL + pl) * TB + tid * 16)), "v"(z) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // ---- pipeline prologue: S(0) = QK^T of tile 0 and its softmax, no overlap ---------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int ib2 = 2 % nt;                                        // in-block index of tile tau + 2 (tau = 0 here), kept by increments
#ifndef VTQ_ATTN_NO_VMASK
    for (int g = 0; g < 3 && g < NT; ++g)
        if (g % nt == nt - 1) zero_masked_v(g);             // tiles 0 .. 2 were staged above and have landed
#endif
    pp_barrier();
    {
        uint32_t kaddr[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) kaddr[tt] = k_lane[tt];
        const uint32_t vnone[2] = {0, 0};
        static_for<0, DIST>([&](auto gc) __attribute__((always_inline)) { issue_g(gc, kaddr, vnone); });
        static_for<0, 8>([&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value, sl = g % NS, kb = g >> 2, tt = g & 3;
            if constexpr (g + DIST < 8) issue_g(std::integral_constant<int, g + DIST>{}, kaddr, vnone);
            constexpr int ahead = ahead_of(g, 8);
            if constexpr (NSPLIT == 3) asm volatile("s_waitcnt lgkmcnt(%c2)" : "+v"(fr.ka[sl]), "+v"(fr.kl[sl]) : "i"(ahead));
            else asm volatile("s_waitcnt lgkmcnt(%c1)" : "+v"(fr.ka[sl]) : "i"(ahead));
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const tx8 kf = __builtin_bit_cast(tx8, fr.ka[sl]);
            sB[kb] = SW_MFMA<T>(kf, qf[0][tt], tt == 0 ? zero16 : sB[kb]);
            if constexpr (NSPLIT == 3) {
                const tx8 kl = __builtin_bit_cast(tx8, fr.kl[sl]);
                sB[kb] = SW_MFMA<T>(kf, qf[1][tt], sB[kb]);
                sB[kb] = SW_MFMA<T>(kl, qf[0][tt], sB[kb]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        mask_tail(0);
        max_part(0);
        max_part(1);
        exp_part(0, 32);
        finish_softmax();
#pragma unroll
        for (int d = 0; d < 2; ++d) sA[d] = sB[d];
        if (nt == 1 && b0 + bstep < b1) {                      // single-tile blocks: the next QK^T already belongs to the next block
            load_q_async(b0 + bstep, qf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pin_q(qf);
            if constexpr (NSPLIT == 3) { if (!q_log2) prescale_q<T>(qf, sc); }
        }
    }
    VTQ_AT_SPAN(dg_pro);

    // Half a step of the P split: 4 probabilities of sA -> 2 hi words + 2 lo words, written IN PLACE over the floats they came from, so
    // that after both halves the 8 registers of a step hold [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] = the two MFMA fragments of the step
    // (no second set of 32 fragment registers).  Half 0 may only overwrite the floats it consumed (0..3): its hi words go to 0, 1 and
    // its lo words wait in lw_keep; half 1 consumes floats 4..7 first and then fills 2, 3 (hi), 4, 5 (the kept lo) and 6, 7 (lo).
    auto split_half = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, step = g >> 1, half = g & 1, kb = step >> 1, base = 8 * (step & 1);
        float p4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) p4[j] = sA[kb][base + 4 * half + j];
        uint32_t hw[2], lw[2];
        if constexpr (NSPLIT == 1) {
            tx4 hv = {(T)p4[0], (T)p4[1], (T)p4[2], (T)p4[3]};
            typedef uint32_t u2 __attribute__((ext_vector_type(2)));
            const u2 w = __builtin_bit_cast(u2, hv);
            hw[0] = w[0]; hw[1] = w[1]; lw[0] = 0; lw[1] = 0;
        } else {
            split_p4<T>(p4, hw, lw);
        }
        // formed HERE (opaque), not sunk to the PV group that consumes them
        asm volatile("" : "+v"(hw[0]), "+v"(hw[1]));
        if constexpr (NSPLIT == 3) asm volatile("" : "+v"(lw[0]), "+v"(lw[1]));
        auto put = [&](int i, uint32_t w) __attribute__((always_inline)) { sA[kb][base + i] = __builtin_bit_cast(float, w); };
        if constexpr (half == 0) {
            put(0, hw[0]); put(1, hw[1]);
            lw_keep[0] = lw[0]; lw_keep[1] = lw[1];
        } else {
            put(2, hw[0]); put(3, hw[1]);
            if constexpr (NSPLIT == 3) { put(4, lw_keep[0]); put(5, lw_keep[1]); put(6, lw[0]); put(7, lw[1]); }
        }
    };
    // the fragments of step (kb * 2 + s2): registers 8 s2 .. + 3 (hi) and + 4 .. + 7 (lo) of sA[kb]
    auto p_frag = [&](int step, int lo) __attribute__((always_inline)) -> tx8 {
        const f32x16& v = sA[step >> 1];
        const int o = 8 * (step & 1) + 4 * lo;
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 w = {v[o], v[o + 1], v[o + 2], v[o + 3]};
        return __builtin_bit_cast(tx8, w);
    };
    // Issue priority of this wave against its SIMD partner (waves w and w + 4 share a SIMD; the older one, w, wins arbitration by age): the stamps show
    // waves 4 - 7 taking 2 462 + 3 312 cycles for the two phases of a tile against 1 674 + 2 837 for waves 0 - 3, which then wait 1 300 cycles longer at the
    // tile's barrier (profiles/r05_attention_prio.txt).  hi(x): this wave is favoured in slot x (phase or fragment group).
    auto prio = [&](int slot) __attribute__((always_inline)) {
#if VTQ_SW_PRIO == 1 || VTQ_SW_PRIO == 2
        if (((wave >> 2) ^ slot) & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#elif VTQ_SW_PRIO == 3
        if (wave >> 2) __builtin_amdgcn_s_setprio(2);
#endif
    };
    auto iteration = [&](auto more_c, int tau) __attribute__((always_inline)) {
        constexpr bool more = decltype(more_c)::value;         // is there a tile tau + 1 (its QK^T and softmax run in this iteration)
        const int tb_next = (ct + 1 == nt) ? 0 : ct + 1;       // its index in its block
        uint32_t kaddr[4], vaddr[2];
        {
            const uint32_t kslot = (uint32_t)((tau + 1) & 3) * STAGE, vslot = (uint32_t)(tau & 3) * STAGE;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) kaddr[tt] = k_lane[tt] + kslot;
#pragma unroll
            for (int d = 0; d < 2; ++d) vaddr[d] = v_lane[d] + vslot;
        }
        // waves whose 32 rows lie behind the sequence in the last query block of a (sequence, head) only load and synchronise
        const int qb_next = tb_next == 0 ? ((cqb + 1 == nqb) ? 0 : cqb + 1) : cqb;                  // query block of tile tau + 1
        const bool active1 = (qb_next * 256 + wave * 32 < Sq);
        const bool active2 = (cqb * 256 + wave * 32 < Sq);
        // Q of the block after the one tile tau + 1 belongs to, when tile tau + 1 is that block's last: loaded into spare registers at
        // the top of the iteration whose phase 1 still needs the current Q, installed behind the end-of-iteration wait.
        // ---------------- phase 1: QK^T(tau + 1) -> sB  ||  split of P(tau) = sA, rescale of O ------------------------------
#if VTQ_SW_PRIO == 1 || VTQ_SW_PRIO == 3
        prio(1);                                   // phase 1: waves 4 - 7 favoured (1); phase 2: waves 0 - 3 (below)
#endif
        // Output of the block that ended with the previous iteration's PV.  Written HERE, at the top: phase 1 does not touch O, the stores are
        // older than this iteration's LDS-DMA and have both phases as cover before the counted wait at the end (all CUs reach their seams
        // together: 16 MB of stores in one burst, which one phase did not cover).
        const bool had_pending = wr_pending;       // the successor block starts from zero: nothing to rescale
#ifdef VTQ_ATTN_DIAG
        dg_it0 = dg_t;
#endif
#if VTQ_SW_EARLY_WRITE
        if (wr_pending) { write_block(l_fin, wr_b, wr_qb); wr_pending = false; }
#endif
#ifdef VTQ_ATTN_DIAG
        if (had_pending) { VTQ_AT_SPAN(dg_wb); }
#endif
        if (rescale && !had_pending) {
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
        }
        if (!active1 && !active2) {
        } else
        if constexpr (more) {
            static_for<0, DIST>([&](auto gc) __attribute__((always_inline)) { issue_g(gc, kaddr, vaddr); });
            static_for<0, 8>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value, sl = g % NS, kb = g >> 2, tt = g & 3;
#if VTQ_SW_PRIO == 2
                prio(g);
#endif
                issue_g(std::integral_constant<int, g + DIST>{}, kaddr, vaddr);
                constexpr int ahead = ahead_of(g, 16);
                if constexpr (NSPLIT == 3) asm volatile("s_waitcnt lgkmcnt(%c2)" : "+v"(fr.ka[sl]), "+v"(fr.kl[sl]) : "i"(ahead));
                else asm volatile("s_waitcnt lgkmcnt(%c1)" : "+v"(fr.ka[sl]) : "i"(ahead));
                __builtin_amdgcn_sched_barrier(0);
                const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const tx8 kf = __builtin_bit_cast(tx8, fr.ka[sl]);
                sB[kb] = SW_MFMA<T>(kf, qf[0][tt], tt == 0 ? zero16 : sB[kb]);
                if constexpr (NSPLIT == 3) {
                    const tx8 kl = __builtin_bit_cast(tx8, fr.kl[sl]);
                    sB[kb] = SW_MFMA<T>(kf, qf[1][tt], sB[kb]);
                    sB[kb] = SW_MFMA<T>(kl, qf[0][tt], sB[kb]);
                }
                if constexpr (!VTQ_SW_NOFILL) split_half(gc);
                if constexpr (NSPLIT == 3) { SW_MFMA_VALU(4); SW_MFMA_VALU(4); SW_MFMA_VALU(4); }
                else { SW_MFMA_VALU(2); }
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
            // last tile of the stream: no QK^T left; split without cover, first PV fragments
            static_for<0, 8>([&](auto gc) __attribute__((always_inline)) { split_half(gc); });
            static_for<8, 8 + DIST>([&](auto gc) __attribute__((always_inline)) { issue_g(gc, kaddr, vaddr); });
        }
        VTQ_AT_SPAN(dg_p1);
#if !VTQ_SW_EARLY_WRITE
        if (wr_pending) { write_block(l_fin, wr_b, wr_qb); wr_pending = false; }
#endif
        // at a seam the finished block's row sums are set aside and the statistics restart before the next tile's softmax
        const bool seam = (ct + 1 == nt);
        if (seam) { l_fin = l_run; m_run = -1e30f; l_run = 0.f; }
        if constexpr (more) mask_tail(tb_next);
        // ---------------- phase 2: PV(tau) into O  ||  softmax of sB (tile tau + 1) -------------------------------------------
#if VTQ_SW_PRIO == 1
        prio(0);
#endif
        bool sent = false, q_loaded = false;
        auto pv_group = [&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value, sl = (g + 8) % NS, step = g >> 1, d = g & 1;
#if VTQ_SW_PRIO == 2
            prio(g);
#endif
            issue_g(std::integral_constant<int, g + 8 + DIST>{}, kaddr, vaddr);          // nothing beyond group 15
            constexpr int ahead = ahead_of(g + 8, 16);
            if constexpr (NSPLIT == 3)
                asm volatile("s_waitcnt lgkmcnt(%c4)" : "+v"(fr.va0[sl]), "+v"(fr.va1[sl]), "+v"(fr.vl0[sl]), "+v"(fr.vl1[sl]) : "i"(ahead));
            else
                asm volatile("s_waitcnt lgkmcnt(%c2)" : "+v"(fr.va0[sl]), "+v"(fr.va1[sl]) : "i"(ahead));
            __builtin_amdgcn_sched_barrier(0);
            const tx8 vf = __builtin_bit_cast(tx8, u32x4{fr.va0[sl][0], fr.va0[sl][1], fr.va1[sl][0], fr.va1[sl][1]});
            const tx8 ph = p_frag(step, 0);
            o_acc[d] = SW_MFMA<T>(vf, ph, o_acc[d]);
            if constexpr (NSPLIT == 3) {
                const tx8 vl = __builtin_bit_cast(tx8, u32x4{fr.vl0[sl][0], fr.vl0[sl][1], fr.vl1[sl][0], fr.vl1[sl][1]});
                const tx8 pl = p_frag(step, 1);
                o_acc[d] = SW_MFMA<T>(vf, pl, o_acc[d]);
                o_acc[d] = SW_MFMA<T>(vl, ph, o_acc[d]);
            }
            if constexpr (more && !VTQ_SW_NOFILL) {
                if constexpr (g == 0) max_part(0);
                else if constexpr (g == 1) max_part(1);
                else if constexpr (g == 2) exp_part(0, 6);
                else if constexpr (g == 3) exp_part(6, 12);
                else if constexpr (g == 4) exp_part(12, 18);
                else if constexpr (g == 5) exp_part(18, 22);
                else if constexpr (g == 6) exp_part(22, 28);
                else exp_part(28, 32);
                if constexpr (NSPLIT == 3) { SW_MFMA_VALU(5); SW_MFMA_VALU(5); SW_MFMA_VALU(5); }
                else { SW_MFMA_VALU(6); }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto phase2 = [&]() __attribute__((always_inline)) { static_for<0, 8>(pv_group); };
        // Q of the next block: its registers are free once the QK^T of this block's last tile has run (phase 1 above); phase 2 covers
        // the loads, the end-of-iteration wait completes them.
        // Q of the block after the one tile tau + 1 belongs to: its registers are free once the QK^T of that block's last tile has run
        // (phase 1 above); loaded in place, hidden from hipcc's vmcnt bookkeeping, complete behind the counted wait at the end
        if constexpr (more) {
            const int bq = (tb_next == 0 ? cb + bstep : cb) + bstep;
            if (tb_next == nt - 1 && bq < b1 && !VTQ_SW_NOQ) { load_q_async(bq, qf); q_loaded = true; }
        }
        sent = issue_tile();                    // tile tau + 3: younger than the Q loads and stores, so vmcnt(NI) below covers them
        if (active1 || active2) phase2();
        // L2 prefetch of the Q rows that will be loaded two iterations from now: one dword per 128-byte row segment (lane = row, half-wave =
        // plane), into a register nothing reads.  All CUs reach their seams together, so the Q loads of a seam are a 16 MB burst that HBM
        // serves in ~3.5 us -- longer than the phase that covers them; pulled into L2 ahead of time they return at L2 latency.  Issued
        // AFTER this iteration's LDS-DMA, behind phase 2 (younger: this iteration's counted wait leaves it in flight, the next one's covers it).
        bool q_pf_sent = false;
#if VTQ_SW_QPF
        if constexpr (more) {
            const int bq = (tb_next == 0 ? cb + bstep : cb) + bstep;
            if (tb_next == nt - 3 && bq < b1) {
                int qb;
                const int64_t base = block_base(bq, qb);
                int qr = qb * 256 + wave * 32 + c;
                qr = qr < Sq ? qr : Sq - 1;
                const T* ptr = qkv + (NPL == 2 ? hh : 0) * plane + base + (int64_t)qr * ld;
                asm volatile("global_load_dword %0, %1, off" : "+v"(q_pf) : "v"(ptr) : "memory");
                q_pf_sent = true;
            }
        }
#endif
        if constexpr (more) finish_softmax(); else rescale = false;
        VTQ_AT_SPAN(dg_p2);
        if (seam) {
            wr_pending = true; wr_b = cb; wr_qb = cqb;
            ct = 0;
            if ((cb += bstep) < b1) block_base(cb, cqb);
        } else {
            ++ct;
        }
#pragma unroll
        for (int d = 0; d < 2; ++d) sA[d] = sB[d];
        VTQ_AT_SPAN(dg_rest);
        if (sent && q_pf_sent) asm volatile("s_waitcnt vmcnt(%c0)" :: "i"(NI + 1) : "memory");
        else if (sent) asm volatile("s_waitcnt vmcnt(%c0)" :: "i"(NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(q_pf));                         // the prefetch's destination stays reserved (it may still be in flight)
        // everything older than tile tau + 3 has landed: tile tau + 2 (its V is read two iterations from now) gets its masked rows zeroed
#ifndef VTQ_ATTN_NO_VMASK
        if (tau + 2 < NT && ib2 == nt - 1) zero_masked_v(tau + 2);
        ib2 = (ib2 + 1 == nt) ? 0 : ib2 + 1;                   // in-block index of tile tau + 3, for the next iteration
#endif
        pin_q(qf);
        if constexpr (NSPLIT == 3) { if (q_loaded && !q_log2) prescale_q<T>(qf, sc); }
        pp_barrier();
        VTQ_AT_SPAN(dg_bar);
#ifdef VTQ_ATTN_DIAG
        { const int kd = had_pending ? 1 : (q_loaded ? 2 : 0); dg_kind[kd] += dg_t - dg_it0; dg_nkind[kd] += 1; }
#endif
    };
    for (int tau = 0; tau < NT - 1; ++tau) iteration(std::true_type{}, tau);
    iteration(std::false_type{}, NT - 1);
#ifdef VTQ_ATTN_DIAG
    dg_tail0 = dg_t;
#endif
    write_block(l_fin, wr_b, wr_qb);             // the last block of the list
    if (out8_scale > 0.f) fp8_report(obs, amax8, out8_scale);
#ifdef VTQ_ATTN_DIAG
    if (diag) {
        unsigned long long dg_k1, dg_r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_k1), "=s"(dg_r1) :: "memory");
        if (lane == 0) {
            unsigned long long* dq = diag + ((size_t)blockIdx.x * 8 + wave) * 16;
            dq[0] = dg_k1 - dg_k0; dq[1] = dg_r1 - dg_r0; dq[2] = dg_p1; dq[3] = dg_p2; dq[4] = dg_rest; dq[5] = dg_bar; dq[6] = 1; dq[7] = NT; dq[8] = 0;
            dq[9] = dg_pro;
            dq[10] = dg_kind[0]; dq[11] = dg_kind[1]; dq[12] = dg_kind[2]; dq[13] = dg_nkind[0] | (dg_nkind[1] << 20) | (dg_nkind[2] << 40); dq[14] = dg_wb; dq[15] = dg_k1 - dg_tail0;
        }
    }
#endif
}

}  // namespace

template <typename T, int NSPLIT>
hipError_t launch_attention_t(const void* qkv, int64_t plane, void* out, int64_t o_plane, int nseq, int S, int S_pad, int H, hipStream_t s,
                              float out8_scale, Fp8Obs obs, bool q_log2, int q0 = 0) {
#ifdef VTQ_ATTN_DIAG
    static const int lds_pad = VTQ_MEASURE_ENV("VTQ_ATTN_LDS_PAD") ? atoi(VTQ_MEASURE_ENV("VTQ_ATTN_LDS_PAD")) : 0;   // occupancy experiments
    const int LDS = 2 * 2 * 64 * 128 * (NSPLIT == 1 ? 1 : 2) + lds_pad;
#else
    constexpr int LDS = 2 * 2 * 64 * 128 * (NSPLIT == 1 ? 1 : 2);
#endif
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
    const dim3 grid(((S_pad - q0 + 127) / 128) * (H / 64) * nseq), blk(256);
    hipLaunchKernelGGL((attention_kernel<T, NSPLIT>), grid, blk, LDS, s, (const T*)qkv, plane, (T*)out, o_plane, S, S_pad, H, out8_scale, obs,
                       gemm_diag_buffer(), q_log2 ? 1 : 0, q0, S_pad);
    return hipGetLastError();
}

static std::atomic<int> g_attn_map{0};          // measurement hook: 0 = XCD-strided block walk (default), 1 = the round-3 contiguous / paired walk
void attention_set_map(int m) { g_attn_map.store(m, std::memory_order_relaxed); }

// Pipelined kernel: persistent grid of at most one workgroup per CU (160 KB of LDS in the 3-term formats), `per` consecutive blocks each.
template <typename T, int NSPLIT>
hipError_t launch_attention_sw_t(const void* qkv, int64_t plane, void* out, int64_t o_plane, int nseq, int S, int S_pad, int H, hipStream_t s,
                                 float out8_scale, Fp8Obs obs, int cus, bool q_log2, int Sq = 0) {
    if (Sq <= 0) Sq = S_pad;
    constexpr int LDS = 4 * 2 * 64 * 128 * (NSPLIT == 1 ? 1 : 2) + 8 * 4096;     // K/V ring + output staging
    static std::mutex mu;
    static bool configured[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!configured[dev]) {
            e = hipFuncSetAttribute((const void*)attention_sw_kernel<T, NSPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) return e;
            configured[dev] = true;
        }
    }
    const int nblk = ((Sq + 255) / 256) * (H / 64) * nseq;
    int per = (nblk + cus - 1) / cus;
    dim3 grid((nblk + per - 1) / per), blk(512);
    if (g_attn_map.load(std::memory_order_relaxed) != 1 && cus >= 8 && nblk >= 8) {
        // XCD-strided walk (see the kernel): a grid that is a multiple of 8, at most one workgroup per CU; the rounds of the busiest XCD are
        // ceil(its blocks / its workgroups), so fewer workgroups than CUs are used when that does not add a round (fewer half-empty last rounds)
        const int nqb = (Sq + 255) / 256, items = nblk / nqb;
        const int xmax = ((items + 7) / 8) * nqb;                      // blocks of the busiest XCD
        int nx = cus / 8 < xmax ? cus / 8 : xmax;
        const int rounds = (xmax + nx - 1) / nx;
        nx = (xmax + rounds - 1) / rounds;
        grid = dim3(8 * nx);
        per = 0;
    }
    hipLaunchKernelGGL((attention_sw_kernel<T, NSPLIT>), grid, blk, LDS, s, (const T*)qkv, plane, (T*)out, o_plane, S, S_pad, H, nblk, per, out8_scale, obs,
                       gemm_diag_buffer(), q_log2 ? 1 : 0, Sq);
    return hipGetLastError();
}

static int g_attn_variant = -1;                 // -1: the rule below (or VTQ_ATTN_VARIANT), 0: 4-wave kernel, 1: pipelined kernel
void attention_set_variant(int v) { g_attn_variant = v; }

static std::atomic<int> g_attn_cus{0};          // measurement hook (a launch on a CU-masked stream): 0 = the device's CU count
void attention_set_cus(int cus) { g_attn_cus.store(cus, std::memory_order_relaxed); }

int device_cus(int* cus) {
    static std::mutex mu;
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1;
    std::lock_guard<std::mutex> lk(mu);
    if (!cached[dev] && hipDeviceGetAttribute(&cached[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 1;
    *cus = cached[dev];
    return 0;
}

// Which kernel: the pipelined one pays in the 3-term formats (profiles/r03_attention_anatomy.txt: -7 .. -17 % at the encoder's shapes; the
// single-plane formats are equal or slower), and only when its 256-row blocks keep the CUs busy: at least 74 % (round 6, sustained timing
// with the XCD-strided walk: 8 sequences x 501 = 75 % of one round is 11 % faster than the 4-wave kernel, 16 and 24 sequences tie; 85 % before) of the block slots of the
// persistent grid filled, and no more than 15 % more padded query rows than the 128-row blocks of the 4-wave kernel (waves without rows
// skip their arithmetic, so padding costs the skeleton only: S = 1025 pads 11 % more rows and is 6 % faster; S = 257 pads 33 % more
// and is 34 % slower).
// SPLIT (2), a measurement form since round 6: the pipelined kernel on the full 256-row blocks and ONE launch of the 4-wave kernel on the rows behind them
// (sequences a few rows longer than a multiple of 256: 521 = the reference's default topology, 1025 = configs[3]).  It was the rule's choice in round 5; with
// the XCD-strided block walk a last block with one active wave costs 0.7 of a block and the rule's own two cases are as fast or faster at every such shape
// (profiles/r06_attention_loop.txt section 9: S = 1025 -2 .. -5 % against the split form, S = 521 / 545 / 769 within 1 %).  vtq_debug_attention_variant(2) still
// runs it.  Same arithmetic per query row in every form: outputs are bit-identical (tests/test_gpu_kernels.py).
int attention_rule(int nseq, int S_pad, int H, int terms, int cus) {
    if (terms != 3 || cus < 1 || nseq < 1 || S_pad < 1 || H % 64) return 0;
    auto fills = [&](int rows) {
        const int nblk = ((rows + 255) / 256) * (H / 64) * nseq;
        const int per = (nblk + cus - 1) / cus;
        return (double)nblk >= 0.74 * (double)per * cus;
    };
    const bool rows_ok = ((S_pad + 255) / 256) * 256 * 100 <= ((S_pad + 127) / 128) * 128 * 115;
    return (fills(S_pad) && rows_ok) ? 1 : 0;
}

static int pick_variant(int nseq, int S_pad, int H, int terms, int cus) {
    static const int env = VTQ_MEASURE_ENV("VTQ_ATTN_VARIANT") ? atoi(VTQ_MEASURE_ENV("VTQ_ATTN_VARIANT")) : -1;   // -DVTQ_MEASURE builds only
    const int forced = g_attn_variant >= 0 ? g_attn_variant : env;
    if (forced == 2) return (S_pad >= 256 && S_pad % 256) ? 2 : 1;      // forced split needs a full block and a rest
    if (forced >= 0) return forced == 1 ? 1 : 0;
    return attention_rule(nseq, S_pad, H, terms, cus);
}

template <typename T, int NSPLIT>
static hipError_t launch_attention_v(int variant, const void* qkv, int64_t plane, void* out, int64_t o_plane, int nseq, int S, int S_pad, int H,
                                     hipStream_t s, float out8_scale, Fp8Obs obs, int cus, bool q_log2) {
    if (variant == 0) return launch_attention_t<T, NSPLIT>(qkv, plane, out, o_plane, nseq, S, S_pad, H, s, out8_scale, obs, q_log2);
    if (variant == 1) return launch_attention_sw_t<T, NSPLIT>(qkv, plane, out, o_plane, nseq, S, S_pad, H, s, out8_scale, obs, cus, q_log2);
    const int full = (S_pad / 256) * 256;
    const hipError_t e = launch_attention_sw_t<T, NSPLIT>(qkv, plane, out, o_plane, nseq, S, S_pad, H, s, out8_scale, obs, cus, q_log2, full);
    if (e != hipSuccess) return e;
    return launch_attention_t<T, NSPLIT>(qkv, plane, out, o_plane, nseq, S, S_pad, H, s, out8_scale, obs, q_log2, full);
}

hipError_t launch_attention(const void* qkv, int64_t plane, void* out, int64_t o_plane, int nseq, int S, int S_pad, int H,
                            Num num, hipStream_t s, float out8_scale, Fp8Obs obs, bool q_log2) {
    if (H % 64 || S < 1 || S > S_pad || S <= S_pad - 64 || (num.terms != 1 && num.terms != 3) || num.f16 > 1) return hipErrorInvalidValue;
    if (q_log2 && num.terms != 3) return hipErrorInvalidValue;
    int cus = 0;
    if (device_cus(&cus) || cus < 1) return hipErrorInvalidDevice;
    if (const int o = g_attn_cus.load(std::memory_order_relaxed); o > 0) cus = o;
    const int v = pick_variant(nseq, S_pad, H, num.terms, cus);
    if (!num.f16) {
        if (num.terms == 1) return launch_attention_v<bf16, 1>(v, qkv, plane, out, o_plane, nseq, S, S_pad, H, s, out8_scale, obs, cus, q_log2);
        return launch_attention_v<bf16, 3>(v, qkv, plane, out, o_plane, nseq, S, S_pad, H, s, out8_scale, obs, cus, q_log2);
    }
    if (num.terms == 1) return launch_attention_v<f16, 1>(v, qkv, plane, out, o_plane, nseq, S, S_pad, H, s, out8_scale, obs, cus, q_log2);
    return launch_attention_v<f16, 3>(v, qkv, plane, out, o_plane, nseq, S, S_pad, H, s, out8_scale, obs, cus, q_log2);
}

}  // namespace vtq
