// Device helpers shared by the attention kernels (attention.hip, attention_w4.hip): the probability split, the packed exponent step, the Q pre-scale,
// asm LDS fragment reads hidden from hipcc's waitcnt bookkeeping, the rolling fragment buffers of the pipelined forms.
#pragma once
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

}  // namespace
}  // namespace vtq
