// Pipelined attention, ONE wave per SIMD: 4 waves x 64 query rows (two 32-row blocks per wave), 3-term operand formats, hand-placed stream.
//
// Same algorithm, LDS tile layout, ring, seams and per-query-row arithmetic (in the same order) as attention_sw_kernel (attention.hip) -- outputs are
// bit-identical -- but the two row blocks that share a SIMD are two instruction streams of ONE wave: every K / V fragment read feeds 6 MFMAs, and
// the vector work sits between the wave's own MFMAs in program order (<= 5 issues per MFMA gap, cdna_hip_programming.md "4-wave, one-wave-per-SIMD,
// persistent structure") instead of being arbitrated between two waves on the SIMD's one vector issue port.  The wave owns the SIMD's whole
// 512-entry register file: O (64) and Q (64) live in the AGPR half, operands of asm MFMA statements ("a" constraints; never copied in the loop),
// scores / probabilities, fragments and everything else in arch VGPRs.  hipcc schedules nothing here: every MFMA statement is followed by its
// filler chunk and a scheduling fence.  It also pads no hazard around an asm MFMA: results are read by the next MFMA of the same chain (legal
// back to back) or by vector code at least six MFMAs later; the few places where vector code writes an MFMA operand late (Q re-scale at a seam
// in the raw-Q test path, masked scores of a last tile) end with an explicit s_nop.
#include <atomic>
#include <cstdlib>
#include <mutex>

#include "dev_common.h"
#include "kernels.h"
#include "attention_common.h"

namespace vtq {
namespace {

template <typename T> struct W4Mfma;
template <> struct W4Mfma<f16> {
    template <typename X8> static __device__ __forceinline__ void qk0(f32x16& acc, X8 a, const X8& b) { if (W4_NOMFMA) asm volatile("" : "=&v"(acc) : "v"(a), "a"(b)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b)); }
    template <typename X8> static __device__ __forceinline__ void qk(f32x16& acc, X8 a, const X8& b) { if (W4_NOMFMA) asm volatile("" : "+v"(acc) : "v"(a), "a"(b)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b)); }
    template <typename X8> static __device__ __forceinline__ void pv(f32x16& acc, X8 a, X8 b) { if (W4_NOMFMA) asm volatile("" : "+a"(acc) : "v"(a), "v"(b)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); }
};
template <> struct W4Mfma<bf16> {
    template <typename X8> static __device__ __forceinline__ void qk0(f32x16& acc, X8 a, const X8& b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b)); }
    template <typename X8> static __device__ __forceinline__ void qk(f32x16& acc, X8 a, const X8& b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b)); }
    template <typename X8> static __device__ __forceinline__ void pv(f32x16& acc, X8 a, X8 b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); }
};

#define W4_FENCE __builtin_amdgcn_sched_barrier(0)
#ifndef W4_NOFILL
#define W4_NOFILL 0
#endif
#ifndef W4_NOMFMA
#define W4_NOMFMA 0
#endif

// one pair of probabilities -> hi word, lo word (split_p4's arithmetic for one pair: same bits)
template <typename T>
__device__ __forceinline__ void split_p2(float p0, float p1, uint32_t& hw, uint32_t& lw) {
    if constexpr (std::is_same<T, f16>::value) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        const auto hp = __builtin_amdgcn_cvt_pkrtz(p0, p1);
        const h2 hh = __builtin_bit_cast(h2, hp);
        hw = __builtin_bit_cast(uint32_t, hp);
        lw = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(p0 - (float)hh[0], p1 - (float)hh[1]));
    } else {
        uint32_t h[2], l[2];
        const float p[2] = {p0, p1};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t hb = __builtin_bit_cast(uint32_t, p[e]) & 0xFFFF0000u;
            const float hf = __builtin_bit_cast(float, hb);
            h[e] = hb >> 16;
            l[e] = (uint32_t)__builtin_bit_cast(unsigned short, (bf16)(p[e] - hf));
        }
        hw = h[0] | (h[1] << 16);
        lw = l[0] | (l[1] << 16);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attention_w4_kernel(const T* __restrict__ qkv, int64_t plane, T* __restrict__ out, int64_t o_plane,
                                                           int S, int S_pad, int H, int nblk, int q_log2, int Sq) {
    typedef typename Vec<T>::x8 tx8;
    typedef typename Vec<T>::x4 tx4;
    constexpr int NSPLIT = 3, NPL = 2, RB = 2;
    constexpr int KT = 64, TB = KT * 128, STAGE = TB * NPL * 2, NI = 4 * NPL;       // NI: LDS-DMA pieces of one tile per wave (2 rounds x K, V x planes)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hh = lane >> 5;
    const int nqb = (Sq + 255) / 256, nh = H / 64;
    const int nt = (S + KT - 1) / KT;
    const int ld = 3 * H;
    // XCD-strided block walk (attention_sw_kernel): XCD x owns a contiguous run of (sequence, head) pairs, its workgroups take that run's blocks side by side
    int b0, b1, bstep;
    {
        const int x = blockIdx.x & 7, sl = blockIdx.x >> 3;
        const int nx = ((int)gridDim.x - x + 7) >> 3;
        const int items = nblk / nqb;
        const int i0 = (int)((long long)items * x >> 3), i1 = (int)((long long)items * (x + 1) >> 3);
        b0 = i0 * nqb + sl;
        bstep = nx;
        b1 = i1 * nqb;
    }
    if (b0 >= b1) return;
    const int NT = ((b1 - b0 + bstep - 1) / bstep) * nt;
    const float sc = 0.125f * 1.4426950408889634f;

    auto block_base = [&](int b, int& qb) __attribute__((always_inline)) -> int64_t {
        qb = b % nqb;
        const int head = (b / nqb) % nh, seq = b / (nqb * nh);
        return (int64_t)seq * S_pad * ld + head * 64;
    };
    auto block_out = [&](int b) __attribute__((always_inline)) -> int64_t {
        const int head = (b / nqb) % nh, seq = b / (nqb * nh);
        return (int64_t)seq * S_pad * H + head * 64;
    };
    auto q_row_of = [&](int qb, int rb) __attribute__((always_inline)) -> int {
        const int qr = qb * 256 + wave * 64 + rb * 32 + c;
        return qr < Sq ? qr : Sq - 1;
    };
    auto load_q = [&](int b, tx8 (&qf)[RB][2][4]) __attribute__((always_inline)) {
        int qb;
        const int64_t base = block_base(b, qb);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int qr = q_row_of(qb, rb);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int t = 0; t < 4; ++t) qf[rb][pl][t] = *(const tx8*)(qkv + pl * plane + base + (int64_t)qr * ld + 16 * t + 8 * hh);
        }
    };
    // the same loads straight into the AGPR half, hidden from hipcc's vmcnt bookkeeping (in place); the caller waits with a counted vmcnt
    auto load_q_async = [&](int b, tx8 (&qf)[RB][2][4]) __attribute__((always_inline)) {
        int qb;
        const int64_t base = block_base(b, qb);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int qr = q_row_of(qb, rb);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const T* ptr = qkv + pl * plane + base + (int64_t)qr * ld + 16 * t + 8 * hh;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "+a"(qf[rb][pl][t]) : "v"(ptr) : "memory");
                }
        }
    };
    auto pin_q = [&](tx8 (&qf)[RB][2][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int t = 0; t < 4; ++t) asm volatile("" : "+a"(qf[rb][pl][t]));
    };
    auto prescale = [&](tx8 (&qf)[RB][2][4]) __attribute__((always_inline)) {        // raw-Q entry (unit tests) only: the engine's Q is in log2 units already
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) prescale_q<T>(qf[rb], sc);
        pin_q(qf);
        asm volatile("s_nop 4" ::: "memory");              // accvgpr writes -> MFMA operand reads
    };

    // LDS-DMA: 256 threads cover a 64-row x 128-byte plane of a tile in two rounds (slot = r * 256 + tid: row = slot >> 3, 16-byte chunk = slot & 7)
    uint32_t k_off[2], v_off[2];
    int d_row[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int slot = r * 256 + tid;
        const int row = slot >> 3, s = slot & 7;
        d_row[r] = row;
        k_off[r] = (uint32_t)(row * ld + H + ((s ^ ((row >> 1) & 7)) << 3));
        v_off[r] = (uint32_t)(row * ld + 2 * H + ((s ^ (((row >> 1) & 1) << 2)) << 3));
    }
    int ib = b0, it = 0, itau = 0, iqb;
    int64_t ibase = block_base(ib, iqb);
    // tile tau + 3 of the stream: begun once per iteration (pointers + bookkeeping), its 8 pieces issued one per PV group
    char* t_sb = smem;
    const T* t_base = qkv;
    auto tile_begin = [&]() __attribute__((always_inline)) -> bool {
        if (ib >= b1) return false;
        t_sb = smem + (itau & 3) * STAGE + wave * 1024;
        t_base = qkv + ibase + (int64_t)it * KT * ld;
        ++itau;
        if (++it == nt) {
            it = 0;
            if ((ib += bstep) < b1) ibase = block_base(ib, iqb);
        }
        return true;
    };
    auto tile_piece = [&](int k) __attribute__((always_inline)) {          // piece k = pl * 4 + r * 2 + (K | V)
        const int pl = k >> 2, r = (k >> 1) & 1;
        if (k & 1) glds16(t_base + pl * plane + v_off[r], t_sb + (NPL + pl) * TB + r * 4096);
        else glds16(t_base + pl * plane + k_off[r], t_sb + pl * TB + r * 4096);
    };
    auto issue_tile = [&]() __attribute__((always_inline)) -> bool {
        if (!tile_begin()) return false;
#pragma unroll
        for (int k = 0; k < 8; ++k) tile_piece(k);
        return true;
    };

    tx8 qf[RB][2][4];
    load_q(b0, qf);
    if (!q_log2) prescale(qf);
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

    f32x16 o_acc[RB][2], sA[RB][2], sB[RB][2];   // per row block: O (AGPR half); P of the current tile; scores, then P, of the next one
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[rb][d][r] = 0.f;
            asm volatile("" : "+a"(o_acc[rb][d]));
        }
    uint32_t q_pf = 0;                           // destination of the Q prefetch (never read)
    uint32_t lw_keep[RB][2] = {{0, 0}, {0, 0}}, lw_keep2[RB] = {0, 0};
    float m_run[RB] = {-1e30f, -1e30f}, l_run[RB] = {0.f, 0.f}, l_fin[RB] = {0.f, 0.f};
    float alpha[RB] = {1.f, 1.f};
    bool rescale[RB] = {false, false};

    int cb = b0, ct = 0, cqb;                    // block / tile-in-block of the tile whose PV runs in this iteration
    block_base(cb, cqb);
    bool wr_pending = false;
    int wr_b = 0, wr_qb = 0;

    // Finished block: O / l formed and split once per row block, staged through 4 KB of LDS per (wave, row block) so that every store instruction
    // writes eight whole 128-byte row segments (attention_sw_kernel's write_block).  O comes out of the AGPR half here and goes back as zeros.
    auto write_block = [&](int b, int qb) __attribute__((always_inline)) {
        const int64_t obase = block_out(b);
        const int r_row = lane >> 3, r_chunk = lane & 7;
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");                 // the last PV MFMAs' results -> accvgpr reads
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            char* const o_stage = smem + 4 * STAGE + (wave * RB + rb) * 4096;
            const float l_tot = l_fin[rb] + __shfl_xor(l_fin[rb], 32, 64);
            const float inv = 1.0f / l_tot;
            asm volatile("" : "+a"(o_acc[rb][0]), "+a"(o_acc[rb][1]));
            tx4 lo_keep[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float v4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = o_acc[rb][k >> 2][4 * (k & 3) + e] * inv;
                tx4 hv;
                if constexpr (std::is_same<T, f16>::value) {
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
                    const int qr = qb * 256 + wave * 64 + rb * 32 + row;
                    if (qr < Sq) *(uint4*)(out + pl * o_plane + obase + (int64_t)qr * H + 8 * r_chunk) = w;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int d = 0; d < 2; ++d) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[rb][d][r] = 0.f;
                asm volatile("" : "+a"(o_acc[rb][d]));
            }
        }
        asm volatile("s_nop 4" ::: "memory");                             // accvgpr writes -> MFMA reads of O
    };

    // softmax pieces on sB[rb] (tile index in its block: tb)
    auto mask_tail = [&](int tb) __attribute__((always_inline)) {
        if ((tb + 1) * KT > S) {
            int hq = 4 * hh;
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(hq));               // the last QK^T MFMAs' results -> vector code
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = tb * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + hq;
                        if (key >= S) sB[rb][kb][r] = -INFINITY;
                    }
        }
    };
    float mx[RB] = {0.f, 0.f}, m_new[RB] = {0.f, 0.f}, nm[RB] = {0.f, 0.f};
    f32x2 rs2[RB] = {{0.f, 0.f}, {0.f, 0.f}};
    // max over sB[rb][half], registers [r0, r1): the v_max3 chain of attention_sw_kernel's max_part in two pieces, then its tail
    auto max_piece = [&](int rb, int half, int r0, int r1) __attribute__((always_inline)) {
        if (half == 0 && r0 == 0) mx[rb] = sB[rb][0][0];
#pragma unroll
        for (int r = r0; r < r1; r += 2) mx[rb] = fmaxf(fmaxf(mx[rb], sB[rb][half][r]), sB[rb][half][r + 1]);
        asm volatile("" : "+v"(mx[rb]));
    };
    auto max_tail = [&](int rb) __attribute__((always_inline)) {
        uint32_t ma = __builtin_bit_cast(uint32_t, mx[rb]), mb = ma;
        asm volatile("" : "+v"(mb));
        const auto sw = __builtin_amdgcn_permlane32_swap(ma, mb, false, false);
        uint32_t s0 = sw[0], s1 = sw[1];
        asm volatile("" : "+v"(s0), "+v"(s1));
        mx[rb] = fmaxf(__builtin_bit_cast(float, s0), __builtin_bit_cast(float, s1));
        m_new[rb] = fmaxf(m_run[rb], mx[rb]);
        nm[rb] = -m_new[rb];
        rs2[rb] = f32x2{0.f, 0.f};
        asm volatile("" : "+v"(m_new[rb]), "+v"(rs2[rb]));
    };
    auto exp_one = [&](int rb, int i) __attribute__((always_inline)) {              // scores i, i + 1 of the 32 (i even)
        exp_pair<NSPLIT>(sB[rb][i >> 4], i & 15, m_new[rb], sc, nm[rb], rs2[rb]);
        asm volatile("" : "+v"(rs2[rb]), "+v"(sB[rb][i >> 4][i & 15]), "+v"(sB[rb][i >> 4][(i & 15) + 1]));
    };
    auto exp_part = [&](int rb, int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
        for (int i = i0; i < i1; i += 2) exp_one(rb, i);
    };
    auto finish_softmax = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            rescale[rb] = __builtin_amdgcn_ballot_w64(m_new[rb] > m_run[rb]) != 0;
            if (rescale[rb]) {
                alpha[rb] = __builtin_amdgcn_exp2f(m_run[rb] - m_new[rb]);
                l_run[rb] *= alpha[rb];
            }
            m_run[rb] = m_new[rb];
            l_run[rb] += rs2[rb][0] + rs2[rb][1];
        }
    };

    PPFrags<T, NSPLIT> fr;
    constexpr int NS = VTQ_SW_DIST + 1, DIST = VTQ_SW_DIST;
    auto issue_k = [&fr](auto gc, const uint32_t (&kaddr)[4]) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, sl = g % NS, kb = g >> 2, tt = g & 3, off = kb * 4096;
        PP_DS_B128(fr.ka[sl], kaddr[tt], off);
        PP_DS_B128(fr.kl[sl], kaddr[tt], off + TB);
    };
    auto issue_v = [&fr](auto gc, const uint32_t (&vaddr)[2]) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, sl = (g + 8) % NS, step = g >> 1, d = g & 1, off = (step >> 1) * 4096 + (step & 1) * 2048;
        PP_DS_TR(fr.va0[sl], vaddr[d], off);
        PP_DS_TR(fr.va1[sl], vaddr[d], off + 1024);
        PP_DS_TR(fr.vl0[sl], vaddr[d], off + TB);
        PP_DS_TR(fr.vl1[sl], vaddr[d], off + TB + 1024);
    };
    auto issue_g = [&](auto Gc, const uint32_t (&kaddr)[4], const uint32_t (&vaddr)[2]) __attribute__((always_inline)) {
        constexpr int G = decltype(Gc)::value;
        if constexpr (G < 8) issue_k(std::integral_constant<int, G>{}, kaddr);
        else if constexpr (G < 16) issue_v(std::integral_constant<int, G - 8>{}, vaddr);
    };
    auto ahead_of = [](int G, int end) constexpr { return sw_ahead<NSPLIT>(G, end); };

    // masked keys' V rows of a block's last tile become zeros in LDS (attention_kernel zero_masked_v): every thread its own pieces
    const int tail_valid = S - (nt - 1) * KT;
    auto zero_masked_v = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (d_row[r] >= tail_valid) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    asm volatile("ds_write_b128 %0, %1" ::"v"(lds0 + (uint32_t)((g & 3) * STAGE + (NPL + pl) * TB + r * 4096 + tid * 16)), "v"(z) : "memory");
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // half a step of the P split of row block rb, pair j (floats 4 half + 2 j, + 1 of the step), in place (attention_sw_kernel split_half):
    // after both halves the 8 registers of a step hold [hi01 hi23 hi45 hi67 | lo01 lo23 lo45 lo67]
    auto split_pair = [&](auto gc, int rb, int j) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, step = g >> 1, half = g & 1, kb = step >> 1, base = 8 * (step & 1);
        uint32_t hw, lw;
        split_p2<T>(sA[rb][kb][base + 4 * half + 2 * j], sA[rb][kb][base + 4 * half + 2 * j + 1], hw, lw);
        asm volatile("" : "+v"(hw), "+v"(lw));
        auto put = [&](int i, uint32_t w) __attribute__((always_inline)) { sA[rb][kb][base + i] = __builtin_bit_cast(float, w); };
        if constexpr (half == 0) {
            put(j, hw);                                   // floats 0, 1 (j = 0) resp. 2, 3 (j = 1) are consumed: hi words to registers 0, 1
            lw_keep[rb][j] = lw;
        } else {
            put(2 + j, hw);                               // floats 2, 3 were consumed by half 0
            if (j == 0) {
                lw_keep2[rb] = lw;                        // register 6 still holds a float pair 1 needs
            } else {
                put(4, lw_keep[rb][0]); put(5, lw_keep[rb][1]); put(6, lw_keep2[rb]); put(7, lw);
            }
        }
    };
    auto p_frag = [&](int rb, int step, int lo) __attribute__((always_inline)) -> tx8 {
        const f32x16& v = sA[rb][step >> 1];
        const int o = 8 * (step & 1) + 4 * lo;
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 w = {v[o], v[o + 1], v[o + 2], v[o + 3]};
        return __builtin_bit_cast(tx8, w);
    };

    // one QK^T fragment group (K fragment hi, lo against both row blocks' Q: 6 MFMAs); FILL(k) runs behind MFMA k
    auto qk_group = [&](auto gc, auto&& fill) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, sl = g % NS, kb = g >> 2, tt = g & 3;
        const tx8 kf = __builtin_bit_cast(tx8, fr.ka[sl]);
        const tx8 kl = __builtin_bit_cast(tx8, fr.kl[sl]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if constexpr (tt == 0) W4Mfma<T>::qk0(sB[rb][kb], kf, qf[rb][0][tt]);
            else W4Mfma<T>::qk(sB[rb][kb], kf, qf[rb][0][tt]);
            fill(3 * rb + 0); W4_FENCE;
            W4Mfma<T>::qk(sB[rb][kb], kf, qf[rb][1][tt]);
            fill(3 * rb + 1); W4_FENCE;
            W4Mfma<T>::qk(sB[rb][kb], kl, qf[rb][0][tt]);
            fill(3 * rb + 2); W4_FENCE;
        }
    };

    // ---- pipeline prologue: S(0) and its softmax, no overlap ------------------------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int ib2 = 2 % nt;
    for (int g = 0; g < 3 && g < NT; ++g)
        if (g % nt == nt - 1) zero_masked_v(g);
    pp_barrier();
    {
        uint32_t kaddr[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) kaddr[tt] = k_lane[tt];
        const uint32_t vnone[2] = {0, 0};
        static_for<0, DIST>([&](auto gc) __attribute__((always_inline)) { issue_g(gc, kaddr, vnone); });
        static_for<0, 8>([&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value, sl = g % NS;
            if constexpr (g + DIST < 8) issue_g(std::integral_constant<int, g + DIST>{}, kaddr, vnone);
            constexpr int ahead = ahead_of(g, 8);
            asm volatile("s_waitcnt lgkmcnt(%c2)" : "+v"(fr.ka[sl]), "+v"(fr.kl[sl]) : "i"(ahead));
            W4_FENCE;
            qk_group(gc, [](int) {});
        });
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        mask_tail(0);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            max_piece(rb, 0, 0, 16);
            max_piece(rb, 1, 0, 16);
            max_tail(rb);
            exp_part(rb, 0, 32);
        }
        finish_softmax();
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int d = 0; d < 2; ++d) sA[rb][d] = sB[rb][d];
        if (nt == 1 && b0 + bstep < b1) {
            load_q_async(b0 + bstep, qf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pin_q(qf);
            if (!q_log2) prescale(qf);
        }
    }

    auto iteration = [&](auto more_c, int tau) __attribute__((always_inline)) {
        constexpr bool more = decltype(more_c)::value;
        const int tb_next = (ct + 1 == nt) ? 0 : ct + 1;
        uint32_t kaddr[4], vaddr[2];
        {
            const uint32_t kslot = (uint32_t)((tau + 1) & 3) * STAGE, vslot = (uint32_t)(tau & 3) * STAGE;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) kaddr[tt] = k_lane[tt] + kslot;
#pragma unroll
            for (int d = 0; d < 2; ++d) vaddr[d] = v_lane[d] + vslot;
        }
        // a wave whose 64 rows lie behind the sequence in the last query block of a pair only loads and synchronises
        const int qb_next = tb_next == 0 ? ((cqb + 1 == nqb) ? 0 : cqb + 1) : cqb;
        const bool active1 = (qb_next * 256 + wave * 64 < Sq);
        const bool active2 = (cqb * 256 + wave * 64 < Sq);
        const bool had_pending = wr_pending;
        if (wr_pending) { write_block(wr_b, wr_qb); wr_pending = false; }
        if (!had_pending) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                if (rescale[rb]) {                  // rare: the running maximum moved.  O leaves the AGPR half and returns inside this branch only
                    asm volatile("s_nop 7\n\ts_nop 7" : "+a"(o_acc[rb][0]), "+a"(o_acc[rb][1]));
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) o_acc[rb][d][r] *= alpha[rb];
                        asm volatile("" : "+a"(o_acc[rb][d]));
                    }
                    asm volatile("s_nop 4" ::: "memory");
                }
        }
        // ---------------- phase 1: QK^T(tau + 1) -> sB  ||  split of P(tau) = sA ------------------------------------------------------
        if (!active1 && !active2) {
        } else
        if constexpr (more) {
            static_for<0, DIST>([&](auto gc) __attribute__((always_inline)) { issue_g(gc, kaddr, vaddr); });
            static_for<0, 8>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value, sl = g % NS;
                issue_g(std::integral_constant<int, g + DIST>{}, kaddr, vaddr);
                constexpr int ahead = ahead_of(g, 16);
                asm volatile("s_waitcnt lgkmcnt(%c2)" : "+v"(fr.ka[sl]), "+v"(fr.kl[sl]) : "i"(ahead));
                W4_FENCE;
                // fillers: MFMA 0 -> row block 0 pair 0, 1 -> row block 0 pair 1, 2 -> row block 1 pair 0, 3 -> row block 1 pair 1 (6 issues each), 4, 5 bare
                qk_group(gc, [&](int k) __attribute__((always_inline)) {
                    if (k < 4 && !W4_NOFILL) split_pair(gc, k >> 1, k & 1);
                });
            });
        } else {
            static_for<0, 8>([&](auto gc) __attribute__((always_inline)) { split_pair(gc, 0, 0); split_pair(gc, 0, 1); split_pair(gc, 1, 0); split_pair(gc, 1, 1); });
            static_for<8, 8 + DIST>([&](auto gc) __attribute__((always_inline)) { issue_g(gc, kaddr, vaddr); });
        }
        const bool seam = (ct + 1 == nt);
        if (seam) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) { l_fin[rb] = l_run[rb]; m_run[rb] = -1e30f; l_run[rb] = 0.f; }
        }
        if constexpr (more) mask_tail(tb_next);
        // ---------------- phase 2: PV(tau) into O  ||  softmax of sB (tile tau + 1)  ||  the LDS-DMA of tile tau + 3, one piece per group ----------
        bool sent = false, q_loaded = false;
        if constexpr (more) {
            const int bq = (tb_next == 0 ? cb + bstep : cb) + bstep;
            if (tb_next == nt - 1 && bq < b1) { load_q_async(bq, qf); q_loaded = true; }
        }
        sent = tile_begin();
        auto soft_chunk = [&](auto gc, int k) __attribute__((always_inline)) {      // the softmax work of group g behind MFMA k (k = 0 .. 5)
            constexpr int g = decltype(gc)::value;
            if (W4_NOFILL) return;
            const int rb = k / 3, kk = k % 3;
            if constexpr (g == 0) {
                if (kk == 0) max_piece(rb, 0, 0, 8);
                else if (kk == 1) max_piece(rb, 0, 8, 16);
            } else if constexpr (g == 1) {
                if (kk == 0) max_piece(rb, 1, 0, 8);
                else if (kk == 1) max_piece(rb, 1, 8, 16);
                else max_tail(rb);
            } else {
                constexpr int i0 = g == 2 ? 0 : g == 3 ? 6 : g == 4 ? 12 : g == 5 ? 18 : g == 6 ? 22 : 28;
                constexpr int npairs = (g == 5 || g == 7) ? 2 : 3;
                if (kk < npairs) exp_one(rb, i0 + 2 * kk);
            }
        };
        auto pv_group = [&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value, sl = (g + 8) % NS, step = g >> 1, d = g & 1;
            issue_g(std::integral_constant<int, g + 8 + DIST>{}, kaddr, vaddr);          // nothing beyond group 15
            constexpr int ahead = ahead_of(g + 8, 16);
            asm volatile("s_waitcnt lgkmcnt(%c4)" : "+v"(fr.va0[sl]), "+v"(fr.va1[sl]), "+v"(fr.vl0[sl]), "+v"(fr.vl1[sl]) : "i"(ahead));
            W4_FENCE;
            const tx8 vf = __builtin_bit_cast(tx8, u32x4{fr.va0[sl][0], fr.va0[sl][1], fr.va1[sl][0], fr.va1[sl][1]});
            const tx8 vl = __builtin_bit_cast(tx8, u32x4{fr.vl0[sl][0], fr.vl0[sl][1], fr.vl1[sl][0], fr.vl1[sl][1]});
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const tx8 ph = p_frag(rb, step, 0);
                const tx8 pl = p_frag(rb, step, 1);
                W4Mfma<T>::pv(o_acc[rb][d], vf, ph);
                if constexpr (more) soft_chunk(gc, 3 * rb + 0);
                W4_FENCE;
                W4Mfma<T>::pv(o_acc[rb][d], vf, pl);
                if constexpr (more) soft_chunk(gc, 3 * rb + 1);
                W4_FENCE;
                W4Mfma<T>::pv(o_acc[rb][d], vl, ph);
                if constexpr (more) soft_chunk(gc, 3 * rb + 2);
                if (rb == 1 && sent) tile_piece(g);
                W4_FENCE;
            }
        };
        if (active1 || active2) static_for<0, 8>(pv_group);
        else if (sent) {
#pragma unroll
            for (int k = 0; k < 8; ++k) tile_piece(k);
        }
        // L2 prefetch of the Q rows loaded two iterations from now (one dword per 128-byte row segment; lane = row, half-wave = plane)
        bool q_pf_sent = false;
        if constexpr (more) {
            const int bq = (tb_next == 0 ? cb + bstep : cb) + bstep;
            if (tb_next == nt - 3 && bq < b1) {
                int qb;
                const int64_t base = block_base(bq, qb);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const T* ptr = qkv + hh * plane + base + (int64_t)q_row_of(qb, rb) * ld;
                    asm volatile("global_load_dword %0, %1, off" : "+v"(q_pf) : "v"(ptr) : "memory");
                }
                q_pf_sent = true;
            }
        }
        if constexpr (more) finish_softmax(); else { rescale[0] = false; rescale[1] = false; }
        if (seam) {
            wr_pending = true; wr_b = cb; wr_qb = cqb;
            ct = 0;
            if ((cb += bstep) < b1) block_base(cb, cqb);
        } else {
            ++ct;
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int d = 0; d < 2; ++d) sA[rb][d] = sB[rb][d];
        if (sent && q_pf_sent) asm volatile("s_waitcnt vmcnt(%c0)" :: "i"(NI + RB) : "memory");
        else if (sent) asm volatile("s_waitcnt vmcnt(%c0)" :: "i"(NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(q_pf));
        if (tau + 2 < NT && ib2 == nt - 1) zero_masked_v(tau + 2);
        ib2 = (ib2 + 1 == nt) ? 0 : ib2 + 1;
        pin_q(qf);
        if (q_loaded && !q_log2) prescale(qf);
        pp_barrier();
    };
    for (int tau = 0; tau < NT - 1; ++tau) iteration(std::true_type{}, tau);
    iteration(std::false_type{}, NT - 1);
    write_block(wr_b, wr_qb);
}

}  // namespace

// one workgroup per CU (160 KB of LDS): 4-deep K/V ring + output staging
template <typename T>
static hipError_t launch_attention_w4_t(const void* qkv, int64_t plane, void* out, int64_t o_plane, int nseq, int S, int S_pad, int H, hipStream_t s,
                                        int cus, bool q_log2, int Sq) {
    if (Sq <= 0) Sq = S_pad;
    constexpr int LDS = 4 * 2 * 64 * 128 * 2 + 8 * 4096;
    static std::mutex mu;
    static bool configured[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!configured[dev]) {
            e = hipFuncSetAttribute((const void*)attention_w4_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) return e;
            configured[dev] = true;
        }
    }
    const int nqb = (Sq + 255) / 256;
    const int nblk = nqb * (H / 64) * nseq;
    if (cus < 8 || nblk < 8) return hipErrorInvalidValue;
    const int items = nblk / nqb;
    const int xmax = ((items + 7) / 8) * nqb;
    int nx = cus / 8 < xmax ? cus / 8 : xmax;
    const int rounds = (xmax + nx - 1) / nx;
    nx = (xmax + rounds - 1) / rounds;
    hipLaunchKernelGGL((attention_w4_kernel<T>), dim3(8 * nx), dim3(256), LDS, s, (const T*)qkv, plane, (T*)out, o_plane, S, S_pad, H, nblk, q_log2 ? 1 : 0, Sq);
    return hipGetLastError();
}

hipError_t launch_attention_w4(const void* qkv, int64_t plane, void* out, int64_t o_plane, int nseq, int S, int S_pad, int H, int f16_, hipStream_t s,
                               int cus, bool q_log2, int Sq) {
    return f16_ ? launch_attention_w4_t<f16>(qkv, plane, out, o_plane, nseq, S, S_pad, H, s, cus, q_log2, Sq)
                : launch_attention_w4_t<bf16>(qkv, plane, out, o_plane, nseq, S, S_pad, H, s, cus, q_log2, Sq);
}

}  // namespace vtq
