// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  with fused epilogues.
//
// Replaces every torch.nn.Linear / patch Conv2d on the ViT path of the reference
// (modules/VisionTransformer/transformer.py:138-140,154-156,169 [QKV/out], :205-206,212-215 [MLP],
//  :475-480,531-532 [patch embedding as GEMM]).
//
// Design (cdna_hip_programming.md section 5, "256^2 8-phase template", re-derived for this shape family):
//   * 256x256 output tile per 512-thread workgroup, 8 waves, mfma_f32_16x16x32_bf16, 128 fp32 accumulators per lane.
//   * Both operands are K-contiguous (torch Linear layout) and go global -> LDS by 16-byte LDS-DMA
//     (global_load_lds_dwordx4).  The LDS image is lane-linear (DMA constraint); bank conflicts of the ds_read_b128
//     fragment reads are removed by XOR-swizzling the 16-byte chunk index on the SOURCE address and on the read
//     (SQ_LDS_BANK_CONFLICT = 0 measured).
//   * A K tile is four half-tile REGIONS (A rows 0-127 | 128-255, W rows 0-127 | 128-255).  Every wave owns 64 rows
//     of EACH A half and 32 columns of EACH W half, so an MFMA cluster on quadrant (mh, nh) touches exactly one A
//     region and one W region and regions are released early -> a DMA ring over two K-tile buffers that runs up to
//     4 half-tiles ahead under a COUNTED vmcnt (never 0 in the steady state) and raw s_barrier.
//   * Ping-pong: waves 0-3 and 4-7 (the two waves of each SIMD) run one barrier apart, so while one wave of a SIMD
//     issues its MFMA cluster the other issues LDS reads + DMA and waits for them.
//       2-phase schedule (default): per K tile   A: read A0,B0,B1 | 32 (x3: 48) MFMAs on quadrants (0,0),(0,1)
//                                                 B: read A1       | 32 (x3: 48) MFMAs on quadrants (1,1),(1,0)
//       LDS reads are retired (lgkmcnt(0)) BEFORE the phase barrier: the MFMA cluster starts right after the barrier and a
//       region may be re-staged one phase after its last read.  Measured fixed cost per barrier interval ~250 cycles, so
//       the longer clusters lift MFMA utilisation over a 4-phase form (16 MFMAs per interval; measured, removed).
//   * Operands are swapped in the MFMA (W fragment as A-operand): a lane holds 4 consecutive output columns of one row.
//   * Epilogues stage through LDS so every global access is a full row segment, 16 bytes per lane.
//   * NSPLIT == 3 ("bf16x3"): A and W arrive as hi/lo bf16 planes; each product is hi*hi + hi*lo + lo*hi into the same fp32
//     accumulator (3 MFMAs per 4 fragment reads); BK is 32 instead of 64 so regions keep their 16 KiB.
//   * Workgroups follow a host-built tile schedule (build_schedule): XCD-contiguous tile runs whose last round is filled
//     with 128-row half tiles.
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "dev_common.h"
#include "kernels.h"

namespace vtq {

namespace {

// chunk swizzle of a region row: BK=64 -> 128-byte rows / 8 chunks; BK=32 -> 64-byte rows / 4 chunks (DESIGN.md "LDS swizzle")
template <int BK> __device__ __forceinline__ int swz(int row) {
    if constexpr (BK == 64) return (row >> 1) & 7;
    else return ((row >> 3) & 1) * 3;
}

template <int N> __device__ __forceinline__ void wait_vm() {
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// allow `n` LDS-DMA instructions (2 per half-tile) to stay in flight; n is wave-uniform
__device__ __forceinline__ void wait_dma(int n) {
    if (n >= 8) wait_vm<8>();
    else if (n >= 6) wait_vm<6>();
    else if (n >= 4) wait_vm<4>();
    else if (n >= 2) wait_vm<2>();
    else wait_vm<0>();
}

// ---- shared epilogue of the ping-pong kernels -------------------------------------------------------------------------
template <int NSPLIT, int EPI, int MH>      // MH = 2: 256-row tile, MH = 1: 128-row half tile (rows m0 .. m0+127)
__device__ __forceinline__ void pp_epilogue(const GemmArgs& p, f32x4 (&acc)[2][2][4][2], char* smem, int tid, int wr, int wc,
                                            int fr, int fq, int64_t m0, int n0) {
    // ---- epilogue ------------------------------------------------------------------------------------------------------
    // acc[mh][nh][mi][ni][reg]: m = m0 + mh*128 + wr*64 + mi*16 + fr ; n = n0 + nh*128 + wc*32 + ni*16 + fq*4 + reg
    // bf16 outputs and the fp32 residual update go through LDS (free after the main loop) so that every global access is a
    // full row segment: one wave instruction = 2 rows x 512 B (bf16) or 1 row x 1 KiB (fp32), 16 bytes per lane.
    float4 b4[2][2], g4[2][2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + nh * 128 + wc * 32 + ni * 16 + fq * 4;
            b4[nh][ni] = *(const float4*)(p.bias + n);
            if constexpr (EPI == EPI_RESID) g4[nh][ni] = p.gamma ? *(const float4*)(p.gamma + n) : float4{1.f, 1.f, 1.f, 1.f};
        }

    // The staged forms work in CHUNKS through two LDS images (above the first K-tile buffer, see STG below): in every barrier
    // interval the workgroup copies chunk k-1 out to global memory (ds_read_b128 -> 16-byte row-segment stores) and converts
    // chunk k into the other image, so the stores' address-path time hides behind the next chunk's VALU / LDS writes.
    constexpr int STG = 65536;                          // staging base: the ring's second K-tile buffer and the slack above it
    auto interval_end = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
        // chunk (mh, q): the 64 rows {mh*128 + wr*64 + (2q + e)*16 + fr}, image row = wr*32 + e*16 + fr; one pass per plane
        constexpr int RS = 528;                         // 256 bf16 + 16 B pad: rows stay 16-B aligned for ds_read_b128
        constexpr int IMG = 64 * RS;
        constexpr int NP = (NSPLIT == 1) ? 1 : 2;
        constexpr int NPASS = MH * 2 * NP;
        bf16x4 lo[(NSPLIT == 1) ? 1 : 8];               // lo plane of the chunk in flight (written by the next pass)
        auto convert = [&](int pass) {                  // pass = (mh*2 + q)*NP + pl; all indices fold after unrolling
            const int pl = pass % NP, cq = pass / NP, mh = cq >> 1, q = cq & 1;
            char* img = smem + STG + (pass & 1) * IMG;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const int lrow = wr * 32 + e * 16 + fr;
                        const int col = nh * 128 + wc * 32 + ni * 16 + fq * 4;
                        const int li = (e * 2 + nh) * 2 + ni;
                        bf16x4 h;
                        if (pl == 0) {
                            const f32x4 a = acc[mh][nh][2 * q + e][ni];
                            const float4 bb = b4[nh][ni];
                            float v[4] = {a[0] + bb.x, a[1] + bb.y, a[2] + bb.z, a[3] + bb.w};
                            if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) v[k] = gelu_erf(v[k]);
                            }
                            if constexpr (NSPLIT == 1) {
                                h = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                            } else {
                                bf16x4 l;
#pragma unroll
                                for (int k = 0; k < 4; ++k) { bf16 x, y; split2(v[k], x, y); h[k] = x; l[k] = y; }
                                lo[li] = l;
                            }
                        } else {
                            h = lo[(NSPLIT == 1) ? 0 : li];
                        }
                        *(bf16x4*)(img + lrow * RS + col * 2) = h;
                    }
        };
        auto copy_out = [&](int pass) {
            const int pl = pass % NP, cq = pass / NP, mh = cq >> 1, q = cq & 1;
            const char* img = smem + STG + (pass & 1) * IMG;
            const int c16 = tid & 31, r0 = tid >> 5;
            bf16* og = (bf16*)p.out + pl * p.o_plane + m0 * p.ldo + n0 + c16 * 8;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {            // image row ps*16 + r0 = (wr = ps>>1, e = ps&1, fr = r0)
                const int grow = mh * 128 + (ps >> 1) * 64 + (2 * q + (ps & 1)) * 16 + r0;
                const uint4 v = *(const uint4*)(img + (ps * 16 + r0) * RS + c16 * 16);
                store_nt16(og + (int64_t)grow * p.ldo, v);
            }
        };
        convert(0);
        interval_end();
#pragma unroll
        for (int pass = 1; pass < NPASS; ++pass) {
            copy_out(pass - 1);
            convert(pass);
            interval_end();
        }
        copy_out(NPASS - 1);
    } else if constexpr (EPI == EPI_RESID) {
        // chunk (mh, mi): the 32 rows {mh*128 + wr*64 + mi*16 + fr}, image row = wr*16 + fr, fp32; the residual rows of chunk
        // k are requested one interval before they are needed
        constexpr int RS = 1040;                        // 256 fp32 + 16 B pad (odd multiple of 16: conflict-free b128 writes)
        constexpr int IMG = 32 * RS;
        constexpr int NCH = MH * 4;
        const int c16 = tid & 63, r0 = tid >> 6;
        auto grow_of = [&](int ch, int ps) { return (ch >> 2) * 128 + (ps >> 1) * 64 + (ch & 3) * 16 + (ps & 1) * 8 + r0; };
        auto convert = [&](int ch) {
            const int mh = ch >> 2, mi = ch & 3;
            char* img = smem + STG + (ch & 1) * IMG;
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int col = nh * 128 + wc * 32 + ni * 16 + fq * 4;
                    const f32x4 a = acc[mh][nh][mi][ni];
                    const float4 bb = b4[nh][ni], gg = g4[nh][ni];
                    const float4 v = {gg.x * (a[0] + bb.x), gg.y * (a[1] + bb.y), gg.z * (a[2] + bb.z), gg.w * (a[3] + bb.w)};
                    *(float4*)(img + (wr * 16 + fr) * RS + col * 4) = v;
                }
        };
        float* xg = p.x + m0 * p.N + n0 + c16 * 4;
        float4 xv[2][4];
        auto load_x = [&](int ch) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) xv[ch & 1][ps] = *(const float4*)(xg + (int64_t)grow_of(ch, ps) * p.N);
        };
        auto copy_out = [&](int ch) {
            const char* img = smem + STG + (ch & 1) * IMG;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {            // image row ps*8 + r0 = (wr = ps>>1, fr = (ps&1)*8 + r0)
                const float4 d = *(const float4*)(img + (ps * 8 + r0) * RS + c16 * 16);
                float4 x = xv[ch & 1][ps];
                x.x += d.x; x.y += d.y; x.z += d.z; x.w += d.w;
                *(float4*)(xg + (int64_t)grow_of(ch, ps) * p.N) = x;
            }
        };
        load_x(0);
        convert(0);
        interval_end();
#pragma unroll
        for (int ch = 1; ch < NCH; ++ch) {
            load_x(ch);
            copy_out(ch - 1);
            convert(ch);
            interval_end();
        }
        copy_out(NCH - 1);
    } else {  // EPI_EMBED: scattered rows + table gathers, once per forward: direct from registers
#pragma unroll
        for (int mh = 0; mh < MH; ++mh)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int64_t m = m0 + mh * 128 + wr * 64 + mi * 16 + fr;
                const int orow = p.row_map[m];
                const int i1 = p.idx1[m];
                const int i2 = p.table2 ? p.idx2[m] : 0;
                float4 t1[2][2], t2[2][2];
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const int n = n0 + nh * 128 + wc * 32 + ni * 16 + fq * 4;
                        t1[nh][ni] = *(const float4*)(p.table1 + (int64_t)i1 * p.N + n);
                        t2[nh][ni] = p.table2 ? *(const float4*)(p.table2 + (int64_t)i2 * p.N + n) : float4{0.f, 0.f, 0.f, 0.f};
                    }
                if (orow >= 0) {
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            const int n = n0 + nh * 128 + wc * 32 + ni * 16 + fq * 4;
                            const f32x4 a = acc[mh][nh][mi][ni];
                            const float4 bb = b4[nh][ni];
                            float4 r = {a[0] + bb.x + t1[nh][ni].x + t2[nh][ni].x, a[1] + bb.y + t1[nh][ni].y + t2[nh][ni].y,
                                        a[2] + bb.z + t1[nh][ni].z + t2[nh][ni].z, a[3] + bb.w + t1[nh][ni].w + t2[nh][ni].w};
                            *(float4*)(p.x + (int64_t)orow * p.N + n) = r;
                        }
                }
            }
    }
}

// ---- common prologue of both schedules ----------------------------------------------------------------------------------
// Tile descriptor of a workgroup.  With a schedule (GemmArgs::sched, built on the host by build_schedule below) block b runs
// sched[b] = (tile << 2) | kind, kind 0 = full 256x256 tile, 1 / 2 = top / bottom 128-row half of it; without one the blocks
// are remapped so that consecutive tiles (sharing an A row panel) land on one XCD's L2.
struct Tile { int64_t m0; int n0; int half; };
__device__ __forceinline__ Tile tile_of_block(const GemmArgs& p) {
    const int ntn = p.N / 256, ntm = p.M / 256;
    int bid = blockIdx.x, kind = 0;
    if (p.sched) {
        const int d = p.sched[bid];
        bid = d >> 2;
        kind = d & 3;
    } else {
        const int nwg = ntn * ntm;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;      // bijective XCD-contiguous remap
    }
    const int tm = bid / ntn, tn = bid - tm * ntn;
    return Tile{(int64_t)tm * 256 + (kind == 2 ? 128 : 0), tn * 256, kind != 0};
}

#define VTQ_PP_COMMON()                                                                                                  \
    constexpr int BK = (NSPLIT == 1) ? 64 : 32;                                                                          \
    constexpr int ROWB = BK * 2;                                                                                          \
    constexpr int REG_B = 16384; /* one half-tile region (all planes) */                                                  \
    constexpr int BUF_B = 4 * REG_B; /* one K tile: regions 0 = A half 0, 1 = W half 0, 2 = W half 1, 3 = A half 1 */     \
    constexpr int NFA = 8, NFB = 4;                                                                                       \
    extern __shared__ __attribute__((aligned(16))) char smem[];                                                           \
    const int tid = threadIdx.x;                                                                                          \
    const int lane = tid & 63;                                                                                            \
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                            \
    const int wr = wave >> 2, wc = wave & 3;                                                                              \
    const int fr = lane & 15, fq = lane >> 4;                                                                             \
    const Tile tl = tile_of_block(p);                                                                              \
    const int64_t m0 = tl.m0;                                                                                             \
    const int n0 = tl.n0;                                                                                                 \
    const bf16* __restrict__ Ag = (const bf16*)p.A + m0 * p.lda;                                                          \
    const bf16* __restrict__ Wg = (const bf16*)p.W + (int64_t)n0 * p.K;                                                   \
    const int nkt = p.K / BK;                                                                                             \
    /* DMA source offsets of this thread inside a half-tile (two rounds of 512 x 16 B; NSPLIT 3: round == plane) */       \
    int64_t a_off[2], w_off[2];                                                                                           \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                                                       \
        if constexpr (NSPLIT == 1) {                                                                                      \
            const int slot = r * 512 + tid;                                                                               \
            const int row = slot >> 3, c = (slot & 7) ^ swz<64>(row);                                                     \
            a_off[r] = (int64_t)row * p.lda + c * 8;                                                                      \
            w_off[r] = (int64_t)row * p.K + c * 8;                                                                        \
        } else {                                                                                                          \
            const int row = tid >> 2, c = (tid & 3) ^ swz<32>(row);                                                       \
            a_off[r] = r * p.a_plane + (int64_t)row * p.lda + c * 8;                                                      \
            w_off[r] = r * p.w_plane + (int64_t)row * p.K + c * 8;                                                        \
        }                                                                                                                 \
    }                                                                                                                     \
    /* stage half-tile region r of K tile kt (caller guarantees kt < nkt) */                                              \
    auto stage = [&](int kt, int r) {                                                                                     \
        char* dst = smem + (kt & 1) * BUF_B + r * REG_B + wave * 1024;                                                    \
        const bool isA = (r == 0 || r == 3);                                                                              \
        const int half = (r >= 2) ? 1 : 0;                                                                                \
        const bf16* base = isA ? Ag + (int64_t)half * 128 * p.lda + kt * BK : Wg + (int64_t)half * 128 * p.K + kt * BK;   \
        glds16(base + (isA ? a_off[0] : w_off[0]), dst);                                                                  \
        glds16(base + (isA ? a_off[1] : w_off[1]), dst + 8192);                                                           \
    };                                                                                                                    \
    /* fragment read offsets inside a region: x = k-step (NSPLIT 1) or plane (NSPLIT 3) */                                \
    int a_rd[2], b_rd[2];                                                                                                 \
    _Pragma("unroll") for (int x = 0; x < 2; ++x) {                                                                       \
        if constexpr (NSPLIT == 1) {                                                                                      \
            const int ch = ((x * 4 + fq) ^ swz<64>(fr)) << 4;                                                             \
            a_rd[x] = (wr * 64 + fr) * ROWB + ch;                                                                         \
            b_rd[x] = (wc * 32 + fr) * ROWB + ch;                                                                         \
        } else {                                                                                                          \
            const int ch = (fq ^ swz<32>(fr)) << 4;                                                                       \
            a_rd[x] = x * 8192 + (wr * 64 + fr) * ROWB + ch;                                                              \
            b_rd[x] = x * 8192 + (wc * 32 + fr) * ROWB + ch;                                                              \
        }                                                                                                                 \
    }                                                                                                                     \
    f32x4 acc[2][2][4][2];                                                                                                \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                                         \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                     \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};                \
    bf16x8 fa[NFA], fb0[NFB], fb1[NFB];                                                                                   \
    auto read_a = [&](const char* reg) {                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                     \
            _Pragma("unroll") for (int x = 0; x < 2; ++x) fa[i * 2 + x] = *(const bf16x8*)(reg + a_rd[x] + i * 16 * ROWB); \
    };                                                                                                                    \
    auto read_b = [&](const char* reg, bf16x8(&fb)[NFB]) {                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                     \
            _Pragma("unroll") for (int x = 0; x < 2; ++x) fb[j * 2 + x] = *(const bf16x8*)(reg + b_rd[x] + j * 16 * ROWB); \
    };                                                                                                                    \
    auto mma = [&](f32x4(&c)[4][2], const bf16x8(&fb)[NFB]) {                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                     \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                               \
                if constexpr (NSPLIT == 1) {                                                                              \
                    c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j * 2 + 0], fa[i * 2 + 0], c[i][j], 0, 0, 0);    \
                    c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j * 2 + 1], fa[i * 2 + 1], c[i][j], 0, 0, 0);    \
                } else {                                                                                                  \
                    c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j * 2 + 0], fa[i * 2 + 0], c[i][j], 0, 0, 0);    \
                    c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j * 2 + 0], fa[i * 2 + 1], c[i][j], 0, 0, 0);    \
                    c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j * 2 + 1], fa[i * 2 + 0], c[i][j], 0, 0, 0);    \
                }                                                                                                         \
            }                                                                                                             \
    };

// =====================================================================================================================
// 2-phase ping-pong schedule.
//   DMA groups in issue order: g = 2k: {A0, W0, W1} of K tile k (6 DMA instructions); g = 2k+1: {A1} of tile k (2).
//   prologue issues groups 0,1,2; phase A(kt) issues group 2kt+3, phase B(kt) issues group 2kt+4.
//   phase A(kt) reads A0,W0,W1 (waited for in B(kt-1) / prologue) and waits for group 2kt+1 (read in B(kt));
//   phase B(kt) reads A1 and waits for group 2kt+2 (read in A(kt+1)).
//   WAR: a region is read (and the read retired by lgkmcnt(0)) before its reader passes the phase barrier; the other wave
//        group re-stages it in the NEXT phase, i.e. after that barrier.  RAW: a group is waited for by every wave one phase
//        before the first read, and both wave groups' waits precede the barrier that opens the reading phase.
//
// Half tiles (128 x 256, rows of A half 0 only) are phase A alone: one group {A0, W0, W1} per K tile on a ring of THREE
// 48 KiB buffers, tile kt+2 staged in phase kt into the buffer read in phase kt-1 (same WAR/RAW argument), vmcnt(6) steady.
// They exist for the tile schedule (build_schedule), which fills the last round of a launch in half-tile granules.
template <int NSPLIT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_pp2_kernel(GemmArgs p) {
    VTQ_PP_COMMON()
#define VTQ_SYNC_OPEN(n_dma)                                   \
    wait_dma(n_dma);                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_sched_barrier(0);                         \
    __builtin_amdgcn_s_barrier();                              \
    __builtin_amdgcn_sched_barrier(0);                         \
    __builtin_amdgcn_s_setprio(1);
#define VTQ_SYNC_CLOSE()                                       \
    __builtin_amdgcn_s_setprio(0);                             \
    __builtin_amdgcn_sched_barrier(0);                         \
    __builtin_amdgcn_s_barrier();                              \
    __builtin_amdgcn_sched_barrier(0);

    if (!tl.half) {
        // ---- prologue: groups 0, 1, 2 ----------------------------------------------------------------------------
        stage(0, 0); stage(0, 1); stage(0, 2); stage(0, 3);
        if (nkt > 1) { stage(1, 0); stage(1, 1); stage(1, 2); }
        wait_dma(nkt > 1 ? 8 : 2);
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();        // second wave group runs one barrier behind

        for (int kt = 0; kt < nkt; ++kt) {
            const char* buf = smem + (kt & 1) * BUF_B;
            // ---- phase A --------------------------------------------------------------------------------------------
            read_b(buf + 1 * REG_B, fb0);
            read_a(buf + 0 * REG_B);
            read_b(buf + 2 * REG_B, fb1);
            if (kt + 1 < nkt) stage(kt + 1, 3);
            VTQ_SYNC_OPEN(kt + 1 < nkt ? 8 : 0)
            mma(acc[0][0], fb0);
            mma(acc[0][1], fb1);
            VTQ_SYNC_CLOSE()
            // ---- phase B --------------------------------------------------------------------------------------------
            read_a(buf + 3 * REG_B);
            if (kt + 2 < nkt) { stage(kt + 2, 0); stage(kt + 2, 1); stage(kt + 2, 2); }
            VTQ_SYNC_OPEN(kt + 2 < nkt ? 8 : (kt + 1 < nkt ? 2 : 0))
            mma(acc[1][1], fb1);
            mma(acc[1][0], fb0);
            VTQ_SYNC_CLOSE()
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();        // match the extra barrier of the second group
        pp_epilogue<NSPLIT, EPI, 2>(p, acc, smem, tid, wr, wc, fr, fq, m0, n0);
    } else {
        constexpr int BUF_H = 3 * REG_B;
        auto stage_h = [&](int kt, int slot) {             // {A0, W0, W1} of K tile kt into ring slot
            char* dst = smem + slot * BUF_H + wave * 1024;
            const bf16* ab = Ag + kt * BK;
            const bf16* w0 = Wg + kt * BK;
            const bf16* w1 = Wg + (int64_t)128 * p.K + kt * BK;
            glds16(ab + a_off[0], dst);
            glds16(ab + a_off[1], dst + 8192);
            glds16(w0 + w_off[0], dst + REG_B);
            glds16(w0 + w_off[1], dst + REG_B + 8192);
            glds16(w1 + w_off[0], dst + 2 * REG_B);
            glds16(w1 + w_off[1], dst + 2 * REG_B + 8192);
        };
        stage_h(0, 0);
        if (nkt > 1) stage_h(1, 1);
        wait_dma(nkt > 1 ? 6 : 0);
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();
        int slot = 0;                                      // kt % 3
        for (int kt = 0; kt < nkt; ++kt) {
            const char* buf = smem + slot * BUF_H;
            read_b(buf + 1 * REG_B, fb0);
            read_a(buf + 0 * REG_B);
            read_b(buf + 2 * REG_B, fb1);
            const int nslot = slot == 0 ? 2 : slot - 1;    // (kt + 2) % 3
            if (kt + 2 < nkt) stage_h(kt + 2, nslot);
            VTQ_SYNC_OPEN(kt + 2 < nkt ? 6 : 0)
            mma(acc[0][0], fb0);
            mma(acc[0][1], fb1);
            VTQ_SYNC_CLOSE()
            slot = slot == 2 ? 0 : slot + 1;
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
        pp_epilogue<NSPLIT, EPI, 1>(p, acc, smem, tid, wr, wc, fr, fq, m0, n0);
    }
#undef VTQ_SYNC_OPEN
#undef VTQ_SYNC_CLOSE
}

// =====================================================================================================================
// Tile schedule (host).  Workgroups are dispatched in blockIdx order, round-robin over the 8 XCDs, one per CU (LDS-bound),
// and a CU takes the next block when it retires one.  The schedule keeps each XCD on a contiguous run of tiles (A row panels
// and W stay in its L2) and ends the run with 128-row HALF tiles, so the last round is filled in half-tile granules: N = 768
// GEMMs are 1.5 rounds at 64 pairs, qkv 4.5 -- measured -10 % (out-proj), -6 % (fc2), -3 % (qkv) against whole tiles only.
// The number of split tiles is chosen by simulating the greedy dispatch per XCD.  Output never depends on the schedule:
// every element is accumulated over K in the same order in both tile forms.
// (Starting half of the CUs on a half tile, to keep neighbours' epilogue write bursts out of phase, was measured too:
//  no gain -- a write burst starves every CU's operand stream, not only the writers'.)
constexpr int kXcds = 8, kCusPerXcd = 32;
constexpr double kHalfCost = 0.57;              // measured: a launch of half tiles only takes 1.14x the whole-tile launch

double greedy_makespan(int n_full, int n_half) {
    double cu[kCusPerXcd] = {0};
    auto put = [&](double d) {
        int best = 0;
        for (int i = 1; i < kCusPerXcd; ++i) if (cu[i] < cu[best]) best = i;
        cu[best] += d;
    };
    for (int i = 0; i < n_full; ++i) put(1.0);
    for (int i = 0; i < n_half; ++i) put(kHalfCost);
    double m = 0;
    for (double v : cu) m = v > m ? v : m;
    return m;
}

// sequence of one XCD owning tiles [t0, t0 + cnt): whole tiles, then `tail` tiles as top/bottom halves
void xcd_sequence(const int* order, int cnt, std::vector<int>& out) {
    static const bool all_halves = [] { const char* v = getenv("VTQ_GEMM_SCHED"); return v && v[0] == '2'; }();   // measurement knob
    int best_tail = 0;
    double best = 1e30;
    for (int tail = 0; tail <= cnt && tail <= 2 * kCusPerXcd; ++tail) {
        const double m = greedy_makespan(cnt - tail, 2 * tail) + 1e-3 * tail;
        if (m < best) { best = m; best_tail = tail; }
    }
    if (all_halves) best_tail = cnt;
    for (int i = 0; i < cnt - best_tail; ++i) out.push_back(order[i] << 2);
    for (int i = cnt - best_tail; i < cnt; ++i) { out.push_back((order[i] << 2) | 1); out.push_back((order[i] << 2) | 2); }
}

std::vector<int> build_schedule(int ntm, int ntn) {
    const int nt = ntm * ntn, q = nt / kXcds, r = nt % kXcds;
    // tile order the XCD runs are cut from: row-major, i.e. the column tiles of one A row panel are neighbours.  (Cutting the
    // runs from column BANDS of 3 / 4 / 6 tiles instead -- fewer distinct W tiles per XCD round -- measured no different.)
    std::vector<int> order(nt);
    for (int i = 0; i < nt; ++i) order[i] = i;
    std::vector<std::vector<int>> seq(kXcds);
    size_t total = 0;
    for (int x = 0; x < kXcds; ++x) {
        const int t0 = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        xcd_sequence(order.data() + t0, q + (x < r ? 1 : 0), seq[x]);
        total += seq[x].size();
    }
    std::vector<int> out;
    out.reserve(total);
    for (size_t j = 0; out.size() < total; ++j)             // block b -> XCD b % 8 while every XCD still has entries
        for (int x = 0; x < kXcds; ++x)
            if (j < seq[x].size()) out.push_back(seq[x][j]);
    return out;
}

struct DevSched { int* dev; int n; };
// process-wide cache keyed by the tile grid; entries are a few KiB and live until exit
hipError_t schedule_for(int ntm, int ntn, DevSched& ds) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, DevSched> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find({ntm, ntn});
    if (it != cache.end()) { ds = it->second; return hipSuccess; }
    const std::vector<int> h = build_schedule(ntm, ntn);
    DevSched d{nullptr, (int)h.size()};
    hipError_t e = hipMalloc(&d.dev, h.size() * sizeof(int));
    if (e != hipSuccess) return e;
    e = hipMemcpy(d.dev, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);     // blocking; first launch of a shape only
    if (e != hipSuccess) return e;
    cache[{ntm, ntn}] = d;
    ds = d;
    return hipSuccess;
}

template <int NSPLIT, int EPI> hipError_t launch_t(GemmArgs a, hipStream_t s) {
    constexpr int LDS = 147456;                    // full tiles: DMA ring 128 KiB, epilogue images 256x528 B; half tiles: 3 x 48 KiB
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_pp2_kernel<NSPLIT, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        configured = true;
    }
    static const bool use_sched = [] { const char* v = getenv("VTQ_GEMM_SCHED"); return !(v && v[0] == '0'); }();
    int nwg = (a.M / 256) * (a.N / 256);
    a.sched = nullptr;
    if (use_sched) {
        DevSched ds;
        hipError_t e = schedule_for(a.M / 256, a.N / 256, ds);
        if (e != hipSuccess) return e;
        a.sched = ds.dev;
        nwg = ds.n;
    }
    hipLaunchKernelGGL((gemm_pp2_kernel<NSPLIT, EPI>), dim3(nwg), dim3(512), LDS, s, a);
    return hipGetLastError();
}

}  // namespace

std::vector<int> gemm_tile_schedule(int ntm, int ntn) { return build_schedule(ntm, ntn); }

hipError_t launch_gemm(const GemmArgs& a, int nsplit, int epilogue, hipStream_t s) {
    if (a.M <= 0 || a.M % 256 || a.N % 256 || a.K % 64 || a.lda % 8 || (nsplit != 1 && nsplit != 3)) return hipErrorInvalidValue;
#define VTQ_CASE(NS, EP) if (nsplit == NS && epilogue == EP) return launch_t<NS, EP>(a, s);
    VTQ_CASE(1, EPI_BIAS) VTQ_CASE(1, EPI_BIAS_GELU) VTQ_CASE(1, EPI_RESID) VTQ_CASE(1, EPI_EMBED)
    VTQ_CASE(3, EPI_BIAS) VTQ_CASE(3, EPI_BIAS_GELU) VTQ_CASE(3, EPI_RESID) VTQ_CASE(3, EPI_EMBED)
#undef VTQ_CASE
    return hipErrorInvalidValue;
}

}  // namespace vtq
