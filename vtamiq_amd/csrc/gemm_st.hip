// Small-tile MFMA GEMM for gfx950: the same contraction as gemm.hip's 256x256 persistent kernel, for launches whose 256x256 tile count
// cannot fill 256 CUs -- the small-batch / single-query regime (the reference's own FLOP probe is batch 1 x 500 patches,
// modules/utils.py:68-78) and the N = 768 launches of mid-size batches.
//
//   C[M,N] = A[M,K] * W[N,K]^T  with the epilogues of kernels.h (EPI_BIAS, EPI_BIAS_GELU, EPI_RESID, EPI_EMBED); replaces the same
//   torch.nn.Linear / patch Conv2d call sites as gemm.hip (modules/VisionTransformer/transformer.py:138-140,154-156,169,205-215,475-480).
//
// BITWISE CONTRACT.  Every output element goes through exactly the instruction sequence gemm_pp2_kernel gives it: the accumulator
// starts at the bias (zero for the patch embedding), takes one mfma_f32_16x16x32 per (32-deep k-step, term) in ascending k with the terms
// in the order  w_hi*a_hi, w_hi*a_lo, w_lo*a_hi  (2-term: w*a_hi, w*a_lo), the W fragment as the MFMA's first operand, lane group fq
// holding k elements 8 fq .. 8 fq + 7 of the step -- and the epilogue arithmetic is the same operations in the same order (gelu_erf4 /
// split4_f16 / split2 from dev_common.h; gamma * acc rounded BEFORE the residual add).  So which kernel a launch uses is a pure speed
// choice: scores do not depend on the batch size (tests/test_gpu_kernels.py: st tiles == 256x256 tiles bit for bit; the batch-invariance
// tests of test_gpu_parity.py run across the selection boundary).
//
// Structure (cdna_hip_programming.md section 5: "128^2 tile / grouped GEMM at 2-3 blocks/CU", glds + counted vmcnt + raw barrier):
//   * one workgroup per BM x BN tile: WR x WC MFMA waves, each a (BM/WR) x (BN/WC) sub-tile of 16x16 accumulator blocks, plus NPW PRODUCER waves that
//     do nothing but issue the LDS-DMA of the ring, wait for it (counted vmcnt) and synchronise: with one MFMA wave per SIMD nothing else would run
//     beside a wave's DMA issue and landing wait (the load path alone and the compute path alone each take ~2/3 of what they took together);
//   * operands global -> LDS by 16-byte LDS-DMA into a ring of NST stages; a stage is one 64-deep K slice (two 32-deep k-steps side
//     by side) with the LDS image [plane][row][128 B], so that every DMA wave-instruction reads 8 full 128-byte lines
//     ("x through LDS in full 128-B lines"); the image is lane-linear per DMA piece (NT/8 rows), the 16-byte chunk index
//     XOR-swizzled on the SOURCE address and on the fragment read (the involution of gemm.hip's 128-byte rows: swz(row) = (row >> 1) & 7;
//     conflict-free for ds_read_b128 over fr = 0..15, fq = 0..3);
//   * one raw s_barrier per stage; a stage is waited for by every wave with a COUNTED vmcnt (NST - 2 stages stay in flight) before
//     the barrier that precedes its first read, and re-staged after the barrier that follows its last read (retired by lgkmcnt(0));
//   * epilogue straight from the accumulators (a lane holds 4 consecutive columns of one row: 8-byte plane stores, 16-byte fp32
//     read-modify-write of the residual stream);
//   * tile -> workgroup map: workgroup b runs on XCD b % 8 (round-robin dispatch); the 8 XCDs form a gr x gc grid over the tile grid, chosen on the host
//     so that what an XCD pulls through its L2 -- 1/gr of A (freshly written: Infinity Cache) + 1/gc of W (a layer's weights are COLD inside a
//     forward: HBM) -- is smallest.
// What bounds it (profiles/r05_gemm_tile_shapes.txt): with WARM operands (a benchmark that re-uses one weight buffer) a CU's L1 -> LDS fill rate (~90 GB/s
// with the loads alone) -- a 64x64 tile moves 32 KiB per 24 MFMAs per wave -- and how well the load and compute paths overlap: DMA-only producer waves
// beside the MFMA waves take 24 - 27 % off the B = 1 out-proj / fc2 there.  Inside a forward every layer's weights are COLD (HBM: 344 MB of weights against a
// 256 MB Infinity Cache) and the W stream of a column tile is one dependent chain of 128-byte-per-row stage fetches that all the XCD's row panels wait on:
// the B = 1 fc2 takes ~31 us cold whatever the ring depth, the XCD grid or a W-prefetching wave (built, slower), against 23 us warm; the producers' gain
// inside a forward is 3 - 10 %.  So small tiles only pay while the 256x256 form leaves most CUs idle (gemm.hip gemm_tile_rule).
#include "dev_common.h"
#include "kernels.h"

#include <cstdlib>
#include <mutex>

namespace vtq {

namespace {

template <int N> __device__ __forceinline__ void st_wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MODE (exploration builds only): 0 = the kernel; 1 = no fragment reads / MFMAs (the load path alone); 2 = no LDS-DMA (compute + barriers
// alone); 3 = 2 without the barriers; 4 = 2 without the fragment reads; 5 = MFMAs alone
// NPW: producer waves.  0: the WR x WC MFMA waves issue the LDS-DMA themselves.  > 0: NPW extra waves do nothing but issue the DMA, wait for it and
// synchronise -- the MFMA waves then spend no issue slots on DMA addressing and never stall on vmcnt (one wave per SIMD has nothing else to overlap with).
template <typename T, int TERMS, int EPI, int BM, int BN, int WR, int WC, int NST, int MODE = 0, int NPW = 0>
__global__ __launch_bounds__(64 * (WR * WC + NPW)) void gemm_st_kernel(GemmArgs p) {
    typedef typename Vec<T>::x8 tx8;
    typedef typename Vec<T>::x4 tx4;
    constexpr int NCW = WR * WC;                          // MFMA waves
    constexpr int NT = 64 * (NPW ? NPW : NCW);            // threads that issue the DMA
    constexpr int APL = (TERMS == 1) ? 1 : 2, WPL = (TERMS == 3) ? 2 : 1;
    constexpr int WM = BM / WR, WN = BN / WC, MI = WM / 16, NJ = WN / 16;
    constexpr int KS = 2, ROWB = 128;                     // k-steps per stage; bytes of an LDS row
    constexpr int PR = NT / 8;                            // rows of one DMA piece: every thread moves 16 bytes
    static_assert(BM % PR == 0 && BN % PR == 0 && WM % 16 == 0 && WN % 16 == 0, "tile / workgroup shape");
    constexpr int PA = BM / PR, PW = BN / PR;             // pieces per plane
    constexpr int SUB_A = APL * BM * ROWB, SUB_W = WPL * BN * ROWB, STAGE = SUB_A + SUB_W;
    constexpr int G = APL * PA + WPL * PW;                // LDS-DMA instructions per thread and stage
    static_assert(NST >= 2 && NST <= 5 && (NST - 2) * G <= 63, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- tile of this workgroup.  Workgroup b runs on XCD b % 8 (round-robin dispatch).  The 8 XCDs form a gr x gc grid over the tile grid (host-chosen,
    // launch_gemm_st): XCD (xr, xc) owns the row panels [xr ntm / gr, (xr + 1) ntm / gr) x the column tiles [xc ntn / gc, ...), walked column-major, so that
    // consecutive workgroups share a W tile.  What an XCD pulls into its L2 is then 1/gr of A and 1/gc of W.
    const int ntn = p.N / BN, ntm = p.M / BM;
    const int gr = p.st_grid >> 4, gc = p.st_grid & 15;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int xr = xcd / gc, xc = xcd - xr * gc;
    const int r0 = xr * ntm / gr, nr = (xr + 1) * ntm / gr - r0;
    const int c0 = xc * ntn / gc, nc = (xc + 1) * ntn / gc - c0;
    if (j >= nr * nc) return;
    const int tn = c0 + j / nr, tm = r0 + (j - (j / nr) * nr);
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave - wr * WC;
    const int lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int nkt = p.K / 64;
    const bool producer = NPW > 0 && wave >= NCW;         // wave-uniform
    const int dtid = NPW ? tid - NCW * 64 : tid, dwave = NPW ? wave - NCW : wave;      // position among the DMA-issuing threads / waves

    // ---- DMA addressing ---------------------------------------------------------------------------------------------------------
    const int prow = dtid >> 3;
    const int pch = (dtid & 7) ^ ((prow >> 1) & 7);
    const uint32_t a_off = (uint32_t)(prow * p.lda + pch * 8) * 2u;
    const uint32_t w_off = (uint32_t)(prow * p.K + pch * 8) * 2u;
    const char* Ag = (const char*)p.A + m0 * p.lda * 2;
    const char* Wg = (const char*)p.W + (int64_t)n0 * p.K * 2;
    const int64_t a_pl = p.a_plane * 2, w_pl = p.w_plane * 2;
    const int64_t a_rb = (int64_t)PR * p.lda * 2, w_rb = (int64_t)PR * p.K * 2;
    auto issue = [&](int kt, int s) {
        if constexpr (MODE >= 2) return;
        char* dst = smem + s * STAGE + dwave * 1024;
        const int kb = kt * ROWB;                         // byte column of the stage
#pragma unroll
        for (int pl = 0; pl < APL; ++pl)
#pragma unroll
            for (int rb = 0; rb < PA; ++rb) glds16(Ag + pl * a_pl + rb * a_rb + kb + a_off, dst + (pl * BM + rb * PR) * ROWB);
#pragma unroll
        for (int pl = 0; pl < WPL; ++pl)
#pragma unroll
            for (int rb = 0; rb < PW; ++rb) glds16(Wg + pl * w_pl + rb * w_rb + kb + w_off, dst + SUB_A + (pl * BN + rb * PR) * ROWB);
    };
    // ---- producer waves: the DMA ring and nothing else (they leave BEFORE the tile's ordinary loads: their wave index is no MFMA-wave position) -------
    if constexpr (NPW > 0) {
        if (producer) {
#pragma unroll
            for (int s = 0; s < NST - 1; ++s)
                if (s < nkt) issue(s, s);
            int slot = 0;
            for (int kt = 0; kt < nkt; ++kt) {
                const int rem = nkt - kt - 1, st = rem < NST - 2 ? rem : NST - 2;
                if (st <= 0) st_wait_vm<0>();
                else if (st == 1) st_wait_vm<(G <= 63) ? G : 63>();
                else if (st == 2) st_wait_vm<(2 * G <= 63) ? 2 * G : 63>();
                else st_wait_vm<(3 * G <= 63) ? 3 * G : 63>();
                __builtin_amdgcn_s_barrier();             // stage kt is published; the MFMA waves have retired their reads of stage kt - 1
                if (kt + NST - 1 < nkt) issue(kt + NST - 1, slot == 0 ? NST - 1 : slot - 1);
                slot = (slot == NST - 1) ? 0 : slot + 1;
            }
            return;
        }
    }

    // fragment read offsets: k-step ks of a stage = chunks 4 ks .. 4 ks + 3 of the 128-byte rows
    int rdA[KS], rdW[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int rch = ((ks * 4 + fq) ^ ((fr >> 1) & 7)) << 4;
        rdA[ks] = (wr * WM + fr) * ROWB + rch;
        rdW[ks] = SUB_A + (wc * WN + fr) * ROWB + rch;
    }

    // ---- ordinary loads first, LDS-DMA prologue behind them, first use after it -----------------------------------------------------
    // The bias (the accumulators start at it) and, for the residual form, this lane's x values are requested BEFORE the prologue's
    // DMA and consumed after it: one memory latency instead of three in a row (hipcc drains vmcnt(0) at the first use of a
    // VGPR-destination load while a DMA is in flight -- here that drain is the wait for stage 0 the loop would do anyway).
    constexpr bool kBiasInAcc = EPI != EPI_EMBED;
    f32x4 bias4[NJ];
    float4 xv[(EPI == EPI_RESID) ? MI : 1][(EPI == EPI_RESID) ? NJ : 1];
    int e_orow[(EPI == EPI_EMBED) ? MI : 1], e_i1[(EPI == EPI_EMBED) ? MI : 1], e_i2[(EPI == EPI_EMBED) ? MI : 1];
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) {
        bias4[jn] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (kBiasInAcc) bias4[jn] = *(const f32x4*)(p.bias + n0 + wc * WN + jn * 16 + fq * 4);
    }
    if constexpr (EPI == EPI_RESID) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) xv[i][jn] = *(const float4*)(p.x + (m0 + wr * WM + i * 16 + fr) * p.N + n0 + wc * WN + jn * 16 + fq * 4);
    }
    if constexpr (EPI == EPI_EMBED) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int64_t m = m0 + wr * WM + i * 16 + fr;
            e_orow[i] = p.row_map[m];
            e_i1[i] = p.idx1[m];
            e_i2[i] = p.table2 ? p.idx2[m] : 0;
        }
    }
    if constexpr (NPW == 0) {
#pragma unroll
        for (int s = 0; s < NST - 1; ++s)
            if (s < nkt) issue(s, s);
    }
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][jn] = bias4[jn];

    // ---- fragment reads and the MFMA block of one k-step ------------------------------------------------------------------------
    struct Frag { tx8 a[MI][APL], b[NJ][WPL]; };
    auto read_frag = [&](Frag& f, const char* buf, int ks) {
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
            for (int pl = 0; pl < WPL; ++pl) f.b[jn][pl] = *(const tx8*)(buf + rdW[ks] + pl * BN * ROWB + jn * 16 * ROWB);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int pl = 0; pl < APL; ++pl) f.a[i][pl] = *(const tx8*)(buf + rdA[ks] + pl * BM * ROWB + i * 16 * ROWB);
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) {
                f32x4 c = acc[i][jn];
                if constexpr (TERMS == 1) {
                    c = mfma16<T>(f.b[jn][0], f.a[i][0], c);
                } else if constexpr (TERMS == 2) {
                    c = mfma16<T>(f.b[jn][0], f.a[i][0], c);
                    c = mfma16<T>(f.b[jn][0], f.a[i][1], c);
                } else {
                    c = mfma16<T>(f.b[jn][0], f.a[i][0], c);
                    c = mfma16<T>(f.b[jn][0], f.a[i][1], c);
                    c = mfma16<T>(f.b[jn][1], f.a[i][0], c);
                }
                acc[i][jn] = c;
            }
    };
    // counted wait: all but the `stages` youngest stages of this thread's LDS-DMA have landed (stages is clamped to what the ring holds)
    auto wait_stages = [&](int stages) {
        if constexpr (NPW > 0) return;                    // the MFMA waves issue no DMA
        if (stages <= 0) st_wait_vm<0>();
        else if (stages == 1) st_wait_vm<(G <= 63) ? G : 63>();
        else if (stages == 2) st_wait_vm<(2 * G <= 63) ? 2 * G : 63>();
        else st_wait_vm<(3 * G <= 63) ? 3 * G : 63>();
    };

    int slot = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        // stage kt landed (this wave's part), the previous stage's fragment reads retired; then everyone's
        const int rem = nkt - kt - 1;                 // stages issued behind this one (at most NST - 2 at this point)
        wait_stages(rem < NST - 2 ? rem : NST - 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (MODE != 3 && MODE != 5) __builtin_amdgcn_s_barrier();
        // re-stage the slot read in iteration kt - 1
        if constexpr (NPW == 0)
            if (kt + NST - 1 < nkt) issue(kt + NST - 1, slot == 0 ? NST - 1 : slot - 1);
        const char* buf = smem + slot * STAGE;
        if constexpr (MODE >= 4) {         // exploration: no fragment reads (the same registers every k-step)
            Frag f;
            if (kt == 0) read_frag(f, buf, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(f.a[i][0]));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) mma(f);
        } else
        if constexpr (MODE != 1) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag f;
                read_frag(f, buf, ks);
                mma(f);
            }
        }
        slot = (slot == NST - 1) ? 0 : slot + 1;
    }

    // ---- epilogue (same arithmetic, same order as gemm.hip pp_epilogue; every DMA has landed: plain loads are safe) -------------
    // acc[i][jn][reg]: m = m0 + wr*WM + i*16 + fr ; n = n0 + wc*WN + jn*16 + fq*4 + reg
    if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int64_t m = m0 + wr * WM + i * 16 + fr;
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) {
                const int n = n0 + wc * WN + jn * 16 + fq * 4;
                const f32x4 a = acc[i][jn];
                float v[4] = {a[0], a[1], a[2], a[3]};
                if constexpr (EPI == EPI_BIAS_GELU) gelu_erf4(v);
                T* og = (T*)p.out + m * p.ldo + n;
                if constexpr (APL == 1) {
                    *(tx4*)og = tx4{(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
                } else {
                    tx4 h, l;
                    if constexpr (std::is_same<T, f16>::value) split4_f16(v, h, l);
                    else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { T x, y; split2<T>(v[k], x, y); h[k] = x; l[k] = y; }
                    }
                    *(tx4*)og = h;
                    *(tx4*)(og + p.o_plane) = l;
                }
            }
        }
    } else if constexpr (EPI == EPI_RESID) {
        float4 g4[NJ];
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) {
            const int n = n0 + wc * WN + jn * 16 + fq * 4;
            g4[jn] = p.gamma ? *(const float4*)(p.gamma + n) : float4{1.f, 1.f, 1.f, 1.f};
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int64_t m = m0 + wr * WM + i * 16 + fr;
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) {
                const f32x4 a = acc[i][jn];
                // gamma * acc is ROUNDED to fp32 before the add (the 256x256 kernel passes it through an fp32 LDS image): no fma contraction
                float d0 = g4[jn].x * a[0], d1 = g4[jn].y * a[1], d2 = g4[jn].z * a[2], d3 = g4[jn].w * a[3];
                asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
                float4 x = xv[i][jn];
                x.x += d0; x.y += d1; x.z += d2; x.w += d3;
                *(float4*)(p.x + m * p.N + n0 + wc * WN + jn * 16 + fq * 4) = x;
            }
        }
    } else {  // EPI_EMBED: + bias + pos_table[idx1] (+ scale_table[idx2]), scattered to the token rows (transformer.py:531-558)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int orow = e_orow[i], i1 = e_i1[i], i2 = e_i2[i];
            if (orow < 0) continue;
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) {
                const int n = n0 + wc * WN + jn * 16 + fq * 4;
                const float4 bb = *(const float4*)(p.bias + n);
                const float4 t1 = *(const float4*)(p.table1 + (int64_t)i1 * p.N + n);
                const float4 t2 = p.table2 ? *(const float4*)(p.table2 + (int64_t)i2 * p.N + n) : float4{0.f, 0.f, 0.f, 0.f};
                const f32x4 a = acc[i][jn];
                float4 rr = {a[0] + bb.x + t1.x + t2.x, a[1] + bb.y + t1.y + t2.y, a[2] + bb.z + t1.z + t2.z, a[3] + bb.w + t1.w + t2.w};
                *(float4*)(p.x + (int64_t)orow * p.N + n) = rr;
            }
        }
    }
}

// ---- host ----------------------------------------------------------------------------------------------------------------------
// The XCD grid (rows x columns, product 8) that minimises what one XCD reads: M / gr rows of A + kColdW x N / gc rows of W (same K, same planes);
// W counts double: inside a forward it comes from HBM, A from the Infinity Cache.  Only grids that leave no XCD without tiles.
constexpr int kColdW = 2;
int st_xcd_grid(int M, int N, int ntm, int ntn) {
    static const int forced = [] { const char* v = VTQ_MEASURE_ENV("VTQ_ST_GRID"); return v ? atoi(v) : 0; }();   // measurement builds: rows * 16 + columns
    if (forced) return forced;
    int best = (8 << 4) | 1;
    long best_cost = -1;
    for (int gr = 8; gr >= 1; gr >>= 1) {
        const int gc = 8 / gr;
        if (gr > ntm || gc > ntn) continue;
        const long cost = (long)M / gr + (long)kColdW * N / gc;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = (gr << 4) | gc; }
    }
    return best;
}

template <typename T, int TERMS, int EPI, int BM, int BN, int WR, int WC, int NST, int MODE = 0, int NPW = 0>
hipError_t launch_v(const GemmArgs& a, hipStream_t s) {
    constexpr int APL = (TERMS == 1) ? 1 : 2, WPL = (TERMS == 3) ? 2 : 1;
    constexpr int LDS = NST * (APL * BM + WPL * BN) * 128;
    static_assert(LDS <= 163840, "LDS ring");
    auto kern = gemm_st_kernel<T, TERMS, EPI, BM, BN, WR, WC, NST, MODE, NPW>;
    if constexpr (LDS > 65536) {
        static std::mutex mu;
        static bool configured[64] = {false};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!configured[dev]) {
            e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) return e;
            configured[dev] = true;
        }
    }
    const int ntm = a.M / BM, ntn = a.N / BN;
    GemmArgs b = a;
    b.st_grid = st_xcd_grid(a.M, a.N, ntm, ntn);
    const int gr = b.st_grid >> 4, gc = b.st_grid & 15;
    const int grid = 8 * ((ntm + gr - 1) / gr) * ((ntn + gc - 1) / gc);       // workgroups beyond an XCD's rectangle leave at once
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (WR * WC + NPW)), LDS, s, b);
    return hipGetLastError();
}

template <typename T, int TERMS, int EPI> hipError_t launch_shape(const GemmArgs& a, int variant, hipStream_t s) {
    switch (variant) {
        // MFMA waves 2x2 (64x64) or 2x4 (128x128) + DMA-only producer waves (profiles/r05_gemm_tile_shapes.txt: -24 .. -27 % on the B = 1 out-proj / fc2,
        // -16 .. -18 % on the 128x128 fc2 of B = 3 .. 5 against the same tiles with the MFMA waves issuing the DMA themselves)
        case GEMM_ST_64: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 3, 0, 2>(a, s);    // ring of 3: 96 KiB (3-term formats), one workgroup per CU; 4 + 2 waves
        case GEMM_ST_64X2: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 2, 0, 4>(a, s);  // ring of 2: 64 KiB, two workgroups per CU; 4 + 4 waves
        case GEMM_ST_128: return launch_v<T, TERMS, EPI, 128, 128, 2, 4, 2, 0, 8>(a, s); // ring of 2: 128 KiB; 8 + 8 waves
#ifdef VTQ_GEMM_ST_EXPLORE                                   // tile-shape exploration builds (tools/st_bench.py): not in the product library
        case 9: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 3>(a, s);                   // the three shapes WITHOUT producer waves
        case 10: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 2>(a, s);
        case 18: return launch_v<T, TERMS, EPI, 128, 128, 2, 4, 2>(a, s);
        case 4: return launch_v<T, TERMS, EPI, 128, 64, 2, 2, 3>(a, s);
        case 5: return launch_v<T, TERMS, EPI, 64, 128, 2, 2, 3>(a, s);
        case 6: return launch_v<T, TERMS, EPI, 128, 128, 2, 2, 2>(a, s);
        case 7: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 4>(a, s);
        case 8: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 5>(a, s);
        case 20: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 3, 0, 4>(a, s);
        case 21: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 2, 0, 4>(a, s);
        case 22: return launch_v<T, TERMS, EPI, 128, 128, 2, 4, 2, 0, 4>(a, s);
        case 23: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 3, 0, 2>(a, s);
        case 24: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 4, 0, 4>(a, s);
        case 25: return launch_v<T, TERMS, EPI, 128, 128, 2, 4, 2, 0, 8>(a, s);
        case 11: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 3, 1>(a, s);
        case 12: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 3, 2>(a, s);
        case 13: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 3, 3>(a, s);
        case 14: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 3, 4>(a, s);
        case 15: return launch_v<T, TERMS, EPI, 64, 64, 2, 2, 3, 5>(a, s);
        case 16: return launch_v<T, TERMS, EPI, 128, 128, 2, 4, 2, 1>(a, s);
        case 17: return launch_v<T, TERMS, EPI, 128, 128, 2, 4, 2, 2>(a, s);
#endif
    }
    return hipErrorInvalidValue;
}

template <typename T, int TERMS> hipError_t launch_epi(const GemmArgs& a, int epilogue, int variant, hipStream_t s) {
    switch (epilogue) {
        case EPI_BIAS: return launch_shape<T, TERMS, EPI_BIAS>(a, variant, s);
        case EPI_BIAS_GELU: return launch_shape<T, TERMS, EPI_BIAS_GELU>(a, variant, s);
        case EPI_RESID: return launch_shape<T, TERMS, EPI_RESID>(a, variant, s);
        case EPI_EMBED: return launch_shape<T, TERMS, EPI_EMBED>(a, variant, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_gemm_st(const GemmArgs& a, Num num, int epilogue, int variant, hipStream_t s) {
    if (a.M <= 0 || a.M % 256 || a.N <= 0 || a.N % 256 || a.K <= 0 || a.K % 64 || a.lda % 16 || !num_valid(num) || num.f16 == 2) return hipErrorInvalidValue;
    if (!num.f16) {
        if (num.terms == 1) return launch_epi<bf16, 1>(a, epilogue, variant, s);
        if (num.terms == 3) return launch_epi<bf16, 3>(a, epilogue, variant, s);
    } else {
        if (num.terms == 1) return launch_epi<f16, 1>(a, epilogue, variant, s);
        if (num.terms == 2) return launch_epi<f16, 2>(a, epilogue, variant, s);
        if (num.terms == 3) return launch_epi<f16, 3>(a, epilogue, variant, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace vtq
