// MFMA GEMM for gfx950 (16-bit and e4m3 operands):  C[M,N] = A[M,K] * W[N,K]^T  with fused epilogues, persistent over a tile schedule.
//
// Replaces every torch.nn.Linear / patch Conv2d on the ViT path of the reference
// (modules/VisionTransformer/transformer.py:138-140,154-156,169 [QKV/out], :205-206,212-215 [MLP],
//  :475-480,531-532 [patch embedding as GEMM]).
//
// Design (cdna_hip_programming.md section 5, "256^2 8-phase template", re-derived for this shape family):
//   * 256x256 output tile per 512-thread workgroup, 8 waves, mfma_f32_16x16x32_{bf16,f16}, 128 fp32 accumulators per lane.
//   * Both operands are K-contiguous (torch Linear layout) and go global -> LDS by 16-byte LDS-DMA
//     (global_load_lds_dwordx4).  The LDS image is lane-linear (DMA constraint); bank conflicts of the ds_read_b128
//     fragment reads are removed by XOR-swizzling the 16-byte chunk index on the SOURCE address and on the read
//     (SQ_LDS_BANK_CONFLICT: 2 % of the launch's cycles for the 16-bit forms, profiles/archive/r02_gemm_fc1_pmc.txt).
//   * A K tile is four half-tile REGIONS (A rows 0-127 | 128-255, W rows 0-127 | 128-255).  Every wave owns 64 rows
//     of EACH A half and 32 columns of EACH W half, so an MFMA cluster on quadrant (mh, nh) touches exactly one A
//     region and one W region and regions are released early -> a DMA ring over two K-tile buffers that runs up to
//     4 half-tiles ahead under a COUNTED vmcnt (never 0 in the steady state) and raw s_barrier.
//   * Ping-pong: waves 0-3 and 4-7 (the two waves of each SIMD) run one barrier apart, so while one wave of a SIMD
//     issues its MFMA cluster the other issues LDS reads + DMA and waits for them.
//       2-phase schedule: per K tile   A: read A0,W0,W1 | MFMAs on quadrants (0,0),(0,1)
//                                      B: read A1       | MFMAs on quadrants (1,1),(1,0)
//       LDS reads are retired (lgkmcnt(0)) BEFORE the phase barrier: the MFMA cluster starts right after the barrier and a
//       region may be re-staged one phase after its last read.
//   * Operands are swapped in the MFMA (W fragment as A-operand): a lane holds 4 consecutive output columns of one row.
//   * Epilogues stage through LDS so every global access is a full row segment, 16 bytes per lane.
//   * TERMS = MFMAs per product (operand format):
//       1: a*w                                   one 16-bit plane each, BK = 64
//       2: (a_hi + a_lo)*w                       activation hi/lo planes, weight single plane, BK = 32
//       3: a_hi*w_hi + a_lo*w_hi + a_hi*w_lo     hi/lo planes for both, BK = 32
//     into the same fp32 accumulator.
//     T = f8 (TERMS 1): OCP e4m3 bytes on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales, BK = 128; the epilogue first
//     multiplies the accumulators by wscale[n] * ascale_inv (per-output-channel weight scale, static activation scale).
//   * PERSISTENT: at most one workgroup per CU walks its own list of tiles (host-built schedule).  The DMA ring runs
//     straight through a tile boundary: the next tile's first K tile is staged during the current tile's last K tile, so
//     it lands under the epilogue; only the K tile that would overwrite the epilogue's LDS staging area is deferred to
//     the end of the epilogue.  Epilogue stores are never drained (counted vmcnt steps over them).
//   * Schedule: each XCD (workgroups b, b+8, ...) owns a contiguous run of tiles, ordered column-group-major so that the
//     32 tiles it works on at a time are (32/cg row panels) x (cg column tiles) with the cg W tiles L2-resident; the run
//     ends with 128-row half tiles so the last round is filled in half-tile granules.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
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
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifndef VTQ_RESID_DEFER
#define VTQ_RESID_DEFER 0             // 1: residual epilogue with the copy-out's LDS reads issued before the next chunk's conversion (A/B: profiles/r05_epilogue_balanced.txt)
#endif
#ifndef VTQ_EPI_BALANCED
#define VTQ_EPI_BALANCED 1            // 0: the plane-alternating passes of rounds 2 - 4 (A/B: profiles/r05_epilogue_balanced.txt)
#endif

constexpr int kHalfRows = 128;

// ---- shared epilogue of the ping-pong kernels -------------------------------------------------------------------------
// VMEM instructions one wave issues in the epilogue after its bias / gamma loads (all unconditional in the staged forms):
// stores, and for the residual form the interleaved loads of x.  The counted wait of a following tile steps over them.
template <int OPL, int EPI, int MH> constexpr int epilogue_vmem_ops() {
    return (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) ? MH * 2 * OPL * 4 : MH * 4 * 8;
}

template <typename T, int OPL, int EPI, int MH> constexpr int epilogue_ops_of() {     // T = OUTPUT element type
    return std::is_same<T, f8>::value ? ((EPI == EPI_BIAS_GELU) ? MH * 2 * 2 : epilogue_vmem_ops<OPL, EPI, MH>()) : epilogue_vmem_ops<OPL, EPI, MH>();
}

// T = OUTPUT element type of the 16-bit / 8-bit forms; SCALED: fp8 operands (acc * wscale[n] * ascale_inv before the bias)
#ifdef VTQ_GEMM_DIAG
struct EpiDiag { unsigned long long conv, copy, wait; };
#define VTQ_EPI_T0() unsigned long long dg_e0; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_e0) :: "memory");
#define VTQ_EPI_T1(field) { unsigned long long dg_e1; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_e1) :: "memory"); ed.field += dg_e1 - dg_e0; dg_e0 = dg_e1; }
#else
struct EpiDiag {};
#define VTQ_EPI_T0()
#define VTQ_EPI_T1(field)
#endif

// BIASED: the accumulators were initialised with the bias (gemm_pp2_kernel init_acc), so the forms below add nothing
template <typename T, int OPL, int EPI, int MH, bool SCALED, bool BIASED>      // MH = 2: 256-row tile, MH = 1: 128-row half tile (rows m0 .. m0+127)
__device__ __forceinline__ void pp_epilogue(const GemmArgs& p, f32x4 (&acc)[2][2][4][2], char* smem, int tid, int wr, int wc,
                                            int fr, int fq, int64_t m0, int n0, EpiDiag& ed) {
    typedef typename Vec<T>::x4 tx4;
    // opaque copies of the lane indices: every per-lane address below is then formed HERE, after the main loop, instead of being
    // hoisted above it and kept (or spilled) across it
    asm volatile("" : "+v"(tid), "+v"(fr), "+v"(fq));
    if constexpr (SCALED) {           // de-scale the accumulators once, in place: the forms below then see plain values
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int n = n0 + nh * 128 + wc * 32 + ni * 16 + fq * 4;
                float4 s4 = *(const float4*)(p.wscale + n);
                s4.x *= p.ascale_inv; s4.y *= p.ascale_inv; s4.z *= p.ascale_inv; s4.w *= p.ascale_inv;
#pragma unroll
                for (int mh = 0; mh < MH; ++mh)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        f32x4& a = acc[mh][nh][mi][ni];
                        a[0] *= s4.x; a[1] *= s4.y; a[2] *= s4.z; a[3] *= s4.w;
                    }
            }
    }
    // acc[mh][nh][mi][ni][reg]: m = m0 + mh*128 + wr*64 + mi*16 + fr ; n = n0 + nh*128 + wc*32 + ni*16 + fq*4 + reg
    // 16-bit outputs and the fp32 residual update go through LDS (free after the main loop) so that every global access is a
    // full row segment: one wave instruction = 2 rows x 512 B (16-bit) or 1 row x 1 KiB (fp32), 16 bytes per lane.
    float4 b4[2][2], g4[2][2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + nh * 128 + wc * 32 + ni * 16 + fq * 4;
            if constexpr (!BIASED) b4[nh][ni] = *(const float4*)(p.bias + n);
            if constexpr (EPI == EPI_RESID) g4[nh][ni] = p.gamma ? *(const float4*)(p.gamma + n) : float4{1.f, 1.f, 1.f, 1.f};
        }

    // The staged forms work in CHUNKS through two LDS images (in the ring's second K-tile buffer and the slack above it; the
    // first buffer may already be receiving the next tile): in every barrier interval the workgroup copies chunk k-1 out to
    // global memory (ds_read_b128 -> 16-byte row-segment stores) and converts chunk k into the other image.
    constexpr int STG = 65536;
    auto interval_end = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    if constexpr ((EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) && std::is_same<T, f8>::value) {
        // e4m3 output (the GELU form of the fp8 mode): chunk (mh, q) as below, image rows of 256 bytes, one pass per chunk
        constexpr int RS = 272;                         // 256 B + 16 B pad
        constexpr int IMG = 64 * RS;
        constexpr int NPASS = MH * 2;
        float amax8 = 0.f;                              // max |GELU output| of this lane before scaling (kernels.h Fp8Obs)
        auto convert = [&](int pass) {
            const int mh = pass >> 1, q = pass & 1;
            char* img = smem + STG + (pass & 1) * IMG;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const int lrow = wr * 32 + e * 16 + fr;
                        const int col = nh * 128 + wc * 32 + ni * 16 + fq * 4;
                        const f32x4 a = acc[mh][nh][2 * q + e][ni];
                        const float4 bb = b4[nh][ni];
                        float v[4] = {a[0] + bb.x, a[1] + bb.y, a[2] + bb.z, a[3] + bb.w};
                        if constexpr (EPI == EPI_BIAS_GELU) gelu_erf4(v);
                        amax8 = amax4(amax8, v[0], v[1], v[2], v[3]);
                        *(uint32_t*)(img + lrow * RS + col) = pack_fp8x4(v[0] * p.out_scale, v[1] * p.out_scale, v[2] * p.out_scale, v[3] * p.out_scale);
                    }
        };
        auto copy_out = [&](int pass) {
            const int mh = pass >> 1, q = pass & 1;
            const char* img = smem + STG + (pass & 1) * IMG;
            const int c16 = tid & 15, r0 = tid >> 4;    // 16 threads x 16 B per 256-byte row, 32 rows per sweep
            char* og = (char*)p.out + m0 * p.ldo + n0 + c16 * 16;
            u32x4 v[2];
            lds_read_rows2<32 * RS>(img + r0 * RS + c16 * 16, v);
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {            // image row ps*32 + r0 = (wr = ps, e = r0 >> 4, fr = r0 & 15)
                const int grow = mh * 128 + ps * 64 + (2 * q + (r0 >> 4)) * 16 + (r0 & 15);
                store_nt16(og + (int64_t)grow * p.ldo, uint4{v[ps][0], v[ps][1], v[ps][2], v[ps][3]});
            }
        };
        convert(0);
        interval_end();
#pragma unroll
        for (int pass = 1; pass < NPASS; ++pass) {
            if (wr == 0) { copy_out(pass - 1); convert(pass); }
            else { convert(pass); copy_out(pass - 1); }
            interval_end();
        }
        copy_out(NPASS - 1);
        fp8_report(p.obs, amax8, p.out_scale);
#if VTQ_EPI_BALANCED
    } else if constexpr ((EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) && OPL == 2) {
        // Two output planes, BALANCED passes (round 5): chunk (mh, mi) = the 32 rows {mh*128 + wr*64 + mi*16 + fr} with BOTH planes per pass --
        // image = [plane][row wr*16 + fr][256 x 16 bit + pad] -- so that every barrier interval holds the same work (the form below alternates a pass
        // with all of a chunk's arithmetic and a pass that only writes the kept lo plane, and needs 16 registers for that plane).  Within an
        // interval every wave ISSUES the LDS reads of the previous chunk's copy-out, converts the next chunk under them, then waits and stores:
        // the read latency sits under the conversion and both waves of a SIMD convert at the same time (two waves issue vector instructions
        // every 3.3 cycles, a lone one every 6.6: profiles/r05_gelu_packed.txt section 1).  Same values, same stores per thread.
        constexpr int RS = 528;
        constexpr int IMG = 64 * RS;                    // 2 planes x 32 rows
        constexpr int NPASS = MH * 4;
        auto convert = [&](int pass) {
            const int mh = pass >> 2, mi = pass & 3;
            char* img = smem + STG + (pass & 1) * IMG;
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int lrow = wr * 16 + fr;
                    const int col = nh * 128 + wc * 32 + ni * 16 + fq * 4;
                    const f32x4 a = acc[mh][nh][mi][ni];
                    float v[4] = {a[0], a[1], a[2], a[3]};
                    if constexpr (!BIASED) {
                        const float4 bb = b4[nh][ni];
                        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                    }
                    if constexpr (EPI == EPI_BIAS_GELU) gelu_erf4(v);
                    tx4 h, l;
                    if constexpr (std::is_same<T, f16>::value) {
                        split4_f16(v, h, l);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { T x, y; split2<T>(v[k], x, y); h[k] = x; l[k] = y; }
                    }
                    *(tx4*)(img + lrow * RS + col * 2) = h;
                    *(tx4*)(img + (32 + lrow) * RS + col * 2) = l;
                }
        };
        const int c16 = tid & 31, r0 = tid >> 5;
        u32x4 cv[4];                                    // image rows r0, 16 + r0 of plane 0, then of plane 1 (16 * RS apart)
        auto copy_issue = [&](int pass) {               // hidden from hipcc's waitcnt bookkeeping (see lds_read_rows4); copy_store waits
            const char* img = smem + STG + (pass & 1) * IMG;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%c5\n\tds_read_b128 %2, %4 offset:%c6\n\tds_read_b128 %3, %4 offset:%c7"
                         : "=&v"(cv[0]), "=&v"(cv[1]), "=&v"(cv[2]), "=&v"(cv[3])
                         : "v"(lds_addr(img + r0 * RS + c16 * 16)), "i"(16 * RS), "i"(32 * RS), "i"(48 * RS) : "memory");
        };
        auto copy_store = [&](int pass) {
            const int mh = pass >> 2, mi = pass & 3;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cv[0]), "+v"(cv[1]), "+v"(cv[2]), "+v"(cv[3]) :: "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) {               // k = plane * 2 + wr
                const int grow = mh * 128 + (k & 1) * 64 + mi * 16 + r0;
                T* og = (T*)p.out + (k >> 1) * p.o_plane + (m0 + grow) * p.ldo + n0 + c16 * 8;
                store_nt16(og, uint4{cv[k][0], cv[k][1], cv[k][2], cv[k][3]});
            }
        };
        VTQ_EPI_T0()
        convert(0);
        VTQ_EPI_T1(conv)
        interval_end();
        VTQ_EPI_T1(wait)
#pragma unroll
        for (int pass = 1; pass < NPASS; ++pass) {
            copy_issue(pass - 1);
            convert(pass);
            VTQ_EPI_T1(conv)
            copy_store(pass - 1);
            VTQ_EPI_T1(copy)
            interval_end();
            VTQ_EPI_T1(wait)
        }
        copy_issue(NPASS - 1);
        copy_store(NPASS - 1);
        VTQ_EPI_T1(copy)
#endif
    } else if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
        // chunk (mh, q): the 64 rows {mh*128 + wr*64 + (2q + e)*16 + fr}, image row = wr*32 + e*16 + fr; one pass per plane
        constexpr int RS = 528;                         // 256 x 16 bit + 16 B pad: rows stay 16-B aligned for ds_read_b128
        constexpr int IMG = 64 * RS;
        constexpr int NP = OPL;
        constexpr int NPASS = MH * 2 * NP;
        tx4 lo[(NP == 1) ? 1 : 8];                      // lo plane of the chunk in flight (written by the next pass)
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
                        tx4 h;
                        if (pl == 0) {
                            const f32x4 a = acc[mh][nh][2 * q + e][ni];
                            float v[4] = {a[0], a[1], a[2], a[3]};
                            if constexpr (!BIASED) {
                                const float4 bb = b4[nh][ni];
                                v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                            }
#if !(defined(VTQ_EPI_ABL) && VTQ_EPI_ABL == 1)                           // measurement build 1: no GELU arithmetic
                            if constexpr (EPI == EPI_BIAS_GELU) gelu_erf4(v);
#endif
                            if constexpr (NP == 1) {
                                h = tx4{(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
                            } else if constexpr (std::is_same<T, f16>::value) {
                                tx4 l;
                                split4_f16(v, h, l);
                                lo[li] = l;
                            } else {
                                tx4 l;
#pragma unroll
                                for (int k = 0; k < 4; ++k) { T x, y; split2<T>(v[k], x, y); h[k] = x; l[k] = y; }
                                lo[li] = l;
                            }
                        } else {
                            h = lo[(NP == 1) ? 0 : li];
                        }
#if defined(VTQ_EPI_ABL) && VTQ_EPI_ABL == 4                               // measurement build 4: no LDS staging either
                        asm volatile("" ::"v"(h));
                        (void)img; (void)lrow; (void)col;
#else
                        *(tx4*)(img + lrow * RS + col * 2) = h;
#endif
                    }
        };
        auto copy_out = [&](int pass) {
            const int pl = pass % NP, cq = pass / NP, mh = cq >> 1, q = cq & 1;
            const char* img = smem + STG + (pass & 1) * IMG;
            const int c16 = tid & 31, r0 = tid >> 5;
            T* og = (T*)p.out + pl * p.o_plane + m0 * p.ldo + n0 + c16 * 8;
#if defined(VTQ_EPI_ABL) && (VTQ_EPI_ABL == 2 || VTQ_EPI_ABL == 4)          // measurement builds: no copy-out at all
            (void)img; (void)og; (void)mh; (void)q; (void)r0; (void)c16;
#else
            u32x4 v[4];
            lds_read_rows4<16 * RS>(img + r0 * RS + c16 * 16, v);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {            // image row ps*16 + r0 = (wr = ps>>1, e = ps&1, fr = r0)
                const int grow = mh * 128 + (ps >> 1) * 64 + (2 * q + (ps & 1)) * 16 + r0;
#if defined(VTQ_EPI_ABL) && VTQ_EPI_ABL == 3                               // measurement build: LDS read kept, global store dropped
                asm volatile("" ::"v"(v[ps])); (void)grow;
#else
                store_nt16(og + (int64_t)grow * p.ldo, uint4{v[ps][0], v[ps][1], v[ps][2], v[ps][3]});
#endif
            }
#endif
        };
        // Within an interval the two wave groups (the two waves of every SIMD) run the two steps in OPPOSITE order -- they touch
        // different images, so either order is valid -- so that one wave's VALU conversion runs while its SIMD partner sits in
        // the store-issue queue, instead of both queueing and then both converting.
        VTQ_EPI_T0()
        convert(0);
        VTQ_EPI_T1(conv)
        interval_end();
        VTQ_EPI_T1(wait)
#pragma unroll
        for (int pass = 1; pass < NPASS; ++pass) {
#if defined(VTQ_EPI_ORDER) && VTQ_EPI_ORDER == 1          // measurement builds: both wave groups copy first / convert first
            const bool copy_first = true;
#elif defined(VTQ_EPI_ORDER) && VTQ_EPI_ORDER == 2
            const bool copy_first = false;
#else
            const bool copy_first = (wr == 0);
#endif
            if (copy_first) { copy_out(pass - 1); VTQ_EPI_T1(copy) convert(pass); VTQ_EPI_T1(conv) }
            else { convert(pass); VTQ_EPI_T1(conv) copy_out(pass - 1); VTQ_EPI_T1(copy) }
            interval_end();
            VTQ_EPI_T1(wait)
        }
        copy_out(NPASS - 1);
        VTQ_EPI_T1(copy)
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
                    const float4 gg = g4[nh][ni];
                    float4 v;
                    if constexpr (BIASED) v = float4{gg.x * a[0], gg.y * a[1], gg.z * a[2], gg.w * a[3]};
                    else {
                        const float4 bb = b4[nh][ni];
                        v = float4{gg.x * (a[0] + bb.x), gg.y * (a[1] + bb.y), gg.z * (a[2] + bb.z), gg.w * (a[3] + bb.w)};
                    }
                    *(float4*)(img + (wr * 16 + fr) * RS + col * 4) = v;
                }
        };
        float* xg = p.x + m0 * p.N + n0 + c16 * 4;
        float4 xv[2][4];
        auto load_x = [&](int ch) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) xv[ch & 1][ps] = *(const float4*)(xg + (int64_t)grow_of(ch, ps) * p.N);
        };
        f32x4 dv[4];
#if VTQ_RESID_DEFER
        // the previous chunk's image rows are requested BEFORE the next chunk's conversion and waited for behind it (as in the balanced two-plane form)
        auto copy_issue = [&](int ch) {
            const char* img = smem + STG + (ch & 1) * IMG;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%c5\n\tds_read_b128 %2, %4 offset:%c6\n\tds_read_b128 %3, %4 offset:%c7"
                         : "=&v"(dv[0]), "=&v"(dv[1]), "=&v"(dv[2]), "=&v"(dv[3])
                         : "v"(lds_addr(img + r0 * RS + c16 * 16)), "i"(8 * RS), "i"(16 * RS), "i"(24 * RS) : "memory");
        };
#endif
        auto copy_out = [&](int ch) {
#if VTQ_RESID_DEFER
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dv[0]), "+v"(dv[1]), "+v"(dv[2]), "+v"(dv[3]) :: "memory");
#else
            const char* img = smem + STG + (ch & 1) * IMG;
            lds_read_rows4<8 * RS>(img + r0 * RS + c16 * 16, dv);
#endif
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {            // image row ps*8 + r0 = (wr = ps>>1, fr = (ps&1)*8 + r0)
                float4 x = xv[ch & 1][ps];
                x.x += dv[ps][0]; x.y += dv[ps][1]; x.z += dv[ps][2]; x.w += dv[ps][3];
                *(float4*)(xg + (int64_t)grow_of(ch, ps) * p.N) = x;
#ifdef VTQ_RESID_PLANES
                // Pricing build (tools/ln_fold_price.py): what the producer side of a LayerNorm fold would add to this epilogue -- the new
                // residual row also as the consumer's hi / lo operand planes, and the (mean, M2) of its 256 columns for a Chan combination.
                if (p.out) {
                    typedef typename Vec<T>::x4 tx4;
                    const int64_t gr = m0 + grow_of(ch, ps);
                    T* pg = (T*)p.out + gr * p.ldo + n0 + c16 * 4;
                    const float xs[4] = {x.x, x.y, x.z, x.w};
                    tx4 h, l;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { T a, b; split2<T>(xs[k], a, b); h[k] = a; l[k] = b; }
                    *(tx4*)pg = h;
                    *(tx4*)(pg + p.o_plane) = l;
                    const float mt = wave_sum((x.x + x.y) + (x.z + x.w)) * (1.0f / 256.0f);
                    const float d0 = x.x - mt, d1 = x.y - mt, d2 = x.z - mt, d3 = x.w - mt;
                    const float m2 = wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
                    if (c16 == 0 && p.row_stats) *(float2*)(p.row_stats + (gr * (p.N >> 8) + (n0 >> 8)) * 2) = float2{mt, m2};
                }
#endif
            }
        };
        load_x(0);
        convert(0);
        interval_end();
#pragma unroll
        for (int ch = 1; ch < NCH; ++ch) {
            load_x(ch);
#if VTQ_RESID_DEFER
            copy_issue(ch - 1);
            convert(ch);
            copy_out(ch - 1);
#else
            copy_out(ch - 1);
            convert(ch);
#endif
            interval_end();
        }
#if VTQ_RESID_DEFER
        copy_issue(NCH - 1);
#endif
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

// =====================================================================================================================
// Persistent 2-phase ping-pong kernel.
//
// DMA groups of a FULL tile, in issue order: g(2k) = {A0, W0, W1} of K tile k (GE instructions per wave), g(2k+1) = {A1} of K
// tile k (GO).  Phase A(kt) issues g(2kt+3), phase B(kt) issues g(2kt+4); the K-tile index runs on INTO THE NEXT TILE of this
// workgroup's list (K tile nkt + j of the current tile = K tile j of the next), except g(2 nkt + 2) = {A0, W0, W1} of the next
// tile's K tile 1: its buffer is the epilogue's staging area, so it is issued after the epilogue.
//   phase A(kt) reads A0,W0,W1 (waited for in B(kt-1) / before the loop) and waits for g(2kt+1) (read in B(kt));
//   phase B(kt) reads A1 and waits for g(2kt+2) (read in A(kt+1)); the last B of a tile waits for nothing.
//   WAR: a region is read (and the read retired by lgkmcnt(0)) before its reader passes the phase barrier; the other wave
//        group re-stages it in the NEXT phase, i.e. after that barrier.  RAW: a group is waited for by every wave one phase
//        before the first read, and both wave groups' waits precede the barrier that opens the reading phase.
//   vmcnt counts loads, LDS-DMA and stores together in issue order, so the wait that opens a chained tile allows for the
//   epilogue's VMEM operations (EOPS, all unconditional) behind the groups it needs: stores are never drained for their own
//   sake; the first group issued AFTER the stores (g2) is waited for one K tile later.
//
// Half tiles (128 x 256, rows of A half 0 only) are phase A alone: one group {A0, W0, W1} per K tile on a ring of THREE
// 48 KiB buffers, tile kt+2 staged in phase kt into the buffer read in phase kt-1 (same WAR/RAW argument).  They take no part in
// the cross-tile prefetch: a half tile is entered and left with `chained == false`, so it stages its own K tiles from scratch and the
// tile after it starts through the unchained entry path.  build_schedule puts them at the END of a workgroup's list, or -- the
// epilogue stagger on the odd XCDs -- a single closing half tile at the FRONT; the half -> full transition is safe because the half
// tile's epilogue ends with the lgkmcnt(0) + barrier below (its LDS images are retired) and its stores are OLDER than the next tile's
// first DMA group: the in-order vmcnt of the unchained entry (wait_vm<GO + GE>) retires them together with that group.
template <typename T, int TERMS, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_pp2_kernel(GemmArgs p) {
    typedef typename Vec<T>::x8 tx8;                 // one MFMA operand fragment: 16 bytes (16-bit forms) / 32 bytes (fp8)
    constexpr bool F8 = std::is_same<T, f8>::value;
    static_assert(!F8 || TERMS == 1, "fp8 operands are single-plane");
    constexpr int ES = F8 ? 1 : 2;                   // bytes per operand element
    constexpr int BK = F8 ? 128 : ((TERMS == 1) ? 64 : 32);
    constexpr int ROWB = BK * ES;                    // bytes of a K-tile row: 128 (one plane, fp8 too) | 64
    constexpr int APL = (TERMS == 1) ? 1 : 2;        // activation planes (also the planes of a 16-bit output)
    constexpr int REG_B = 16384;                     // one half-tile region (A: all planes; W with one plane of BK = 32 fills half)
    constexpr int BUF_B = 4 * REG_B;                 // one K tile: regions 0 = A half 0, 1 = W half 0, 2 = W half 1, 3 = A half 1
    constexpr int NA = 2, NW = (TERMS == 2) ? 1 : 2; // LDS-DMA instructions per wave per A / W region
    constexpr int GE = NA + 2 * NW, GO = NA;
    constexpr int NFA = F8 ? 4 : 8, NFB = (F8 || TERMS == 2) ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntn = p.N / 256;
    const int nkt = p.K / BK;
    // Per-lane addressing state.  It is RE-DERIVED at the top of every tile (tile_setup) from an opaque copy of the thread id,
    // so that none of it stays live across the epilogue, whose accumulators + staging values already fill the register file.
    int tid, fr, fq;
    uint32_t a_off_, w_off_;         // DMA source byte offset of this thread inside a half-tile plane (first of its two 16-byte pieces)
    int a_rd[2], b_rd[2];            // fragment read offsets inside a region: x = k-step (TERMS 1) or plane (TERMS 2, 3)
    auto tile_setup = [&]() {
        tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        fr = lane & 15;
        fq = lane >> 4;
        if constexpr (TERMS == 1) {          // piece r = rows r*64 + (tid >> 3): the second piece is 64 rows further (same swizzle)
            const int row = tid >> 3, c = (tid & 7) ^ swz<64>(row);
            a_off_ = (uint32_t)(row * p.lda * ES + c * 16);
            w_off_ = (uint32_t)(row * p.K * ES + c * 16);
        } else {                             // piece r = plane r: same offset, the plane stride goes on the scalar base
            const int row = tid >> 2, c = (tid & 3) ^ swz<32>(row);
            a_off_ = (uint32_t)(row * p.lda + c * 8) * 2u;
            w_off_ = (uint32_t)(row * p.K + c * 8) * 2u;
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            if constexpr (TERMS == 1) {
                // 16-bit: x = k-step (lane group fq takes chunk 4x + fq); fp8: the two chunks of the lane's 32 bytes (2 fq + x)
                const int ch = ((F8 ? (2 * fq + x) : (x * 4 + fq)) ^ swz<64>(fr)) << 4;
                a_rd[x] = (wr * 64 + fr) * ROWB + ch;
                b_rd[x] = (wc * 32 + fr) * ROWB + ch;
            } else {
                const int ch = (fq ^ swz<32>(fr)) << 4;
                a_rd[x] = x * 8192 + (wr * 64 + fr) * ROWB + ch;
                b_rd[x] = x * 8192 + (wc * 32 + fr) * ROWB + ch;
            }
        }
    };
    // byte distance (wave-uniform) from a thread's first DMA piece to its second: 64 rows on (one plane) or one plane on
    const int64_t a_d2 = (TERMS == 1) ? (int64_t)64 * p.lda * ES : p.a_plane * 2;
    const int64_t w_d2 = (TERMS == 1) ? (int64_t)64 * p.K * ES : ((TERMS == 3) ? p.w_plane * 2 : 0);
    f32x4 acc[2][2][4][2];
    tx8 fa[NFA], fb0[NFB], fb1[NFB];
    // The accumulators start at the BIAS instead of at zero (16-bit forms with a staged epilogue): the bias vector lives in LDS for
    // the whole persistent launch (kBiasLds .. + 4 N bytes, above the ring and the epilogue images) and every accumulator block
    // is initialised by one ds_read_b128 of its four columns.  That removes two vector instructions per output from the
    // vector-issue-bound epilogue -- the v_mov that zeroed the register and the bias add (profiles/r03_gemm_epilogue_stamps.txt).
    constexpr bool kBiasInAcc = !F8 && EPI != EPI_EMBED;
    constexpr int kBiasLds = 147456;
    if constexpr (kBiasInAcc) {
        for (int i = threadIdx.x * 4; i < p.N; i += 2048) *(float4*)(smem + kBiasLds + i * 4) = *(const float4*)(p.bias + i);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    auto init_acc = [&](int n0) {
        int boff = kBiasLds + (n0 + wc * 32 + fq * 4) * 4;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        asm volatile("" : "+v"(boff));          // a load of its own for every block: a copy would be a vector instruction again
                        acc[a][b][i][j] = *(const f32x4*)(smem + boff + (b * 128 + j * 16) * 4);
                    }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    typedef __attribute__((ext_vector_type(4))) int v4i;
    auto read32 = [&](const char* a0, const char* a1) {      // fp8 fragment: two 16-byte chunks of one row
        const v4i lo = *(const v4i*)a0, hi = *(const v4i*)a1;
        return v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto read_a = [&](const char* reg) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (F8) fa[i] = read32(reg + a_rd[0] + i * 16 * ROWB, reg + (a_rd[0] ^ 16) + i * 16 * ROWB);   // chunks 2fq, 2fq+1: the XOR swizzle keeps them 16 B apart
            else {
#pragma unroll
                for (int x = 0; x < 2; ++x) fa[i * 2 + x] = *(const tx8*)(reg + a_rd[x] + i * 16 * ROWB);
            }
        }
    };
    auto read_b = [&](const char* reg, tx8(&fb)[NFB]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (F8) fb[j] = read32(reg + b_rd[0] + j * 16 * ROWB, reg + (b_rd[0] ^ 16) + j * 16 * ROWB);
            else if constexpr (TERMS == 2) fb[j] = *(const tx8*)(reg + b_rd[0] + j * 16 * ROWB);
            else {
#pragma unroll
                for (int x = 0; x < 2; ++x) fb[j * 2 + x] = *(const tx8*)(reg + b_rd[x] + j * 16 * ROWB);
            }
        }
    };
    auto mma = [&](f32x4(&c)[4][2], const tx8(&fb)[NFB]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (F8) {
                    c[i][j] = mfma_f8(fb[j], fa[i], c[i][j]);
                } else if constexpr (TERMS == 1) {
                    c[i][j] = mfma16<T>(fb[j * 2 + 0], fa[i * 2 + 0], c[i][j]);
                    c[i][j] = mfma16<T>(fb[j * 2 + 1], fa[i * 2 + 1], c[i][j]);
                } else if constexpr (TERMS == 2) {
                    c[i][j] = mfma16<T>(fb[j], fa[i * 2 + 0], c[i][j]);
                    c[i][j] = mfma16<T>(fb[j], fa[i * 2 + 1], c[i][j]);
                } else {
                    c[i][j] = mfma16<T>(fb[j * 2 + 0], fa[i * 2 + 0], c[i][j]);
                    c[i][j] = mfma16<T>(fb[j * 2 + 0], fa[i * 2 + 1], c[i][j]);
                    c[i][j] = mfma16<T>(fb[j * 2 + 1], fa[i * 2 + 0], c[i][j]);
                }
            }
    };
    // stage region r (0 = A half 0, 1 = W half 0, 2 = W half 1, 3 = A half 1) of K tile kt of the tile with bases (ag, wg)
    auto stage = [&](const T* ag, const T* wg, int kt, int r) {
        char* dst = smem + (kt & 1) * BUF_B + r * REG_B + wave * 1024;
        const bool isA = (r == 0 || r == 3);
        const int half = (r >= 2) ? 1 : 0;
        // the per-lane part stays a 32-bit offset next to a scalar base (opaque here: otherwise hipcc pre-adds it to every
        // tile base as 64-bit VGPR pairs that live -- or spill -- across the K loop)
        uint32_t a_off = a_off_, w_off = w_off_;
        asm volatile("" : "+v"(a_off), "+v"(w_off));
        if (isA) {
            const char* base = (const char*)(ag + (int64_t)half * kHalfRows * p.lda + kt * BK);
            glds16(base + a_off, dst);
            glds16(base + a_d2 + a_off, dst + 8192);
        } else {
            const char* base = (const char*)(wg + (int64_t)half * kHalfRows * p.K + kt * BK);
            glds16(base + w_off, dst);
            if constexpr (NW == 2) glds16(base + w_d2 + w_off, dst + 8192);
        }
    };

#define VTQ_SYNC_OPEN_WAITED()                                 \
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

    // output form: the operand type's planes; fp8 operands: BIAS -> fp16 hi/lo planes (attention input), GELU -> e4m3 bytes
    typedef typename std::conditional<F8, typename std::conditional<EPI == EPI_BIAS_GELU, f8, f16>::type, T>::type TO;
    constexpr int OPL = F8 ? 1 : APL;      // fp8 mode: one fp16 plane (QKV -> single-MFMA attention) or e4m3 bytes (GELU)
    constexpr int EOPS = epilogue_ops_of<TO, OPL, EPI, 2>();
    constexpr int NCHAIN = (EOPS + GE > 63) ? 63 : EOPS + GE;   // youngest ops that may stay in flight while g0/g1 of a chained tile are waited for
    constexpr bool kChain = (EPI != EPI_EMBED);                 // cross-tile DMA chaining (EMBED's stores are conditional: not countable)
    // the schedule is read through the scalar cache (constant address space): a vector load here would make hipcc drain
    // vmcnt -- i.e. the previous tile's epilogue stores -- at the top of every tile
    const __attribute__((address_space(4))) int* sch = (const __attribute__((address_space(4))) int*)p.sched;
    const int it_beg = sch[blockIdx.x], it_end = sch[blockIdx.x + 1];
    EpiDiag epi_diag{};
#ifdef VTQ_GEMM_DIAG
    // Diagnostic build only (MI355X_MICROARCH.md 'DVFS give-back' item 6): shader-clock and 100 MHz real-time stamps around every
    // K loop and around the whole kernel, summed in scalar registers and written to a buffer nothing else reads.
    unsigned long long dg_lt = 0, dg_lr = 0, dg_t0, dg_r0, dg_kt0, dg_kr0;
    auto stamp = [&](unsigned long long& t, unsigned long long& r) {
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r) :: "memory");
    };
    stamp(dg_kt0, dg_kr0);
    float dg_v[4] = {1.0f, 1.5f, 2.0f, 2.5f};
    auto shadow_valu = [&]() {           // p.shadow x 8 independent FMAs in the load phase (beside the partner wave's MFMA cluster)
        for (int i = 0; i < p.shadow; ++i)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %0\n\tv_fma_f32 %3, %3, %0, %1\n\t"
                         "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %0\n\tv_fma_f32 %3, %3, %0, %1"
                         : "+v"(dg_v[0]), "+v"(dg_v[1]), "+v"(dg_v[2]), "+v"(dg_v[3]));
    };
#define VTQ_DIAG_SHADOW() shadow_valu();
#define VTQ_DIAG_LOOP_BEGIN() stamp(dg_t0, dg_r0);
#define VTQ_DIAG_LOOP_END() { unsigned long long t1, r1; stamp(t1, r1); dg_lt += t1 - dg_t0; dg_lr += r1 - dg_r0; }
#else
#define VTQ_DIAG_SHADOW()
#define VTQ_DIAG_LOOP_BEGIN()
#define VTQ_DIAG_LOOP_END()
#endif
    bool chained = false;       // this tile's K tile 0 (both groups) was staged by the previous tile; K tile 1's g2 is still to issue
    for (int it = it_beg; it < it_end; ++it) {
        const int d = sch[it];
        const int tile = d >> 2, kind = d & 3;
        const int tm = tile / ntn, tn = tile - tm * ntn;
        const int64_t m0 = (int64_t)tm * 256 + (kind == 2 ? kHalfRows : 0);
        const int n0 = tn * 256;
        const bool wrapl = (gemm_flags(p) & GEMM_FLAG_WRAP_LOADS) != 0;     // measurement: every tile reads the first two A / W panels (L2 hits)
        const T* __restrict__ Ag = (const T*)p.A + (wrapl ? (m0 & 511) : m0) * p.lda;
        const T* __restrict__ Wg = (const T*)p.W + (int64_t)(wrapl ? (n0 & 511) : n0) * p.K;
        // the next entry of this workgroup's list, if the DMA ring may run on into it
        const T* Agn = Ag;
        const T* Wgn = Wg;
        bool has_next = false;
        if constexpr (kChain) {
            if (kind == 0 && it + 1 < it_end && !(gemm_flags(p) & GEMM_FLAG_NO_CHAIN)) {
                const int dn = sch[it + 1];
                if ((dn & 3) == 0) {
                    const int tmn = (dn >> 2) / ntn, tnn = (dn >> 2) - tmn * ntn;
                    Agn = (const T*)p.A + (int64_t)(wrapl ? (tmn & 1) : tmn) * 256 * p.lda;
                    Wgn = (const T*)p.W + (int64_t)(wrapl ? (tnn & 1) : tnn) * 256 * p.K;
                    has_next = true;
                }
            }
        }
        tile_setup();
        if constexpr (!kBiasInAcc) zero_acc();
        if (kind == 0) {
            // ---- entry: groups 0, 1, 2 in flight, group 0 landed for everyone ------------------------------------------
            if (!chained) {
                stage(Ag, Wg, 0, 0); stage(Ag, Wg, 0, 1); stage(Ag, Wg, 0, 2); stage(Ag, Wg, 0, 3);
                if (nkt > 1) { stage(Ag, Wg, 1, 0); stage(Ag, Wg, 1, 1); stage(Ag, Wg, 1, 2); wait_vm<GO + GE>(); }
                else wait_vm<GO>();
            } else {
                // g0, g1 were issued during the previous tile's last K tile; its epilogue stores came after them; g2 now
                stage(Ag, Wg, 1, 0); stage(Ag, Wg, 1, 1); stage(Ag, Wg, 1, 2);
                if constexpr (kChain) wait_vm<NCHAIN>();             // g0 AND g1 landed: they are older than every epilogue op
            }
            __builtin_amdgcn_s_barrier();
            if (wr == 1) __builtin_amdgcn_s_barrier();        // second wave group runs one barrier behind
            if constexpr (kBiasInAcc) init_acc(n0);           // here, not at the top of the tile: the accumulators are dead until now

            VTQ_DIAG_LOOP_BEGIN()
            for (int kt = 0; kt < nkt; ++kt) {
                const char* buf = smem + (kt & 1) * BUF_B;
                // ---- phase A --------------------------------------------------------------------------------------------
                read_b(buf + 1 * REG_B, fb0);
                read_a(buf + 0 * REG_B);
                read_b(buf + 2 * REG_B, fb1);
                if (kt + 1 < nkt) {
                    stage(Ag, Wg, kt + 1, 3);
                    if (kt == 0 && chained) { if constexpr (kChain) wait_vm<63>(); }   // g1 was covered at entry; do not drain the stores
                    else wait_vm<GE + GO>();
                } else if (has_next) {
                    stage(Agn, Wgn, 0, 3);                    // next tile's g1 (nkt is even: its K tile 0 lives in buffer 0)
                    wait_vm<GE + GO>();
                } else {
                    wait_vm<0>();
                }
                VTQ_DIAG_SHADOW()
                VTQ_SYNC_OPEN_WAITED()
                mma(acc[0][0], fb0);
                mma(acc[0][1], fb1);
                VTQ_SYNC_CLOSE()
                // ---- phase B --------------------------------------------------------------------------------------------
                read_a(buf + 3 * REG_B);
                if (kt + 2 < nkt) {
                    stage(Ag, Wg, kt + 2, 0); stage(Ag, Wg, kt + 2, 1); stage(Ag, Wg, kt + 2, 2);
                    wait_vm<GO + GE>();
                } else if (kt + 2 == nkt) {
                    if (has_next) { stage(Agn, Wgn, 0, 0); stage(Agn, Wgn, 0, 1); stage(Agn, Wgn, 0, 2); wait_vm<GO + GE>(); }
                    else wait_vm<GO>();
                }                                             // last K tile: nothing of this tile is left to land
                VTQ_DIAG_SHADOW()
                VTQ_SYNC_OPEN_WAITED()
                mma(acc[1][1], fb1);
                mma(acc[1][0], fb0);
                VTQ_SYNC_CLOSE()
            }
            VTQ_DIAG_LOOP_END()
            if (wr == 0) __builtin_amdgcn_s_barrier();        // match the extra barrier of the second group
            if (!(gemm_flags(p) & GEMM_FLAG_NO_EPILOGUE))
                pp_epilogue<TO, OPL, EPI, 2, F8, kBiasInAcc>(p, acc, smem, tid, wr, wc, fr, fq, (gemm_flags(p) & GEMM_FLAG_WRAP_ROWS) ? 0 : m0, n0, epi_diag);
            else {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[a][b][i][j]));
            }
            chained = has_next;
            if (it + 1 < it_end) {                            // the staging images are re-staged / re-used next: retire their reads
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        } else {
            constexpr int BUF_H = 3 * REG_B;
            constexpr int GH = NA + 2 * NW;
            auto stage_h = [&](int kt, int slot) {             // {A0, W0, W1} of K tile kt into ring slot
                uint32_t a_off = a_off_, w_off = w_off_;
                asm volatile("" : "+v"(a_off), "+v"(w_off));
                char* dst = smem + slot * BUF_H + wave * 1024;
                const char* ab = (const char*)(Ag + kt * BK);
                const char* w0 = (const char*)(Wg + kt * BK);
                const char* w1 = (const char*)(Wg + (int64_t)kHalfRows * p.K + kt * BK);
                glds16(ab + a_off, dst);
                glds16(ab + a_d2 + a_off, dst + 8192);
                glds16(w0 + w_off, dst + REG_B);
                if constexpr (NW == 2) glds16(w0 + w_d2 + w_off, dst + REG_B + 8192);
                glds16(w1 + w_off, dst + 2 * REG_B);
                if constexpr (NW == 2) glds16(w1 + w_d2 + w_off, dst + 2 * REG_B + 8192);
            };
            stage_h(0, 0);
            if (nkt > 1) { stage_h(1, 1); wait_vm<GH>(); }
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            if (wr == 1) __builtin_amdgcn_s_barrier();
            if constexpr (kBiasInAcc) init_acc(n0);
            int slot = 0;                                      // kt % 3
            VTQ_DIAG_LOOP_BEGIN()
            for (int kt = 0; kt < nkt; ++kt) {
                const char* buf = smem + slot * BUF_H;
                read_b(buf + 1 * REG_B, fb0);
                read_a(buf + 0 * REG_B);
                read_b(buf + 2 * REG_B, fb1);
                const int nslot = slot == 0 ? 2 : slot - 1;    // (kt + 2) % 3
                if (kt + 2 < nkt) { stage_h(kt + 2, nslot); wait_vm<GH>(); }
                else wait_vm<0>();
                VTQ_DIAG_SHADOW()
                VTQ_SYNC_OPEN_WAITED()
                mma(acc[0][0], fb0);
                mma(acc[0][1], fb1);
                VTQ_SYNC_CLOSE()
                slot = slot == 2 ? 0 : slot + 1;
            }
            VTQ_DIAG_LOOP_END()
            if (wr == 0) __builtin_amdgcn_s_barrier();
            pp_epilogue<TO, OPL, EPI, 1, F8, kBiasInAcc>(p, acc, smem, tid, wr, wc, fr, fq, (gemm_flags(p) & GEMM_FLAG_WRAP_ROWS) ? 0 : m0, n0, epi_diag);
            chained = false;
            if (it + 1 < it_end) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }
#ifdef VTQ_GEMM_DIAG
    {
        unsigned long long t1, r1;
        stamp(t1, r1);
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("" ::"v"(dg_v[0]), "v"(dg_v[1]), "v"(dg_v[2]), "v"(dg_v[3]));
        if (p.diag && (threadIdx.x & 63) == 0) {
            unsigned long long* d = p.diag + (size_t)blockIdx.x * 64;
            if (threadIdx.x == 0) { d[0] = dg_lt; d[1] = dg_lr; d[2] = t1 - dg_kt0; d[3] = r1 - dg_kr0; d[4] = (unsigned long long)(it_end - it_beg); d[5] = xcc & 0xf; }
            unsigned long long* w = d + 8 + (threadIdx.x >> 6) * 4;          // per wave: cycles in convert / copy-out / interval-end waits
            w[0] = epi_diag.conv; w[1] = epi_diag.copy; w[2] = epi_diag.wait;
        }
    }
#endif
#undef VTQ_DIAG_SHADOW
#undef VTQ_DIAG_LOOP_BEGIN
#undef VTQ_DIAG_LOOP_END
#undef VTQ_SYNC_OPEN_WAITED
#undef VTQ_SYNC_CLOSE
}

// =====================================================================================================================
// Tile schedule (host).  The launch is PERSISTENT: kNumCus workgroups, dealt round-robin over the 8 XCDs by the dispatcher
// (workgroup b lands on XCD b % 8, one per CU: 144 KiB of LDS each), every workgroup walks its own list.
//   * XCD x owns a contiguous run of the row-major tile order (its A row panels stay in its L2).
//   * Inside the run the order is column-group-major: (column group of cg tiles) > row panel > column.  The XCD's 32
//     workgroups take consecutive entries, so at any time they work on 32/cg row panels x cg column tiles, and the cg W tiles
//     (cg * 256 * K * planes * 2 B, sized to ~2.5 MiB) stay L2-resident while the panels stream past.
//   * The run ENDS with 128x256 half tiles, so the last round is filled in half-tile granules; how many tiles to split is
//     chosen by simulating the greedy list assignment (half tile = 0.57 of a tile, measured).  On the odd XCDs a workgroup whose
//     list closes with ONE half tile runs it first instead (stagger, below).
//   * Entries are dealt to the XCD's workgroups in order, each to the least loaded one (what a dynamic queue would do with
//     these costs).  Output never depends on the schedule: every element is accumulated over K in the same order in both
//     tile forms, so results are bitwise independent of batch size and placement.
constexpr int kXcds = 8, kCusPerXcd = 32, kNumCus = kXcds * kCusPerXcd;
constexpr double kHalfCost = 0.57;

// Workgroups per XCD of the persistent launch: all 32 CUs, unless a measurement narrows the grid -- vtq_debug_gemm_cus (a launch on a
// CU-masked stream, tools/cu_partition.py: the grid must match the CUs the stream owns, the schedule is rebuilt for it) or, in
// -DVTQ_MEASURE builds, VTQ_GEMM_CUS.  Results never depend on it (the tile -> workgroup assignment does not enter the arithmetic).
std::atomic<int> g_cus_per_xcd{0};
int cus_per_xcd() {
    static const int env = [] { const char* v = VTQ_MEASURE_ENV("VTQ_GEMM_CUS"); const int k = v ? atoi(v) : kCusPerXcd; return (k >= 1 && k <= kCusPerXcd) ? k : kCusPerXcd; }();
    const int g = g_cus_per_xcd.load(std::memory_order_relaxed);
    return (g >= 1 && g <= kCusPerXcd) ? g : env;
}

double greedy_makespan(int n_full, int n_half) {
    double cu[kCusPerXcd] = {0};
    const int ncu = cus_per_xcd();
    auto put = [&](double d) {
        int best = 0;
        for (int i = 1; i < ncu; ++i) if (cu[i] < cu[best]) best = i;
        cu[best] += d;
    };
    for (int i = 0; i < n_full; ++i) put(1.0);
    for (int i = 0; i < n_half; ++i) put(kHalfCost);
    double m = 0;
    for (double v : cu) m = v > m ? v : m;
    return m;
}

// entries of one XCD owning `order[0..cnt)`: whole tiles, then `tail` tiles as top/bottom halves
void xcd_sequence(const int* order, int cnt, std::vector<int>& out) {
    static const bool all_halves = [] { const char* v = VTQ_MEASURE_ENV("VTQ_GEMM_SCHED"); return v && v[0] == '2'; }();   // measurement knob
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

// -> [offsets (kNumCus + 1 entries, absolute indices into the same array)] [entries: (tile << 2) | kind]
std::vector<int> build_schedule(int ntm, int ntn, int cg) {
    const int nt = ntm * ntn, q = nt / kXcds, r = nt % kXcds;
    if (cg < 1) cg = 1;
    std::vector<std::vector<int>> lists(kNumCus);
    for (int x = 0; x < kXcds; ++x) {
        const int t0 = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q, cnt = q + (x < r ? 1 : 0);
        std::vector<std::tuple<int, int, int, int>> keyed;       // (column group, row panel, column, tile)
        keyed.reserve(cnt);
        for (int t = t0; t < t0 + cnt; ++t) keyed.emplace_back((t % ntn) / cg, t / ntn, t % ntn, t);
        std::sort(keyed.begin(), keyed.end());
        std::vector<int> order(cnt), seq;
        for (int i = 0; i < cnt; ++i) order[i] = std::get<3>(keyed[i]);
        xcd_sequence(order.data(), cnt, seq);
        double load[kCusPerXcd] = {0};
        const int ncu = cus_per_xcd();
        for (int e : seq) {
            int best = 0;
            for (int i = 1; i < ncu; ++i) if (load[i] < load[best] - 1e-9) best = i;
            load[best] += (e & 3) ? kHalfCost : 1.0;
            lists[x + kXcds * best].push_back(e);
        }
    }
    // Stagger: on the odd XCDs a workgroup runs its closing half tile FIRST.  All tiles of a launch take the same time, so otherwise every
    // workgroup of the chip reaches its epilogue -- a burst of stores and, in the residual form, loads -- at the same moments; with the
    // half tile in front, the odd XCDs' epilogues fall about half a tile after the even ones'.  Whole XCDs, not every second CU of one:
    // the column tiles of a row panel that share their A panel through the XCD's L2 stay in step (per-CU staggering cost fc2 3.5 %).
    // Same work, same results (profiles/r03_gemm_stagger.txt: out-proj -6 %, QKV -3 %, fc2 and fc1 unchanged; -DVTQ_MEASURE builds: VTQ_GEMM_STAGGER=0 turns it off).
    static const bool stagger = [] { const char* v = VTQ_MEASURE_ENV("VTQ_GEMM_STAGGER"); return !(v && v[0] == '0'); }();
    if (stagger)
        for (int b = 0; b < kNumCus; ++b) {
            std::vector<int>& l = lists[b];
            if (((b % kXcds) & 1) && l.size() >= 2 && (l.back() & 3) && !(l[l.size() - 2] & 3)) { const int h = l.back(); l.pop_back(); l.insert(l.begin(), h); }
        }
    std::vector<int> out(kNumCus + 1);
    int off = kNumCus + 1;
    for (int b = 0; b < kNumCus; ++b) { out[b] = off; off += (int)lists[b].size(); }
    out[kNumCus] = off;
    for (int b = 0; b < kNumCus; ++b) out.insert(out.end(), lists[b].begin(), lists[b].end());
    return out;
}

// Measurement form (GEMM_FLAG_DYNAMIC): the same per-XCD sequences, but one entry per workgroup in hardware dispatch order
// (block b -> XCD b % 8): offsets[b] = first entry of block b, one entry each.
std::vector<int> build_schedule_dynamic(int ntm, int ntn, int cg) {
    const int nt = ntm * ntn, q = nt / kXcds, r = nt % kXcds;
    if (cg < 1) cg = 1;
    std::vector<std::vector<int>> seq(kXcds);
    size_t total = 0;
    for (int x = 0; x < kXcds; ++x) {
        const int t0 = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q, cnt = q + (x < r ? 1 : 0);
        std::vector<std::tuple<int, int, int, int>> keyed;
        for (int t = t0; t < t0 + cnt; ++t) keyed.emplace_back((t % ntn) / cg, t / ntn, t % ntn, t);
        std::sort(keyed.begin(), keyed.end());
        std::vector<int> order(cnt);
        for (int i = 0; i < cnt; ++i) order[i] = std::get<3>(keyed[i]);
        xcd_sequence(order.data(), cnt, seq[x]);
        total += seq[x].size();
    }
    std::vector<int> ent;
    for (size_t j = 0; ent.size() < total; ++j)
        for (int x = 0; x < kXcds; ++x)
            if (j < seq[x].size()) ent.push_back(seq[x][j]);
    std::vector<int> out(total + 1);
    for (size_t b = 0; b <= total; ++b) out[b] = (int)(total + 1 + b);
    out.insert(out.end(), ent.begin(), ent.end());
    return out;
}

int column_group(int ntn, int K, int wpl) {
    const double tile_bytes = 256.0 * K * 2.0 * wpl;
    int cg = (int)(2.5 * 1048576.0 / tile_bytes);
    if (cg < 1) cg = 1;
    if (cg >= ntn || ntn <= 4) cg = ntn;
    return cg;
}

// Device-resident schedules, per DEVICE (a second engine on another GPU of the same process gets its own copies) and per
// (tile grid, column group); entries are a few KiB and live until exit.
struct DevSched { int* dev; int nwg; };
hipError_t schedule_for(int ntm, int ntn, int cg, bool dynamic, DevSched& ds, hipStream_t s) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int, int>, DevSched> cache;
    static std::vector<std::vector<int>*> staged;              // host images of uploads in flight on some stream: kept for the process lifetime
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_tuple(dev, ntm, ntn, cg, (int)dynamic, cus_per_xcd());
    auto it = cache.find(key);
    if (it != cache.end()) { ds = it->second; return hipSuccess; }
    // First use of a shape on this device.  The engine does this from vtq_reserve / the top of vtq_forward (gemm_prepare), never
    // between the launches of a forward; the per-kernel test entry points may land here from a launch.  The upload is an async
    // copy on the launch stream (ordered before the kernel that reads it) from a host image that stays alive.
    std::vector<int>* h = new std::vector<int>(dynamic ? build_schedule_dynamic(ntm, ntn, cg) : build_schedule(ntm, ntn, cg));
    DevSched d{nullptr, dynamic ? (*h)[0] - 1 : kXcds * cus_per_xcd()};
    e = hipMalloc(&d.dev, h->size() * sizeof(int));
    if (e != hipSuccess) { delete h; return e; }
    e = hipMemcpyAsync(d.dev, h->data(), h->size() * sizeof(int), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) { (void)hipFree(d.dev); delete h; return e; }
    staged.push_back(h);
    cache[key] = d;
    ds = d;
    return hipSuccess;
}

int env_cg() {
    static const int v = [] { const char* c = VTQ_MEASURE_ENV("VTQ_GEMM_CG"); return c ? atoi(c) : 0; }();    // measurement knob: column-group width of the tile order
    return v;
}

unsigned long long* g_diag_buf = nullptr;      // diagnostic builds: stamp buffer and shadow-VALU count (gemm_set_diag)
int g_diag_shadow = 0;

int env_flags() {
    static const int f = [] { const char* v = VTQ_MEASURE_ENV("VTQ_GEMM_FLAGS"); return v ? atoi(v) : 0; }();   // measurement knobs (kernels.h)
    return f;
}

template <typename T, int TERMS, int EPI> hipError_t launch_t(GemmArgs a, hipStream_t s) {
    constexpr int LDS = 163840;                    // full tiles: DMA ring 128 KiB, epilogue images 256x528 B; half tiles: 3 x 48 KiB; bias vector (N <= 4096) above 144 KiB
    static std::mutex mu;
    static bool configured[64] = {false};          // hipFuncSetAttribute is per device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!configured[dev]) {
            e = hipFuncSetAttribute((const void*)gemm_pp2_kernel<T, TERMS, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) return e;
            configured[dev] = true;
        }
    }
    constexpr int WPL = (TERMS == 3) ? 2 : 1;
    DevSched ds;
    a.flags = env_flags();
    const int cg = env_cg() > 0 ? env_cg() : column_group(a.N / 256, a.K, WPL);
    e = schedule_for(a.M / 256, a.N / 256, cg, (a.flags & GEMM_FLAG_DYNAMIC) != 0, ds, s);
    if (e != hipSuccess) return e;
    a.sched = ds.dev;
    a.diag = g_diag_buf;
    a.shadow = g_diag_shadow;
#ifdef VTQ_RESID_PLANES
    {   // pricing build: a statistics buffer nothing reads (one per process, sized for the largest M x N / 256 asked for so far)
        static float* stats = nullptr;
        static size_t cap = 0;
        const size_t need = (size_t)a.M * (a.N / 256) * 2 * sizeof(float);
        if (need > cap) { if (stats) (void)hipFree(stats); if (hipMalloc((void**)&stats, need) != hipSuccess) return hipErrorOutOfMemory; cap = need; }
        a.row_stats = stats;
    }
#endif
    hipLaunchKernelGGL((gemm_pp2_kernel<T, TERMS, EPI>), dim3(ds.nwg), dim3(512), LDS, s, a);
    return hipGetLastError();
}

template <typename T, int TERMS> hipError_t launch_e(const GemmArgs& a, int epilogue, hipStream_t s) {
    switch (epilogue) {
        case EPI_BIAS: return launch_t<T, TERMS, EPI_BIAS>(a, s);
        case EPI_BIAS_GELU: return launch_t<T, TERMS, EPI_BIAS_GELU>(a, s);
        case EPI_RESID: return launch_t<T, TERMS, EPI_RESID>(a, s);
        case EPI_EMBED: return launch_t<T, TERMS, EPI_EMBED>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

void gemm_set_diag(unsigned long long* buf, int shadow) { g_diag_buf = buf; g_diag_shadow = shadow; }
unsigned long long* gemm_diag_buffer() { return g_diag_buf; }
bool gemm_is_diag_build() {
#ifdef VTQ_GEMM_DIAG
    return true;
#else
    return false;
#endif
}

hipError_t gemm_prepare(int M, int N, int K, int wpl, hipStream_t s) {
    if (M <= 0 || M % 256 || N <= 0 || N % 256) return hipErrorInvalidValue;
    DevSched ds;
    const int cg = env_cg() > 0 ? env_cg() : column_group(N / 256, K, wpl);
    return schedule_for(M / 256, N / 256, cg, (env_flags() & GEMM_FLAG_DYNAMIC) != 0, ds, s);
}

std::vector<int> gemm_tile_schedule(int ntm, int ntn, int K, int wpl) {
    return build_schedule(ntm, ntn, column_group(ntn, K, wpl));
}

namespace { std::atomic<int> g_tile_variant{GEMM_TILE_AUTO}; }
void gemm_set_variant(int v) { g_tile_variant.store(v, std::memory_order_relaxed); }
void gemm_set_cus_per_xcd(int n) { g_cus_per_xcd.store(n, std::memory_order_relaxed); }

// Which tile shape serves an (M, N, K) launch (profiles/r05_gemm_tile_shapes.txt, fp16x3 on the encoder's four GEMMs at the row counts of
// B = 1 .. 32 pairs).  The persistent 256x256 kernel wins from 64 of its tiles up (it is MFMA-bound per tile, the small tiles are bound by
// a CU's L1 -> LDS fill rate: 4x the operand bytes per flop at 64x64); below that most CUs would idle, and one workgroup per small tile
// is faster: 64x64 tiles while they fit one or two co-resident workgroups per CU, 128x128 beyond.  Pure speed choice (bitwise contract
// of gemm_st.hip); fp8 operands have the 256x256 kernel only.
int gemm_tile_rule(int M, int N, int K, Num num, int cus) {
    if (cus < 1) cus = kNumCus;
    if (num.f16 == 2 || M <= 0 || M % 256 || N <= 0 || N % 256 || K % 64) return GEMM_TILE_256;
    const int t256 = (M / 256) * (N / 256);
    if (t256 >= 64) return GEMM_TILE_256;
    const int n64 = 16 * t256;
    if (n64 <= cus) return GEMM_ST_64;                     // one workgroup per CU, ring of 3
    if (n64 <= 2 * cus) return GEMM_ST_64X2;               // two co-resident workgroups per CU, ring of 2
    return GEMM_ST_128;
}

hipError_t launch_gemm(const GemmArgs& a, Num num, int epilogue, hipStream_t s) {
    if (num_valid(num) && num.f16 != 2) {
        const int forced = g_tile_variant.load(std::memory_order_relaxed);
        int cus = 0;
        if (forced == GEMM_TILE_AUTO && device_cus(&cus)) return hipErrorInvalidDevice;      // the CU count of the CURRENT device (a partitioned part has fewer)
        const int v = forced != GEMM_TILE_AUTO ? forced : gemm_tile_rule(a.M, a.N, a.K, num, cus);
        if (v != GEMM_TILE_256) return launch_gemm_st(a, num, epilogue, v, s);
    }
    const int bk2 = (num.f16 == 2) ? 256 : ((num.terms == 1) ? 128 : 64);     // two K tiles: the DMA ring's buffer parity is fixed across tiles
    if (a.M <= 0 || a.M % 256 || a.N % 256 || a.N > 4096 || a.K <= 0 || a.K % bk2 || a.lda % 16 || !num_valid(num)) return hipErrorInvalidValue;
    if (num.f16 == 2) {
#ifdef VTQ_WITH_FP8                                   // the fp8 experiment (include/vtamiq_hip_fp8.h): not in the product library
        if (!a.wscale) return hipErrorInvalidValue;
        return launch_e<f8, 1>(a, epilogue, s);
#else
        return hipErrorNotSupported;
#endif
    }
    if (!num.f16) {
        if (num.terms == 1) return launch_e<bf16, 1>(a, epilogue, s);
        if (num.terms == 3) return launch_e<bf16, 3>(a, epilogue, s);
    } else {
        if (num.terms == 1) return launch_e<f16, 1>(a, epilogue, s);
        if (num.terms == 2) return launch_e<f16, 2>(a, epilogue, s);
        if (num.terms == 3) return launch_e<f16, 3>(a, epilogue, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace vtq
