// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  with fused epilogues.
//
// Replaces every torch.nn.Linear / patch Conv2d on the ViT path of the reference
// (modules/VisionTransformer/transformer.py:138-140,154-156,169 [QKV/out], :205-206,212-215 [MLP],
//  :475-480,531-532 [patch embedding as GEMM]).
//
// Design (cdna_hip_programming.md section 5):
//   * 256x256 output tile per 512-thread workgroup (8 waves as 2(M) x 4(N); 128x64 per wave, 32 accumulators
//     of mfma_f32_16x16x32_bf16 = 128 acc VGPRs).
//   * both operands are K-contiguous (torch Linear weight layout), staged global -> LDS with 16-byte LDS-DMA
//     (global_load_lds_dwordx4), double buffered, one barrier per K tile.
//   * LDS image is lane-linear (DMA constraint); bank conflicts of the ds_read_b128 fragment reads are removed
//     by XOR-swizzling the 16-byte chunk index on the SOURCE address and on the read (rule 21).
//   * operands are swapped in the MFMA (W fragment as A-operand) so each lane ends up with 4 consecutive output
//     columns of one row: bias/gamma/residual are float4 accesses and bf16 results are 8-byte stores.
//   * NSPLIT == 3 ("bf16x3"): A and W arrive as hi/lo bf16 planes, each product is hi*hi + hi*lo + lo*hi into the
//     same fp32 accumulator (3 MFMAs per 4 fragment reads); BK is halved to keep the 128 KiB LDS budget.
//   * workgroup id is remapped so that consecutive tiles (sharing an A row panel) land on one XCD's L2 (T1).
#include "dev_common.h"
#include "kernels.h"

namespace vtq {

namespace {

template <int BK> __device__ __forceinline__ int swz(int row) {
    // BK=64: 128-byte rows, 8 chunks; BK=32: 64-byte rows, 4 chunks.  See DESIGN.md "LDS swizzle".
    if constexpr (BK == 64) return (row >> 1) & 7;
    else return ((row >> 3) & 1) * 3;
}

template <int NSPLIT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_kernel(GemmArgs p) {
    constexpr int BM = 256, BN = 256;
    constexpr int BK = (NSPLIT == 1) ? 64 : 32;
    constexpr int NPL = (NSPLIT == 1) ? 1 : 2;      // bf16 planes per operand
    constexpr int ROWB = BK * 2;                    // bytes per LDS tile row
    constexpr int SPR = ROWB / 16;                  // 16-byte chunks per row
    constexpr int TILE_B = BM * ROWB;               // bytes per plane tile
    constexpr int ROUNDS = TILE_B / 8192;           // 512 threads x 16 B per round
    constexpr int STAGE_B = TILE_B * NPL * 2;       // A planes + W planes = 64 KiB
    constexpr int KSTEPS = BK / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;

    // ---- XCD-aware, bijective block -> tile map --------------------------------------------------------
    const int ntn = p.N / BN, ntm = p.M / BM;
    const int nwg = ntn * ntm;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;

    const bf16* __restrict__ Ag = (const bf16*)p.A + m0 * p.lda;
    const bf16* __restrict__ Wg = (const bf16*)p.W + (int64_t)n0 * p.K;

    // ---- per-thread DMA source offsets (elements), identical for every K tile --------------------------
    uint32_t a_off[ROUNDS], w_off[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int slot = r * 512 + tid;
        const int row = slot / SPR, s = slot % SPR;
        const int c = s ^ swz<BK>(row);
        a_off[r] = (uint32_t)(row * p.lda + c * 8);
        w_off[r] = (uint32_t)(row * p.K + c * 8);
    }

    auto stage = [&](int kt, int buf) {
        char* sb = smem + buf * STAGE_B + wave * 1024;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            const bf16* Ap = Ag + pl * p.a_plane + kt * BK;
            const bf16* Wp = Wg + pl * p.w_plane + kt * BK;
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                glds16(Ap + a_off[r], sb + pl * TILE_B + r * 8192);
                glds16(Wp + w_off[r], sb + (NPL + pl) * TILE_B + r * 8192);
            }
        }
    };

    // ---- fragment read addresses (bytes inside a plane tile) -------------------------------------------
    // row = w*.. + i*16 + fr ; chunk = ks*4 + fq ; swizzle depends only on fr (see swz<>): one base per ks.
    int a_rd[KSTEPS], w_rd[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const int ch = (ks * 4 + fq) ^ swz<BK>(fr);
        a_rd[ks] = (wm * 128 + fr) * ROWB + ch * 16;
        w_rd[ks] = (wn * 64 + fr) * ROWB + ch * 16;
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const char* sa = smem + buf * STAGE_B;
        const char* sw = sa + NPL * TILE_B;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            bf16x8 wf[NPL][4];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    wf[pl][j] = *(const bf16x8*)(sw + pl * TILE_B + w_rd[ks] + j * 16 * ROWB);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bf16x8 af[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) af[pl] = *(const bf16x8*)(sa + pl * TILE_B + a_rd[ks] + i * 16 * ROWB);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // swapped operands: D[n_local][m_local]
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][j], af[0], acc[i][j], 0, 0, 0);
                    if constexpr (NSPLIT == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][j], af[1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][j], af[0], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    };

    // ---- main loop: DMA of tile t+1 in flight under the MFMAs of tile t --------------------------------
    const int nkt = p.K / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) stage(kt + 1, cur ^ 1);
        compute(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane holds rows m = .. + fr, columns n .. n+3 ----------------------------------------
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t m = m0 + wm * 128 + i * 16 + fr;
        int orow = 0;
        int i1 = 0, i2 = 0;
        if constexpr (EPI == EPI_EMBED) {
            orow = p.row_map[m];
            i1 = p.idx1[m];
            if (p.table2) i2 = p.idx2[m];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fq * 4;
            const float4 b4 = *(const float4*)(p.bias + n);
            float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
            if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
                if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                }
                bf16* o = (bf16*)p.out + m * p.ldo + n;
                if constexpr (NSPLIT == 1) {
                    bf16x4 h = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    *(bf16x4*)o = h;
                } else {
                    bf16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bf16 a, b; split2(v[e], a, b); h[e] = a; l[e] = b; }
                    *(bf16x4*)o = h;
                    *(bf16x4*)(o + p.o_plane) = l;
                }
            } else if constexpr (EPI == EPI_RESID) {
                float* xp = p.x + m * p.N + n;
                float4 xv = *(const float4*)xp;
                if (p.gamma) {
                    const float4 g4 = *(const float4*)(p.gamma + n);
                    xv.x += g4.x * v[0]; xv.y += g4.y * v[1]; xv.z += g4.z * v[2]; xv.w += g4.w * v[3];
                } else {
                    xv.x += v[0]; xv.y += v[1]; xv.z += v[2]; xv.w += v[3];
                }
                *(float4*)xp = xv;
            } else {  // EPI_EMBED
                if (orow >= 0) {
                    const float4 t1 = *(const float4*)(p.table1 + (int64_t)i1 * p.N + n);
                    float4 r4 = {v[0] + t1.x, v[1] + t1.y, v[2] + t1.z, v[3] + t1.w};
                    if (p.table2) {
                        const float4 t2 = *(const float4*)(p.table2 + (int64_t)i2 * p.N + n);
                        r4.x += t2.x; r4.y += t2.y; r4.z += t2.z; r4.w += t2.w;
                    }
                    *(float4*)(p.x + (int64_t)orow * p.N + n) = r4;
                }
            }
        }
    }
}

template <int NSPLIT, int EPI> hipError_t launch_t(const GemmArgs& a, hipStream_t s) {
    constexpr int LDS = 131072;
    static bool configured = false;
    auto kfn = gemm_bf16_kernel<NSPLIT, EPI>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int nwg = (a.M / 256) * (a.N / 256);
    hipLaunchKernelGGL(kfn, dim3(nwg), dim3(512), LDS, s, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm(const GemmArgs& a, int nsplit, int epilogue, hipStream_t s) {
    if (a.M <= 0 || a.M % 256 || a.N % 256 || a.K % 64 || a.lda % 8 || (nsplit != 1 && nsplit != 3)) return hipErrorInvalidValue;
#define VTQ_CASE(NS, EP) if (nsplit == NS && epilogue == EP) return launch_t<NS, EP>(a, s);
    VTQ_CASE(1, EPI_BIAS) VTQ_CASE(1, EPI_BIAS_GELU) VTQ_CASE(1, EPI_RESID) VTQ_CASE(1, EPI_EMBED)
    VTQ_CASE(3, EPI_BIAS) VTQ_CASE(3, EPI_BIAS_GELU) VTQ_CASE(3, EPI_RESID) VTQ_CASE(3, EPI_EMBED)
#undef VTQ_CASE
    return hipErrorInvalidValue;
}

}  // namespace vtq
