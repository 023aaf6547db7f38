// Whole-row residual GEMM with LayerNorm in its epilogue (gfx950), N = H = 768:
//
//     x[m, :]  +=  gamma * (A[m, :] W^T + bias)                      (out-proj / fc2 + LayerScale + residual, transformer.py:279, 284)
//     out[m,:]  =  planes( LayerNorm(x[m, :]; ln_w, ln_b, eps 1e-6) ) (the NEXT block's attention_norm / ffn_norm, :276, :281)
//
// The north star's "fused LayerNorm + QKV projection", built from the producer side: the workgroup that owns a 128-row panel owns every
// one of its 768 columns, so after the residual add it holds whole rows, forms mean / variance in-workgroup and writes the fp32 stream
// AND the operand planes the following QKV / fc1 GEMM reads.  The stand-alone LayerNorm launch (and its re-read of x) disappears.
//
// Structure (3-term hi/lo operands; one MFMA k-step = 32):
//   * tile 128 x 768 per 256-thread workgroup = FOUR waves, one per SIMD, each with the whole 512-entry register file: wave w owns all
//     128 rows x columns [192 w, 192 w + 192) = 8 x 12 accumulator blocks of 16 x 16 = 384 registers per lane -- 256 in the AGPR half
//     (column blocks 0..7) and 128 in the VGPR half (8..11).  hipcc cannot place that with the MFMA builtins (it spills); the MFMAs are
//     asm statements whose accumulator operand is constrained "+a" / "+v", everything else is allocated by the compiler as usual.
//   * W is wave-PRIVATE (a wave's 192 weight rows are read by nobody else): each wave streams its own rows by LDS-DMA into its own ring
//     of 16 slots of 2 KiB (slot = one 16-row column block, hi + lo plane, of one K tile), 16 blocks ahead of the block it computes on,
//     under a counted vmcnt and with NO workgroup barrier.  Only the A panel (16 KiB per K tile, shared by the four waves) needs one:
//     two slots, the next-but-one K tile requested one K tile ahead; ONE s_barrier per K tile.  LDS: 32 KiB (A) + 4 x 32 KiB (W) = 160 KiB.
//   * per K tile a wave keeps its 16 A fragments in registers (read once), walks its 12 column blocks (W fragment of block j + 1 read
//     while block j's 24 MFMAs issue) and reloads the A fragments for the next K tile inside the last block: per 288 MFMAs 40
//     ds_read_b128 (35 B/clk/CU against 62 for the 256 x 256 ping-pong tile) and 28 LDS-DMA instructions, all in the MFMAs' shadow.
//   * accumulation order per output element is the 256 x 256 kernel's (K tiles ascending, W_hi A_hi, W_hi A_lo, W_lo A_hi; accumulators
//     start at the bias): x is BIT-IDENTICAL to gemm_pp2_kernel<.., EPI_RESID>, and the LayerNorm arithmetic is elementwise.hip
//     ln_row's, so the planes are bit-identical to layernorm_kernel's (tests/test_gpu_kernels.py compares them bitwise).
//   * epilogue: eight passes of 16 rows through two fp32 LDS images (row pitch 3088 B: conflict-free ds_write_b128 from the accumulator
//     registers -- AGPRs are read by the LDS instruction directly); a wave then owns 4 WHOLE rows of the pass: 3 float4 per lane, the
//     residual row brought into LDS by LDS-DMA one pass ahead, two wave reductions, x stored as 1-KiB row segments, the planes as 512-B ones.
//   * measured (profiles/r04_rowln_*.txt): bit-identical and NOT faster than the two launches it replaces -- out-proj ties,
//     fc2 is 7 % slower -- so the engine uses it only under vtq_config.options & VTQ_OPT_FUSED_LAYERNORM.
#include <mutex>

#include "dev_common.h"
#include "kernels.h"

namespace vtq {

namespace {

constexpr int kH = 768;             // N of the GEMM = hidden size (whole rows)
constexpr int kRows = 128;          // tile height
constexpr int kASlot = 16384;       // one K tile of the A panel: [plane 2][row block 8][16 rows][64 B]
constexpr int kWBase = 2 * kASlot;  // W rings start behind the two A slots
constexpr int kWWave = 32768;       // one wave's ring: 16 slots
constexpr int kWSlot = 2048;        // [plane 2][16 rows][64 B]
constexpr int kRing = 16;
constexpr int kCb = 12;             // 16-column blocks per wave
constexpr int kLds = 163840;
constexpr int kPitch = kH * 4 + 16; // epilogue image row pitch (bytes)
constexpr int kImg = 16 * kPitch;   // one 16-row fp32 image

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
#ifndef VTQ_RL_PRE
#define VTQ_RL_PRE "s_nop 1\n\t"       // measurement builds may empty these two (wrong results possible)
#endif
#ifndef VTQ_RL_POST
#define VTQ_RL_POST "\n\ts_nop 7"
#endif
#define VTQ_RL_MM3(TY) "v_mfma_f32_16x16x32_" TY " %0, %1, %3, %0\n\tv_mfma_f32_16x16x32_" TY " %0, %1, %4, %0\n\tv_mfma_f32_16x16x32_" TY " %0, %2, %3, %0"
// two row blocks at once, their chains interleaved (each accumulator still sees hi*hi, hi*lo, lo*hi in that order):
// %0 %1 = accumulators, %2 %3 = w_hi w_lo, %4 %5 = a_hi a_lo of the first row block, %6 %7 of the second
#define VTQ_RL_MM6(TY)                                                                                             \
    "v_mfma_f32_16x16x32_" TY " %0, %2, %4, %0\n\tv_mfma_f32_16x16x32_" TY " %1, %2, %6, %1\n\t"                   \
    "v_mfma_f32_16x16x32_" TY " %0, %2, %5, %0\n\tv_mfma_f32_16x16x32_" TY " %1, %2, %7, %1\n\t"                   \
    "v_mfma_f32_16x16x32_" TY " %0, %3, %4, %0\n\tv_mfma_f32_16x16x32_" TY " %1, %3, %6, %1"
template <typename T, bool AG>
__device__ __forceinline__ void mma3(f32x4& c, const u32x4& w_hi, const u32x4& w_lo, const u32x4& a_hi, const u32x4& a_lo) {
    if constexpr (std::is_same<T, f16>::value) {
        if constexpr (AG) asm volatile(VTQ_RL_PRE VTQ_RL_MM3("f16") : "+a"(c) : "v"(w_hi), "v"(w_lo), "v"(a_hi), "v"(a_lo));
        else asm volatile(VTQ_RL_PRE VTQ_RL_MM3("f16") VTQ_RL_POST : "+v"(c) : "v"(w_hi), "v"(w_lo), "v"(a_hi), "v"(a_lo));
    } else {
        if constexpr (AG) asm volatile(VTQ_RL_PRE VTQ_RL_MM3("bf16") : "+a"(c) : "v"(w_hi), "v"(w_lo), "v"(a_hi), "v"(a_lo));
        else asm volatile(VTQ_RL_PRE VTQ_RL_MM3("bf16") VTQ_RL_POST : "+v"(c) : "v"(w_hi), "v"(w_lo), "v"(a_hi), "v"(a_lo));
    }
}
template <typename T, bool AG>
__device__ __forceinline__ void mma6(f32x4& c0, f32x4& c1, const u32x4& w_hi, const u32x4& w_lo, const u32x4& a0_hi, const u32x4& a0_lo,
                                     const u32x4& a1_hi, const u32x4& a1_lo) {
    if constexpr (std::is_same<T, f16>::value) {
        if constexpr (AG) asm volatile(VTQ_RL_PRE VTQ_RL_MM6("f16") : "+a"(c0), "+a"(c1) : "v"(w_hi), "v"(w_lo), "v"(a0_hi), "v"(a0_lo), "v"(a1_hi), "v"(a1_lo));
        else asm volatile(VTQ_RL_PRE VTQ_RL_MM6("f16") VTQ_RL_POST : "+v"(c0), "+v"(c1) : "v"(w_hi), "v"(w_lo), "v"(a0_hi), "v"(a0_lo), "v"(a1_hi), "v"(a1_lo));
    } else {
        if constexpr (AG) asm volatile(VTQ_RL_PRE VTQ_RL_MM6("bf16") : "+a"(c0), "+a"(c1) : "v"(w_hi), "v"(w_lo), "v"(a0_hi), "v"(a0_lo), "v"(a1_hi), "v"(a1_lo));
        else asm volatile(VTQ_RL_PRE VTQ_RL_MM6("bf16") VTQ_RL_POST : "+v"(c0), "+v"(c1) : "v"(w_hi), "v"(w_lo), "v"(a0_hi), "v"(a0_lo), "v"(a1_hi), "v"(a1_lo));
    }
}

template <int OFF> __device__ __forceinline__ void lds_rd(u32x4& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(d) : "v"(addr), "i"(OFF));
}
template <int OFF, bool AG> __device__ __forceinline__ void lds_wr(uint32_t addr, const f32x4& v) {
    if constexpr (AG) asm volatile("ds_write_b128 %0, %1 offset:%c2" ::"v"(addr), "a"(v), "i"(OFF) : "memory");
    else asm volatile("ds_write_b128 %0, %1 offset:%c2" ::"v"(addr), "v"(v), "i"(OFF) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct RowLnKArgs {
    const void* A; int64_t a_plane; int lda;
    const void* W; int64_t w_plane;
    int M, K;
    const float* bias; const float* gamma;
    float* x;
    const float* ln_w; const float* ln_b;
    void* out; int64_t o_plane;
    unsigned long long* diag;                     // -DVTQ_GEMM_DIAG builds: gemm_diag_buffer(); never read otherwise
};

enum { KT_FIRST = 0, KT_MID = 1, KT_PENULT = 2, KT_LAST = 3 };     // position of a K tile in its row tile (nkt >= 4)

template <typename T, bool LN>
__global__ __launch_bounds__(256, 1) void gemm_rowln_kernel(RowLnKArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nkt = p.K / 32;
    const int ntiles = p.M / kRows;
    const uint32_t smem0 = lds_addr(smem);

    f32x4 acc[8][kCb];                               // [row block][column block]; column blocks 0..7 in AGPRs, 8..11 in VGPRs
    u32x4 ah[8], al[8], wh[2], wl[2];

#ifdef VTQ_GEMM_DIAG
    // diagnostic build only: shader-clock / 100 MHz stamps per phase, summed over this workgroup's tiles, into a buffer nothing else reads
    unsigned long long dg_c[4] = {0, 0, 0, 0}, dg_r[4] = {0, 0, 0, 0}, dg_t, dg_rt, dg_aw = 0, dg_ww = 0;
    auto tick = [&]() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; };
    auto stamp = [&](int k) {
        unsigned long long t, r;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r) :: "memory");
        if (k >= 0) { dg_c[k] += t - dg_t; dg_r[k] += r - dg_rt; }
        dg_t = t; dg_rt = r;
    };
#define VTQ_RL_STAMP(k) stamp(k);
#else
#define VTQ_RL_STAMP(k)
#endif
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t m0 = (int64_t)tile * kRows;
        VTQ_RL_STAMP(-1)
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));               // per-lane addressing is re-derived here for every tile (nothing lives across the epilogue)
        const int fr = lane & 15, fq = lane >> 4;
        const int drow = lane >> 2, dch = (lane & 3) ^ (((drow >> 3) & 1) * 3);      // LDS-DMA piece: 16 rows x 64 B, chunk swizzled on the source
        const uint32_t a_off = (uint32_t)(drow * p.lda + dch * 8) * 2u;
        const uint32_t w_off = (uint32_t)(drow * p.K + dch * 8) * 2u;
        const uint32_t frag = (uint32_t)(fr * 64 + ((fq ^ (((fr >> 3) & 1) * 3)) << 4));   // fragment read: row fr, 16-byte chunk fq (swizzled)
        const uint32_t a_rd = smem0 + frag;
        const uint32_t w_rd = smem0 + kWBase + wave * kWWave + frag;
        const char* Ag = (const char*)p.A + (m0 + wave * 32) * (int64_t)p.lda * 2;           // this wave's quarter of the A panel: 32 rows
        const char* Wg = (const char*)p.W + (int64_t)(wave * 192) * p.K * 2;               // this wave's 192 weight rows
        const int64_t a_pl = p.a_plane * 2, w_pl = p.w_plane * 2;
        const uint32_t a_blk = (uint32_t)p.lda * 32u, w_blk = (uint32_t)p.K * 32u;         // bytes between 16-row blocks

        // Addresses are (scalar base of the K tile and plane) + (32-bit per-lane offset incl. the row block): two scalar pairs live, not 24
        auto stage_a = [&](int kt, int piece) {      // my quarter of A(kt) -> slot kt & 1; piece = plane * 2 + row block (of my two)
            uint32_t off = a_off + (uint32_t)(piece & 1) * a_blk;
            asm volatile("" : "+v"(off));
            char* dst = smem + (kt & 1) * kASlot + wave * 2048 + (piece >> 1) * 8192 + (piece & 1) * 1024;
            const char* src = Ag + kt * 64 + (piece >> 1) * a_pl;
            glds16(src + off, dst);
        };
        auto stage_w = [&](int kt, int j, int pl) {  // column block j of K tile kt (step g = 12 kt + j), plane pl: 16 weight rows -> ring slot g % 16
            uint32_t off = w_off + (uint32_t)j * w_blk;
            asm volatile("" : "+v"(off));
            const int g = kt * kCb + j;
            char* dst = smem + kWBase + wave * kWWave + (g & (kRing - 1)) * kWSlot + pl * 1024;
            const char* src = Wg + kt * 64 + pl * w_pl;
            glds16(src + off, dst);
        };
        auto read_a = [&](int i, uint32_t base) {
            switch (i) {                             // the offset is an instruction immediate
#define VTQ_RA(k) case k: lds_rd<k * 1024>(ah[k], base); lds_rd<8192 + k * 1024>(al[k], base); break;
                VTQ_RA(0) VTQ_RA(1) VTQ_RA(2) VTQ_RA(3) VTQ_RA(4) VTQ_RA(5) VTQ_RA(6) VTQ_RA(7)
#undef VTQ_RA
            }
        };

        // ---- accumulators start at the bias (as gemm_pp2_kernel's do) --------------------------------------------------------------
        {
            f32x4 b4[kCb];
#pragma unroll
            for (int j = 0; j < kCb; ++j) b4[j] = *(const f32x4*)(p.bias + wave * 192 + j * 16 + fq * 4);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < kCb; ++j) acc[i][j] = b4[j];
        }
        // ---- prologue: A(0), A(1), the first 16 column blocks of W ------------------------------------------------------------------
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) stage_a(0, pc);
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) stage_a(1, pc);
#pragma unroll
        for (int j = 0; j < kCb; ++j) { stage_w(0, j, 0); stage_w(0, j, 1); }
#pragma unroll
        for (int j = 0; j < kRing - kCb; ++j) { stage_w(1, j, 0); stage_w(1, j, 1); }
        // Requests of a wave retire in issue order; a wait names how many of the YOUNGEST may stay in flight (over-waiting is safe).
        //   step g = 12 kt + j issues the refill W(g + 16) (while that block exists), then waits for W(g + 1): 15 W pairs are younger in
        //   the steady state, r - 2 in the last 16 steps (r = steps left incl. this one); the A pieces between them are waited for too.
        //   step 12 kt + 11 first waits for A(kt + 1), requested 12 steps (12 refills = 24 pieces) earlier -- 9 refills at the
        //   penultimate K tile; 16 + 11 W pairs after the prologue's A(1) -- then the barrier, then requests A(kt + 2).
        wait_vm<36>();                               // A(0) landed; younger: A(1)'s 4 pieces + 16 W pairs
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) read_a(i, a_rd);
        wait_vm<30>();                               // W block 0 landed
        lds_rd<0>(wh[0], w_rd);
        lds_rd<1024>(wl[0], w_rd);
        // the accumulator initialisation above was VALU / v_accvgpr_write: pad it against the first MFMA reading it as C
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");

        // One column-block step = 24 MFMAs (8 row blocks x 3 terms) = 384 matrix-pipe cycles, issued by ONE wave per SIMD: everything else
        // of the step rides between the MFMA groups (a clustered LDS-DMA issue of ~60 cycles each would idle the pipe):
        // the MFMAs go out as four statements of two row blocks each (six MFMAs, the two accumulation chains interleaved); behind
        //   statement 0: refill request, hi plane      statement 1: refill request, lo plane
        //   statement 2: wait for block g + 1, read its fragment (needed at the top of the next step)
        //   last step of a K tile: A(kt + 1) is waited for + barrier at its top; the A fragments of the row blocks one statement back are
        //   reloaded behind statements 1, 2, 3 (the last two blocks behind the step); one piece of A(kt + 2) is requested behind each.
        auto ktile = [&](int kt, auto pos_c) {
            constexpr int POS = decltype(pos_c)::value;
#pragma unroll
            for (int j = 0; j < kCb; ++j) {
                const int g = kt * kCb + j;
                const int r = (POS == KT_PENULT) ? 2 * kCb - j : ((POS == KT_LAST) ? kCb - j : 1000);    // steps left incl. this one (tail tiles)
                const bool last_step = (j == kCb - 1 && POS != KT_LAST);
                const bool refill = r > kRing;
                if (last_step) {
                    // A(kt + 1) landed for everyone; everyone has long finished reading A(kt) into registers -> its slot takes A(kt + 2).
                    // Younger than A(kt + 1)'s last piece: the refills of 11 steps (8 at the penultimate K tile; 16 + 11 W pairs behind the
                    // prologue's A(1))
#ifdef VTQ_GEMM_DIAG
                    const unsigned long long ta = tick();
#endif
                    if (POS == KT_FIRST) wait_vm<54>(); else if (POS == KT_MID) wait_vm<22>(); else wait_vm<16>();
                    __builtin_amdgcn_s_barrier();
#ifdef VTQ_GEMM_DIAG
                    dg_aw += tick() - ta;            // cycles in the wait for A(kt + 1) + the K tile's barrier
#endif
                }
                // W fragment of block j (and, at j = 0, the A fragments reloaded during the previous step) have arrived; the statement
                // names them so that no compiler copy of these registers can sit between the reads and the wait
                if (j == 0)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wh[0]), "+v"(wl[0]), "+v"(ah[0]), "+v"(al[0]), "+v"(ah[1]), "+v"(al[1]), "+v"(ah[2]), "+v"(al[2]),
                                 "+v"(ah[3]), "+v"(al[3]), "+v"(ah[4]), "+v"(al[4]), "+v"(ah[5]), "+v"(al[5]), "+v"(ah[6]), "+v"(al[6]), "+v"(ah[7]), "+v"(al[7]) :: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wh[j & 1]), "+v"(wl[j & 1]) :: "memory");
                const int kt_n = kt + 1 + (j + 4 >= kCb ? 1 : 0), j_n = (j + 4) % kCb;       // block g + 16: its slot (g % 16) is free now
                const uint32_t an = a_rd + (uint32_t)(((kt + 1) & 1) * kASlot);
#ifndef VTQ_RL_MM
#define VTQ_RL_MM 1                   // 1: two row blocks per MFMA statement, chains interleaved; 0: one row block (three dependent MFMAs back to back)
#endif
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#if VTQ_RL_MM == 1
                    if ((i & 1) == 0) {
                        if (j < 8) mma6<T, true>(acc[i][j], acc[i + 1][j], wh[j & 1], wl[j & 1], ah[i], al[i], ah[i + 1], al[i + 1]);
                        else mma6<T, false>(acc[i][j], acc[i + 1][j], wh[j & 1], wl[j & 1], ah[i], al[i], ah[i + 1], al[i + 1]);
                        continue;
                    }
#else
                    if (j < 8) mma3<T, true>(acc[i][j], wh[j & 1], wl[j & 1], ah[i], al[i]);
                    else mma3<T, false>(acc[i][j], wh[j & 1], wl[j & 1], ah[i], al[i]);
#endif
                    // fillers behind the MFMAs of row blocks (i - 1, i) [pair form: i odd] / of row block i
#if !(defined(VTQ_RL_ABL) && (VTQ_RL_ABL & 1))              // measurement build bit 0: no W refills in the K loop (results wrong by design)
                    if (i == 1 && refill) stage_w(kt_n, j_n, 0);
                    if (i == 3 && refill) stage_w(kt_n, j_n, 1);
#endif
                    if (i == 5 && r > 1) {
                        // W block g + 1 landed: 15 W pairs are younger in the steady state (r - 2 in the last 16 steps); A pieces among
                        // them are waited for as well
#if defined(VTQ_RL_ABL) && (VTQ_RL_ABL & 1)
                        if (false) {}
#else
#ifdef VTQ_GEMM_DIAG
                        if (r > kRing) { const unsigned long long tw = tick(); wait_vm<30>(); dg_ww += tick() - tw; }      // cycles in the wait for W(g + 1)
#else
                        if (r > kRing) wait_vm<30>();
#endif
#endif
                        else if (r == 16) wait_vm<28>(); else if (r == 15) wait_vm<26>(); else if (r == 14) wait_vm<24>(); else if (r == 13) wait_vm<22>();
                        else if (r == 12) wait_vm<20>(); else if (r == 11) wait_vm<18>(); else if (r == 10) wait_vm<16>(); else if (r == 9) wait_vm<14>();
                        else if (r == 8) wait_vm<12>(); else if (r == 7) wait_vm<10>(); else if (r == 6) wait_vm<8>(); else if (r == 5) wait_vm<6>();
                        else if (r == 4) wait_vm<4>(); else if (r == 3) wait_vm<2>(); else wait_vm<0>();
#if !(defined(VTQ_RL_ABL) && (VTQ_RL_ABL & 2))              // measurement build bit 1: no W fragment reads
                        const uint32_t wa = w_rd + (uint32_t)(((g + 1) & (kRing - 1)) * kWSlot);
                        lds_rd<0>(wh[(j + 1) & 1], wa);
                        lds_rd<1024>(wl[(j + 1) & 1], wa);
#endif
                    }
                    if (last_step) {
                        // the fragments of the row blocks one filler slot behind are dead: reload them from A(kt + 1)
                        if (i == 3) { read_a(0, an); read_a(1, an); }
                        if (i == 5) { read_a(2, an); read_a(3, an); }
                        if (i == 7) { read_a(4, an); read_a(5, an); }
                        if (POS != KT_PENULT) {
                            if (i == 1) stage_a(kt + 2, 0); else if (i == 3) stage_a(kt + 2, 1); else if (i == 5) stage_a(kt + 2, 2); else if (i == 7) stage_a(kt + 2, 3);
                        }
                    }
                }
                if (last_step) { read_a(6, an); }
                if (last_step) read_a(7, an);
            }
        };
        VTQ_RL_STAMP(0)                              // [0] prologue: bias, first requests, A(0) / W(0) landed
        ktile(0, std::integral_constant<int, KT_FIRST>());
        for (int kt = 1; kt < nkt - 2; ++kt) ktile(kt, std::integral_constant<int, KT_MID>());
        ktile(nkt - 2, std::integral_constant<int, KT_PENULT>());
        ktile(nkt - 1, std::integral_constant<int, KT_LAST>());
        // ---- epilogue ------------------------------------------------------------------------------------------------------------------
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMA's result before any reader of the accumulators
        VTQ_RL_STAMP(1)                                         // [1] the K loop
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();                           // every wave is done with the rings: the LDS is the epilogue's now
        // LDS now: two fp32 images of 16 rows (accumulators of a pass, row pitch 3088 B) | the residual rows of a pass (16 x 3072 B, filled
        // by LDS-DMA one pass ahead: no registers, 12 KiB in flight per wave) | gamma, ln_w, ln_b (3 x 3072 B)
        constexpr int kXImg = 2 * kImg, kConst = kXImg + 16 * kH * 4;
        static_assert(kConst + 3 * kH * 4 <= kLds, "epilogue LDS map");
        if (threadIdx.x < 192) {
            const int c = threadIdx.x * 4;
            *(f32x4*)(smem + kConst + c * 4) = p.gamma ? *(const f32x4*)(p.gamma + c) : f32x4{1.f, 1.f, 1.f, 1.f};
            if constexpr (LN) {
                *(f32x4*)(smem + kConst + kH * 4 + c * 4) = *(const f32x4*)(p.ln_w + c);
                *(f32x4*)(smem + kConst + 2 * kH * 4 + c * 4) = *(const f32x4*)(p.ln_b + c);
            }
        }
        // read side: wave w owns rows 4 w .. 4 w + 3 of every 16-row pass (whole rows: the LayerNorm statistics are two wave reductions);
        // a lane holds columns 4 lane + 256 i (i = 0..2) of its rows -- elementwise.hip ln_row's layout
        const float* xg = p.x + (m0 + wave * 4) * kH + lane * 4;
        auto dma_x = [&](int pass, int q) {                     // residual row (pass, q) of this wave -> its place in the x image
            const char* src = (const char*)(xg + (int64_t)(pass * 16 + q) * kH);
            char* dst = smem + kXImg + (wave * 4 + q) * (kH * 4);
#pragma unroll
            for (int i = 0; i < 3; ++i) glds16(src + i * 1024, dst + i * 1024);
        };
#pragma unroll
        for (int q = 0; q < 4; ++q) dma_x(0, q);
        // write side: lane (fr, fq) holds row fr, columns wave * 192 + cb * 16 + fq * 4 .. + 3 of every row block
        const uint32_t wr_addr = smem0 + (uint32_t)(fr * kPitch + (wave * 192 + fq * 4) * 4);
        const uint32_t rd_acc = smem0 + (uint32_t)(wave * 4 * kPitch + lane * 16);
        const uint32_t rd_x = smem0 + (uint32_t)(kXImg + wave * 4 * kH * 4 + lane * 16);
        const uint32_t rd_c = smem0 + (uint32_t)(kConst + lane * 16);
        auto write_pass = [&](int pass) {                       // pass is a compile-time constant after unrolling
            const uint32_t base = wr_addr + (uint32_t)((pass & 1) * kImg);
#pragma unroll
            for (int j = 0; j < kCb; ++j) {
                if (j < 8) lds_wr<0, true>(base + j * 64, acc[pass][j]);
                else lds_wr<0, false>(base + j * 64, acc[pass][j]);
            }
        };
        write_pass(0);
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();
        constexpr int S = LN ? 9 : 3;                           // global stores per row: 3 x (16 B of x) [+ 3 x 2 planes x 8 B]
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            if (pass + 1 < 8) write_pass(pass + 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // the residual row (pass, q) has landed.  Requests behind its three pieces, in issue order: (pass 0) the later rows' pieces and,
                // per earlier row of this pass, its next-pass pieces + stores; (pass > 0) the stores of row q of the previous pass, then per
                // row in between 3 pieces (while a next pass exists) + S stores
                constexpr int kDummy = 0; (void)kDummy;
                const int younger = pass == 0 ? (3 - q) * 3 + q * (3 + S) : S + (3 - q) * (3 + S) + q * ((pass < 7 ? 3 : 0) + S);
                switch (younger) {
#define VTQ_C(k) case k: wait_vm<k>(); break;
                    VTQ_C(0) VTQ_C(3) VTQ_C(6) VTQ_C(9) VTQ_C(12) VTQ_C(15) VTQ_C(18) VTQ_C(21) VTQ_C(24) VTQ_C(27) VTQ_C(30) VTQ_C(33) VTQ_C(36) VTQ_C(39) VTQ_C(42) VTQ_C(45)
#undef VTQ_C
                    default: wait_vm<0>(); break;
                }
                u32x4 dv[3], dx[3], dg[3], dw[3], db[3];
                const uint32_t ba = rd_acc + (uint32_t)((pass & 1) * kImg + q * kPitch), bx = rd_x + (uint32_t)(q * kH * 4);
                asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:1024\n\tds_read_b128 %2, %6 offset:2048\n\t"
                             "ds_read_b128 %3, %7\n\tds_read_b128 %4, %7 offset:1024\n\tds_read_b128 %5, %7 offset:2048"
                             : "=&v"(dv[0]), "=&v"(dv[1]), "=&v"(dv[2]), "=&v"(dx[0]), "=&v"(dx[1]), "=&v"(dx[2]) : "v"(ba), "v"(bx) : "memory");
                asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048"
                             : "=&v"(dg[0]), "=&v"(dg[1]), "=&v"(dg[2]) : "v"(rd_c) : "memory");
                if constexpr (LN)
                    asm volatile("ds_read_b128 %0, %6 offset:3072\n\tds_read_b128 %1, %6 offset:4096\n\tds_read_b128 %2, %6 offset:5120\n\t"
                                 "ds_read_b128 %3, %6 offset:6144\n\tds_read_b128 %4, %6 offset:7168\n\tds_read_b128 %5, %6 offset:8192"
                                 : "=&v"(dw[0]), "=&v"(dw[1]), "=&v"(dw[2]), "=&v"(db[0]), "=&v"(db[1]), "=&v"(db[2]) : "v"(rd_c) : "memory");
                // the x row is in registers after this wait: its LDS place takes the same row of the next pass
                asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(dv[0]), "+v"(dv[1]), "+v"(dv[2]), "+v"(dx[0]), "+v"(dx[1]), "+v"(dx[2]) : "n"(LN ? 9 : 3) : "memory");
                if (pass + 1 < 8) dma_x(pass + 1, q);
                if constexpr (LN)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dg[0]), "+v"(dg[1]), "+v"(dg[2]), "+v"(dw[0]), "+v"(dw[1]), "+v"(dw[2]), "+v"(db[0]), "+v"(db[1]), "+v"(db[2]) :: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dg[0]), "+v"(dg[1]), "+v"(dg[2]) :: "memory");
                const int64_t row = m0 + pass * 16 + wave * 4 + q;
                float* xr = p.x + row * kH;
                f32x4 v[3];
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const f32x4 a = __builtin_bit_cast(f32x4, dv[i]), g4 = __builtin_bit_cast(f32x4, dg[i]);
                    f32x4 xx = __builtin_bit_cast(f32x4, dx[i]);
                    // gemm_pp2_kernel's residual epilogue: v = gamma * acc (the accumulator already holds the bias), rounded; x += v -- two
                    // roundings there (the product goes through an LDS image).  hipcc contracts even __fadd_rn(__fmul_rn()) into one FMA:
                    // the product is made opaque first
                    f32x4 t = {g4[0] * a[0], g4[1] * a[1], g4[2] * a[2], g4[3] * a[3]};
                    asm volatile("" : "+v"(t));
                    xx[0] += t[0]; xx[1] += t[1]; xx[2] += t[2]; xx[3] += t[3];
                    *(f32x4*)(xr + (i * 64 + lane) * 4) = xx;
                    v[i] = xx;
                    s += (xx[0] + xx[1]) + (xx[2] + xx[3]);
                }
                if constexpr (LN) {                              // elementwise.hip ln_row, operation for operation
                    const float mean = wave_sum(s) * (1.0f / kH);
                    float qq = 0.f;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        v[i][0] -= mean; v[i][1] -= mean; v[i][2] -= mean; v[i][3] -= mean;
                        qq += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
                    }
                    const float rstd = 1.0f / sqrtf(wave_sum(qq) * (1.0f / kH) + 1e-6f);
                    T* o = (T*)p.out + row * kH;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        typedef typename Vec<T>::x4 tx4;
                        const f32x4 lw4 = __builtin_bit_cast(f32x4, dw[i]), lb4 = __builtin_bit_cast(f32x4, db[i]);
                        tx4 h, l;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float y = v[i][k] * rstd * lw4[k] + lb4[k];
                            T a_, b_;
                            split2<T>(y, a_, b_);
                            h[k] = a_; l[k] = b_;
                        }
                        *(tx4*)(o + (i * 64 + lane) * 4) = h;
                        *(tx4*)(o + p.o_plane + (i * 64 + lane) * 4) = l;
                    }
                }
            }
            if (pass + 1 < 8) {
                wait_lgkm0();                                    // my writes of image (pass + 1) are in the LDS
                __builtin_amdgcn_s_barrier();                    // ... everyone's are, and everyone has read image pass
            }
        }
        VTQ_RL_STAMP(2)                                         // [2] the epilogue up to its last store's issue
        wait_vm<0>();
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();                           // the images are dead before the next tile's DMA lands on them
        VTQ_RL_STAMP(3)                                         // [3] the drain of the stores
    }
#ifdef VTQ_GEMM_DIAG
    if (p.diag && threadIdx.x == 0) {
        unsigned long long* d = p.diag + (size_t)blockIdx.x * 64;
        for (int k = 0; k < 4; ++k) { d[k] = dg_c[k]; d[4 + k] = dg_r[k]; }
        d[8] = dg_aw; d[9] = dg_ww;
    }
#endif
#undef VTQ_RL_STAMP
}

}  // namespace

hipError_t launch_gemm_rowln(const RowLnArgs& a, Num num, hipStream_t s) {
    if (a.M <= 0 || a.M % kRows || a.K <= 0 || a.K % 32 || a.K < 128 || a.lda % 8 || a.N != kH || num.terms != 3 || num.f16 > 1) return hipErrorInvalidValue;
    if (!a.A || !a.W || !a.bias || !a.x || (a.ln_w && (!a.ln_b || !a.out))) return hipErrorInvalidValue;
    static std::mutex mu;
    static bool configured[64][4] = {};
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    const bool ln = a.ln_w != nullptr;
    typedef void (*kern_t)(RowLnKArgs);
    const kern_t kern = num.f16 ? (ln ? (kern_t)gemm_rowln_kernel<f16, true> : (kern_t)gemm_rowln_kernel<f16, false>)
                                : (ln ? (kern_t)gemm_rowln_kernel<bf16, true> : (kern_t)gemm_rowln_kernel<bf16, false>);
    {
        std::lock_guard<std::mutex> lk(mu);
        bool& done = configured[dev][num.f16 * 2 + (ln ? 1 : 0)];
        if (!done) {
            e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
            if (e != hipSuccess) return e;
            done = true;
        }
    }
    RowLnKArgs k{a.A, a.a_plane, a.lda, a.W, a.w_plane, a.M, a.K, a.bias, a.gamma, a.x, a.ln_w, a.ln_b, a.out, a.o_plane, gemm_diag_buffer()};
    const int ntiles = a.M / kRows;
    const int grid = ntiles < cus ? ntiles : cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kLds, s, k);
    return hipGetLastError();
}

}  // namespace vtq
