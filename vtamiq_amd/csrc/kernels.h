// Host-side launch prototypes of the gfx950 kernels (internal; the public ABI is include/vtamiq_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>
#include <stdint.h>

namespace vtq {

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_RESID = 2, EPI_EMBED = 3 };

// Operand format of a dense contraction: element type and MFMAs per product.
//   terms 1: a*w (one 16-bit plane each);  2: (a_hi + a_lo)*w (activation hi/lo planes, single weight plane);
//   terms 3: a_hi*w_hi + a_lo*w_hi + a_hi*w_lo (hi/lo planes for both).  ABI code (include/vtamiq_hip.h): terms | (f16 ? 16 : 0).
//   f16 == 2: e4m3 bytes on the MX-scaled MFMA (unit block scales), one plane each, terms = 1; ABI code 33.
struct Num {
    int f16;      // 0 = bf16 planes, 1 = fp16 planes, 2 = fp8 (e4m3) bytes
    int terms;    // 1 | 2 | 3
    int apl() const { return terms == 1 ? 1 : 2; }     // planes of an activation tensor
    int wpl() const { return terms == 3 ? 2 : 1; }     // planes of a weight tensor
};
inline Num num_from_code(int code) { return Num{(code >> 4) & 3, code & 15}; }
inline bool num_valid(Num n) {
    if (n.f16 == 2) return n.terms == 1;
    return (n.terms == 1 || n.terms == 3 || (n.terms == 2 && n.f16 == 1)) && (n.f16 == 0 || n.f16 == 1);
}

// ---- Every build switch of csrc/, in one place.  The PRODUCT library (python -m vtamiq_amd.build) defines NONE of them: the
// code under them is measurement or experiment code that no shipped kernel contains (tests/test_layout.py greps the shipped objects).  Builds with
// switches are made by tools/build_abl.sh NAME "-D..." into tools/_abl/NAME.so and loaded through VTQ_LIB_PATH.
//   VTQ_WITH_FP8          the fp8 experiment (include/vtamiq_hip_fp8.h; python -m vtamiq_amd.build --fp8): e4m3 GEMM instantiations, fp8 entry points
//   VTQ_MEASURE           makes VTQ_MEASURE_ENV getenv: VTQ_GEMM_FLAGS / _CUS / _CG / _SCHED / _STAGGER, VTQ_ATTN_VARIANT, VTQ_ATTN_LDS_PAD (A/B runs)
//   VTQ_GEMM_DIAG         gemm.hip: s_memtime / s_memrealtime stamps around K loops, kernel and epilogue steps; shadow-VALU filler (profiles/r03_clock.txt)
//   VTQ_EPI_ABL=1..4      gemm.hip epilogue ablations: 1 no GELU arithmetic, 2 no copy-out, 3 no global stores, 4 no LDS staging (r03_gemm_epilogue_ablation.txt)
//   VTQ_EPI_ORDER=1|2     gemm.hip: both wave groups copy first / convert first in an epilogue interval (the plane-alternating form only)
//   VTQ_RESID_DEFER=1     gemm.hip: residual epilogue with the copy-out's LDS reads issued before the next chunk's conversion (equal: profiles/r05_epilogue_balanced.txt)
//   VTQ_EPI_BALANCED=0    gemm.hip: the plane-alternating passes of the two-plane bias / GELU epilogue (rounds 2 - 4) instead of the balanced ones (profiles/r05_epilogue_balanced.txt)
//   VTQ_RESID_PLANES      gemm.hip: pricing build of a LayerNorm fold's producer side (profiles/r03_ln_fold_price.txt)
//   VTQ_GEMM_ST_EXPLORE   gemm_st.hip: extra tile shapes and the load-only / compute-only modes of tools/st_bench.py (profiles/r05_gemm_tile_shapes.txt)
//   VTQ_RL_ABL, VTQ_RL_MM, VTQ_RL_PRE, VTQ_RL_POST   gemm_rowln.hip: K-loop ablations, MFMA statement form, waits (profiles/r04_rowln_anatomy.txt)
//   VTQ_ATTN_DIAG         attention.hip: per-wave phase stamps (profiles/r03_attention_anatomy.txt)
//   VTQ_ATTN_NO_VMASK     attention.hip: without the zeroing of masked keys' V rows (profiles/r04_attention_vmask_ab.txt)
//   VTQ_SW_NOFILL / NOMFMA / NOSTORE / NOQ / NODMA / PAIRED / DIST / HALFREADS   attention.hip pipelined kernel: skeleton ablations, read-ahead distance
//   VTQ_SW_QPF=0 / VTQ_SW_EARLY_WRITE=0   attention.hip pipelined kernel: without the L2 prefetch of the next block's Q / with the finished block's output written mid-iteration (profiles/r06_attention_loop.txt)
//   VTQ_SW_PRIO=1|2|3     attention.hip: s_setprio alternating between the two waves of a SIMD per phase / per fragment group / static for waves 4-7 (profiles/r05_attention_prio.txt)
//   VTQ_SW_SEAM_STAGGER=n attention.hip: every second workgroup of an XCD starts n us late (profiles/r05_attention_seams.txt)
//   VTQ_LIBM_ERF          dev_common.h: erff() instead of the fitted exact-erf GELU (accuracy cross-check)
//   VTQ_GELU_PACKED=1     dev_common.h: the GELU polynomial as v_pk_fma_f32 (same bits, 30 instead of 44 instructions per 4 values; equal time: profiles/r05_gelu_packed.txt)
//
// Measurement knobs.  The PRODUCT library reads no environment variable and executes no measurement branch: every knob below exists only
// in builds with -DVTQ_MEASURE (tools/build_abl.sh), where VTQ_MEASURE_ENV is getenv; in the shipped build it is a null constant (the
// variable names do not even appear in the objects: tests/test_layout.py greps for them) and GemmArgs::flags is ignored by the kernels.
#ifdef VTQ_MEASURE
#define VTQ_MEASURE_ENV(name) getenv(name)
#else
#define VTQ_MEASURE_ENV(name) ((const char*)nullptr)
#endif

// GemmArgs::flags, measurement knobs (-DVTQ_MEASURE builds: environment VTQ_GEMM_FLAGS, read once per process)
enum { GEMM_FLAG_WRAP_ROWS = 1,     // every tile writes the rows of row panel 0: no HBM write stream (timing experiments only)
       GEMM_FLAG_NO_CHAIN = 2,      // no DMA chaining across a workgroup's consecutive tiles
       GEMM_FLAG_DYNAMIC = 4,       // one schedule entry per workgroup, as many workgroups as entries (hardware dispatch order
                                    // instead of the persistent lists): the round-1 launch form, for A/B timing
       GEMM_FLAG_NO_EPILOGUE = 8,   // skip the epilogue (timing experiments only: output is not written)
       GEMM_FLAG_WRAP_LOADS = 16 }; // every tile loads the first two A and W panels: operands always hit L2 (timing experiments only)

// Where the sequences live in the row-major activation buffers: sequence s starts at row s*pitch + (s/per)*gap -- `per`
// sequences per part-batch, each part-batch padded by `gap` rows to a multiple of 256 (the GEMM tile height).
struct SeqMap { int pitch, per, gap; };
__host__ __device__ inline int64_t seq_row(const SeqMap& m, int s) { return (int64_t)s * m.pitch + (int64_t)(s / m.per) * m.gap; }

// Optional 16-bit plane copy of a kernel's fp32 row output, in the form the next skinny stage reads (PReLU(*slope) first when
// slope != NULL); p == NULL: none.
struct PlaneOut { void* p; int64_t plane; int ld; int f16; int planes; const float* slope; };

// fp8 mode: what a kernel that WRITES e4m3 activation bytes reports besides them (both optional):
//   amax: running maximum of |value| BEFORE scaling, as float bits (atomicMax on the int view: values are >= 0) -- the calibration
//         forward reads it to choose the tensor's power-of-two scale;
//   err:  the engine's error word; bit 2 is set when a value * scale left e4m3's range and was clamped to +-448.
struct Fp8Obs { float* amax; int* err; };

struct GemmArgs {
    const void* A; int64_t a_plane; int lda;      // 16-bit planes [M, lda]
    const void* W; int64_t w_plane;               // bf16 planes [N, K]
    int M, N, K;                                  // M % 256 == 0, N % 256 == 0, K % 128 == 0 (terms 1) | K % 64 == 0
    const float* bias;                            // [N]
    const float* gamma;                           // [N] or nullptr (EPI_RESID)
    float* x;                                     // fp32 [*, N]: EPI_RESID in/out, EPI_EMBED out
    void* out; int64_t o_plane; int ldo;          // bf16 planes (EPI_BIAS / EPI_BIAS_GELU)
    const int* row_map;                           // EPI_EMBED: output row of x for GEMM row m, or -1
    const int* idx1; const float* table1;         // EPI_EMBED: + table1[idx1[m]]  (position embedding)
    const int* idx2; const float* table2;         // EPI_EMBED: + table2[idx2[m]]  (scale embedding) or nullptr
    // fp8 operands only: acc is de-scaled by wscale[n] * ascale_inv before the bias (per-output-channel weight scale, static
    // activation scale); a GELU output is written as e4m3(value * out_scale); a BIAS output as fp16 hi/lo planes
    const float* wscale; float ascale_inv; float out_scale;
    Fp8Obs obs;                                   // fp8 GELU form only (the one GEMM epilogue that writes e4m3 bytes)
    const int* sched;                             // set by launch_gemm: per-workgroup tile lists (gemm.hip build_schedule)
    int st_grid;                                  // set by launch_gemm_st: the XCD grid of the small-tile kernels, rows * 16 + columns (gemm_st.hip)
    int flags;                                    // set by launch_gemm: GEMM_FLAG_* (-DVTQ_MEASURE builds; the product kernels read 0, gemm_flags())
    // diagnostic builds (-DVTQ_GEMM_DIAG, tools/build_abl.sh) only; set by launch_gemm from gemm_set_diag, never read otherwise:
    unsigned long long* diag;                     //   per workgroup 8 words: K-loop and whole-kernel s_memtime / s_memrealtime sums
    int shadow;                                   //   dummy v_fma_f32 issued in every load phase (x8): the price of VALU work beside the partner's MFMAs
    float* row_stats;                             //   -DVTQ_RESID_PLANES pricing build only: per row and column tile (mean, M2) of the new residual row
};

#ifdef VTQ_MEASURE
__host__ __device__ inline int gemm_flags(const GemmArgs& a) { return a.flags; }
#else
__host__ __device__ constexpr int gemm_flags(const GemmArgs&) { return 0; }
#endif

hipError_t launch_gemm(const GemmArgs& a, Num num, int epilogue, hipStream_t s);
// Tile shape of a GEMM launch.  launch_gemm picks one from (M, N, K, operand format) by gemm_tile_rule -- a pure speed choice: every
// shape gives every output element the same MFMA sequence and epilogue arithmetic (gemm_st.hip "bitwise contract"), so results do not
// depend on it.  GEMM_TILE_256: the persistent 256x256 / 128x256 kernel of gemm.hip; GEMM_ST_*: one workgroup per small tile (gemm_st.hip).
enum { GEMM_TILE_AUTO = -1, GEMM_TILE_256 = 0, GEMM_ST_64 = 1, GEMM_ST_64X2 = 2, GEMM_ST_128 = 3 };
hipError_t launch_gemm_st(const GemmArgs& a, Num num, int epilogue, int variant, hipStream_t s);
int gemm_tile_rule(int M, int N, int K, Num num, int cus = 0);     // host only: the shape launch_gemm would pick on a device of `cus` CUs (0: 256)
void gemm_set_variant(int v);                        // test / measurement hook: GEMM_TILE_AUTO (default) or a forced shape
void gemm_set_cus_per_xcd(int n);                    // measurement hook: workgroups per XCD of the persistent 256x256 launch (0 = all 32)
hipError_t launch_cu_map(unsigned* out, int nblocks, int spin_us, hipStream_t s);   // measurement: (XCC id, HW_ID) of each workgroup, one per CU
int device_cus(int* cus);                            // CU count of the current device (cached per device); non-zero on failure
// Whole-row residual GEMM with LayerNorm in its epilogue (gemm_rowln.hip): x[M, 768] += gamma * (A W^T + bias), then (ln_w != NULL)
// out planes = LayerNorm(x; ln_w, ln_b).  N = 768, 3-term operand formats, M % 128 == 0, K % 32 == 0.
struct RowLnArgs {
    const void* A; int64_t a_plane; int lda;      // activation planes [M, lda] (hi, lo)
    const void* W; int64_t w_plane;               // weight planes [768, K]
    int M, N, K;
    const float* bias; const float* gamma;        // [768]; gamma NULL = 1
    float* x;                                     // fp32 [M, 768], in / out
    const float* ln_w; const float* ln_b;         // [768]; NULL: no LayerNorm output
    void* out; int64_t o_plane;                   // planes [M, 768] of the normalised rows
};
hipError_t launch_gemm_rowln(const RowLnArgs& a, Num num, hipStream_t s);
// diagnostic builds: stamp buffer (256 workgroups x 8 words, device memory, or NULL) and shadow-VALU count of the following launches
int attention_rule(int nseq, int S_pad, int H, int terms, int cus);    // host-only: the form launch_attention picks: 0 four-wave kernel, 1 pipelined kernel, 2 split
void attention_set_variant(int v);               // test / measurement hook: -1 the rule, 0 / 1 / 2 as above
void attention_set_map(int m);                   // measurement hook: block walk of the pipelined kernel, 0 = XCD-strided (default), 1 = contiguous / paired (round 3)
void attention_set_cus(int cus);                 // measurement hook: size the persistent attention grid for `cus` CUs (0 = the device's)
unsigned long long* gemm_diag_buffer();         // the buffer of gemm_set_diag (attention's diagnostic build shares it)
void gemm_set_diag(unsigned long long* buf, int shadow);
bool gemm_is_diag_build();
// Build and upload (async on s) the persistent tile schedule of an (M, N, K) GEMM with wpl weight planes on the current device, if
// it is not cached yet: called by the engine before the first launch of a forward so that launch_gemm itself never allocates
hipError_t gemm_prepare(int M, int N, int K, int wpl, hipStream_t s);
// Persistent schedule of a (ntm x ntn)-tile GEMM with K columns and wpl weight planes (host only): 257 offsets, then the
// per-workgroup lists; entry = (tile << 2) | kind, kind 0 full, 1 / 2 top / bottom 128-row half
std::vector<int> gemm_tile_schedule(int ntm, int ntn, int K, int wpl);

// fp32 -> 16-bit planes (f16: 0 = bf16, 1 = fp16; planes: 1 = single, 2 = hi + lo with lo `plane` elements behind hi), or
// f16 == 2: e4m3 bytes of value * scale
hipError_t launch_split(const float* src, void* dst, int64_t plane, int64_t numel, int f16, int planes, hipStream_t s, float scale = 1.0f,
                        Fp8Obs obs = Fp8Obs{nullptr, nullptr});
// dst[i] = src[i] * mul (weight ingestion: the softmax scale folded into the query projection, engine.hip kQLog2Scale)
hipError_t launch_scale_copy(const float* src, float* dst, int64_t n, float mul, hipStream_t s);
// fp8 weights: W[N][K] fp32 -> e4m3 rows with a per-row power-of-two scale; inv_scale[n] = 1 / scale
hipError_t launch_quant_rows_fp8(const float* src, void* dst, float* inv_scale, int N, int K, hipStream_t s, int Kp = 0);
// rows of K floats -> planes with row pitch Kp >= K, zero beyond K
hipError_t launch_split_rows_pad(const float* src, void* dst, int64_t plane, int rows, int K, int Kp, int f16, int planes, hipStream_t s);

// nimg images (ref, dist[, dist2]) of fp32 patches [B*N, K] each -> 16-bit planes [rows_pad, K], rows >= nimg*B*N zero-filled
hipError_t launch_pack_patches(const float* const* imgs, int nimg, void* dst, int64_t plane, int BN, int K, int rows_pad, int f16,
                               int planes, hipStream_t s, float scale = 1.0f, int Kp = 0, Fp8Obs obs = Fp8Obs{nullptr, nullptr});

// per patch row r in [0, rows_pad): pos index, scale index (sc == nullptr: none), destination row in the residual stream (or -1);
// positions outside [0, 1) are clamped into the table and flagged in *err (bit 0)
hipError_t launch_embed_index(const float* const* pos, const float* const* sc, int nimg, int* pidx, int* sidx, int* row_map, int B, int N,
                              int rows_pad, SeqMap sm, int T, int grid, int num_scales, int* err, hipStream_t s);

// CLS (+pos row 0) and register tokens into the first T rows of every sequence
// pre-embedded input (transformer.py:534-535): x[row_map[r]] = feats[r] + table1[pidx[r]] (+ table2[sidx[r]]); feats: nimg pointers to (B*N, H) fp32
hipError_t launch_embed_rows(const float* const* feats, int nimg, int BN, const int* row_map, const int* pidx, const int* sidx,
                             const float* table1, const float* table2, float* x, int H, hipStream_t s);
hipError_t launch_tokens(float* x, const float* cls, const float* pos_table, const float* extra, int nseq, SeqMap sm,
                         int T, int H, hipStream_t s);

hipError_t launch_layernorm(const float* x, const float* w, const float* b, void* out, int64_t o_plane, int rows, int H,
                            int f16, int planes, hipStream_t s, float scale = 1.0f, Fp8Obs obs = Fp8Obs{nullptr, nullptr});

// num.terms: 1 = single planes, 3 = hi/lo planes for Q, K, V and P (the 2-term form is not offered: DESIGN.md section 2)
// out8_scale > 0: the output is written as e4m3 bytes of value * out8_scale ([rows][H] bytes) instead of planes (fp8 mode)
// q_log2 (3-term formats only): Q already carries the softmax scale 1/sqrt(64) * log2(e) (the engine folds it into the query
// projection's weights); otherwise the kernels fold it into their Q fragments themselves (attention.hip prescale_q)
hipError_t launch_attention(const void* qkv, int64_t plane, void* out, int64_t o_plane, int nseq, int S, int S_pad, int H,
                            Num num, hipStream_t s, float out8_scale = 0.0f, Fp8Obs obs = Fp8Obs{nullptr, nullptr}, bool q_log2 = false);

// zero the rows of the residual stream that belong to no token: per-sequence pads, per-part tails, and everything up to rows_total
hipError_t launch_zero_pad_rows(float* x, int nseq, int S, SeqMap sm, int H, int rows_total, hipStream_t s);

// copy token rows (first T rows of each sequence) of x into trace[nseq][T][H]
hipError_t launch_copy_tokens(const float* x, float* dst, int nseq, SeqMap sm, int T, int H, hipStream_t s);

// d[j*B + b] = gamma * (LN(x[row(b)]) - LN(x[row((j+1)*B + b)])), j < ndist  (final encoder_norm on the CLS rows only; vtamiq.py:104-111)
hipError_t launch_final_diff(const float* x, const float* ln_w, const float* ln_b, const float* gamma, float* d, int B, int ndist,
                             SeqMap sm, int H, PlaneOut po, hipStream_t s, int* err = nullptr);

// one-time RCAB weight fold [Wc ; Wd Wc], bcat = [bc ; Wd bc + bd] (head.hip)
hipError_t launch_fold_ca(const float* Wc, const float* bc, const float* Wd, const float* bd, float* Wcat, float* bcat, int H, int hid,
                          hipStream_t s);

// ---- skinny linear stages on MFMA (skinny.hip): CLS tail of the last layer and the DiffNet head --------------------------
enum { SK_PLAIN = 0, SK_GELU = 1, SK_PRELU = 2, SK_RESID = 3, SK_GATE = 4, SK_CONVCAT = 5 };
struct SkinnyArgs {
    const void* xa; int64_t xa_plane; int ldx;    // activation planes [apl][>= ceil64(R)][ldx] (16-bit), ldx >= K, ldx % 8 == 0
    const void* W; int64_t w_plane;               // weight planes [wpl][ceil16(N)][K]
    int R, N, K;                                  // valid rows / outputs; K % 32 == 0 (zero-padded)
    const float* bias;                            // [N]
    int epi;                                      // SK_*: v = acc + bias, then
                                                  //   GELU gelu(v) | PRELU prelu(v, *post_slope) | RESID res + gamma * v (gamma NULL = 1)
                                                  //   GATE res + aux * sigmoid(v) | CONVCAT relu(v) for columns >= nsplit
    const float* post_slope;
    const float* gamma;
    const float* res; const float* aux; int ldr;  // fp32 [R][ldr]
    int nsplit;
    float* y; int ldy; int ycols;                 // fp32 output (or NULL): columns [0, ycols)
    void* ya; int64_t ya_plane; int ldya;         // 16-bit plane output (or NULL): columns [pcol0, N) stored at column c - pcol0,
    int ya_planes; int pcol0;                     //   as prelu(value, *next_slope) when next_slope != NULL (the consumer's pre-activation)
    const float* next_slope;
};
hipError_t launch_skinny(const SkinnyArgs& a, Num num, hipStream_t s);
// fp32 rows [R][ldx] -> 16-bit planes [R][ldo] (K columns), PReLU(*slope) first when slope != NULL
hipError_t launch_rows_to_planes(const float* x, int ldx, const float* slope, void* out, int64_t plane, int ldo, int R, int K, int f16,
                                 int planes, hipStream_t s);

// ---- CLS-only tail of the last encoder layer (cls_tail.hip) -----------------------------------------------------------
hipError_t launch_rows_ln(const float* src, int64_t stride, const float* w, const float* b, float* ln, float* copy, int rows, int H,
                          PlaneOut po, hipStream_t s);
// K, V rows of the packed qkv planes (f16, planes); any S (the score buffer is dynamic LDS)
// q_log2: q already carries 1/sqrt(64) * log2(e) (see launch_attention)
hipError_t launch_cls_attention(const float* q, const void* qkv, int64_t plane, float* out, int nseq, int S, int S_pad, int H,
                                int f16, int planes, PlaneOut po, hipStream_t s, bool q_log2 = false);
// largest S launch_cls_attention accepts (LDS score buffer); longer sequences run the full last layer instead
int cls_attention_max_seq();

// ---- on-device image -> patch tensor (patches.hip) -------------------------------------------------------
hipError_t launch_image_normalize(const uint8_t* in, float* out, int NI, int H, int W, const int* flips, const float* mean, const float* sd,
                                  hipStream_t s);
hipError_t launch_avgpool2(const float* in, float* out, int NC, int H, int W, hipStream_t s);
hipError_t launch_gather_patches(const float* const* levels, const int* hs, const int* ws, int nlevels, const int* samples,
                                 const int* scale_ids, float* patches, float* pos, float* scales, int NI, int N, hipStream_t s, int P = 16);

// ---- measurement: the matrix pipe's sustained rate on this device (mfma_stream.hip) -----------------------------------------
// f16: 0 bf16, 1 fp16 operands; data: 0 the 3-term mix of gaussian hi / lo planes, 1 zeros, 2 uniform random
hipError_t mfma_stream_measure(int f16, int data, double warm_s, double timed_s, double* tflops, double* ghz, hipStream_t s);

// ---- validation reductions (metrics.hip) --------------------------------------------------------------------------------
hipError_t launch_repeat_mean(const float* q, double* out, int R, int N, hipStream_t s);
hipError_t launch_rank_metrics(const double* a, const double* b, int N, int normalize, double* aa, double* bb, double* ra, double* rb,
                               long long* counts, double* out, hipStream_t s);

#if defined(__HIPCC__) && defined(VTQ_DEV_COMMON)
// 4 consecutive values of one row -> the plane sink (device side; dev_common.h must be included first)
__device__ __forceinline__ void plane_store4(const PlaneOut& o, int64_t row, int col, float a, float b, float c, float d) {
    if (!o.p) return;
    if (o.slope) {
        const float sl = *o.slope;
        a = a >= 0.f ? a : sl * a; b = b >= 0.f ? b : sl * b; c = c >= 0.f ? c : sl * c; d = d >= 0.f ? d : sl * d;
    }
    if (o.f16) {
        f16* dst = (f16*)o.p + row * o.ld + col;
        if (o.planes == 1) *(f16x4*)dst = f16x4{(f16)a, (f16)b, (f16)c, (f16)d};
        else {
            f16x4 h, l; f16 x, y;
            split2<f16>(a, x, y); h[0] = x; l[0] = y; split2<f16>(b, x, y); h[1] = x; l[1] = y;
            split2<f16>(c, x, y); h[2] = x; l[2] = y; split2<f16>(d, x, y); h[3] = x; l[3] = y;
            *(f16x4*)dst = h; *(f16x4*)(dst + o.plane) = l;
        }
    } else {
        bf16* dst = (bf16*)o.p + row * o.ld + col;
        if (o.planes == 1) *(bf16x4*)dst = bf16x4{(bf16)a, (bf16)b, (bf16)c, (bf16)d};
        else {
            bf16x4 h, l; bf16 x, y;
            split2<bf16>(a, x, y); h[0] = x; l[0] = y; split2<bf16>(b, x, y); h[1] = x; l[1] = y;
            split2<bf16>(c, x, y); h[2] = x; l[2] = y; split2<bf16>(d, x, y); h[3] = x; l[3] = y;
            *(bf16x4*)dst = h; *(bf16x4*)(dst + o.plane) = l;
        }
    }
}
#endif

}  // namespace vtq
