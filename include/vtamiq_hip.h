/*
 * vtamiq_hip.h -- C ABI of libvtamiq_hip.so, the MI355X (gfx950) engine for VTAMIQ's ViT patch-pair forward.
 *
 * The reference (ch-andrei/VTAMIQ) is pure Python/PyTorch and has no FFI of its own; this header is the
 * boundary a maintainer binds with ctypes.  Each entry point names the reference code
 * it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated otherwise; the caller (torch) owns all inputs/outputs,
 *     the library owns only packed weights and its workspace;
 *   - all calls are asynchronous on the hipStream_t passed in (void* here so the header needs no HIP include);
 *     no internal synchronisation except where stated;
 *   - int return: 0 = ok, non-zero = error; text via vtq_last_error(); no exceptions cross the ABI;
 *   - a handle is not thread-safe; one process per GPU.
 */
#ifndef VTAMIQ_HIP_H
#define VTAMIQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VTQ_ABI_VERSION 9

/* numerics mode of the dense contractions (fp32 accumulate, fp32 LayerNorm/softmax/residual in all of them).
 * bf16 and fp16 MFMAs run at the same rate on gfx950; fp16 carries 11 significand bits instead of 8 in the range the reference's
 * own GPU path uses for these contractions (torch.cuda.amp.autocast(float16), train.py:602). */
#define VTQ_PREC_BF16   0   /* one bf16 MFMA per product                                                                      */
#define VTQ_PREC_BF16X3 1   /* operands split hi+lo bf16: a_hi*w_hi + a_lo*w_hi + a_hi*w_lo (3 MFMAs)                          */
#define VTQ_PREC_FP16   2   /* one fp16 MFMA per product                                                                      */
#define VTQ_PREC_FP16X3 3   /* operands split hi+lo fp16, 3 MFMAs per product: at the fp32 reference's own noise floor         */
#define VTQ_PREC_FP16X2 4   /* linear layers: activations split hi+lo fp16, weights single fp16 (2 MFMAs per product);         */
                            /* attention (QK^T, PV): the 3-term fp16 form                                                      */
/* 5 is the fp8 EXPERIMENT (BASELINE configs[4]): not a scoring mode and not in this library -- vtq_create rejects it unless the library
 * was built with -DVTQ_WITH_FP8; its constants and entry points live in include/vtamiq_hip_fp8.h. */

/* operand-format code of the per-kernel entry points: MFMAs per product (1 | 2 | 3) + 16 for fp16 planes (0 = bf16):
 *   1 / 17 single plane each; 18 = activation hi/lo planes x single weight plane (fp16 only); 3 / 19 = hi/lo planes for both */
#define VTQ_NUM_BF16   1
#define VTQ_NUM_BF16X3 3
#define VTQ_NUM_FP16   17
#define VTQ_NUM_FP16X2 18
#define VTQ_NUM_FP16X3 19

typedef struct vtq_config {
    int32_t hidden_size;       /* 768 | 1024                (transformer.py:68-98)                    */
    int32_t mlp_dim;           /* 3072 | 4096                                                         */
    int32_t num_heads;         /* 12 | 16 ; head_dim must be 64                                       */
    int32_t num_layers;        /* kept layers               (transformer.py:342-345)                  */
    int32_t patch_dim;         /* 3*16*16 = 768 | 3*8*8 = 192 (ViT-B/8, transformer.py:81-85)         */
    int32_t pos_grid;          /* img_dim / patch: 24 | 48   (transformer.py:411)                     */
    int32_t num_extra_tokens;  /* register tokens           (transformer.py:487-492)                  */
    int32_t num_scales;        /* scale embedding iff > 1   (transformer.py:500)                      */
    int32_t use_layer_scale;   /* ls1/ls2 gamma             (transformer.py:270-271)                  */
    int32_t calibrate;         /* DiffNet on/off            (vtamiq.py:63-69)                         */
    int32_t diff_scale;        /* LayerScale on the CLS diff (vtamiq.py:61)                           */
    int32_t num_rgs;           /* vtamiq.py:34                                                        */
    int32_t num_rcabs;         /* vtamiq.py:35                                                        */
    int32_t ca_hidden;         /* hidden_size / ca_reduction (channel_attention.py:75)                */
    int32_t precision;         /* VTQ_PREC_*                                                          */
    int32_t num_adapters;      /* Adapter pairs per layer (transformer.py:260-269); pair 0 is applied   */
    int32_t options;           /* VTQ_OPT_* bit mask (tests and measurement; 0 = the product path)      */
    int32_t reserved[3];
} vtq_config;

/* vtq_config.options.  The library reads NO environment variable: what rounds 1-3 steered through VTQ_NO_CLS_PRUNE /
 * VTQ_FP8_STATIC_SCALES is an explicit field of the configuration the caller hands over. */
#define VTQ_OPT_FULL_LAST_LAYER   1   /* run the last encoder layer on every token row instead of on the CLS rows only (same result) */
                                      /* (2 belongs to the fp8 experiment: include/vtamiq_hip_fp8.h)                                  */
#define VTQ_OPT_FUSED_LAYERNORM   4   /* LayerNorm inside the residual GEMMs: the out-proj launch also writes LayerNorm 2's operand planes, */
                                      /* the fc2 launch the next layer's LayerNorm 1 planes (csrc/gemm_rowln.hip; hidden 768, 3-term          */
                                      /* formats, no adapters: vtq_create fails otherwise).  Bit-identical scores; measured 1.3 % SLOWER at   */
                                      /* B = 32 than separate LayerNorm launches, hence opt-in                               */

typedef struct vtq_tensor_desc {
    const char*  name;         /* HOST string: the reference's state_dict key          */
    const float* data;         /* DEVICE pointer, fp32, contiguous                                    */
    int64_t      numel;
} vtq_tensor_desc;

typedef struct vtq_engine* vtq_handle;

int         vtq_abi_version(void);
const char* vtq_last_error(void);

/* Replaces VTAMIQ.__init__ / VisionTransformer.__init__ for the inference path (vtamiq.py:27-79). */
int  vtq_create(const vtq_config* cfg, vtq_handle* out);
void vtq_destroy(vtq_handle h);

/* Replaces Module.load_state_dict for the engine: borrows fp32 device tensors for the duration of the call
 * (which synchronises the stream before returning) and packs them to bf16 (hi[,lo]) / fp32 buffers owned by
 * the handle.  Unknown names are an error; every tensor of the configured topology must be present. */
int  vtq_load_weights(vtq_handle h, const vtq_tensor_desc* descs, int32_t n, void* stream);

/* Bytes of device workspace the handle holds for a (B pairs, N patches) call. */
size_t vtq_workspace_bytes(vtq_handle h, int32_t B, int32_t N);
/* Grows the workspace ahead of time (hipMalloc happens here, or lazily on the first larger vtq_forward). */
int  vtq_reserve(vtq_handle h, int32_t B, int32_t N);

/* Replaces VTAMIQ.forward (vtamiq.py:94-119): q_out[B] = score of each (ref, dist) pair.
 *   patches_* : [B, N, 3, 16, 16] fp32 contiguous      pos_* : [B, N, 2] fp32 in [0,1)
 *   scales_*  : [B, N] fp32-cast scale ids, or NULL when the model has no scale embedding
 *               (NULL with num_scales > 1 is an error, as in transformer.py:547-548). */
int  vtq_forward(vtq_handle h,
                 const float* patches_ref, const float* patches_dist,
                 const float* pos_ref, const float* pos_dist,
                 const float* scales_ref, const float* scales_dist,
                 int32_t B, int32_t N, float* q_out, void* stream);

/* VTAMIQ.forward on PRE-EMBEDDED input: Embeddings.forward takes the (B, N, H) branch (transformer.py:534-535) -- what a model built with
 * use_patch_embedding=False is fed, and what the reference does with ANY 3-D `patches` tensor.  feats_*: (B, N, hidden_size) fp32, contiguous; everything
 * else as vtq_forward.  The patch-embedding weights are not used (they must still be loaded: zeros do). */
int  vtq_forward_tokens(vtq_handle h, const float* feats_ref, const float* feats_dist, const float* pos_ref, const float* pos_dist,
                        const float* scales_ref, const float* scales_dist, int32_t B, int32_t N, float* q_out, void* stream);

/* Pairwise items (train.predict, train.py:281-301: two model calls sharing the reference image): patches/pos/scales are HOST
 * arrays of 3 DEVICE pointers {ref, dist1, dist2}, each as in vtq_forward (scales may be NULL); the reference image is
 * encoded ONCE (3B sequences instead of 4B).  q_out[2B]: q_out[b] = score(ref_b, dist1_b), q_out[B + b] = score(ref_b, dist2_b);
 * bit-identical to two vtq_forward calls. */
int  vtq_forward_pairwise(vtq_handle h, const float* const* patches, const float* const* pos, const float* const* scales,
                          int32_t B, int32_t N, float* q_out, void* stream);

/* vtq_forward_pairwise on PRE-EMBEDDED input (the (B, N, H) branch of Embeddings.forward, as vtq_forward_tokens): feats = HOST array of 3 DEVICE
 * pointers {ref, dist1, dist2}, each (B, N, hidden_size) fp32 contiguous.  Bit-identical to two vtq_forward_tokens calls. */
int  vtq_forward_pairwise_tokens(vtq_handle h, const float* const* feats, const float* const* pos, const float* const* scales,
                                 int32_t B, int32_t N, float* q_out, void* stream);

/* Input check.  The reference raises (IndexError / device assert) when a position lies outside [0, 1)
 * (transformer.py:417-421); vtq_forward clamps such an index into the table instead of gathering out of bounds and records it.
 * Bit 1: the CLS difference of some pair was not finite -- an operand left its format's range upstream (the fp16 operand modes
 * carry |v| <= 65504; VTQ_PREC_BF16X3 has the fp32 range), or the inputs / weights held inf / NaN.
 * (Bit 2 is raised by the fp8 experiment only: include/vtamiq_hip_fp8.h.)
 * This call synchronises `stream`, returns the flags accumulated since the last call and clears them. */
int  vtq_input_errors(vtq_handle h, int32_t* flags, void* stream);

/* Which token row of the encoder output the head consumes: VTAMIQ.token_num (vtamiq.py:57, 107-108: "can be CLS token or
 * extra_token").  0 = CLS (the reference's and this library's default), 1 .. num_extra_tokens = a register token.  Applies to every
 * later vtq_forward / vtq_forward_pairwise; an index outside the model's tokens is refused (the reference would read a PATCH row). */
int  vtq_set_iqa_token(vtq_handle h, int32_t token);

/* Debug tap: when buf != NULL, every later vtq_forward also writes the pre-final-LN token rows after the
 * embedding and after each layer: buf[(L+1)][2B][T][H] fp32 (ref sequences first).  Mirrors
 * vit_config["return_layers"] (transformer.py:369-372, 632-636). */
int  vtq_set_token_trace(vtq_handle h, float* buf);
/* Test hooks for localising a divergence: leave the encoder after stage `layer * 7 + k` of the NEXT forwards (k = 0 LayerNorm 1,
 * 1 QKV, 2 attention, 3 out-proj, 4 LayerNorm 2, 5 fc1, 6 fc2; -1 = run everything; the scores of such a forward are
 * meaningless), and borrow the workspace: x = fp32 residual stream [rows, H], lnbuf = LayerNorm / attention output planes,
 * big = QKV / fc1 output planes, in the layouts DESIGN.md section 3 gives for the engine's precision. */
/* Measurement hook: how the pipelined attention kernel's persistent workgroups walk the (sequence, head, 256-row block) items -- 0 (default) XCD-strided:
 * the workgroups of an XCD take the blocks of the same few (sequence, head) pairs side by side, so a pair's K / V tiles are fetched once per XCD;
 * 1: the round-3 walk (consecutive blocks per workgroup; paired when a pair has two blocks).  Same results either way. */
int  vtq_debug_attention_map(int32_t m);
/* Measurement hook for launches on CU-masked streams (hipExtStreamCreateWithCUMask; tools/cu_partition.py): size the persistent grids of
 * the following launches for the CUs such a stream owns -- the 256x256 GEMM for `gemm_cus_per_xcd` workgroups on each of the 8 XCDs (its tile
 * schedule is rebuilt for that grid), the pipelined attention kernel for `attention_cus` CUs.  0 = the whole device (the default).  Process-wide;
 * results never depend on it. */
int  vtq_debug_cu_partition(int32_t gemm_cus_per_xcd, int32_t attention_cus);
/* Which CUs does a stream own?  Launches `nblocks` workgroups that hold a whole CU each (144 KiB of LDS) for ~spin_us microseconds;
 * out[2 b] = XCC id, out[2 b + 1] = HW_ID register (SE / SH / CU fields) of workgroup b.  out: 2 * nblocks uint32 of device memory. */
int  vtq_debug_cu_map(uint32_t* out, int32_t nblocks, int32_t spin_us, void* stream);
int  vtq_debug_stop_after(vtq_handle h, int32_t stage);
int  vtq_debug_buffers(vtq_handle h, void** x, void** lnbuf, void** big, int64_t* rows);
/* Clock diagnostic of the GEMM kernel (MI355X_MICROARCH.md 'DVFS give-back' item 6).  Only a library built with -DVTQ_GEMM_DIAG
 * (tools/build_abl.sh) executes stamps; the shipped build returns 0 and does nothing.  buf = 256 x 64 uint64 of device memory that
 * nothing else reads: per workgroup {sum over its K loops of s_memtime, of s_memrealtime; the same two over the whole kernel;
 * tiles; XCC id; then per wave the cycles spent in the epilogue conversion, copy-out and waits}; shadow = dummy VALU instructions (x8) issued in every LDS-read phase of the main loop (what vector work beside
 * the partner wave's MFMAs costs).  Returns 1 in a diagnostic build. */
int  vtq_debug_gemm_diag(void* buf, int32_t shadow);
/* Which of the two fused-attention kernels vtq_k_attention and the engine launch (process-wide; tests and measurement):
 * 0 = the 4-wave kernel, 1 = the 8-wave software-pipelined kernel, 2 = split (the pipelined kernel on the full 256-row query blocks, the
 * 4-wave kernel on the few rows behind them: S = 521, the reference-default topology), -1 = the library's rule (the pipelined kernel for
 * the 3-term formats when its 256-row blocks fill the chip; split when S is at most 64 rows past a multiple of 256).  Every form
 * computes the same arithmetic in the same order per query row: outputs are bit-identical. */
int  vtq_debug_attention_variant(int32_t variant);
/* Which tile shape vtq_k_gemm and the engine's GEMM launches use (process-wide; tests and measurement): -1 = the library's rule
 * (vtq_k_gemm_tile_rule), 0 = the persistent 256x256 kernel, 1 = 64x64 tiles, 2 = 128x128 tiles (one workgroup per tile,
 * csrc/gemm_st.hip).  Every shape gives every output element the same MFMA sequence and epilogue arithmetic: outputs are bit-identical. */
int  vtq_debug_gemm_variant(int32_t variant);
/* Host-only: the tile shape the library's rule gives an (M, N, K) launch in operand format num (VTQ_NUM_*); -1 = bad format code. */
int  vtq_k_gemm_tile_rule(int32_t M, int32_t N, int32_t K, int32_t num);
/* Host-only: which kernel the library's rule gives nseq sequences of pitch S_pad, hidden size H, operand format num (VTQ_NUM_*) on a
 * device with `cus` compute units: 2 = split, 1 = pipelined, 0 = 4-wave, -1 = bad format code. */
int  vtq_k_attention_rule(int32_t nseq, int32_t S_pad, int32_t H, int32_t num, int32_t cus);

/* The practical ceiling of the matrix pipe on this device (csrc/mfma_stream.hip; bench.py roofline.practical_peak_tflops_measured_here): a bare
 * stream of back-to-back mfma_f32_16x16x32 on register operands on every CU for >= timed_s seconds after >= warm_s seconds of the same
 * load (each <= 30).  f16: 0 = bf16, 1 = fp16 operands; data: 0 = the operand bits of a 3-term product (hi x hi, hi x lo, lo x hi of
 * gaussian planes), 1 = zeros (the issue limit), 2 = uniform random.  *tflops = MFMA-issue TFLOP/s, *ghz (may be NULL) = the clock it
 * implies.  Blocks the calling thread; measurement only. */
int  vtq_debug_mfma_stream(int32_t f16, int32_t data, double warm_s, double timed_s, double* tflops, double* ghz, void* stream);

/* ---- measurement: per-kernel-class HIP-event timing on the launch stream ------------------------------ */
#define VTQ_K_CONVERT  0
#define VTQ_K_PATCH    1   /* patch-embedding GEMM + pos/scale gather epilogue */
#define VTQ_K_LN       2
#define VTQ_K_QKV      3
#define VTQ_K_ATTN     4
#define VTQ_K_OUTPROJ  5
#define VTQ_K_FC1      6
#define VTQ_K_FC2      7
#define VTQ_K_HEAD     8
#define VTQ_K_COUNT    9
/* mask: bit k set = bracket every launch of class k with hipEvents (0 disables). */
int  vtq_profile_enable(vtq_handle h, uint32_t class_mask);
/* Synchronises the recorded events; ms_sum[k], launches[k] for k < VTQ_K_COUNT (HOST arrays); resets. */
int  vtq_profile_collect(vtq_handle h, double* ms_sum, int64_t* launches);

/* ---- per-kernel entry points (unit tests call these through the same ABI) ----------------------------- */
/* fp32 [rows, cols] -> 16-bit planes (f16: 0 = bf16, 1 = fp16): dst (hi) and, when planes == 2, dst + plane_stride (lo). */
int  vtq_k_split(const float* src, void* dst, int64_t plane_stride, int64_t numel, int32_t f16, int32_t planes, void* stream);

/* C[M,N] = A[M,K] * W[N,K]^T (+epilogue); A/W are 16-bit planes as above, format `num` = VTQ_NUM_*; M%256==0, N%256==0,
 * K%128==0 (one MFMA per product) or K%64==0.
 *   epilogue 0: out16  = acc + bias                 (1 or 2 planes, as the activations of `num`)
 *            1: out16  = gelu_erf(acc + bias)                              (transformer.py:212-215)
 *            2: x_f32 += gamma * (acc + bias)   (gamma NULL = 1)           (transformer.py:279,284) */
int  vtq_k_gemm(const void* A, int64_t a_plane, int32_t lda, const void* W, int64_t w_plane,
                int32_t M, int32_t N, int32_t K, int32_t num, int32_t epilogue,
                const float* bias, const float* gamma, float* x_f32,
                void* out16, int64_t o_plane, int32_t ldo, void* stream);

/* Whole-row residual GEMM with LayerNorm in its epilogue (csrc/gemm_rowln.hip; N = 768, num = VTQ_NUM_BF16X3 | VTQ_NUM_FP16X3, M % 128 == 0,
 * K % 32 == 0, K >= 128):
 *     x_f32[M, 768] += gamma * (A[M, K] * W[768, K]^T + bias)          (out-proj / fc2 + LayerScale + residual, transformer.py:279, 284)
 *     out16 planes   = LayerNorm(x_f32; ln_w, ln_b, eps 1e-6)           (the NEXT block's attention_norm / ffn_norm, transformer.py:276, 281)
 * in one launch: the workgroup that owns a 128-row panel owns whole rows.  ln_w == NULL: no LayerNorm output (the x update only).
 * x_f32 is bit-identical to vtq_k_gemm epilogue 2, out16 to vtq_k_layernorm applied to it. */
int  vtq_k_gemm_rowln(const void* A, int64_t a_plane, int32_t lda, const void* W, int64_t w_plane, int32_t M, int32_t K, int32_t num,
                      const float* bias, const float* gamma, float* x_f32, const float* ln_w, const float* ln_b,
                      void* out16, int64_t o_plane, void* stream);

/* HOST-only: the persistent schedule vtq_k_gemm uses for an [M, N] output (M, N multiples of 256) with K columns and `wplanes`
 * weight planes: out[0..256] = begin offsets of the 256 workgroups' lists (out[256] = total length), then the lists:
 * entry = (tile << 2) | kind with tile = row_tile * (N/256) + col_tile, kind 0 = 256x256 tile, 1 / 2 = its top / bottom 128 rows.
 * Returns the total length (writes at most cap entries; out may be NULL), or -1 on a bad shape.  No GPU needed. */
int  vtq_k_gemm_schedule(int32_t M, int32_t N, int32_t K, int32_t wplanes, int32_t* out, int32_t cap);

/* LayerNorm(eps=1e-6) rows of x[rows, H] fp32 -> 16-bit planes (transformer.py:253-254, 276, 281). */
int  vtq_k_layernorm(const float* x, const float* w, const float* b, void* out, int64_t o_plane,
                     int32_t rows, int32_t H, int32_t f16, int32_t planes, void* stream);

/* softmax(Q K^T / sqrt(64)) V per (sequence, head) on the packed qkv[rows, 3H] planes (num = VTQ_NUM_* with 1 or 3 terms);
 * sequences are S_pad rows apart, keys >= S are masked; out[rows, H] planes, heads merged (transformer.py:153-166). */
int  vtq_k_attention(const void* qkv, int64_t plane, void* out, int64_t o_plane,
                     int32_t nseq, int32_t S, int32_t S_pad, int32_t H, int32_t num, void* stream);

/* One skinny linear stage on the MFMA pipe (CLS tail of the last layer, DiffNet head): out[R, N] = act[R, K] * W[N, K]^T + bias,
 * operands as 16-bit planes in format `num` (VTQ_NUM_*): xa [planes][>= ceil64(R)][ldx], W [planes][ceil16(N)][K], K % 32 == 0.
 *   epi 0 plain | 1 gelu_erf | 2 prelu(*post_slope) | 3 res + gamma * v (gamma NULL = 1) | 4 res + aux * sigmoid(v) |
 *       5 relu for columns >= nsplit (RCAB conv with the channel-attention squeeze folded in, channel_attention.py:45, 58-61)
 * Outputs: y fp32 [R][ldy], columns [0, ycols) (may be NULL); ya: 16-bit planes of columns [pcol0, N), stored at column
 * c - pcol0, as prelu(value, *next_slope) when next_slope != NULL (may be NULL). */
int  vtq_k_skinny_linear(const void* xa, int64_t xa_plane, int32_t ldx, const void* W, int64_t w_plane, int32_t R, int32_t N, int32_t K,
                         int32_t num, int32_t epi, const float* bias, const float* post_slope, const float* gamma, const float* res,
                         const float* aux, int32_t ldr, int32_t nsplit, float* y, int32_t ldy, int32_t ycols, void* ya, int64_t ya_plane,
                         int32_t ldya, int32_t pcol0, const float* next_slope, void* stream);

/* The DiffNet head + quality predictor alone (quality_decoder -> q_predictor, vtamiq.py:114-117, channel_attention.py:13-86) with
 * the handle's loaded weights: d fp32 [HB][H] = diff_scale(cls_ref - cls_dist) -> q_out fp32 [HB]. */
int  vtq_k_diffnet_head(vtq_handle h, const float* d, int32_t HB, float* q_out, void* stream);

/* ---- on-device image -> patch tensor; replaces the CPU loader's transform_img (data/utils.py:76-94) and the
 * gather / position / pyramid part of get_iqa_patches (data/patch_sampling.py:529-611) for GIVEN sample coordinates ---------- */
/* images uint8 [NI, H, W, 3] -> out fp32 [NI, 3, H, W] = ((x / 255) - mean[c]) / std[c]; flips: DEVICE int32 [NI][2] = (hflip, vflip)
 * or NULL; mean/std: HOST float[3]. */
int  vtq_k_image_normalize(const uint8_t* images, float* out, int32_t NI, int32_t H, int32_t W, const int32_t* flips,
                           const float* mean, const float* std_, void* stream);
/* torch.nn.AvgPool2d(2): in [NC, H, W] -> out [NC, H/2, W/2]. */
int  vtq_k_avgpool2(const float* in, float* out, int32_t NC, int32_t H, int32_t W, void* stream);
/* levels: HOST array of nlevels (<= 4) DEVICE pointers to [NI, 3, hs[l], ws[l]]; samples DEVICE int32 [NI, N, 2] (row, col at the
 * patch's own scale); scale_ids DEVICE int32 [NI, N] or NULL (all scale 0); patch_size P = 16 | 8.  Outputs: patches [NI, N, 3, P, P],
 * pos [NI, N, 2] = clamp((sample + P/2) / (dim - P/2), 0, 1 - 1e-6), scales [NI, N] (fp32-cast ids, may be NULL). */
int  vtq_k_gather_patches(const float* const* levels, const int32_t* hs, const int32_t* ws, int32_t nlevels, const int32_t* samples,
                          const int32_t* scale_ids, float* patches, float* pos, float* scales, int32_t NI, int32_t N, int32_t patch_size,
                          void* stream);

/* ---- validation-loop reductions; fp64 like the reference's numpy arrays ------------------------------------ */
/* average_over_repeats (train.py:398-400): q fp32 [R, N] (repeat-major, as the concatenated passes of do_validation) ->
 * out fp64 [N] = mean over the R repeats, summed in repeat order. */
int  vtq_k_repeat_mean(const float* q, double* out, int32_t R, int32_t N, void* stream);
/* The fit-free part of compute_correlations (utils/misc/correlations.py:21-33) on two fp64 score vectors a, b [N]:
 *   work[0:N], work[N:2N]  = normalize_array(a), normalize_array(b) (image_tools.py:17-21; plain copies when normalize == 0)
 *   work[2N:3N], [3N:4N]   = their average-tie ranks
 *   counts[0] = 2 (concordant - discordant pairs), counts[1] = 2 (pairs tied in a), counts[2] = 2 (pairs tied in b)   (exact)
 *   out[0] = Spearman (Pearson of the ranks), out[1] = Pearson, out[2] = RMSE of the normalised vectors.
 * work: DEVICE fp64 [4N]; counts: DEVICE int64 [3]; out: DEVICE fp64 [3].  The host finishes Kendall's tau-b from the counts. */
int  vtq_k_rank_metrics(const double* a, const double* b, int32_t N, int32_t normalize, double* work, int64_t* counts, double* out,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VTAMIQ_HIP_H */
