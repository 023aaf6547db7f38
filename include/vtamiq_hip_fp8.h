/*
 * vtamiq_hip_fp8.h -- the fp8 EXPERIMENT of libvtamiq_hip (BASELINE.json configs[4]: "fp8 weights on CDNA4 fp8 MFMA").
 *
 * NOT part of the product library.  It is a THROUGHPUT experiment, not a scoring mode: 3 mantissa bits of activation precision put the
 * scores tens of percent from the fp32 model's on random-init weights and SROCC 0.66 - 0.84 from it on a distortion ladder with every scale
 * granularity tried (profiles/r04_fp8_study.txt); its parity statement is against its own fake-quant oracle (oracle/fp8_oracle.py), which
 * nothing from the reference pins.  Round 5 therefore moved it out of the shipped .so, out of include/vtamiq_hip.h and out of
 * vtamiq_amd.VTAMIQ: the symbols below exist only in a library built with -DVTQ_WITH_FP8
 *     python -m vtamiq_amd.build --fp8        ->  vtamiq_amd/libvtamiq_hip_fp8.so   (load it with VTQ_LIB_PATH=...)
 * and the host side is vtamiq_amd/experimental_fp8.py (class VTAMIQFp8).  bench.py reports the mode's throughput only when run on that build.
 *
 * The mode: linear layers on OCP e4m3 operands with the MX-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales, 2x the bf16
 * MFMA rate): weights e4m3 with per-output-channel power-of-two scales, activations e4m3 with calibrated per-tensor scales; attention
 * single fp16; DiffNet head fp16 hi/lo.
 */
#ifndef VTAMIQ_HIP_FP8_H
#define VTAMIQ_HIP_FP8_H

#include "vtamiq_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VTQ_PREC_FP8    5             /* vtq_config.precision */
#define VTQ_NUM_FP8    33             /* operand-format code: e4m3 bytes, one plane each (vtq_k_gemm_fp8) */
#define VTQ_OPT_FP8_STATIC_SCALES 2   /* vtq_config.options: keep the static default activation scales, never calibrate */
/* vtq_input_errors bit 2: an activation times its scale exceeded e4m3's largest value and was clamped to +-448 (the scales no longer fit
 * the data: calibrate again, vtq_fp8_calibrate). */

/* VTQ_PREC_FP8: per-tensor power-of-two activation scales, one per quantisation point -- [0] the packed patches, then for every
 * layer l: [1 + 4l] LayerNorm-1 output, [2 + 4l] attention context, [3 + 4l] LayerNorm-2 output, [4 + 4l] GELU output.
 * An engine starts with static defaults (256, 8, 16, 8, 4) and CALIBRATES on the batch of its first vtq_forward: every producing
 * kernel reports max |value|, the largest power of two mapping it to <= 224 becomes the scale, the producer is run again with
 * it (one stream synchronisation per point, that forward only; its scores are computed with the final scales).  vtq_config.options
 * & VTQ_OPT_FP8_STATIC_SCALES keeps the defaults.  vtq_load_weights marks the engine uncalibrated again (scales fitted to other
 * weights would clamp) unless the current scales were installed through vtq_fp8_set_scales.  vtq_fp8_calibrate repeats the calibration on a batch of the caller's choice
 * (arguments as vtq_forward); get / set expose the 1 + 4 * num_layers values (get returns the count, -1 for a non-fp8 engine;
 * set requires positive powers of two and marks the engine calibrated). */
int  vtq_fp8_calibrate(vtq_handle h, const float* patches_ref, const float* patches_dist, const float* pos_ref, const float* pos_dist,
                       const float* scales_ref, const float* scales_dist, int32_t B, int32_t N, float* q_out, void* stream);
int  vtq_fp8_get_scales(vtq_handle h, float* out, int32_t cap);
int  vtq_fp8_set_scales(vtq_handle h, const float* scales, int32_t n);
/* Forget the current scales (calibrated or installed): the next vtq_forward calibrates on its own batch again.  The host calls this when
 * it drops scales it had saved for an engine whose weights changed (a re-installed set must not outlive the weights it fits). */
int  vtq_fp8_reset(vtq_handle h);

/* fp8 (VTQ_PREC_FP8) building blocks.  vtq_k_quant_rows_fp8: W[N][K] fp32 -> e4m3 rows, each scaled by the largest power of two
 * that keeps its maximum <= 448, inv_scale[n] = 1 / scale.  vtq_k_quant_fp8: e4m3(src * scale), clamped to +-448.
 * vtq_k_gemm_fp8: C = A8[M,K] * W8[N,K]^T on v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales), acc * wscale[n] * ascale_inv,
 * then the epilogue: 0 -> out = fp16 (one plane) of (v + bias); 1 -> out = e4m3(gelu(v + bias) * out_scale), ldo bytes per row;
 * 2 -> x_f32 += gamma * (v + bias).  M%256==0, N%256==0, K%256==0. */
int  vtq_k_quant_rows_fp8(const float* W, void* dst, float* inv_scale, int32_t N, int32_t K, void* stream);
int  vtq_k_quant_fp8(const float* src, void* dst, int64_t numel, float scale, void* stream);
int  vtq_k_gemm_fp8(const void* A8, int32_t lda, const void* W8, const float* wscale, float ascale_inv, int32_t M, int32_t N, int32_t K,
                    int32_t epilogue, const float* bias, const float* gamma, float* x_f32, void* out, int64_t o_plane, int32_t ldo,
                    float out_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
