// Device-side helpers shared by the gfx950 kernels.  CDNA4 only: wave = 64 lanes, MFMA bf16, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 16-byte async global -> LDS copy (global_load_lds_dwordx4).  LDS destination = wave-uniform base + lane*16;
// the per-lane part is the SOURCE address (cdna_hip_programming.md section 5 caveat).
// 16-byte streaming store (nt policy): outputs that the next kernel re-reads from HBM anyway; measured -3..5 % on the
// GEMM epilogue's write burst versus the default write-back policy.
__device__ __forceinline__ void store_nt16(void* dst, uint4 v) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, (u32x4_t*)dst);
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }

// hi = RNE(v); lo = RNE(v - hi): v ~= hi + lo to ~16 mantissa bits.
__device__ __forceinline__ void split2(float v, bf16& hi, bf16& lo) {
    hi = (bf16)v;
    lo = (bf16)(v - (float)hi);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// exact-erf GELU, the default of torch.nn.functional.gelu (transformer.py:54-57).
// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32 rounding level on the activation) with the
// hardware rcp/exp2: ~14 VALU ops instead of libm erff's ~45 -- the fc1 epilogue evaluates 10^8 of these per layer.
__device__ __forceinline__ float gelu_erf(float x) {
#ifdef VTQ_LIBM_ERF
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
#else
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float hc = 0.5f * x * (poly * t * e);                 // 0.5 x erfc(|x|/sqrt2): no cancellation in either tail
    return x >= 0.f ? x - hc : hc;
#endif
}

#define VTQ_WAVE 64
