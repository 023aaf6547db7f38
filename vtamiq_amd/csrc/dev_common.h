// Device-side helpers shared by the gfx950 kernels.  CDNA4 only: wave = 64 lanes, MFMA bf16 / f16, LDS-DMA.
#pragma once
#define VTQ_DEV_COMMON 1
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

typedef __bf16 bf16;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int v8i;
struct f8 { uint8_t v; };      // one OCP e4m3 operand byte (the "fp8" numerics mode: MX-scaled MFMA with unit block scales)

// 16-bit MFMA operand element: bf16 (8 significand bits, fp32 range) or f16 (11 bits, max 65504 -- the range the reference's
// own GPU path runs these contractions in: torch.cuda.amp.autocast(float16), train.py:602).  Same MFMA rate on gfx950.
template <typename T> struct Vec;
template <> struct Vec<bf16> { typedef bf16x8 x8; typedef bf16x4 x4; };
template <> struct Vec<f16> { typedef f16x8 x8; typedef f16x4 x4; };
template <> struct Vec<f8> { typedef v8i x8; typedef uint32_t x4; };     // x8: one 16x16x128 MFMA fragment (32 bytes); x4: 4 packed bytes

template <typename T>
__device__ __forceinline__ f32x4 mfma16(typename Vec<T>::x8 a, typename Vec<T>::x8 b, f32x4 c) {
    if constexpr (std::is_same<T, bf16>::value) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
template <typename T>
__device__ __forceinline__ f32x16 mfma32(typename Vec<T>::x8 a, typename Vec<T>::x8 b, f32x16 c) {
    if constexpr (std::is_same<T, bf16>::value) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// 16x16x128 e4m3 x e4m3 with unit block scales (E8M0 byte 127 = 2^0): twice the bf16 MFMA rate.  D rows come from the first
// operand; both operands hold row (lane & 15), bytes 32 * (lane >> 4) .. + 31 of a K = 128 row (tools/micro/mx_probe.hip).
__device__ __forceinline__ f32x4 mfma_f8(v8i a, v8i b, f32x4 c) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

// 4 floats -> 4 e4m3 bytes (RNE, subnormals kept; the conversion turns |x| >= 464 into NaN, so clamp to the format's 448 first:
// the same two steps as torch's clamp + .to(float8_e4m3fn) in oracle/fp8_oracle.py)
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    return (uint32_t)p;
}

// max |.| of four values folded into a running maximum (NaNs are ignored by v_max)
__device__ __forceinline__ float amax4(float m, float a, float b, float c, float d) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(a), fabsf(b))), fmaxf(fabsf(c), fabsf(d)));
}

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 16-byte streaming store (nt policy): outputs that the next kernel re-reads from HBM anyway; measured -3..5 % on the
// GEMM epilogue's write burst versus the default write-back policy.
__device__ __forceinline__ void store_nt16(void* dst, uint4 v) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, (u32x4_t*)dst);
}

// 16-byte async global -> LDS copy (global_load_lds_dwordx4).  LDS destination = wave-uniform base + lane*16;
// the per-lane part is the SOURCE address (cdna_hip_programming.md section 5 caveat).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }

// hi = RNE(v); lo = RNE(v - hi): v ~= hi + lo to ~16 (bf16) / ~22 (f16) significand bits.  f16 subnormals are kept by the
// conversion and by the MFMA (tools/micro/f16_probe.hip), so no pre-scaling is needed for small values.
// The value is made opaque first: with fp contraction hipcc otherwise forms hi twice -- once from the fp32-rounded v (the copy
// that is stored) and once fused with the producing multiply (v_fma_mixlo_f16 on the exact product, the copy lo is taken
// against); at f16 ties the two differ by one ulp and lo gets the wrong sign (measured: 1e-5 of attention outputs off by an ulp).
template <typename T>
__device__ __forceinline__ void split2(float v, T& hi, T& lo) {
    asm volatile("" : "+v"(v));
    hi = (T)v;
    lo = (T)(v - (float)hi);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// exact-erf GELU, the default of torch.nn.functional.gelu (transformer.py:54-57), to fp32 rounding level:
//   gelu(x) = max(x, 0) - |x| g(|x|),   g(a) = 0.5 erfc(a / sqrt 2) = 2^(-1 - a Q(a))
// (one form for both signs, no cancellation in either tail).  Q is a degree-6 polynomial fitted on a in [0, 5.7] to the
// weighted minimax of the GELU's absolute error (tools/fit_gelu.py): |gelu - exact| <= 2.8e-7 over x in [-8, 8], i.e. half an
// fp32 ulp of the result where the bound is attained (x ~ 4.2).  Beyond a = 5.7 g is held at g(5.7) = 6e-9 (relative error of
// the result < 1e-8).  9 FMA / min / max + 1 transcendental (exp2) per element -- the round-1 form (Abramowitz-Stegun 7.1.26:
// rcp + exp2 + 11 VALU) cost 19 issue slots against these 13, and the fc1 epilogue evaluates 10^8 of these per layer, VALU-bound.
__device__ __forceinline__ float gelu_erf(float x) {
#ifdef VTQ_LIBM_ERF
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
#else
    const float ax = fabsf(x);
    const float a = fminf(ax, 5.7f);
    float q = fmaf(-4.525732038018759e-06f, a, 1.5414956578752026e-05f);
    q = fmaf(q, a, 0.0005650819512084126f);
    q = fmaf(q, a, -0.007649366278201342f);
    q = fmaf(q, a, 0.05292675271630287f);
    q = fmaf(q, a, 0.45905444025993347f);
    q = fmaf(q, a, 1.1511250734329224f);
    const float g = __builtin_amdgcn_exp2f(fmaf(-a, q, -1.0f));
    return fmaf(-ax, g, fmaxf(x, 0.0f));
#endif
}

// gelu_erf on FOUR values, the same operations in the same order (bit-identical results), as one hand-ordered instruction block:
// the four polynomial chains advance in lock step, so every instruction's operand was produced four instructions earlier (no
// dependent-issue stall, no s_nop behind the v_exp: hipcc, left alone, either serialises the chains -- one element at a time with
// an s_nop after every v_exp -- or packs pairs into v_pk_*_f32, which issue slower than what they replace), and the clamp / max
// are single instructions on the raw MFMA accumulators (as C++ fminf / fmaxf they each get a canonicalising v_max in front).
// 11 instructions per value.  The GEMM epilogue is bound by vector-instruction issue (profiles/r03_gemm_epilogue_stamps.txt).
#ifndef VTQ_GELU_PACKED
#define VTQ_GELU_PACKED 0             // measurement builds: the polynomial as v_pk_fma_f32 (same operations, same bits); profiles/r05_gelu_packed.txt
#endif
#if VTQ_GELU_PACKED
// The same value two at a time: the six polynomial steps and -a q - 1 as seven v_pk_fma_f32 per PAIR, clamp / exp2 / max / last fma as before:
// 30 instructions per 4 values instead of 44, ONE statement (no compiler-placed nops between its parts).  The pairs live in v[248:255], named
// as clobbers (an asm operand has no sub-register syntax, and the scalar instructions address the halves): the kernel's allocation becomes 256
// registers, which the two-waves-per-SIMD GEMMs have.  Constants: the low dword of an SGPR pair, broadcast to both halves by op_sel_hi; made
// opaque HERE so that they are not materialised above the main loop and kept (spilled) across it.
__device__ __forceinline__ void gelu_erf4(float (&x)[4]) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    uint64_t k0 = 0x37814f5eu, k1 = 0x3a142202u, k2 = 0xbbfaa789u, k3 = 0x3d58c9b9u, k4 = 0x3eeb092fu, k5 = 0x3f935811u, km1 = 0xbf800000u;
    asm volatile("" : "+s"(k0), "+s"(k1), "+s"(k2), "+s"(k3), "+s"(k4), "+s"(k5), "+s"(km1));
    const f2 c6 = {-4.525732038018759e-06f, -4.525732038018759e-06f};      // in a register pair: an SGPR beside k0 would be two constant-bus operands
    float r0, r1, r2, r3;
    asm("v_min_f32_e64 v248, |%4|, %8\n\tv_min_f32_e64 v249, |%5|, %8\n\tv_min_f32_e64 v250, |%6|, %8\n\tv_min_f32_e64 v251, |%7|, %8\n\t"
        "v_pk_fma_f32 v[252:253], %16, v[248:249], %9 op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[254:255], %16, v[250:251], %9 op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 v[252:253], v[252:253], v[248:249], %10 op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[254:255], v[254:255], v[250:251], %10 op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 v[252:253], v[252:253], v[248:249], %11 op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[254:255], v[254:255], v[250:251], %11 op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 v[252:253], v[252:253], v[248:249], %12 op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[254:255], v[254:255], v[250:251], %12 op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 v[252:253], v[252:253], v[248:249], %13 op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[254:255], v[254:255], v[250:251], %13 op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 v[252:253], v[252:253], v[248:249], %14 op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[254:255], v[254:255], v[250:251], %14 op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 v[252:253], v[248:249], v[252:253], %15 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 v[254:255], v[250:251], v[254:255], %15 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_exp_f32_e32 v252, v252\n\tv_exp_f32_e32 v253, v253\n\tv_exp_f32_e32 v254, v254\n\tv_exp_f32_e32 v255, v255\n\t"
        "v_max_f32_e32 v248, 0, %4\n\tv_max_f32_e32 v249, 0, %5\n\tv_max_f32_e32 v250, 0, %6\n\tv_max_f32_e32 v251, 0, %7\n\t"
        "v_fma_f32 %0, -|%4|, v252, v248\n\tv_fma_f32 %1, -|%5|, v253, v249\n\tv_fma_f32 %2, -|%6|, v254, v250\n\tv_fma_f32 %3, -|%7|, v255, v251"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "s"(5.7f), "s"(k0), "s"(k1), "s"(k2), "s"(k3), "s"(k4), "s"(k5), "s"(km1), "v"(c6)
        : "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
    x[0] = r0; x[1] = r1; x[2] = r2; x[3] = r3;
}
#else
__device__ __forceinline__ void gelu_erf4(float (&x)[4]) {
    float t0, t1, t2, t3, q0, q1, q2, q3;
    const float c6 = -4.525732038018759e-06f, c57 = 5.7f;
    asm("v_min_f32_e64 %0, |%8|, %13\n\tv_min_f32_e64 %1, |%9|, %13\n\tv_min_f32_e64 %2, |%10|, %13\n\tv_min_f32_e64 %3, |%11|, %13\n\t"
        "v_fmaak_f32 %4, %12, %0, 0x37814f5e\n\tv_fmaak_f32 %5, %12, %1, 0x37814f5e\n\tv_fmaak_f32 %6, %12, %2, 0x37814f5e\n\tv_fmaak_f32 %7, %12, %3, 0x37814f5e\n\t"
        "v_fmaak_f32 %4, %4, %0, 0x3a142202\n\tv_fmaak_f32 %5, %5, %1, 0x3a142202\n\tv_fmaak_f32 %6, %6, %2, 0x3a142202\n\tv_fmaak_f32 %7, %7, %3, 0x3a142202\n\t"
        "v_fmaak_f32 %4, %4, %0, 0xbbfaa789\n\tv_fmaak_f32 %5, %5, %1, 0xbbfaa789\n\tv_fmaak_f32 %6, %6, %2, 0xbbfaa789\n\tv_fmaak_f32 %7, %7, %3, 0xbbfaa789\n\t"
        "v_fmaak_f32 %4, %4, %0, 0x3d58c9b9\n\tv_fmaak_f32 %5, %5, %1, 0x3d58c9b9\n\tv_fmaak_f32 %6, %6, %2, 0x3d58c9b9\n\tv_fmaak_f32 %7, %7, %3, 0x3d58c9b9\n\t"
        "v_fmaak_f32 %4, %4, %0, 0x3eeb092f\n\tv_fmaak_f32 %5, %5, %1, 0x3eeb092f\n\tv_fmaak_f32 %6, %6, %2, 0x3eeb092f\n\tv_fmaak_f32 %7, %7, %3, 0x3eeb092f\n\t"
        "v_fmaak_f32 %4, %4, %0, 0x3f935811\n\tv_fmaak_f32 %5, %5, %1, 0x3f935811\n\tv_fmaak_f32 %6, %6, %2, 0x3f935811\n\tv_fmaak_f32 %7, %7, %3, 0x3f935811\n\t"
        "v_fma_f32 %4, -%0, %4, -1.0\n\tv_fma_f32 %5, -%1, %5, -1.0\n\tv_fma_f32 %6, -%2, %6, -1.0\n\tv_fma_f32 %7, -%3, %7, -1.0\n\t"
        "v_exp_f32_e32 %4, %4\n\tv_exp_f32_e32 %5, %5\n\tv_exp_f32_e32 %6, %6\n\tv_exp_f32_e32 %7, %7\n\t"
        "v_max_f32_e32 %0, 0, %8\n\tv_max_f32_e32 %1, 0, %9\n\tv_max_f32_e32 %2, 0, %10\n\tv_max_f32_e32 %3, 0, %11\n\t"
        "v_fma_f32 %4, -|%8|, %4, %0\n\tv_fma_f32 %5, -|%9|, %5, %1\n\tv_fma_f32 %6, -|%10|, %6, %2\n\tv_fma_f32 %7, -|%11|, %7, %3"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(c6), "s"(c57));       // c6 in a VGPR: an SGPR beside the literal would be two constant-bus operands
    x[0] = q0; x[1] = q1; x[2] = q2; x[3] = q3;
}
#endif

// hi = RNE(v), lo = RNE(v - hi) for FOUR values of the f16 format in 6 instructions: two v_cvt_pk_f16_f32, then
// lo = fma(v, 1.0, -hi) by v_fma_mixlo / mixhi_f16 with hi read from its f16 half (computed in fp32, rounded once) -- bit for bit
// what split2 gives (tools/micro/mix_probe.hip: 0 mismatches on 4M values incl. ties and subnormals).  The conversions sit inside
// the statement too: left to hipcc the cast is contracted with the multiply-add that produced v (the double-rounding trap
// described at split2).  The two halves of one destination are written two instructions apart (partial-register write, then a
// read-modify-write of the same register: gfx950 dst-sel forwarding hazard, nothing is padded inside an asm statement).
__device__ __forceinline__ void split4_f16(const float (&v)[4], f16x4& hi, f16x4& lo) {
    uint32_t h0, h1, l0, l1;
    asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_cvt_pk_f16_f32 %1, %6, %7\n\t"
        "v_fma_mixlo_f16 %2, %4, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %6, 1.0, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %2, %5, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %7, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    typedef __attribute__((ext_vector_type(2))) uint32_t u2;
    hi = __builtin_bit_cast(f16x4, u2{h0, h1});
    lo = __builtin_bit_cast(f16x4, u2{l0, l1});
}

// N x ds_read_b128 (N = 2 | 4, rows STEP bytes apart) issued back to back and waited for inside ONE statement, hidden from hipcc's
// waitcnt bookkeeping.  Why: beside an LDS-DMA in flight (the GEMM's chained prefetch of the next tile) hipcc puts
// `s_waitcnt vmcnt(0)` in front of every ordinary LDS read (the DMA might write what is read) -- a drain of the prefetch AND of the
// previous interval's output stores -- and emits read / wait / store one after the other: 4 exposed LDS latencies + a memory
// round trip per epilogue interval (profiles/r03_gemm_epilogue_ablation.txt).  The image read here was written by ds_write
// before the workgroup barrier the caller has passed; the LDS-DMA never targets it while it is live.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
// V = u32x4 | f32x4: take the element type the values are USED as (hipcc miscompiles __builtin_bit_cast(float, v[k][j]) on a u32x4
// statement output -- every element becomes element 0, seen in the residual epilogue, ROCm 7.2)
template <int STEP, typename V>
__device__ __forceinline__ void lds_read_rows4(const void* p, V (&v)[4]) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%c5\n\tds_read_b128 %2, %4 offset:%c6\n\tds_read_b128 %3, %4 offset:%c7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(lds_addr(p)), "i"(STEP), "i"(2 * STEP), "i"(3 * STEP) : "memory");
}
template <int STEP, typename V>
__device__ __forceinline__ void lds_read_rows2(const void* p, V (&v)[2]) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%c3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]) : "v"(lds_addr(p)), "i"(STEP) : "memory");
}

#define VTQ_WAVE 64

#ifdef __HIPCC__
// End of an fp8 producer (kernels.h Fp8Obs): m = this lane's max |value| before scaling.  One atomic per wave, and only when asked.
template <typename Obs>
__device__ __forceinline__ void fp8_report(const Obs& o, float m, float scale) {
    if (!o.amax && !o.err) return;                       // kernel argument: wave-uniform
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) {
        if (o.amax) atomicMax((int*)o.amax, __float_as_int(m));
        if (o.err && m * scale > 448.0f) atomicOr(o.err, 4);
    }
}
#endif

