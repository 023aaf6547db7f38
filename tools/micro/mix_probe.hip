// Probe (GPU box): split4_f16 (v_cvt_pk_f16_f32 + v_fma_mixlo/mixhi_f16, below) against split2 (cast, subtract, cast; dev_common.h) bit for bit:
// random values over the fp16 range, exact ties of the hi rounding, values whose lo part is an fp16 subnormal, zeros, negatives.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vtamiq_amd/csrc tools/micro/mix_probe.hip -o tools/micro/mix_probe && tools/micro/mix_probe
#include "dev_common.h"

// EXPERIMENT RECORD (round 3, commit 80f4162 had this in the GEMM epilogue and, as an 8-value form, in the attention softmax):
// bit-exact (0 mismatches on 4M values) and 11 % fewer vector instructions in the fc1 epilogue, but no measurable time saved
// (profiles/r03_valu_diet_ab.txt), so the kernels keep the plain C++ split.
// The same split for FOUR values of the f16 format in 6 vector instructions instead of 14: two v_cvt_pk_f16_f32 for the hi pairs
// (RNE, like the cast), and lo = RNE(v - hi) by v_fma_mixlo/mixhi_f16 -- fma(v, 1.0, -hi) with hi read straight from its f16
// half, computed in fp32 and rounded once into the low / high half of the destination: bit for bit what split2 gives
// (checked below on the device, ties and subnormals included).  The epilogues and the attention
// softmax are bound by vector-instruction ISSUE (one per ~4-5 cycles per SIMD for its two waves together, measured with
// in-kernel stamps: profiles/r03_gemm_epilogue_stamps.txt), so instructions saved there are time saved.
// The two halves of one destination are written by instructions two apart (partial-register write followed by a read of the
// same register needs a wait state on gfx950: LLVM's dst-sel forwarding hazard; inside an asm statement nothing is padded).
// The inputs must not come straight from a transcendental (v_exp ...): its result may not be read by the next vector
// instruction (trans forwarding hazard) and hipcc pads nothing for asm operands -- callers pass values that went through
// an ordinary VALU instruction, or make the statement depend on one that did (`after`).
__device__ __forceinline__ void split4_f16(const float (&v)[4], f16x4& hi, f16x4& lo) {
    // the hi conversions sit inside the statement too: left to hipcc, the cast is contracted with the multiply-add that
    // produced v (v_fma_mixlo_f16 on the unrounded product) -- the double-rounding trap described at split2
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

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k(const float* in, uint32_t* a, uint32_t* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * i + 3 >= n) return;
    float v[4];
    for (int j = 0; j < 4; ++j) { v[j] = in[4 * i + j]; asm volatile("" : "+v"(v[j])); }
    f16x4 h, l;
    split4_f16(v, h, l);
    typedef __attribute__((ext_vector_type(2))) uint32_t u2;
    const u2 hb = __builtin_bit_cast(u2, h), lb = __builtin_bit_cast(u2, l);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f16 x, y;
        split2<f16>(v[j], x, y);
        const uint32_t hj = (hb[j >> 1] >> (16 * (j & 1))) & 0xffffu, lj = (lb[j >> 1] >> (16 * (j & 1))) & 0xffffu;
        a[4 * i + j] = (hj << 16) | lj;
        b[4 * i + j] = ((uint32_t)__builtin_bit_cast(unsigned short, x) << 16) | __builtin_bit_cast(unsigned short, y);
    }
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const int kind = i & 7;
        const float u = (float)rand() / RAND_MAX, w = (float)rand() / RAND_MAX;
        float v;
        if (kind == 0) v = (u - 0.5f) * 2.0f;                                   // [-1, 1]
        else if (kind == 1) v = (u - 0.5f) * 131000.0f;                          // the fp16 range
        else if (kind == 2) v = ldexpf(1.0f + floorf(u * 1024.0f) / 1024.0f + 1.0f / 2048.0f, (int)(w * 30) - 15);   // exact ties of the hi rounding
        else if (kind == 3) v = ldexpf(u, -14 - (int)(w * 12));                  // hi (and lo) in the fp16 subnormal range
        else if (kind == 4) v = (i & 8) ? 0.0f : -0.0f;
        else if (kind == 5) v = ldexpf(1.0f + u, (int)(w * 8) - 4) * ((i & 16) ? -1.f : 1.f);
        else if (kind == 6) v = expf(-20.0f * u);                                // softmax-like
        else v = (u - 0.5f) * 20.0f;
        h[i] = v;
    }
    float* d; uint32_t *da, *db;
    (void)hipMalloc(&d, n * 4); (void)hipMalloc(&da, n * 4); (void)hipMalloc(&db, n * 4);
    (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 4 / 256), dim3(256), 0, 0, d, da, db, n);
    std::vector<uint32_t> a(n), b(n);
    (void)hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i)
        if (a[i] != b[i]) { if (bad < 10) printf("MISMATCH v=%.9g  mix hi/lo %04x %04x  split2 %04x %04x\n", h[i], a[i] >> 16, a[i] & 0xffff, b[i] >> 16, b[i] & 0xffff); ++bad; }
    printf("split4_f16 vs split2 on %d values: %ld mismatches\n", n, bad);
    return bad != 0;
}
