// Micro-benchmark: issue cost of fp32 vector instructions in an MFMA-FREE stretch (the GEMM epilogues: GELU polynomial, hi / lo split),
// scalar against packed, one and two waves per SIMD.  MI355X_MICROARCH.md prices packed fp32 only BESIDE MFMAs (an anti-lever there);
// the fc1 epilogue is 24 % of a tile and bound by vector-instruction issue (profiles/r03_gemm_epilogue_stamps.txt), so what a
// v_pk_fma_f32 costs there against the two v_fma_f32 it replaces decides whether a packed GELU polynomial is worth building.
// Every stream is 16 independent chains (no dependent-issue stall), cycles from s_memtime around 64 x 16 instructions, per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_issue.hip -o tools/micro/valu_issue && tools/micro/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND 0: v_fma_f32 x 2 per step pair (32 scalar fmas = 32 values)   1: v_pk_fma_f32 (16 packed = 32 values)
//      2: v_pk_mul_f32   3: v_pk_add_f32   4: v_fmaak_f32 (literal constant, the shipped polynomial's form)   5: v_exp_f32
//      6: mixed as the shipped gelu_erf4 (per 4 values: 4 min, 24 fmaak, 4 fma, 4 exp, 4 max, 4 fma)   7: the same with the polynomial packed
template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned long long* cyc, float* out, int reps) {
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = 0.5f + 1e-3f * ((threadIdx.x + 7 * i) & 63);
    f32x2 p[16];
    for (int i = 0; i < 16; ++i) p[i] = f32x2{v[2 * i], v[2 * i + 1]};
    const float c1 = 0.9999999f, c2 = 1e-7f;
    const f32x2 c1p = {c1, c1}, c2p = {c2, c2};
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int r = 0; r < reps; ++r) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c1p), "v"(c2p));
        } else if constexpr (KIND == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c1p));
        } else if constexpr (KIND == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2p));
        } else if constexpr (KIND == 4) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fmaak_f32 %0, %1, %0, 0x33d6bf95" : "+v"(v[i]) : "v"(c1));
        } else if constexpr (KIND == 5) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(v[i]));
        } else if constexpr (KIND == 6) {
            // the shipped gelu_erf4 shape on 8 groups of 4 values (32 values): same instruction mix, chains in lock step
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float* x = v + 4 * g;
                float t[4], q[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_min_f32_e64 %0, |%1|, %2" : "=v"(t[e]) : "v"(x[e]), "s"(5.7f));
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_fmaak_f32 %0, %1, %2, 0x37814f5e" : "=v"(q[e]) : "v"(c2), "v"(t[e]));
#pragma unroll
                for (int s = 0; s < 5; ++s)
#pragma unroll
                    for (int e = 0; e < 4; ++e) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3a142202" : "+v"(q[e]) : "v"(t[e]));
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_fma_f32 %0, -%1, %0, -1.0" : "+v"(q[e]) : "v"(t[e]));
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(q[e]));
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_max_f32_e32 %0, 0, %1" : "=v"(t[e]) : "v"(x[e]));
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_fma_f32 %0, -|%1|, %2, %3" : "=v"(x[e]) : "v"(x[e]), "v"(q[e]), "v"(t[e]));
            }
        } else {
            // the same with the six polynomial steps and the -a q - 1 step as packed instructions (constants in register pairs)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float* x = v + 4 * g;
                f32x2 t[2], q[2];
                float m[4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float a, b;
                    asm volatile("v_min_f32_e64 %0, |%1|, %2" : "=v"(a) : "v"(x[2 * e]), "s"(5.7f));
                    asm volatile("v_min_f32_e64 %0, |%1|, %2" : "=v"(b) : "v"(x[2 * e + 1]), "s"(5.7f));
                    t[e] = f32x2{a, b};
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(q[e]) : "v"(c2p), "v"(t[e]), "v"(c1p));
#pragma unroll
                for (int s = 0; s < 5; ++s)
#pragma unroll
                    for (int e = 0; e < 2; ++e) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[e]) : "v"(t[e]), "v"(c1p));
#pragma unroll
                for (int e = 0; e < 2; ++e) asm volatile("v_pk_fma_f32 %0, %1, %0, %2 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(q[e]) : "v"(t[e]), "v"(c2p));
                float qs[4] = {q[0][0], q[0][1], q[1][0], q[1][1]};
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(qs[e]));
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_max_f32_e32 %0, 0, %1" : "=v"(m[e]) : "v"(x[e]));
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_fma_f32 %0, -|%1|, %2, %3" : "=v"(x[e]) : "v"(x[e]), "v"(qs[e]), "v"(m[e]));
            }
        }
    }
    asm volatile("s_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float r = 0;
    for (int i = 0; i < 32; ++i) r += v[i];
    for (int i = 0; i < 16; ++i) r += p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND> void run(const char* name, int values_per_rep, int instr_per_rep, unsigned long long* dcyc, float* dout) {
    for (int threads : {256, 512, 1024}) {               // one / two / four waves per SIMD
        const int wgs = 256, waves = wgs * threads / 64, reps = 64;
        hipLaunchKernelGGL((k<KIND>), dim3(wgs), dim3(threads), 0, 0, dcyc, dout, reps);
        hipLaunchKernelGGL((k<KIND>), dim3(wgs), dim3(threads), 0, 0, dcyc, dout, reps);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(waves);
        (void)hipMemcpy(h.data(), dcyc, waves * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        // s_memtime counts shader-clock cycles (s_memrealtime is the constant 100 MHz one)
        const double med = (double)h[waves / 2];
        printf("%-44s %d wave(s)/SIMD: %9.0f cycles per wave for %d instr (%d values) x %d  -> %.4f cycles / instr, %.4f cycles / value\n", name,
               threads / 256, med, instr_per_rep, values_per_rep, reps, med / (instr_per_rep * reps), med / (values_per_rep * reps));
    }
}

int main() {
    unsigned long long* dcyc; float* dout;
    (void)hipMalloc(&dcyc, 256 * 16 * 8); (void)hipMalloc(&dout, 256 * 1024 * 4);
    run<0>("v_fma_f32 (32 per rep)", 32, 32, dcyc, dout);
    run<4>("v_fmaak_f32 literal (32 per rep)", 32, 32, dcyc, dout);
    run<1>("v_pk_fma_f32 (16 per rep)", 32, 16, dcyc, dout);
    run<2>("v_pk_mul_f32 (16 per rep)", 32, 16, dcyc, dout);
    run<3>("v_pk_add_f32 (16 per rep)", 32, 16, dcyc, dout);
    run<5>("v_exp_f32 (32 per rep)", 32, 32, dcyc, dout);
    run<6>("gelu_erf4 shape, scalar polynomial", 32, 8 * 44, dcyc, dout);
    run<7>("gelu_erf4 shape, packed polynomial", 32, 8 * 30, dcyc, dout);
    return 0;
}
