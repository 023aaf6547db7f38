// Micro-benchmark: cycles per v_mfma_f32_16x16x32_f16 for ONE wave per SIMD issuing the whole-row GEMM's statement forms
// (tools/micro/mfma_stmt.hip; hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form).  Operands live in registers; 96 accumulator
// blocks (64 in AGPRs, 32 in VGPRs) as in csrc/gemm_rowln.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MM3 "v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %4, %0\n\tv_mfma_f32_16x16x32_f16 %0, %2, %3, %0"
#define MM6 "v_mfma_f32_16x16x32_f16 %0, %2, %4, %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %6, %1\n\tv_mfma_f32_16x16x32_f16 %0, %2, %5, %0\n\t" \
            "v_mfma_f32_16x16x32_f16 %1, %2, %7, %1\n\tv_mfma_f32_16x16x32_f16 %0, %3, %4, %0\n\tv_mfma_f32_16x16x32_f16 %1, %3, %6, %1"

// MODE 0: three dependent MFMAs per statement, AGPR accumulators, no nops      1: + "s_nop 1" in front
//      2: pairs interleaved (6 per statement), no nops                         3: + "s_nop 1" in front
//      4: pairs, VGPR accumulators, no nops                                    5: pairs, VGPR, s_nop 1 + s_nop 7
//      6: pairs, AGPR, a v_add_u32 + s_add between statements (filler)          7: singles: one MFMA per statement, AGPR
template <int MODE>
__global__ __launch_bounds__(256, 1) void kern(float* out, unsigned long long* cyc, int iters) {
    u32x4 wh, wl, ah[8], al[8];
    for (int k = 0; k < 4; ++k) { wh[k] = 0x3c003c00u + threadIdx.x * 7 + k; wl[k] = 0x1c001c00u + k; }
    for (int i = 0; i < 8; ++i) for (int k = 0; k < 4; ++k) { ah[i][k] = 0x38003a00u + i * 131 + k + threadIdx.x; al[i][k] = 0x18001a00u + i + k; }
    f32x4 acc[8][12];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 12; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int filler = threadIdx.x;
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            constexpr bool kAllV = (MODE == 4 || MODE == 5);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool ag = !kAllV && j < 8;
                if constexpr (MODE == 0 || MODE == 1) {
                    if (ag) { if (MODE == 1) asm volatile("s_nop 1\n\t" MM3 : "+a"(acc[i][j]) : "v"(wh), "v"(wl), "v"(ah[i]), "v"(al[i]));
                              else asm volatile(MM3 : "+a"(acc[i][j]) : "v"(wh), "v"(wl), "v"(ah[i]), "v"(al[i])); }
                    else asm volatile(MM3 "\n\ts_nop 7" : "+v"(acc[i][j]) : "v"(wh), "v"(wl), "v"(ah[i]), "v"(al[i]));
                } else if constexpr (MODE == 7) {
                    if (ag) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(wh), "v"(ah[i]));
                              asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(wh), "v"(al[i]));
                              asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(wl), "v"(ah[i])); }
                    else asm volatile(MM3 "\n\ts_nop 7" : "+v"(acc[i][j]) : "v"(wh), "v"(wl), "v"(ah[i]), "v"(al[i]));
                } else if ((i & 1) == 0) {
                    if (ag) {
                        if (MODE == 3) asm volatile("s_nop 1\n\t" MM6 : "+a"(acc[i][j]), "+a"(acc[i + 1][j]) : "v"(wh), "v"(wl), "v"(ah[i]), "v"(al[i]), "v"(ah[i + 1]), "v"(al[i + 1]));
                        else asm volatile(MM6 : "+a"(acc[i][j]), "+a"(acc[i + 1][j]) : "v"(wh), "v"(wl), "v"(ah[i]), "v"(al[i]), "v"(ah[i + 1]), "v"(al[i + 1]));
                    } else {
                        if (MODE == 4) asm volatile(MM6 : "+v"(acc[i][j]), "+v"(acc[i + 1][j]) : "v"(wh), "v"(wl), "v"(ah[i]), "v"(al[i]), "v"(ah[i + 1]), "v"(al[i + 1]));
                        else asm volatile("s_nop 1\n\t" MM6 "\n\ts_nop 7" : "+v"(acc[i][j]), "+v"(acc[i + 1][j]) : "v"(wh), "v"(wl), "v"(ah[i]), "v"(al[i]), "v"(ah[i + 1]), "v"(al[i + 1]));
                    }
                    if (MODE == 6) asm volatile("v_add_u32 %0, %0, %1\n\ts_nop 0" : "+v"(filler) : "v"(filler));
                }
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = (float)filler;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 12; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE> void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 60;
    hipLaunchKernelGGL(kern<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[1024]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < 1024; ++i) sum += (double)c[i];
    const double n = 288.0 * iters;
    printf("%-66s %6.2f cycles/MFMA (mean over 1024 waves), %6.2f ns/MFMA by events -> %.2f GHz\n", name, sum / 1024 / n, ms * 1e6 / n, (sum / 1024 / n) / (ms * 1e6 / n));
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    run<0>("0: 3 dependent MFMAs / statement, AGPR+VGPR blocks, no front nop", out, cyc);
    run<1>("1: same + s_nop 1 in front", out, cyc);
    run<2>("2: pairs (6 / statement), no front nop; VGPR blocks s_nop 1 + 7", out, cyc);
    run<3>("3: pairs + s_nop 1 in front of the AGPR ones too", out, cyc);
    run<4>("4: pairs, ALL accumulators in VGPRs (can't: 384 regs) -> see note", out, cyc);
    run<6>("6: pairs + one VALU + s_nop 0 behind every statement", out, cyc);
    run<7>("7: one MFMA per statement (AGPR blocks)", out, cyc);
    return 0;
}
