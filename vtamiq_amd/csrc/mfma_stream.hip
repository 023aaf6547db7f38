// The practical ceiling of the matrix pipe, measured on THIS device (BASELINE.json "% bf16 MFMA roofline").
//
// A bare stream of independent back-to-back mfma_f32_16x16x32 on register operands -- no LDS, no memory -- on every CU (one workgroup of 8
// waves per CU, two waves per SIMD), with the operand BITS of a 3-term product: of every three MFMAs one multiplies hi x hi planes, one
// hi x lo, one lo x hi of gaussian data (the all-CU MFMA clock is power-bound and follows the operand contents: MI355X_MICROARCH.md 'DVFS
// give-back'; profiles/r04_mfma_power.txt: zeros 2.3 GHz, random fp16 1.75 - 1.82 GHz, this mix 1.85 GHz).  The A operand is held over four
// consecutive MFMAs, as the GEMM K loops hold theirs.  What this stream sustains is the most MFMA issue any kernel can get from the chip
// with such operands before a single byte has moved; bench.py prints it beside the nominal peak (roofline.practical_peak_tflops_measured_here).
// Measurement only: nothing on the forward path calls it.
#include <chrono>
#include <mutex>

#include "dev_common.h"
#include "kernels.h"

namespace vtq {
namespace {

__device__ inline unsigned ms_rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ inline float ms_gauss(unsigned& s) {
    float a = 0.f;
    for (int i = 0; i < 12; ++i) a += (ms_rnd(s) >> 8) * (1.f / 16777216.f);
    return a - 6.f;
}

// data: 0 = the 3-term mix of gaussian hi / lo planes, 1 = zeros (the issue limit), 2 = uniform random single planes
template <typename T>
__global__ __launch_bounds__(512) void mfma_stream_kernel(float* out, int iters, int data) {
    typedef typename Vec<T>::x8 tx8;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 12345u;
    tx8 A[3][4], B[3][4];
    for (int f = 0; f < 4; ++f) {
        tx8 ahi, alo, bhi, blo;
        for (int i = 0; i < 8; ++i) {
            float x, y;
            if (data == 1) { x = 0.f; y = 0.f; }
            else if (data == 2) { x = ((int)(ms_rnd(s) >> 8) % 2001 - 1000) * 1e-3f; y = ((int)(ms_rnd(s) >> 8) % 2001 - 1000) * 1e-3f; }
            else { x = ms_gauss(s); y = 0.02f * ms_gauss(s); }
            ahi[i] = (T)x; alo[i] = (T)(x - (float)ahi[i]);
            bhi[i] = (T)y; blo[i] = (T)(y - (float)bhi[i]);
        }
        A[0][f] = ahi; B[0][f] = bhi;
        A[1][f] = ahi; B[1][f] = (data == 0) ? blo : bhi;
        A[2][f] = (data == 0) ? alo : ahi; B[2][f] = bhi;
    }
    f32x4 acc[16];
    for (int n = 0; n < 16; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int h = 0; h < 8; ++h) {                 // 24 MFMAs per iteration
                const int n = (t * 8 + h) & 15;
                acc[n] = mfma16<T>(A[t][h >> 2], B[t][h & 3], acc[n]);
            }
    }
    float r = 0.f;
    for (int n = 0; n < 16; ++n) r += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// Where do the workgroups of a launch land?  One workgroup per CU (144 KiB of LDS), each records XCC id and HW_ID (SE / SH / CU fields) and
// stays for ~spin_us so that all of them are resident together.  tools/cu_partition.py reads it back to check what a CU-masked stream owns.
__global__ __launch_bounds__(256) void cu_map_kernel(unsigned* out, long long spin_ticks) {
    extern __shared__ char lds_hold[];
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        lds_hold[0] = (char)xcc;
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
}

}  // namespace

hipError_t launch_cu_map(unsigned* out, int nblocks, int spin_us, hipStream_t s) {
    constexpr int LDS = 144 * 1024;
    static std::once_flag once;
    static hipError_t attr = hipSuccess;
    std::call_once(once, [] { attr = hipFuncSetAttribute((const void*)cu_map_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); });
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(cu_map_kernel, dim3(nblocks), dim3(256), LDS, s, out, (long long)spin_us * 100);      // wall_clock64: 100 MHz
    return hipGetLastError();
}

// -> tflops: MFMA-issue TFLOP/s of the stream over >= timed_s seconds after >= warm_s seconds of the same load; ghz: the clock that rate
// implies (2 waves per SIMD, 16 cycles per MFMA)
hipError_t mfma_stream_measure(int fp16, int data, double warm_s, double timed_s, double* tflops, double* ghz, hipStream_t s) {
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    float* out = nullptr;
    e = hipMalloc((void**)&out, (size_t)cus * 512 * sizeof(float));
    if (e != hipSuccess) return e;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const int iters = 20000;                              // ~ 45 ms per launch: the host keeps two launches queued, the GPU never idles
    auto launch = [&]() {
        if (fp16) hipLaunchKernelGGL((mfma_stream_kernel<f16>), dim3(cus), dim3(512), 0, s, out, iters, data);
        else hipLaunchKernelGGL((mfma_stream_kernel<bf16>), dim3(cus), dim3(512), 0, s, out, iters, data);
    };
    auto spin = [&](double seconds, int& n) -> hipError_t {
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
        n = 0;
        hipEvent_t prev = nullptr;
        do {
            launch();
            ++n;
            hipEvent_t ev;
            hipError_t er = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (er != hipSuccess) return er;
            (void)hipEventRecord(ev, s);
            if (prev) { (void)hipEventSynchronize(prev); (void)hipEventDestroy(prev); }    // at most two launches in flight
            prev = ev;
        } while (std::chrono::steady_clock::now() < t_end);
        if (prev) { (void)hipEventSynchronize(prev); (void)hipEventDestroy(prev); }
        return hipGetLastError();
    };
    int n = 0;
    if ((e = hipEventCreate(&e0)) == hipSuccess && (e = hipEventCreate(&e1)) == hipSuccess && (e = spin(warm_s, n)) == hipSuccess) {
        (void)hipEventRecord(e0, s);
        e = spin(timed_s, n);
        (void)hipEventRecord(e1, s);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && ms > 0.f) {
            const double mfma_per_wave = 24.0 * iters * n;
            const double flops = mfma_per_wave * 8.0 * cus * 2.0 * 16 * 16 * 32;
            if (tflops) *tflops = flops / (ms * 1e-3) / 1e12;
            if (ghz) *ghz = 2.0 * mfma_per_wave * 16.0 / (ms * 1e-3) / 1e9;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(out);
    return e;
}

}  // namespace vtq
