// Validation-loop reductions on the device: the per-image mean over test-time repeats
// (train.py:398-400, average_over_repeats) and the fit-free part of utils/misc/correlations.py:21-33 --
// normalize_array (utils/image_processing/image_tools.py:17-21), Spearman (Pearson of average-tie ranks), the integer
// pair counts of Kendall's tau-b, Pearson and RMSE -- so a validation pass keeps its scores on the GPU and brings back
// eight numbers instead of synchronising on q.cpu() after every batch (train.py:617-618).
// fp64 throughout, like the reference's numpy arrays; the pair counts are exact integers.
// Bound: the O(N^2) pair kernel is VALU/LDS work on N <= 1e5 scores (1e10 compares ~ 10 ms); nothing here is on the
// forward's critical path.
#include "dev_common.h"
#include "kernels.h"

namespace vtq {

namespace {

constexpr int kT = 256;

__device__ __forceinline__ double block_sum(double v, double* sh) {      // all threads get the sum; fixed tree order
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if (t < o) sh[t] += sh[t + o];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_minmax(double v, double* sh, bool want_max) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if (t < o) sh[t] = want_max ? fmax(sh[t], sh[t + o]) : fmin(sh[t], sh[t + o]);
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// numpy reduces axis 0 of the (R, N) view row by row: ((x0 + x1) + x2) + ... then one true-divide
__global__ void repeat_mean_kernel(const float* __restrict__ q, double* __restrict__ out, int R, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double s = (double)q[i];
    for (int r = 1; r < R; ++r) s += (double)q[(int64_t)r * N + i];
    out[i] = s / (double)R;
}

// one block: b = a - min(a); if |max(b)| > 1e-6: b /= max(b)   (normalize == 0: plain copy)
__global__ void normalize_kernel(const double* __restrict__ a, double* __restrict__ b, int N, int normalize) {
    __shared__ double sh[1024];
    const int t = threadIdx.x;
    if (!normalize) {
        for (int i = t; i < N; i += blockDim.x) b[i] = a[i];
        return;
    }
    double mn = INFINITY;
    for (int i = t; i < N; i += blockDim.x) mn = fmin(mn, a[i]);
    mn = block_minmax(mn, sh, false);
    double mx = -INFINITY;
    for (int i = t; i < N; i += blockDim.x) { const double v = a[i] - mn; b[i] = v; mx = fmax(mx, v); }
    mx = block_minmax(mx, sh, true);
    if (fabs(mx) > 1e-6)
        for (int i = t; i < N; i += blockDim.x) b[i] = b[i] / mx;
}

// thread i against every j: average-tie ranks of both arrays and the Kendall pair sums over ordered pairs
//   counts[0] += sum_j sign(a_i - a_j) * sign(b_i - b_j)   (= 2 (concordant - discordant))
//   counts[1] += #{j != i : a_j == a_i}                    (= 2 xtie)       counts[2]: same for b (= 2 ytie)
__global__ __launch_bounds__(kT) void pair_kernel(const double* __restrict__ a, const double* __restrict__ b, int N,
                                                  double* __restrict__ ra, double* __restrict__ rb, long long* __restrict__ counts) {
    __shared__ double sa[kT], sb[kT];
    __shared__ long long red[3][kT];
    const int t = threadIdx.x, i = blockIdx.x * kT + t;
    const bool live = i < N;
    const double ai = live ? a[i] : 0.0, bi = live ? b[i] : 0.0;
    long long less_a = 0, eq_a = 0, less_b = 0, eq_b = 0, sgn = 0;
    for (int j0 = 0; j0 < N; j0 += kT) {
        const int j = j0 + t;
        sa[t] = j < N ? a[j] : 0.0;
        sb[t] = j < N ? b[j] : 0.0;
        __syncthreads();
        const int lim = min(kT, N - j0);
        for (int k = 0; k < lim; ++k) {
            const double aj = sa[k], bj = sb[k];
            const int da = (ai > aj) - (ai < aj), db = (bi > bj) - (bi < bj);
            less_a += da > 0; eq_a += da == 0;
            less_b += db > 0; eq_b += db == 0;
            sgn += da * db;
        }
        __syncthreads();
    }
    if (live) {                                              // eq counts include j == i
        ra[i] = (double)less_a + 0.5 * (double)(eq_a + 1);
        rb[i] = (double)less_b + 0.5 * (double)(eq_b + 1);
    }
    red[0][t] = live ? sgn : 0;
    red[1][t] = live ? eq_a - 1 : 0;
    red[2][t] = live ? eq_b - 1 : 0;
    __syncthreads();
    for (int o = kT >> 1; o > 0; o >>= 1) {
        if (t < o) { red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o]; red[2][t] += red[2][t + o]; }
        __syncthreads();
    }
    if (t < 3) atomicAdd((unsigned long long*)&counts[t], (unsigned long long)red[t][0]);
}

// one block: Pearson r of (x, y) the scipy.stats.pearsonr way (centre, normalise, dot, clamp) and sqrt(mean((x - y)^2))
__global__ void pearson_rmse_kernel(const double* __restrict__ x, const double* __restrict__ y, int N, double* __restrict__ r_out,
                                    double* __restrict__ rmse_out) {
    __shared__ double sh[1024];
    const int t = threadIdx.x;
    double sx = 0, sy = 0;
    for (int i = t; i < N; i += blockDim.x) { sx += x[i]; sy += y[i]; }
    const double mx = block_sum(sx, sh) / N, my = block_sum(sy, sh) / N;
    double xx = 0, yy = 0, xy = 0, dd = 0;
    for (int i = t; i < N; i += blockDim.x) {
        const double u = x[i] - mx, v = y[i] - my, d = x[i] - y[i];
        xx += u * u; yy += v * v; xy += u * v; dd += d * d;
    }
    xx = block_sum(xx, sh); yy = block_sum(yy, sh); xy = block_sum(xy, sh); dd = block_sum(dd, sh);
    if (t == 0) {
        double r = xy / (sqrt(xx) * sqrt(yy));
        r = fmax(fmin(r, 1.0), -1.0);
        if (r_out) *r_out = r;
        if (rmse_out) *rmse_out = sqrt(dd / N);
    }
}

}  // namespace

hipError_t launch_repeat_mean(const float* q, double* out, int R, int N, hipStream_t s) {
    hipLaunchKernelGGL(repeat_mean_kernel, dim3((N + 255) / 256), dim3(256), 0, s, q, out, R, N);
    return hipGetLastError();
}

hipError_t launch_rank_metrics(const double* a, const double* b, int N, int normalize, double* aa, double* bb, double* ra, double* rb,
                               long long* counts, double* out, hipStream_t s) {
    hipError_t e = hipMemsetAsync(counts, 0, 3 * sizeof(long long), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(1024), 0, s, a, aa, N, normalize);
    hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(1024), 0, s, b, bb, N, normalize);
    hipLaunchKernelGGL(pair_kernel, dim3((N + kT - 1) / kT), dim3(kT), 0, s, (const double*)aa, (const double*)bb, N, ra, rb, counts);
    hipLaunchKernelGGL(pearson_rmse_kernel, dim3(1), dim3(1024), 0, s, (const double*)ra, (const double*)rb, N, out + 0, (double*)nullptr);
    hipLaunchKernelGGL(pearson_rmse_kernel, dim3(1), dim3(1024), 0, s, (const double*)aa, (const double*)bb, N, out + 1, out + 2);
    return hipGetLastError();
}

}  // namespace vtq
