/*
 * compile.hip -- power-amp filter compilation on the device:
 *   effects/poweramp.go:25-127   per slot: Reduce(order) -> Normalize -> Multiply(level), then Add in slot order
 *   filter/filter.go:520-604     Reduce = real FFT -> Lanczos-3 resampling of the positive-frequency bins -> inverse real FFT
 *   filter/filter.go:127-138, :270-336   Normalize / Multiply / Add
 *   resample/resample.go:109-142 resample.Frequency
 * Set-up time work (a knob move recompiles 8 slots x up to 2^20 taps): every step is a plain elementwise / reduction
 * kernel over HBM; the FFT is a radix-2 Stockham transform, one launch per stage, any power of two (no LDS limit).
 */
#include "../../include/gdg.h"
#include "gdg_internal.h"
#include <math.h>

typedef double2 cplx;

__global__ void __launch_bounds__(256)
real_to_complex_kernel(const double *__restrict__ src, int n, cplx *__restrict__ dst, int n_fft) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_fft; i += gridDim.x * 256)
        dst[i] = make_double2(i < n ? src[i] : 0.0, 0.0);
}

/* one radix-2 Stockham stage: sub-transforms of size ns -> 2 ns; sign = -1 forward, +1 inverse */
__global__ void __launch_bounds__(256)
fft_stage_kernel(const cplx *__restrict__ in, cplx *__restrict__ out, int half, int ns, double sign) {
    for (int j = blockIdx.x * 256 + threadIdx.x; j < half; j += gridDim.x * 256) {
        int k = j & (ns - 1);
        double s, c;
        sincospi((double)k / (double)ns, &s, &c);              /* angle pi k / ns = 2 pi k / (2 ns) */
        s *= sign;
        cplx a = in[j], b = in[j + half];
        cplx wb = make_double2(b.x * c - b.y * s, b.x * s + b.y * c);
        int j0 = ((j - k) << 1) + k;
        out[j0] = make_double2(a.x + wb.x, a.y + wb.y);
        out[j0 + ns] = make_double2(a.x - wb.x, a.y - wb.y);
    }
}

/* resample/resample.go:10-66 on one array component */
__device__ __forceinline__ double lanczos3(double x) {
    if (x == 0) return 1.0;
    if ((-3.0 < x) && (x < 3.0)) {
        double pi_x = M_PI * x;
        double pi_xa = pi_x / 3.0;
        double pi_x_squared = pi_x * pi_x;
        double prod = sin(pi_x) * sin(pi_xa);
        double arg = 3.0 * prod;
        return arg / pi_x_squared;
    }
    return 0.0;
}

/* resample.Frequency (resample.go:109-142): real and imaginary parts interpolated separately */
__global__ void __launch_bounds__(256)
resample_frequency_kernel(const cplx *__restrict__ bins, int n_src, cplx *__restrict__ out, int n_tgt) {
    const double dx = (double)n_src / (double)n_tgt;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_tgt; i += gridDim.x * 256) {
        double x = (double)i * dx;
        int idx = (int)floor(x);
        double re = 0.0, im = 0.0;
#pragma unroll
        for (int j = idx - 2; j < idx + 4; j++) {
            if (j >= 0 && j < n_src) {
                double w = lanczos3(x - (double)j);
                re += bins[j].x * w;
                im += bins[j].y * w;
            }
        }
        out[i] = make_double2(re, im);
    }
}

/* filter.go:563-583 + what RealInverseFourier keeps of its input (fft.go: the real parts of bins 0 and n/2) */
__global__ void __launch_bounds__(256)
hermitian_kernel(const cplx *__restrict__ pos, int n, cplx *__restrict__ full) {
    const int half = n >> 1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        cplx v;
        if (i == 0 || i == half) v = make_double2(pos[i].x, 0.0);
        else if (i < half) v = pos[i];
        else v = make_double2(pos[n - i].x, -pos[n - i].y);
        full[i] = v;
    }
}

__global__ void __launch_bounds__(256)
take_real_kernel(const cplx *__restrict__ src, double scale, int n, double *__restrict__ dst) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i].x * scale;
}

/* sum of squares: fixed-shape two-level reduction (deterministic; differs from the reference's left-to-right sum by rounding only) */
__global__ void __launch_bounds__(256)
sumsq_partial_kernel(const double *__restrict__ src, int n, double *__restrict__ partial) {
    __shared__ double s[256];
    double acc = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) acc += src[i] * src[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}
__global__ void __launch_bounds__(256)
sumsq_final_kernel(const double *__restrict__ partial, int n, double *__restrict__ out) {
    __shared__ double s[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = s[0];
}

/* Normalize (fac = compensation / sqrt(sum)), Multiply(level), Add into the composite: composite += level * (fac * x) */
__global__ void __launch_bounds__(256)
normalize_scale_add_kernel(const double *__restrict__ src, int n, const double *__restrict__ sumsq, double compensation, double level,
                           double *__restrict__ composite) {
    const double gain = sqrt(sumsq[0]);
    const double fac = compensation / gain;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        double normalized = fac * src[i];
        double scaled = level * normalized;
        composite[i] = composite[i] + scaled;
    }
}

static int grid_of(long long n) {
    long long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    return (int)(b < 1 ? 1 : b);
}

/* complex FFT of n = 2^m points; data in `a`, scratch `b`; returns the buffer that holds the result */
static cplx *fft_pow2(cplx *a, cplx *b, int n, double sign, hipStream_t s) {
    int half = n >> 1;
    for (int ns = 1; ns < n; ns <<= 1) {
        fft_stage_kernel<<<grid_of(half), 256, 0, s>>>(a, b, half, ns, sign);
        cplx *t = a; a = b; b = t;
    }
    return a;
}

static unsigned long long next_pow2(unsigned long long v) {
    unsigned long long p = 1;
    while (p < v) p <<= 1;
    return p;
}

/*
 * filter.Reduce (filter.go:520-604): d_taps[n] -> d_out[order]; the caller guarantees n > order > 0.
 * work_a, work_b: nextpow2(n) complex points each; work_pos: nextpow2(order) / 2 + 1 complex points.
 */
hipError_t gdg_launch_filter_reduce(const double *d_taps, int n, unsigned order, cplx *work_a, cplx *work_b, cplx *work_pos, double *d_out,
                                    hipStream_t s) {
    const int n_src = (int)next_pow2((unsigned long long)n), n_tgt = (int)next_pow2(order);
    const int pos_src = (n_src >> 1) + 1, pos_tgt = (n_tgt >> 1) + 1;
    real_to_complex_kernel<<<grid_of(n_src), 256, 0, s>>>(d_taps, n, work_a, n_src);
    cplx *spec = fft_pow2(work_a, work_b, n_src, -1.0, s);
    cplx *other = (spec == work_a) ? work_b : work_a;
    resample_frequency_kernel<<<grid_of(pos_tgt), 256, 0, s>>>(spec, pos_src, work_pos, pos_tgt);
    if (n_tgt == 1) {           /* one-point inverse: x[0] = Re X[0] */
        take_real_kernel<<<1, 256, 0, s>>>(work_pos, 1.0, 1, d_out);
        return hipGetLastError();
    }
    hermitian_kernel<<<grid_of(n_tgt), 256, 0, s>>>(work_pos, n_tgt, other);         /* n_tgt <= n_src: fits */
    cplx *res = fft_pow2(other, spec, n_tgt, +1.0, s);
    take_real_kernel<<<grid_of(order), 256, 0, s>>>(res, 1.0 / (double)n_tgt, (int)order, d_out);
    return hipGetLastError();
}

void gdg_filter_reduce_sizes(int n, unsigned order, size_t *work_points, size_t *pos_points) {
    *work_points = (size_t)next_pow2((unsigned long long)(n > 0 ? n : 1));
    *pos_points = (size_t)(next_pow2(order > 0 ? order : 1) / 2 + 1);
}

hipError_t gdg_launch_normalize_scale_add(const double *d_src, int n, double compensation, double level, double *d_partial, double *d_composite,
                                          hipStream_t s) {
    int g = grid_of(n);
    if (g > 256) g = 256;
    sumsq_partial_kernel<<<g, 256, 0, s>>>(d_src, n, d_partial);
    sumsq_final_kernel<<<1, 256, 0, s>>>(d_partial, g, d_partial + 256);
    normalize_scale_add_kernel<<<grid_of(n), 256, 0, s>>>(d_src, n, d_partial + 256, compensation, level, d_composite);
    return hipGetLastError();
}
