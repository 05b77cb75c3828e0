/*
 * io.hip -- the data formats either side of the hot path:
 *   wave sample codecs  wave/wave.go:275-735   LPCM 8/16/24/32 and IEEE 32/64 <-> float64, bit exact
 *   resample.Time       resample/resample.go:72-103   Lanczos-3 rate conversion of whole files
 * Both are embarrassingly parallel and HBM bound (1..8 B in + 8 B out per sample for the codecs;
 * ~6 x 8 B gathered (cache hits) + 8 B out for the resampler, which is sin()-bound in FP64).
 */
#include "gdg_internal.h"
#include <math.h>

#define MAX_INT24 0x007fffff
#define MIN_INT24 (-(MAX_INT24 + 1))
#define SIGN_BIT_INT24 0x00800000

__device__ __forceinline__ double clamp1(double s) { return s < -1.0 ? -1.0 : (s > 1.0 ? 1.0 : s); }

template <int FMT>
__global__ void __launch_bounds__(256)
wave_decode_kernel(const unsigned char *__restrict__ data, size_t n, double *__restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double r;
        if (FMT == GDG_FMT_LPCM8) {                              /* wave.go:316-342 */
            short temp = (short)((short)data[i] + (-128));
            r = (1.0 / 127.0) * (double)temp;
            r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
        } else if (FMT == GDG_FMT_LPCM16) {                      /* wave.go:400-427 */
            short s = (short)(unsigned short)(data[2 * i] | (data[2 * i + 1] << 8));
            r = (2.0 / 65535.0) * (double)s;
        } else if (FMT == GDG_FMT_LPCM24) {                      /* wave.go:475-514 */
            unsigned w = (unsigned)data[3 * i] | ((unsigned)data[3 * i + 1] << 8) | ((unsigned)data[3 * i + 2] << 16);
            int v = (int)w;
            if (w & SIGN_BIT_INT24) v = MIN_INT24 + (v & MAX_INT24);
            r = (2.0 / 16777215.0) * (double)v;
        } else if (FMT == GDG_FMT_LPCM32) {                      /* wave.go:567-594 */
            unsigned w = reinterpret_cast<const unsigned *>(data)[i];
            r = (2.0 / 4294967295.0) * (double)(int)w;
        } else if (FMT == GDG_FMT_IEEE32) {                      /* wave.go:640-669 */
            r = (double)reinterpret_cast<const float *>(data)[i];
        } else {                                                 /* wave.go:695-714 */
            r = reinterpret_cast<const double *>(data)[i];
        }
        out[i] = r;
    }
}

template <int FMT>
__global__ void __launch_bounds__(256)
wave_encode_kernel(const double *__restrict__ in, size_t n, unsigned char *__restrict__ data) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double sample = in[i];
        if (FMT == GDG_FMT_LPCM8) {                              /* wave.go:275-311 */
            sample = clamp1(sample);
            short temp = (short)(127.0 * sample);
            int res = temp + 128;
            data[i] = (unsigned char)(res < 0 ? 0 : (res > 255 ? 255 : res));
        } else if (FMT == GDG_FMT_LPCM16) {                      /* wave.go:347-395 */
            sample = clamp1(sample);
            int tmp = (int)((0.5 * 65535.0) * sample);
            tmp = tmp > 32767 ? 32767 : (tmp < -32768 ? -32768 : tmp);
            reinterpret_cast<short *>(data)[i] = (short)tmp;
        } else if (FMT == GDG_FMT_LPCM24) {                      /* wave.go:433-470 */
            sample = clamp1(sample);
            int tmp = (int)((0.5 * 16777215.0) * sample);
            tmp = tmp > MAX_INT24 ? MAX_INT24 : (tmp < MIN_INT24 ? MIN_INT24 : tmp);
            unsigned u = (unsigned)tmp;
            data[3 * i] = (unsigned char)(u & 0xff);
            data[3 * i + 1] = (unsigned char)((u >> 8) & 0xff);
            data[3 * i + 2] = (unsigned char)((u >> 16) & 0xff);
        } else if (FMT == GDG_FMT_LPCM32) {                      /* wave.go:519-562 */
            sample = clamp1(sample);
            long long tmp = (long long)((0.5 * 4294967295.0) * sample);
            tmp = tmp > 2147483647LL ? 2147483647LL : (tmp < -2147483648LL ? -2147483648LL : tmp);
            reinterpret_cast<int *>(data)[i] = (int)tmp;
        } else if (FMT == GDG_FMT_IEEE32) {                      /* wave.go:599-635 */
            reinterpret_cast<float *>(data)[i] = (float)clamp1(sample);
        } else {                                                 /* wave.go:674-690: no clipping */
            reinterpret_cast<double *>(data)[i] = sample;
        }
    }
}

/* resample/resample.go:10-31 */
__device__ __forceinline__ double lanczos_kernel(double x, double a) {
    if (x == 0) return 1.0;
    if ((-a < x) && (x < a)) {
        double pi_x = M_PI * x;
        double pi_xa = pi_x / a;
        double pi_x_squared = pi_x * pi_x;
        double prod = sin(pi_x) * sin(pi_xa);
        double arg = a * prod;
        return arg / pi_x_squared;
    }
    return 0.0;
}

/* resample.Time: out[i] = sum_{j = floor(x)-2}^{floor(x)+3} s[j] L3(x - j), x = i * (src / dst) (resample.go:36-103) */
__global__ void __launch_bounds__(256)
resample_time_kernel(const double *__restrict__ s, int n, double dx, double *__restrict__ out, int n_out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_out; i += gridDim.x * 256) {
        double x = (double)i * dx;
        int idx = (int)floor(x);
        double sum = 0.0;
#pragma unroll
        for (int j = idx - 2; j < idx + 4; j++) {
            if (j >= 0 && j < n) {
                double diff = x - (double)j;
                sum += s[j] * lanczos_kernel(diff, 3.0);
            }
        }
        out[i] = sum;
    }
}

static int grid_for(size_t n) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    return (int)(blocks ? blocks : 1);
}

hipError_t gdg_launch_wave_decode(int fmt, const void *d_bytes, size_t n, double *d_out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const unsigned char *p = static_cast<const unsigned char *>(d_bytes);
    int g = grid_for(n);
    switch (fmt) {
    case GDG_FMT_LPCM8: wave_decode_kernel<GDG_FMT_LPCM8><<<g, 256, 0, s>>>(p, n, d_out); break;
    case GDG_FMT_LPCM16: wave_decode_kernel<GDG_FMT_LPCM16><<<g, 256, 0, s>>>(p, n, d_out); break;
    case GDG_FMT_LPCM24: wave_decode_kernel<GDG_FMT_LPCM24><<<g, 256, 0, s>>>(p, n, d_out); break;
    case GDG_FMT_LPCM32: wave_decode_kernel<GDG_FMT_LPCM32><<<g, 256, 0, s>>>(p, n, d_out); break;
    case GDG_FMT_IEEE32: wave_decode_kernel<GDG_FMT_IEEE32><<<g, 256, 0, s>>>(p, n, d_out); break;
    case GDG_FMT_IEEE64: wave_decode_kernel<GDG_FMT_IEEE64><<<g, 256, 0, s>>>(p, n, d_out); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t gdg_launch_wave_encode(int fmt, const double *d_in, size_t n, void *d_bytes, hipStream_t s) {
    if (n == 0) return hipSuccess;
    unsigned char *p = static_cast<unsigned char *>(d_bytes);
    int g = grid_for(n);
    switch (fmt) {
    case GDG_FMT_LPCM8: wave_encode_kernel<GDG_FMT_LPCM8><<<g, 256, 0, s>>>(d_in, n, p); break;
    case GDG_FMT_LPCM16: wave_encode_kernel<GDG_FMT_LPCM16><<<g, 256, 0, s>>>(d_in, n, p); break;
    case GDG_FMT_LPCM24: wave_encode_kernel<GDG_FMT_LPCM24><<<g, 256, 0, s>>>(d_in, n, p); break;
    case GDG_FMT_LPCM32: wave_encode_kernel<GDG_FMT_LPCM32><<<g, 256, 0, s>>>(d_in, n, p); break;
    case GDG_FMT_IEEE32: wave_encode_kernel<GDG_FMT_IEEE32><<<g, 256, 0, s>>>(d_in, n, p); break;
    case GDG_FMT_IEEE64: wave_encode_kernel<GDG_FMT_IEEE64><<<g, 256, 0, s>>>(d_in, n, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t gdg_launch_resample_time(const double *d_in, int n, double dx, double *d_out, int n_out, hipStream_t s) {
    if (n_out <= 0) return hipSuccess;
    resample_time_kernel<<<grid_for((size_t)n_out), 256, 0, s>>>(d_in, n, dx, d_out, n_out);
    return hipGetLastError();
}
