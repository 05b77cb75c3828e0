/*
 * io.hip -- the data formats either side of the hot path:
 *   wave sample codecs  wave/wave.go:275-735   LPCM 8/16/24/32 and IEEE 32/64 <-> float64, bit exact
 *   resample.Time       resample/resample.go:72-103   Lanczos-3 rate conversion of whole files
 * Both are embarrassingly parallel and HBM bound (1..8 B in + 8 B out per sample for the codecs;
 * ~6 x 8 B gathered (cache hits) + 8 B out for the resampler, which is sin()-bound in FP64).
 */
#include "../../include/gdg.h"
#include "gdg_internal.h"
#include <math.h>
#include <stdlib.h>

#define MAX_INT24 0x007fffff
#define MIN_INT24 (-(MAX_INT24 + 1))
#define SIGN_BIT_INT24 0x00800000

__device__ __forceinline__ double clamp1(double s) { return s < -1.0 ? -1.0 : (s > 1.0 ? 1.0 : s); }

template <int FMT>
__global__ void __launch_bounds__(256)
wave_decode_kernel(const unsigned char *__restrict__ data, size_t n, double *__restrict__ out, unsigned channels, size_t per) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double r;
        if (FMT == GDG_FMT_LPCM8) {                              /* wave.go:316-342 */
            short temp = (short)((short)data[i] + (-128));
            r = (1.0 / 127.0) * (double)temp;
            r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
        } else if (FMT == GDG_FMT_LPCM16) {                      /* wave.go:400-428 */
            short s = (short)(unsigned short)(data[2 * i] | (data[2 * i + 1] << 8));
            r = (2.0 / 65535.0) * (double)s;
        } else if (FMT == GDG_FMT_LPCM24) {                      /* wave.go:483-531 */
            unsigned w = (unsigned)data[3 * i] | ((unsigned)data[3 * i + 1] << 8) | ((unsigned)data[3 * i + 2] << 16);
            int v = (int)w;
            if (w & SIGN_BIT_INT24) v = MIN_INT24 + (v & MAX_INT24);
            r = (2.0 / 16777215.0) * (double)v;
        } else if (FMT == GDG_FMT_LPCM32) {                      /* wave.go:589-617 */
            unsigned w = reinterpret_cast<const unsigned *>(data)[i];
            r = (2.0 / 4294967295.0) * (double)(int)w;
        } else if (FMT == GDG_FMT_IEEE32) {                      /* wave.go:662-689 */
            r = (double)reinterpret_cast<const float *>(data)[i];
        } else {                                                 /* wave.go:714-732 */
            r = reinterpret_cast<const double *>(data)[i];
        }
        /* samplesToChannels (wave.go:237-270): interleaved sample i belongs to channel i % C, position i / C */
        out[channels == 1 ? i : (size_t)(i % channels) * per + i / channels] = r;
    }
}

template <int FMT>
__global__ void __launch_bounds__(256)
wave_encode_kernel(const double *__restrict__ in, size_t n, unsigned char *__restrict__ data, unsigned channels, size_t per) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        /* channelsToSamples (wave.go:173-232) */
        double sample = in[channels == 1 ? i : (size_t)(i % channels) * per + i / channels];
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
        } else if (FMT == GDG_FMT_LPCM24) {                      /* wave.go:433-478 */
            sample = clamp1(sample);
            int tmp = (int)((0.5 * 16777215.0) * sample);
            tmp = tmp > MAX_INT24 ? MAX_INT24 : (tmp < MIN_INT24 ? MIN_INT24 : tmp);
            unsigned u = (unsigned)tmp;
            data[3 * i] = (unsigned char)(u & 0xff);
            data[3 * i + 1] = (unsigned char)((u >> 8) & 0xff);
            data[3 * i + 2] = (unsigned char)((u >> 16) & 0xff);
        } else if (FMT == GDG_FMT_LPCM32) {                      /* wave.go:536-584 */
            sample = clamp1(sample);
            long long tmp = (long long)((0.5 * 4294967295.0) * sample);
            tmp = tmp > 2147483647LL ? 2147483647LL : (tmp < -2147483648LL ? -2147483648LL : tmp);
            reinterpret_cast<int *>(data)[i] = (int)tmp;
        } else if (FMT == GDG_FMT_IEEE32) {                      /* wave.go:622-657 */
            reinterpret_cast<float *>(data)[i] = (float)clamp1(sample);
        } else {                                                 /* wave.go:694-709: no clipping */
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

/* ------------------------------------------------------------------------------------------------
 * level/level.go:147-210 channel meter, one workgroup per port, at most GDG_METER_SEG samples per launch.
 *   current:  c <- max(c * d, |x|)  ==  max(c0 d^n, max_i |x_i| d^(n-1-i))            (weighted max reduction)
 *   peak/hold: the reference's per-sample state machine (decay once the counter passed `hold`, a sample
 *   >= peak records and restarts the hold) has, within a segment of n <= hold samples, exactly one
 *   interesting event: the FIRST record r.  Before it the peak is p0 (hold phase) or p0 d^k (decay phase,
 *   closed form); after it the peak is the running maximum of |x_r..|, the counter the distance to the LAST
 *   sample attaining that maximum.  The host guarantees n <= hold (gdg_meter_process splits).
 * ---------------------------------------------------------------------------------------------- */
#define METER_T 1024
#define METER_CHK (GDG_METER_SEG / METER_T)

/* Two values reduced over the workgroup in one go: the maximum of `v` and the maximum of the pair (key, idx) in the order "larger key, then
 * larger idx".  Wave level by shuffles, the 16 wave results through LDS, and EVERY wave finishes the reduction from those 16 cells with four
 * more shuffle steps (no serial walk over the partials: that walk alone was ~1 000 cycles, and the kernel had four such reductions). */
struct MeterRed { double v; double key; int idx; };
__device__ __forceinline__ MeterRed meter_reduce(MeterRed x, double *scr_v, double *scr_k, int *scr_i) {
    const int tid = threadIdx.x, lane = tid & 63;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_down(x.v, o), ok = __shfl_down(x.key, o);
        const int oi = __shfl_down(x.idx, o);
        x.v = fmax(x.v, ov);
        if (ok > x.key || (ok == x.key && oi > x.idx)) { x.key = ok; x.idx = oi; }
    }
    __syncthreads();                                    /* the cells of the reduction before are no longer read */
    if (lane == 0) { scr_v[tid >> 6] = x.v; scr_k[tid >> 6] = x.key; scr_i[tid >> 6] = x.idx; }
    __syncthreads();
    MeterRed r = { scr_v[lane & 15], scr_k[lane & 15], scr_i[lane & 15] };
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const double ov = __shfl_xor(r.v, o), ok = __shfl_xor(r.key, o);
        const int oi = __shfl_xor(r.idx, o);
        r.v = fmax(r.v, ov);
        if (ok > r.key || (ok == r.key && oi > r.idx)) { r.key = ok; r.idx = oi; }
    }
    return r;
}
static_assert(METER_T / 64 == 16, "meter_reduce finishes from 16 wave results");

__global__ void __launch_bounds__(METER_T)
meter_kernel(const double *__restrict__ rows, size_t stride, int n, gdg_meter_rec *__restrict__ st,
             double decay, unsigned long long hold) {
    __shared__ double scr_v[METER_T / 64], scr_k[METER_T / 64];
    __shared__ int scr_i[METER_T / 64];
    /* a thread works on METER_CHK CONSECUTIVE samples (the follower is sequential inside a chunk), but the port is read the way memory
     * likes it -- lane after lane, all loads of the thread in flight -- and turned through LDS (one pad cell per 32, as in the segment
     * kernel).  Reading x[base + k] directly gave every lane its own 64-byte piece of a 4 KiB span per instruction: eight times the
     * port's bytes between L2 and L1. */
    __shared__ double turn[GDG_METER_SEG + GDG_METER_SEG / 32];
    /* decay^m for the exponents the closed forms need (0 <= m <= n <= 8192): three small tables made by 73 threads with pow(), every
     * other power is a product of three entries (pow() in every thread, three calls of ~300 FP64 instructions each, kept all SIMDs busy
     * for ~11 us per round of 256 workgroups). */
    __shared__ double pw_lo[32], pw_hi[33], pw_one[8];      /* (d^8)^j, (d^256)^i, d^j */
    const int tid = threadIdx.x;
    gdg_meter_rec *m = st + blockIdx.x;
    const double *x = rows + (size_t)blockIdx.x * stride;
    /* a disabled port's row is never touched: gdg_meter_process_device takes caller-supplied rows, and a caller may leave the rows of
     * disabled ports unbacked (`enabled` is uniform over the workgroup: one scalar load in front of the row loads) */
    if (!m->enabled) return;
    double t[METER_CHK];
#pragma unroll
    for (int k = 0; k < METER_CHK; k++) { const int i = tid + METER_T * k; t[k] = (i < n) ? x[i] : 0.0; }     /* in flight while the tables are made */
    const double c0 = m->current, p0 = m->peak;
    const unsigned long long cnt0 = m->counter;
    const int base = tid * METER_CHK;
    if (tid < 32) pw_lo[tid] = pow(decay, (double)(8 * tid));
    else if (tid < 65) pw_hi[tid - 32] = pow(decay, (double)(256 * (tid - 32)));
    else if (tid < 73) pw_one[tid - 65] = pow(decay, (double)(tid - 65));
#pragma unroll
    for (int k = 0; k < METER_CHK; k++) { const int i = tid + METER_T * k; turn[i + (i >> 5)] = t[k]; }
    __syncthreads();
    auto dpow = [&](long long e) -> double {                /* 0 <= e <= 8192 + 7 */
        const int a8 = (int)(e >> 3);
        return (pw_hi[a8 >> 5] * pw_lo[a8 & 31]) * pw_one[(int)(e & 7)];
    };
    double a[METER_CHK];
#pragma unroll
    for (int k = 0; k < METER_CHK; k++) { const int i = base + k; a[k] = (i < n) ? fabs(turn[i + (i >> 5)]) : -1.0; }

    /* current value: zero-state follower over the chunk, then its decay to the end of the segment */
    double b = 0.0;
    int len = 0;
#pragma unroll
    for (int k = 0; k < METER_CHK; k++)
        if (base + k < n) { b *= decay; if (a[k] > b) b = a[k]; len++; }
    const double contrib = (len > 0) ? b * dpow(n - base - len) : 0.0;

    /* first record: smallest i with |x_i| >= p_i;  p_i = p0 for i < e, p0 d^(i-e+1) from e on */
    const long long e = (cnt0 > hold) ? 0 : (long long)(hold - cnt0) + 1;
    double p = p0;
    if ((long long)base >= e) p = p0 * dpow((long long)base - e);      /* value before sample `base` decays */
    int first = n;
#pragma unroll
    for (int k = 0; k < METER_CHK; k++) {
        int i = base + k;
        if (i < n) {
            if ((long long)i >= e) p *= decay;
            if (first == n && a[k] >= p) first = i;
        }
    }
    /* one reduction for both: the largest contribution, and the smallest `first` (as the largest n - first) */
    const MeterRed r1 = meter_reduce(MeterRed{ contrib, 0.0, n - first }, scr_v, scr_k, scr_i);
    const double cur = fmax(r1.v, c0 * dpow(n));
    const int r = n - r1.idx;

    double peak;
    unsigned long long counter;
    if (r >= n) {       /* no record in this segment */
        long long dec = (long long)n - e;
        peak = (dec > 0) ? p0 * dpow(dec) : p0;
        counter = (cnt0 > hold) ? cnt0 : ((cnt0 + (unsigned long long)n < hold + 1) ? cnt0 + (unsigned long long)n : hold + 1);
    } else {
        /* from the record on the peak is the running maximum, the counter the distance to the LAST sample attaining it: the pair
         * (value, index) reduced in the order "larger value, then larger index" */
        double mx = -1.0;
        int last = -1;
#pragma unroll
        for (int k = 0; k < METER_CHK; k++)
            if (base + k >= r && base + k < n && a[k] >= mx) { mx = a[k]; last = base + k; }
        const MeterRed r2 = meter_reduce(MeterRed{ 0.0, mx, last }, scr_v, scr_k, scr_i);
        peak = r2.key;
        counter = (unsigned long long)(n - 1 - r2.idx);
    }
    if (tid == 0) { m->current = cur; m->peak = peak; m->counter = counter; }
}

hipError_t gdg_launch_meter(const double *d_rows, size_t stride, int n_ports, int n, gdg_meter_rec *d_state,
                            double decay, unsigned long long hold, hipStream_t s) {
    if (n_ports <= 0 || n <= 0) return hipSuccess;
    if (n > GDG_METER_SEG) return hipErrorInvalidValue;
    meter_kernel<<<n_ports, METER_T, 0, s>>>(d_rows, stride, n, d_state, decay, hold);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------------
 * metronome/metronome.go:63-131.  The two counters have closed forms inside a buffer: up to and including sample j0 (the
 * last one before the first beat boundary) the sample counter is sc0 + i; after it sample i lies m = i - j0 - 1 samples
 * into a run of whole beats: counter = m mod spb, resets so far = 1 + m div spb.  The host keeps the two counters.
 * ---------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
metronome_kernel(const double *__restrict__ tick, unsigned n_tick, const double *__restrict__ tock, unsigned n_tock, double *__restrict__ out, int n,
                 unsigned sc0, unsigned tc0, unsigned spb, unsigned beats, unsigned j0) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        unsigned sc, tc;
        if ((unsigned)i <= j0) { sc = sc0 + (unsigned)i; tc = tc0; }
        else {
            unsigned m = (unsigned)i - j0 - 1u;
            unsigned resets = 1u + (spb ? m / spb : m);
            sc = spb ? m % spb : 0u;
            tc = ((tc0 + 1u) % beats + (resets - 1u) % beats) % beats;
        }
        double sample = 0.0;
        if (tc == 0) { if (tick && sc < n_tick) sample = tick[sc]; }
        else { if (tock && sc < n_tock) sample = tock[sc]; }
        out[i] = sample;
    }
}

__global__ void __launch_bounds__(256) add_aux_kernel(double *__restrict__ a, double *__restrict__ b, const double *__restrict__ src, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const double v = src[i]; a[i] += v; b[i] += v; }
}
hipError_t gdg_launch_add_aux(double *d_a, double *d_b, const double *d_src, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    add_aux_kernel<<<(n + 255) / 256, 256, 0, s>>>(d_a, d_b, d_src, n);
    return hipGetLastError();
}

/* dst[i] += src[i]: the shards' partial master sums, added in shard order (gdg_batch_finish_master) */
__global__ void __launch_bounds__(256) accumulate_kernel(double *__restrict__ dst, const double *__restrict__ src, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
hipError_t gdg_launch_accumulate(double *d_dst, const double *d_src, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    accumulate_kernel<<<(n + 255) / 256, 256, 0, s>>>(d_dst, d_src, n);
    return hipGetLastError();
}

hipError_t gdg_launch_metronome(const double *d_tick, unsigned n_tick, const double *d_tock, unsigned n_tock, double *d_out, int n,
                                unsigned sc0, unsigned tc0, unsigned spb, unsigned beats, unsigned j0, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    metronome_kernel<<<(n + 255) / 256, 256, 0, s>>>(d_tick, n_tick, d_tock, n_tock, d_out, n, sc0, tc0, spb, beats, j0);
    return hipGetLastError();
}

static int grid_for(size_t n) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    return (int)(blocks ? blocks : 1);
}

/* ---- mono fast path: four samples per thread, word-sized loads and stores ---------------------------------------- */
template <int FMT> struct fmt_width { static const int W = FMT == GDG_FMT_LPCM8 ? 1 : FMT == GDG_FMT_LPCM16 ? 2 : FMT == GDG_FMT_LPCM24 ? 3 : 4; };
__device__ __forceinline__ unsigned getb(const unsigned *w, int k) { return (w[k >> 2] >> ((k & 3) * 8)) & 0xffu; }

template <int FMT>
__device__ __forceinline__ double decode_code(unsigned code) {       /* code = the sample's little-endian bytes */
    if (FMT == GDG_FMT_LPCM8) {
        short temp = (short)((short)code + (-128));
        double r = (1.0 / 127.0) * (double)temp;
        return r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    } else if (FMT == GDG_FMT_LPCM16) {
        return (2.0 / 65535.0) * (double)(short)(unsigned short)code;
    } else if (FMT == GDG_FMT_LPCM24) {
        int v = (int)code;
        if (code & SIGN_BIT_INT24) v = MIN_INT24 + (v & MAX_INT24);
        return (2.0 / 16777215.0) * (double)v;
    } else if (FMT == GDG_FMT_LPCM32) {
        return (2.0 / 4294967295.0) * (double)(int)code;
    } else {
        return (double)__uint_as_float(code);
    }
}

template <int FMT>
__device__ __forceinline__ unsigned encode_code(double sample) {
    sample = clamp1(sample);
    if (FMT == GDG_FMT_LPCM8) {
        short temp = (short)(127.0 * sample);
        int res = temp + 128;
        return (unsigned)(res < 0 ? 0 : (res > 255 ? 255 : res));
    } else if (FMT == GDG_FMT_LPCM16) {
        int tmp = (int)((0.5 * 65535.0) * sample);
        tmp = tmp > 32767 ? 32767 : (tmp < -32768 ? -32768 : tmp);
        return (unsigned)tmp & 0xffffu;
    } else if (FMT == GDG_FMT_LPCM24) {
        int tmp = (int)((0.5 * 16777215.0) * sample);
        tmp = tmp > MAX_INT24 ? MAX_INT24 : (tmp < MIN_INT24 ? MIN_INT24 : tmp);
        return (unsigned)tmp & 0xffffffu;
    } else if (FMT == GDG_FMT_LPCM32) {
        long long tmp = (long long)((0.5 * 4294967295.0) * sample);
        tmp = tmp > 2147483647LL ? 2147483647LL : (tmp < -2147483648LL ? -2147483648LL : tmp);
        return (unsigned)(int)tmp;
    } else {
        return __float_as_uint((float)sample);
    }
}

typedef double v2d __attribute__((ext_vector_type(2)));

template <int FMT, bool NT>
__global__ void __launch_bounds__(256)
wave_decode4_kernel(const unsigned *__restrict__ words, size_t groups, v2d *__restrict__ out) {
    constexpr int W = fmt_width<FMT>::W;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (size_t)gridDim.x * 256) {
        unsigned w[W];
#pragma unroll
        for (int k = 0; k < W; k++) w[k] = NT ? __builtin_nontemporal_load(words + g * W + k) : words[g * W + k];
        double r[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            unsigned code = 0;
#pragma unroll
            for (int k = 0; k < W; k++) code |= getb(w, s * W + k) << (8 * k);
            r[s] = decode_code<FMT>(code);
        }
        v2d a = { r[0], r[1] }, b = { r[2], r[3] };
        if (NT) { __builtin_nontemporal_store(a, out + 2 * g); __builtin_nontemporal_store(b, out + 2 * g + 1); }
        else { out[2 * g] = a; out[2 * g + 1] = b; }
    }
}

template <int FMT, bool NT>
__global__ void __launch_bounds__(256)
wave_encode4_kernel(const v2d *__restrict__ in, size_t groups, unsigned *__restrict__ words) {
    constexpr int W = fmt_width<FMT>::W;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (size_t)gridDim.x * 256) {
        v2d a = NT ? __builtin_nontemporal_load(in + 2 * g) : in[2 * g], b = NT ? __builtin_nontemporal_load(in + 2 * g + 1) : in[2 * g + 1];
        double r[4] = { a.x, a.y, b.x, b.y };
        unsigned w[W];
#pragma unroll
        for (int k = 0; k < W; k++) w[k] = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            unsigned code = encode_code<FMT>(r[s]);
#pragma unroll
            for (int k = 0; k < W; k++) {
                int byte = s * W + k;
                w[byte >> 2] |= ((code >> (8 * k)) & 0xffu) << ((byte & 3) * 8);
            }
        }
#pragma unroll
        for (int k = 0; k < W; k++) { if (NT) __builtin_nontemporal_store(w[k], words + g * W + k); else words[g * W + k] = w[k]; }
    }
}

template <int FMT>
static void launch_decode(const unsigned char *p, size_t per, unsigned channels, double *d_out, hipStream_t s) {
    size_t n = per * channels, done = 0;
    if (FMT != GDG_FMT_IEEE64 && channels == 1 && n >= 4 && ((uintptr_t)p & 3) == 0 && ((uintptr_t)d_out & 15) == 0) {
        size_t groups = n / 4;
        /* measured (profiles/io_variants_r01.txt): one group per thread without a grid cap; plain stores for decode, non-temporal for encode */
        wave_decode4_kernel<FMT, false><<<(unsigned)((groups + 255) / 256), 256, 0, s>>>(reinterpret_cast<const unsigned *>(p), groups, reinterpret_cast<v2d *>(d_out));
        done = groups * 4;
    }
    if (done < n) {         /* tail, multi-channel files, unaligned buffers: the one-sample-per-thread kernel */
        size_t off = done * gdg_wave_bytes_per_sample(FMT);
        if (channels == 1) wave_decode_kernel<FMT><<<grid_for(n - done), 256, 0, s>>>(p + off, n - done, d_out + done, 1, n - done);
        else wave_decode_kernel<FMT><<<grid_for(n), 256, 0, s>>>(p, n, d_out, channels, per);
    }
}

template <int FMT>
static void launch_encode(const double *d_in, size_t per, unsigned channels, unsigned char *p, hipStream_t s) {
    size_t n = per * channels, done = 0;
    if (FMT != GDG_FMT_IEEE64 && channels == 1 && n >= 4 && ((uintptr_t)p & 3) == 0 && ((uintptr_t)d_in & 15) == 0) {
        size_t groups = n / 4;
        wave_encode4_kernel<FMT, true><<<(unsigned)((groups + 255) / 256), 256, 0, s>>>(reinterpret_cast<const v2d *>(d_in), groups, reinterpret_cast<unsigned *>(p));
        done = groups * 4;
    }
    if (done < n) {
        size_t off = done * gdg_wave_bytes_per_sample(FMT);
        if (channels == 1) wave_encode_kernel<FMT><<<grid_for(n - done), 256, 0, s>>>(d_in + done, n - done, p + off, 1, n - done);
        else wave_encode_kernel<FMT><<<grid_for(n), 256, 0, s>>>(d_in, n, p, channels, per);
    }
}

template <int FMT>
__device__ __forceinline__ void decode_piece(const gdg_decode_row &r) {
    constexpr int W = fmt_width<FMT>::W;
    unsigned done = 0;
    if ((((uintptr_t)r.src) & 3) == 0 && (((uintptr_t)r.dst) & 15) == 0) {
        /* word-sized loads, four samples per thread: byte loads fetch every line W times over (the batch run's pieces are 16-byte aligned) */
        const unsigned *words = reinterpret_cast<const unsigned *>(r.src);
        v2d *out = reinterpret_cast<v2d *>(r.dst);
        const unsigned groups = r.count / 4;
        for (unsigned g = blockIdx.x * 256 + threadIdx.x; g < groups; g += gridDim.x * 256) {
            unsigned w[W];
#pragma unroll
            for (int k = 0; k < W; k++) w[k] = __builtin_nontemporal_load(words + (size_t)g * W + k);
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                unsigned code = 0;
#pragma unroll
                for (int k = 0; k < W; k++) code |= getb(w, q * W + k) << (8 * k);
                v[q] = decode_code<FMT>(code);
            }
            v2d a = { v[0], v[1] }, b = { v[2], v[3] };
            out[2 * (size_t)g] = a;
            out[2 * (size_t)g + 1] = b;
        }
        done = groups * 4;
    }
    for (unsigned i = done + blockIdx.x * 256 + threadIdx.x; i < r.count; i += gridDim.x * 256) {
        unsigned code = 0;
#pragma unroll
        for (int k = 0; k < W; k++) code |= (unsigned)r.src[(size_t)i * W + k] << (8 * k);
        r.dst[i] = decode_code<FMT>(code);
    }
}

/* blockIdx.y = piece; every piece has its own format (uniform per workgroup), source and destination */
__global__ void __launch_bounds__(256)
wave_decode_rows_kernel(const gdg_decode_row *__restrict__ rows) {
    const gdg_decode_row r = rows[blockIdx.y];
    switch (r.fmt) {
    case GDG_FMT_LPCM8: decode_piece<GDG_FMT_LPCM8>(r); break;
    case GDG_FMT_LPCM16: decode_piece<GDG_FMT_LPCM16>(r); break;
    case GDG_FMT_LPCM24: decode_piece<GDG_FMT_LPCM24>(r); break;
    case GDG_FMT_LPCM32: decode_piece<GDG_FMT_LPCM32>(r); break;
    case GDG_FMT_IEEE32: decode_piece<GDG_FMT_IEEE32>(r); break;
    default:                                                   /* IEEE64 (wave.go:714-732): the bytes are the sample; src is 8-byte aligned */
        for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < r.count; i += gridDim.x * 256) r.dst[i] = reinterpret_cast<const double *>(r.src)[i];
        break;
    }
}

hipError_t gdg_launch_wave_decode_rows(const gdg_decode_row *d_rows, int n_rows, unsigned max_count, hipStream_t s) {
    if (n_rows <= 0 || max_count == 0) return hipSuccess;
    unsigned tiles = (max_count + 1023) / 1024;                /* four samples per thread */
    if (tiles > 64) tiles = 64;
    wave_decode_rows_kernel<<<dim3(tiles, (unsigned)n_rows), dim3(256), 0, s>>>(d_rows);
    return hipGetLastError();
}

hipError_t gdg_launch_wave_decode(int fmt, const void *d_bytes, size_t per, unsigned channels, double *d_out, hipStream_t s) {
    size_t n = per * channels;
    if (n == 0) return hipSuccess;
    const unsigned char *p = static_cast<const unsigned char *>(d_bytes);
    switch (fmt) {
    case GDG_FMT_LPCM8: launch_decode<GDG_FMT_LPCM8>(p, per, channels, d_out, s); break;
    case GDG_FMT_LPCM16: launch_decode<GDG_FMT_LPCM16>(p, per, channels, d_out, s); break;
    case GDG_FMT_LPCM24: launch_decode<GDG_FMT_LPCM24>(p, per, channels, d_out, s); break;
    case GDG_FMT_LPCM32: launch_decode<GDG_FMT_LPCM32>(p, per, channels, d_out, s); break;
    case GDG_FMT_IEEE32: launch_decode<GDG_FMT_IEEE32>(p, per, channels, d_out, s); break;
    case GDG_FMT_IEEE64: launch_decode<GDG_FMT_IEEE64>(p, per, channels, d_out, s); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t gdg_launch_wave_encode(int fmt, const double *d_in, size_t per, unsigned channels, void *d_bytes, hipStream_t s) {
    size_t n = per * channels;
    if (n == 0) return hipSuccess;
    unsigned char *p = static_cast<unsigned char *>(d_bytes);
    switch (fmt) {
    case GDG_FMT_LPCM8: launch_encode<GDG_FMT_LPCM8>(d_in, per, channels, p, s); break;
    case GDG_FMT_LPCM16: launch_encode<GDG_FMT_LPCM16>(d_in, per, channels, p, s); break;
    case GDG_FMT_LPCM24: launch_encode<GDG_FMT_LPCM24>(d_in, per, channels, p, s); break;
    case GDG_FMT_LPCM32: launch_encode<GDG_FMT_LPCM32>(d_in, per, channels, p, s); break;
    case GDG_FMT_IEEE32: launch_encode<GDG_FMT_IEEE32>(d_in, per, channels, p, s); break;
    case GDG_FMT_IEEE64: launch_encode<GDG_FMT_IEEE64>(d_in, per, channels, p, s); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

/* n_rows rows of row_len samples each, row r at d_in + r * row_stride, encoded into COMPACT rows of row_len * width bytes: the batch
 * run's window keeps one row stride for every step (full windows and the tail), so one plan serves the whole batch; blockIdx.y = row.
 * row_len a multiple of 4, rows 16-byte aligned (the batch run's rows are multiples of 8192 samples). */
template <int FMT>
__global__ void __launch_bounds__(256)
wave_encode4_rows_kernel(const double *__restrict__ in, size_t row_stride, size_t groups_per_row, unsigned *__restrict__ words) {
    constexpr int W = fmt_width<FMT>::W;
    const v2d *row = reinterpret_cast<const v2d *>(in + (size_t)blockIdx.y * row_stride);
    unsigned *dst = words + (size_t)blockIdx.y * groups_per_row * W;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < groups_per_row; g += (size_t)gridDim.x * 256) {
        v2d a = __builtin_nontemporal_load(row + 2 * g), b = __builtin_nontemporal_load(row + 2 * g + 1);
        double r[4] = { a.x, a.y, b.x, b.y };
        unsigned w[W];
#pragma unroll
        for (int k = 0; k < W; k++) w[k] = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            unsigned code = encode_code<FMT>(r[s]);
#pragma unroll
            for (int k = 0; k < W; k++) {
                int byte = s * W + k;
                w[byte >> 2] |= ((code >> (8 * k)) & 0xffu) << ((byte & 3) * 8);
            }
        }
#pragma unroll
        for (int k = 0; k < W; k++) __builtin_nontemporal_store(w[k], dst + g * W + k);
    }
}

template <int FMT>
static void launch_encode_rows(const double *d_in, size_t row_stride, size_t row_len, unsigned n_rows, unsigned char *p, hipStream_t s) {
    size_t groups = row_len / 4;
    unsigned tiles = (unsigned)((groups + 255) / 256);
    wave_encode4_rows_kernel<FMT><<<dim3(tiles, n_rows), dim3(256), 0, s>>>(d_in, row_stride, groups, reinterpret_cast<unsigned *>(p));
}

hipError_t gdg_launch_wave_encode_rows(int fmt, const double *d_in, size_t row_stride, size_t row_len, unsigned n_rows, void *d_bytes, hipStream_t s) {
    if (n_rows == 0 || row_len == 0) return hipSuccess;
    if ((row_len & 3) || (row_stride & 1) || ((uintptr_t)d_in & 15) || ((uintptr_t)d_bytes & 3)) return hipErrorInvalidValue;
    unsigned char *p = static_cast<unsigned char *>(d_bytes);
    switch (fmt) {
    case GDG_FMT_LPCM8: launch_encode_rows<GDG_FMT_LPCM8>(d_in, row_stride, row_len, n_rows, p, s); break;
    case GDG_FMT_LPCM16: launch_encode_rows<GDG_FMT_LPCM16>(d_in, row_stride, row_len, n_rows, p, s); break;
    case GDG_FMT_LPCM24: launch_encode_rows<GDG_FMT_LPCM24>(d_in, row_stride, row_len, n_rows, p, s); break;
    case GDG_FMT_LPCM32: launch_encode_rows<GDG_FMT_LPCM32>(d_in, row_stride, row_len, n_rows, p, s); break;
    case GDG_FMT_IEEE32: launch_encode_rows<GDG_FMT_IEEE32>(d_in, row_stride, row_len, n_rows, p, s); break;
    case GDG_FMT_IEEE64:                                       /* wave.go:694-709: the sample's bytes, no clipping */
        return hipMemcpy2DAsync(p, row_len * sizeof(double), d_in, row_stride * sizeof(double), row_len * sizeof(double), n_rows, hipMemcpyDeviceToDevice, s);
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t gdg_launch_resample_time(const double *d_in, int n, double dx, double *d_out, int n_out, hipStream_t s) {
    if (n_out <= 0) return hipSuccess;
    resample_time_kernel<<<grid_for((size_t)n_out), 256, 0, s>>>(d_in, n, dx, d_out, n_out);
    return hipGetLastError();
}
