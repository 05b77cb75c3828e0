/*
 * fir.hip -- the FFT convolution behind the power amp (effects/poweramp.go:186-216 ->
 * filter/filter.go:342-515 -> fft/fft.go:744-990), redesigned for gfx950.
 *
 * The reference runs an UNPARTITIONED overlap-add: one 2*nextpow2(L)-point real FFT pair per
 * frame and channel.  Here the same y = clip(x * h) is computed as a uniformly partitioned
 * overlap-save convolution with partition P = frame:
 *
 *   fir_fwd   one workgroup per channel: [x_{t-1} | x_t] (2P reals) -> packed P-point complex
 *             Stockham FFT held entirely in registers + LDS (P = 8192: 16 points per thread,
 *             512 threads, 139 KiB of LDS -- only possible with CDNA4's 160 KiB) -> half spectrum
 *             written to slot pos % R of the channel's frequency-domain delay line (FDL) in HBM.
 *   fir_inv   one workgroup per channel.  Default (FUSED): the spectrum multiply-accumulate
 *             Y[b] = sum_k FDL[(pos - k) % R][b] * H[k][b] runs HERE, partition by partition through all bins
 *             (two long sequential streams per workgroup, 32 sixteen-byte loads per lane in flight), and lands
 *             directly in the inverse transform's first stage in LDS -> packed inverse FFT -> second half ->
 *             clip -> out.  This is the HBM-bound kernel (the roofline one).
 *   fir_mac   the same multiply-accumulate as a kernel of its own writing Y (GDG_FIR_FUSED=0, kept for A/B runs:
 *             the Y store alone costs it 20 % of its time, profiles/probes/).
 *
 * FP64 throughout; no MFMA (there is no dense contraction here).  Spectra are stored as P
 * complex128 per slot: bin 0 carries (Re X[0], Re X[P]) since both are real.
 */
#include "gdg_internal.h"
#include <hip/hip_ext.h>
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

typedef double2 cplx;

__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
/* Rounding is spelled out in this file: no contraction by the compiler (which product of a sum of two it fuses depends on the code around
 * it -- two kernels inlining the same butterfly could differ in the last bit, and the window / chained / one-buffer variants of a
 * transform must give the SAME bits as the per-frame kernels), fused multiply-adds written where they are wanted. */
#pragma clang fp contract(off)
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return make_double2(__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x));
}

typedef double v2d __attribute__((ext_vector_type(2)));
/* Frames and spectra live in HBM, but their pointers come out of the channel descriptors in memory, so the compiler only knows
 * "generic" and would emit FLAT loads / stores -- which also count as LDS operations, so every wait for an LDS read would
 * wait for the outstanding HBM accesses too.  These accessors restore the global address space. */
#define GDG_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ cplx gload(const cplx *p) {
    v2d v = *reinterpret_cast<const GDG_GLOBAL v2d *>((const GDG_GLOBAL void *)p);
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void gstore(cplx *p, cplx c) {
    v2d v = { c.x, c.y };
    *reinterpret_cast<GDG_GLOBAL v2d *>((GDG_GLOBAL void *)p) = v;
}
/* written once, read a gigabyte of traffic later (a window's spectra and products): past the caches */
__device__ __forceinline__ void gstore_nt(cplx *p, cplx c) {
    v2d v = { c.x, c.y };
    __builtin_nontemporal_store(v, reinterpret_cast<GDG_GLOBAL v2d *>((GDG_GLOBAL void *)p));
}
__device__ __forceinline__ double gload1(const double *p) { return *(const GDG_GLOBAL double *)p; }
__device__ __forceinline__ void gstore1(double *p, double x) { *(GDG_GLOBAL double *)p = x; }


#define GDG_C1 0.92387953251128673848      /* cos(pi/8) */
#define GDG_S1 0.38268343236508978178      /* sin(pi/8) */
#define GDG_RH 0.70710678118654752440      /* sqrt(1/2) */

/* multiply by W_16^M = exp(-2 pi i M / 16) (forward) or its conjugate (inverse), M known at compile time */
template <int M, bool INV>
__device__ __forceinline__ cplx mul_w16(cplx a) {
    constexpr int m = M & 15;
    if constexpr (m == 0) {
        return a;
    } else if constexpr (m == 4) {
        return INV ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
    } else if constexpr (m == 2) {
        return INV ? make_double2((a.x - a.y) * GDG_RH, (a.x + a.y) * GDG_RH)
                   : make_double2((a.x + a.y) * GDG_RH, (a.y - a.x) * GDG_RH);
    } else if constexpr (m == 6) {
        return INV ? make_double2((-a.x - a.y) * GDG_RH, (a.x - a.y) * GDG_RH)
                   : make_double2((a.y - a.x) * GDG_RH, (-a.x - a.y) * GDG_RH);
    } else {
        constexpr double wr = (m == 1) ? GDG_C1 : (m == 3) ? GDG_S1 : (m == 5) ? -GDG_S1 : -GDG_C1;
        constexpr double wi_f = (m == 1) ? -GDG_S1 : (m == 3) ? -GDG_C1 : (m == 5) ? -GDG_C1 : -GDG_S1;
        constexpr double wi = INV ? -wi_f : wi_f;
        return make_double2(__builtin_fma(a.x, wr, -(a.y * wi)), __builtin_fma(a.x, wi, a.y * wr));
    }
}

/* small DFTs on registers, natural order in and out (radix-2 decimation in time, fully unrolled) */
template <int R, bool INV> struct Dft;

template <int R, int K, bool INV> struct DftCombine {
    static __device__ __forceinline__ void run(cplx (&v)[R], const cplx (&e)[R / 2], const cplx (&o)[R / 2]) {
        if constexpr (K < R / 2) {
            cplx t = mul_w16<K * (16 / R), INV>(o[K]);
            v[K] = cadd(e[K], t);
            v[K + R / 2] = csub(e[K], t);
            DftCombine<R, K + 1, INV>::run(v, e, o);
        }
    }
};

template <bool INV> struct Dft<2, INV> {
    static __device__ __forceinline__ void run(cplx (&v)[2]) {
        cplx a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <int R, bool INV> struct Dft {
    static __device__ __forceinline__ void run(cplx (&v)[R]) {
        cplx e[R / 2], o[R / 2];
#pragma unroll
        for (int i = 0; i < R / 2; i++) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
        Dft<R / 2, INV>::run(e);
        Dft<R / 2, INV>::run(o);
        DftCombine<R, 0, INV>::run(v, e, o);
    }
};

/* pass schedule: log2 radices, largest first (so that the radix-16 pass has no twiddles) */
__host__ __device__ constexpr int sched_lr(int logn, int p) {
    switch (logn) {
    case 13: return p == 0 ? 4 : (p < 4 ? 3 : 0);
    case 12: return p < 3 ? 4 : 0;
    case 11: return p < 2 ? 4 : (p == 2 ? 3 : 0);
    case 10: return p == 0 ? 4 : (p < 3 ? 3 : 0);
    case 9: return p < 3 ? 3 : 0;
    case 8: return p < 2 ? 4 : 0;
    case 7: return p == 0 ? 4 : (p == 1 ? 3 : 0);
    case 6: return p < 2 ? 3 : 0;
    default: return 0;
    }
}
__host__ __device__ constexpr int sched_lns(int logn, int p) {
    int s = 0;
    for (int i = 0; i < p; i++) s += sched_lr(logn, i);
    return s;
}
__host__ __device__ constexpr int sched_npass(int logn) {
    int n = 0;
    while (n < 4 && sched_lr(logn, n) != 0) n++;
    return n;
}

#define GDG_PAD(e) ((e) + ((e) >> 4))

template <int LOGN> struct FftCfg {
    static constexpr int N = 1 << LOGN;
    static constexpr int T = N / 16;
    static constexpr int LDS = N + N / 16 + 16;
};

/* registers <- LDS for one pass: butterfly j = tid + T*b takes elements j + t*N/R */
template <int LOGN, int LOGR>
__device__ __forceinline__ void pass_load(cplx (&v)[16], const double *sre, const double *sim, int tid) {
    constexpr int N = 1 << LOGN, T = N / 16, R = 1 << LOGR, B = 16 / R;
#pragma unroll
    for (int b = 0; b < B; b++) {
        int j = tid + T * b;
#pragma unroll
        for (int t = 0; t < R; t++) {
            int e = j + t * (N / R);
            v[b * R + t] = make_double2(sre[GDG_PAD(e)], sim[GDG_PAD(e)]);
        }
    }
}

/* twiddle + radix-R DFT on the registers */
template <int LOGN, int LOGR, int LOGNS, bool INV, int TWS = 1>
__device__ __forceinline__ void pass_compute(cplx (&v)[16], const cplx *__restrict__ tw, int tid) {
    constexpr int N = 1 << LOGN, T = N / 16, R = 1 << LOGR, B = 16 / R, NS = 1 << LOGNS;
#pragma unroll
    for (int b = 0; b < B; b++) {
        int j = tid + T * b;
        cplx u[R];
#pragma unroll
        for (int t = 0; t < R; t++) u[t] = v[b * R + t];
        if constexpr (NS > 1) {
            int kk = j & (NS - 1);
            constexpr int stepm = N / (NS * R);
            /* the R - 1 twiddles w^t of this butterfly from ONE table load: w, w^2, w^4, w^8 by squaring, the others as
             * products of two of them (depth <= 4 multiplications, error ~4 ulp) instead of R - 1 scattered 16-byte loads */
            cplx wp[R];
            wp[1] = tw[kk * stepm * TWS];            /* TWS: the table belongs to a transform TWS times as long */
            if constexpr (INV) wp[1].y = -wp[1].y;
#pragma unroll
            for (int t = 2; t < R; t++) {
                const int hi = 1 << (31 - __builtin_clz(t)), lo = t - hi;       /* t = hi + lo, hi a power of two */
                wp[t] = (lo == 0) ? cmul(wp[t / 2], wp[t / 2]) : cmul(wp[hi], wp[lo]);
            }
#pragma unroll
            for (int t = 1; t < R; t++) u[t] = cmul(u[t], wp[t]);
        }
        Dft<R, INV>::run(u);
#pragma unroll
        for (int t = 0; t < R; t++) v[b * R + t] = u[t];
    }
}

/* registers -> LDS, Stockham auto-sort addressing */
template <int LOGN, int LOGR, int LOGNS>
__device__ __forceinline__ void pass_store(const cplx (&v)[16], double *sre, double *sim, int tid) {
    constexpr int N = 1 << LOGN, T = N / 16, R = 1 << LOGR, B = 16 / R, NS = 1 << LOGNS;
#pragma unroll
    for (int b = 0; b < B; b++) {
        int j = tid + T * b;
        int base = ((j >> LOGNS) << (LOGNS + LOGR)) + (j & (NS - 1));
#pragma unroll
        for (int t = 0; t < R; t++) {
            int e = base + t * NS;
            sre[GDG_PAD(e)] = v[b * R + t].x;
            sim[GDG_PAD(e)] = v[b * R + t].y;
        }
    }
}

/* passes FIRST..LAST-1 entirely through LDS (input already in LDS, output left in LDS) */
template <int LOGN, int P, int LAST, bool INV>
__device__ __forceinline__ void run_lds_passes(cplx (&v)[16], double *sre, double *sim, const cplx *tw, int tid) {
    if constexpr (P < LAST) {
        constexpr int LR = sched_lr(LOGN, P), LNS = sched_lns(LOGN, P);
        pass_load<LOGN, LR>(v, sre, sim, tid);
        __syncthreads();
        pass_compute<LOGN, LR, LNS, INV>(v, tw, tid);
        pass_store<LOGN, LR, LNS>(v, sre, sim, tid);
        __syncthreads();
        run_lds_passes<LOGN, P + 1, LAST, INV>(v, sre, sim, tw, tid);
    }
}

/* ---- one component at a time through LDS ------------------------------------------------------------------------------------------
 * The 8192-point transforms hold their data in registers (16 points per thread); LDS is only the exchange between passes.  With the real
 * parts and the imaginary parts taking turns in ONE buffer the exchange needs 70 KiB instead of 140, and TWO workgroups fit a CU: one
 * streams its frame in or its spectrum out while the other computes (a lone workgroup per CU adds its HBM, LDS and FP64 phases up:
 * profiles/experiments/README.md).  Same passes, same arithmetic, same bits. */
template <int C> __device__ __forceinline__ double &part(cplx &z) { if constexpr (C == 0) return z.x; else return z.y; }
template <int C> __device__ __forceinline__ double part(const cplx &z) { if constexpr (C == 0) return z.x; else return z.y; }

template <int LOGN, int LOGR, int C>
__device__ __forceinline__ void pass_load1(cplx (&v)[16], const double *s, int tid) {
    constexpr int N = 1 << LOGN, T = N / 16, R = 1 << LOGR, B = 16 / R;
#pragma unroll
    for (int b = 0; b < B; b++) {
        int j = tid + T * b;
#pragma unroll
        for (int t = 0; t < R; t++) part<C>(v[b * R + t]) = s[GDG_PAD(j + t * (N / R))];
    }
}
template <int LOGN, int LOGR, int LOGNS, int C>
__device__ __forceinline__ void pass_store1(const cplx (&v)[16], double *s, int tid) {
    constexpr int N = 1 << LOGN, T = N / 16, R = 1 << LOGR, B = 16 / R, NS = 1 << LOGNS;
#pragma unroll
    for (int b = 0; b < B; b++) {
        int j = tid + T * b;
        int base = ((j >> LOGNS) << (LOGNS + LOGR)) + (j & (NS - 1));
#pragma unroll
        for (int t = 0; t < R; t++) s[GDG_PAD(base + t * NS)] = part<C>(v[b * R + t]);
    }
}
/* results of pass P -> operands of pass P + 1, across the workgroup (WAVE = false) or inside one wave's region (WAVE = true: a wave's
 * LDS operations execute in order, nothing to wait for) */
template <int LOGN, int P, bool WAVE>
__device__ __forceinline__ void exchange1(cplx (&v)[16], double *s, int tid) {
    constexpr int LR = sched_lr(LOGN, P), LNS = sched_lns(LOGN, P), LR1 = sched_lr(LOGN, P + 1);
    auto meet = [&]() { if constexpr (WAVE) __builtin_amdgcn_wave_barrier(); else __syncthreads(); };
    meet();                                                   /* the buffer's previous contents have been read */
    pass_store1<LOGN, LR, LNS, 0>(v, s, tid);
    meet();
    pass_load1<LOGN, LR1, 0>(v, s, tid);
    meet();
    pass_store1<LOGN, LR, LNS, 1>(v, s, tid);
    meet();
    pass_load1<LOGN, LR1, 1>(v, s, tid);
}

/*
 * Forward: real [a | b] (2N reals) -> packed half spectrum (N complex).
 * IRJOB = false: a = prev ring half, b = current frame; writes FDL slot and the other prev half.
 * IRJOB = true : a = one zero-padded IR partition, b = zeros; result scaled (1/(2P) folded into H).
 */
template <int LOGN, bool IRJOB>
__global__ void __launch_bounds__(FftCfg<LOGN>::T)
fir_fwd_kernel(const gdg_fir_chan *__restrict__ chans, const gdg_fir_irjob *__restrict__ jobs, double scale, gdg_shift shift,
               const cplx *__restrict__ tw, const cplx *__restrict__ tw2) {
    constexpr int N = FftCfg<LOGN>::N, T = FftCfg<LOGN>::T;
    __shared__ double sre[FftCfg<LOGN>::LDS];
    __shared__ double sim[FftCfg<LOGN>::LDS];
    const int tid = threadIdx.x;
    const double *a, *bsrc;
    double *prev_out = nullptr;
    cplx *out;
    int hop = N;
    if constexpr (IRJOB) {
        gdg_fir_irjob jb = jobs[blockIdx.x];
        a = jb.a;
        bsrc = jb.b;
        out = jb.out;
        if (jb.hop > 0) hop = jb.hop;                    /* [a (hop) | b (hop) | zeros]; hop == 0: [a (N) | zeros] */
        else bsrc = nullptr;
    } else {
        gdg_fir_chan ch = chans[blockIdx.x];
        int pos = *ch.pos;
        a = ch.prev + (size_t)((pos + 1) & 1) * N;       /* previous frame */
        prev_out = ch.prev + (size_t)(pos & 1) * N;       /* where this frame is kept for the next call */
        bsrc = (ch.flags & GDG_SRC_IS_INPUT) ? ch.src + shift.in : ch.src;
        out = ch.fdl + (size_t)(pos % ch.R) * N;
        hop = ch.hop;
    }

    /* pass 0 straight from global memory: packed element e = (r[2e], r[2e+1]) */
    constexpr int LR0 = sched_lr(LOGN, 0), R0 = 1 << LR0, B0 = 16 / R0;
    cplx v[16];
#pragma unroll
    for (int b = 0; b < B0; b++) {
        int j = tid + T * b;
#pragma unroll
        for (int t = 0; t < R0; t++) {
            int e = j + t * (N / R0);
            cplx val;
            if ((!IRJOB && hop != N) || (IRJOB && bsrc != nullptr)) {
                /* frame shorter than the transform half: r = [previous (hop) | current (hop) | zeros], any parity of hop
                 * (also the re-partitioning jobs, whatever their hop: their frames need not be 16-byte aligned) */
                double r2[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int i = 2 * e + h;
                    double x = 0.0;
                    if (i < hop) x = gload1(a + i);
                    else if (i < 2 * hop) { x = gload1(bsrc + (i - hop)); if constexpr (!IRJOB) gstore1(prev_out + (i - hop), x); }
                    r2[h] = x;
                }
                val = make_double2(r2[0], r2[1]);
            } else if (e < N / 2) {
                val = gload(reinterpret_cast<const cplx *>(a + 2 * e));
            } else {
                if constexpr (IRJOB) val = make_double2(0.0, 0.0);
                else {
                    val = gload(reinterpret_cast<const cplx *>(bsrc + 2 * (e - N / 2)));
                    gstore(reinterpret_cast<cplx *>(prev_out + 2 * (e - N / 2)), val);
                }
            }
            v[b * R0 + t] = val;
        }
    }
    pass_compute<LOGN, LR0, 0, false>(v, tw, tid);
    pass_store<LOGN, LR0, 0>(v, sre, sim, tid);
    __syncthreads();
    run_lds_passes<LOGN, 1, sched_npass(LOGN), false>(v, sre, sim, tw, tid);

    /* un-pack: X[k] and X[N-k] from Z[k], Z[N-k] (Z natural order in LDS) */
    constexpr int ITER = (N / 2) / T;      /* = 8 */
#pragma unroll
    for (int i = 0; i < ITER; i++) {
        int k = tid + T * i;
        if (k == 0) {
            double zx = sre[0], zy = sim[0];
            gstore(out, make_double2((zx + zy) * scale, (zx - zy) * scale));
            double hx = sre[GDG_PAD(N / 2)], hy = sim[GDG_PAD(N / 2)];
            gstore(out + N / 2, make_double2(hx * scale, -hy * scale));
        } else {
            int n = N - k;
            cplx zk = make_double2(sre[GDG_PAD(k)], sim[GDG_PAD(k)]);
            cplx zn = make_double2(sre[GDG_PAD(n)], sim[GDG_PAD(n)]);
            cplx A = make_double2(zk.x + zn.x, zk.y - zn.y);
            cplx Bv = make_double2(zk.x - zn.x, zk.y + zn.y);
            cplx cw = cmul(tw2[k], Bv);
            double hs = 0.5 * scale;
            gstore(out + k, make_double2((A.x + cw.y) * hs, (A.y - cw.x) * hs));
            gstore(out + n, make_double2((A.x - cw.y) * hs, (-A.y - cw.x) * hs));
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * 8192-point transforms as 8 x 1024 (decimation in frequency): one radix-8 step across the workgroup, then every WAVE runs a
 * 1024-point Stockham transform on its own LDS region with wave-local exchanges only -- the eight waves stop meeting at a
 * barrier after every pass (2 workgroup barriers per forward transform instead of 8).
 *   X[8 k1 + k2] = sum_n1 [ W_N^(n1 k2) * ( sum_n2 x[n1 + 1024 n2] W_8^(n2 k2) ) ] W_1024^(n1 k1)
 * Region r = k2 holds y[.][k2] (1024 points, padded); RL mod 32 = 4 keeps the k-order reads of the un-packing conflict free.
 * ---------------------------------------------------------------------------------------------- */
#define GDG_W_RL 1124
#define GDG_W_LDS (8 * GDG_W_RL)

/* 1024-point transform of one region by one wave: passes [16, 8, 8]; results stay in v (natural order:
 * v[b * 8 + t] = X[(lane + 64 b) + 128 t]) */
template <bool INV>
__device__ __forceinline__ void wave_fft1024(cplx (&v)[16], double *rre, double *rim, const cplx *__restrict__ tw, int lane) {
    constexpr int L = 10;
    static_assert(sched_lr(L, 0) == 4 && sched_lr(L, 1) == 3 && sched_lr(L, 2) == 3 && sched_npass(L) == 3, "1024 = 16 x 8 x 8");
    pass_load<L, 4>(v, rre, rim, lane);
    __builtin_amdgcn_wave_barrier();
    pass_compute<L, 4, 0, INV, 8>(v, tw, lane);
    pass_store<L, 4, 0>(v, rre, rim, lane);
    __builtin_amdgcn_wave_barrier();
    pass_load<L, 3>(v, rre, rim, lane);
    __builtin_amdgcn_wave_barrier();
    pass_compute<L, 3, 4, INV, 8>(v, tw, lane);
    pass_store<L, 3, 4>(v, rre, rim, lane);
    __builtin_amdgcn_wave_barrier();
    pass_load<L, 3>(v, rre, rim, lane);
    __builtin_amdgcn_wave_barrier();
    pass_compute<L, 3, 7, INV, 8>(v, tw, lane);
}

/* y[k2] *= w^k2, k2 = 1..7, from w alone (w^2, w^4 by squaring, the rest as products) */
__device__ __forceinline__ void twiddle_powers8(cplx (&u)[8], cplx w) {
    cplx w2 = cmul(w, w), w4 = cmul(w2, w2), w3 = cmul(w2, w), w5 = cmul(w4, w), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
    u[1] = cmul(u[1], w); u[2] = cmul(u[2], w2); u[3] = cmul(u[3], w3); u[4] = cmul(u[4], w4);
    u[5] = cmul(u[5], w5); u[6] = cmul(u[6], w6); u[7] = cmul(u[7], w7);
}

/* TB = 0: one frame per channel (blockIdx.x = channel).  TB = 1: a window of W consecutive frames per channel (time blocking):
 * blockIdx.x = channel * W + j; frame j's predecessor is frame j - 1 of the same call (only frame 0 looks into `prev`), nothing is
 * written to `prev` and the frame counter is left alone -- fir_tb_finish_kernel does both once the window is through. */
template <int TB>
__global__ void __launch_bounds__(512)
fir_fwd13w_kernel(const gdg_fir_chan *__restrict__ chans, int W, gdg_shift shift, const cplx *__restrict__ tw, const cplx *__restrict__ tw2) {
    constexpr int N = 8192, T = 512;
    __shared__ double sre[GDG_W_LDS];
    __shared__ double sim[GDG_W_LDS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int jw = TB ? (int)(blockIdx.x % (unsigned)W) : 0;
    gdg_fir_chan ch = chans[TB ? blockIdx.x / (unsigned)W : blockIdx.x];
    if (ch.flags & GDG_SRC_IS_INPUT) ch.src += shift.in;
    const int pos = *ch.pos;
    const double *a = ch.prev + (size_t)((pos + 1) & 1) * N;       /* previous frame */
    if (TB && jw > 0) a = ch.src + (size_t)jw * N - N;
    double *prev_out = ch.prev + (size_t)(pos & 1) * N;            /* where this frame is kept for the next call */
    const double *bsrc = ch.src + (size_t)jw * N;
    cplx *out = ch.fdl + (size_t)((pos + jw) % ch.R) * N;

    /* step A: radix-8 across the workgroup, straight from global memory (element e = n1 + 1024 n2: e < N/2 is the previous frame).
     * All sixteen frame loads and both twiddle loads of the thread are issued before anything is consumed: the copy of the frame
     * into `prev` used to sit right behind each load, and the compiler waited for the loads one by one. */
    cplx ua[2][8], wa[2];
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int n1 = tid + T * b;
        wa[b] = tw[n1];
#pragma unroll
        for (int n2 = 0; n2 < 8; n2++) {
            const int e = n1 + 1024 * n2;
            ua[b][n2] = (n2 < 4) ? gload(reinterpret_cast<const cplx *>(a + 2 * e)) : gload(reinterpret_cast<const cplx *>(bsrc + 2 * (e - N / 2)));
        }
    }
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int n1 = tid + T * b;
        if constexpr (!TB) {
#pragma unroll
            for (int n2 = 4; n2 < 8; n2++) gstore(reinterpret_cast<cplx *>(prev_out + 2 * (n1 + 1024 * n2 - N / 2)), ua[b][n2]);
        }
        Dft<8, false>::run(ua[b]);
        twiddle_powers8(ua[b], wa[b]);
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) { sre[k2 * GDG_W_RL + GDG_PAD(n1)] = ua[b][k2].x; sim[k2 * GDG_W_RL + GDG_PAD(n1)] = ua[b][k2].y; }
    }
    __syncthreads();
    /* step B: wave w transforms region w; X[8 k1 + w] back into the region in natural k1 order */
    {
        double *rre = sre + wave * GDG_W_RL, *rim = sim + wave * GDG_W_RL;
        cplx v[16];
        wave_fft1024<false>(v, rre, rim, tw, lane);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int k1 = (lane + 64 * b) + 128 * t;
                rre[GDG_PAD(k1)] = v[b * 8 + t].x;
                rim[GDG_PAD(k1)] = v[b * 8 + t].y;
            }
    }
    __syncthreads();
    /* un-pack: X[k] and X[N-k] from Z[k], Z[N-k]; Z[k] = region[k & 7][k >> 3] */
    auto Z = [&](int k) { return make_double2(sre[(k & 7) * GDG_W_RL + GDG_PAD(k >> 3)], sim[(k & 7) * GDG_W_RL + GDG_PAD(k >> 3)]); };
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int k = tid + T * i;
        if (k == 0) {
            cplx z0 = Z(0), zh = Z(N / 2);
            gstore(out, make_double2(z0.x + z0.y, z0.x - z0.y));
            gstore(out + N / 2, make_double2(zh.x, -zh.y));
        } else {
            const int n = N - k;
            cplx zk = Z(k), zn = Z(n);
            cplx A = make_double2(zk.x + zn.x, zk.y - zn.y);
            cplx Bv = make_double2(zk.x - zn.x, zk.y + zn.y);
            cplx cw = cmul(tw2[k], Bv);
            gstore(out + k, make_double2((A.x + cw.y) * 0.5, (A.y - cw.x) * 0.5));
            gstore(out + n, make_double2((A.x - cw.y) * 0.5, (-A.y - cw.x) * 0.5));
        }
    }
}

/* fir_fwd13w_kernel with one LDS buffer (see exchange1): two workgroups per CU, 128 VGPRs */
template <int TB>
__global__ void __launch_bounds__(512, 4)
fir_fwd13wh_kernel(const gdg_fir_chan *__restrict__ chans, int W, gdg_shift shift, const cplx *__restrict__ tw, const cplx *__restrict__ tw2, int by_xcd) {
    constexpr int N = 8192, T = 512;
    __shared__ double s[GDG_W_LDS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    /* by_xcd (channel count a multiple of 8): workgroups go round the eight XCDs, so XCD x takes the channels = x (mod 8), their frames in
     * order: frame j is read as "current" and, by the next workgroup of the same XCD, as "previous" -- the second read meets the first in L2 */
    unsigned cw = TB ? blockIdx.x / (unsigned)W : blockIdx.x, jw = TB ? blockIdx.x % (unsigned)W : 0u;
    if (TB && by_xcd) {
        const unsigned x = blockIdx.x & 7u, q = blockIdx.x >> 3;
        cw = x + 8u * (q / (unsigned)W);
        jw = q % (unsigned)W;
    }
    gdg_fir_chan ch = chans[cw];
    if (ch.flags & GDG_SRC_IS_INPUT) ch.src += shift.in;
    const int pos = *ch.pos;
    const double *a = ch.prev + (size_t)((pos + 1) & 1) * N;
    if (TB && jw > 0) a = ch.src + (size_t)jw * N - N;
    double *prev_out = ch.prev + (size_t)(pos & 1) * N;
    const double *bsrc = ch.src + (size_t)jw * N;
    cplx *out = ch.fdl + (size_t)((pos + jw) % ch.R) * N;

    cplx ua[2][8], wa[2];
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int n1 = tid + T * b;
        wa[b] = tw[n1];
#pragma unroll
        for (int n2 = 0; n2 < 8; n2++) {
            const int e = n1 + 1024 * n2;
            ua[b][n2] = (n2 < 4) ? gload(reinterpret_cast<const cplx *>(a + 2 * e)) : gload(reinterpret_cast<const cplx *>(bsrc + 2 * (e - N / 2)));
        }
    }
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int n1 = tid + T * b;
        if constexpr (!TB) {
#pragma unroll
            for (int n2 = 4; n2 < 8; n2++) gstore(reinterpret_cast<cplx *>(prev_out + 2 * (n1 + 1024 * n2 - N / 2)), ua[b][n2]);
        }
        Dft<8, false>::run(ua[b]);
        twiddle_powers8(ua[b], wa[b]);
        /* step A -> step B: thread (n1) holds y[n1][k2], wave k2 wants region k2; the real parts go first */
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) s[k2 * GDG_W_RL + GDG_PAD(n1)] = ua[b][k2].x;
    }
    double *rs = s + wave * GDG_W_RL;
    cplx v[16];
    __syncthreads();
    pass_load1<10, 4, 0>(v, rs, lane);
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) s[k2 * GDG_W_RL + GDG_PAD(tid + T * b)] = ua[b][k2].y;
    __syncthreads();
    pass_load1<10, 4, 1>(v, rs, lane);
    /* wave_fft1024 on the wave's own region */
    pass_compute<10, 4, 0, false, 8>(v, tw, lane);
    exchange1<10, 0, true>(v, rs, lane);
    pass_compute<10, 3, 4, false, 8>(v, tw, lane);
    exchange1<10, 1, true>(v, rs, lane);
    pass_compute<10, 3, 7, false, 8>(v, tw, lane);
    /* X[8 k1 + wave] back into the region in natural k1 order, then the un-packing in k order: real parts, then imaginary parts */
    auto Zs = [&](int k) { return s[(k & 7) * GDG_W_RL + GDG_PAD(k >> 3)]; };
    double zkx[8], znx[8];
    __builtin_amdgcn_wave_barrier();
    int ln = lane;
    asm volatile("" : "+v"(ln));                    /* likewise: these sixteen addresses are made here */
    double *rn = rs + GDG_PAD(ln);                  /* PAD(ln + c) = PAD(ln) + c + c / 16 for multiples of 16 */
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int t = 0; t < 8; t++) rn[GDG_PAD(64 * b + 128 * t)] = v[b * 8 + t].x;
    __syncthreads();
    int tu = tid;
    asm volatile("" : "+v"(tu));                    /* the un-packing's sixteen LDS addresses are made here, not hoisted to the top and spilled */
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int k = tu + T * i, n = (k == 0) ? N / 2 : N - k;
        zkx[i] = Zs(k);
        znx[i] = Zs(n);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int t = 0; t < 8; t++) rn[GDG_PAD(64 * b + 128 * t)] = v[b * 8 + t].y;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int k = tu + T * i;
        if (k == 0) {
            cplx z0 = make_double2(zkx[i], Zs(0)), zh = make_double2(znx[i], Zs(N / 2));
            gstore(out, make_double2(z0.x + z0.y, z0.x - z0.y));
            gstore(out + N / 2, make_double2(zh.x, -zh.y));
        } else {
            const int n = N - k;
            cplx zk = make_double2(zkx[i], Zs(k)), zn = make_double2(znx[i], Zs(n));
            cplx A = make_double2(zk.x + zn.x, zk.y - zn.y);
            cplx Bv = make_double2(zk.x - zn.x, zk.y + zn.y);
            cplx cw = cmul(tw2[k], Bv);
            gstore(out + k, make_double2((A.x + cw.y) * 0.5, (A.y - cw.x) * 0.5));
            gstore(out + n, make_double2((A.x - cw.y) * 0.5, (-A.y - cw.x) * 0.5));
        }
    }
}

/* The window's forward transforms, one workgroup per CHANNEL walking its W frames: the second half of a transform's input (the
 * current frame) is the first half of the next one's (the previous frame) and stays in the registers of the very threads that need it
 * -- element e = n1 + 1024 n2 of the packed sequence: n2 >= 4 now, n2 - 4 next time -- so every frame is read from HBM once instead of
 * twice (the per-frame workgroups of fir_fwd13w_kernel<1> read 1.07 GB per launch where 0.81 suffice, profiles/r02_window8_pmc.txt). */
__global__ void __launch_bounds__(512)
fir_fwd13w_chan_kernel(const gdg_fir_chan *__restrict__ chans, int W, gdg_shift shift, const cplx *__restrict__ tw, const cplx *__restrict__ tw2) {
    constexpr int N = 8192, T = 512;
    __shared__ double sre[GDG_W_LDS];
    __shared__ double sim[GDG_W_LDS];
    const int tid = threadIdx.x;
    gdg_fir_chan ch = chans[blockIdx.x];
    if (ch.flags & GDG_SRC_IS_INPUT) ch.src += shift.in;
    const int pos = *ch.pos;
    /* `fresh()` hides the thread index from the optimiser: otherwise every address of the loop body is loop-invariant, gets hoisted in
     * front of the loop and spills (profiles/experiments/README.md) */
    auto fresh = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
    cplx wa[2], pv[2][4];
    {
        const double *a = ch.prev + (size_t)((pos + 1) & 1) * N;
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int n1 = tid + T * b;
            wa[b] = tw[n1];
#pragma unroll
            for (int m = 0; m < 4; m++) pv[b][m] = gload(reinterpret_cast<const cplx *>(a + 2 * (n1 + 1024 * m)));
        }
    }
    for (int jw = 0; jw < W; jw++) {
        const double *bsrc = ch.src + (size_t)jw * N;
        cplx *out = ch.fdl + (size_t)((pos + jw) % ch.R) * N;
        cplx ua[2][8];
        {
            const int t0 = fresh();
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const int n1 = t0 + T * b;
#pragma unroll
                for (int m = 0; m < 4; m++) ua[b][4 + m] = gload(reinterpret_cast<const cplx *>(bsrc + 2 * (n1 + 1024 * m)));
            }
        }
        /* step A: radix-8 across the workgroup */
        {
            const int ta = fresh();
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const int n1 = ta + T * b;
#pragma unroll
                for (int m = 0; m < 4; m++) { ua[b][m] = pv[b][m]; pv[b][m] = ua[b][4 + m]; }
                Dft<8, false>::run(ua[b]);
                twiddle_powers8(ua[b], wa[b]);
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) { sre[k2 * GDG_W_RL + GDG_PAD(n1)] = ua[b][k2].x; sim[k2 * GDG_W_RL + GDG_PAD(n1)] = ua[b][k2].y; }
            }
        }
        __syncthreads();
        /* step B: wave w transforms region w; X[8 k1 + w] back into the region in natural k1 order */
        {
            const int tb = fresh(), wv = tb >> 6, ln = tb & 63;
            double *rre = sre + wv * GDG_W_RL, *rim = sim + wv * GDG_W_RL;
            cplx v[16];
            wave_fft1024<false>(v, rre, rim, tw, ln);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const int k1 = (ln + 64 * b) + 128 * t;
                    rre[GDG_PAD(k1)] = v[b * 8 + t].x;
                    rim[GDG_PAD(k1)] = v[b * 8 + t].y;
                }
        }
        __syncthreads();
        /* un-pack: X[k] and X[N-k] from Z[k], Z[N-k]; Z[k] = region[k & 7][k >> 3] */
        auto Z = [&](int k) { return make_double2(sre[(k & 7) * GDG_W_RL + GDG_PAD(k >> 3)], sim[(k & 7) * GDG_W_RL + GDG_PAD(k >> 3)]); };
        const int tu = fresh();
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int k = tu + T * i;
            if (k == 0) {
                cplx z0 = Z(0), zh = Z(N / 2);
                gstore(out, make_double2(z0.x + z0.y, z0.x - z0.y));
                gstore(out + N / 2, make_double2(zh.x, -zh.y));
            } else {
                const int n = N - k;
                cplx zk = Z(k), zn = Z(n);
                cplx A = make_double2(zk.x + zn.x, zk.y - zn.y);
                cplx Bv = make_double2(zk.x - zn.x, zk.y + zn.y);
                cplx cw = cmul(tw2[k], Bv);
                gstore(out + k, make_double2((A.x + cw.y) * 0.5, (A.y - cw.x) * 0.5));
                gstore(out + n, make_double2((A.x - cw.y) * 0.5, (-A.y - cw.x) * 0.5));
            }
        }
        __syncthreads();                                             /* the regions are rewritten by the next frame's step A */
    }
}

/* Y[b] = sum_k FDL[(pos - k) mod K][b] * H[k][b]; bin 0 is the (DC, Nyquist) pair of reals.
 * UNROLL partitions are loaded before any is used (2 * UNROLL * BPT 16-byte loads in flight per lane);
 * BPT adjacent bins per lane; NT: non-temporal loads (each spectrum is read exactly once per launch). */

template <bool NT>
__device__ __forceinline__ cplx mac_load(const cplx *p) {
    if constexpr (NT) {
        v2d v = __builtin_nontemporal_load(reinterpret_cast<const GDG_GLOBAL v2d *>((const GDG_GLOBAL void *)p));
        return make_double2(v.x, v.y);
    } else {
        return gload(p);
    }
}

template <int UNROLL, int BPT, bool NT, bool SWAP = false, bool HNT = NT>
__global__ void __launch_bounds__(1024)
fir_mac_kernel(const gdg_fir_chan *__restrict__ chans, int P, int k_lo) {
#pragma clang fp contract(off)      /* the sums of products as the reference forms them (no fused multiply-add): all three multiply-accumulate kernels give the same bits */
    gdg_fir_chan ch = chans[SWAP ? blockIdx.x : blockIdx.y];
    const int b0 = ((SWAP ? blockIdx.y : blockIdx.x) * blockDim.x + threadIdx.x) * BPT;
    if (b0 >= P) return;
    const int K = ch.K, R = ch.R;
    const int cur = (*ch.pos) % R;
    const cplx *__restrict__ fdl = ch.fdl + b0;
    const cplx *__restrict__ H = ch.H + b0;
    double ar[BPT], ai[BPT], br = 0.0, bi = 0.0;       /* complex product sums; component-wise sum for bin 0 */
#pragma unroll
    for (int q = 0; q < BPT; q++) { ar[q] = 0.0; ai[q] = 0.0; }
    /* THE ORDER OF THE SUM (all three multiply-accumulate kernels, and what makes their results bit-identical): k DESCENDING, K - 1 first,
     * the newest partition (k = 0, the frame that has just been transformed) LAST.  Every prefix of that order is computable before the
     * frame exists: k = K - 1 .. K_LO with K_LO = 1 is the "premac" launch that runs beside the previous call's last segment, and the
     * inverse kernel then only adds the k = 0 term (fir_inv_kernel, FUSED = 4) -- the same additions in the same order, the same bits. */
    int k = K - 1;
    /* U partitions' loads are issued before any is used; all of them non-temporal when NT: a spectrum is read once per launch, and a cached
     * read of one evicts what the kernels beside this one keep in L2 (the premac beside a small shard's segment: 96 channels 172 -> 148 us
     * per frame when its tail stopped reading through the cache, profiles/premac_loads_ab_r06.txt) */
    auto chunk = [&](auto u_tag) {
        constexpr int U = decltype(u_tag)::value;
        cplx x[U][BPT], h[U][BPT];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int slot = cur - (k - u);
            if (slot < 0) slot += R;
#pragma unroll
            for (int q = 0; q < BPT; q++) {
                x[u][q] = mac_load<NT>(fdl + (size_t)slot * P + q);
                h[u][q] = mac_load<HNT>(H + (size_t)(k - u) * P + q);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int q = 0; q < BPT; q++) {
                ar[q] += x[u][q].x * h[u][q].x - x[u][q].y * h[u][q].y;
                ai[q] += x[u][q].x * h[u][q].y + x[u][q].y * h[u][q].x;
            }
            br += x[u][0].x * h[u][0].x;
            bi += x[u][0].y * h[u][0].y;
        }
        k -= U;
    };
    constexpr int TAIL = UNROLL >= 6 ? 3 : (UNROLL >= 4 ? 2 : 1);
    while (k - (UNROLL - 1) >= k_lo) chunk(std::integral_constant<int, UNROLL>());
    if (k_lo > 0) {
        /* the premac runs BESIDE other kernels: what is left goes the same way */
        if constexpr (TAIL > 1) { while (k - (TAIL - 1) >= k_lo) chunk(std::integral_constant<int, TAIL>()); }
        while (k >= k_lo) chunk(std::integral_constant<int, 1>());
    }
    /* the whole sum on the call's own stream (few channels, short filters): alone on the chip, and a small shard's spectra are few enough to
     * still be in the last-level cache from the frame before -- through the cache (64 channels x 4 partitions, BASELINE config 3: 98.5 us per
     * frame against 101.2 with non-temporal loads here too, profiles/premac_loads_ab_r06.txt) */
    for (; k >= k_lo; k--) {
        int slot = cur - k;
        if (slot < 0) slot += R;
#pragma unroll
        for (int q = 0; q < BPT; q++) {
            cplx x = gload(fdl + (size_t)slot * P + q), h = gload(H + (size_t)k * P + q);
            ar[q] += x.x * h.x - x.y * h.y;
            ai[q] += x.x * h.y + x.y * h.x;
            if (q == 0) { br += x.x * h.x; bi += x.y * h.y; }
        }
    }
#pragma unroll
    for (int q = 0; q < BPT; q++) gstore(ch.Y + b0 + q, (b0 + q == 0) ? make_double2(br, bi) : make_double2(ar[q], ai[q]));
}

/* Inverse: packed half spectrum Y -> second half of the 2N-point real sequence -> clip -> dst */
/* first stage of the packed-real inverse: the spectrum pair (Y[k], Y[n]) -> Z[k], Z[n] in LDS (k = 0: n = N/2) */
template <int LOGN>
__device__ __forceinline__ void inv_head_values(int k, cplx yk, cplx yn, const cplx *__restrict__ tw2, cplx &zk, cplx &zn) {
    if (k == 0) {
        zk = make_double2(yk.x + yk.y, yk.x - yk.y);
        zn = make_double2(2.0 * yn.x, -2.0 * yn.y);
    } else {
        cplx A = make_double2(yk.x + yn.x, yk.y - yn.y);
        cplx Bv = make_double2(yk.x - yn.x, yk.y + yn.y);
        cplx w = tw2[k];
        w.y = -w.y;
        cplx O = cmul(Bv, w);
        zk = make_double2(A.x - O.y, A.y + O.x);
        zn = make_double2(A.x + O.y, -A.y + O.x);
    }
}
template <int LOGN>
__device__ __forceinline__ void inv_head_store(int k, cplx yk, cplx yn, double *sre, double *sim, const cplx *__restrict__ tw2) {
    constexpr int N = FftCfg<LOGN>::N;
    const int n = (k == 0) ? N / 2 : N - k;
    cplx zk, zn;
    inv_head_values<LOGN>(k, yk, yn, tw2, zk, zn);
    sre[GDG_PAD(k)] = zk.x;
    sim[GDG_PAD(k)] = zk.y;
    sre[GDG_PAD(n)] = zn.x;
    sim[GDG_PAD(n)] = zn.y;
}

/* The fused head: the workgroup walks ONE partition at a time through all its bins (partition-major), accumulating the
 * thread's eight bin pairs in registers.  Per partition it reads the delay-line slot and the IR partition front to back
 * (bins k ascending from 0, n descending from N - 1): two long sequential streams per workgroup.  The bin-major order (all K
 * partitions of one bin pair at once) had 32 interleaved streams per workgroup, 8192 on the chip, and lost 25 % of the
 * bandwidth to DRAM page conflicts (profiles/experiments/README.md). */
/* PRE: the terms k = K - 1 .. 1 are already summed in Y (the premac launch, fir_mac_kernel with k_lo = 1): the sums start from there and
 * only the newest partition is added -- two spectra per channel instead of 2 K on the critical path of a small shard. */
template <int LOGN, bool HNT, bool PRE = false>
__device__ __forceinline__ void mac_head(const gdg_fir_chan &ch, int cur, int tid, double *sre, double *sim, const cplx *__restrict__ tw2) {
#pragma clang fp contract(off)
    constexpr int N = FftCfg<LOGN>::N, T = FftCfg<LOGN>::T, ITER = (N / 2) / T;
    const int K = ch.K;
    double kr[ITER], ki[ITER], nr[ITER], ni[ITER], br = 0.0, bi = 0.0;
#pragma unroll
    for (int i = 0; i < ITER; i++) { kr[i] = 0.0; ki[i] = 0.0; nr[i] = 0.0; ni[i] = 0.0; }
    if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < ITER; i++) {
            const int k = tid + T * i, n = (k == 0) ? N / 2 : N - k;
            const cplx yk = gload(ch.Y + k), yn = gload(ch.Y + n);
            if (k == 0) { br = yk.x; bi = yk.y; } else { kr[i] = yk.x; ki[i] = yk.y; }
            nr[i] = yn.x; ni[i] = yn.y;
        }
    }
    for (int u = PRE ? 0 : K - 1; u >= 0; u--) {        /* k descending: the order of the sum (fir_mac_kernel) */
        int slot = cur - u;
        if (slot < 0) slot += ch.R;
        const cplx *__restrict__ x = ch.fdl + (size_t)slot * N;
        const cplx *__restrict__ h = ch.H + (size_t)u * N;
        cplx xk[ITER], hk[ITER], xn[ITER], hn[ITER];
#pragma unroll
        for (int i = 0; i < ITER; i++) {
            const int k = tid + T * i, n = (k == 0) ? N / 2 : N - k;
            xk[i] = mac_load<true>(x + k);
            hk[i] = mac_load<HNT>(h + k);
            xn[i] = mac_load<true>(x + n);
            hn[i] = mac_load<HNT>(h + n);
        }
        /* The waves of the workgroup meet HERE, with the partition's loads in flight: nobody runs more than one partition ahead,
         * and a wave that waits for the others waits with 32 loads outstanding.  Without the barrier the waves drift apart by up to
         * a fifth of the phase (cycle stamps, profiles/experiments/README.md: wave 0 spent 40 000 of the workgroup's 186 000 cycles
         * at the barrier in front of the transform while the late waves finished with a fraction of the CU's loads in flight):
         * 198 -> 190 us per launch. */
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < ITER; i++) {
            kr[i] += xk[i].x * hk[i].x - xk[i].y * hk[i].y;
            ki[i] += xk[i].x * hk[i].y + xk[i].y * hk[i].x;
            nr[i] += xn[i].x * hn[i].x - xn[i].y * hn[i].y;
            ni[i] += xn[i].x * hn[i].y + xn[i].y * hn[i].x;
        }
        br += xk[0].x * hk[0].x;                       /* bin 0 = (DC, Nyquist) as two reals: component-wise */
        bi += xk[0].y * hk[0].y;
    }
#pragma unroll
    for (int i = 0; i < ITER; i++) {
        const int k = tid + T * i;
        inv_head_store<LOGN>(k, (k == 0) ? make_double2(br, bi) : make_double2(kr[i], ki[i]), make_double2(nr[i], ni[i]), sre, sim, tw2);
    }
}

/* FUSED 4: Y holds the sum of the terms k = K - 1 .. 1 (fir_mac_kernel with k_lo = 1, launched ahead of the frame); the newest term is added here.
 * FUSED 0: Y comes from fir_mac_kernel.  FUSED 1 / 2: the multiply-accumulate runs here, straight into the inverse's
 * first stage (no Y round trip through HBM: the 6 % of extra bytes cost the separate MAC 20 % of its time, see
 * profiles/probes/); 2 = the IR spectra are shared between channels and read with cacheable loads. */
/* CHAIN (8192-point frames only): the NEXT unit of every channel is a power amp too (the benchmark chain: cabinet IR, then reverb IR).
 * The inverse's last pass leaves the clipped output frame in exactly the registers the next amp's forward transform (fir_fwd13w_kernel,
 * 8 x 1024) wants as its "current" half -- packed element n1 + 1024 m of the frame sits in thread n1 mod 512, slot (n1 div 512, 4 + m) on
 * both sides -- so that transform runs right here: the frame goes to the next amp's overlap-save history and, transformed together
 * with the previous one, into its delay line, without the round trip through the intermediate frame buffer and without a launch.
 * Same arithmetic in the same order as the stand-alone forward kernel: the delay line gets the same bits. */
template <int LOGN, int FUSED, bool CHAIN = false>
__global__ void __launch_bounds__(FftCfg<LOGN>::T)
fir_inv_kernel(const gdg_fir_chan *__restrict__ chans, int W, gdg_shift shift, const cplx *__restrict__ tw, const cplx *__restrict__ tw2,
               const gdg_fir_chan *__restrict__ next_chans = nullptr) {
    constexpr int N = FftCfg<LOGN>::N, T = FftCfg<LOGN>::T;
    static_assert(!CHAIN || (LOGN == 13 && FUSED != 3), "the chained forward transform is the 8 x 1024 one of the batch block size");
    constexpr int LDS_WORDS = (CHAIN && GDG_W_LDS > FftCfg<LOGN>::LDS) ? GDG_W_LDS : FftCfg<LOGN>::LDS;
    __shared__ double sre[LDS_WORDS];
    __shared__ double sim[LDS_WORDS];
    const int tid = threadIdx.x;
    /* FUSED 3: frame j of a window of W frames (blockIdx.x = channel * W + j): Y holds W spectra, dst W frames, the frame
     * counter stays (fir_tb_finish_kernel) */
    const int jw = (FUSED == 3) ? (int)(blockIdx.x % (unsigned)W) : 0;
    gdg_fir_chan ch = chans[(FUSED == 3) ? blockIdx.x / (unsigned)W : blockIdx.x];
    const cplx *__restrict__ Y = ch.Y + (size_t)jw * N;
    int cur = 0;
    if constexpr (FUSED == 1 || FUSED == 2 || FUSED == 4) cur = (*ch.pos) % ch.R;

    if constexpr (FUSED == 1) mac_head<LOGN, true>(ch, cur, tid, sre, sim, tw2);
    else if constexpr (FUSED == 2) mac_head<LOGN, false>(ch, cur, tid, sre, sim, tw2);
    else if constexpr (FUSED == 4) mac_head<LOGN, false, true>(ch, cur, tid, sre, sim, tw2);       /* Y holds the terms k >= 1 (premac) */
    else {
        constexpr int ITER = (N / 2) / T;
#pragma unroll
        for (int i = 0; i < ITER; i++) {
            const int k = tid + T * i, n = (k == 0) ? N / 2 : N - k;
            inv_head_store<LOGN>(k, gload(Y + k), gload(Y + n), sre, sim, tw2);
        }
    }
    __syncthreads();

    /* CHAIN: the next amp's previous frame and the radix-8 step's twiddles, requested now, consumed after the inverse's passes */
    cplx nx_prev[2][4], nx_w[2];
    cplx *nx_out = nullptr;
    double *nx_keep = nullptr;
    if constexpr (CHAIN) {
        const gdg_fir_chan nx = next_chans[blockIdx.x];
        const int pos2 = *nx.pos;
        const double *a2 = nx.prev + (size_t)((pos2 + 1) & 1) * N;
        nx_keep = nx.prev + (size_t)(pos2 & 1) * N;
        nx_out = nx.fdl + (size_t)(pos2 % nx.R) * N;
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int n1 = tid + T * b;
            nx_w[b] = tw[n1];
#pragma unroll
            for (int m = 0; m < 4; m++) nx_prev[b][m] = gload(reinterpret_cast<const cplx *>(a2 + 2 * (n1 + 1024 * m)));
        }
    }

    cplx v[16];
    constexpr int NP = sched_npass(LOGN);
    run_lds_passes<LOGN, 0, NP - 1, true>(v, sre, sim, tw, tid);

    /* last pass: its outputs n = j + t*N/R are already in natural order; only n >= N/2 is kept */
    constexpr int LR = sched_lr(LOGN, NP - 1), LNS = sched_lns(LOGN, NP - 1), R = 1 << LR, B = 16 / R;
    pass_load<LOGN, LR>(v, sre, sim, tid);
    pass_compute<LOGN, LR, LNS, true>(v, tw, tid);
    const int hop = ch.hop;
    double *__restrict__ dst = ch.dst + (size_t)jw * hop + ((ch.flags & GDG_DST_IS_OUTPUT) ? shift.out : 0);
    if constexpr (CHAIN) {
        static_assert(R == 8 && B == 2, "last pass of the 8192-point inverse: radix 8, two butterflies per thread");
        /* frame -> next amp's history; [previous | frame] -> radix-8 step of its forward transform (fir_fwd13w_kernel, step A) */
        cplx ua[2][8];
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int n1 = tid + T * b;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                cplx z = v[b * R + 4 + m];
                z.x = fmin(1.0, fmax(-1.0, z.x));           /* filter/filter.go:487-493 */
                z.y = fmin(1.0, fmax(-1.0, z.y));
                ua[b][m] = nx_prev[b][m];
                ua[b][4 + m] = z;
                gstore(reinterpret_cast<cplx *>(nx_keep + 2 * (n1 + 1024 * m)), z);
                if (!(ch.flags & GDG_DST_UNUSED)) gstore(reinterpret_cast<cplx *>(dst + 2 * (n1 + 1024 * m)), z);
            }
        }
        __syncthreads();                                    /* the inverse's last pass has been read by everybody */
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int n1 = tid + T * b;
            Dft<8, false>::run(ua[b]);
            twiddle_powers8(ua[b], nx_w[b]);
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) { sre[k2 * GDG_W_RL + GDG_PAD(n1)] = ua[b][k2].x; sim[k2 * GDG_W_RL + GDG_PAD(n1)] = ua[b][k2].y; }
        }
        __syncthreads();
        {
            const int wave = tid >> 6, lane = tid & 63;
            double *rre = sre + wave * GDG_W_RL, *rim = sim + wave * GDG_W_RL;
            cplx f[16];
            wave_fft1024<false>(f, rre, rim, tw, lane);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const int k1 = (lane + 64 * b) + 128 * t;
                    rre[GDG_PAD(k1)] = f[b * 8 + t].x;
                    rim[GDG_PAD(k1)] = f[b * 8 + t].y;
                }
        }
        __syncthreads();
        auto Z = [&](int k) { return make_double2(sre[(k & 7) * GDG_W_RL + GDG_PAD(k >> 3)], sim[(k & 7) * GDG_W_RL + GDG_PAD(k >> 3)]); };
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int k = tid + T * i;
            if (k == 0) {
                cplx z0 = Z(0), zh = Z(N / 2);
                gstore(nx_out, make_double2(z0.x + z0.y, z0.x - z0.y));
                gstore(nx_out + N / 2, make_double2(zh.x, -zh.y));
            } else {
                const int n = N - k;
                cplx zk = Z(k), zn = Z(n);
                cplx A = make_double2(zk.x + zn.x, zk.y - zn.y);
                cplx Bv = make_double2(zk.x - zn.x, zk.y + zn.y);
                cplx cw = cmul(tw2[k], Bv);
                gstore(nx_out + k, make_double2((A.x + cw.y) * 0.5, (A.y - cw.x) * 0.5));
                gstore(nx_out + n, make_double2((A.x - cw.y) * 0.5, (-A.y - cw.x) * 0.5));
            }
        }
    } else if (hop == N) {
#pragma unroll
        for (int b = 0; b < B; b++) {
            int j = tid + T * b;
#pragma unroll
            for (int t = R / 2; t < R; t++) {
                int n = j + t * (N / R);
                cplx z = v[b * R + t];
                /* filter/filter.go:487-493: the emitted samples are clipped to [-1, 1] */
                z.x = fmin(1.0, fmax(-1.0, z.x));
                z.y = fmin(1.0, fmax(-1.0, z.y));
                gstore(reinterpret_cast<cplx *>(dst + 2 * (n - N / 2)), z);
            }
        }
    } else {
        /* frame shorter than the transform half: the valid outputs are the real samples [hop, 2 hop) */
#pragma unroll
        for (int b = 0; b < B; b++) {
            int j = tid + T * b;
#pragma unroll
            for (int t = 0; t < R; t++) {
                int n = j + t * (N / R);
                cplx z = v[b * R + t];
                int i0 = 2 * n - hop;
                if (i0 >= 0 && i0 < hop) gstore1(dst + i0, fmin(1.0, fmax(-1.0, z.x)));
                if (i0 + 1 >= 0 && i0 + 1 < hop) gstore1(dst + i0 + 1, fmin(1.0, fmax(-1.0, z.y)));
            }
        }
    }
    if (FUSED != 3 && tid == 0) {
        int pos = *ch.pos + 1;
        int wrap = 2 * ch.R;
        *ch.pos = (pos >= wrap) ? pos - wrap : pos;
    }
}

/* fir_inv_kernel<13, 3> (frame j of a window, Y from the time-blocked multiply-accumulate) with one LDS buffer (see exchange1): two
 * workgroups per CU, 128 VGPRs */
__global__ void __launch_bounds__(512, 4)
fir_inv13h_kernel(const gdg_fir_chan *__restrict__ chans, int W, gdg_shift shift, const cplx *__restrict__ tw, const cplx *__restrict__ tw2) {
    constexpr int LOGN = 13, N = 8192, T = 512, ITER = (N / 2) / T;
    __shared__ double s[FftCfg<LOGN>::LDS];
    const int tid = threadIdx.x;
    const int jw = (int)(blockIdx.x % (unsigned)W);
    const gdg_fir_chan ch = chans[blockIdx.x / (unsigned)W];
    const cplx *__restrict__ Y = ch.Y + (size_t)jw * N;
    cplx zk[ITER], zn[ITER];
    {
        cplx yk[ITER], yn[ITER];
#pragma unroll
        for (int i = 0; i < ITER; i++) {
            const int k = tid + T * i, n = (k == 0) ? N / 2 : N - k;
            yk[i] = gload(Y + k);
            yn[i] = gload(Y + n);
        }
#pragma unroll
        for (int i = 0; i < ITER; i++) inv_head_values<LOGN>(tid + T * i, yk[i], yn[i], tw2, zk[i], zn[i]);
    }
    cplx v[16];
#pragma unroll
    for (int i = 0; i < ITER; i++) {
        const int k = tid + T * i, n = (k == 0) ? N / 2 : N - k;
        s[GDG_PAD(k)] = zk[i].x;
        s[GDG_PAD(n)] = zn[i].x;
    }
    __syncthreads();
    pass_load1<LOGN, 4, 0>(v, s, tid);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITER; i++) {
        const int k = tid + T * i, n = (k == 0) ? N / 2 : N - k;
        s[GDG_PAD(k)] = zk[i].y;
        s[GDG_PAD(n)] = zn[i].y;
    }
    __syncthreads();
    pass_load1<LOGN, 4, 1>(v, s, tid);
    static_assert(sched_npass(LOGN) == 4 && sched_lr(LOGN, 0) == 4 && sched_lr(LOGN, 3) == 3, "8192 = 16 x 8 x 8 x 8");
    pass_compute<LOGN, 4, 0, true>(v, tw, tid);
    exchange1<LOGN, 0, false>(v, s, tid);
    pass_compute<LOGN, 3, 4, true>(v, tw, tid);
    exchange1<LOGN, 1, false>(v, s, tid);
    pass_compute<LOGN, 3, 7, true>(v, tw, tid);
    exchange1<LOGN, 2, false>(v, s, tid);
    pass_compute<LOGN, 3, 10, true>(v, tw, tid);
    /* outputs n = j + t N/8 in natural order; only n >= N/2 is kept (frames of 8192 samples: hop == N) */
    double *__restrict__ dst = ch.dst + (size_t)jw * N + ((ch.flags & GDG_DST_IS_OUTPUT) ? shift.out : 0);
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int j = tid + T * b;
#pragma unroll
        for (int t = 4; t < 8; t++) {
            const int n = j + t * (N / 8);
            cplx z = v[b * 8 + t];
            z.x = fmin(1.0, fmax(-1.0, z.x));               /* filter/filter.go:487-493 */
            z.y = fmin(1.0, fmax(-1.0, z.y));
            gstore(reinterpret_cast<cplx *>(dst + 2 * (n - N / 2)), z);
        }
    }
}

/* Raw inverse (frame size change with live convolution state): one delay-line slot -- the packed half spectrum of
 * [x_{t-1} | x_t | zeros] -- back to its two frames, unclipped.  The forward transform is unscaled and this inverse is too, so
 * scale = 1 / (2N) restores the samples. */
template <int LOGN>
__global__ void __launch_bounds__(FftCfg<LOGN>::T)
fir_raw_inv_kernel(const gdg_fir_rawjob *__restrict__ jobs, double scale, const cplx *__restrict__ tw, const cplx *__restrict__ tw2) {
    constexpr int N = FftCfg<LOGN>::N, T = FftCfg<LOGN>::T;
    __shared__ double sre[FftCfg<LOGN>::LDS];
    __shared__ double sim[FftCfg<LOGN>::LDS];
    const int tid = threadIdx.x;
    const gdg_fir_rawjob jb = jobs[blockIdx.x];
    constexpr int ITER = (N / 2) / T;
#pragma unroll
    for (int i = 0; i < ITER; i++) {
        const int k = tid + T * i, n = (k == 0) ? N / 2 : N - k;
        inv_head_store<LOGN>(k, gload(jb.Y + k), gload(jb.Y + n), sre, sim, tw2);
    }
    __syncthreads();
    cplx v[16];
    constexpr int NP = sched_npass(LOGN);
    run_lds_passes<LOGN, 0, NP - 1, true>(v, sre, sim, tw, tid);
    constexpr int LR = sched_lr(LOGN, NP - 1), LNS = sched_lns(LOGN, NP - 1), R = 1 << LR, B = 16 / R;
    pass_load<LOGN, LR>(v, sre, sim, tid);
    pass_compute<LOGN, LR, LNS, true>(v, tw, tid);
    const int hop = jb.hop;
#pragma unroll
    for (int b = 0; b < B; b++) {
        const int j = tid + T * b;
#pragma unroll
        for (int t = 0; t < R; t++) {
            const int n = j + t * (N / R);
            const cplx z = v[b * R + t];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int i = 2 * n + h;                    /* real sample i of [first (hop) | second (hop) | zeros] */
                const double x = (h ? z.y : z.x) * scale;
                if (i < hop) { if (jb.first) gstore1(jb.first + i, x); }
                else if (i < 2 * hop) gstore1(jb.second + (i - hop), x);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Time blocking.  A caller that holds W consecutive frames of every channel (the batch run: whole files live in HBM) gets the W
 * outputs  Y_j = sum_k H[k] X[t0 + j - k],  j = 0 .. W - 1,  from ONE pass over the IR spectra and one over the K + W - 1 delay
 * line slots they touch: (2 K + W - 1) spectrum reads for W frames instead of 2 K W.  The ring has R >= K + W - 1 slots so the
 * W new spectra do not overwrite what the early frames of the window still need.
 *
 * One thread per bin.  Partitions are walked in chunks of C (= W up to 8, 8 for W = 16): a chunk needs C IR partitions and the
 * W + C - 1 delay-line slots they meet, all loaded before the C x W multiply-adds, whose indices are compile-time constants.
 * Every Y_j accumulates its terms in DESCENDING k -- the order of the per-frame kernels (fir_mac_kernel) -- and, like them, without fused
 * multiply-adds (`fp contract(off)` in all three multiply-accumulate kernels): the results are bit-identical to W single-frame calls.
 * ---------------------------------------------------------------------------------------------- */
template <int W, int C, bool HNT>
__global__ void __launch_bounds__(256)
fir_mac_tb_kernel(const gdg_fir_chan *__restrict__ chans, int P) {
#pragma clang fp contract(off)
    gdg_fir_chan ch = chans[blockIdx.y];
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= P) return;
    const int K = ch.K, R = ch.R;
    const int pos0 = (*ch.pos) % R;                       /* slot of X[t0] (frame 0 of the window) */
    const cplx *__restrict__ fdl = ch.fdl + b;
    const cplx *__restrict__ H = ch.H + b;
    double ar[W], ai[W];
#pragma unroll
    for (int j = 0; j < W; j++) { ar[j] = 0.0; ai[j] = 0.0; }
    /* partitions in chunks of C: chunk c0 needs H[c0 .. c0 + C - 1] and the W + C - 1 slots X[t0 - c0 + d - (C - 1)], d = 0 .. W + C - 2 */
    for (int c0 = ((K - 1) / C) * C; c0 >= 0; c0 -= C) {      /* k descending: the order of the sum (fir_mac_kernel) */
        cplx x[W + C - 1], h[C];
#pragma unroll
        for (int d = 0; d < W + C - 1; d++) {
            int slot = (pos0 + (d - (C - 1)) - c0) % R;      /* the remainder has the argument's sign: one correction */
            if (slot < 0) slot += R;
            x[d] = mac_load<true>(fdl + (size_t)slot * P);
        }
#pragma unroll
        for (int i = 0; i < C; i++) {
            const int k = c0 + i;
            h[i] = (k < K) ? mac_load<HNT>(H + (size_t)k * P) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int i = C - 1; i >= 0; i--) {
            if (c0 + i < K) {
#pragma unroll
                for (int j = 0; j < W; j++) {
                    const cplx xv = x[j - i + C - 1];
                    ar[j] += xv.x * h[i].x - xv.y * h[i].y;
                    ai[j] += xv.x * h[i].y + xv.y * h[i].x;
                }
            }
        }
    }
    if (b != 0) {
#pragma unroll
        for (int j = 0; j < W; j++) gstore_nt(ch.Y + (size_t)j * P + b, make_double2(ar[j], ai[j]));
    }
    /* bin 0 = (DC, Nyquist) as two reals: component-wise products, the same descending order.  Lane j of the channel's first workgroup
     * takes frame j: its operands were loaded a moment ago (cache), eight partitions' worth are issued before they are consumed */
    if (blockIdx.x == 0 && threadIdx.x < W) {
        const int j = threadIdx.x;
        double br = 0.0, bi = 0.0;
        for (int k0 = ((K - 1) / 8) * 8; k0 >= 0; k0 -= 8) {
            cplx xv[8], hv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int k = min(k0 + u, K - 1);
                int slot = (pos0 + j - k) % R;
                if (slot < 0) slot += R;
                xv[u] = gload(ch.fdl + (size_t)slot * P);
                hv[u] = gload(ch.H + (size_t)k * P);
            }
#pragma unroll
            for (int u = 7; u >= 0; u--) {
                if (k0 + u < K) {
                    br += xv[u].x * hv[u].x;
                    bi += xv[u].y * hv[u].y;
                }
            }
        }
        gstore(ch.Y + (size_t)j * P, make_double2(br, bi));
    }
}

/* Time blocking + adjacent power amps: one workgroup per CHANNEL walks the W frames of the window; per frame the inverse transform of
 * amp 1 (product spectrum Y_j -> clipped frame) runs straight into the forward transform of amp 2 ([previous | frame] -> delay-line
 * slot), the frame staying in registers as in fir_inv_kernel<.., CHAIN> and the previous frame as in fir_fwd13w_chan_kernel.  Against
 * the two separate launches (inverse: 128 KiB in, 64 out per channel-frame; forward: 64 in, 128 out) the frame's round trip through
 * HBM is gone: 256 KiB instead of 384.  Only the LAST frame of the window is also written to amp 1's output buffer: amp 2's
 * fir_tb_finish_kernel takes its overlap-save history from there.  Same arithmetic in the same order as the separate kernels. */
__global__ void __launch_bounds__(512)
fir_inv_fwd_chain_chan_kernel(const gdg_fir_chan *__restrict__ chans, const gdg_fir_chan *__restrict__ next_chans, int W, gdg_shift shift,
                              const cplx *__restrict__ tw, const cplx *__restrict__ tw2) {
    constexpr int LOGN = 13, N = 8192, T = 512;
    constexpr int LDS_WORDS = GDG_W_LDS > FftCfg<LOGN>::LDS ? GDG_W_LDS : FftCfg<LOGN>::LDS;
    __shared__ double sre[LDS_WORDS];
    __shared__ double sim[LDS_WORDS];
    const int tid = threadIdx.x;
    const gdg_fir_chan ch = chans[blockIdx.x], nx = next_chans[blockIdx.x];
    const int pos2 = *nx.pos;
    double *dst_last = ch.dst + (size_t)(W - 1) * N + ((ch.flags & GDG_DST_IS_OUTPUT) ? shift.out : 0);
    /* hides the thread index from the optimiser (see fir_fwd13w_chan_kernel): otherwise the loop body's addresses are hoisted and spill */
    auto fresh = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
    cplx wa[2], pv[2][4];
    {
        const double *a2 = nx.prev + (size_t)((pos2 + 1) & 1) * N;
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int n1 = tid + T * b;
            wa[b] = tw[n1];
#pragma unroll
            for (int m = 0; m < 4; m++) pv[b][m] = gload(reinterpret_cast<const cplx *>(a2 + 2 * (n1 + 1024 * m)));
        }
    }
    for (int jw = 0; jw < W; jw++) {
        const cplx *__restrict__ Y = ch.Y + (size_t)jw * N;
        cplx *nx_out = nx.fdl + (size_t)((pos2 + jw) % nx.R) * N;
        /* inverse of amp 1 (fir_inv_kernel<13, 3>) */
        {
            const int t0 = fresh();
            constexpr int ITER = (N / 2) / T;
            cplx yk[ITER], yn[ITER];
#pragma unroll
            for (int i = 0; i < ITER; i++) {
                const int k = t0 + T * i, n = (k == 0) ? N / 2 : N - k;
                yk[i] = gload(Y + k);
                yn[i] = gload(Y + n);
            }
#pragma unroll
            for (int i = 0; i < ITER; i++) inv_head_store<LOGN>(t0 + T * i, yk[i], yn[i], sre, sim, tw2);
        }
        __syncthreads();
        cplx ua[2][8];
        {
            const int t1 = fresh();
            cplx v[16];
            constexpr int NP = sched_npass(LOGN);
            run_lds_passes<LOGN, 0, NP - 1, true>(v, sre, sim, tw, t1);
            constexpr int LR = sched_lr(LOGN, NP - 1), LNS = sched_lns(LOGN, NP - 1), R = 1 << LR;
            static_assert(R == 8, "last pass of the 8192-point inverse: radix 8");
            pass_load<LOGN, LR>(v, sre, sim, t1);
            pass_compute<LOGN, LR, LNS, true>(v, tw, t1);
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const int n1 = t1 + T * b;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    cplx z = v[b * R + 4 + m];
                    z.x = fmin(1.0, fmax(-1.0, z.x));           /* filter/filter.go:487-493 */
                    z.y = fmin(1.0, fmax(-1.0, z.y));
                    ua[b][m] = pv[b][m];
                    ua[b][4 + m] = z;
                    pv[b][m] = z;
                    if (jw == W - 1 || !(ch.flags & GDG_DST_UNUSED))
                        gstore(reinterpret_cast<cplx *>((jw == W - 1 ? dst_last : dst_last - (size_t)(W - 1 - jw) * N) + 2 * (n1 + 1024 * m)), z);
                }
            }
        }
        __syncthreads();                                        /* the inverse's last pass has been read by everybody */
        /* forward of amp 2 (fir_fwd13w_chan_kernel) */
        {
            const int ta = fresh();
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const int n1 = ta + T * b;
                Dft<8, false>::run(ua[b]);
                twiddle_powers8(ua[b], wa[b]);
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) { sre[k2 * GDG_W_RL + GDG_PAD(n1)] = ua[b][k2].x; sim[k2 * GDG_W_RL + GDG_PAD(n1)] = ua[b][k2].y; }
            }
        }
        __syncthreads();
        {
            const int tb = fresh(), wv = tb >> 6, ln = tb & 63;
            double *rre = sre + wv * GDG_W_RL, *rim = sim + wv * GDG_W_RL;
            cplx f[16];
            wave_fft1024<false>(f, rre, rim, tw, ln);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const int k1 = (ln + 64 * b) + 128 * t;
                    rre[GDG_PAD(k1)] = f[b * 8 + t].x;
                    rim[GDG_PAD(k1)] = f[b * 8 + t].y;
                }
        }
        __syncthreads();
        auto Z = [&](int k) { return make_double2(sre[(k & 7) * GDG_W_RL + GDG_PAD(k >> 3)], sim[(k & 7) * GDG_W_RL + GDG_PAD(k >> 3)]); };
        const int tu = fresh();
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int k = tu + T * i;
            if (k == 0) {
                cplx z0 = Z(0), zh = Z(N / 2);
                gstore(nx_out, make_double2(z0.x + z0.y, z0.x - z0.y));
                gstore(nx_out + N / 2, make_double2(zh.x, -zh.y));
            } else {
                const int n = N - k;
                cplx zk = Z(k), zn = Z(n);
                cplx A = make_double2(zk.x + zn.x, zk.y - zn.y);
                cplx Bv = make_double2(zk.x - zn.x, zk.y + zn.y);
                cplx cw = cmul(tw2[k], Bv);
                gstore(nx_out + k, make_double2((A.x + cw.y) * 0.5, (A.y - cw.x) * 0.5));
                gstore(nx_out + n, make_double2((A.x - cw.y) * 0.5, (-A.y - cw.x) * 0.5));
            }
        }
        __syncthreads();                                        /* the regions are rewritten by the next frame's inverse head */
    }
}

/* after a window of W frames: the last frame becomes the overlap-save history of the next call, the frame counter moves on */
__global__ void __launch_bounds__(256)
fir_tb_finish_kernel(const gdg_fir_chan *__restrict__ chans, int W, gdg_shift shift) {
    gdg_fir_chan ch = chans[blockIdx.x];
    if (ch.flags & GDG_SRC_IS_INPUT) ch.src += shift.in;
    const int N = ch.hop;
    const int pos = *ch.pos;
    __syncthreads();
    const cplx *src = reinterpret_cast<const cplx *>(ch.src + (size_t)(W - 1) * N);
    cplx *dst = reinterpret_cast<cplx *>(ch.prev + (size_t)((pos + W + 1) & 1) * N);
    for (int i = threadIdx.x; i < N / 2; i += 256) gstore(dst + i, gload(src + i));
    if (threadIdx.x == 0) *ch.pos = (pos + W) % (2 * ch.R);
}

/* ---- host side ------------------------------------------------------------------------------- */

/* ------------------------------------------------------------------------------------------------
 * Small transforms, P = 1 ... 32 packed points (n = 2 ... 64 real samples): below the FIR path's smallest frame, here so that the
 * stand-alone gdg_fft_real / gdg_fft_real_inverse cover the sizes of the reference's own known answers (fft/fft_test.go:237-271:
 * eight points).  Same algorithm as the large kernels -- packed-real transform: P complex points z[e] = (r[2e], r[2e+1]), one
 * complex FFT, the un-packing / re-packing stage with tw2[k] = exp(-i pi k / P) -- as a plain radix-2 decimation-in-time in LDS,
 * one wave per job, one butterfly per lane and stage.
 * ---------------------------------------------------------------------------------------------- */
template <bool INV>
__device__ __forceinline__ void small_fft(cplx *z, int P, int logp, const cplx *__restrict__ tw, int lane) {
    /* z holds the input in bit-reversed order; stage s joins transforms of length 2^s */
    for (int s = 0; s < logp; s++) {
        const int half = 1 << s;
        if (lane < P / 2) {
            const int j = lane & (half - 1), base = ((lane >> s) << (s + 1)) + j;
            cplx w = tw[j * (P >> (s + 1))];                  /* exp(-2 pi i j / 2^(s+1)) */
            if (INV) w.y = -w.y;
            const cplx a = z[base], b = cmul(z[base + half], w);
            z[base] = cadd(a, b);
            z[base + half] = csub(a, b);
        }
        __builtin_amdgcn_wave_barrier();
    }
}
__device__ __forceinline__ int bitrev(int v, int bits) { return bits ? (int)(__brev((unsigned)v) >> (32 - bits)) : 0; }

__global__ void __launch_bounds__(64)
fft_small_fwd_kernel(const gdg_fir_irjob *__restrict__ jobs, int P, int logp, double scale, const cplx *__restrict__ tw, const cplx *__restrict__ tw2) {
    __shared__ cplx z[32];
    const int lane = threadIdx.x;
    const gdg_fir_irjob jb = jobs[blockIdx.x];
    if (lane < P) {
        /* r = [a (P) | b (P)] (hop == P) or [a (P) | zeros] (hop == 0) */
        double r2[2];
        for (int h = 0; h < 2; h++) {
            const int i = 2 * lane + h;
            r2[h] = (i < P) ? gload1(jb.a + i) : ((jb.hop > 0 && jb.b) ? gload1(jb.b + (i - P)) : 0.0);
        }
        z[bitrev(lane, logp)] = make_double2(r2[0], r2[1]);
    }
    __builtin_amdgcn_wave_barrier();
    small_fft<false>(z, P, logp, tw, lane);
    if (lane <= P / 2) {
        const int k = lane;
        if (k == 0) {
            gstore(jb.out, make_double2((z[0].x + z[0].y) * scale, (z[0].x - z[0].y) * scale));
        } else {
            const int n = P - k;
            const cplx zk = z[k], zn = z[n];
            const cplx A = make_double2(zk.x + zn.x, zk.y - zn.y), Bv = make_double2(zk.x - zn.x, zk.y + zn.y);
            const cplx cw = cmul(tw2[k], Bv);
            const double hs = 0.5 * scale;
            gstore(jb.out + k, make_double2((A.x + cw.y) * hs, (A.y - cw.x) * hs));
            if (n != k) gstore(jb.out + n, make_double2((A.x - cw.y) * hs, (-A.y - cw.x) * hs));
        }
    }
}

__global__ void __launch_bounds__(64)
fft_small_inv_kernel(const gdg_fir_rawjob *__restrict__ jobs, int P, int logp, double scale, const cplx *__restrict__ tw, const cplx *__restrict__ tw2) {
    __shared__ cplx z[32];
    const int lane = threadIdx.x;
    const gdg_fir_rawjob jb = jobs[blockIdx.x];
    if (lane <= P / 2) {
        const int k = lane;
        if (k == 0) {
            const cplx y0 = gload(jb.Y);
            z[0] = make_double2(y0.x + y0.y, y0.x - y0.y);
        } else {
            const int n = P - k;
            const cplx yk = gload(jb.Y + k), yn = gload(jb.Y + n);
            const cplx A = make_double2(yk.x + yn.x, yk.y - yn.y), Bv = make_double2(yk.x - yn.x, yk.y + yn.y);
            cplx w = tw2[k];
            w.y = -w.y;
            const cplx O = cmul(Bv, w);
            z[bitrev(k, logp)] = make_double2(A.x - O.y, A.y + O.x);
            if (n != k) z[bitrev(n, logp)] = make_double2(A.x + O.y, -A.y + O.x);
        }
    }
    __builtin_amdgcn_wave_barrier();
    small_fft<true>(z, P, logp, tw, lane);
    if (lane < P) {
        for (int h = 0; h < 2; h++) {
            const int i = 2 * lane + h;
            const double x = (h ? z[lane].y : z[lane].x) * scale;
            if (i < jb.hop) { if (jb.first) gstore1(jb.first + i, x); }
            else if (i < 2 * jb.hop) gstore1(jb.second + (i - jb.hop), x);
        }
    }
}

static int ilog2_exact(int P) {
    int l = 0;
    while ((1 << l) < P) l++;
    return ((1 << l) == P) ? l : -1;
}

/* twiddle tables: tw[m] = exp(-2 pi i m / P), m < P;  tw2[k] = exp(-i pi k / P), k <= P/2 */
hipError_t gdg_fir_tables_create(int P, cplx **d_tw, cplx **d_tw2) {
    std::vector<cplx> tw((size_t)P), tw2((size_t)P / 2 + 1);
    const long double pi = 3.141592653589793238462643383279502884L;
    for (int m = 0; m < P; m++) {
        long double ang = -2.0L * pi * (long double)m / (long double)P;
        tw[m] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    for (int k = 0; k <= P / 2; k++) {
        long double ang = -pi * (long double)k / (long double)P;
        tw2[k] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    hipError_t e = hipMalloc((void **)d_tw, sizeof(cplx) * tw.size());
    if (e != hipSuccess) return e;
    e = hipMalloc((void **)d_tw2, sizeof(cplx) * tw2.size());
    if (e != hipSuccess) return e;
    e = hipMemcpy(*d_tw, tw.data(), sizeof(cplx) * tw.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    return hipMemcpy(*d_tw2, tw2.data(), sizeof(cplx) * tw2.size(), hipMemcpyHostToDevice);
}

#define GDG_DISPATCH_LOGN(L, STMT)                                                                  \
    switch (L) {                                                                                    \
    case 6: { constexpr int LG = 6; STMT; break; }                                                  \
    case 7: { constexpr int LG = 7; STMT; break; }                                                  \
    case 8: { constexpr int LG = 8; STMT; break; }                                                  \
    case 9: { constexpr int LG = 9; STMT; break; }                                                  \
    case 10: { constexpr int LG = 10; STMT; break; }                                                \
    case 11: { constexpr int LG = 11; STMT; break; }                                                \
    case 12: { constexpr int LG = 12; STMT; break; }                                                \
    case 13: { constexpr int LG = 13; STMT; break; }                                                \
    default: return hipErrorInvalidValue;                                                           \
    }

template <int LG> static void launch_fwd(const gdg_fir_chan *d_chans, int n, const cplx *tw, const cplx *tw2, gdg_shift shift, hipStream_t s) {
    fir_fwd_kernel<LG, false><<<dim3(n), dim3(FftCfg<LG>::T), 0, s>>>(d_chans, nullptr, 1.0, shift, tw, tw2);
}
template <int LG> static void launch_ir(const gdg_fir_irjob *d_jobs, int n, double scale, const cplx *tw, const cplx *tw2, hipStream_t s) {
    fir_fwd_kernel<LG, true><<<dim3(n), dim3(FftCfg<LG>::T), 0, s>>>(nullptr, d_jobs, scale, gdg_shift{ 0, 0 }, tw, tw2);
}
template <int LG> static void launch_raw_inv(const gdg_fir_rawjob *d_jobs, int n, double scale, const cplx *tw, const cplx *tw2, hipStream_t s) {
    fir_raw_inv_kernel<LG><<<dim3(n), dim3(FftCfg<LG>::T), 0, s>>>(d_jobs, scale, tw, tw2);
}
/* e0 / e1: HIP events that take the KERNEL's own begin and end timestamps (hipExtLaunchKernelGGL) -- what rocprofv3 reports as its duration.
 * A pair of hipEventRecord calls around the launch measures 4-5 us more (two marker packets, the dispatch latency between them) and keeps the
 * kernel from overlapping its neighbours' ramp-up and tail. */
#define GDG_LAUNCH_INV(KERNEL, ...)                                                                                                   \
    do {                                                                                                                              \
        if (e0) hipExtLaunchKernelGGL((KERNEL), dim3(n), dim3(FftCfg<LG>::T), 0, s, e0, e1, 0, __VA_ARGS__);                           \
        else hipLaunchKernelGGL((KERNEL), dim3(n), dim3(FftCfg<LG>::T), 0, s, __VA_ARGS__);                                           \
    } while (0)
template <int LG> static void launch_inv(const gdg_fir_chan *d_chans, int n, const cplx *tw, const cplx *tw2, int fused, gdg_shift shift, hipStream_t s,
                                         const gdg_fir_chan *d_next, hipEvent_t e0, hipEvent_t e1) {
    const gdg_fir_chan *none = nullptr;
    if constexpr (LG == 13) {
        if (d_next) {       /* the next unit of every channel is a power amp: its forward transform rides along */
            if (fused == 0) GDG_LAUNCH_INV((fir_inv_kernel<13, 0, true>), d_chans, 1, shift, tw, tw2, d_next);
            else if (fused == 4) GDG_LAUNCH_INV((fir_inv_kernel<13, 4, true>), d_chans, 1, shift, tw, tw2, d_next);
            else if (fused == 1) GDG_LAUNCH_INV((fir_inv_kernel<13, 1, true>), d_chans, 1, shift, tw, tw2, d_next);
            else GDG_LAUNCH_INV((fir_inv_kernel<13, 2, true>), d_chans, 1, shift, tw, tw2, d_next);
            return;
        }
    }
    if (fused == 0) GDG_LAUNCH_INV((fir_inv_kernel<LG, 0>), d_chans, 1, shift, tw, tw2, none);
    else if (fused == 4) { if constexpr (LG == 13) GDG_LAUNCH_INV((fir_inv_kernel<13, 4>), d_chans, 1, shift, tw, tw2, none); }
    else if (fused == 1) GDG_LAUNCH_INV((fir_inv_kernel<LG, 1>), d_chans, 1, shift, tw, tw2, none);
    else GDG_LAUNCH_INV((fir_inv_kernel<LG, 2>), d_chans, 1, shift, tw, tw2, none);
}
#undef GDG_LAUNCH_INV

static int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

/* Which 8192-point transforms exchange through ONE LDS buffer, two workgroups per CU (env GDG_FFT_HALF_LDS, bits: 1 the window's forward,
 * 2 the window's inverse, 4 the per-frame forward, 8 no chained inverse -> forward kernel in windows of four frames or more, 32 the window's
 * forward workgroups XCD by XCD).  Default 14, measured at W = 16 with 512 channels, us per frame (profiles/fft_half_lds_r04.txt):
 *   inverse 32 -> 23: it reads 128 KiB and writes 64 per channel-frame, the second workgroup streams while the first computes;
 *   chained inverse -> forward kernel 54.5 -> two launches 23 + 24: the chained kernel needs both LDS buffers and the previous frame in registers;
 *   forward: the walk (one workgroup per channel, previous frame in registers) 24, one buffer and one workgroup per frame 22 (the previous
 *   frame is read again, from the L2 of the XCD that has just read it as "current") -- faster alone, but two such workgroups take a CU's
 *   whole register file, and with the two channel groups of the batch path the walk (one 140 KiB workgroup, half the registers) shares its
 *   CU with the other group's multiply-accumulate: 272-278 against 279-280 us per frame for the window as a whole.  The walk stays (bits 1, 32 off).
 *   per-frame forward of the real-time path (bit 4, on): 38.4 us alone for its 165 MB; with one buffer the step is 7-10 us shorter with one
 *   channel group (593 -> 583-586 us) and a wash to +0.4 % with two (profiles/fft_half_lds_groups_r04.txt).
 * Not kept: the per-channel walk with one buffer (26 us: 512 workgroups, nothing to balance with), non-temporal spectrum stores and product
 * loads (no difference). */
/* Launch shapes of the transforms / the tuner that the launchers (which have no context) choose between: process-wide values, set through
 * gdg_ctx_set_option (api_ctx.cpp: the keys, what they mean, their ranges); first use takes the environment variable of the same name as a debug
 * override, else the measured default. */
static int g_knob[GDG_KNOB_COUNT];
static bool g_knob_set[GDG_KNOB_COUNT];
int gdg_knob_get(int which) {
    static const struct { const char *env; int def; } K[GDG_KNOB_COUNT] = {
        { "GDG_FFT_HALF_LDS", 14 }, { "GDG_FWD_PER_CHANNEL", 1 }, { "GDG_WAVE_FFT", 1 }, { "GDG_MAC_VARIANT", 0 }, { "GDG_TUNER_PARTS", 0 },
    };
    if (which < 0 || which >= GDG_KNOB_COUNT) return 0;
    if (!g_knob_set[which]) { const char *e = getenv(K[which].env); g_knob[which] = e ? atoi(e) : K[which].def; g_knob_set[which] = true; }
    return g_knob[which];
}
void gdg_knob_set(int which, int value) {
    if (which < 0 || which >= GDG_KNOB_COUNT) return;
    g_knob[which] = value;
    g_knob_set[which] = true;
}
static int fft_half_lds() { return gdg_knob_get(GDG_KNOB_FFT_HALF_LDS); }

/* a window of W frames of 8192 samples per channel (W in {2, 4, 8, 16}); the four launches of one power-amp step */
template <int W, int C> static void launch_mac_tb(const gdg_fir_chan *d_chans, int n, bool shared, hipStream_t s) {
    if (shared) fir_mac_tb_kernel<W, C, false><<<dim3(8192 / 256, n), dim3(256), 0, s>>>(d_chans, 8192);
    else fir_mac_tb_kernel<W, C, true><<<dim3(8192 / 256, n), dim3(256), 0, s>>>(d_chans, 8192);
}
hipError_t gdg_launch_fir_window(int W, const gdg_fir_chan *d_chans, int n_chans, int shared_spectra, const cplx *d_tw, const cplx *d_tw2, int what,
                                 gdg_shift shift, hipStream_t s) {
    if (n_chans <= 0) return hipSuccess;
    if (W != 2 && W != 4 && W != 8 && W != 16) return hipErrorInvalidValue;
    const int per_channel = gdg_knob_get(GDG_KNOB_FWD_PER_CHANNEL);
    const int half = fft_half_lds();
    if (what == 0 && ((half & 1) || n_chans < cu_count())) {
        /* one buffer, two workgroups per CU.  Off by default for a chip's worth of channels (the walk leaves room for the other channel group's
         * multiply-accumulate, see above); below that there is no second group to make room for and the (channel, frame) grid of one-buffer
         * workgroups, dealt XCD by XCD so that a frame's second read meets its first in L2, is the fastest forward launch:
         * 64 channels 8.3 -> 6.2 us per frame, the window 51.5 -> 49.8; 128 channels 87.1 -> 83.4 (profiles/small_shards_r05.txt) */
        const bool deal = n_chans % 8 == 0 && ((half & 32) || n_chans < cu_count());
        fir_fwd13wh_kernel<1><<<dim3(n_chans * W), dim3(512), 0, s>>>(d_chans, W, shift, d_tw, d_tw2, deal ? 1 : 0);
    } else if (what == 0) {
        /* one workgroup per channel needs a chip's worth of channels; below that the (channel, frame) grid fills the CUs better
         * (64 channels: 9.7 vs 18.8 us per frame) */
        if (per_channel && n_chans >= cu_count()) fir_fwd13w_chan_kernel<<<dim3(n_chans), dim3(512), 0, s>>>(d_chans, W, shift, d_tw, d_tw2);
        else fir_fwd13w_kernel<1><<<dim3(n_chans * W), dim3(512), 0, s>>>(d_chans, W, shift, d_tw, d_tw2);
    } else if (what == 1) {
        if (W == 2) launch_mac_tb<2, 2>(d_chans, n_chans, shared_spectra != 0, s);
        else if (W == 4) launch_mac_tb<4, 4>(d_chans, n_chans, shared_spectra != 0, s);
        else if (W == 8) launch_mac_tb<8, 8>(d_chans, n_chans, shared_spectra != 0, s);
        else launch_mac_tb<16, 8>(d_chans, n_chans, shared_spectra != 0, s);
    } else if (what == 2 && (half & 2)) fir_inv13h_kernel<<<dim3(n_chans * W), dim3(512), 0, s>>>(d_chans, W, shift, d_tw, d_tw2);
    else if (what == 2) fir_inv_kernel<13, 3><<<dim3(n_chans * W), dim3(FftCfg<13>::T), 0, s>>>(d_chans, W, shift, d_tw, d_tw2);
    else fir_tb_finish_kernel<<<dim3(n_chans), dim3(256), 0, s>>>(d_chans, W, shift);
    return hipGetLastError();
}

/* 1 when a window's inverse transforms of one power amp can produce the forward transforms of the next (one workgroup per channel
 * walking the frames: needs a chip's worth of channels, like the per-channel forward kernel) */
int gdg_fir_window_chain_ok(int n_chans, int W) {
    const int per_channel = gdg_knob_get(GDG_KNOB_FWD_PER_CHANNEL);
    if ((fft_half_lds() & 8) && W >= 4) return 0;       /* two frames: the chained kernel still wins (513 against 518 us per frame) */
    return per_channel && n_chans >= cu_count();
}

/* the window's inverse transforms of d_chans and the forward transforms of d_next_chans (the power amp that follows in every channel) */
hipError_t gdg_launch_fir_window_chain(int W, const gdg_fir_chan *d_chans, const gdg_fir_chan *d_next_chans, int n_chans, const cplx *d_tw, const cplx *d_tw2,
                                       gdg_shift shift, hipStream_t s) {
    if (n_chans <= 0) return hipSuccess;
    if (W != 2 && W != 4 && W != 8 && W != 16) return hipErrorInvalidValue;
    fir_inv_fwd_chain_chan_kernel<<<dim3(n_chans), dim3(512), 0, s>>>(d_chans, d_next_chans, W, shift, d_tw, d_tw2);
    return hipGetLastError();
}

hipError_t gdg_launch_fir_fwd(int P, int hop, const gdg_fir_chan *d_chans, int n_chans, const cplx *d_tw, const cplx *d_tw2, gdg_shift shift, hipStream_t s) {
    if (n_chans <= 0) return hipSuccess;
    const int wave_fft = gdg_knob_get(GDG_KNOB_WAVE_FFT);
    if (P == 8192 && hop == P && (fft_half_lds() & 4)) {
        fir_fwd13wh_kernel<0><<<dim3(n_chans), dim3(512), 0, s>>>(d_chans, 1, shift, d_tw, d_tw2, 0);
        return hipGetLastError();
    }
    if (P == 8192 && hop == P && (wave_fft & 1)) {
        fir_fwd13w_kernel<0><<<dim3(n_chans), dim3(512), 0, s>>>(d_chans, 1, shift, d_tw, d_tw2);
        return hipGetLastError();
    }
    int L = ilog2_exact(P);
    GDG_DISPATCH_LOGN(L, launch_fwd<LG>(d_chans, n_chans, d_tw, d_tw2, shift, s));
    return hipGetLastError();
}

hipError_t gdg_launch_fir_ir(int P, const gdg_fir_irjob *d_jobs, int n_jobs, double scale, const cplx *d_tw, const cplx *d_tw2, hipStream_t s) {
    if (n_jobs <= 0) return hipSuccess;
    int L = ilog2_exact(P);
    if (L >= 0 && L <= 5) {
        fft_small_fwd_kernel<<<dim3(n_jobs), dim3(64), 0, s>>>(d_jobs, P, L, scale, d_tw, d_tw2);
        return hipGetLastError();
    }
    GDG_DISPATCH_LOGN(L, launch_ir<LG>(d_jobs, n_jobs, scale, d_tw, d_tw2, s));
    return hipGetLastError();
}

hipError_t gdg_launch_fir_raw_inv(int P, const gdg_fir_rawjob *d_jobs, int n_jobs, double scale, const cplx *d_tw, const cplx *d_tw2, hipStream_t s) {
    if (n_jobs <= 0) return hipSuccess;
    int L = ilog2_exact(P);
    if (L >= 0 && L <= 5) {
        fft_small_inv_kernel<<<dim3(n_jobs), dim3(64), 0, s>>>(d_jobs, P, L, scale, d_tw, d_tw2);
        return hipGetLastError();
    }
    GDG_DISPATCH_LOGN(L, launch_raw_inv<LG>(d_jobs, n_jobs, scale, d_tw, d_tw2, s));
    return hipGetLastError();
}

template <int UNROLL, int BPT, bool NT, bool SWAP = false, bool HNT = NT>
static void launch_mac(int P, const gdg_fir_chan *d_chans, int n_chans, hipStream_t s, int block = 256, int k_lo = 0, int lds_bytes = 0) {
    int per_block = block * BPT;
    int threads = P < per_block ? (P / BPT) : block;
    if (threads < 1) threads = 1;
    unsigned tiles = (unsigned)((P + threads * BPT - 1) / (threads * BPT));
    dim3 grid = SWAP ? dim3((unsigned)n_chans, tiles) : dim3(tiles, (unsigned)n_chans);
    fir_mac_kernel<UNROLL, BPT, NT, SWAP, HNT><<<grid, dim3(threads), (size_t)lds_bytes, s>>>(d_chans, P, k_lo);
}

hipError_t gdg_launch_fir_mac(int P, const gdg_fir_chan *d_chans, int n_chans, int shared_spectra, hipStream_t s, int k_lo, int lds_bytes) {
    if (n_chans <= 0) return hipSuccess;
    /* IR spectra shared between channels are worth caching; private ones are read once: non-temporal like the delay line */
    if (shared_spectra) { launch_mac<8, 1, true, false, false>(P, d_chans, n_chans, s, 256, k_lo, lds_bytes); return hipGetLastError(); }
    /* the premac (k_lo = 1, launched beside the segments of a small shard): seven partitions' loads in flight -- all of 65536 taps' terms but
     * the newest -- and `lds_bytes` of LDS it never touches: a workgroup that asks for them does not fit on a CU beside a segment workgroup
     * (159 KiB, or two of 80), so the sums run on the CUs the segments leave idle instead of among their waves (api_process.cpp) */
    if (k_lo > 0) { launch_mac<7, 1, true>(P, d_chans, n_chans, s, 256, k_lo, lds_bytes); return hipGetLastError(); }
    /* GDG_MAC_VARIANT: tuning knob for profiles/mac_variants.py; the default is the measured best */
    const int variant = gdg_knob_get(GDG_KNOB_MAC_VARIANT);
    switch (variant) {
    case 1: launch_mac<8, 1, false>(P, d_chans, n_chans, s); break;
    case 2: launch_mac<4, 1, true>(P, d_chans, n_chans, s); break;
    case 3: launch_mac<8, 1, true>(P, d_chans, n_chans, s); break;
    case 4: launch_mac<4, 2, false>(P, d_chans, n_chans, s); break;
    case 5: launch_mac<4, 2, true>(P, d_chans, n_chans, s); break;
    case 6: launch_mac<8, 2, true>(P, d_chans, n_chans, s); break;
    case 7: launch_mac<2, 2, false>(P, d_chans, n_chans, s); break;
    case 8: launch_mac<2, 4, true>(P, d_chans, n_chans, s); break;
    case 9: launch_mac<4, 1, false>(P, d_chans, n_chans, s); break;
    case 10: launch_mac<8, 1, true>(P, d_chans, n_chans, s, 64); break;
    case 11: launch_mac<8, 1, true>(P, d_chans, n_chans, s, 128); break;
    case 12: launch_mac<8, 1, true>(P, d_chans, n_chans, s, 512); break;
    case 13: launch_mac<8, 1, true>(P, d_chans, n_chans, s, 1024); break;
    case 14: launch_mac<8, 1, true, true>(P, d_chans, n_chans, s, 256); break;
    case 15: launch_mac<8, 1, true, true>(P, d_chans, n_chans, s, 1024); break;
    default: launch_mac<8, 1, true>(P, d_chans, n_chans, s); break;      /* measured best: profiles/mac_variants_r01.txt */
    }
    return hipGetLastError();
}

/* fused: 0 = inverse only (Y from gdg_launch_fir_mac), 1 = MAC + inverse, 2 = MAC + inverse with shared (cacheable) IR spectra,
 * 4 (P = 8192 only) = Y from gdg_launch_fir_mac(.., k_lo = 1) + the newest partition's term + inverse */
hipError_t gdg_launch_fir_inv(int P, const gdg_fir_chan *d_chans, int n_chans, const cplx *d_tw, const cplx *d_tw2, int fused, gdg_shift shift, hipStream_t s,
                              const gdg_fir_chan *d_next_chans, hipEvent_t ev_begin, hipEvent_t ev_end) {
    if (n_chans <= 0) return hipSuccess;
    int L = ilog2_exact(P);
    if ((d_next_chans || fused == 4) && L != 13) return hipErrorInvalidValue;
    GDG_DISPATCH_LOGN(L, launch_inv<LG>(d_chans, n_chans, d_tw, d_tw2, fused, shift, s, d_next_chans, ev_begin, ev_end));
    return hipGetLastError();
}

#pragma clang fp contract(fast)     /* the tuner has no second path to agree with bit for bit */
#include "tuner_kernels.h"
