/*
 * tuner_kernels.h -- tuner.Process / tuner.Analyze (tuner/tuner.go:379-587) for a batch of
 * independent tuners, one per channel.  Included at the end of fir.hip because it reuses the
 * register/LDS Stockham passes defined there.
 *
 * Analyze = autocorrelation through a 262144-point real FFT (tuner.go:388-444).  The packed
 * complex length N = 131072 does not fit any LDS, so the transform is a four-step FFT over HBM:
 *   N = N1 x N2 = 512 x 256,  n = n1*N2 + n2,  k = k1 + N1*k2
 *   col pass : 256 column FFTs of length 512 (8 adjacent columns per workgroup so that every
 *              global access is a 128-byte run), times the twiddle W_N^(n2*k1)
 *   row pass : 512 row FFTs of length 256 (16 rows per workgroup), result left in the
 *              "transposed" order  pos(k) = (k mod 512)*256 + (k div 512)
 *   square   : X[k] from Z[k], Z[N-k] -> |X[k]|^2 (real, even) -> re-packed spectrum of the inverse
 *   col pass + row pass with conjugated twiddles
 *   pick     : first maximum of r[lag] in the reference's lag window, 3-point parabola, note match.
 * The 1/N scale of the inverse is skipped: arg-max and the parabola's shift are scale free.
 */

#define TUNER_N1 512
#define TUNER_N2 256
#define TUNER_N (TUNER_N1 * TUNER_N2)
#define TUNER_COLS_PER_WG 8
#define TUNER_ROWS_PER_WG 16

/* tuner.Process: append `frames` samples to every ring (circular/circular.go:33-71) */
__global__ void __launch_bounds__(256)
tuner_enqueue_kernel(double *__restrict__ rings, int wp, const double *__restrict__ samples, int stride, int frames) {
    const int c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= frames) return;
    int first = frames > GDG_TUNER_RING ? frames - GDG_TUNER_RING : 0;
    if (i < first) return;
    int p = (wp + (i - first)) % GDG_TUNER_RING;
    rings[(size_t)c * GDG_TUNER_RING + p] = samples[(size_t)c * stride + i];
}

/* column pass: FFTs of length 512 along n1 for 8 adjacent columns */
template <bool INV, bool FROM_RING>
__global__ void __launch_bounds__(256)
tuner_col_kernel(const double *__restrict__ rings, int wp, const cplx *__restrict__ src, cplx *__restrict__ dst,
                 const cplx *__restrict__ tw512, const cplx *__restrict__ twN) {
    constexpr int LOGN = 9, NF = 512, TF = 32, LDSF = FftCfg<LOGN>::LDS;
    __shared__ double sre[TUNER_COLS_PER_WG * LDSF];
    __shared__ double sim[TUNER_COLS_PER_WG * LDSF];
    const int ch = blockIdx.y;
    const int col = threadIdx.x & (TUNER_COLS_PER_WG - 1);
    const int lt = threadIdx.x >> 3;
    const int n2 = blockIdx.x * TUNER_COLS_PER_WG + col;
    double *mre = sre + col * LDSF, *mim = sim + col * LDSF;
    const size_t base = (size_t)ch * TUNER_N;

    constexpr int LR0 = sched_lr(LOGN, 0), R0 = 1 << LR0, B0 = 16 / R0;
    cplx v[16];
#pragma unroll
    for (int b = 0; b < B0; b++) {
        int j = lt + TF * b;
#pragma unroll
        for (int t = 0; t < R0; t++) {
            int e = j + t * (NF / R0);                 /* n1 */
            int n = e * TUNER_N2 + n2;
            cplx val;
            if constexpr (FROM_RING) {
                /* packed real input: z[n] = x[2n] + i x[2n+1], x = ring oldest-first, zero beyond 96000 */
                int i0 = 2 * n, i1 = 2 * n + 1;
                const double *ring = rings + (size_t)ch * GDG_TUNER_RING;
                val.x = (i0 < GDG_TUNER_RING) ? ring[(wp + i0) % GDG_TUNER_RING] : 0.0;
                val.y = (i1 < GDG_TUNER_RING) ? ring[(wp + i1) % GDG_TUNER_RING] : 0.0;
            } else {
                val = src[base + n];
            }
            v[b * R0 + t] = val;
        }
    }
    pass_compute<LOGN, LR0, 0, INV>(v, tw512, lt);
    pass_store<LOGN, LR0, 0>(v, mre, mim, lt);
    __syncthreads();
    run_lds_passes<LOGN, 1, sched_npass(LOGN), INV>(v, mre, mim, tw512, lt);
    /* times W_N^(n2 k1), written back in the same [k1][n2] layout */
#pragma unroll
    for (int i = 0; i < NF / TF; i++) {
        int k1 = lt + TF * i;
        cplx z = make_double2(mre[GDG_PAD(k1)], mim[GDG_PAD(k1)]);
        cplx w = twN[n2 * k1];
        if constexpr (INV) w.y = -w.y;
        dst[base + (size_t)k1 * TUNER_N2 + n2] = cmul(z, w);
    }
}

/* row pass: FFTs of length 256 along n2, 16 rows per workgroup, in place */
template <bool INV>
__global__ void __launch_bounds__(256)
tuner_row_kernel(cplx *__restrict__ data, const cplx *__restrict__ tw256) {
    constexpr int LOGN = 8, NF = 256, TF = 16, LDSF = FftCfg<LOGN>::LDS;
    __shared__ double sre[TUNER_ROWS_PER_WG * LDSF];
    __shared__ double sim[TUNER_ROWS_PER_WG * LDSF];
    const int ch = blockIdx.y;
    const int lt = threadIdx.x & (TF - 1);
    const int row_l = threadIdx.x >> 4;
    const int k1 = blockIdx.x * TUNER_ROWS_PER_WG + row_l;
    double *mre = sre + row_l * LDSF, *mim = sim + row_l * LDSF;
    cplx *row = data + (size_t)ch * TUNER_N + (size_t)k1 * TUNER_N2;

    constexpr int LR0 = sched_lr(LOGN, 0), R0 = 1 << LR0, B0 = 16 / R0;
    cplx v[16];
#pragma unroll
    for (int b = 0; b < B0; b++) {
        int j = lt + TF * b;
#pragma unroll
        for (int t = 0; t < R0; t++) v[b * R0 + t] = row[j + t * (NF / R0)];
    }
    pass_compute<LOGN, LR0, 0, INV>(v, tw256, lt);
    pass_store<LOGN, LR0, 0>(v, mre, mim, lt);
    __syncthreads();
    run_lds_passes<LOGN, 1, sched_npass(LOGN), INV>(v, mre, mim, tw256, lt);
#pragma unroll
    for (int i = 0; i < NF / TF; i++) {
        int k2 = lt + TF * i;
        row[k2] = make_double2(mre[GDG_PAD(k2)], mim[GDG_PAD(k2)]);
    }
}

__device__ __forceinline__ int tuner_pos(int k) { return (k & (TUNER_N1 - 1)) * TUNER_N2 + (k >> 9); }

/* |X[k]|^2 of the real spectrum and the packed input of the inverse real transform; out in natural index order */
__global__ void __launch_bounds__(256)
tuner_square_kernel(const cplx *__restrict__ Z, cplx *__restrict__ out, const cplx *__restrict__ twM) {
    const int ch = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;          /* 0 .. N/2 - 1 */
    const cplx *z = Z + (size_t)ch * TUNER_N;
    cplx *o = out + (size_t)ch * TUNER_N;
    if (k == 0) {
        cplx z0 = z[0];
        double x0 = z0.x + z0.y, xn = z0.x - z0.y;         /* X[0], X[N] (both real) */
        double s0 = x0 * x0, sn = xn * xn;
        o[0] = make_double2(s0 + sn, s0 - sn);
        cplx zh = z[tuner_pos(TUNER_N / 2)];               /* X[N/2] = conj(Z[N/2]) */
        double sh = zh.x * zh.x + zh.y * zh.y;
        o[TUNER_N / 2] = make_double2(2.0 * sh, 0.0);
        return;
    }
    const int n = TUNER_N - k;
    cplx zk = z[tuner_pos(k)], zn = z[tuner_pos(n)];
    cplx A = make_double2(zk.x + zn.x, zk.y - zn.y);
    cplx Bv = make_double2(zk.x - zn.x, zk.y + zn.y);
    cplx w = twM[k];                                       /* exp(-i pi k / N) */
    cplx cw = cmul(w, Bv);
    cplx xk = make_double2(0.5 * (A.x + cw.y), 0.5 * (A.y - cw.x));
    cplx xn = make_double2(0.5 * (A.x - cw.y), 0.5 * (-A.y - cw.x));
    /* elem * conj(elem), tuner.go:424-427 */
    double sk = xk.x * xk.x + xk.y * xk.y;
    double sn = xn.x * xn.x + xn.y * xn.y;
    double a = sk + sn, d = sk - sn;
    o[k] = make_double2(a + d * w.y, d * w.x);
    o[n] = make_double2(a - d * w.y, d * w.x);
}

/* r[lag] (unscaled) from the transposed inverse result: lag = 2m + (0 | 1) */
__device__ __forceinline__ double tuner_corr(const cplx *r, int lag) {
    cplx z = r[tuner_pos(lag >> 1)];
    return (lag & 1) ? z.y : z.x;
}

/* tuner.go:446-567: arg-max in the lag window, parabolic refinement, nearest note.  corr(lag) = r[lag] up to a positive scale.
 * One workgroup of 256 threads; s_val / s_idx: 256 entries of LDS each. */
__device__ __forceinline__ void tuner_lag_window(double sample_rate, const double *__restrict__ note_freqs, int n_notes, long long *low_idx, long long *high_idx) {
    const long long two_n = 2LL * GDG_TUNER_RING;
    const double low_freq = note_freqs[0], high_freq = note_freqs[n_notes - 1];
    double lo_f = (sample_rate / high_freq) + 0.5, hi_f = (sample_rate / low_freq) + 0.5;
    long long lo = (lo_f == lo_f && fabs(lo_f) < 9e18) ? (long long)lo_f : -1;
    if (lo < 0 || lo >= two_n) lo = 0;
    long long hi = (hi_f == hi_f && fabs(hi_f) < 9e18) ? (long long)hi_f : -1;
    if (hi < 0 || hi >= two_n) hi = two_n - 1;
    *low_idx = lo;
    *high_idx = hi;
}

template <class Corr>
__device__ __forceinline__ void tuner_pick(Corr corr, double sample_rate, const double *__restrict__ note_freqs, int n_notes,
                                           gdg_tuner_out *__restrict__ out_ch, double *s_val, int *s_idx, unsigned seq) {
    const int tid = threadIdx.x;
    const int n = GDG_TUNER_RING;
    long long low_idx, high_idx;
    tuner_lag_window(sample_rate, note_freqs, n_notes, &low_idx, &high_idx);
    double best = -INFINITY;
    int best_i = -1;
    for (long long i = low_idx + tid; i < high_idx; i += 256) {
        double v = corr((int)i);
        if (v > best) { best = v; best_i = (int)i; }            /* ascending i per thread: first maximum wins */
    }
    s_val[tid] = best; s_idx[tid] = best_i;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            double v2 = s_val[tid + s]; int i2 = s_idx[tid + s];
            double v1 = s_val[tid]; int i1 = s_idx[tid];
            bool take = (i2 >= 0) && (i1 < 0 || v2 > v1 || (v2 == v1 && i2 < i1));
            if (take) { s_val[tid] = v2; s_idx[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        double max_val = s_val[0];
        int idx = (s_idx[0] >= 0) ? s_idx[0] : (int)low_idx - 1;
        int idx_up = idx + 1; if (idx_up > n) idx_up = n;
        int idx_down = idx - 1; if (idx_down < 0) idx_down = 0;
        double value_left = corr(idx_down), value_right = corr(idx_up);
        double idx_float = (double)idx;
        double value_diff = value_right - value_left;
        double value_sum = value_right + value_left;
        double half_diff = 0.5 * value_diff;
        double double_max = 2.0 * max_val;
        double denominator = double_max - value_sum;
        double shift = half_diff / denominator;
        if (shift < -0.5) shift = -0.5; else if (shift > 0.5) shift = 0.5;
        idx_float += shift;
        s_val[0] = sample_rate / idx_float;
    }
    __syncthreads();
    /* the nearest note (tuner.go:470-497): one note per thread instead of one thread walking all of them through log2 (~10 us of a 190 us
     * analysis); ties go to the first note, as in the reference's strict "<" walk */
    const double freq = s_val[0];
    __syncthreads();
    double da = INFINITY, dc = INFINITY;
    int note = -1;
    for (int k = tid; k < n_notes; k += 256) {
        double ratio = freq / note_freqs[k];
        double c = 1200.0 * log2(ratio);
        double ca = fabs(c);
        if (ca < da) { note = k; dc = c; da = ca; }
    }
    __shared__ double s_cents[256];
    s_val[tid] = da; s_idx[tid] = note; s_cents[tid] = dc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const double a2 = s_val[tid + s], a1 = s_val[tid];
            const int k2 = s_idx[tid + s], k1 = s_idx[tid];
            const bool take = (k2 >= 0) && (k1 < 0 || a2 < a1 || (a2 == a1 && k2 < k1));
            if (take) { s_val[tid] = a2; s_idx[tid] = k2; s_cents[tid] = s_cents[tid + s]; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double cents = s_cents[0];
        int cents_int = 0;
        if (!(isinf(cents) || isnan(cents))) cents_int = (int)(signed char)(int)cents;
        out_ch->frequency = freq;
        out_ch->note_index = s_idx[0];
        out_ch->cents = cents_int;
        /* the record is complete for whoever reads `seq` (the host polls it in mapped host memory): system-scope release, then the number */
        __threadfence_system();
        __hip_atomic_store(&out_ch->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void __launch_bounds__(256)
tuner_pick_kernel(const cplx *__restrict__ R, double sample_rate, const double *__restrict__ note_freqs, int n_notes,
                  gdg_tuner_out *__restrict__ out, unsigned seq) {
    __shared__ double s_val[256];
    __shared__ int s_idx[256];
    const cplx *r = R + (size_t)blockIdx.x * TUNER_N;
    tuner_pick([&](int lag) { return tuner_corr(r, lag); }, sample_rate, note_freqs, n_notes, out + blockIdx.x, s_val, s_idx, seq);
}

/* ------------------------------------------------------------------------------------------------
 * The short-lag analysis (every standard rate: the lag window ends at sr / 61.7354 + 0.5 <= 3111 at 192 kHz).
 *
 * Only r[lag] for lag <= high index + 1 is ever looked at, so the 262144-point transform pair is not needed: the linear
 * autocorrelation of the zero-padded signal (what tuner.go:408-444 computes) is, for lag < 4097,
 *     r[l] = sum_b sum_{n in block b} x[n] x[n + l],      blocks of B = 4096 samples,
 * and per block  sum_n a_b[n] c_b[n + l] = IFFT_8192( conj(A_b) C_b )[l]  with a_b = block b zero-padded to 8192 and
 * c_b = blocks b, b + 1.  Since c_b = a_b + shift_4096(a_{b+1}), C_b = A_b + (-1)^k A_{b+1}: ONE forward transform per block,
 *     S[k] = sum_b |A_b[k]|^2 + (-1)^k sum_b conj(A_b[k]) A_{b+1}[k],     r = IFFT_8192(S).
 * One workgroup per channel: 24 packed-real forward transforms of 8192 points (4096 complex, registers + 70 KiB LDS), the
 * running spectrum A_b and the sum S in registers, one inverse, the pick on the result in LDS.  HBM traffic = the ring, read
 * ONCE: 768 kB per analysis -- the algorithmic minimum -- instead of ~19 MiB of four-step passes.
 * Differs from the long transform by rounding only (~1e-15 relative in r).
 * ---------------------------------------------------------------------------------------------- */
#define TUNER_BLK 4096
#define TUNER_SHORT_MAX_LAG TUNER_BLK          /* r[0 .. 4096] are free of wrap-around */

/* blocks [blk_begin, blk_end) of one channel's ring into the running sums (sk, sn); `lead`: block blk_begin - 1 is transformed first, only to
 * serve as "previous block" of the first cross term (a part of a split analysis that does not start at block 0) */
template <int LOGN>
__device__ __forceinline__ void tuner_accumulate_blocks(const double *ring, int wp, int blk_begin, int blk_end, bool lead, double *sre, double *sim,
                                                        const cplx *__restrict__ tw, const cplx *__restrict__ tw2,
                                                        cplx (&sk)[(FftCfg<LOGN>::N / 2) / FftCfg<LOGN>::T], cplx (&sn)[(FftCfg<LOGN>::N / 2) / FftCfg<LOGN>::T]) {
    constexpr int N = FftCfg<LOGN>::N, T = FftCfg<LOGN>::T, ITER = (N / 2) / T;
    constexpr int LR0 = sched_lr(LOGN, 0), R0 = 1 << LR0, B0 = 16 / R0;
    const int tid = threadIdx.x;
    /* per thread: bins k = tid + T i and n = N - k (i = 0: thread 0 holds (X[0], X[N]) as two reals and X[N/2]) */
    cplx pk[ITER], pn[ITER];
#pragma unroll
    for (int i = 0; i < ITER; i++) { pk[i] = pn[i] = make_double2(0.0, 0.0); }
    /* the samples of a block: packed element e = (a[2e], a[2e+1]) for e < 2048 (the other half of the transform's input is zeros); a thread's
     * eight non-zero elements are the t < R0 / 2 of its butterflies.  They are requested ONE BLOCK AHEAD: a block's loads used to be waited
     * for at the top of its transform, 24 exposed HBM latencies per analysis (~2 us each of the ~6 us a block took). */
    constexpr int NZ = 16 / 2;
    static_assert(B0 * (R0 / 2) == NZ && (N / R0) * (R0 / 2) == TUNER_BLK / 2, "the first half of every butterfly's inputs is the block, the second half zeros");
    cplx nx[NZ];
    auto request = [&](int blk) {
#pragma unroll
        for (int b = 0; b < B0; b++) {
            const int j = tid + T * b;
#pragma unroll
            for (int t = 0; t < R0 / 2; t++) {
                const int e = j + t * (N / R0);
                cplx val = make_double2(0.0, 0.0);
                const int i0 = blk * TUNER_BLK + 2 * e;                                         /* oldest-first sample index */
                if (i0 < GDG_TUNER_RING) { int p = wp + i0; if (p >= GDG_TUNER_RING) p -= GDG_TUNER_RING; val.x = gload1(ring + p); }
                if (i0 + 1 < GDG_TUNER_RING) { int p = wp + i0 + 1; if (p >= GDG_TUNER_RING) p -= GDG_TUNER_RING; val.y = gload1(ring + p); }
                nx[b * (R0 / 2) + t] = val;
            }
        }
    };
    const int blk_first = lead ? blk_begin - 1 : blk_begin;
    if (blk_first < blk_end) request(blk_first);
    for (int blk = blk_first; blk < blk_end; blk++) {
        const bool count = blk >= blk_begin;
        cplx v[16];
#pragma unroll
        for (int b = 0; b < B0; b++)
#pragma unroll
            for (int t = 0; t < R0; t++) v[b * R0 + t] = (t < R0 / 2) ? nx[b * (R0 / 2) + t] : make_double2(0.0, 0.0);
        if (blk + 1 < blk_end) request(blk + 1);                                                /* in flight during this block's transform */
        pass_compute<LOGN, LR0, 0, false>(v, tw, tid);
        pass_store<LOGN, LR0, 0>(v, sre, sim, tid);
        __syncthreads();
        run_lds_passes<LOGN, 1, sched_npass(LOGN), false>(v, sre, sim, tw, tid);
        /* un-pack Z -> the real sequence's bins, accumulate |A|^2 and (-1)^k conj(A_prev) A */
#pragma unroll
        for (int i = 0; i < ITER; i++) {
            const int k = tid + T * i;
            cplx ak, an;
            if (k == 0) {
                const double zx = sre[0], zy = sim[0];
                ak = make_double2(zx + zy, zx - zy);                                            /* X[0], X[N]: two reals */
                an = make_double2(sre[GDG_PAD(N / 2)], -sim[GDG_PAD(N / 2)]);                   /* X[N/2] */
                if (count) {
                    sk[i].x += ak.x * ak.x + pk[i].x * ak.x;                                    /* bins 0 and N (= 4096) are even */
                    sk[i].y += ak.y * ak.y + pk[i].y * ak.y;
                    sn[i].x += an.x * an.x + an.y * an.y + (pn[i].x * an.x + pn[i].y * an.y);   /* N/2 = 2048 is even */
                    sn[i].y += pn[i].x * an.y - pn[i].y * an.x;
                }
            } else {
                const int n = N - k;
                const cplx zk = make_double2(sre[GDG_PAD(k)], sim[GDG_PAD(k)]), zn = make_double2(sre[GDG_PAD(n)], sim[GDG_PAD(n)]);
                const cplx A = make_double2(zk.x + zn.x, zk.y - zn.y), Bv = make_double2(zk.x - zn.x, zk.y + zn.y);
                const cplx cw = cmul(tw2[k], Bv);
                ak = make_double2((A.x + cw.y) * 0.5, (A.y - cw.x) * 0.5);
                an = make_double2((A.x - cw.y) * 0.5, (-A.y - cw.x) * 0.5);
                if (count) {
                    const double sg = (k & 1) ? -1.0 : 1.0;                                     /* n = N - k has the parity of k */
                    sk[i].x += ak.x * ak.x + ak.y * ak.y + sg * (pk[i].x * ak.x + pk[i].y * ak.y);      /* conj(p) a */
                    sk[i].y += sg * (pk[i].x * ak.y - pk[i].y * ak.x);
                    sn[i].x += an.x * an.x + an.y * an.y + sg * (pn[i].x * an.x + pn[i].y * an.y);
                    sn[i].y += sg * (pn[i].x * an.y - pn[i].y * an.x);
                }
            }
            pk[i] = ak; pn[i] = an;
        }
        __syncthreads();                                                                        /* everyone is done reading Z */
    }
}

/* the sums S -> r (inverse packed-real transform, unscaled: arg-max and parabola are scale free) -> the pick */
template <int LOGN>
__device__ __forceinline__ void tuner_finish(const cplx (&sk)[(FftCfg<LOGN>::N / 2) / FftCfg<LOGN>::T], const cplx (&sn)[(FftCfg<LOGN>::N / 2) / FftCfg<LOGN>::T],
                                             double *s_all, double *s_val, int *s_idx, double sample_rate, const cplx *__restrict__ tw,
                                             const cplx *__restrict__ tw2, const double *__restrict__ note_freqs, int n_notes, gdg_tuner_out *out_ch, unsigned seq) {
    constexpr int N = FftCfg<LOGN>::N, T = FftCfg<LOGN>::T, ITER = (N / 2) / T;
    double *sre = s_all, *sim = s_all + FftCfg<LOGN>::LDS;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ITER; i++) inv_head_store<LOGN>(tid + T * i, sk[i], sn[i], sre, sim, tw2);
    __syncthreads();
    cplx v[16];
    constexpr int NP = sched_npass(LOGN);
    run_lds_passes<LOGN, 0, NP - 1, true>(v, sre, sim, tw, tid);
    constexpr int LR = sched_lr(LOGN, NP - 1), LNS = sched_lns(LOGN, NP - 1), R = 1 << LR, B = 16 / R;
    pass_load<LOGN, LR>(v, sre, sim, tid);
    pass_compute<LOGN, LR, LNS, true>(v, tw, tid);
    __syncthreads();                                                                            /* all inputs of the last pass are in registers */
#pragma unroll
    for (int b = 0; b < B; b++) {
        const int j = tid + T * b;
#pragma unroll
        for (int t = 0; t < R; t++) {
            const int n = j + t * (N / R);
            s_all[2 * n] = v[b * R + t].x;                                                      /* linear: 8192 <= 2 * LDS */
            s_all[2 * n + 1] = v[b * R + t].y;
        }
    }
    __syncthreads();
    tuner_pick([&](int lag) { return s_all[lag]; }, sample_rate, note_freqs, n_notes, out_ch, s_val, s_idx, seq);
}

#define TUNER_NBLK ((GDG_TUNER_RING + TUNER_BLK - 1) / TUNER_BLK)                                /* 24 */

/* one workgroup per channel: all 24 blocks, then the finish (a chip's worth of channels) */
__global__ void __launch_bounds__(256)
tuner_short_kernel(const double *__restrict__ rings, int wp, double sample_rate, const cplx *__restrict__ tw, const cplx *__restrict__ tw2,
                   const double *__restrict__ note_freqs, int n_notes, gdg_tuner_out *__restrict__ out, unsigned seq) {
    constexpr int LOGN = 12, ITER = (FftCfg<LOGN>::N / 2) / FftCfg<LOGN>::T;                    /* 4096 complex points, 256 threads */
    __shared__ double s_all[2 * FftCfg<LOGN>::LDS];
    __shared__ double s_val[256];
    __shared__ int s_idx[256];
    const int ch = blockIdx.x;
    cplx sk[ITER], sn[ITER];
#pragma unroll
    for (int i = 0; i < ITER; i++) { sk[i] = sn[i] = make_double2(0.0, 0.0); }
    tuner_accumulate_blocks<LOGN>(rings + (size_t)ch * GDG_TUNER_RING, wp, 0, TUNER_NBLK, false, s_all, s_all + FftCfg<LOGN>::LDS, tw, tw2, sk, sn);
    tuner_finish<LOGN>(sk, sn, s_all, s_val, s_idx, sample_rate, tw, tw2, note_freqs, n_notes, out + ch, seq);
}

/* Fewer channels than CUs (BASELINE config 5 on 8 GPUs: 32 tuners per GPU): a channel's 24 blocks are cut into `parts` runs, one workgroup
 * each (blockIdx.x = channel * parts + part; a run that does not start at block 0 transforms its predecessor once more for the cross term);
 * the partial sums S_p go to HBM (64 KiB each) and tuner_short_finish_kernel adds them in part order, inverts and picks. */
__global__ void __launch_bounds__(256)
tuner_short_part_kernel(const double *__restrict__ rings, int wp, int parts, const cplx *__restrict__ tw, const cplx *__restrict__ tw2, cplx *__restrict__ partial) {
    constexpr int LOGN = 12, N = FftCfg<LOGN>::N, T = FftCfg<LOGN>::T, ITER = (N / 2) / T;
    __shared__ double s_all[2 * FftCfg<LOGN>::LDS];
    const int ch = blockIdx.x / parts, part = blockIdx.x % parts, tid = threadIdx.x;
    const int b0 = part * TUNER_NBLK / parts, b1 = (part + 1) * TUNER_NBLK / parts;
    cplx sk[ITER], sn[ITER];
#pragma unroll
    for (int i = 0; i < ITER; i++) { sk[i] = sn[i] = make_double2(0.0, 0.0); }
    tuner_accumulate_blocks<LOGN>(rings + (size_t)ch * GDG_TUNER_RING, wp, b0, b1, b0 > 0, s_all, s_all + FftCfg<LOGN>::LDS, tw, tw2, sk, sn);
    cplx *dst = partial + (size_t)blockIdx.x * N;                  /* [i][0 = k side, 1 = n side][tid] */
#pragma unroll
    for (int i = 0; i < ITER; i++) { gstore(dst + (2 * i) * T + tid, sk[i]); gstore(dst + (2 * i + 1) * T + tid, sn[i]); }
}

__global__ void __launch_bounds__(256)
tuner_short_finish_kernel(const cplx *__restrict__ partial, int parts, double sample_rate, const cplx *__restrict__ tw, const cplx *__restrict__ tw2,
                          const double *__restrict__ note_freqs, int n_notes, gdg_tuner_out *__restrict__ out, unsigned seq) {
    constexpr int LOGN = 12, N = FftCfg<LOGN>::N, T = FftCfg<LOGN>::T, ITER = (N / 2) / T;
    __shared__ double s_all[2 * FftCfg<LOGN>::LDS];
    __shared__ double s_val[256];
    __shared__ int s_idx[256];
    const int ch = blockIdx.x, tid = threadIdx.x;
    cplx sk[ITER], sn[ITER];
#pragma unroll
    for (int i = 0; i < ITER; i++) { sk[i] = sn[i] = make_double2(0.0, 0.0); }
    /* in part order; FOUR parts' loads (4 x 2 ITER sixteen-byte loads per lane) are in flight before the first is added: one part per round
     * was eight exposed HBM latencies on the one CU that finishes a channel (12.6 us at 8 parts) */
    int p = 0;
#pragma unroll 1
    for (; p + 4 <= parts; p += 4) {
        cplx a[4][ITER], b[4][ITER];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const cplx *src = partial + ((size_t)ch * parts + p + q) * N;
#pragma unroll
            for (int i = 0; i < ITER; i++) { a[q][i] = gload(src + (2 * i) * T + tid); b[q][i] = gload(src + (2 * i + 1) * T + tid); }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int i = 0; i < ITER; i++) { sk[i].x += a[q][i].x; sk[i].y += a[q][i].y; sn[i].x += b[q][i].x; sn[i].y += b[q][i].y; }
    }
    for (; p < parts; p++) {
        const cplx *src = partial + ((size_t)ch * parts + p) * N;
#pragma unroll
        for (int i = 0; i < ITER; i++) {
            const cplx a = gload(src + (2 * i) * T + tid), b = gload(src + (2 * i + 1) * T + tid);
            sk[i].x += a.x; sk[i].y += a.y; sn[i].x += b.x; sn[i].y += b.y;
        }
    }
    tuner_finish<LOGN>(sk, sn, s_all, s_val, s_idx, sample_rate, tw, tw2, note_freqs, n_notes, out + ch, seq);
}

hipError_t gdg_launch_tuner_enqueue(double *d_rings, int nch, int wp, const double *d_samples, int stride, int frames, hipStream_t s) {
    if (frames <= 0) return hipSuccess;
    tuner_enqueue_kernel<<<dim3((frames + 255) / 256, nch), dim3(256), 0, s>>>(d_rings, wp, d_samples, stride, frames);
    return hipGetLastError();
}

/* d_tw_n: exp(-2 pi i m / N), m < N;  d_tw_m: exp(-i pi k / N), k <= N/2 */
hipError_t gdg_tuner_tables_create(cplx **d_tw_n, cplx **d_tw_m) { return gdg_fir_tables_create(TUNER_N, d_tw_n, d_tw_m); }

/* 1 when every lag the analysis can look at (window end + 1, tuner.go:454-500) lies inside the short-lag kernel's range */
int gdg_tuner_short_ok(double sample_rate, double lowest_note_frequency) {
    double hi_f = (sample_rate / lowest_note_frequency) + 0.5;
    if (!(hi_f == hi_f) || hi_f < 0.0) return 0;
    return hi_f + 1.0 <= (double)TUNER_SHORT_MAX_LAG;
}

/* workgroups per channel of the short-lag analysis: one from a chip's worth of channels on, else enough runs of blocks to put a workgroup
 * on every CU (at most 8: three blocks + the repeated predecessor each) */
int gdg_tuner_short_parts(int nch) {
    const int forced = gdg_knob_get(GDG_KNOB_TUNER_PARTS);        /* 0: by channel count (gdg_ctx_set_option "tuner_parts") */
    /* as many runs per channel as keep the launch within ONE workgroup per CU (floor: 48 channels x 6 runs = 288 workgroups on 256 CUs took 79 us where
     * x 4 takes 62, profiles/tuner32_parts_poll_r06.txt) */
    int parts = forced > 0 ? forced : (nch >= cu_count() ? 1 : cu_count() / nch);
    if (parts > (forced > 0 ? TUNER_NBLK : 8)) parts = forced > 0 ? TUNER_NBLK : 8;
    return parts < 1 ? 1 : parts;
}

hipError_t gdg_launch_tuner_short(const double *d_rings, int nch, int wp, double sample_rate, const cplx *tw4096, const cplx *tw2_4096,
                                  const double *d_note_freqs, int n_notes, gdg_tuner_out *d_out, cplx *d_partial, int parts, hipStream_t s, unsigned seq) {
    if (parts <= 1 || !d_partial) {
        tuner_short_kernel<<<dim3(nch), dim3(256), 0, s>>>(d_rings, wp, sample_rate, tw4096, tw2_4096, d_note_freqs, n_notes, d_out, seq);
        return hipGetLastError();
    }
    tuner_short_part_kernel<<<dim3(nch * parts), dim3(256), 0, s>>>(d_rings, wp, parts, tw4096, tw2_4096, d_partial);
    tuner_short_finish_kernel<<<dim3(nch), dim3(256), 0, s>>>(d_partial, parts, sample_rate, tw4096, tw2_4096, d_note_freqs, n_notes, d_out, seq);
    return hipGetLastError();
}

hipError_t gdg_launch_tuner_analyze(const double *d_rings, int nch, int wp, double sample_rate, cplx *d_work,
                                    const cplx *d_tw_n, const cplx *d_tw_m, const cplx *tw512, const cplx *tw256,
                                    const double *d_note_freqs, int n_notes, gdg_tuner_out *d_out, hipStream_t s, unsigned seq) {
    cplx *bufA = d_work, *bufB = d_work + (size_t)nch * TUNER_N;
    tuner_col_kernel<false, true><<<dim3(TUNER_N2 / TUNER_COLS_PER_WG, nch), dim3(256), 0, s>>>(d_rings, wp, nullptr, bufA, tw512, d_tw_n);
    tuner_row_kernel<false><<<dim3(TUNER_N1 / TUNER_ROWS_PER_WG, nch), dim3(256), 0, s>>>(bufA, tw256);
    tuner_square_kernel<<<dim3(TUNER_N / 2 / 256, nch), dim3(256), 0, s>>>(bufA, bufB, d_tw_m);
    tuner_col_kernel<true, false><<<dim3(TUNER_N2 / TUNER_COLS_PER_WG, nch), dim3(256), 0, s>>>(nullptr, 0, bufB, bufA, tw512, d_tw_n);
    tuner_row_kernel<true><<<dim3(TUNER_N1 / TUNER_ROWS_PER_WG, nch), dim3(256), 0, s>>>(bufA, tw256);
    tuner_pick_kernel<<<dim3(nch), dim3(256), 0, s>>>(bufA, sample_rate, d_note_freqs, n_notes, d_out, seq);
    return hipGetLastError();
}
