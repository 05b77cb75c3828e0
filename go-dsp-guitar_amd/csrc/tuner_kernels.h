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

/* tuner.go:446-567: arg-max in the lag window, parabolic refinement, nearest note */
__global__ void __launch_bounds__(256)
tuner_pick_kernel(const cplx *__restrict__ R, double sample_rate, const double *__restrict__ note_freqs, int n_notes,
                  gdg_tuner_out *__restrict__ out) {
    __shared__ double s_val[256];
    __shared__ int s_idx[256];
    const int ch = blockIdx.x, tid = threadIdx.x;
    const cplx *r = R + (size_t)ch * TUNER_N;
    const int n = GDG_TUNER_RING;
    const long long two_n = 2LL * n;
    const double low_freq = note_freqs[0], high_freq = note_freqs[n_notes - 1];
    double lo_f = (sample_rate / high_freq) + 0.5, hi_f = (sample_rate / low_freq) + 0.5;
    long long low_idx = (lo_f == lo_f && fabs(lo_f) < 9e18) ? (long long)lo_f : -1;
    if (low_idx < 0 || low_idx >= two_n) low_idx = 0;
    long long high_idx = (hi_f == hi_f && fabs(hi_f) < 9e18) ? (long long)hi_f : -1;
    if (high_idx < 0 || high_idx >= two_n) high_idx = two_n - 1;
    double best = -INFINITY;
    int best_i = -1;
    for (long long i = low_idx + tid; i < high_idx; i += 256) {
        double v = tuner_corr(r, (int)i);
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
        double value_left = tuner_corr(r, idx_down), value_right = tuner_corr(r, idx_up);
        double idx_float = (double)idx;
        double value_diff = value_right - value_left;
        double value_sum = value_right + value_left;
        double half_diff = 0.5 * value_diff;
        double double_max = 2.0 * max_val;
        double denominator = double_max - value_sum;
        double shift = half_diff / denominator;
        if (shift < -0.5) shift = -0.5; else if (shift > 0.5) shift = 0.5;
        idx_float += shift;
        double freq = sample_rate / idx_float;
        int note = -1;
        double cents = INFINITY, cents_abs = INFINITY;
        for (int k = 0; k < n_notes; k++) {
            double ratio = freq / note_freqs[k];
            double dc = 1200.0 * log2(ratio);
            double da = fabs(dc);
            if (da < cents_abs) { note = k; cents = dc; cents_abs = da; }
        }
        int cents_int = 0;
        if (!(isinf(cents) || isnan(cents))) cents_int = (int)(signed char)(int)cents;
        out[ch].frequency = freq;
        out[ch].note_index = note;
        out[ch].cents = cents_int;
    }
}

hipError_t gdg_launch_tuner_enqueue(double *d_rings, int nch, int wp, const double *d_samples, int stride, int frames, hipStream_t s) {
    if (frames <= 0) return hipSuccess;
    tuner_enqueue_kernel<<<dim3((frames + 255) / 256, nch), dim3(256), 0, s>>>(d_rings, wp, d_samples, stride, frames);
    return hipGetLastError();
}

/* d_tw_n: exp(-2 pi i m / N), m < N;  d_tw_m: exp(-i pi k / N), k <= N/2 */
hipError_t gdg_tuner_tables_create(cplx **d_tw_n, cplx **d_tw_m) { return gdg_fir_tables_create(TUNER_N, d_tw_n, d_tw_m); }

hipError_t gdg_launch_tuner_analyze(const double *d_rings, int nch, int wp, double sample_rate, cplx *d_work,
                                    const cplx *d_tw_n, const cplx *d_tw_m, const cplx *tw512, const cplx *tw256,
                                    const double *d_note_freqs, int n_notes, gdg_tuner_out *d_out, hipStream_t s) {
    cplx *bufA = d_work, *bufB = d_work + (size_t)nch * TUNER_N;
    tuner_col_kernel<false, true><<<dim3(TUNER_N2 / TUNER_COLS_PER_WG, nch), dim3(256), 0, s>>>(d_rings, wp, nullptr, bufA, tw512, d_tw_n);
    tuner_row_kernel<false><<<dim3(TUNER_N1 / TUNER_ROWS_PER_WG, nch), dim3(256), 0, s>>>(bufA, tw256);
    tuner_square_kernel<<<dim3(TUNER_N / 2 / 256, nch), dim3(256), 0, s>>>(bufA, bufB, d_tw_m);
    tuner_col_kernel<true, false><<<dim3(TUNER_N2 / TUNER_COLS_PER_WG, nch), dim3(256), 0, s>>>(nullptr, 0, bufB, bufA, tw512, d_tw_n);
    tuner_row_kernel<true><<<dim3(TUNER_N1 / TUNER_ROWS_PER_WG, nch), dim3(256), 0, s>>>(bufA, tw256);
    tuner_pick_kernel<<<dim3(nch), dim3(256), 0, s>>>(bufA, sample_rate, d_note_freqs, n_notes, d_out);
    return hipGetLastError();
}
