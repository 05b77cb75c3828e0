/*
 * api_plan.cpp -- the launch plan: per-unit constants in the reference's arithmetic, scan tables, IR spectra and delay lines, the steps of a call.
 * Part of the host side of libgdg.so (the C-ABI of include/gdg.h on top of the HIP kernels; see ctx.h for the map).
 * There is no CPU compute path here: every sample is produced by a HIP kernel.
 */
#include "ctx.h"

/* ---- plan ----------------------------------------------------------------------------------------- */

static int ensure_hist(gdg_ctx *ctx, Unit &u, size_t len, long long key) {
    if (u.hist_key == key && u.hist_len == len) return GDG_OK;
    if (u.d_hist) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); ctx->arena.release(u.d_hist); u.d_hist = nullptr; }
    u.hist_len = len;
    u.hist_key = key;
    if (len > 0) {
        HIP_TRY(ctx, ctx->arena.alloc_zeroed((void **)&u.d_hist, len * sizeof(double), ctx->stream));
    }
    return GDG_OK;
}

static int zero_is(gdg_ctx *ctx, Unit &u, int first, int count) {
    HIP_TRY(ctx, hipMemsetAsync(u.d_is + first, 0, (size_t)count * sizeof(int), ctx->stream));
    return GDG_OK;
}

/* ---- scan tables of the constant-coefficient recurrences (seg.hip: lin_scan, lin2_scan) ----------------------------------------------
 * Powers of the 8-sample chunk map of a one-pole section / follower (scalar A = keep^8) or of a high-pass feeding a low-pass (2 x 2
 * lower-triangular P = M^8), by binary exponentiation in plain FP64 multiplications -- the arithmetic the kernel itself used while
 * it still built them on every call.  They depend on the coefficients only, i.e. on parameters and the sample rate: built when a plan
 * is built, one copy in HBM per distinct coefficient set (512 channels with the same tone-stack settings share one). */
static double pow_u(double A, int e) {
    double r = 1.0, b = A;
    for (int k = 0; k < 10; k++) { if (e & (1 << k)) r *= b; b *= b; }
    return r;
}
/* chk = samples per thread of the kernel that will read the table: 8 (seg_kernel) or 16 (the two-per-CU kernel, GDG_CHK_FAST) */
static void lin_tab_host(bool maxop, double a, double keep, double *tab, int chk) {
    const double k2 = keep * keep, k4 = k2 * k2, A8 = k4 * k4, A = chk == 16 ? A8 * A8 : A8;
    for (int lane = 0; lane < 64; lane++) {
        tab[LT_PC_(chk) + lane] = pow_u(A, lane);
        if (lane < 16) tab[LT_PA_(chk) + lane] = pow_u(A, lane + 1);
        if (lane >= 32) tab[LT_PB_(chk) + lane - 32] = pow_u(A, lane - 31);
        if (lane < 10) tab[LT_ST_(chk) + lane] = pow_u(A, 1 << lane);
        if (lane < chk) tab[LT_W_(chk) + lane] = (maxop ? 1.0 : a) * pow_u(keep, chk - 1 - lane);
    }
}
struct Tri { double a, b, c; };                                /* [[a, 0], [b, c]] */
static Tri tri_mul(const Tri &x, const Tri &y) { Tri r = { x.a * y.a, (x.b * y.a) + (x.c * y.b), x.c * y.c }; return r; }
static Tri tri_pow(Tri M, int e) {
    Tri r = { 1.0, 0.0, 1.0 };
    for (int k = 0; k < 10; k++) { if (e & (1 << k)) r = tri_mul(r, M); M = tri_mul(M, M); }
    return r;
}
static void tri_store(double *p, const Tri &t) { p[0] = t.a; p[1] = t.b; p[2] = t.c; }
/* per sample (h, l) <- M (h, l) + (aH, aL) x, M = [[1-aH, 0], [-aL, 1-aL]] */
static void lin2_tab_host(double aH, double aL, double *tab, int chk) {
    const Tri M = { 1.0 - aH, -aL, 1.0 - aL };
    const Tri P = tri_pow(M, chk);
    for (int lane = 0; lane < 64; lane++) {
        tri_store(tab + L2_PC_(chk) + 3 * lane, tri_pow(P, lane));
        if (lane < 16) tri_store(tab + L2_PA_(chk) + 3 * lane, tri_pow(P, lane + 1));
        if (lane >= 32) tri_store(tab + L2_PB_(chk) + 3 * (lane - 32), tri_pow(P, lane - 31));
        if (lane < 10) tri_store(tab + L2_ST_(chk) + 3 * lane, tri_pow(P, 1 << lane));
        if (lane < chk) {
            Tri G = tri_pow(M, chk - 1 - lane);
            tab[L2_W_(chk) + 2 * lane] = G.a * aH;
            tab[L2_W_(chk) + 2 * lane + 1] = (G.b * aH) + (G.c * aL);
        }
    }
}
/* the tables of `key` (a tag + the coefficients): the device copy, BUILT and uploaded on first use only -- 512 channels with the same
 * tone-stack setting asked for the same 12 KB table 512 times, and building it (4 bands x ~200 triangular matrix powers) before the
 * look-up cost 13 us each: 6.6 of the 6.2-6.8 ms a plan of 512 channels took to rebuild */
static int scan_tables(gdg_ctx *ctx, const std::vector<double> &key, size_t n_doubles, const std::function<void(double *)> &build, const double **out) {
    auto it = ctx->scan_tabs.find(key);
    if (it == ctx->scan_tabs.end()) {
        std::vector<double> tab(n_doubles, 0.0);
        build(tab.data());
        double *d = nullptr;
        HIP_TRY(ctx, ctx->arena.alloc((void **)&d, tab.size() * sizeof(double)));
        HIP_TRY(ctx, hipMemcpy(d, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
        it = ctx->scan_tabs.emplace(key, d).first;
    }
    *out = it->second;
    return GDG_OK;
}
/* [follower | coupling capacitor] of fuzz / octaver, or the follower alone (compressor): follow 0 = peak (max-affine), 1 = level */
static int follower_tables(gdg_ctx *ctx, int follow, double d_inv, double d, bool with_cap, const double **out, int chk) {
    std::vector<double> key = { 1.0 + 0.001 * chk, (double)follow, d_inv, d, with_cap ? 1.0 : 0.0 };
    return scan_tables(ctx, key, (size_t)(with_cap ? 2 : 1) * LT_SIZE_(chk), [&](double *tab) {
        if (follow == 0) lin_tab_host(true, 0.0, d_inv, tab, chk);
        else lin_tab_host(false, d, d_inv, tab, chk);
        if (with_cap) lin_tab_host(false, d, 1.0 - d, tab + LT_SIZE_(chk), chk);
    }, out);
}

/* May this unit run on the two-per-CU segment kernel (seg.hip compiled with SEG_FAST: 8192-sample frames, one LDS frame buffer, every
 * unit in place)?  By type (gdg_segf_supported), then by what the in-place variants assume: no oversampling (the oversampled shapers
 * stage a whole output frame in a second buffer); the reverb with every tap at least a frame back (rates from 42.7 kHz) and all-pass
 * rings of at most 3072 / 1024 values for the two short ones (rates up to 226 kHz). */
bool segf_unit_ok(const Unit &u, int frames, uint32_t sample_rate) {
    if (frames != GDG_MAX_FRAMES || !gdg_segf_supported(u.type)) return false;
    const int32_t *p = u.params;
    const double sr = (double)sample_rate;
    switch (u.type) {
    case GDG_UNIT_OVERDRIVE: return p[5] == 0;
    case GDG_UNIT_DISTORTION: return p[3] == 0;
    case GDG_UNIT_EXCESS: return p[2] == 0;
    case GDG_UNIT_REVERB:
        return (uint32_t)round(0.19196 * sr) >= (uint32_t)frames && (int)round(0.01348 * sr) - 1 <= 3072 && (int)round(0.00452 * sr) - 1 <= 1024 &&
               (int)round(0.00452 * sr) - 1 >= 1;
    default: return true;
    }
}

/* May the wet path of a reverb be made ahead of the frame's own samples (seg.hip, REVERB_AHEAD)?  The batch block size, and every tap at least
 * a frame back, so that the tapped sums need nothing of the frame itself (rates from 42.7 kHz). */
bool reverb_ahead_ok(int frames, uint32_t sample_rate) {
    return frames == GDG_MAX_FRAMES && (uint32_t)round(0.19196 * (double)sample_rate) >= (uint32_t)frames;
}

/* Fill the device-side description of one non-FIR unit; (re)build its history for this rate / frame size. */
int prepare_unit(gdg_ctx *ctx, Unit &u, int frames, uint32_t sample_rate, gdg_seg_unit &d, int chk) {
    memset(&d, 0, sizeof(d));
    d.type = u.type;
    for (int i = 0; i < GDG_MAX_PARAMS; i++) d.ip[i] = u.params[i];
    const int32_t *p = u.params;
    const double sr = (double)sample_rate;
    int rc = GDG_OK;
    switch (u.type) {
    case GDG_UNIT_COMPRESSOR: {
        d.dp[0] = decibels_to_factor(p[1]);
        d.dp[1] = decibels_to_factor(p[2]);
        d.dp[2] = exp(-20.0 / sr);
        d.dp[3] = 1.0 - d.dp[2];
        rc = follower_tables(ctx, p[0], d.dp[2], d.dp[3], false, &d.tab, chk);
        break;
    }
    case GDG_UNIT_OVERDRIVE:
    case GDG_UNIT_DISTORTION:
    case GDG_UNIT_EXCESS: {
        int os_idx;
        if (u.type == GDG_UNIT_OVERDRIVE) {
            d.dp[0] = decibels_to_factor(p[0] + p[1]);
            d.dp[1] = 0.01 * (double)p[2];
            d.dp[2] = 1.0 - d.dp[1];
            d.dp[3] = decibels_to_factor(p[3]);
            d.ip[4] = p[4];
            os_idx = p[5];
        } else if (u.type == GDG_UNIT_DISTORTION) {
            d.dp[0] = decibels_to_factor(p[0] + p[1]);
            d.dp[3] = decibels_to_factor(p[2]);
            os_idx = p[3];
        } else {
            d.dp[0] = decibels_to_factor(p[0]);
            d.dp[3] = decibels_to_factor(p[1]);
            os_idx = p[2];
        }
        int f = (os_idx == 1) ? 2 : (os_idx == 2) ? 4 : 1;
        d.jp[0] = f;
        if (f > 1) {
            /* one history block per oversampler object (oversamplerTwo / oversamplerFour keep separate state) */
            const size_t len2 = 8 + 76, len4 = 8 + 154;
            rc = ensure_hist(ctx, u, len2 + len4, 1);
            if (rc != GDG_OK) return rc;
            double *base = u.d_hist + (f == 2 ? 0 : len2);
            int which = (f == 2) ? 0 : 1;
            if (u.os_frames[which] != frames) {
                /* bufferPreUpsampling is re-made when the frame size changes (oversampling.go:86-89) */
                if (u.os_frames[which] >= 0) HIP_TRY(ctx, hipMemsetAsync(base, 0, 8 * sizeof(double), ctx->stream));
                u.os_frames[which] = frames;
            }
            d.hist = base;
        }
        break;
    }
    case GDG_UNIT_TONESTACK: {
        static const double freqs[5] = { 20.0, 300.0, 3000.0, 6000.0, 20000.0 };
        double m2pi_sr = -GO_MATH_TWO_PI / sr;
        for (int j = 0; j < 4; j++) {
            d.dp[j] = decibels_to_factor(p[j]);
            d.dp[4 + j] = 1.0 - exp(m2pi_sr * freqs[j]);
            d.dp[8 + j] = 1.0 - exp(m2pi_sr * freqs[j + 1]);
        }
        std::vector<double> key = { 2.0 + 0.001 * chk };
        for (int j = 0; j < 4; j++) { key.push_back(d.dp[4 + j]); key.push_back(d.dp[8 + j]); }
        rc = scan_tables(ctx, key, 4 * L2_SIZE_(chk), [&](double *tab) { for (int j = 0; j < 4; j++) lin2_tab_host(d.dp[4 + j], d.dp[8 + j], tab + j * L2_SIZE_(chk), chk); }, &d.tab);
        break;
    }
    case GDG_UNIT_CABINET: {
        static const double f[7] = { 300.0, 120.0, 80.0, 3000.0, 4000.0, 5000.0, 6000.0 };
        double m2pi_sr = -GO_MATH_TWO_PI / sr;
        for (int j = 0; j < 7; j++) d.dp[j] = 1.0 - exp(m2pi_sr * f[j]);
        std::vector<double> key = { 3.0 + 0.001 * chk };
        for (int j = 0; j < 7; j++) key.push_back(d.dp[j]);
        rc = scan_tables(ctx, key, 7 * LT_SIZE_(chk), [&](double *tab) { for (int j = 0; j < 7; j++) lin_tab_host(false, d.dp[j], 1.0 - d.dp[j], tab + j * LT_SIZE_(chk), chk); }, &d.tab);
        break;
    }
    case GDG_UNIT_CHORUS: {
        double depth = 0.1 * (double)p[0];
        if (depth < 0.0) depth = 0.0; else if (depth > 10.0) depth = 10.0;
        d.dp[0] = depth;
        d.dp[1] = GO_MATH_PI_THOUSANDTH * (double)p[1];
        d.dp[2] = sr;
        int C = (int)floor((0.05 * sr) + 0.5);
        d.jp[0] = C;
        /* the history ring holds the reference's C samples PLUS one frame (the frame is appended BEFORE the delays are read, so every
         * tap -- in the frame or before it -- is one ring access), rounded up to a power of two (index masks) + one guard cell that
         * mirrors cell 0 (a sample pair never wraps).  Re-made (zeroed) when the reference re-makes its buffer: when C changes. */
        size_t cp = 1;
        while (cp < (size_t)C + (size_t)ctx->max_frames) cp <<= 1;
        d.jp[1] = (int)(cp - 1);
        if (u.hist_key != C) rc = zero_is(ctx, u, 0, 1);
        if (rc == GDG_OK) rc = ensure_hist(ctx, u, cp + 1, C);
        d.hist = u.d_hist;
        break;
    }
    case GDG_UNIT_FLANGER:
    case GDG_UNIT_PHASER: {
        double depth = 0.01 * (double)p[0];
        if (depth < 0.0) depth = 0.0; else if (depth > 1.0) depth = 1.0;
        d.dp[0] = depth;
        d.dp[1] = GO_MATH_TWO_PI_HUNDREDTH * (double)p[1];
        d.dp[2] = sr;
        d.dp[3] = 1.0 / sr;
        d.dp[4] = 0.5;
        d.dp[5] = 0.5;
        if (u.type == GDG_UNIT_PHASER) {
            double radians = GO_MATH_DEGREE_TO_RADIANS * (double)p[2];
            d.dp[5] = 0.5 * sin(radians);
            d.dp[4] = 1.0 - fabs(d.dp[5]);
        }
        int C = (int)floor((0.002 * sr) + 0.5);
        d.jp[0] = C;
        if (u.hist_key != C) rc = zero_is(ctx, u, 0, 1);
        if (rc == GDG_OK) rc = ensure_hist(ctx, u, (size_t)C, C);
        d.hist = u.d_hist;
        break;
    }
    case GDG_UNIT_DELAY: {
        double seconds = 0.001 * (double)p[0];
        int D = (int)floor((seconds * sr) + 0.5);
        d.dp[0] = decibels_to_factor(p[1]);
        d.dp[1] = decibels_to_factor(p[2]);
        d.jp[0] = D;
        if (u.hist_key != D) rc = zero_is(ctx, u, 0, 1);
        if (rc == GDG_OK) rc = ensure_hist(ctx, u, (size_t)D, D);
        d.hist = u.d_hist;
        break;
    }
    case GDG_UNIT_RINGMODULATOR: {
        double angular = GO_MATH_TWO_PI * (double)p[0];
        d.dp[0] = angular / sr;
        break;
    }
    case GDG_UNIT_TREMOLO: {
        double frequency = 0.1 * (double)p[0];
        double period_f = sr / frequency;
        uint32_t period = (uint32_t)period_f;
        double phase = 0.01 * (double)p[1];
        uint32_t unatt = (uint32_t)(period_f * phase);
        uint32_t att = period - unatt;
        d.dp[0] = decibels_to_factor(p[2]);
        d.jp[0] = (int)unatt;
        d.jp[1] = (int)att;
        break;
    }
    case GDG_UNIT_SIGNALGENERATOR: {
        d.dp[0] = (0.01 * (double)p[0]) * decibels_to_factor(p[1]);
        double fac_signal_gain = decibels_to_factor(p[5]);
        d.dp[1] = (0.01 * (double)p[4]) * fac_signal_gain;
        d.dp[2] = GO_MATH_TWO_PI * ((double)p[3] / sr);
        break;
    }
    case GDG_UNIT_REVERB: {
        static const double ap_delays[3] = { 0.04204, 0.01348, 0.00452 };
        static const double tap_times[4] = { 0.19196, 0.19996, 0.21596, 0.23204 };
        double wet = 0.01 * (double)p[0];
        d.dp[0] = 1.0 - wet;
        d.dp[1] = 0.5 * wet;
        uint32_t max_index = 0;
        for (int i = 0; i < 4; i++) {
            uint32_t t = (uint32_t)round(tap_times[i] * sr);
            d.jp[i] = (int)t;
            if (t > max_index) max_index = t;
        }
        /* the delay line holds the longest tap PLUS one frame of the batch block size: the in-place reverb of the two-per-CU kernel appends
         * the frame before it reads the taps (seg.hip); the general kernel only sees a longer ring */
        d.jp[4] = (int)max_index + GDG_MAX_FRAMES;
        size_t len = (size_t)max_index + GDG_MAX_FRAMES;
        for (int i = 0; i < 3; i++) {
            int D = (int)round(ap_delays[i] * sr);
            d.jp[5 + i] = D;
            len += (size_t)(D > 1 ? D - 1 : 0);
        }
        /* behind the rings (even offset): the sums  tapped + all-passed  of a frame whose wet path an earlier launch of the call makes (seg.hip, REVERB_AHEAD) */
        len = ((len + 1) & ~(size_t)1) + GDG_MAX_FRAMES;
        d.ip[7] = 0;                    /* 1: an earlier launch of the call makes the unit's wet path (build_plan decides) */
        /* the reference rebuilds every reverb buffer when the sample rate changes (reverb.go:207-271) */
        if (u.hist_key != (long long)sample_rate) rc = zero_is(ctx, u, 0, 4);
        if (rc == GDG_OK) rc = ensure_hist(ctx, u, len, (long long)sample_rate);
        d.hist = u.d_hist;
        break;
    }
    case GDG_UNIT_FUZZ: {
        int f = (p[6] == 1) ? 2 : (p[6] == 2) ? 4 : 1;
        d.jp[0] = f;
        d.dp[0] = 0.01 * (double)p[1];
        d.dp[1] = decibels_to_factor(p[2] + p[3]);
        d.dp[2] = 0.01 * (double)p[4];
        d.dp[3] = 1.0 - d.dp[2];
        d.dp[4] = decibels_to_factor(p[5]);
        /* the follower and the coupling capacitor run at the OVERSAMPLED rate (fuzz.go:42-45, :167-168) */
        double inner_rate = (double)((uint32_t)f * sample_rate);
        d.dp[5] = exp(-20.0 / inner_rate);
        d.dp[6] = 1.0 - d.dp[5];
        rc = follower_tables(ctx, p[0], d.dp[5], d.dp[6], true, &d.tab, GDG_CHK);
        if (rc != GDG_OK) return rc;
        if (f > 1) {
            const size_t len2 = 8 + 76, len4 = 8 + 154;
            rc = ensure_hist(ctx, u, len2 + len4, 1);
            if (rc != GDG_OK) return rc;
            double *base = u.d_hist + (f == 2 ? 0 : len2);
            int which = (f == 2) ? 0 : 1;
            if (u.os_frames[which] != frames) {
                if (u.os_frames[which] >= 0) HIP_TRY(ctx, hipMemsetAsync(base, 0, 8 * sizeof(double), ctx->stream));
                u.os_frames[which] = frames;
            }
            d.hist = base;
        }
        break;
    }
    case GDG_UNIT_AUTOYOY: {
        int32_t level_a = p[1], level_b = p[2];
        double depth_a = 0.0, depth_b = 0.01 * (double)p[3];
        if (level_a > level_b) { std::swap(level_a, level_b); std::swap(depth_a, depth_b); }
        double la = (double)level_a, lb = (double)level_b;
        double sr_inv = 1.0 / sr;
        d.dp[0] = la; d.dp[1] = lb; d.dp[2] = depth_a; d.dp[3] = depth_b;
        d.dp[4] = (depth_b - depth_a) / (lb - la);
        d.dp[5] = exp(-20.0 * sr_inv);
        d.dp[6] = 1.0 - d.dp[5];
        d.dp[7] = sr;
        int C = (int)floor((0.01 * sr) + 0.5);
        d.jp[0] = C;
        if (u.hist_key != C) rc = zero_is(ctx, u, 0, 1);
        if (rc == GDG_OK) rc = ensure_hist(ctx, u, (size_t)C, C);
        d.hist = u.d_hist;
        break;
    }
    case GDG_UNIT_AUTOWAH: {
        int32_t level_a = p[1], level_b = p[2], freq_a = p[3], freq_b = p[4];
        if (level_a > level_b) { std::swap(level_a, level_b); std::swap(freq_a, freq_b); }
        double la = (double)level_a, lb = (double)level_b, fa = (double)freq_a, fb = (double)freq_b;
        d.dp[0] = la; d.dp[1] = lb; d.dp[2] = fa; d.dp[3] = fb;
        d.dp[4] = (fb - fa) / (lb - la);
        d.dp[5] = exp(-20.0 / sr);
        d.dp[6] = 1.0 - d.dp[5];
        d.dp[7] = sr;
        break;
    }
    case GDG_UNIT_BANDPASS: {
        static const int orders[4] = { 2, 4, 6, 8 };
        int half = (p[0] >= 0 && p[0] < 4) ? orders[p[0]] >> 1 : 0;
        int32_t fa = p[1], fb = p[2];
        if (fa > fb) std::swap(fa, fb);
        double m2pi_sr = -GO_MATH_TWO_PI / sr;
        d.dp[0] = 1.0 - exp(m2pi_sr * (double)fa);
        d.dp[1] = 1.0 - exp(m2pi_sr * (double)fb);
        {
            std::vector<double> key = { 4.0, d.dp[0], d.dp[1] };
            rc = scan_tables(ctx, key, L2_SIZE, [&](double *tab) { lin2_tab_host(d.dp[0], d.dp[1], tab, GDG_CHK); }, &d.tab);
            if (rc != GDG_OK) return rc;
        }
        d.jp[0] = half;
        if (u.bp_half_order != half) {
            /* bandpass.go:40-49: both capacitor slices are re-made when the order changes */
            if (u.bp_half_order >= 0) HIP_TRY(ctx, hipMemsetAsync(u.d_ds, 0, 8 * sizeof(double), ctx->stream));
            u.bp_half_order = half;
        }
        break;
    }
    case GDG_UNIT_OCTAVER: {
        for (int i = 0; i < 6; i++) d.dp[i] = decibels_to_factor(p[1 + i]);
        d.dp[6] = exp(-20.0 / sr);
        d.dp[7] = 1.0 - d.dp[6];
        rc = follower_tables(ctx, p[0], d.dp[6], d.dp[7], true, &d.tab, GDG_CHK);
        break;
    }
    case GDG_UNIT_NOISEGATE: {
        d.dp[0] = decibels_to_factor(p[0]);
        d.dp[1] = decibels_to_factor(p[1]);
        double hold_seconds = 0.001 * (double)p[2];
        d.jp[0] = (int)(uint32_t)floor((hold_seconds * sr) + 0.5);
        d.jp[1] = (p[0] < p[1]) ? 1 : 0;
        break;
    }
    default:
        return fail(ctx, GDG_ERR_UNSUPPORTED, "unit type %d has no HIP implementation yet", u.type);
    }
    if (rc != GDG_OK) return rc;
    d.ds = u.d_ds;
    d.is = u.d_is;
    return GDG_OK;
}

int fir_tables(gdg_ctx *ctx, int P, double2 **tw, double2 **tw2) {
    auto it = ctx->fir_tables.find(P);
    if (it == ctx->fir_tables.end()) {
        double2 *a = nullptr, *b = nullptr;
        HIP_TRY(ctx, gdg_fir_tables_create(P, &a, &b));
        it = ctx->fir_tables.emplace(P, std::make_pair(a, b)).first;
    }
    *tw = it->second.first;
    *tw2 = it->second.second;
    return GDG_OK;
}

/* transform half size for a frame of `frames` samples: the next power of two, at least GDG_MIN_FIR_FRAMES */
int fir_transform_size(int frames) {
    int P = GDG_MIN_FIR_FRAMES;
    while (P < frames) P <<= 1;
    return P;
}

/* filter.Process walks the frame in blocks of nextpow2(L) samples but counts them on nextpow2(N): when N is not a power of two a
 * block can start beyond the frame and the reference panics on the slice bounds (filter/filter.go:370-382, :443-453).  Such a
 * (frame size, filter length) pair is rejected instead of replicated. */
static bool reference_panics(int frames, int taps) {
    if (taps <= 0 || frames <= 0) return false;
    uint64_t n_power = 1, block = 1;
    while (n_power < (uint64_t)frames) n_power <<= 1;
    while (block < (uint64_t)taps) block <<= 1;
    uint64_t blocks = n_power / block + ((n_power % block) ? 1 : 0);
    return blocks > 0 && (blocks - 1) * block > (uint64_t)frames;
}

/* IR spectra of `taps` for frames of `hop` samples: reuse a live copy of the same taps at the same partition size, else build one */
static int fir_spectra(gdg_ctx *ctx, Unit &u, int hop, int P, int K) {
    const int L = (int)u.taps.size();
    size_t spec = (size_t)K * (size_t)P * sizeof(double2);
    uint64_t key = 1469598103934665603ull;                              /* FNV-1a style over the taps' 64-bit patterns, P, hop and L (byte-wise
                                                                         * it cost 0.5 ms per 65536-tap filter: half a second for 1024 of them) */
    {
        for (size_t i = 0; i < u.taps.size(); i++) { uint64_t w; memcpy(&w, &u.taps[i], sizeof(w)); key ^= w; key *= 1099511628211ull; key ^= key >> 29; }
        key ^= (uint64_t)P; key *= 1099511628211ull;
        key ^= (uint64_t)hop; key *= 1099511628211ull;
        key ^= (uint64_t)L; key *= 1099511628211ull;
    }
    u.H.reset();
    if (ctx->share_spectra) {
        auto range = ctx->spectra.equal_range(key);
        for (auto it = range.first; it != range.second;) {
            std::shared_ptr<SharedSpectra> sp = it->second.lock();
            if (!sp) { it = ctx->spectra.erase(it); continue; }
            if (sp->P == P && sp->hop == hop && sp->taps == u.taps) { u.H = sp; break; }      /* compared in full: a hash match alone is not trusted */
            ++it;
        }
    }
    if (u.H) return GDG_OK;
    auto sp = std::make_shared<SharedSpectra>();
    sp->taps = u.taps;
    sp->P = P;
    sp->K = K;
    sp->hop = hop;
    sp->arena = &ctx->arena;
    HIP_TRY(ctx, ctx->arena.alloc((void **)&sp->d_H, spec));
    if (L > 0) ctx->pending_ir.push_back(sp);                           /* transformed with the plan's other new filters: flush_ir */
    else HIP_TRY(ctx, hipMemsetAsync(sp->d_H, 0, spec, ctx->stream));   /* filter.Empty: one all-zero partition */
    u.H = sp;
    if (ctx->share_spectra) ctx->spectra.emplace(key, sp);
    return GDG_OK;
}

/* The IR spectra of every filter the plan under construction brought in, in a few large launches: a 512-channel context with private IRs
 * has 1024 of them, and one upload + launch + synchronise + release each cost 1.1 ms a piece (1.2 s before the first frame).  Filters of
 * the same transform size go together, at most ~256 MiB of zero-padded taps per round: partition k = taps [k hop, (k + 1) hop) padded to P. */
static int flush_ir_body(gdg_ctx *ctx, const std::vector<std::shared_ptr<SharedSpectra>> &todo);
static int flush_ir(gdg_ctx *ctx) {
    if (ctx->pending_ir.empty()) return GDG_OK;
    int rc = flush_ir_body(ctx, ctx->pending_ir);
    if (rc == GDG_OK) ctx->pending_ir.clear();          /* on failure the list stays: the caller marks these filters for a fresh start */
    return rc;
}
static int flush_ir_body(gdg_ctx *ctx, const std::vector<std::shared_ptr<SharedSpectra>> &todo) {
    std::map<int, std::vector<SharedSpectra *>> by_P;
    for (auto &sp : todo) by_P[sp->P].push_back(sp.get());
    for (auto &kv : by_P) {
        const int P = kv.first;
        double2 *tw, *tw2;
        int rc = fir_tables(ctx, P, &tw, &tw2);
        if (rc != GDG_OK) return rc;
        const size_t budget = ((size_t)256 << 20) / ((size_t)P * sizeof(double));       /* partitions per round */
        size_t at = 0;
        while (at < kv.second.size()) {
            size_t end = at, parts = 0;
            while (end < kv.second.size() && (parts == 0 || parts + (size_t)kv.second[end]->K <= budget)) parts += (size_t)kv.second[end++]->K;
            std::vector<double> padded(parts * (size_t)P, 0.0);
            std::vector<gdg_fir_irjob> jobs(parts);
            struct Temps {
                DevArena &a; hipStream_t st; void *p = nullptr, *q = nullptr;
                ~Temps() { hipStreamSynchronize(st); a.release(p); a.release(q); }
            } tmp{ ctx->arena, ctx->stream };
            HIP_TRY(ctx, ctx->arena.alloc(&tmp.p, padded.size() * sizeof(double)));
            HIP_TRY(ctx, ctx->arena.alloc(&tmp.q, jobs.size() * sizeof(gdg_fir_irjob)));
            double *d_taps = static_cast<double *>(tmp.p);
            size_t j = 0;
            for (size_t i = at; i < end; i++) {
                const SharedSpectra &sp = *kv.second[i];
                const int L = (int)sp.taps.size();
                for (int k = 0; k < sp.K; k++, j++) {
                    const int n = std::min(sp.hop, L - k * sp.hop);
                    if (n > 0) memcpy(padded.data() + j * (size_t)P, sp.taps.data() + (size_t)k * sp.hop, (size_t)n * sizeof(double));
                    memset(&jobs[j], 0, sizeof(gdg_fir_irjob));
                    jobs[j].a = d_taps + j * (size_t)P;
                    jobs[j].out = sp.d_H + (size_t)k * P;
                }
            }
            HIP_TRY(ctx, hipMemcpyAsync(d_taps, padded.data(), padded.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(tmp.q, jobs.data(), jobs.size() * sizeof(gdg_fir_irjob), hipMemcpyHostToDevice, ctx->stream));
            /* 1/(2P): the inverse real transform's scale, folded into the IR spectra */
            HIP_TRY(ctx, gdg_launch_fir_ir(P, static_cast<const gdg_fir_irjob *>(tmp.q), (int)parts, 1.0 / (2.0 * (double)P), tw, tw2, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                     /* `padded` and `jobs` are locals */
            at = end;
        }
    }
    return GDG_OK;
}

/* (Re)build the partitioned spectra of one power amp for frames of `hop` samples: IR partitions of `hop` taps, transforms of 2 P
 * points with P = fir_transform_size(hop) (hop == P for the power-of-two frame sizes).
 *   - new filter / new sample rate / reset: the convolution state starts from zero (poweramp.go:131-203);
 *   - frame size changed while the filter is live: the reference's tail and transform sizes depend on L only, so any sequence of
 *     frame sizes is one continuous convolution (filter.go:370-428).  Here the partition size follows the frame, so the delay
 *     line is RE-PARTITIONED: its slots are transformed back to the last (K + 1) hop input samples (raw inverse, ~1e-16), which
 *     are re-cut into frames of the new size and transformed into the new delay line.  Samples older than that only ever meet
 *     zero-padded taps, so the continuation is exact.  Happens once per change, never in the steady state. */
static int prepare_fir(gdg_ctx *ctx, Unit &u, int hop, uint32_t sample_rate) {
    const int P = fir_transform_size(hop);
    if (u.fir_sr != sample_rate) {
        /* poweramp.go:191-203: a sample-rate change recompiles the filter, i.e. fresh state */
        u.fir_sr = sample_rate;
        u.fir_dirty = true;
        u.fir_live = false;
    }
    const int W = (hop == GDG_MAX_FRAMES) ? ctx->window : 1;      /* time blocking exists for the batch block size only */
    if (!u.fir_dirty && u.fir_hop == hop && u.fir_R == u.fir_K + W - 1) return GDG_OK;
    int L = (int)u.taps.size();
    if (reference_panics(hop, L))
        return fail(ctx, GDG_ERR_UNSUPPORTED, "frame size %d with a %d-tap filter: the reference panics on this pair (filter/filter.go:443-453: a block of "
                    "nextpow2(L) samples starts beyond the frame); rejected, not replicated", hop, L);
    int K = (L + hop - 1) / hop;
    if (K < 1) K = 1;                 /* filter.Empty: one all-zero partition => zeros out */
    const int R = K + W - 1;
    /* a live delay line moves into the new layout when the frame size OR the ring size (gdg_ctx_set_window) changes */
    const bool carry = !u.fir_dirty && u.fir_live && (u.fir_hop != hop || u.fir_R != R) && L > 0 && u.d_fdl && u.d_pos;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    /* the old state, kept until the new delay line is built */
    double *o_prev = u.d_prev; double2 *o_fdl = u.d_fdl, *o_Y = u.d_Y; int *o_pos = u.d_pos;
    const int K1 = u.fir_K, P1 = u.fir_P, hop1 = u.fir_hop, R1 = u.fir_R;
    DevArena &arena = ctx->arena;
    auto free_old = [&]() { arena.release(o_prev); arena.release(o_fdl); arena.release(o_Y); arena.release(o_pos); };
    u.d_prev = nullptr; u.d_fdl = nullptr; u.d_Y = nullptr; u.d_pos = nullptr;
    double *d_old_hist = nullptr, *d_new_hist = nullptr;
    void *d_jobs = nullptr;
    size_t old_len = 0;
    int rc = GDG_OK;
    auto body = [&]() -> int {
        double2 *tw, *tw2;
        if (carry) {
            /* 1. the last (K1 + 1) hop1 input samples out of the old delay line, oldest first */
            int pos = 0;
            HIP_TRY(ctx, hipMemcpy(&pos, o_pos, sizeof(int), hipMemcpyDeviceToHost));
            old_len = (size_t)(K1 + 1) * (size_t)hop1;
            HIP_TRY(ctx, arena.alloc((void **)&d_old_hist, old_len * sizeof(double)));
            std::vector<gdg_fir_rawjob> jobs((size_t)K1);
            for (int j = 0; j < K1; j++) {
                int m = K1 - 1 - j;                                   /* frame t - m, t = the latest one, sits in slot (pos - 1 - m) mod R1 */
                int slot = (((pos - 1 - m) % R1) + R1) % R1;
                jobs[(size_t)j].Y = o_fdl + (size_t)slot * P1;
                jobs[(size_t)j].first = (j == 0) ? d_old_hist : nullptr;
                jobs[(size_t)j].second = d_old_hist + (size_t)(j + 1) * hop1;
                jobs[(size_t)j].hop = hop1;
            }
            HIP_TRY(ctx, arena.alloc(&d_jobs, jobs.size() * sizeof(gdg_fir_rawjob)));
            HIP_TRY(ctx, hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(gdg_fir_rawjob), hipMemcpyHostToDevice, ctx->stream));
            int r = fir_tables(ctx, P1, &tw, &tw2);
            if (r != GDG_OK) return r;
            HIP_TRY(ctx, gdg_launch_fir_raw_inv(P1, (const gdg_fir_rawjob *)d_jobs, K1, 1.0 / (2.0 * (double)P1), tw, tw2, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            arena.release(d_jobs); d_jobs = nullptr;
        }
        size_t spec = (size_t)R * (size_t)P * sizeof(double2);
        HIP_TRY(ctx, arena.alloc_zeroed((void **)&u.d_prev, 2 * (size_t)P * sizeof(double), ctx->stream));
        HIP_TRY(ctx, arena.alloc_zeroed((void **)&u.d_fdl, spec, ctx->stream));
        HIP_TRY(ctx, arena.alloc((void **)&u.d_Y, (size_t)W * (size_t)P * sizeof(double2)));
        HIP_TRY(ctx, arena.alloc_zeroed((void **)&u.d_pos, sizeof(int), ctx->stream));
        int r = fir_spectra(ctx, u, hop, P, K);
        if (r != GDG_OK) return r;
        if (carry) {
            /* 2. the newest K hop samples, re-cut into K frames of the new size (zeros where the old line does not reach) */
            size_t new_len = (size_t)K * (size_t)hop;
            HIP_TRY(ctx, arena.alloc((void **)&d_new_hist, new_len * sizeof(double)));
            HIP_TRY(ctx, hipMemsetAsync(d_new_hist, 0, new_len * sizeof(double), ctx->stream));
            size_t n = std::min(old_len, new_len);
            HIP_TRY(ctx, hipMemcpyAsync(d_new_hist + (new_len - n), d_old_hist + (old_len - n), n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            /* 3. slot f = spectrum of [frame f - 1 | frame f], f = 1 .. K - 1; the next frame goes to slot K mod R */
            if (K > 1) {
                std::vector<gdg_fir_irjob> jobs((size_t)(K - 1));
                for (int f = 1; f < K; f++) {
                    jobs[(size_t)(f - 1)].a = d_new_hist + (size_t)(f - 1) * hop;
                    jobs[(size_t)(f - 1)].b = d_new_hist + (size_t)f * hop;
                    jobs[(size_t)(f - 1)].hop = hop;
                    jobs[(size_t)(f - 1)].out = u.d_fdl + (size_t)f * P;
                }
                HIP_TRY(ctx, arena.alloc(&d_jobs, jobs.size() * sizeof(gdg_fir_irjob)));
                HIP_TRY(ctx, hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(gdg_fir_irjob), hipMemcpyHostToDevice, ctx->stream));
                r = fir_tables(ctx, P, &tw, &tw2);
                if (r != GDG_OK) return r;
                HIP_TRY(ctx, gdg_launch_fir_ir(P, (const gdg_fir_irjob *)d_jobs, K - 1, 1.0, tw, tw2, ctx->stream));
            }
            /* 4. the overlap-save history = the newest frame, where the forward transform of frame counter K looks for it */
            HIP_TRY(ctx, hipMemcpyAsync(u.d_prev + (size_t)((K + 1) & 1) * P, d_new_hist + (size_t)(K - 1) * hop, (size_t)hop * sizeof(double),
                                        hipMemcpyDeviceToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(u.d_pos, &K, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        return GDG_OK;
    };
    rc = body();
    if (rc != GDG_OK || carry) hipStreamSynchronize(ctx->stream);          /* nothing in flight reads what is released below */
    arena.release(d_old_hist); arena.release(d_new_hist); arena.release(d_jobs);
    free_old();
    if (rc != GDG_OK) { u.fir_dirty = true; u.fir_live = false; return rc; }
    u.fir_P = P;
    u.fir_K = K;
    u.fir_R = R;
    u.fir_hop = hop;
    u.fir_dirty = false;
    u.fir_live = carry;
    return GDG_OK;
}

struct Op { bool is_fir; std::vector<int> handles; };

/* `active`: the channels taking part in this call; row i of d_in / d_out belongs to channel active[i] */
int build_plan(gdg_ctx *ctx, const std::vector<int> &active, const double *d_in, double *d_out, int frames, uint32_t sample_rate,
                      int stride, int stride_out, bool rows_by_channel, int G, const std::vector<size_t> &bounds) {
    const int nch = ctx->nch;
    int ptrace = 0;
    { const char *e = getenv("GDG_PLAN_TRACE"); ptrace = e ? atoi(e) : 0; }        /* read per plan: a test switches it on */
    auto pnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_fir = 0.0, t_unit = 0.0;
    const double t_plan0 = pnow();
    join_groups(ctx);                 /* a new plan replaces descriptors (and possibly unit state) the group streams may still be reading */
    join_premac(ctx, false);          /* ... and the sums made ahead belong to the old plan's next frame */
    /* Scan tables live as long as some plan's descriptors point at them -- there is one plan, this one.  A caller that sweeps a parameter
     * through thousands of values would let the cache grow without bound (12 KB per tone-stack setting): past the limit everything is
     * dropped once the work in flight has drained, and this plan re-makes the few tables it needs. */
    {
        long limit = ctx->scan_tables_max;                         /* gdg_ctx_set_option "scan_tables_max": a test lowers it */
        if (limit < 1) limit = 1;
        if ((long)ctx->scan_tabs.size() > limit) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            for (auto &kv : ctx->scan_tabs) ctx->arena.release(kv.second);
            ctx->scan_tabs.clear();
        }
    }
    /* channel groups: contiguous runs of `active`, group g = [bounds[g], bounds[g + 1]) (equal shares unless the caller weights them) */
    std::vector<int> group_of((size_t)nch, 0);
    for (int g = 0; g < G; g++)
        for (size_t i = bounds[(size_t)g]; i < bounds[(size_t)g + 1]; i++) group_of[(size_t)active[i]] = g;
    ctx->plan_groups = G;
    /* per channel: ops placed on a common grid of slots: 2k = segment k, 2k+1 = FIR k */
    std::map<int, std::vector<std::pair<int, Op>>> by_slot;           /* slot -> (channel, op) */
    std::vector<int> n_ops((size_t)nch, 0);
    std::vector<int> row_of((size_t)nch, -1);
    for (size_t i = 0; i < active.size(); i++) row_of[(size_t)active[i]] = rows_by_channel ? active[i] : (int)i;
    bool any_fir = false;
    /* Slots: the ops of all channels are laid on one grid so that one launch takes the same op of every channel that has it.  Power amp number k
     * of a channel sits at slot_key(k, 63, fir); what stands between power amps k - 1 and k is segment pieces (kind 0) and -- when the channels
     * are too few to fill the chip -- oversampled shapers as launches of their own (kinds 2 / 3 for 2 x / 4 x: seg.hip os_tiles_kernel), numbered
     * j = 0, 1, 2 ... in the order they come.  A channel without such a shaper has ONE piece (j = 0), as before. */
    enum { K_SEG = 0, K_FIR = 1, K_OS2 = 2, K_OS4 = 3 };
    auto slot_key = [](int k, int j, int kind) { return (k * 64 + j) * 4 + kind; };
    const bool os_tiles = ctx->seg_os_tiles_max > 0 && (int)active.size() <= ctx->seg_os_tiles_max && frames == GDG_MAX_FRAMES;
    for (int c : active) {
        std::vector<int> seg;
        int k = 0, j = 0, count = 0;
        auto close_seg = [&]() { if (!seg.empty()) { by_slot[slot_key(k, j, K_SEG)].push_back({ c, Op{ false, seg } }); seg.clear(); count++; j++; } };
        for (auto &s : ctx->chains[(size_t)c]) {
            if (s.bypass) continue;                                   /* signal.go:390-401 */
            Unit &u = ctx->units[(size_t)s.handle];
            if (u.type == GDG_UNIT_POWERAMP) {
                close_seg();
                by_slot[slot_key(k, 63, K_FIR)].push_back({ c, Op{ true, { s.handle } } });
                count++;
                k++;
                j = 0;
                any_fir = true;
            } else {
                if (!gdg_seg_supported(u.type)) return fail(ctx, GDG_ERR_UNSUPPORTED, "unit type %d has no HIP implementation yet", u.type);
                const bool shaper = u.type == GDG_UNIT_OVERDRIVE || u.type == GDG_UNIT_DISTORTION || u.type == GDG_UNIT_EXCESS;
                const int os_index = !shaper ? 0 : u.params[u.type == GDG_UNIT_OVERDRIVE ? 5 : (u.type == GDG_UNIT_DISTORTION ? 3 : 2)];      /* 0 none, 1 "2", 2 "4" */
                if (os_tiles && os_index > 0 && j < 60) {
                    close_seg();
                    by_slot[slot_key(k, j, os_index == 1 ? K_OS2 : K_OS4)].push_back({ c, Op{ false, { s.handle } } });
                    count++;
                    j++;
                } else seg.push_back(s.handle);
            }
        }
        close_seg();
        if (count == 0) { by_slot[slot_key(0, 0, K_SEG)].push_back({ c, Op{ false, {} } }); count = 1; }     /* empty chain: copy */
        n_ops[(size_t)c] = count;
    }
    (void)any_fir;
    /* counters of the WAVE launches: tickets per (segment step, channel group), then one frame counter per unit that sits in a segment */
    {
        /* 8 cells per unit in a segment + a flag and an arrival counter per oversampled shaper that is a launch of its own (each such unit is one descriptor of one step) */
        const size_t need = (size_t)GDG_WAVE_STEPS * GDG_WAVE_GROUPS + 10 * ctx->units.size() + (size_t)nch + 64;
        if (need > ctx->d_wave_cap) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(ctx->d_wave);
            ctx->d_wave = nullptr;
            ctx->d_wave_cap = need * 2;
            HIP_TRY(ctx, hipMalloc((void **)&ctx->d_wave, ctx->d_wave_cap * sizeof(int)));
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_wave, 0, ctx->d_wave_cap * sizeof(int), ctx->stream));
        }
    }
    size_t wave_next = (size_t)GDG_WAVE_STEPS * GDG_WAVE_GROUPS;
    int seg_steps = 0;
    /* blob layout: [step 0 descs][step 1 descs]...[seg units] */
    std::vector<gdg_seg_unit> seg_units;
    std::vector<std::pair<int, int>> tile_conditional;     /* (step, index into seg_units) of reverbs in steps that could run tiled: they can if their wet path is made ahead */
    std::vector<std::vector<int>> ahead_lists; /* per step: reverbs of later steps (indices into seg_units) whose wet path the step's launch makes (seg.hip REVERB_AHEAD) */
    int ahead_host = -1;                       /* the general-kernel segment step, among those already laid out, that hosts them */
    double ahead_host_weight = 0.0;
    std::vector<std::vector<gdg_seg_chan>> seg_descs;
    std::vector<std::vector<gdg_fir_chan>> fir_descs;
    std::vector<int> done((size_t)nch, 0);
    std::vector<const double *> cur((size_t)nch);
    for (int c : active) cur[(size_t)c] = d_in + (size_t)row_of[(size_t)c] * stride;
    ctx->steps.clear();
    ctx->plan_unit_slot.assign(ctx->units.size(), -1);
    ctx->patch_units.clear();                  /* this plan reads every unit's current parameters */
    ctx->plan_unit_fast.assign(ctx->units.size(), 0);
    ctx->plan_unit_fast_ok.assign(ctx->units.size(), 0);
    ctx->plan_fir_steps = 0;
    for (auto &kv : by_slot) if ((kv.first & 3) == K_FIR) ctx->plan_fir_steps++;
    for (auto &kv : by_slot) {
        const int kind = kv.first & 3;
        const bool is_fir = kind == K_FIR, is_os = kind == K_OS2 || kind == K_OS4;
        std::vector<gdg_seg_chan> sd;
        std::vector<gdg_fir_chan> fd;
        /* a segment step goes to the two-per-CU kernel when EVERY unit of EVERY channel in it can (one launch per step) */
        const int n_act = (int)active.size();
        const int fast_min = ctx->seg_fast_min;
        bool step_fast = !is_fir && !is_os && ctx->seg_fast && frames == GDG_MAX_FRAMES && n_act >= fast_min;
        if (step_fast)
            for (auto &entry : kv.second)
                for (int h : entry.second.handles) if (!segf_unit_ok(ctx->units[(size_t)h], frames, sample_rate)) { step_fast = false; break; }
        for (auto &entry : kv.second) {
            int c = entry.first;
            Op &op = entry.second;
            bool last = (done[(size_t)c] + 1 == n_ops[(size_t)c]);
            double *dst;
            if (last) dst = d_out + (size_t)row_of[(size_t)c] * stride_out;
            else dst = ((done[(size_t)c] & 1) ? ctx->d_w1 : ctx->d_w0) + (size_t)c * ctx->w_stride;
            if (is_fir) {
                Unit &u = ctx->units[(size_t)op.handles[0]];
                const double tq = pnow();
                int rc = prepare_fir(ctx, u, frames, sample_rate);
                t_fir += pnow() - tq;
                if (rc != GDG_OK) return rc;
                gdg_fir_chan f;
                memset(&f, 0, sizeof(f));
                f.src = cur[(size_t)c]; f.dst = dst; f.prev = u.d_prev; f.fdl = u.d_fdl; f.H = u.H->d_H; f.Y = u.d_Y;
                f.pos = u.d_pos; f.K = u.fir_K; f.R = u.fir_R; f.hop = frames;
                f.flags = (done[(size_t)c] == 0 ? GDG_SRC_IS_INPUT : 0) | (last ? GDG_DST_IS_OUTPUT : 0);
                fd.push_back(f);
                u.fir_live = true;
            } else {
                gdg_seg_chan s;
                memset(&s, 0, sizeof(s));
                s.src = cur[(size_t)c]; s.dst = dst;
                s.flags = (done[(size_t)c] == 0 ? GDG_SRC_IS_INPUT : 0) | (last ? GDG_DST_IS_OUTPUT : 0);
                s.scratch = ctx->d_scratch + (size_t)c * ctx->max_frames;
                s.unit_begin = (int)seg_units.size();
                s.unit_count = (int)op.handles.size();
                s.wave = ctx->d_wave + wave_next;
                wave_next += 8 * op.handles.size();                     /* GDG_WAVE_CELLS counters per unit (seg.hip) */
                {   /* which units meet their predecessor frame in a WAVE launch, and whether their stores are write-through there (seg.hip, wt) */
                    unsigned mask = 0;
                    for (size_t ui = 0; ui < op.handles.size(); ui++) {
                        const Unit &wu = ctx->units[(size_t)op.handles[ui]];
                        const bool shaper = wu.type == GDG_UNIT_OVERDRIVE || wu.type == GDG_UNIT_DISTORTION || wu.type == GDG_UNIT_EXCESS;
                        const int os_param = wu.type == GDG_UNIT_OVERDRIVE ? 5 : (wu.type == GDG_UNIT_DISTORTION ? 3 : 2);
                        if (shaper && wu.params[os_param] == 0) continue;                       /* memoryless: no state, no meeting */
                        if (ui < 15) mask |= 1u << ui;
                        if (ui < 15 && (wu.type == GDG_UNIT_COMPRESSOR || wu.type == GDG_UNIT_TONESTACK || wu.type == GDG_UNIT_CABINET))
                            mask |= 1u << (16 + ui);                                              /* all their state is a few cells, read past the L1 */
                        const bool write_through = wu.type == GDG_UNIT_COMPRESSOR || wu.type == GDG_UNIT_TONESTACK || wu.type == GDG_UNIT_CABINET ||
                                                   wu.type == GDG_UNIT_CHORUS || wu.type == GDG_UNIT_REVERB || (shaper && !step_fast);
                        if (!write_through) mask |= 1u << 31;
                    }
                    s.wave_mask = (int)mask;
                }
                for (int h : op.handles) {
                    gdg_seg_unit du;
                    const double tq = pnow();
                    int rc = prepare_unit(ctx, ctx->units[(size_t)h], frames, sample_rate, du, step_fast ? GDG_CHK_FAST : GDG_CHK);
                    t_unit += pnow() - tq;
                    if (rc != GDG_OK) return rc;
                    /* both reverbs that append the frame BEFORE they tap (two-per-CU kernel; general kernel in a WAVE launch) rely on a delay
                     * line exactly one batch frame longer than the longest tap (seg.hip): a change of one side without the other stops here */
                    if (du.type == GDG_UNIT_REVERB && du.jp[4] != std::max(std::max(du.jp[0], du.jp[1]), std::max(du.jp[2], du.jp[3])) + GDG_MAX_FRAMES)
                        return fail(ctx, GDG_ERR_INVALID, "reverb delay line of %d cells, expected the longest tap + %d", du.jp[4], GDG_MAX_FRAMES);
                    /* a reverb behind an earlier general-kernel segment launch of the same call: that launch makes its wet path beside its own
                     * channels (extra workgroups: the channels are too few to fill the chip), the unit itself only mixes */
                    if (du.type == GDG_UNIT_REVERB && !step_fast && !is_os && ahead_host >= 0 && ctx->seg_reverb_ahead_max > 0 && n_act <= std::max(ctx->seg_reverb_ahead_max, 127) &&
                        reverb_ahead_ok(frames, sample_rate)) {
                        du.ip[7] = 1;
                        ahead_lists[(size_t)ahead_host].push_back((int)seg_units.size());
                    }
                    ctx->plan_unit_slot[(size_t)h] = (int)seg_units.size();
                    ctx->plan_unit_fast[(size_t)h] = step_fast ? 1 : 0;
                    ctx->plan_unit_fast_ok[(size_t)h] = segf_unit_ok(ctx->units[(size_t)h], frames, sample_rate) ? 1 : 0;
                    seg_units.push_back(du);
                }
                sd.push_back(s);
            }
            cur[(size_t)c] = dst;
            done[(size_t)c]++;
        }
        StepDesc st;
        st.is_fir = is_fir;
        st.os_factor = is_os ? (kind == K_OS2 ? 2 : 4) : 0;
        st.fast = step_fast;
        st.n = is_fir ? (int)fd.size() : (int)sd.size();
        for (auto &d : sd) if ((unsigned)d.wave_mask >> 31) st.wave_release = true;
        if (!is_fir && !is_os && !step_fast && frames == GDG_MAX_FRAMES && n_act <= ctx->seg_tile_max && !sd.empty()) {
            /* a channel's frame on two workgroups in per-frame calls: unit types, no oversampling, the exchange ids of a segment within the area */
            st.tile_ok = true;
            for (auto &entry : kv.second) {
                int xids = 0;
                for (int h : entry.second.handles) {
                    const Unit &tu = ctx->units[(size_t)h];
                    const bool shaper = tu.type == GDG_UNIT_OVERDRIVE || tu.type == GDG_UNIT_DISTORTION || tu.type == GDG_UNIT_EXCESS;
                    const int os_param = tu.type == GDG_UNIT_OVERDRIVE ? 5 : (tu.type == GDG_UNIT_DISTORTION ? 3 : 2);
                    if (!gdg_segt_supported(tu.type) || (shaper && tu.params[os_param] != 0)) st.tile_ok = false;
                    if (gdg_segt_supported(tu.type) == 2) tile_conditional.push_back(std::make_pair((int)ctx->steps.size(), ctx->plan_unit_slot[(size_t)h]));   /* a reverb: only as a mix (below) */
                    xids += gdg_segt_exchanges(tu.type);
                }
                if (xids > 32 || entry.second.handles.size() > 16 || entry.second.handles.empty()) st.tile_ok = false;
            }
        }
        st.offset = 0;
        if (!is_fir && seg_steps < GDG_WAVE_STEPS && G <= GDG_WAVE_GROUPS) st.wave_tickets = GDG_WAVE_GROUPS * seg_steps++;
        if (is_os) { st.os_flags = (int)wave_next; wave_next += sd.size(); st.os_arrive = (int)wave_next; wave_next += sd.size(); }    /* a flag and an arrival counter per channel of the launch (os_tiles_kernel) */
        if (is_os && ctx->seg_os_prefix && !ctx->steps.empty() && !ctx->steps.back().is_fir && !ctx->steps.back().os_factor && !ctx->steps.back().fast &&
            ctx->steps.back().n == (int)sd.size() && seg_descs.back().size() == sd.size() && G == 1) {
            /* the step in front of this launch: the same channels in the same order, each with ONE unit, a compressor, feeding this shaper */
            bool lone = true;
            for (size_t i = 0; i < sd.size(); i++) {
                const gdg_seg_chan &pc = seg_descs.back()[i];
                if (pc.unit_count != 1 || seg_units[(size_t)pc.unit_begin].type != GDG_UNIT_COMPRESSOR || pc.dst != sd[i].src) lone = false;
            }
            if (lone) {
                st.os_prefix_step = (int)ctx->steps.size() - 1;
                ctx->steps.back().absorbed_per_frame = true;
                /* a step that per-frame calls do not launch cannot carry other steps' extra workgroups (nothing has been given to it yet: it is the step right in front) */
                if (ahead_host == st.os_prefix_step) { ahead_host = -1; ahead_host_weight = 0.0; }
            }
        }
        /* descriptors are in `active` order, so every channel group owns one contiguous run of them */
        st.group_range.assign((size_t)G, std::make_pair(0, 0));
        {
            int pos = 0;
            for (auto &entry : kv.second) {
                auto &r = st.group_range[(size_t)group_of[(size_t)entry.first]];
                if (r.second == 0) r.first = pos;
                r.second++;
                pos++;
            }
        }
        if (is_fir) {
            std::vector<const void *> hp;
            for (auto &f : fd) hp.push_back(f.H);
            std::sort(hp.begin(), hp.end());
            st.shared_spectra = std::adjacent_find(hp.begin(), hp.end()) != hp.end();
            /* the terms k >= 1 ahead of the frame (premac): the split launch shape of few channels, one group, batch frames, every channel K >= 2 */
            const bool split = ctx->fir_fused < 0 ? (st.n <= fir_split_limit(ctx)) : (ctx->fir_fused == 0);
            long partitions = 0;
            for (auto &f : fd) partitions += f.K;
            const long premac_min = ctx->plan_fir_steps >= 2 ? std::min(ctx->fir_premac_min, ctx->fir_premac_min_two) : ctx->fir_premac_min;
            st.premac_ok = ctx->fir_premac != 0 && split && G == 1 && frames == GDG_MAX_FRAMES && partitions >= premac_min;
            for (auto &f : fd) if (f.K < 2 || f.hop != frames) st.premac_ok = false;
            /* LDS the premac's workgroups ask for and never touch: such a workgroup does not fit on a CU beside a general or tile segment
             * workgroup (159 KiB), and at most one fits beside a two-per-CU one (80 KiB), so the sums run on the CUs the segments leave idle
             * and not among their waves.  Pays where the sums are neither a sliver nor the whole frame -- 5 .. 32 partitions per channel:
             * 48 channels x 65536 taps 124.5 -> 119.9 us per frame, 128 channels 195.3 -> 185.2; 128 x 32768 taps 168 -> 172 and 64 x 1048576
             * taps 305 -> 320 WITH it, hence the bounds (profiles/premac_loads_ab_r06.txt) */
            const long per_channel = st.n > 0 ? partitions / st.n : 0;
            st.premac_lds = ctx->fir_premac_lds >= 0 ? ctx->fir_premac_lds : ((per_channel >= 5 && per_channel <= 32) ? (st.n < 120 ? 16384 : 49152) : 0);
        }
        if (!is_fir && !is_os && !step_fast && !sd.empty()) {
            /* the host of later reverbs' wet paths: the earlier general-kernel launch that lives longest (the extra workgroups take ~15 us at
             * 192 kHz; behind a short launch -- a lone compressor -- they would BE the launch).  Rough unit times in us, one workgroup per CU
             * (profiles/seg_latency_by_channels_r04.txt) */
            double w = 0.0;
            for (int h : kv.second[0].second.handles) {
                switch (ctx->units[(size_t)h].type) {
                case GDG_UNIT_COMPRESSOR: w += 2.6; break;
                case GDG_UNIT_TONESTACK: case GDG_UNIT_CABINET: w += 6.5; break;
                case GDG_UNIT_CHORUS: w += 9.2; break;
                case GDG_UNIT_REVERB: w += 18.0; break;
                case GDG_UNIT_AUTOWAH: w += 14.0; break;
                default: w += 4.0; break;
                }
            }
            if (ahead_host < 0 || w > ahead_host_weight) { ahead_host = (int)ctx->steps.size(); ahead_host_weight = w; }
        }
        ahead_lists.emplace_back();
        ctx->steps.push_back(st);
        seg_descs.push_back(sd);
        fir_descs.push_back(fd);
    }
    {   /* the reverbs' wet paths as extra workgroups share the chip with the premac's launches: with both, fewer channels (ctx.h) */
        bool any_premac = false;
        for (auto &st : ctx->steps) any_premac = any_premac || (st.is_fir && st.premac_ok);
        if (any_premac && (int)active.size() > ctx->seg_reverb_ahead_max) {
            for (auto &l : ahead_lists) { for (int idx : l) seg_units[(size_t)idx].ip[7] = 0; l.clear(); }
        }
    }
    {   /* two power amps' sums made ahead stream twice as long beside the segments: the tile kernel then only pays up to the channel count the
         * reverbs' extra workgroups pay to (bench chain, tile kernel off / on: 80 channels 162.4 -> 157.8 us, 96: 173.4 -> 177.4; one amp, 96: 112.9 -> 104.2) */
        int premac_steps = 0;
        for (auto &st : ctx->steps) premac_steps += (st.is_fir && st.premac_ok) ? 1 : 0;
        if (premac_steps >= 2 && (int)active.size() > ctx->seg_reverb_ahead_max) for (auto &st : ctx->steps) st.tile_ok = false;
    }
    /* a tiled step runs a reverb only as the mix of a wet path made by an earlier launch (seg.hip unit_reverb_mix_tile) */
    for (auto &tc : tile_conditional)
        if (tc.second < 0 || !seg_units[(size_t)tc.second].ip[7]) ctx->steps[(size_t)tc.first].tile_ok = false;
    {   /* the new filters' spectra, all together */
        const double tq = pnow();
        int rc = flush_ir(ctx);
        if (ptrace) fprintf(stderr, "[plan] prepare_fir %.1f ms, prepare_unit %.1f ms, flush_ir %.1f ms, so far %.1f ms\n", t_fir, t_unit, pnow() - tq, pnow() - t_plan0);
        if (rc != GDG_OK) return rc;
    }
    /* adjacent power amps (the benchmark chain: cabinet IR, then reverb IR): when EVERY channel of a FIR step hands its frame to the
     * next FIR step, that step's forward transform is produced by this step's inverse kernel -- no launch, no round trip of the frame */
    if (ctx->fir_chain && frames == GDG_MAX_FRAMES) {
        for (size_t i = 0; i + 1 < ctx->steps.size(); i++) {
            if (!ctx->steps[i].is_fir || !ctx->steps[i + 1].is_fir) continue;
            auto &a = fir_descs[i], &b = fir_descs[i + 1];
            bool ok = !a.empty() && a.size() == b.size() && ctx->steps[i].group_range == ctx->steps[i + 1].group_range;
            for (size_t k = 0; ok && k < a.size(); k++) ok = a[k].dst == b[k].src && !(a[k].flags & GDG_DST_IS_OUTPUT) && a[k].hop == frames && b[k].hop == frames;
            if (!ok) continue;
            ctx->steps[i].chain_next = true;
            for (auto &f : a) f.flags |= GDG_DST_UNUSED;      /* only the chained transform reads the frame (window-mode kernels ignore the flag) */
        }
    }
    /* serialise */
    ctx->blob.clear();
    auto append = [&](const void *p, size_t bytes) {
        size_t off = (ctx->blob.size() + 255) & ~(size_t)255;
        ctx->blob.resize(off + bytes);
        if (bytes) memcpy(ctx->blob.data() + off, p, bytes);
        return off;
    };
    for (size_t i = 0; i < ctx->steps.size(); i++) {
        if (ctx->steps[i].is_fir) ctx->steps[i].offset = append(fir_descs[i].data(), fir_descs[i].size() * sizeof(gdg_fir_chan));
        else ctx->steps[i].offset = append(seg_descs[i].data(), seg_descs[i].size() * sizeof(gdg_seg_chan));
    }
    ctx->units_offset = append(seg_units.data(), seg_units.size() * sizeof(gdg_seg_unit));
    for (size_t i = 0; i < ctx->steps.size(); i++) {
        ctx->steps[i].ahead_n = (int)ahead_lists[i].size();
        if (ctx->steps[i].ahead_n) ctx->steps[i].ahead_offset = append(ahead_lists[i].data(), ahead_lists[i].size() * sizeof(int));
    }
    if (ctx->blob.size() > ctx->d_blob_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        hipFree(ctx->d_blob);
        ctx->d_blob = nullptr;
        ctx->d_blob_cap = ctx->blob.size() * 2 + 4096;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_blob, ctx->d_blob_cap));
    } else {
        /* the previous plan may still be in use by launches in flight */
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->arena.trim();                         /* the stream is drained: the one place where giving spare chunks back stalls nobody */
    /* The counters are laid out afresh by every plan: a cell of this layout may sit on a word the previous layout used for something else (an
     * os_tiles flag keeps its launch's epoch, a chorus "done" mark keeps epoch * 32 + f + 1).  Every stream is joined and drained here. */
    if (wave_next > ctx->d_wave_cap) return fail(ctx, GDG_ERR_INVALID, "counter layout of %zu words exceeds the %zu allocated", wave_next, ctx->d_wave_cap);
    if (ctx->d_wave) HIP_TRY(ctx, hipMemsetAsync(ctx->d_wave, 0, ctx->d_wave_cap * sizeof(int), ctx->stream));
    {   /* the tiles' exchange area: one block per descriptor of the largest tile launch; zero with every plan (tags of another layout) */
        size_t need = 0;
        for (auto &st : ctx->steps) if (st.tile_ok) need = std::max(need, (size_t)st.n * gdg_segt_xch_words());
        if (need > ctx->d_tile_xch_cap) {
            hipFree(ctx->d_tile_xch);
            ctx->d_tile_xch = nullptr;
            ctx->d_tile_xch_cap = 0;
            HIP_TRY(ctx, hipMalloc((void **)&ctx->d_tile_xch, need * sizeof(unsigned long long)));
            ctx->d_tile_xch_cap = need;
        }
        if (ctx->d_tile_xch && need > 0) HIP_TRY(ctx, hipMemsetAsync(ctx->d_tile_xch, 0, ctx->d_tile_xch_cap * sizeof(unsigned long long), ctx->stream));
    }
    if (!ctx->blob.empty())
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_blob, ctx->blob.data(), ctx->blob.size(), hipMemcpyHostToDevice, ctx->stream));
    ctx->plan_frames = frames;
    ctx->plan_sr = sample_rate;
    ctx->plan_in = d_in;
    ctx->plan_out = d_out;
    ctx->dirty = false;
    return GDG_OK;
}
