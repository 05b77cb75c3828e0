/*
 * seg.hip -- the serial per-sample stage: every effects.Unit.Process() that is not the FIR power
 * amp, fused per channel into ONE launch per chain segment (the units between two FIR units).
 *
 * Design (gfx950): one workgroup of 1024 threads (16 wavefronts, 4 per SIMD) per channel; the frame (<= 8192
 * float64) is loaded once from HBM into LDS, ping-pongs between two LDS buffers from unit to
 * unit and is written once (16 B per sample of frame traffic, plus each unit's own state).
 * The 160 KiB LDS of CDNA4 is what lets two 8192-sample FP64 frames plus a 26 KiB tile live
 * on-chip.  Every unit is a non-inlined device function over the file-scope LDS arrays (all 21 inlined
 * into one kernel spilled hundreds of bytes per lane).  Recurrences are not run sample by sample by one lane:
 *   - one-pole sections and the "level" follower are affine maps, the peak follower is a
 *     max-affine map; both compose associatively, so every thread reduces its 8-sample chunk
 *     to one map, the 1024 maps are scanned across the workgroup (DPP row shifts inside 16-lane rows,
 *     readlane across rows, a 16-lane DPP scan of the wave totals through LDS) and every thread then
 *     replays its chunk from the exact incoming state IN THE REFERENCE'S OPERATION ORDER (so only the
 *     chunk-start state carries scan rounding, ~1e-16 relative);
 *   - the noise gate and the octaver's polarity logic are finite-state machines: scans of small
 *     function tables;
 *   - feed-forward delays (chorus, flanger, phaser, delay, reverb taps) read the unit's INPUT
 *     history and are embarrassingly parallel; history lives in an HBM ring per unit;
 *   - the reverb all-passes are true feedback loops of length M >= 99 samples that only couple samples
 *     M apart: the chains n, n + M, n + 2M, ... are walked independently, in place;
 *   - 2x/4x oversampling never leaves the CU: Lanczos-3 up-sampling (precomputed 6-tap polyphase
 *     weights), the waveshaper and the 77/155-tap decimator run tile by tile through LDS (polyphase
 *     layout, taps through the scalar cache).
 *
 * Compiled with -ffp-contract=off: Go never fuses multiply-add.
 */
/*
 * This file is compiled TWICE (csrc/Makefile):
 *   seg.o                 the general kernel: 1024 threads per channel (8-sample chunks), two LDS frame buffers + a 26 KiB tile = 159 KiB,
 *                         ONE workgroup per CU; every unit type, every frame size;
 *   segf.o  (-DSEG_FAST)  the kernel of the batch block size (8192 frames) for segments made of units that work IN PLACE:
 *                         512 threads per channel (16-sample chunks), ONE LDS frame buffer + a 13 KiB tile = 80 KiB, TWO workgroups per CU.
 *                         A workgroup's life is a chain of dependent phases (descriptor -> state -> data, scan -> replay, barriers; ~25 us
 *                         per segment whatever the frame size) plus its VALU work (~18 us per channel); with one workgroup per CU a
 *                         512-channel launch pays the chain twice (two rounds), with two per CU once (profiles/experiments/README.md, r04).
 * Same source, same arithmetic, same state layout in HBM (a stream may move between the two from call to call); the association of the
 * workgroup scans differs (512 chunks of 16 instead of 1024 of 8), i.e. results differ by the scans' rounding (~1e-16).
 */
/*   segt.o  (-DSEG_TILE)  round 6: per-frame calls of few channels with a channel's frame on SEG_TILES = 2 workgroups (512 threads, 8-sample
 *                         chunks, a tile of 4096 samples each): the units of a per-frame segment of few channels are VALU-bound on ONE CU
 *                         while 3/4 of the chip idles.  Same arithmetic, same chunks and -- the point -- THE SAME ASSOCIATION as the general
 *                         kernel: a scan's 16 wave totals still meet in one 16-lane scan, only that eight of them now come from the other
 *                         workgroup through HBM (tile_put / tile_get below): bit-identical results.  Units: compressor, shapers without
 *                         oversampling, tone stack, cabinet, chorus; the launch also carries the extra workgroups of the general kernel
 *                         (reverbs' wet paths, REVERB_AHEAD).  Two LDS frame buffers as in the general build. */
#ifdef SEG_FAST
#define GDG_CHK 16                        /* before gdg_internal.h: the scan-table layouts follow the chunk size */
#endif
#include "gdg_internal.h"
#include "go_consts.h"
#include "../../include/gdg.h"
#include <math.h>
#include <stdlib.h>

#ifdef SEG_FAST
#define SEG_T 512                         /* 8 waves per workgroup, two workgroups per CU = 4 waves per SIMD (128 VGPRs each) */
#define SEG_MIN_WAVES_PER_EU 4
#define SEG_SCR 1600                      /* the tone stack's four tables (4 x L2_SIZE = 1592 doubles) are the largest tenant */
#define seg_kernel segf_kernel
#define gdg_launch_seg gdg_launch_segf
#define gdg_seg_supported gdg_segf_supported
#elif defined(SEG_TILE)
#define SEG_T 512                         /* a tile of 4096 samples per workgroup, 8-sample chunks as in the general kernel */
#define SEG_MIN_WAVES_PER_EU 2
#define SEG_SCR 3328
#define SEG_TILES 2
#else
#define SEG_T 1024                        /* 16 waves = 4 per SIMD: hides the FP64 / LDS / HBM latencies of one workgroup per CU */
#define SEG_MIN_WAVES_PER_EU 4
#define SEG_SCR 3328
#endif
#if defined(SEG_FAST) || defined(SEG_TILE)
#define SEG_SUBSET                        /* a build that runs a subset of the units: the general kernel's other units and its side kernels live in seg.o */
#endif
#ifndef SEG_TILES
#define SEG_TILES 1
#endif
#define SEG_WAVES (SEG_T / 64)
#define SEG_LBUF (8192 + 256 + 8)
#define LX(e) ((e) + ((e) >> 5))
#define SEG_N (GDG_MAX_FRAMES / SEG_TILES)   /* samples a workgroup holds at the batch block size */
#define CHK (SEG_N / SEG_T)                  /* samples per thread at the batch block size */
static_assert(CHK * SEG_T * SEG_TILES == GDG_MAX_FRAMES, "chunk size");
static_assert(CHK == GDG_CHK, "chunk size of the scan tables");

/* the workgroup's LDS, at file scope so that the (non-inlined) unit functions address it as LDS, not through flat pointers */
__shared__ double s_a[SEG_LBUF];                     /* frame ping */
#ifdef SEG_FAST
#define s_b s_a                                      /* ONE frame buffer: every unit of this configuration works in place */
#else
__shared__ double s_b[SEG_LBUF];                     /* frame pong */
#endif
__shared__ double s_scr[SEG_SCR];                    /* oversampling / run-list tile */

#define SEG_STASH 128                             /* first state cell inside s_tmp: behind the scan scratch (block_scan: 2 x 4 recurrences x 16 waves;
                                                   * lin_scan / lin2_scan: two alternating pairs of 17-cell exchange slots = 68) */
__shared__ double s_tmp[SEG_STASH + 32 + 8 + 2];    /* scan scratch + 32 state cells + 16 unit types + the workgroup's ticket (WAVE) */
static_assert(2 * 4 * (SEG_T / 64) <= SEG_STASH, "block_scan scratch");
#ifdef SEG_FAST
static_assert(sizeof(double) * (SEG_LBUF + SEG_SCR + SEG_STASH + 32 + 8 + 2) <= 81920, "two workgroups per CU: 80 KiB of LDS each");
static_assert(4 * L2_SIZE <= SEG_SCR && 7 * LT_SIZE <= SEG_SCR, "scan tables fit the tile");
#endif

/* The units are INLINED into the kernel (`flip` says which LDS frame is the input).  A call costs more than it looks: the callee saves and
 * restores the callee-saved half of the vector registers it touches (v40-47, v56-63, ...: 20 to 36 dwords per lane and unit, straight to scratch
 * memory) -- 84 dwords per lane for the bench's first segment = 88 MB written and read back per launch, more than the frames themselves
 * (rocprofv3 WRITE_SIZE: 146 MB where 67 are data).  Inlining alone does not work: every unit's thread-index arithmetic and every constant of
 * the transcendental functions is hoisted to the top of the kernel and lives through all units (128 registers, 772 bytes of scratch).  Two
 * things keep values where they are used: seg_tid() hands every use of the thread index its own opaque copy, and this file is compiled with
 * -mllvm -disable-machine-licm (Makefile): 115-120 registers, no spills in the two-per-CU build, three spill stores in the general one. */
#define UNIT_FN static __device__ __forceinline__ void
__device__ __forceinline__ unsigned seg_tid() {
    unsigned t = __builtin_amdgcn_workitem_id_x();
    asm volatile("" : "+v"(t));
    return t;
}
/* wt ("write-through"): the launch runs a channel's frames on several workgroups (WAVE, at the segment kernel): what a unit leaves in HBM
 * for its next frame -- state cells, ring cells -- is stored with sc1 (st_* below), so that the hand-off needs no write-back of the XCD's L2.
 * A compile-time constant in every instantiation (the units are inlined). */
#define UNIT_ARGS const gdg_seg_unit *U, int flip, int N, const bool wt
#define UNIT_PROLOGUE                                                                     \
    double *in = flip ? s_b : s_a;                                                        \
    double *out = flip ? s_a : s_b;                                                       \
    double *tmp = s_tmp;                                                                  \
    double *scr = s_scr;                                                                  \
    (void)in; (void)out; (void)tmp; (void)scr;


#define ATTENUATION_HALF_DECIBEL 0.9440608762859234   /* oversampling/oversampling.go:13 */

__device__ __forceinline__ double clip1(double v) { return v < -1.0 ? -1.0 : (v > 1.0 ? 1.0 : v); }

/* math.Mod(x, 2 pi) for 0 <= x < ~8 pi (LFO phases): while x >= 2 pi the subtraction x - 2 pi is exact
 * (Sterbenz), so this is bit-identical to fmod and costs two compares instead of ocml's general fmod */
__device__ __forceinline__ double fmod_2pi(double x) {
    if (x >= 4.0 * GO_MATH_TWO_PI || x < 0.0) return fmod(x, GO_MATH_TWO_PI);
    if (x >= 2.0 * GO_MATH_TWO_PI) x -= 2.0 * GO_MATH_TWO_PI;
    if (x >= GO_MATH_TWO_PI) x -= GO_MATH_TWO_PI;
    return x;
}

/* ---- workgroup scan of (A, B) maps:  affine x -> A x + B   or   max-affine x -> max(A x, B) ---- */
template <bool MAXOP>
__device__ __forceinline__ void compose(double &A, double &B, double A1, double B1) {
    /* (A, B) := "first (A1, B1), then (A, B)" */
    double nb = A * B1;
    B = MAXOP ? fmax(nb, B) : nb + B;
    A = A * A1;
}

/*
 * In: this thread's chunk map (A, B).  Out: (Ap, Bp) = composition of the maps of all threads
 * with a lower index (identity for thread 0).  tmp: 2 * C * 4 doubles of LDS.
 */
/* DPP row shift right by D lanes inside each 16-lane row (register-to-register, no LDS crossbar);
 * lanes whose source falls outside the row keep their own value -- callers guard with (lane & 15) >= D */
template <int D>
__device__ __forceinline__ int row_shr_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x110 | D, 0xf, 0xf, false); }
template <int D>
__device__ __forceinline__ double row_shr(double v) {
    int lo = row_shr_i<D>(__double2loint(v)), hi = row_shr_i<D>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <int C, bool MAXOP, int D>
__device__ __forceinline__ void scan_row_step(double (&a)[C], double (&b)[C], int lane) {
#pragma unroll
    for (int c = 0; c < C; c++) {
        double ao = row_shr<D>(a[c]), bo = row_shr<D>(b[c]);
        if ((lane & 15) >= D) compose<MAXOP>(a[c], b[c], ao, bo);
    }
}

template <int C, bool MAXOP>
__device__ __forceinline__ void block_scan(const double (&A)[C], const double (&B)[C], double (&Ap)[C], double (&Bp)[C], double *tmp) {
    const int tid = seg_tid(), lane = tid & 63, wave = tid >> 6, row = lane >> 4;
    double a[C], b[C];
#pragma unroll
    for (int c = 0; c < C; c++) { a[c] = A[c]; b[c] = B[c]; }
    /* inclusive scan inside each row of 16 lanes with DPP shifts */
    scan_row_step<C, MAXOP, 1>(a, b, lane);
    scan_row_step<C, MAXOP, 2>(a, b, lane);
    scan_row_step<C, MAXOP, 4>(a, b, lane);
    scan_row_step<C, MAXOP, 8>(a, b, lane);
    /* row totals (lanes 15, 31, 47) -> prefix of the rows below, through scalar registers */
    double pa[C], pb[C];                               /* composition of all lower rows of this lane's row */
#pragma unroll
    for (int c = 0; c < C; c++) {
        double a0 = read_lane(a[c], 15), b0 = read_lane(b[c], 15);
        double a1 = read_lane(a[c], 31), b1 = read_lane(b[c], 31);
        double a2 = read_lane(a[c], 47), b2 = read_lane(b[c], 47);
        compose<MAXOP>(a1, b1, a0, b0);                /* rows 0..1 */
        compose<MAXOP>(a2, b2, a1, b1);                /* rows 0..2 */
        pa[c] = (row == 1) ? a0 : (row == 2) ? a1 : a2;
        pb[c] = (row == 1) ? b0 : (row == 2) ? b1 : b2;
        if (row == 0) { pa[c] = 1.0; pb[c] = 0.0; }
        else compose<MAXOP>(a[c], b[c], pa[c], pb[c]);
    }
    if (lane == 63) {
#pragma unroll
        for (int c = 0; c < C; c++) { tmp[(wave * C + c) * 2] = a[c]; tmp[(wave * C + c) * 2 + 1] = b[c]; }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; c++) {
        /* exclusive value: the inclusive one of the lane below; at a row start that is the rows-below prefix */
        double ae = row_shr<1>(a[c]), be = row_shr<1>(b[c]);
        if ((lane & 15) == 0) { ae = pa[c]; be = pb[c]; }
        /* maps of the preceding waves: lanes 0..15 of every wave scan the 16 wave totals with DPP (no serial
         * loop over LDS), then everybody reads the entry of wave - 1 through a scalar register */
        double wa = 1.0, wb = 0.0;
        if (lane < SEG_WAVES) { wa = tmp[(lane * C + c) * 2]; wb = tmp[(lane * C + c) * 2 + 1]; }
        {
            double t[1] = { wa }, u[1] = { wb };
            scan_row_step<1, MAXOP, 1>(t, u, lane);
            scan_row_step<1, MAXOP, 2>(t, u, lane);
            scan_row_step<1, MAXOP, 4>(t, u, lane);
            scan_row_step<1, MAXOP, 8>(t, u, lane);
            wa = t[0]; wb = u[0];
        }
        const int wsel = __builtin_amdgcn_readfirstlane(wave > 0 ? wave - 1 : 0);
        double aw = read_lane(wa, wsel), bw = read_lane(wb, wsel);
        if (wave == 0) { aw = 1.0; bw = 0.0; }
        compose<MAXOP>(ae, be, aw, bw);            /* first the preceding waves, then the lanes below */
        Ap[c] = ae; Bp[c] = be;
    }
    __syncthreads();
}

template <bool MAXOP>
__device__ __forceinline__ double apply_map(double A, double B, double s) {
    double v = A * s;
    return MAXOP ? fmax(v, B) : v + B;
}

/* ---- constant-coefficient recurrences on the batch block (8192 = 1024 threads x 8 samples) ---------------------------
 * A one-pole section (or follower) with a FIXED coefficient maps a thread's chunk start state s to  A s + B  (max(A s, B)
 * for the peak follower) with the SAME A = keep^8 for every thread.  The general block_scan above carries (A, B) pairs
 * through every step (4 DPP moves + 3 flops + a select, ~130 instructions per wave and scan: 70 % of the cabinet's
 * instruction stream, and these units are VALU-issue bound).  With A constant only B travels:
 *   - the chunk's zero-state result B is a dot product with precomputed weights a keep^(7-i)   (8 fma instead of 32 flops);
 *   - a scan step is  B += A^d * shifted(B): two DPP moves with zero fill (bound_ctrl) + one fma, no select;
 *   - rows are joined with row_bcast:15 / row_bcast:31 and per-lane powers A^(q+1), A^(lane-31) from a small LDS table;
 *   - the 16 wave totals are scanned by lanes 0..15 of every wave with the A^64 ladder, seeded with the state before the
 *     frame in lane 0, so lane w holds the state entering wave w;
 *   - start state of the thread's chunk = exclusive in-wave value (wave_shr:1) + A^lane * (state entering the wave).
 * One workgroup barrier per scan (the exchange slots alternate).  The tables depend on the coefficient only: the host builds them at
 * plan time (api_plan.cpp: scan_tables) and the unit copies them from HBM into LDS.  The exact replay from the start state is unchanged,
 * so only that start state carries the scan's rounding (~1e-16 relative), as before.
 * 2 x 2 variant (tone stack band = high-pass feeding a low-pass): the pair (h, l) evolves linearly with the constant
 * lower-triangular matrix [[1-aH, 0], [-aL, 1-aL]] per sample, so ONE scan of vectors replaces two scans and one of the
 * three passes. */
/* table layout (LT_*, L2_*): gdg_internal.h -- the tables are built on the host at plan time (api_plan.cpp scan_tables) */
#define LX_SLOT 17                /* exchange slot: [state before the frame | 16 wave totals] */

template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp0(double v) {       /* DPP move, lanes without a source (and rows outside ROWMASK) read 0 */
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
#define DPP_ROW_SHR(d) (0x110 | (d))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
#define DPP_WAVE_SHR1 0x138

/* the unit's tables: HBM (shared by every unit with the same coefficients, so usually L2 hits) -> LDS, all threads, issued together with
 * the frame chunk.  Round 2 built them here on every call, one wave per section: a dependent chain of ~400 FP64 operations (4 000 of the
 * tone stack's 24 000 cycles) for values that only change with the sample rate or a parameter. */
__device__ __forceinline__ void tab_fetch(double *dst, const double *src_generic, int n) {
    const double *src = src_generic;
    for (int i = seg_tid(); i < n; i += blockDim.x) dst[i] = *(const __attribute__((address_space(1))) double *)(src + i);
}

#ifdef SEG_TILE
/* ---- a channel's frame on several workgroups (SEG_TILE): what crosses between them ---------------------------------------------------------
 * Values travel as GRANULES (MI355X guide, hand-off price list): a naturally aligned 8-byte { 32 data bits | 32-bit tag } written by ONE
 * agent-scope store and read by ONE agent-scope load -- no flag, no fence, no ordering between granules: a granule is either this launch's
 * (its tag is the launch's epoch, a number no earlier launch of the context used for these cells) or it is not there yet.  A double is two
 * granules.  Slot layout per channel (gdg_seg_chan.scratch): [exchange id][16 waves][2 values][2 granules].  Every wait is bounded in time
 * (wave_spin_expired: the context's error word, never a hung device). */
struct TileCtx { unsigned long long *xch; int *d_error; int tile; int epoch; int xid; int pad; };
__shared__ TileCtx s_tc;
#define GDG_TILE_XIDS 32                  /* exchange ids per segment (api_plan.cpp counts them: compressor 1, tone stack 4, cabinet 7, chorus 1) */
__device__ __forceinline__ void tile_put(int xid, int wave, int value, double v);       /* (defined behind the hand-off primitives below) */
__device__ __forceinline__ double tile_get(int xid, int wave, int value);
#endif

template <bool MAXOP>
__device__ __forceinline__ double lin_comb(double acc, double f, double v) { return MAXOP ? fmax(acc, f * v) : fma(f, v, acc); }

/* zero-state result of this thread's chunk */
template <bool MAXOP, bool ABS = MAXOP>
__device__ __forceinline__ double lin_chunk_map(const double (&x)[CHK], const double *tab) {
    double B = 0.0;
#pragma unroll
    for (int i = 0; i < CHK; i++) B = lin_comb<MAXOP>(B, tab[LT_W + i], ABS ? fabs(x[i]) : x[i]);
    return B;
}

/* B = zero-state chunk result of this thread; *s0 (LDS) = state before the frame; returns the state at this thread's chunk start */
/* SEG_TILE: `wave` is the wave's place in the FRAME (tile x 8 + its place in the workgroup); the totals of the waves of lower tiles come in
 * through HBM (exchange id s_tc.xid + xslot), fetched by wave 0 in front of the scan's one barrier -- the 16-lane scan below then runs on the
 * very 16 values it runs on in the general kernel */
template <bool MAXOP>
__device__ __forceinline__ double lin_scan(double B, const double *tab, const double *s0, double *xch, const int xslot = 0) {
#ifdef SEG_TILE
    const int lane = seg_tid() & 63, lwave = seg_tid() >> 6, wave = s_tc.tile * SEG_WAVES + lwave;
#else
    const int lane = seg_tid() & 63, wave = seg_tid() >> 6;
    (void)xslot;
#endif
    double I = B;
    I = lin_comb<MAXOP>(I, tab[LT_ST + 0], dpp0<DPP_ROW_SHR(1), 0xf>(I));
    I = lin_comb<MAXOP>(I, tab[LT_ST + 1], dpp0<DPP_ROW_SHR(2), 0xf>(I));
    I = lin_comb<MAXOP>(I, tab[LT_ST + 2], dpp0<DPP_ROW_SHR(4), 0xf>(I));
    I = lin_comb<MAXOP>(I, tab[LT_ST + 3], dpp0<DPP_ROW_SHR(8), 0xf>(I));
    I = lin_comb<MAXOP>(I, tab[LT_PA + (lane & 15)], dpp0<DPP_ROW_BCAST15, 0xa>(I));     /* rows 1, 3 <- totals of rows 0, 2 */
    I = lin_comb<MAXOP>(I, tab[LT_PB + (lane & 31)], dpp0<DPP_ROW_BCAST31, 0xc>(I));     /* rows 2, 3 <- total of rows 0..1 */
    if (lane == 63) xch[1 + wave] = I;
#ifdef SEG_TILE
    if (lane == 63 && s_tc.tile + 1 < SEG_TILES) tile_put(s_tc.xid + xslot, wave, 0, I);
    if (lwave == 0 && lane < s_tc.tile * SEG_WAVES) xch[1 + lane] = tile_get(s_tc.xid + xslot, lane, 0);
#endif
    __syncthreads();
    const double E = dpp0<DPP_WAVE_SHR1, 0xf>(I);                 /* exclusive inside the wave */
    double T = (lane == 0) ? *s0 : xch[lane & 15];                /* lanes 0..15: [s0, total of wave 0, ..., of wave 14] */
    T = lin_comb<MAXOP>(T, tab[LT_ST + 6], dpp0<DPP_ROW_SHR(1), 0xf>(T));
    T = lin_comb<MAXOP>(T, tab[LT_ST + 7], dpp0<DPP_ROW_SHR(2), 0xf>(T));
    T = lin_comb<MAXOP>(T, tab[LT_ST + 8], dpp0<DPP_ROW_SHR(4), 0xf>(T));
    T = lin_comb<MAXOP>(T, tab[LT_ST + 9], dpp0<DPP_ROW_SHR(8), 0xf>(T));
    const double V = read_lane(T, __builtin_amdgcn_readfirstlane(wave));     /* state entering this wave */
    return lin_comb<MAXOP>(E, tab[LT_PC + lane], V);
}

/* ---- 2 x 2: lower-triangular matrices (m00, m10, m11) --------------------------------------------------------------- */
__device__ __forceinline__ void lin2_comb(double &h, double &l, const double *m, double sh, double sl) {
    h = fma(m[0], sh, h);
    l = fma(m[2], sl, fma(m[1], sh, l));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ void lin2_step(double &h, double &l, const double *m) {
    const double sh = dpp0<CTRL, ROWMASK>(h), sl = dpp0<CTRL, ROWMASK>(l);
    lin2_comb(h, l, m, sh, sl);
}
/* (ch, cl): zero-state chunk result; s0h / s0l (LDS): state before the frame; returns the chunk start state in (ch, cl) */
__device__ __forceinline__ void lin2_scan(double &ch, double &cl, const double *tab, const double *s0h, const double *s0l, double *xch, const int xslot = 0) {
#ifdef SEG_TILE
    const int lane = seg_tid() & 63, lwave = seg_tid() >> 6, wave = s_tc.tile * SEG_WAVES + lwave;
#else
    const int lane = seg_tid() & 63, wave = seg_tid() >> 6;
    (void)xslot;
#endif
    double h = ch, l = cl;
    lin2_step<DPP_ROW_SHR(1), 0xf>(h, l, tab + L2_ST + 0);
    lin2_step<DPP_ROW_SHR(2), 0xf>(h, l, tab + L2_ST + 3);
    lin2_step<DPP_ROW_SHR(4), 0xf>(h, l, tab + L2_ST + 6);
    lin2_step<DPP_ROW_SHR(8), 0xf>(h, l, tab + L2_ST + 9);
    lin2_step<DPP_ROW_BCAST15, 0xa>(h, l, tab + L2_PA + 3 * (lane & 15));
    lin2_step<DPP_ROW_BCAST31, 0xc>(h, l, tab + L2_PB + 3 * (lane & 31));
    if (lane == 63) { xch[1 + wave] = h; xch[LX_SLOT + 1 + wave] = l; }
#ifdef SEG_TILE
    if (lane == 63 && s_tc.tile + 1 < SEG_TILES) { tile_put(s_tc.xid + xslot, wave, 0, h); tile_put(s_tc.xid + xslot, wave, 1, l); }
    if (lwave == 0 && lane < s_tc.tile * SEG_WAVES) { xch[1 + lane] = tile_get(s_tc.xid + xslot, lane, 0); xch[LX_SLOT + 1 + lane] = tile_get(s_tc.xid + xslot, lane, 1); }
#endif
    __syncthreads();
    const double eh = dpp0<DPP_WAVE_SHR1, 0xf>(h), el = dpp0<DPP_WAVE_SHR1, 0xf>(l);
    double th = (lane == 0) ? *s0h : xch[lane & 15];
    double tl = (lane == 0) ? *s0l : xch[LX_SLOT + (lane & 15)];
    lin2_step<DPP_ROW_SHR(1), 0xf>(th, tl, tab + L2_ST + 18);
    lin2_step<DPP_ROW_SHR(2), 0xf>(th, tl, tab + L2_ST + 21);
    lin2_step<DPP_ROW_SHR(4), 0xf>(th, tl, tab + L2_ST + 24);
    lin2_step<DPP_ROW_SHR(8), 0xf>(th, tl, tab + L2_ST + 27);
    const int w = __builtin_amdgcn_readfirstlane(wave);
    const double vh = read_lane(th, w), vl = read_lane(tl, w);
    ch = eh; cl = el;
    lin2_comb(ch, cl, tab + L2_PC + 3 * lane, vh, vl);
}

/* ---- history rings in HBM --------------------------------------------------------------------
 * A ring of capacity C holds the last C inputs of a unit: oldest at wp, newest at wp - 1.
 * Sample with frame-relative index idx (-C <= idx < 0) is ring[(wp + idx) mod C].
 */
/* Rings, unit states and frames live in HBM, but their pointers reach the unit functions through structs in memory, so the
 * compiler only knows "generic" and emits FLAT loads (which also count as LDS operations and stall on both counters).  The
 * accessors below restore the global address space. */
#define GDG_GLOBAL __attribute__((address_space(1)))
typedef double seg_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ const GDG_GLOBAL double *as_global(const double *p) { return (const GDG_GLOBAL double *)p; }
__device__ __forceinline__ GDG_GLOBAL double *as_global(double *p) { return (GDG_GLOBAL double *)p; }
__device__ __forceinline__ GDG_GLOBAL int *as_global(int *p) { return (GDG_GLOBAL int *)p; }

/* Stores of what the NEXT frame of the channel reads (unit state, rings).  Plain, unless that frame runs on another workgroup (wt): then
 * write-through (sc1: relaxed agent-scope atomics for the cells, a 16-byte sc1 store for ring pairs), which leaves nothing in this XCD's
 * L2 for a release fence to write back (MI355X_MICROARCH.md, visibility: "sc1 payload -> every storing wave drains -> flag"). */
__device__ __forceinline__ void st_f64(GDG_GLOBAL double *p, double v, bool wt) {
    if (wt) __hip_atomic_store((GDG_GLOBAL unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
/* ... and the matching load of a state cell: sc1 (past this CU's L1, which no other CU's store refreshes) when the producer stored sc1 -- a
 * unit whose whole state is read this way needs no acquire fence at its hand-off (MI355X_MICROARCH.md: "sc1 loads may replace the acquire
 * only when the producer stored sc1") */
__device__ __forceinline__ double ld_f64(const GDG_GLOBAL double *p, bool wt) {
    if (wt) return __longlong_as_double((long long)__hip_atomic_load((const GDG_GLOBAL unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return *p;
}
__device__ __forceinline__ void st_i32(GDG_GLOBAL int *p, int v, bool wt) {
    if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ void st_v2d(GDG_GLOBAL seg_v2d *p, seg_v2d v, bool wt) {
    if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    else *p = v;
}

/* ---- frames of one channel in flight on SEVERAL workgroups (WAVE) ---------------------------------------------------------------
 * A window of W frames of a channel is W x U cells (frame f, unit u); cell (f, u) needs the frame from (f, u - 1) -- LDS of the same
 * workgroup -- and unit u's state from (f - 1, u) -- HBM.  The walk (one workgroup per channel, frame after frame) runs the cells one
 * at a time on one CU: fine when the channels fill the chip, 3/4 of it idle with 64 channels (a GPU's share of the 512-channel job on
 * eight).  Here workgroup f takes frame f and meets its predecessor once per unit: before unit u touches its state it waits until the
 * unit's cell of the channel (wave[u], HBM) says "frame f", after the unit it posts f + 1 (the window's last frame posts 0 for the
 * next launch).  A unit still sees its frames strictly in order -- state, rings and per-call quirks exactly as in the walk, the same
 * bits -- but unit u of frame f runs beside unit u + 1 of frame f - 1 on another CU: a window takes (sum of the units) + (W - 1) x
 * (slowest unit) instead of W x (sum).  Release / acquire at agent scope (the workgroups of a channel may sit on different XCDs, each
 * with an L2 of its own): every wave writes back before the barrier, one lane posts; one lane polls, every wave invalidates after the
 * barrier.  Workgroups take their frame by TICKET (seg_kernel), so a workgroup never waits for one that has not started. */
/* light: the unit reads everything its predecessor frame left it through sc1 loads (ld_f64): no acquire fence (1.7 us) */
/* Every spin is BOUNDED IN TIME (1 s unless option wave_spin_limit_ms says otherwise: the context's error block, d_error[1]): a counter that never
 * comes -- a launch that broke the protocol, a workgroup that died -- must end as an error code in the context's error word (the next
 * synchronize reports it), never as a hung device.  The clock is the constant 100 MHz one (s_memrealtime), read every 1024th poll; the first
 * 1024 polls (~1 ms) are free. */
#define GDG_WAVE_TIMEOUT_CODE 0x57415645           /* "WAVE" */
__device__ __forceinline__ bool wave_spin_expired(int &spins, unsigned long long &t0, int *d_error) {
    if ((++spins & 1023) != 0) return false;
    const unsigned long long now = wall_clock64();
    if (spins == 1024) { t0 = now; return false; }
    const int ms = d_error ? __hip_atomic_load(as_global(d_error) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    return now - t0 > (unsigned long long)(ms > 0 ? ms : 1000) * 100000ull;
}
__device__ __forceinline__ void wave_wait(int *cell, int want, bool light = false, int *d_error = nullptr) {
    if (seg_tid() == 0) {
        int spins = 0;
        unsigned long long t0 = 0;
        while (__hip_atomic_load(as_global(cell), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
            __builtin_amdgcn_s_sleep(1);
            if (wave_spin_expired(spins, t0, d_error)) { if (d_error) atomicExch(d_error, GDG_WAVE_TIMEOUT_CODE); break; }
        }
        if (!light) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          /* ONE buffer_inv sc1 per workgroup: this CU's L1 (MI355X_MICROARCH.md, visibility) */
    }
    __syncthreads();
}
__device__ __forceinline__ void wave_post(int *cell, int value, bool release) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                /* every storing wave drains */
    __syncthreads();
    if (seg_tid() == 0) {
        /* units whose stores for the next frame are write-through (UNIT_ARGS, wt) need no write-back; a segment with any other unit pays
         * ONE buffer_wbl2 sc1 -- this XCD's dirty lines, 2-8 us -- per unit and frame (`release`, bit 31 of the channel's wave_mask) */
        if (release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            /* the compiler may drop the wait behind the write-back (guide, pitfall 12) */
        __hip_atomic_store(as_global(cell), value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#ifdef SEG_TILE
__device__ __forceinline__ unsigned long long *tile_slot(int xid, int wave, int value) {
    return s_tc.xch + (((size_t)xid * 16 + (size_t)wave) * 2 + (size_t)value) * 2;
}
__device__ __forceinline__ void tile_put(int xid, int wave, int value, double v) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v), tag = (unsigned long long)(unsigned)s_tc.epoch << 32;
    GDG_GLOBAL unsigned long long *p = (GDG_GLOBAL unsigned long long *)tile_slot(xid, wave, value);
    __hip_atomic_store(p, (bits & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, (bits >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double tile_get(int xid, int wave, int value) {
    const GDG_GLOBAL unsigned long long *p = (const GDG_GLOBAL unsigned long long *)tile_slot(xid, wave, value);
    const unsigned tag = (unsigned)s_tc.epoch;
    int spins = 0;
    unsigned long long t0 = 0, lo, hi;
    for (;;) {
        lo = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hi = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag) break;
        __builtin_amdgcn_s_sleep(1);
        if (wave_spin_expired(spins, t0, s_tc.d_error)) { if (s_tc.d_error) atomicExch(s_tc.d_error, GDG_WAVE_TIMEOUT_CODE); break; }
    }
    return __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32)));
}
#endif

/* what a unit that meets its predecessor frame TWICE needs (the reverb: delay line, then all-pass rings): its second counter */
struct WaveGate { int *cell; int wf, wf_next; bool release; int epoch; int *d_error; };
#define GDG_WAVE_CELLS 8      /* counters per unit: [0] the unit's, [1] its second meeting (reverb), [2..5] "frame f is done" marks (chorus), spare */

__device__ __forceinline__ double ring_read(const double *ring, int C, int wp, int idx) {
    int p = wp + idx;
    if (p < 0) p += C;
    return as_global(ring)[p];
}

/* append the frame held in LDS buffer `in` to the ring (all threads), then thread 0 advances wp */
__device__ __forceinline__ void ring_append(double *ring, int C, int *wp_ptr, const double *in, int N, const bool wt = false) {
    if (C <= 0) return;
    const int wp = *wp_ptr;
    int first = N > C ? N - C : 0;
    if (((N - first) & 1) == 0) {
        /* sample pairs: one 16-byte store per pair (two stores where the ring wraps inside the pair) */
        GDG_GLOBAL double *g = as_global(ring);
        for (int i = first + 2 * (int)seg_tid(); i < N; i += 2 * SEG_T) {
            int p = (wp + i) % C;
            const double a = in[LX(i)], b = in[LX(i + 1)];
            if (p + 1 < C) { seg_v2d v = { a, b }; st_v2d((GDG_GLOBAL seg_v2d *)(g + p), v, wt); }
            else { st_f64(g + p, a, wt); st_f64(g, b, wt); }
        }
    } else {
        for (int i = first + (int)seg_tid(); i < N; i += SEG_T) {
            int p = (wp + i) % C;
            st_f64(as_global(ring) + p, in[LX(i)], wt);
        }
    }
    if (wt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* the sc1 pair stores are inline assembly: the compiler does not wait for them */
    __syncthreads();
    if (seg_tid() == 0) st_i32(as_global(wp_ptr), (wp + N) % C, wt);
}
/* the reference's fractional delay read (e.g. effects/flanger.go:63-90): both weights are 1 when the delay is integral */
__device__ __forceinline__ double frac_delay(const double *in, const double *ring, int C, int wp, int i, double delay_samples) {
    double early = floor(delay_samples), late = ceil(delay_samples);
    int ie = i - (int)early, il = i - (int)late;
    double se, sl;
    int pl = wp + il;
    if (pl < 0) pl += C;
    if (ie < 0 && il == ie - 1 && pl + 1 < C) {
        /* the two neighbours lie side by side in the HBM ring: one 16-byte load */
        seg_v2d v = *(const GDG_GLOBAL seg_v2d *)(as_global(ring) + pl);
        sl = v.x; se = v.y;
    } else {
        se = (ie >= 0) ? in[LX(ie)] : ring_read(ring, C, wp, ie);
        sl = (il >= 0) ? in[LX(il)] : ring_read(ring, C, wp, il);
    }
    double we = 1.0 - (delay_samples - early);
    double wl = 1.0 - (late - delay_samples);
    return (we * se) + (wl * sl);
}

/* ---- envelope follower shared by compressor / fuzz / octaver / auto-wah / auto-yoy -------------
 * (e.g. effects/compressor.go:37-58).  follow 0 = "envelope" (peak), 1 = "level", other = 1.0.
 * Leaves env[i] (the follower value AFTER sample i) in dst[LX(i)] and returns nothing; the new
 * state is written by the thread that owns the last sample.
 */
__device__ __forceinline__ void envelope_to(const double *in, double *dst, int N, int follow, double d_inv, double d,
                                            double *state, double *tmp) {
    const int tid = seg_tid();
    const int m = (N + SEG_T - 1) / SEG_T;
    const int c0 = min(N, tid * m), c1 = min(N, c0 + m);
    double s0 = *state;
    __syncthreads();                                /* everybody has read the state before it is rewritten */
    if (follow != 0 && follow != 1) {
        for (int i = c0; i < c1; i++) dst[LX(i)] = 1.0;
        if (c1 == N && c0 < N) *state = 1.0;
        return;
    }
    double A[1] = { 1.0 }, B[1] = { 0.0 }, Ap[1], Bp[1];
    if (follow == 0) {
        for (int i = c0; i < c1; i++) { A[0] *= d_inv; B[0] *= d_inv; double a = fabs(in[LX(i)]); if (a > B[0]) B[0] = a; }
        block_scan<1, true>(A, B, Ap, Bp, tmp);
        double e = apply_map<true>(Ap[0], Bp[0], s0);
        for (int i = c0; i < c1; i++) {
            e *= d_inv;
            double a = fabs(in[LX(i)]);
            if (a > e) e = a;
            dst[LX(i)] = e;
        }
        if (c1 == N && c0 < N) *state = e;
    } else {
        for (int i = c0; i < c1; i++) { A[0] *= d_inv; double diff = fabs(in[LX(i)]) - B[0]; B[0] += diff * d; }
        block_scan<1, false>(A, B, Ap, Bp, tmp);
        double e = apply_map<false>(Ap[0], Bp[0], s0);
        for (int i = c0; i < c1; i++) {
            double diff = fabs(in[LX(i)]) - e;
            e += diff * d;
            dst[LX(i)] = e;
        }
        if (c1 == N && c0 < N) *state = e;
    }
}

/* ---- register-resident chunks --------------------------------------------------------------------
 * With 1024 threads a thread owns at most CHK = 8 consecutive samples, so a whole recurrence chain (zero-state
 * pass, scan, exact replay, next section ...) runs on registers: one LDS read and one LDS write per sample and unit. */

/* A thread's chunk of the frame.  FULL = the frame is exactly CHK * SEG_T samples (the batch block size): every chunk is complete
 * and the per-sample "inside the chunk?" guards -- a v_cndmask pair and an exec-mask update per sample and per section, 3/4
 * of the instructions of the cabinet -- fold away at compile time. */
template <bool FULL> struct ChunkT {
    int c0, len; bool last;
    static constexpr bool full = FULL;
    __device__ __forceinline__ bool has(int i) const { return FULL || i < len; }
};
typedef ChunkT<false> Chunk;

__device__ __forceinline__ Chunk my_chunk(int N) {
    const int m = (N + SEG_T - 1) / SEG_T;
    Chunk c;
    c.c0 = min(N, (int)seg_tid() * m);
    c.len = min(N, c.c0 + m) - c.c0;
    c.last = (c.len > 0) && (c.c0 + c.len == N);
    return c;
}
__device__ __forceinline__ ChunkT<true> full_chunk() {
    ChunkT<true> c;
    c.c0 = (int)seg_tid() * CHK;
    c.len = CHK;
#ifdef SEG_TILE
    c.last = seg_tid() == SEG_T - 1 && s_tc.tile == SEG_TILES - 1;       /* the frame's last sample: its thread leaves the unit's state */
#else
    c.last = seg_tid() == SEG_T - 1;
#endif
    return c;
}
template <class C>
__device__ __forceinline__ void chunk_load(const double *buf, const C &c, double (&v)[CHK]) {
#pragma unroll
    for (int i = 0; i < CHK; i++) v[i] = c.has(i) ? buf[LX(c.c0 + i)] : 0.0;
}
template <class C>
__device__ __forceinline__ void chunk_store(double *buf, const C &c, const double (&v)[CHK]) {
#pragma unroll
    for (int i = 0; i < CHK; i++) if (c.has(i)) buf[LX(c.c0 + i)] = v[i];
}

/* one-pole section on a register chunk; s is the section's state before the frame on entry, after it on exit (valid in the `last` thread) */
enum { OP_DIFF_OLD = 0, OP_OLD = 1, OP_NEW = 2, OP_DIFF_NEW = 3 };    /* what a one-pole section emits, see onepole() below */

template <int MODE, class C>
__device__ __forceinline__ void onepole_reg(double (&v)[CHK], const C &c, double a, double &s, double *tmp) {
    double A[1] = { 1.0 }, B[1] = { 0.0 }, Ap[1], Bp[1];
    const double keep = 1.0 - a;
#pragma unroll
    for (int i = 0; i < CHK; i++) if (c.has(i)) { A[0] *= keep; double diff = v[i] - B[0]; B[0] += diff * a; }
    block_scan<1, false>(A, B, Ap, Bp, tmp);
    s = apply_map<false>(Ap[0], Bp[0], s);
#pragma unroll
    for (int i = 0; i < CHK; i++) {
        if (c.has(i)) {
            double x = v[i];
            double diff = x - s;
            double s_old = s;
            s += diff * a;
            v[i] = (MODE == OP_DIFF_OLD) ? diff : (MODE == OP_OLD) ? s_old : (MODE == OP_NEW) ? s : x - s;
        }
    }
}

/* the same with a per-sample coefficient (auto-wah) */
template <int MODE, class C>
__device__ __forceinline__ void onepole_reg_var(double (&v)[CHK], const double (&a)[CHK], const C &c, double &s, double *tmp) {
    double A[1] = { 1.0 }, B[1] = { 0.0 }, Ap[1], Bp[1];
#pragma unroll
    for (int i = 0; i < CHK; i++) if (c.has(i)) { A[0] *= (1.0 - a[i]); double diff = v[i] - B[0]; B[0] += diff * a[i]; }
    block_scan<1, false>(A, B, Ap, Bp, tmp);
    s = apply_map<false>(Ap[0], Bp[0], s);
#pragma unroll
    for (int i = 0; i < CHK; i++) {
        if (c.has(i)) {
            double x = v[i];
            double diff = x - s;
            double s_old = s;
            s += diff * a[i];
            v[i] = (MODE == OP_DIFF_OLD) ? diff : (MODE == OP_OLD) ? s_old : (MODE == OP_NEW) ? s : x - s;
        }
    }
}

/* follower on a register chunk: x -> e (value after each sample); same conventions as onepole_reg */
template <class C>
__device__ __forceinline__ void envelope_reg(const double (&x)[CHK], double (&e)[CHK], const C &c, int follow, double d_inv, double d,
                                             double &s, double *tmp) {
    double A[1] = { 1.0 }, B[1] = { 0.0 }, Ap[1], Bp[1];
#pragma unroll
    for (int i = 0; i < CHK; i++) e[i] = 1.0;
    if (follow == 0) {
#pragma unroll
        for (int i = 0; i < CHK; i++) if (c.has(i)) { A[0] *= d_inv; B[0] *= d_inv; double q = fabs(x[i]); if (q > B[0]) B[0] = q; }
        block_scan<1, true>(A, B, Ap, Bp, tmp);
        s = apply_map<true>(Ap[0], Bp[0], s);
#pragma unroll
        for (int i = 0; i < CHK; i++) if (c.has(i)) { s *= d_inv; double q = fabs(x[i]); if (q > s) s = q; e[i] = s; }
    } else if (follow == 1) {
#pragma unroll
        for (int i = 0; i < CHK; i++) if (c.has(i)) { A[0] *= d_inv; double diff = fabs(x[i]) - B[0]; B[0] += diff * d; }
        block_scan<1, false>(A, B, Ap, Bp, tmp);
        s = apply_map<false>(Ap[0], Bp[0], s);
#pragma unroll
        for (int i = 0; i < CHK; i++) if (c.has(i)) { double diff = fabs(x[i]) - s; s += diff * d; e[i] = s; }
    } else {
#pragma unroll
        for (int i = 0; i < CHK; i++) e[i] = 1.0;
        s = 1.0;
    }
}

/* The unit descriptor is the same for the whole workgroup and never written by the kernel: read it through scalar loads
 * (wave-uniform pointer in SGPRs, constant address space) instead of FLAT vector loads from the generic pointer. */
#define GDG_CONST __attribute__((address_space(4)))
__device__ __forceinline__ const GDG_CONST gdg_seg_unit *uniform_unit(const gdg_seg_unit *p) {
    unsigned long long v = (unsigned long long)p;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const GDG_CONST gdg_seg_unit *)(((unsigned long long)hi << 32) | lo);
}

/* ---- compressor: effects/compressor.go:18-84 ---------------------------------------------------
 * ip0 follow; dp0 gain limit factor, dp1 target factor, dp2 exp(-20/sr), dp3 1 - dp2; ds0 envelope */
template <class C>
__device__ __forceinline__ void compressor_body(const gdg_seg_unit *U, int flip, int N, const C &c, const bool wt) {
    UNIT_PROLOGUE
    double s = ld_f64(as_global(U->ds), wt);
    double x[CHK], e[CHK];
    chunk_load(in, c, x);
    __syncthreads();                                /* everybody holds the old state before the last thread rewrites it */
    envelope_reg(x, e, c, U->ip[0], U->dp[2], U->dp[3], s, tmp);
    const double limit = U->dp[0], target = U->dp[1];
#pragma unroll
    for (int i = 0; i < CHK; i++) {
        double gain = target / e[i];
        if (gain > limit) gain = limit;
        x[i] = clip1(gain * x[i]);
    }
    chunk_store(out, c, x);
    if (c.last) st_f64(as_global(U->ds), s, wt);
}
/* the batch block size: constant-coefficient scan (see lin_scan) */
/* defer != NULL (an LDS cell): the unit's new state goes there instead of to HBM -- os_tiles_kernel runs the unit in every tile's workgroup
 * and lets ONE of them store the state, once all have read the old one */
__device__ __forceinline__ void compressor_full(const gdg_seg_unit *Ug, int flip, const bool wt, double *defer = nullptr) {
    UNIT_PROLOGUE
    const GDG_CONST gdg_seg_unit *U = uniform_unit(Ug);
    double *st = tmp + SEG_STASH;
    const int tid = seg_tid();
    const int follow = U->ip[0];
    const double d_inv = U->dp[2], d = U->dp[3], limit = U->dp[0], target = U->dp[1];
    GDG_GLOBAL double *ds = as_global(U->ds);
    if (tid == 0) st[0] = ld_f64(ds, wt);
    tab_fetch(scr, U->tab, LT_SIZE);
    const ChunkT<true> c = full_chunk();
    double x[CHK], e[CHK];
    chunk_load(in, c, x);
    __syncthreads();
    double s = 1.0;
    if (follow == 0) {
        s = lin_scan<true>(lin_chunk_map<true>(x, scr), scr, &st[0], tmp);
#pragma unroll
        for (int i = 0; i < CHK; i++) { s *= d_inv; double q = fabs(x[i]); if (q > s) s = q; e[i] = s; }
    } else if (follow == 1) {
        s = lin_scan<false>(lin_chunk_map<false, true>(x, scr), scr, &st[0], tmp);
#pragma unroll
        for (int i = 0; i < CHK; i++) { double diff = fabs(x[i]) - s; s += diff * d; e[i] = s; }
    } else {
#pragma unroll
        for (int i = 0; i < CHK; i++) e[i] = 1.0;
    }
#pragma unroll
    for (int i = 0; i < CHK; i++) {
        double gain = target / e[i];
        if (gain > limit) gain = limit;
        x[i] = clip1(gain * x[i]);
    }
    chunk_store(out, c, x);
    if (c.last) { if (defer) *defer = s; else st_f64(ds, s, wt); }
}
UNIT_FN unit_compressor(UNIT_ARGS) {
    if (N == CHK * SEG_T) compressor_full(U, flip, wt);
    else compressor_body(U, flip, N, my_chunk(N), wt);
}

/* ---- memoryless waveshapers ---------------------------------------------------------------------- */
struct Shaper { int type; int valve; double gain, drive, clean, level; };

__device__ __forceinline__ double shape(const Shaper &S, double sample) {
    if (S.type == GDG_UNIT_OVERDRIVE) {             /* effects/overdrive.go:57-76 */
        double arg = S.gain * sample;
        double dist = 0.0;
        if (S.valve == 0) {
            double aarg = GO_MATH_QUARTER_PI * arg;
            dist = GO_MATH_TWO_OVER_PI * atan(aarg);
        } else if (S.valve == 1) {
            double x = exp(-arg);
            dist = (2.0 / (1.0 + x)) - 1.0;
        }
        double mix = (S.drive * dist) + (S.clean * sample);
        return S.level * mix;
    } else if (S.type == GDG_UNIT_DISTORTION) {     /* effects/distortion.go:34-47 */
        return S.level * clip1(S.gain * sample);
    } else {                                        /* excess, effects/excess.go:33-64 */
        double pre = S.gain * sample;
        double abs_pre = fabs(pre);
        bool exceeded = abs_pre > 1.0;
        bool negative = pre < 0.0;
        double fl = floor(abs_pre + 1.0);
        int section = (int)(0.5 * fl);
        bool section_odd = (section % 2) != 0;
        bool inverted = section_odd != (exceeded && negative);
        double excess = fmod(abs_pre + 1.0, 2.0);
        if (exceeded) pre = inverted ? 1.0 - excess : excess - 1.0;
        return S.level * pre;
    }
}

/*
 * overdrive / distortion / excess incl. the oversampling wrapper (e.g. effects/overdrive.go:83-144,
 * oversampling/oversampling.go:54-184, resample/resample.go:148-176).
 * dp0 gain, dp1 drive, dp2 clean, dp3 level; ip[4] valve (overdrive); jp0 factor (1, 2, 4).
 * hist: [0..7] the last 8 inputs, [8 .. 8+TAPS-2] the last TAPS-1 waveshaped oversampled samples.
 */
/* Oversampled shaping (oversampling/oversampling.go:160-236 + the unit's curve), F = 2 or 4, in tiles through LDS:
 *   stage 1: one thread per INPUT sample: the six-sample Lanczos window is read once and gives all F phases
 *            (resample.go:148-176: phase 0 is the input sample itself), each shaped and stored in a POLYPHASE layout
 *            (sample m at [m mod F][m div F]), so that stage 2's lanes walk consecutive LDS words;
 *   stage 2: y = clip(sum_k h[k] w[F o - k]) * 0.944 (filter.Process semantics; the reference computes this sum by FFT, so there
 *            is no operation order to keep: fused multiply-adds).  The decimator was LDS-bandwidth bound -- 155 eight-byte reads
 *            per output sample -- so every thread now makes R = 2 (4x) or 4 (2x) CONSECUTIVE outputs from one sliding window:
 *            16-byte loads of slot pairs, each loaded value feeding up to R accumulators (taps through the scalar cache).
 * To give all 1024 threads R outputs a tile has S = 2048 (4x) / 4096 (2x) outputs: the staging area is the whole OUTPUT frame
 * buffer (8448 words), and the tiles are walked from the END of the frame so that a tile's results can overwrite its own inputs in
 * the INPUT buffer (lower tiles only read inputs below it): the unit works in place and the caller does not flip the buffers.
 * 4 (or 2) tiles and 8 (4) barriers per frame instead of 11 tiles of 792 outputs with 23 % of the lanes idle.
 * hist: [8 inputs | TAPS - 1 oversampled samples of the previous call]; scr: the previous call's tail, saved before it is replaced. */
__device__ __forceinline__ const double *uniform_ptr(const double *p) {
    unsigned long long v = (unsigned long long)p;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const double *)(((unsigned long long)hi << 32) | lo);
}

template <int F> struct OsCfg {
    static constexpr int TAPS = GDG_OS_TAPS(F);
    static constexpr int BACK = GDG_OS_BACK(F);                   /* input samples reached back by the filter: ceil((TAPS-1)/F) */
    static constexpr int R = GDG_OS_R(F);                         /* outputs per thread */
    static constexpr int S = R * SEG_T;                           /* outputs per tile */
    static constexpr int NC = GDG_OS_NC;                          /* slots per phase a thread walks: R t .. R t + NC - 1 */
    static constexpr int PH = (S - R + NC + 1) & ~1;              /* capacity of one phase array (even: pair loads stay aligned) */
    static constexpr int NE = GDG_OS_NE(F), PADLO = GDG_OS_PADLO(F);
    static_assert(NC >= BACK + R && NC % 8 == 0, "slot window");
    static_assert(F * PH <= SEG_LBUF, "the staging area is one frame buffer");
};

/* The decimating filter for the R consecutive outputs of thread t.  Output j reads slot R t + c of phase r with tap
 * F (j + BACK - c) - r; the phase-major table tp[r][.] is zero where that tap does not exist, so the walk over c = 0 .. NC - 1 is
 * branch free.  Eight slots (four 16-byte LDS loads) and their 8 + R - 1 table entries (scalar loads, consecutive) per step. */
template <int F>
__device__ __forceinline__ void os_decimate(const double *stage, const GDG_CONST double *tp, int t, double (&acc)[OsCfg<F>::R]) {
    constexpr int BACK = OsCfg<F>::BACK, R = OsCfg<F>::R, PH = OsCfg<F>::PH, NC = OsCfg<F>::NC, NE = OsCfg<F>::NE, PADLO = OsCfg<F>::PADLO;
#pragma unroll
    for (int j = 0; j < R; j++) acc[j] = 0.0;
#pragma unroll 1
    for (int r = 0; r < F; r++) {
        const double *ph = stage + r * PH + R * t;                /* R t is even and r PH is even: 16-byte aligned pairs */
        const GDG_CONST double *tr = tp + r * NE;
#pragma unroll 1
        for (int cb = 0; cb < NC / 8; cb++) {
            seg_v2d w2[4];
#pragma unroll
            for (int q = 0; q < 4; q++) w2[q] = *reinterpret_cast<const seg_v2d *>(ph + 8 * cb + 2 * q);
            /* entry of (slot c = 8 cb + i, output j): (j + BACK - c) + PADLO = (NC - 1 - 8 cb - 7) + (j - i + 7) */
            const GDG_CONST double *te = tr + (NC - 8 - 8 * cb);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const double w = (i & 1) ? w2[i >> 1].y : w2[i >> 1].x;
#pragma unroll
                for (int j = 0; j < R; j++) acc[j] = fma(te[j - i + 7], w, acc[j]);
            }
        }
    }
    (void)BACK; (void)PADLO;
}

/* DBG (gdg_debug_oversample_decimate only): no waveshaper between the two halves, and the oversampled stream -- sample F i + r of the
 * call = phase r of input slot i -- also goes to `dbg_up`, so that oversampling_test.go's vectors meet these tiles directly */
template <int F, bool DBG = false>
__device__ __forceinline__ void shaper_oversampled(const Shaper &S, double *in, double *stage, double *scr, double *hist_generic,
                                                   const double *taps_generic, const double *lw_generic, int N, double *dbg_up = nullptr, bool wt = false) {
    constexpr int TAPS = OsCfg<F>::TAPS, BACK = OsCfg<F>::BACK, R = OsCfg<F>::R, TILE = OsCfg<F>::S, PH = OsCfg<F>::PH;
    /* wave-uniform table pointers in SGPRs + constant address space: the taps arrive through scalar loads */
    const GDG_CONST double *taps = (const GDG_CONST double *)uniform_ptr(taps_generic);      /* phase-major, zero padded */
    const GDG_CONST double *lw = (const GDG_CONST double *)uniform_ptr(lw_generic);
    GDG_GLOBAL double *hist = as_global(hist_generic);
    const int tid = seg_tid();
    /* the previous call's state, before it is replaced: scr[0 .. TAPS-2] the oversampled tail, scr[TAPS-1 .. TAPS+6] the 8 inputs */
    for (int q = tid; q < TAPS - 1 + 8; q += SEG_T) scr[q] = (q < TAPS - 1) ? hist[8 + q] : hist[q - (TAPS - 1)];
    __syncthreads();
    /* stream sample s[k]: k < 0 from the 8-sample history, else the frame */
    auto s_at = [&](int k) -> double { return k >= 0 ? in[LX(k)] : scr[TAPS - 1 + 8 + k]; };
    /* last 8 inputs (concatenation with the old history when N < 8): taken now, the frame is overwritten below */
    double keep = 0.0;
    if (tid < 8) keep = s_at(N - 8 + tid);
    const int n_tiles = (N + TILE - 1) / TILE;
    for (int tile = n_tiles - 1; tile >= 0; tile--) {
        const int o0 = tile * TILE;
        const int S_out = min(TILE, N - o0);
        const int I0 = o0 - BACK;                             /* input index held at slot 0 of every phase array */
        const int slots = S_out + BACK;                       /* slots I0 .. o0 + S_out - 1 */
        for (int idx = tid; idx < min(slots + OsCfg<F>::NC, PH); idx += SEG_T) {
            const int i = I0 + idx;
            if (idx >= slots) {
                /* a thread's walk reaches up to NC - BACK - 1 slots past the tile: defined values that only ever meet zero entries */
#pragma unroll
                for (int r = 0; r < F; r++) stage[r * PH + idx] = 0.0;
            } else if (i < 0) {
                /* oversampled samples of the previous call (m = F i + r < 0); older than the stored tail: never read */
#pragma unroll
                for (int r = 0; r < F; r++) {
                    const int m = F * i + r;
                    stage[r * PH + idx] = (m >= -(TAPS - 1)) ? scr[(TAPS - 1) + m] : 0.0;
                }
            } else {
                double w6[6];
#pragma unroll
                for (int t = 0; t < 6; t++) w6[t] = s_at(i - 6 + t);
                stage[idx] = DBG ? w6[2] : shape(S, w6[2]);   /* phase 0: s[i - 4], resample.go:160-164 */
                if (DBG && idx >= BACK) dbg_up[F * i] = w6[2];
#pragma unroll
                for (int r = 1; r < F; r++) {
                    double up = 0.0;
#pragma unroll
                    for (int t = 0; t < 6; t++) up += w6[t] * lw[(r - 1) * 6 + t];
                    stage[r * PH + idx] = DBG ? up : shape(S, up);
                    if (DBG && idx >= BACK) dbg_up[F * i + r] = up;
                }
            }
        }
        __syncthreads();
        if (tile == n_tiles - 1) {
            /* keep the last TAPS - 1 oversampled samples m = F N - (TAPS - 1) .. F N - 1 */
            for (int q = tid; q < TAPS - 1; q += SEG_T) {
                const int m = F * N - (TAPS - 1) + q;
                const int i = (m >= 0) ? m / F : -((-m + F - 1) / F);
                const int r = m - i * F;
                st_f64(hist + 8 + q, stage[r * PH + (i - I0)], wt);       /* i >= I0 always: F * BACK >= TAPS - 1 */
            }
        }
        if (R * tid < S_out) {
            double acc[R];
            os_decimate<F>(stage, taps, tid, acc);
#pragma unroll
            for (int j = 0; j < R; j++)
                if (R * tid + j < S_out) in[LX(o0 + R * tid + j)] = ATTENUATION_HALF_DECIBEL * clip1(acc[j]);     /* in place */
        }
        __syncthreads();
    }
    if (tid < 8) st_f64(hist + tid, keep, wt);
}

/* returns 1 when the result is in the INPUT buffer (oversampled: in place), 0 when it is in the other one */
/* (the general build keeps the call: the oversampled variants are large) */
#ifdef SEG_SUBSET
static __device__ __forceinline__ int unit_shaper(UNIT_ARGS, const gdg_os_tables &os) {
#else
__device__ __attribute__((noinline)) int unit_shaper(UNIT_ARGS, const gdg_os_tables &os) {
#endif
    UNIT_PROLOGUE
    Shaper S;
    S.type = U->type; S.valve = U->ip[4];
    S.gain = U->dp[0]; S.drive = U->dp[1]; S.clean = U->dp[2]; S.level = U->dp[3];
    const int f = U->jp[0];
    if (f <= 1) {
        for (int i = seg_tid(); i < N; i += SEG_T) out[LX(i)] = shape(S, in[LX(i)]);
        return 0;
    }
#ifndef SEG_SUBSET                                  /* the oversampled shapers stage a whole output frame in the second buffer (general build only) */
    if (f == 2) shaper_oversampled<2>(S, in, out, scr, U->hist, os.tapsP2, os.lanczos2, N, nullptr, wt);
    else shaper_oversampled<4>(S, in, out, scr, U->hist, os.tapsP4, os.lanczos4, N, nullptr, wt);
#endif
    return 1;
}

/* ---- tone stack: effects/tonestack.go:19-100 ------------------------------------------------------
 * dp0..3 band factors, dp4..7 (1 - exp(-2 pi fA/sr)), dp8..11 (1 - exp(-2 pi fB/sr)); ds0..3 hcv, ds4..7 lcv */
template <class C>
__device__ __forceinline__ void tonestack_body(const gdg_seg_unit *U, int flip, int N, const C &c, const bool wt) {
    UNIT_PROLOGUE
    double *st = tmp + SEG_STASH;                    /* the eight capacitor voltages, stashed in LDS (not in live registers) */
    if (seg_tid() < 8) st[seg_tid()] = ld_f64(as_global(U->ds) + seg_tid(), wt);
    double x[CHK];
    chunk_load(in, c, x);
    __syncthreads();
    /* two bands at a time (all four at once do not fit 128 VGPRs and spill); the band sum keeps the reference's
     * order j = 0..3 because the partial sum is carried across the two rounds (in the thread's own cells of `out`) */
#pragma unroll 1
    for (int pair = 0; pair < 2; pair++) {
        double aH[2], aL[2], fac[2];
#pragma unroll
        for (int j = 0; j < 2; j++) { aH[j] = U->dp[4 + 2 * pair + j]; aL[j] = U->dp[8 + 2 * pair + j]; fac[j] = U->dp[2 * pair + j]; }
        double A[2], B[2], Ap[2], Bp[2];
        /* pass 1: chunk maps of the high-pass capacitors */
#pragma unroll
        for (int j = 0; j < 2; j++) { A[j] = 1.0; B[j] = 0.0; }
#pragma unroll
        for (int i = 0; i < CHK; i++) {
            if (c.has(i)) {
#pragma unroll
                for (int j = 0; j < 2; j++) { A[j] *= (1.0 - aH[j]); double diff = x[i] - B[j]; B[j] += diff * aH[j]; }
            }
        }
        block_scan<2, false>(A, B, Ap, Bp, tmp);
        double h[2], hs[2];
#pragma unroll
        for (int j = 0; j < 2; j++) { hs[j] = apply_map<false>(Ap[j], Bp[j], st[2 * pair + j]); h[j] = hs[j]; }
        /* pass 2: exact high-pass, chunk maps of the low-pass capacitors */
#pragma unroll
        for (int j = 0; j < 2; j++) { A[j] = 1.0; B[j] = 0.0; }
#pragma unroll
        for (int i = 0; i < CHK; i++) {
            if (c.has(i)) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    double diff = x[i] - h[j];
                    h[j] += diff * aH[j];
                    A[j] *= (1.0 - aL[j]);
                    diff -= B[j];
                    B[j] += diff * aL[j];
                }
            }
        }
        block_scan<2, false>(A, B, Ap, Bp, tmp);
        double l[2];
#pragma unroll
        for (int j = 0; j < 2; j++) { l[j] = apply_map<false>(Ap[j], Bp[j], st[4 + 2 * pair + j]); h[j] = hs[j]; }
        /* pass 3: the reference's loop body from the exact chunk-start state */
#pragma unroll
        for (int i = 0; i < CHK; i++) {
            if (c.has(i)) {
                double sum = (pair == 0) ? 0.0 : out[LX(c.c0 + i)];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    double diff = x[i] - h[j];
                    h[j] += diff * aH[j];
                    diff -= l[j];
                    double pre = l[j];
                    l[j] += diff * aL[j];
                    sum += fac[j] * pre;
                }
                out[LX(c.c0 + i)] = (pair == 0) ? sum : clip1(sum);
            }
        }
        if (c.last) {
#pragma unroll
            for (int j = 0; j < 2; j++) { st_f64(as_global(U->ds) + 2 * pair + j, h[j], wt); st_f64(as_global(U->ds) + 4 + 2 * pair + j, l[j], wt); }
        }
    }
}
/* the batch block size: per band ONE scan of the (high-pass, low-pass) state pair with the constant 2 x 2 chunk matrix, then the
 * reference's loop body from the scanned chunk-start state; the band sum stays in registers (j = 0..3 as in the reference) */
__device__ __forceinline__ void tonestack_full(const gdg_seg_unit *Ug, int flip, const bool wt) {
    UNIT_PROLOGUE
    const GDG_CONST gdg_seg_unit *U = uniform_unit(Ug);
    double *st = tmp + SEG_STASH;                    /* [0..7] the capacitor voltages, [16..27] factors and coefficients */
    const int tid = seg_tid();
    GDG_GLOBAL double *ds = as_global(U->ds);
    if (tid < 8) st[tid] = ld_f64(ds + tid, wt);
    if (tid < 12) st[16 + tid] = U->dp[tid];
    tab_fetch(scr, U->tab, 4 * L2_SIZE);
    const ChunkT<true> c = full_chunk();
    double x[CHK], sum[CHK];
    chunk_load(in, c, x);
#pragma unroll
    for (int i = 0; i < CHK; i++) sum[i] = 0.0;
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < 4; j++) {
        const double *tab = scr + j * L2_SIZE;
        const double fac = st[16 + j], aH = st[20 + j], aL = st[24 + j];
        double h = 0.0, l = 0.0;
#pragma unroll
        for (int i = 0; i < CHK; i++) { h = fma(tab[L2_W + 2 * i], x[i], h); l = fma(tab[L2_W + 2 * i + 1], x[i], l); }
        lin2_scan(h, l, tab, &st[j], &st[4 + j], tmp + (j & 1) * 2 * LX_SLOT, j);
#pragma unroll
        for (int i = 0; i < CHK; i++) {
            double diff = x[i] - h;
            h += diff * aH;
            diff -= l;
            double pre = l;
            l += diff * aL;
            sum[i] += fac * pre;
        }
        if (c.last) { st_f64(ds + j, h, wt); st_f64(ds + 4 + j, l, wt); }
    }
#pragma unroll
    for (int i = 0; i < CHK; i++) sum[i] = clip1(sum[i]);
    chunk_store(out, c, sum);
}
UNIT_FN unit_tonestack(UNIT_ARGS) {
    if (N == CHK * SEG_T) tonestack_full(U, flip, wt);
    else tonestack_body(U, flip, N, my_chunk(N), wt);
}

/* ---- cabinet (IIR): effects/cabinet.go:27-162 -------------------------------------------------------
 * dp0..2 high-pass (1 - exp(-2 pi f/sr)) for 300/120/80 Hz, dp3..6 low-pass for 3/4/5/6 kHz; ds0..2 hcv, ds3..6 lcv.
 * Seven one-pole sections in series on a register chunk: per section a zero-state pass, a workgroup scan, an exact replay. */
template <class C>
__device__ __forceinline__ void cabinet_body(const gdg_seg_unit *U, int flip, int N, const C &c, const bool wt) {
    UNIT_PROLOGUE
    double *st = tmp + SEG_STASH;                    /* the seven capacitor voltages, fetched once, kept in LDS */
    if (seg_tid() < 7) st[seg_tid()] = ld_f64(as_global(U->ds) + seg_tid(), wt);
    double v[CHK];
    chunk_load(in, c, v);
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < 7; p++) {
        const double a = U->dp[p];
        double s = st[p];
        if (p < 3) onepole_reg<OP_DIFF_OLD>(v, c, a, s, tmp);
        else onepole_reg<OP_OLD>(v, c, a, s, tmp);
        if (c.last) st_f64(as_global(U->ds) + p, s, wt);
    }
#pragma unroll
    for (int i = 0; i < CHK; i++) v[i] = clip1(v[i]);
    chunk_store(out, c, v);
}
/* the batch block size: per section a dot product, a constant-coefficient scan (lin_scan) and the exact replay */
__device__ __forceinline__ void cabinet_full(const gdg_seg_unit *Ug, int flip, const bool wt) {
    UNIT_PROLOGUE
    const GDG_CONST gdg_seg_unit *U = uniform_unit(Ug);
    double *st = tmp + SEG_STASH;                    /* [0..6] the capacitor voltages, [16..22] the coefficients */
    const int tid = seg_tid();
    GDG_GLOBAL double *ds = as_global(U->ds);
    if (tid < 7) { st[tid] = ld_f64(ds + tid, wt); st[16 + tid] = U->dp[tid]; }
    tab_fetch(scr, U->tab, 7 * LT_SIZE);
    const ChunkT<true> c = full_chunk();
    double v[CHK];
    chunk_load(in, c, v);
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < 7; p++) {
        const double *tab = scr + p * LT_SIZE;
        const double a = st[16 + p];
        double s = lin_scan<false>(lin_chunk_map<false>(v, tab), tab, &st[p], tmp + (p & 1) * 2 * LX_SLOT, p);
        if (p < 3) {
#pragma unroll
            for (int i = 0; i < CHK; i++) { double diff = v[i] - s; s += diff * a; v[i] = diff; }          /* cabinet.go:114-118 */
        } else {
#pragma unroll
            for (int i = 0; i < CHK; i++) { double diff = v[i] - s; v[i] = s; s += diff * a; }             /* cabinet.go:135-139 */
        }
        if (c.last) st_f64(ds + p, s, wt);
    }
#pragma unroll
    for (int i = 0; i < CHK; i++) v[i] = clip1(v[i]);
    chunk_store(out, c, v);
}
UNIT_FN unit_cabinet(UNIT_ARGS) {
    if (N == CHK * SEG_T) cabinet_full(U, flip, wt);
    else cabinet_body(U, flip, N, my_chunk(N), wt);
}

/* ---- chorus: effects/chorus.go:19-131 ------------------------------------------------------------------
 * dp0 depth (0..10), dp1 angular speed, dp2 sample rate; jp0 history length C of the reference, jp1 ring mask (capacity - 1,
 * capacity = a power of two >= C + max frames); ds0 previousPhase; is0 ring write position.
 * The frame is appended to the ring FIRST: sample t of the virtual sequence [history | frame] (t = -C .. N - 1) is then
 * ring[(wp + t) & mask] whether it lies in the frame or before it -- one 16-byte load fetches the two neighbours of a fractional
 * delay, no LDS-or-ring branch, no modulo.  (The first version branched per tap between LDS and ring and between pair and
 * single loads: ~60 instructions and 6 branches per tap, 300 per sample.) */
/* wt (a frame per workgroup): what the next frame needs from this one is the appended ring and the two state cells -- both known as soon
 * as the frame is in the ring.  So the unit posts its counter right there and the next frame's chorus runs beside this one's taps: the
 * unit's serial part shrinks from the whole unit (10 us) to the append (2 us).  What bounds the overlap is the ring: the append of frame
 * f + s + 1 reaches cells frame f's taps may still read once (s + 2) N + C + 1 >= capacity, so a frame waits for frame f - s - 1 to be DONE
 * before it appends (s = 1 at 192 kHz: 32768 cells, C = 9600; s = 0, i.e. no overlap and the plain hand-off, at 96 kHz).  "Done" marks are
 * per frame (cell 2 + (f & 3)) and carry the launch's epoch, so a mark left by an earlier launch never passes for this one's.
 * *posted says to the caller that the unit's counter has been posted. */
#ifdef SEG_TILE
/* (the tile build's form: a tile of the frame per workgroup, two LFO states per thread; the other builds keep theirs, below, to the letter --
 * moving the LFO's rotation out of the sample loop cost the two-per-CU kernel 2.5 us per launch at 512 channels) */
UNIT_FN unit_chorus(UNIT_ARGS, const WaveGate &gate, int *posted) {
    UNIT_PROLOGUE
    const GDG_CONST gdg_seg_unit *Uc = uniform_unit(U);
    const double depth = Uc->dp[0], angular = Uc->dp[1], sr = Uc->dp[2];
    const int C = Uc->jp[0], mask = Uc->jp[1];
    GDG_GLOBAL double *ring = as_global(Uc->hist);
    GDG_GLOBAL int *is = as_global(Uc->is);
    GDG_GLOBAL double *ds = as_global(Uc->ds);
    const int wp = is[0];
    const double prev = ds[0];
#ifdef SEG_TILE
    /* a tile of the frame: its samples are g0 .. g0 + N - 1 of the frame; what it appends, the tiles behind it may tap (write-through) */
    const int g0 = s_tc.tile * SEG_N;
    const bool wts = true;
#else
    const int g0 = 0;
    const bool wts = wt;
#endif
    if (wt) {
        const int s_ok = ((mask + 1) - C - 2) / N - 1;
        if (s_ok >= 1 && s_ok <= 2 && gate.wf - s_ok - 1 >= 0) {
            const int fd = gate.wf - s_ok - 1;                        /* the frame whose taps this frame's append would run into */
            wave_wait(gate.cell + 1 + (fd & 3), gate.epoch * 32 + fd + 1, false, gate.d_error);
        }
    }
    /* 1. append the frame (pairs where possible); cell 0 is mirrored into the guard cell mask + 1 */
    if ((N & 1) == 0 && (wp & 1) == 0) {
        for (int i = 2 * (int)seg_tid(); i < N; i += 2 * SEG_T) {
            const int p = (wp + g0 + i) & mask;                     /* even, so p + 1 <= mask */
            seg_v2d v = { in[LX(i)], in[LX(i + 1)] };
            st_v2d((GDG_GLOBAL seg_v2d *)(ring + p), v, wts);
            if (p == 0) st_f64(ring + mask + 1, v.x, wts);
        }
    } else {
        for (int i = seg_tid(); i < N; i += SEG_T) {
            const int p = (wp + g0 + i) & mask;
            const double v = in[LX(i)];
            st_f64(ring + p, v, wts);
            if (p == 0) st_f64(ring + mask + 1, v, wts);
        }
    }
    if (wts) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     /* the sc1 pair stores are inline assembly: the compiler does not wait for them */
    __syncthreads();                                                /* the frame is in the ring (visible to the whole workgroup) */
#ifdef SEG_TILE
    /* the tiles behind this one tap what it appended (a tap reaches at most C + 1 samples back: never forwards); every wave has drained its
     * write-through stores in front of the barrier above, so one granule says "tile t is in the ring" */
    if (seg_tid() == 0) {
        if (s_tc.tile + 1 < SEG_TILES) tile_put(s_tc.xid, s_tc.tile, 0, 1.0);
        for (int lt = 0; lt < s_tc.tile; lt++) (void)tile_get(s_tc.xid, lt, 0);
        if (s_tc.tile > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      /* ONE buffer_inv sc1: this CU's L1 holds none of the ring's old lines any more */
    }
    __syncthreads();
#endif
    /* (this frame's taps read t = -C - 1 .. N - 1 relative to wp; frame f + j appends at wp + j N ..) */
    const int slack = wt ? ((mask + 1) - C - 2) / N - 1 : 0;        /* s: how many later frames may append while this one still reads */
    const bool early = wt && slack >= 1 && slack <= 2;
    if (early) {
        if (seg_tid() == 0) {
            st_f64(ds, fmod(prev + (angular * ((double)C / sr)), GO_MATH_TWO_PI), true);      /* the end-of-unit update below, same expression */
            st_i32(is, (wp + N) & mask, true);
        }
        wave_post(gate.cell - 1, gate.wf_next, gate.release);
        *posted = 1;
    }
    /* sin(zero_phase + j 2pi/5) by the angle-addition formula from ONE sincos (the five LFOs are 72 degrees apart):
     * differs from the reference's sin(fmod(zero_phase + j 2pi/5, 2pi)) by ~1e-16, i.e. ~1e-13 samples of delay */
    const double cj[5] = { 1.0, 0.30901699437494742410, -0.80901699437494742410, -0.80901699437494742410, 0.30901699437494742410 };
    const double sj[5] = { 0.0, 0.95105651629515357212, 0.58778525229247312917, -0.58778525229247312917, -0.95105651629515357212 };
    /* one sincos per thread: a thread's samples are SEG_T apart, so its LFO phase advances by a fixed angle from one to the
     * next and (sin, cos) follow by rotation (error ~1e-16 per step, eight steps).  (SEG_TILE: the thread of the GENERAL kernel that owns a
     * sample is its frame index mod 1024, and the value it uses is that thread's sincos rotated (index div 1024) times: the same here.) */
    constexpr int LFO_STRIDE = SEG_T * SEG_TILES;
    double sd, cd;
    sincos(angular * ((double)LFO_STRIDE / sr), &sd, &cd);
    auto lfo_start = [&](int index, double &s_, double &c_) {
        double time = (double)index / sr;
        double zero_phase = fmod_2pi(prev + (angular * time));
        sincos(zero_phase, &s_, &c_);
    };
    auto lfo_step = [&](double &s_, double &c_) {
        double sn = (s_ * cd) + (c_ * sd), cn = (c_ * cd) - (s_ * sd);
        s_ = sn; c_ = cn;
    };
    /* G samples at a time: first every address and ALL 5 G tap loads (16 bytes each), then the arithmetic -- written as two
     * loops because the compiler otherwise waits for each load right where it is used: 40 exposed L2 / HBM latencies per thread
     * were the whole cost of this unit (the ALU work is a third of it).  sv / cv: the LFO's (sin, cos) at the two samples */
    auto samples = [&](const int (&idx)[2], int G, const double (&sv)[2], const double (&cv)[2]) {
        seg_v2d v[2][5];
        double frs[2][5];
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if (g < G) {
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    double offset = depth * ((sv[g] * cj[j]) + (cv[g] * sj[j]));
                    double delay_time = 0.001 * (40.0 + offset);
                    double delay_samples = delay_time * sr;
                    /* chorus.go:63-90: early = floor, late = ceil, weights 1 - (d - early) and 1 - (late - d).  With fr = d - early
                     * (exact): fr != 0: late = early + 1 and the weights are exactly 1 - fr and fr; fr == 0: late = early, both 1 */
                    const double early = floor(delay_samples);
                    frs[g][j] = delay_samples - early;
                    const int t = g0 + idx[g] - (int)early - 1;     /* the older neighbour; t >= -C - 1, and t = -C - 1 only with fr == 0 */
                    v[g][j] = *(const GDG_GLOBAL seg_v2d *)(ring + ((wp + t) & mask));      /* (V[t], V[t + 1]) */
                }
            }
        }
        /* an integral delay (both weights 1, the sample counted twice) is rare -- depth 0 or a lucky phase -- and costs six selects per
         * tap: the wave asks once per sample pair whether any of its lanes has one and otherwise takes the plain interpolation (the same
         * operations in the same order: the same bits) */
        bool any_whole = false;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if (g < G) {
#pragma unroll
                for (int j = 0; j < 5; j++) any_whole |= frs[g][j] == 0.0;
            }
        }
        const bool plain = __builtin_amdgcn_ballot_w64(any_whole) == 0;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if (g < G) {
                double effected = 0.0;
                if (plain) {
#pragma unroll
                    for (int j = 0; j < 5; j++) {
                        const double fr = frs[g][j];
                        effected += 0.2 * (((1.0 - fr) * v[g][j].y) + (fr * v[g][j].x));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 5; j++) {
                        const double fr = frs[g][j];
                        const bool whole = fr == 0.0;
                        const double se = v[g][j].y, sl = whole ? v[g][j].y : v[g][j].x;
                        const double we = whole ? 1.0 : 1.0 - fr, wl = whole ? 1.0 : fr;
                        effected += 0.2 * ((we * se) + (wl * sl));
                    }
                }
                out[LX(idx[g])] = (0.5 * in[LX(idx[g])]) + (0.5 * effected);
            }
        }
    };
#ifdef SEG_TILE
    {
        /* local samples tid + 512 q, q = 0 .. 7: frame index g0 + tid + 512 q = (tid + 512 (q & 1)) + 1024 (4 tile + q / 2): two LFO states per
         * thread (the general kernel's threads tid and tid + 512), each already 4 x tile steps along when the tile starts */
        double sa, ca, sb, cb;
        lfo_start((int)seg_tid(), sa, ca);
        lfo_start((int)seg_tid() + SEG_T, sb, cb);
        for (int r = 0; r < 4 * s_tc.tile; r++) { lfo_step(sa, ca); lfo_step(sb, cb); }
#pragma unroll
        for (int q = 0; q < CHK; q += 2) {
            const int idx[2] = { (int)seg_tid() + q * SEG_T, (int)seg_tid() + (q + 1) * SEG_T };
            const double sv[2] = { sa, sb }, cv[2] = { ca, cb };
            samples(idx, 2, sv, cv);
            lfo_step(sa, ca);
            lfo_step(sb, cb);
        }
    }
#else
    double s0, c0;
    lfo_start((int)seg_tid(), s0, c0);
    if (N == CHK * SEG_T) {
#pragma unroll
        for (int q = 0; q < CHK; q += 2) {                         /* the batch block size: a fixed trip count */
            const int idx[2] = { (int)seg_tid() + q * SEG_T, (int)seg_tid() + (q + 1) * SEG_T };
            double sv[2], cv[2];
            sv[0] = s0; cv[0] = c0;
            lfo_step(s0, c0);
            sv[1] = s0; cv[1] = c0;
            lfo_step(s0, c0);
            samples(idx, 2, sv, cv);
        }
    } else {
        for (int i = seg_tid(); i < N; i += 2 * SEG_T) {
            const int idx[2] = { i, i + SEG_T };
            double sv[2], cv[2];
            sv[0] = s0; cv[0] = c0;
            lfo_step(s0, c0);
            sv[1] = s0; cv[1] = c0;
            lfo_step(s0, c0);
            samples(idx, (i + SEG_T < N) ? 2 : 1, sv, cv);
        }
    }
#endif
#ifdef SEG_TILE
    if (seg_tid() == 0 && s_tc.tile == SEG_TILES - 1) {
        /* the frame's last tile leaves the state: it has seen every other tile's "in the ring", which they posted after reading wp and the phase */
        st_f64(ds, fmod(prev + (angular * ((double)C / sr)), GO_MATH_TWO_PI), false);
        st_i32(is, (wp + GDG_MAX_FRAMES) & mask, false);
    }
#else
    if (seg_tid() == 0 && !early) {
        double buffer_time = (double)C / sr;          /* quirk: advances by the buffer length, not by N */
        st_f64(ds, fmod(prev + (angular * buffer_time), GO_MATH_TWO_PI), wt);
        st_i32(is, (wp + N) & mask, wt);
    }
#endif
    if (early) wave_post(gate.cell + 1 + (gate.wf & 3), gate.epoch * 32 + gate.wf + 1, false);      /* this frame's taps are done (nothing to publish: no release) */
}
#else
UNIT_FN unit_chorus(UNIT_ARGS, const WaveGate &gate, int *posted) {
    UNIT_PROLOGUE
    const GDG_CONST gdg_seg_unit *Uc = uniform_unit(U);
    const double depth = Uc->dp[0], angular = Uc->dp[1], sr = Uc->dp[2];
    const int C = Uc->jp[0], mask = Uc->jp[1];
    GDG_GLOBAL double *ring = as_global(Uc->hist);
    GDG_GLOBAL int *is = as_global(Uc->is);
    GDG_GLOBAL double *ds = as_global(Uc->ds);
    const int wp = is[0];
    const double prev = ds[0];
    if (wt) {
        const int s_ok = ((mask + 1) - C - 2) / N - 1;
        if (s_ok >= 1 && s_ok <= 2 && gate.wf - s_ok - 1 >= 0) {
            const int fd = gate.wf - s_ok - 1;                        /* the frame whose taps this frame's append would run into */
            wave_wait(gate.cell + 1 + (fd & 3), gate.epoch * 32 + fd + 1, false, gate.d_error);
        }
    }
    /* 1. append the frame (pairs where possible); cell 0 is mirrored into the guard cell mask + 1 */
    if ((N & 1) == 0 && (wp & 1) == 0) {
        for (int i = 2 * (int)seg_tid(); i < N; i += 2 * SEG_T) {
            const int p = (wp + i) & mask;                          /* even, so p + 1 <= mask */
            seg_v2d v = { in[LX(i)], in[LX(i + 1)] };
            st_v2d((GDG_GLOBAL seg_v2d *)(ring + p), v, wt);
            if (p == 0) st_f64(ring + mask + 1, v.x, wt);
        }
    } else {
        for (int i = seg_tid(); i < N; i += SEG_T) {
            const int p = (wp + i) & mask;
            const double v = in[LX(i)];
            st_f64(ring + p, v, wt);
            if (p == 0) st_f64(ring + mask + 1, v, wt);
        }
    }
    if (wt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* the sc1 pair stores are inline assembly: the compiler does not wait for them */
    __syncthreads();                                                /* the frame is in the ring (visible to the whole workgroup) */
    /* (this frame's taps read t = -C - 1 .. N - 1 relative to wp; frame f + j appends at wp + j N ..) */
    const int slack = wt ? ((mask + 1) - C - 2) / N - 1 : 0;        /* s: how many later frames may append while this one still reads */
    const bool early = wt && slack >= 1 && slack <= 2;
    if (early) {
        if (seg_tid() == 0) {
            st_f64(ds, fmod(prev + (angular * ((double)C / sr)), GO_MATH_TWO_PI), true);      /* the end-of-unit update below, same expression */
            st_i32(is, (wp + N) & mask, true);
        }
        wave_post(gate.cell - 1, gate.wf_next, gate.release);
        *posted = 1;
    }
    /* sin(zero_phase + j 2pi/5) by the angle-addition formula from ONE sincos (the five LFOs are 72 degrees apart):
     * differs from the reference's sin(fmod(zero_phase + j 2pi/5, 2pi)) by ~1e-16, i.e. ~1e-13 samples of delay */
    const double cj[5] = { 1.0, 0.30901699437494742410, -0.80901699437494742410, -0.80901699437494742410, 0.30901699437494742410 };
    const double sj[5] = { 0.0, 0.95105651629515357212, 0.58778525229247312917, -0.58778525229247312917, -0.95105651629515357212 };
    /* one sincos per thread: a thread's samples are SEG_T apart, so its LFO phase advances by a fixed angle from one to the
     * next and (sin, cos) follow by rotation (error ~1e-16 per step, eight steps) */
    double s0, c0, sd, cd;
    {
        double time = (double)seg_tid() / sr;
        double zero_phase = fmod_2pi(prev + (angular * time));
        sincos(zero_phase, &s0, &c0);
        sincos(angular * ((double)SEG_T / sr), &sd, &cd);
    }
    /* G samples at a time: first every address and ALL 5 G tap loads (16 bytes each), then the arithmetic -- written as two
     * loops because the compiler otherwise waits for each load right where it is used: 40 exposed L2 / HBM latencies per thread
     * were the whole cost of this unit (the ALU work is a third of it) */
    auto samples = [&](const int (&idx)[2], int G) {
        seg_v2d v[2][5];
        double frs[2][5];
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if (g < G) {
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    double offset = depth * ((s0 * cj[j]) + (c0 * sj[j]));
                    double delay_time = 0.001 * (40.0 + offset);
                    double delay_samples = delay_time * sr;
                    /* chorus.go:63-90: early = floor, late = ceil, weights 1 - (d - early) and 1 - (late - d).  With fr = d - early
                     * (exact): fr != 0: late = early + 1 and the weights are exactly 1 - fr and fr; fr == 0: late = early, both 1 */
                    const double early = floor(delay_samples);
                    frs[g][j] = delay_samples - early;
                    const int t = idx[g] - (int)early - 1;          /* the older neighbour; t >= -C - 1, and t = -C - 1 only with fr == 0 */
                    v[g][j] = *(const GDG_GLOBAL seg_v2d *)(ring + ((wp + t) & mask));      /* (V[t], V[t + 1]) */
                }
                double sn = (s0 * cd) + (c0 * sd), cn = (c0 * cd) - (s0 * sd);
                s0 = sn; c0 = cn;
            }
        }
        /* an integral delay (both weights 1, the sample counted twice) is rare -- depth 0 or a lucky phase -- and costs six selects per
         * tap: the wave asks once per sample pair whether any of its lanes has one and otherwise takes the plain interpolation (the same
         * operations in the same order: the same bits) */
        bool any_whole = false;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if (g < G) {
#pragma unroll
                for (int j = 0; j < 5; j++) any_whole |= frs[g][j] == 0.0;
            }
        }
        const bool plain = __builtin_amdgcn_ballot_w64(any_whole) == 0;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if (g < G) {
                double effected = 0.0;
                if (plain) {
#pragma unroll
                    for (int j = 0; j < 5; j++) {
                        const double fr = frs[g][j];
                        effected += 0.2 * (((1.0 - fr) * v[g][j].y) + (fr * v[g][j].x));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 5; j++) {
                        const double fr = frs[g][j];
                        const bool whole = fr == 0.0;
                        const double se = v[g][j].y, sl = whole ? v[g][j].y : v[g][j].x;
                        const double we = whole ? 1.0 : 1.0 - fr, wl = whole ? 1.0 : fr;
                        effected += 0.2 * ((we * se) + (wl * sl));
                    }
                }
                out[LX(idx[g])] = (0.5 * in[LX(idx[g])]) + (0.5 * effected);
            }
        }
    };
    if (N == CHK * SEG_T) {
#pragma unroll
        for (int q = 0; q < CHK; q += 2) {                         /* the batch block size: a fixed trip count */
            const int idx[2] = { (int)seg_tid() + q * SEG_T, (int)seg_tid() + (q + 1) * SEG_T };
            samples(idx, 2);
        }
    } else {
        for (int i = seg_tid(); i < N; i += 2 * SEG_T) {
            const int idx[2] = { i, i + SEG_T };
            samples(idx, (i + SEG_T < N) ? 2 : 1);
        }
    }
    if (seg_tid() == 0 && !early) {
        double buffer_time = (double)C / sr;          /* quirk: advances by the buffer length, not by N */
        st_f64(ds, fmod(prev + (angular * buffer_time), GO_MATH_TWO_PI), wt);
        st_i32(is, (wp + N) & mask, wt);
    }
    if (early) wave_post(gate.cell + 1 + (gate.wf & 3), gate.epoch * 32 + gate.wf + 1, false);      /* this frame's taps are done (nothing to publish: no release) */
}
#endif

#ifndef SEG_SUBSET
/* ---- flanger / phaser: effects/flanger.go:19-119, effects/phaser.go:19-125 --------------------------------
 * dp0 depth (0..1), dp1 angular speed, dp2 sr, dp3 1/sr, dp4 dry factor, dp5 wet factor; jp0 ring capacity */
UNIT_FN unit_flanger(UNIT_ARGS) {
    UNIT_PROLOGUE
    const double depth = U->dp[0], angular = U->dp[1], sr = U->dp[2], sr_inv = U->dp[3];
    const double mix_dry = U->dp[4], mix_wet = U->dp[5];
    const int C = U->jp[0], wp = U->is[0];
    const double prev = U->ds[0];
    const double *ring = U->hist;
    /* one sincos per thread, then rotation by the fixed phase step between a thread's samples (as in the chorus) */
    double s0, c0, sd, cd;
    {
        double time = (double)seg_tid() * sr_inv;
        sincos(fmod_2pi(prev + (angular * time)), &s0, &c0);
        sincos(angular * ((double)SEG_T * sr_inv), &sd, &cd);
    }
    for (int i = seg_tid(); i < N; i += SEG_T) {
        double offset = depth * s0;
        double delay_time = 0.001 * (depth + offset);
        double delay_samples = delay_time * sr;
        double delayed = frac_delay(in, ring, C, wp, i, delay_samples);
        out[LX(i)] = (mix_dry * in[LX(i)]) + (mix_wet * delayed);
        double sn = (s0 * cd) + (c0 * sd), cn = (c0 * cd) - (s0 * sd);
        s0 = sn; c0 = cn;
    }
    __syncthreads();
    if (seg_tid() == 0) {
        double duration = (double)C * sr_inv;
        U->ds[0] = fmod(prev + (angular * duration), GO_MATH_TWO_PI);
    }
    ring_append(U->hist, C, &U->is[0], in, N);
}

/* ---- delay: effects/delay.go:18-89 ------------------------------------------------------------------------
 * dp0 feedback factor, dp1 level factor; jp0 delay in samples (= ring capacity) */
UNIT_FN unit_delay(UNIT_ARGS) {
    UNIT_PROLOGUE
    const double feedback = U->dp[0], level = U->dp[1];
    const int D = U->jp[0], wp = U->is[0];
    const double *ring = U->hist;
    for (int i = seg_tid(); i < N; i += SEG_T) {
        int idx = i - D;
        double delayed = (idx >= 0) ? in[LX(idx)] : ring_read(ring, D, wp, idx);
        out[LX(i)] = clip1(level * (in[LX(i)] + (feedback * delayed)));
    }
    __syncthreads();
    ring_append(U->hist, D, &U->is[0], in, N);
}

#endif

/* ---- ring modulator: effects/ringmodulator.go:18-45.  dp0 phase increment per sample; ds0 phase ---------- */
UNIT_FN unit_ringmod(UNIT_ARGS) {
    UNIT_PROLOGUE
    const double fraction = U->dp[0], phase = U->ds[0];
    for (int i = seg_tid(); i < N; i += SEG_T) {
        double cur = fmod(phase + ((double)i * fraction), GO_MATH_TWO_PI);
        out[LX(i)] = sin(cur) * in[LX(i)];
    }
    __syncthreads();
    if (seg_tid() == 0) U->ds[0] = fmod(phase + ((double)N * fraction), GO_MATH_TWO_PI);
}

/* ---- tremolo: effects/tremolo.go:15-65 ----------------------------------------------------------------------
 * dp0 attenuation factor; jp0 samplesUnattenuated, jp1 samplesAttenuated (uint32); is0 attenuated, is1 inStateSince.
 * The counter FSM is data independent: thread 0 walks it in runs (a run ends where the reference flips state). */
UNIT_FN unit_tremolo(UNIT_ARGS) {
    UNIT_PROLOGUE
    const double fac = U->dp[0];
    int *runs = reinterpret_cast<int *>(scr);          /* pairs (start, attenuated), terminated by start = N */
    if (seg_tid() == 0) {
        const unsigned on = (unsigned)U->jp[0], off = (unsigned)U->jp[1];
        int att = U->is[0];
        unsigned cnt = (unsigned)U->is[1];
        int i = 0, nr = 0;
        const int max_runs = (SEG_SCR * 2 - 4) / 2;
        while (i < N) {
            unsigned thr = att ? off : on;
            if (cnt >= thr) { att = !att; cnt = 0; }
            unsigned thr2 = att ? off : on;
            unsigned k = (thr2 > cnt) ? thr2 - cnt : 1u;
            if (k < 1u) k = 1u;
            if (k > (unsigned)(N - i)) k = (unsigned)(N - i);
            if (nr < max_runs) { runs[2 * nr] = i; runs[2 * nr + 1] = att; nr++; }
            cnt += k;
            i += (int)k;
        }
        runs[2 * nr] = N; runs[2 * nr + 1] = 0;
        U->is[0] = att;
        U->is[1] = (int)cnt;
    }
    __syncthreads();
    for (int i = seg_tid(); i < N; i += SEG_T) {
        int r = 0;
        while (runs[2 * (r + 1)] <= i) r++;
        double v = in[LX(i)];
        if (runs[2 * r + 1]) v *= fac;
        out[LX(i)] = v;
    }
}

/* ---- signal generator: effects/signalgenerator.go:20-153 ------------------------------------------------------
 * ip2 signal type; dp0 input factor, dp1 signal factor, dp2 phase increment; ds0 phase; is0 LCG state, is1 LCG seeded */
__device__ __forceinline__ unsigned lcg_mulmod(unsigned a, unsigned b) {
    return (unsigned)(((unsigned long long)a * (unsigned long long)b) % 2147483647ull);
}
UNIT_FN unit_siggen(UNIT_ARGS) {
    UNIT_PROLOGUE
    const int type = U->ip[2];
    const double fac_in = U->dp[0], fac_sig = U->dp[1], inc = U->dp[2], phase = U->ds[0];
    if (type == 4) {                                    /* "noise": random/random.go LCG, seed 1337 */
        unsigned x0 = U->is[1] ? (unsigned)U->is[0] : (unsigned)((64979ull * 1337ull + 83ull) % 2147483647ull);
        __syncthreads();
        const int tid = seg_tid(), m = (N + SEG_T - 1) / SEG_T;
        const int c0 = min(N, tid * m), c1 = min(N, c0 + m);
        /* jump ahead: x_{c0} = 16807^c0 * x0 mod (2^31 - 1) */
        unsigned p = 1u, base = 16807u;
        for (int e = c0; e > 0; e >>= 1) { if (e & 1) p = lcg_mulmod(p, base); base = lcg_mulmod(base, base); }
        unsigned x = lcg_mulmod(p, x0);
        for (int i = c0; i < c1; i++) {
            x = lcg_mulmod(16807u, x);
            double r = (double)x / 2147483646.0;
            double uniform = (1.0 - (2.0 * r));
            out[LX(i)] = (fac_in * in[LX(i)]) + (fac_sig * uniform);
        }
        if (c1 == N && c0 < N) { U->is[0] = (int)x; U->is[1] = 1; }
        return;
    }
    for (int i = seg_tid(); i < N; i += SEG_T) {
        double cur = fmod(phase + ((double)i * inc), GO_MATH_TWO_PI);
        double signal = 0.0;
        switch (type) {
        case 0: signal = sin(cur); break;
        case 1: signal = (cur < M_PI) ? (GO_MATH_TWO_OVER_PI * cur) - 1.0 : 3.0 - (GO_MATH_TWO_OVER_PI * cur); break;
        case 2: { double d = M_PI - cur; signal = d < 0.0 ? -1.0 : (d > 0.0 ? 1.0 : 0.0); break; }
        case 3: signal = cur / M_PI; if (cur > M_PI) signal -= 2.0; break;
        }
        out[LX(i)] = (fac_in * in[LX(i)]) + (fac_sig * signal);
    }
    __syncthreads();
    if (seg_tid() == 0) {
        double ph = phase + ((double)N * inc);
        U->ds[0] = fmod(ph, GO_MATH_TWO_PI);
    }
}

/* ---- reverb: effects/reverb.go:41-116, :179-338 ------------------------------------------------------------------
 * dp0 dry, dp1 0.5 * wet; jp0..3 tap offsets, jp4 delay-line ring capacity (the longest tap + one frame of 8192: the in-place variant
 * appends the frame BEFORE it reads the taps), jp5..7 all-pass ring sizes D_k.
 * hist: [delay-line ring | all-pass 1 ring | all-pass 2 ring | all-pass 3 ring]; is0 delay-line wp, is1..3 all-pass ring pos.
 * An all-pass ring of size D delays by M = D - 1 samples (write at ptr, read at ptr + 1, reverb.go:51-58).
 * Here each ring keeps the last M values of p[n] = in[n] - g p[n - M]; o[n] = g p[n] + p[n - M].
 */
#define REVERB_QMAX (GDG_MAX_FRAMES / SEG_T)
#define REVERB_G 0.7                                  /* reverb.go:34 */

/* One Schroeder all-pass over the frame in `buf`, in place.  The recurrence p[n] = x[n] - g p[n - M] only couples samples
 * M apart, so sample chains r, r + M, r + 2M, ... are independent: a thread walks its chains in the reference's operation
 * order (no tiles, no barriers inside), starting from the ring value p[r - M] it fetched into `pm0` at the top of the unit
 * (one exposed HBM latency for the whole unit), and leaves the last p of every chain in the ring. */
template <int Q>
__device__ __forceinline__ void allpass_fetch(const double *ring, int M, int rp, int N, double (&pm0)[Q]) {
    const int cnt = min(M, N);
    /* unconditional loads (index clamped into the ring): a load under a lane condition is waited for at the end of its branch, which
     * turned these twelve "prefetches" into twelve exposed latencies */
#pragma unroll
    for (int q = 0; q < Q; q++) {
        int r = (int)seg_tid() + q * SEG_T;
        /* (the modulo costs ~40 instructions, but the conditional-subtraction form makes the in-place reverb of the two-per-CU build allocate 128
         * registers and 236 bytes of scratch instead of 120 / 132 -- and the KERNEL's register count is the largest of its units': every
         * segment, also those without a reverb, then ran 12 us slower, profiles/experiments/README.md r04) */
        pm0[q] = as_global(ring)[(r < cnt) ? (rp + r) % M : 0];
    }
}
template <int Q>
__device__ __forceinline__ void allpass_chains(double *buf, double *ring, int M, int rp, int N, const double (&pm0)[Q], int *rp_out, const bool wt = false) {
    const int cnt = min(M, N);
#pragma unroll
    for (int q = 0; q < Q; q++) {
        int r = (int)seg_tid() + q * SEG_T;
        if (r < cnt) {
            double pm = pm0[q], p;
            int n = r;
            do {
                p = buf[LX(n)] - (REVERB_G * pm);           /* reverb.go:51-58: write in - g * delayed, emit g * written + delayed */
                buf[LX(n)] = (REVERB_G * p) + pm;
                pm = p;
                n += M;
            } while (n < N);
            if (N >= M) st_f64(as_global(ring) + (n - N), p, wt);      /* the last M values of p, oldest first (n - M is this chain's last index) */
            else { int at = rp + r; if (at >= M) at -= M; st_f64(as_global(ring) + at, p, wt); }
        }
    }
    if (seg_tid() == 0) st_i32(as_global(rp_out), (N >= M) ? 0 : (rp + N) % M, wt);
    __syncthreads();
}
/* any ring size: fetch at use (exposes the latency; only sample rates far above 192 kHz come here) */
__device__ __attribute__((noinline)) void allpass_generic(double *buf, double *ring, int M, int rp, int N, int *rp_out, bool wt) {
    double pm0[REVERB_QMAX];
    allpass_fetch<REVERB_QMAX>(ring, M, rp, N, pm0);
    __syncthreads();                                        /* every old ring value is in a register before anyone overwrites the ring */
    if (wt) allpass_chains<REVERB_QMAX>(buf, ring, M, rp, N, pm0, rp_out, true);
    else allpass_chains<REVERB_QMAX>(buf, ring, M, rp, N, pm0, rp_out, false);
}

#ifdef SEG_FAST
/* The reverb IN PLACE (one frame buffer), for the shape the host checks (gdg_segf_supported + api_plan.cpp segf_unit_ok): N = 8192, every tap
 * at least a frame back (rates from 42.7 kHz), all-pass rings of at most 16 / 6 / 2 values per thread.  Same arithmetic as below; what moves
 * is WHERE values wait: the tapped sums and dry * x stay in registers while the all-passes run in the buffer, the frame joins the delay
 * line (from registers) as soon as every thread has consumed its taps, and the tap loads come in two batches (16 sixteen-byte loads in
 * flight per lane instead of 32: the register file is shared by two workgroups). */
/* wt (a frame per workgroup, WAVE): the two meetings of the general build's reverb (below) -- the delay line is handed on once this frame's
 * pairs are in it and its taps are loaded, the all-pass rings are waited for separately */
UNIT_FN unit_reverb(UNIT_ARGS, const WaveGate &gate, const int /* ahead: general build only */ = 0) {
    UNIT_PROLOGUE
    const GDG_CONST gdg_seg_unit *Uc = uniform_unit(U);
    const int tid = seg_tid();
    const double dry = Uc->dp[0], half_wet = Uc->dp[1];
    const double coeff[4] = { 0.1855, 0.18325, 0.17875, 0.17425 };
    int taps[4];
#pragma unroll
    for (int j = 0; j < 4; j++) taps[j] = Uc->jp[j];
    const int DL = Uc->jp[4];
    double *dl_ring = Uc->hist;
    int *is_state = Uc->is;
    const int dl_wp = as_global(is_state)[0];
    int M[3], rp[3];
    double *ring[3];
    {
        double *r = dl_ring + DL;
#pragma unroll
        for (int k = 0; k < 3; k++) { M[k] = Uc->jp[5 + k] - 1; rp[k] = wt ? 0 : as_global(is_state)[1 + k]; ring[k] = r; r += (M[k] > 0 ? M[k] : 0); }
    }
    constexpr int QA = REVERB_QMAX, QB = 3 * 1024 / SEG_T, QC = 1024 / SEG_T, NP = REVERB_QMAX / 2;      /* NP sample pairs per thread */
    GDG_GLOBAL double *g = as_global(dl_ring);
    const double g0 = g[0];
    /* 1. The frame's pairs (2p, 2p + 1), p = tid + q SEG_T, join the delay line and the tapped sums (reverb.go:65-116) take their place in the
     * buffer (the all-passes' input).  The ring holds one frame more than the longest tap (jp4 = taps[3] + 8192), so the cells the frame
     * overwrites are older than anything a tap of this frame reads and the taps -- all at least a frame back -- never meet the new cells:
     * appending and tapping commute. */
    /* the union of the four tap windows fits REVERB_QMAX pairs per thread: rates up to 204 kHz (below ~43 kHz the unit is not here at all) */
    const bool union_ok = taps[3] - taps[0] + N + 1 <= 2 * REVERB_QMAX * SEG_T;
    if (union_ok) {
        /* The four tap windows of a frame overlap: 4 x 64 KiB are read for a union
         * of (taps[3] - taps[0] + 8192) samples = 124 KiB at 192 kHz, and with 64 workgroups per XCD the overlaps no longer meet in L2 -- the
         * segment is HBM bound.  So every value of the union is loaded ONCE and handed to the (up to) four outputs it belongs to:
         * value r (relative to the frame's start) adds coeff[j] v to sample n = r + taps[j], tap after tap with a barrier between: a cell
         * receives its terms in the order j = 0, 1, 2, 3 -- tap 0 STORES (0.0 + c v: the reference's first addition), the others add: the
         * same sums in the same order, bit for bit, as the four-stream form below and the general kernel. */
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const int i0 = 2 * (tid + q * SEG_T);
            const double x0 = in[LX(i0)], x1 = in[LX(i0 + 1)];
            int p = dl_wp + i0;                             /* dl_wp < DL, i0 < N <= DL */
            if (p >= DL) p -= DL;
            if (p + 1 < DL) { seg_v2d v = { x0, x1 }; st_v2d((GDG_GLOBAL seg_v2d *)(g + p), v, wt); }
            else { st_f64(g + p, x0, wt); st_f64(g, x1, wt); }
        }
        __syncthreads();                                    /* every thread has its pair out of the buffer: tap 0 may overwrite it */
        const int rtop = N - 1 - taps[0];                   /* the newest value any tap of this frame reads (negative) */
        const int npairs = (taps[3] - taps[0] + N + 1) / 2; /* pair k = values (rtop - 2 k - 1, rtop - 2 k) */
        auto fetch = [&](int k) -> seg_v2d {
            const int kk = min(k, npairs - 1);
            const int r_old = rtop - 2 * kk - 1;            /* the pair's older value, relative to the frame's start */
            /* a union of an odd number of values: the last pair's older half would be value -taps[3] - 1 -- with DL = taps[3] + N that is the
             * cell this frame's sample N - 1 has just been stored to: never consumed (no tap reaches it), but a load racing a store all the
             * same.  That one lane loads the pair one cell up and hands its older half out as the newer one (a select, not a branch: a
             * branch here cost the unit 8 %) */
            const bool over = r_old < -taps[3];
            int pl = dl_wp + r_old + (over ? 1 : 0);        /* ring cell of the first value loaded; >= -DL */
            if (pl < 0) pl += DL;
            seg_v2d v;
            if (pl + 1 < DL) v = *(const GDG_GLOBAL seg_v2d *)(g + pl);
            else { v.x = g[pl]; v.y = g0; }
            v.y = over ? v.x : v.y;
            return v;
        };
        /* all of the thread's pairs first (up to 16 sixteen-byte loads in flight: nothing else of the unit is live yet), then the steps */
        seg_v2d pr[REVERB_QMAX];
#pragma unroll
        for (int it = 0; it < REVERB_QMAX; it++) pr[it] = fetch(it * SEG_T + tid);
        /* tap by tap: within one tap every value goes to a cell of its own (n = r + taps[j] is one to one), so FOUR barriers order the
         * four additions of every cell (walking the window in steps, tap order from step to step, took sixteen) */
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int it = 0; it < REVERB_QMAX; it++) {
                const int k = it * SEG_T + tid;
                const int rh = rtop - 2 * k;                /* k >= npairs: r + taps[j] < 0 for every tap, nothing is written */
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int n = rh - h + taps[j];
                    if (n >= 0 && n < N) {
                        const double term = coeff[j] * (h ? pr[it].x : pr[it].y);
                        out[LX(n)] = (j == 0 ? 0.0 : out[LX(n)]) + term;
                    }
                }
            }
            __syncthreads();
        }
    } else {
    /* per sample pair everything in the thread's OWN cells: the frame's pair joins the delay line, the
     * tapped sums (reverb.go:65-116) take its place in the buffer (the all-passes' input).  The ring holds one frame more than the longest
     * tap (jp4 = taps[3] + 8192), so the cells the frame overwrites are older than anything a tap of this frame reads and the taps -- all
     * at least a frame back -- never meet the new cells: appending and tapping commute, nobody waits for anybody.  A real loop of NB
     * batches of eight loads: unrolled, the compiler hoists all 32 loads to the top and parks them in scratch memory. */
    constexpr int NB = 4, PB = NP / NB;
#pragma unroll 1
    for (int b = 0; b < NB; b++) {
        seg_v2d tv[PB][4];
        int wrapm = 0;
#pragma unroll
        for (int qq = 0; qq < PB; qq++) {
            const int i0 = 2 * (tid + (b * PB + qq) * SEG_T);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int p = dl_wp + (i0 - taps[j]);
                if (p < 0) p += DL;
                const bool wraps = p + 1 >= DL;
                wrapm |= wraps ? (1 << (qq * 4 + j)) : 0;
                tv[qq][j] = *(const GDG_GLOBAL seg_v2d *)(g + (wraps ? DL - 2 : p));
            }
        }
#pragma unroll
        for (int qq = 0; qq < PB; qq++) {
            const int i0 = 2 * (tid + (b * PB + qq) * SEG_T);
            const double x0 = in[LX(i0)], x1 = in[LX(i0 + 1)];
            int p = dl_wp + i0;                             /* dl_wp < DL, i0 < N <= DL */
            if (p >= DL) p -= DL;
            if (p + 1 < DL) { seg_v2d v = { x0, x1 }; st_v2d((GDG_GLOBAL seg_v2d *)(g + p), v, wt); }
            else { st_f64(g + p, x0, wt); st_f64(g, x1, wt); }
            double pre0 = 0.0, pre1 = 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool wraps = (wrapm >> (qq * 4 + j)) & 1;
                const double c0 = wraps ? tv[qq][j].y : tv[qq][j].x, c1 = wraps ? g0 : tv[qq][j].y;
                pre0 += coeff[j] * c0;
                pre1 += coeff[j] * c1;
            }
            out[LX(i0)] = pre0;
            out[LX(i0 + 1)] = pre1;
        }
    }
    }
    if (tid == 0) st_i32(as_global(is_state), (dl_wp + N) % DL, wt);
    if (wt) {
        /* every tap of this frame is in registers / in the buffer (the union path ends on a barrier; the batches consumed their loads before
         * they stored): the delay line goes to the next frame, the all-pass rings are waited for */
        wave_post(gate.cell - 1, gate.wf_next, gate.release);
        wave_wait(gate.cell, gate.wf, false, gate.d_error);
#pragma unroll
        for (int k = 0; k < 3; k++) rp[k] = as_global(is_state)[1 + k];
    }
    /* 2. ring heads of the three all-passes, and the tapped sums back into registers (own cells, written above: the mix wants them after
     * the all-passes have replaced them in the buffer) */
    double pm_a[QA], pm_b[QB], pm_c[QC];
    allpass_fetch<QA>(ring[0], M[0], rp[0], N, pm_a);
    allpass_fetch<QB>(ring[1], M[1], rp[1], N, pm_b);
    allpass_fetch<QC>(ring[2], M[2], rp[2], N, pm_c);
    double dlr[REVERB_QMAX];
#pragma unroll
    for (int q = 0; q < REVERB_QMAX; q++) dlr[q] = out[LX(2 * (tid + (q >> 1) * SEG_T) + (q & 1))];
    /* every old all-pass value must have arrived before any thread overwrites the rings below: vmcnt(0), then the barrier (which also
     * completes the buffer) */
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    allpass_chains<QA>(out, ring[0], M[0], rp[0], N, pm_a, &is_state[1], wt);
    allpass_chains<QB>(out, ring[1], M[1], rp[1], N, pm_b, &is_state[2], wt);
    /* the dry samples come back from the delay line (this thread's own stores of step 1) while the last all-pass runs */
    seg_v2d xr[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) {
        int p = dl_wp + 2 * (tid + q * SEG_T);
        if (p >= DL) p -= DL;
        if (p + 1 < DL) xr[q] = *(const GDG_GLOBAL seg_v2d *)(g + p);
        else { xr[q].x = g[p]; xr[q].y = g[0]; }
    }
    allpass_chains<QC>(out, ring[2], M[2], rp[2], N, pm_c, &is_state[3], wt);
#pragma unroll
    for (int q = 0; q < REVERB_QMAX; q++) {
        const int i = 2 * (tid + (q >> 1) * SEG_T) + (q & 1);
        const double x = (q & 1) ? xr[q >> 1].y : xr[q >> 1].x;
        const double sum = dlr[q] + out[LX(i)];
        out[LX(i)] = clip1((dry * x) + (half_wet * sum));
    }
}
#else
/* wt (a frame per workgroup): the unit meets its predecessor frame twice.  Delay line: entered when the caller's wait on the unit's first
 * counter returns; the frame is appended FIRST (the ring is a frame longer than the longest tap: the cells it overwrites are the predecessor's
 * oldest tap window, consumed by then, and no tap of this frame reads them), the taps are loaded, gate.cell - 1 is posted -- the next frame may
 * append.  All-pass rings: wait on gate.cell, fetch, run, mix; the caller posts gate.cell.  So frame f + 1 taps while frame f runs its all-passes. */
/* The wet path AHEAD of the frame (round 6).  With every tap at least a frame back (rates from 42.7 kHz at the batch block size) the tapped
 * sums of a frame -- and with them the three all-passes, whose only input they are -- depend on EARLIER frames' inputs alone: all the unit
 * needs of the frame itself is the final mix  clip(dry x + 0.5 wet (tapped + all-passed))  (reverb.go:300-317).  Per-frame calls of few
 * channels leave 3/4 of the chip idle, and the reverb is the longest unit of its segment (18 of the 33 us a cabinet > reverb workgroup lives).
 * So the FIRST segment launch of such a call carries extra workgroups, one per reverb of the call's later segment steps, that make
 * tapped + all-passed  for this very frame (REVERB_AHEAD: the same expressions in the same order as the unit itself) beside the channels'
 * own workgroups, and the unit, when its turn comes behind the power amps, only mixes (REVERB_CONSUME).  Everything stays inside one call and
 * one stream: nothing speculative, no event.  (A side stream that made the NEXT call's wet paths cost more in cross-stream hops than the
 * unit takes: 153 -> 172 us per step at 64 channels, profiles/experiments/README.md r06.)  The extra workgroup leaves the all-pass rings
 * exactly as the whole unit would (nobody but this unit reads them, and its mix comes later in the same call) and the sums
 * tapped + all-passed  in a staging area behind the rings (even offset, 8192 values); the frame joins the delay line at the mix.
 * The host decides (api_plan.cpp reverb_ahead_ok, build_plan). */
enum { REVERB_FULL = 0, REVERB_AHEAD = 1, REVERB_CONSUME = 2 };
template <int MODE>
static __device__ __forceinline__ void reverb_general(UNIT_ARGS, const WaveGate &gate) {
    UNIT_PROLOGUE
    const GDG_CONST gdg_seg_unit *Uc = uniform_unit(U);
    const int tid = seg_tid();
    const double dry = Uc->dp[0], half_wet = Uc->dp[1];
    const double coeff[4] = { 0.1855, 0.18325, 0.17875, 0.17425 };
    int taps[4];
#pragma unroll
    for (int j = 0; j < 4; j++) taps[j] = Uc->jp[j];
    const int DL = Uc->jp[4];
    double *dl_ring = Uc->hist;
    int *is_state = Uc->is;
    const int dl_wp = as_global(is_state)[0];
    int M[3], rp[3];
    double *ring[3];
    double *ahead;                                      /* the staging area (see above) */
    {
        double *r = dl_ring + DL;
#pragma unroll
        for (int k = 0; k < 3; k++) { M[k] = Uc->jp[5 + k] - 1; rp[k] = wt ? 0 : as_global(is_state)[1 + k]; ring[k] = r; r += (M[k] > 0 ? M[k] : 0); }
        ahead = dl_ring + (((r - dl_ring) + 1) & ~(ptrdiff_t)1);
    }
    if (MODE == REVERB_CONSUME) {
        /* the sums made ahead (16-byte pairs, the mapping of the pair path below), the mix, the frame into the delay line */
        seg_v2d sm[REVERB_QMAX / 2];
#pragma unroll
        for (int q = 0; q < REVERB_QMAX / 2; q++) sm[q] = *(const GDG_GLOBAL seg_v2d *)(as_global(ahead) + 2 * (tid + q * SEG_T));
#pragma unroll
        for (int q = 0; q < REVERB_QMAX; q++) {
            const int i = 2 * (tid + (q >> 1) * SEG_T) + (q & 1);
            const double sum = (q & 1) ? sm[q >> 1].y : sm[q >> 1].x;
            out[LX(i)] = clip1((dry * in[LX(i)]) + (half_wet * sum));
        }
        __syncthreads();
        ring_append(dl_ring, DL, &is_state[0], in, N);
        return;
    }
    /* ring heads of the three all-passes first (in-order return: they are home before the tap loads below are consumed) */
    constexpr int QB = 3 * 1024 / SEG_T, QC = 1024 / SEG_T;       /* ring values per thread of the two short all-passes (up to 3072 / 1024 values) */
    const bool fast = min(M[1], N) <= QB * SEG_T && min(M[2], N) <= QC * SEG_T;
    double pm_a[REVERB_QMAX], pm_b[QB], pm_c[QC];
    if (!wt) {
        if (M[0] >= 1) allpass_fetch<REVERB_QMAX>(ring[0], M[0], rp[0], N, pm_a);
        if (fast) {
            if (M[1] >= 1) allpass_fetch<QB>(ring[1], M[1], rp[1], N, pm_b);
            if (M[2] >= 1) allpass_fetch<QC>(ring[2], M[2], rp[2], N, pm_c);
        }
    } else {
        ring_append(dl_ring, DL, &is_state[0], in, N, true);       /* dl_wp above is the write position BEFORE the frame: the taps below count from it */
    }
    double dlr[REVERB_QMAX];
    /* tapped delay line over the input history (reverb.go:65-116) */
    const bool pairs = N == GDG_MAX_FRAMES && taps[0] >= N && taps[1] >= N && taps[2] >= N && taps[3] >= N;
    if (pairs) {
        /* the usual case (taps 192..232 ms back, frames <= 43 ms, the batch block size): every tap lies in the HBM ring.  A thread
         * takes sample PAIRS (2p, 2p + 1), p = tid + q * SEG_T: one 16-byte load per tap and pair (8-byte aligned; the ring's
         * wrap between the two samples of a pair falls back to two loads), half the load instructions of the sample-wise walk.
         * dlr[2q], dlr[2q + 1] hold the pair; the final mix below uses the same mapping. */
        const GDG_GLOBAL double *g = as_global((const double *)dl_ring);
        /* all sixteen pair loads first, branch free (a pair that would wrap -- p = DL - 1, once per tap and frame -- is served from
         * the pair below it and from cell 0), then the arithmetic */
        const double g0 = g[0];
        seg_v2d tv[REVERB_QMAX / 2][4];
        int wrapm = 0;
#pragma unroll
        for (int q = 0; q < REVERB_QMAX / 2; q++) {
            const int i0 = 2 * (tid + q * SEG_T);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int p = dl_wp + (i0 - taps[j]);
                if (p < 0) p += DL;
                const bool wraps = p + 1 >= DL;
                wrapm |= wraps ? (1 << (q * 4 + j)) : 0;
                tv[q][j] = *(const GDG_GLOBAL seg_v2d *)(g + (wraps ? DL - 2 : p));
            }
        }
#pragma unroll
        for (int q = 0; q < REVERB_QMAX / 2; q++) {
            const int i0 = 2 * (tid + q * SEG_T);
            double pre0 = 0.0, pre1 = 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool wraps = (wrapm >> (q * 4 + j)) & 1;
                const double c0 = wraps ? tv[q][j].y : tv[q][j].x, c1 = wraps ? g0 : tv[q][j].y;
                pre0 += coeff[j] * c0;
                pre1 += coeff[j] * c1;
            }
            out[LX(i0)] = pre0;
            out[LX(i0 + 1)] = pre1;
            dlr[2 * q] = pre0;
            dlr[2 * q + 1] = pre1;
        }
    } else {
#pragma unroll
        for (int q = 0; q < REVERB_QMAX; q++) {
            int i = tid + q * SEG_T;
            double pre = 0.0;
            if (i < N) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int idx = i - taps[j];
                    double cur = 0.0;
                    if (idx >= 0) cur = in[LX(idx)];
                    else if (idx >= -DL) cur = ring_read(dl_ring, DL, dl_wp, idx);
                    pre += coeff[j] * cur;
                }
                out[LX(i)] = pre;
            }
            dlr[q] = pre;
        }
    }
    /* every old ring value must have arrived before any thread overwrites the rings below: vmcnt(0) (the tap loads were
     * needed here anyway), then the barrier */
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (wt) {
        wave_post(gate.cell - 1, gate.wf_next, gate.release);       /* the delay line is free for the next frame */
        wave_wait(gate.cell, gate.wf, false, gate.d_error);         /* the all-pass rings are ours */
#pragma unroll
        for (int k = 0; k < 3; k++) rp[k] = as_global(is_state)[1 + k];
        if (M[0] >= 1) allpass_fetch<REVERB_QMAX>(ring[0], M[0], rp[0], N, pm_a);
        if (fast) {
            if (M[1] >= 1) allpass_fetch<QB>(ring[1], M[1], rp[1], N, pm_b);
            if (M[2] >= 1) allpass_fetch<QC>(ring[2], M[2], rp[2], N, pm_c);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();                                            /* every old ring value is in a register before anyone overwrites the rings */
    }
    if (M[0] >= 1) allpass_chains<REVERB_QMAX>(out, ring[0], M[0], rp[0], N, pm_a, &is_state[1], wt);
    if (fast) {
        if (M[1] >= 1) allpass_chains<QB>(out, ring[1], M[1], rp[1], N, pm_b, &is_state[2], wt);
        if (M[2] >= 1) allpass_chains<QC>(out, ring[2], M[2], rp[2], N, pm_c, &is_state[3], wt);
    } else {
        if (M[1] >= 1) allpass_generic(out, ring[1], M[1], rp[1], N, &is_state[2], wt);
        if (M[2] >= 1) allpass_generic(out, ring[2], M[2], rp[2], N, &is_state[3], wt);
    }
    if (MODE == REVERB_AHEAD) {
        /* tapped + all-passed, the first addition of the mix (reverb.go:303), for the frame to come */
#pragma unroll
        for (int q = 0; q < REVERB_QMAX / 2; q++) {
            const int i0 = 2 * (tid + q * SEG_T);
            seg_v2d v = { dlr[2 * q] + out[LX(i0)], dlr[2 * q + 1] + out[LX(i0 + 1)] };
            *(GDG_GLOBAL seg_v2d *)(as_global(ahead) + i0) = v;
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < REVERB_QMAX; q++) {
        int i = pairs ? 2 * (tid + (q >> 1) * SEG_T) + (q & 1) : tid + q * SEG_T;
        if (i < N) {
            double sum = dlr[q] + out[LX(i)];
            out[LX(i)] = clip1((dry * in[LX(i)]) + (half_wet * sum));
        }
    }
    __syncthreads();
    if (!wt) ring_append(dl_ring, DL, &is_state[0], in, N);
}
#ifdef SEG_TILE
/* The tile build runs a reverb only as the MIX of a wet path an earlier launch of the call made (REVERB_CONSUME above, api_plan.cpp): a tile's
 * share of the sums, the mix, its share of the frame into the delay line.  Both tiles read the line's write position; the frame's last tile
 * advances it once tile 0 has said (a granule) that it has read it. */
UNIT_FN unit_reverb_mix_tile(UNIT_ARGS) {
    UNIT_PROLOGUE
    const GDG_CONST gdg_seg_unit *Uc = uniform_unit(U);
    const int tid = seg_tid(), g0 = s_tc.tile * SEG_N;
    const double dry = Uc->dp[0], half_wet = Uc->dp[1];
    const int DL = Uc->jp[4];
    GDG_GLOBAL double *g = as_global(Uc->hist);
    GDG_GLOBAL int *is_state = as_global(Uc->is);
    const int dl_wp = is_state[0];
    int rings = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) { const int M = Uc->jp[5 + k] - 1; rings += M > 0 ? M : 0; }
    const GDG_GLOBAL double *ahead = g + ((DL + rings + 1) & ~1) + g0;       /* the sums of this tile's samples (reverb_general: `ahead`) */
    seg_v2d sm[CHK / 2];
#pragma unroll
    for (int q = 0; q < CHK / 2; q++) sm[q] = *(const GDG_GLOBAL seg_v2d *)(ahead + 2 * (tid + q * SEG_T));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              /* every lane HOLDS the write position (and its sums) before the workgroup says so */
    __syncthreads();
    if (tid == 0 && s_tc.tile + 1 < SEG_TILES) tile_put(s_tc.xid, s_tc.tile, 0, 1.0);
#pragma unroll
    for (int q = 0; q < CHK; q++) {
        const int i = 2 * (tid + (q >> 1) * SEG_T) + (q & 1);
        const double sum = (q & 1) ? sm[q >> 1].y : sm[q >> 1].x;
        out[LX(i)] = clip1((dry * in[LX(i)]) + (half_wet * sum));
    }
    /* the tile's samples into the delay line (ring_append's pairs, offset by the tile's place in the frame) */
#pragma unroll
    for (int q = 0; q < CHK / 2; q++) {
        const int i = 2 * (tid + q * SEG_T);
        const int p = (dl_wp + g0 + i) % DL;
        const double a = in[LX(i)], b = in[LX(i + 1)];
        if (p + 1 < DL) { seg_v2d v = { a, b }; *(GDG_GLOBAL seg_v2d *)(g + p) = v; }
        else { g[p] = a; g[0] = b; }
    }
    if (tid == 0 && s_tc.tile == SEG_TILES - 1) {
        for (int lt = 0; lt < s_tc.tile; lt++) (void)tile_get(s_tc.xid, lt, 0);       /* every other tile has read the old position */
        is_state[0] = (dl_wp + GDG_MAX_FRAMES) % DL;
    }
}
#endif

/* ahead: one frame per launch (kernel argument); ip[7]: an earlier launch of this call made the unit's wet path (api_plan.cpp build_plan) */
UNIT_FN unit_reverb(UNIT_ARGS, const WaveGate &gate, const int ahead = 0) {
    if (!wt && ahead && uniform_unit(U)->ip[7]) reverb_general<REVERB_CONSUME>(U, flip, N, false, gate);
    else reverb_general<REVERB_FULL>(U, flip, N, wt, gate);
}
#endif

/* ---- generic one-pole section on an in-place sequence ----------------------------------------------------------------
 * state recurrence (every reference unit writes it the same way):  diff = v - s;  s += diff * a
 * emitted value:  DIFF_OLD v - s_old (high-pass, effects/cabinet.go:114-118)   OLD s_old (low-pass, cabinet.go:135-139)
 *                 NEW s_new (auto-wah low-pass, autowah.go:106-109)            DIFF_NEW v - s_new (coupling capacitor, fuzz.go:92-94)
 * VAR: the coefficient varies per sample and is read from abuf (auto-wah). */
template <int MODE, bool VAR>
__device__ __forceinline__ void onepole(double *buf, const double *abuf, double a_const, double *state, int N, double *tmp) {
    const int tid = seg_tid(), m = (N + SEG_T - 1) / SEG_T;
    const int c0 = min(N, tid * m), c1 = min(N, c0 + m);
    const double s0 = *state;                       /* read before the scan's barriers, rewritten after them */
    double A[1] = { 1.0 }, B[1] = { 0.0 }, Ap[1], Bp[1];
    for (int i = c0; i < c1; i++) {
        double a = VAR ? abuf[LX(i)] : a_const;
        A[0] *= (1.0 - a);
        double diff = buf[LX(i)] - B[0];
        B[0] += diff * a;
    }
    block_scan<1, false>(A, B, Ap, Bp, tmp);
    double s = apply_map<false>(Ap[0], Bp[0], s0);
    for (int i = c0; i < c1; i++) {
        double a = VAR ? abuf[LX(i)] : a_const;
        double v = buf[LX(i)];
        double diff = v - s;
        double s_old = s;
        s += diff * a;
        double o;
        if (MODE == OP_DIFF_OLD) o = diff;
        else if (MODE == OP_OLD) o = s_old;
        else if (MODE == OP_NEW) o = s;
        else o = v - s;
        buf[LX(i)] = o;
    }
    if (c1 == N && c0 < N) *state = s;
}

/* ---- workgroup scan of small integer maps (FSM compositions) ------------------------------------------------------------ */
struct IMap { int f[5]; };

template <class Compose>
__device__ __forceinline__ IMap block_scan_imap(IMap mine, IMap identity, Compose comp, int *itmp) {
    const int tid = seg_tid(), lane = tid & 63, wave = tid >> 6, row = lane >> 4;
    IMap a = mine;
    {
        IMap o;
#pragma unroll
        for (int k = 0; k < 5; k++) o.f[k] = row_shr_i<1>(a.f[k]);
        if ((lane & 15) >= 1) a = comp(o, a);       /* first the lower lanes, then this one */
#pragma unroll
        for (int k = 0; k < 5; k++) o.f[k] = row_shr_i<2>(a.f[k]);
        if ((lane & 15) >= 2) a = comp(o, a);
#pragma unroll
        for (int k = 0; k < 5; k++) o.f[k] = row_shr_i<4>(a.f[k]);
        if ((lane & 15) >= 4) a = comp(o, a);
#pragma unroll
        for (int k = 0; k < 5; k++) o.f[k] = row_shr_i<8>(a.f[k]);
        if ((lane & 15) >= 8) a = comp(o, a);
    }
    IMap t0, t1, t2, rowpre = identity;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        t0.f[k] = __builtin_amdgcn_readlane(a.f[k], 15);
        t1.f[k] = __builtin_amdgcn_readlane(a.f[k], 31);
        t2.f[k] = __builtin_amdgcn_readlane(a.f[k], 47);
    }
    t1 = comp(t0, t1);
    t2 = comp(t1, t2);
    if (row == 1) rowpre = t0; else if (row == 2) rowpre = t1; else if (row == 3) rowpre = t2;
    if (row != 0) a = comp(rowpre, a);
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 5; k++) itmp[wave * 5 + k] = a.f[k];
    }
    __syncthreads();
    IMap e;
#pragma unroll
    for (int k = 0; k < 5; k++) e.f[k] = row_shr_i<1>(a.f[k]);
    if ((lane & 15) == 0) e = rowpre;
    IMap w = identity;
    if (lane < SEG_WAVES) {
#pragma unroll
        for (int k = 0; k < 5; k++) w.f[k] = itmp[lane * 5 + k];
    }
    {
        IMap o;
#pragma unroll
        for (int k = 0; k < 5; k++) o.f[k] = row_shr_i<1>(w.f[k]);
        if ((lane & 15) >= 1) w = comp(o, w);
#pragma unroll
        for (int k = 0; k < 5; k++) o.f[k] = row_shr_i<2>(w.f[k]);
        if ((lane & 15) >= 2) w = comp(o, w);
#pragma unroll
        for (int k = 0; k < 5; k++) o.f[k] = row_shr_i<4>(w.f[k]);
        if ((lane & 15) >= 4) w = comp(o, w);
#pragma unroll
        for (int k = 0; k < 5; k++) o.f[k] = row_shr_i<8>(w.f[k]);
        if ((lane & 15) >= 8) w = comp(o, w);
    }
    const int wsel = __builtin_amdgcn_readfirstlane(wave > 0 ? wave - 1 : 0);
    IMap wp;
#pragma unroll
    for (int k = 0; k < 5; k++) wp.f[k] = __builtin_amdgcn_readlane(w.f[k], wsel);
    if (wave == 0) wp = identity;
    e = comp(wp, e);
    __syncthreads();
    return e;
}

#ifndef SEG_SUBSET        /* from here to the segment kernel: units of the general kernel only */
/* ---- fuzz without oversampling: effects/fuzz.go:24-108 ------------------------------------------------------------------------
 * ip0 follow; dp0 bias, dp1 gain, dp2 fuzz, dp3 1 - fuzz, dp4 level, dp5 exp(-20/sr), dp6 1 - dp5; ds0 envelope, ds1 coupling cap */
UNIT_FN unit_fuzz(UNIT_ARGS) {
    UNIT_PROLOGUE
    envelope_to(in, out, N, U->ip[0], U->dp[5], U->dp[6], &U->ds[0], tmp);
    const double bias = U->dp[0], gain = U->dp[1], fuzz = U->dp[2], fuzz_inv = U->dp[3], level = U->dp[4];
    const int tid = seg_tid(), m = (N + SEG_T - 1) / SEG_T;
    const int c0 = min(N, tid * m), c1 = min(N, c0 + m);
    for (int i = c0; i < c1; i++) {
        double sample = in[LX(i)];
        double bias_voltage = bias * out[LX(i)];
        double pre = clip1(gain * (sample - bias_voltage));
        double fuzz_fraction = fuzz * pre;
        double clean_fraction = fuzz_inv * sample;
        out[LX(i)] = fuzz_fraction + clean_fraction;
    }
    onepole<OP_DIFF_NEW, false>(out, nullptr, U->dp[6], &U->ds[1], N, tmp);
    for (int i = c0; i < c1; i++) out[LX(i)] = level * clip1(out[LX(i)]);
}

/* ---- fuzz WITH 2x / 4x oversampling: effects/fuzz.go:113-174 around fuzz.go:24-108 at the oversampled rate -----------------
 * Same tiling as unit_shaper, but the shaper has memory (envelope follower + coupling capacitor run at f * sr), so every
 * tile runs two workgroup scans over its f * S_out new oversampled samples, with the state carried from tile to tile in LDS.
 * dp5 / dp6 are exp(-20 / (f sr)) and its complement; hist as in unit_shaper. */
__device__ __forceinline__ void lin_chunk(int cnt, int &c0, int &c1) {
    const int m = (cnt + SEG_T - 1) / SEG_T;
    c0 = min(cnt, (int)seg_tid() * m);
    c1 = min(cnt, c0 + m);
}

/* follower over a linear LDS array: v[i] (input) -> env[i] in e[i]; state in *st (LDS) */
__device__ __forceinline__ void envelope_lin(const double *v, double *e, int cnt, int follow, double d_inv, double d, double *st, double *tmp) {
    int c0, c1;
    lin_chunk(cnt, c0, c1);
    const double s0 = *st;
    double A[1] = { 1.0 }, B[1] = { 0.0 }, Ap[1], Bp[1];
    if (follow == 0) {
        for (int i = c0; i < c1; i++) { A[0] *= d_inv; B[0] *= d_inv; double a = fabs(v[i]); if (a > B[0]) B[0] = a; }
        block_scan<1, true>(A, B, Ap, Bp, tmp);
        double s = apply_map<true>(Ap[0], Bp[0], s0);
        for (int i = c0; i < c1; i++) { s *= d_inv; double a = fabs(v[i]); if (a > s) s = a; e[i] = s; }
        if (c1 == cnt && c0 < cnt) *st = s;
    } else if (follow == 1) {
        for (int i = c0; i < c1; i++) { A[0] *= d_inv; double diff = fabs(v[i]) - B[0]; B[0] += diff * d; }
        block_scan<1, false>(A, B, Ap, Bp, tmp);
        double s = apply_map<false>(Ap[0], Bp[0], s0);
        for (int i = c0; i < c1; i++) { double diff = fabs(v[i]) - s; s += diff * d; e[i] = s; }
        if (c1 == cnt && c0 < cnt) *st = s;
    } else {
        __syncthreads();
        for (int i = c0; i < c1; i++) e[i] = 1.0;
        if (c1 == cnt && c0 < cnt) *st = 1.0;
    }
    __syncthreads();
}

/* decimating FIR over a LINEAR tile (the fuzz needs the oversampled stream in time order for its scans): taps through the
 * scalar cache, k ascending as in the reference */
template <int F>
__device__ __forceinline__ void fir_decimate_linear(const double *scr, const double *taps_generic, double *out, int o0, int S_out) {
    constexpr int TAPS = (F == 2) ? 77 : 155;
    const GDG_CONST double *taps = (const GDG_CONST double *)uniform_ptr(taps_generic);
    for (int o = seg_tid(); o < S_out; o += SEG_T) {
        const double *base = scr + F * o + (TAPS - 1);
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < TAPS; k++) acc += taps[k] * base[-k];
        out[LX(o0 + o)] = ATTENUATION_HALF_DECIBEL * clip1(acc);
    }
}

/* Frames that are a multiple of the big tile (the batch block size): the oversampled stream of a tile is exactly CHK = 8 samples per
 * thread, so the follower and the coupling capacitor run as constant-coefficient scans on register chunks (lin_scan), the shaped
 * samples go to the polyphase staging area in the OUTPUT frame buffer and the register-blocked decimator of the memoryless shapers
 * finishes the tile in place.  Tiles go FORWARD (the follower's state does), so the last eight inputs of a tile are saved before its
 * results overwrite them.  4 (4x) / 2 (2x) tiles per frame instead of 21 / 11 with ten barriers each and 40 % of the lanes idle. */
template <int F>
__device__ __forceinline__ void fuzz_os_tiles(const gdg_seg_unit *Ug, double *in, double *stage, double *scr, double *tmp, const gdg_os_tables &os, int N) {
    constexpr int TAPS = OsCfg<F>::TAPS, BACK = OsCfg<F>::BACK, R = OsCfg<F>::R, TILE = OsCfg<F>::S, PH = OsCfg<F>::PH;
    static_assert(F * R == CHK, "a thread's share of a tile is one register chunk");
    const GDG_CONST gdg_seg_unit *U = uniform_unit(Ug);
    const int tid = seg_tid(), follow = U->ip[0];
    const double bias = U->dp[0], gain = U->dp[1], fuzz = U->dp[2], fuzz_inv = U->dp[3], level = U->dp[4];
    const double d_inv = U->dp[5], d = U->dp[6];
    const GDG_CONST double *tp = (const GDG_CONST double *)uniform_ptr(F == 2 ? os.tapsP2 : os.tapsP4);
    const GDG_CONST double *lw = (const GDG_CONST double *)uniform_ptr(F == 2 ? os.lanczos2 : os.lanczos4);
    GDG_GLOBAL double *hist = as_global(U->hist);
    GDG_GLOBAL double *ds = as_global(U->ds);
    double *tab_env = scr, *tab_cap = scr + LT_SIZE, *carry8 = scr + 2 * LT_SIZE, *old_tail = scr + 2 * LT_SIZE + 8;
    /* the two carried states, double buffered by tile: a fast wave finishes a tile's replay (and stores the new state) before a
     * slow one has read the old state inside the same scan */
    double *st = tmp + SEG_STASH + 16;               /* [2 * (tile & 1) + {0: envelope, 1: coupling capacitor}] */
    if (tid == 0) { st[0] = ds[0]; st[1] = ds[1]; }
    tab_fetch(tab_env, U->tab, 2 * LT_SIZE);                    /* [follower | coupling capacitor]; tab_cap = tab_env + LT_SIZE */
    for (int q = tid; q < TAPS - 1 + 8; q += SEG_T) { if (q < 8) carry8[q] = hist[q]; else old_tail[q - 8] = hist[q]; }
    __syncthreads();
    int parity = 0, tile = 0;
    for (int o0 = 0; o0 < N; o0 += TILE, tile++) {
        const int I0 = o0 - BACK;
        const double *st_in = st + 2 * (tile & 1);
        double *st_out = st + 2 * ((tile + 1) & 1);
        /* inputs before this tile: the eight saved ones (previous tile's last inputs, or the previous call's) */
        auto s_at = [&](int k) -> double { return k >= o0 ? in[LX(k)] : carry8[k - (o0 - 8)]; };
        /* 1. this thread's R inputs -> CHK oversampled samples (oversampling.go:54-115) */
        double w[CHK];
#pragma unroll
        for (int g = 0; g < R; g++) {
            const int i = o0 + R * tid + g;
            double w6[6];
#pragma unroll
            for (int t = 0; t < 6; t++) w6[t] = s_at(i - 6 + t);
            w[F * g] = w6[2];
#pragma unroll
            for (int r = 1; r < F; r++) {
                double up = 0.0;
#pragma unroll
                for (int t = 0; t < 6; t++) up += w6[t] * lw[(r - 1) * 6 + t];
                w[F * g + r] = up;
            }
        }
        double keep8 = 0.0;
        if (tid < 8) keep8 = in[LX(o0 + TILE - 8 + tid)];
        /* 2. follower at the oversampled rate, fuzz curve, coupling capacitor (fuzz.go:47-106) */
        double e[CHK];
        if (follow == 0) {
            double s = lin_scan<true>(lin_chunk_map<true>(w, tab_env), tab_env, &st_in[0], tmp + parity * 2 * LX_SLOT);
#pragma unroll
            for (int i = 0; i < CHK; i++) { s *= d_inv; double q = fabs(w[i]); if (q > s) s = q; e[i] = s; }
            if (tid == SEG_T - 1) st_out[0] = s;
        } else if (follow == 1) {
            double s = lin_scan<false>(lin_chunk_map<false, true>(w, tab_env), tab_env, &st_in[0], tmp + parity * 2 * LX_SLOT);
#pragma unroll
            for (int i = 0; i < CHK; i++) { double diff = fabs(w[i]) - s; s += diff * d; e[i] = s; }
            if (tid == SEG_T - 1) st_out[0] = s;
        } else {
#pragma unroll
            for (int i = 0; i < CHK; i++) e[i] = 1.0;
            if (tid == SEG_T - 1) st_out[0] = 1.0;
        }
        parity ^= 1;
#pragma unroll
        for (int i = 0; i < CHK; i++) {
            double sample = w[i];
            double bias_voltage = bias * e[i];
            double pre = clip1(gain * (sample - bias_voltage));
            w[i] = (fuzz * pre) + (fuzz_inv * sample);
        }
        {
            double s = lin_scan<false>(lin_chunk_map<false>(w, tab_cap), tab_cap, &st_in[1], tmp + parity * 2 * LX_SLOT);
#pragma unroll
            for (int i = 0; i < CHK; i++) { double diff = w[i] - s; s += diff * d; w[i] = level * clip1(w[i] - s); }
            if (tid == SEG_T - 1) st_out[1] = s;
        }
        parity ^= 1;
        /* 3. polyphase staging: slot of input i is i - I0; the BACK slots before the tile come from the previous tile / call */
#pragma unroll
        for (int g = 0; g < R; g++)
#pragma unroll
            for (int r = 0; r < F; r++) stage[r * PH + BACK + R * tid + g] = w[F * g + r];
        if (o0 == 0) {
            for (int q = tid; q < F * BACK; q += SEG_T) {
                const int r = q / BACK, idx = q - r * BACK, m = F * (I0 + idx) + r;          /* m < 0: the previous call's tail */
                stage[r * PH + idx] = (m >= -(TAPS - 1)) ? old_tail[(TAPS - 1) + m] : 0.0;
            }
        }
        for (int q = tid; q < F * (OsCfg<F>::NC - BACK); q += SEG_T) {                        /* the walk's overrun: zeros */
            const int r = q / (OsCfg<F>::NC - BACK), idx = q - r * (OsCfg<F>::NC - BACK);
            if (BACK + TILE + idx < PH) stage[r * PH + BACK + TILE + idx] = 0.0;
        }
        __syncthreads();
        if (tid < 8) carry8[tid] = keep8;
        if (o0 + TILE >= N) {
            for (int q = tid; q < TAPS - 1; q += SEG_T) {
                const int m = F * N - (TAPS - 1) + q;
                const int i = m / F, r = m - i * F;                                              /* m >= 0: N >= TILE > TAPS */
                hist[8 + q] = stage[r * PH + (i - I0)];
            }
        }
        /* 4. decimate in place */
        double acc[R];
        os_decimate<F>(stage, tp, tid, acc);
#pragma unroll
        for (int j = 0; j < R; j++) in[LX(o0 + R * tid + j)] = ATTENUATION_HALF_DECIBEL * clip1(acc[j]);
        /* 5. slide: the last BACK slots of every phase become the head of the next tile */
        double keep = 0.0;
        const bool mine = tid < F * BACK;
        const int kr = tid / BACK, kidx = tid - kr * BACK;
        if (mine) keep = stage[kr * PH + TILE + kidx];
        __syncthreads();
        if (mine) stage[kr * PH + kidx] = keep;
        __syncthreads();
    }
    if (tid < 8) hist[tid] = carry8[tid];
    if (tid == 0) { ds[0] = st[2 * (tile & 1)]; ds[1] = st[2 * (tile & 1) + 1]; }
}

/* returns 1 when the result is in the INPUT buffer */
__device__ __attribute__((noinline)) int unit_fuzz_os(UNIT_ARGS, const gdg_os_tables &os) {
    UNIT_PROLOGUE
    if (U->jp[0] == 4 && N % OsCfg<4>::S == 0) { fuzz_os_tiles<4>(U, in, out, scr, tmp, os, N); return 1; }
    if (U->jp[0] == 2 && N % OsCfg<2>::S == 0) { fuzz_os_tiles<2>(U, in, out, scr, tmp, os, N); return 1; }
    const int tid = seg_tid();
    const int f = U->jp[0];
    const int follow = U->ip[0];
    const double bias = U->dp[0], gain = U->dp[1], fuzz = U->dp[2], fuzz_inv = U->dp[3], level = U->dp[4];
    const double d_inv = U->dp[5], d = U->dp[6];
    const int TAPS = (f == 2) ? 77 : 155;
    const double *taps = (f == 2) ? os.taps2 : os.taps4;
    const double *lw = (f == 2) ? os.lanczos2 : os.lanczos4;
    double *hist = U->hist;
    /* scr: [w: TAPS-1 old + f*S_out new | e: f*S_out envelope values]; the two carried states live in tmp[120..121] */
    const int TILE = ((SEG_SCR - TAPS) / 2) / f;
    double *st = tmp + SEG_STASH + 16;
    if (tid == 0) { st[0] = U->ds[0]; st[1] = U->ds[1]; }
    __syncthreads();
    auto s_at = [&](int k) -> double { return k >= 0 ? in[LX(k)] : hist[8 + k]; };
    for (int o0 = 0; o0 < N; o0 += TILE) {
        const int S_out = min(TILE, N - o0);
        const int cnt = f * S_out;
        double *w = scr + (TAPS - 1), *e = scr + (TAPS - 1) + f * TILE;
        /* tail: the last TAPS-1 outputs of the shaper (previous call for the first tile, previous tile otherwise) */
        if (o0 == 0) for (int q = tid; q < TAPS - 1; q += SEG_T) scr[q] = hist[8 + q];
        for (int q = tid; q < cnt; q += SEG_T) {
            int m = f * o0 + q;
            int i = m / f, r = m - i * f;
            double up;
            if (r == 0) up = s_at(i - 4);
            else {
                const double *wq = lw + (r - 1) * 6;
                up = 0.0;
#pragma unroll
                for (int t = 0; t < 6; t++) up += s_at(i - 6 + t) * wq[t];
            }
            w[q] = up;
        }
        __syncthreads();
        envelope_lin(w, e, cnt, follow, d_inv, d, &st[0], tmp);
        int c0, c1;
        lin_chunk(cnt, c0, c1);
        for (int q = c0; q < c1; q++) {
            double sample = w[q];
            double bias_voltage = bias * e[q];
            double pre = clip1(gain * (sample - bias_voltage));
            w[q] = (fuzz * pre) + (fuzz_inv * sample);
        }
        {   /* coupling capacitor: diff = p - c; c += diff d; out = p - c (fuzz.go:92-94) */
            const double s0 = st[1];
            double A[1] = { 1.0 }, B[1] = { 0.0 }, Ap[1], Bp[1];
            for (int q = c0; q < c1; q++) { A[0] *= (1.0 - d); double diff = w[q] - B[0]; B[0] += diff * d; }
            block_scan<1, false>(A, B, Ap, Bp, tmp);
            double s = apply_map<false>(Ap[0], Bp[0], s0);
            for (int q = c0; q < c1; q++) { double diff = w[q] - s; s += diff * d; w[q] = level * clip1(w[q] - s); }
            if (c1 == cnt && c0 < cnt) st[1] = s;
        }
        __syncthreads();
        if (f == 2) fir_decimate_linear<2>(scr, taps, out, o0, S_out);
        else fir_decimate_linear<4>(scr, taps, out, o0, S_out);
        __syncthreads();
        /* slide: the last TAPS-1 shaped samples become the head of the next tile (and of the next call) */
        double keep = 0.0;
        const bool mine = tid < TAPS - 1;
        if (mine) keep = scr[cnt + tid];
        __syncthreads();
        if (mine) { scr[tid] = keep; if (o0 + TILE >= N) hist[8 + tid] = keep; }
        __syncthreads();
    }
    double keep8 = 0.0;
    if (tid < 8) keep8 = s_at(N - 8 + tid);
    __syncthreads();
    if (tid < 8) hist[tid] = keep8;
    if (tid == 0) { U->ds[0] = st[0]; U->ds[1] = st[1]; }
    return 0;
}

/* ---- auto-yoy: effects/autoyoy.go:19-157 ----------------------------------------------------------------------------------------
 * ip0 follow; dp0 level A, dp1 level B, dp2 depth A, dp3 depth B, dp4 slope, dp5 exp(-20/sr), dp6 1 - dp5, dp7 sr; jp0 ring capacity */
UNIT_FN unit_autoyoy(UNIT_ARGS) {
    UNIT_PROLOGUE
    envelope_to(in, out, N, U->ip[0], U->dp[5], U->dp[6], &U->ds[0], tmp);
    const double la = U->dp[0], lb = U->dp[1], da = U->dp[2], db = U->dp[3], slope = U->dp[4], sr = U->dp[7];
    const int C = U->jp[0], wp = U->is[0];
    const double *ring = U->hist;
    const int tid = seg_tid(), m = (N + SEG_T - 1) / SEG_T;
    const int c0 = min(N, tid * m), c1 = min(N, c0 + m);
    for (int i = c0; i < c1; i++) {
        double level = 20.0 * log10(out[LX(i)]);
        double delay_fac;
        if (level <= la) delay_fac = da;
        else if (level >= lb) delay_fac = db;
        else delay_fac = da + (slope * (level - la));
        double delay_time = 0.01 * delay_fac;
        double delay_samples = delay_time * sr;
        double delayed = frac_delay(in, ring, C, wp, i, delay_samples);
        out[LX(i)] = (0.5 * in[LX(i)]) + (0.5 * delayed);
    }
    __syncthreads();
    ring_append(U->hist, C, &U->is[0], in, N);
}

/* ---- auto-wah: effects/autowah.go:20-130 -------------------------------------------------------------------------------------------
 * ip0 follow; dp0 level A, dp1 level B, dp2 freq A, dp3 freq B, dp4 slope, dp5 exp(-20/sr), dp6 1 - dp5, dp7 sr;
 * ds0 envelope, ds1..8 hcv, ds9..16 lcv.  CLOBBERS its input buffer (it is free: the next unit overwrites it anyway). */
/* the batch block size: the frame chunk, the per-sample coefficients and the running signal stay in registers through all sixteen
 * time-varying sections (the generic version below walks two LDS arrays four times per section and is LDS bound) */
__device__ __forceinline__ void autowah_full(const gdg_seg_unit *Ug, int flip) {
    UNIT_PROLOGUE
    const GDG_CONST gdg_seg_unit *U = uniform_unit(Ug);
    double *st = tmp + SEG_STASH;                    /* [0] envelope, [1..8] hcv, [9..16] lcv */
    GDG_GLOBAL double *ds = as_global(U->ds);
    if (seg_tid() < 17) st[seg_tid()] = ds[seg_tid()];
    const double la = U->dp[0], lb = U->dp[1], fa = U->dp[2], fb = U->dp[3], slope = U->dp[4], sr = U->dp[7];
    const ChunkT<true> c = full_chunk();
    double v[CHK], al[CHK];
    chunk_load(in, c, v);
    __syncthreads();
    {
        double e[CHK], s = st[0];
        envelope_reg(v, e, c, U->ip[0], U->dp[5], U->dp[6], s, tmp);
        if (c.last) ds[0] = s;
#pragma unroll
        for (int i = 0; i < CHK; i++) {
            double level = 20.0 * log10(e[i]);
            double frequency;
            if (level <= la) frequency = fa;
            else if (level >= lb) frequency = fb;
            else frequency = fa + (slope * (level - la));
            double arg = -frequency / sr;
            al[i] = 1.0 - exp(arg);
        }
    }
#pragma unroll 1
    for (int j = 0; j < 8; j++) {
        double hs = st[1 + j], ls = st[9 + j];
        onepole_reg_var<OP_DIFF_OLD>(v, al, c, hs, tmp);
        onepole_reg_var<OP_NEW>(v, al, c, ls, tmp);
        if (c.last) { ds[1 + j] = hs; ds[9 + j] = ls; }
    }
#pragma unroll
    for (int i = 0; i < CHK; i++) v[i] = clip1(256.0 * v[i]);
    chunk_store(out, c, v);
}

UNIT_FN unit_autowah(UNIT_ARGS) {
    if (N == CHK * SEG_T) { autowah_full(U, flip); return; }
    UNIT_PROLOGUE
    envelope_to(in, out, N, U->ip[0], U->dp[5], U->dp[6], &U->ds[0], tmp);
    const double la = U->dp[0], lb = U->dp[1], fa = U->dp[2], fb = U->dp[3], slope = U->dp[4], sr = U->dp[7];
    const int tid = seg_tid(), m = (N + SEG_T - 1) / SEG_T;
    const int c0 = min(N, tid * m), c1 = min(N, c0 + m);
    for (int i = c0; i < c1; i++) {
        double level = 20.0 * log10(out[LX(i)]);
        double frequency;
        if (level <= la) frequency = fa;
        else if (level >= lb) frequency = fb;
        else frequency = fa + (slope * (level - la));
        double arg = -frequency / sr;
        double alpha = 1.0 - exp(arg);
        out[LX(i)] = in[LX(i)];                      /* the signal */
        in[LX(i)] = alpha;                           /* the per-sample coefficient */
    }
    for (int j = 0; j < 8; j++) {
        onepole<OP_DIFF_OLD, true>(out, in, 0.0, &U->ds[1 + j], N, tmp);
        onepole<OP_NEW, true>(out, in, 0.0, &U->ds[9 + j], N, tmp);
    }
    for (int i = c0; i < c1; i++) out[LX(i)] = clip1(256.0 * out[LX(i)]);
}

/* ---- bandpass: effects/bandpass.go:20-98.  jp0 half order; dp0 high-pass coefficient, dp1 low-pass coefficient; ds0..3 hcv, ds4..7 lcv */
/* the batch block size: every stage is a high-pass feeding a low-pass with the SAME two coefficients -- one 2 x 2 scan per stage
 * (lin2_scan, one table for all stages), then the reference's loop body and the clip between stages */
__device__ __forceinline__ void bandpass_full(const gdg_seg_unit *Ug, int flip) {
    UNIT_PROLOGUE
    const GDG_CONST gdg_seg_unit *U = uniform_unit(Ug);
    double *st = tmp + SEG_STASH;                    /* [0..3] hcv, [4..7] lcv */
    GDG_GLOBAL double *ds = as_global(U->ds);
    const double aH = U->dp[0], aL = U->dp[1];
    const int half = U->jp[0];
    if (seg_tid() < 8) st[seg_tid()] = ds[seg_tid()];
    tab_fetch(scr, U->tab, L2_SIZE);
    const ChunkT<true> c = full_chunk();
    double v[CHK];
    chunk_load(in, c, v);
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < half; j++) {
        double h = 0.0, l = 0.0;
#pragma unroll
        for (int i = 0; i < CHK; i++) { h = fma(scr[L2_W + 2 * i], v[i], h); l = fma(scr[L2_W + 2 * i + 1], v[i], l); }
        lin2_scan(h, l, scr, &st[j], &st[4 + j], tmp + (j & 1) * 2 * LX_SLOT);
#pragma unroll
        for (int i = 0; i < CHK; i++) {                      /* bandpass.go:70-91 */
            double diff = v[i] - h;
            h += diff * aH;
            diff -= l;
            double iv = l;
            l += diff * aL;
            v[i] = clip1(iv);
        }
        if (c.last) { ds[j] = h; ds[4 + j] = l; }
    }
    chunk_store(out, c, v);
}

UNIT_FN unit_bandpass(UNIT_ARGS) {
    if (N == CHK * SEG_T) { bandpass_full(U, flip); return; }
    UNIT_PROLOGUE
    const int tid = seg_tid(), m = (N + SEG_T - 1) / SEG_T;
    const int c0 = min(N, tid * m), c1 = min(N, c0 + m);
    for (int i = c0; i < c1; i++) out[LX(i)] = in[LX(i)];
    const int half = U->jp[0];
    for (int j = 0; j < half; j++) {
        onepole<OP_DIFF_OLD, false>(out, nullptr, U->dp[0], &U->ds[j], N, tmp);
        onepole<OP_OLD, false>(out, nullptr, U->dp[1], &U->ds[4 + j], N, tmp);
        for (int i = c0; i < c1; i++) out[LX(i)] = clip1(out[LX(i)]);     /* the clip between stages (bandpass.go:85-91) */
    }
}

/* ---- octaver: effects/octaver.go:21-139 -----------------------------------------------------------------------------------------------
 * ip0 follow; dp0 up, dp1 clean, dp2 dist, dp3 down1, dp4 down2, dp5 hysteresis factors, dp6 exp(-20/sr), dp7 1 - dp6;
 * ds0 envelope, ds1 coupling cap; is0 previousPolarity (-1, 0, 1), is1 octaveRegister.
 * The polarity FSM is scanned as a map  pp_in -> (pp_out, register increment):  f = {has, s1, drest, pp_out}. */
/* the batch block size: follower and coupling capacitor as constant-coefficient scans on the register chunk, the polarity FSM as
 * before (a scan of small maps); nothing but the frame itself goes through LDS */
__device__ __forceinline__ void octaver_full(const gdg_seg_unit *Ug, int flip) {
    UNIT_PROLOGUE
    const GDG_CONST gdg_seg_unit *U = uniform_unit(Ug);
    double *st = tmp + SEG_STASH;                    /* [0] envelope, [1] coupling capacitor */
    GDG_GLOBAL double *ds = as_global(U->ds);
    GDG_GLOBAL int *is = as_global(U->is);
    const int tid = seg_tid(), follow = U->ip[0];
    const double f_up = U->dp[0], f_clean = U->dp[1], f_dist = U->dp[2], f_d1 = U->dp[3], f_d2 = U->dp[4], f_hyst = U->dp[5];
    const double d_inv = U->dp[6], d = U->dp[7];
    double *tab_env = scr, *tab_cap = scr + LT_SIZE;
    if (tid < 2) st[tid] = ds[tid];
    tab_fetch(tab_env, U->tab, 2 * LT_SIZE);                    /* [follower | coupling capacitor] */
    const int pp0 = is[0], reg0 = is[1];
    const ChunkT<true> c = full_chunk();
    double x[CHK], e[CHK];
    chunk_load(in, c, x);
    __syncthreads();
    if (follow == 0) {
        double s = lin_scan<true>(lin_chunk_map<true>(x, tab_env), tab_env, &st[0], tmp);
#pragma unroll
        for (int i = 0; i < CHK; i++) { s *= d_inv; double q = fabs(x[i]); if (q > s) s = q; e[i] = s; }
        if (c.last) ds[0] = s;
    } else if (follow == 1) {
        double s = lin_scan<false>(lin_chunk_map<false, true>(x, tab_env), tab_env, &st[0], tmp);
#pragma unroll
        for (int i = 0; i < CHK; i++) { double diff = fabs(x[i]) - s; s += diff * d; e[i] = s; }
        if (c.last) ds[0] = s;
    } else {
#pragma unroll
        for (int i = 0; i < CHK; i++) e[i] = 1.0;
        if (c.last) ds[0] = 1.0;
    }
    /* polarity FSM: chunk map pp_in -> (pp_out, register increment), f = {has, first sign, flips after it, last sign} */
    IMap mine = { { 0, 0, 0, 0, 0 } }, ident = { { 0, 0, 0, 0, 0 } };
#pragma unroll
    for (int i = 0; i < CHK; i++) {
        const double sample = x[i];
        const int sg = sample < 0.0 ? -1 : (sample > 0.0 ? 1 : 0);
        if (sg != 0 && fabs(sample) > e[i] * f_hyst) {
            if (!mine.f[0]) { mine.f[0] = 1; mine.f[1] = sg; mine.f[2] = 0; mine.f[3] = sg; }
            else if (sg != mine.f[3]) { mine.f[2]++; mine.f[3] = sg; }
        }
    }
    auto comp = [](const IMap &f, const IMap &g) -> IMap {
        if (!g.f[0]) return f;
        if (!f.f[0]) return g;
        IMap r = f;
        r.f[2] = f.f[2] + (f.f[3] != g.f[1] ? 1 : 0) + g.f[2];
        r.f[3] = g.f[3];
        return r;
    };
    IMap pre = block_scan_imap(mine, ident, comp, reinterpret_cast<int *>(tmp + 2 * LX_SLOT));
    int pp = pp0;
    unsigned reg = (unsigned)reg0;
    if (pre.f[0]) { reg = (reg + (unsigned)pre.f[2] + (pp0 != pre.f[1] ? 1u : 0u)) & 7u; pp = pre.f[3]; }
    double p[CHK];
#pragma unroll
    for (int i = 0; i < CHK; i++) {                  /* octaver.go:86-131 */
        const double sample = x[i], envelope = e[i];
        const double sample_abs = fabs(sample), square = sample * sample;
        const int sg = sample < 0.0 ? -1 : (sample > 0.0 ? 1 : 0);
        const double sign = (double)sg, hysteresis = envelope * f_hyst;
        if ((sg != 0) && (sg != pp) && (sample_abs > hysteresis)) { reg = (reg + 1u) & 7u; pp = sg; }
        const double first_down = (reg & 2u) ? -1.0 : 1.0, second_down = (reg & 4u) ? -1.0 : 1.0;
        double q = f_clean * sample;
        if (envelope > 0.0001) q += f_up * (square / envelope);
        q += f_dist * (sign * envelope);
        q += f_d1 * (first_down * envelope);
        q += f_d2 * (second_down * envelope);
        p[i] = q;
    }
    if (c.last) { is[0] = pp; is[1] = (int)reg; }
    {
        double s = lin_scan<false>(lin_chunk_map<false>(p, tab_cap), tab_cap, &st[1], tmp + 2 * LX_SLOT);
#pragma unroll
        for (int i = 0; i < CHK; i++) { double diff = p[i] - s; s += diff * d; p[i] = clip1(p[i] - s); }
        if (c.last) ds[1] = s;
    }
    chunk_store(out, c, p);
}

UNIT_FN unit_octaver(UNIT_ARGS) {
    if (N == CHK * SEG_T) { octaver_full(U, flip); return; }
    UNIT_PROLOGUE
    envelope_to(in, out, N, U->ip[0], U->dp[6], U->dp[7], &U->ds[0], tmp);
    const double f_up = U->dp[0], f_clean = U->dp[1], f_dist = U->dp[2], f_d1 = U->dp[3], f_d2 = U->dp[4], f_hyst = U->dp[5];
    const int tid = seg_tid(), m = (N + SEG_T - 1) / SEG_T;
    const int c0 = min(N, tid * m), c1 = min(N, c0 + m);
    const int pp0 = U->is[0], reg0 = U->is[1];
    IMap mine = { { 0, 0, 0, 0, 0 } }, ident = { { 0, 0, 0, 0, 0 } };
    for (int i = c0; i < c1; i++) {
        double sample = in[LX(i)];
        int s = sample < 0.0 ? -1 : (sample > 0.0 ? 1 : 0);
        double hysteresis = out[LX(i)] * f_hyst;
        if (s != 0 && fabs(sample) > hysteresis) {
            if (!mine.f[0]) { mine.f[0] = 1; mine.f[1] = s; mine.f[2] = 0; mine.f[3] = s; }
            else if (s != mine.f[3]) { mine.f[2]++; mine.f[3] = s; }
        }
    }
    auto comp = [](const IMap &f, const IMap &g) -> IMap {
        if (!g.f[0]) return f;
        if (!f.f[0]) return g;
        IMap r = f;
        r.f[2] = f.f[2] + (f.f[3] != g.f[1] ? 1 : 0) + g.f[2];
        r.f[3] = g.f[3];
        return r;
    };
    IMap pre = block_scan_imap(mine, ident, comp, reinterpret_cast<int *>(tmp));
    int pp = pp0;
    unsigned reg = (unsigned)reg0;
    if (pre.f[0]) { reg = (reg + (unsigned)pre.f[2] + (pp0 != pre.f[1] ? 1u : 0u)) & 7u; pp = pre.f[3]; }
    for (int i = c0; i < c1; i++) {
        double sample = in[LX(i)];
        double sample_abs = fabs(sample);
        double envelope = out[LX(i)];
        double square = sample * sample;
        int s = sample < 0.0 ? -1 : (sample > 0.0 ? 1 : 0);
        double sign = (double)s;
        double hysteresis = envelope * f_hyst;
        if ((s != 0) && (s != pp) && (sample_abs > hysteresis)) { reg = (reg + 1u) & 7u; pp = s; }
        double first_down = (reg & 2u) ? -1.0 : 1.0;
        double second_down = (reg & 4u) ? -1.0 : 1.0;
        double p = f_clean * sample;
        if (envelope > 0.0001) p += f_up * (square / envelope);
        p += f_dist * (sign * envelope);
        p += f_d1 * (first_down * envelope);
        p += f_d2 * (second_down * envelope);
        out[LX(i)] = p;
    }
    if (c1 == N && c0 < N) { U->is[0] = pp; U->is[1] = (int)reg; }
    onepole<OP_DIFF_NEW, false>(out, nullptr, U->dp[7], &U->ds[1], N, tmp);
    for (int i = c0; i < c1; i++) out[LX(i)] = clip1(out[LX(i)]);
}

/* ---- noise gate: effects/noisegate.go:19-96 ------------------------------------------------------------------------------------------------
 * dp0 open factor, dp1 close factor; jp0 hold samples, jp1 pass-through (threshold_open < threshold_close); is0 gateOpen, is1 onHoldSince.
 * FSM map (open_in, since_in) -> (open_out, since_out):  f = {has_reset, off, forced, fv, T}
 *   since_out = has_reset ? off : since_in + off;   open_out = forced ? fv : (open_in && since_in < T)
 * (every sample above the open threshold is also above the close threshold, so before the first reset only closing can happen). */
#define GATE_INF 0x3fffffff
UNIT_FN unit_noisegate(UNIT_ARGS) {
    UNIT_PROLOGUE
    const int tid = seg_tid(), m = (N + SEG_T - 1) / SEG_T;
    const int c0 = min(N, tid * m), c1 = min(N, c0 + m);
    if (U->jp[1]) {
        for (int i = c0; i < c1; i++) out[LX(i)] = in[LX(i)];
        __syncthreads();
        if (tid == 0) { U->is[0] = 1; U->is[1] = 0; }
        return;
    }
    const double fac_open = U->dp[0], fac_close = U->dp[1];
    const int hold = U->jp[0];
    const int open0 = U->is[0], since0 = U->is[1];
    IMap mine = { { 0, 0, 0, 0, GATE_INF } }, ident = { { 0, 0, 0, 0, GATE_INF } };
    {
        int has = 0, r1 = 0, since = 0, forced = 0, fv = 0;
        for (int i = c0; i < c1; i++) {
            double a = fabs(in[LX(i)]);
            if (!has) {
                if (a > fac_close) { has = 1; r1 = i - c0; } else continue;
            }
            if (a > fac_open) { forced = 1; fv = 1; }
            if (a > fac_close) since = 0;
            if (since >= hold) { forced = 1; fv = 0; }
            since++;
        }
        int len = c1 - c0;
        int prefix = has ? r1 : len;
        int T = GATE_INF;
        if (prefix > 0) { T = hold - prefix + 1; if (T < 0) T = 0; }
        mine.f[0] = has; mine.f[1] = has ? since : len; mine.f[2] = forced; mine.f[3] = fv; mine.f[4] = T;
    }
    auto comp = [](const IMap &f, const IMap &g) -> IMap {
        IMap r;
        r.f[0] = (g.f[0] || f.f[0]) ? 1 : 0;
        r.f[1] = g.f[0] ? g.f[1] : f.f[1] + g.f[1];
        r.f[2] = 0; r.f[3] = 0; r.f[4] = GATE_INF;
        if (g.f[2]) { r.f[2] = 1; r.f[3] = g.f[3]; }
        else if (f.f[0]) {
            int cond2 = f.f[1] < g.f[4];
            if (f.f[2]) { r.f[2] = 1; r.f[3] = (f.f[3] && cond2) ? 1 : 0; }
            else if (cond2) { r.f[4] = f.f[4]; }
            else { r.f[2] = 1; r.f[3] = 0; }
        } else {
            int t2 = g.f[4];
            if (t2 != GATE_INF) { t2 -= f.f[1]; if (t2 < 0) t2 = 0; }
            r.f[4] = f.f[4] < t2 ? f.f[4] : t2;
        }
        return r;
    };
    IMap pre = block_scan_imap(mine, ident, comp, reinterpret_cast<int *>(tmp));
    int since = pre.f[0] ? pre.f[1] : since0 + pre.f[1];
    int gate = pre.f[2] ? pre.f[3] : ((open0 && since0 < pre.f[4]) ? 1 : 0);
    for (int i = c0; i < c1; i++) {
        double sample = in[LX(i)];
        double a = fabs(sample);
        if (a > fac_open) gate = 1;
        if (a > fac_close) since = 0;
        if (since >= hold) gate = 0;
        double fac = gate ? 1.0 : 0.0;
        out[LX(i)] = fac * sample;
        since++;
    }
    if (c1 == N && c0 < N) { U->is[0] = gate; U->is[1] = since; }
}

/* ---- debug: the oversampler / decimator tiles on their own (gdg_debug_oversample_decimate) ------------------------------------
 * OversamplerDecimator.Oversample followed by Decimate on its output (oversampling/oversampling_test.go:84-128 does exactly that):
 * `hist` = the object's state ([8 inputs | TAPS - 1 oversampled samples], zeros = a fresh object), `up` F N samples, `down` N. */
template <int F>
__global__ void __launch_bounds__(SEG_T)
os_debug_kernel(const double *__restrict__ in, int N, double *hist, double *__restrict__ up, double *__restrict__ down, gdg_os_tables os) {
    const int tid = seg_tid();
    for (int i = tid; i < N; i += SEG_T) s_a[LX(i)] = in[i];
    __syncthreads();
    Shaper S = {};
    shaper_oversampled<F, true>(S, s_a, s_b, s_scr, hist, F == 2 ? os.tapsP2 : os.tapsP4, F == 2 ? os.lanczos2 : os.lanczos4, N, up);
    __syncthreads();
    for (int i = tid; i < N; i += SEG_T) down[i] = s_a[LX(i)];
}

hipError_t gdg_launch_os_debug(int factor, const double *d_in, int n, double *d_hist, double *d_up, double *d_down, gdg_os_tables os, hipStream_t s) {
    if (factor == 2) hipLaunchKernelGGL(os_debug_kernel<2>, dim3(1), dim3(SEG_T), 0, s, d_in, n, d_hist, d_up, d_down, os);
    else if (factor == 4) hipLaunchKernelGGL(os_debug_kernel<4>, dim3(1), dim3(SEG_T), 0, s, d_in, n, d_hist, d_up, d_down, os);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

/* ---- an oversampled shaper as a launch of its own: one workgroup per TILE (round 5) -----------------------------------------------------------
 * The oversampled overdrive / distortion / excess is the one unit of the chains that is heavy (46 us per frame on one CU at 4 x: 32768
 * exp + a 155-tap decimator) AND feed-forward: a tile of 2048 (4 x) / 4096 (2 x) outputs needs the inputs below it and nothing a neighbour
 * computes.  With the chip full that does not matter; with 64 channels (a GPU's share of a split job, BASELINE config 3) the unit runs on
 * 64 of 256 CUs for 1/3 of the step.  The plan then cuts the segment at the unit (api_plan.cpp) and this kernel runs one workgroup per
 * (channel, frame of the window, tile): the tile's inputs and the BACK + 6 samples below them come straight from the caller's row in HBM
 * (frames of a window are consecutive in it, so the previous frame's tail is simply below the frame), the oversampled, shaped stream is
 * made in LDS in the same polyphase layout by the same code (os_decimate, shape) as the in-segment unit, the outputs go to the output row.
 * What the in-segment unit keeps as STATE between frames -- the last 8 inputs and the last TAPS - 1 shaped oversampled samples -- is
 * recomputed here from the previous frame's inputs for every frame but a call's first (the same expressions on the same operands: the same
 * bits); the call's first frame reads the unit's state, its last frame writes it.  One flag per channel orders that pair: the last
 * workgroup does not store before the first has loaded (a bounded spin; both are in flight together in every launch that has both). */
/* pre_chans != NULL (a per-frame call, n_frames == 1; round 6): the segment step in FRONT of the shaper is a lone compressor in every channel
 * (BASELINE config 3: compressor > 4 x overdrive > ...), pre_chans[c] its descriptor.  A launch of its own for it was 8.6 us of which 6.4 are a
 * launch's fixed chain; here every tile's workgroup runs the compressor over the whole frame itself (2.6 us of vector work, the unit's own code:
 * compressor_full) and takes its tile's inputs from LDS.  All tiles of a channel read the compressor's state; the channel's LAST workgroup stores the
 * new one once an arrival counter (arrive[c], zero between launches) says that every tile has read the old one. */
template <int F>
__global__ void __launch_bounds__(SEG_T)
os_tiles_kernel(const gdg_seg_chan *__restrict__ chans, const gdg_seg_unit *__restrict__ units, int N, int n_frames, gdg_shift shift,
                gdg_os_tables os, int *__restrict__ flags, int epoch, int *d_error, const gdg_seg_chan *__restrict__ pre_chans, int *__restrict__ arrive) {
    constexpr int TAPS = OsCfg<F>::TAPS, BACK = OsCfg<F>::BACK, R = OsCfg<F>::R, TILE = OsCfg<F>::S, PH = OsCfg<F>::PH, HALO = BACK + 6;
    static_assert(TILE + HALO + 8 <= 8192 + 256, "a tile's inputs fit one frame buffer");
    const int tid = seg_tid();
    const int tiles = (N + TILE - 1) / TILE;
    const int f = blockIdx.x / tiles, tile = blockIdx.x - f * tiles;
    gdg_seg_chan ch = chans[blockIdx.y];
    if (ch.flags & GDG_SRC_IS_INPUT) ch.src += shift.in;
    if (ch.flags & GDG_DST_IS_OUTPUT) ch.dst += shift.out;
    const gdg_seg_unit *U = units + ch.unit_begin;
    Shaper S;
    S.type = U->type; S.valve = U->ip[4];
    S.gain = U->dp[0]; S.drive = U->dp[1]; S.clean = U->dp[2]; S.level = U->dp[3];
    const GDG_CONST double *taps = (const GDG_CONST double *)uniform_ptr(F == 2 ? os.tapsP2 : os.tapsP4);
    const GDG_CONST double *lw = (const GDG_CONST double *)uniform_ptr(F == 2 ? os.lanczos2 : os.lanczos4);
    GDG_GLOBAL double *hist = as_global(U->hist);
    const GDG_GLOBAL double *src = as_global(ch.src) + (size_t)f * N;      /* sample 0 of this frame; negative indices: the previous frame of the row */
    GDG_GLOBAL double *dst = as_global(ch.dst) + (size_t)f * N;
    double *sin_ = s_a, *stage = s_b, *scr = s_scr;
    const bool first = f == 0, last = f == n_frames - 1 && tile == tiles - 1;
    const int o0 = tile * TILE, S_out = min(TILE, N - o0), I0 = o0 - BACK, b0 = I0 - 6;       /* b0: the lowest input index the tile touches */
    double *pre_state = s_tmp + SEG_STASH + 30;              /* the compressor's new state, parked (cells the compressor itself does not use) */
    double pre_keep = 0.0;
    if (pre_chans) {
        /* the whole frame through the compressor in front of the shaper (n_frames == 1: the launcher sees to it), then this tile's inputs
         * out of its result */
        gdg_seg_chan pch = pre_chans[blockIdx.y];
        if (pch.flags & GDG_SRC_IS_INPUT) pch.src += shift.in;
        const GDG_GLOBAL seg_v2d *s2 = (const GDG_GLOBAL seg_v2d *)pch.src;
        seg_v2d fv[CHK / 2];
#pragma unroll
        for (int q = 0; q < CHK / 2; q++) fv[q] = s2[tid + q * SEG_T];
#pragma unroll
        for (int q = 0; q < CHK / 2; q++) { const int i = tid + q * SEG_T; s_a[LX(2 * i)] = fv[q].x; s_a[LX(2 * i + 1)] = fv[q].y; }
        __syncthreads();
        compressor_full(units + pch.unit_begin, 0, false, pre_state);      /* s_a -> s_b; its first barrier lies behind its read of the old state */
        __syncthreads();
        if (tid == 0) atomicAdd(arrive + blockIdx.y, 1);         /* this workgroup has read the compressor's state */
        constexpr int NK = (TILE + HALO + SEG_T - 1) / SEG_T;    /* a tile's inputs per thread: 3 (4 x) / 5 (2 x) */
        double keepv[NK];
#pragma unroll
        for (int q = 0; q < NK; q++) {
            const int k = tid + q * SEG_T, i = b0 + k;
            keepv[q] = (k < S_out + HALO) ? ((i >= 0) ? s_b[LX(i)] : ((i >= -8) ? hist[8 + i] : 0.0)) : 0.0;
        }
        if (last && tid < 8) pre_keep = s_b[LX(N - 8 + tid)];     /* the call's last 8 inputs of the shaper (its state for the next call) */
        __syncthreads();                                         /* everybody holds its inputs: the frame buffers are free */
#pragma unroll
        for (int q = 0; q < NK; q++) { const int k = tid + q * SEG_T; if (k < S_out + HALO) sin_[LX(k)] = keepv[q]; }
    } else {
    /* inputs b0 .. o0 + S_out - 1 -> sin_[0 ..]; below the call's first frame: the unit's 8-sample history, older ones are never used */
    for (int k = tid; k < S_out + HALO; k += SEG_T) {
        const int i = b0 + k;
        double v;
        if (i >= 0 || !first) v = src[i];
        else v = (i >= -8) ? hist[8 + i] : 0.0;
        sin_[LX(k)] = v;
    }
    }
    if (first && tile == 0) for (int q = tid; q < TAPS - 1; q += SEG_T) scr[q] = hist[8 + q];      /* the previous call's oversampled tail */
    double keep = 0.0;
    if (last && tid < 8) {                                  /* the last 8 inputs of the call (older frames of the row, or the old history, when N < 8 never happens: N = 8192) */
        keep = pre_chans ? pre_keep : src[N - 8 + tid];      /* (with the compressor in front: its output) */
    }
    __syncthreads();
    if (first && tile == 0 && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    /* the state is in LDS / registers: the call's last workgroup may replace it */
        __hip_atomic_store(as_global(flags + blockIdx.y), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    auto s_at = [&](int k) -> double { return sin_[LX(k - b0)]; };
    const int slots = S_out + BACK;
    for (int idx = tid; idx < min(slots + OsCfg<F>::NC, PH); idx += SEG_T) {
        const int i = I0 + idx;
        if (idx >= slots) {
#pragma unroll
            for (int r = 0; r < F; r++) stage[r * PH + idx] = 0.0;
        } else if (i < 0 && first) {
            /* oversampled samples of the previous CALL: the stored tail (older than it: never read) */
#pragma unroll
            for (int r = 0; r < F; r++) {
                const int m = F * i + r;
                stage[r * PH + idx] = (m >= -(TAPS - 1)) ? scr[(TAPS - 1) + m] : 0.0;
            }
        } else {
            /* (i < 0 in a later frame of the window: the previous frame's samples, made again from its inputs) */
            double w6[6];
#pragma unroll
            for (int t = 0; t < 6; t++) w6[t] = s_at(i - 6 + t);
            stage[idx] = shape(S, w6[2]);
#pragma unroll
            for (int r = 1; r < F; r++) {
                double up = 0.0;
#pragma unroll
                for (int t = 0; t < 6; t++) up += w6[t] * lw[(r - 1) * 6 + t];
                stage[r * PH + idx] = shape(S, up);
            }
        }
    }
    __syncthreads();
    if (last) {
        /* the call's new state, once the call's first workgroup has read the old one */
        if (tid == 0 && !(first && tile == 0)) {
            int spins = 0;
            unsigned long long t0 = 0;
            while (__hip_atomic_load(as_global(flags + blockIdx.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (wave_spin_expired(spins, t0, d_error)) { atomicExch(d_error, GDG_WAVE_TIMEOUT_CODE); break; }
            }
        }
        if (pre_chans && tid == 0) {
            /* ... and the compressor's: every tile of the channel has read the old state */
            const int tiles_all = tiles * n_frames;
            int spins = 0;
            unsigned long long t0 = 0;
            while (__hip_atomic_load(as_global(arrive + blockIdx.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tiles_all) {
                __builtin_amdgcn_s_sleep(1);
                if (wave_spin_expired(spins, t0, d_error)) { atomicExch(d_error, GDG_WAVE_TIMEOUT_CODE); break; }
            }
            __hip_atomic_store(as_global(arrive + blockIdx.y), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       /* ready for the next launch */
            as_global(units[pre_chans[blockIdx.y].unit_begin].ds)[0] = *pre_state;
        }
        __syncthreads();
        for (int q = tid; q < TAPS - 1; q += SEG_T) {
            const int m = F * N - (TAPS - 1) + q;
            const int i = m / F, r = m - i * F;             /* m >= 0: N = 8192 */
            hist[8 + q] = stage[r * PH + (i - I0)];
        }
        if (tid < 8) hist[tid] = keep;
    }
    if (R * tid < S_out) {
        double acc[R];
        os_decimate<F>(stage, taps, tid, acc);
#pragma unroll
        for (int j = 0; j < R; j += 2) {
            seg_v2d v = { ATTENUATION_HALF_DECIBEL * clip1(acc[j]), ATTENUATION_HALF_DECIBEL * clip1(acc[j + 1]) };
            *(GDG_GLOBAL seg_v2d *)(dst + o0 + R * tid + j) = v;
        }
    }
}

/* n_chans x (n_frames x tiles) workgroups; d_flags: one int per channel of the launch (any value but this launch's epoch) */
hipError_t gdg_launch_os_tiles(int factor, const gdg_seg_chan *d_chans, int n_chans, const gdg_seg_unit *d_units, int frames, int n_frames,
                               gdg_shift shift, gdg_os_tables os, int *d_flags, int epoch, int *d_error, hipStream_t s,
                               const gdg_seg_chan *d_pre_chans, int *d_arrive) {
    if (n_chans <= 0 || n_frames <= 0) return hipSuccess;
    if (frames != GDG_MAX_FRAMES) return hipErrorInvalidValue;
    if (n_frames != 1 || !d_arrive) d_pre_chans = nullptr;
    if (factor == 2)
        hipLaunchKernelGGL(os_tiles_kernel<2>, dim3(n_frames * (frames / OsCfg<2>::S), n_chans), dim3(SEG_T), 0, s, d_chans, d_units, frames, n_frames, shift, os, d_flags, epoch, d_error,
                           d_pre_chans, d_arrive);
    else if (factor == 4)
        hipLaunchKernelGGL(os_tiles_kernel<4>, dim3(n_frames * (frames / OsCfg<4>::S), n_chans), dim3(SEG_T), 0, s, d_chans, d_units, frames, n_frames, shift, os, d_flags, epoch, d_error,
                           d_pre_chans, d_arrive);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
#endif

/* ---- the segment kernel ------------------------------------------------------------------------------------------ */
/* one frame of one channel: HBM -> LDS, the segment's units, LDS -> HBM */
template <bool WAVE>
__device__ __forceinline__ void seg_frame(const double *src, double *dst, const gdg_seg_unit *units, int unit_begin, int unit_count, int N,
                                          const gdg_os_tables &os, int *d_error, int my_type, int *wave = nullptr, int wf = 0, int wf_next = 0,
                                          unsigned wave_mask = 0, int epoch = 0, int ahead = 0, int stall = 0) {
    int tid = seg_tid();
#ifdef SEG_FAST
    /* opaque per call: in the window walk the compiler otherwise hoists every per-thread address of the frame's load and store loops out of
     * the frame loop and keeps them in callee-saved vector registers across the unit calls (v120-v123: over the 120 the units need) */
    asm volatile("" : "+v"(tid));
#endif
    int *s_types = reinterpret_cast<int *>(s_tmp + SEG_STASH + 32);       /* 16 ints behind the stash cells */
    const bool aligned = ((N & 1) | (int)((size_t)src & 15) | (int)((size_t)dst & 15)) == 0;
    if (aligned && N == CHK * SEG_T) {
        /* the batch block size: every load of the thread is in flight before the first one is consumed (a loop waits for
         * each load in turn: four exposed HBM latencies per workgroup) */
        const GDG_GLOBAL seg_v2d *s2 = (const GDG_GLOBAL seg_v2d *)src;
        seg_v2d v[CHK / 2];
#pragma unroll
        for (int q = 0; q < CHK / 2; q++) v[q] = s2[tid + q * SEG_T];
#pragma unroll
        for (int q = 0; q < CHK / 2; q++) { const int i = tid + q * SEG_T; s_a[LX(2 * i)] = v[q].x; s_a[LX(2 * i + 1)] = v[q].y; }
    } else if (aligned) {
        /* 16 bytes per lane: half the load instructions, 1 KiB per wave access */
        const GDG_GLOBAL seg_v2d *s2 = (const GDG_GLOBAL seg_v2d *)src;
        for (int i = tid; i < N / 2; i += SEG_T) { seg_v2d v = s2[i]; s_a[LX(2 * i)] = v.x; s_a[LX(2 * i + 1)] = v.y; }
    } else {
        for (int i = tid; i < N; i += SEG_T) s_a[LX(i)] = as_global(src)[i];
    }
    if (tid < unit_count && tid < 16) s_types[tid] = my_type;
    __syncthreads();
    int flip = 0;                                   /* 0: s_a holds the current frame, 1: s_b */
#ifdef SEG_TILE
    int xid = 0;                                    /* the next unit's first exchange id (tile_put / tile_get) */
#endif
    for (int u = 0; u < unit_count; u++) {
        double *out = flip ? s_a : s_b;
        const gdg_seg_unit *U = units + unit_begin + u;
        const int type = (u < 16) ? __builtin_amdgcn_readfirstlane(s_types[u]) : U->type;
        int inplace = 0;
#ifdef SEG_TILE
        /* (the barrier that ended the previous unit lies between its last read of s_tc.xid and this write; every unit has a barrier of its
         * own between here and its first exchange) */
        if (tid == 0) s_tc.xid = xid;
        xid += type == GDG_UNIT_COMPRESSOR ? 1 : type == GDG_UNIT_TONESTACK ? 4 : type == GDG_UNIT_CABINET ? 7 : (type == GDG_UNIT_CHORUS || type == GDG_UNIT_REVERB) ? 1 : 0;
#endif
        /* a unit that keeps nothing from frame to frame (a shaper without oversampling) meets nobody */
        const bool gated = WAVE && (u >= 15 || ((wave_mask >> u) & 1u));
        if (gated) wave_wait(wave + GDG_WAVE_CELLS * u, wf, u < 15 && ((wave_mask >> (16 + u)) & 1u), d_error);
        int second = 0;                              /* 1: the unit posted its first counter itself and waits on the second (general reverb) */
        bool skip_post = false;                      /* the unit posted its counter itself and has no second one (chorus) */
        switch (type) {
        case GDG_UNIT_COMPRESSOR: unit_compressor(U, flip, N, WAVE); break;
        case GDG_UNIT_OVERDRIVE:
        case GDG_UNIT_DISTORTION:
#ifdef SEG_SUBSET                                   /* no oversampling here: the six table pointers need not live across the unit calls */
        case GDG_UNIT_EXCESS: { gdg_os_tables none = {}; inplace = unit_shaper(U, flip, N, WAVE, none); break; }
#else
        case GDG_UNIT_EXCESS: inplace = unit_shaper(U, flip, N, WAVE, os); break;
#endif
        case GDG_UNIT_TONESTACK: unit_tonestack(U, flip, N, WAVE); break;
        case GDG_UNIT_CABINET: unit_cabinet(U, flip, N, WAVE); break;
        case GDG_UNIT_CHORUS: {
            const WaveGate gate = { wave + GDG_WAVE_CELLS * u + 1, wf, wf_next, (wave_mask >> 31) != 0, epoch, d_error };
            int posted = 0;
            unit_chorus(U, flip, N, WAVE, gate, &posted);
            skip_post = WAVE && __builtin_amdgcn_readfirstlane(posted);
            break;
        }
#ifdef SEG_TILE
        case GDG_UNIT_REVERB: unit_reverb_mix_tile(U, flip, N, false); break;      /* (only reverbs whose wet path an earlier launch made come here: api_plan.cpp) */
#endif
#ifndef SEG_TILE                                    /* (a tile sees a part of the frame: these three and the whole reverb keep to whole frames) */
        case GDG_UNIT_RINGMODULATOR: unit_ringmod(U, flip, N, WAVE); break;
        case GDG_UNIT_TREMOLO: unit_tremolo(U, flip, N, WAVE); break;
        case GDG_UNIT_SIGNALGENERATOR: unit_siggen(U, flip, N, WAVE); break;
        case GDG_UNIT_REVERB: {
            const WaveGate gate = { wave + GDG_WAVE_CELLS * u + 1, wf, wf_next, (wave_mask >> 31) != 0, epoch, d_error };
            unit_reverb(U, flip, N, WAVE, gate, ahead);
            second = WAVE ? 1 : 0;
            break;
        }
#endif
#ifndef SEG_SUBSET                                  /* units that read other threads' cells of their input after writing output, or stage a frame: two buffers */
        case GDG_UNIT_FLANGER:
        case GDG_UNIT_PHASER: unit_flanger(U, flip, N, WAVE); break;
        case GDG_UNIT_DELAY: unit_delay(U, flip, N, WAVE); break;
        case GDG_UNIT_FUZZ:
            if (U->jp[0] > 1) inplace = unit_fuzz_os(U, flip, N, WAVE, os);
            else unit_fuzz(U, flip, N, WAVE);
            break;
        case GDG_UNIT_AUTOYOY: unit_autoyoy(U, flip, N, WAVE); break;
        case GDG_UNIT_AUTOWAH: unit_autowah(U, flip, N, WAVE); break;
        case GDG_UNIT_BANDPASS: unit_bandpass(U, flip, N, WAVE); break;
        case GDG_UNIT_OCTAVER: unit_octaver(U, flip, N, WAVE); break;
        case GDG_UNIT_NOISEGATE: unit_noisegate(U, flip, N, WAVE); break;
#endif
        default:
            if (tid == 0) atomicExch(d_error, 1 + type);
            for (int i = tid; i < N; i += SEG_T) out[LX(i)] = 0.0;
            break;
        }
        /* (stall: option debug_stall_unit -- frame 0 withholds this unit's counter, so that frame 1's bounded wait can be seen to expire) */
        if (gated && !skip_post && !(WAVE && stall == unit_begin + u + 1 && wf == 0)) wave_post(wave + GDG_WAVE_CELLS * u + second, wf_next, (wave_mask >> 31) != 0);
        else __syncthreads();
        if (!__builtin_amdgcn_readfirstlane(inplace)) flip ^= 1;
    }
    const double *fin = flip ? s_b : s_a;
    if (aligned) {
        GDG_GLOBAL seg_v2d *d2 = (GDG_GLOBAL seg_v2d *)dst;
        for (int i = tid; i < N / 2; i += SEG_T) { seg_v2d v = { fin[LX(2 * i)], fin[LX(2 * i + 1)] }; d2[i] = v; }
    } else {
        for (int i = tid; i < N; i += SEG_T) as_global(dst)[i] = fin[LX(i)];
    }
}
/* MULTI = false: one frame per launch (the loop below disappears: this is the kernel of the per-frame calls, and a loop around its
 * body costs it 8 us in scalar register spills).  MULTI = true: a window of n_frames frames per launch. */
#ifdef SEG_FAST
/* A kernel's register count is the largest of its own body and of every unit it can call, and it applies to ALL its waves.  The units of
 * this configuration stay at 120; a build in which one unit (or the window loop's scalars parked in vector lanes) took 124-128 ran EVERY
 * segment 15 % slower -- also segments that never call that unit (profiles/experiments/README.md, r04).  Hence the explicit ceiling. */
#define SEG_KERNEL_ATTR __attribute__((amdgpu_num_vgpr(60)))      /* counted in pairs on the unified register file of gfx90a and later: 120 */
#else
#define SEG_KERNEL_ATTR
#endif
#ifdef SEG_TILE
/* ---- per-frame calls of few channels: a channel's frame on SEG_TILES workgroups (see the head of this file) ---------------------------------
 * n_chans x SEG_TILES workgroups take (tile, channel) by TICKET, tile-major: a workgroup only ever waits for LOWER tiles of its channel, and
 * those hold smaller tickets -- they are running or done whatever the order of dispatch.  Beyond them: one workgroup per listed reverb of a
 * later segment step (REVERB_AHEAD, as in the general kernel).  xch: GDG_TILE_XCH_WORDS 8-byte words per descriptor of the launch. */
#define GDG_TILE_XCH_WORDS (GDG_TILE_XIDS * 16 * 2 * 2)
__global__ void __launch_bounds__(SEG_T, SEG_MIN_WAVES_PER_EU)
segt_kernel(const gdg_seg_chan *__restrict__ chans, const gdg_seg_unit *__restrict__ units, gdg_shift shift, gdg_os_tables os, int *d_error,
            int n_chans, int *ticket, int epoch, unsigned long long *xch, const int *__restrict__ ahead_list) {
    const int tid = seg_tid();
    if ((int)blockIdx.x >= SEG_TILES * n_chans) {
        const WaveGate none = {};
        reverb_general<REVERB_AHEAD>(units + ahead_list[blockIdx.x - SEG_TILES * n_chans], 0, GDG_MAX_FRAMES, false, none);
        return;
    }
    int *s_ticket = reinterpret_cast<int *>(s_tmp + SEG_STASH + 32) + 16;
    if (tid == 0) {
        const int t = atomicAdd(ticket, 1);
        if (t == SEG_TILES * n_chans - 1) __hip_atomic_store(as_global(ticket), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     /* everybody has drawn: ready for the next launch */
        *s_ticket = t;
    }
    __syncthreads();
    const int t = __builtin_amdgcn_readfirstlane(*s_ticket);
    const int tile = t / n_chans, block = t - tile * n_chans;
    gdg_seg_chan ch = chans[block];
    if (ch.flags & GDG_SRC_IS_INPUT) ch.src += shift.in;
    if (ch.flags & GDG_DST_IS_OUTPUT) ch.dst += shift.out;
    if (tid == 0) { s_tc.xch = xch + (size_t)block * GDG_TILE_XCH_WORDS; s_tc.d_error = d_error; s_tc.tile = tile; s_tc.epoch = epoch; s_tc.xid = 0; }
    int my_type = 0;
    if (tid < ch.unit_count && tid < 16) my_type = *(const GDG_GLOBAL int *)&units[ch.unit_begin + tid].type;
    /* (seg_frame's barrier behind the frame load publishes s_tc to the workgroup) */
    seg_frame<false>(ch.src + (size_t)tile * SEG_N, ch.dst + (size_t)tile * SEG_N, units, ch.unit_begin, ch.unit_count, SEG_N, os, d_error, my_type);
}

int gdg_segt_supported(int unit_type) {
    switch (unit_type) {
    case GDG_UNIT_COMPRESSOR: case GDG_UNIT_OVERDRIVE: case GDG_UNIT_DISTORTION: case GDG_UNIT_EXCESS: case GDG_UNIT_TONESTACK:
    case GDG_UNIT_CABINET: case GDG_UNIT_CHORUS:
        return 1;
    case GDG_UNIT_REVERB:
        return 2;                           /* only as the mix of a wet path an earlier launch of the call makes (the plan checks) */
    default:
        return 0;
    }
}
/* exchange ids a unit of that type uses (the plan keeps a segment's sum within GDG_TILE_XIDS) */
int gdg_segt_exchanges(int unit_type) {
    return unit_type == GDG_UNIT_COMPRESSOR ? 1 : unit_type == GDG_UNIT_TONESTACK ? 4 : unit_type == GDG_UNIT_CABINET ? 7 :
           (unit_type == GDG_UNIT_CHORUS || unit_type == GDG_UNIT_REVERB) ? 1 : 0;
}
size_t gdg_segt_xch_words(void) { return GDG_TILE_XCH_WORDS; }

hipError_t gdg_launch_segt(const gdg_seg_chan *d_chans, int n_chans, const gdg_seg_unit *d_units, gdg_shift shift, gdg_os_tables os, int *d_error,
                           hipStream_t s, int *d_ticket, int epoch, unsigned long long *d_xch, const int *d_ahead_list, int n_ahead) {
    if (n_chans <= 0) return hipSuccess;
    if (!d_ahead_list) n_ahead = 0;
    hipLaunchKernelGGL(segt_kernel, dim3(SEG_TILES * n_chans + n_ahead), dim3(SEG_T), 0, s, d_chans, d_units, shift, os, d_error, n_chans, d_ticket, epoch, d_xch, d_ahead_list);
    return hipGetLastError();
}
#else
template <int MODE>             /* 0: one frame per launch; 1: the walk; 2: WAVE (a workgroup per frame, above) */
__global__ void __launch_bounds__(SEG_T, SEG_MIN_WAVES_PER_EU) SEG_KERNEL_ATTR
seg_kernel(const gdg_seg_chan *__restrict__ chans, const gdg_seg_unit *__restrict__ units, int N, int n_frames, gdg_shift shift,
           gdg_os_tables os, int *d_error, int n_chans, int *ticket, int epoch, int ahead, const int *__restrict__ ahead_list) {
    constexpr bool MULTI = MODE == 1;
    int block = blockIdx.x, wf0 = 0;
#ifndef SEG_FAST
    if (MODE == 0 && block >= n_chans) {
        /* an extra workgroup: the wet path of a reverb of a LATER segment step of this call (reverb_general, REVERB_AHEAD) */
        const WaveGate none = {};
        reverb_general<REVERB_AHEAD>(units + ahead_list[block - n_chans], 0, GDG_MAX_FRAMES, false, none);
        return;
    }
#endif
    if (MODE == 2) {
        /* frames are dealt in the order the workgroups START, frame-major (all channels' frame 0, then frame 1, ...): whoever a
         * workgroup waits for holds a smaller ticket and is already running (or done), whatever the order of dispatch */
        int *s_ticket = reinterpret_cast<int *>(s_tmp + SEG_STASH + 32) + 16;
        if (threadIdx.x == 0) {
            const int t = atomicAdd(ticket, 1);
            if (t == n_chans * n_frames - 1) __hip_atomic_store(as_global(ticket), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   /* everybody has drawn: ready for the next launch */
            *s_ticket = t;
        }
        __syncthreads();
        const int t = __builtin_amdgcn_readfirstlane(*s_ticket);
        wf0 = t / n_chans;
        block = t - wf0 * n_chans;
    }
    gdg_seg_chan ch = chans[block];
    if (ch.flags & GDG_SRC_IS_INPUT) ch.src += shift.in;
    if (ch.flags & GDG_DST_IS_OUTPUT) ch.dst += shift.out;
    const int tid = seg_tid();
    /* the unit types of this segment, fetched together with the frame (one exposed latency instead of one per unit) */
    int my_type = 0;
    if (tid < ch.unit_count && tid < 16) my_type = *(const GDG_GLOBAL int *)&units[ch.unit_begin + tid].type;
    if (MODE == 2) {
        seg_frame<true>(ch.src + (size_t)wf0 * N, ch.dst + (size_t)wf0 * N, units, ch.unit_begin, ch.unit_count, N, os, d_error, my_type,
                        ch.wave, wf0, wf0 + 1 < n_frames ? wf0 + 1 : 0, (unsigned)ch.wave_mask, epoch, 0, ahead /* WAVE: the debug stall (launcher) */);
        return;
    }
    /* a window of n_frames consecutive frames (gdg_process_window_device): the workgroup walks them in order, the units' state going
     * through global memory from one frame to the next exactly as from one launch to the next (same CU, same L1; a barrier between) */
    for (int wf = 0; wf < (MULTI ? n_frames : 1); wf++, ch.src += N, ch.dst += N) {
        seg_frame<false>(ch.src, ch.dst, units, ch.unit_begin, ch.unit_count, N, os, d_error, my_type, nullptr, 0, 0, 0, 0, MODE == 0 ? ahead : 0);
        if (MULTI && wf + 1 < n_frames) { __threadfence_block(); __syncthreads(); }      /* state and LDS frames before the next frame touches them */
    }
}

int gdg_seg_supported(int unit_type) {
#ifdef SEG_FAST
    /* by type; the host adds the per-unit conditions (no oversampling, the reverb's shape: api_plan.cpp segf_unit_ok) */
    switch (unit_type) {
    case GDG_UNIT_COMPRESSOR: case GDG_UNIT_OVERDRIVE: case GDG_UNIT_DISTORTION: case GDG_UNIT_EXCESS: case GDG_UNIT_TONESTACK:
    case GDG_UNIT_CABINET: case GDG_UNIT_CHORUS: case GDG_UNIT_RINGMODULATOR: case GDG_UNIT_TREMOLO: case GDG_UNIT_SIGNALGENERATOR:
    case GDG_UNIT_REVERB:
        return 1;
    default:
        return 0;
    }
#endif
    switch (unit_type) {
    case GDG_UNIT_COMPRESSOR: case GDG_UNIT_OVERDRIVE: case GDG_UNIT_DISTORTION: case GDG_UNIT_EXCESS:
    case GDG_UNIT_TONESTACK: case GDG_UNIT_CABINET: case GDG_UNIT_CHORUS: case GDG_UNIT_FLANGER:
    case GDG_UNIT_PHASER: case GDG_UNIT_DELAY: case GDG_UNIT_RINGMODULATOR: case GDG_UNIT_TREMOLO:
    case GDG_UNIT_SIGNALGENERATOR: case GDG_UNIT_REVERB: case GDG_UNIT_FUZZ: case GDG_UNIT_AUTOYOY:
    case GDG_UNIT_AUTOWAH: case GDG_UNIT_BANDPASS: case GDG_UNIT_OCTAVER: case GDG_UNIT_NOISEGATE:
        return 1;
    default:
        return 0;
    }
}

hipError_t gdg_launch_seg(const gdg_seg_chan *d_chans, int n_chans, const gdg_seg_unit *d_units, int frames, int n_frames, gdg_shift shift,
                          gdg_os_tables os, int *d_error, hipStream_t s, int *d_wave_ticket, int epoch, int ahead, const int *d_ahead_list, int n_ahead) {
    if (n_chans <= 0 || n_frames <= 0) return hipSuccess;
    if (n_frames > 1 && d_wave_ticket)
        hipLaunchKernelGGL(seg_kernel<2>, dim3(n_chans * n_frames), dim3(SEG_T), 0, s, d_chans, d_units, frames, n_frames, shift, os, d_error, n_chans, d_wave_ticket, epoch,
                           ahead /* a WAVE launch has no use for it: 1 + the plan index of the unit whose first counter stays away (option debug_stall_unit) */, nullptr);
    else if (n_frames > 1) hipLaunchKernelGGL(seg_kernel<1>, dim3(n_chans), dim3(SEG_T), 0, s, d_chans, d_units, frames, n_frames, shift, os, d_error, n_chans, nullptr, 0, 0, nullptr);
    else {
#ifdef SEG_FAST
        n_ahead = 0;
#endif
        if (!d_ahead_list || frames != GDG_MAX_FRAMES) n_ahead = 0;
        hipLaunchKernelGGL(seg_kernel<0>, dim3(n_chans + n_ahead), dim3(SEG_T), 0, s, d_chans, d_units, frames, 1, shift, os, d_error, n_chans, nullptr, 0, ahead, d_ahead_list);
    }
    return hipGetLastError();
}
#endif      /* SEG_TILE */
