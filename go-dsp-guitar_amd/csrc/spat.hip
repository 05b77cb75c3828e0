/*
 * spat.hip -- spatializer.Process (spatializer/spatializer.go:140-335): the partial N -> 2 stereo
 * mixdown of one shard of channels.  Memory bound (8 B per channel-sample in, 16 B per sample out).
 *
 * Two levels so that the launch fills the chip: groups of 16 channels are summed in channel order, the
 * group partials in group order.  (The reference adds all channels in index order; the different
 * association changes the result by ~1e-16 * N, far inside the 1e-9 RMS bar.  Across shards the host adds
 * the partial pairs and the aux buffer, spatializer.go:300-310.)
 * Per-channel gains, delays and interpolation weights are computed on the host in the reference's
 * arithmetic (api_tuner_spat.cpp), including its quirk that the delay is always computed for 96 kHz.
 */
#include "gdg_internal.h"

#define SPAT_GROUP 16                    /* channels summed in index order by one lane */
#define SPAT_TILE 32                     /* samples per workgroup */
#define SPAT_SUB 16                      /* channel groups a workgroup works on at the same time (512 threads = 32 samples x 16) */
#define SPAT_T (SPAT_TILE * SPAT_SUB)
#define SPAT_UNROLL 16                   /* channels whose loads are in flight together */
#define SPAT_MAX_GROUPS 64               /* group partials held in LDS at a time: 64 x 2 x 32 doubles = 32 KiB = 1024 channels per round;
                                          * wider shards take several rounds (any channel count runs) */
#define SPAT_LDS_DESC_MAX 512             /* up to this many channels the descriptors are staged in LDS (24 KiB), beyond it read from HBM */

/* ONE launch per block (round 2: partial sums, reduce and history update were three launches, 23 us per 8192-frame block of 256
 * channels, all latency).  Workgroup t < tiles mixes samples [32 t, 32 t + 32): lane (s, q) sums channel groups q, q + 16, ... of 16
 * channels each in channel order, the group partials meet in LDS and are added in group order (an association of its own: groups of
 * 16 since late in round 3, of 32 before -- results differ from those versions in the last bits, ~1e-16 N, as they do from the reference's
 * plain channel order).  Shards of more than SPAT_MAX_GROUPS groups go through the LDS partials in rounds; the running sums carry over,
 * so the order of additions -- and the bits -- do not depend on the round size.  The history (last H inputs of every channel, spatializer.go:313-331) is double buffered: this block reads
 * `hist_read` and the workgroups t >= tiles write `hist_write`, so nobody waits for anybody inside the launch. */
template <bool LDS_DESC>
__global__ void __launch_bounds__(SPAT_TILE * SPAT_SUB)
spat_kernel(const gdg_spat_chan *__restrict__ chans, int nch, const double *__restrict__ in, int in_stride,
            const double *__restrict__ hist_read, double *__restrict__ hist_write, int H, double *__restrict__ out_lr, int out_stride,
            int frames, int tiles) {
    extern __shared__ double part[];                 /* [groups][2][SPAT_TILE] doubles, then nch channel descriptors */
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= tiles) {
        /* history: entry k of channel c = element k + frames - H of [old history | frame] shifted by H */
        const int total = nch * H;
        for (int e = ((int)blockIdx.x - tiles) * SPAT_T + tid; e < total; e += ((int)gridDim.x - tiles) * SPAT_T) {
            const int c = e / H, k = e - c * H, src = k + frames - H;
            hist_write[e] = (src >= 0) ? in[(size_t)c * in_stride + src] : hist_read[(size_t)c * H + k + frames];
        }
        return;
    }
    /* Workgroups are dealt to the eight XCDs round robin, and a tile reads three places of every channel row: its own 32 samples and the two
     * neighbours of the delayed sample, up to H samples back -- other tiles' samples.  With tile = blockIdx a cache line's three readers sat on
     * three XCDs, each fetching its own copy into its own L2; dealt so that an XCD works on a contiguous run of tiles (tiles / 8 of them = 1024
     * samples at the batch block size) they share one: 11.5-12.1 -> 10.6-10.9 us per launch (same box, profiles/spat_shapes_r05.txt; wider tiles
     * and more threads all lose: 64 x 8 15 us, 64 x 16 17, 128 x 4 23, 16 x 32 17). */
    const int bx = (int)blockIdx.x;
    const int tile = (tiles % 8 == 0) ? (bx & 7) * (tiles >> 3) + (bx >> 3) : bx;
    const int s = tid & (SPAT_TILE - 1), q = tid / SPAT_TILE;
    const int j = min(tile * SPAT_TILE + s, frames - 1);                 /* lanes past the end repeat the last sample and are not stored */
    const int groups = (nch + SPAT_GROUP - 1) / SPAT_GROUP;
    /* the channel descriptors, once per workgroup, into LDS (behind the partials); very wide shards read them from HBM instead */
    gdg_spat_chan *l_ch = reinterpret_cast<gdg_spat_chan *>(part + (size_t)groups * 2 * SPAT_TILE);
    if constexpr (LDS_DESC) {
        for (int c = tid; c < nch; c += SPAT_T) l_ch[c] = chans[c];
        __syncthreads();
    }
#define s_ch (LDS_DESC ? (const gdg_spat_chan *)l_ch : chans)
    const int side = tid / SPAT_TILE, jj = tile * SPAT_TILE + s;            /* the 2 x 32 finishing threads: (side, sample) */
    double acc = 0.0;
    for (int g0 = 0; g0 < groups; g0 += SPAT_MAX_GROUPS) {
    const int g1 = min(groups, g0 + SPAT_MAX_GROUPS);
    for (int g = g0 + q; g < g1; g += SPAT_SUB) {
        const int c_begin = g * SPAT_GROUP, c_end = min(nch, c_begin + SPAT_GROUP);
        double L = 0.0, R = 0.0;
        /* SPAT_UNROLL channels at a time: all their loads (current sample, the two neighbours of the delayed one -- from the block or from
         * the history, one address either way) are issued before the first is consumed; the sums keep the channel order */
        for (int c0 = c_begin; c0 < c_end; c0 += SPAT_UNROLL) {
            double cur[SPAT_UNROLL], se[SPAT_UNROLL], sl[SPAT_UNROLL];
#pragma unroll
            for (int u = 0; u < SPAT_UNROLL; u++) {
                const int c = min(c0 + u, c_end - 1);
                const double *x = in + (size_t)c * in_stride;
                const double *hb = hist_read + (size_t)c * H;
                const int delayed = s_ch[c].mode != 0;
                const int ie = delayed ? j - s_ch[c].early : j, il = delayed ? j - s_ch[c].late : j;
                cur[u] = x[j];
                se[u] = *((ie >= 0) ? x + ie : hb + (H + ie));
                sl[u] = *((il >= 0) ? x + il : hb + (H + il));
            }
#pragma unroll
            for (int u = 0; u < SPAT_UNROLL; u++) {
                if (c0 + u < c_end) {
                    const gdg_spat_chan ch = s_ch[c0 + u];
                    if (ch.mode == 0) {
                        L += ch.fac_left * cur[u];
                        R += ch.fac_right * cur[u];
                    } else {
                        double early_sample = ch.w_early * se[u];
                        double late_sample = ch.w_late * sl[u];
                        double delayed = early_sample + late_sample;
                        if (ch.mode == 1) { L += ch.fac_left * delayed; R += ch.fac_right * cur[u]; }
                        else { L += ch.fac_left * cur[u]; R += ch.fac_right * delayed; }
                    }
                }
            }
        }
        part[((g - g0) * 2 + 0) * SPAT_TILE + s] = L;
        part[((g - g0) * 2 + 1) * SPAT_TILE + s] = R;
    }
    __syncthreads();
    if (tid < 2 * SPAT_TILE)
        for (int g = g0; g < g1; g++) acc += part[((g - g0) * 2 + side) * SPAT_TILE + s];
    if (g1 < groups) __syncthreads();                    /* the next round overwrites the partials */
    }
    if (tid < 2 * SPAT_TILE && jj < frames) out_lr[(size_t)side * out_stride + jj] = acc;
}

#undef s_ch

hipError_t gdg_launch_spatializer(const gdg_spat_chan *d_chans, int nch, const double *d_in, int in_stride, const double *d_hist_read,
                                  double *d_hist_write, int H, double *d_out_lr, int out_stride, int frames, hipStream_t s) {
    if (H > 1024 || nch < 1) return hipErrorInvalidValue;
    const int tiles = (frames + SPAT_TILE - 1) / SPAT_TILE;
    int hist_blocks = (nch * H + 2047) / 2048;                    /* eight entries per thread */
    if (hist_blocks < 1) hist_blocks = 1;
    const int groups = (nch + SPAT_GROUP - 1) / SPAT_GROUP;
    const size_t part_bytes = (size_t)(groups < SPAT_MAX_GROUPS ? groups : SPAT_MAX_GROUPS) * 2 * SPAT_TILE * sizeof(double);
    if (nch <= SPAT_LDS_DESC_MAX)
        spat_kernel<true><<<dim3(tiles + hist_blocks), dim3(SPAT_T), part_bytes + (size_t)nch * sizeof(gdg_spat_chan), s>>>(d_chans, nch, d_in, in_stride, d_hist_read, d_hist_write, H, d_out_lr, out_stride, frames, tiles);
    else
        spat_kernel<false><<<dim3(tiles + hist_blocks), dim3(SPAT_T), part_bytes, s>>>(d_chans, nch, d_in, in_stride, d_hist_read, d_hist_write, H, d_out_lr, out_stride, frames, tiles);
    return hipGetLastError();
}
