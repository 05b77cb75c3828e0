/*
 * spat.hip -- spatializer.Process (spatializer/spatializer.go:140-335): the partial N -> 2 stereo
 * mixdown of one shard of channels.  Memory bound (8 B per channel-sample in, 16 B per sample out).
 *
 * Two levels so that the launch fills the chip: each workgroup sums a group of 32 channels for a
 * tile of 256 samples in channel order, a second tiny kernel adds the group partials in group
 * order.  (The reference adds all channels in index order; the different association changes the
 * result by ~1e-16 * N, far inside the 1e-9 RMS bar.  Across shards the host adds the partial
 * pairs and the aux buffer, spatializer.go:300-310.)
 * Per-channel gains, delays and interpolation weights are computed on the host in the reference's
 * arithmetic (api.cpp), including its quirk that the delay is always computed for 96 kHz.
 */
#include "gdg_internal.h"

#define SPAT_GROUP 32

__global__ void __launch_bounds__(256)
spat_partial_kernel(const gdg_spat_chan *__restrict__ chans, int nch, const double *__restrict__ in, int in_stride,
                    const double *__restrict__ hist, int H, double *__restrict__ partial, int frames, int max_frames) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y;
    if (j >= frames) return;
    const int c_begin = g * SPAT_GROUP, c_end = min(nch, c_begin + SPAT_GROUP);
    double L = 0.0, R = 0.0;
    for (int c = c_begin; c < c_end; c++) {
        const gdg_spat_chan ch = chans[c];
        const double *x = in + (size_t)c * in_stride;
        const double cur = x[j];
        if (ch.mode == 0) {
            L += ch.fac_left * cur;
            R += ch.fac_right * cur;
        } else {
            const double *hb = hist + (size_t)c * H;
            int ie = j - ch.early, il = j - ch.late;
            double se = (ie >= 0) ? x[ie] : hb[H + ie];
            double sl = (il >= 0) ? x[il] : hb[H + il];
            double early_sample = ch.w_early * se;
            double late_sample = ch.w_late * sl;
            double delayed = early_sample + late_sample;
            if (ch.mode == 1) { L += ch.fac_left * delayed; R += ch.fac_right * cur; }
            else { L += ch.fac_left * cur; R += ch.fac_right * delayed; }
        }
    }
    partial[((size_t)g * 2 + 0) * max_frames + j] = L;
    partial[((size_t)g * 2 + 1) * max_frames + j] = R;
}

__global__ void __launch_bounds__(256)
spat_reduce_kernel(const double *__restrict__ partial, int groups, double *__restrict__ out_lr, int out_stride, int frames, int max_frames) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= frames) return;
    double L = 0.0, R = 0.0;
    for (int g = 0; g < groups; g++) {
        L += partial[((size_t)g * 2 + 0) * max_frames + j];
        R += partial[((size_t)g * 2 + 1) * max_frames + j];
    }
    out_lr[j] = L;
    out_lr[(size_t)out_stride + j] = R;
}

/* history = the last H inputs of every channel (spatializer.go:313-331); one workgroup per channel */
__global__ void __launch_bounds__(256)
spat_hist_kernel(const double *__restrict__ in, int in_stride, double *__restrict__ hist, int H, int frames) {
    const int c = blockIdx.x;
    double *hb = hist + (size_t)c * H;
    const double *x = in + (size_t)c * in_stride;
    double keep[4];
    int cnt = 0;
    for (int k = threadIdx.x; k < H && cnt < 4; k += 256, cnt++) {
        int src = k + frames - H;                    /* index into the concatenation [old history | frame] shifted by H */
        keep[cnt] = (src >= 0) ? x[src] : hb[k + frames];
    }
    __syncthreads();
    cnt = 0;
    for (int k = threadIdx.x; k < H && cnt < 4; k += 256, cnt++) hb[k] = keep[cnt];
}

hipError_t gdg_launch_spatializer(const gdg_spat_chan *d_chans, int nch, const double *d_in, int in_stride, double *d_hist, int H,
                                  double *d_partial, double *d_out_lr, int out_stride, int frames, int max_frames, hipStream_t s) {
    if (H > 1024) return hipErrorInvalidValue;
    const int groups = (nch + SPAT_GROUP - 1) / SPAT_GROUP;
    const int tiles = (frames + 255) / 256;
    spat_partial_kernel<<<dim3(tiles, groups), dim3(256), 0, s>>>(d_chans, nch, d_in, in_stride, d_hist, H, d_partial, frames, max_frames);
    spat_reduce_kernel<<<dim3(tiles), dim3(256), 0, s>>>(d_partial, groups, d_out_lr, out_stride, frames, max_frames);
    spat_hist_kernel<<<dim3(nch), dim3(256), 0, s>>>(d_in, in_stride, d_hist, H, frames);
    return hipGetLastError();
}

int gdg_spat_groups(int nch) { return (nch + SPAT_GROUP - 1) / SPAT_GROUP; }
