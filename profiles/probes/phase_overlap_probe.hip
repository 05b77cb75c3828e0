// Does a second resident workgroup per CU overlap the memory phases of a segment-like kernel with the compute phases of its
// neighbour -- and does it need a stagger?  Synthetic workgroup = [load frame 64 KiB -> LDS] [VALU C1] [fetch RING bytes, discard]
// [VALU C2] [append 64 KiB] [store frame 64 KiB], 512 channels, shaped like the cabinet > reverb segment of the bench.
//   A: 1024 threads, 160 KiB LDS (1 workgroup per CU, two rounds)          -- today's seg_kernel shape
//   B: 512 threads, 80 KiB LDS (2 per CU, one round), no stagger
//   C: like B, workgroups with blockIdx >= half start `delay` cycles late
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/phase_probe profiles/probes/phase_overlap_probe.hip && /tmp/phase_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int T>
__global__ void __launch_bounds__(T) probe(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ rings,
                                           double *__restrict__ appendbuf, int ring_doubles, int valu1, int valu2, long long delay, int half) {
    extern __shared__ double lds[];
    constexpr int N = 8192;
    const int tid = threadIdx.x, ch = blockIdx.x;
    if (delay > 0 && (int)blockIdx.x >= half) {
        long long t0 = __builtin_readcyclecounter();
        while ((long long)__builtin_readcyclecounter() - t0 < delay) __builtin_amdgcn_s_sleep(32);
    }
    const v2d *src = (const v2d *)(in + (size_t)ch * N);
    v2d v[N / 2 / T];
#pragma unroll
    for (int q = 0; q < N / 2 / T; q++) v[q] = src[tid + q * T];
#pragma unroll
    for (int q = 0; q < N / 2 / T; q++) { lds[2 * (tid + q * T)] = v[q].x; lds[2 * (tid + q * T) + 1] = v[q].y; }
    __syncthreads();
    // VALU phase 1: dependent fp64 chain per sample (scaled so that the whole workgroup spends `valu1` fma per sample)
    double acc[N / T];
#pragma unroll
    for (int q = 0; q < N / T; q++) acc[q] = lds[tid + q * T];
    for (int it = 0; it < valu1; it++) {
#pragma unroll
        for (int q = 0; q < N / T; q++) acc[q] = fma(acc[q], 0.999999, 1e-9);
    }
    __syncthreads();
    // memory phase: fetch the "ring" (16-byte loads, all in flight in batches of 8), fold into acc
    const v2d *ring = (const v2d *)(rings + (size_t)ch * ring_doubles);
    const int n2 = ring_doubles / 2;
    for (int base = 0; base < n2; base += 8 * T) {
        v2d r[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { int i = base + u * T + tid; r[u] = (i < n2) ? ring[i] : v2d{0.0, 0.0}; }
#pragma unroll
        for (int u = 0; u < 8; u++) acc[u % (N / T)] += r[u].x * 1e-30 + r[u].y * 1e-30;
    }
    __syncthreads();
    for (int it = 0; it < valu2; it++) {
#pragma unroll
        for (int q = 0; q < N / T; q++) acc[q] = fma(acc[q], 0.999999, 1e-9);
    }
    // append 64 KiB + store frame
    v2d *ap = (v2d *)(appendbuf + (size_t)ch * N);
    v2d *dst = (v2d *)(out + (size_t)ch * N);
#pragma unroll
    for (int q = 0; q < N / 2 / T; q++) {
        v2d w = { acc[(2 * q) % (N / T)], acc[(2 * q + 1) % (N / T)] };
        ap[tid + q * T] = w;
        dst[tid + q * T] = w;
    }
}

int main() {
    const int nch = 512, N = 8192, ring_doubles = 44800;            // ~350 KB per channel like reverb's tap window + all-pass rings
    double *in, *out, *rings, *app, *flush;
    hipMalloc(&in, (size_t)nch * N * 8); hipMalloc(&out, (size_t)nch * N * 8); hipMalloc(&app, (size_t)nch * N * 8);
    hipMalloc(&rings, (size_t)nch * ring_doubles * 8);
    size_t flush_n = (size_t)1 << 27;
    hipMalloc(&flush, flush_n * 8);
    hipMemset(in, 0, (size_t)nch * N * 8); hipMemset(rings, 0, (size_t)nch * ring_doubles * 8);
    hipFuncSetAttribute((const void *)probe<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)probe<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    // valu counts: a cabinet + all-pass/mix worth of issue slots: ~30 000 + 11 000 cycles per workgroup of 1024 threads
    // per fma iteration: (N / T) fma per thread = 8192 lane-fma per workgroup = 128 wave-instr = 128 cycles per CU with 4 SIMDs busy
    struct Cfg { const char *name; int T; size_t lds; long long delay; } cfgs[] = {
        { "A: 1 WG/CU, 1024 threads, 2 rounds", 1024, 159 * 1024, 0 },
        { "B: 2 WG/CU,  512 threads, no stagger", 512, 79 * 1024, 0 },
        { "C: 2 WG/CU, stagger 10 000 cycles", 512, 79 * 1024, 10000 },
        { "C: 2 WG/CU, stagger 20 000 cycles", 512, 79 * 1024, 20000 },
        { "C: 2 WG/CU, stagger 30 000 cycles", 512, 79 * 1024, 30000 },
        { "C: 2 WG/CU, stagger 45 000 cycles", 512, 79 * 1024, 45000 },
    };
    for (int valu = 0; valu < 2; valu++) {
        int v1 = valu ? 230 : 0, v2 = valu ? 90 : 0;
        printf("VALU phases: %d + %d fma iterations (%s)\n", v1, v2, valu ? "cabinet > reverb like" : "memory phases only");
        for (auto &c : cfgs) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; rep++) {
                hipMemsetAsync(flush, rep, flush_n * 8, 0);           // evict L2 / MALL like the MAC's 1 GB stream does
                hipEventRecord(a, 0);
                if (c.T == 1024) probe<1024><<<nch, 1024, c.lds, 0>>>(in, out, rings, app, ring_doubles, v1, v2, c.delay, nch / 2);
                else probe<512><<<nch, 512, c.lds, 0>>>(in, out, rings, app, ring_doubles, v1, v2, c.delay, nch / 2);
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("  %-40s %8.1f us\n", c.name, best * 1e3f);
        }
    }
    return 0;
}
