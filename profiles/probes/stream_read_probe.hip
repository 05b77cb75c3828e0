// Practical streaming-read ceiling of the box: sum-reduce 1 GiB with 16-byte loads per lane, several shapes.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/stream_read_probe profiles/probes/stream_read_probe.hip && gpurun_out/stream_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) read_kernel(const v2d *__restrict__ p, size_t n, double *__restrict__ out) {
    // block b reads UNROLL chunks of 256 elements (4 KiB) that lie `stride` apart, like the MAC's 16 streams
    size_t chunk = (size_t)blockIdx.x;
    size_t nchunks = n / 256;
    size_t per = nchunks / UNROLL;
    double acc = 0.0;
    v2d v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        const v2d *q = p + ((size_t)u * per + chunk) * 256 + threadIdx.x;
        v[u] = NT ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc += v[u].x * v[u].y;
    if (acc == 123.456) out[0] = acc;
}

template <bool NT>
__global__ void __launch_bounds__(256) linear_kernel(const v2d *__restrict__ p, size_t n, double *__restrict__ out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        v2d v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        acc += v.x * v.y;
    }
    if (acc == 123.456) out[0] = acc;
}

// the MAC's real situation: every channel's delay line and IR spectra are separate 1 MiB allocations
template <bool NT>
__global__ void __launch_bounds__(256) multi_alloc_kernel(const v2d *const *__restrict__ bufs, double *__restrict__ out) {
    // grid (32 tiles, pairs): pair y reads bufs[2y] and bufs[2y+1], 8 chunks of 4 KiB each, 128 KiB apart
    const v2d *a = bufs[2 * blockIdx.y], *b = bufs[2 * blockIdx.y + 1];
    double acc = 0.0;
    v2d va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        size_t off = (size_t)u * 8192 + (size_t)blockIdx.x * 256 + threadIdx.x;
        va[u] = NT ? __builtin_nontemporal_load(a + off) : a[off];
        vb[u] = NT ? __builtin_nontemporal_load(b + off) : b[off];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) acc += va[u].x * vb[u].y;
    if (acc == 123.456) out[0] = acc;
}

// ... plus what the real kernel does before its first load: descriptor fetch, *pos fetch, slot rotation; and its Y store
struct Desc { const v2d *a, *b; v2d *y; const int *pos; int K; int pad; };
template <bool NT, bool CHASE, int STORE>
__global__ void __launch_bounds__(256) mac_like_kernel(const Desc *__restrict__ descs, double *__restrict__ out) {
    Desc ch = descs[blockIdx.y];
    int cur = CHASE ? (*ch.pos) % ch.K : 3;
    const int b0 = blockIdx.x * 256 + threadIdx.x;
    double ar = 0.0, ai = 0.0;
    v2d va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        int slot = cur - u; if (slot < 0) slot += 8;
        va[u] = NT ? __builtin_nontemporal_load(ch.a + (size_t)slot * 8192 + b0) : ch.a[(size_t)slot * 8192 + b0];
        vb[u] = NT ? __builtin_nontemporal_load(ch.b + (size_t)u * 8192 + b0) : ch.b[(size_t)u * 8192 + b0];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) { ar += va[u].x * vb[u].x - va[u].y * vb[u].y; ai += va[u].x * vb[u].y + va[u].y * vb[u].x; }
    if (STORE == 1) { v2d r = { ar, ai }; ch.y[b0] = r; }
    else if (STORE == 2) { v2d r = { ar, ai }; __builtin_nontemporal_store(r, ch.y + b0); }
    else if (ar == 123.456) out[0] = ai;
}

__global__ void fill_kernel(double *p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned long long x = (i + seed) * 6364136223846793005ull + 1442695040888963407ull;
        x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32;
        p[i] = (double)(long long)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}
static void fill(void *p, size_t bytes, unsigned seed) { fill_kernel<<<4096, 256>>>((double *)p, bytes / 8, seed); }

// the fused kernel's shape: ONE 512-thread workgroup per channel walks partition by partition through all bins, k ascending
// and n = N - k descending (REV), 32 loads per lane per partition
template <bool REV>
__global__ void __launch_bounds__(512) fused_shape_kernel(const Desc *__restrict__ descs, double *__restrict__ out) {
    Desc ch = descs[blockIdx.x];
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int u = 0; u < 8; u++) {
        const v2d *x = ch.a + (size_t)u * 8192, *h = ch.b + (size_t)u * 8192;
        v2d xk[8], hk[8], xn[8], hn[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int k = tid + 512 * i, n = REV ? 8191 - k : 4096 + k;
            xk[i] = __builtin_nontemporal_load(x + k); hk[i] = __builtin_nontemporal_load(h + k);
            xn[i] = __builtin_nontemporal_load(x + n); hn[i] = __builtin_nontemporal_load(h + n);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) acc += xk[i].x * hk[i].y + xn[i].x * hn[i].y;
    }
    if (acc == 123.456) out[0] = acc;
}

template <typename F> static float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 10; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}

int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    v2d *d; double *o;
    hipMalloc(&d, bytes); hipMalloc(&o, 8); fill(d, bytes, 1);   /* random bits (zero-filled buffers give the same numbers) */
    auto rep = [&](const char *name, float ms) { printf("%-44s %8.1f us  %7.0f GB/s  %.3f of 8 TB/s\n", name, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / 8000); };
    rep("16 streams x 4 KiB per block, NT", timeit([&] { read_kernel<16, true><<<n / 256 / 16, 256>>>(d, n, o); }));
    rep("16 streams x 4 KiB per block", timeit([&] { read_kernel<16, false><<<n / 256 / 16, 256>>>(d, n, o); }));
    rep("8 streams x 4 KiB per block, NT", timeit([&] { read_kernel<8, true><<<n / 256 / 8, 256>>>(d, n, o); }));
    rep("4 streams x 4 KiB per block, NT", timeit([&] { read_kernel<4, true><<<n / 256 / 4, 256>>>(d, n, o); }));
    rep("1 stream  x 4 KiB per block, NT", timeit([&] { read_kernel<1, true><<<n / 256, 256>>>(d, n, o); }));
    for (int g : {1024, 2048, 4096, 16384}) {
        char nm[64];
        snprintf(nm, sizeof nm, "grid-stride linear, %d blocks, NT", g);
        rep(nm, timeit([&] { linear_kernel<true><<<g, 256>>>(d, n, o); }));
        snprintf(nm, sizeof nm, "grid-stride linear, %d blocks", g);
        rep(nm, timeit([&] { linear_kernel<false><<<g, 256>>>(d, n, o); }));
    }
    {
        // 1024 separate 1 MiB allocations vs the same 1024 pieces carved from the one slab
        std::vector<v2d *> sep(1024), slab(1024);
        for (int i = 0; i < 1024; i++) { hipMalloc(&sep[i], 1 << 20); fill(sep[i], 1 << 20, 100 + i); slab[i] = d + (size_t)i * 65536; }
        const v2d **d_sep, **d_slab;
        hipMalloc(&d_sep, 1024 * sizeof(void *)); hipMalloc(&d_slab, 1024 * sizeof(void *));
        hipMemcpy(d_sep, sep.data(), 1024 * sizeof(void *), hipMemcpyHostToDevice);
        hipMemcpy(d_slab, slab.data(), 1024 * sizeof(void *), hipMemcpyHostToDevice);
        rep("MAC pattern, 1024 separate 1 MiB hipMallocs, NT", timeit([&] { multi_alloc_kernel<true><<<dim3(32, 512), 256>>>(d_sep, o); }));
        rep("MAC pattern, same pieces of one 1 GiB slab, NT", timeit([&] { multi_alloc_kernel<true><<<dim3(32, 512), 256>>>(d_slab, o); }));
        rep("MAC pattern, 1024 separate 1 MiB hipMallocs", timeit([&] { multi_alloc_kernel<false><<<dim3(32, 512), 256>>>(d_sep, o); }));
        rep("MAC pattern, same pieces of one 1 GiB slab", timeit([&] { multi_alloc_kernel<false><<<dim3(32, 512), 256>>>(d_slab, o); }));
    }
    {
        std::vector<Desc> h(512);
        int *d_pos; hipMalloc(&d_pos, 512 * sizeof(int)); hipMemset(d_pos, 0, 512 * sizeof(int));
        for (int i = 0; i < 512; i++) {
            v2d *a, *b, *y;
            hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20); hipMalloc(&y, 1 << 17);
            fill(a, 1 << 20, 7 * i); fill(b, 1 << 20, 7 * i + 3);
            h[i] = Desc{ a, b, y, d_pos + i, 8, 0 };
        }
        Desc *dd; hipMalloc(&dd, 512 * sizeof(Desc)); hipMemcpy(dd, h.data(), 512 * sizeof(Desc), hipMemcpyHostToDevice);
        rep("fused shape: 512-thread block per channel, k asc + n desc", timeit([&] { fused_shape_kernel<true><<<512, 512>>>(dd, o); }));
        rep("fused shape: 512-thread block per channel, both ascending", timeit([&] { fused_shape_kernel<false><<<512, 512>>>(dd, o); }));
        rep("MAC-like, descriptors only, NT", timeit([&] { mac_like_kernel<true, false, 0><<<dim3(32, 512), 256>>>(dd, o); }));
        rep("MAC-like, + *pos chase, NT", timeit([&] { mac_like_kernel<true, true, 0><<<dim3(32, 512), 256>>>(dd, o); }));
        rep("MAC-like, + *pos chase + Y store, NT", timeit([&] { mac_like_kernel<true, true, 1><<<dim3(32, 512), 256>>>(dd, o); }));
        rep("MAC-like, + *pos chase + NT Y store, NT", timeit([&] { mac_like_kernel<true, true, 2><<<dim3(32, 512), 256>>>(dd, o); }));
        rep("MAC-like, descriptors + Y store, NT", timeit([&] { mac_like_kernel<true, false, 1><<<dim3(32, 512), 256>>>(dd, o); }));
    }
    return 0;
}
