// Does a buffer written by one kernel come back from the 256 MB infinity cache (MALL) when the next kernel reads it?
// For footprints of 16 MB .. 2 GB: W = write the buffer, R = read it; cycles W R W R ... and prints the time of each.
// If writes allocate in the cache, R after W is fast (and W over the same lines again too) while the footprint fits.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_probe profiles/probes/mall_probe.hip && /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v2d __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) write_kernel(v2d *__restrict__ p, size_t n, double x) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { v2d v = { x, (double)i }; p[i] = v; }
}
__global__ void __launch_bounds__(256) read_kernel(const v2d *__restrict__ p, size_t n, double *__restrict__ out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { v2d v = p[i]; acc += v.x * v.y; }
    if (acc == 123.456) out[0] = acc;
}
__global__ void __launch_bounds__(256) read_nt_kernel(const v2d *__restrict__ p, size_t n, double *__restrict__ out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { v2d v = __builtin_nontemporal_load(p + i); acc += v.x * v.y; }
    if (acc == 123.456) out[0] = acc;
}

int main() {
    const size_t cap = (size_t)2 << 30;
    v2d *buf; double *out; v2d *flush;
    hipMalloc(&buf, cap); hipMalloc(&out, 8); hipMalloc(&flush, cap);
    hipEvent_t e[4]; for (auto &x : e) hipEventCreate(&x);
    printf("MB,write_us,read_after_write_us,read_again_us,read_after_flush_us,write_GBs,read_after_write_GBs,read_again_GBs,cold_read_GBs\n");
    for (size_t mb : { 16, 32, 64, 96, 128, 192, 256, 512, 2048 }) {
        const size_t bytes = mb << 20, n = bytes / 16;
        const int grid = 256 * 16;
        float tw = 0, tr = 0, tr2 = 0, tc = 0;
        for (int rep = 0; rep < 5; rep++) {
            write_kernel<<<grid, 256>>>(flush, cap / 16, 1.0);            // push everything else out
            hipDeviceSynchronize();
            hipEventRecord(e[0]); write_kernel<<<grid, 256>>>(buf, n, 2.0 + rep);
            hipEventRecord(e[1]); read_kernel<<<grid, 256>>>(buf, n, out);
            hipEventRecord(e[2]); read_kernel<<<grid, 256>>>(buf, n, out);
            hipEventRecord(e[3]); hipDeviceSynchronize();
            hipEventElapsedTime(&tw, e[0], e[1]); hipEventElapsedTime(&tr, e[1], e[2]); hipEventElapsedTime(&tr2, e[2], e[3]);
            write_kernel<<<grid, 256>>>(flush, cap / 16, 1.0);
            hipDeviceSynchronize();
            hipEventRecord(e[0]); read_kernel<<<grid, 256>>>(buf, n, out); hipEventRecord(e[1]); hipDeviceSynchronize();
            hipEventElapsedTime(&tc, e[0], e[1]);
        }
        printf("%zu,%.1f,%.1f,%.1f,%.1f,%.0f,%.0f,%.0f,%.0f\n", mb, tw * 1e3, tr * 1e3, tr2 * 1e3, tc * 1e3, bytes / tw / 1e6, bytes / tr / 1e6, bytes / tr2 / 1e6, bytes / tc / 1e6);
    }
    // steady state: W R W R over the SAME footprint without flushes in between (what a reused scratch buffer sees)
    printf("MB,steady_write_us,steady_read_us,write_GBs,read_GBs\n");
    for (size_t mb : { 32, 64, 128, 256, 1024 }) {
        const size_t bytes = mb << 20, n = bytes / 16;
        const int grid = 256 * 16;
        float tw = 0, tr = 0;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(e[0]); write_kernel<<<grid, 256>>>(buf, n, 2.0 + rep);
            hipEventRecord(e[1]); read_nt_kernel<<<grid, 256>>>(buf, n, out);
            hipEventRecord(e[2]); hipDeviceSynchronize();
            hipEventElapsedTime(&tw, e[0], e[1]); hipEventElapsedTime(&tr, e[1], e[2]);
        }
        printf("%zu,%.1f,%.1f,%.0f,%.0f\n", mb, tw * 1e3, tr * 1e3, bytes / tw / 1e6, bytes / tr / 1e6);
    }
    return 0;
}
