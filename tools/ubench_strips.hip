// Is a column-strip march (the access pattern of the spline sweeps) slower than a flat copy for DRAM reasons alone?
//   hipcc -O3 --offload-arch=gfx950 -o build/ubench_strips tools/ubench_strips.hip && build/ubench_strips
// Every kernel copies nt levels of ny x nx doubles (read once, written once): flat (consecutive threads, consecutive
// elements, 16 B per lane), and strip marches -- a thread owns one column element and walks down the rows, R rows of loads in
// flight, then R stores -- with workgroups of 256 / 512 / 1024 threads (a workgroup's strip is 2 / 4 / 8 KB of a 16 KB row).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int R>
__global__ void strip_march(const double *__restrict__ in, double *__restrict__ out, int nt, int ny, int nx) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)nt * nx) return;
    const size_t t = i / nx, x = i - t * nx;
    const double *src = in + t * (size_t)ny * nx + x;
    double *dst = out + t * (size_t)ny * nx + x;
    for (int r0 = 0; r0 < ny; r0 += R) {
        double a[R];
#pragma unroll
        for (int q = 0; q < R; ++q) a[q] = src[(size_t)(r0 + q) * nx];
#pragma unroll
        for (int q = 0; q < R; ++q) dst[(size_t)(r0 + q) * nx] = a[q] * 1.0000001;
    }
}
// the same march with a dependent chain of CH fused multiply-adds per element (what a recursion adds)
template <int R, int CH>
__global__ void strip_march_chain(const double *__restrict__ in, double *__restrict__ out, int nt, int ny, int nx) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)nt * nx) return;
    const size_t t = i / nx, x = i - t * nx;
    const double *src = in + t * (size_t)ny * nx + x;
    double *dst = out + t * (size_t)ny * nx + x;
    double prev = 0.0;
    for (int r0 = 0; r0 < ny; r0 += R) {
        double a[R];
#pragma unroll
        for (int q = 0; q < R; ++q) a[q] = src[(size_t)(r0 + q) * nx];
#pragma unroll
        for (int q = 0; q < R; ++q) {
#pragma unroll
            for (int c = 0; c < CH; ++c) prev = fma(-0.2679, prev, a[q]);
            dst[(size_t)(r0 + q) * nx] = prev;
        }
    }
}
__global__ void flat_copy(const double2 *__restrict__ in, double2 *__restrict__ out, size_t n2) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        double2 v = in[i];
        v.x *= 1.0000001;
        out[i] = v;
    }
}

int main() {
    const int nt = 201, ny = 1024, nx = 2048;
    const size_t n = (size_t)nt * ny * nx;
    double *a, *b;
    hipMalloc(&a, n * 8);
    hipMalloc(&b, n * 8);
    hipMemset(a, 0, n * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms, 2.0 * n * 8 / ms / 1e9);
    };
    const size_t lines = (size_t)nt * nx;
    time("flat copy, 16 B per lane", [&] { hipLaunchKernelGGL(flat_copy, dim3(8192), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, n / 2); });
    time("strip march, 16 rows in flight, block 256", [&] { hipLaunchKernelGGL(strip_march<16>, dim3((lines + 255) / 256), dim3(256), 0, 0, a, b, nt, ny, nx); });
    time("strip march, 16 rows in flight, block 512", [&] { hipLaunchKernelGGL(strip_march<16>, dim3((lines + 511) / 512), dim3(512), 0, 0, a, b, nt, ny, nx); });
    time("strip march, 16 rows in flight, block 1024", [&] { hipLaunchKernelGGL(strip_march<16>, dim3((lines + 1023) / 1024), dim3(1024), 0, 0, a, b, nt, ny, nx); });
    time("strip march, 8 rows in flight, block 256", [&] { hipLaunchKernelGGL(strip_march<8>, dim3((lines + 255) / 256), dim3(256), 0, 0, a, b, nt, ny, nx); });
    time("strip march, 32 rows in flight, block 256", [&] { hipLaunchKernelGGL(strip_march<32>, dim3((lines + 255) / 256), dim3(256), 0, 0, a, b, nt, ny, nx); });
    time("strip march + chain of 4 fma, block 256", [&] { hipLaunchKernelGGL((strip_march_chain<16, 4>), dim3((lines + 255) / 256), dim3(256), 0, 0, a, b, nt, ny, nx); });
    time("strip march + chain of 8 fma, block 256", [&] { hipLaunchKernelGGL((strip_march_chain<16, 8>), dim3((lines + 255) / 256), dim3(256), 0, 0, a, b, nt, ny, nx); });
    time("strip march + chain of 16 fma, block 256", [&] { hipLaunchKernelGGL((strip_march_chain<16, 16>), dim3((lines + 255) / 256), dim3(256), 0, 0, a, b, nt, ny, nx); });
    return 0;
}
