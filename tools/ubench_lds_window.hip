// LDS cost of a 4 x 4 window of {u, v} float32 nodes in two tile forms, with the two-seed order-3 kernel's lane pattern
// (round-3 review item 6: "{node, x-difference} entries read as 8 ds_read_b128 instead of 16 ds_read_b64"):
//   A  8-byte nodes {u, v}, pitch 36 nodes (advect_lds2_o3_kernel's tile): rows of 4 x ds_read_b64              = 16 reads
//   B  16-byte entries {u, v, u[x+1]-u, v[x+1]-v}, pitch 36 or 40 entries: rows of 2 x ds_read_b128 (x0, x0+2) =  8 reads
// One wave = 8 x 16 seeds (two per lane, stacked); seeds DENS per node in x and y (C3: 4 per node in x, 5.3 in y -> 4;
// 2 = half as dense).  Every wave loops over windows with an s_waitcnt per window (the kernel consumes each window
// before the next); W waves per SIMD on all 4 SIMDs of all 256 CUs.  Reported: SIMD cycles per window per wave-slot,
// i.e. (a wave's cycles) / (windows x W) -- what one more window costs the SIMD -- and the LDS bytes per workgroup.
//   hipcc -O3 --offload-arch=gfx950 -o build/ubench_lds_window tools/ubench_lds_window.hip && build/ubench_lds_window
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int COLS = 32, ROWS = 16;

template <int FORM, int PITCH, int DENS>
__global__ void __launch_bounds__(256) k(float *out, long long *cyc, int iters) {
    constexpr int NODE = FORM == 0 ? 8 : 16;
    __shared__ __attribute__((aligned(16))) char tile[4][ROWS * PITCH * NODE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < ROWS * PITCH * NODE / 4; i += 64) ((float *)tile[wave])[i] = (float)(i & 255) * 0.001f;
    __syncthreads();
    // window origins of the lane's two seeds inside the tile (patch at the tile's middle, as the kernel anchors it)
    const int lx = lane & 7, ly = lane >> 3;
    const int ox = (COLS - 4) / 2 - 1 + lx / DENS, oy0 = (ROWS - 4) / 2 - 2 + ly / DENS, oy1 = (ROWS - 4) / 2 - 2 + (ly + 8) / DENS;
    typedef __attribute__((address_space(3))) const f2 lds_f2;
    typedef __attribute__((address_space(3))) const f4 lds_f4;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)tile[wave];
    const unsigned a0 = base + (unsigned)(min(oy0, ROWS - 4) * PITCH + ox) * NODE, a1 = base + (unsigned)(min(oy1, ROWS - 4) * PITCH + ox) * NODE;
    f2 acc = {0.0f, 0.0f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned a = (q ? a1 : a0) + (unsigned)((it & 1) * NODE);  // (moves by one node every other window: no hoisting)
            if (FORM == 0) {
                f2 n[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) n[r][c] = *(lds_f2 *)(size_t)(a + (unsigned)(r * PITCH + c) * 8u);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc += n[r][0] + n[r][1] + n[r][2] + n[r][3];
            } else {
                f4 e[4][2];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) e[r][c] = *(lds_f4 *)(size_t)(a + (unsigned)(r * PITCH + 2 * c) * 16u);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc += e[r][0].xy + e[r][0].zw + e[r][1].xy + e[r][1].zw;
            }
            asm volatile("" : "+v"(acc));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (acc.x == 12345.678f) out[threadIdx.x] = acc.y;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int FORM, int PITCH, int DENS>
static void run(const char *name) {
    float *out;
    long long *cyc;
    hipMalloc(&out, 4096);
    hipMalloc(&cyc, sizeof(long long) * 256 * 8 * 4);
    const int iters = 4000;
    const int lds = 4 * ROWS * PITCH * (FORM == 0 ? 8 : 16);
    printf("%-44s LDS %5d B/workgroup (%d per CU):", name, lds, 163840 / lds > 8 ? 8 : 163840 / lds);
    for (int wps : {1, 2, 4, 5, 8}) {
        if (wps * lds > 163840) {
            printf("  W=%d: does not fit", wps);
            continue;
        }
        const int blocks = 256 * wps;
        hipLaunchKernelGGL((k<FORM, PITCH, DENS>), dim3(blocks), dim3(256), 0, 0, out, cyc, 100);
        hipDeviceSynchronize();
        hipLaunchKernelGGL((k<FORM, PITCH, DENS>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
        double avg = 0;
        for (long long v : h) avg += (double)v;
        avg /= (double)h.size();
        printf("  W=%d: %6.1f", wps, avg / (2.0 * iters) / wps);
    }
    printf("   [s_memtime ticks per window and wave-slot]\n");
    hipFree(out);
    hipFree(cyc);
}

int main() {
    printf("seeds 4 per node (C3-like patch)\n");
    run<0, 36, 4>("A  {u,v} 8 B, pitch 36, 16 x ds_read_b64");
    run<1, 36, 4>("B  {u,v,du,dv} 16 B, pitch 36, 8 x ds_read_b128");
    run<1, 40, 4>("B  {u,v,du,dv} 16 B, pitch 40, 8 x ds_read_b128");
    run<1, 34, 4>("B  {u,v,du,dv} 16 B, pitch 34, 8 x ds_read_b128");
    printf("seeds 2 per node\n");
    run<0, 36, 2>("A  {u,v} 8 B, pitch 36, 16 x ds_read_b64");
    run<1, 36, 2>("B  {u,v,du,dv} 16 B, pitch 36, 8 x ds_read_b128");
    run<1, 40, 2>("B  {u,v,du,dv} 16 B, pitch 40, 8 x ds_read_b128");
    run<1, 34, 2>("B  {u,v,du,dv} 16 B, pitch 34, 8 x ds_read_b128");
    return 0;
}
