// Issue-rate microbenchmark: cycles one SIMD needs per wave64 instruction, by opcode and by waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o ubench_valu tools/ubench_valu.hip && ./ubench_valu
// Each kernel runs N independent-chain instructions per wave in a loop; the grid is 256 CUs x 4 SIMDs x W
// waves.  Reported: SIMD cycles per wave-instruction = time x clock / (instructions per wave x W), with the
// clock taken from s_memtime / wall time inside the same launch.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define BODY(asm_line)                                                                   \
    for (int it = 0; it < iters; ++it) {                                                 \
        REP8(REP8(asm volatile(asm_line : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)) \
    }

enum Op { FMA, PKFMA, PKMUL, PKADD, CVTFLR, FRACT, MED3, MADU24, CMP, ADD, MIXPK, NOPS, VS, VN, SALU, VS2, LDSB64, LDSREAD2, FMA64, MUL64, ADD64, RCP64, FLOOR64, CVTI64, DIVSC64, DIVFIX64, CNDMASK };

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, long long *cyc, int iters, long long *rt) {
    __shared__ f2 lds[2048];
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a0}, p3 = {a3, a2};
    const f2 pb = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f};
    const float b = 1.0001f, c = 0.5f;
    unsigned u0 = threadIdx.x, u1 = u0 + 7, u2 = u0 + 9, u3 = u0 + 11;
    lds[threadIdx.x] = p0;
    lds[threadIdx.x + 256] = p1;
    __syncthreads();
    const unsigned la = (threadIdx.x & 63) * 8;
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (OP == FMA) {
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));))
        }
    } else if (OP == ADD) {
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));))
        }
    } else if (OP == PKFMA) {
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));))
        }
    } else if (OP == PKMUL) {
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));))
        }
    } else if (OP == PKADD) {
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));))
        }
    } else if (OP == MIXPK) {  // 2 packed + 2 plain
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_fma_f32 %2, %2, %6, %7\n v_pk_fma_f32 %1, %1, %4, %5\n v_fma_f32 %3, %3, %6, %7"
                                   : "+v"(p0), "+v"(p1), "+v"(a2), "+v"(a3) : "v"(pb), "v"(pc), "v"(b), "v"(c));))
        }
    } else if (OP == CVTFLR) {
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_cvt_flr_i32_f32 %0, %4\n v_cvt_flr_i32_f32 %1, %5\n v_cvt_flr_i32_f32 %2, %4\n v_cvt_flr_i32_f32 %3, %5"
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(b), "v"(c));))
        }
    } else if (OP == FRACT) {
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));))
        }
    } else if (OP == MED3) {
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));))
        }
    } else if (OP == MADU24) {
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5"
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0), "v"(u1));))
        }
    } else if (OP == CMP) {
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4"
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0), "v"(u1) : "vcc");))
        }
    } else if (OP == NOPS) {
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));))
        }
    } else if (OP == VS) {  // 4 VALU (4-cycle class) + 4 SALU interleaved
        unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_fract_f32 %0, %0\n s_add_u32 %4, %4, 1\n v_fract_f32 %1, %1\n s_add_u32 %5, %5, 1\n v_fract_f32 %2, %2\n s_add_u32 %6, %6, 1\n v_fract_f32 %3, %3\n s_add_u32 %7, %7, 1"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");))
        }
        u0 += s0 + s1 + s2 + s3;
    } else if (OP == VS2) {  // 4 VALU + 8 SALU
        unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_fract_f32 %0, %0\n s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n v_fract_f32 %1, %1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n v_fract_f32 %2, %2\n s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n v_fract_f32 %3, %3\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");))
        }
        u0 += s0 + s1 + s2 + s3;
    } else if (OP == VN) {  // 4 VALU + 4 s_nop
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_fract_f32 %0, %0\n s_nop 0\n v_fract_f32 %1, %1\n s_nop 0\n v_fract_f32 %2, %2\n s_nop 0\n v_fract_f32 %3, %3\n s_nop 0"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));))
        }
    } else if (OP == SALU) {
        unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1"
                                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");))
        }
        u0 += s0 + s1 + s2 + s3;
    } else if (OP == LDSB64) {
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:512\n ds_read_b64 %3, %4 offset:520\n s_waitcnt lgkmcnt(0)"
                                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(la));))
        }
    } else if (OP == LDSREAD2) {
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("ds_read2_b64 %0, %2 offset1:1\n ds_read2_b64 %1, %2 offset0:64 offset1:65\n s_waitcnt lgkmcnt(0)"
                                   : "+v"(*(float __attribute__((ext_vector_type(4))) *)&p0), "+v"(*(float __attribute__((ext_vector_type(4))) *)&p2) : "v"(la));))
        }
    }
    double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
    const double db = 1.0001, dc = 0.5;
#define D64(line)                                                                                            \
    for (int it = 0; it < iters; ++it) {                                                                     \
        REP8(REP8(asm volatile(line : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));))          \
    }
    if (OP == FMA64) {
        D64("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5")
    } else if (OP == MUL64) {
        D64("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4")
    } else if (OP == ADD64) {
        D64("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4")
    } else if (OP == RCP64) {
        D64("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3")
    } else if (OP == FLOOR64) {
        D64("v_floor_f64 %0, %0\n v_floor_f64 %1, %1\n v_floor_f64 %2, %2\n v_floor_f64 %3, %3")
    } else if (OP == CVTI64) {
        for (int it = 0; it < iters; ++it) {
            REP8(REP8(asm volatile("v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %4\n v_cvt_i32_f64 %3, %5"
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(db), "v"(dc));))
        }
    } else if (OP == DIVSC64) {
        D64("v_div_scale_f64 %0, vcc, %0, %4, %5\n v_div_scale_f64 %1, vcc, %1, %4, %5\n v_div_scale_f64 %2, vcc, %2, %4, %5\n v_div_scale_f64 %3, vcc, %3, %4, %5")
    } else if (OP == DIVFIX64) {
        D64("v_div_fixup_f64 %0, %0, %4, %5\n v_div_fixup_f64 %1, %1, %4, %5\n v_div_fixup_f64 %2, %2, %4, %5\n v_div_fixup_f64 %3, %3, %4, %5")
    } else if (OP == CNDMASK) {
        BODY("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc")
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(d0 + d1 + d2 + d3) + a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p3.y + (float)(u0 + u1 + u2 + u3);
    if (threadIdx.x == 0) {
        cyc[blockIdx.x] = t1 - t0;
        rt[blockIdx.x] = r1 - r0;
    }
}

template <int OP>
void run(const char *name, int per_body) {
    float *out;
    long long *cyc, *rt;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 8 * 4);
    hipMalloc(&cyc, sizeof(long long) * 256 * 8 * 4);
    hipMalloc(&rt, sizeof(long long) * 256 * 8 * 4);
    const int iters = OP >= LDSB64 ? 2000 : 8000;  // (the float64 cases share the short count)
    printf("%-10s", name);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;  // 256-thread blocks = 4 waves, one per SIMD; all resident at once
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 100, rt);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, rt);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks), hr(blocks);
        hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
        hipMemcpy(hr.data(), rt, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
        double avg = 0, clk = 0;
        for (int i = 0; i < blocks; ++i) {
            avg += (double)h[i];
            clk += (double)h[i] / (double)hr[i] * 0.1;  // GHz: memtime ticks per 100 MHz memrealtime tick
        }
        avg /= blocks;
        clk /= blocks;
        const double n_inst = (double)iters * 64 * per_body;  // per wave
        // SIMD cycles per wave-instruction: (in-kernel cycles of one wave) / (instructions of the W waves sharing its SIMD)
        printf("  W=%d: %.2f SIMD-cyc/inst (clock %.2f GHz, %.2f ms; wall-based %.2f)", wps, avg / n_inst / wps, clk, ms,
               ms * 1e-3 * clk * 1e9 / n_inst / wps);
    }
    printf("\n");
    hipFree(out);
    hipFree(cyc);
    hipFree(rt);
}

int main() {
    run<FMA>("v_fma", 4);
    run<PKFMA>("v_pk_fma", 4);
    run<CVTFLR>("cvt_flr", 4);
    run<FRACT>("v_fract", 4);
    run<NOPS>("s_nop", 4);
    run<SALU>("s_add", 4);
    run<VS>("4v+4s", 4);    // per_body counts the VALU instructions only: compare with v_fract
    run<VS2>("4v+8s", 4);
    run<VN>("4v+4nop", 4);
    run<LDSB64>("ds_b64x4", 4);
    run<LDSREAD2>("ds_rd2x2", 2);
    run<FMA64>("v_fma_f64", 4);
    run<MUL64>("v_mul_f64", 4);
    run<ADD64>("v_add_f64", 4);
    run<RCP64>("v_rcp_f64", 4);
    run<FLOOR64>("v_floor64", 4);
    run<CVTI64>("cvt_i32f64", 4);
    run<DIVSC64>("divscale64", 4);
    run<DIVFIX64>("divfixup64", 4);
    run<CNDMASK>("v_cndmask", 4);
    return 0;
}
