// Field preparation kernels: interleave (u,v), cubic B-spline prefilter, mirror pads.
//
// Replaces the seed-independent work the reference repeats inside every
// tools.xr_map_coordinates call (LCS/tools.py:12-14, and for order 3 the spline
// prefilter scipy.ndimage.map_coordinates runs on the whole field each call,
// LCS/tools.py:26-30).  Done once per wind time series here.
//
// Image layout per time level: (ny_f+3) x (nx_f+3) nodes, node = {u, v}.
// Node (y, x) of the field sits at padded position (y+1, x+1).  Pad nodes hold
// the mirrored interior value (i -> |i| reflected about 0 and n-1), which is
// the tap rule scipy applies to out-of-range spline taps.
#include <type_traits>

#include "lcs_common.h"
#include "launch_plan.h"

namespace {

__device__ __forceinline__ int mirror_index(int i, int n) {
    // scipy ni_interpolation.c tap mirroring; here |i| never exceeds n+1
    if (i < 0) i = -i;
    if (i > n - 1) i = 2 * (n - 1) - i;
    return i;
}

// interior nodes: packed[t][y+1][x+1] = {u[t][y][x], v[t][y][x]}
// TIN: the element type of the raw planes (float for LC_F64_WIND_F32: a float32 wind whose spline coefficients scipy forms
// in double -- spline_filter(output=float64) inside map_coordinates, LCS/tools.py:26-30)
template <typename T, typename TIN = T>
__global__ void pack_interior_kernel(const TIN *__restrict__ u, const TIN *__restrict__ v, T *__restrict__ packed,
                                     int ny, int nx, size_t n_nodes_total) {
    const size_t plane = (size_t)ny * nx;
    const int pitch = nx + LC_PAD;
    const size_t level = (size_t)(ny + LC_PAD) * pitch;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_nodes_total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / plane;
        const size_t r = i - t * plane;
        const int y = (int)(r / nx);
        const int x = (int)(r - (size_t)y * nx);
        const size_t o = (t * level + (size_t)(y + LC_PAD_LO) * pitch + (x + LC_PAD_LO)) * 2;
        packed[o] = (T)u[i];
        packed[o + 1] = (T)v[i];
    }
}

// pad nodes <- mirrored interior nodes (runs after the interior is final): the 3 (ny + nx + 3) pad nodes of a level,
// one thread per pad node (grid: pad chunks x levels)
template <typename T>
__global__ void __launch_bounds__(256) pads_only_kernel(T *__restrict__ packed, int nt, int ny, int nx) {
    const int pitch = nx + LC_PAD, nrowpad = LC_PAD * pitch, npad = nrowpad + LC_PAD * ny;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npad) return;
    int py, px;
    if (i < nrowpad) {  // the LC_PAD whole pad rows: padded row 0, then ny + 1, ny + 2
        const int k = i / pitch;
        px = i - k * pitch;
        py = k < LC_PAD_LO ? k : ny + k;
    } else {            // the LC_PAD pad columns of the interior rows: padded column 0, then nx + 1, nx + 2
        const int j = i - nrowpad, y = j / LC_PAD, k = j - y * LC_PAD;
        py = y + LC_PAD_LO;
        px = k < LC_PAD_LO ? k : nx + k;
    }
    const int sy = mirror_index(py - LC_PAD_LO, ny), sx = mirror_index(px - LC_PAD_LO, nx);
    const size_t level = (size_t)(ny + LC_PAD) * pitch;
    typedef T T2 __attribute__((ext_vector_type(2)));
    for (int t = blockIdx.y; t < nt; t += gridDim.y) {
        T2 *img = (T2 *)packed + (size_t)t * level;
        img[(size_t)py * pitch + px] = img[(size_t)(sy + LC_PAD_LO) * pitch + (sx + LC_PAD_LO)];
    }
}

// One line of the separable cubic B-spline prefilter, in place, mirror boundary.
// Same recursion as scipy ni_splines.c (_apply_filter_gain, _init_causal_mirror,
// _init_anticausal_mirror) for the single cubic pole z = sqrt(3) - 2.
// Arithmetic in double whatever T is; the stored coefficients are T.
template <typename T>
__device__ void prefilter_line(T *c, size_t stride, int n) {
    const double z = -0.26794919243112270647;  // sqrt(3) - 2
    const double gain = 6.0;                   // (1 - z)(1 - 1/z)
    if (n < 2) return;
    // causal initialisation: exact sum over the mirrored line
    const double zn1 = pow(z, (double)(n - 1));
    const double last = gain * (double)c[(size_t)(n - 1) * stride];
    double c0 = gain * (double)c[0] + zn1 * last;
    double zi = z;
    for (int i = 1; i < n - 1; ++i) {
        if (zi == 0.0) break;  // every remaining term is exactly zero
        c0 += zi * (gain * (double)c[(size_t)i * stride] + zn1 * gain * (double)c[(size_t)(n - 1 - i) * stride]);
        zi *= z;
    }
    c0 /= (1.0 - zn1 * zn1);
    double prev = c0;
    c[0] = (T)prev;
    for (int i = 1; i < n; ++i) {
        prev = gain * (double)c[(size_t)i * stride] + z * prev;
        c[(size_t)i * stride] = (T)prev;
    }
    // anticausal
    double cn1 = (double)c[(size_t)(n - 1) * stride];
    double cn2 = (double)c[(size_t)(n - 2) * stride];
    double next = (z * cn2 + cn1) * z / (z * z - 1.0);
    c[(size_t)(n - 1) * stride] = (T)next;
    for (int i = n - 2; i >= 0; --i) {
        next = z * (next - (double)c[(size_t)i * stride]);
        c[(size_t)i * stride] = (T)next;
    }
}

// Same recursion, marching in blocks of 8 elements whose loads are issued together: the march
// is a dependent chain per line, so memory latency (not bandwidth) is what must be overlapped.
// ``src`` / ``sstride``: where the causal pass READS the line (the raw field -- of element type TIN --, so that the
// interleave and the latitude sweep are one pass over the data).
template <typename T, typename TIN = T>
__device__ void prefilter_line_blocked(T *c, size_t stride, int n, const TIN *src, size_t sstride) {
    constexpr int B = 8;
    const double z = -0.26794919243112270647, gain = 6.0;
    if (n < 2 * B) {
        for (int i = 0; i < n; ++i) c[(size_t)i * stride] = (T)src[(size_t)i * sstride];
        prefilter_line<T>(c, stride, n);
        return;
    }
    // Causal initial value (scipy _init_causal_mirror).  For n >= 64 the mirrored terms carry
    // z^(n-1) <= 1e-36 and the direct terms beyond the 64th are below |z|^64 = 2.5e-37 of the line's
    // scale -- far under double rounding -- so they are not read (same horizon as the longitude sweep).
    constexpr int HORIZON = 64;
    const bool mirrored = n < HORIZON;
    const double zn1 = mirrored ? pow(z, (double)(n - 1)) : 0.0;
    double c0 = gain * (double)src[0];
    if (mirrored) c0 += zn1 * (gain * (double)src[(size_t)(n - 1) * sstride]);
    double zi = z;
    const int last = mirrored ? n - 1 : HORIZON;  // direct terms i = 1 .. last-1
    for (int i0 = 1; i0 < last; i0 += B) {
        double a[B], b[B];
#pragma unroll
        for (int q = 0; q < B; ++q) {
            const int i = min(i0 + q, n - 2);
            a[q] = (double)src[(size_t)i * sstride];
            b[q] = mirrored ? (double)src[(size_t)(n - 1 - i) * sstride] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < B; ++q) {
            if (i0 + q < last) {
                c0 += zi * (gain * a[q] + zn1 * gain * b[q]);
                zi *= z;
            }
        }
    }
    c0 /= (1.0 - zn1 * zn1);
    double prev = c0;
    c[0] = (T)prev;
    for (int i0 = 1; i0 < n; i0 += B) {
        double a[B];
#pragma unroll
        for (int q = 0; q < B; ++q) a[q] = (double)src[(size_t)min(i0 + q, n - 1) * sstride];
#pragma unroll
        for (int q = 0; q < B; ++q) {
            if (i0 + q < n) {
                prev = gain * a[q] + z * prev;
                c[(size_t)(i0 + q) * stride] = (T)prev;
            }
        }
    }
    const double cn1 = (double)c[(size_t)(n - 1) * stride], cn2 = (double)c[(size_t)(n - 2) * stride];
    double next = (z * cn2 + cn1) * z / (z * z - 1.0);
    c[(size_t)(n - 1) * stride] = (T)next;
    for (int i0 = n - 2; i0 >= 0; i0 -= B) {
        double a[B];
#pragma unroll
        for (int q = 0; q < B; ++q) a[q] = (double)c[(size_t)max(i0 - q, 0) * stride];
#pragma unroll
        for (int q = 0; q < B; ++q) {
            if (i0 - q >= 0) {
                next = z * (next - a[q]);
                c[(size_t)(i0 - q) * stride] = (T)next;
            }
        }
    }
}

// Orders 2, 4, 5: the same recursion with scipy's pole lists (ni_splines.c get_filter_poles); the total gain
// prod (1 - z)(1 - 1/z) is applied while the first pole's causal pass reads the line, as _apply_filter_gain does
// before any pole.  Thread per line (these orders take the generic direct advect kernel; speed is not the point).
struct PoleList {
    double z[2];
    int n;
    double gain;
};

inline PoleList spline_poles(int order) {
    PoleList p = {};
    if (order == 2) {
        p.n = 1;
        p.z[0] = sqrt(8.0) - 3.0;
    } else if (order == 3) {
        p.n = 1;
        p.z[0] = sqrt(3.0) - 2.0;
    } else if (order == 4) {
        p.n = 2;
        p.z[0] = sqrt(664.0 - sqrt(438976.0)) + sqrt(304.0) - 19.0;
        p.z[1] = sqrt(664.0 + sqrt(438976.0)) - sqrt(304.0) - 19.0;
    } else {
        p.n = 2;
        p.z[0] = sqrt(67.5 - sqrt(4436.25)) + sqrt(26.25) - 6.5;
        p.z[1] = sqrt(67.5 + sqrt(4436.25)) - sqrt(26.25) - 6.5;
    }
    p.gain = 1.0;
    for (int i = 0; i < p.n; ++i) p.gain *= (1.0 - p.z[i]) * (1.0 - 1.0 / p.z[i]);
    return p;
}

template <typename T>
__device__ void prefilter_line_poles(T *c, size_t stride, int n, const PoleList P) {
    if (n < 2) return;
    for (int k = 0; k < P.n; ++k) {
        const double z = P.z[k], gain = k == 0 ? P.gain : 1.0;
        const double zn1 = pow(z, (double)(n - 1));
        double c0 = gain * (double)c[0] + zn1 * (gain * (double)c[(size_t)(n - 1) * stride]);
        double zi = z;
        for (int i = 1; i < n - 1; ++i) {
            if (zi == 0.0) break;
            c0 += zi * (gain * (double)c[(size_t)i * stride] + zn1 * (gain * (double)c[(size_t)(n - 1 - i) * stride]));
            zi *= z;
        }
        c0 /= (1.0 - zn1 * zn1);
        double prev = c0;
        c[0] = (T)prev;
        for (int i = 1; i < n; ++i) {
            prev = gain * (double)c[(size_t)i * stride] + z * prev;
            c[(size_t)i * stride] = (T)prev;
        }
        const double cn1 = (double)c[(size_t)(n - 1) * stride], cn2 = (double)c[(size_t)(n - 2) * stride];
        double next = (z * cn2 + cn1) * z / (z * z - 1.0);
        c[(size_t)(n - 1) * stride] = (T)next;
        for (int i = n - 2; i >= 0; --i) {
            next = z * (next - (double)c[(size_t)i * stride]);
            c[(size_t)i * stride] = (T)next;
        }
    }
}

template <typename T>
__global__ void prefilter_general_kernel(T *__restrict__ packed, int nt, int ny, int nx, int axis, const PoleList P) {
    const int pitch = nx + LC_PAD;
    const size_t level = (size_t)(ny + LC_PAD) * pitch * 2;
    const size_t per_level = (size_t)(axis == 0 ? nx : ny) * 2;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= per_level * nt) return;
    const size_t t = i / per_level, r = i - t * per_level;
    const int k = (int)(r >> 1), comp = (int)(r & 1);
    T *base = packed + t * level + ((size_t)LC_PAD_LO * pitch + LC_PAD_LO) * 2 + comp;
    if (axis == 0)
        prefilter_line_poles<T>(base + (size_t)k * 2, (size_t)pitch * 2, ny, P);
    else
        prefilter_line_poles<T>(base + (size_t)k * pitch * 2, 2, nx, P);
}

// axis 0 (latitude): one thread per (level, column, component); consecutive
// threads touch consecutive elements, so every step of the march is coalesced.
// The causal pass reads the RAW field (u / v planes) and writes the interleaved image: the separate interleave
// pass (pack_interior_kernel, one write + one read of the whole image) is folded into the sweep.
template <typename T, typename TIN = T>
__global__ void prefilter_cols_kernel(const TIN *__restrict__ u, const TIN *__restrict__ v, T *__restrict__ packed, int nt, int ny,
                                      int nx) {
    const int pitch = nx + LC_PAD;
    const size_t level = (size_t)(ny + LC_PAD) * pitch * 2;
    const size_t lines = (size_t)nt * nx * 2;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= lines) return;
    const size_t t = i / ((size_t)nx * 2);
    const size_t xc = i - t * (size_t)nx * 2;  // x*2 + component
    T *c = packed + t * level + ((size_t)LC_PAD_LO * pitch + LC_PAD_LO) * 2 + xc;
    const TIN *src = ((xc & 1) ? v : u) + t * (size_t)ny * nx + (xc >> 1);
    prefilter_line_blocked<T, TIN>(c, (size_t)pitch * 2, ny, src, (size_t)nx);
}

// axis 1 (longitude): one thread per (level, row, component).
template <typename T>
__global__ void prefilter_rows_kernel(T *__restrict__ packed, int nt, int ny, int nx) {
    const int pitch = nx + LC_PAD;
    const size_t level = (size_t)(ny + LC_PAD) * pitch * 2;
    const size_t lines = (size_t)nt * ny * 2;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= lines) return;
    const size_t t = i / ((size_t)ny * 2);
    const size_t r = i - t * (size_t)ny * 2;
    const int y = (int)(r >> 1), comp = (int)(r & 1);
    T *c = packed + t * level + ((size_t)(y + LC_PAD_LO) * pitch + LC_PAD_LO) * 2 + comp;
    prefilter_line<T>(c, 2, nx);
}

// axis 1 (longitude), fast form for nx >= 64: one wave filters 32 rows x 2 components.  The
// row chunks travel through an LDS tile so that global traffic is coalesced row segments
// (the thread-per-line kernel above strides by a whole row between lanes) while each lane
// walks its own line out of LDS.  Causal sweep left->right, anticausal right->left; the
// running value crosses chunk boundaries in a register.
// The causal initial value sums the first 64 mirrored terms instead of all n-2: the dropped
// terms are below |z|^64 = 2.5e-37 of the line's scale (and z^(n-1) <= 1e-36 for n >= 64),
// far under double rounding -- indistinguishable from scipy's exact sum.
constexpr int PR_ROWS = 32;   // rows per wave
constexpr int PR_CHUNK = 64;  // nodes per chunk

template <typename T>
__global__ void __launch_bounds__(64) prefilter_rows_lds_kernel(T *__restrict__ packed, int ny, int nx) {
    __shared__ T tile[PR_ROWS][2 * PR_CHUNK + 1];
    const double z = -0.26794919243112270647, gain = 6.0;
    const int pitch = nx + LC_PAD;
    const size_t level = (size_t)(ny + LC_PAD) * pitch * 2;
    const int t = blockIdx.y;
    const int r0 = blockIdx.x * PR_ROWS;
    const int lane = threadIdx.x;
    const int row = lane >> 1, comp = lane & 1;
    const bool line_ok = r0 + row < ny;
    T *base = packed + (size_t)t * level + ((size_t)(r0 + LC_PAD_LO) * pitch + LC_PAD_LO) * 2;  // row r0, node 0
    const int nrows = min(PR_ROWS, ny - r0);

    auto load_chunk = [&](int x0, int cnt) {  // nodes [x0, x0+cnt) of every row -> tile
        // 8 rows x 2 half-rows of loads in flight before the first LDS write: the sweep is
        // latency-bound, so the global loads must overlap each other
        for (int rb = 0; rb < nrows; rb += 8) {
            T va[8], vb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int rr = min(rb + q, nrows - 1);
                const T *src = base + (size_t)rr * pitch * 2 + (size_t)x0 * 2;
                va[q] = lane < 2 * cnt ? src[lane] : T(0);
                vb[q] = lane + 64 < 2 * cnt ? src[lane + 64] : T(0);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (rb + q < nrows) {
                    tile[rb + q][lane] = va[q];
                    tile[rb + q][lane + 64] = vb[q];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto store_chunk = [&](int x0, int cnt) {
        __builtin_amdgcn_wave_barrier();
        for (int rr = 0; rr < nrows; ++rr) {
            T *dst = base + (size_t)rr * pitch * 2 + (size_t)x0 * 2;
            for (int j = lane; j < 2 * cnt; j += 64) dst[j] = tile[rr][j];
        }
        __builtin_amdgcn_wave_barrier();
    };

    // ---- causal sweep ----
    double prev = 0.0;
    for (int x0 = 0; x0 < nx; x0 += PR_CHUNK) {
        const int cnt = min(PR_CHUNK, nx - x0);
        load_chunk(x0, cnt);
        if (line_ok) {
            int i = 0;
            if (x0 == 0) {  // initial value from the first 64 terms (cnt == 64 here: nx >= 64)
                double c0 = gain * (double)tile[row][comp], zi = z;
                for (int k = 1; k < PR_CHUNK; ++k) {
                    c0 += zi * (gain * (double)tile[row][2 * k + comp]);
                    zi *= z;
                }
                prev = c0;
                tile[row][comp] = (T)prev;
                i = 1;
            }
            for (; i < cnt; ++i) {
                prev = gain * (double)tile[row][2 * i + comp] + z * prev;
                tile[row][2 * i + comp] = (T)prev;
            }
        }
        store_chunk(x0, cnt);
    }
    // ---- anticausal sweep: chunks aligned to the END of the line ----
    double next = 0.0;
    for (int xe = nx; xe > 0; xe -= PR_CHUNK) {
        const int x0 = max(0, xe - PR_CHUNK), cnt = xe - x0;
        load_chunk(x0, cnt);
        if (line_ok) {
            int i = cnt - 1;
            if (xe == nx) {  // last chunk holds n-1 and n-2 (cnt >= 2)
                const double cn1 = (double)tile[row][2 * (cnt - 1) + comp], cn2 = (double)tile[row][2 * (cnt - 2) + comp];
                next = (z * cn2 + cn1) * z / (z * z - 1.0);
                tile[row][2 * (cnt - 1) + comp] = (T)next;
                i = cnt - 2;
            }
            for (; i >= 0; --i) {
                next = z * (next - (double)tile[row][2 * i + comp]);
                tile[row][2 * i + comp] = (T)next;
            }
        }
        store_chunk(x0, cnt);
    }
}

// ======================================================================================
// float64, order 3: each sweep as ONE read and ONE write of the image (the two kernels above read and write it
// twice: causal march, then anticausal march over what the first wrote; config 2: 3.4 + 3.2 ms for 27 GB).
// The anticausal recursion c[i] = z (c[i+1] - c+[i]) forgets its start value by |z| per step, |z|^32 = 5e-19: started
// LOOKAHEAD = 32 nodes further on from the mirror formula applied there (an O(|z|) guess), it reaches the nodes that
// are kept with an error below 5e-19 of the line's scale -- under a quarter of a double's last bit, the same argument
// as the 64-term horizon of the causal start value.  So a line is filtered in one streaming pass: the causal march runs
// 32 nodes ahead of the nodes being finished and only finished coefficients are written.  The last window of a line
// starts from scipy's exact mirror formula at n-1 (bit-identical tail).  LCS_FIR_PREFILTER=0: the two-march kernels.
// ======================================================================================
constexpr int PS_C = 16, PS_H = 32, PS_W = PS_C + PS_H;  // latitude sweep: nodes finished per round, lookahead, register window

template <typename TIN>
__global__ void __launch_bounds__(256) prefilter_cols_stream_kernel(const TIN *__restrict__ u, const TIN *__restrict__ v,
                                                                    double *__restrict__ packed, int nt, int ny, int nx) {
    const double z = -0.26794919243112270647, gain = 6.0, zend = z / (z * z - 1.0);
    const int pitch = nx + LC_PAD;
    const size_t level = (size_t)(ny + LC_PAD) * pitch * 2;
    const size_t lines = (size_t)nt * nx * 2;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= lines) return;
    const size_t t = i / ((size_t)nx * 2);
    const size_t xc = i - t * (size_t)nx * 2;  // x*2 + component
    double *c = packed + t * level + ((size_t)LC_PAD_LO * pitch + LC_PAD_LO) * 2 + xc;
    const size_t cs = (size_t)pitch * 2, ss = (size_t)nx;
    const TIN *src = ((xc & 1) ? v : u) + t * (size_t)ny * nx + (xc >> 1);
    const int n = ny;  // >= 64 (the launcher checks)
    // causal start value: the first 64 terms (prefilter_line_blocked's horizon), loads 8 deep
    double c0 = gain * src[0], zi = z;
    for (int i0 = 1; i0 < 64; i0 += 8) {
        double a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = src[(size_t)min(i0 + q, 63) * ss];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (i0 + q < 64) {
                c0 += zi * (gain * a[q]);
                zi *= z;
            }
        }
    }
    double r[PS_W];  // causal values of nodes s .. s + PS_W - 1
    r[0] = c0;
    {
        double prev = c0;
#pragma unroll
        for (int j0 = 1; j0 < PS_W; j0 += 16) {
            double a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = src[(size_t)min(j0 + q, PS_W - 1) * ss];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (j0 + q < PS_W) {
                    prev = gain * a[q] + z * prev;
                    r[j0 + q] = prev;
                }
            }
        }
    }
    for (int s = 0;; s += PS_C) {
        const int rem = n - s;
        if (rem <= PS_W) {  // the window reaches the end of the line: scipy's start value at n-1, every node kept
            const int jl = rem - 1;  // >= PS_H
            double next = 0.0;
#pragma unroll
            for (int j = PS_W - 1; j >= 0; --j) {
                if (j == jl)
                    next = (z * r[j > 0 ? j - 1 : 0] + r[j]) * zend;
                else
                    next = z * (next - r[j]);
                if (j <= jl) c[(size_t)(s + j) * cs] = next;
            }
            return;
        }
        // the next PS_C inputs, in flight during the anticausal walk
        double a[PS_C];
#pragma unroll
        for (int q = 0; q < PS_C; ++q) a[q] = src[(size_t)min(s + PS_W + q, n - 1) * ss];
        double next = (z * r[PS_W - 2] + r[PS_W - 1]) * zend;
#pragma unroll
        for (int j = PS_W - 2; j >= PS_C; --j) next = z * (next - r[j]);
#pragma unroll
        for (int j = PS_C - 1; j >= 0; --j) {
            next = z * (next - r[j]);
            c[(size_t)(s + j) * cs] = next;
        }
        double prev = r[PS_W - 1];
#pragma unroll
        for (int j = 0; j < PS_H; ++j) r[j] = r[j + PS_C];
#pragma unroll
        for (int q = 0; q < PS_C; ++q) {  // (values past n-1 repeat the last input: never read, the last window stops at jl)
            prev = gain * a[q] + z * prev;
            r[PS_H + q] = prev;
        }
    }
}

// Both sweeps in ONE pass over the raw planes (round 5; float64 order 3 moved 13.6 GB in the two sweeps above, this one
// 7.7; BASELINE configs[1]: 3.23 -> 1.70 ms, profiles/r05/c2_o3_fused_prefilter_ab.txt): the latitude march of prefilter_cols_stream_kernel, a lane per (column, component) walking down its column, and --
// instead of storing the finished latitude values for a second kernel to transpose through LDS -- the longitude recursion
// ACROSS THE LANES of the workgroup, on the FS_C rows a round finishes.  A first-order recursion s[i] = a[i] + z s[i-1] is a
// scan.  Within a DPP row (16 lanes = 8 nodes x 2 components) three doubling steps s[i] += z^(2^k) s[i - 2^k] by row_shr
// (lanes without a source read 0: no select); the rows then publish their last value E in LDS and every lane adds
// z^(l+1) (E[R-1] + z^8 E[R-2] + z^16 E[R-3] + z^24 E[R-4]) of the four rows before its own (l: its node within the row):
// at least 32 predecessors in every sum, the same |z|^32 = 5e-19 horizon the anticausal walks above start from.  The
// anticausal pass is the same mirrored (row_shl, the rows' first values, the four rows after).  No ds_bpermute: the first
// form of this kernel did all five doubling steps of a 64-lane scan with it and was bound by the LDS crossbar (3.1 ms).
// The first and the last wave of a workgroup are halo (their columns are marched and their row ends feed the neighbours,
// nothing of theirs is kept); at the ends of a line the halo columns are the mirrored ones (x -> -x, x -> 2 (nx - 1) - x),
// which IS scipy's mirror boundary: its causal start value and its closed form at n - 1 are the infinite mirrored sums
// these truncate at |z|^32.  Results differ from the two-sweep kernels in the last bits only (summation order).  A
// workgroup is 4..12 waves (the launcher picks what marches the fewest halo columns for this nx:
// lcplan::fused_prefilter_waves); blocks are dealt so that the workgroups of one level run on the same XCD (their halo
// columns are each other's interior: L2 hits).  FS_C = 8 rows per round: 168 registers, three waves per SIMD (16 rows: 228).
constexpr int FS_MAXW = 12;
constexpr int FS_C = 8, FS_H = 32, FS_W = FS_C + FS_H;
constexpr int FS_ROWS = FS_MAXW * 4 + 8;  // DPP rows of a workgroup + 4 rows of zeros on either side

template <int CTRL>
__device__ __forceinline__ double row_shift64(double v) {  // DPP row_shr:n (0x110 + n) / row_shl:n (0x100 + n); no source lane: 0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void workgroup_barrier_lds() {  // LDS traffic only: the global loads in flight stay in flight
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The longitude filter of rows 0 .. FS_C - 1 of the register window, in place.  Every thread of the workgroup calls it
// (two workgroup barriers).  ends[0] / ends[1]: the rows' last causal / first anticausal values, [4 + DPP row][q][component].
template <int P>
__device__ __forceinline__ void lanes_prefilter(double (&r)[FS_W], double (&ends)[2][FS_ROWS][FS_C][2], const double (&zp)[2][8], int tid) {
    constexpr int B = FS_C * P;  // the window's first row sits at r[B] in phase P (fused_round)
    const double z = -0.26794919243112270647, gain = 6.0;
    const double z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    const int R = 4 + (tid >> 4), l = (tid & 15) >> 1, comp = tid & 1;  // zp[0][l] = z^(l + 1), zp[1][l] = z^(8 - l): read when used (registers)
    // causal: s[i] = gain a[i] + z s[i-1]
#pragma unroll
    for (int q = 0; q < FS_C; ++q) {
        double s = gain * r[B + q];
        s = fma(z, row_shift64<0x112>(s), s);
        s = fma(z2, row_shift64<0x114>(s), s);
        s = fma(z4, row_shift64<0x118>(s), s);
        r[B + q] = s;
    }
    if (l == 7) {
#pragma unroll
        for (int q = 0; q < FS_C; ++q) ends[0][R][q][comp] = r[B + q];
    }
    workgroup_barrier_lds();
    // ... the four rows before, then the anticausal input b[i] = -z c+[i]: c[i] = b[i] + z c[i+1]
    const double zp_in = zp[0][l];
#pragma unroll
    for (int q = 0; q < FS_C; ++q) {
        const double t = fma(z8, fma(z8, fma(z8, ends[0][R - 4][q][comp], ends[0][R - 3][q][comp]), ends[0][R - 2][q][comp]), ends[0][R - 1][q][comp]);
        double s = -z * fma(zp_in, t, r[B + q]);
        s = fma(z, row_shift64<0x102>(s), s);
        s = fma(z2, row_shift64<0x104>(s), s);
        s = fma(z4, row_shift64<0x108>(s), s);
        r[B + q] = s;
        if (q % 4 == 3) __builtin_amdgcn_sched_barrier(0);  // four rows' LDS reads in flight, not all FS_C (registers: 6 spilled otherwise)
    }
    if (l == 0) {
#pragma unroll
        for (int q = 0; q < FS_C; ++q) ends[1][R][q][comp] = r[B + q];
    }
    workgroup_barrier_lds();
    const double zp_out = zp[1][l];
#pragma unroll
    for (int q = 0; q < FS_C; ++q) {
        const double t = fma(z8, fma(z8, fma(z8, ends[1][R + 4][q][comp], ends[1][R + 3][q][comp]), ends[1][R + 2][q][comp]), ends[1][R + 1][q][comp]);
        r[B + q] = fma(zp_out, t, r[B + q]);
        if (q % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
}

// The causal value of row b from the 64 rows up to and including it, c+[b] = sum_{k < 64} z^k gain a[b - k] (rows before the
// first: mirrored, scipy's start value at b = 0; rows past the last: the mirror extension the march runs on), by Horner from
// the far end: |z|^64 = 2.5e-37 of what lies beyond.  ONE function for the start of a row piece and for the restart every
// march makes at every multiple of FUSED_PIECE_ALIGN rows: a level's coefficients are the same bits whether this launch cut
// the level there or not.
template <typename TIN>
__device__ __forceinline__ double fused_causal_restart(const TIN *__restrict__ src, int b, int n, size_t ss) {
    constexpr double z = -0.26794919243112270647, gain = 6.0;
    auto row = [&](int i) {
        i = abs(i);
        return (size_t)(i > n - 1 ? 2 * (n - 1) - i : i) * ss;
    };
    double acc = 0.0;
    for (int i0 = b - 63; i0 <= b; i0 += 8) {
        double a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = src[row(i0 + q)];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = __builtin_fma(z, acc, gain * a[q]);
    }
    return acc;
}

// One round of the fused kernel in phase P: the window (FS_W rows, first row s) sits at r[(j + FS_C P) % FS_W] -- the
// round loop is unrolled over the FS_W / FS_C phases, so the window never moves (a shift by FS_C rows per round is 2 FS_H
// register moves and, worse, doubles the window's live range at the point where the finished rows are still needed).
template <int P, typename TIN>
__device__ __forceinline__ void fused_round(double (&r)[FS_W], double (&ends)[2][FS_ROWS][FS_C][2], const double (&zp)[2][8], const TIN *__restrict__ src,
                                            double *__restrict__ c, int s, int y1, int n, size_t ss, size_t cs, int tid) {
    constexpr double z = -0.26794919243112270647, gain = 6.0, zend = z / (z * z - 1.0);
    constexpr int B = FS_C * P;
    auto W = [&](int j) -> double & { return r[(B + j) % FS_W]; };
    // (the row index reaches the address arithmetic through an opaque scalar: left to itself, loop strength reduction keeps one
    // 64-bit per-lane pointer for every row of the window and of the store list across the rounds -- 80 registers, spilled)
    asm volatile("" : "+s"(s));
    double a[FS_C];  // the next FS_C inputs, in flight during the anticausal walk and the lane scans
#pragma unroll
    for (int q = 0; q < FS_C; ++q) {
        const int i = s + FS_W + q;
        a[q] = src[(size_t)(i > n - 1 ? 2 * (n - 1) - i : i) * ss];
    }
    double prev = W(FS_W - 1);
    // (the lookahead stretch in closed form -- four Horner chains in z^4, 8 dependent operations instead of 62 -- was tried:
    // the register allocator answers with 90 spilled registers, whatever the scheduling barriers)
    double next = (z * W(FS_W - 2) + W(FS_W - 1)) * zend;
#pragma unroll
    for (int j = FS_W - 2; j >= FS_C; --j) next = z * (next - W(j));
#pragma unroll
    for (int j = FS_C - 1; j >= 0; --j) {
        next = z * (next - W(j));
        W(j) = next;
    }
    lanes_prefilter<P>(r, ends, zp, tid);
#pragma unroll
    for (int j = 0; j < FS_C; ++j)
        if (s + j < y1) c[(size_t)(s + j) * cs] = W(j);
    // row s + FS_W a multiple of FUSED_PIECE_ALIGN (uniform; s is a multiple of FS_C, so only the first of the new rows can
    // be one): the march restarts there (fused_causal_restart: the finished rows' registers are free again by now)
    if (((s + FS_W) & (lcplan::FUSED_PIECE_ALIGN - 1)) == 0) {
        prev = fused_causal_restart(src, s + FS_W, n, ss);
        W(0) = prev;
#pragma unroll
        for (int q = 1; q < FS_C; ++q) {
            prev = __builtin_fma(z, prev, gain * a[q]);
            W(q) = prev;
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < FS_C; ++q) {  // rows s + FS_W + q take the places of the rows just stored
        prev = __builtin_fma(z, prev, gain * a[q]);
        W(q) = prev;
    }
}

template <typename TIN>
__global__ void __launch_bounds__(FS_MAXW * 64) prefilter_fused_stream_kernel(const TIN *__restrict__ u, const TIN *__restrict__ v,
                                                                              double *__restrict__ packed, int nt, int ny, int nx, int nxb,
                                                                              const lcplan::FusedSplit split) {
    __shared__ double ends[2][FS_ROWS][FS_C][2];
    __shared__ double zp[2][8];
    for (int i = threadIdx.x; i < 2 * FS_ROWS * FS_C * 2; i += blockDim.x) (&ends[0][0][0][0])[i] = 0.0;  // (the rows beside the workgroup's stay 0)
    const double z = -0.26794919243112270647, gain = 6.0;
    const int nw = blockDim.x >> 6, xout = 32 * (nw - 2);
    // the 8 XCDs take consecutive runs of (level, x block) items: blocks i, i + 8, ... (one XCD) are neighbours in x
    // (the items past split.n_whole come in split.pieces row pieces each: lcplan::fused_prefilter_split; every XCD takes an
    // eighth of the whole items and an eighth of the pieces)
    const unsigned items = (unsigned)nxb * nt, n_pieces = (items - split.n_whole) * split.pieces;
    const unsigned whole_per = (split.n_whole + 7) / 8, piece_per = (n_pieces + 7) / 8;
    const unsigned xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    unsigned item;
    int piece = 0;
    if (j < whole_per) {
        item = xcd * whole_per + j;
        if (item >= (unsigned)split.n_whole) return;
    } else {
        const unsigned q = xcd * piece_per + (j - whole_per);
        if (q >= n_pieces) return;
        item = split.n_whole + q / split.pieces;
        piece = q % split.pieces;
    }
    const int y0 = piece * split.piece_rows;                                                         // this workgroup's rows [y0, y1)
    const int y1 = item < (unsigned)split.n_whole ? ny : min(ny, y0 + split.piece_rows);
    const int t = item / nxb, xb = item - t * nxb;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, comp = lane & 1;
    const int x = xb * xout - 32 + (tid >> 1);  // this lane's column; outside [0, nx): the mirrored one (beyond that: unused)
    int xm = x < 0 ? -x : x;
    xm = xm > nx - 1 ? 2 * (nx - 1) - xm : xm;
    xm = min(max(xm, 0), nx - 1);
    const bool keep = w >= 1 && w <= nw - 2 && x < nx;
    if (threadIdx.x < 8) {  // z^(l + 1), z^(8 - l) for the node l of a DPP row
        double p = z;
        for (int i = 0; i < (int)threadIdx.x; ++i) p *= z;
        zp[0][threadIdx.x] = p;
        zp[1][7 - threadIdx.x] = p;
    }
    __syncthreads();  // the zeroing of `ends` above strides over all threads: done before any wave stores its DPP rows' end values

    const int pitch = nx + LC_PAD;
    const size_t level = (size_t)(ny + LC_PAD) * pitch * 2;
    // a lane that keeps nothing (halo waves, columns past nx) stores like the others, into the pad column of the same rows: the
    // pads are written after this kernel (pads_ext_kernel / pads_only_kernel)
    double *c = packed + (size_t)t * level + ((size_t)LC_PAD_LO * pitch + (keep ? LC_PAD_LO + x : 0)) * 2 + comp;
    const size_t cs = (size_t)pitch * 2, ss = (size_t)nx;
    const TIN *src = (comp ? v : u) + (size_t)t * ny * nx + xm;
    const int n = ny;  // >= 64 (the launcher checks)
    // the latitude march: prefilter_cols_stream_kernel's, with the finished values kept in the window instead of stored.  The
    // causal value at the first row y0 is the sum of the 64 rows above it, c+[y0] = sum_k z^k gain a[|y0 - k|]: scipy's mirror
    // start value at y0 = 0 (prefilter_line_blocked's horizon), and the restart of a row piece anywhere else (|z|^64 = 2.5e-37)
    auto row = [&](int i) { return (size_t)(i > n - 1 ? 2 * (n - 1) - i : i) * ss; };   // rows past the end: mirrored
    const double c0 = fused_causal_restart(src, y0, n, ss);   // (y0 = 0 or a multiple of FUSED_PIECE_ALIGN: where every march restarts)
    double r[FS_W];
    r[0] = c0;
    {
        double prev = c0;
#pragma unroll
        for (int j0 = 1; j0 < FS_W; j0 += 16) {
            double a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = src[row(y0 + min(j0 + q, FS_W - 1))];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (j0 + q < FS_W) {
                    prev = __builtin_fma(z, prev, gain * a[q]);
                    r[j0 + q] = prev;
                }
            }
        }
    }
    // Past the end of the column the march goes on over the mirrored rows (i -> 2 (n - 1) - i): the anticausal walk of the
    // last rounds then starts 32 rows beyond n - 1 on scipy's mirror extension, which its closed form at n - 1 sums exactly
    // (the same |z|^32 horizon as everywhere else), and every round is the same code.
    static_assert(FS_W % FS_C == 0 && FS_W / FS_C == 5, "the round loop is unrolled over five phases");
    for (int s = y0; s < y1;) {   // (y0, y1 are the same for every thread: the barriers inside are reached by all or none)
        fused_round<0>(r, ends, zp, src, c, s, y1, n, ss, cs, tid);
        if ((s += FS_C) >= y1) break;
        fused_round<1>(r, ends, zp, src, c, s, y1, n, ss, cs, tid);
        if ((s += FS_C) >= y1) break;
        fused_round<2>(r, ends, zp, src, c, s, y1, n, ss, cs, tid);
        if ((s += FS_C) >= y1) break;
        fused_round<3>(r, ends, zp, src, c, s, y1, n, ss, cs, tid);
        if ((s += FS_C) >= y1) break;
        fused_round<4>(r, ends, zp, src, c, s, y1, n, ss, cs, tid);
        s += FS_C;
    }
}

// Longitude sweep, same scheme on the wave-per-32-rows LDS tile of prefilter_rows_lds_kernel: the tile is a ring of
// 64 nodes (two chunks of 32); a lane walks its line (row, component) through it.  Round k: the ring holds the causal
// values of chunks k and k+1; the lane starts the anticausal walk at the end of chunk k+1, walks chunk k+1 without
// writing, finishes chunk k in place; the wave stores chunk k as whole 512-byte row segments, drops chunk k+2 (loaded
// into registers before the walk, so its latency is behind it) into the freed half and runs the causal march over it.
constexpr int RS_ROWS = 32, RS_C = 32, RS_RING = 2 * RS_C;

__global__ void __launch_bounds__(64) prefilter_rows_stream_kernel(double *__restrict__ packed, int ny, int nx) {
    __shared__ double ring[RS_ROWS][2 * RS_RING + 1];
    const double z = -0.26794919243112270647, gain = 6.0, zend = z / (z * z - 1.0);
    const int pitch = nx + LC_PAD;
    const size_t level = (size_t)(ny + LC_PAD) * pitch * 2;
    const int t = blockIdx.y;
    const int r0 = blockIdx.x * RS_ROWS;
    const int lane = threadIdx.x;
    const int row = lane >> 1, comp = lane & 1;
    const bool line_ok = r0 + row < ny;
    double *base = packed + (size_t)t * level + ((size_t)(r0 + LC_PAD_LO) * pitch + LC_PAD_LO) * 2;  // row r0, node 0
    const int nrows = min(RS_ROWS, ny - r0);
    double *mine = &ring[row][comp];  // node i of this lane's line: mine[2 * (i & 63)]

    auto chunk_cnt = [&](int k) { return min(RS_C, nx - k * RS_C); };
    auto fetch = [&](int k, double (&reg)[RS_ROWS]) {  // chunk k of every row -> registers (one 512-byte segment per row)
        const int cnt = chunk_cnt(k);
#pragma unroll
        for (int q = 0; q < RS_ROWS; ++q) {
            const int rr = min(q, nrows - 1);
            reg[q] = lane < 2 * cnt ? base[(size_t)rr * pitch * 2 + (size_t)k * RS_C * 2 + lane] : 0.0;
        }
    };
    auto drop = [&](int k, const double (&reg)[RS_ROWS]) {
        const int h = (k & 1) * RS_C * 2;
#pragma unroll
        for (int q = 0; q < RS_ROWS; ++q) ring[q][h + lane] = reg[q];
        __builtin_amdgcn_wave_barrier();
    };
    auto store = [&](int k) {
        __builtin_amdgcn_wave_barrier();
        const int cnt = chunk_cnt(k), h = (k & 1) * RS_C * 2;
        if (lane < 2 * cnt)
            for (int rr = 0; rr < nrows; ++rr) base[(size_t)rr * pitch * 2 + (size_t)k * RS_C * 2 + lane] = ring[rr][h + lane];
        __builtin_amdgcn_wave_barrier();
    };
    double prev = 0.0;
    auto causal = [&](int k, int first) {  // march over chunk k from its node ``first``
        const int cnt = chunk_cnt(k);
        double *p = mine + (k & 1) * RS_C * 2;
#pragma unroll 8
        for (int i = first; i < cnt; ++i) {
            prev = gain * p[2 * i] + z * prev;
            p[2 * i] = prev;
        }
    };

    double reg[RS_ROWS];
    fetch(0, reg);
    drop(0, reg);
    fetch(1, reg);
    drop(1, reg);  // nx >= 64: chunks 0 and 1 are whole
    if (line_ok) {
        double c0 = gain * mine[0], zi = z;  // start value from the first 64 terms
#pragma unroll 8
        for (int k = 1; k < RS_RING; ++k) {
            c0 += zi * (gain * mine[2 * k]);
            zi *= z;
        }
        prev = c0;
        mine[0] = prev;
        causal(0, 1);
        causal(1, 0);
    }
    const int nch = (nx + RS_C - 1) / RS_C;
    for (int k = 0;; ++k) {
        const bool last = k + 2 >= nch;  // the ring reaches the end of the line
        if (!last) fetch(k + 2, reg);
        if (line_ok) {
            if (last) {  // scipy's start value at nx-1; chunks k and k+1 are both finished
                constexpr int M = RS_RING - 1;
                double next = (z * mine[2 * ((nx - 2) & M)] + mine[2 * ((nx - 1) & M)]) * zend;
                mine[2 * ((nx - 1) & M)] = next;
#pragma unroll 8
                for (int i = nx - 2; i >= k * RS_C; --i) {
                    next = z * (next - mine[2 * (i & M)]);
                    mine[2 * (i & M)] = next;
                }
            } else {
                const double *pa = mine + ((k + 1) & 1) * RS_C * 2;
                double next = (z * pa[2 * (RS_C - 2)] + pa[2 * (RS_C - 1)]) * zend;
#pragma unroll 8
                for (int j = RS_C - 2; j >= 0; --j) next = z * (next - pa[2 * j]);
                double *p = mine + (k & 1) * RS_C * 2;
#pragma unroll 8
                for (int j = RS_C - 1; j >= 0; --j) {
                    next = z * (next - p[2 * j]);
                    p[2 * j] = next;
                }
            }
        }
        store(k);
        if (last) {
            store(k + 1);
            return;
        }
        drop(k + 2, reg);
        if (line_ok) causal(k + 2, 0);
    }
}

// Order 1 in one pass over the PADDED image: every node (pads included) reads its mirrored
// source once and writes lin[t] and, when asked, ext[t] = 2 F[t] - F[t+1].
// A block takes 256 consecutive nodes of one padded row of one level (grid: x chunks, padded rows, levels): no
// integer division anywhere -- the first form, one flat index per node decomposed with two 64-bit divisions, was bound
// by that arithmetic, not by memory (C3: 0.60 ms for 3.4 GB; config 2 in float64 with both images: 2.5 ms for 10 GB).
// Non-temporal stores of the images (nothing reads them again before the advect kernel streams them): config 2's float64
// pack 2.07 -> 1.95 ms, C3's 0.54 -> 0.50 (profiles/r03).  -DLCS_PACK_PLAIN: plain stores (A/B).
#ifdef LCS_PACK_PLAIN
#define LCS_PACK_STORE(ptr, val) (*(ptr) = (val))
#else
#define LCS_PACK_STORE(ptr, val) __builtin_nontemporal_store(val, ptr)
#endif
// A thread walks PACK_LV consecutive time levels of its node and carries level t+1 forward, so every raw level is read
// ONCE (plus one level in PACK_LV at the chunk's end) instead of twice -- as F[t] for lin[t] / ext[t] and again as F[t+1]
// for ext[t-1]: float64 config 2 moved 13.9 GB for 10.1 GB compulsory that way (profiles/r03/c2_pmc_traffic.json).  The
// loads of a chunk are issued together (static trip count), then the stores.
// lin == NULL: the fused-level image alone (lc_field_pack(order 1, packed_dev = NULL): the caller samples order 1 from
// the raw planes, lc_advect_ex).
#ifndef LCS_PACK_LV
#define LCS_PACK_LV 2
#endif
constexpr int PACK_LV = LCS_PACK_LV;
template <typename T>
__global__ void __launch_bounds__(256) pack_fused_kernel(const T *__restrict__ u, const T *__restrict__ v, T *__restrict__ lin,
                                                         T *__restrict__ ext, int nt, int ny, int nx) {
    const int pitch = nx + LC_PAD;
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= pitch) return;
    const size_t plane = (size_t)ny * nx, level = (size_t)(ny + LC_PAD) * pitch;
    const int sx = mirror_index(px - LC_PAD_LO, nx);
    typedef T T2 __attribute__((ext_vector_type(2)));
    for (int py = blockIdx.y; py < ny + LC_PAD; py += gridDim.y) {     // (the grid covers every row and level chunk unless
        const int sy = mirror_index(py - LC_PAD_LO, ny);               //  a dimension exceeds 65535 blocks)
        const size_t so = (size_t)sy * nx + sx, po = (size_t)py * pitch + px;
        for (int t0 = blockIdx.z * PACK_LV; t0 < nt; t0 += gridDim.z * PACK_LV) {
            T a[PACK_LV + 1], b[PACK_LV + 1];
#pragma unroll
            for (int q = 0; q <= PACK_LV; ++q) {
                const size_t t = (size_t)min(t0 + q, nt - 1);          // (past the series: the last level again, never used)
                a[q] = u[t * plane + so];
                b[q] = v[t * plane + so];
            }
#pragma unroll
            for (int q = 0; q < PACK_LV; ++q) {
                const int t = t0 + q;
                if (t >= nt) break;
                if (lin) LCS_PACK_STORE((T2 *)(lin + ((size_t)t * level + po) * 2), ((T2){a[q], b[q]}));
                if (ext && t + 1 < nt)
                    LCS_PACK_STORE((T2 *)(ext + ((size_t)t * level + po) * 2), ((T2){T(2) * a[q] - a[q + 1], T(2) * b[q] - b[q + 1]}));
            }
        }
    }
}

// Pads of img and the whole of ext = 2*img[t] - img[t+1] in one pass over the padded levels: every padded node
// reads its mirrored interior source at levels t and t+1 (interior nodes are final by now), writes its own pad
// of img[t] if it is one, and its node of ext[t].  (Without ext: pads_only_kernel.)  Level t+1 is
// carried forward over PACK_LV levels as in pack_fused_kernel: each level of the image is read once (float64 config 2 at
// order 3: 10.4 -> 7.0 GB moved).
template <typename T>
__global__ void __launch_bounds__(256) pads_ext_kernel(T *__restrict__ img, T *__restrict__ ext, int nt, int ny, int nx) {
    // same decomposition as pack_fused_kernel: 256 nodes of one padded row per block, no integer division
    const int pitch = nx + LC_PAD;
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= pitch) return;
    const size_t level = (size_t)(ny + LC_PAD) * pitch;
    const int x = px - LC_PAD_LO, sx = mirror_index(x, nx);
    typedef T T2 __attribute__((ext_vector_type(2)));
    for (int py = blockIdx.y; py < ny + LC_PAD; py += gridDim.y) {
        const int y = py - LC_PAD_LO, sy = mirror_index(y, ny);
        const bool pad = !(y >= 0 && y < ny && x >= 0 && x < nx);
        const size_t so = ((size_t)(sy + LC_PAD_LO) * pitch + (sx + LC_PAD_LO)) * 2, po = ((size_t)py * pitch + px) * 2;
        for (int t0 = blockIdx.z * PACK_LV; t0 < nt; t0 += gridDim.z * PACK_LV) {
            T2 c[PACK_LV + 1];
#pragma unroll
            for (int q = 0; q <= PACK_LV; ++q) c[q] = *(const T2 *)(img + (size_t)min(t0 + q, nt - 1) * level * 2 + so);
#pragma unroll
            for (int q = 0; q < PACK_LV; ++q) {
                const int t = t0 + q;
                if (t >= nt) break;
                if (pad) *(T2 *)(img + (size_t)t * level * 2 + po) = c[q];
                if (t + 1 < nt) LCS_PACK_STORE((T2 *)(ext + (size_t)t * level * 2 + po), T(2) * c[q] - c[q + 1]);
            }
        }
    }
}

// ext[t] = 2*img[t] - img[t+1] over whole padded levels (linear, so pads stay mirrored)
template <typename T>
__global__ void extrapolate_kernel(const T *__restrict__ img, T *__restrict__ ext, size_t level_elems, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        ext[i] = T(2) * img[i] - img[i + level_elems];
}


// ======================================================================================
// float32, order 3: the whole cubic prefilter as ONE pass over the raw field.
// The recursive filter of scipy (pole z = sqrt(3) - 2 on a mirror-extended line) is the convolution with
//   h[n] = -6 z / (1 - z^2) * z^|n|,
// and |z|^15 = 2.6e-9: cut at |n| <= 14 the filter differs from the recursion by 1.2e-8 of the line's scale, a tenth
// of a float32 ulp -- so a workgroup can filter a 32 x 32-node tile from its own 60 x 60 mirrored neighbourhood,
// both axes, without the two whole-image sweeps (2.5 -> 0.9 ms for the 97-level 720 x 1440 series).  kind 0 writes
// img[t] = P(F[t]); kind 1 writes the fused-level image ext[t] = P(2 F[t] - F[t+1]) (P is linear; the float32
// kernels read ext as given, all of them the same one).  Pads (mirrored coefficients) are written by the tile that
// owns their source node.  float64 keeps the exact recursion (prefilter_cols / prefilter_rows kernels).
// ======================================================================================
typedef float pf2 __attribute__((ext_vector_type(2)));
constexpr int FT = 32, FHALO = 14, FR = FT + 2 * FHALO, FRUN = 8, FWIN = FRUN + 2 * FHALO;
struct FirTaps {
    pf2 h[FHALO + 1];  // {h[n], h[n]}: an SGPR pair is a packed operand as it is
};

__device__ __forceinline__ int reflect_clamped(int i, int n) {
    if (i < 0) i = -i;
    if (i > n - 1) i = 2 * (n - 1) - i;
    return min(max(i, 0), n - 1);  // rows / columns of a ragged last tile beyond one reflection feed no kept output
}

template <int RUN>
__device__ __forceinline__ void fir_run(const pf2 (&win)[RUN + 2 * FHALO], const FirTaps &taps, pf2 (&acc)[RUN]) {
#pragma unroll
    for (int o = 0; o < RUN; ++o) {
        pf2 a = taps.h[0] * win[o + FHALO];
#pragma unroll
        for (int k = 1; k <= FHALO; ++k)  // symmetric pairs, nearest last: the small terms are summed first
            a = __builtin_elementwise_fma(taps.h[FHALO + 1 - k], win[o + FHALO - (FHALO + 1 - k)] + win[o + FHALO + (FHALO + 1 - k)], a);
        acc[o] = a;
    }
}

__global__ void __launch_bounds__(256) prefilter_fir_kernel(const float *__restrict__ u, const float *__restrict__ v,
                                                            float *__restrict__ img, float *__restrict__ ext, int nt, int ny,
                                                            int nx, const FirTaps taps) {
#pragma clang fp contract(off)
    __shared__ pf2 R[FR][FR + 1];   // raw neighbourhood; rows FHALO.. are overwritten by the latitude pass, then the
                                    // tile's corner by the results (29 KB: five workgroups per CU)
    const int nk = ext ? 2 : 1;
    const int t = blockIdx.z / nk, kind = blockIdx.z - t * nk;
    if (kind == 1 && t + 1 >= nt) return;
    const int gy0 = blockIdx.y * FT, gx0 = blockIdx.x * FT;
    const size_t plane = (size_t)ny * nx;
    const float *ut = u + (size_t)t * plane, *vt = v + (size_t)t * plane;
    // all loads of the neighbourhood are issued before the first LDS store (the trip count is static)
    constexpr int NLOAD = (FR * FR + 255) / 256;
    pf2 ld[NLOAD], ld2[NLOAD];
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
        const int i = min((int)threadIdx.x + it * 256, FR * FR - 1);
        const int ry = i / FR, rx = i - ry * FR;
        const size_t src = (size_t)reflect_clamped(gy0 - FHALO + ry, ny) * nx + reflect_clamped(gx0 - FHALO + rx, nx);
        ld[it] = (pf2){ut[src], vt[src]};
        if (kind) ld2[it] = (pf2){ut[src + plane], vt[src + plane]};
    }
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < FR * FR) {
            const int ry = i / FR, rx = i - ry * FR;
            R[ry][rx] = kind ? 2.0f * ld[it] - ld2[it] : ld[it];
        }
    }
    __syncthreads();
    // latitude (axis 0 first, as scipy): 60 columns x 4 runs of 8 rows, in place (rows FHALO .. FHALO + FT - 1)
    {
        const bool on = threadIdx.x < FR * (FT / FRUN);
        const int c = threadIdx.x % FR, r0 = min((int)threadIdx.x / FR, FT / FRUN - 1) * FRUN;
        pf2 win[FWIN], acc[FRUN];
#pragma unroll
        for (int j = 0; j < FWIN; ++j) win[j] = R[r0 + j][c];
        fir_run<FRUN>(win, taps, acc);
        __syncthreads();  // every window is in registers
        if (on) {
#pragma unroll
            for (int o = 0; o < FRUN; ++o) R[FHALO + r0 + o][c] = acc[o];
        }
    }
    __syncthreads();
    // longitude: 32 rows x 8 runs of 4 columns, results to the tile's corner R[0..31][0..31]
    {
        constexpr int XRUN = 4;
        const int r = threadIdx.x % FT, c0 = (threadIdx.x / FT) * XRUN;
        pf2 win[XRUN + 2 * FHALO], acc[XRUN];
#pragma unroll
        for (int j = 0; j < XRUN + 2 * FHALO; ++j) win[j] = R[FHALO + r][c0 + j];
        fir_run<XRUN>(win, taps, acc);
        __syncthreads();
#pragma unroll
        for (int o = 0; o < XRUN; ++o) R[r][c0 + o] = acc[o];
    }
    __syncthreads();
    // store the tile, and the pads whose mirrored source it holds (rows 1, ny-2, ny-3 -> pads -1, ny, ny+1; same in x)
    const int pitch = nx + LC_PAD;
    pf2 *dst = reinterpret_cast<pf2 *>(kind ? ext : img) + (size_t)t * (size_t)(ny + LC_PAD) * pitch;
    for (int i = threadIdx.x; i < FT * FT; i += 256) {
        const int r = i / FT, c = i - r * FT;
        const int gy = gy0 + r, gx = gx0 + c;
        if (gy >= ny || gx >= nx) continue;
        const pf2 val = R[r][c];
        const int py0 = gy + LC_PAD_LO, px0 = gx + LC_PAD_LO;
        const int py1 = gy == 1 ? 0 : (gy == ny - 2 ? ny + 1 : (gy == ny - 3 ? ny + 2 : -1));
        const int px1 = gx == 1 ? 0 : (gx == nx - 2 ? nx + 1 : (gx == nx - 3 ? nx + 2 : -1));
        dst[(size_t)py0 * pitch + px0] = val;
        if (py1 >= 0) dst[(size_t)py1 * pitch + px0] = val;
        if (px1 >= 0) dst[(size_t)py0 * pitch + px1] = val;
        if (py1 >= 0 && px1 >= 0) dst[(size_t)py1 * pitch + px1] = val;
    }
}

static FirTaps cubic_fir_taps() {
    FirTaps T;
    const double z = sqrt(3.0) - 2.0, g = -6.0 * z / (1.0 - z * z);
    for (int n = 0; n <= FHALO; ++n) {
        const float h = (float)(g * pow(z, n));
        T.h[n] = (pf2){h, h};
    }
    return T;
}

template <typename T, typename TIN = T>
int pack_impl(lc_ctx *ctx, const TIN *u, const TIN *v, int nt, int ny, int nx, int order, T *packed, T *ext) {
    const size_t nodes = (size_t)nt * ny * nx;
    const int threads = 256;
    if constexpr (std::is_same<T, TIN>::value) if (order == 1) {
        // (grid.y and grid.z are capped at 65535 blocks: the kernel loops over what is beyond)
        const int nchunk = (nt + PACK_LV - 1) / PACK_LV;
        hipLaunchKernelGGL(pack_fused_kernel<T>, dim3((nx + LC_PAD + 255) / 256, ny + LC_PAD < 65535 ? ny + LC_PAD : 65535, nchunk < 65535 ? nchunk : 65535),
                           dim3(threads), 0, ctx->stream, u, v, packed, ext, nt, ny, nx);
        ctx->last_pack_kernel = "pack_fused_kernel";
        LC_HIP_CHECK(hipGetLastError());
        return LC_OK;
    }
    const int blocks = (int)((nodes + threads - 1) / threads < 8192 ? (nodes + threads - 1) / threads : 8192);
    if constexpr (sizeof(T) == 4 && std::is_same<T, TIN>::value) {
        // float32, order 3: truncated-convolution prefilter, pads and the fused-level image in one pass over the raw
        // field (each reflection of the 14-node halo must stay inside the grid: n >= 16)
        if (order == 3 && ny >= FHALO + 2 && nx >= FHALO + 2 && ctx->fir_prefilter) {
            const bool both = ext && nt >= 2 && ctx->fir_prefilter == 2;  // 2: ext as a second filtered image (measured slower)
            const dim3 grid((nx + FT - 1) / FT, (ny + FT - 1) / FT, nt * (both ? 2 : 1));
            hipLaunchKernelGGL(prefilter_fir_kernel, grid, dim3(256), 0, ctx->stream, u, v, packed, both ? ext : nullptr, nt, ny, nx,
                               cubic_fir_taps());
            ctx->last_pack_kernel = "prefilter_fir_kernel";
            if (ext && nt >= 2 && !both) {  // ext = 2 img[t] - img[t+1] from the finished coefficients (pads rewritten, same values)
                const int nchunk = (nt + PACK_LV - 1) / PACK_LV;
                hipLaunchKernelGGL(pads_ext_kernel<T>, dim3((nx + LC_PAD + 255) / 256, ny + LC_PAD < 65535 ? ny + LC_PAD : 65535, nchunk < 65535 ? nchunk : 65535), dim3(threads), 0, ctx->stream, packed, ext, nt, ny, nx);
            }
            LC_HIP_CHECK(hipGetLastError());
            return LC_OK;
        }
    }
    if (order == 3) {
        // scipy filters axis 0 first, then axis 1 (spline_filter loops over axes in order); the latitude sweep
        // reads the raw field and writes the interleaved image
        size_t lines = (size_t)nt * nx * 2;
        // float64: one read + one write per sweep (lines of 64 nodes or more; LCS_FIR_PREFILTER=0: the two-march kernels)
        const bool stream = sizeof(T) == 8 && ctx->fir_prefilter;
        const bool cols_stream = stream && ny >= 64;
        bool fused = false;
        if constexpr (sizeof(T) == 8) {
            if (cols_stream && nx >= 64 && ctx->fused_prefilter) {   // both sweeps in one pass over the raw planes
                const int nw = lcplan::fused_prefilter_waves(nx, FS_MAXW), nxb = (nx + 32 * (nw - 2) - 1) / (32 * (nw - 2));
                const lcplan::FusedSplit split = lcplan::fused_prefilter_split(nxb * nt, ctx->n_cus, ny);
                const unsigned per = ((unsigned)split.n_whole + 7) / 8 + (((unsigned)nxb * nt - split.n_whole) * split.pieces + 7) / 8;
                hipLaunchKernelGGL(prefilter_fused_stream_kernel<TIN>, dim3(per * 8), dim3(nw * 64), 0, ctx->stream, u, v, packed, nt, ny, nx, nxb,
                                   split);
                fused = true;
                ctx->last_pack_kernel = std::is_same<TIN, float>::value ? "prefilter_fused_stream_kernel<float>" : "prefilter_fused_stream_kernel<double>";
            }
            if (cols_stream && !fused)
                hipLaunchKernelGGL(prefilter_cols_stream_kernel<TIN>, dim3((unsigned)((lines + 255) / 256)), dim3(256), 0, ctx->stream, u, v,
                                   packed, nt, ny, nx);
        }
        if (!cols_stream)
            hipLaunchKernelGGL((prefilter_cols_kernel<T, TIN>), dim3((unsigned)((lines + 255) / 256)), dim3(256), 0,
                               ctx->stream, u, v, packed, nt, ny, nx);
        if (!fused)   // one sweep per axis: the streaming form where the axis is long enough (float64), the two-march kernels otherwise
            ctx->last_pack_kernel = (cols_stream && stream && nx >= RS_RING) ? "prefilter_cols_stream_kernel + prefilter_rows_stream_kernel"
                                    : cols_stream                            ? "prefilter_cols_stream_kernel + prefilter_rows_kernel"
                                    : (stream && nx >= RS_RING)              ? "prefilter_cols_kernel + prefilter_rows_stream_kernel"
                                                                             : "prefilter_cols_kernel + prefilter_rows_kernel";
        if (fused) {
        } else if (stream && nx >= RS_RING) {
            if constexpr (sizeof(T) == 8)
                hipLaunchKernelGGL(prefilter_rows_stream_kernel, dim3((ny + RS_ROWS - 1) / RS_ROWS, nt), dim3(64), 0, ctx->stream, packed,
                                   ny, nx);
        } else if (nx >= PR_CHUNK) {
            hipLaunchKernelGGL(prefilter_rows_lds_kernel<T>, dim3((ny + PR_ROWS - 1) / PR_ROWS, nt), dim3(64), 0,
                               ctx->stream, packed, ny, nx);
        } else {
            lines = (size_t)nt * ny * 2;
            hipLaunchKernelGGL(prefilter_rows_kernel<T>, dim3((unsigned)((lines + 63) / 64)), dim3(64), 0,
                               ctx->stream, packed, nt, ny, nx);
        }
    } else {  // orders 2, 4, 5: generic pole lists, thread per line
        hipLaunchKernelGGL((pack_interior_kernel<T, TIN>), dim3(blocks), dim3(threads), 0, ctx->stream, u, v, packed, ny, nx,
                           nodes);
        ctx->last_pack_kernel = "pack_interior_kernel + prefilter_general_kernel";
        const PoleList P = spline_poles(order);
        const size_t l0 = (size_t)nt * nx * 2, l1 = (size_t)nt * ny * 2;
        hipLaunchKernelGGL(prefilter_general_kernel<T>, dim3((unsigned)((l0 + 63) / 64)), dim3(64), 0, ctx->stream, packed, nt,
                           ny, nx, 0, P);
        hipLaunchKernelGGL(prefilter_general_kernel<T>, dim3((unsigned)((l1 + 63) / 64)), dim3(64), 0, ctx->stream, packed, nt,
                           ny, nx, 1, P);
    }
    const int nchunk = (nt + PACK_LV - 1) / PACK_LV;   // (grid.y and grid.z are capped at 65535 blocks: the kernels loop over what is beyond)
    if (ext && nt >= 2)   // pads + fused-level image in one pass
        hipLaunchKernelGGL(pads_ext_kernel<T>, dim3((nx + LC_PAD + 255) / 256, ny + LC_PAD < 65535 ? ny + LC_PAD : 65535, nchunk < 65535 ? nchunk : 65535), dim3(threads), 0, ctx->stream, packed, ext, nt, ny, nx);
    else                  // the pads alone (no fused-level image: lc_advect_args.fuse_levels_raw, or the reference's two-sample form)
        hipLaunchKernelGGL(pads_only_kernel<T>, dim3((LC_PAD * (nx + LC_PAD) + LC_PAD * ny + 255) / 256, nt < 65535 ? nt : 65535), dim3(threads), 0, ctx->stream, packed, nt, ny, nx);
    LC_HIP_CHECK(hipGetLastError());
    return LC_OK;
}

}  // namespace

int lc_launch_extrapolate(lc_ctx *ctx, const void *img, int dtype, int nt, int ny_f, int nx_f, void *ext) {
    const size_t le = lc_level_elems(ny_f, nx_f), total = le * (size_t)(nt - 1);
    if (dtype == LC_F32)
        hipLaunchKernelGGL(extrapolate_kernel<float>, dim3(8192), dim3(256), 0, ctx->stream, (const float *)img,
                           (float *)ext, le, total);
    else
        hipLaunchKernelGGL(extrapolate_kernel<double>, dim3(8192), dim3(256), 0, ctx->stream, (const double *)img,
                           (double *)ext, le, total);
    LC_HIP_CHECK(hipGetLastError());
    return LC_OK;
}

int lc_launch_pack(lc_ctx *ctx, const void *u, const void *v, int dtype, int nt, int ny_f, int nx_f, int order,
                   void *packed, void *ext) {
    if (dtype == LC_F32)
        return pack_impl<float>(ctx, (const float *)u, (const float *)v, nt, ny_f, nx_f, order, (float *)packed,
                                (float *)ext);
    if (dtype == LC_F64_WIND_F32)   // float32 planes in, float64 spline coefficients out (orders 2..5: lc_field_pack checked)
        return pack_impl<double, float>(ctx, (const float *)u, (const float *)v, nt, ny_f, nx_f, order, (double *)packed, (double *)ext);
    return pack_impl<double>(ctx, (const double *)u, (const double *)v, nt, ny_f, nx_f, order, (double *)packed,
                             (double *)ext);
}
