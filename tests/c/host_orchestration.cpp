// The host orchestration of liblcs_hip -- lc_lcs_host, lc_lcs_global_host (csrc/api.hip), the spectral-truncation operator
// cache (csrc/preprocess.hip), lc_advect_ex's launcher with its level chunks and the outer-clamp bookkeeping (csrc/advect.hip),
// the wave-state audit's counters -- run against tests/c/fake_hip.c under AddressSanitizer + UBSan (+ LeakSanitizer at exit):
//   * the plain routes in float32 / float64, orders 1 and 3, trajectories, Gaussian smoothing, both fidelity modes;
//   * every bad-argument refusal returns its status and leaves nothing allocated;
//   * an allocation failure injected at EVERY allocation of a route in turn, and a copy failure at every copy: the call
//     returns an error (never crashes, never reads a freed buffer) and the context can still be destroyed with nothing live.
// Built and run by tests/test_host_orchestration_asan.py.  Prints "OK <n checks>" on success.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/lcs_hip.h"

extern "C" {
int fake_hip_live(void);
int fake_hip_mallocs(void);
int fake_hip_copies(void);
int fake_hip_launches(void);
int fake_hip_bad_frees(void);
void fake_hip_fail_malloc_at(int n);
void fake_hip_fail_memcpy_at(int n);
void fake_hip_fail_host_malloc(int on);
void fake_hip_reset_counts(void);
// "device" memory of the stand-in runtime, for the staged-copy calls (hipError_t is an enum: 0 = success)
int hipMalloc(void **p, size_t bytes);
int hipFree(void *p);
}

static int g_checks = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        ++g_checks;                                                                  \
        if (!(cond)) {                                                               \
            fprintf(stderr, "%s:%d: CHECK failed: %s   [%s]\n", __FILE__, __LINE__, #cond, lc_last_error()); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

template <typename T>
struct Case {
    int nt, ny_f, nx_f, ny, nx;
    std::vector<T> u, v, lat, lon, slat, slon, sigma, x, y, tx, ty;
    Case(int nt_, int nyf, int nxf, int ny_, int nx_) : nt(nt_), ny_f(nyf), nx_f(nxf), ny(ny_), nx(nx_) {
        u.assign((size_t)nt * ny_f * nx_f, T(3));
        v.assign(u.size(), T(-1));
        for (size_t i = 0; i < u.size(); ++i) u[i] += T(0.01) * T(i % 97), v[i] += T(0.02) * T(i % 53);
        lat.resize(ny_f), lon.resize(nx_f), slat.resize(ny), slon.resize(nx);
        for (int i = 0; i < ny_f; ++i) lat[i] = T(-80.0 + 160.0 * i / (ny_f - 1));
        for (int i = 0; i < nx_f; ++i) lon[i] = T(-180.0 + 360.0 * i / nx_f);
        for (int i = 0; i < ny; ++i) slat[i] = T(-80.0 + 160.0 * i / (ny - 1));
        for (int i = 0; i < nx; ++i) slon[i] = T(-180.0 + 359.0 * i / (nx - 1));
        sigma.resize((size_t)ny * nx), x.resize(sigma.size()), y.resize(sigma.size());
        tx.resize((size_t)nt * ny * nx), ty.resize(tx.size());
    }
    int run(lc_ctx *ctx, int dtype, int K, int order, int cyclic, double gauss, bool traj) {
        return lc_lcs_host(ctx, u.data(), v.data(), dtype, nt, ny_f, nx_f, lat.data(), lon.data(), slat.data(), ny, slon.data(), nx,
                           -900.0, K, order, cyclic, 0, nt - 1, gauss, 1, LC_LAYOUT_REFERENCE, sigma.data(), x.data(), y.data(),
                           traj ? tx.data() : nullptr, traj ? ty.data() : nullptr);
    }
};

// a route run once clean (to count its allocations and copies), then once per allocation / copy with that one failing
template <typename F>
static void sweep_failures(lc_ctx *ctx, const char *what, F route) {
    // (lc_lcs_host keeps its device buffers on the context for the next call: lc_ctx_trim before every count and every run, so
    //  that each run allocates everything anew and every allocation can be the one that fails)
    auto live = [&] { lc_ctx_trim(ctx); return fake_hip_live(); };
    fake_hip_reset_counts();
    const int live0 = live();
    CHECK(route() == LC_OK);
    CHECK(live() == live0);   // (the truncation operators stay cached on the context: counted in live0 from the second run on)
    const int live1 = live();
    fake_hip_reset_counts();
    CHECK(route() == LC_OK);
    const int n_malloc = fake_hip_mallocs(), n_copy = fake_hip_copies();
    CHECK(live() == live1 && n_malloc > 0 && n_copy > 0 && fake_hip_launches() > 0);
    for (int k = 1; k <= n_malloc; ++k) {
        fake_hip_fail_malloc_at(k);
        const int rc = route();
        fake_hip_fail_malloc_at(0);
        if (!(rc == LC_ENOMEM || rc == LC_EHIP || rc == LC_OK)) {   // (LC_OK: an optional scratch buffer whose absence has a fallback)
            fprintf(stderr, "%s: allocation %d of %d failing gave status %d [%s]\n", what, k, n_malloc, rc, lc_last_error());
            exit(1);
        }
        if (live() != live1) {
            fprintf(stderr, "%s: allocation %d of %d failing left %d buffers live (%d before) [%s]\n", what, k, n_malloc, fake_hip_live(), live1, lc_last_error());
            exit(1);
        }
        ++g_checks;
    }
    for (int k = 1; k <= n_copy; ++k) {
        fake_hip_fail_memcpy_at(k);
        const int rc = route();
        fake_hip_fail_memcpy_at(0);
        if (rc == LC_OK || live() != live1) {
            fprintf(stderr, "%s: copy %d of %d failing gave status %d, %d buffers live (%d before) [%s]\n", what, k, n_copy, rc, fake_hip_live(), live1, lc_last_error());
            exit(1);
        }
        ++g_checks;
    }
    CHECK(route() == LC_OK && live() == live1);
}

// buffers allocated right now, not counting what lc_lcs_host keeps on the context for its next call
#define LIVE_AFTER_TRIM() (lc_ctx_trim(ctx), fake_hip_live())

int main() {
    CHECK(lc_version() == LC_VERSION);
    lc_ctx *ctx = nullptr;
    CHECK(lc_ctx_create(3, &ctx) == LC_EINVAL && ctx == nullptr);   // the fake machine has one device
    CHECK(lc_ctx_create(0, &ctx) == LC_OK && ctx != nullptr);
    {   // the staging ring of the host routes (csrc/hostxfer.h): created by the first call that needs it, kept on the context
        const int before = fake_hip_live();
        Case<float> w(5, 24, 40, 33, 47);
        fake_hip_fail_host_malloc(1);     // no pinned memory to be had: the route falls back to plain copies and keeps nothing
        CHECK(w.run(ctx, LC_F32, 4, 1, 1, 0.0, false) == LC_OK && LIVE_AFTER_TRIM() == before);
        fake_hip_fail_host_malloc(0);
        CHECK(w.run(ctx, LC_F32, 4, 1, 1, 0.0, false) == LC_OK && LIVE_AFTER_TRIM() == before + 4);   // four pinned pieces
        CHECK(w.run(ctx, LC_F32, 4, 1, 1, 0.0, false) == LC_OK && LIVE_AFTER_TRIM() == before + 4);   // ... once
        // ONE ring per device, shared by the contexts: a second context's calls pin nothing more, destroying either context
        // leaves the ring to the other, the last one takes it down (and the next context builds it again)
        lc_ctx *other = nullptr;
        CHECK(lc_ctx_create(0, &other) == LC_OK);
        unsigned char bytes[64] = {1, 2, 3}, back[64] = {0};
        void *dev = nullptr;
        CHECK(hipMalloc(&dev, sizeof bytes) == 0);
        const int with_dev = fake_hip_live();
        CHECK(w.run(other, LC_F32, 4, 1, 1, 0.0, false) == LC_OK && lc_ctx_trim(other) == LC_OK && fake_hip_live() == with_dev);
        CHECK(lc_copy_to_device(other, dev, bytes, sizeof bytes) == LC_OK && lc_copy_to_host(ctx, back, dev, sizeof bytes) == LC_OK);
        CHECK(std::memcmp(bytes, back, sizeof bytes) == 0 && fake_hip_live() == with_dev);
        CHECK(lc_ctx_destroy(other) == LC_OK && fake_hip_live() == with_dev);                           // the ring stays: `ctx` holds it
        CHECK(lc_copy_to_host(ctx, back, dev, sizeof bytes) == LC_OK && hipFree(dev) == 0);
        // ... and two threads, a context each, on that one ring at once: staged copies longer than a piece and the pipelined route
        // (the ring's `use` lock takes them in turn; ThreadSanitizer watches the slots, the workers and the books)
        CHECK(lc_ctx_create(0, &other) == LC_OK);
        std::atomic<int> bad{0};
        auto hammer = [&](lc_ctx *c, unsigned seed) {
            const size_t n = ((size_t)40 << 20) + seed;
            std::vector<unsigned char> src(n), got(n, 0);
            for (size_t i = 0; i < n; i += 4099) src[i] = (unsigned char)(i * 13 + seed);
            void *d = nullptr;
            if (hipMalloc(&d, n) != 0) { ++bad; return; }
            Case<float> mine(37, 24, 40, 33, 47);
            for (int rep = 0; rep < 3; ++rep) {
                if (lc_copy_to_device(c, d, src.data(), n) != LC_OK || lc_copy_to_host(c, got.data(), d, n) != LC_OK) ++bad;
                if (std::memcmp(src.data(), got.data(), n) != 0) ++bad;
                if (mine.run(c, LC_F32, 4, 1, 1, 0.0, false) != LC_OK) ++bad;
            }
            if (hipFree(d) != 0) ++bad;
        };
        std::thread t1(hammer, ctx, 1u), t2(hammer, other, 2u);
        t1.join();
        t2.join();
        CHECK(bad.load() == 0 && lc_ctx_trim(other) == LC_OK && lc_ctx_destroy(other) == LC_OK && LIVE_AFTER_TRIM() == before + 4);
    }
    {   // the device buffers of a call stay on the context for the next one: the second call of a shape allocates nothing,
        // lc_ctx_trim returns them, lc_ctx_set_host_cache(0) stops keeping them
        Case<float> w(5, 24, 40, 33, 47);
        const int before = LIVE_AFTER_TRIM();
        CHECK(w.run(ctx, LC_F32, 4, 1, 1, 0.0, false) == LC_OK && fake_hip_live() > before);
        const int kept = fake_hip_live();
        fake_hip_reset_counts();
        CHECK(w.run(ctx, LC_F32, 4, 1, 1, 0.0, false) == LC_OK && fake_hip_mallocs() == 0 && fake_hip_live() == kept);
        CHECK(lc_ctx_trim(ctx) == LC_OK && fake_hip_live() == before && lc_ctx_trim(nullptr) == LC_EINVAL);
        CHECK(lc_ctx_set_host_cache(ctx, 0) == LC_OK && w.run(ctx, LC_F32, 4, 1, 1, 0.0, false) == LC_OK && fake_hip_live() == before);
        CHECK(lc_ctx_set_host_cache(ctx, 2) == LC_EINVAL && lc_ctx_set_host_cache(ctx, 1) == LC_OK);
        double marks[4] = {-1, -1, -1, -1};
        CHECK(lc_ctx_last_host_marks(ctx, marks) == LC_OK && marks[0] >= 0 && marks[3] >= marks[1] && lc_ctx_last_host_marks(ctx, nullptr) == LC_EINVAL);
        CHECK(lc_ctx_set_host_pipeline(ctx, 0) == LC_OK && w.run(ctx, LC_F32, 4, 1, 1, 0.0, false) == LC_OK);     // plain copies on request
        CHECK(lc_ctx_set_host_pipeline(ctx, 3) == LC_EINVAL && lc_ctx_set_host_pipeline(ctx, 1) == LC_OK);
        CHECK(lc_ctx_set_xcd_split(ctx, 8) == LC_OK && lc_ctx_set_xcd_split(ctx, -2) == LC_EINVAL && lc_ctx_set_xcd_split(ctx, -1) == LC_OK);
    }
    {   // staged copies on their own (lc_copy_to_device / lc_copy_to_host): sizes below, at and across the ring's 32 MB pieces and
        // longer than the whole ring; with the ring switched off; refusals; a failing DMA is reported and leaves nothing behind
        const int before = LIVE_AFTER_TRIM();
        for (size_t n : {(size_t)1, (size_t)4097, ((size_t)32 << 20), ((size_t)32 << 20) + 13, ((size_t)150 << 20) + 7}) {
            std::vector<unsigned char> src(n), back(n, 0);
            for (size_t i = 0; i < n; i += 4093) src[i] = (unsigned char)(i * 31 + 7);
            src[n - 1] = 0xA5;
            void *dev = nullptr;
            CHECK(hipMalloc(&dev, n) == 0);
            for (int on = 1; on >= 0; --on) {
                std::fill(back.begin(), back.end(), 0);
                CHECK(lc_ctx_set_host_pipeline(ctx, on) == LC_OK);
                CHECK(lc_copy_to_device(ctx, dev, src.data(), n) == LC_OK && lc_copy_to_host(ctx, back.data(), dev, n) == LC_OK);
                CHECK(std::memcmp(src.data(), back.data(), n) == 0);
            }
            CHECK(lc_ctx_set_host_pipeline(ctx, 1) == LC_OK);
            fake_hip_fail_memcpy_at(2);
            CHECK(lc_copy_to_device(ctx, dev, src.data(), n) != LC_OK || n <= ((size_t)32 << 20));   // (one piece: one copy, the second never happens)
            fake_hip_fail_memcpy_at(0);
            CHECK(lc_copy_to_device(ctx, dev, src.data(), n) == LC_OK);
            CHECK(hipFree(dev) == 0);
        }
        CHECK(lc_copy_to_device(ctx, nullptr, nullptr, 0) == LC_OK && lc_copy_to_device(ctx, nullptr, nullptr, 8) == LC_EINVAL);
        CHECK(lc_copy_to_host(nullptr, nullptr, nullptr, 8) == LC_EINVAL && LIVE_AFTER_TRIM() == before);
    }
    const int base = LIVE_AFTER_TRIM();
    {
        Case<float> c(5, 24, 40, 33, 47);
        Case<double> d(5, 24, 40, 33, 47);
        for (int order = 1; order <= 3; order += 2)
            for (int cyclic = 0; cyclic <= 2; ++cyclic) {   // LC_X_CLAMP_POINT, LC_X_CYCLIC, LC_X_CLAMP_REFERENCE_OUTER
                CHECK(c.run(ctx, LC_F32, 4, order, cyclic, 0.0, false) == LC_OK && LIVE_AFTER_TRIM() == base);
                CHECK(d.run(ctx, LC_F64, 2, order, cyclic, 1.5, true) == LC_OK && LIVE_AFTER_TRIM() == base);
            }
        CHECK(lc_ctx_set_f64_fidelity(ctx, LC_F64_FAST) == LC_OK && d.run(ctx, LC_F64, 4, 3, 1, 0.0, false) == LC_OK);
        CHECK(lc_ctx_set_f64_fidelity(ctx, LC_F64_EXACT_ORDER) == LC_OK && d.run(ctx, LC_F64, 4, 1, 1, 0.0, true) == LC_OK);
        CHECK(lc_ctx_set_f64_fidelity(ctx, 7) == LC_EINVAL && lc_ctx_set_f64_fidelity(ctx, LC_F64_AUTO) == LC_OK);
        CHECK(lc_ctx_set_level_chunk(ctx, 2) == LC_OK && c.run(ctx, LC_F32, 4, 1, 1, 0.0, true) == LC_OK);   // several launches per call
        CHECK(lc_ctx_set_level_chunk(ctx, -1) == LC_OK && LIVE_AFTER_TRIM() == base);
        // refusals: nothing may stay allocated
        CHECK(c.run(ctx, 9, 4, 1, 1, 0.0, false) == LC_EINVAL);
        CHECK(c.run(ctx, LC_F32, -1, 1, 1, 0.0, false) == LC_EINVAL);
        CHECK(c.run(ctx, LC_F32, 4, 0, 1, 0.0, false) == LC_EUNSUPPORTED);
        CHECK(c.run(ctx, LC_F32, 4, 6, 1, 0.0, false) == LC_EUNSUPPORTED);
        CHECK(c.run(ctx, LC_F32, 4, 1, 5, 0.0, false) == LC_EINVAL);
        CHECK(lc_lcs_host(ctx, nullptr, c.v.data(), LC_F32, c.nt, c.ny_f, c.nx_f, c.lat.data(), c.lon.data(), c.slat.data(), c.ny, c.slon.data(), c.nx,
                          -900.0, 4, 1, 1, 0, 4, 0.0, 1, 0, c.sigma.data(), nullptr, nullptr, nullptr, nullptr) == LC_EINVAL);
        CHECK(lc_lcs_host(ctx, c.u.data(), c.v.data(), LC_F32, c.nt, c.ny_f, c.nx_f, c.lat.data(), c.lon.data(), c.slat.data(), c.ny, c.slon.data(), c.nx,
                          -900.0, 4, 1, 1, 3, 4, 0.0, 1, 0, c.sigma.data(), nullptr, nullptr, nullptr, nullptr) == LC_EINVAL);   // steps beyond the series
        CHECK(lc_lcs_host(nullptr, c.u.data(), c.v.data(), LC_F32, c.nt, c.ny_f, c.nx_f, c.lat.data(), c.lon.data(), c.slat.data(), c.ny, c.slon.data(), c.nx,
                          -900.0, 4, 1, 1, 0, 4, 0.0, 1, 0, c.sigma.data(), nullptr, nullptr, nullptr, nullptr) == LC_EINVAL);
        CHECK(LIVE_AFTER_TRIM() == base && fake_hip_bad_frees() == 0);
        sweep_failures(ctx, "lc_lcs_host float32 order 1", [&] { return c.run(ctx, LC_F32, 4, 1, 1, 0.0, true); });
        sweep_failures(ctx, "lc_lcs_host float32 order 3 + gauss", [&] { return c.run(ctx, LC_F32, 4, 3, 1, 2.0, false); });
        sweep_failures(ctx, "lc_lcs_host float64 order 3", [&] { return d.run(ctx, LC_F64, 4, 3, 1, 0.0, true); });
        sweep_failures(ctx, "lc_lcs_host float64 outer clamp", [&] { return d.run(ctx, LC_F64, 2, 1, 2, 0.0, false); });
        // a series long enough for the pipelined form (upload of level chunk c + 1 while chunk c is packed and advected: three
        // chunks of 16 levels here), both orders and dtypes, a sub-range of the series, and every allocation / copy of it failing
        Case<float> p(41, 24, 40, 33, 47);
        Case<double> q(41, 24, 40, 33, 47);
        CHECK(p.run(ctx, LC_F32, 4, 1, 1, 0.0, false) == LC_OK && p.run(ctx, LC_F32, 4, 3, 1, 0.0, false) == LC_OK && LIVE_AFTER_TRIM() == base);
        CHECK(lc_ctx_set_f64_fidelity(ctx, LC_F64_FAST) == LC_OK && q.run(ctx, LC_F64, 4, 1, 1, 0.0, false) == LC_OK && q.run(ctx, LC_F64, 2, 3, 1, 1.0, false) == LC_OK);
        CHECK(lc_lcs_host(ctx, p.u.data(), p.v.data(), LC_F32, p.nt, p.ny_f, p.nx_f, p.lat.data(), p.lon.data(), p.slat.data(), p.ny, p.slon.data(), p.nx,
                          -900.0, 4, 1, 1, 3, 35, 0.0, 1, 0, p.sigma.data(), p.x.data(), p.y.data(), nullptr, nullptr) == LC_OK);   // levels 3 .. 38 of 41
        CHECK(lc_lcs_host(ctx, p.u.data(), p.v.data(), LC_F32, p.nt, p.ny_f, p.nx_f, p.lat.data(), p.lon.data(), p.slat.data(), p.ny, p.slon.data(), p.nx,
                          -900.0, 4, 1, 1, 5, 7, 0.0, 1, 0, p.sigma.data(), p.x.data(), p.y.data(), nullptr, nullptr) == LC_OK);    // a short sub-range: serial form, those levels only
        sweep_failures(ctx, "lc_lcs_host float32 order 1, pipelined", [&] { return p.run(ctx, LC_F32, 4, 1, 1, 0.0, false); });
        sweep_failures(ctx, "lc_lcs_host float64 order 3, pipelined", [&] { return q.run(ctx, LC_F64, 4, 3, 1, 0.0, false); });
        CHECK(lc_ctx_set_f64_fidelity(ctx, LC_F64_AUTO) == LC_OK && LIVE_AFTER_TRIM() == base);
    }
    {   // the reference's default global call form: regrid + T20 truncation (operator cache) + the path
        int gy = 0, gx = 0;
        CHECK(lc_common_grid(&gy, &gx, nullptr, nullptr) == LC_OK && gy == 360 && gx == 721);
        const int nt = 3, ny_f = 45, nx_f = 90;   // (an odd number of equally spaced latitudes includes the poles: windspharm's rule)
        std::vector<double> u((size_t)nt * ny_f * nx_f, 5.0), v(u.size(), 1.0), lat(ny_f), lon(nx_f);
        for (size_t i = 0; i < u.size(); ++i) u[i] += 0.1 * std::sin(0.01 * (double)i);
        for (int i = 0; i < ny_f; ++i) lat[i] = -90.0 + 180.0 * i / (ny_f - 1);
        for (int i = 0; i < nx_f; ++i) lon[i] = -180.0 + 360.0 * i / nx_f;
        std::vector<double> sg((size_t)gy * gx), x(sg.size()), y(sg.size());
        auto global = [&](int common, int trunc) {
            return lc_lcs_global_host(ctx, u.data(), v.data(), LC_F64, nt, ny_f, nx_f, lat.data(), lon.data(), common, trunc, -21600.0, 4, 3,
                                      0.0, 1, LC_LAYOUT_REFERENCE, sg.data(), x.data(), y.data());
        };
        CHECK(global(0, 20) == LC_OK);            // the operators of (45, 90, T20) are now cached on the context
        CHECK(global(0, 10) == LC_OK);            // another truncation: the cache is replaced, the old operators freed
        CHECK(global(0, -1) == LC_OK);
        CHECK(global(0, 200) != LC_OK);           // beyond the grid's resolution
        CHECK(lc_lcs_global_host(ctx, u.data(), v.data(), 7, nt, ny_f, nx_f, lat.data(), lon.data(), 0, 20, -21600.0, 4, 3, 0.0, 1, 0, sg.data(), nullptr, nullptr) == LC_EINVAL);
        sweep_failures(ctx, "lc_lcs_global_host own grid, T10", [&] { return global(0, 10); });
        sweep_failures(ctx, "lc_lcs_global_host common grid, T20", [&] { return global(1, 20); });
    }
    {   // the wave-state audit's counters
        unsigned out[LC_VERIFY_WORDS];
        CHECK(lc_ctx_read_verify(ctx, out, 1) == LC_EINVAL);
        CHECK(lc_ctx_set_verify(ctx, 3) == LC_EINVAL && lc_ctx_set_verify(ctx, 2) == LC_OK);
        CHECK(lc_ctx_read_verify(ctx, out, 1) == LC_OK && out[15] == 0xBADu && out[0] == 0);
        CHECK(lc_ctx_set_verify(ctx, 1) == LC_OK && lc_ctx_read_verify(ctx, out, 0) == LC_OK && out[15] == 0);
        fake_hip_fail_malloc_at(1);
        CHECK(lc_ctx_set_verify(ctx, 0) == LC_OK && lc_ctx_set_verify(ctx, 1) == LC_ENOMEM && lc_ctx_read_verify(ctx, out, 0) == LC_EINVAL);
        fake_hip_fail_malloc_at(0);
        CHECK(lc_ctx_set_verify(ctx, 1) == LC_OK);   // left on: lc_ctx_destroy frees the counters
    }
    CHECK(lc_ctx_destroy(ctx) == LC_OK);
    CHECK(fake_hip_live() == 0 && fake_hip_bad_frees() == 0);
    printf("OK %d checks\n", g_checks);
    return 0;
}
