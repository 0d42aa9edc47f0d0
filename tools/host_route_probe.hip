// What the one-call host route (lc_lcs_host) can choose between for moving configs[2]'s 805 MB of wind up and 200 MB of results
// down: pageable hipMemcpy, hipHostRegister + DMA, a pinned staging ring filled by host threads, and what hipMalloc / hipFree
// of its 2.6 GB cost per call.   hipcc -O3 --offload-arch=gfx950 -o build/host_route_probe tools/host_route_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(e)                                                                 \
    do {                                                                      \
        hipError_t _e = (e);                                                  \
        if (_e != hipSuccess) {                                               \
            std::printf("%s failed: %s\n", #e, hipGetErrorString(_e));        \
            return 1;                                                         \
        }                                                                     \
    } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t up = (size_t)97 * 720 * 1440 * 4 * 2, down = (size_t)4096 * 4096 * 4 * 3;
    char *h = (char *)std::malloc(up), *hd = (char *)std::malloc(down);
    std::memset(h, 1, up);
    std::memset(hd, 0, down);
    void *d = nullptr, *dd = nullptr;
    double t = now();
    CK(hipMalloc(&d, up));
    CK(hipMalloc(&dd, down));
    void *big = nullptr;
    CK(hipMalloc(&big, (size_t)1700 << 20));
    std::printf("hipMalloc of %.0f + %.0f + 1700 MB: %.2f ms\n", up / 1e6, down / 1e6, (now() - t) * 1e3);
    t = now();
    CK(hipFree(big));
    std::printf("hipFree of 1700 MB: %.2f ms\n", (now() - t) * 1e3);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 2; ++rep) {
        t = now();
        CK(hipMemcpyAsync(d, h, up, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double a = now() - t;
        t = now();
        CK(hipMemcpyAsync(hd, dd, down, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        std::printf("pageable: H2D %.0f MB %.2f ms (%.1f GB/s), D2H %.0f MB %.2f ms (%.1f GB/s)\n", up / 1e6, a * 1e3, up / a / 1e9, down / 1e6,
                    (now() - t) * 1e3, down / (now() - t) / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        t = now();
        CK(hipHostRegister(h, up, hipHostRegisterDefault));
        double r = now() - t;
        t = now();
        CK(hipMemcpyAsync(d, h, up, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double a = now() - t;
        t = now();
        CK(hipHostUnregister(h));
        std::printf("registered: hipHostRegister %.2f ms, H2D %.2f ms (%.1f GB/s), hipHostUnregister %.2f ms\n", r * 1e3, a * 1e3, up / a / 1e9,
                    (now() - t) * 1e3);
        t = now();
        CK(hipHostRegister(hd, down, hipHostRegisterDefault));
        r = now() - t;
        t = now();
        CK(hipMemcpyAsync(hd, dd, down, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        a = now() - t;
        t = now();
        CK(hipHostUnregister(hd));
        std::printf("registered: hipHostRegister(out) %.2f ms, D2H %.2f ms (%.1f GB/s), unregister %.2f ms\n", r * 1e3, a * 1e3, down / a / 1e9,
                    (now() - t) * 1e3);
    }
    // pinned staging ring: NT host threads copy slices of a chunk into a pinned buffer, the DMA of chunk k overlaps the copy of k+1
    for (int nthreads : {1, 2, 4, 8}) {
        const size_t chunk = (size_t)32 << 20;
        const int ring = 4;
        char *pin[ring];
        hipEvent_t ev[ring];
        for (int i = 0; i < ring; ++i) {
            CK(hipHostMalloc((void **)&pin[i], chunk, hipHostMallocDefault));
            CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        for (int rep = 0; rep < 2; ++rep) {
            t = now();
            size_t off = 0;
            int k = 0;
            while (off < up) {
                const size_t n = std::min(chunk, up - off);
                const int slot = k % ring;
                if (k >= ring) CK(hipEventSynchronize(ev[slot]));
                std::vector<std::thread> th;
                const size_t per = (n + nthreads - 1) / nthreads;
                for (int q = 1; q < nthreads; ++q) {
                    const size_t b = std::min(n, q * per), e = std::min(n, (q + 1) * per);
                    th.emplace_back([=] { std::memcpy(pin[slot] + b, h + off + b, e - b); });
                }
                std::memcpy(pin[slot], h + off, std::min(n, per));
                for (auto &x : th) x.join();
                CK(hipMemcpyAsync((char *)d + off, pin[slot], n, hipMemcpyHostToDevice, st));
                CK(hipEventRecord(ev[slot], st));
                off += n;
                ++k;
            }
            CK(hipStreamSynchronize(st));
            const double a = now() - t;
            if (rep) std::printf("staging ring, %d host thread(s), 32 MB chunks: H2D %.2f ms (%.1f GB/s)\n", nthreads, a * 1e3, up / a / 1e9);
        }
        for (int i = 0; i < ring; ++i) {
            CK(hipHostFree(pin[i]));
            CK(hipEventDestroy(ev[i]));
        }
    }
    {   // fully pinned source: the DMA rate itself
        char *p = nullptr;
        CK(hipHostMalloc((void **)&p, up, hipHostMallocDefault));
        std::memcpy(p, h, up);
        t = now();
        CK(hipMemcpyAsync(d, p, up, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        const double a = now() - t;
        std::printf("hipHostMalloc'ed source: H2D %.2f ms (%.1f GB/s)\n", a * 1e3, up / a / 1e9);
        CK(hipHostFree(p));
    }
    return 0;
}
