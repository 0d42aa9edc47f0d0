// Host <-> device transfers of the one-call host routes (lc_lcs_host): a pinned staging ring fed by a few host threads.
//
// What the reference-side binding hands lc_lcs_host is ordinary (pageable) memory: numpy arrays.  Measured on an MI355X box
// (tools/host_route_probe.hip, profiles/r06/host_route_probe.txt), configs[2]'s 805 MB of wind up and 201 MB of results down:
//   hipMemcpy from / to pageable memory the runtime has not seen before   14.4 GB/s up, 12.2 GB/s down (it pins the pages
//                                                                         on the fly: the same range again runs at 56 GB/s)
//   hipHostRegister of the range, then DMA                                53.6 ms to register 805 MB, then 56.6 GB/s
//   a ring of 32 MB pinned buffers filled by >= 2 host threads            52.8-53.8 GB/s, whatever the caller's pages are
// -- so the ring: the DMA of piece k runs while the threads copy piece k + 1, and the upload is cut at time-level boundaries
// so that the pack and advect kernels of level chunk c run while chunk c + 1 is on the bus (api.hip: lc_lcs_host).
// Downloads mirror it (DMA into the ring, threads copy out); the pages of the caller's fresh output arrays are touched by a
// background thread during the upload, so the copy-out does not pay their faults.
//
// ONE ring per device and process, shared by every context (acquire / release, `use` held for the length of a transfer
// sequence).  Measured (profiles/r06/host_route.txt): with two rings alive -- the Engine's context had staged an upload, the
// one-call route's context then made its own -- the SECOND ring's copy stream moves the same bytes 25 % slower (805 MB up in
// 18.0 instead of 14.2 ms, whichever context created it second; the first one's stays fast): the runtime gives a process's
// later copy streams another DMA engine.
#pragma once
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

struct lc_host_xfer {
    static constexpr int RING = 4;
    size_t piece = (size_t)32 << 20;
    char *pin[RING] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t dma_done[RING] = {nullptr, nullptr, nullptr, nullptr};  // the last DMA out of / into the slot
    bool in_flight[RING] = {false, false, false, false};
    int next = 0;
    hipStream_t copy = nullptr;  // H2D / D2H DMAs (own non-blocking stream)

    // a few persistent worker threads: parallel_copy() splits one memcpy between them and the calling thread
    struct Slice {
        char *dst;
        const char *src;
        size_t n;
    };
    std::vector<std::thread> workers;
    std::vector<Slice> slices;  // one per worker, valid while `generation` is odd-numbered work
    std::mutex m;
    std::condition_variable wake, finished;
    unsigned long generation = 0;
    int pending = 0;
    bool quit = false;
    double last_down_wait_ms = 0.0, last_down_copy_ms = 0.0;  // the last download(): waiting for DMAs / copying out of the ring
    std::mutex use;  // held by the one transfer sequence that is using the ring (slots, `next`, the copy stream's order)
    int refs = 0, device = -1;

    void worker(int id) {
        unsigned long seen = 0;
        for (;;) {
            Slice s;
            {
                std::unique_lock<std::mutex> lk(m);
                wake.wait(lk, [&] { return quit || generation != seen; });
                if (quit) return;
                seen = generation;
                s = slices[(size_t)id];
            }
            if (s.n) std::memcpy(s.dst, s.src, s.n);
            {
                std::lock_guard<std::mutex> lk(m);
                if (--pending == 0) finished.notify_one();
            }
        }
    }

    void parallel_copy(void *dst, const void *src, size_t n) {
        const size_t parts = workers.size() + 1;
        if (workers.empty() || n < ((size_t)1 << 20)) {
            std::memcpy(dst, src, n);
            return;
        }
        const size_t per = ((n + parts - 1) / parts + 63) / 64 * 64;
        {
            std::lock_guard<std::mutex> lk(m);
            for (size_t w = 0; w < workers.size(); ++w) {
                const size_t b = std::min(n, (w + 1) * per), e = std::min(n, (w + 2) * per);
                slices[w] = Slice{(char *)dst + b, (const char *)src + b, e - b};
            }
            pending = (int)workers.size();
            ++generation;
        }
        wake.notify_all();
        std::memcpy(dst, src, std::min(n, per));
        std::unique_lock<std::mutex> lk(m);
        finished.wait(lk, [&] { return pending == 0; });
    }

    // ---- life cycle -------------------------------------------------------------------------------------------------
    // threads < 0 / piece_bytes == 0: the defaults below (LCS_HOST_THREADS, LCS_HOST_PIECE_MB at context creation: experiments)
    static lc_host_xfer *create(hipError_t *err, int threads = -1, size_t piece_bytes = 0) {
        lc_host_xfer *x = new (std::nothrow) lc_host_xfer;
        if (!x) {
            *err = hipErrorOutOfMemory;
            return nullptr;
        }
        if (piece_bytes) x->piece = piece_bytes;
        hipError_t e = hipStreamCreateWithFlags(&x->copy, hipStreamNonBlocking);
        for (int i = 0; i < RING && e == hipSuccess; ++i) {
            e = hipHostMalloc((void **)&x->pin[i], x->piece, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&x->dma_done[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            *err = e;
            destroy(x);
            return nullptr;
        }
        unsigned hw = std::thread::hardware_concurrency();
        // + the calling thread.  Two threads already feed the bus on the way up (the DMA of piece k hides behind the copy of
        // k + 1); the way down ends with the copy-out of the last pieces, which is as fast as the threads are many
        const int nw = threads >= 0 ? std::min(threads, 63) : (hw >= 16 ? 7 : (hw >= 8 ? 3 : (hw >= 4 ? 1 : 0)));
        x->slices.resize((size_t)nw);
        try {
            for (int i = 0; i < nw; ++i) x->workers.emplace_back(&lc_host_xfer::worker, x, i);
        } catch (...) {  // no threads to be had: the calling thread copies alone
            x->slices.resize(x->workers.size());
        }
        *err = hipSuccess;
        return x;
    }

    // the device's ring, created by the first context that asks (that context's thread / piece settings); NULL with *err set
    // when it cannot be had.  Every acquire is paired with one release; the last one destroys the ring.
    static constexpr int MAX_DEVICES = 64;
    static std::mutex &registry_lock() {
        static std::mutex m;
        return m;
    }
    static lc_host_xfer **registry() {
        static lc_host_xfer *rings[MAX_DEVICES] = {};
        return rings;
    }
    static lc_host_xfer *acquire(int device, hipError_t *err, int threads = -1, size_t piece_bytes = 0) {
        *err = hipSuccess;
        if (device < 0 || device >= MAX_DEVICES) {
            *err = hipErrorInvalidDevice;
            return nullptr;
        }
        std::lock_guard<std::mutex> lk(registry_lock());
        lc_host_xfer *&slot = registry()[device];
        if (!slot) {
            slot = create(err, threads, piece_bytes);
            if (!slot) return nullptr;
            slot->device = device;
        }
        ++slot->refs;
        return slot;
    }
    static void release(lc_host_xfer *x) {
        if (!x) return;
        std::lock_guard<std::mutex> lk(registry_lock());
        if (--x->refs > 0) return;
        if (x->device >= 0 && x->device < MAX_DEVICES && registry()[x->device] == x) registry()[x->device] = nullptr;
        destroy(x);
    }

    static void destroy(lc_host_xfer *x) {
        if (!x) return;
        {
            std::lock_guard<std::mutex> lk(x->m);
            x->quit = true;
        }
        x->wake.notify_all();
        for (auto &t : x->workers) t.join();
        if (x->copy) (void)hipStreamSynchronize(x->copy);
        for (int i = 0; i < RING; ++i) {
            if (x->dma_done[i]) (void)hipEventDestroy(x->dma_done[i]);
            if (x->pin[i]) (void)hipHostFree(x->pin[i]);
        }
        if (x->copy) (void)hipStreamDestroy(x->copy);
        delete x;
    }

    // ---- transfers ---------------------------------------------------------------------------------------------------
    // host -> device, enqueued on `copy`; returns when the last piece has been HANDED to the DMA engine (not when it arrived:
    // record an event on `copy` for that)
    hipError_t upload(void *dev, const void *host, size_t bytes) {
        for (size_t off = 0; off < bytes; off += piece) {
            const size_t n = std::min(piece, bytes - off);
            const int s = next;
            next = (next + 1) % RING;
            if (in_flight[s]) {
                hipError_t e = hipEventSynchronize(dma_done[s]);
                if (e != hipSuccess) return e;
                in_flight[s] = false;
            }
            parallel_copy(pin[s], (const char *)host + off, n);
            hipError_t e = hipMemcpyAsync((char *)dev + off, pin[s], n, hipMemcpyHostToDevice, copy);
            if (e == hipSuccess) e = hipEventRecord(dma_done[s], copy);
            if (e != hipSuccess) return e;
            in_flight[s] = true;
        }
        return hipSuccess;
    }

    // device -> host (after everything enqueued on `copy` so far); returns when the bytes are in the host buffers.  Several
    // buffers go through ONE pipelined sequence of pieces: the DMA of piece k + 1 ... k + RING - 1 runs while the threads copy
    // piece k out, across buffer boundaries (three 67 MB results one after the other would each pay the ring's fill and drain).
    struct Range {
        void *host;
        const void *dev;
        size_t bytes;
    };
    hipError_t download(const std::vector<Range> &ranges) {
        struct Piece {
            char *host;
            const char *dev;
            size_t n;
        };
        std::vector<Piece> pieces;
        for (const Range &r : ranges)
            for (size_t off = 0; off < r.bytes; off += piece)
                pieces.push_back(Piece{(char *)r.host + off, (const char *)r.dev + off, std::min(piece, r.bytes - off)});
        const size_t np = pieces.size();
        auto issue = [&](size_t i) -> hipError_t {
            const int s = (int)(i % RING);
            // (a slot still feeding an upload's DMA: that DMA is earlier on the same stream, the stream orders them)
            hipError_t e = hipMemcpyAsync(pin[s], pieces[i].dev, pieces[i].n, hipMemcpyDeviceToHost, copy);
            if (e == hipSuccess) e = hipEventRecord(dma_done[s], copy);
            in_flight[s] = e == hipSuccess;
            return e;
        };
        for (size_t i = 0; i < std::min(np, (size_t)RING); ++i) {
            hipError_t e = issue(i);
            if (e != hipSuccess) return e;
        }
        using clk = std::chrono::steady_clock;
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        last_down_wait_ms = last_down_copy_ms = 0.0;
        for (size_t j = 0; j < np; ++j) {
            const int s = (int)(j % RING);
            const auto t0 = clk::now();
            hipError_t e = hipEventSynchronize(dma_done[s]);
            if (e != hipSuccess) return e;
            in_flight[s] = false;
            const auto t1 = clk::now();
            parallel_copy(pieces[j].host, pin[s], pieces[j].n);
            last_down_wait_ms += ms(t0, t1);
            last_down_copy_ms += ms(t1, clk::now());
            if (j + RING < np) {
                e = issue(j + RING);
                if (e != hipSuccess) return e;
            }
        }
        next = 0;
        return hipSuccess;
    }
    hipError_t download(void *host, const void *dev, size_t bytes) { return download(std::vector<Range>{Range{host, dev, bytes}}); }

    // every DMA handed over so far has finished (before device buffers they touch are freed, or after a failure)
    void drain() {
        if (copy) (void)hipStreamSynchronize(copy);
        for (int i = 0; i < RING; ++i) in_flight[i] = false;
    }
};

// Populates the pages of a caller's output buffers from background threads (a fresh numpy array is unfaulted: the copy-out
// would pay 4 KB faults at a quarter of the copy rate).  What populating costs is the kernel ZEROING the pages -- 201 MB of
// results is 20 ms of one thread, longer than the upload it hides behind -- so the ranges are cut between a few threads.
// Joined by the destructor.
struct lc_prefault {
    std::vector<std::thread> ts;
    static void populate(char *b, size_t n) {
        volatile char *p = (volatile char *)b;
        if (!p || !n) return;
#ifdef MADV_POPULATE_WRITE
        // whole pages inside the range in one call (Linux 5.14: no fault per page); the partial pages at either end, and
        // everything when the call is refused, by touching
        const uintptr_t lo = ((uintptr_t)b + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)b + n) & ~(uintptr_t)4095;
        if (hi > lo && madvise((void *)lo, hi - lo, MADV_POPULATE_WRITE) == 0) {
            p[0] = 0;
            p[n - 1] = 0;
            return;
        }
#endif
        for (size_t o = 0; o < n; o += 4096) p[o] = 0;
        p[n - 1] = 0;
    }
    void start(std::vector<std::pair<void *, size_t>> ranges) {
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t parts = hw >= 16 ? 4 : (hw >= 4 ? 2 : 1);
        for (size_t k = 0; k < parts; ++k) {
            try {
                ts.emplace_back([ranges, k, parts] {
                    for (auto &r : ranges) {
                        if (!r.first) continue;
                        const size_t per = ((r.second + parts - 1) / parts + 4095) & ~(size_t)4095;
                        const size_t b = std::min(r.second, k * per), e = std::min(r.second, (k + 1) * per);
                        populate((char *)r.first + b, e - b);
                    }
                });
            } catch (...) {  // no thread: the copy-out pays the faults of this part
            }
        }
    }
    void join() {
        for (auto &t : ts)
            if (t.joinable()) t.join();
        ts.clear();
    }
    ~lc_prefault() { join(); }
};
