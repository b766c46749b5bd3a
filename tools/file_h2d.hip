// file_h2d.hip -- how fast can the bytes of a FILE (page cache / tmpfs) reach the device, and at what host cost?  Decides the read
// feeder's transport.  Variants, T host threads each moving 32-MB slices on a stream of its own:
//   pread   pread() into a page-locked buffer (huge-page mapping, hipHostRegister once), then hipMemcpyAsync            (one host copy per byte)
//   direct  hipMemcpyAsync straight from the mmap()ed file (pageable source: the runtime stages it)                      (no host code touches a byte)
//   reg     hipHostRegister the mmap()ed slice, hipMemcpyAsync, hipHostUnregister                                        (zero copy)
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/file_h2d tools/file_h2d.hip -lpthread ; run: tools/bin/file_h2d [file] [GiB]
// Diagnostic tool only; not part of the product path.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using Clock = std::chrono::steady_clock;
static double secs(Clock::time_point a) { return std::chrono::duration<double>(Clock::now() - a).count(); }
constexpr size_t SLICE = 32u << 20;

static void *pinned(size_t n) {
    void *p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(p, n, MADV_HUGEPAGE);
    for (size_t o = 0; o < n; o += 4096) ((volatile char *)p)[o] = 0;
    if (hipHostRegister(p, n, hipHostRegisterDefault) != hipSuccess) { printf("hipHostRegister(anon) failed\n"); exit(1); }
    return p;
}

int main(int argc, char **argv) {
    const std::string path = argc > 1 ? argv[1] : "/dev/shm/file_h2d.bin";
    const size_t gib = argc > 2 ? strtoull(argv[2], nullptr, 10) : 4;
    const size_t total = gib << 30;
    hipSetDevice(0);
    hipFree(nullptr);
    {
        int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
        std::vector<char> blk(64u << 20);
        for (size_t i = 0; i < blk.size(); ++i) blk[i] = "ACGT"[(i * 2654435761u >> 13) & 3];
        for (size_t o = 0; o < total; o += blk.size())
            if (write(fd, blk.data(), blk.size()) != (ssize_t)blk.size()) { printf("write failed\n"); return 1; }
        close(fd);
    }
    const int fd = open(path.c_str(), O_RDONLY);
    const size_t n_slices = total / SLICE;
    // every run maps the file afresh: the page tables of a new mapping are empty, and filling them (one minor fault per 16 pages, or
    // madvise(MADV_POPULATE_READ) per slice = "pop") is part of what a zero-copy path costs
    for (const char *mode : {"pread-only", "pread", "direct", "reg", "only-pop", "xpop+direct", "ypop+reg"}) {
        for (int T : {1, 2, 4, 8}) {
            const uint8_t *map = (const uint8_t *)mmap(nullptr, total, PROT_READ, MAP_SHARED, fd, 0);
            if (map == MAP_FAILED) { printf("mmap failed\n"); return 1; }
            std::atomic<size_t> next{0};
            std::atomic<int> bad{0};
            const auto t0 = Clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    hipSetDevice(0);
                    hipStream_t st;
                    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
                    void *d[2] = {nullptr, nullptr};
                    hipMalloc(&d[0], SLICE);
                    hipMalloc(&d[1], SLICE);
                    void *h[2] = {nullptr, nullptr};
                    const bool pr = mode[0] == 'p';
                    if (pr) { h[0] = pinned(SLICE); h[1] = pinned(SLICE); }
                    hipEvent_t ev[2];
                    hipEventCreate(&ev[0]);
                    hipEventCreate(&ev[1]);
                    bool used[2] = {false, false};
                    const uint8_t *reg_prev[2] = {nullptr, nullptr};
                    for (int k = 0;; ++k) {
                        const size_t s = next.fetch_add(1);
                        if (s >= n_slices) break;
                        const int b = k & 1;
                        if (used[b]) {
                            hipEventSynchronize(ev[b]);
                            if (reg_prev[b]) { hipHostUnregister((void *)reg_prev[b]); reg_prev[b] = nullptr; }
                        }
                        const uint8_t *src = map + s * SLICE;
                        if (mode[1] == 'p' || mode[0] == 'o') {
                            if (madvise((void *)src, SLICE, 22 /* MADV_POPULATE_READ */) != 0) { bad = 3; break; }
                            if (mode[0] == 'o') { used[b] = false; continue; }
                        }
                        if (pr) {
                            size_t got = 0;
                            while (got < SLICE) {
                                const ssize_t r = pread(fd, (char *)h[b] + got, SLICE - got, (off_t)(s * SLICE + got));
                                if (r <= 0) { bad = 1; break; }
                                got += (size_t)r;
                            }
                            if (strcmp(mode, "pread-only") == 0) { used[b] = false; continue; }
                            if (hipMemcpyAsync(d[b], h[b], SLICE, hipMemcpyHostToDevice, st) != hipSuccess) bad = 1;
                        } else if (mode[0] == 'd' || mode[0] == 'x') {
                            if (hipMemcpyAsync(d[b], src, SLICE, hipMemcpyHostToDevice, st) != hipSuccess) bad = 1;
                        } else {
                            if (hipHostRegister((void *)src, SLICE, hipHostRegisterDefault) != hipSuccess) { bad = 2; break; }
                            reg_prev[b] = src;
                            if (hipMemcpyAsync(d[b], src, SLICE, hipMemcpyHostToDevice, st) != hipSuccess) bad = 1;
                        }
                        hipEventRecord(ev[b], st);
                        used[b] = true;
                    }
                    hipStreamSynchronize(st);
                    for (int b = 0; b < 2; ++b)
                        if (reg_prev[b]) hipHostUnregister((void *)reg_prev[b]);
                    hipFree(d[0]);
                    hipFree(d[1]);
                    if (pr) {
                        for (int b = 0; b < 2; ++b) { hipHostUnregister(h[b]); munmap(h[b], SLICE); }
                    }
                    hipStreamDestroy(st);
                });
            for (auto &x : th) x.join();
            const double dt = secs(t0);
            printf("%-10s T=%d  %.2f GB/s  (%.3f s)%s\n", mode, T, total / dt / 1e9, dt, bad.load() == 2 ? "  [hipHostRegister on the file mapping FAILED]" : bad.load() == 3 ? "  [madvise(MADV_POPULATE_READ) failed]" : bad.load() ? "  [error]" : "");
            fflush(stdout);
            (void)hipGetLastError();
            munmap((void *)map, total);
            if (bad.load() == 2) break;
        }
    }
    close(fd);
    unlink(path.c_str());
    return 0;
}
