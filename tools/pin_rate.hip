// pin_rate.hip -- how fast can host memory be page-locked on this box?  Decides how the native driver's chunk pool is built.
//   build: hipcc -O2 -o tools/bin/pin_rate tools/pin_rate.hip ; run: tools/bin/pin_rate [MiB]
// Diagnostic tool only.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

using Clock = std::chrono::steady_clock;
static double secs(Clock::time_point a) { return std::chrono::duration<double>(Clock::now() - a).count(); }

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1024;
    const size_t n = mib << 20;
    hipFree(nullptr);
    void *d = nullptr;
    hipMalloc(&d, 64 << 20);
    auto h2d = [&](void *p, const char *what) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        for (size_t o = 0; o + (64u << 20) <= n; o += 64u << 20) hipMemcpyAsync(d, (char *)p + o, 64u << 20, hipMemcpyHostToDevice, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("    H2D from %-28s %.1f GB/s\n", what, n / ms / 1e6);
    };
    {
        auto t = Clock::now();
        void *p = nullptr;
        hipHostMalloc(&p, n, hipHostMallocDefault);
        printf("hipHostMalloc %zu MiB: %.3f s (%.2f GB/s)\n", mib, secs(t), n / secs(t) / 1e9);
        h2d(p, "hipHostMalloc");
        t = Clock::now();
        hipHostFree(p);
        printf("    hipHostFree %.3f s\n", secs(t));
    }
    {
        auto t = Clock::now();
        const int nt = 8;
        std::vector<void *> ps(nt);
        std::vector<std::thread> th;
        for (int i = 0; i < nt; ++i) th.emplace_back([&, i] { hipHostMalloc(&ps[i], n / nt, hipHostMallocDefault); });
        for (auto &x : th) x.join();
        printf("hipHostMalloc %zu MiB as %d concurrent pieces: %.3f s (%.2f GB/s)\n", mib, nt, secs(t), n / secs(t) / 1e9);
        for (auto p : ps) hipHostFree(p);
    }
    {
        auto t = Clock::now();
        void *p = aligned_alloc(2u << 20, n);
        memset(p, 1, n);
        const double t_touch = secs(t);
        t = Clock::now();
        hipError_t e = hipHostRegister(p, n, hipHostRegisterDefault);
        printf("malloc + touch %.3f s, hipHostRegister %.3f s (%.2f GB/s) %s\n", t_touch, secs(t), n / secs(t) / 1e9, hipGetErrorString(e));
        if (e == hipSuccess) {
            h2d(p, "registered malloc");
            hipHostUnregister(p);
        }
        free(p);
    }
    {
        auto t = Clock::now();
        void *p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(p, n, MADV_HUGEPAGE);
        memset(p, 1, n);
        const double t_touch = secs(t);
        t = Clock::now();
        hipError_t e = hipHostRegister(p, n, hipHostRegisterDefault);
        printf("mmap + MADV_HUGEPAGE + touch %.3f s, hipHostRegister %.3f s (%.2f GB/s) %s\n", t_touch, secs(t), n / secs(t) / 1e9, hipGetErrorString(e));
        if (e == hipSuccess) {
            h2d(p, "registered THP mmap");
            hipHostUnregister(p);
        }
        munmap(p, n);
    }
    {
        auto t = Clock::now();
        void *p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
        const double t_touch = secs(t);
        const int nt = 8;
        t = Clock::now();
        std::vector<std::thread> th;
        for (int i = 0; i < nt; ++i) th.emplace_back([&, i] { hipHostRegister((char *)p + (n / nt) * i, n / nt, hipHostRegisterDefault); });
        for (auto &x : th) x.join();
        printf("mmap MAP_POPULATE %.3f s, hipHostRegister in %d concurrent pieces %.3f s (%.2f GB/s)\n", t_touch, nt, secs(t), n / secs(t) / 1e9);
        for (int i = 0; i < nt; ++i) hipHostUnregister((char *)p + (n / nt) * i);
        munmap(p, n);
    }
    {
        void *p = malloc(n);
        memset(p, 1, n);
        h2d(p, "pageable malloc");
        free(p);
    }
    return 0;
}
