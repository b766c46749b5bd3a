// alloc_probe.hip -- what a large device allocation costs on this platform (the index table: 17 GB for a human genome), and whether
// it can be had faster: one hipMalloc, hipMalloc + hipMemset, the same size again after hipFree, several threads allocating parts
// at once, hipMallocAsync from a pool.  Diagnostic tool only; not part of the product path.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/alloc_probe tools/alloc_probe.hip -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(int argc, char **argv) {
    const size_t GB = 1ull << 30;
    const size_t total = (argc > 1 ? atoll(argv[1]) : 16) * GB;
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    void *p = nullptr;
    double t0 = now();
    CK(hipMalloc(&p, 64 << 20));
    printf("first hipMalloc (64 MB, runtime warm-up): %.1f ms\n", (now() - t0) * 1e3);
    CK(hipFree(p));
    const bool only_vmm = argc > 2 && std::string(argv[2]) == "vmm";  // the virtual-memory path alone, on memory nobody has touched yet
    if (!only_vmm)
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now();
        CK(hipMalloc(&p, total));
        double t1 = now();
        CK(hipMemset(p, 0, total));
        CK(hipDeviceSynchronize());
        double t2 = now();
        CK(hipMemset(p, 0, total));
        CK(hipDeviceSynchronize());
        double t3 = now();
        CK(hipFree(p));
        double t4 = now();
        printf("rep %d: hipMalloc(%zu GB) %.1f ms, first hipMemset %.1f ms, second hipMemset %.1f ms, hipFree %.1f ms\n", rep, total / GB, (t1 - t0) * 1e3, (t2 - t1) * 1e3,
               (t3 - t2) * 1e3, (t4 - t3) * 1e3);
    }
    if (!only_vmm)
    for (int nt : {2, 4, 8, 16}) {
        std::vector<void *> ps(nt, nullptr);
        std::vector<std::thread> th;
        t0 = now();
        for (int i = 0; i < nt; ++i)
            th.emplace_back([&, i] {
                hipSetDevice(0);
                hipMalloc(&ps[i], total / nt);
            });
        for (auto &t : th) t.join();
        double t1 = now();
        for (int i = 0; i < nt; ++i) hipMemsetAsync(ps[i], 0, total / nt, 0);
        CK(hipDeviceSynchronize());
        double t2 = now();
        for (int i = 0; i < nt; ++i) hipFree(ps[i]);
        printf("%2d threads x hipMalloc(%zu MB): %.1f ms, memset %.1f ms, free %.1f ms\n", nt, total / nt >> 20, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (now() - t2) * 1e3);
    }
    if (!only_vmm) {  // pool allocation
        hipMemPool_t pool;
        CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thresh = ~0ull;
        CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thresh));
        for (int rep = 0; rep < 2; ++rep) {
            t0 = now();
            CK(hipMallocAsync(&p, total, 0));
            CK(hipStreamSynchronize(0));
            double t1 = now();
            CK(hipMemsetAsync(p, 0, total, 0));
            CK(hipStreamSynchronize(0));
            double t2 = now();
            CK(hipFreeAsync(p, 0));
            CK(hipStreamSynchronize(0));
            printf("pool rep %d: hipMallocAsync %.1f ms, memset %.1f ms, hipFreeAsync %.1f ms\n", rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (now() - t2) * 1e3);
        }
    }
    {  // virtual memory API: reserve once, map physical chunks
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess) {
            printf("vmm granularity %zu\n", gran);
            t0 = now();
            hipMemGenericAllocationHandle_t h;
            hipError_t e = hipMemCreate(&h, total, &prop, 0);
            double t1 = now();
            printf("hipMemCreate(%zu GB): %s, %.1f ms\n", total / GB, hipGetErrorString(e), (t1 - t0) * 1e3);
            if (e == hipSuccess) {
                void *va = nullptr;
                double a0 = now();
                hipError_t e1 = hipMemAddressReserve(&va, total, 0, nullptr, 0);
                double a1 = now();
                hipError_t e2 = e1 == hipSuccess ? hipMemMap(va, total, 0, h, 0) : e1;
                double a2 = now();
                hipMemAccessDesc acc = {};
                acc.location = prop.location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                hipError_t e3 = e2 == hipSuccess ? hipMemSetAccess(va, total, &acc, 1) : e2;
                double a3 = now();
                printf("vmm: reserve %s %.1f ms, map %s %.1f ms, set access %s %.1f ms\n", hipGetErrorString(e1), (a1 - a0) * 1e3, hipGetErrorString(e2), (a2 - a1) * 1e3,
                       hipGetErrorString(e3), (a3 - a2) * 1e3);
                if (e3 == hipSuccess) {
                    double m0 = now();
                    hipError_t e4 = hipMemset(va, 0, total);
                    hipDeviceSynchronize();
                    double m1 = now();
                    hipMemset(va, 0, total);
                    hipDeviceSynchronize();
                    printf("vmm: first memset %s %.1f ms, second %.1f ms\n", hipGetErrorString(e4), (m1 - m0) * 1e3, (now() - m1) * 1e3);
                    hipMemUnmap(va, total);
                }
                if (e1 == hipSuccess) hipMemAddressFree(va, total);
                hipMemRelease(h);
            }
        }
    }
    return 0;
}
