/* tools/thp_probe.c -- diagnostic: is a large anonymous mapping that is filled by pread() huge-page backed on this box, and what does
 * unmapping it cost?  (The native driver reads the reference FASTA into such a buffer: ref_loader.hpp.)
 *   gcc -O2 -o tools/bin/thp_probe tools/thp_probe.c -lpthread;  tools/bin/thp_probe <file> <threads> */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static int fd; static char *buf; static size_t size; static volatile long next_blk; static const size_t BLK = 16u << 20;
static void *work(void *a) {
    (void)a;
    for (;;) {
        long b = __sync_fetch_and_add(&next_blk, 1);
        size_t lo = (size_t)b * BLK; if (lo >= size) break;
        size_t n = size - lo < BLK ? size - lo : BLK, got = 0;
        while (got < n) { ssize_t r = pread(fd, buf + lo + got, n - got, (off_t)(lo + got)); if (r <= 0) return 0; got += (size_t)r; }
    }
    return 0;
}
static void huge(const char *tag) {
    FILE *f = fopen("/proc/self/smaps_rollup", "r"); char l[256];
    while (f && fgets(l, 256, f)) if (strstr(l, "AnonHuge") || strstr(l, "Rss:")) printf("  %s: %s", tag, l);
    if (f) fclose(f);
}
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    fd = open(argv[1], O_RDONLY); struct stat st; fstat(fd, &st); size = (size_t)st.st_size; int T = atoi(argv[2]);
    for (int adv = 1; adv >= 0; --adv) {
        size_t mapped = (size + (2u << 20) - 1) / (2u << 20) * (2u << 20);
        double t0 = now();
        buf = mmap(0, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (adv) madvise(buf, mapped, MADV_HUGEPAGE);
        next_blk = 0; pthread_t th[64];
        for (int t = 0; t < T; ++t) pthread_create(&th[t], 0, work, 0);
        for (int t = 0; t < T; ++t) pthread_join(th[t], 0);
        double t1 = now();
        huge(adv ? "MADV_HUGEPAGE" : "plain");
        double t2 = now();
        munmap(buf, mapped);
        double t3 = now();
        printf("%s: %zu MB read by %d threads in %.3f s (%.1f GB/s), munmap %.3f s\n", adv ? "MADV_HUGEPAGE" : "plain", size >> 20, T, t1 - t0, size / (t1 - t0) / 1e9, t3 - t2);
    }
    printf("thp enabled: "); fflush(stdout); system("cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag");
    return 0;
}
