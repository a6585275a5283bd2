// write_bench -- what writing nh_run's plain outputs costs on a box, and whether MORE THREADS per file help (VERDICT r5 item 4:
// "positional parallel writes").  nh_run writes each output file from one thread with writev() (two files: two threads).
//   write_bench <dir> [GB per file = 4]
// Prints GB/s for: one file from one thread (write); TWO files, one thread each (what nh_run does); one file from T threads
// with pwrite() at disjoint offsets; one file grown with ftruncate() and filled through mmap() by T threads; the same two ways
// on two files at once.  tmpfs (and ext4 / xfs buffered writes) take the inode's lock exclusively for a write, so pwrite()
// from several threads into ONE file serialises; page faults into a shared mapping do not take it, but pay a fault and a
// zeroed page per 4 KiB.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const size_t BATCH = 96u << 20;  // one batch of nh_run: ~45 MB a file at 150 bp, up to 512 MB for long reads
static char *src;

static void one_file(const std::string &path, size_t total, int T, int how) {  // how 0: write / pwrite, 1: mmap
    int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    for (size_t off = 0; off < total; off += BATCH) {
        if (how == 0 && T == 1) {
            size_t left = BATCH;
            const char *p = src;
            while (left) {
                ssize_t w = write(fd, p, left);
                if (w <= 0) exit(1);
                left -= (size_t)w;
                p += w;
            }
            continue;
        }
        char *m = nullptr;
        if (how == 1) {
            if (ftruncate(fd, (off_t)(off + BATCH))) exit(1);
            m = (char *)mmap(nullptr, BATCH, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)off);
            if (m == MAP_FAILED) exit(1);
        }
        std::vector<std::thread> th;
        for (int i = 0; i < T; i++)
            th.emplace_back([=] {
                const size_t a = BATCH / (size_t)T * (size_t)i, b = i == T - 1 ? BATCH : BATCH / (size_t)T * (size_t)(i + 1);
                if (how == 1) {
                    memcpy(m + a, src + a, b - a);
                    return;
                }
                size_t o = a;
                while (o < b) {
                    ssize_t w = pwrite(fd, src + o, b - o, (off_t)(off + o));
                    if (w <= 0) exit(1);
                    o += (size_t)w;
                }
            });
        for (auto &x : th) x.join();
        if (m) munmap(m, BATCH);
    }
    close(fd);
}

static double run(const std::string &dir, size_t total, int files, int T, int how) {
    std::vector<std::string> paths;
    for (int f = 0; f < files; f++) paths.push_back(dir + "/wb_" + std::to_string(f) + ".bin");
    const double t = now();
    std::vector<std::thread> th;
    for (int f = 1; f < files; f++) th.emplace_back(one_file, paths[(size_t)f], total, T, how);
    one_file(paths[0], total, T, how);
    for (auto &x : th) x.join();
    const double dt = now() - t;
    for (auto &p : paths) unlink(p.c_str());
    return (double)total * files / dt / 1e9;
}

int main(int argc, char **argv) {
    if (argc < 2) return fprintf(stderr, "usage: write_bench <dir> [GB per file]\n"), 2;
    const std::string dir = argv[1];
    const size_t total = (size_t)((argc > 2 ? atof(argv[2]) : 4.0) * (double)(1u << 30)) / BATCH * BATCH;
    src = (char *)malloc(BATCH);
    memset(src, 'A', BATCH);
    run(dir, total / 4, 1, 1, 0);  // (warm-up: the first pages of a fresh tmpfs come slower)
    printf("%.1f GB per file, batches of %zu MB, GB/s (all files together)\n", (double)total / 1e9, BATCH >> 20);
    printf("  %-34s %6.2f\n", "1 file,  write(), 1 thread", run(dir, total, 1, 1, 0));
    printf("  %-34s %6.2f   <- what nh_run does\n", "2 files, write(), 1 thread each", run(dir, total, 2, 1, 0));
    for (int T : {2, 4, 8}) {
        char name[64];
        snprintf(name, sizeof name, "1 file,  pwrite(), %d threads", T);
        printf("  %-34s %6.2f\n", name, run(dir, total, 1, T, 0));
    }
    for (int T : {2, 4, 8}) {
        char name[64];
        snprintf(name, sizeof name, "2 files, pwrite(), %d threads each", T);
        printf("  %-34s %6.2f\n", name, run(dir, total, 2, T, 0));
    }
    for (int T : {1, 2, 4, 8}) {
        char name[64];
        snprintf(name, sizeof name, "1 file,  mmap + memcpy, %d threads", T);
        printf("  %-34s %6.2f\n", name, run(dir, total, 1, T, 1));
    }
    for (int T : {2, 4, 8}) {
        char name[64];
        snprintf(name, sizeof name, "2 files, mmap + memcpy, %d each", T);
        printf("  %-34s %6.2f\n", name, run(dir, total, 2, T, 1));
    }
    return 0;
}
