#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const char *path = argv[1];
    size_t total = (size_t)atof(argv[2]) * (1u << 30);
    size_t batch = 128u << 20;
    char *src = (char *)malloc(batch);
    memset(src, 'A', batch);
    {
        int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        double t = now();
        for (size_t off = 0; off < total; off += batch) {
            size_t left = batch; const char *p = src;
            while (left) { ssize_t w = write(fd, p, left); if (w <= 0) return 1; left -= w; p += w; }
        }
        close(fd);
        printf("write() 1 thread: %.2f GB/s\n", total / (now() - t) / 1e9);
        unlink(path);
    }
    for (int T : {1, 2, 4, 8}) {
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t = now();
        for (size_t off = 0; off < total; off += batch) {
            if (ftruncate(fd, off + batch)) return 1;
            char *m = (char *)mmap(nullptr, batch, PROT_READ | PROT_WRITE, MAP_SHARED, fd, off);
            if (m == MAP_FAILED) return 1;
            std::vector<std::thread> th;
            for (int i = 0; i < T; i++) th.emplace_back([=] { size_t a = batch / T * i, b = i == T - 1 ? batch : batch / T * (i + 1); memcpy(m + a, src + a, b - a); });
            for (auto &x : th) x.join();
            munmap(m, batch);
        }
        close(fd);
        printf("mmap+memcpy %d threads: %.2f GB/s\n", T, total / (now() - t) / 1e9);
        unlink(path);
    }
    for (int T : {2, 4}) {  // parallel pwrite to one file
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t = now();
        for (size_t off = 0; off < total; off += batch) {
            std::vector<std::thread> th;
            for (int i = 0; i < T; i++) th.emplace_back([=] { size_t a = batch / T * i, b = i == T - 1 ? batch : batch / T * (i + 1); size_t o = a; while (o < b) { ssize_t w = pwrite(fd, src + o, b - o, off + o); if (w <= 0) break; o += w; } });
            for (auto &x : th) x.join();
        }
        close(fd);
        printf("pwrite %d threads: %.2f GB/s\n", T, total / (now() - t) / 1e9);
        unlink(path);
    }
    return 0;
}
