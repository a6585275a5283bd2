// pinned_read_bench.hip -- does read() into page-locked memory cost more than into malloc'ed memory?
//   hipcc -O2 tools/pinned_read_bench.hip -o tools/pinned_read_bench ; ./pinned_read_bench file
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double slurp(const char *path, char *buf, size_t cap, size_t chunk) {
    int fd = open(path, O_RDONLY);
    double t = now();
    size_t got = 0;
    for (;;) {
        size_t want = cap - got < chunk ? cap - got : chunk;
        if (!want) break;
        ssize_t n = read(fd, buf + got, want);
        if (n <= 0) break;
        got += n;
    }
    double dt = now() - t;
    close(fd);
    return got / dt / 1e9;
}
int main(int argc, char **argv) {
    size_t cap = 1u << 30;
    char *m = (char *)malloc(cap);
    memset(m, 1, cap);
    unsigned flags[] = {hipHostMallocDefault, hipHostMallocPortable, hipHostMallocNonCoherent, hipHostMallocNumaUser};
    const char *names[] = {"default", "portable", "noncoherent", "numa-user"};
    printf("malloc: %.2f GB/s, %.2f GB/s\n", slurp(argv[1], m, cap, 4u << 20), slurp(argv[1], m, cap, 4u << 20));
    for (int i = 0; i < 4; i++) {
        char *p = nullptr;
        double t = now();
        if (hipHostMalloc((void **)&p, cap, flags[i]) != hipSuccess) { printf("%s: alloc failed\n", names[i]); continue; }
        double ta = now() - t;
        printf("hipHostMalloc %-12s (alloc %.3f s = %.2f GB/s): read %.2f GB/s, %.2f GB/s; memcpy from malloc ", names[i], ta, cap / ta / 1e9,
               slurp(argv[1], p, cap, 4u << 20), slurp(argv[1], p, cap, 4u << 20));
        t = now();
        memcpy(p, m, cap);
        printf("%.2f GB/s; memchr scan ", cap / (now() - t) / 1e9);
        t = now();
        size_t c = 0;
        for (char *q = p; (q = (char *)memchr(q, '\n', p + cap - q)); q++) c++;
        printf("%.2f GB/s (%zu)\n", cap / (now() - t) / 1e9, c);
        hipHostFree(p);
    }
    {
        double t = now();
        hipHostRegister(m, cap, hipHostRegisterDefault);
        printf("hipHostRegister of malloc'ed 1 GiB: %.3f s; read %.2f GB/s\n", now() - t, slurp(argv[1], m, cap, 4u << 20));
    }
    return 0;
}
