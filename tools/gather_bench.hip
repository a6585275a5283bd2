// gather_bench.hip -- microbenchmark: how many random 16-byte probes per second can one MI355X
// sustain into a multi-GB table?  This is the practical ceiling of the CompactHashTable::Get step
// (one 64-B line per lookup in the roofline formula of BASELINE.md section 4).
//
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o tools/gather_bench
//   ./gather_bench [table_GiB=6] [loads_per_thread=64]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));             \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// ILP independent loads in flight per lane; WIDTH = bytes per load (4 or 16)
template <int ILP, int WIDTH>
__global__ __launch_bounds__(256) void k_gather(const uint32_t *__restrict__ table, uint64_t n_chunks,
                                                int iters, uint32_t *__restrict__ sink) {
    uint64_t s = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1);
    uint32_t acc = 0;
    for (int it = 0; it < iters; it += ILP) {
        uint64_t idx[ILP];
#pragma unroll
        for (int j = 0; j < ILP; j++) {
            s = mix(s + j + 1);
            idx[j] = __umul64hi(s, n_chunks);  // uniform in [0, n_chunks)
        }
        if (WIDTH == 16) {
            uint4 v[ILP];
#pragma unroll
            for (int j = 0; j < ILP; j++) v[j] = reinterpret_cast<const uint4 *>(table)[idx[j]];
#pragma unroll
            for (int j = 0; j < ILP; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        } else {
            uint32_t v[ILP];
#pragma unroll
            for (int j = 0; j < ILP; j++) v[j] = table[idx[j] * 4];
#pragma unroll
            for (int j = 0; j < ILP; j++) acc += v[j];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

static uint64_t n_chunks0(uint64_t bytes) { return bytes / 16; }

template <int ILP, int WIDTH>
static void run(const uint32_t *d_table, uint64_t n_chunks, int iters, uint32_t *d_sink, int blocks) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather<ILP, WIDTH>), dim3(blocks), dim3(256), 0, 0, d_table, n_chunks, iters, d_sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather<ILP, WIDTH>), dim3(blocks), dim3(256), 0, 0, d_table, n_chunks, iters, d_sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double n = (double)blocks * 256 * iters;
    printf("ILP=%d width=%2d blocks=%5d: %8.3f ms  %7.2f G probes/s  = %7.1f GB/s at 64 B/probe\n", ILP,
           WIDTH, blocks, ms, n / ms / 1e6, n * 64 / ms / 1e6);
}

// dependent chain: one lane, each load's address comes from the previous value (idle latency)
__global__ void k_chase(const uint32_t *__restrict__ table, uint64_t n_chunks, int iters, uint32_t *sink,
                        long long *cycles) {
    uint64_t s = 12345;
    uint32_t acc = 0;
    long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        s = mix(s + acc);
        uint64_t idx = __umul64hi(s, n_chunks);
        acc += table[idx * 4];
    }
    long long t1 = wall_clock64();
    sink[1] = acc;
    cycles[0] = t1 - t0;
}

template <int TPB>
__global__ __launch_bounds__(TPB) void k_gather1(const uint32_t *__restrict__ table, uint64_t n_chunks,
                                                 int iters, uint32_t *__restrict__ sink) {
    uint64_t s = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1);
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        s = mix(s + 1);
        const uint64_t idx = __umul64hi(s, n_chunks);
        const uint4 v = reinterpret_cast<const uint4 *>(table)[idx];
        acc += v.x ^ v.y ^ v.z ^ v.w;
        s += acc & 1;  // serialise: the next address depends on this load
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// cache-policy variants of the same dependent 16-B gather: does any of them make the L2 fetch less
// than a 128-B line from HBM?  MODE 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt, 5 sc0
template <int MODE>
__global__ __launch_bounds__(64) void k_gather_mode(const uint32_t *__restrict__ table, uint64_t n_chunks,
                                                    int iters, uint32_t *__restrict__ sink) {
    uint64_t s = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1);
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        s = mix(s + 1);
        const uint64_t idx = __umul64hi(s, n_chunks);
        const uint4 *p = reinterpret_cast<const uint4 *>(table) + idx;
        uint4 v;
        if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if (MODE == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        acc += v.x ^ v.y ^ v.z ^ v.w;
        s += acc & 1;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
static void run_mode(const char *name, const uint32_t *d_table, uint64_t n_chunks, uint32_t *d_sink) {
    const int iters = 256, waves = 8192;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather_mode<MODE>), dim3(waves), dim3(64), 0, 0, d_table, n_chunks, iters, d_sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather_mode<MODE>), dim3(waves), dim3(64), 0, 0, d_table, n_chunks, iters, d_sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  policy %-12s: %7.2f G probes/s\n", name, (double)waves * 64 * iters / ms / 1e6);
}


// Two probes per lane and round: one 16-B load at a random 64-B sector and a second one `stride`
// bytes further on (XOR 64 = the other half of the same
// 128-byte line).  Question: is the ~50 G/s ceiling a rate of 64-B requests, or of
// 128-byte lines / DRAM rows -- i.e. is a second sector next to the first one (almost) free?
// MODE 0: second = first ^ 64 (same 128-B line); 1: first + stride; 2: independent random sector.
template <int MODE>
__global__ __launch_bounds__(64) void k_pair(const uint32_t *__restrict__ table, uint64_t n_sectors,
                                             uint64_t stride, int iters, uint32_t *__restrict__ sink) {
    uint64_t s = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1);
    uint32_t acc = 0;
    const char *base = reinterpret_cast<const char *>(table);
    for (int it = 0; it < iters; it++) {
        s = mix(s + 1);
        const uint64_t a = __umul64hi(s, n_sectors) * 64;
        uint64_t b;
        if (MODE == 0) b = a ^ 64;
        else if (MODE == 1) { b = a + stride; if (b >= n_sectors * 64) b -= n_sectors * 64; }
        else b = __umul64hi(mix(s ^ 0x5555), n_sectors) * 64;
        const uint4 v = *reinterpret_cast<const uint4 *>(base + a);
        const uint4 w = *reinterpret_cast<const uint4 *>(base + b);
        acc += v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y ^ w.z ^ w.w;
        s += acc & 1;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
static void run_pair(const char *name, const uint32_t *d_table, uint64_t bytes, uint64_t stride, uint32_t *d_sink) {
    const int iters = 256;
    for (int waves : {4096, 8192, 16384}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k_pair<MODE>), dim3(waves), dim3(64), 0, 0, d_table, bytes / 64, stride, iters, d_sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_pair<MODE>), dim3(waves), dim3(64), 0, 0, d_table, bytes / 64, stride, iters, d_sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double pairs = (double)waves * 64 * iters;
        printf("  pair %-22s waves=%5d: %7.2f G pairs/s = %7.2f G sector requests/s\n", name, waves,
               pairs / ms / 1e6, 2 * pairs / ms / 1e6);
    }
}

// TA study (round 2): how fast does the vector memory pipeline take a wave-load whose lanes form groups of G
// consecutive lanes reading G x 16 contiguous bytes of one random line (G = 1: fully divergent)?  Small table
// (L2-resident) so that neither HBM nor the fabric is the limit.  ALIGNED: the group's bytes start on a
// (16 G)-byte boundary.
template <int G, bool ALIGNED>
__global__ __launch_bounds__(256) void k_groups(const uint32_t *__restrict__ table, uint64_t n_lines, int iters,
                                                uint32_t *__restrict__ sink) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t grp = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    uint64_t s = mix((uint64_t)grp * 0x9E3779B97F4A7C15ull + 1);
    uint32_t acc = 0;
    const char *base = reinterpret_cast<const char *>(table);
    for (int it = 0; it < iters; it++) {
        s = mix(s + 1);
        const uint64_t line = __umul64hi(s, n_lines);
        uint32_t off = (uint32_t)(s >> 3) & (128u - 16u * G) & ~15u;  // start of the group's bytes within the line
        if (ALIGNED) off &= ~(16u * G - 1u);
        const uint4 v = *reinterpret_cast<const uint4 *>(base + line * 128 + off + 16u * (lane % G));
        acc += v.x ^ v.y ^ v.z ^ v.w;
        s += acc & 1;  // dependent: one load in flight per lane
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int G, bool ALIGNED>
static void run_groups(const uint32_t *d_table, uint64_t bytes, uint32_t *d_sink) {
    const int iters = 2048, blocks = 256 * 5;  // 20 waves per CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_groups<G, ALIGNED>), dim3(blocks), dim3(256), 0, 0, d_table, bytes / 128, iters, d_sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_groups<G, ALIGNED>), dim3(blocks), dim3(256), 0, 0, d_table, bytes / 128, iters, d_sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double wl = (double)blocks * 4 * iters;  // wave-loads
    printf("  groups of %d lanes x 16 B %-9s: %7.2f G wave-loads/s chip = %6.1f cycles per wave-load per CU (2.4 GHz), %7.2f G groups/s\n",
           G, ALIGNED ? "aligned" : "unaligned", wl / ms / 1e6, 2.4e9 / (wl / ms * 1e3 / 256), wl * (64 / G) / ms / 1e6);
}

static void curve(const uint32_t *d_table, uint64_t n_chunks, uint32_t *d_sink) {
    printf("latency/throughput curve: every lane keeps exactly ONE dependent 16-B probe in flight\n");
    const int iters = 256;
    for (int waves : {256, 512, 1024, 2048, 4096, 8192, 16384, 32768}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k_gather1<64>), dim3(waves), dim3(64), 0, 0, d_table, n_chunks, iters, d_sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_gather1<64>), dim3(waves), dim3(64), 0, 0, d_table, n_chunks, iters, d_sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double lanes = (double)waves * 64;
        double rate = lanes * iters / ms / 1e6;  // G probes/s
        // resident lanes are capped by the chip (256 CUs x 32 waves x 64)
        double resident = lanes < 256.0 * 32 * 64 ? lanes : 256.0 * 32 * 64;
        printf("  waves=%6d lanes=%8.0f: %7.2f G probes/s, latency per probe ~ %6.2f us\n", waves, lanes, rate,
               resident / (rate * 1e3));
    }
}

int main(int argc, char **argv) {
    double gib = argc > 1 ? atof(argv[1]) : 6.0;
    int iters = argc > 2 ? atoi(argv[2]) : 64;
    uint64_t bytes = (uint64_t)(gib * (1ull << 30)) & ~15ull;
    uint32_t *d_table, *d_sink;
    CK(hipMalloc((void **)&d_table, bytes));
    CK(hipMalloc((void **)&d_sink, 64));
    CK(hipMemset(d_table, 1, bytes));
    uint64_t n_chunks = bytes / 16;
    printf("table %.2f GiB, %d loads per thread\n", gib, iters);



    if (argc > 3 && atoi(argv[3]) == 3) {  // memory-type study: does an uncached / fine-grained table fetch less than 128 B per probe?
        static const unsigned flags[] = {0u, hipDeviceMallocFinegrained, hipDeviceMallocUncached};
        static const char *names[] = {"hipMalloc (default)", "hipDeviceMallocFinegrained", "hipDeviceMallocUncached"};
        for (int f = 0; f < 3; f++) {
            uint32_t *t = nullptr;
            hipError_t e = f == 0 ? hipMalloc((void **)&t, bytes) : hipExtMallocWithFlags((void **)&t, bytes, flags[f]);
            if (e != hipSuccess) {
                printf("%s: allocation failed (%s)\n", names[f], hipGetErrorString(e));
                (void)hipGetLastError();
                continue;
            }
            CK(hipMemset(t, 1, bytes));
            printf("%s:\n", names[f]);
            run_groups<1, true>(t, bytes, d_sink);
            run_groups<4, true>(t, bytes, d_sink);
            CK(hipFree(t));
        }
        return 0;
    }
    if (argc > 3 && atoi(argv[3]) == 2) {  // TA study: table should be small (L2-resident), e.g. 0.002 GiB
        printf("TA study: dependent 16-byte loads, lanes grouped on contiguous bytes of one random line\n");
        run_groups<1, true>(d_table, bytes, d_sink);
        run_groups<2, true>(d_table, bytes, d_sink);
        run_groups<2, false>(d_table, bytes, d_sink);
        run_groups<4, true>(d_table, bytes, d_sink);
        run_groups<4, false>(d_table, bytes, d_sink);
        run_groups<8, true>(d_table, bytes, d_sink);
        return 0;
    }
    if (argc > 3 && atoi(argv[3]) == 1) {  // pair study only
        printf("pair study: two 16-B probes per lane per round (one dependent round in flight per lane)\n");
        run_pair<2>("random + random", d_table, bytes, 0, d_sink);
        run_pair<0>("same 128-B line (^64)", d_table, bytes, 0, d_sink);
        char nm[64];
        for (uint64_t st : {64ull, 128ull, 256ull, 512ull, 1024ull, 2048ull, 4096ull, 65536ull, 2097152ull}) {
            snprintf(nm, sizeof nm, "+%llu B", (unsigned long long)st);
            run_pair<1>(nm, d_table, bytes, st, d_sink);
        }
        return 0;
    }
    {
        long long *d_cyc, cyc = 0;
        CK(hipMalloc((void **)&d_cyc, 8));
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, d_table, n_chunks0(bytes), 2000, d_sink, d_cyc);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, d_table, n_chunks0(bytes), 2000, d_sink, d_cyc);
        CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        printf("idle dependent-chain latency: %.1f ticks of the 100 MHz wall clock per probe = %.2f us\n",
               cyc / 2000.0, cyc / 2000.0 / 100.0);
        curve(d_table, n_chunks0(bytes), d_sink);
        printf("cache-policy variants (8192 waves, one dependent probe per lane):\n");
        run_mode<0>("plain", d_table, n_chunks0(bytes), d_sink);
        run_mode<1>("nt", d_table, n_chunks0(bytes), d_sink);
        run_mode<2>("sc1", d_table, n_chunks0(bytes), d_sink);
        run_mode<3>("sc0 sc1", d_table, n_chunks0(bytes), d_sink);
        run_mode<4>("sc0 sc1 nt", d_table, n_chunks0(bytes), d_sink);
        run_mode<5>("sc0", d_table, n_chunks0(bytes), d_sink);
    }
    for (int blocks : {256 * 4, 256 * 8, 256 * 16}) {
        run<1, 16>(d_table, n_chunks, iters, d_sink, blocks);
        run<2, 16>(d_table, n_chunks, iters, d_sink, blocks);
        run<4, 16>(d_table, n_chunks, iters, d_sink, blocks);
        run<8, 16>(d_table, n_chunks, iters, d_sink, blocks);
        run<4, 4>(d_table, n_chunks, iters, d_sink, blocks);
    }
    return 0;
}
