// bin_bench.hip -- prototype with full cost accounting: do table look-ups get cheaper when they are BINNED by
// table region first, so that the lines of a region are fetched from HBM once and the look-ups that share
// them (4.3 per 128-byte line in a launch of 2.5 M read pairs against a 5.7 GB table) hit the L2?
// (profiles/r03_binning.txt holds the numbers and the go / no-go.)
//
//   pass D  direct: every lane probes its look-up where it falls (what k_classify_short does today)
//   pass A  every look-up becomes a 12-byte tuple (home cell, key, return slot) appended to the bin of its
//           table region -- one buffer per (XCD, bin), cursor bumped by an L2 atomic (workgroup scope: the
//           buffer belongs to one XCD), the tuples of a bin's open line merge in that XCD's write-back L2
//   pass B  the bins of an XCD are walked by all its workgroups together (one slice of the table at a time,
//           1 MB: it stays in the XCD's 4 MB L2), tuples are read back coalesced and probed
//   result  a look-up's value goes to its return slot; the prototype adds them up (checksum D == B) and
//           counts hits; a product would scatter only HITS (fragment, taxon, run length) in a third pass
//
//   hipcc --offload-arch=gfx950 -O3 tools/bin_bench.hip -o tools/bin_bench
//   ./bin_bench [million look-ups = 193.6] [bin_shift = 18] [table cells = 1431655765]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));             \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

constexpr uint32_t VBITS = 5, VMASK = 31;  // value bits of the synthetic table (bench.py: 30-node chain)
constexpr int NXCD = 8;

__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t mod_cap(uint64_t hc, uint64_t cap, uint64_t magic) {
    uint64_t q = __umul64hi(hc, magic);
    uint64_t r = hc - q * cap;
    if (r >= cap) r -= cap;
    if (r >= cap) r -= cap;
    return r;
}

// table with kraken2's insertion rule (linear probing, first empty cell or same key), constant value
__global__ void k_fill(uint32_t *table, uint64_t cap, uint64_t magic, uint64_t n_keys, uint64_t seed, uint32_t value) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_keys; i += stride) {
        const uint64_t hc = fmix64(splitmix64(seed + i) & ((1ull << 62) - 1));
        const uint32_t compacted = (uint32_t)(hc >> (32 + VBITS));
        const uint32_t cell = (compacted << VBITS) | value;
        uint64_t idx = mod_cap(hc, cap, magic);
        for (;;) {
            const uint32_t old = atomicCAS(&table[idx], 0u, cell);
            if (old == 0 || (old >> VBITS) == compacted) break;
            if (++idx >= cap) idx = 0;
        }
    }
}

// look-up i of the launch: half of them (hit_pct of 256) are keys that were inserted
__device__ __forceinline__ uint64_t lookup_hc(uint64_t i, uint64_t n_keys, uint32_t hit_256, uint64_t seed) {
    const uint64_t r = splitmix64(i * 0xD1B54A32D192ED03ull + 77);
    if ((uint32_t)(r & 255) < hit_256) return fmix64(splitmix64(seed + (r >> 8) % n_keys) & ((1ull << 62) - 1));
    return fmix64(r);
}

// CompactHashTable::Get from (home, ckey): 16-byte rounds that never leave the 128-byte line, as the product does
__device__ __forceinline__ uint32_t probe(const uint32_t *__restrict__ table, uint32_t cap, uint32_t pos, uint32_t ckey) {
    for (;;) {
        const uint32_t in_line = 32u - (pos & 31u);
        const uint32_t room = cap - pos;
        uint32_t nvalid = in_line < room ? in_line : room;
        if (nvalid > 4) nvalid = 4;
        const uint32_t lo = in_line < 4u ? 4u - in_line : 0u;
        const uint4 c = *reinterpret_cast<const uint4 *>(table + pos - lo);
        const uint32_t cells[4] = {c.x, c.y, c.z, c.w};
        uint32_t res = 0, resj = 64;
#pragma unroll
        for (int j = 3; j >= 0; j--) {
            const uint32_t x = cells[j] ^ ckey;
            const bool stop = ((uint32_t)j >= lo) & ((x <= VMASK) | ((cells[j] & VMASK) == 0));
            res = stop ? x : res;
            resj = stop ? (uint32_t)j : resj;
        }
        if (resj < lo + nvalid) return res <= VMASK ? res : 0u;
        pos += nvalid;
        if (pos >= cap) pos = 0;
    }
}

__device__ __forceinline__ void wave_add(unsigned long long *dst, unsigned long long v) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, v);
}

// ---- pass D ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_direct(const uint32_t *__restrict__ table, uint64_t cap, uint64_t magic,
                                                uint64_t n, uint64_t n_keys, uint32_t hit_256, uint64_t seed,
                                                unsigned long long *sums) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0, hits = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t hc = lookup_hc(i, n_keys, hit_256, seed);
        const uint32_t home = (uint32_t)mod_cap(hc, cap, magic);
        const uint32_t ckey = (uint32_t)(hc >> (32 + VBITS)) << VBITS;
        const uint32_t v = probe(table, (uint32_t)cap, home, ckey);
        acc += (unsigned long long)v * ((i & 1023) + 1);
        hits += v != 0;
    }
    wave_add(&sums[0], acc);
    wave_add(&sums[1], hits);
}

// ---- pass A ------------------------------------------------------------------------------------------
struct Bins {
    uint32_t *cursor;   // [NXCD][nbins]
    uint32_t *tuples;   // [NXCD][nbins][bin_cap][3]
    uint32_t nbins, bin_cap, bin_shift, pad;
    unsigned long long *overflow;
};

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}

template <int MODE>  // 0: generate only (what pass A costs without the binning), 1: atomics + stores
__global__ __launch_bounds__(256) void k_bin_write(Bins B, uint64_t cap, uint64_t magic, uint64_t n, uint64_t n_keys,
                                                   uint32_t hit_256, uint64_t seed, unsigned long long *sums) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t xcc = xcc_id();
    uint32_t *const cur = B.cursor + (size_t)xcc * B.nbins;
    uint32_t *const tup = B.tuples + (size_t)xcc * B.nbins * B.bin_cap * 3;
    unsigned long long over = 0, acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t hc = lookup_hc(i, n_keys, hit_256, seed);
        const uint32_t home = (uint32_t)mod_cap(hc, cap, magic);
        const uint32_t ckey = (uint32_t)(hc >> (32 + VBITS)) << VBITS;
        const uint32_t bin = home >> B.bin_shift;
        if (MODE == 0) {
            acc += home ^ ckey ^ bin;
            continue;
        }
        // the cursor is only ever touched from this XCD: an L2 atomic (workgroup scope), not a device-scope one
        const uint32_t idx = __hip_atomic_fetch_add(&cur[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (idx < B.bin_cap) {
            uint32_t *t = tup + ((size_t)bin * B.bin_cap + idx) * 3;
            t[0] = home;
            t[1] = ckey;
            t[2] = (uint32_t)(i & 0xFFFFFFFFu);
        } else {
            over++;
        }
    }
    if (MODE == 0) wave_add(&sums[2], acc);
    wave_add(B.overflow, over);
}

// ---- pass B ------------------------------------------------------------------------------------------
// Every workgroup finds its XCD and its rank among the workgroups there; the XCD's bins are walked in the
// same order by all of them, each taking a strided share of a bin's tuples (from all eight source buffers).
template <bool PREFETCH>
__global__ __launch_bounds__(256) void k_bin_probe(Bins B, const uint32_t *__restrict__ table, uint64_t cap, uint32_t *xcd_rank,
                                                   uint32_t wgs_per_xcd, unsigned long long *sums) {
    __shared__ uint32_t s_rank;
    const uint32_t xcc = xcc_id();
    if (threadIdx.x == 0) s_rank = atomicAdd(&xcd_rank[xcc], 1u);
    __syncthreads();
    const uint32_t rank = s_rank;
    if (rank >= wgs_per_xcd) return;  // (more workgroups landed here than planned: the others cover everything)
    unsigned long long acc = 0, hits = 0;
    const uint32_t nb = B.nbins;
    for (uint32_t bin = xcc; bin < nb; bin += NXCD) {
        if (PREFETCH && bin + NXCD < nb) {
            // touch this workgroup's share of the NEXT slice so that its lines are on their way
            const uint64_t s0 = (uint64_t)(bin + NXCD) << B.bin_shift;
            const uint64_t cells = 1ull << B.bin_shift;
            for (uint64_t c = ((uint64_t)rank * 256 + threadIdx.x) * 32; c < cells; c += (uint64_t)wgs_per_xcd * 256 * 32) {
                if (s0 + c < cap) acc += __builtin_nontemporal_load(table + s0 + c) & 0;  // keeps the load, adds nothing
            }
        }
        for (int src = 0; src < NXCD; src++) {
            const uint32_t cnt0 = B.cursor[(size_t)src * nb + bin];
            const uint32_t cnt = cnt0 < B.bin_cap ? cnt0 : B.bin_cap;
            const uint32_t *tup = B.tuples + ((size_t)src * nb + bin) * B.bin_cap * 3;
            for (uint32_t t = rank * 256 + threadIdx.x; t < cnt; t += wgs_per_xcd * 256) {
                const uint32_t home = tup[3 * (size_t)t], ckey = tup[3 * (size_t)t + 1], id = tup[3 * (size_t)t + 2];
                const uint32_t v = probe(table, (uint32_t)cap, home, ckey);
                acc += (unsigned long long)v * ((id & 1023) + 1);
                hits += v != 0;
            }
        }
    }
    wave_add(&sums[0], acc);
    wave_add(&sums[1], hits);
}

// ---- pass P: one level of a radix partition with LDS staging --------------------------------------------
// What a binned design pays per LEVEL beyond the first (whose reading half is the scan itself): read 12-byte
// tuples, split 64 ways by six bits of the home cell, write them back.  A workgroup takes 4096 tuples at a
// time: LDS histogram, ONE global atomic per non-empty bin and batch to reserve its run, tuples moved into
// bin order through LDS and written as runs (64 tuples = 768 bytes on average).
constexpr int P_BATCH = 4096, P_WAYS = 64;
__global__ __launch_bounds__(256) void k_partition64(const uint32_t *__restrict__ in, uint64_t n, uint32_t shift,
                                                     uint32_t *__restrict__ out, uint32_t *cursors, uint64_t cap_per_bin) {
    __shared__ uint32_t hist[P_WAYS], base_g[P_WAYS], base_l[P_WAYS];
    __shared__ uint32_t stage[P_BATCH * 3];
    const int tid = threadIdx.x;
    const uint64_t n_batches = (n + P_BATCH - 1) / P_BATCH;
    for (uint64_t b = blockIdx.x; b < n_batches; b += gridDim.x) {
        if (tid < P_WAYS) hist[tid] = 0;
        __syncthreads();
        uint32_t t0[16], t1[16], t2[16], slot[16];
        const uint64_t first = b * P_BATCH;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint64_t i = first + (uint64_t)j * 256 + tid;
            slot[j] = 0xFFFFFFFFu;
            if (i < n) {
                t0[j] = in[3 * i];
                t1[j] = in[3 * i + 1];
                t2[j] = in[3 * i + 2];
                slot[j] = atomicAdd(&hist[(t0[j] >> shift) & (P_WAYS - 1)], 1u);  // rank inside its bin (LDS atomic)
            }
        }
        __syncthreads();
        if (tid < P_WAYS) base_g[tid] = hist[tid] ? atomicAdd(&cursors[tid], hist[tid]) : 0u;
        if (tid == 0) {
            uint32_t acc = 0;
            for (int w = 0; w < P_WAYS; w++) {
                base_l[w] = acc;
                acc += hist[w];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (slot[j] != 0xFFFFFFFFu) {
                const uint32_t w = (t0[j] >> shift) & (P_WAYS - 1);
                const uint32_t p = base_l[w] + slot[j];
                stage[3 * p] = t0[j];
                stage[3 * p + 1] = t1[j];
                stage[3 * p + 2] = t2[j];
            }
        __syncthreads();
        // bin runs out: thread k of the batch order writes dwords of tuple k (runs are contiguous in `stage`)
        const uint32_t cnt = (uint32_t)((first + P_BATCH <= n) ? P_BATCH : n - first);
        for (uint32_t k = tid; k < cnt; k += 256) {
            // which bin holds staged tuple k: binary search over base_l
            uint32_t lo = 0, hi = P_WAYS - 1;
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                if (base_l[mid] <= k) lo = mid;
                else hi = mid - 1;
            }
            const uint64_t dst = (uint64_t)lo * cap_per_bin + base_g[lo] + (k - base_l[lo]);
            if (base_g[lo] + (k - base_l[lo]) < cap_per_bin) {
                out[3 * dst] = stage[3 * k];
                out[3 * dst + 1] = stage[3 * k + 1];
                out[3 * dst + 2] = stage[3 * k + 2];
            }
        }
        __syncthreads();
    }
}

// ---- pass B, second version ---------------------------------------------------------------------------
// The workgroups of an XCD form GROUPS; group g walks the XCD's bins g, g + G, ...: G slices of the table are
// live in the XCD's L2 at a time.  A bin's tuples (all eight source buffers, flattened) are spread over the
// threads of the group, and a thread loads its tuple of the NEXT bin before it probes the current one.
struct Tup {
    uint32_t home, ckey, id, ok;
};
__device__ __forceinline__ Tup load_tuple(const Bins &B, uint32_t bin, uint32_t q) {
    Tup t = {0, 0, 0, 0};
    if (bin >= B.nbins) return t;
    uint32_t acc = 0;
#pragma unroll
    for (int src = 0; src < NXCD; src++) {
        const uint32_t c0 = B.cursor[(size_t)src * B.nbins + bin];
        const uint32_t c = c0 < B.bin_cap ? c0 : B.bin_cap;
        if (!t.ok && q < acc + c) {
            const uint32_t *p = B.tuples + (((size_t)src * B.nbins + bin) * B.bin_cap + (q - acc)) * 3;
            t.home = p[0];
            t.ckey = p[1];
            t.id = p[2];
            t.ok = 1;
        }
        acc += c;
    }
    return t;
}
__device__ __forceinline__ uint32_t bin_total(const Bins &B, uint32_t bin) {
    uint32_t acc = 0;
    if (bin >= B.nbins) return 0;
#pragma unroll
    for (int src = 0; src < NXCD; src++) {
        const uint32_t c0 = B.cursor[(size_t)src * B.nbins + bin];
        acc += c0 < B.bin_cap ? c0 : B.bin_cap;
    }
    return acc;
}

__global__ __launch_bounds__(256) void k_bin_probe2(Bins B, const uint32_t *__restrict__ table, uint64_t cap, uint32_t *xcd_rank,
                                                    uint32_t wgs_per_xcd, uint32_t groups, unsigned long long *sums) {
    __shared__ uint32_t s_rank;
    const uint32_t xcc = xcc_id();
    if (threadIdx.x == 0) s_rank = atomicAdd(&xcd_rank[xcc], 1u);
    __syncthreads();
    const uint32_t rank = s_rank;
    if (rank >= wgs_per_xcd) return;
    const uint32_t wpg = wgs_per_xcd / groups;         // workgroups per group
    const uint32_t g = rank / wpg, r = rank % wpg;
    if (g >= groups) return;
    const uint32_t gthreads = wpg * 256, q0 = r * 256 + threadIdx.x;
    unsigned long long acc = 0, hits = 0;
    const uint32_t step = NXCD * groups;
    uint32_t bin = xcc + NXCD * g;
    Tup cur = load_tuple(B, bin, q0);
    for (; bin < B.nbins; bin += step) {
        const Tup nxt = load_tuple(B, bin + step, q0);  // in flight while this bin is probed
        if (cur.ok) {
            const uint32_t v = probe(table, (uint32_t)cap, cur.home, cur.ckey);
            acc += (unsigned long long)v * ((cur.id & 1023) + 1);
            hits += v != 0;
        }
        const uint32_t tot = bin_total(B, bin);
        for (uint32_t q = q0 + gthreads; q < tot; q += gthreads) {  // bins with more tuples than the group has threads
            const Tup t = load_tuple(B, bin, q);
            const uint32_t v = probe(table, (uint32_t)cap, t.home, t.ckey);
            acc += (unsigned long long)v * ((t.id & 1023) + 1);
            hits += v != 0;
        }
        cur = nxt;
    }
    wave_add(&sums[0], acc);
    wave_add(&sums[1], hits);
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) {
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main(int argc, char **argv) {
    const double mlook = argc > 1 ? atof(argv[1]) : 193.6;
    const uint32_t bin_shift = argc > 2 ? (uint32_t)atoi(argv[2]) : 18;
    const uint64_t cap = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1431655765ull;
    const uint32_t hit_256 = argc > 4 ? (uint32_t)atoi(argv[4]) : 0;  // look-ups of inserted keys, in 1/256
    const uint64_t n = (uint64_t)(mlook * 1e6);
    const uint64_t n_keys = (uint64_t)(cap * 0.7);
    const uint64_t magic = ~0ull / cap;
    const uint64_t seed = 20250101;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *table;
    CK(hipMalloc(&table, (cap + 64) * 4));
    CK(hipMemset(table, 0, (cap + 64) * 4));
    hipLaunchKernelGGL(k_fill, dim3(cus * 16), dim3(256), 0, 0, table, cap, magic, n_keys, seed, 30u);
    CK(hipDeviceSynchronize());
    Bins B;
    B.bin_shift = bin_shift;
    B.nbins = (uint32_t)((cap + (1ull << bin_shift) - 1) >> bin_shift);
    const double mean = (double)n / NXCD / B.nbins;
    B.bin_cap = (uint32_t)(mean * 1.4 + 6 * __builtin_sqrt((double)mean) + 64);
    B.pad = 0;
    CK(hipMalloc(&B.cursor, (size_t)NXCD * B.nbins * 4));
    CK(hipMalloc(&B.tuples, (size_t)NXCD * B.nbins * B.bin_cap * 12));
    CK(hipMalloc(&B.overflow, 8));
    unsigned long long *sums;
    CK(hipMalloc(&sums, 64));
    uint32_t *xcd_rank;
    CK(hipMalloc(&xcd_rank, 64));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&e2));
    printf("table %llu cells = %.2f GB at load 0.70, %.1f M look-ups (%.1f per 128-byte line), %u/256 of them hits;\n"
           "bins of %u cells = %.2f MB: %u bins x %d XCDs, %u tuples of 12 bytes each (%.2f GB of tuple buffers)\n",
           (unsigned long long)cap, cap * 4 / 1e9, n / 1e6, (double)n / (cap / 32.0), hit_256, 1u << bin_shift,
           (1u << bin_shift) * 4 / 1e6, B.nbins, NXCD, B.bin_cap, (double)NXCD * B.nbins * B.bin_cap * 12 / 1e9);
    unsigned long long h[8];
    // ---- D ----
    for (int wpc : {8, 16, 20, 32}) {  // resident waves per CU
        const int blocks = cus * wpc / 4;
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(sums, 0, 64));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_direct, dim3(blocks), dim3(256), 0, 0, table, cap, magic, n, n_keys, hit_256, seed, sums);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            best = time_ms(e0, e1) < best ? time_ms(e0, e1) : best;
        }
        CK(hipMemcpy(h, sums, 64, hipMemcpyDeviceToHost));
        printf("D  direct, %2d waves/CU:            %7.3f ms  %6.1f G look-ups/s   checksum %llx hits %llu\n", wpc, best,
               n / best / 1e6, h[0], h[1]);
    }
    const unsigned long long want_sum = h[0], want_hits = h[1];
    // ---- A ----
    for (int wpc : {16, 32}) {
        const int blocks = cus * wpc / 4;
        float best0 = 1e9, best1 = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(B.cursor, 0, (size_t)NXCD * B.nbins * 4));
            CK(hipMemset(B.overflow, 0, 8));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_bin_write<0>), dim3(blocks), dim3(256), 0, 0, B, cap, magic, n, n_keys, hit_256, seed, sums);
            CK(hipEventRecord(e1));
            hipLaunchKernelGGL((k_bin_write<1>), dim3(blocks), dim3(256), 0, 0, B, cap, magic, n, n_keys, hit_256, seed, sums);
            CK(hipEventRecord(e2));
            CK(hipEventSynchronize(e2));
            best0 = time_ms(e0, e1) < best0 ? time_ms(e0, e1) : best0;
            best1 = time_ms(e1, e2) < best1 ? time_ms(e1, e2) : best1;
        }
        CK(hipMemcpy(h, B.overflow, 8, hipMemcpyDeviceToHost));
        printf("A  bin write, %2d waves/CU:         %7.3f ms  %6.1f G tuples/s   (generation alone %.3f ms; %llu tuples did not fit)\n",
               wpc, best1, n / best1 / 1e6, best0, h[0]);
    }
    // ---- B ---- (bins as the last pass A left them)
    for (int pf = 0; pf < 2; pf++)
        for (int wpc : {20}) {
            const uint32_t wgs_per_xcd = (uint32_t)(cus / NXCD * wpc / 4);
            const int blocks = cus * wpc / 4;
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemset(sums, 0, 64));
                CK(hipMemset(xcd_rank, 0, 64));
                CK(hipEventRecord(e0));
                if (pf)
                    hipLaunchKernelGGL((k_bin_probe<true>), dim3(blocks), dim3(256), 0, 0, B, table, cap, xcd_rank, wgs_per_xcd, sums);
                else
                    hipLaunchKernelGGL((k_bin_probe<false>), dim3(blocks), dim3(256), 0, 0, B, table, cap, xcd_rank, wgs_per_xcd, sums);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                best = time_ms(e0, e1) < best ? time_ms(e0, e1) : best;
            }
            CK(hipMemcpy(h, sums, 64, hipMemcpyDeviceToHost));
            uint32_t ranks[16];
            CK(hipMemcpy(ranks, xcd_rank, 64, hipMemcpyDeviceToHost));
            printf("B  bin probe, %2d waves/CU%s: %7.3f ms  %6.1f G look-ups/s   checksum %s hits %s   (workgroups per XCD %u..%u of %u)\n",
                   wpc, pf ? ", prefetch" : "          ", best, n / best / 1e6, h[0] == want_sum ? "ok" : "DIFFERS",
                   h[1] == want_hits ? "ok" : "DIFFER", ranks[0], ranks[7], wgs_per_xcd);
        }
    // ---- B, second version ----
    for (uint32_t groups : {4u, 8u, 16u})
        for (int wpc : {16, 32}) {
            const uint32_t wgs_per_xcd = (uint32_t)(cus / NXCD * wpc / 4);
            const int blocks = cus * wpc / 4;
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemset(sums, 0, 64));
                CK(hipMemset(xcd_rank, 0, 64));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_bin_probe2, dim3(blocks), dim3(256), 0, 0, B, table, cap, xcd_rank, wgs_per_xcd, groups, sums);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                best = time_ms(e0, e1) < best ? time_ms(e0, e1) : best;
            }
            CK(hipMemcpy(h, sums, 64, hipMemcpyDeviceToHost));
            printf("B2 bin probe, %2d waves/CU, %2u groups per XCD (%4.1f MB of slices live): %7.3f ms  %6.1f G look-ups/s   checksum %s hits %s\n",
                   wpc, groups, groups * (double)(1u << bin_shift) * 4 / 1e6, best, n / best / 1e6,
                   h[0] == want_sum ? "ok" : "DIFFERS", h[1] == want_hits ? "ok" : "DIFFER");
        }
    // ---- P: one 64-way partition level over n tuples (input: random tuples) ----
    {
        uint32_t *tin, *tout, *cur;
        const uint64_t cap_per_bin = n / P_WAYS + n / P_WAYS / 8 + 65536;
        CK(hipMalloc(&tin, n * 12));
        CK(hipMalloc(&tout, cap_per_bin * P_WAYS * 12));
        CK(hipMalloc(&cur, P_WAYS * 4));
        // any 12-byte records will do: reuse the first n tuples' worth of the bin buffers (random homes)
        CK(hipMemcpy(tin, B.tuples, n * 12 < (size_t)NXCD * B.nbins * B.bin_cap * 12 ? n * 12 : (size_t)NXCD * B.nbins * B.bin_cap * 12,
                     hipMemcpyDeviceToDevice));
        for (int wpc : {8, 16, 20}) {
            const int blocks = cus * wpc / 4;
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemset(cur, 0, P_WAYS * 4));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_partition64, dim3(blocks), dim3(256), 0, 0, tin, n, 20u, tout, cur, cap_per_bin);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                best = time_ms(e0, e1) < best ? time_ms(e0, e1) : best;
            }
            printf("P  one 64-way partition level, %2d waves/CU: %7.3f ms  %6.1f G tuples/s  (%.2f TB/s of tuple traffic)\n", wpc, best,
                   n / best / 1e6, 2.0 * n * 12 / best / 1e9);
        }
    }
    return 0;
}
