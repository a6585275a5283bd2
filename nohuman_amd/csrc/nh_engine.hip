// nh_engine.hip -- host side of libnohuman_engine.so: database images -> HBM, batch staging,
// and the extern "C" ABI declared in include/nohuman_engine.h.
//
// Reference units replaced (file:line under /root/reference): the kraken2 process boundary
// src/lib.rs:22-48 / src/main.rs:270; DB directory contract src/lib.rs:119-141; the kraken2 units
// behind it are external (pinned Dockerfile:15,35-38) and specified in SURVEY.md Appendix A.
// No CPU fallback exists in this file: without a HIP device every entry fails with NH_EDEVICE.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <algorithm>
#include <thread>
#include <unistd.h>
#include <fcntl.h>

#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "nh_device.h"
#include <map>

#include "nh_internal.h"
#include "nohuman_engine.h"

namespace nh {

// ---- logical devices (nh_internal.h) ----------------------------------------------------------------------------------
namespace {
int fake_devices() {
    static const int n = getenv("NOHUMAN_FAKE_DEVICES") ? std::max(0, atoi(getenv("NOHUMAN_FAKE_DEVICES"))) : 0;
    return n;
}
thread_local int tl_ldev = -1;
std::mutex g_dev_mu;
std::string g_dev_violation;
struct DevAlloc {
    size_t bytes;
    int ldev;
};
std::map<uintptr_t, DevAlloc> g_dev_allocs;  // debug mode: device allocations of this library, by start address
void dev_register(const void *p, size_t bytes, int ldev) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    const uintptr_t a = (uintptr_t)p;
    // (what overlaps a fresh allocation was freed in the meantime)
    auto it = g_dev_allocs.lower_bound(a);
    if (it != g_dev_allocs.begin()) {
        auto pr = std::prev(it);
        if (pr->first + pr->second.bytes > a) it = pr;
    }
    while (it != g_dev_allocs.end() && it->first < a + bytes) it = g_dev_allocs.erase(it);
    g_dev_allocs[a] = {bytes, ldev};
}
}  // namespace
bool dev_debug() {
    static const bool on = getenv("NOHUMAN_DEBUG_DEVICE") && getenv("NOHUMAN_DEBUG_DEVICE")[0] != '0';
    return on;
}
int dev_count() {
    if (fake_devices() > 0) return fake_devices();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
int dev_phys(int ldev) { return fake_devices() > 0 ? 0 : ldev; }
hipError_t dev_set(int ldev) {
    if (ldev < 0 || (fake_devices() > 0 && ldev >= fake_devices())) return hipErrorInvalidDevice;
    const hipError_t e = hipSetDevice(dev_phys(ldev));
    if (e == hipSuccess) tl_ldev = ldev;
    return e;
}
int dev_current() { return tl_ldev; }
static void dev_violate(const std::string &m) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (g_dev_violation.empty()) {
        g_dev_violation = m;
        fprintf(stderr, "nohuman: DEVICE DISCIPLINE: %s\n", m.c_str());
    }
}
std::string dev_violation(bool clear) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    std::string m = g_dev_violation;
    if (clear) g_dev_violation.clear();
    return m;
}
void dev_check(int owner, const char *where) {
    if (!dev_debug()) return;
    int hip_dev = -1;
    (void)hipGetDevice(&hip_dev);
    if (tl_ldev != owner || hip_dev != dev_phys(owner))
        dev_violate(std::string(where) + ": the thread's device is " + std::to_string(tl_ldev) + " (HIP " + std::to_string(hip_dev) +
                    "), the work belongs to device " + std::to_string(owner));
}
void dev_check_ptr(const void *p, int owner, const char *where) {
    if (!dev_debug() || !p) return;
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        const uintptr_t a = (uintptr_t)p;
        auto it = g_dev_allocs.upper_bound(a);
        if (it != g_dev_allocs.begin()) {
            --it;
            if (a < it->first + it->second.bytes && it->second.ldev != owner) {
                const int have = it->second.ldev;
                g_dev_mu.unlock();
                dev_violate(std::string(where) + ": a buffer of device " + std::to_string(have) + " is used as one of device " + std::to_string(owner));
                g_dev_mu.lock();
                return;
            }
        }
    }
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return;  // (not known to the runtime: pageable host memory)
    }
    if (at.type == hipMemoryTypeDevice && at.device != dev_phys(owner))
        dev_violate(std::string(where) + ": the buffer lies on HIP device " + std::to_string(at.device) + ", its owner is device " +
                    std::to_string(owner) + " (HIP " + std::to_string(dev_phys(owner)) + ")");
}
hipError_t dev_copy_between(void *dst, int dst_ldev, const void *src, int src_ldev, size_t n, hipStream_t stream) {
    if (!n) return hipSuccess;
    dev_check(dst_ldev, "dev_copy_between");
    dev_check_ptr(dst, dst_ldev, "dev_copy_between (destination)");
    dev_check_ptr(src, src_ldev, "dev_copy_between (source)");
    if (dst_ldev == src_ldev) return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, stream);
    static const bool no_peer = getenv("NOHUMAN_NO_PEER") && getenv("NOHUMAN_NO_PEER")[0] != '0';
    const int pd = dev_phys(dst_ldev), ps = dev_phys(src_ldev);
    if (no_peer) {
        // over the host: D2H on the source's device, H2D on the destination's (page-locked staging, one piece at a time)
        const size_t piece = std::min<size_t>(n, (size_t)64u << 20);
        void *h = nullptr;
        hipError_t e = host_malloc(&h, piece);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);  // (what the stream had queued before this copy comes first)
        for (size_t off = 0; off < n && e == hipSuccess; off += piece) {
            const size_t m = std::min(piece, n - off);
            e = dev_set(src_ldev);
            if (e == hipSuccess) e = hipMemcpy(h, (const char *)src + off, m, hipMemcpyDeviceToHost);
            const hipError_t e2 = dev_set(dst_ldev);
            if (e == hipSuccess) e = e2;
            if (e == hipSuccess) e = hipMemcpy((char *)dst + off, h, m, hipMemcpyHostToDevice);
        }
        (void)dev_set(dst_ldev);
        (void)hipHostFree(h);
        return e;
    }
    if (pd != ps) {
        static std::mutex mu;
        static std::vector<std::pair<int, int>> tried;
        std::lock_guard<std::mutex> lk(mu);
        if (std::find(tried.begin(), tried.end(), std::make_pair(pd, ps)) == tried.end()) {
            tried.push_back({pd, ps});
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, pd, ps) == hipSuccess && can) {
                const hipError_t pe = hipDeviceEnablePeerAccess(ps, 0);  // (the current device is pd: dev_check above)
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                    fprintf(stderr, "nohuman: WARN peer access from GPU %d to GPU %d could not be enabled (%s); copies between them are staged by the runtime\n",
                            pd, ps, hipGetErrorString(pe));
            } else if (getenv("NOHUMAN_TRACE")) {
                fprintf(stderr, "[nohuman trace] no peer access from GPU %d to GPU %d: copies between them are staged by the runtime\n", pd, ps);
            }
            (void)hipGetLastError();
        }
    }
    return hipMemcpyPeerAsync(dst, pd, src, ps, n, stream);
}

static hipError_t alloc_retry(void **p, size_t bytes, bool host, unsigned flags) {
    hipError_t e = host ? hipHostMalloc(p, bytes, flags) : hipMalloc(p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        const int dev = dev_current();
        run_cache_trim();
        dev_cache_trim();  // (frees on every device it holds buffers of)
        if (dev >= 0) (void)dev_set(dev);
        e = host ? hipHostMalloc(p, bytes, flags) : hipMalloc(p, bytes);
        if (e != hipSuccess) (void)hipGetLastError();
    }
    if (e == hipSuccess && !host && dev_debug()) dev_register(*p, bytes, dev_current());
    return e;
}
hipError_t dev_malloc(void **p, size_t bytes) { return alloc_retry(p, bytes, false, 0); }
hipError_t host_malloc(void **p, size_t bytes, unsigned flags) { return alloc_retry(p, bytes, true, flags); }


thread_local std::string g_last_error;

int set_error(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return set_error(_e == hipErrorOutOfMemory ? NH_EOOM : NH_EDEVICE, "%s: %s", #expr, \
                             hipGetErrorString(_e));                                         \
    } while (0)

static uint64_t rd64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}

static int check_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_error(NH_EDEVICE, "no HIP device available (%s)",
                         e == hipSuccess ? "count 0" : hipGetErrorString(e));
    n = dev_count();  // (logical devices: nh_internal.h)
    if (device < 0 || device >= n)
        return set_error(NH_EDEVICE, "device %d out of range (have %d)", device, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev_phys(device)));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(NH_EDEVICE, "device %d is %s; this library is built for gfx950 only",
                         device, prop.gcnArchName);
    return NH_OK;
}

static int parse_opts(Engine *e, const void *opts, size_t opts_len) {
    uint8_t ob[64];
    memset(ob, 0, sizeof ob);
    memcpy(ob, opts, opts_len < 64 ? opts_len : 64);
    e->opts_image.assign((const uint8_t *)opts, (const uint8_t *)opts + opts_len);
    nh_db_info &i = e->info;
    i.k = rd64(ob);
    i.l = rd64(ob + 8);
    i.spaced_seed_mask = rd64(ob + 16);
    i.toggle_mask = rd64(ob + 24);
    i.dna_db = ob[32];
    i.minimum_acceptable_hash_value = rd64(ob + 40);
    memcpy(&i.revcom_version, ob + 48, 4);
    if (!i.dna_db) return set_error(NH_EDB, "opts.k2d: protein databases are not supported");
    if (i.l == 0 || i.l > 31 || i.l > i.k)
        return set_error(NH_EDB, "opts.k2d: unsupported k=%llu l=%llu", (unsigned long long)i.k,
                         (unsigned long long)i.l);
    if (i.k - i.l > 64)
        return set_error(NH_EDB, "opts.k2d: k-l=%llu exceeds the 64 supported by the tile layout",
                         (unsigned long long)(i.k - i.l));
    return NH_OK;
}

static int parse_taxonomy(Engine *e, const void *taxo, size_t taxo_len) {
    const uint8_t *tb = (const uint8_t *)taxo;
    if (taxo_len < 32 || memcmp(tb, "K2TAXDAT", 8) != 0)
        return set_error(NH_EDB, "taxo.k2d: bad magic");
    uint64_t nc = rd64(tb + 8), nl = rd64(tb + 16), rl = rd64(tb + 24);
    if (nc == 0 || nc > 0xFFFFFFF0ull || taxo_len != 32 + 56 * nc + nl + rl)
        return set_error(NH_EDB, "taxo.k2d: size does not match its header");
    e->taxo_image.assign(tb, tb + taxo_len);
    e->info.node_count = nc;
    e->parent.resize(nc);
    e->external.resize(nc);
    for (uint64_t i = 0; i < nc; i++) {
        const uint8_t *n = tb + 32 + 56 * i;
        uint64_t p = rd64(n);
        // (the kernels climb with `while (b > a) b = parent[b]`: every parent below its child, the root's parent 0 --
        //  a root that names another node as its parent would be a loop on the GPU.  Node 0 is kraken2's unused dummy.)
        if (i >= 1 && p >= i) return set_error(NH_EDB, "taxo.k2d: parent id %llu of node %llu is not below it", (unsigned long long)p, (unsigned long long)i);
        e->parent[i] = i == 0 ? 0u : (uint32_t)p;
        e->external[i] = rd64(n + 40);
        // what the report writer (-r, nh_run.hip) walks on the host: the children's range and the two string offsets
        const uint64_t first = rd64(n + 8), cnt = rd64(n + 16), name_off = rd64(n + 24), rank_off = rd64(n + 32);
        if (cnt && (first <= i || first >= nc || cnt > nc - first))
            return set_error(NH_EDB, "taxo.k2d: children [%llu, +%llu) of node %llu are not nodes behind it", (unsigned long long)first,
                             (unsigned long long)cnt, (unsigned long long)i);
        if ((nl && name_off >= nl) || (rl && rank_off >= rl) || (!nl && name_off) || (!rl && rank_off))
            return set_error(NH_EDB, "taxo.k2d: name / rank offset of node %llu lies outside the string tables", (unsigned long long)i);
    }
    // (the two string tables are read as C strings)
    if ((nl && tb[32 + 56 * nc + nl - 1] != 0) || (rl && tb[32 + 56 * nc + nl + rl - 1] != 0))
        return set_error(NH_EDB, "taxo.k2d: a string table does not end with a NUL byte");
    return NH_OK;
}

// the probe queue packs (home cell << key_bits | compacted key) into 63 bits
static int check_queue_packing(Engine *e) {
    uint32_t cap_bits = 0;
    while (cap_bits < 64 && (e->info.capacity >> cap_bits) != 0) cap_bits++;
    if (cap_bits + e->info.key_bits > 63)
        return set_error(NH_EDB, "hash.k2d: capacity (%u bits) + key_bits (%llu) exceeds the 63 bits "
                         "of a probe-queue entry", cap_bits, (unsigned long long)e->info.key_bits);
    return NH_OK;
}

static void finish_devdb(Engine *e) {
    DevDB &d = e->dev;
    const nh_db_info &i = e->info;
    d.table = e->d_table;
    d.copy_stride = e->copy_stride;
    d.n_copies = e->n_copies;
    d.copy_shift = e->n_copies == 8 ? 2u : e->n_copies == 4 ? 3u : e->n_copies == 2 ? 4u : 5u;
    d.capacity = i.capacity;
    d.cap_magic = ~0ull / i.capacity;
    d.parent = e->d_parent;
    d.node_count = (uint32_t)i.node_count;
    d.value_bits = (uint32_t)i.value_bits;
    d.vmask = (uint32_t)((1ull << i.value_bits) - 1);
    d.k = (uint32_t)i.k;
    d.l = (uint32_t)i.l;
    d.window = (uint32_t)(i.k - i.l);
    d.lmer_mask = (1ull << (2 * i.l)) - 1;
    d.spaced_mask = i.spaced_seed_mask ? i.spaced_seed_mask : ~0ull;
    d.toggle = i.toggle_mask & d.lmer_mask;
    d.min_hash = i.minimum_acceptable_hash_value;
    d.revcom_version = i.revcom_version;
    d.linear_probing = e->options.linear_probing;
    d.reset_per_mate = e->options.reset_per_mate;
    d.min_hit_groups = e->options.minimum_hit_groups;
    d.ambig_rule = e->options.ambiguity_rule == NH_AMBIGUITY_LAST_LMER ? 0 : 1;  // (the kernels' numbering: 0 last l-mer, 1 queue)
    // bound of the probe loop: every round advances by at least one cell
    d.max_chunks = i.capacity + 1 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(i.capacity + 1);
}

void finish_devdb_public(Engine *e) { finish_devdb(e); }

static int alloc_table(Engine *e, uint64_t capacity) {
    // padded so that 16-byte chunk loads at the end of the table stay in bounds
    e->table_cells_alloc = ((capacity + 3) & ~3ull) + 32;
    // Staggered copies (DevDB::copy_stride): the gather ceiling of the chip is a rate of 128-byte lines,
    // and a probe run that starts in the first half of its line leaves it 3.5x less often than one
    // that starts anywhere (tools/probe_cost_model.py: 1.158 -> 1.045 lines per lookup at load 0.7).  HBM is not the scarce resource here (4 x 5.7 GB on a 288 GB device): a table too
    // large for four copies gets as many as fit (the loop below halves on out-of-memory).  NOHUMAN_TABLE_COPIES=1|2|4|8.
    uint32_t want = 4;  // with quad probing: 787 / 964 / 1000 Mreads/s for 1 / 2 / 4 copies (profiles/r02_tuning.txt)
    if (const char *env = getenv("NOHUMAN_TABLE_COPIES")) {
        const int v = atoi(env);
        want = v >= 8 ? 8u : v >= 4 ? 4u : v >= 2 ? 2u : 1u;
    }
    // Tables of 2^32 - 256 cells and more: only the quad-probing rounds (default geometry, linear probing) read the copies
    // through 64-bit positions; the per-lane rounds every other configuration takes read copy 0 alone (three
    // copies of >= 16 GiB allocated and refreshed for nothing, and the batch buffers or the gzip encoder starved)
    {
        const nh_db_info &i = e->info;
        const bool std_geom = i.k == 35 && i.l == 31 && i.revcom_version != 0 && i.minimum_acceptable_hash_value == 0;
        bool quad = std_geom && e->options.linear_probing != 0;
#ifdef NH_NO_QUAD
        quad = false;
#endif
        if (capacity >= 0xFFFFFF00ull && !quad) want = 1;
    }
    bool trimmed = false;
    for (;; want >>= 1) {
        const uint64_t sh = 32 / want;
        const uint64_t stride = ((e->table_cells_alloc + 32 + 31) & ~31ull) - (want > 1 ? sh : 0);
        const size_t bytes = (size_t)(stride * want + 64) * sizeof(uint32_t) + 256;
        hipError_t he = hipMalloc(&e->d_table_raw, bytes);
        if (he != hipSuccess && !trimmed) {
            // what earlier runs of the process keep for their next one (the gzip reader's buffers in HBM, page-locked batch
            // buffers) must not cost a database its table copies: the stores are emptied and the allocation tried again
            // (not up front: giving 100 GB back to the driver and taking them again costs a run seconds)
            (void)hipGetLastError();
            run_cache_trim();
            dev_cache_trim();  // (frees on every device it holds buffers of, and leaves the last one selected)
            (void)dev_set(e->device);
            trimmed = true;
            he = hipMalloc(&e->d_table_raw, bytes);
        }
        if (he == hipSuccess) {
            if (dev_debug()) dev_register(e->d_table_raw, bytes, e->device);
            e->d_table = (uint32_t *)(((uintptr_t)e->d_table_raw + 127) & ~(uintptr_t)127);
            e->n_copies = want;
            e->copy_stride = stride;
            if (getenv("NOHUMAN_TRACE"))
                fprintf(stderr, "[nohuman trace] hash table: %u staggered cop%s of %.2f GB\n", want, want == 1 ? "y" : "ies",
                        (double)e->table_cells_alloc * 4 / 1e9);
            return NH_OK;
        }
        (void)hipGetLastError();
        if (want == 1) return set_error(NH_EOOM, "cannot allocate the hash table (%zu bytes)", bytes);
    }
}

// Copies 1.. are refreshed from copy 0 whenever its cells changed (load, synthetic inserts).  Done at
// that time, after a device-wide sync, never lazily in the launch path: launches come in on several
// streams and host threads at once.
int refresh_table_copies(Engine *e) {
    HIP_TRY(dev_set(e->device));
    HIP_TRY(hipDeviceSynchronize());
    for (uint32_t j = 1; j < e->n_copies; j++)
        HIP_TRY(hipMemcpyAsync(e->d_table + j * e->copy_stride, e->d_table, e->table_cells_alloc * sizeof(uint32_t),
                               hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return NH_OK;
}

LaunchKnobs read_launch_knobs() {
    LaunchKnobs kn;
    if (const char *env = getenv("NOHUMAN_FRAG_CHUNK")) kn.frag_chunk = (uint32_t)std::max(1, atoi(env));
    if (const char *env = getenv("NOHUMAN_SEG_CAP")) {
        const long v = atol(env);
        if (v >= 1) kn.seg_cap = (uint64_t)v;
    }
    kn.sched = parse_sched_knobs(getenv("NOHUMAN_SCHED"));
    return kn;
}

// Reads every cell of copy 0 once (k_validate_table, ~1.3 ms per 5.7 GB) and refuses what the kernels could not survive
// or what cannot be a table kraken2-build wrote (TableCheck, nh_internal.h).  Reference: the three files are only checked for
// existence (src/lib.rs:119-141), and several database versions can be installed side by side (src/download.rs:178-222).
static int validate_table(Engine *e, bool check_size) {
    auto t0 = std::chrono::steady_clock::now();
    unsigned long long *d_out = (unsigned long long *)e->d_counters + CNT_N;  // (two of the spare words behind the counters)
    unsigned long long out[2] = {0, 0};
    HIP_TRY(hipMemsetAsync(d_out, 0, sizeof out, e->stream));
    HIP_TRY(launch_validate_table(e->d_table, e->table_cells_alloc & ~3ull, (uint32_t)((1ull << e->info.value_bits) - 1), d_out, e->stream));
    HIP_TRY(hipMemcpyAsync(out, d_out, sizeof out, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemsetAsync(d_out, 0, sizeof out, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->check.non_empty = out[0];
    e->check.max_value = out[1];
    e->check.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("NOHUMAN_TRACE"))
        fprintf(stderr, "[nohuman trace] hash table checked in %.4f s: %llu of %llu cells in use (load %.4f), largest value %llu, taxonomy nodes %llu\n",
                e->check.seconds, out[0], (unsigned long long)e->info.capacity, (double)out[0] / (double)e->info.capacity, out[1],
                (unsigned long long)e->info.node_count);
    if (out[1] >= e->info.node_count)
        return set_error(NH_EDB, "hash.k2d holds the taxon value %llu but taxo.k2d has only %llu nodes: the two files are not of one database",
                         out[1], (unsigned long long)e->info.node_count);
    if (check_size && out[0] != e->info.size)
        return set_error(NH_EDB, "hash.k2d: %llu cells are in use but its header says size %llu", out[0], (unsigned long long)e->info.size);
    return NH_OK;
}

static int common_open(Engine *e, int device) {
    int rc = check_device(device);
    if (rc) return rc;
    e->device = device;
    e->info.device = device;
    HIP_TRY(dev_set(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev_phys(device)));
    e->n_cu = prop.multiProcessorCount;
    e->grid_blocks = e->n_cu * classify_blocks_per_cu();  // every resident wave slot, persistent
    e->options.minimum_hit_groups = 2;
    e->options.linear_probing = 1;
    e->options.reset_per_mate = 1;
    e->options.ambiguity_rule = NH_AMBIGUITY_DEFAULT;
    // per-process overrides of the "verify first" switches (parity_vs_kraken2.sh walks their lattice through the CLI)
    // (NOHUMAN_OPT_AMBIGUITY_RULE: 0 = last l-mer, 1 = queue -- a plain index, as the lattice walker passes it).
    // They change what every engine of the process classifies: each one applied is said once on stderr.
    bool overridden = false;
    if (const char *v = getenv("NOHUMAN_OPT_AMBIGUITY_RULE")) e->options.ambiguity_rule = atoi(v) != 0 ? NH_AMBIGUITY_QUEUE : NH_AMBIGUITY_LAST_LMER, overridden = true;
    if (const char *v = getenv("NOHUMAN_OPT_LINEAR_PROBING")) e->options.linear_probing = atoi(v) != 0, overridden = true;
    if (const char *v = getenv("NOHUMAN_OPT_RESET_PER_MATE")) e->options.reset_per_mate = atoi(v) != 0, overridden = true;
    if (const char *v = getenv("NOHUMAN_OPT_MIN_HIT_GROUPS")) e->options.minimum_hit_groups = (uint32_t)atoi(v), overridden = true;
    if (overridden) {
        static std::atomic<bool> said{false};
        if (!said.exchange(true))
            fprintf(stderr, "nohuman: WARN NOHUMAN_OPT_* overrides the classification switches of every engine of this process: ambiguity rule %s, "
                            "linear probing %d, per-mate reset %d, minimum hit groups %u\n",
                    e->options.ambiguity_rule == NH_AMBIGUITY_QUEUE ? "queue" : "last l-mer", e->options.linear_probing, e->options.reset_per_mate,
                    e->options.minimum_hit_groups);
    }
    e->knobs = read_launch_knobs();
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    for (unsigned i = 0; i < LAUNCH_SLOTS; i++) HIP_TRY(hipEventCreateWithFlags(&e->slot_ev[i], hipEventDisableTiming));
    HIP_TRY(dev_malloc((void **)&e->d_counters, (CNT_N + 12) * sizeof(uint64_t)));
    HIP_TRY(hipMemset(e->d_counters, 0, (CNT_N + 12) * sizeof(uint64_t)));
    // launch slots start CLEAN and every launch leaves its slot clean again (k_finish_launch)
    HIP_TRY(dev_malloc((void **)&e->d_work, LAUNCH_SLOTS * WORK_PASSES * WORK_WORDS * WORK_STRIDE * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(e->d_work, 0, LAUNCH_SLOTS * WORK_PASSES * WORK_WORDS * WORK_STRIDE * sizeof(unsigned long long)));
    HIP_TRY(dev_malloc((void **)&e->d_cshard, LAUNCH_SLOTS * COUNTER_SHARDS * COUNTER_STRIDE * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(e->d_cshard, 0, LAUNCH_SLOTS * COUNTER_SHARDS * COUNTER_STRIDE * sizeof(unsigned long long)));
    // [0, LAUNCH_SLOTS) "fragments left to the BIG variant" per launch slot, then the sticky error bits,
    // then [LAUNCH_SLOTS + 1, 2 LAUNCH_SLOTS + 1) "chunks left to the generic kernel" per launch slot
    HIP_TRY(dev_malloc((void **)&e->d_error, (2 * LAUNCH_SLOTS + 1) * sizeof(int)));
    HIP_TRY(hipMemset(e->d_error, 0, (2 * LAUNCH_SLOTS + 1) * sizeof(int)));
    HIP_TRY(dev_malloc((void **)&e->d_defer, LAUNCH_SLOTS * DEFER_WORDS * sizeof(uint32_t)));
    HIP_TRY(hipMemset(e->d_defer, 0, LAUNCH_SLOTS * DEFER_WORDS * sizeof(uint32_t)));
    return NH_OK;
}

static int upload_taxonomy(Engine *e) {
    HIP_TRY(dev_malloc((void **)&e->d_parent, e->parent.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(e->d_parent, e->parent.data(), e->parent.size() * sizeof(uint32_t),
                      hipMemcpyHostToDevice));
    return NH_OK;
}

void destroy(Engine *e) {
    if (!e) return;
    if (e->device >= 0) (void)dev_set(e->device);
    if (e->d_table_raw) (void)hipFree(e->d_table_raw);
    if (e->d_parent) (void)hipFree(e->d_parent);
    if (e->d_counters) (void)hipFree(e->d_counters);
    if (e->d_error) (void)hipFree(e->d_error);
    if (e->d_work) (void)hipFree(e->d_work);
    if (e->d_cshard) (void)hipFree(e->d_cshard);
    if (e->d_defer) (void)hipFree(e->d_defer);
    for (SplitBufs &sb : e->split)
        for (void *p : {(void *)sb.hdr, (void *)sb.items_multi, (void *)sb.items_single, (void *)sb.part, (void *)sb.part_done})
            if (p) (void)hipFree(p);
    for (void *p : {e->st.d_bases, e->st.d_offsets, e->st.d_results, e->st.d_taxa, e->st.d_taxa_off})
        if (p) (void)hipFree(p);
    for (hipEvent_t ev : e->slot_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int open_images(const void *opts, size_t opts_len, const void *taxo, size_t taxo_len,
                const void *hash, size_t hash_len, int device, Engine **out) {
    if (!opts || !taxo || !hash || !out) return set_error(NH_EINVAL, "null argument");
    Engine *e = new Engine();
    int rc = common_open(e, device);
    if (!rc) rc = parse_opts(e, opts, opts_len);
    if (!rc) rc = parse_taxonomy(e, taxo, taxo_len);
    if (!rc) {
        const uint8_t *hb = (const uint8_t *)hash;
        if (hash_len < 32) rc = set_error(NH_EDB, "hash.k2d: truncated header");
        else {
            e->info.capacity = rd64(hb);
            e->info.size = rd64(hb + 8);
            e->info.key_bits = rd64(hb + 16);
            e->info.value_bits = rd64(hb + 24);
            if (e->info.key_bits + e->info.value_bits != 32 || e->info.value_bits == 0 ||
                e->info.value_bits > 31)
                rc = set_error(NH_EDB, "hash.k2d: key_bits + value_bits != 32");
            else if (e->info.capacity == 0 || hash_len != 32 + 4 * e->info.capacity)
                rc = set_error(NH_EDB, "hash.k2d: size %zu != 32 + 4*capacity (%llu)", hash_len,
                               (unsigned long long)e->info.capacity);
        }
        if (!rc) rc = check_queue_packing(e);
        if (!rc) rc = alloc_table(e, e->info.capacity);
        if (!rc) {
            hipError_t he = hipMemset(e->d_table, 0, e->table_cells_alloc * sizeof(uint32_t));
            if (he == hipSuccess)
                he = hipMemcpy(e->d_table, hb + 32, 4 * e->info.capacity, hipMemcpyHostToDevice);
            if (he != hipSuccess) rc = set_error(NH_EDEVICE, "table upload: %s", hipGetErrorString(he));
        }
    }
    if (!rc) rc = validate_table(e, true);
    if (!rc) rc = upload_taxonomy(e);
    if (!rc) rc = refresh_table_copies(e);
    if (rc) {
        destroy(e);
        return rc;
    }
    finish_devdb(e);
    *out = e;
    return NH_OK;
}

static bool file_exists(const std::string &p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

static int slurp(const std::string &path, std::vector<uint8_t> &buf) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return set_error(NH_EIO, "cannot open %s", path.c_str());
    struct stat st;
    if (fstat(fileno(f), &st) != 0) {
        fclose(f);
        return set_error(NH_EIO, "cannot stat %s", path.c_str());
    }
    buf.resize((size_t)st.st_size);
    size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), f);
    fclose(f);
    if (got != buf.size()) return set_error(NH_EIO, "short read on %s", path.c_str());
    return NH_OK;
}

// validate_db_directory semantics (/root/reference/src/lib.rs:119-141)
int resolve_db_dir(const char *db_dir, std::string &resolved) {
    static const char *req[3] = {"hash.k2d", "opts.k2d", "taxo.k2d"};
    for (const char *sub : {"", "/db"}) {
        std::string d = std::string(db_dir) + sub;
        bool ok = true;
        for (const char *r : req) ok = ok && file_exists(d + "/" + r);
        if (ok) {
            resolved = d;
            return NH_OK;
        }
    }
    return set_error(NH_EDB,
                     "Required files (hash.k2d, opts.k2d, taxo.k2d) not found in \"%s\" or its "
                     "'db' subdirectory",
                     db_dir);
}

int open_dir(const char *db_dir, int device, Engine **out) {
    if (!db_dir || !out) return set_error(NH_EINVAL, "null argument");
    std::string dir;
    int rc = resolve_db_dir(db_dir, dir);
    if (rc) return rc;
    std::vector<uint8_t> ob, tb;
    if ((rc = slurp(dir + "/opts.k2d", ob))) return rc;
    if ((rc = slurp(dir + "/taxo.k2d", tb))) return rc;
    // hash.k2d can be many GB: stream it to the device through a pinned bounce buffer
    Engine *e = new Engine();
    rc = common_open(e, device);
    if (!rc) rc = parse_opts(e, ob.data(), ob.size());
    if (!rc) rc = parse_taxonomy(e, tb.data(), tb.size());
    FILE *f = nullptr;
    if (!rc) {
        std::string hp = dir + "/hash.k2d";
        f = fopen(hp.c_str(), "rb");
        uint8_t hdr[32];
        struct stat st;
        if (!f || fstat(fileno(f), &st) != 0 || fread(hdr, 1, 32, f) != 32)
            rc = set_error(NH_EIO, "cannot read %s", hp.c_str());
        else {
            e->info.capacity = rd64(hdr);
            e->info.size = rd64(hdr + 8);
            e->info.key_bits = rd64(hdr + 16);
            e->info.value_bits = rd64(hdr + 24);
            if (e->info.key_bits + e->info.value_bits != 32 || e->info.value_bits == 0 ||
                e->info.value_bits > 31)
                rc = set_error(NH_EDB, "hash.k2d: key_bits + value_bits != 32");
            else if (e->info.capacity == 0 ||
                     (uint64_t)st.st_size != 32 + 4 * e->info.capacity)
                rc = set_error(NH_EDB, "hash.k2d: file size does not match capacity");
        }
    }
    const bool trace = getenv("NOHUMAN_TRACE") != nullptr;
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = now_s();
    if (!rc) rc = check_queue_packing(e);
    if (!rc) rc = alloc_table(e, e->info.capacity);
    const double t_b = now_s();
    double t_c = t_b;
    if (!rc) {
        // The cells go page cache -> page-locked bounce buffers -> HBM.  One thread's read() moves ~10 GB/s, a PCIe 5 x16 link
        // five times that: several loaders (NOHUMAN_DB_LOADERS, default 4) take the file's 64 MiB chunks in turn, each with its
        // own descriptor, two bounce buffers and a stream (round 5; one loader: 0.50 s for the 5.73 GB table, tools/db_load_bench.py).
        const size_t CH = 64u << 20;
        const uint64_t bytes = 4 * e->info.capacity;
        const uint64_t n_chunks = (bytes + CH - 1) / CH;
        int T = 4;
        if (const char *env = getenv("NOHUMAN_DB_LOADERS")) T = atoi(env);
        T = std::max(1, std::min<int>(T, (int)std::min<uint64_t>(16, n_chunks)));
        hipError_t he0 = hipMemset(e->d_table, 0, e->table_cells_alloc * sizeof(uint32_t));
        if (he0 == hipSuccess) he0 = hipDeviceSynchronize();  // (the loaders' streams do not wait for the legacy stream's memset)
        t_c = now_s();
        if (he0 != hipSuccess) rc = set_error(NH_EDEVICE, "table upload: %s", hipGetErrorString(he0));
        const std::string hp = dir + "/hash.k2d";
        std::vector<int> trc((size_t)T, NH_OK);
        std::vector<std::string> terr((size_t)T);
        auto loader = [&](int t) {
            auto failed = [&](int code, const std::string &m) {
                trc[(size_t)t] = code;
                terr[(size_t)t] = m;
            };
            const int fd = ::open(hp.c_str(), O_RDONLY | O_CLOEXEC);
            if (fd < 0) return failed(NH_EIO, "cannot read " + hp);
            void *pin[2] = {nullptr, nullptr};
            hipEvent_t ev[2] = {nullptr, nullptr};
            hipStream_t st = nullptr;
            hipError_t he = dev_set(device);
            if (he == hipSuccess) he = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            for (int i = 0; i < 2 && he == hipSuccess; i++) {
                he = host_malloc(&pin[i], CH, hipHostMallocDefault);
                if (he == hipSuccess) he = hipEventCreate(&ev[i]);
            }
            bool used[2] = {false, false};
            int slot = 0;
            for (uint64_t c = (uint64_t)t; c < n_chunks && he == hipSuccess && trc[(size_t)t] == NH_OK; c += (uint64_t)T) {
                const uint64_t off = c * CH;
                const size_t n = (size_t)std::min<uint64_t>(CH, bytes - off);
                if (used[slot]) he = hipEventSynchronize(ev[slot]);
                if (he != hipSuccess) break;
                size_t got = 0;
                while (got < n) {
                    const ssize_t r = pread(fd, (char *)pin[slot] + got, n - got, (off_t)(32 + off + got));
                    if (r <= 0) break;
                    got += (size_t)r;
                }
                if (got != n) {
                    failed(NH_EIO, "short read on hash.k2d");
                    break;
                }
                he = hipMemcpyAsync((uint8_t *)e->d_table + off, pin[slot], n, hipMemcpyHostToDevice, st);
                if (he == hipSuccess) he = hipEventRecord(ev[slot], st);
                used[slot] = true;
                slot ^= 1;
            }
            if (he == hipSuccess && st) he = hipStreamSynchronize(st);
            for (int i = 0; i < 2; i++) {
                if (pin[i]) (void)hipHostFree(pin[i]);
                if (ev[i]) (void)hipEventDestroy(ev[i]);
            }
            if (st) (void)hipStreamDestroy(st);
            ::close(fd);
            if (he != hipSuccess && trc[(size_t)t] == NH_OK) failed(NH_EDEVICE, std::string("table upload: ") + hipGetErrorString(he));
        };
        if (!rc) {
            std::vector<std::thread> th;
            for (int t = 1; t < T; t++) th.emplace_back(loader, t);
            loader(0);
            for (auto &x : th) x.join();
            (void)dev_set(device);
            for (int t = 0; t < T && !rc; t++)
                if (trc[(size_t)t] != NH_OK) rc = set_error(trc[(size_t)t], "%s", terr[(size_t)t].c_str());
        }
    }
    if (f) fclose(f);
    const double t_d = now_s();
    if (!rc) rc = validate_table(e, true);
    if (!rc) rc = upload_taxonomy(e);
    if (!rc) rc = refresh_table_copies(e);
    if (trace && !rc)
        fprintf(stderr, "[nohuman trace] database: table allocated in %.3f s, cleared in %.3f s, %.2f GB uploaded in %.3f s, taxonomy + copies %.3f s\n", t_b - t_a,
                t_c - t_b, 4.0 * e->info.capacity / 1e9, t_d - t_c, now_s() - t_d);
    if (rc) {
        destroy(e);
        return rc;
    }
    finish_devdb(e);
    *out = e;
    return NH_OK;
}

// chain taxonomy 1 -> 2 -> ... -> depth, external ids: 1 (root), 2..depth-1, 9606 for the leaf
static std::vector<uint8_t> chain_taxonomy(uint32_t depth) {
    const uint64_t nc = (uint64_t)depth + 1;
    std::string names, ranks = std::string("no rank") + '\0';
    std::vector<uint64_t> nodes(7 * nc, 0);
    for (uint64_t i = 0; i < nc; i++) {
        uint64_t *n = &nodes[7 * i];
        n[0] = i >= 2 ? i - 1 : 0;
        n[1] = (i >= 1 && i + 1 < nc) ? i + 1 : 0;
        n[2] = (i >= 1 && i + 1 < nc) ? 1 : 0;
        n[3] = names.size();
        n[4] = 0;
        n[5] = i == 0 ? 0 : (i + 1 == nc ? 9606 : i);
        char nm[32];
        snprintf(nm, sizeof nm, "node%llu", (unsigned long long)i);
        names += nm;
        names += '\0';
    }
    std::vector<uint8_t> img;
    img.insert(img.end(), (const uint8_t *)"K2TAXDAT", (const uint8_t *)"K2TAXDAT" + 8);
    uint64_t hdr[3] = {nc, names.size(), ranks.size()};
    img.insert(img.end(), (uint8_t *)hdr, (uint8_t *)hdr + 24);
    img.insert(img.end(), (uint8_t *)nodes.data(), (uint8_t *)nodes.data() + 56 * nc);
    img.insert(img.end(), names.begin(), names.end());
    img.insert(img.end(), ranks.begin(), ranks.end());
    return img;
}

int open_synthetic(uint64_t capacity, uint64_t n_keys, uint32_t depth, uint64_t seed, int device,
                   Engine **out) {
    if (!out || capacity == 0 || depth == 0 || n_keys >= capacity)
        return set_error(NH_EINVAL, "open_synthetic: need 0 < n_keys < capacity and depth > 0");
    Engine *e = new Engine();
    int rc = common_open(e, device);
    // kraken2 nucleotide defaults: k=35 l=31, 7 spaced positions, default toggle
    uint8_t ob[64];
    memset(ob, 0, sizeof ob);
    uint64_t v[4] = {35, 31, 0x3FFFFFFFF3333333ull, 0xe37e28c4271b5a2dull};
    memcpy(ob, v, 32);
    ob[32] = 1;
    int32_t rv = 1;
    memcpy(ob + 48, &rv, 4);
    if (!rc) rc = parse_opts(e, ob, sizeof ob);
    std::vector<uint8_t> tx = chain_taxonomy(depth);
    if (!rc) rc = parse_taxonomy(e, tx.data(), tx.size());
    if (!rc) {
        uint32_t vb = 1;
        while ((1ull << vb) < e->info.node_count) vb++;
        e->info.capacity = capacity;
        e->info.value_bits = vb;
        e->info.key_bits = 32 - vb;
        rc = check_queue_packing(e);
        if (!rc) rc = alloc_table(e, capacity);
    }
    if (!rc) rc = upload_taxonomy(e);
    if (!rc) {
        hipError_t he = hipMemsetAsync(e->d_table, 0, e->table_cells_alloc * sizeof(uint32_t), e->stream);
        unsigned long long *d_size = (unsigned long long *)e->d_counters;
        if (he == hipSuccess)
            he = launch_synth_insert(e->d_table, capacity, ~0ull / capacity,
                                     (uint32_t)e->info.value_bits, depth, n_keys, seed,
                                     (1ull << 62) - 1, d_size, e->stream);
        unsigned long long sz = 0;
        if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
        if (he == hipSuccess) he = hipMemcpy(&sz, d_size, 8, hipMemcpyDeviceToHost);
        if (he == hipSuccess) he = hipMemset(e->d_counters, 0, (CNT_N + 12) * sizeof(uint64_t));
        if (he != hipSuccess) rc = set_error(NH_EDEVICE, "synthetic table: %s", hipGetErrorString(he));
        e->info.size = sz;
    }
    if (!rc) rc = validate_table(e, true);  // (the generator checked like a file: what it inserted is what the table holds)
    if (!rc) rc = refresh_table_copies(e);
    if (rc) {
        destroy(e);
        return rc;
    }
    finish_devdb(e);
    *out = e;
    return NH_OK;
}

static int ensure(void **p, size_t *cap, size_t need) {
    if (*cap >= need) return NH_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    size_t want = need + need / 4 + 256;
    HIP_TRY(dev_malloc(p, want));
    *cap = want;
    return NH_OK;
}

uint64_t kmer_taxa_entries(const Engine *e, const uint64_t *seq_offsets, uint64_t n_frag, int mates,
                           uint64_t *offsets_out) {
    const uint64_t k = e->info.k;
    uint64_t total = 0;
    for (uint64_t f = 0; f < n_frag; f++) {
        if (offsets_out) offsets_out[f] = total;
        for (int m = 0; m < mates; m++) {
            uint64_t s = f * (uint64_t)mates + (uint64_t)m;
            uint64_t len = seq_offsets[s + 1] - seq_offsets[s];
            if (len >= k) total += len - k + 1;
        }
        if (mates == 2) total += 1;
    }
    if (offsets_out) offsets_out[n_frag] = total;
    return total;
}


// Every launch takes the next of LAUNCH_SLOTS (scheduling counter, "BIG pass pending" word), so
// launches in flight on different streams of one engine never share them; the launch that takes a slot
// again waits for the slot's previous launch (Engine::slot_ev): more than LAUNCH_SLOTS launches in flight
// on one engine are serialised slot by slot, never mixed.
// Fragments a wave claims at a time.  Their offsets (mates * n + 1) must fit the 64 lanes; larger
// chunks amortise the claim and the offsets load (measured best: 24 paired, 32 single-end), but a
// small batch is cut finer so that every resident wave still gets about two chunks.
static uint32_t frag_chunk_for(const Engine *e, uint32_t flags, uint64_t n_frag) {
    if (flags & NH_FLAG_LONG) return 1;
    const bool paired = (flags & NH_FLAG_PAIRED) != 0;
    const uint32_t cap = paired ? 31u : 63u;
    if (e->knobs.frag_chunk) {  // NOHUMAN_FRAG_CHUNK (tuning / test knob, read when the engine was opened)
        const uint32_t c = e->knobs.frag_chunk;
        return c > cap ? cap : c;
    }
    // 24 pairs / 32 reads = whole batches of 4 tiles (measured best: 31 / 16 / 12 pairs are 1.5-2 % slower, and for a
    // small launch finer chunks lose more at their boundaries than they win at the tail: 1 M single reads 634
    // Mreads/s in chunks of 12 against 765 in chunks of 32); only a launch too small to give every wave two
    // chunks is cut finer.
    // round 4 (tools/sweep_sched.py sechunk, three interleaved passes, 1 M single reads): 32 -> 1.087 ms, 40 / 48 / 56 ->
    // 1.075, 60 -> 1.062 (+2.3 %): single-end chunks take the 60 reads whose 61 offsets the lanes can hold
    uint32_t c = paired ? 24u : 60u;
    const uint64_t waves = (uint64_t)e->grid_blocks * 4;
    const uint64_t fair = n_frag / (2 * waves);
    if (fair < c) c = (uint32_t)fair;
    const uint32_t step = paired ? 2u : 4u;  // one batch of 4 tiles
    c = c / step * step;
    return c < step ? step : c;
}

// Item buffers of a launch slot for a launch of n_frag long reads: one list entry per read, and items /
// partial slots for the segments of the reads that get cut (8 per read on average, 2^16 .. 2^20; what does
// not fit is classified whole).  NOHUMAN_SEG_CAP (LaunchKnobs) overrides the number of segments (test knob).
static int ensure_split(Engine *e, unsigned slot, uint64_t n_frag) {
    std::lock_guard<std::mutex> lock(e->split_mu);
    SplitBufs &sb = e->split[slot];
    uint64_t want_seg = n_frag * 8;
    if (want_seg < (1u << 16)) want_seg = 1u << 16;
    if (want_seg > (1u << 20)) want_seg = 1u << 20;
    if (e->knobs.seg_cap) want_seg = e->knobs.seg_cap;
    if (sb.hdr && sb.seg_cap >= want_seg && e->split_single_cap[slot] >= n_frag) return NH_OK;
    if (sb.hdr && e->knobs.seg_cap && e->split_single_cap[slot] >= n_frag) return NH_OK;
    // (re)allocate: the slot's previous launch, if any, has long finished when its turn comes again
    for (void *p : {(void *)sb.hdr, (void *)sb.items_multi, (void *)sb.items_single, (void *)sb.part, (void *)sb.part_done})
        if (p) (void)hipFree(p);
    sb = SplitBufs{};
    e->split_single_cap[slot] = 0;
    const uint64_t nsingle = n_frag + n_frag / 4 + 1024;
    if (dev_malloc((void **)&sb.hdr, sizeof(SplitHdr)) != hipSuccess ||
        dev_malloc((void **)&sb.items_multi, want_seg * sizeof(SplitItem)) != hipSuccess ||
        dev_malloc((void **)&sb.items_single, nsingle * sizeof(uint32_t)) != hipSuccess ||
        dev_malloc((void **)&sb.part, want_seg * PART_DWORDS * sizeof(uint32_t)) != hipSuccess ||
        dev_malloc((void **)&sb.part_done, want_seg * sizeof(uint32_t)) != hipSuccess) {
        (void)hipGetLastError();
        for (void *p : {(void *)sb.hdr, (void *)sb.items_multi, (void *)sb.items_single, (void *)sb.part, (void *)sb.part_done})
            if (p) (void)hipFree(p);
        sb = SplitBufs{};
        return NH_OK;  // no buffers: the launch goes by whole reads (slower tail, same results)
    }
    // (the header of fresh buffers is cleared by the first launch that uses them, on ITS stream: a hipMemset here would
    //  only be ordered with the legacy stream, and the launches run on non-blocking ones; later launches find the
    //  header cleared by k_finish_launch)
    e->split_fresh[slot] = true;
    sb.seg_cap = (uint32_t)want_seg;
    e->split_single_cap[slot] = nsingle;
    return NH_OK;
}

int classify_device(Engine *e, const void *d_bases, const void *d_seq_off, uint64_t n_frag,
                         uint32_t flags, double confidence, void *d_results, void *d_kmer_taxa,
                         const void *d_kmer_taxa_off, void *d_counters, hipStream_t stream,
                         const void *d_seq_len, uint64_t bases_end) {
    dev_check(e->device, "classify_device");  // (NOHUMAN_DEBUG_DEVICE: the launch belongs to the engine's device, and so do its buffers)
    dev_check_ptr(d_bases, e->device, "classify_device (sequence text)");
    dev_check_ptr(d_results, e->device, "classify_device (results)");
    if (!(confidence >= 0.0 && confidence <= 1.0))
        return set_error(NH_EINVAL, "Confidence score must be in the closed interval [0, 1]");
    if ((d_kmer_taxa != nullptr) != (d_kmer_taxa_off != nullptr))
        return set_error(NH_EINVAL, "kmer_taxa and kmer_taxa_offsets go together");
    if (((uintptr_t)d_bases & 3) != 0) return set_error(NH_EINVAL, "d_bases must be 4-byte aligned");
    DevDB db;
    {   // options may be set from another thread: build this launch's parameter block under the lock
        std::lock_guard<std::mutex> lock(e->db_mu);
        finish_devdb(e);
        db = e->dev;
    }
    LaunchIO io;
    io.d_bases = d_bases;
    io.d_seq_off = d_seq_off;
    io.d_seq_len = d_seq_len;
    io.bases_end = bases_end;
    io.n_frag = n_frag;
    io.mates = (flags & NH_FLAG_PAIRED) ? 2 : 1;
    io.long_reads = (flags & NH_FLAG_LONG) != 0;
    io.d_out = d_results;
    io.d_kmer_taxa = d_kmer_taxa;
    io.d_kmer_taxa_off = d_kmer_taxa_off;
    io.d_counters = d_counters;
    // The launch takes the next slot and holds its mutex until its event is recorded: the slot's next user -- this thread
    // or another, sixteen launches on -- finds the event of the launch before it and makes ITS stream wait for it.
    const unsigned slot = e->launch_seq.fetch_add(1) % LAUNCH_SLOTS;
    std::lock_guard<std::mutex> slot_lock(e->slot_mu[slot]);
#ifndef NH_NO_SLOT_WAIT  // (-DNH_NO_SLOT_WAIT: rounds 1-5's behaviour, to show that tests/test_gpu_threads.py sees the difference)
    // (also when the slot's last launch went to this very stream handle: the runtime drops a wait for an event of the same
    //  stream itself, and a handle can be a NEW stream that took a destroyed one's place)
    if (e->slot_used[slot]) {
        const hipError_t we = hipStreamWaitEvent(stream, e->slot_ev[slot], 0);
        if (we != hipSuccess) return set_error(NH_EDEVICE, "classify launch (slot wait): %s", hipGetErrorString(we));
    }
#endif
    LaunchSlot sl;
    sl.d_error = e->d_error + LAUNCH_SLOTS;
    sl.d_pending = e->d_error + slot;
    sl.d_pending_long = e->d_error + LAUNCH_SLOTS + 1 + slot;
    sl.d_work = e->d_work + (size_t)slot * WORK_PASSES * WORK_WORDS * WORK_STRIDE;
    sl.d_cshard = e->d_cshard + (size_t)slot * COUNTER_SHARDS * COUNTER_STRIDE;
    sl.d_defer = e->d_defer + (size_t)slot * DEFER_WORDS;
    sl.defer_cap_bits = DEFER_WORDS * 32;
    if ((flags & NH_FLAG_LONG) && !(flags & NH_FLAG_PAIRED) && n_frag < 0xFFFFFFFFull) {
        (void)ensure_split(e, slot, n_frag);
        sl.split = e->split[slot];
        sl.split_single_cap = e->split_single_cap[slot];
        {
            std::lock_guard<std::mutex> lock(e->split_mu);
            sl.split_fresh = e->split_fresh[slot];
            e->split_fresh[slot] = false;
        }
    }
    hipError_t he = launch_classify(db, io, confidence, sl, frag_chunk_for(e, flags, n_frag), e->grid_blocks, stream, e->knobs.sched);
    if (he == hipSuccess && n_frag) {
        he = hipEventRecord(e->slot_ev[slot], stream);
        e->slot_used[slot] = true;
    }
    if (he != hipSuccess) return set_error(NH_EDEVICE, "classify launch: %s", hipGetErrorString(he));
    return NH_OK;
}

int check_error_flag(Engine *e) {
    int flag = 0;
    dev_check(e->device, "check_error_flag");
    HIP_TRY(hipMemcpy(&flag, e->d_error + LAUNCH_SLOTS, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) {
        HIP_TRY(hipMemset(e->d_error + LAUNCH_SLOTS, 0, sizeof(int)));
        if (flag & 2)
            return set_error(NH_EINVAL, "a sequence of 2^31 bases or more is not supported");
        return set_error(NH_ECAPACITY, "a fragment hit more than 2048 distinct taxa");
    }
    return NH_OK;
}

int classify_host(Engine *e, const uint8_t *bases, const uint64_t *seq_offsets, uint64_t n_frag,
                  uint32_t flags, double confidence, nh_result *results, uint32_t *kmer_taxa,
                  uint64_t *kmer_taxa_offsets, uint64_t kmer_taxa_cap) {
    if (!seq_offsets || !results || (!bases && n_frag))
        return set_error(NH_EINVAL, "null argument");
    if ((kmer_taxa != nullptr) != (kmer_taxa_offsets != nullptr))
        return set_error(NH_EINVAL, "kmer_taxa and kmer_taxa_offsets go together");
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(dev_set(e->device));
    const int mates = (flags & NH_FLAG_PAIRED) ? 2 : 1;
    const uint64_t n_seq = n_frag * (uint64_t)mates;
    if (n_frag == 0) return NH_OK;
    const uint64_t base0 = seq_offsets[0];
    const uint64_t total = seq_offsets[n_seq] - base0;
    for (uint64_t s = 0; s < n_seq; s++)
        if (seq_offsets[s + 1] < seq_offsets[s]) return set_error(NH_EINVAL, "seq_offsets not monotone");
    auto t0 = std::chrono::steady_clock::now();
    Staging &st = e->st;
    int rc;
    if ((rc = ensure(&st.d_bases, &st.cap_bases, total + 64))) return rc;
    if ((rc = ensure(&st.d_offsets, &st.cap_offsets, (n_seq + 1) * 8))) return rc;
    if ((rc = ensure(&st.d_results, &st.cap_results, n_frag * sizeof(nh_result)))) return rc;
    // offsets are rebased so that the staged bases start at 0 (4-byte aligned by hipMalloc)
    std::vector<uint64_t> rebased;
    const uint64_t *offs = seq_offsets;
    if (base0 != 0) {
        rebased.resize(n_seq + 1);
        for (uint64_t s = 0; s <= n_seq; s++) rebased[s] = seq_offsets[s] - base0;
        offs = rebased.data();
    }
    uint64_t n_taxa = 0;
    if (kmer_taxa) {
        n_taxa = kmer_taxa_entries(e, offs, n_frag, mates, kmer_taxa_offsets);
        if (n_taxa > kmer_taxa_cap) return set_error(NH_EINVAL, "kmer_taxa buffer too small");
        if ((rc = ensure(&st.d_taxa, &st.cap_taxa, (n_taxa + 1) * 4))) return rc;
        if ((rc = ensure(&st.d_taxa_off, &st.cap_taxa_off, (n_frag + 1) * 8))) return rc;
        HIP_TRY(hipMemcpyAsync(st.d_taxa_off, kmer_taxa_offsets, (n_frag + 1) * 8,
                               hipMemcpyHostToDevice, e->stream));
    }
    // One H2D copy, one launch, one D2H copy on one stream.  (Measured on the MI355X box: cutting the
    // batch into pieces on two streams to overlap the copy with the kernel is slower, 11.5 vs 9.4 ms
    // per 1 M pairs -- tools/host_batch_bench.py.)
    if (total) HIP_TRY(hipMemcpyAsync(st.d_bases, bases + base0, total, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemsetAsync((uint8_t *)st.d_bases + total, 'A', 64, e->stream));
    HIP_TRY(hipMemcpyAsync(st.d_offsets, offs, (n_seq + 1) * 8, hipMemcpyHostToDevice, e->stream));
    if (total / n_frag > 2000) flags |= NH_FLAG_LONG;  // long reads: finer dynamic scheduling
    rc = classify_device(e, st.d_bases, st.d_offsets, n_frag, flags, confidence, st.d_results,
                         kmer_taxa ? st.d_taxa : nullptr, kmer_taxa ? st.d_taxa_off : nullptr,
                         e->d_counters, e->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(results, st.d_results, n_frag * sizeof(nh_result), hipMemcpyDeviceToHost,
                           e->stream));
    if (kmer_taxa && n_taxa)
        HIP_TRY(hipMemcpyAsync(kmer_taxa, st.d_taxa, n_taxa * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if ((rc = check_error_flag(e))) return rc;
    e->seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return NH_OK;
}

}  // namespace nh

// ------------------------------------------------------------------------------------------------
using nh::Engine;
using nh::set_error;
using nh::finish_devdb_public;

extern "C" {

const char *nh_last_error(void) { return nh::g_last_error.c_str(); }
int nh_abi_version(void) { return NH_ABI_VERSION; }

int nh_device_count(int *count) {
    if (!count) return set_error(NH_EINVAL, "null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return set_error(NH_EDEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n > 0 ? nh::dev_count() : 0;  // (logical devices: NOHUMAN_FAKE_DEVICES on a test box)
    return NH_OK;
}

int nh_probe(char *msg, size_t msg_len) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    int rc = NH_OK;
    char buf[256];
    if (e != hipSuccess || n <= 0) {
        snprintf(buf, sizeof buf, "no HIP device available (%s)",
                 e == hipSuccess ? "count 0" : hipGetErrorString(e));
        rc = NH_EDEVICE;
    } else {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, 0);
        if (e != hipSuccess) {
            snprintf(buf, sizeof buf, "hipGetDeviceProperties: %s", hipGetErrorString(e));
            rc = NH_EDEVICE;
        } else if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            snprintf(buf, sizeof buf, "device 0 is %s, need gfx950", prop.gcnArchName);
            rc = NH_EDEVICE;
        } else {
            snprintf(buf, sizeof buf, "%d x %s (%s), %d CUs, %.0f GiB", n, prop.name,
                     prop.gcnArchName, prop.multiProcessorCount,
                     (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
        }
    }
    if (msg && msg_len) snprintf(msg, msg_len, "%s", buf);
    if (rc) set_error(rc, "%s", buf);
    return rc;
}

int nh_open(const char *db_dir, int device, nh_engine **out) {
    return nh::open_dir(db_dir, device, (Engine **)out);
}
int nh_open_images(const void *opts, size_t opts_len, const void *taxo, size_t taxo_len,
                   const void *hash, size_t hash_len, int device, nh_engine **out) {
    return nh::open_images(opts, opts_len, taxo, taxo_len, hash, hash_len, device, (Engine **)out);
}
int nh_open_synthetic(uint64_t capacity, uint64_t n_keys, uint32_t depth, uint64_t seed, int device,
                      nh_engine **out) {
    return nh::open_synthetic(capacity, n_keys, depth, seed, device, (Engine **)out);
}
int nh_close(nh_engine *e) {
    nh::destroy((Engine *)e);
    nh::run_cache_trim();  // (what the runs kept for one another goes with the engine)
    nh::dev_cache_trim();
    return NH_OK;
}

int nh_db_info_get(const nh_engine *e, nh_db_info *info) {
    if (!e || !info) return set_error(NH_EINVAL, "null argument");
    *info = ((const Engine *)e)->info;
    return NH_OK;
}
int nh_options_get(const nh_engine *e, nh_options *o) {
    if (!e || !o) return set_error(NH_EINVAL, "null argument");
    *o = ((const Engine *)e)->options;
    return NH_OK;
}
int nh_options_set(nh_engine *e, const nh_options *o) {
    if (!e || !o) return set_error(NH_EINVAL, "null argument");
    if (o->ambiguity_rule < NH_AMBIGUITY_ENGINE_DEFAULT || o->ambiguity_rule > NH_AMBIGUITY_QUEUE)
        return set_error(NH_EINVAL, "nh_options.ambiguity_rule must be 0 (the engine's default), 1 (last l-mer) or 2 (queue)");
    std::lock_guard<std::mutex> lock(((Engine *)e)->db_mu);
    ((Engine *)e)->options = *o;
    if (o->ambiguity_rule == NH_AMBIGUITY_ENGINE_DEFAULT) ((Engine *)e)->options.ambiguity_rule = NH_AMBIGUITY_DEFAULT;
    return NH_OK;
}
int nh_taxon_external(const nh_engine *e_, uint32_t internal, uint64_t *external) {
    const Engine *e = (const Engine *)e_;
    if (!e || !external) return set_error(NH_EINVAL, "null argument");
    if (internal >= e->external.size()) return set_error(NH_EINVAL, "taxon id out of range");
    *external = e->external[internal];
    return NH_OK;
}
int nh_table_download(const nh_engine *e_, uint32_t *cells, uint64_t n_cells) {
    const Engine *e = (const Engine *)e_;
    if (!e || !cells) return set_error(NH_EINVAL, "null argument");
    if (n_cells != e->info.capacity) return set_error(NH_EINVAL, "n_cells != capacity");
    HIP_TRY(nh::dev_set(e->device));
    HIP_TRY(hipMemcpy(cells, e->d_table, n_cells * 4, hipMemcpyDeviceToHost));
    return NH_OK;
}
static int copy_image(const std::vector<uint8_t> &img, void *buf, size_t cap, size_t *len) {
    if (!len) return set_error(NH_EINVAL, "null argument");
    *len = img.size();
    if (buf) {
        if (cap < img.size()) return set_error(NH_EINVAL, "buffer too small");
        memcpy(buf, img.data(), img.size());
    }
    return NH_OK;
}
int nh_taxonomy_image(const nh_engine *e, void *buf, size_t cap, size_t *len) {
    if (!e) return set_error(NH_EINVAL, "null argument");
    return copy_image(((const Engine *)e)->taxo_image, buf, cap, len);
}
int nh_opts_image(const nh_engine *e, void *buf, size_t cap, size_t *len) {
    if (!e) return set_error(NH_EINVAL, "null argument");
    return copy_image(((const Engine *)e)->opts_image, buf, cap, len);
}

int nh_classify_batch(nh_engine *e, const uint8_t *bases, const uint64_t *seq_offsets,
                      uint64_t n_frag, uint32_t flags, double confidence, nh_result *results,
                      uint32_t *kmer_taxa, uint64_t *kmer_taxa_offsets, uint64_t kmer_taxa_cap) {
    if (!e) return set_error(NH_EINVAL, "null engine");
    return nh::classify_host((Engine *)e, bases, seq_offsets, n_frag, flags, confidence, results,
                             kmer_taxa, kmer_taxa_offsets, kmer_taxa_cap);
}

uint64_t nh_kmer_taxa_entries(const nh_engine *e, const uint64_t *seq_offsets, uint64_t n_frag,
                              uint32_t flags) {
    if (!e || !seq_offsets) return 0;
    return nh::kmer_taxa_entries((const Engine *)e, seq_offsets, n_frag,
                                 (flags & NH_FLAG_PAIRED) ? 2 : 1, nullptr);
}

int nh_classify_batch_device(nh_engine *e_, const void *d_bases, const void *d_seq_offsets,
                             uint64_t n_frag, uint32_t flags, double confidence, void *d_results,
                             void *d_kmer_taxa, const void *d_kmer_taxa_offsets, void *d_counters,
                             void *stream) {
    Engine *e = (Engine *)e_;
    if (!e || !d_bases || !d_seq_offsets || !d_results) return set_error(NH_EINVAL, "null argument");
    HIP_TRY(nh::dev_set(e->device));  // (the launch goes to the engine's device whatever the calling thread had selected)
    return nh::classify_device(e, d_bases, d_seq_offsets, n_frag, flags, confidence, d_results,
                               d_kmer_taxa, d_kmer_taxa_offsets, d_counters, (hipStream_t)stream);
}

int nh_classify_records_device(nh_engine *e_, const void *d_text, uint64_t text_len, const void *d_seq_starts,
                               const void *d_seq_lens, uint64_t n_frag, uint32_t flags, double confidence,
                               void *d_results, void *d_kmer_taxa, const void *d_kmer_taxa_offsets,
                               void *d_counters, void *stream) {
    Engine *e = (Engine *)e_;
    if (!e || !d_text || !d_seq_starts || !d_seq_lens || !d_results) return set_error(NH_EINVAL, "null argument");
    HIP_TRY(nh::dev_set(e->device));
    return nh::classify_device(e, d_text, d_seq_starts, n_frag, flags, confidence, d_results, d_kmer_taxa,
                               d_kmer_taxa_offsets, d_counters, (hipStream_t)stream, d_seq_lens, text_len);
}

int nh_synthetic_add_sequences(nh_engine *e_, const void *d_bases, const void *d_seq_offsets,
                               uint64_t n_seq, uint32_t value, void *stream) {
    Engine *e = (Engine *)e_;
    if (!e || !d_bases || !d_seq_offsets) return set_error(NH_EINVAL, "null argument");
    if (value == 0 || value >= e->info.node_count) return set_error(NH_EINVAL, "value is not a taxon id");
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(nh::dev_set(e->device));
    finish_devdb_public(e);
    unsigned long long *d_ins = nullptr, ins = 0;
    HIP_TRY(nh::dev_malloc((void **)&d_ins, 8));
    HIP_TRY(hipMemsetAsync(d_ins, 0, 8, (hipStream_t)stream));
    hipError_t he = nh::launch_insert_sequences(e->dev, d_bases, d_seq_offsets, n_seq, value, d_ins,
                                                e->grid_blocks, (hipStream_t)stream);
    if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)stream);
    if (he == hipSuccess) he = hipMemcpy(&ins, d_ins, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_ins);
    if (he != hipSuccess)
        return set_error(NH_EDEVICE, "insert sequences (default k=35/l=31 linear-probing DBs only): %s",
                         hipGetErrorString(he));
    e->info.size += ins;
    return refresh_table_copies(e);
}

// test hook (not part of the ABI in include/nohuman_engine.h): the claim map a launch of n_frag fragments
// would get -- out = {n0, n01, total, base1, base2, c0, c1, c2}
void nh_debug_sched(uint64_t n_frag, uint32_t c0, int mates, uint64_t waves, uint64_t *out) {
    const nh::Sched sc = nh::make_sched(n_frag, c0, mates, waves, nh::parse_sched_knobs(getenv("NOHUMAN_SCHED")));
    const uint64_t v[8] = {sc.n0, sc.n01, sc.total, sc.base1, sc.base2, sc.c0, sc.c1, sc.c2};
    memcpy(out, v, sizeof v);
}

// test / tuning hook (not part of the ABI): read NOHUMAN_FRAG_CHUNK / NOHUMAN_SEG_CAP / NOHUMAN_SCHED again for an open
// engine (tools/sweep_sched.py changes them between launches); no launch of the engine may be in flight
void nh_debug_reload_knobs(nh_engine *e) {
    if (e) ((Engine *)e)->knobs = nh::read_launch_knobs();
}

int nh_db_check_get(const nh_engine *e_, nh_db_check *c) {
    const Engine *e = (const Engine *)e_;
    if (!e || !c) return set_error(NH_EINVAL, "null argument");
    c->non_empty_cells = e->check.non_empty;
    c->max_value = e->check.max_value;
    c->load_factor = e->info.capacity ? (double)e->check.non_empty / (double)e->info.capacity : 0.0;
    c->seconds = e->check.seconds;
    return NH_OK;
}

int nh_stats_get(nh_engine *e_, nh_stats *s) {
    Engine *e = (Engine *)e_;
    if (!e || !s) return set_error(NH_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(nh::dev_set(e->device));
    uint64_t c[nh::CNT_N];
    HIP_TRY(hipMemcpy(c, e->d_counters, sizeof c, hipMemcpyDeviceToHost));
    s->total_sequences = c[nh::CNT_FRAGMENTS];
    s->classified = c[nh::CNT_CLASSIFIED];
    s->unclassified = c[nh::CNT_FRAGMENTS] - c[nh::CNT_CLASSIFIED];
    s->total_bases = c[nh::CNT_BASES];
    s->table_lookups = c[nh::CNT_LOOKUPS];
    s->seconds = e->seconds;
    return NH_OK;
}
int nh_stats_reset(nh_engine *e_) {
    Engine *e = (Engine *)e_;
    if (!e) return set_error(NH_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(nh::dev_set(e->device));
    HIP_TRY(hipMemset(e->d_counters, 0, (nh::CNT_N + 12) * sizeof(uint64_t)));
    HIP_TRY(hipDeviceSynchronize());  // (launches run on non-blocking streams, which a legacy-stream memset does not order)
    e->seconds = 0;
    return NH_OK;
}

}  // extern "C"
