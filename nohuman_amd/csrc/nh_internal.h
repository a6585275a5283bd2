// nh_internal.h -- host-side engine state shared by nh_engine.hip and nh_run.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "nh_device.h"
#include "nohuman_engine.h"

namespace nh {
constexpr unsigned LAUNCH_SLOTS = 16;  // launches of one engine that may be in flight at once
constexpr uint64_t DEFER_WORDS = 1u << 15;  // 1 Mi chunks per launch (24 M pairs); larger launches skip the short-read kernel
}

namespace nh {

struct Staging {  // grow-only device buffers behind nh_classify_batch (host-buffer entry)
    void *d_bases = nullptr, *d_offsets = nullptr, *d_results = nullptr, *d_taxa = nullptr,
         *d_taxa_off = nullptr;
    size_t cap_bases = 0, cap_offsets = 0, cap_results = 0, cap_taxa = 0, cap_taxa_off = 0;
};

// Tuning / test knobs of a launch, read from the environment ONCE, when the engine is opened (round 6; rounds 1-5 called
// getenv at every launch): NOHUMAN_FRAG_CHUNK (fragments a wave claims at a time), NOHUMAN_SEG_CAP (segments a long-read
// launch may cut), NOHUMAN_SCHED (claim map: "off" or c1,c2,p1,p2).  nh_debug_reload_knobs (a test hook like
// nh_debug_sched, not part of the ABI) reads them again for the sweeps of tools/.
struct LaunchKnobs {
    uint32_t frag_chunk = 0;  // 0: the default for the launch's shape
    uint64_t seg_cap = 0;     // 0: 8 segments per read, 2^16 .. 2^20
    SchedKnobs sched;
};
LaunchKnobs read_launch_knobs();
SchedKnobs parse_sched_knobs(const char *env);  // NOHUMAN_SCHED's grammar (nh_kernels.hip, beside make_sched)

// What nh_open* found when it read every cell of the table once (k_validate_table): a cell's value indexes the taxonomy
// in every kernel, so a value >= node_count (a hash.k2d beside another database's taxo.k2d) is refused with NH_EDB
// instead of faulting on the GPU; so is a table whose non-empty cells are not the `size` its header states.
struct TableCheck {
    uint64_t non_empty = 0, max_value = 0;
    double seconds = 0;
};

struct Engine {
    int device = -1;
    int n_cu = 0;
    int grid_blocks = 0;
    nh_db_info info{};
    nh_options options{};
    DevDB dev{};
    void *d_table_raw = nullptr;    // one allocation: n_copies staggered copies of the table (DevDB::copy_stride)
    uint32_t *d_table = nullptr;    // copy 0, 128-byte aligned
    uint32_t n_copies = 1;
    uint64_t copy_stride = 0;       // cells
    uint64_t table_cells_alloc = 0; // cells of one copy incl. padding
    uint32_t *d_parent = nullptr;
    uint64_t *d_counters = nullptr;
    int *d_error = nullptr;
    unsigned long long *d_work = nullptr;  // work counters of k_classify: WORK_WORDS per launch slot (nh_device.h)
    unsigned long long *d_cshard = nullptr; // counter rows the waves of a launch add to, COUNTER_SHARDS per launch slot
    uint32_t *d_defer = nullptr;           // per launch slot DEFER_WORDS words: chunks the short-read kernel left behind
    // long-read item buffers per launch slot (SplitBufs, nh_device.h), allocated when a slot first carries a
    // launch of long single-end reads and grown when a larger one comes
    SplitBufs split[LAUNCH_SLOTS] = {};
    uint64_t split_single_cap[LAUNCH_SLOTS] = {};
    bool split_fresh[LAUNCH_SLOTS] = {};  // buffers allocated, header not yet cleared (the first launch does it)
    std::mutex split_mu;
    std::atomic<unsigned> launch_seq{0};
    // A launch slot is used by one launch at a time: the launch that takes a slot waits (on the device, hipStreamWaitEvent)
    // for the event the slot's previous launch recorded behind its k_finish_launch.  With at most LAUNCH_SLOTS launches
    // in flight -- what every caller in this repo does -- the event has long fired; more than that are ORDERED, not
    // undefined (rounds 1-5 documented "not supported" and nothing detected it).  slot_mu orders the hosts' enqueues.
    hipEvent_t slot_ev[LAUNCH_SLOTS] = {};
    bool slot_used[LAUNCH_SLOTS] = {};
    std::mutex slot_mu[LAUNCH_SLOTS];
    LaunchKnobs knobs;
    TableCheck check;
    hipStream_t stream = nullptr;
    std::vector<uint32_t> parent;
    std::vector<uint64_t> external;
    std::vector<uint8_t> taxo_image, opts_image;
    Staging st;
    std::mutex mu;
    std::mutex db_mu;  // options / DevDB parameter block (set from one thread, read by launches on others)
    double seconds = 0;
};

extern thread_local std::string g_last_error;
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

hipError_t launch_classify(const DevDB &db, const LaunchIO &io, double confidence, const LaunchSlot &sl,
                           uint32_t frag_chunk, int grid_blocks, hipStream_t stream, const SchedKnobs &sched_knobs);
Sched make_sched(uint64_t n_frag, uint32_t c0, int mates, uint64_t waves, const SchedKnobs &kn);
hipError_t launch_validate_table(const uint32_t *table, uint64_t n_cells, uint32_t vmask, unsigned long long *d_out2,
                                 hipStream_t stream);
int classify_blocks_per_cu();
hipError_t launch_insert_sequences(const DevDB &db, const void *d_bases, const void *d_seq_off,
                                   uint64_t n_seq, uint32_t value, unsigned long long *d_inserted,
                                   int grid_blocks, hipStream_t stream);
hipError_t launch_synth_insert(uint32_t *table, uint64_t capacity, uint64_t cap_magic,
                               uint32_t value_bits, uint32_t value, uint64_t n_keys, uint64_t seed,
                               uint64_t key_mask, unsigned long long *d_size, hipStream_t stream);

int open_dir(const char *db_dir, int device, Engine **out);
int open_images(const void *opts, size_t opts_len, const void *taxo, size_t taxo_len,
                const void *hash, size_t hash_len, int device, Engine **out);
int open_synthetic(uint64_t capacity, uint64_t n_keys, uint32_t depth, uint64_t seed, int device,
                   Engine **out);
void destroy(Engine *e);
void finish_devdb_public(Engine *e);
int refresh_table_copies(Engine *e);  // after the cells of copy 0 changed (load, inserts): device-wide sync + re-copy
int resolve_db_dir(const char *db_dir, std::string &resolved);
int classify_device(Engine *e, const void *d_bases, const void *d_seq_off, uint64_t n_frag,
                    uint32_t flags, double confidence, void *d_results, void *d_kmer_taxa,
                    const void *d_kmer_taxa_off, void *d_counters, hipStream_t stream,
                    const void *d_seq_len = nullptr, uint64_t bases_end = 0);
int classify_host(Engine *e, const uint8_t *bases, const uint64_t *seq_offsets, uint64_t n_frag,
                  uint32_t flags, double confidence, nh_result *results, uint32_t *kmer_taxa,
                  uint64_t *kmer_taxa_offsets, uint64_t kmer_taxa_cap);
int check_error_flag(Engine *e);
// buffers kept between the runs of a process (nh_run.hip: page-locked batch text; nh_gunzip.hip: the gzip reader's HBM and
// staging): emptied when an engine is closed
void run_cache_trim();
void dev_cache_trim();
// ---- devices.  Every device id in this library (nh_open's `device`, nh_run_args.device_ids, Engine::device, a batch's
// dev_device ...) is a LOGICAL device; dev_set() is the only way a thread selects one.  By default logical == HIP ordinal.
// NOHUMAN_FAKE_DEVICES=N (test mode for a box with one GPU): N logical devices, all on HIP device 0 -- the run's engines,
// reader lanes, encoders, streams and buffers then have N distinct owners, copies between them take the peer route, and
// NOHUMAN_DEBUG_DEVICE=1 checks the discipline a real multi-GPU node enforces by failing: dev_check() that the calling
// thread's current logical device is the owner's at every launch / copy / allocation site, dev_check_ptr() that a buffer
// was allocated under its owner (this library's allocations are registered; hipPointerGetAttributes has the last word on
// the physical device).  The first violation is kept (dev_violation()) and fails the run that meets it.
int dev_count();                 // logical devices (<= 0: none / HIP error)
int dev_phys(int ldev);          // HIP ordinal behind a logical device
hipError_t dev_set(int ldev);    // hipSetDevice(dev_phys(ldev)) and the thread's logical device
int dev_current();               // the thread's logical device (-1: none selected yet)
bool dev_debug();
void dev_check(int owner, const char *where);
void dev_check_ptr(const void *p, int owner, const char *where);
std::string dev_violation(bool clear = false);
// n bytes from device memory of src_ldev to device memory of dst_ldev on `stream` (a stream of dst_ldev, the calling
// thread's current device): the same logical device: an ordinary device-to-device copy; otherwise hipMemcpyPeerAsync --
// peer access between the two HIP devices is enabled the first time where hipDeviceCanAccessPeer allows it (without it the
// runtime stages the copy itself) -- or, under NOHUMAN_NO_PEER=1, through a page-locked host buffer (D2H, H2D; synchronous).
hipError_t dev_copy_between(void *dst, int dst_ldev, const void *src, int src_ldev, size_t n, hipStream_t stream);
// Every allocation of the run path goes through these two: when the device (or the page-locked pool) is full, what the
// process keeps between runs is given back first and the allocation tried once more -- idle buffers never fail a run.
hipError_t dev_malloc(void **p, size_t bytes);
hipError_t host_malloc(void **p, size_t bytes, unsigned flags = hipHostMallocDefault);
template <class T>
inline hipError_t dev_malloc(T **p, size_t bytes) {
    return dev_malloc((void **)p, bytes);
}
template <class T>
inline hipError_t host_malloc(T **p, size_t bytes, unsigned flags = hipHostMallocDefault) {
    return host_malloc((void **)p, bytes, flags);
}
// the path's only collective: rows[g] = counters of device ids[g] -> every row = the sum (RCCL, nh_collective.hip)
// (d_src[g] != NULL: the four counters lie in device ids[g]'s memory and are reduced from there)
int allreduce_counters(const int *ids, int n_dev, uint64_t *rows, std::string &backend,
                       const uint64_t *const *d_src = nullptr);
uint64_t kmer_taxa_entries(const Engine *e, const uint64_t *seq_offsets, uint64_t n_frag, int mates,
                           uint64_t *offsets_out);

}  // namespace nh
