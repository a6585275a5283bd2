// nh_device.h -- device-side parameter block shared by the gfx950 kernels and the host launcher.
#pragma once
#include <stdint.h>

namespace nh {

// Everything a classify wave needs to know about the database, passed by value as a kernel
// argument (lives in SGPRs / the kernarg segment; no global loads for it).
struct DevDB {
    const uint32_t *table;  // hash.k2d cells, capacity entries (+ padding to a multiple of 4, +32); 128-byte aligned
    // The table is resident n_copies times (1, 2, 4 or 8).  Copy j starts copy_stride cells after copy
    // j-1 and lies 32/n_copies cells (mod 32) further to the LEFT on the 128-byte line grid, so cell i
    // of copy j sits (i - j * 32/n_copies) mod 32 cells into its line.  A probe round at cell p uses
    // copy (p mod 32) / (32/n_copies): there p lies in the first 32/n_copies cells of a line and the
    // run has at least 32 - 32/n_copies + 1 cells before it leaves the line -- the unit the fabric
    // fetches (profiles/r02_pair_study.txt: the gather ceiling is a rate of 128-byte lines).
    uint64_t copy_stride;   // cells from one copy to the next
    uint32_t n_copies;
    uint32_t copy_shift;    // log2(32 / n_copies): 5, 4, 3 or 2
    uint64_t capacity;
    uint64_t cap_magic;     // floor((2^64 - 1) / capacity): exact `hc % capacity` without a divide
    const uint32_t *parent; // taxonomy: internal parent ids [node_count]
    uint32_t node_count;
    uint32_t value_bits;
    uint32_t vmask;
    uint32_t k, l;
    uint32_t window;        // k - l
    uint64_t lmer_mask;     // low 2l bits
    uint64_t spaced_mask;   // spaced_seed_mask, or ~0 when the DB has none
    uint64_t toggle;        // toggle_mask & lmer_mask
    uint64_t min_hash;      // minimum_acceptable_hash_value
    int32_t revcom_version;
    int32_t linear_probing;
    int32_t reset_per_mate;
    uint32_t min_hit_groups;
    uint32_t max_chunks;    // bound of the linear-probe loop in rounds (>= 1 cell per round)
};

// per-launch device counters (uint64 each)
enum { CNT_FRAGMENTS = 0, CNT_CLASSIFIED = 1, CNT_BASES = 2, CNT_LOOKUPS = 3, CNT_N = 4 };

struct Result {
    uint32_t call, total_kmers, clade_hits, hit_groups;
};

// The one kernel argument of k_classify.  Device code never names the parameter: it reads the
// fields it needs, phase by phase, through the kernarg-segment pointer, so that the ~50 dwords of
// arguments are not all kept live in SGPRs for the whole kernel (see nh_kernels.hip).
struct KArgs {
    DevDB db;
    const uint8_t *bases;
    // sequence i = bases[seq_off[i], seq_off[i+1])  (seq_len == NULL, n_seq + 1 offsets), or
    //            = bases[seq_off[i], seq_off[i] + seq_len[i])  (n_seq offsets: records classified in place
    //              inside their FASTQ / FASTA text, which is bases_end bytes long)
    const uint64_t *seq_off;
    const uint32_t *seq_len;
    uint64_t bases_end;
    uint64_t n_frag;
    int32_t mates;
    uint32_t frag_chunk;
    double confidence;
    Result *out;
    uint32_t *kmer_taxa;
    const uint64_t *kmer_taxa_off;
    unsigned long long *counters;
    int *error_flag;  // sticky error bits of the engine (1: > 2048 distinct taxa, 2: sequence too long)
    int *pending;     // per launch slot: fragments were left to the BIG variant
    // short-read kernel -> generic kernel: chunks that hold a sequence of more than one tile (one bit per
    // chunk of frag_chunk fragments), and "there are such chunks"
    uint32_t *defer_bits;
    int *pending_long;
    int32_t only_deferred;  // generic kernel: classify only the chunks marked in defer_bits
    unsigned long long *work;
};

// ---- host side of a launch ----
// what one classify launch reads and writes (device pointers)
struct LaunchIO {
    const void *d_bases = nullptr;
    const void *d_seq_off = nullptr;   // n_seq + 1 offsets, or n_seq starts when d_seq_len is set
    const void *d_seq_len = nullptr;   // NULL, or n_seq uint32 lengths (sequences in place inside their records' text)
    uint64_t bases_end = 0;            // with d_seq_len: bytes of d_bases (readable for 8 more)
    uint64_t n_frag = 0;
    int mates = 1;
    bool long_reads = false;           // hint: skip the short-read kernel
    void *d_out = nullptr, *d_kmer_taxa = nullptr;
    const void *d_kmer_taxa_off = nullptr;
    void *d_counters = nullptr;
};
// the per-launch words of one of the engine's LAUNCH_SLOTS
struct LaunchSlot {
    int *d_error = nullptr;             // sticky error bits of the engine
    int *d_pending = nullptr;           // fragments were left to the BIG variant
    int *d_pending_long = nullptr;      // chunks were left to the generic kernel
    unsigned long long *d_work = nullptr;
    uint32_t *d_defer = nullptr;        // one bit per chunk
    uint64_t defer_cap_bits = 0;
};

constexpr uint32_t TAXON_AMBIGUOUS = 0xFFFFFFFFu;
constexpr uint32_t TAXON_MATE_BORDER = 0xFFFFFFFEu;

constexpr int WAVE = 64;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int TL = 128;      // l-mers per tile (2 per lane)
constexpr int LIST_CAP = 64; // distinct taxa per fragment held in LDS

}  // namespace nh
