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
    int32_t ambig_rule;     // nh_options.ambiguity_rule: 0 = ambiguous byte in the last l bases, 1 = in the last k-1 (scan_body)
    uint32_t max_chunks;    // bound of the linear-probe loop in rounds (>= 1 cell per round)
};

// per-launch device counters (uint64 each)
enum { CNT_FRAGMENTS = 0, CNT_CLASSIFIED = 1, CNT_BASES = 2, CNT_LOOKUPS = 3, CNT_N = 4 };

struct Result {
    uint32_t call, total_kmers, clade_hits, hit_groups;
};

// How the fragments of a launch are handed out: claim i of the launch's work counters is a chunk of c0
// fragments while i < n0 (the body), of c1 while i < n01, of c2 up to `total` (the guided tail: smaller chunks
// for the waves that finish early; a fixed map, a claim index is also the chunk's bit in KArgs::defer_bits).
// make_sched in nh_kernels.hip says what it is worth; n0 == total is the flat map (NOHUMAN_SCHED=off).
struct Sched {
    uint64_t n0, n01, total;  // claims in chunks of c0 / up to here in chunks of c1 / all claims
    uint64_t base1, base2;    // first fragment of the c1 chunks, of the c2 chunks
    uint32_t c0, c1, c2, pad;
};

// One hot word is a bottleneck on this chip: a device-scope atomic on ONE address retires at 46-88 per
// microsecond (profiles/r03_sched.txt), and the memory pipeline of a CU that holds such atomics backs up behind
// them.  Round 2 had every wave of a launch add its four counters to the caller's four words as it ended:
// 20480 atomics on one 32-byte stretch in the last ~300 us of every launch, which is where the launches' "tail"
// came from (a wave's last chunk took 100+ us longer than any other).  Now the waves add to one of
// COUNTER_SHARDS rows (by workgroup), each in a line of its own, and a one-wave kernel behind the launch folds
// the rows into the caller's counters: the fixed cost of a launch fell from 0.30 to 0.13 ms.
// The WORK counter: one word hands out the claims of the launch's BODY (chunks of c0; ~27 claims per
// microsecond in steady state are no problem for one word, and one word balances the XCDs, which differ by
// ~8 % in speed).  Sharding it per XCD for the whole launch was tried and is slower (+3 % per read: the faster
// XCDs run out of their share early and every later claim of theirs starts with a failed atomic).  The claims
// of a guided TAIL (Sched: chunks of c1, c2 < c0, NOHUMAN_SCHED) come four times as often in the launch's last
// moments -- more than one word retires -- so they are dealt from TAIL_SHARDS words: tail claim j belongs to
// shard j mod TAIL_SHARDS, a wave draws from its XCD's shard and moves on to the next when that is empty.
constexpr uint32_t TAIL_SHARDS = 8;                 // a power of two
constexpr uint32_t WORK_WORDS = 1 + TAIL_SHARDS;    // counters of a launch: body, then the tail shards
constexpr uint32_t WORK_STRIDE = 16;                // uint64 words from one counter to the next (128 bytes)
constexpr uint32_t WORK_PASSES = 3;                 // short-read / first pass, deferred pass, BIG pass: a set of counters each
constexpr uint32_t COUNTER_SHARDS = 64;
constexpr uint32_t COUNTER_STRIDE = 16;   // uint64 words per row (128 bytes; four are used)

// ---- long reads: work items instead of whole fragments ------------------------------------------------
// A launch of long single-end reads (NH_FLAG_LONG) is handed out as ITEMS built by a prepass (k_prep_items):
// a read of more than SPLIT_MIN_TILES tiles is cut into segments of SEG_TILES tiles that any wave can
// classify -- a 200 kb read is 50 items, not one wave's 10 ms -- and the items go out largest first: all
// segments, then the whole reads of 8 tiles and more, then the small ones.  A segment (a) finds kraken2's
// last_minimizer at its start by scanning the tile before it (further back while that one holds no
// unambiguous k-mer), (b) looks that minimizer up once, uncounted, for last_taxon, (c) leaves its (taxon,
// count) list and hit groups in a 256-byte PARTIAL slot; the segment that finishes last (a counter per
// fragment) adds the partials up and resolves the fragment.  More than PART_CAP distinct taxa in one
// segment: the fragment is left to the BIG variant (whole fragment, one wave), as a list overflow is.
constexpr uint32_t SEG_TILES = 32;        // tiles per segment: 3968 k-mers at k=35/l=31
constexpr uint32_t SPLIT_MIN_TILES = 48;  // reads of more tiles than this are cut
constexpr uint32_t MID_TILES = 8;         // whole reads of at least this many tiles go out before the small ones
constexpr uint32_t PART_DWORDS = 64;      // a partial: [0] hit groups, [1] entries, [2] overflow, then (taxon, count) pairs from [4]
constexpr uint32_t PART_CAP = 30;
struct SplitItem {
    uint32_t f, seg, nseg, slot;  // fragment, segment, segments of the fragment, first partial slot of the fragment
};
struct SplitHdr {
    uint32_t n_multi, n_mid, n_small, seg_used;
};
struct SplitBufs {
    SplitHdr *hdr;            // NULL: the launch is handed out by fragments (Sched)
    SplitItem *items_multi;   // [seg_cap]
    uint32_t *items_single;   // [n_frag]: reads of >= MID_TILES tiles from the front, smaller ones from the back
    uint32_t *part;           // [seg_cap][PART_DWORDS]
    uint32_t *part_done;      // [seg_cap]: segments finished, at the fragment's first slot
    uint32_t seg_cap, pad;
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
    uint32_t frag_chunk;  // = sched.c0: the largest number of fragments a wave claims at a time
    Sched sched;
    double confidence;
    Result *out;
    uint32_t *kmer_taxa;
    const uint64_t *kmer_taxa_off;
    unsigned long long *counters;  // the caller's {fragments, classified, bases, lookups} (may be NULL)
    unsigned long long *cshard;    // COUNTER_SHARDS rows the waves add to; folded into `counters` behind the launch
    int *error_flag;  // sticky error bits of the engine (1: > 2048 distinct taxa, 2: sequence too long)
    int *pending;     // per launch slot: fragments were left to the BIG variant
    // short-read kernel -> generic kernel: chunks that hold a sequence of more than one tile (one bit per
    // chunk of frag_chunk fragments), and "there are such chunks"
    uint32_t *defer_bits;
    int *pending_long;
    int32_t only_deferred;  // generic kernel: classify only the chunks marked in defer_bits
    unsigned long long *work;
    SplitBufs split;
    // tuning aid (tools/timeline.py; NULL in normal use): 32 x uint64 per wave of k_classify_short -- 100 MHz
    // timestamps of start, first claim, first batch encoded, first probe phase done, last claim, end; chunks
    // taken; XCC id; then (start << 8 | fragments) of the first 24 chunks
    unsigned long long *timeline;
};

// ---- host side of a launch ----
// NOHUMAN_SCHED, parsed once when the engine is opened (nh_internal.h LaunchKnobs): "off" = the flat claim map, or
// c1,c2,p1,p2 = sizes of the guided tail's chunks and its two lengths in percent
struct SchedKnobs {
    bool set = false, off = false;
    uint32_t c1 = 0, c2 = 0, p1 = 100, p2 = 100;
};
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
    unsigned long long *d_work = nullptr;    // WORK_PASSES x WORK_WORDS counters, WORK_STRIDE words apart
    unsigned long long *d_cshard = nullptr;  // COUNTER_SHARDS rows of COUNTER_STRIDE words
    uint32_t *d_defer = nullptr;        // one bit per chunk
    uint64_t defer_cap_bits = 0;
    SplitBufs split = {};               // long-read item buffers of this slot (hdr == NULL: none)
    uint64_t split_single_cap = 0;      // entries of split.items_single
    bool split_fresh = false;           // split.hdr has never been cleared: the launch clears it first
};

constexpr uint32_t TAXON_AMBIGUOUS = 0xFFFFFFFFu;
constexpr uint32_t TAXON_MATE_BORDER = 0xFFFFFFFEu;

constexpr int WAVE = 64;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int TL = 128;      // l-mers per tile (2 per lane)
constexpr int LIST_CAP = 64; // distinct taxa per fragment held in LDS

}  // namespace nh
