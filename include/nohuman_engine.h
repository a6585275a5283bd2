/*
 * nohuman_engine.h -- C ABI of libnohuman_engine.so, the MI355X (gfx950) in-process replacement
 * for the `kraken2` subprocess that nohuman spawns.
 *
 * What it replaces in the reference (file:line under /root/reference):
 *   - the process boundary  CommandRunner::run -> Command::new("kraken2").args(..).output()
 *     (src/lib.rs:22-48) called once, blocking, from main() (src/main.rs:270);
 *   - the information carried by the argv built at src/main.rs:210-267
 *     (--threads, --db, --output, --confidence, --report, --paired,
 *      --classified-out | --unclassified-out, input paths);
 *   - the three integers scraped from kraken2's stderr by parse_kraken_stderr
 *     (src/lib.rs:61-97), which the engine returns directly in nh_stats;
 *   - the dependency check CommandRunner::is_executable (src/lib.rs:50-57) -> nh_probe;
 *   - the database directory contract validate_db_directory (src/lib.rs:119-141) -> nh_open.
 *
 * Conventions: every function returns 0 on success or a negative nh_status; nothing throws or
 * aborts across the ABI; the caller owns every buffer it passes; the engine handle is opaque and
 * only created/destroyed by nh_open... / nh_close; nh_last_error() is a thread-local string valid
 * until the next call on that thread.  There is NO CPU fallback: without a usable gfx950 device
 * nh_open / nh_probe fail with NH_EDEVICE.
 */
#ifndef NOHUMAN_ENGINE_H
#define NOHUMAN_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NH_ABI_VERSION 5

typedef enum {
    NH_OK = 0,
    NH_EINVAL = -1,   /* bad argument */
    NH_EIO = -2,      /* file could not be read / written */
    NH_EDB = -3,      /* malformed or unsupported database (e.g. protein DB) */
    NH_EDEVICE = -4,  /* no usable gfx950 device, HIP error */
    NH_EOOM = -5,     /* host or device allocation failed */
    NH_ECAPACITY = -6 /* an internal per-fragment capacity was exceeded (reported, never silent) */
} nh_status;

typedef struct nh_engine nh_engine;

/* One record per fragment (read or read pair).  16 bytes; the "result record" of the roofline
 * formula in BASELINE.md section 4.  Replaces the C/U column, the taxid column and the hit list
 * totals of kraken2's per-read output line. */
typedef struct {
    uint32_t call;        /* internal taxon id of the call; 0 = unclassified (kept by nohuman) */
    uint32_t total_kmers; /* k-mers of both mates, ambiguous ones included (confidence denominator) */
    uint32_t clade_hits;  /* k-mer hits in the clade rooted at `call` (confidence numerator) */
    uint32_t hit_groups;  /* kraken2 minimizer_hit_groups */
} nh_result;

/* per-k-mer taxa list markers (the "A:n" and "|:|" items of kraken2's hit list) */
#define NH_TAXON_AMBIGUOUS 0xFFFFFFFFu
#define NH_TAXON_MATE_BORDER 0xFFFFFFFEu

/* Replaces the three summary lines of kraken2's stderr (src/lib.rs:67-89). */
typedef struct {
    uint64_t total_sequences; /* fragments processed */
    uint64_t classified;      /* "sequences classified" == human for nohuman */
    uint64_t unclassified;
    uint64_t total_bases;
    uint64_t table_lookups;   /* D of the roofline formula, summed */
    double seconds;           /* classify wall time, database load excluded (as kraken2's timer) */
} nh_stats;

typedef struct {
    uint64_t k, l, spaced_seed_mask, toggle_mask, minimum_acceptable_hash_value;
    int32_t revcom_version, dna_db;
    uint64_t capacity, size, key_bits, value_bits;
    uint64_t node_count;
    int32_t device;
    int32_t reserved;
} nh_db_info;

/* Behaviour switches; defaults reproduce kraken2 as nohuman invokes it (src/main.rs:215-224:
 * no --minimum-hit-groups, no --quick, no quality masking). */
typedef struct {
    uint32_t minimum_hit_groups; /* default 2 */
    int32_t linear_probing;      /* default 1 (kraken2 builds with -DLINEAR_PROBING) */
    int32_t reset_per_mate;      /* default 1 (last minimizer/taxon reset for each mate) */
    /* ABI 4 (ABI 2: `reserved`, 0; ABI 3: 0 / 1): which k-mers next to an ambiguous base count as ambiguous ("A:n" in the
     * hit list, no look-up) -- the two recollections of kraken2's scanner, switchable until a binary has been diffed
     * (BASELINE.md section 2).  ZERO MEANS "THE ENGINE'S DEFAULT": a caller that fills the struct
     * from scratch and leaves the former `reserved` word 0 keeps the pinned default instead of silently selecting a rule:
     *   NH_AMBIGUITY_ENGINE_DEFAULT (0)  whatever NH_AMBIGUITY_DEFAULT names (nh_options_get reports the rule in force);
     *   NH_AMBIGUITY_LAST_LMER (1)  the bool* flag of MinimizerScanner::NextMinimizer: an ambiguous byte among the
     *                               k-mer's last l bases; an isolated N costs l = 31 k-mers, the next three are
     *                               looked up with windows of 1, 2, 3 l-mers;
     *   NH_AMBIGUITY_QUEUE (2)      mmscanner.h is_ambiguous() = (queue_pos_ < k_ - l_) || !!last_ambig_, what
     *                               classify.cc / build_db.cc ask: also ambiguous until k - l l-mers have been queued
     *                               behind the base, i.e. an ambiguous byte among the last k - 1 bases; an isolated N
     *                               costs k - 1 = 34 k-mers.  The default. */
    int32_t ambiguity_rule;
} nh_options;
#define NH_AMBIGUITY_ENGINE_DEFAULT 0
#define NH_AMBIGUITY_LAST_LMER 1
#define NH_AMBIGUITY_QUEUE 2
#define NH_AMBIGUITY_DEFAULT NH_AMBIGUITY_QUEUE

/* flags of nh_classify_* */
#define NH_FLAG_PAIRED 1u /* sequences 2f and 2f+1 are the mates of fragment f (--paired) */
#define NH_FLAG_LONG 2u   /* scheduling hint: fragments are long (kilobases); hand them out one by one */

const char *nh_last_error(void);
int nh_abi_version(void);

/* 0 if a gfx950 device is usable; msg receives a one-line description either way.
 * Replaces CommandRunner::is_executable (src/lib.rs:50-57), used by `nohuman --check`. */
int nh_probe(char *msg, size_t msg_len);
int nh_device_count(int *count);

/* Load <db_dir>/{hash,opts,taxo}.k2d -- or <db_dir>/db/... (src/lib.rs:119-141) -- into the HBM
 * of `device`.  Replaces kraken2's "Loading database information..." phase.
 * The CONTENT is checked too (ABI 5), because the kernels index the taxonomy with the cells' values and nohuman can have
 * several database versions installed side by side (src/download.rs:178-222): every cell is read once on the device
 * (about a millisecond per 5 GB); a value >= taxo.k2d's node count, a count of non-empty cells that is not the header's
 * `size`, or a taxonomy whose parent ids do not lie below their children fail with NH_EDB -- never with a GPU fault. */
int nh_open(const char *db_dir, int device, nh_engine **out);
/* Same from in-memory images of the three files (borrowed for the duration of the call). */
int nh_open_images(const void *opts, size_t opts_len, const void *taxo, size_t taxo_len,
                   const void *hash, size_t hash_len, int device, nh_engine **out);
/* Bench/test support: a database whose hash table is generated directly in HBM (n_keys pseudo-
 * random minimizers inserted with kraken2's CompareAndSet linear-probing rule; every value is the
 * deepest node of a `depth`-node chain taxonomy).  Stands in for HPRC.r2, which cannot be
 * downloaded on the build or GPU boxes. */
int nh_open_synthetic(uint64_t capacity, uint64_t n_keys, uint32_t depth, uint64_t seed, int device,
                      nh_engine **out);
/* Bench/test support: inserts every minimizer of the given device-resident sequences into the
 * table with internal taxon id `value` (kraken2 build semantics: CompareAndSet, linear probing), so
 * that reads drawn from them hit.  Default k=35/l=31 geometry only. */
int nh_synthetic_add_sequences(nh_engine *e, const void *d_bases, const void *d_seq_offsets,
                               uint64_t n_seq, uint32_t value, void *stream);
int nh_close(nh_engine *e);

int nh_db_info_get(const nh_engine *e, nh_db_info *info);
/* What the content check of nh_open* measured (ABI 5). */
typedef struct {
    uint64_t non_empty_cells; /* cells whose value field is not 0; == nh_db_info.size of a database that opened */
    uint64_t max_value;       /* largest value field in the table; < nh_db_info.node_count */
    double load_factor;       /* non_empty_cells / capacity */
    double seconds;           /* what the pass cost */
} nh_db_check;
int nh_db_check_get(const nh_engine *e, nh_db_check *c);
/* The defaults of a freshly opened engine can be overridden per process by NOHUMAN_OPT_AMBIGUITY_RULE,
 * NOHUMAN_OPT_LINEAR_PROBING, NOHUMAN_OPT_RESET_PER_MATE, NOHUMAN_OPT_MIN_HIT_GROUPS (integers): how
 * scripts/parity_vs_kraken2.sh walks the switch lattice through nh_run and the CLI host. */
int nh_options_get(const nh_engine *e, nh_options *o);
int nh_options_set(nh_engine *e, const nh_options *o);
/* internal taxon id -> external (NCBI) id, as printed in kraken2's output / "kraken:taxid|N" */
int nh_taxon_external(const nh_engine *e, uint32_t internal, uint64_t *external);
/* copies of the database images back to the host (tests and the bench's CPU-baseline leg) */
int nh_table_download(const nh_engine *e, uint32_t *cells, uint64_t n_cells);
int nh_taxonomy_image(const nh_engine *e, void *buf, size_t cap, size_t *len);
int nh_opts_image(const nh_engine *e, void *buf, size_t cap, size_t *len);

/*
 * Classify one batch held in HOST memory; blocking.  Replaces kraken2's per-block loop
 * (classify.cc ProcessFiles/ClassifySequence) for the reads of one batch.
 *   bases        concatenated sequence bytes exactly as in the FASTQ/FASTA sequence lines
 *   seq_offsets  n_seq+1 offsets into bases; n_seq = n_frag * (paired ? 2 : 1)
 *   confidence   the f64 value kraken2 would parse from --confidence (src/main.rs:213)
 *   results      n_frag records
 *   kmer_taxa / kmer_taxa_offsets (both NULL or both set): per-k-mer internal taxon ids with the
 *                two markers above -- the material of the `-k` hit list.
 *                kmer_taxa_offsets[f] = sum over fragments < f of (k-mers of both mates + paired);
 *                the caller sizes kmer_taxa with nh_kmer_taxa_entries.
 */
int nh_classify_batch(nh_engine *e, const uint8_t *bases, const uint64_t *seq_offsets,
                      uint64_t n_frag, uint32_t flags, double confidence, nh_result *results,
                      uint32_t *kmer_taxa, uint64_t *kmer_taxa_offsets, uint64_t kmer_taxa_cap);
uint64_t nh_kmer_taxa_entries(const nh_engine *e, const uint64_t *seq_offsets, uint64_t n_frag,
                              uint32_t flags);

/*
 * Same with every buffer already resident in the engine device's HBM; asynchronous on `stream`
 * (a hipStream_t, NULL = the default stream).  d_bases must be 4-byte aligned and readable for
 * 8 bytes past the last base.  d_kmer_taxa / d_kmer_taxa_offsets may be NULL.  d_counters (may be
 * NULL) points at 4 uint64 accumulators {fragments, classified, bases, table_lookups} that the
 * kernel adds to.  This is the entry the roofline number of bench.py is measured on.
 * Thread-safe per engine: any number of host threads may call it on one engine, each on its own stream.  The engine has
 * 16 launch slots; a launch that finds its slot still taken by the launch sixteen before it waits for that one ON THE
 * DEVICE (an event wait queued on `stream`), so more than 16 launches in flight are ordered, never mixed.
 */
int nh_classify_batch_device(nh_engine *e, const void *d_bases, const void *d_seq_offsets,
                             uint64_t n_frag, uint32_t flags, double confidence, void *d_results,
                             void *d_kmer_taxa, const void *d_kmer_taxa_offsets, void *d_counters,
                             void *stream);

/*
 * The same launch with the sequences classified IN PLACE inside a device-resident buffer of record text
 * (what nh_run does with the FASTQ / FASTA text of a batch, copied to the device as read -- no pass that
 * gathers the sequence lines): sequence i is d_text[d_seq_starts[i], +d_seq_lens[i]) (uint64 / uint32,
 * n_frag * (paired ? 2 : 1) entries each; pairs: entries 2f and 2f+1; the first sequence must be the
 * one at the lowest address and all of a launch must lie within 4 GB).  Whatever surrounds a sequence
 * (header, '+' line, qualities, newlines) is never interpreted.  d_text must be 4-byte aligned and readable
 * for 8 bytes past text_len.
 */
int nh_classify_records_device(nh_engine *e, const void *d_text, uint64_t text_len, const void *d_seq_starts,
                               const void *d_seq_lens, uint64_t n_frag, uint32_t flags, double confidence,
                               void *d_results, void *d_kmer_taxa, const void *d_kmer_taxa_offsets,
                               void *d_counters, void *stream);

/* running totals over every nh_classify_batch* call on this engine since open / reset */
int nh_stats_get(nh_engine *e, nh_stats *s);
int nh_stats_reset(nh_engine *e);

/* container of the kept reads (CompressionFormat, /root/reference/src/compression.rs:12-20) */
typedef enum nh_codec {
    NH_CODEC_NONE = 0,
    NH_CODEC_BZIP2 = 1,
    NH_CODEC_GZIP = 2,
    NH_CODEC_XZ = 3,
    NH_CODEC_ZSTD = 4 /* through the system's libzstd.so.1 (level 3, frame checksum, `threads` workers) */
} nh_codec;

/*
 * Whole-run entry: the information of the argv at src/main.rs:210-267, outputs written
 * UNCOMPRESSED (out_codec 0) to the given paths exactly where kraken2 would write kraken_out.fq /
 * kraken_out_1.fq + kraken_out_2.fq (src/main.rs:252-256,308-309,333), so nohuman's compress stage
 * (src/main.rs:342-368) is untouched -- or, with out_codec set, already in their final container.
 */
typedef struct {
    const char *db_dir;        /* --db */
    const char *in1;           /* first input (plain or gzip FASTQ/FASTA) */
    const char *in2;           /* second input or NULL (--paired when set) */
    const char *out1;          /* kraken_out.fq / kraken_out_1.fq */
    const char *out2;          /* kraken_out_2.fq or NULL */
    const char *kraken_output; /* --output; NULL or "/dev/null" = none */
    const char *report;        /* --report; NULL = none */
    double confidence;         /* --confidence */
    uint32_t threads;          /* --threads: host reader/writer workers */
    int32_t keep_human;        /* 0: --unclassified-out (default), 1: --classified-out (-H) */
    int32_t n_devices;         /* 0 = all visible devices */
    const int32_t *device_ids; /* NULL = 0..n_devices-1 */
    /* ABI 2 -- SURVEY.md 8f-4: the kept reads can leave the engine already compressed, written straight to
     * their final paths by a streaming encoder, which removes the temporary uncompressed files and the
     * second pass of the reference's compress stage (src/compression.rs:182-268, src/main.rs:342-368).
     * 0 (NH_CODEC_NONE) keeps the kraken2 behaviour: plain text at out1 / out2. */
    int32_t out_codec;         /* nh_codec of out1 / out2 */
    uint32_t codec_threads;    /* encoder workers per output file (0 = 1) */
} nh_run_args;

int nh_run(const nh_run_args *args, nh_stats *stats);
/* Host-side check of the sequence reader nh_run uses (kraken2 record semantics, SURVEY.md A.6;
 * plain / gzip / bzip2, FASTA / FASTQ): number of records, total bases, and an FNV-1a-64 digest
 * over every record's header line, sequence and qualities (each followed by one 0 byte).
 * Needs no GPU. */
int nh_fastx_scan(const char *path, uint64_t *n_records, uint64_t *n_bases, uint64_t *digest);
/* Output compression stage: replaces CompressionFormat::compress
 * (/root/reference/src/compression.rs:182-200, called from src/main.rs:342-368).  Compresses file
 * `in` to file `out`; gzip runs block-parallel on `threads` workers like the reference's gzp
 * encoder (compression.rs:214-233) and produces one ordinary gzip member at level 6; bzip2 and xz
 * go through libbz2.so.1 / liblzma.so.5 (one thread / `threads` workers, preset 6, CRC64), zstd through
 * libzstd.so.1; NH_CODEC_NONE copies.  Parity target is the decompressed content and
 * the container magic (compression.rs:282-288).  Needs no GPU. */
int nh_compress_file(const char *in, const char *out, int codec, uint32_t threads);
/* The gzip case of that stage on the GPU (nohuman_amd/csrc/nh_deflate.hip; replaces gzip_compress,
 * /root/reference/src/compression.rs:214-233): n bytes at `in` (host memory) become ONE ordinary gzip member in
 * file `out` -- 64 KiB regions of the text, a wave each, dynamic Huffman blocks, regions joined by empty stored
 * blocks the way gzp / pigz join theirs.  It is what nh_run's writer feeds when out_codec is NH_CODEC_GZIP
 * (NOHUMAN_GZIP=host selects the host encoder above).  stats (may be NULL): [0] bytes of the file, [1] microseconds
 * of kernel time (HIP events).  Parity target as above: the decompressed content. */
int nh_gzip_gpu_file(int32_t device, const void *in, uint64_t n, const char *out, uint64_t *stats);
/* nh_compress_file for a host that keeps the reference's two stages (kraken2-style temporary file, then compress,
 * src/main.rs:342-368) but has the GPU at hand: NH_CODEC_GZIP is encoded on `device` as above, the other codecs as in
 * nh_compress_file. */
int nh_compress_file_device(const char *in, const char *out, int codec, uint32_t threads, int32_t device);
/* Test / tool support for the multi-threaded gzip input decoder nh_run reads .gz inputs with
 * (kraken2's wrapper pipes them through `gzip -dc`): decompress `in` to
 * `out` on `threads` workers, cutting the compressed file every chunk_bytes (0 = default).
 * stats3 (optional) receives {chunks accepted, chunks rejected, bytes decoded in order by the
 * consumer}.  Needs no GPU. */
int nh_gunzip_file(const char *in, const char *out, uint32_t threads, uint64_t chunk_bytes, uint64_t *stats3);
/* The gzip READER on the GPU (nohuman_amd/csrc/nh_gunzip.hip; what nh_run reads .gz inputs with: replaces the `gzip -dc`
 * pipe of kraken2's wrapper): decompress `in` to `out` on `device` -- block search, decode to
 * 16-bit symbols with markers, the chunks' windows by a prefix scan, marker replacement and the members' CRC-32 on the
 * device (BGZF files: the chunks' starts from the members' headers, no search); seg_bytes / stretch_bytes = compressed
 * bytes per piece / per chunk (0 = defaults).  stats8 (optional):
 * {pieces, chunks, chunks decoded again after a false block start, pieces the host decoder took over, members, text
 * bytes, gzip bytes, kernel microseconds (NOHUMAN_TRACE only)}.  Test / tool support like nh_gunzip_file. */
int nh_gunzip_device_file(const char *in, const char *out, int32_t device, uint64_t seg_bytes, uint64_t stretch_bytes,
                          uint64_t *stats8);
/* Bytes of idle buffers the process keeps between runs for the gzip reader on `device` (page_locked != 0: its
 * page-locked staging, all devices): bounded by bytes, oldest evicted first, given back before any allocation of the
 * run path fails, emptied by nh_close.  Diagnostic / test support. */
uint64_t nh_cache_bytes(int32_t device, int32_t page_locked);
/* The one collective of the path: `counters` holds n_devices rows of 4 uint64
 * {fragments, classified, bases, table_lookups}, row g being the totals of device_ids[g] (NULL =
 * 0..n_devices-1); on return every row is the sum over the devices -- one ncclAllReduce(4 x uint64, sum)
 * over RCCL / xGMI, single process, ncclCommInitAll.  nh_run calls it at the end of a multi-device run
 * (and checks it against the host-side sum, which stays the fallback when RCCL cannot be loaded).
 * `backend` (optional) receives a one-line description of what ran.  Replaces nothing in the
 * reference (kraken2 sums its thread-local counters with atomics); it is the multi-GPU form of the
 * three integers of src/lib.rs:61-97. */
int nh_allreduce_counters(const int32_t *device_ids, int32_t n_devices, uint64_t *counters, char *backend,
                          size_t backend_len);
/* nh_run on an already opened engine (single device) */
int nh_run_engine(nh_engine *e, const nh_run_args *args, nh_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* NOHUMAN_ENGINE_H */
