/*
 * k2_oracle.h -- CPU restatement of the kraken2 `classify` hot path (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED vs kraken2: the algorithm lives in the third-party program
 *   github.com/DerrickWood/kraken2 @ f885f832c9863704638dece6c29f0fa4bc19a2e3 (K2VER 2.17),
 * which nohuman only pins in its container recipe (/root/reference/Dockerfile:15,35-38) and
 * spawns as a subprocess (/root/reference/src/main.rs:170,215-270; src/lib.rs:22-48).  Neither
 * its source nor a binary nor a .k2d database exists in the build container, and the reference's
 * own tests hold no classification vectors.  This file restates
 * kraken2's published algorithm (mmscanner.cc NextMinimizer, kv_store.h MurmurHash3,
 * compact_hash.cc Get, classify.cc ClassifySequence/ResolveTree, taxonomy.cc) from its
 * behavioural description in SURVEY.md Appendix A.  It is pinned against (i) an independent
 * closed-form restatement in oracle/k2_literal.py and (ii) the committed fixtures under
 * tests/golden/ produced by that restatement.
 *
 * Nothing in the product path (nohuman_amd/, include/) may link or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker.
 */
#ifndef K2_ORACLE_H
#define K2_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* opts.k2d (kraken2 IndexOptions) */
typedef struct {
    uint64_t k;
    uint64_t l;
    uint64_t spaced_seed_mask;
    uint64_t toggle_mask;
    uint64_t dna_db;
    uint64_t minimum_acceptable_hash_value;
    int32_t revcom_version;
    int32_t db_version;
    int32_t db_type;
} k2o_opts;

typedef struct {
    k2o_opts opts;
    /* hash.k2d */
    uint64_t capacity, size, key_bits, value_bits;
    const uint32_t *cells; /* capacity cells, not owned unless own_cells */
    /* taxo.k2d */
    uint64_t node_count;
    uint32_t *parent;      /* [node_count] internal parent ids (owned) */
    uint64_t *external_id; /* [node_count] (owned) */
    /* behaviour switches (SURVEY.md A.4/A.5 "verify" items) */
    int linear_probing;      /* 1 = -DLINEAR_PROBING build (default), 0 = double hashing */
    int reset_per_mate;      /* 1 = last_minimizer/last_taxon reset per mate (default) */
    uint32_t minimum_hit_groups; /* default 2 */
    int ambiguity_rule;      /* 0 = ambiguous byte in the last l bases; 1 = mmscanner.h is_ambiguous():
                              * queue_pos < k-l || last_ambig, i.e. in the last k-1 bases (default) */
    void *own_cells;         /* malloc'd copy of the cells when loaded from a directory */
} k2o_db;

/* per-fragment record; identical layout to nh_result in include/nohuman_engine.h */
typedef struct {
    uint32_t call;        /* internal taxon id, 0 = unclassified */
    uint32_t total_kmers; /* denominator of the confidence score (ambiguous k-mers included) */
    uint32_t clade_hits;  /* numerator: hits in the clade rooted at `call` (0 if call == 0) */
    uint32_t hit_groups;  /* minimizer_hit_groups */
} k2o_result;

#define K2O_AMBIG_DEFAULT 1
#define K2O_TAXON_AMBIGUOUS 0xFFFFFFFFu /* kraken2 AMBIGUOUS_SPAN_TAXON, "A:n" in the hit list */
#define K2O_TAXON_MATE_BORDER 0xFFFFFFFEu /* kraken2 MATE_PAIR_BORDER_TAXON, "|:|" */

/* Parse the three raw-struct images.  `hash` is borrowed (must outlive the db). */
int k2o_db_from_images(k2o_db *db, const void *opts, size_t opts_len, const void *taxo,
                       size_t taxo_len, const void *hash, size_t hash_len);
/* Load <dir>/{opts,taxo,hash}.k2d (or <dir>/db/...; /root/reference/src/lib.rs:119-141). */
int k2o_db_load_dir(k2o_db *db, const char *dir);
void k2o_db_free(k2o_db *db);
const char *k2o_last_error(void);

uint64_t k2o_fmix64(uint64_t key);
uint64_t k2o_reverse_complement(uint64_t kmer, unsigned n, int revcom_version);
uint32_t k2o_table_get(const k2o_db *db, uint64_t minimizer);

/* Scan one sequence; writes per-k-mer minimizers (value undefined where ambig[i] != 0).
 * Returns the number of k-mers (max(0, len-k+1)), or fewer than cap are written if cap is small. */
size_t k2o_scan_minimizers(const k2o_db *db, const uint8_t *seq, size_t len, uint64_t *minimizers,
                           uint8_t *ambig, size_t cap);

/* Number of entries of the per-k-mer taxa list of one fragment (k-mers of both mates + border). */
uint64_t k2o_taxa_entries(const k2o_db *db, const uint64_t *seq_offsets, uint64_t frag, int paired);

/*
 * Classify n_frag fragments.  bases = concatenated sequence bytes; sequence s occupies
 * bases[seq_offsets[s] .. seq_offsets[s+1]); paired => sequences 2f, 2f+1 are the mates of f.
 * out[n_frag] required.  lookups[n_frag] (optional) receives D = table lookups issued.
 * taxa / taxa_offsets (optional, both or neither): taxa_offsets[n_frag+1] is filled with the
 * prefix of k2o_taxa_entries and taxa[...] with internal ids / the two marker values.
 */
int k2o_classify(const k2o_db *db, const uint8_t *bases, const uint64_t *seq_offsets,
                 uint64_t n_frag, int paired, double confidence, k2o_result *out,
                 uint32_t *lookups, uint32_t *taxa, uint64_t *taxa_offsets, uint64_t taxa_cap);

/* Same, fragments split dynamically over n_threads pthreads (no taxa list). */
int k2o_classify_mt(const k2o_db *db, const uint8_t *bases, const uint64_t *seq_offsets,
                    uint64_t n_frag, int paired, double confidence, k2o_result *out,
                    uint32_t *lookups, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
