/*
 * k2_oracle.c -- CPU restatement of the kraken2 `classify` hot path.  TEST INFRASTRUCTURE ONLY:
 * see k2_oracle.h for the parity statement ("parity unpinned" vs kraken2) and the import rule.
 *
 * Each function names the kraken2 unit it restates (third-party, pinned at
 * /root/reference/Dockerfile:15,35-38) and the SURVEY.md Appendix A paragraph that specifies it;
 * the nohuman call site of the whole path is /root/reference/src/main.rs:215-270 and
 * /root/reference/src/lib.rs:22-48.
 */
#define _GNU_SOURCE
#include "k2_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

static __thread char g_err[512];
const char *k2o_last_error(void) { return g_err; }
#define FAIL(...)                                 \
    do {                                          \
        snprintf(g_err, sizeof g_err, __VA_ARGS__); \
        return -1;                                \
    } while (0)

/* ---- kv_store.h MurmurHash3 finaliser (A.4) ------------------------------------------- */
uint64_t k2o_fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

/* ---- mmscanner.cc reverse_complement / canonical_representation (A.2) ------------------ */
uint64_t k2o_reverse_complement(uint64_t kmer, unsigned n, int revcom_version) {
    kmer = ((kmer & 0xCCCCCCCCCCCCCCCCULL) >> 2) | ((kmer & 0x3333333333333333ULL) << 2);
    kmer = ((kmer & 0xF0F0F0F0F0F0F0F0ULL) >> 4) | ((kmer & 0x0F0F0F0F0F0F0F0FULL) << 4);
    kmer = ((kmer & 0xFF00FF00FF00FF00ULL) >> 8) | ((kmer & 0x00FF00FF00FF00FFULL) << 8);
    kmer = ((kmer & 0xFFFF0000FFFF0000ULL) >> 16) | ((kmer & 0x0000FFFF0000FFFFULL) << 16);
    kmer = (kmer >> 32) | (kmer << 32);
    uint64_t mask = (n >= 32) ? ~0ULL : ((1ULL << (2 * n)) - 1);
    if (revcom_version == 0) /* legacy DBs: complement masked without the shift */
        return (~kmer) & mask;
    return ((~kmer) >> (64 - 2 * n)) & mask;
}

static inline uint64_t canonical(uint64_t kmer, unsigned n, int rv) {
    uint64_t rc = k2o_reverse_complement(kmer, n, rv);
    return kmer < rc ? kmer : rc;
}

/* ---- compact_hash.cc CompactHashTable::Get (A.4) ---------------------------------------- */
static inline uint32_t table_get_hc(const k2o_db *db, uint64_t hc) {
    const uint32_t vbits = (uint32_t)db->value_bits;
    const uint32_t vmask = (uint32_t)((1ULL << vbits) - 1);
    uint64_t compacted_key = hc >> (32 + vbits);
    uint64_t idx = hc % db->capacity;
    uint64_t first_idx = idx;
    uint64_t step = 0;
    for (;;) {
        uint32_t cell = db->cells[idx];
        if ((cell & vmask) == 0) break;
        if ((uint64_t)(cell >> vbits) == compacted_key) return cell & vmask;
        if (step == 0) step = db->linear_probing ? 1 : ((hc >> 8) | 1);
        idx += step;
        idx %= db->capacity;
        if (idx == first_idx) break;
    }
    return 0;
}
uint32_t k2o_table_get(const k2o_db *db, uint64_t minimizer) {
    return table_get_hc(db, k2o_fmix64(minimizer));
}

/* ---- mmscanner.cc MinimizerScanner (A.2/A.3), state-machine form ------------------------ */
typedef struct {
    uint64_t cand;
    int64_t pos;
} mm_entry;

typedef struct {
    const uint8_t *str;
    size_t str_pos, finish;
    int64_t k, l;
    uint64_t lmer, lmer_mask, last_ambig, spaced_mask, toggle;
    int64_t loaded_ch, queue_pos;
    int rv;
    int ambig_rule; /* k2o_db.ambiguity_rule */
    uint64_t last_minimizer;
    /* monotone deque as a ring; never holds more than k-l+2 live entries */
    mm_entry *q;
    int64_t qcap, qhead, qlen;
} mm_scanner;

static uint8_t g_code[256];
static pthread_once_t g_code_once = PTHREAD_ONCE_INIT;
static void init_code(void) {
    memset(g_code, 0xFF, sizeof g_code);
    g_code['A'] = g_code['a'] = 0;
    g_code['C'] = g_code['c'] = 1;
    g_code['G'] = g_code['g'] = 2;
    g_code['T'] = g_code['t'] = 3;
}

static void mm_init(mm_scanner *sc, const k2o_db *db) {
    pthread_once(&g_code_once, init_code);
    memset(sc, 0, sizeof *sc);
    sc->k = (int64_t)db->opts.k;
    sc->l = (int64_t)db->opts.l;
    sc->lmer_mask = (sc->l >= 32) ? ~0ULL : ((1ULL << (2 * sc->l)) - 1);
    sc->spaced_mask = db->opts.spaced_seed_mask;
    sc->toggle = db->opts.toggle_mask & sc->lmer_mask;
    sc->rv = db->opts.revcom_version;
    sc->ambig_rule = db->ambiguity_rule;
    sc->qcap = sc->k - sc->l + 4;
    sc->q = (mm_entry *)malloc((size_t)sc->qcap * sizeof(mm_entry));
}
static void mm_free(mm_scanner *sc) { free(sc->q); }

static void mm_load(mm_scanner *sc, const uint8_t *seq, size_t len) {
    sc->str = seq;
    sc->str_pos = 0;
    sc->finish = len;
    sc->lmer = 0;
    sc->last_ambig = 0;
    sc->loaded_ch = 0;
    sc->queue_pos = 0;
    sc->qhead = 0;
    sc->qlen = 0;
    sc->last_minimizer = ~0ULL;
}

#define QAT(sc, i) ((sc)->q[((sc)->qhead + (i)) % (sc)->qcap])

/* Is the k-mer NextMinimizer just returned ambiguous?  Two recollections of upstream (SURVEY.md A.3 (i)/(ii),
 * VERDICT r3), switchable until a kraken2 binary has been diffed:
 *   rule 0  the flag NextMinimizer hands back through its bool* argument: an ambiguous byte among the last l bases;
 *   rule 1  mmscanner.h is_ambiguous(): (queue_pos_ < k_ - l_) || !!last_ambig_ -- also ambiguous until k - l l-mers
 *           have been queued since the reset, i.e. an ambiguous byte among the last k - 1 bases (isolated N: A:34). */
static inline int mm_is_ambiguous(const mm_scanner *sc) {
    if (sc->ambig_rule == 0) return sc->last_ambig != 0;
    return sc->queue_pos < sc->k - sc->l || sc->last_ambig != 0;
}

/* returns 1 and sets *minimizer, *ambig for the next k-mer; 0 at end of sequence */
static int mm_next(mm_scanner *sc, uint64_t *minimizer, int *ambig) {
    if (sc->str_pos >= sc->finish) return 0;
    int changed = 0;
    while (!changed) {
        if (sc->loaded_ch == sc->l) sc->loaded_ch--;
        while (sc->loaded_ch < sc->l && sc->str_pos < sc->finish) {
            sc->loaded_ch++;
            sc->lmer <<= 2;
            sc->last_ambig <<= 2;
            uint8_t code = g_code[sc->str[sc->str_pos++]];
            if (code == 0xFF) {
                sc->qlen = 0;
                sc->qhead = 0;
                sc->queue_pos = 0;
                sc->lmer = 0;
                sc->loaded_ch = 0;
                sc->last_ambig |= 1;
            } else {
                sc->lmer |= code;
            }
            sc->lmer &= sc->lmer_mask;
            sc->last_ambig &= sc->lmer_mask;
            if ((int64_t)sc->str_pos >= sc->k && sc->loaded_ch < sc->l) {
                *ambig = mm_is_ambiguous(sc);
                *minimizer = sc->last_minimizer;
                return 1;
            }
        }
        if (sc->loaded_ch < sc->l) return 0;
        uint64_t canon = canonical(sc->lmer, (unsigned)sc->l, sc->rv);
        if (sc->spaced_mask) canon &= sc->spaced_mask;
        uint64_t cand = canon ^ sc->toggle;
        if (sc->k == sc->l) {
            sc->last_minimizer = cand ^ sc->toggle;
            *ambig = mm_is_ambiguous(sc); /* (queue_pos stays 0 on this path: 0 < k - l is false) */
            *minimizer = sc->last_minimizer;
            return 1;
        }
        while (sc->qlen > 0 && QAT(sc, sc->qlen - 1).cand > cand) sc->qlen--;
        if (sc->qlen == 0 && sc->queue_pos >= sc->k - sc->l) changed = 1;
        QAT(sc, sc->qlen).cand = cand;
        QAT(sc, sc->qlen).pos = sc->queue_pos;
        sc->qlen++;
        if (QAT(sc, 0).pos < sc->queue_pos - sc->k + sc->l) {
            sc->qhead = (sc->qhead + 1) % sc->qcap;
            sc->qlen--;
            changed = 1;
        }
        if (sc->queue_pos == sc->k - sc->l) changed = 1;
        sc->queue_pos++;
        if ((int64_t)sc->str_pos >= sc->k) break;
    }
    sc->last_minimizer = QAT(sc, 0).cand ^ sc->toggle;
    *ambig = mm_is_ambiguous(sc);
    *minimizer = sc->last_minimizer;
    return 1;
}

size_t k2o_scan_minimizers(const k2o_db *db, const uint8_t *seq, size_t len, uint64_t *minimizers,
                           uint8_t *ambig, size_t cap) {
    mm_scanner sc;
    mm_init(&sc, db);
    mm_load(&sc, seq, len);
    size_t n = 0;
    uint64_t m;
    int a;
    while (mm_next(&sc, &m, &a)) {
        if (n < cap) {
            minimizers[n] = m;
            ambig[n] = (uint8_t)a;
        }
        n++;
    }
    mm_free(&sc);
    return n;
}

/* ---- taxonomy.cc IsAAncestorOfB / LowestCommonAncestor (A.5) ----------------------------- */
static inline int is_a_ancestor_of_b(const k2o_db *db, uint32_t a, uint32_t b) {
    if (!a || !b) return 0;
    while (b > a) b = db->parent[b];
    return b == a;
}
static inline uint32_t lca(const k2o_db *db, uint32_t a, uint32_t b) {
    if (!a || !b) return a ? a : b;
    while (a != b) {
        if (a > b)
            a = db->parent[a];
        else
            b = db->parent[b];
    }
    return a;
}

/* hit_counts: insertion-ordered (taxon,count) list; ResolveTree is order independent */
typedef struct {
    uint32_t *taxon, *count;
    size_t n, cap;
} hitmap;
static void hm_add(hitmap *h, uint32_t t) {
    for (size_t i = 0; i < h->n; i++)
        if (h->taxon[i] == t) {
            h->count[i]++;
            return;
        }
    if (h->n == h->cap) {
        h->cap = h->cap ? 2 * h->cap : 16;
        h->taxon = (uint32_t *)realloc(h->taxon, h->cap * sizeof(uint32_t));
        h->count = (uint32_t *)realloc(h->count, h->cap * sizeof(uint32_t));
    }
    h->taxon[h->n] = t;
    h->count[h->n] = 1;
    h->n++;
}
static uint32_t hm_get(const hitmap *h, uint32_t t) {
    for (size_t i = 0; i < h->n; i++)
        if (h->taxon[i] == t) return h->count[i];
    return 0;
}

/* ---- classify.cc ResolveTree (A.5) ------------------------------------------------------ */
static uint32_t resolve_tree(const k2o_db *db, const hitmap *h, uint32_t total_kmers,
                             double confidence, uint32_t *clade_hits) {
    uint32_t max_taxon = 0, max_score = 0;
    uint32_t required = (uint32_t)ceil(confidence * (double)total_kmers);
    for (size_t i = 0; i < h->n; i++) {
        uint32_t taxon = h->taxon[i], score = 0;
        for (size_t j = 0; j < h->n; j++)
            if (is_a_ancestor_of_b(db, h->taxon[j], taxon)) score += h->count[j];
        if (score > max_score) {
            max_score = score;
            max_taxon = taxon;
        } else if (score == max_score) {
            max_taxon = lca(db, max_taxon, taxon);
        }
    }
    max_score = hm_get(h, max_taxon);
    while (max_taxon && max_score < required) {
        max_score = 0;
        for (size_t j = 0; j < h->n; j++)
            if (is_a_ancestor_of_b(db, max_taxon, h->taxon[j])) max_score += h->count[j];
        if (max_score >= required) break;
        max_taxon = db->parent[max_taxon];
    }
    /* numerator of the per-read confidence score: hits in the clade of the call */
    uint32_t ch = 0;
    if (max_taxon)
        for (size_t j = 0; j < h->n; j++)
            if (is_a_ancestor_of_b(db, max_taxon, h->taxon[j])) ch += h->count[j];
    *clade_hits = ch;
    return max_taxon;
}

uint64_t k2o_taxa_entries(const k2o_db *db, const uint64_t *seq_offsets, uint64_t frag,
                          int paired) {
    uint64_t k = db->opts.k, n = 0;
    int mates = paired ? 2 : 1;
    for (int m = 0; m < mates; m++) {
        uint64_t s = frag * (uint64_t)mates + (uint64_t)m;
        uint64_t len = seq_offsets[s + 1] - seq_offsets[s];
        if (len >= k) n += len - k + 1;
    }
    return n + (paired ? 1 : 0);
}

/* ---- classify.cc ClassifySequence, nucleotide branch (A.5) ------------------------------- */
typedef struct {
    mm_scanner sc;
    hitmap h;
} worker_state;

static void classify_one(const k2o_db *db, worker_state *ws, const uint8_t *bases,
                         const uint64_t *seq_offsets, uint64_t frag, int paired, double confidence,
                         k2o_result *out, uint32_t *lookups, uint32_t *taxa) {
    int mates = paired ? 2 : 1;
    ws->h.n = 0;
    uint32_t hit_groups = 0, n_lookups = 0;
    uint64_t n_taxa = 0;
    uint64_t last_minimizer = ~0ULL;
    uint32_t last_taxon = 0xFFFFFFFFu;
    for (int m = 0; m < mates; m++) {
        uint64_t s = frag * (uint64_t)mates + (uint64_t)m;
        mm_load(&ws->sc, bases + seq_offsets[s], (size_t)(seq_offsets[s + 1] - seq_offsets[s]));
        if (db->reset_per_mate) {
            last_minimizer = ~0ULL;
            last_taxon = 0xFFFFFFFFu;
        }
        uint64_t minimizer;
        int ambig;
        while (mm_next(&ws->sc, &minimizer, &ambig)) {
            uint32_t taxon;
            if (ambig) {
                taxon = K2O_TAXON_AMBIGUOUS;
            } else {
                if (minimizer != last_minimizer) {
                    int skip = 0;
                    uint64_t hc = k2o_fmix64(minimizer);
                    if (db->opts.minimum_acceptable_hash_value &&
                        hc < db->opts.minimum_acceptable_hash_value)
                        skip = 1;
                    taxon = 0;
                    if (!skip) {
                        taxon = table_get_hc(db, hc);
                        n_lookups++;
                    }
                    last_taxon = taxon;
                    last_minimizer = minimizer;
                    if (taxon) hit_groups++;
                } else {
                    taxon = last_taxon;
                }
                if (taxon) hm_add(&ws->h, taxon);
            }
            if (taxa) taxa[n_taxa] = taxon;
            n_taxa++;
        }
        if (paired && m == 0) {
            if (taxa) taxa[n_taxa] = K2O_TAXON_MATE_BORDER;
            n_taxa++;
        }
    }
    uint32_t total_kmers = (uint32_t)(n_taxa - (paired ? 1 : 0));
    uint32_t clade_hits = 0;
    uint32_t call = resolve_tree(db, &ws->h, total_kmers, confidence, &clade_hits);
    if (call && hit_groups < db->minimum_hit_groups) {
        call = 0;
        clade_hits = 0;
    }
    out->call = call;
    out->total_kmers = total_kmers;
    out->clade_hits = clade_hits;
    out->hit_groups = hit_groups;
    if (lookups) *lookups = n_lookups;
}

int k2o_classify(const k2o_db *db, const uint8_t *bases, const uint64_t *seq_offsets,
                 uint64_t n_frag, int paired, double confidence, k2o_result *out,
                 uint32_t *lookups, uint32_t *taxa, uint64_t *taxa_offsets, uint64_t taxa_cap) {
    if (!db->opts.dna_db) FAIL("protein databases are out of scope");
    if ((taxa != NULL) != (taxa_offsets != NULL)) FAIL("taxa and taxa_offsets go together");
    worker_state ws;
    memset(&ws, 0, sizeof ws);
    mm_init(&ws.sc, db);
    uint64_t toff = 0;
    for (uint64_t f = 0; f < n_frag; f++) {
        uint32_t *tp = NULL;
        if (taxa) {
            uint64_t ne = k2o_taxa_entries(db, seq_offsets, f, paired);
            taxa_offsets[f] = toff;
            if (toff + ne > taxa_cap) {
                mm_free(&ws.sc);
                free(ws.h.taxon);
                free(ws.h.count);
                FAIL("taxa buffer too small");
            }
            tp = taxa + toff;
            toff += ne;
        }
        classify_one(db, &ws, bases, seq_offsets, f, paired, confidence, &out[f],
                     lookups ? &lookups[f] : NULL, tp);
    }
    if (taxa) taxa_offsets[n_frag] = toff;
    mm_free(&ws.sc);
    free(ws.h.taxon);
    free(ws.h.count);
    return 0;
}

/* ---- classify.cc's OpenMP block loop (A.7), restated with pthreads ------------------------ */
typedef struct {
    const k2o_db *db;
    const uint8_t *bases;
    const uint64_t *seq_offsets;
    uint64_t n_frag;
    int paired;
    double confidence;
    k2o_result *out;
    uint32_t *lookups;
    uint64_t *next;
} mt_job;

#define MT_BLOCK 256

static void *mt_worker(void *arg) {
    mt_job *j = (mt_job *)arg;
    worker_state ws;
    memset(&ws, 0, sizeof ws);
    mm_init(&ws.sc, j->db);
    for (;;) {
        uint64_t b = __atomic_fetch_add(j->next, MT_BLOCK, __ATOMIC_RELAXED);
        if (b >= j->n_frag) break;
        uint64_t e = b + MT_BLOCK < j->n_frag ? b + MT_BLOCK : j->n_frag;
        for (uint64_t f = b; f < e; f++)
            classify_one(j->db, &ws, j->bases, j->seq_offsets, f, j->paired, j->confidence,
                         &j->out[f], j->lookups ? &j->lookups[f] : NULL, NULL);
    }
    mm_free(&ws.sc);
    free(ws.h.taxon);
    free(ws.h.count);
    return NULL;
}

int k2o_classify_mt(const k2o_db *db, const uint8_t *bases, const uint64_t *seq_offsets,
                    uint64_t n_frag, int paired, double confidence, k2o_result *out,
                    uint32_t *lookups, int n_threads) {
    if (!db->opts.dna_db) FAIL("protein databases are out of scope");
    if (n_threads < 1) n_threads = 1;
    uint64_t next = 0;
    mt_job job = {db, bases, seq_offsets, n_frag, paired, confidence, out, lookups, &next};
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, mt_worker, &job);
    for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    free(th);
    return 0;
}

/* ---- DB images (A.1) --------------------------------------------------------------------- */
static uint64_t rd64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}

int k2o_db_from_images(k2o_db *db, const void *opts, size_t opts_len, const void *taxo,
                       size_t taxo_len, const void *hash, size_t hash_len) {
    memset(db, 0, sizeof *db);
    db->linear_probing = 1;
    db->reset_per_mate = 1;
    db->minimum_hit_groups = 2;
    db->ambiguity_rule = K2O_AMBIG_DEFAULT;
    { /* tests run whole suites under the other rule in a child process (the engine reads NOHUMAN_OPT_AMBIGUITY_RULE) */
        const char *env = getenv("K2O_AMBIGUITY_RULE");
        if (env && *env) db->ambiguity_rule = atoi(env) != 0;
    }
    /* opts.k2d: read min(filesize, sizeof IndexOptions) bytes into a zeroed struct */
    uint8_t ob[64];
    memset(ob, 0, sizeof ob);
    memcpy(ob, opts, opts_len < 64 ? opts_len : 64);
    db->opts.k = rd64(ob + 0);
    db->opts.l = rd64(ob + 8);
    db->opts.spaced_seed_mask = rd64(ob + 16);
    db->opts.toggle_mask = rd64(ob + 24);
    db->opts.dna_db = ob[32];
    db->opts.minimum_acceptable_hash_value = rd64(ob + 40);
    memcpy(&db->opts.revcom_version, ob + 48, 4);
    memcpy(&db->opts.db_version, ob + 52, 4);
    memcpy(&db->opts.db_type, ob + 56, 4);
    if (db->opts.l == 0 || db->opts.l > 31 || db->opts.l > db->opts.k)
        FAIL("opts.k2d: bad k=%llu l=%llu", (unsigned long long)db->opts.k,
             (unsigned long long)db->opts.l);
    /* hash.k2d */
    if (hash_len < 32) FAIL("hash.k2d: truncated header");
    const uint8_t *hb = (const uint8_t *)hash;
    db->capacity = rd64(hb);
    db->size = rd64(hb + 8);
    db->key_bits = rd64(hb + 16);
    db->value_bits = rd64(hb + 24);
    if (db->key_bits + db->value_bits != 32) FAIL("hash.k2d: key_bits + value_bits != 32");
    if (db->capacity == 0 || hash_len != 32 + 4 * db->capacity)
        FAIL("hash.k2d: size %zu != 32 + 4*capacity(%llu)", hash_len,
             (unsigned long long)db->capacity);
    db->cells = (const uint32_t *)(hb + 32);
    /* taxo.k2d */
    const uint8_t *tb = (const uint8_t *)taxo;
    if (taxo_len < 32 || memcmp(tb, "K2TAXDAT", 8) != 0) FAIL("taxo.k2d: bad magic");
    db->node_count = rd64(tb + 8);
    uint64_t name_len = rd64(tb + 16), rank_len = rd64(tb + 24);
    if (taxo_len != 32 + 56 * db->node_count + name_len + rank_len)
        FAIL("taxo.k2d: size mismatch");
    db->parent = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(db->node_count + 1));
    db->external_id = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(db->node_count + 1));
    for (uint64_t i = 0; i < db->node_count; i++) {
        const uint8_t *n = tb + 32 + 56 * i;
        db->parent[i] = (uint32_t)rd64(n + 0);
        db->external_id[i] = rd64(n + 40);
        if (i >= 2 && db->parent[i] >= i) FAIL("taxo.k2d: parent id not below child id");
    }
    return 0;
}

static void *slurp(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    struct stat st;
    if (fstat(fileno(f), &st) != 0) {
        fclose(f);
        return NULL;
    }
    void *buf = malloc((size_t)st.st_size ? (size_t)st.st_size : 1);
    if (fread(buf, 1, (size_t)st.st_size, f) != (size_t)st.st_size) {
        free(buf);
        fclose(f);
        return NULL;
    }
    fclose(f);
    *len = (size_t)st.st_size;
    return buf;
}

int k2o_db_load_dir(k2o_db *db, const char *dir) {
    char p[3][4096];
    const char *names[3] = {"opts.k2d", "taxo.k2d", "hash.k2d"};
    void *buf[3] = {0, 0, 0};
    size_t len[3];
    for (int attempt = 0; attempt < 2; attempt++) {
        int ok = 1;
        for (int i = 0; i < 3; i++) {
            snprintf(p[i], sizeof p[i], attempt ? "%s/db/%s" : "%s/%s", dir, names[i]);
            struct stat st;
            if (stat(p[i], &st) != 0) ok = 0;
        }
        if (!ok) continue;
        for (int i = 0; i < 3; i++) {
            buf[i] = slurp(p[i], &len[i]);
            if (!buf[i]) FAIL("cannot read %.400s", p[i]);
        }
        int rc = k2o_db_from_images(db, buf[0], len[0], buf[1], len[1], buf[2], len[2]);
        free(buf[0]);
        free(buf[1]);
        if (rc != 0) {
            free(buf[2]);
            return rc;
        }
        db->own_cells = buf[2];
        return 0;
    }
    FAIL("Required files (hash.k2d, opts.k2d, taxo.k2d) not found in %.300s or its 'db' subdirectory",
         dir);
}

void k2o_db_free(k2o_db *db) {
    free(db->parent);
    free(db->external_id);
    free(db->own_cells);
    memset(db, 0, sizeof *db);
}
