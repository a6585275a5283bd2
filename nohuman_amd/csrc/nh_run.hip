// nh_run.hip -- whole-run entry nh_run(): files in, kraken2-compatible files out.
//
// Replaces, for the caller at /root/reference/src/main.rs:270, what the kraken2 subprocess does
// with the argv of src/main.rs:210-267: read 1-2 FASTQ/FASTA inputs (plain/gzip/bzip2), classify
// every fragment, write the kept fragments UNCOMPRESSED to kraken_out.fq or kraken_out_1.fq +
// kraken_out_2.fq (src/main.rs:252-256,308-309,333), optionally the per-read kraken output
// (--output) and return the counts nohuman scrapes from stderr (src/lib.rs:61-97).
// kraken2 units restated: classify.cc ProcessFiles / output formatting.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include "nh_fastx.h"
#include "nh_internal.h"
#include "nohuman_engine.h"

namespace nh {

struct OutFile {
    FILE *f = nullptr;
    std::vector<char> iobuf;
    int open(const char *path) {
        f = fopen(path, "wb");
        if (!f) return set_error(NH_EIO, "cannot create %s", path);
        iobuf.resize(4u << 20);
        setvbuf(f, iobuf.data(), _IOFBF, iobuf.size());
        return NH_OK;
    }
    int close() {
        int rc = NH_OK;
        if (f && fclose(f) != 0) rc = set_error(NH_EIO, "write error on output file");
        f = nullptr;
        return rc;
    }
    ~OutFile() {
        if (f) fclose(f);
    }
};

static void append_record(std::string &dst, const SeqRecord &r, const char *suffix) {
    dst += r.header;
    if (suffix) dst += suffix;
    dst += '\n';
    dst += r.seq;
    if (r.format == FMT_FASTQ) {
        dst += "\n+\n";
        dst += r.quals;
    }
    dst += '\n';
}

// kraken2 TrimPairInfo: drop a trailing /1 or /2 from ids longer than two characters
static void trim_pair_info(std::string &id) {
    size_t sz = id.size();
    if (sz <= 2) return;
    if (id[sz - 2] == '/' && (id[sz - 1] == '1' || id[sz - 1] == '2')) id.erase(sz - 2);
}

// kraken2 AddHitlistString
static void append_hitlist(std::string &dst, const Engine *e, const uint32_t *taxa, uint64_t n) {
    if (n == 0) {
        dst += "0:0";
        return;
    }
    char tmp[64];
    uint64_t i = 0;
    bool first = true;
    while (i < n) {
        uint32_t t = taxa[i];
        uint64_t j = i;
        while (j < n && taxa[j] == t) j++;
        if (t == TAXON_MATE_BORDER) {
            for (uint64_t r = i; r < j; r++) {
                if (!first) dst += ' ';
                dst += "|:|";
                first = false;
            }
        } else {
            if (!first) dst += ' ';
            if (t == TAXON_AMBIGUOUS)
                snprintf(tmp, sizeof tmp, "A:%llu", (unsigned long long)(j - i));
            else
                snprintf(tmp, sizeof tmp, "%llu:%llu", (unsigned long long)e->external[t],
                         (unsigned long long)(j - i));
            dst += tmp;
            first = false;
        }
        i = j;
    }
}

// kraken2 reports.cc ReportKrakenStyle / KrakenReportDFS (reached through
// nohuman's -r/--kraken-report, /root/reference/src/main.rs:97-99,226-228).  Lowest-confidence part
// of the restatement: the line format is "%6.2f\tclade\ttaxon\trank\ttaxid\t<2*depth spaces>name".
struct TaxoView {
    uint64_t n = 0;
    const uint8_t *nodes = nullptr;
    const char *names = nullptr, *ranks = nullptr;
    uint64_t field(uint64_t i, int k) const {
        uint64_t v;
        memcpy(&v, nodes + 56 * i + 8 * k, 8);
        return v;
    }
};

static void report_dfs(FILE *f, const TaxoView &t, const std::vector<uint64_t> &clade,
                       const std::vector<uint64_t> &own, uint64_t total, uint64_t id, char rank_code,
                       int rank_depth, int depth) {
    if (clade[id] == 0) return;  // clades absent from the sample are not printed
    const std::string rank = t.ranks + t.field(id, 4);
    if (rank == "superkingdom" || rank == "domain") { rank_code = 'D'; rank_depth = 0; }
    else if (rank == "kingdom") { rank_code = 'K'; rank_depth = 0; }
    else if (rank == "phylum") { rank_code = 'P'; rank_depth = 0; }
    else if (rank == "class") { rank_code = 'C'; rank_depth = 0; }
    else if (rank == "order") { rank_code = 'O'; rank_depth = 0; }
    else if (rank == "family") { rank_code = 'F'; rank_depth = 0; }
    else if (rank == "genus") { rank_code = 'G'; rank_depth = 0; }
    else if (rank == "species") { rank_code = 'S'; rank_depth = 0; }
    else rank_depth++;
    std::string rank_str(1, rank_code);
    if (rank_depth != 0) rank_str += std::to_string(rank_depth);
    fprintf(f, "%6.2f\t%llu\t%llu\t%s\t%llu\t", 100.0 * (double)clade[id] / (double)total,
            (unsigned long long)clade[id], (unsigned long long)own[id], rank_str.c_str(),
            (unsigned long long)t.field(id, 5));
    for (int i = 0; i < depth; i++) fputs("  ", f);
    fprintf(f, "%s\n", t.names + t.field(id, 3));
    const uint64_t first = t.field(id, 1), cnt = t.field(id, 2);
    std::vector<uint64_t> kids(cnt);
    for (uint64_t i = 0; i < cnt; i++) kids[i] = first + i;
    std::stable_sort(kids.begin(), kids.end(),
                     [&](uint64_t x, uint64_t y) { return clade[x] > clade[y]; });
    for (uint64_t c : kids) report_dfs(f, t, clade, own, total, c, rank_code, rank_depth, depth + 1);
}

static int write_report(const Engine *e, const char *path, const std::vector<uint64_t> &own,
                        uint64_t total, uint64_t unclassified) {
    TaxoView t;
    const uint8_t *img = e->taxo_image.data();
    memcpy(&t.n, img + 8, 8);
    uint64_t name_len;
    memcpy(&name_len, img + 16, 8);
    t.nodes = img + 32;
    t.names = (const char *)(img + 32 + 56 * t.n);
    t.ranks = t.names + name_len;
    std::vector<uint64_t> clade(own);
    for (uint64_t i = t.n; i-- > 1;)  // children have larger ids than their parents
        if (clade[i]) clade[t.field(i, 0)] += clade[i];
    FILE *f = fopen(path, "w");
    if (!f) return set_error(NH_EIO, "cannot create %s", path);
    if (unclassified != 0)
        fprintf(f, "%6.2f\t%llu\t%llu\tU\t0\tunclassified\n",
                100.0 * (double)unclassified / (double)total, (unsigned long long)unclassified,
                (unsigned long long)unclassified);
    if (t.n > 1 && total) report_dfs(f, t, clade, own, total, 1, 'R', -1, 0);
    if (fclose(f) != 0) return set_error(NH_EIO, "write error on %s", path);
    return NH_OK;
}

int run_engine(Engine *e, const nh_run_args *a, nh_stats *stats) {
    if (!a || !a->in1 || !a->out1) return set_error(NH_EINVAL, "nh_run: in1 and out1 are required");
    const bool paired = a->in2 != nullptr;
    if (paired && !a->out2) return set_error(NH_EINVAL, "nh_run: paired input needs out2");
    if (!(a->confidence >= 0.0 && a->confidence <= 1.0))
        return set_error(NH_EINVAL, "Confidence score must be in the closed interval [0, 1]");
    const bool want_k =
        a->kraken_output && a->kraken_output[0] && strcmp(a->kraken_output, "/dev/null") != 0;

    std::string err;
    FastxReader r1, r2;
    if (r1.open(a->in1, err) != 0) return set_error(NH_EIO, "%s", err.c_str());
    if (paired && r2.open(a->in2, err) != 0) return set_error(NH_EIO, "%s", err.c_str());
    OutFile o1, o2, ok;
    int rc;
    if ((rc = o1.open(a->out1))) return rc;
    if (paired && (rc = o2.open(a->out2))) return rc;
    if (want_k && (rc = ok.open(a->kraken_output))) return rc;

    const size_t BATCH_FRAGS = 1u << 18;
    const size_t BATCH_BYTES = 256u << 20;
    std::vector<SeqRecord> recs1(BATCH_FRAGS), recs2(paired ? BATCH_FRAGS : 0);
    std::vector<uint8_t> bases;
    std::vector<uint64_t> offsets;
    std::vector<nh_result> results;
    std::vector<uint32_t> taxa;
    std::vector<uint64_t> taxa_off;
    std::string buf1, buf2, bufk;
    uint64_t total = 0, classified = 0, total_bases = 0;
    std::vector<uint64_t> call_counts(e->external.size(), 0);  // fragments called at each taxon
    const uint32_t flags = paired ? NH_FLAG_PAIRED : 0;
    auto t0 = std::chrono::steady_clock::now();
    bool done = false;
    while (!done) {
        size_t n = 0;
        bases.clear();
        offsets.clear();
        offsets.push_back(0);
        while (n < BATCH_FRAGS && bases.size() < BATCH_BYTES) {
            int g1 = r1.next(recs1[n], err);
            if (g1 < 0) return set_error(NH_EIO, "%s", err.c_str());
            if (g1 == 0) {
                done = true;
                break;
            }
            if (paired) {
                int g2 = r2.next(recs2[n], err);
                if (g2 < 0) return set_error(NH_EIO, "%s", err.c_str());
                if (g2 == 0) {
                    done = true;
                    break;
                }
            }
            bases.insert(bases.end(), recs1[n].seq.begin(), recs1[n].seq.end());
            offsets.push_back(bases.size());
            if (paired) {
                bases.insert(bases.end(), recs2[n].seq.begin(), recs2[n].seq.end());
                offsets.push_back(bases.size());
            }
            n++;
        }
        if (n == 0) break;
        results.resize(n);
        uint64_t ntaxa = 0;
        if (want_k) {
            ntaxa = kmer_taxa_entries(e, offsets.data(), n, paired ? 2 : 1, nullptr);
            taxa.resize(ntaxa + 1);
            taxa_off.resize(n + 1);
        }
        rc = classify_host(e, bases.data(), offsets.data(), n, flags, a->confidence, results.data(),
                           want_k ? taxa.data() : nullptr, want_k ? taxa_off.data() : nullptr,
                           ntaxa + 1);
        if (rc) return rc;
        buf1.clear();
        buf2.clear();
        bufk.clear();
        char tmp[128];
        for (size_t i = 0; i < n; i++) {
            const uint32_t call = results[i].call;
            const bool is_class = call != 0;
            total++;
            classified += is_class;
            call_counts[call] += is_class;
            total_bases += recs1[i].seq.size() + (paired ? recs2[i].seq.size() : 0);
            const uint64_t ext = is_class ? e->external[call] : 0;
            if (is_class == (a->keep_human != 0)) {
                const char *suffix = nullptr;
                if (is_class) {
                    snprintf(tmp, sizeof tmp, " kraken:taxid|%llu", (unsigned long long)ext);
                    suffix = tmp;
                }
                append_record(buf1, recs1[i], suffix);
                if (paired) append_record(buf2, recs2[i], suffix);
            }
            if (want_k) {
                std::string id = recs1[i].id;
                if (paired) trim_pair_info(id);
                bufk += is_class ? "C\t" : "U\t";
                bufk += id;
                snprintf(tmp, sizeof tmp, "\t%llu\t", (unsigned long long)ext);
                bufk += tmp;
                if (paired)
                    snprintf(tmp, sizeof tmp, "%zu|%zu\t", recs1[i].seq.size(), recs2[i].seq.size());
                else
                    snprintf(tmp, sizeof tmp, "%zu\t", recs1[i].seq.size());
                bufk += tmp;
                append_hitlist(bufk, e, taxa.data() + taxa_off[i], taxa_off[i + 1] - taxa_off[i]);
                bufk += '\n';
            }
        }
        if (!buf1.empty() && fwrite(buf1.data(), 1, buf1.size(), o1.f) != buf1.size())
            return set_error(NH_EIO, "write error on %s", a->out1);
        if (paired && !buf2.empty() && fwrite(buf2.data(), 1, buf2.size(), o2.f) != buf2.size())
            return set_error(NH_EIO, "write error on %s", a->out2);
        if (want_k && !bufk.empty() && fwrite(bufk.data(), 1, bufk.size(), ok.f) != bufk.size())
            return set_error(NH_EIO, "write error on %s", a->kraken_output);
    }
    if (a->report && a->report[0] &&
        (rc = write_report(e, a->report, call_counts, total, total - classified)))
        return rc;
    if ((rc = o1.close())) return rc;
    if (paired && (rc = o2.close())) return rc;
    if (want_k && (rc = ok.close())) return rc;
    if (stats) {
        nh_stats st;
        memset(&st, 0, sizeof st);
        st.total_sequences = total;
        st.classified = classified;
        st.unclassified = total - classified;
        st.total_bases = total_bases;
        st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        *stats = st;
    }
    return NH_OK;
}

}  // namespace nh

extern "C" {

int nh_run_engine(nh_engine *e, const nh_run_args *args, nh_stats *stats) {
    if (!e) return nh::set_error(NH_EINVAL, "null engine");
    return nh::run_engine((nh::Engine *)e, args, stats);
}

int nh_run(const nh_run_args *args, nh_stats *stats) {
    if (!args || !args->db_dir) return nh::set_error(NH_EINVAL, "nh_run: db_dir is required");
    int device = (args->device_ids && args->n_devices > 0) ? args->device_ids[0] : 0;
    nh::Engine *e = nullptr;
    int rc = nh::open_dir(args->db_dir, device, &e);
    if (rc) return rc;
    rc = nh::run_engine(e, args, stats);
    nh::destroy(e);
    return rc;
}

}  // extern "C"
