// deflate_model.cpp -- host model of the GPU gzip encoder (nohuman_amd/csrc/nh_deflate.hip): the SAME lane-level
// and sequential functions (nh_deflate_core.h), with the wave's 64 lanes run by a loop.  Used to check the format
// logic against zlib's inflate on a CPU (tests/test_deflate_core.py) and to tune the match heuristics for ratio
// without a GPU.  Not part of the product: nothing in nohuman_amd/ calls it.
//   g++ -O2 -std=c++17 -I nohuman_amd/csrc tools/deflate_model.cpp -o /tmp/deflate_model -lz
//   deflate_model <in> <out.gz> [region_bytes] [block_bytes]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "nh_deflate_core.h"

using namespace nh::dfl;

struct BitWriter {
    std::vector<uint8_t> out;
    uint64_t acc = 0;
    uint32_t nacc = 0;
    void put(uint32_t v, uint32_t n) {
        acc |= (uint64_t)v << nacc;
        nacc += n;
        while (nacc >= 8) {
            out.push_back((uint8_t)acc);
            acc >>= 8;
            nacc -= 8;
        }
    }
    void align() {
        if (nacc) put(0, 8 - nacc);
    }
};

struct Tree {
    uint8_t lens[NLIT];
    uint16_t codes[NLIT];
};

static void build_tree(const uint32_t *freq_in, int nsym, int maxbits, Tree &t) {
    uint32_t freq[NLIT];
    for (int s = 0; s < nsym; s++) freq[s] = freq_in[s];
    int used = 0;
    for (int s = 0; s < nsym; s++) used += freq[s] != 0;
    for (int s = 0; used < 2 && s < nsym; s++)  // a code needs two symbols to be complete
        if (!freq[s]) {
            freq[s] = 1;
            used++;
        }
    std::vector<uint32_t> keys;
    for (int s = 0; s < nsym; s++)
        if (freq[s]) keys.push_back((freq[s] << 9) | (uint32_t)s);
    std::sort(keys.begin(), keys.end());
    uint32_t a[NLIT];
    uint16_t ss[NLIT];
    for (size_t i = 0; i < keys.size(); i++) {
        a[i] = keys[i] >> 9;
        ss[i] = (uint16_t)(keys[i] & 511u);
    }
    huff_lengths_sorted(a, ss, (int)keys.size(), nsym, maxbits, t.lens);
    huff_codes(t.lens, nsym, maxbits, t.codes);
}

struct Block {
    std::vector<uint32_t> tok;
    uint32_t lfreq[NLIT], dfreq[NDIST];
    uint32_t in_from = 0;
    void reset(uint32_t from) {
        tok.clear();
        memset(lfreq, 0, sizeof lfreq);
        memset(dfreq, 0, sizeof dfreq);
        in_from = from;
    }
};

static uint64_t g_cat_bits[4], g_cat_lit[4], g_cat_match[4], g_cat_mbytes[4];
static uint32_t g_reclen = 0, g_hdr = 0, g_seq = 0;  // DFL_REC="reclen,hdr,seq": bits by line of a fixed-length FASTQ record
static int cat_of(uint64_t pos) {
    const uint32_t o = (uint32_t)(pos % g_reclen);
    return o < g_hdr ? 0 : o < g_hdr + g_seq + 1 ? 1 : o < g_hdr + g_seq + 3 ? 2 : 3;
}
static uint64_t g_region_base = 0;
static uint64_t g_dyn = 0, g_stored = 0, g_hdr_bits = 0, g_tok = 0, g_match = 0, g_match_bytes = 0;

static void emit_block(BitWriter &bw, Block &b, const uint8_t *src, uint32_t in_to, Tree &lt, Tree &dt) {
    b.lfreq[256] = 1;
    build_tree(b.lfreq, NLIT_USED, MAXBITS, lt);
    build_tree(b.dfreq, NDIST_USED, MAXBITS, dt);
    int hlit = NLIT_USED, hdist = NDIST_USED;
    while (hlit > 257 && lt.lens[hlit - 1] == 0) hlit--;
    while (hdist > 1 && dt.lens[hdist - 1] == 0) hdist--;
    uint8_t all[NLIT + NDIST];
    for (int i = 0; i < hlit; i++) all[i] = lt.lens[i];
    for (int i = 0; i < hdist; i++) all[hlit + i] = dt.lens[i];
    uint16_t items[NLIT + NDIST];
    uint32_t clfreq[NCL];
    const int ni = rle_lengths(all, hlit + hdist, items, clfreq);
    Tree ct;
    build_tree(clfreq, NCL, MAXBITS_CL, ct);
    int hclen = NCL;
    while (hclen > 4 && ct.lens[cl_order(hclen - 1)] == 0) hclen--;
    uint64_t bits = 3 + 5 + 5 + 4 + 3 * (uint64_t)hclen;
    for (int i = 0; i < ni; i++) bits += ct.lens[items[i] & 31u] + cl_extra_bits(items[i] & 31u);
    const uint64_t hdr = bits;
    for (int s = 0; s < NLIT_USED; s++) bits += (uint64_t)b.lfreq[s] * (lt.lens[s] + (s > 256 ? len_extra_bits(s) : 0));
    for (int s = 0; s < NDIST_USED; s++) bits += (uint64_t)b.dfreq[s] * (dt.lens[s] + dist_extra_bits(s));
    const uint32_t nbytes = in_to - b.in_from;
    if (bits >= 8ull * nbytes + 40) {  // stored
        g_stored++;
        bw.put(0, 3);
        bw.align();
        bw.put(nbytes & 0xFFFF, 16);
        bw.put(~nbytes & 0xFFFF, 16);
        for (uint32_t i = 0; i < nbytes; i++) bw.put(src[b.in_from + i], 8);
        return;
    }
    g_dyn++;
    g_hdr_bits += hdr;
    bw.put(0, 1);
    bw.put(2, 2);
    bw.put(hlit - 257, 5);
    bw.put(hdist - 1, 5);
    bw.put(hclen - 4, 4);
    for (int i = 0; i < hclen; i++) bw.put(ct.lens[cl_order(i)], 3);
    for (int i = 0; i < ni; i++) {
        const uint32_t s = items[i] & 31u;
        bw.put(ct.codes[s], ct.lens[s]);
        if (cl_extra_bits(s)) bw.put(items[i] >> 5, cl_extra_bits(s));
    }
    uint64_t pos = g_region_base + b.in_from;
    for (uint32_t t : b.tok) {
        if (g_reclen) {
            const int c = cat_of(pos);
            if (t & 0x80000000u) {
                uint32_t sym, eb, ev, dsym, deb, dev;
                len_symbol((t >> 16) & 0xFFu, sym, eb, ev);
                dist_symbol(t & 0x7FFFu, dsym, deb, dev);
                g_cat_bits[c] += lt.lens[sym] + eb + dt.lens[dsym] + deb;
                g_cat_match[c]++;
                g_cat_mbytes[c] += ((t >> 16) & 0xFFu) + 3;
                pos += ((t >> 16) & 0xFFu) + 3;
            } else {
                g_cat_bits[c] += lt.lens[t];
                g_cat_lit[c]++;
                pos++;
            }
        }
        if (t & 0x80000000u) {
            uint32_t sym, eb, ev, dsym, deb, dev;
            len_symbol((t >> 16) & 0xFFu, sym, eb, ev);
            dist_symbol(t & 0x7FFFu, dsym, deb, dev);
            bw.put(lt.codes[sym], lt.lens[sym]);
            if (eb) bw.put(ev, eb);
            bw.put(dt.codes[dsym], dt.lens[dsym]);
            if (deb) bw.put(dev, deb);
        } else {
            bw.put(lt.codes[t], lt.lens[t]);
        }
    }
    bw.put(lt.codes[256], lt.lens[256]);
}

// one region, as one wave would do it.  Knobs (environment, tuning only): DFL_WAYS = 4 | 8 | 16 bucket width, DFL_BB =
// bucket bits, DFL_INIT = the first prices (0: literals 6 bits like zlib's rules assume, 1: A C G T N 2 bits -- what the
// encoder uses), DFL_KEEP = 1: the prices carry over from region to region (as from chunk to chunk on the GPU).
static void deflate_region(BitWriter &bw, const uint8_t *src, uint32_t n, uint32_t block_bytes) {
    static const uint32_t WAYS = getenv("DFL_WAYS") ? atoi(getenv("DFL_WAYS")) : 8;
    static const uint32_t BB = getenv("DFL_BB") ? atoi(getenv("DFL_BB")) : BUCKET_BITS;
    static const int keep_prices = getenv("DFL_KEEP") ? atoi(getenv("DFL_KEEP")) : 0;
    static const int init = getenv("DFL_INIT") ? atoi(getenv("DFL_INIT")) : 1;
    std::vector<uint16_t> bucket((size_t)WAYS << BB, (uint16_t)EMPTY_ENTRY);
    static Tree lt, dt;
    static bool have_prices = false;
    if (!(keep_prices && have_prices)) {
        for (int s = 0; s < NLIT; s++) lt.lens[s] = s < 256 ? 6 : 7;
        for (int s = 0; s < NDIST; s++) dt.lens[s] = 5;
        if (init == 1)
            for (const char *c = "ACGTN"; *c; c++) lt.lens[(int)*c] = 2;
        have_prices = true;
    }
    Block b;
    b.reset(0);
    uint32_t carry = 0;
    for (uint32_t s = 0; s < n; s += 64) {
        uint32_t L[64], D[64], H[64];
        const Costs costs{lt.lens, dt.lens};
        const bool any = carry < s + 64;
        for (uint32_t lane = 0; lane < 64; lane++) {  // every lane: hash, bucket, match
            const uint32_t p = s + lane;
            L[lane] = 0;
            D[lane] = 0;
            H[lane] = 0;
            if (p >= n) continue;
            if (p + HASH_BYTES <= n) H[lane] = hash_at(load8(src + p));
            if (!any || p < carry) continue;
            int gain = 0;
            const uint16_t *e = &bucket[WAYS * (H[lane] >> (32 - BB))];
            const Bytes16 c16 = load16(src + p);
            L[lane] = WAYS == 4    ? find_match<4>(src, p, n, c16, e, costs, D[lane], gain)
                      : WAYS == 16 ? find_match<16>(src, p, n, c16, e, costs, D[lane], gain)
                                   : find_match<8>(src, p, n, c16, e, costs, D[lane], gain);
        }
        for (uint32_t lane = 0; lane < 64; lane++) {  // the step's positions enter the buckets after the look-ups
            const uint32_t p = s + lane;
            if (p + HASH_BYTES <= n) bucket[WAYS * (H[lane] >> (32 - BB)) + ((p >> 6) % WAYS)] = make_entry(p);
        }
        if (any) {
            for (uint32_t lane = 0; lane < 64; lane++) {  // lazy rule: a longer match one position on wins
                const uint32_t nx = lane < 63 ? L[lane + 1] : 0;
                if (L[lane] && L[lane] < 16 && nx > L[lane]) L[lane] = 0;  // (the kernel: all lanes at once, L[lane + 1] as found)
            }
            uint32_t q = carry > s ? carry - s : 0;
            while (q < 64 && s + q < n) {
                const uint32_t p = s + q;
                uint32_t l = L[q];
                if (l == SCAN_CAP) {
                    const uint32_t room = n - p, cap = room < MAX_MATCH ? room : MAX_MATCH;
                    l = common_prefix(src + p - D[q], src + p, SCAN_CAP, cap);
                }
                if (l) {
                    uint32_t sym, eb, ev, dsym, deb, dev;
                    len_symbol(l - 3, sym, eb, ev);
                    dist_symbol(D[q] - 1, dsym, deb, dev);
                    b.tok.push_back(tok_match(l, D[q]));
                    b.lfreq[sym]++;
                    b.dfreq[dsym]++;
                    g_match++;
                    g_match_bytes += l;
                } else {
                    b.tok.push_back(src[p]);
                    b.lfreq[src[p]]++;
                }
                g_tok++;
                q += l ? l : 1;
            }
            carry = s + q;
        }
        if (carry - b.in_from >= block_bytes || s + 64 >= n) {
            const uint32_t to = carry < n ? carry : n;
            emit_block(bw, b, src, to, lt, dt);
            b.reset(to);
        }
    }
    bw.put(0, 3);  // empty stored block: the region ends on a byte boundary
    bw.align();
    bw.put(0x0000, 16);
    bw.put(0xFFFF, 16);
}

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: deflate_model <in> <out.gz> [region_bytes] [block_bytes]\n");
        return 2;
    }
    const uint32_t region = argc > 3 ? (uint32_t)atol(argv[3]) : 65536u;
    const uint32_t block = argc > 4 ? (uint32_t)atol(argv[4]) : 32768u;
    if (region > MAX_REGION || region < 64) {
        fprintf(stderr, "region out of range\n");
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<uint8_t> in;
    {
        std::vector<uint8_t> buf(1 << 20);
        size_t r;
        while ((r = fread(buf.data(), 1, buf.size(), f)) > 0) in.insert(in.end(), buf.begin(), buf.begin() + r);
    }
    fclose(f);
    const size_t n = in.size();
    in.resize(n + 64, 0);  // the match finder reads a few bytes past the end
    BitWriter bw;
    static const uint8_t header[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
    for (uint8_t c : header) bw.out.push_back(c);
    if (getenv("DFL_REC")) sscanf(getenv("DFL_REC"), "%u,%u,%u", &g_reclen, &g_hdr, &g_seq);
    for (size_t r = 0; r < n; r += region) {
        g_region_base = r;
        const uint32_t len = (uint32_t)std::min<size_t>(region, n - r);
        deflate_region(bw, in.data() + r, len, block);
    }
    bw.put(3, 10);  // final block: fixed codes, end of block only
    bw.align();
    const uint32_t crc = (uint32_t)crc32(0, in.data(), (uInt)n);
    for (int i = 0; i < 4; i++) bw.out.push_back((uint8_t)(crc >> (8 * i)));
    for (int i = 0; i < 4; i++) bw.out.push_back((uint8_t)((uint32_t)n >> (8 * i)));
    f = fopen(argv[2], "wb");
    if (!f) return 1;
    fwrite(bw.out.data(), 1, bw.out.size(), f);
    fclose(f);
    fprintf(stderr,
            "in %zu out %zu ratio %.3f  blocks dyn %llu stored %llu  header %.1f B/block  tokens %llu matches %llu "
            "(avg len %.1f)\n",
            n, bw.out.size(), (double)n / bw.out.size(), (unsigned long long)g_dyn, (unsigned long long)g_stored,
            g_dyn ? g_hdr_bits / 8.0 / g_dyn : 0.0, (unsigned long long)g_tok, (unsigned long long)g_match,
            g_match ? (double)g_match_bytes / g_match : 0.0);
    if (g_reclen) {
        const char *nm[4] = {"header", "bases", "plus", "quals"};
        const double nrec = (double)n / g_reclen;
        for (int c = 0; c < 4; c++)
            fprintf(stderr, "  %-6s %6.2f B/record  literals %5.1f matches %5.2f (avg len %.1f) per record\n", nm[c],
                    g_cat_bits[c] / 8.0 / nrec, g_cat_lit[c] / nrec, g_cat_match[c] / nrec,
                    g_cat_match[c] ? (double)g_cat_mbytes[c] / g_cat_match[c] : 0.0);
    }
    return 0;
}
