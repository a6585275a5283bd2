// nh_deflate_core.h -- the lane-level and sequential pieces of the DEFLATE (RFC 1951) encoder behind the gzip
// output of nh_run (the reference's stage is gzip_compress,
// /root/reference/src/compression.rs:214-233: gzp's block-parallel encoder at the default level).  Everything here
// is plain integer code that compiles for the device (nh_deflate.hip: one wave per region of the text) and for the
// host (tools/deflate_model.cpp, tests/test_deflate_core.py: the same functions driven by a 64-lane loop), so that
// the format logic -- symbol mapping, code construction, the block header -- is checked on a CPU against zlib's
// inflate before a GPU sees it.  Written from RFC 1951; the code-length construction is the in-place minimum
// redundancy algorithm of Moffat and Katajainen (1995), the length limit a Kraft-sum repair.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NH_HD __host__ __device__
#else
#define NH_HD
#endif

namespace nh {
namespace dfl {

constexpr int NLIT = 288;   // literal / length alphabet (286 used; 288 keeps tables a multiple of 32)
constexpr int NLIT_USED = 286;
constexpr int NDIST = 32;   // distance alphabet (30 used)
constexpr int NDIST_USED = 30;
constexpr int NCL = 19;     // code-length alphabet
constexpr int MAXBITS = 15, MAXBITS_CL = 7;
constexpr uint32_t MIN_MATCH = 3, MAX_MATCH = 258, WINDOW = 32768;

// a token: literal = the byte; match = bit 31 | (length - 3) << 16 | (distance - 1)
NH_HD inline uint32_t tok_match(uint32_t len, uint32_t dist) { return 0x80000000u | ((len - 3u) << 16) | (dist - 1u); }

NH_HD inline uint32_t ilog2(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }

// length - 3 -> symbol 257.., number of extra bits and their value (RFC 1951 3.2.5)
NH_HD inline void len_symbol(uint32_t lc, uint32_t &sym, uint32_t &eb, uint32_t &ev) {
    if (lc < 8u) {
        sym = 257u + lc;
        eb = 0;
        ev = 0;
    } else if (lc == 255u) {
        sym = 285u;
        eb = 0;
        ev = 0;
    } else {
        eb = ilog2(lc) - 2u;
        sym = 261u + 4u * eb + ((lc - (4u << eb)) >> eb);
        ev = lc & ((1u << eb) - 1u);
    }
}
// distance - 1 -> symbol 0..29
NH_HD inline void dist_symbol(uint32_t d, uint32_t &sym, uint32_t &eb, uint32_t &ev) {
    if (d < 4u) {
        sym = d;
        eb = 0;
        ev = 0;
    } else {
        const uint32_t hb = ilog2(d);
        eb = hb - 1u;
        sym = 2u * hb + ((d >> (hb - 1u)) & 1u);
        ev = d & ((1u << eb) - 1u);
    }
}
NH_HD inline uint32_t len_extra_bits(uint32_t sym) {  // sym 257..285
    return (sym < 265u || sym == 285u) ? 0u : (sym - 261u) >> 2;
}
NH_HD inline uint32_t dist_extra_bits(uint32_t dsym) { return dsym < 4u ? 0u : (dsym >> 1) - 1u; }

NH_HD inline uint32_t bitrev(uint32_t v, uint32_t n) {  // the low n bits of v, reversed (n >= 1)
#if defined(__clang__)
    return __builtin_bitreverse32(v) >> (32u - n);
#else
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1u - i);
    return r;
#endif
}

// In: a[0..n) = the frequencies of the n >= 2 used symbols in ascending order (T holds their sum: 16 bits do for
// a block of up to 65535 tokens).  Out: a[i] = the code length of the
// i-th of them in an unrestricted Huffman code (non-increasing in i).  Three passes, in place.
template <typename T>
NH_HD inline void huff_depths_sorted(T *a, int n) {
    if (n == 2) {
        a[0] = a[1] = 1;
        return;
    }
    a[0] = (T)(a[0] + a[1]);
    int root = 0, leaf = 2;
    for (int next = 1; next < n - 1; next++) {
        if (leaf >= n || a[root] < a[leaf]) {
            a[next] = a[root];
            a[root++] = (T)next;
        } else {
            a[next] = a[leaf++];
        }
        if (leaf >= n || (root < next && a[root] < a[leaf])) {
            a[next] = (T)(a[next] + a[root]);
            a[root++] = (T)next;
        } else {
            a[next] = (T)(a[next] + a[leaf++]);
        }
    }
    a[n - 2] = 0;
    for (int next = n - 3; next >= 0; next--) a[next] = (T)(a[a[next]] + 1u);
    int avail = 1, used = 0, depth = 0;
    int r = n - 2, next = n - 1;
    while (avail > 0) {
        while (r >= 0 && (int)a[r] == depth) {
            used++;
            r--;
        }
        while (avail > used) {
            a[next--] = (T)depth;
            avail--;
        }
        avail = 2 * used;
        depth++;
        used = 0;
    }
}

// Lengths of a code over nsym symbols, none longer than maxbits, from the used symbols sorted by ascending
// frequency: sorted_sym[i] with frequency a[i] (a is overwritten).  lens[] gets 0 for unused symbols.
template <typename T>
NH_HD inline void huff_lengths_sorted(T *a, const uint16_t *sorted_sym, int n, int nsym, int maxbits, uint8_t *lens) {
    for (int s = 0; s < nsym; s++) lens[s] = 0;
    huff_depths_sorted(a, n);
    uint32_t cnt[MAXBITS + 2];
    for (int l = 0; l <= maxbits + 1; l++) cnt[l] = 0;
    for (int i = 0; i < n; i++) cnt[a[i] > (uint32_t)maxbits ? (uint32_t)maxbits : a[i]]++;
    // Kraft sum in units of 2^-maxbits; while it is above one: a code of the longest length is given up and a
    // shorter code is lengthened by one bit to take its sibling's place -- one unit less each time
    uint32_t total = 0;
    for (int l = 1; l <= maxbits; l++) total += cnt[l] << (maxbits - l);
    while (total > (1u << maxbits)) {
        cnt[maxbits]--;
        for (int l = maxbits - 1; l > 0; l--)
            if (cnt[l]) {
                cnt[l]--;
                cnt[l + 1] += 2;
                break;
            }
        total--;
    }
    int i = 0;  // the rarest symbols take the longest codes
    for (int l = maxbits; l >= 1; l--)
        for (uint32_t c = 0; c < cnt[l]; c++) lens[sorted_sym[i++]] = (uint8_t)l;
}

// canonical codes (RFC 1951 3.2.2), stored bit-reversed: ready to be emitted least significant bit first
NH_HD inline void huff_codes(const uint8_t *lens, int nsym, int maxbits, uint16_t *codes) {
    uint32_t cnt[MAXBITS + 2], next[MAXBITS + 2];
    for (int l = 0; l <= maxbits; l++) cnt[l] = 0;
    for (int s = 0; s < nsym; s++) cnt[lens[s]]++;
    cnt[0] = 0;
    uint32_t code = 0;
    next[0] = 0;
    for (int l = 1; l <= maxbits; l++) {
        code = (code + cnt[l - 1]) << 1;
        next[l] = code;
    }
    for (int s = 0; s < nsym; s++) {
        const uint32_t l = lens[s];
        codes[s] = l ? (uint16_t)bitrev(next[l]++, l) : (uint16_t)0;
    }
}

// Run-length form of the hlit + hdist code lengths (RFC 1951 3.2.7): item i = symbol 0..18 in the low 5 bits, the
// value of its extra bits above.  Returns the number of items (at most n) and counts the symbols into clfreq[19].
NH_HD inline int rle_lengths(const uint8_t *lens, int n, uint16_t *items, uint32_t *clfreq) {
    for (int s = 0; s < NCL; s++) clfreq[s] = 0;
    int ni = 0, i = 0;
    while (i < n) {
        const uint32_t v = lens[i];
        int run = 1;
        while (i + run < n && lens[i + run] == v) run++;
        i += run;
        if (v == 0) {
            while (run >= 11) {
                const int r = run > 138 ? 138 : run;
                items[ni++] = (uint16_t)(18u | ((uint32_t)(r - 11) << 5));
                clfreq[18]++;
                run -= r;
            }
            if (run >= 3) {
                items[ni++] = (uint16_t)(17u | ((uint32_t)(run - 3) << 5));
                clfreq[17]++;
                run = 0;
            }
        } else {
            items[ni++] = (uint16_t)v;  // the length itself, then copies of it
            clfreq[v]++;
            run--;
            while (run >= 3) {
                const int r = run > 6 ? 6 : run;
                items[ni++] = (uint16_t)(16u | ((uint32_t)(r - 3) << 5));
                clfreq[16]++;
                run -= r;
            }
        }
        for (; run > 0; run--) {
            items[ni++] = (uint16_t)v;
            clfreq[v]++;
        }
    }
    return ni;
}
NH_HD inline uint32_t cl_extra_bits(uint32_t sym) { return sym == 16u ? 2u : sym == 17u ? 3u : sym == 18u ? 7u : 0u; }
// the order in which the code-length code's own lengths are stored
NH_HD inline uint32_t cl_order(int i) {
    const uint8_t o[NCL] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    return o[i];
}

// ---- match finding, one position ------------------------------------------------------------------------------
// the bucket and the tag of the four bytes at a position
// the hash of the HASH_BYTES bytes at a position (lo = its first eight bytes).  Five bytes: on FASTQ text a 5-gram
// bucket holds fewer useless candidates among the bases than a 4-gram one and the header fields still fit (six and
// more lose them): 0.6 % smaller files than with four.
constexpr uint32_t HASH_BYTES = 5;
NH_HD inline uint32_t hash_at(uint64_t lo) {
    return (uint32_t)(((lo & ((1ull << (8u * HASH_BYTES)) - 1ull)) * 0x9E3779B97F4A7C15ull) >> 32);
}
#ifndef NH_BUCKET_BITS
#define NH_BUCKET_BITS 10
#endif
constexpr uint32_t BUCKET_BITS = NH_BUCKET_BITS;
// a bucket entry is a position of the region in 16 bits (regions are at most 64 KiB); 0xFFFF = empty: the last
// position of a full region can never be a candidate (candidates lie before the position that asks)
constexpr uint32_t EMPTY_ENTRY = 0xFFFFu;
constexpr uint32_t MAX_REGION = 65536u;
NH_HD inline uint32_t hash_bucket(uint32_t h) { return h >> (32u - BUCKET_BITS); }
NH_HD inline uint16_t make_entry(uint32_t pos) { return (uint16_t)pos; }

NH_HD inline uint64_t load8(const uint8_t *p) {  // eight bytes at any address (little endian)
    struct __attribute__((packed)) U {
        uint64_t v;
    };
    return ((const U *)p)->v;
}
struct Bytes16 {
    uint64_t lo, hi;
};
NH_HD inline Bytes16 load16(const uint8_t *p) {  // sixteen bytes at any address
    struct __attribute__((packed)) U {
        uint64_t lo, hi;
    };
    const U *u = (const U *)p;
    return Bytes16{u->lo, u->hi};
}
NH_HD inline uint32_t tz32(uint32_t x) { return x ? (uint32_t)__builtin_ctz(x) : 32u; }
NH_HD inline uint32_t equal_bytes16(const Bytes16 &a, const Bytes16 &b) {  // 0..16 equal leading bytes
    const uint64_t xl = a.lo ^ b.lo, xh = a.hi ^ b.hi;
#if defined(__HIP_DEVICE_COMPILE__)
    // v_ffbl_b32 answers 0xFFFFFFFF for a word without a set bit: "never", which an OR of the word's offset leaves as it is and an
    // unsigned minimum passes over -- fifteen instructions instead of twenty-two (a step runs ten of these a lane)
    uint32_t b0, b1, b2, b3;
    asm("v_ffbl_b32 %0, %1" : "=v"(b0) : "v"((uint32_t)xl));
    asm("v_ffbl_b32 %0, %1" : "=v"(b1) : "v"((uint32_t)(xl >> 32)));
    asm("v_ffbl_b32 %0, %1" : "=v"(b2) : "v"((uint32_t)xh));
    asm("v_ffbl_b32 %0, %1" : "=v"(b3) : "v"((uint32_t)(xh >> 32)));
    b1 |= 32u;
    b2 |= 64u;
    b3 |= 96u;
    const uint32_t m01 = b0 < b1 ? b0 : b1, m23 = b2 < b3 ? b2 : b3;
    uint32_t bits = m01 < m23 ? m01 : m23;
    bits = bits < 128u ? bits : 128u;
    return bits >> 3;
#else
    // straight-line code: trailing zeros of the four words of the difference, chained by selects
    const uint32_t t0 = tz32((uint32_t)xl), t1 = tz32((uint32_t)(xl >> 32)), t2 = tz32((uint32_t)xh), t3 = tz32((uint32_t)(xh >> 32));
    uint32_t bits = t2 + (t2 == 32u ? t3 : 0u);
    bits = t1 + (t1 == 32u ? bits : 0u);
    bits = t0 + (t0 == 32u ? bits : 0u);
    return bits >> 3;
#endif
}
// number of equal leading bytes of the strings at a and b, at most cap (reads up to 7 bytes past cap)
NH_HD inline uint32_t common_prefix(const uint8_t *a, const uint8_t *b, uint32_t from, uint32_t cap) {
    uint32_t len = from;
    while (len < cap) {
        const uint64_t x = load8(a + len) ^ load8(b + len);
        if (x) {
            len += (uint32_t)__builtin_ctzll(x) >> 3;
            break;
        }
        len += 8;
    }
    return len < cap ? len : cap;
}

#ifndef NH_SCAN_CAP
#define NH_SCAN_CAP 32
#endif
constexpr uint32_t SCAN_CAP = NH_SCAN_CAP;  // match lengths are measured up to here for every position; a match the parse takes is extended

// what a match costs and what it saves, in bits, under the code lengths of the previous block (lit_cost: 288 + 32
// entries, 0 = the symbol did not occur there)
NH_HD inline uint32_t cost_or(uint32_t len_bits, uint32_t absent) { return len_bits ? len_bits : absent; }
constexpr uint32_t ABSENT_LIT = 12u, ABSENT_LEN = 10u, ABSENT_DIST = 8u;  // what a symbol the previous block did not use is priced at
// PRICED = false: the two rows are code lengths (0 = absent), priced on every look-up.  PRICED = true: the rows hold prices already
// (price_of() of every entry, refreshed when a block's codes are built): the device's form -- three look-ups in four are literals,
// and a compare + select per look-up is a sixth of the match finder's pricing instructions.
template <bool PRICED>
struct CostsT {
    const uint8_t *llen;  // literal / length lengths (or prices)
    const uint8_t *dlen;
    NH_HD uint32_t lit(uint32_t b) const { return PRICED ? (uint32_t)llen[b] : cost_or(llen[b], ABSENT_LIT); }
    NH_HD uint32_t len(uint32_t sym) const { return PRICED ? (uint32_t)llen[sym] : cost_or(llen[sym], ABSENT_LEN); }
    NH_HD uint32_t dist(uint32_t dsym) const { return PRICED ? (uint32_t)dlen[dsym] : cost_or(dlen[dsym], ABSENT_DIST); }
};
using Costs = CostsT<false>;
NH_HD inline uint8_t price_of_litlen(uint32_t sym, uint32_t len_bits) { return (uint8_t)cost_or(len_bits, sym < 256u ? ABSENT_LIT : ABSENT_LEN); }
NH_HD inline uint8_t price_of_dist(uint32_t len_bits) { return (uint8_t)cost_or(len_bits, ABSENT_DIST); }
// prices of the first eight literals at a position as eight running sums, one per byte of the result
template <typename C>
NH_HD inline uint64_t literal_prices8(const C &c, uint64_t cur8) {
    uint64_t packed = 0;
    uint32_t acc = 0;
    for (uint32_t j = 0; j < 8; j++) {
        acc += c.lit((uint32_t)(cur8 >> (8u * j)) & 0xFFu);
        packed |= (uint64_t)acc << (8u * j);
    }
    return packed;
}
// bits saved by coding `len` bytes as a match at `dist` instead of literals (the literals behind the eighth are
// priced like the first eight on average)
template <typename C>
NH_HD inline int match_gain(const C &c, uint64_t lit8, uint32_t len, uint32_t dist) {
    uint32_t sym, eb, ev, dsym, deb, dev;
    len_symbol(len - 3u, sym, eb, ev);
    dist_symbol(dist - 1u, dsym, deb, dev);
    const int cost = (int)(c.len(sym) + eb + c.dist(dsym) + deb);
    const uint32_t lit = len <= 8u ? (uint32_t)(lit8 >> (8u * (len - 1u))) & 0xFFu : ((uint32_t)(lit8 >> 56) * len) >> 3;
    return (int)lit - cost;
}

// The best match for position p of the region src[0..n): candidates are the WAYS entries of the position's
// bucket (older positions with the same hash) and distance 1 (a run: the one short distance the bucket cannot
// hold, because a step's own positions enter it after the look-ups; distances 2..4 and the last match's distance
// were candidates too and found 0.03 % on FASTQ text).  cur16 = the sixteen bytes at p (the caller has them from
// its previous step).  All candidates are compared sixteen bytes at a time in lockstep and the loads of a round are
// unconditional -- issued back to back, waited for once: on a GPU the rounds' latency is what a step costs.  The
// longest far candidate (the nearest among equals) and the run are priced under `costs`; the one that saves more
// bits wins.  Returns the length (0: none worth taking, else 3..SCAN_CAP, SCAN_CAP meaning "at least").
// The two halves of it: match_probe() reads the bucket and issues the first round's loads (it needs the position and the
// bucket only), match_finish() does everything that looks at the bytes.  The kernel runs the probe of the NEXT step before this
// step's parse, so that the gathers are in flight while the scalar walk along the tokens runs; find_match() = one after the other.
template <int NC>
struct MatchProbe {
    uint32_t d[NC];   // candidate distances: [0] the run (1), then the bucket's ways; 0 = none
    Bytes16 x[NC];    // the sixteen bytes at each candidate (at the position itself where there is none)
};
template <int WAYS, typename EntryPtr>
NH_HD inline void match_probe(const uint8_t *src, uint32_t p, uint32_t n, EntryPtr entries, MatchProbe<1 + WAYS> &pr) {
    constexpr int NC = 1 + WAYS;
    const bool live = p < n && n - p >= MIN_MATCH;  // (a position at which match_finish() gives up before it looks: no candidates)
    const uint32_t room = live ? n - p : 0u;
    // (every address below is the region's base -- uniform -- plus a 32-bit offset that cannot be negative: d <= p.  Written so,
    //  the compiler keeps the base in scalar registers and the loads cost one subtraction each instead of a 64-bit one)
    pr.d[0] = live && p >= 1u ? 1u : 0u;
    for (int k = 0; k < WAYS; k++) {
        const uint32_t c = entries[k];
        const bool ok = room >= HASH_BYTES && c < p && p - c <= WINDOW && p - c > 1u;
        pr.d[1 + k] = ok ? p - c : 0u;
    }
    for (int k = 0; k < NC; k++) pr.x[k] = load16(src + (p - pr.d[k]));  // (distance 0 reads the position itself: readable up to n + 64)
}
template <int WAYS, typename C>
NH_HD inline uint32_t match_finish(const uint8_t *src, uint32_t p, uint32_t n, const Bytes16 &cur16, MatchProbe<1 + WAYS> &pr,
                                   const C &costs, uint32_t &dist_out, int &gain_out) {
    constexpr int NC = 1 + WAYS;
    const uint32_t room = n - p;
    const uint32_t cap = room < SCAN_CAP ? room : SCAN_CAP;
    if (cap < MIN_MATCH) return 0;
    uint32_t len[NC];
    const uint32_t *d = pr.d;
    Bytes16 *x = pr.x;
    for (int k = 0; k < NC; k++) len[k] = d[k] == 0u ? 0u : equal_bytes16(x[k], cur16);
    if (cap > 16u) {
        bool any = false;
        for (int k = 0; k < NC; k++) any |= len[k] == 16u;
        if (any) {
            const Bytes16 c2 = load16(src + (p + 16u));
            for (int k = 0; k < NC; k++) x[k] = load16(src + (p + 16u - (len[k] == 16u ? d[k] : 0u)));  // (the others: lines the wave reads anyway)
            for (int k = 0; k < NC; k++) {
                const uint32_t more = equal_bytes16(x[k], c2);
                len[k] += len[k] == 16u ? more : 0u;
            }
        }
    }
    uint32_t nl = len[0], fl = 0, fd = 0;
    for (int k = 1; k < NC; k++)
        if (len[k] > fl || (len[k] == fl && fl != 0u && d[k] < fd)) {
            fl = len[k];
            fd = d[k];
        }
    nl = nl < cap ? nl : cap;
    fl = fl < cap ? fl : cap;
    const uint64_t lit8 = literal_prices8(costs, cur16.lo);
    uint32_t best = 0, bdist = 0;
    int bgain = 0;
    if (nl >= MIN_MATCH) {
        bgain = match_gain(costs, lit8, nl, 1u);
        if (bgain > 0) {
            best = nl;
            bdist = 1u;
        } else {
            bgain = 0;
        }
    }
    if (fl >= MIN_MATCH) {
        const int g = match_gain(costs, lit8, fl, fd);
        if (g > bgain) {
            bgain = g;
            best = fl;
            bdist = fd;
        }
    }
    if (best < MIN_MATCH) return 0;
    dist_out = bdist;
    gain_out = bgain;
    return best;
}
template <int WAYS, typename EntryPtr, typename C>
NH_HD inline uint32_t find_match(const uint8_t *src, uint32_t p, uint32_t n, const Bytes16 &cur16, EntryPtr entries,
                                 const C &costs, uint32_t &dist_out, int &gain_out) {
    MatchProbe<1 + WAYS> pr;
    match_probe<WAYS>(src, p, n, entries, pr);
    return match_finish<WAYS>(src, p, n, cur16, pr, costs, dist_out, gain_out);
}

// ---- CRC-32 of the text (the gzip member's check value), computed where the text is ---------------------------
// Reflected polynomial 0xEDB88320 as in RFC 1952.  A wave's lanes each take a slice of the region; slices and
// regions are joined by crc(A || B) = crc(A) * x^(8 |B|) + crc(B) over GF(2)[x] mod P.
constexpr uint32_t CRC_POLY = 0xEDB88320u;
NH_HD inline uint32_t crc32_bytes(uint32_t crc, const uint8_t *p, uint32_t n) {  // one bit at a time, no table
    crc = ~crc;
    for (uint32_t i = 0; i < n; i++) {
        crc ^= p[i];
        for (int k = 0; k < 8; k++) crc = (crc >> 1) ^ (CRC_POLY & (0u - (crc & 1u)));
    }
    return ~crc;
}
// a(x) * b(x) mod P, both in the reflected representation (bit 31 = x^0)
NH_HD inline uint32_t gf2_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        p ^= b & (0u - ((a >> (31 - i)) & 1u));
        b = (b >> 1) ^ (CRC_POLY & (0u - (b & 1u)));
    }
    return p;
}
NH_HD inline uint32_t gf2_xpow8(uint64_t n_bytes) {  // x^(8 n) mod P
    uint32_t r = 0x80000000u;        // x^0
    uint32_t sq = 0x00800000u;       // x^8
    while (n_bytes) {
        if (n_bytes & 1u) r = gf2_mul(r, sq);
        sq = gf2_mul(sq, sq);
        n_bytes >>= 1;
    }
    return r;
}
NH_HD inline uint32_t crc32_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    return gf2_mul(crc_a, gf2_xpow8(len_b)) ^ crc_b;
}

}  // namespace dfl
}  // namespace nh
