// nh_inflate.cpp -- see nh_inflate.h.  Deflate (RFC 1951) + gzip framing (RFC 1952) decoder with a
// speculative multi-threaded mode.  Written from the specifications.
#include "nh_inflate.h"

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <atomic>
#include <chrono>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>  // crc32_combine only

#include <condition_variable>
#include <functional>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace nh {

// ---- CRC-32 ----------------------------------------------------------------------------------------
namespace {
struct CrcTables {
    uint32_t t[16][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1)));
            t[0][i] = c;
        }
        for (int s = 1; s < 16; s++)
            for (uint32_t i = 0; i < 256; i++) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 255];
    }
};
}  // namespace

#if defined(__x86_64__)
// CRC-32 by carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using
// PCLMULQDQ", Intel 2009; the bit-reflected constants for the zlib polynomial as given there): folds 64 bytes
// per step, 12 GB/s.  `c` and the result are in the inverted domain (~crc); n >= 64 and a multiple of 16.  The
// slicing-by-16 code below ran at 1.9-2.3 GB/s and was a quarter of the inflate threads' time
// (round 3, gprof: FASTQ text inflates as 97 % match bytes, ~14 ns per match, so the CRC stood out).
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul(uint32_t c, const uint8_t *buf, size_t len) {
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)c));
    x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64;
    len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
        y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
        y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64;
        len -= 64;
    }
    x0 = _mm_load_si128((const __m128i *)k3k4);  // four lanes -> one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16;
        len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);  // 128 -> 64 bits
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i *)poly);  // Barrett reduction to 32 bits
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

uint32_t crc32_fast(uint32_t crc, const uint8_t *p, size_t n) {
    static const CrcTables T;
    uint32_t c = ~crc;
#if defined(__x86_64__)
    static const bool have_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") &&
                                   getenv("NOHUMAN_NO_CLMUL") == nullptr;
    if (have_clmul && n >= 64) {
        const size_t bulk = n & ~(size_t)15;
        c = crc32_clmul(c, p, bulk);
        p += bulk;
        n -= bulk;
    }
#endif
    while (n && ((uintptr_t)p & 7)) {
        c = (c >> 8) ^ T.t[0][(c ^ *p++) & 255];
        n--;
    }
    while (n >= 16) {
        uint64_t a, b;
        memcpy(&a, p, 8);
        memcpy(&b, p + 8, 8);
        a ^= c;
        c = T.t[15][a & 255] ^ T.t[14][(a >> 8) & 255] ^ T.t[13][(a >> 16) & 255] ^ T.t[12][(a >> 24) & 255] ^
            T.t[11][(a >> 32) & 255] ^ T.t[10][(a >> 40) & 255] ^ T.t[9][(a >> 48) & 255] ^ T.t[8][a >> 56] ^
            T.t[7][b & 255] ^ T.t[6][(b >> 8) & 255] ^ T.t[5][(b >> 16) & 255] ^ T.t[4][(b >> 24) & 255] ^
            T.t[3][(b >> 32) & 255] ^ T.t[2][(b >> 40) & 255] ^ T.t[1][(b >> 48) & 255] ^ T.t[0][b >> 56];
        p += 16;
        n -= 16;
    }
    while (n--) c = (c >> 8) ^ T.t[0][(c ^ *p++) & 255];
    return ~c;
}

namespace {

const size_t WSIZE = 32768;

// ---- bit reader (LSB first) ------------------------------------------------------------------------
struct BitIn {
    const uint8_t *base = nullptr, *p = nullptr, *end = nullptr;
    uint64_t buf = 0;
    unsigned cnt = 0;  // valid bits in buf
    unsigned pad = 0;  // zero bytes appended past `end`
    void init(const uint8_t *b, const uint8_t *e, uint64_t bitpos) {
        base = b;
        end = e;
        p = b + (bitpos >> 3);
        if (p > e) p = e;
        buf = 0;
        cnt = 0;
        pad = 0;
        refill();
        drop((unsigned)(bitpos & 7));
    }
    inline void refill() {
        if (end - p >= 8) {
            uint64_t w;
            memcpy(&w, p, 8);
            buf |= w << cnt;
            p += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56) {
                if (p < end)
                    buf |= (uint64_t)*p++ << cnt;
                else
                    pad++;
                cnt += 8;
            }
        }
    }
    inline uint32_t peek(unsigned n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    inline void drop(unsigned n) {
        buf >>= n;
        cnt -= n;
    }
    inline uint32_t take(unsigned n) {
        uint32_t v = peek(n);
        drop(n);
        return v;
    }
    uint64_t bitpos() const { return (uint64_t)(p - base) * 8 + (uint64_t)pad * 8 - cnt; }
    bool overrun() const { return (uint64_t)pad * 8 > cnt; }  // bits past the end were consumed
    // continue at a byte boundary: drop the bits up to it and hand the whole bytes back
    const uint8_t *byte_pos() {
        drop(cnt & 7);
        const uint8_t *q = p + pad - cnt / 8;  // pad > 0 only if p == end
        return q;
    }
};

// ---- Huffman decode tables ---------------------------------------------------------------------------
// entry: bits 0-3 codeword bits to drop, 4-7 extra bits (K_SUB: index bits of the subtable),
//        8-11 kind, 16-31 value (literal, length / distance base, subtable start)
enum { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4 };
const int LIT_ROOT = 11, DIST_ROOT = 8, PRE_ROOT = 7;
const int LIT_CAP = 2048 + 1024, DIST_CAP = 256 + 512;
inline uint32_t mk(uint32_t value, uint32_t kind, uint32_t extra) { return value << 16 | kind << 8 | extra << 4; }
inline uint32_t e_len(uint32_t e) { return e & 15; }
inline uint32_t e_extra(uint32_t e) { return (e >> 4) & 15; }
inline uint32_t e_kind(uint32_t e) { return (e >> 8) & 15; }
inline uint32_t e_val(uint32_t e) { return e >> 16; }

struct Tables {
    uint32_t lit[LIT_CAP];
    uint32_t dist[DIST_CAP];
};

struct SymEntries {
    uint32_t lit[288], dist[32], pre[19];
    SymEntries() {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (uint32_t i = 0; i < 256; i++) lit[i] = mk(i, K_LIT, 0);
        lit[256] = mk(0, K_EOB, 0);
        for (int i = 0; i < 29; i++) lit[257 + i] = mk(lbase[i], K_LEN, lext[i]);
        lit[286] = lit[287] = mk(0, K_BAD, 0);
        for (int i = 0; i < 30; i++) dist[i] = mk(dbase[i], K_LEN, dext[i]);
        dist[30] = dist[31] = mk(0, K_BAD, 0);
        for (uint32_t i = 0; i < 19; i++) pre[i] = mk(i, K_LIT, 0);
    }
};
const SymEntries SYM;

inline uint32_t rev16(uint32_t x) {
    x = ((x & 0x5555) << 1) | ((x >> 1) & 0x5555);
    x = ((x & 0x3333) << 2) | ((x >> 2) & 0x3333);
    x = ((x & 0x0F0F) << 4) | ((x >> 4) & 0x0F0F);
    return ((x & 0x00FF) << 8) | ((x >> 8) & 0x00FF);
}

// Canonical Huffman code -> lookup table.  Accepts what zlib accepts: a complete code, or a single
// one-bit code, or (allow_empty) no code at all.  false = invalid lengths.
bool build_table(const uint8_t *lens, int n, int root, uint32_t *table, int cap, const uint32_t *sym_entry,
                 bool allow_empty) {
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    int maxlen = 15;
    while (maxlen > 0 && !count[maxlen]) maxlen--;
    const int size = 1 << root;
    for (int i = 0; i < size; i++) table[i] = mk(0, K_BAD, 0);
    if (maxlen == 0) return allow_empty;
    int left = 1;
    for (int len = 1; len <= 15; len++) {
        left = (left << 1) - count[len];
        if (left < 0) return false;  // over-subscribed
    }
    if (left > 0 && maxlen != 1) return false;  // incomplete
    uint32_t next[16];
    uint32_t code = 0;
    for (int len = 1; len <= 15; len++) {
        code = (code + (uint32_t)count[len - 1]) << 1;
        next[len] = code;
    }
    uint16_t rcode[288];
    uint8_t submax[1 << LIT_ROOT];
    const bool has_sub = maxlen > root;
    if (has_sub) memset(submax, 0, (size_t)size);
    for (int s = 0; s < n; s++) {
        const int len = lens[s];
        if (!len) continue;
        const uint32_t r = rev16(next[len]++) >> (16 - len);
        rcode[s] = (uint16_t)r;
        if (len > root) {
            const uint32_t prefix = r & (uint32_t)(size - 1);
            if (submax[prefix] < len) submax[prefix] = (uint8_t)len;
        }
    }
    int used = size;
    for (int s = 0; s < n; s++) {
        const int len = lens[s];
        if (!len) continue;
        const uint32_t r = rcode[s];
        if (len <= root) {
            const uint32_t e = sym_entry[s] | (uint32_t)len;
            for (uint32_t i = r; i < (uint32_t)size; i += 1u << len) table[i] = e;
        } else {
            const uint32_t prefix = r & (uint32_t)(size - 1);
            if (e_kind(table[prefix]) != K_SUB) {
                const int sb = submax[prefix] - root;
                if (used + (1 << sb) > cap) return false;
                for (int i = 0; i < (1 << sb); i++) table[used + i] = mk(0, K_BAD, 0);
                table[prefix] = mk((uint32_t)used, K_SUB, (uint32_t)sb) | (uint32_t)root;
                used += 1 << sb;
            }
            const uint32_t sub = table[prefix];
            const uint32_t e = sym_entry[s] | (uint32_t)(len - root);
            for (uint32_t i = r >> root; i < (1u << e_extra(sub)); i += 1u << (len - root)) table[e_val(sub) + i] = e;
        }
    }
    return true;
}

struct FixedTables {
    Tables t;
    FixedTables() {
        uint8_t l[288], d[32];
        for (int i = 0; i < 144; i++) l[i] = 8;
        for (int i = 144; i < 256; i++) l[i] = 9;
        for (int i = 256; i < 280; i++) l[i] = 7;
        for (int i = 280; i < 288; i++) l[i] = 8;
        for (int i = 0; i < 32; i++) d[i] = 5;
        build_table(l, 288, LIT_ROOT, t.lit, LIT_CAP, SYM.lit, false);
        build_table(d, 32, DIST_ROOT, t.dist, DIST_CAP, SYM.dist, true);
    }
};
const FixedTables FIXED;

// Reads the code lengths of a dynamic block (the bits after BTYPE) and builds its tables.
// false = not a valid header.
bool read_dynamic(BitIn &in, Tables &T) {
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    in.refill();
    const int nlit = (int)in.take(5) + 257, ndist = (int)in.take(5) + 1, ncode = (int)in.take(4) + 4;
    if (nlit > 286 || ndist > 30) return false;
    uint8_t plen[19] = {0};
    in.refill();  // >= 56 bits: 14 now + up to 57 for the code length code, refill once in between
    for (int i = 0; i < ncode; i++) {
        if (i == 12) in.refill();
        plen[order[i]] = (uint8_t)in.take(3);
    }
    uint32_t pre[1 << PRE_ROOT];
    {  // the code length code must be complete (zlib: "invalid code lengths set")
        int left = 1 << 7, any = 0;
        for (int i = 0; i < 19; i++)
            if (plen[i]) {
                left -= 1 << (7 - plen[i]);
                any = 1;
            }
        if (!any || left != 0) return false;
    }
    if (!build_table(plen, 19, PRE_ROOT, pre, 1 << PRE_ROOT, SYM.pre, false)) return false;
    uint8_t lens[320];
    const int total = nlit + ndist;
    int i = 0;
    while (i < total) {
        in.refill();
        const uint32_t e = pre[in.peek(PRE_ROOT)];
        if (e_kind(e) != K_LIT) return false;
        in.drop(e_len(e));
        const uint32_t sym = e_val(e);
        if (sym < 16) {
            lens[i++] = (uint8_t)sym;
            continue;
        }
        int rep;
        uint8_t v = 0;
        if (sym == 16) {
            if (i == 0) return false;
            v = lens[i - 1];
            rep = 3 + (int)in.take(2);
        } else if (sym == 17) {
            rep = 3 + (int)in.take(3);
        } else {
            rep = 11 + (int)in.take(7);
        }
        if (i + rep > total) return false;
        while (rep--) lens[i++] = v;
    }
    if (in.overrun()) return false;
    if (lens[256] == 0) return false;  // no end-of-block code
    if (!build_table(lens, nlit, LIT_ROOT, T.lit, LIT_CAP, SYM.lit, false)) return false;
    if (!build_table(lens + nlit, ndist, DIST_ROOT, T.dist, DIST_CAP, SYM.dist, true)) return false;
    return true;
}

// Does a non-final dynamic block header start at this bit position?  (the chunk seam test)
bool is_candidate(const uint8_t *base, const uint8_t *end, uint64_t bitpos, Tables &scratch) {
    const uint8_t *p = base + (bitpos >> 3);
    if (end - p < 16) return false;
    uint64_t w;
    memcpy(&w, p, 8);
    w >>= bitpos & 7;
    if ((w & 7) != 4) return false;              // BFINAL = 0, BTYPE = 10b
    if (((w >> 3) & 31) > 29) return false;      // HLIT
    if (((w >> 8) & 31) > 29) return false;      // HDIST
    BitIn in;
    in.init(base, end, bitpos + 3);
    return read_dynamic(in, scratch);
}

// First candidate in [from, to) (bit positions), or (uint64_t)-1
uint64_t find_block(const uint8_t *base, const uint8_t *end, uint64_t from, uint64_t to, Tables &scratch) {
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    (void)order;
    for (uint64_t b = from; b < to; b++) {
        const uint8_t *p = base + (b >> 3);
        if (end - p < 24) return (uint64_t)-1;
        uint64_t w;
        memcpy(&w, p, 8);
        w >>= b & 7;  // >= 57 valid bits
        if ((w & 7) != 4) continue;
        if (((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;
        // quick Kraft test of the code length code on the bits at hand (17 + 3*ncode <= 74: the first
        // 13 lengths are in w, the rest in the next word)
        const int ncode = (int)((w >> 13) & 15) + 4;
        uint64_t w2;
        memcpy(&w2, p + 7, 8);
        w2 >>= b & 7;  // bits 56.. of the window
        int left = 128, any = 0;
        for (int i = 0; i < ncode; i++) {
            const unsigned at = 17 + 3 * (unsigned)i;
            const unsigned l = at + 3 <= 56 ? (unsigned)(w >> at) & 7 : (unsigned)(w2 >> (at - 56)) & 7;
            if (l) {
                left -= 128 >> l;
                any = 1;
            }
        }
        if (!any || left != 0) continue;
        if (is_candidate(base, end, b, scratch)) return b;
    }
    return (uint64_t)-1;
}

// ---- block decoders ---------------------------------------------------------------------------------
enum { D_BLOCK_END = 0, D_OUT_FULL = 1, D_CLEAN = 2, D_ERR = -1 };

// Bytes with known history: sources may reach back to min_src.  Stops when op >= limit; the caller
// keeps 258 + 16 bytes of slack after limit.
int decode_bytes(BitIn &in, const Tables &T, const uint8_t *min_src, uint8_t *&op_ref, uint8_t *limit) {
    uint8_t *op = op_ref;
    int rc;
    for (;;) {
        if (op >= limit) {
            rc = D_OUT_FULL;
            break;
        }
        in.refill();
        uint32_t e = T.lit[in.peek(LIT_ROOT)];
        if (e_kind(e) == K_SUB) {
            in.drop(LIT_ROOT);
            e = T.lit[e_val(e) + in.peek(e_extra(e))];
        }
        in.drop(e_len(e));
        if (e_kind(e) == K_LIT) {
            *op++ = (uint8_t)e_val(e);
            // a second and third literal fit in the bits at hand (3 x 15 <= 56)
            e = T.lit[in.peek(LIT_ROOT)];
            if (e_kind(e) == K_LIT) {
                in.drop(e_len(e));
                *op++ = (uint8_t)e_val(e);
                e = T.lit[in.peek(LIT_ROOT)];
                if (e_kind(e) == K_LIT) {
                    in.drop(e_len(e));
                    *op++ = (uint8_t)e_val(e);
                }
            }
            continue;
        }
        if (e_kind(e) == K_LEN) {
            const uint32_t len = e_val(e) + in.take(e_extra(e));
            uint32_t d = T.dist[in.peek(DIST_ROOT)];
            if (e_kind(d) == K_SUB) {
                in.drop(DIST_ROOT);
                d = T.dist[e_val(d) + in.peek(e_extra(d))];
            }
            if (e_kind(d) != K_LEN) {
                rc = D_ERR;
                break;
            }
            in.drop(e_len(d));
            const uint32_t dist = e_val(d) + in.take(e_extra(d));
            if (dist > (size_t)(op - min_src)) {
                rc = D_ERR;
                break;
            }
            const uint8_t *src = op - dist;
            uint8_t *stop = op + len;
            if (dist >= 8) {
                // sixteen bytes without a question (most matches end there: no loop exit to mispredict), then the rest
                uint64_t w;
                memcpy(&w, src, 8);
                memcpy(op, &w, 8);
                memcpy(&w, src + 8, 8);
                memcpy(op + 8, &w, 8);
                if (len > 16) {
                    src += 16;
                    op += 16;
                    do {
                        memcpy(&w, src, 8);
                        memcpy(op, &w, 8);
                        src += 8;
                        op += 8;
                    } while (op < stop);
                }
            } else if (dist == 1) {
                memset(op, *src, len);
            } else {
                do *op++ = *src++;
                while (op < stop);
            }
            op = stop;
            continue;
        }
        rc = e_kind(e) == K_EOB ? D_BLOCK_END : D_ERR;
        break;
    }
    op_ref = op;
    if (in.overrun()) return D_ERR;
    return rc;
}

// 16-bit symbols with unknown history: `start` is the first symbol of the chunk; a source before it
// becomes the marker 0x8000 | index into the 32 KiB window that precedes the chunk.  marker_end = one
// past the index of the last symbol that is (or may be a copy of) a marker.  Returns D_CLEAN as soon
// as the last 32 KiB are free of markers.
int decode_markers(BitIn &in, const Tables &T, uint16_t *start, uint16_t *&op_ref, uint16_t *limit,
                   size_t &marker_end) {
    uint16_t *op = op_ref;
    int rc;
    for (;;) {
        if ((size_t)(op - start) >= marker_end + WSIZE) {
            rc = D_CLEAN;
            break;
        }
        if (op >= limit) {
            rc = D_OUT_FULL;
            break;
        }
        in.refill();
        uint32_t e = T.lit[in.peek(LIT_ROOT)];
        if (e_kind(e) == K_SUB) {
            in.drop(LIT_ROOT);
            e = T.lit[e_val(e) + in.peek(e_extra(e))];
        }
        in.drop(e_len(e));
        if (e_kind(e) == K_LIT) {
            *op++ = (uint16_t)e_val(e);
            e = T.lit[in.peek(LIT_ROOT)];  // up to two more literals from the bits at hand
            if (e_kind(e) == K_LIT) {
                in.drop(e_len(e));
                *op++ = (uint16_t)e_val(e);
                e = T.lit[in.peek(LIT_ROOT)];
                if (e_kind(e) == K_LIT) {
                    in.drop(e_len(e));
                    *op++ = (uint16_t)e_val(e);
                }
            }
            continue;
        }
        if (e_kind(e) == K_LEN) {
            const uint32_t len = e_val(e) + in.take(e_extra(e));
            uint32_t d = T.dist[in.peek(DIST_ROOT)];
            if (e_kind(d) == K_SUB) {
                in.drop(DIST_ROOT);
                d = T.dist[e_val(d) + in.peek(e_extra(d))];
            }
            if (e_kind(d) != K_LEN) {
                rc = D_ERR;
                break;
            }
            in.drop(e_len(d));
            const uint32_t dist = e_val(d) + in.take(e_extra(d));
            const size_t produced = (size_t)(op - start);
            uint32_t i = 0;
            if (dist > produced) {  // starts in the unknown window
                if (dist - produced > WSIZE) {
                    rc = D_ERR;
                    break;
                }
                const uint32_t in_window = (uint32_t)(dist - produced) < len ? (uint32_t)(dist - produced) : len;
                const uint32_t w0 = (uint32_t)(WSIZE - (dist - produced));
                for (; i < in_window; i++) op[i] = (uint16_t)(0x8000u | (w0 + i));
                marker_end = produced + len;
                const uint16_t *src = op - dist;
                for (; i < len; i++) op[i] = src[i];
            } else {
                const uint16_t *src = op - dist;
                uint64_t acc = 0;
#if defined(__x86_64__)
                if (dist >= 8) {
                    // eight symbols at a time, past the end of the match like the byte decoder (the buffer has the
                    // slack; what is written beyond the match is overwritten by the next symbols; a marker among the
                    // extra symbols only makes marker_end conservative)
                    // (sixteen symbols without a question: most matches end there)
                    const __m128i w0 = _mm_loadu_si128((const __m128i *)src);
                    _mm_storeu_si128((__m128i *)op, w0);
                    const __m128i w1 = _mm_loadu_si128((const __m128i *)(src + 8));
                    _mm_storeu_si128((__m128i *)(op + 8), w1);
                    __m128i a = _mm_or_si128(w0, w1);
                    if (len > 16) {
                        const uint16_t *s2 = src + 16;
                        uint16_t *o = op + 16, *stop = op + len;
                        do {
                            const __m128i w = _mm_loadu_si128((const __m128i *)s2);
                            a = _mm_or_si128(a, w);
                            _mm_storeu_si128((__m128i *)o, w);
                            s2 += 8;
                            o += 8;
                        } while (o < stop);
                    }
                    if (_mm_movemask_epi8(a) & 0xAAAA) marker_end = produced + len;
                    i = len;
                } else
#endif
                if (dist >= 4) {  // four symbols at a time: the source word ends before the target word
                    for (; i + 4 <= len; i += 4) {
                        uint64_t w;
                        memcpy(&w, src + i, 8);
                        acc |= w;
                        memcpy(op + i, &w, 8);
                    }
                } else if (dist == 1) {
                    const uint64_t w = (uint64_t)src[0] * 0x0001000100010001ull;
                    acc = w;
                    for (; i + 4 <= len; i += 4) memcpy(op + i, &w, 8);
                }
                for (; i < len; i++) {
                    const uint16_t v = src[i];
                    acc |= v;
                    op[i] = v;
                }
                if (acc & 0x8000800080008000ull) marker_end = produced + len;
            }
            op += len;
            continue;
        }
        rc = e_kind(e) == K_EOB ? D_BLOCK_END : D_ERR;
        break;
    }
    op_ref = op;
    if (in.overrun()) return D_ERR;
    return rc;
}

// ---- gzip framing -----------------------------------------------------------------------------------
// Parses a member header at p.  Returns the position of the deflate data, nullptr if p does not hold
// a gzip header (not an error after the first member: trailing bytes are ignored like gzip does),
// or (const uint8_t *)1 if the header is cut short.
const uint8_t *TRUNCATED = (const uint8_t *)1;
const uint8_t *parse_gzip_header(const uint8_t *p, const uint8_t *end) {
    if (end - p < 10) return end - p >= 2 && p[0] == 0x1f && p[1] == 0x8b ? TRUNCATED : nullptr;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return nullptr;
    const int flg = p[3];
    p += 10;
    if (flg & 4) {
        if (end - p < 2) return TRUNCATED;
        const size_t xlen = p[0] | (size_t)p[1] << 8;
        p += 2;
        if ((size_t)(end - p) < xlen) return TRUNCATED;
        p += xlen;
    }
    for (int bit : {8, 16})  // file name, comment: zero-terminated
        if (flg & bit) {
            const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(end - p));
            if (!z) return TRUNCATED;
            p = z + 1;
        }
    if (flg & 2) {
        if (end - p < 2) return TRUNCATED;
        p += 2;
    }
    return p;
}

struct MemberEnd {
    uint64_t out_pos;  // offset in the owning piece of output
    uint32_t crc, isize;
};

// Growable output of one chunk: 16-bit part (speculative start) followed by an 8-bit part.
struct ByteBuf {  // bytes with WSIZE of history room in front
    uint8_t *mem = nullptr;
    size_t cap = 0;  // capacity after the history room
    ~ByteBuf() { free(mem); }
    uint8_t *data() { return mem + WSIZE; }
    bool reserve(size_t n) {
        if (n <= cap) return true;
        size_t c = cap ? cap : (1u << 20);
        while (c < n) c += c / 2;
        uint8_t *q = (uint8_t *)realloc(mem, c + WSIZE);
        if (!q) return false;
        mem = q;
        cap = c;
        return true;
    }
};

struct U16Buf {  // 16-bit symbols, never zero-filled
    uint16_t *mem = nullptr;
    size_t cap = 0;
    ~U16Buf() { free(mem); }
    uint16_t *data() { return mem; }
    size_t size() const { return cap; }
    bool reserve(size_t n) {
        if (n <= cap) return true;
        size_t c = cap ? cap : (1u << 19);
        while (c < n) c += c / 2;
        uint16_t *q = (uint16_t *)realloc(mem, c * 2);
        if (!q) return false;
        mem = q;
        cap = c;
        return true;
    }
};

enum StopKind { STOP_NONE = 0, STOP_BOUNDARY, STOP_STREAM_END, STOP_ERROR };

// Decodes blocks (and walks over member trailers / headers) from a block boundary until `want_stop`
// says so at a block boundary, or the stream ends.
struct ChunkDecoder {
    const uint8_t *base, *end;
    Tables T, scratch;
    // output
    U16Buf out16;
    size_t n16 = 0;
    ByteBuf out8;
    size_t n8 = 0;
    ptrdiff_t floor_off = 0;  // sources may reach back to out8.data() + floor_off (history / member start)
    bool bytes_mode = false;
    size_t marker_end = 0;
    std::vector<MemberEnd> members;
    size_t out_limit = (size_t)-1;  // stop at the first block boundary with this much output (inflate_from)
    // result
    uint64_t start_bit = 0, end_bit = 0;
    StopKind stop = STOP_NONE;
    std::string error;

    size_t total() const { return n16 + n8; }

    void reset(const uint8_t *b, const uint8_t *e) {
        base = b;
        end = e;
        n16 = n8 = marker_end = 0;
        floor_off = 0;
        at_block = nullptr;
        bytes_mode = false;
        members.clear();
        stop = STOP_NONE;
        error.clear();
    }
    // start with a known window (the WSIZE bytes before the start) or with none at all (member start)
    void start_bytes(const uint8_t *window, size_t window_len) {
        bytes_mode = true;
        out8.reserve(1u << 20);
        if (window_len) memcpy(out8.data() - window_len, window, window_len);
        floor_off = -(ptrdiff_t)window_len;
    }
    bool fail(const char *msg) {
        error = msg;
        stop = STOP_ERROR;
        return false;
    }
    // Called at block boundaries while the output is still 16-bit (speculative chunks): the owner may
    // hand over the window that precedes the chunk as soon as it is known, see adopt_window().
    // Returning false aborts the chunk (its start turned out not to be a block boundary).
    std::function<bool(ChunkDecoder &)> at_block;

    // The WSIZE bytes before the chunk are known now: the last WSIZE symbols (markers replaced) become
    // the byte decoder's history and the rest of the chunk is decoded as bytes.
    bool adopt_window(const uint8_t *window) {
        if (!out8.reserve(1u << 20)) return false;
        uint8_t *h = out8.data() - WSIZE;
        const uint16_t *s = out16.data();
        if (n16 >= WSIZE) {
            s += n16 - WSIZE;
            for (size_t i = 0; i < WSIZE; i++) h[i] = s[i] & 0x8000u ? window[s[i] & 0x7fffu] : (uint8_t)s[i];
        } else {
            memcpy(h, window + n16, WSIZE - n16);
            uint8_t *d = h + (WSIZE - n16);
            for (size_t i = 0; i < n16; i++) d[i] = s[i] & 0x8000u ? window[s[i] & 0x7fffu] : (uint8_t)s[i];
        }
        floor_off = -(ptrdiff_t)WSIZE;
        bytes_mode = true;
        return true;
    }

    void to_bytes_mode() {  // the last WSIZE symbols are clean: they become the byte decoder's history
        out8.reserve(1u << 20);
        uint8_t *h = out8.data() - WSIZE;
        const uint16_t *s = out16.data() + n16 - WSIZE;
        for (size_t i = 0; i < WSIZE; i++) h[i] = (uint8_t)s[i];
        floor_off = -(ptrdiff_t)WSIZE;
        bytes_mode = true;
    }

    // one deflate block starting at the header bits; false on error
    bool block(BitIn &in, bool &final) {
        in.refill();
        final = in.take(1) != 0;
        const uint32_t type = in.take(2);
        const Tables *tab = &T;
        if (type == 0) {  // stored
            const uint8_t *q = in.byte_pos();
            if (end - q < 4) return fail("truncated stored block");
            const uint32_t len = q[0] | (uint32_t)q[1] << 8, nlen = q[2] | (uint32_t)q[3] << 8;
            if ((len ^ nlen) != 0xffff) return fail("invalid stored block lengths");
            q += 4;
            if ((size_t)(end - q) < len) return fail("truncated stored block");
            if (bytes_mode) {
                if (!out8.reserve(n8 + len + 512)) return fail("out of memory");
                memcpy(out8.data() + n8, q, len);
                n8 += len;
            } else {
                if (!out16.reserve(n16 + len + 512)) return fail("out of memory");
                for (uint32_t i = 0; i < len; i++) out16.data()[n16 + i] = q[i];
                n16 += len;
                if (n16 >= marker_end + WSIZE) to_bytes_mode();
            }
            in.init(base, end, (uint64_t)(q + len - base) * 8);
            return true;
        }
        if (type == 3) return fail("invalid block type");
        if (type == 1)
            tab = &FIXED.t;
        else if (!read_dynamic(in, T))
            return fail("invalid code lengths set");
        for (;;) {
            int rc;
            if (bytes_mode) {
                if (!out8.reserve(n8 + (1u << 18))) return fail("out of memory");
                uint8_t *op = out8.data() + n8;
                rc = decode_bytes(in, *tab, out8.data() + floor_off, op, out8.data() + out8.cap - 320);
                n8 = (size_t)(op - out8.data());
                if (rc == D_OUT_FULL) {
                    if (!out8.reserve(out8.cap + out8.cap / 2)) return fail("out of memory");
                    continue;
                }
            } else {
                if (!out16.reserve(n16 + (1u << 17))) return fail("out of memory");
                uint16_t *op = out16.data() + n16;
                rc = decode_markers(in, *tab, out16.data(), op, out16.data() + out16.size() - 320, marker_end);
                n16 = (size_t)(op - out16.data());
                if (rc == D_OUT_FULL) continue;
                if (rc == D_CLEAN) {
                    to_bytes_mode();
                    continue;
                }
            }
            if (rc == D_ERR) return fail("invalid deflate data");
            return true;  // D_BLOCK_END
        }
    }

    // Decode from `from_bit` (a block header).  Stops at the first block boundary b for which
    // b >= soft_bit and the seam test holds, or b >= hard_bit; or at the end of the stream.
    // exact_stop: stop at the first boundary >= soft_bit without the seam test (gap decoding).
    void run(uint64_t from_bit, uint64_t soft_bit, uint64_t hard_bit, bool exact_stop) {
        BitIn in;
        in.init(base, end, from_bit);
        start_bit = from_bit;
        for (;;) {
            if (!bytes_mode && at_block && !at_block(*this)) {
                fail("superseded");
                return;
            }
            bool final = false;
            if (!block(in, final)) return;
            if (final) {
                const uint8_t *q = in.byte_pos();
                if (end - q < 8) {
                    fail("unexpected end of file");
                    return;
                }
                MemberEnd m;
                m.out_pos = total();
                m.crc = q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24;
                m.isize = q[4] | (uint32_t)q[5] << 8 | (uint32_t)q[6] << 16 | (uint32_t)q[7] << 24;
                members.push_back(m);
                q += 8;
                const uint8_t *d = q == end ? nullptr : parse_gzip_header(q, end);
                if (d == TRUNCATED) {
                    fail("unexpected end of file");
                    return;
                }
                if (!d) {  // end of input, or bytes that are not a gzip member: ignored like gzip does
                    end_bit = (uint64_t)(q - base) * 8;
                    stop = STOP_STREAM_END;
                    return;
                }
                // a new member starts with an empty window: no markers can follow
                if (!bytes_mode) {
                    bytes_mode = true;
                    out8.reserve(1u << 20);
                }
                floor_off = (ptrdiff_t)n8;  // sources may not reach before the member's first byte
                in.init(base, end, (uint64_t)(d - base) * 8);
            }
            const uint64_t b = in.bitpos();
            if (b >= hard_bit || total() >= out_limit || (b >= soft_bit && (exact_stop || is_candidate(base, end, b, scratch)))) {
                end_bit = b;
                stop = STOP_BOUNDARY;
                return;
            }
        }
    }
};


// ---- chunk orchestration ---------------------------------------------------------------------------
struct Seg {  // stretch of a piece's output that belongs to one gzip member
    size_t len;
    uint32_t crc;
    bool member_end;
    uint32_t want_crc, want_isize;
};

struct Chunk {
    size_t index = 0;
    ChunkDecoder dec;
    bool found = false;
    bool chained = false;  // decoded from the known end of its predecessor: start and window are certain
    bool spec_done = false, resolved = false;
    uint8_t window_before[WSIZE];
    std::vector<Seg> segs;
    size_t read_off = 0;  // consumer: bytes already handed out
    bool verified = false;
    size_t out_off = 0;   // range mode: where the chunk's text goes in the caller's buffer
};

// the chunk's output bytes [from, to) as one or two contiguous ranges, applied to f(ptr, n); the
// 16-bit part has been narrowed in place by resolve_chunk()
template <class F>
void for_ranges(Chunk &c, size_t from, size_t to, F f) {
    const size_t n16 = c.dec.n16;
    if (from < n16) {
        const size_t e = to < n16 ? to : n16;
        f((const uint8_t *)c.dec.out16.data() + from, e - from);
        from = e;
    }
    if (from < to) f(c.dec.out8.data() + (from - n16), to - from);
}

// Replaces the markers of the 16-bit part from the window before the chunk (narrowing it to bytes in
// place, front to back) and computes the CRC-32 of every member stretch, block by block while the
// bytes are in cache.
void resolve_chunk(Chunk &c) {
    const size_t n16 = c.dec.n16, total = c.dec.total();
    const uint16_t *s = c.dec.out16.data();
    uint8_t *d = (uint8_t *)c.dec.out16.data();
    const uint8_t *w = c.window_before;
    c.segs.clear();
    size_t pos = 0, mi = 0;
    uint32_t crc = 0;
    size_t seg_start = 0;
    const size_t BLK = 16384;
    while (pos < total || mi < c.dec.members.size()) {
        const size_t seg_end = mi < c.dec.members.size() ? (size_t)c.dec.members[mi].out_pos : total;
        while (pos < seg_end) {
            size_t e = pos + BLK < seg_end ? pos + BLK : seg_end;
            if (pos < n16) {
                if (e > n16) e = n16;
                // (the window byte is loaded whether the symbol is a marker or not -- the index is always inside the
                // window -- so that the choice is a conditional move: markers and literals alternate without a pattern)
                for (size_t i = pos; i < e; i++) {
                    const uint16_t v = s[i];
                    const uint8_t from_window = w[v & 0x7fffu];
                    d[i] = v & 0x8000u ? from_window : (uint8_t)v;
                }
                crc = crc32_fast(crc, d + pos, e - pos);
            } else {
                crc = crc32_fast(crc, c.dec.out8.data() + (pos - n16), e - pos);
            }
            pos = e;
        }
        if (mi < c.dec.members.size()) {
            c.segs.push_back({seg_end - seg_start, crc, true, c.dec.members[mi].crc, c.dec.members[mi].isize});
            mi++;
        } else {
            c.segs.push_back({seg_end - seg_start, crc, false, 0, 0});
        }
        seg_start = seg_end;
        crc = 0;
    }
}

}  // namespace

class GunzipImpl {
public:
    ~GunzipImpl() { close(); }

    int open(const char *path, unsigned threads, size_t chunk_bytes, std::string &err) {
        close();
        fd_ = ::open(path, O_RDONLY | O_CLOEXEC);
        if (fd_ < 0) {
            err = std::string("cannot open ") + path;
            return -1;
        }
        struct stat st;
        if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) {
            err = std::string("not a regular gzip file: ") + path;
            close();
            return -1;
        }
        size_ = (size_t)st.st_size;
        void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) {
            err = std::string("cannot map ") + path;
            close();
            return -1;
        }
        base_ = (const uint8_t *)m;
        const uint8_t *d = parse_gzip_header(base_, base_ + size_);
        if (!d || d == TRUNCATED) {
            err = std::string("not in gzip format: ") + path;
            close();
            return -1;
        }
        first_block_bit_ = (uint64_t)(d - base_) * 8;
        threads_ = threads < 1 ? 1 : threads;
        chunk_ = chunk_bytes ? chunk_bytes : (size_t)(4u << 20);
        if (chunk_ < 64) chunk_ = 64;
        n_chunks_ = (size_ + chunk_ - 1) / chunk_;
        max_ahead_ = 2 * (size_t)threads_ + 2;
        force_chain_ = threads_ <= 2;
        next_dispatch_ = next_stitch_ = 0;
        ended_ = false;
        failed_ = false;
        quit_ = false;
        P_ = first_block_bit_;
        memset(window_, 0, sizeof window_);
        cur_crc_ = 0;
        cur_len_ = 0;
        accepted_ = rejected_ = gap_bytes_ = 0;
        for (unsigned i = 0; i < threads_; i++) workers_.emplace_back([this] { worker(); });
        return 0;
    }

    void close() {
        if ((t_wait_ns_ || t_copy_ns_) && getenv("NOHUMAN_TRACE"))
            fprintf(stderr, "[nohuman trace] gunzip consumer: waited %.3f s for chunks, copied for %.3f s; workers: decoded for %.3f s, "
                            "replaced markers + CRC for %.3f s (accepted %llu, rejected %llu chunks)\n",
                    t_wait_ns_ / 1e9, t_copy_ns_ / 1e9, t_spec_ns_.load() / 1e9, t_resolve_ns_.load() / 1e9,
                    (unsigned long long)accepted_, (unsigned long long)rejected_);
        t_wait_ns_ = t_copy_ns_ = 0;
        t_spec_ns_ = 0;
        t_resolve_ns_ = 0;
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_work_.notify_all();
        for (auto &t : workers_) t.join();
        workers_.clear();
        jobs_.clear();
        resolve_jobs_.clear();
        tails_.clear();
        inflight_.clear();
        pieces_.clear();
        spare_.clear();
        if (base_ && !range_) munmap((void *)base_, size_);
        base_ = nullptr;
        if (fd_ >= 0) ::close(fd_);
        fd_ = -1;
        range_ = false;
        range_dst_ = nullptr;
    }

    // ---- range mode (RangeGunzip, nh_inflate.h): the bytes [lo, hi) of a gzip file image that somebody else has mapped are
    // decoded by this object's workers -- every chunk speculatively, at once, without knowing where the stream stands --
    // and stitched later, when the caller knows the stream's position and window at the range (finish_range).
    int open_range(const uint8_t *base, size_t size, uint64_t lo_byte, uint64_t hi_byte, unsigned threads, size_t chunk_bytes) {
        close();
        range_ = true;
        base_ = base;
        size_ = size;
        lo_ = lo_byte < size ? lo_byte : size;
        hi_ = hi_byte < size ? hi_byte : size;
        if (hi_ < lo_) hi_ = lo_;
        threads_ = threads < 1 ? 1 : threads;
        chunk_ = chunk_bytes ? chunk_bytes : (size_t)(4u << 20);
        if (chunk_ < 64) chunk_ = 64;
        n_chunks_ = (size_t)((hi_ - lo_ + chunk_ - 1) / chunk_);
        max_ahead_ = n_chunks_ + 2;  // (every chunk of the range is decoded ahead)
        force_chain_ = false;        // (the start of the range is not known: nothing to chain from)
        first_block_bit_ = 0;
        next_dispatch_ = next_stitch_ = 0;
        ended_ = failed_ = quit_ = false;
        error_.clear();
        P_ = lo_ * 8;
        memset(window_, 0, sizeof window_);
        cur_crc_ = 0;
        cur_len_ = 0;
        accepted_ = rejected_ = gap_bytes_ = 0;
        out_total_ = 0;
        for (unsigned i = 0; i < threads_; i++) workers_.emplace_back([this] { worker(); });
        dispatch();
        return 0;
    }
    void wait_speculated() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] {
            for (auto &c : inflight_)
                if (!c->spec_done) return false;
            return true;
        });
    }
    // The stream stands at from_bit (a block boundary at or behind the range's first byte) with `window` before it: the
    // range's text goes to dst; returns its length (-1: error()), the bit the stream stands at afterwards -- the first block
    // boundary at or behind the range's end that passes the seam test, or the end of the stream --, the window there and the
    // stretches of text by gzip member with their CRC-32 (the caller keeps the members' books: a member may have begun
    // long before the range).
    long finish_range(uint64_t from_bit, const uint8_t *window, uint8_t *dst, size_t cap, uint64_t *end_bit, bool *stream_end,
                      uint8_t *window_after, std::vector<GzSeg> &segs) {
        if (!range_ || failed_) return -1;
        P_ = from_bit;
        memcpy(window_, window, WSIZE);
        range_dst_ = dst;
        range_cap_ = cap;
        out_total_ = 0;
        while (!failed_ && !ended_) {
            if (inflight_.empty() && P_ >= hi_ * 8) break;
            (void)stitch_next(true);
        }
        if (failed_) return -1;
        segs.clear();
        for (auto &c : pieces_) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_done_.wait(lk, [&] { return c->resolved; });
            }
            for (const Seg &sg : c->segs) segs.push_back({(uint64_t)sg.len, sg.crc, sg.member_end, sg.want_crc, sg.want_isize});
        }
        while (!pieces_.empty()) {
            spare_.push_back(std::move(pieces_.front()));
            pieces_.pop_front();
        }
        if (failed_) return -1;
        *end_bit = P_;
        *stream_end = ended_;
        memcpy(window_after, window_, WSIZE);
        return (long)out_total_;
    }

    long read(uint8_t *dst, size_t cap) {
        if (failed_) return -1;
        size_t got = 0;
        while (got < cap) {
            if (pieces_.empty()) {
                if (ended_) break;
                if (!stitch_next(true)) {
                    if (failed_) return -1;
                    continue;
                }
                continue;
            }
            while (!ended_ && pieces_.size() < max_ahead_ && stitch_next(false)) {
            }
            if (failed_) return -1;
            Chunk &c = *pieces_.front();
            const auto w0 = std::chrono::steady_clock::now();
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_done_.wait(lk, [&] { return c.resolved; });
            }
            if (!c.verified) {
                if (!verify(c)) return -1;
                c.verified = true;
            }
            const auto w1 = std::chrono::steady_clock::now();
            const size_t total = c.dec.total();
            const size_t n = std::min(cap - got, total - c.read_off);
            for_ranges(c, c.read_off, c.read_off + n, [&](const uint8_t *p, size_t k) {
                memcpy(dst + got, p, k);
                got += k;
            });
            const auto w2 = std::chrono::steady_clock::now();
            t_wait_ns_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(w1 - w0).count();
            t_copy_ns_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(w2 - w1).count();
            c.read_off += n;
            if (c.read_off == total) {
                spare_.push_back(std::move(pieces_.front()));
                pieces_.pop_front();
            }
        }
        return (long)got;
    }

    const std::string &error() const { return error_; }
    void stats(uint64_t *a, uint64_t *r, uint64_t *g) const {
        if (a) *a = accepted_;
        if (r) *r = rejected_;
        if (g) *g = gap_bytes_;
    }

private:
    struct Job {
        Chunk *c;
        int kind;  // 0 speculative decode, 1 resolve
    };
    // what a successor needs to continue from a finished chunk whose own start was certain
    struct TailInfo {
        uint64_t end_bit = 0;
        bool stream_end = false, failed = false;
        uint8_t window[WSIZE];
    };

    void worker() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return quit_ || !jobs_.empty() || !resolve_jobs_.empty(); });
                if (quit_) return;
                std::deque<Job> &q = resolve_jobs_.empty() ? jobs_ : resolve_jobs_;  // resolving unblocks output
                j = q.front();
                q.pop_front();
            }
            const auto j0 = std::chrono::steady_clock::now();
            if (j.kind == 0)
                speculate(*j.c);
            else {
                resolve_chunk(*j.c);
                if (range_dst_) copy_out(*j.c);
            }
            (j.kind == 0 ? t_spec_ns_ : t_resolve_ns_) +=
                (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - j0).count();
            std::shared_ptr<TailInfo> tail;
            if (j.kind == 0 && j.c->chained && j.c->dec.stop != STOP_ERROR) {
                tail = make_tail(*j.c);
            } else if (j.kind == 0 && force_chain_) {  // successors wait for a tail: tell them there is none
                tail.reset(new TailInfo());
                tail->failed = true;
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (j.kind == 0) {
                    j.c->spec_done = true;
                    if (tail) {
                        tails_[j.c->index] = tail;
                        while (!tails_.empty() && tails_.begin()->first + 4 * max_ahead_ < j.c->index)
                            tails_.erase(tails_.begin());
                    }
                } else {
                    j.c->resolved = true;
                }
            }
            cv_done_.notify_all();
        }
    }

    uint64_t chunk_bit(size_t c) const {  // nominal first bit of chunk c; beyond the file: "never"
        // (range mode: the chunks tile [lo, hi); the last one ends at the first seam at or behind hi, a chunk's length further at most)
        if (range_) return c < n_chunks_ ? (lo_ + (uint64_t)c * chunk_) * 8 : (hi_ + (uint64_t)(c - n_chunks_) * chunk_) * 8;
        return c >= n_chunks_ ? (uint64_t)-1 : (uint64_t)c * chunk_ * 8;
    }
    void copy_out(Chunk &c) {  // range mode: the chunk's (resolved) text to its place in the caller's buffer
        size_t at = c.out_off;
        for_ranges(c, 0, c.dec.total(), [&](const uint8_t *p, size_t k) {
            memcpy(range_dst_ + at, p, k);
            at += k;
        });
    }

    // last WSIZE bytes of the stream after a chunk whose start is certain (c.window_before holds the
    // window it started with: markers of its 16-bit part can be replaced)
    std::shared_ptr<TailInfo> make_tail(Chunk &c) {
        std::shared_ptr<TailInfo> t(new TailInfo());
        t->end_bit = c.dec.end_bit;
        t->stream_end = c.dec.stop == STOP_STREAM_END;
        const size_t n16 = c.dec.n16, n8 = c.dec.n8, total = n16 + n8;
        const uint16_t *s = c.dec.out16.data();
        const uint8_t *w = c.window_before;
        size_t k = 0;
        if (total < WSIZE) {
            memcpy(t->window, w + total, WSIZE - total);
            k = WSIZE - total;
        }
        for (size_t pos = total > WSIZE ? total - WSIZE : 0; pos < n16; pos++, k++)
            t->window[k] = s[pos] & 0x8000u ? w[s[pos] & 0x7fffu] : (uint8_t)s[pos];
        if (k < WSIZE) memcpy(t->window + k, c.dec.out8.data() + n8 - (WSIZE - k), WSIZE - k);
        return t;
    }

    void speculate(Chunk &c) {
        const uint8_t *end = base_ + size_;
        c.dec.reset(base_, end);
        c.found = false;
        c.chained = false;
        if (c.index == 0 && !range_) {
            memset(c.window_before, 0, WSIZE);
            c.dec.start_bytes(nullptr, 0);
            c.dec.run(first_block_bit_, chunk_bit(1), chunk_bit(2), false);
            c.found = true;  // errors of the first chunk are real errors
            c.chained = true;
            return;
        }
        {  // the predecessor is finished and certain: no need to guess
            std::shared_ptr<TailInfo> t;
            {
                std::unique_lock<std::mutex> lk(mu_);
                // few workers: guessing costs more than it gains, decode strictly in order
                if (force_chain_) cv_done_.wait(lk, [&] { return quit_ || tails_.count(c.index - 1); });
                auto it = tails_.find(c.index - 1);
                if (it != tails_.end()) t = it->second;
            }
            if (force_chain_ && !t) return;
            if (t) {
                if (t->stream_end || t->failed) {  // nothing left for this chunk
                    c.dec.stop = STOP_ERROR;  // (with few workers the epilogue passes the word on)
                    return;
                }
                memcpy(c.window_before, t->window, WSIZE);
                c.dec.start_bytes(t->window, WSIZE);
                c.dec.run(t->end_bit, chunk_bit(c.index + 1), chunk_bit(c.index + 2), false);
                c.found = c.dec.stop != STOP_ERROR;
                c.chained = c.found;
                return;
            }
        }
        uint64_t to = chunk_bit(c.index + 1);
        if (to > (uint64_t)size_ * 8) to = (uint64_t)size_ * 8;
        static const bool debug = getenv("NOHUMAN_GZ_DEBUG") != nullptr;
        timespec t0, t1, t2;
        if (debug) clock_gettime(CLOCK_MONOTONIC, &t0);
        // A position that passes the header test by chance decodes into an error sooner or later:
        // go on searching behind it (a few times; the consumer fills in whatever stays undecoded).
        uint64_t s = (uint64_t)-1, from = chunk_bit(c.index);
        bool pred_known = false;  // the predecessor finished meanwhile and this chunk did not start at its end
        for (int attempt = 0; attempt < 8 && !c.found && !pred_known; attempt++) {
            s = find_block(base_, end, from, to, c.dec.scratch);
            if (s == (uint64_t)-1) break;
            if (attempt) c.dec.reset(base_, end);
            // As soon as the predecessor is finished and certain, this chunk either starts exactly at
            // its end -- then it is certain too, takes over the predecessor's last 32 KiB and goes on
            // with the (three times faster) byte decoder -- or it started at a false positive.
            c.dec.at_block = [this, &c, &pred_known](ChunkDecoder &d) {
                std::shared_ptr<TailInfo> t;
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    auto it = tails_.find(c.index - 1);
                    if (it == tails_.end()) return true;
                    t = it->second;
                }
                pred_known = true;
                if (t->failed || t->stream_end || t->end_bit != d.start_bit) return false;
                memcpy(c.window_before, t->window, WSIZE);
                if (!d.adopt_window(t->window)) return false;
                c.chained = true;
                return true;
            };
            c.dec.run(s, chunk_bit(c.index + 1), chunk_bit(c.index + 2), false);
            c.found = c.dec.stop != STOP_ERROR;
            if (!c.found) c.chained = false;
            from = s + 1;
        }
        c.dec.at_block = nullptr;
        if (!c.found && pred_known) {  // start over from the predecessor's end, which is known now
            speculate(c);
            return;
        }
        if (debug) clock_gettime(CLOCK_MONOTONIC, &t1);
        if (s == (uint64_t)-1) return;
        if (debug) {
            clock_gettime(CLOCK_MONOTONIC, &t2);
            fprintf(stderr, "[gz] chunk %zu: search+decode %.1f ms (start +%llu bits), %.1f ms, n16 %zu n8 %zu stop %d\n", c.index,
                    (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6,
                    (unsigned long long)(s - chunk_bit(c.index)),
                    (t2.tv_sec - t1.tv_sec) * 1e3 + (t2.tv_nsec - t1.tv_nsec) * 1e-6, c.dec.n16, c.dec.n8, (int)c.dec.stop);
        }
    }

    std::unique_ptr<Chunk> new_chunk() {
        if (!spare_.empty()) {
            std::unique_ptr<Chunk> c = std::move(spare_.back());
            spare_.pop_back();
            return c;
        }
        return std::unique_ptr<Chunk>(new Chunk());
    }

    void dispatch() {
        while (next_dispatch_ < n_chunks_ && inflight_.size() + pieces_.size() < max_ahead_) {
            std::unique_ptr<Chunk> c = new_chunk();
            c->index = next_dispatch_++;
            c->spec_done = c->resolved = c->verified = false;
            c->read_off = 0;
            Chunk *raw = c.get();
            inflight_.push_back(std::move(c));
            {
                std::lock_guard<std::mutex> lk(mu_);
                jobs_.push_back({raw, 0});
            }
            cv_work_.notify_one();
        }
    }

    bool fail(const std::string &msg) {
        error_ = msg;
        failed_ = true;
        return false;
    }

    // window after a piece of output, given the window before it
    void advance_window(Chunk &c) {
        const size_t total = c.dec.total();
        uint8_t nw[WSIZE];
        if (total >= WSIZE) {
            size_t pos = total - WSIZE, k = 0;
            const size_t n16 = c.dec.n16;
            const uint16_t *s = c.dec.out16.data();
            for (; pos < n16 && k < WSIZE; pos++, k++) {
                const uint16_t v = s[pos];
                nw[k] = v & 0x8000u ? window_[v & 0x7fffu] : (uint8_t)v;
            }
            if (k < WSIZE) memcpy(nw + k, c.dec.out8.data() + (pos - n16), WSIZE - k);
        } else {
            memcpy(nw, window_ + total, WSIZE - total);
            size_t k = WSIZE - total;
            const size_t n16 = c.dec.n16;
            const uint16_t *s = c.dec.out16.data();
            for (size_t pos = 0; pos < n16; pos++, k++) {
                const uint16_t v = s[pos];
                nw[k] = v & 0x8000u ? window_[v & 0x7fffu] : (uint8_t)v;
            }
            memcpy(nw + k, c.dec.out8.data(), c.dec.n8);
        }
        memcpy(window_, nw, WSIZE);
    }

    bool place(Chunk &c) {  // range mode: the chunk's text follows what has been accepted so far
        if (!range_dst_) return true;
        c.out_off = out_total_;
        out_total_ += c.dec.total();
        if (out_total_ > range_cap_) return fail("gzip: the range's text does not fit the buffer");
        return true;
    }
    void accept(std::unique_ptr<Chunk> c) {
        if (!place(*c)) return;
        memcpy(c->window_before, window_, WSIZE);
        advance_window(*c);
        P_ = c->dec.end_bit;
        if (c->dec.stop == STOP_STREAM_END) ended_ = true;
        Chunk *raw = c.get();
        std::shared_ptr<TailInfo> t;
        if (!raw->chained) {  // accepted: its end and the window after it are certain now
            t.reset(new TailInfo());
            t->end_bit = P_;
            t->stream_end = ended_;
            memcpy(t->window, window_, WSIZE);
        }
        pieces_.push_back(std::move(c));
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (t && !tails_.count(raw->index)) tails_[raw->index] = t;
            resolve_jobs_.push_back({raw, 1});
        }
        if (t) cv_done_.notify_all();
        cv_work_.notify_one();
    }

    // the consumer decodes [P_, target) itself, in pieces of about one chunk
    bool gap_decode(uint64_t target) {
        std::unique_ptr<Chunk> g = new_chunk();
        g->index = (size_t)-1;
        g->spec_done = true;
        g->resolved = g->verified = false;
        g->read_off = 0;
        g->dec.reset(base_, base_ + size_);
        g->dec.start_bytes(window_, WSIZE);
        uint64_t soft = P_ + (uint64_t)chunk_ * 8;
        if (soft > target) soft = target;
        g->dec.run(P_, soft, (uint64_t)-1, true);
        if (g->dec.stop == STOP_ERROR) return fail(g->dec.error);
        if (!place(*g)) return false;
        gap_bytes_ += g->dec.total();
        memcpy(g->window_before, window_, WSIZE);
        advance_window(*g);
        P_ = g->dec.end_bit;
        if (g->dec.stop == STOP_STREAM_END) ended_ = true;
        resolve_chunk(*g);
        if (range_dst_) copy_out(*g);
        g->resolved = true;
        pieces_.push_back(std::move(g));
        return true;
    }

    // Stitches the next chunk (or decodes a gap).  block = wait for the chunk's speculative decode.
    // Returns false if nothing could be done (not ready / failed / ended).
    bool stitch_next(bool block) {
        if (ended_ || failed_) return false;
        dispatch();
        if (inflight_.empty()) {
            // no chunk left but the stream goes on (no block start was found in the rest)
            if (range_) return P_ < hi_ * 8 && block ? gap_decode(hi_ * 8) : false;
            return block ? gap_decode((uint64_t)-1) : false;
        }
        Chunk &c = *inflight_.front();
        {
            std::unique_lock<std::mutex> lk(mu_);
            if (!c.spec_done) {
                if (!block) return false;
                cv_done_.wait(lk, [&] { return c.spec_done; });
            }
        }
        if (c.index == 0 && !range_) {
            if (c.dec.stop == STOP_ERROR) return fail(c.dec.error);
            std::unique_ptr<Chunk> own = std::move(inflight_.front());
            inflight_.pop_front();
            accepted_++;
            accept(std::move(own));
            return true;
        }
        if (!c.found || c.dec.start_bit < P_) {  // no block start, garbage, or a start we already passed
            rejected_++;
            const size_t idx = c.index;
            spare_.push_back(std::move(inflight_.front()));
            inflight_.pop_front();
            publish_position(idx);
            return true;
        }
        if (c.dec.start_bit > P_) {
            if (!block) return false;
            const size_t idx = c.index;
            const bool ok = gap_decode(c.dec.start_bit);
            if (ok && idx > 0) publish_position(idx - 1);
            return ok;
        }
        std::unique_ptr<Chunk> own = std::move(inflight_.front());
        inflight_.pop_front();
        accepted_++;
        accept(std::move(own));
        return true;
    }

    // The consumer's own position stands in for the tail of chunk idx (rejected, or followed by a gap the
    // consumer decoded): chunk idx + 1, if still being decoded, can adopt the window or give up.
    void publish_position(size_t idx) {
        if (ended_ || P_ < chunk_bit(idx + 1)) return;
        std::shared_ptr<TailInfo> t(new TailInfo());
        t->end_bit = P_;
        memcpy(t->window, window_, WSIZE);
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!tails_.count(idx)) tails_[idx] = t;
        }
        cv_done_.notify_all();
    }

    bool verify(Chunk &c) {
        for (const Seg &s : c.segs) {
            cur_crc_ = (uint32_t)crc32_combine(cur_crc_, s.crc, (z_off_t)s.len);
            cur_len_ += s.len;
            if (s.member_end) {
                if (cur_crc_ != s.want_crc) return fail("gzip: crc error");
                if ((uint32_t)cur_len_ != s.want_isize) return fail("gzip: length error");
                cur_crc_ = 0;
                cur_len_ = 0;
            }
        }
        return true;
    }

    int fd_ = -1;
    const uint8_t *base_ = nullptr;
    bool range_ = false;  // range mode: [lo_, hi_) of a mapping that belongs to the caller
    uint64_t lo_ = 0, hi_ = 0;
    uint8_t *range_dst_ = nullptr;
    size_t range_cap_ = 0, out_total_ = 0;
    size_t size_ = 0, chunk_ = 0, n_chunks_ = 0, max_ahead_ = 0;
    unsigned threads_ = 1;
    bool force_chain_ = false;
    uint64_t first_block_bit_ = 0;
    // consumer state
    uint64_t P_ = 0;
    uint8_t window_[WSIZE];
    uint32_t cur_crc_ = 0;
    uint64_t cur_len_ = 0;
    size_t next_dispatch_ = 0, next_stitch_ = 0;
    bool ended_ = false, failed_ = false;
    std::string error_;
    std::deque<std::unique_ptr<Chunk>> inflight_, pieces_;
    std::vector<std::unique_ptr<Chunk>> spare_;
    uint64_t accepted_ = 0, rejected_ = 0, gap_bytes_ = 0;
    // shared with the workers
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    std::deque<Job> jobs_, resolve_jobs_;
    std::atomic<uint64_t> t_spec_ns_{0}, t_resolve_ns_{0};  // the workers' time decoding / replacing markers + CRC (NOHUMAN_TRACE)
    uint64_t t_wait_ns_ = 0, t_copy_ns_ = 0;  // the consumer's time waiting for the head chunk / copying it out (NOHUMAN_TRACE)
    std::map<size_t, std::shared_ptr<TailInfo>> tails_;
    bool quit_ = false;
    std::vector<std::thread> workers_;
};

RangeGunzip::RangeGunzip() : impl_(new GunzipImpl()) {}
RangeGunzip::~RangeGunzip() { delete impl_; }
int RangeGunzip::start(const uint8_t *base, size_t size, uint64_t lo_byte, uint64_t hi_byte, unsigned threads, size_t chunk_bytes) {
    return impl_->open_range(base, size, lo_byte, hi_byte, threads, chunk_bytes);
}
void RangeGunzip::wait_speculated() { impl_->wait_speculated(); }
long RangeGunzip::finish(uint64_t from_bit, const uint8_t *window, uint8_t *dst, size_t cap, uint64_t *end_bit, bool *stream_end,
                         uint8_t *window_after, std::vector<GzSeg> &segs) {
    return impl_->finish_range(from_bit, window, dst, cap, end_bit, stream_end, window_after, segs);
}
const std::string &RangeGunzip::error() const { return impl_->error(); }
void RangeGunzip::stats(uint64_t *a, uint64_t *r, uint64_t *g) const { impl_->stats(a, r, g); }
void RangeGunzip::close() { impl_->close(); }

ParallelGunzip::ParallelGunzip() : impl_(new GunzipImpl()) {}
ParallelGunzip::~ParallelGunzip() { delete impl_; }
const uint8_t *gzip_member_body(const uint8_t *p, const uint8_t *end, bool *truncated) {
    const uint8_t *d = parse_gzip_header(p, end);
    if (truncated) *truncated = d == TRUNCATED;
    return d == TRUNCATED ? nullptr : d;
}

int inflate_from(const uint8_t *base, const uint8_t *end, uint64_t from_bit, uint64_t stop_bit, const uint8_t *window,
                 size_t window_len, std::vector<uint8_t> &out, std::vector<GzMemberEnd> &members, uint64_t *end_bit,
                 bool *stream_end, std::string &err, size_t out_limit) {
    std::unique_ptr<ChunkDecoder> d(new ChunkDecoder());
    d->reset(base, end);
    d->out_limit = out_limit;
    if (window_len > WSIZE) {
        window += window_len - WSIZE;
        window_len = WSIZE;
    }
    d->start_bytes(window, window_len);
    d->run(from_bit, stop_bit, stop_bit, true);
    if (d->stop == STOP_ERROR || d->stop == STOP_NONE) {
        err = std::string("gzip: ") + (d->error.empty() ? "invalid deflate data" : d->error);
        return -1;
    }
    out.assign(d->out8.data(), d->out8.data() + d->n8);
    members.clear();
    for (const MemberEnd &m : d->members) members.push_back({m.out_pos, m.crc, m.isize});
    *end_bit = d->end_bit;
    *stream_end = d->stop == STOP_STREAM_END;
    return 0;
}

uint32_t crc32_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    return (uint32_t)crc32_combine(crc_a, crc_b, (z_off_t)len_b);
}

int ParallelGunzip::open(const char *path, unsigned threads, size_t chunk_bytes, std::string &err) {
    return impl_->open(path, threads, chunk_bytes, err);
}
long ParallelGunzip::read(uint8_t *dst, size_t cap) { return impl_->read(dst, cap); }
const std::string &ParallelGunzip::error() const { return impl_->error(); }
void ParallelGunzip::close() { impl_->close(); }
void ParallelGunzip::stats(uint64_t *a, uint64_t *r, uint64_t *g) const { impl_->stats(a, r, g); }

}  // namespace nh
