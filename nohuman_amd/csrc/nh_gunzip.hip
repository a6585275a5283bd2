// nh_gunzip.hip -- the gzip reader on the GPU: block search, decode, window scan, marker resolve, CRC-32 (nh_gunzip.h).
//
// Replaces, for the inputs nohuman passes verbatim to the path (/root/reference/src/main.rs:267), the `gzip -dc` pipe of
// kraken2's wrapper.  DEFLATE / gzip as RFC 1951 / 1952 define them; the parallel scheme
// (speculative chunk starts, 16-bit symbols with markers for the unknown window) is the one of pugz / rapidgzip and of
// this repository's host reader (nh_inflate.cpp); the window chain as a prefix scan over index maps is this file's own.
// gfx950 only: one wavefront decodes a chunk (all decode state is wave-uniform, a match is copied by the 64 lanes).
#include "nh_gunzip.h"

#include <errno.h>
#include <fcntl.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <memory>
#include <mutex>
#include <thread>

#include "nh_inflate.h"
#include "nh_internal.h"
#include "nohuman_engine.h"

namespace nh {
namespace gz {

constexpr uint32_t WSIZE = 32768;
constexpr uint32_t MAX_MEMBERS = 4;  // member ends one chunk can record
constexpr uint64_t NONE = ~0ull;
constexpr uint32_t NOIDX = 0xFFFFFFFFu;

// status of a chunk
enum : uint32_t {
    ST_OK = 0,
    ST_STORED = 2,    // stored block: LEN / NLEN do not match
    ST_BTYPE = 3,     // block type 3
    ST_HEADER = 4,    // invalid code lengths set
    ST_ROOM = 5,      // more text than the chunk's slots hold (beyond 16 : 1)
    ST_LITLEN = 6,    // invalid literal / length code
    ST_INPUT = 7,     // ran past the bytes on the device (a block longer than the look-ahead, or a truncated file)
    ST_DIST = 8,      // invalid distance code
    ST_MEMBERS = 9,   // more member ends in one chunk than it can record
    ST_FAR = 10,      // distance beyond the window
    ST_GZHEAD = 11,   // a member header runs past the bytes on the device
};

struct ChunkDesc {  // one per stretch of the piece
    // plan (k_plan)
    uint64_t bit_start;  // NONE: no chunk starts in this stretch
    uint64_t stop_bit;   // decode to the first block boundary at or behind it
    uint32_t cap;        // symbols of room (the chunk's own slot and those of the empty stretches behind it)
    uint32_t pad0;
    // results (k_inflate)
    uint64_t bit_end;
    uint32_t out_len;    // symbols written
    uint32_t status;
    uint32_t n_members;  // member ends inside the chunk
    uint32_t flags;      // 1: the stream ended in this chunk
    uint32_t blocks, pad1;
    uint32_t m_off[MAX_MEMBERS], m_crc[MAX_MEMBERS], m_isize[MAX_MEMBERS];
    uint32_t piece_crc[MAX_MEMBERS + 1];  // k_crc: CRC-32 of the text between the chunk's start, its member ends, its end
    uint32_t pad2;
    // NOHUMAN_GZDEV_PROF (k_inflate3 built with NH_GZ_PROF): cycles (s_memtime) in header + tables, the lanes' decode, the
    // walk, the expansion, flushes, all; then windows, tokens, matches, generic copies, slow tokens, blocks
    uint64_t prof[12];
};

struct SegResult {  // k_finish
    uint64_t total;      // text bytes of the piece
    uint64_t end_bit;    // where the stream goes on
    uint32_t n_chunks;   // chunks that count (up to the one the stream ended in)
    uint32_t end_chunk;  // index of the last of them
    uint32_t broken;     // NOIDX, or the first chunk that does not start where its predecessor ended
    uint32_t bad_chunk;  // NOIDX, or the first chunk with an error status
    uint32_t bad_status;
    uint32_t stream_end;
    uint32_t members, pad;
    uint64_t first_start;  // bit position of the first chunk (a speculative piece: the first block start the search found)
};

__constant__ uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// one wave per workgroup: LDS operations of a wave execute in order, so lanes see each other's LDS writes without a
// barrier; what is needed is that the compiler keeps the order (and __syncthreads() would also wait for the global
// stores of the copy before it)
#define LDS_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)lane));
}
__device__ __forceinline__ uint32_t wsum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    return v;
}
__device__ __forceinline__ uint64_t bits_at(const uint8_t *bytes, uint64_t b) {  // >= 57 valid bits at bit position b
    struct __attribute__((packed)) U {
        uint64_t v;
    };
    return ((const U *)(bytes + (b >> 3)))->v >> (b & 7);
}

// a symbol whose code is longer than the root table: canonical walk, one bit at a time
__device__ __forceinline__ uint32_t slow_symbol(uint64_t w, const uint16_t *count, const uint16_t *symlist, uint32_t &len_out) {
    uint32_t code = 0, first = 0, index = 0;
    for (uint32_t l = 1; l <= 15; l++) {
        code |= (uint32_t)(w >> (l - 1)) & 1u;
        const uint32_t c = count[l];
        if (code < first + c) {
            len_out = l;
            return symlist[index + (code - first)];
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    len_out = 0;
    return 0xFFFFu;
}

// plan of the piece: chunk c runs from its start to the next start found (the last one to the first block boundary at or
// behind the end of the piece's stretches) and owns the symbol slots up to the next chunk's
__global__ __launch_bounds__(1024) void k_plan(const uint64_t *start, uint32_t n, uint64_t end_bit, uint32_t slot_syms, ChunkDesc *desc) {
    for (uint32_t c = threadIdx.x; c < n; c += 1024) {
        ChunkDesc &d = desc[c];
        // (a piece may end inside its last stretch -- one that is cut at a boundary of the piece grid: what the search found
        //  behind the end belongs to the next piece)
        const uint64_t s = c == 0 || start[c] < end_bit ? start[c] : NONE;
        d.bit_start = s;
        if (s == NONE) {
            d.stop_bit = 0;
            d.cap = 0;
            continue;
        }
        uint32_t nx = c + 1;
        while (nx < n && (start[nx] == NONE || start[nx] >= end_bit)) nx++;
        d.stop_bit = nx < n ? start[nx] : end_bit;
        const uint64_t cap = (uint64_t)(nx - c) * slot_syms;
        d.cap = cap > 0xFFFFF000ull ? 0xFFFFF000u : (uint32_t)cap;
    }
}

// =====================================================================================================================
// The two heavy kernels (round 4; the two versions before them are in the history of this file, what each step bought in
// profiles/r04_inflate_notes.txt).  The first cut spent ~1000 cycles a token in a chain of dependent latencies (bit fetch
// by three v_readlane, root table in LDS, distance table in LDS, the window read of the copy, a store instruction per
// literal, vmcnt(0) waits behind those stores), and the search nine tenths of its time in the wave-serial header parse of
// candidates that fail it.  Now: the bit buffer lives in scalars, tokens are decoded 64 bit positions at a time and
// expanded a window at a time, the ring goes to HBM 64 symbols an instruction; the search collects the candidates that
// pass the cheap test 64 at a time and decodes their headers IN PARALLEL, a lane each (own bit reader, own 7-bit table of
// the code-length code in LDS) -- only what survives that goes to the wave-wide parse and trial walk.
// =====================================================================================================================
struct BitRd {  // wave-uniform bit reader: 128 dwords of input in two registers a lane, 64 bits of them in scalars
    const uint32_t *in;
    uint32_t wbase;  // dword index of winA's lane 0
    uint32_t winA, winB;
    uint32_t dwi;    // next dword to take, relative to wbase (< 64 after rotate())
    uint64_t bb;
    uint32_t bc;
    __device__ __forceinline__ void init(const uint32_t *p, uint64_t bitpos, int lane) {
        in = p;
        wbase = (uint32_t)(bitpos >> 5) & ~63u;
        winA = in[wbase + lane];
        winB = in[wbase + 64 + lane];
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        dwi = (uint32_t)(bitpos >> 5) - wbase;
        bb = 0;
        bc = 0;
        ensure32();
        drop((uint32_t)bitpos & 31u);
    }
    __device__ __forceinline__ void rotate(int lane) {  // winA is used up
        winA = winB;
        wbase += 64u;
        dwi -= 64u;
        winB = in[wbase + 64 + lane];
        // waited for HERE, once per 256 bytes of input (left to the compiler, every later use would wait for vmcnt(0) too)
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __device__ __forceinline__ void ensure32(int lane = -1) {  // at least 32 valid bits
        if (bc < 32u) {
            if (dwi >= 64u) rotate(lane < 0 ? (int)threadIdx.x : lane);
            const uint32_t v = rl(winA, dwi);
            dwi++;
            bb |= (uint64_t)v << bc;
            bc += 32u;
        }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)bb & ((1u << n) - 1u); }  // n <= 31
    __device__ __forceinline__ void drop(uint32_t n) {
        bb >>= n;
        bc -= n;
    }
    __device__ __forceinline__ uint32_t take(uint32_t n) {  // n <= 31, after ensure32()
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    __device__ __forceinline__ uint64_t bitpos() const { return (uint64_t)(wbase + dwi) * 32u - bc; }
    __device__ __forceinline__ void align8() { drop(bc & 7u); }  // (dwords are whole bytes: the buffer's bit count mod 8 is the stream's)
};

// v = its old value, with lane `lane` (uniform) set to `val` (uniform).  (clang has no builtin for v_writelane_b32)
__device__ __forceinline__ uint32_t wlane(uint32_t old, uint32_t val, uint32_t lane) {
    // (one scalar operand per VALU instruction on gfx9: the lane select rides in M0)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(val), "s"(lane) : "m0");
    return old;
}
// ---- k_inflate3: the decode phase lane-parallel, the LDS and register diet for occupancy -------------------------------
// Measured on the version before and on the first cut of this kernel (NH_GZ_PROF, profiles/r04_inflate_notes.txt): a wave alone
// on a chunk runs its uniform control flow at 10-20 cycles an instruction -- nothing in flight but its own dependent
// chain --, 1300 cycles a match in an expansion loop with an integer modulo, 400 a token in the walk; 16 KB of LDS and
// 111 registers let nine waves a CU hide that.  So: (1) the 64 lanes decode the tokens that WOULD start at the next 64
// bit positions (16-bit root tables in LDS: two gathers for all of them), the real chain is walked with one v_readlane
// a token; (2) every match of up to 64 symbols -- overlapping ones too, the run of a quality string -- is ONE read and
// one write of the ring (lane % distance by a reciprocal held in a register); (3) ring of 2048 symbols, 16-bit tables,
// build scratch shared with the input ring: 7.9 KB of LDS a wave, twenty waves a CU.
constexpr uint32_t NEAR3 = 2048, M3 = NEAR3 - 1;
constexpr uint32_t SPAN3 = 232, AHEAD3 = SPAN3 + 257, FLUSH3 = 256;
constexpr uint32_t RING_DMAX3 = NEAR3 - AHEAD3 - 64;
static_assert(RING_DMAX3 >= 2 * AHEAD3 + FLUSH3 + 258, "a source older than the ring must already be flushed");
constexpr int ROOT3 = 9, DROOT3 = 8;

struct Lds3 {
    uint16_t ring[NEAR3];
    // root tables, 16 bits an entry.  literals / lengths: bits 0-3 code length (0: longer than the root), 4-5 kind (0 literal,
    // 1 length, 2 end of block), 6-13 the literal, or length symbol - 257 (5 bits) and its extra bits (3); distances: bits
    // 0-3 code length, 4-8 distance symbol, 9-12 its extra bits
    uint16_t lit[1 << ROOT3];
    uint16_t dist[1 << DROOT3];
    uint16_t clt[128];  // the code-length code: bits 0-3 length, 8-12 symbol
    uint32_t lb[32], db[32];  // base | extra bits << 16 of the length / distance symbols
    uint16_t lsym[288], dsym[32];
    uint16_t lcount[16], dcount[16];
    uint8_t lens[320];
    uint16_t tmp[40];
    union {
        uint32_t inbuf[128];  // 512 bytes of the input as a ring of dwords (dword d of the buffer at d & 127): live inside a block
        uint16_t code[320];   // build(): the symbols' codes, live between the blocks
    };
};

__device__ __forceinline__ uint32_t lit_entry3(uint32_t s, uint32_t l) {
    if (s < 256u) return l | (s << 6);
    if (s == 256u) return l | (2u << 4);
    if (s >= 286u) return 0u;  // (the two length symbols a fixed block has codes for but no meaning: the canonical walk reports them)
    const uint32_t i = s - 257u;
    return l | (1u << 4) | ((i | ((uint32_t)LEXT[i] << 5)) << 6);
}
__device__ __forceinline__ uint32_t dist_entry3(uint32_t s, uint32_t l) { return s >= 30u ? 0u : (l | (s << 4) | ((uint32_t)DEXT[s] << 9)); }

// build() for Lds3: counts, canonical symbol list, codes, 16-bit root table; mode 0 the code-length code, 1 literals /
// lengths, 2 distances.  False: over-subscribed.
__device__ __forceinline__ bool build3(Lds3 &S, const uint8_t *lens, int n, uint16_t *count, uint16_t *symlist, uint16_t *root, int rootbits,
                                       int mode, int lane) {
    LDS_ORDER();
    uint32_t over = 0;
    if (lane == 0) {
        uint16_t *offs = S.tmp, *next = S.tmp + 16;
        for (int l = 0; l < 16; l++) count[l] = 0;
        for (int s = 0; s < n; s++) count[lens[s]]++;
        count[0] = 0;
        uint32_t o = 0, c = 0;
        int left = 1;
        for (int l = 1; l < 16; l++) {
            left = (left << 1) - (int)count[l];
            if (left < 0) over = 1;
            offs[l] = (uint16_t)o;
            o += count[l];
            c = (c + count[l - 1]) << 1;
            next[l] = (uint16_t)c;
        }
        if (!over)
            for (int s = 0; s < n; s++) {
                const int l = lens[s];
                if (l) {
                    symlist[offs[l]++] = (uint16_t)s;
                    S.code[s] = next[l]++;
                }
            }
    }
    if (uni(over)) return false;
    for (int k = lane; k < (1 << rootbits); k += 64) root[k] = 0;
    LDS_ORDER();
    for (int s = lane; s < n; s += 64) {
        const uint32_t l = lens[s];
        if (l && l <= (uint32_t)rootbits) {
            const uint32_t r = __builtin_bitreverse32((uint32_t)S.code[s]) >> (32 - l);
            const uint32_t e = mode == 0 ? (l | ((uint32_t)s << 8)) : mode == 1 ? lit_entry3((uint32_t)s, l) : dist_entry3((uint32_t)s, l);
            for (uint32_t k = r; k < (1u << rootbits); k += 1u << l) root[k] = (uint16_t)e;
        }
    }
    LDS_ORDER();
    return true;
}

__device__ __forceinline__ uint32_t parse_dynamic3(Lds3 &S, BitRd &br, int lane, int &nlit, int &ndist) {
    br.ensure32(lane);
    nlit = (int)br.take(5) + 257;
    ndist = (int)br.take(5) + 1;
    const int ncl = (int)br.take(4) + 4;
    if (nlit > 286 || ndist > 30) return ST_HEADER;
    LDS_ORDER();
    if (lane < 19) S.lens[lane] = 0;
    LDS_ORDER();
    for (int i = 0; i < ncl; i++) {
        br.ensure32(lane);
        const uint32_t v = br.take(3);
        if (lane == 0) S.lens[CLORDER[i]] = (uint8_t)v;
    }
    if (!build3(S, S.lens, 19, S.lcount, S.lsym, S.clt, 7, 0, lane)) return ST_HEADER;
    const uint32_t c0 = S.clt[lane], c1 = S.clt[64 + lane];  // the 7-bit table of the code-length code: two registers
    int i = 0;
    uint32_t prev = 0;
    while (i < nlit + ndist) {
        br.ensure32(lane);
        const uint32_t ix = br.peek(7);
        const uint32_t e = ix < 64u ? rl(c0, ix) : rl(c1, ix - 64u);
        const uint32_t l = e & 15u, sym = e >> 8;
        if (l == 0) return ST_HEADER;
        br.drop(l);
        uint32_t rep = 1, val = sym;
        if (sym == 16) {
            if (i == 0) return ST_HEADER;
            rep = 3 + br.take(2);
            val = prev;
        } else if (sym == 17) {
            rep = 3 + br.take(3);
            val = 0;
        } else if (sym == 18) {
            rep = 11 + br.take(7);
            val = 0;
        }
        if (i + (int)rep > nlit + ndist) return ST_HEADER;
        for (uint32_t k = (uint32_t)lane; k < rep; k += 64) S.lens[i + k] = (uint8_t)val;
        i += (int)rep;
        prev = val;
    }
    LDS_ORDER();
    if (S.lens[256] == 0) return ST_HEADER;  // no end-of-block code
    return 0;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_inflate3(const uint32_t *in, uint64_t valid_bits,
                                                                                              uint32_t at_eof, ChunkDesc *desc, uint16_t *sym,
                                                                                              uint32_t slot_syms, uint32_t only) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Lds3 &S = *(Lds3 *)smem;
    const int lane = (int)threadIdx.x;
    const uint32_t c = only != NOIDX ? only : blockIdx.x;
    ChunkDesc &cd = desc[c];
    if (cd.bit_start == NONE) {
        if (lane == 0) {
            cd.bit_end = 0;
            cd.out_len = 0;
            cd.status = 0;
            cd.n_members = 0;
            cd.flags = 0;
            cd.blocks = 0;
        }
        return;
    }
    // the unknown 32 KiB before the chunk as markers: the newest NEAR3 of them in the ring (position q of the stream,
    // counted from the chunk's first symbol, lives at q & M3; position -k holds marker 0x8000 | (32768 - k))
    for (uint32_t i = (uint32_t)lane; i < NEAR3; i += 64) S.ring[i] = (uint16_t)(0x8000u | (WSIZE - NEAR3 + i));
    if (lane < 29) S.lb[lane] = (uint32_t)LBASE[lane] | ((uint32_t)LEXT[lane] << 16);
    if (lane < 30) S.db[lane] = (uint32_t)DBASE[lane] | ((uint32_t)DEXT[lane] << 16);
    const uint32_t inv = lane ? (65536u + (uint32_t)lane - 1u) / (uint32_t)lane : 0u;  // lane d: ceil(65536 / d): x % d for x < 64 without a divide
    LDS_ORDER();
    const uint64_t stop_bit = cd.stop_bit;
    uint16_t *o = sym + (uint64_t)c * slot_syms;
    uint32_t op = 0, flushed = 0;  // symbols decoded; symbols in HBM
    const uint32_t cap = cd.cap;
    const uint8_t *bytes = (const uint8_t *)in;
    BitRd br;
    br.init(in, cd.bit_start, lane);
    uint32_t status = 0, blocks = 0, n_members = 0, flags = 0;
    uint32_t floor_op = 0;
    bool fresh_member = false;
    uint32_t tok = 0, tpos = 0;  // this lane's token of the window: literal 0x80000000 | byte, match distance | symbols << 16; offset in the window's output
#ifdef NH_GZ_PROF
    uint64_t pf[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tprev = __builtin_amdgcn_s_memtime();
    const uint64_t tstart = tprev;
#define GZ_STAMP(i)                                          \
    do {                                                     \
        const uint64_t t_ = __builtin_amdgcn_s_memtime();    \
        pf[i] += t_ - tprev;                                 \
        tprev = t_;                                          \
    } while (0)
#define GZ_COUNT(i, n) pf[i] += (n)
#else
#define GZ_STAMP(i)
#define GZ_COUNT(i, n)
#endif
    auto flush_to = [&](uint32_t target) {  // ring -> HBM, 64 symbols an instruction
        for (uint32_t p = flushed + (uint32_t)lane; p < target; p += 64) o[p] = S.ring[p & M3];
        flushed = target;
    };
    // the fields of the token whose literal / length code has entry e (this lane's 64 bits of the stream in w):
    // false when its distance code is longer than the root (de == 0)
    auto fields = [&](uint32_t e, uint32_t de_in, bool have_de, uint64_t w, uint32_t &tok_o, uint32_t &info_o) -> bool {
        const uint32_t l = e & 15u, kind = (e >> 4) & 3u, pay = (e >> 6) & 0xFFu;
        const bool ism = kind == 1u;
        const uint32_t lext = ism ? pay >> 5 : 0u;
        const uint32_t w1 = (uint32_t)(w >> l);
        const uint32_t de = have_de ? de_in : S.dist[(w1 >> lext) & ((1u << DROOT3) - 1u)];
        const uint32_t lbv = S.lb[ism ? (pay & 31u) : 0u];
        const uint32_t dl = de & 15u, dsy = (de >> 4) & 31u, dext = (de >> 9) & 15u;
        const uint32_t dbv = S.db[dsy < 30u ? dsy : 0u];
        const uint32_t len = (lbv & 0xFFFFu) + (w1 & ((1u << lext) - 1u));
        const uint32_t dist = (dbv & 0xFFFFu) + ((uint32_t)(w >> (l + lext + dl)) & ((1u << dext) - 1u));
        tok_o = ism ? (dist | (len << 16)) : (0x80000000u | pay);
        info_o = (l + (ism ? lext + dl + dext : 0u)) | (kind << 6) | ((kind == 2u ? 0u : (ism ? len : 1u)) << 9);
        return !(ism && de == 0u);
    };
    while (status == 0) {
        uint64_t bitpos = br.bitpos();
        // A block boundary at or behind the stop position -- and what follows is a block the SEARCH can find (not final, dynamic
        // Huffman): the empty stored block of a flush point (pigz, gzp: one every 128-512 KiB of text), a fixed or a final
        // block are decoded with this chunk.  So a piece ends exactly where the piece decoded ahead of the stream for the
        // next cell of the grid begins (every eighth was refused on gzp's output before this).
        // (at_eof bit 1: BGZF -- every member is one final block and the chunks' starts are the members' own, so any boundary ends the chunk)
        if (bitpos >= stop_bit && ((at_eof & 2u) || bitpos + 3 > valid_bits || (bits_at(bytes, bitpos) & 7u) == 4u)) break;
        if (bitpos + 3 > valid_bits) {
            status = ST_INPUT;
            break;
        }
        br.ensure32(lane);
        const uint32_t bfinal = br.take(1), btype = br.take(2);
        blocks++;
        if (btype == 0) {  // stored: the bytes go straight from the input to the output (and the ring)
            br.align8();
            br.ensure32(lane);
            bitpos = br.bitpos();
            if (bitpos + 32 > valid_bits) {
                status = ST_INPUT;
                break;
            }
            const uint32_t len = br.take(16);
            br.ensure32(lane);
            const uint32_t nlen = br.take(16);
            if ((len ^ nlen) != 0xFFFFu) {
                status = ST_STORED;
                break;
            }
            bitpos += 32;
            if (bitpos + 8ull * len > valid_bits) {
                status = ST_INPUT;
                break;
            }
            if ((uint64_t)op + len > cap) {
                status = ST_ROOM;
                break;
            }
            flush_to(op);
            const uint8_t *src = bytes + (bitpos >> 3);
            asm volatile("" ::: "memory");
            for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                const uint16_t v = src[i];
                S.ring[(op + i) & M3] = v;
                o[op + i] = v;
            }
            LDS_ORDER();
            op += len;
            flushed = op;
            if (len) br.init(in, bitpos + 8ull * len, lane);
        } else if (btype == 3) {
            status = ST_BTYPE;
            break;
        } else {
            int nlit, ndist;
            if (btype == 1) {  // fixed codes
                LDS_ORDER();
                for (int s = lane; s < 288; s += 64) S.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
                if (lane < 32) S.lens[288 + lane] = 5;
                nlit = 288;
                ndist = 30;
            } else {
                status = parse_dynamic3(S, br, lane, nlit, ndist);
                if (status) break;
            }
            if (!build3(S, S.lens, nlit, S.lcount, S.lsym, S.lit, ROOT3, 1, lane) ||
                !build3(S, S.lens + nlit, ndist, S.dcount, S.dsym, S.dist, DROOT3, 2, lane)) {
                status = ST_HEADER;
                break;
            }
            GZ_STAMP(0);
            GZ_COUNT(11, 1);
            // ---- the symbols of the block, a window of 64 bit positions at a time
            bool eob = false;
            uint64_t P = br.bitpos();                      // bit position of the next token
            uint32_t in_upto = (uint32_t)(P >> 5) & ~63u;  // dwords of the input below this index are in the LDS ring (the newest 128)
            LDS_ORDER();
            while (!eob && status == 0) {
                {
                    const uint32_t need = (uint32_t)(P >> 5) + 6u;
                    while (in_upto < need) {
                        const uint32_t v = in[in_upto + (uint32_t)lane];
                        S.inbuf[(in_upto + (uint32_t)lane) & 127u] = v;
                        in_upto += 64u;
                    }
                    LDS_ORDER();
                }
                // phase 1: the token that would start at bit P + lane
                const uint32_t bp = ((uint32_t)P & 31u) + (uint32_t)lane;
                const uint32_t dwi = (uint32_t)(P >> 5) + (bp >> 5);
                const uint32_t d0 = S.inbuf[dwi & 127u], d1 = S.inbuf[(dwi + 1u) & 127u], d2 = S.inbuf[(dwi + 2u) & 127u];
                const uint32_t wlo = __builtin_amdgcn_alignbit(d1, d0, bp & 31u), whi = __builtin_amdgcn_alignbit(d2, d1, bp & 31u);
                const uint64_t w = ((uint64_t)whi << 32) | wlo;  // 64 bits of the stream from bit P + lane
                uint32_t info;  // bits 0-5 the token's bits, 6-7 kind (0 literal, 1 match, 2 end of block), 8 "not decoded yet", 9-17 symbols
                {
                    const uint32_t e = S.lit[wlo & ((1u << ROOT3) - 1u)];
                    const bool okd = fields(e, 0u, false, w, tok, info);
                    if (e == 0u || !okd) info |= 0x100u;
                }
                uint32_t pos = 0, span = 0;
                uint64_t real = 0;
                GZ_STAMP(1);
                GZ_COUNT(6, 1);
                // the walk along the real chain: one v_readlane a token.  The inner loop takes decoded literals and matches only
                // (one test for everything else: bit 8 "not decoded yet", bit 7 = kinds 2 and 3), so that the compiler's loop is
                // the dozen scalar instructions it looks like; the rare tokens are dealt with outside and the walk resumes.
                for (;;) {
                    uint32_t inf = 0;
                    bool special = false;
                    while (pos < 64u && span < SPAN3) {
                        inf = (uint32_t)__builtin_amdgcn_readlane((int)info, (int)pos);
                        if (inf & 0x180u) {
                            special = true;
                            break;
                        }
                        tpos = wlane(tpos, span, pos);
                        real |= 1ull << pos;
                        span += (inf >> 9) & 0x1FFu;
                        pos += inf & 63u;
                    }
                    if (!special) break;  // the window's bits or the span's symbols are used up
                    if (inf & 0x100u) {  // a code longer than a root table: the canonical walk, in the token's own lane
                        GZ_COUNT(10, 1);
                        if ((uint32_t)lane == pos) {
                            uint32_t e = S.lit[wlo & ((1u << ROOT3) - 1u)];
                            if (e == 0u) {
                                uint32_t l2;
                                // (the walk's fifteen single-bit extractions of w are loop invariants to the compiler, which
                                //  moved them in front of the token walk: fifteen vector instructions a WINDOW for a branch a
                                //  FASTQ stream takes once in thousands of tokens.  An opaque copy keeps them in here.)
                                uint64_t wq = w;
                                asm volatile("" : "+v"(wq));
                                const uint32_t sy = slow_symbol(wq, S.lcount, S.lsym, l2);
                                if (l2 != 0u && sy < 286u) e = lit_entry3(sy, l2);
                            }
                            if (e != 0u) {
                                uint32_t t2, i2;
                                bool ok = fields(e, 0u, false, w, t2, i2);
                                if (!ok) {  // the distance code is a long one
                                    const uint32_t l = e & 15u, lext = (e >> 11) & 7u;
                                    uint32_t dl2;
                                    const uint32_t dsy = slow_symbol(w >> (l + lext), S.dcount, S.dsym, dl2);
                                    if (dl2 != 0u && dsy < 30u) ok = fields(e, dist_entry3(dsy, dl2), true, w, t2, i2);
                                }
                                if (ok) {
                                    tok = t2;
                                    info = i2;
                                }
                            }
                        }
                        inf = (uint32_t)__builtin_amdgcn_readlane((int)info, (int)pos);
                        if (inf & 0x100u) {
                            status = ST_LITLEN;
                            break;
                        }
                    }
                    if (((inf >> 6) & 3u) == 2u) {
                        eob = true;
                        pos += inf & 63u;
                        break;
                    }
                    // a token the slow path decoded: taken like the inner loop takes its own
                    tpos = wlane(tpos, span, pos);
                    real |= 1ull << pos;
                    span += (inf >> 9) & 0x1FFu;
                    pos += inf & 63u;
                }
                GZ_STAMP(2);
                GZ_COUNT(7, (uint64_t)__popcll(real));
                if (status) break;
                P += pos;
                if (P > valid_bits) {
                    status = ST_INPUT;
                    break;
                }
                if ((uint64_t)op + span > cap) {
                    status = ST_ROOM;
                    break;
                }
                // phase 2: expand.  LDS operations of one wave execute in order: a read issued behind a write sees it.
                asm volatile("" ::: "memory");
                const bool act = (real >> lane) & 1u;
                const bool isl = act && (tok & 0x80000000u);
                {   // a distance beyond the window (or beyond the start of a member that began in this chunk)
                    const uint32_t at = op + tpos, dd = tok & 0xFFFFu;
                    const bool far = act && !isl && (fresh_member ? dd > at - floor_op : dd > at + WSIZE);
                    if (__ballot(far)) {
                        status = ST_FAR;
                        break;
                    }
                }
                if (isl) S.ring[(op + tpos) & M3] = (uint16_t)(tok & 0xFFu);
                const uint64_t mall = __ballot(act && !isl);
                GZ_COUNT(8, (uint64_t)__popcll(mall));
                // Far sources first.  A distance beyond the ring is longer than any match, so the source neither overlaps its
                // own output nor anything this window writes: it lies in HBM (flushed a window ago at the latest) or before
                // the chunk (markers).  All of the window's far matches of up to 64 symbols are loaded NOW -- one coalesced
                // load each, past the L1 (sc1: the lines were written by this wave) -- and waited for once.
                uint64_t mfar = __ballot(act && !isl && (tok & 0xFFFFu) > RING_DMAX3 && (tok >> 16) <= 64u);
                uint64_t mf_done = 0;
                if (mfar) {
                    // (their place in the order does not matter: they read nothing this window writes and nobody else writes where
                    //  they write; a later match of the ring that copies FROM one finds it written.  So they leave the in-order loop
                    //  below -- six matches in ten on FASTQ text are of this kind.)
                    __builtin_amdgcn_s_waitcnt(0x0F70);  // what was flushed is in HBM
                    uint64_t m2 = mfar;
                    uint16_t fv[4] = {0, 0, 0, 0};
                    uint32_t fP[4] = {0, 0, 0, 0}, fL[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        if (!m2) break;
                        const uint32_t j = (uint32_t)__builtin_ctzll(m2);
                        m2 &= m2 - 1;
                        mf_done |= 1ull << j;
                        const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)tok, (int)j);
                        const uint32_t Pj = op + (uint32_t)__builtin_amdgcn_readlane((int)tpos, (int)j);
                        const uint32_t L = t >> 16, D = t & 0xFFFFu;
                        const int32_t sp = (int32_t)(Pj - D) + lane;  // this lane's source position (negative: before the chunk)
                        uint16_t v = (uint16_t)(0x8000u | (uint32_t)(sp + (int32_t)WSIZE));
                        if ((uint32_t)lane < L && sp >= 0) v = __hip_atomic_load(&o[sp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        fv[g] = v;
                        fP[g] = Pj;
                        fL[g] = L;
                    }
#pragma unroll
                    for (int g = 0; g < 4; g++)
                        if ((uint32_t)lane < fL[g]) S.ring[(fP[g] + (uint32_t)lane) & M3] = fv[g];  // (fL = 0: not one of them)
                }
                // then the other matches in order: the ring's own (one read, one write; with D < L the pattern repeats: lane % D),
                // long ones (and the fifth far one of a window) 64 symbols at a time
                uint64_t mm = mall & ~mf_done;
                while (mm) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(mm);
                    mm &= mm - 1;
                    const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)tok, (int)j);
                    const uint32_t Pj = op + (uint32_t)__builtin_amdgcn_readlane((int)tpos, (int)j);
                    const uint32_t L = t >> 16, D = t & 0xFFFFu;
                    if (L <= 64u && D <= RING_DMAX3) {
                        uint32_t from = (uint32_t)lane;
                        if (D < 64u) {
                            const uint32_t iv = (uint32_t)__builtin_amdgcn_readlane((int)inv, (int)D);
                            from -= D * (((uint32_t)lane * iv) >> 16);
                        }
                        const uint16_t v = S.ring[(Pj - D + from) & M3];
                        if ((uint32_t)lane < L) S.ring[(Pj + (uint32_t)lane) & M3] = v;
                    } else {  // long (or the fifth far one of a window): every lane a symbol, 64 at a time
                        GZ_COUNT(9, 1);
                        if (D > RING_DMAX3) __builtin_amdgcn_s_waitcnt(0x0F70);
                        for (uint32_t i = (uint32_t)lane; i < L; i += 64) {
                            const uint32_t from = D >= L ? i : i % D;
                            uint16_t v;
                            if (D <= RING_DMAX3) {
                                v = S.ring[(Pj - D + from) & M3];
                            } else if (Pj + from >= D) {  // older than the ring, inside the chunk: from the output
                                v = __hip_atomic_load(&o[Pj + from - D], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            } else {  // older than the ring, before the chunk: the marker itself
                                v = (uint16_t)(0x8000u | (Pj + WSIZE - D + from));
                            }
                            S.ring[(Pj + i) & M3] = v;
                        }
                    }
                }
                asm volatile("" ::: "memory");
                op += span;
                GZ_STAMP(3);
                if (op - flushed >= FLUSH3) flush_to(op);
                GZ_STAMP(4);
            }
            if (status) break;
            br.init(in, P, lane);  // the scalar reader goes on behind the block
        }
        if (bfinal) {
            // ---- the member ends: trailer (CRC-32, ISIZE), then another member's header, or the end of the stream
            br.align8();
            bitpos = br.bitpos();
            uint64_t q = bitpos >> 3;
            const uint64_t valid_bytes = valid_bits >> 3;
            if (q + 8 > valid_bytes) {
                status = ST_INPUT;
                break;
            }
            if (n_members == MAX_MEMBERS) {
                status = ST_MEMBERS;
                break;
            }
            const uint64_t tr = bits_at(bytes, q * 8);
            if (lane == 0) {
                cd.m_off[n_members] = op;
                cd.m_crc[n_members] = (uint32_t)tr;
                cd.m_isize[n_members] = (uint32_t)(tr >> 32);
            }
            n_members++;
            q += 8;
            bool member = false, trunc = false;
            if (q == valid_bytes && (at_eof & 1u)) {
                member = false;
            } else if (q + 10 > valid_bytes) {
                const uint64_t hv = bits_at(bytes, q * 8);
                if ((at_eof & 1u) && !(q + 2 <= valid_bytes && (hv & 0xFFFF) == 0x8B1F)) member = false;
                else trunc = true;
            } else {
                const uint64_t hv = bits_at(bytes, q * 8);
                const uint32_t flg = (uint32_t)(hv >> 24) & 0xFFu;
                if ((hv & 0xFFFFFF) != 0x088B1Fu || (flg & 0xE0u)) {
                    member = false;  // bytes that are no gzip member: ignored like gzip does
                } else {
                    member = true;
                    uint64_t p = q + 10;
                    if (flg & 4u) {
                        if (p + 2 > valid_bytes) trunc = true;
                        else {
                            const uint32_t xlen = (uint32_t)bits_at(bytes, p * 8) & 0xFFFFu;
                            p += 2 + xlen;
                            if (p > valid_bytes) trunc = true;
                        }
                    }
                    for (uint32_t bit = 8; bit <= 16 && !trunc; bit <<= 1)
                        if (flg & bit) {
                            for (;;) {
                                const uint64_t pp = p + (uint64_t)lane;
                                const bool z = pp < valid_bytes && bytes[pp] == 0;
                                const uint64_t zm = __ballot(z);
                                if (zm) {
                                    p += (uint64_t)__builtin_ctzll(zm) + 1;
                                    break;
                                }
                                p += 64;
                                if (p >= valid_bytes) {
                                    trunc = true;
                                    break;
                                }
                            }
                        }
                    if (!trunc && (flg & 2u)) {
                        p += 2;
                        if (p > valid_bytes) trunc = true;
                    }
                    q = p;
                }
            }
            if (trunc) {
                status = ST_GZHEAD;
                break;
            }
            if (!member) {
                flags |= 1u;  // the stream ends here
                br.init(in, q * 8, lane);
                break;
            }
            br.init(in, q * 8, lane);
            fresh_member = true;
            floor_op = op;
        }
    }
    flush_to(op);
#ifdef NH_GZ_PROF
    pf[5] = __builtin_amdgcn_s_memtime() - tstart;
    if (lane == 0)
        for (int i = 0; i < 12; i++) cd.prof[i] = pf[i];
#endif
    if (lane == 0) {
        cd.status = status;
        cd.blocks = blocks;
        cd.bit_end = br.bitpos();
        cd.out_len = op;
        cd.n_members = n_members;
        cd.flags = flags;
    }
}

// ---- the search: a candidate's header, one lane each ---------------------------------------------------------------
constexpr uint32_t CLROW = 129;  // bytes per lane of the code-length code's table (odd: the lanes' rows start in different banks)
// One lane, one candidate: does a dynamic block header at bit `cand` decode into two complete codes with an end-of-block
// symbol?  (Code lengths are not kept: Kraft sums as they come.)  tbl = this lane's 128 entries (length << 5 | symbol).
// (the table is addressed as an offset into the workgroup's LDS: a pointer parameter would be a flat pointer here)
__device__ __forceinline__ bool lane_header_ok(const uint8_t *bytes, uint64_t cand, uint64_t valid_bits, uint32_t tbl_off) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_bytes[];
    uint8_t *const tbl = lds_bytes + tbl_off;
    uint64_t b = cand + 3;
    uint64_t w = bits_at(bytes, b);
    const uint32_t nlit = (uint32_t)(w & 31) + 257, ndist = (uint32_t)((w >> 5) & 31) + 1, ncl = (uint32_t)((w >> 10) & 15) + 4;
    if (nlit > 286 || ndist > 30) return false;
    b += 14;
    uint8_t cl[19];
#pragma unroll
    for (int i = 0; i < 19; i++) cl[i] = 0;
    w = bits_at(bytes, b);
    uint64_t w2 = bits_at(bytes + 7, b);
#pragma unroll
    for (uint32_t i = 0; i < 19; i++) {
        const uint32_t at = 3 * i;
        const uint32_t v = i < ncl ? (at + 3 <= 56 ? (uint32_t)(w >> at) & 7 : (uint32_t)(w2 >> (at - 56)) & 7) : 0u;
        // (CLORDER as a switch over the unrolled index: the array stays in registers)
        constexpr uint8_t ORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        cl[ORD[i]] = (uint8_t)v;
    }
    b += 3 * ncl;
    // canonical codes of the code-length code, its 7-bit table (complete by the cheap test that made this a candidate)
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, next[8];
#pragma unroll
    for (int i = 0; i < 19; i++) cnt[cl[i]]++;
    cnt[0] = 0;
    uint32_t code = 0;
#pragma unroll
    for (int l = 1; l < 8; l++) {
        code = (code + cnt[l - 1]) << 1;
        next[l] = code;
    }
    for (uint32_t k = 0; k < 128; k++) tbl[k] = 0;
#pragma unroll
    for (uint32_t s = 0; s < 19; s++) {
        const uint32_t l = cl[s];
        if (l) {
            const uint32_t cdw = next[l]++;
            if (cdw >> l) return false;  // over-subscribed
            const uint32_t r = __builtin_bitreverse32(cdw) >> (32 - l);
            for (uint32_t k = r; k < 128; k += 1u << l) tbl[k] = (uint8_t)(l << 5 | s);
        }
    }
    const uint32_t total = nlit + ndist;
    uint32_t i = 0, prev = 0, kl = 0, kd = 0, cdist = 0, eob = 0;
    while (i < total) {
        if (b + 64 > valid_bits) return false;
        w = bits_at(bytes, b);
        const uint32_t e = tbl[w & 127];
        const uint32_t l = e >> 5, sy = e & 31;
        if (l == 0) return false;
        b += l;
        w >>= l;
        uint32_t rep = 1, val = sy;
        if (sy == 16) {
            if (i == 0) return false;
            rep = 3 + ((uint32_t)w & 3);
            b += 2;
            val = prev;
        } else if (sy == 17) {
            rep = 3 + ((uint32_t)w & 7);
            b += 3;
            val = 0;
        } else if (sy == 18) {
            rep = 11 + ((uint32_t)w & 127);
            b += 7;
            val = 0;
        }
        if (i + rep > total) return false;
        if (val) {
            const uint32_t wgt = 32768u >> val;
            const uint32_t n1 = i >= nlit ? 0u : (i + rep <= nlit ? rep : nlit - i);
            kl += n1 * wgt;
            kd += (rep - n1) * wgt;
            cdist += rep - n1;
            if (i <= 256u && 256u < i + rep) eob = 1;
        }
        i += rep;
        prev = val;
    }
    return eob && kl == 32768u && (kd == 32768u || cdist <= 1u);
}

// ---- k_search3: the block search on the compact tables of k_inflate3 (9.3 KB of LDS: seventeen waves a CU; a stretch's
// wave is latency-bound, 3 ms alone) -----------------------------------------------------------------------------------
__device__ __forceinline__ bool plausible_block3(Lds3 &S, const uint32_t *in, uint64_t cand, uint64_t valid_bits, int lane) {
    BitRd br;
    br.init(in, cand + 3, lane);
    int nlit, ndist;
    if (parse_dynamic3(S, br, lane, nlit, ndist)) return false;
    uint32_t sl = 0, sd = 0, cd = 0;
    for (int k = lane; k < nlit; k += 64) sl += S.lens[k] ? 32768u >> S.lens[k] : 0u;
    for (int k = lane; k < ndist; k += 64) {
        const uint32_t l = S.lens[nlit + k];
        sd += l ? 32768u >> l : 0u;
        cd += l != 0;
    }
    sl = wsum(sl);
    sd = wsum(sd);
    cd = wsum(cd);
    if (!(sl == 32768u && (sd == 32768u || cd <= 1u))) return false;
    if (!build3(S, S.lens, nlit, S.lcount, S.lsym, S.lit, ROOT3, 1, lane)) return false;
    if (!build3(S, S.lens + nlit, ndist, S.dcount, S.dsym, S.dist, DROOT3, 2, lane)) return false;
    uint32_t produced = 0;
    for (int tok = 0; tok < 256; tok++) {  // the block's first tokens must walk
        if (br.bitpos() + 64 > valid_bits) return true;  // (the look-ahead ends here: what was walked was fine)
        br.ensure32(lane);
        uint32_t e = uni(S.lit[br.peek(ROOT3)]);
        if (e == 0) {
            uint32_t l;
            const uint32_t sy = slow_symbol(br.bb, S.lcount, S.lsym, l);
            if (l == 0 || sy >= 286u) return false;
            e = uni(lit_entry3(uni(sy), uni(l)));
        }
        br.drop(e & 15u);
        const uint32_t kind = (e >> 4) & 3u;
        if (kind == 2u) break;
        if (kind == 0u) {
            produced++;
            continue;
        }
        const uint32_t pay = (e >> 6) & 0xFFu;
        const uint32_t len = (uint32_t)LBASE[pay & 31u] + br.take(pay >> 5);
        br.ensure32(lane);
        uint32_t de = uni(S.dist[br.peek(DROOT3)]);
        if (de == 0) {
            uint32_t dl;
            const uint32_t dsy = slow_symbol(br.bb, S.dcount, S.dsym, dl);
            if (dl == 0 || dsy >= 30u) return false;
            de = uni(dist_entry3(uni(dsy), uni(dl)));
        }
        br.drop(de & 15u);
        const uint32_t dist = (uint32_t)DBASE[(de >> 4) & 31u] + br.take((de >> 9) & 15u);
        if (dist > produced + WSIZE) return false;
        produced += len;
    }
    return true;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_search3(const uint32_t *in, uint64_t valid_bits, uint64_t stretch_bits, uint64_t first_bit,
                                                uint64_t *start) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Lds3 &S = *(Lds3 *)smem;  // the wave-wide parse's tables and the lanes' own (64 x CLROW bytes) share the space: never live together
    __shared__ uint64_t cand_list[128];
    __shared__ uint32_t pre_list[128];
    const uint32_t tbl = (uint32_t)threadIdx.x * CLROW;
    const int lane = (int)threadIdx.x;
    const uint32_t c = blockIdx.x;
    // (first_bit == NONE: a piece decoded ahead of the stream -- no known start, stretch 0 is searched like the others)
    if (c == 0 && first_bit != NONE) {
        if (lane == 0) start[0] = first_bit;
        return;
    }
    uint64_t from = (uint64_t)c * stretch_bits, to = from + stretch_bits;
    if (first_bit != NONE && from <= first_bit) from = first_bit + 1;
    if (to + 160 > valid_bits) to = valid_bits > 160 ? valid_bits - 160 : 0;
    const uint8_t *bytes = (const uint8_t *)in;
    uint64_t found = NONE;
    uint32_t nc = 0, np = 0;
    // Two sieves in front of the candidates' evaluation.  The cheap one (three header bits, HLIT, HDIST: 13 bits of one load) runs on
    // every bit position, 64 a round, and passes one in nine; what passes is LISTED, and the dear one -- the Kraft sum of the
    // code-length code's up to nineteen lengths, 150 vector instructions -- runs on full waves of listed positions instead of on
    // every round of 64 positions with at most a handful of live lanes (it did: some lane passes the cheap sieve in nearly every
    // round).  What passes both is listed again and evaluated (the header parsed per lane, the block's first tokens walked by the
    // wave) 64 at a time.  Order is kept throughout -- lists are filled by ballot compaction and emptied from the front -- so the
    // first plausible start wins as before.  ONE loop with one site for each stage (the stages are large: inlined twice they
    // spilled); once the stretch is scanned the rounds go on until the lists are empty.
    const uint64_t base0 = (uint64_t)c * stretch_bits;  // (listed positions as 32-bit offsets from here: a stretch is 2^18 bits)
    for (uint64_t b0 = from; found == NONE; b0 += 64) {
        const bool scanning = b0 < to;
        if (scanning) {
            const uint64_t b = b0 + (uint64_t)lane;
            const uint64_t w = bits_at(bytes, b);
            const bool pre = b < to && (w & 7) == 4 && ((w >> 3) & 31) <= 29 && ((w >> 8) & 31) <= 29;
            const uint64_t m = __ballot(pre);  // (seven of eight positions fail the first three bits)
            if (m) {
                if (pre) pre_list[np + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint32_t)(b - base0);
                np += (uint32_t)__popcll(m);
            }
        }
        if (np >= 64u || (!scanning && np)) {  // the dear sieve on the first 64 (or the last few) listed positions
            const uint32_t count = np < 64u ? np : 64u;
            LDS_ORDER();
            const bool live = (uint32_t)lane < count;
            const uint64_t b = base0 + (live ? pre_list[lane] : 0u);
            bool ok = false;
            if (live) {
                const uint64_t w = bits_at(bytes, b), w2 = bits_at(bytes + 7, b);  // w2: bits 56.. of the window
                const int ncl = (int)((w >> 13) & 15) + 4;
                int left = 128, any = 0;
                for (int i = 0; i < 19; i++) {
                    const unsigned at = 17 + 3 * (unsigned)i;
                    const unsigned l = i < ncl ? (at + 3 <= 56 ? (unsigned)(w >> at) & 7 : (unsigned)(w2 >> (at - 56)) & 7) : 0u;
                    if (l) {
                        left -= 128 >> l;
                        any = 1;
                    }
                }
                ok = any && left == 0;
            }
            const uint64_t m = __ballot(ok);
            if (m) {
                if (ok) cand_list[nc + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = b;
                nc += (uint32_t)__popcll(m);
            }
            const uint32_t keep = (uint32_t)lane + count < np ? pre_list[count + lane] : 0u;
            LDS_ORDER();
            pre_list[lane] = keep;
            np -= count;
            LDS_ORDER();
        }
        if (nc >= 64u || (!scanning && np == 0 && nc)) {  // the first 64 (or the last few) candidates, in order
            const uint32_t count = nc < 64u ? nc : 64u;
            LDS_ORDER();
            const uint64_t mine = (uint32_t)lane < count ? cand_list[lane] : NONE;
            const bool ok = mine != NONE && lane_header_ok(bytes, mine, valid_bits, tbl);
            uint64_t m = __ballot(ok);
            while (m && found == NONE) {
                const uint32_t j = (uint32_t)__builtin_ctzll(m);
                const uint64_t cb = cand_list[j];
                if (plausible_block3(S, in, cb, valid_bits, lane)) found = cb;
                m &= m - 1;
            }
            LDS_ORDER();
            const uint64_t keep = (uint32_t)lane + count < nc ? cand_list[count + lane] : NONE;
            LDS_ORDER();
            cand_list[lane] = keep;
            nc -= count;
            LDS_ORDER();
        }
        if (!scanning && np == 0 && nc == 0) break;
    }
    if (lane == 0) start[c] = found;
}

__global__ void k_test_mark_end(ChunkDesc *desc, uint32_t c) { desc[c].flags |= 1u; }  // NOHUMAN_GZDEV_FAKE_END (tests)

// After the decode: which chunks count, do they chain, where does the text of each begin
__global__ __launch_bounds__(1024) void k_finish(ChunkDesc *desc, uint32_t n, uint64_t *toff, SegResult *res) {
    __shared__ uint32_t s_end, s_broken, s_bad, s_members, s_cnt;
    __shared__ unsigned long long s_scan[1024];
    const uint32_t t = threadIdx.x;
    if (t == 0) {
        s_end = NOIDX;
        s_broken = NOIDX;
        s_bad = NOIDX;
        s_members = 0;
        s_cnt = 0;
    }
    __syncthreads();
    // the chunk the stream ended in, or the last one
    for (uint32_t c = t; c < n; c += 1024)
        if (desc[c].bit_start != NONE && (desc[c].flags & 1u)) atomicMin(&s_end, c);
    __syncthreads();
    uint32_t last = s_end;
    if (last == NOIDX) {
        __syncthreads();
        for (uint32_t c = t; c < n; c += 1024)
            if (desc[c].bit_start != NONE) atomicMax((int *)&s_end, (int)c);  // (NOIDX is -1 as int: any index is larger)
        __syncthreads();
        last = s_end;
    }
    const bool stream_end = last != NOIDX && (desc[last].flags & 1u);
    // chunks behind the end of the stream do not count (what the search found there was never part of it).  Only the start is
    // struck: when the "end" was a false start's garbage, the seam check below strikes THAT chunk, k_plan puts these starts
    // back and what they decoded must still be there (everything that reads a chunk looks at bit_start first).
    for (uint32_t c = t; c < n; c += 1024)
        if (c > last && desc[c].bit_start != NONE) desc[c].bit_start = NONE;
    __syncthreads();
    // errors and seams
    for (uint32_t c = t; c <= last && c < n; c += 1024) {
        const ChunkDesc &d = desc[c];
        if (d.bit_start == NONE) continue;
        if (d.status) atomicMin(&s_bad, c);
        atomicAdd(&s_members, d.n_members);
        if (c < last) {
            uint32_t nx = c + 1;
            while (desc[nx].bit_start == NONE) nx++;
            if (d.bit_end != desc[nx].bit_start) atomicMin(&s_broken, nx);
        }
    }
    // exclusive prefix sum of the lengths: per thread a run of chunks, then a scan over the threads
    const uint32_t per = (n + 1023) / 1024;
    unsigned long long sum = 0;
    uint32_t mine = 0;
    for (uint32_t i = 0; i < per; i++) {
        const uint32_t c = t * per + i;
        if (c < n && desc[c].bit_start != NONE) {
            sum += desc[c].out_len;
            mine++;
        }
    }
    if (mine) atomicAdd(&s_cnt, mine);
    s_scan[t] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const unsigned long long v = t >= o ? s_scan[t - o] : 0;
        __syncthreads();
        s_scan[t] += v;
        __syncthreads();
    }
    unsigned long long run = s_scan[t] - sum;
    for (uint32_t i = 0; i < per; i++) {
        const uint32_t c = t * per + i;
        if (c < n) {
            toff[c] = run;
            if (desc[c].bit_start != NONE) run += desc[c].out_len;
        }
    }
    if (t == 1023) res->total = s_scan[1023];
    if (t == 0) {
        res->n_chunks = s_cnt;  // (counted by all threads above: one thread walking 8192 descriptors was a millisecond a piece)
        res->end_chunk = last;
        res->end_bit = last != NOIDX ? desc[last].bit_end : 0;
        res->broken = s_broken;
        res->bad_chunk = s_bad;
        res->bad_status = s_bad != NOIDX ? desc[s_bad].status : 0;
        res->stream_end = stream_end ? 1u : 0u;
        res->members = s_members;
        uint64_t fs = NONE;
        for (uint32_t c = 0; c < n && fs == NONE; c++) fs = desc[c].bit_start;  // (the first stretches' chunks: a handful of reads)
        res->first_start = fs;
    }
}

// ---- the same windows in three passes instead of log2(n) rounds over all maps (round 4: 13 rounds x 3 x 0.5 GB a piece of
// 7560 chunks were 19 GB of traffic, 7 ms; this is 1.7 GB).  Groups of SCAN_GROUP consecutive chunks:
//   k_scan_local   a workgroup a group walks its chunks in order, the running composition R in LDS (32 Ki x 16 bit, double
//                  buffered): local[c] = map[c] o ... o map[first of the group]  (the maps straight from the symbols)
//   k_scan_groups  one workgroup walks the groups' totals: pre[g] = total[g - 1] o ... o total[0]  (a few hundred steps)
//   k_scan_windows windows[c + 1] = local[c] o pre[group of c] applied to the window before the piece
constexpr uint32_t SCAN_GROUP = 32;
__device__ __forceinline__ uint16_t map_entry(const uint16_t *src, uint32_t nsym, uint32_t j) {
    if (nsym >= WSIZE) return src[nsym - WSIZE + j];
    if (j < WSIZE - nsym) return (uint16_t)(0x8000u | (j + nsym));  // the old window moves up
    return src[j - (WSIZE - nsym)];
}
__global__ __launch_bounds__(1024) void k_scan_local(const ChunkDesc *d, const uint16_t *sym, uint32_t slot_syms, uint32_t n, uint16_t *local) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *R = (uint16_t *)smem;  // WSIZE entries: the running composition (a thread's 32 entries pass through registers)
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    const uint32_t c0 = g * SCAN_GROUP, c1 = c0 + SCAN_GROUP < n ? c0 + SCAN_GROUP : n;
    for (uint32_t c = c0; c < c1; c++) {
        const uint32_t nsym = d[c].bit_start == NONE ? 0u : d[c].out_len;
        const uint16_t *src = sym + (uint64_t)c * slot_syms;
        uint16_t *out = local + (size_t)c * WSIZE;
        uint16_t v[WSIZE / 1024];
#pragma unroll
        for (uint32_t k = 0; k < WSIZE / 1024; k++) {
            uint16_t x = map_entry(src, nsym, t + 1024 * k);
            if (c != c0 && (x & 0x8000u)) x = R[x & 0x7FFFu];
            v[k] = x;
        }
        __syncthreads();  // every read of R is done
#pragma unroll
        for (uint32_t k = 0; k < WSIZE / 1024; k++) {
            R[t + 1024 * k] = v[k];
            out[t + 1024 * k] = v[k];
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void k_scan_groups(const uint16_t *local, uint32_t n, uint16_t *pre) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *R = (uint16_t *)smem;
    const uint32_t t = threadIdx.x;
    const uint32_t ng = (n + SCAN_GROUP - 1) / SCAN_GROUP;
    for (uint32_t j = t; j < WSIZE; j += 1024) {
        R[j] = (uint16_t)(0x8000u | j);  // pre[0]: the identity
        pre[j] = (uint16_t)(0x8000u | j);
    }
    __syncthreads();
    for (uint32_t g = 1; g < ng; g++) {  // pre[g] = total[g - 1] o pre[g - 1]
        const uint32_t last = g * SCAN_GROUP - 1;  // the last chunk of group g - 1
        const uint16_t *tot = local + (size_t)last * WSIZE;
        uint16_t *out = pre + (size_t)g * WSIZE;
        uint16_t v[WSIZE / 1024];
#pragma unroll
        for (uint32_t k = 0; k < WSIZE / 1024; k++) {
            uint16_t x = tot[t + 1024 * k];
            if (x & 0x8000u) x = R[x & 0x7FFFu];
            v[k] = x;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < WSIZE / 1024; k++) {
            R[t + 1024 * k] = v[k];
            out[t + 1024 * k] = v[k];
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_scan_windows(const uint16_t *local, const uint16_t *pre, const uint8_t *w0, uint8_t *windows) {
    const uint32_t c = blockIdx.x;  // the window of chunk c
    uint8_t *w = windows + (size_t)c * WSIZE;
    if (c == 0) {
        for (uint32_t j = blockIdx.y * 256 + threadIdx.x; j < WSIZE; j += gridDim.y * 256) w[j] = w0[j];
        return;
    }
    const uint16_t *m = local + (size_t)(c - 1) * WSIZE;
    const uint16_t *p = pre + (size_t)((c - 1) / SCAN_GROUP) * WSIZE;
    for (uint32_t j = blockIdx.y * 256 + threadIdx.x; j < WSIZE; j += gridDim.y * 256) {
        uint16_t x = m[j];
        if (x & 0x8000u) x = p[x & 0x7FFFu];
        w[j] = x & 0x8000u ? w0[x & 0x7FFFu] : (uint8_t)x;
    }
}

// every chunk's symbols become bytes at their place in the text
__global__ __launch_bounds__(256) void k_resolve(const ChunkDesc *d, const uint16_t *sym, uint32_t slot_syms, const uint8_t *windows,
                                                 const uint64_t *toff, uint8_t *text) {
    const uint32_t c = blockIdx.x;
    if (d[c].bit_start == NONE) return;
    const uint32_t nsym = d[c].out_len;
    const uint16_t *src = sym + (uint64_t)c * slot_syms;
    const uint8_t *w = windows + (size_t)c * WSIZE;
    uint8_t *dst = text + toff[c];
    for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < nsym; i += gridDim.y * 256) {
        const uint16_t x = src[i];
        dst[i] = x & 0x8000u ? w[x & 0x7FFFu] : (uint8_t)x;
    }
}

// the window behind the piece: the last 32 KiB of (old window || text of the piece)
__global__ __launch_bounds__(1024) void k_carry_window(const uint8_t *text, uint64_t total_cap, const SegResult *res, const uint8_t *old_w,
                                                       uint8_t *new_w) {
    const uint64_t total = res->total < total_cap ? res->total : total_cap;
    for (uint32_t j = threadIdx.x; j < WSIZE; j += 1024) {
        uint8_t v;
        if (total >= WSIZE) v = text[total - WSIZE + j];
        else if (j < WSIZE - total) v = old_w[j + total];
        else v = text[j - (WSIZE - total)];
        new_w[j] = v;
    }
}

// ---- CRC-32 of every chunk's pieces (gzip polynomial, reflected) ---------------------------------------------------
constexpr uint32_t CRC_POLY = 0xEDB88320u;
__device__ __forceinline__ uint32_t multmodp(uint32_t a, uint32_t b) {  // a * b mod P in the reflected representation
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = b & 1 ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
__device__ __forceinline__ uint32_t x8n_modp(uint64_t n) {  // x^(8n) mod P
    uint32_t p = 1u << 31, sq = 0x00800000u;  // x^0 ; x^8
    while (n) {
        if (n & 1) p = multmodp(sq, p);
        sq = multmodp(sq, sq);
        n >>= 1;
    }
    return p;
}
__global__ __launch_bounds__(256) void k_crc(ChunkDesc *d, const uint64_t *toff, const uint8_t *text) {
    __shared__ uint32_t T[4][256];
    __shared__ uint32_t part[256];
    __shared__ uint32_t xl[9];  // x^(8 L 2^j) mod P: what a CRC moves by when 2^j slices follow it
    const uint32_t c = blockIdx.x, t = threadIdx.x;
    if (d[c].bit_start == NONE) return;
    {
        uint32_t v = t;
        for (int k = 0; k < 8; k++) v = v & 1 ? (v >> 1) ^ CRC_POLY : v >> 1;
        T[0][t] = v;
    }
    __syncthreads();
    for (int k = 1; k < 4; k++) {
        const uint32_t v = T[k - 1][t];
        T[k][t] = (v >> 8) ^ T[0][v & 0xFF];
    }
    __syncthreads();
    const uint32_t nm = d[c].n_members, total = d[c].out_len;
    const uint8_t *base = text + toff[c];
    uint32_t a = 0;
    for (uint32_t p = 0; p <= nm; p++) {
        const uint32_t b = p < nm ? d[c].m_off[p] : total;
        const uint32_t len = b - a;
        // slices of L bytes, a thread each (L a multiple of 16); standard CRC-32 per slice, sixteen bytes a load (a lane's
        // slice is a kilobyte away from its neighbour's: every load instruction is 64 separate requests, so make them few)
        uint32_t L = (len + 255) / 256;
        L = (L + 15) & ~15u;
        if (L == 0) L = 16;
        const uint32_t s0 = t * L < len ? t * L : len, s1 = (t + 1) * L < len ? (t + 1) * L : len;
        uint32_t crc = 0xFFFFFFFFu;
        const uint8_t *q = base + a + s0;
        uint32_t k = s1 - s0;
        while (k >= 16) {
            struct __attribute__((packed)) U16 {
                uint32_t v[4];
            };
            const U16 w = *(const U16 *)q;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t x = crc ^ w.v[i];
                crc = T[3][x & 0xFF] ^ T[2][(x >> 8) & 0xFF] ^ T[1][(x >> 16) & 0xFF] ^ T[0][x >> 24];
            }
            q += 16;
            k -= 16;
        }
        while (k--) crc = (crc >> 8) ^ T[0][(crc ^ *q++) & 0xFF];
        part[t] = ~crc;
        if (t == 0) {
            uint32_t v = x8n_modp(L);
            for (int j = 0; j < 9; j++) {
                xl[j] = v;
                v = multmodp(v, v);
            }
        }
        __syncthreads();
        // crc(A || B) = crc(A) * x^(8|B|) + crc(B), as a tree over the slices: at level j thread i (a multiple of 2^(j+1)) takes
        // in the 2^j slices to its right (all of length L but the ragged end of the piece, whose thread computes its own power)
        for (int j = 0; j < 8; j++) {
            const uint32_t w = 1u << j;
            if ((t & (2 * w - 1)) == 0) {
                const uint64_t r0 = (uint64_t)(t + w) * L, r1 = (uint64_t)(t + 2 * w) * L;
                const uint32_t rb = (uint32_t)((r1 < len ? r1 : len) - (r0 < len ? r0 : len));  // bytes of the right block
                if (rb) part[t] = multmodp(rb == w * L ? xl[j] : x8n_modp(rb), part[t]) ^ part[t + w];
            }
            __syncthreads();
        }
        if (t == 0) d[c].piece_crc[p] = part[0];
        __syncthreads();
        a = b;
    }
}

}  // namespace gz

// =====================================================================================================================
// host side
// =====================================================================================================================
using namespace gz;

#define GZ_TRY(x)                                                                             \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

// The reader's large buffers are kept for the next run of the process instead of being freed: allocating and freeing them
// cost 0.5-0.9 s of a 2 s run (profiles/r04_e2e.txt).  What that is, per input file at the default piece and chunk sizes
// (512 MiB of gzip in 8192 chunks of 64 KiB): symbol slots 8192 x (16 x 64 Ki + 64 Ki) x 2 B = 18.3 GB, index maps 1.1 GB,
// windows 0.3 GB, input 0.5 GB (and as much page-locked), two text buffers of 3.5 GB, the record index 0.5 GB -- 28 GB of
// HBM a file, 56 GB a paired run.  The store is bounded BY BYTES per device (64 GiB of HBM -- one paired run's buffers --
// and 8 GiB page-locked; NOHUMAN_GZDEV_CACHE_GB / NOHUMAN_GZDEV_CACHE_HOST_GB) and evicts the oldest entries first, so runs
// over inputs of other sizes replace what is kept instead of piling up; every allocation of the run path gives the
// store back before it fails (dev_malloc / host_malloc, nh_internal.h).  NOHUMAN_GZDEV_CACHE=0 turns it off;
// dev_cache_trim() empties it (nh_close).
namespace {
struct CacheEntry {
    int device;
    size_t bytes;
    void *p;
    bool host;
};
std::mutex g_cache_mu;
std::vector<CacheEntry> g_cache;  // oldest first
bool cache_on() {
    static const bool on = !(getenv("NOHUMAN_GZDEV_CACHE") && getenv("NOHUMAN_GZDEV_CACHE")[0] == '0');
    return on;
}
size_t cache_limit(bool host) {
    static const size_t dev_lim = (size_t)((getenv("NOHUMAN_GZDEV_CACHE_GB") ? atof(getenv("NOHUMAN_GZDEV_CACHE_GB")) : 64.0) * 1073741824.0);
    static const size_t host_lim = (size_t)((getenv("NOHUMAN_GZDEV_CACHE_HOST_GB") ? atof(getenv("NOHUMAN_GZDEV_CACHE_HOST_GB")) : 8.0) * 1073741824.0);
    return host ? host_lim : dev_lim;
}
void cache_release(const CacheEntry &e) {
    if (e.host) (void)hipHostFree(e.p);
    else {
        (void)dev_set(e.device);
        (void)hipFree(e.p);
    }
}
}  // namespace
void dev_cache_trim() {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (CacheEntry &e : g_cache) cache_release(e);
    g_cache.clear();
}
// bytes the store holds (test hook: nh_gunzip_cache_bytes)
static size_t cache_bytes(int device, bool host) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    size_t sum = 0;
    for (const CacheEntry &e : g_cache)
        if (e.host == host && (host || device < 0 || e.device == device)) sum += e.bytes;
    return sum;
}
static void *cache_alloc(int device, size_t bytes, bool host) {
    if (!host) dev_check(device, "gzip reader, buffer allocation");
    if (cache_on() && bytes >= ((size_t)16u << 20)) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (size_t i = 0; i < g_cache.size(); i++) {
            CacheEntry &e = g_cache[i];
            if (e.host == host && (host || e.device == device) && e.bytes >= bytes && e.bytes <= bytes + bytes / 2) {
                void *p = e.p;
                g_cache.erase(g_cache.begin() + (long)i);
                return p;
            }
        }
    }
    void *p = nullptr;
    // (on failure: everything the process keeps between runs goes first, then one more try)
    const hipError_t e = host ? host_malloc(&p, bytes) : dev_malloc(&p, bytes);
    if (e != hipSuccess) return nullptr;
    return p;
}
static void cache_free(int device, size_t bytes, void *p, bool host) {
    if (!p) return;
    if (cache_on() && bytes >= ((size_t)16u << 20) && bytes <= cache_limit(host)) {
        std::vector<CacheEntry> out;
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            g_cache.push_back({device, bytes, p, host});
            // the bound: by bytes per device (page-locked memory: one pool), by entries overall; the oldest go first
            size_t sum = 0;
            for (const CacheEntry &e : g_cache)
                if (e.host == host && (host || e.device == device)) sum += e.bytes;
            for (size_t i = 0; i + 1 < g_cache.size() && sum > cache_limit(host);) {  // (the newest, this one, stays)
                const CacheEntry e = g_cache[i];
                if (e.host == host && (host || e.device == device)) {
                    sum -= e.bytes;
                    out.push_back(e);
                    g_cache.erase(g_cache.begin() + (long)i);
                } else {
                    i++;
                }
            }
            while (g_cache.size() > 96) {
                out.push_back(g_cache.front());
                g_cache.erase(g_cache.begin());
            }
        }
        const int dev = dev_current();  // (a LOGICAL device: hipGetDevice's ordinal is 0 for every one of NOHUMAN_FAKE_DEVICES)
        for (const CacheEntry &e : out) cache_release(e);
        if (!out.empty() && dev >= 0) (void)dev_set(dev);
        return;
    }
    if (host) (void)hipHostFree(p);
    else (void)hipFree(p);
}

class DevGunzipImpl {
    friend class DevGunzip;
    // what one piece is decoded from, and what its first phase found (phase_a / phase_b below)
    struct PieceJob {
        uint64_t a_byte = 0;        // the piece's buffer starts here in the file
        uint64_t first_bit = NONE;  // known start (bits from a_byte), or NONE
        uint32_t n = 0;             // stretches
        uint64_t limit_bits = 0;    // the piece ends at the first block boundary at or behind this bit (0: behind its last stretch)
        // phase A's results
        uint32_t n_str = 0, redo = 0;
        uint64_t valid_bits = 0, end_bit = 0;
        bool at_eof = false, valid = false;
        bool by_headers = false;    // BGZF: the chunks start at members (no window, no markers: nothing to scan)
        SegResult r{};
    };

    // The buffers of one device.  A reader that spreads its pieces over several devices (DevFastqReader with the devices
    // of a run: piece i on device i mod G) has one set per device; the stream's state -- position, CRC, the window behind
    // the last piece -- goes from piece to piece, the window by a 32 KiB copy from the set that decoded the piece before.
    struct DevSet {
        int device_ = -1;
        uint8_t *d_in_ = nullptr, *h_in_ = nullptr;
        uint8_t *h_in2_ = nullptr;  // second staging buffer: the next piece's bytes are copied there while this piece's kernels run
        uint64_t *d_start_ = nullptr, *d_toff_ = nullptr;
        ChunkDesc *d_desc_ = nullptr, *h_desc_ = nullptr;
        uint16_t *d_sym_ = nullptr, *d_maps_[2] = {nullptr, nullptr};
        uint8_t *d_windows_ = nullptr, *d_win_[2] = {nullptr, nullptr};
        int win_ = 0;
        SegResult *d_res_ = nullptr, *h_res_ = nullptr;
        hipEvent_t ev_[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        uint64_t pieces = 0;
        uint64_t *h_start_ = nullptr;  // BGZF: the chunks' starts, computed on the host from the members' headers
        PieceJob spec;           // a cell of the piece grid decoded ahead of the stream on this set (phase A only)
        uint64_t spec_cell = 0;
    };
    std::vector<std::unique_ptr<DevSet>> sets_;
    DevSet *s_ = nullptr;  // the set the current piece is decoded with

public:
    ~DevGunzipImpl() { close(); }

    int open(const char *path, int device, size_t seg_bytes, size_t stretch_bytes, std::string &err) {
        close();
        const auto t_open = std::chrono::steady_clock::now();
        path_ = path;
        sets_.emplace_back(new DevSet());
        s_ = sets_[0].get();
        s_->device_ = device;
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
        (void)madvise(m, size_, MADV_SEQUENTIAL);
        bool trunc = false;
        const uint8_t *d = gzip_member_body(base_, base_ + size_, &trunc);
        if (!d) {
            err = std::string("not in gzip format: ") + path;
            close();
            return -1;
        }
        pos_bit_ = (uint64_t)(d - base_) * 8;
        grid0_ = (uint64_t)(d - base_) / ALIGN * ALIGN;
        // BGZF (bgzip, htslib): every member is ONE final block of at most 64 KiB of text and says its own size in the 'B' 'C'
        // subfield of its header.  Nothing for the block search to find -- and nothing to search for: the chunks' starts are
        // read off the headers, a chunk never needs a window, and the members' CRCs are checked like any other's.
        bgzf_ = bgzf_block_size(0) != 0 && !getenv("NOHUMAN_GZDEV_NO_BGZF");
        bgzf_hdr_ = 0;
        if (bgzf_ && !stretch_bytes) stretch_bytes = (size_t)16u << 10;  // (a member of FASTQ text is 15-20 KB: one or two a chunk, four at most)
        if (bgzf_ && !seg_bytes) seg_bytes = (size_t)256u << 20;
        if (const char *e = getenv("NOHUMAN_GZDEV_SEG")) seg_bytes = (size_t)atol(e);
        if (const char *e = getenv("NOHUMAN_GZDEV_STRETCH")) stretch_bytes = (size_t)atol(e);
        // defaults (profiles/r04_inflate_summary.txt, sweep of piece and chunk sizes on 6.4 GB of text): pieces of 512 MiB of gzip in
        // chunks of 64 KiB -- 8192 chunks a piece, the search's work halves with the chunk count -- take 251 ms of kernels
        // where pieces of 256 MiB in chunks of 32 KiB took 314; a file too small for such a piece keeps the 32 KiB chunks
        // (the chip's 5120 wave slots want thousands of chunks a piece)
        const bool big = size_ >= ((size_t)512u << 20);
        stretch_ = stretch_bytes ? stretch_bytes : (size_t)((big ? 64u : 32u) << 10);
        if (stretch_ < 1024) stretch_ = 1024;
        stretch_ = (stretch_ + 63) & ~(size_t)63;
        // A piece is what the chip decodes at once, a wavefront a chunk: thousands of chunks, more than one round of the 5120
        // wave slots (smaller pieces leave them empty: profiles/r04_inflate_summary.txt, 32 MiB pieces take four times as
        // long).  No larger than the file; halved below while the buffers do not fit the HBM.
        seg_ = seg_bytes ? seg_bytes : (size_t)(512u << 20);
        if (seg_ < stretch_) seg_ = stretch_;
        if (seg_ > size_ + stretch_) seg_ = size_ + stretch_;
        seg_ = (seg_ + stretch_ - 1) / stretch_ * stretch_;
        // a chunk's block may run past the piece: that much more of the file is on the device.  zlib's blocks hold at
        // most 64 KiB (stored) or some 16 K symbols; 2 MiB covers encoders with far larger ones
        look_ = std::max<size_t>((size_t)2u << 20, stretch_);
        if (look_ > size_) look_ = (size_ + 4095) & ~(size_t)4095;
        slot_syms_ = (uint32_t)(16 * stretch_ + 65536);  // symbols a stretch's slot holds: text up to 16 : 1
        trace_ = getenv("NOHUMAN_TRACE") != nullptr;
        if (dev_set(s_->device_) != hipSuccess) {
            err = "hipSetDevice failed";
            close();
            return -1;
        }
        bool ok = false;
        for (;;) {
            n_slots_ = (uint32_t)(seg_ / stretch_);
            in_bytes_ = seg_ + look_ + ALIGN + 4096;
            stage_bytes_ = in_bytes_ + look_ + ALIGN;  // (the next piece starts up to a block's length behind where it is expected)
            sym_bytes_ = ((size_t)n_slots_ * slot_syms_ + 1024) * 2;
            maps_bytes_ = (size_t)n_slots_ * WSIZE * 2;
            ok = alloc_set(*s_);
            if (ok || seg_ <= ((size_t)16u << 20)) break;
            (void)hipGetLastError();
            free_set(*s_);
            seg_ = (seg_ / 2 + stretch_ - 1) / stretch_ * stretch_;
        }
        if (!ok) {
            (void)hipGetLastError();
            err = "the gzip reader's device buffers cannot be had";
            close();
            return -1;
        }
        // (the second staging buffer: wanted for files of more than one piece; without it pieces are staged as they come)
        // (page-locking half a gigabyte takes 50-80 ms: the helper that stages the second piece's bytes makes the buffer, while
        //  the first piece's kernels run, not open())
        stage_ahead_ = size_ > seg_ && !(getenv("NOHUMAN_GZDEV_STAGE_AHEAD") && getenv("NOHUMAN_GZDEV_STAGE_AHEAD")[0] == '0');
        if (const char *e = getenv("NOHUMAN_GZDEV_FAKE_START")) fake_start_ = atol(e);  // test knob: a false positive of the search
        if (const char *e = getenv("NOHUMAN_GZDEV_FAKE_SPEC")) fake_spec_ = atol(e);
        if (const char *e = getenv("NOHUMAN_GZDEV_FAKE_CRC")) fake_crc_ = atol(e);  // test knob: from the k-th piece on the text's CRC-32 comes out wrong
        open_s_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_open).count();
        return 0;
    }

    // the buffers of one device (sizes as open() settled them), its events, the kernels' LDS limits there
    bool alloc_set(DevSet &d) {
        if (dev_set(d.device_) != hipSuccess) return false;
        d.d_in_ = (uint8_t *)cache_alloc(d.device_, in_bytes_, false);
        d.h_in_ = (uint8_t *)cache_alloc(d.device_, stage_bytes_, true);
        d.d_sym_ = (uint16_t *)cache_alloc(d.device_, sym_bytes_, false);
        d.d_maps_[0] = (uint16_t *)cache_alloc(d.device_, maps_bytes_, false);
        d.d_maps_[1] = (uint16_t *)cache_alloc(d.device_, maps_bytes_, false);
        d.d_windows_ = (uint8_t *)cache_alloc(d.device_, maps_bytes_ / 2, false);
        bool ok = d.d_in_ && d.h_in_ && d.d_sym_ && d.d_maps_[0] && d.d_maps_[1] && d.d_windows_ &&
                  dev_malloc((void **)&d.d_start_, (size_t)n_slots_ * 8) == hipSuccess &&
                  dev_malloc((void **)&d.d_desc_, (size_t)n_slots_ * sizeof(ChunkDesc)) == hipSuccess &&
                  host_malloc((void **)&d.h_desc_, (size_t)n_slots_ * sizeof(ChunkDesc), hipHostMallocDefault) == hipSuccess &&
                  dev_malloc((void **)&d.d_toff_, (size_t)n_slots_ * 8) == hipSuccess &&
                  dev_malloc((void **)&d.d_win_[0], WSIZE) == hipSuccess && dev_malloc((void **)&d.d_win_[1], WSIZE) == hipSuccess &&
                  dev_malloc((void **)&d.d_res_, sizeof(SegResult)) == hipSuccess &&
                  host_malloc((void **)&d.h_res_, sizeof(SegResult), hipHostMallocDefault) == hipSuccess &&
                  (!bgzf_ || host_malloc((void **)&d.h_start_, (size_t)n_slots_ * 8, hipHostMallocDefault) == hipSuccess);
        if (ok) ok = hipMemset(d.d_win_[0], 0, WSIZE) == hipSuccess;  // (the input buffer's tail is zeroed with every piece's upload)
        if (ok)
            ok = hipFuncSetAttribute((const void *)k_inflate3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds3)) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_scan_local, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * WSIZE)) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_scan_groups, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * WSIZE)) == hipSuccess;
        for (auto &e : d.ev_)
            if (ok && !e) ok = hipEventCreate(&e) == hipSuccess;
        return ok;
    }
    void free_set(DevSet &d) {
        if (d.device_ >= 0) (void)dev_set(d.device_);
        cache_free(d.device_, in_bytes_, d.d_in_, false);
        cache_free(d.device_, stage_bytes_, d.h_in_, true);
        cache_free(d.device_, stage_bytes_, d.h_in2_, true);
        cache_free(d.device_, sym_bytes_, d.d_sym_, false);
        cache_free(d.device_, maps_bytes_, d.d_maps_[0], false);
        cache_free(d.device_, maps_bytes_, d.d_maps_[1], false);
        cache_free(d.device_, maps_bytes_ / 2, d.d_windows_, false);
        for (void *p : {(void *)d.d_start_, (void *)d.d_desc_, (void *)d.d_toff_, (void *)d.d_win_[0], (void *)d.d_win_[1], (void *)d.d_res_})
            if (p) (void)hipFree(p);
        for (void *p : {(void *)d.h_desc_, (void *)d.h_res_, (void *)d.h_start_})
            if (p) (void)hipHostFree(p);
        for (auto &e : d.ev_)
            if (e) {
                (void)hipEventDestroy(e);
                e = nullptr;
            }
        const int dev = d.device_;
        d = DevSet();
        d.device_ = dev;
    }
    // one more device (the same one again is allowed: a set of its own) for next_on(); its index, or -1
    int add_device(int device, std::string &err) {
        std::unique_ptr<DevSet> d(new DevSet());
        d->device_ = device;
        if (!alloc_set(*d)) {
            (void)hipGetLastError();
            free_set(*d);
            err = "the gzip reader's device buffers cannot be had on device " + std::to_string(device);
            return -1;
        }
        sets_.push_back(std::move(d));
        return (int)sets_.size() - 1;
    }
    // The next piece is decoded with set k: the window behind the stream so far moves to it
    bool select_set(int k, hipStream_t stream) {
        DevSet *to = sets_[(size_t)k].get();
        if (to == s_) return true;
        if (dev_set(to->device_) != hipSuccess) return false;
        const hipError_t e = dev_copy_between(to->d_win_[to->win_], to->device_, s_->d_win_[s_->win_], s_->device_, WSIZE, stream);
        if (e != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return false;
        s_ = to;
        return true;
    }

    void close() {
        if (trace_ && st_.segments)
            fprintf(stderr,
                    "[nohuman trace] gzip reader on GPU %d, %s: %llu pieces, %llu chunks (%llu decoded again), %llu pieces by the host decoder, "
                    "%llu members, %.2f GB -> %.2f GB; kernels ms: search %.1f decode %.1f windows %.1f resolve %.1f crc %.1f; host: open %.3f s, copy %.3f s, wait %.3f s\n",
                    sets_.empty() ? -1 : sets_[0]->device_, path_.c_str(), (unsigned long long)st_.segments, (unsigned long long)st_.chunks, (unsigned long long)st_.redecoded,
                    (unsigned long long)st_.fallback_segments, (unsigned long long)st_.members, st_.gzip_bytes / 1e9, st_.text_bytes / 1e9, st_.ms_search,
                    st_.ms_decode, st_.ms_scan, st_.ms_resolve, st_.ms_crc, open_s_, st_.s_host_copy, st_.s_wait);
#ifdef NH_GZ_PROF
        if (st_.chunks)
            fprintf(stderr,
                    "[gz prof] per chunk (x 10 ns): header+tables %.0f, lanes' decode %.0f, walk %.0f, expansion %.0f, flush %.0f, all %.0f; per chunk: "
                    "windows %.0f tokens %.0f matches %.0f generic copies %.0f slow tokens %.0f blocks %.2f\n",
                    (double)prof_[0] / st_.chunks, (double)prof_[1] / st_.chunks, (double)prof_[2] / st_.chunks, (double)prof_[3] / st_.chunks,
                    (double)prof_[4] / st_.chunks, (double)prof_[5] / st_.chunks, (double)prof_[6] / st_.chunks, (double)prof_[7] / st_.chunks,
                    (double)prof_[8] / st_.chunks, (double)prof_[9] / st_.chunks, (double)prof_[10] / st_.chunks, (double)prof_[11] / st_.chunks);
#endif
        if (trace_ && (spec_used_ || spec_refused_))
            fprintf(stderr, "[nohuman trace] gzip reader, %s: %llu pieces decoded ahead of the stream taken, %llu refused (decoded again in order)\n",
                    path_.c_str(), (unsigned long long)spec_used_, (unsigned long long)spec_refused_);
        spec_used_ = spec_refused_ = 0;
        if (trace_ && sets_.size() > 1) {
            std::string per;
            for (auto &d : sets_) per += " " + std::to_string(d->device_) + ":" + std::to_string((unsigned long long)d->pieces);
            fprintf(stderr, "[nohuman trace] gzip reader, %s: pieces by device (device:pieces)%s\n", path_.c_str(), per.c_str());
        }
        if (trace_ && host_pieces_)
            fprintf(stderr, "[nohuman trace] gzip reader, %s: %llu pieces inflated by the host's cores (%u workers; stitching, marker replacement and upload %.3f s)\n",
                    path_.c_str(), (unsigned long long)host_pieces_, host_threads_, host_s_);
        host_pieces_ = 0;
        host_s_ = 0;
        if (hc_) hc_->close();
        hc_.reset();
        hc_valid_ = false;
        if (h_text_ && !sets_.empty()) cache_free(sets_[0]->device_, h_text_cap_, h_text_, true);
        h_text_ = nullptr;
        h_text_cap_ = 0;
        if (pf_.th.joinable()) pf_.th.join();
        pf_.valid = false;
        if (trace_ && (pf_hits_ || pf_misses_))
            fprintf(stderr, "[nohuman trace] gzip reader, %s: %llu pieces found their bytes staged ahead, %llu were staged as they came\n", path_.c_str(),
                    (unsigned long long)pf_hits_, (unsigned long long)pf_misses_);
        pf_hits_ = pf_misses_ = 0;
        for (auto &d : sets_) free_set(*d);
        sets_.clear();
        s_ = nullptr;
        if (base_) munmap((void *)base_, size_);
        base_ = nullptr;
        if (fd_ >= 0) ::close(fd_);
        fd_ = -1;
        st_ = DevGunzipStats();
        ended_ = false;
    }

    int n_sets() const { return (int)sets_.size(); }
    uint64_t pieces_of(int k) const { return sets_[(size_t)k]->pieces; }
    long next_on(int k, void *d_dst, size_t room, hipStream_t stream) {
        if (!error_.empty()) return -1;
        if (k < 0 || k >= (int)sets_.size()) return fail("no such device set");
        if (!select_set(k, stream)) return fail("moving the window between devices failed");
        return next(d_dst, room, stream);
    }

    long next(void *d_dst, size_t room, hipStream_t stream) {
        if (!error_.empty()) return -1;
        for (;;) {
            if (ended_) return 0;
            const long n = piece((uint8_t *)d_dst, room, stream);
            if (n != 0) return n;  // text, or an error
        }
    }

    bool ended_ = false;
    std::string error_;
    DevGunzipStats st_;

private:
    static constexpr size_t ALIGN = 4096;
    static constexpr size_t SEARCH3_LDS = sizeof(Lds3) > 64 * CLROW ? sizeof(Lds3) : 64 * CLROW;
    typedef void (*InflateFn)(const uint32_t *, uint64_t, uint32_t, ChunkDesc *, uint16_t *, uint32_t, uint32_t);
    InflateFn inflate_kernel() const { return k_inflate3; }
    size_t inflate_lds() const { return sizeof(Lds3); }

    long fail(const std::string &m, bool integrity = false) {
        if (error_.empty()) {
            error_ = "gzip: " + m + " (" + path_ + ")";
            integrity_ = integrity;
        }
        return -1;
    }

    // per-member bookkeeping like gzip's: CRC-32 and length of what was decoded against the trailer
    bool account(uint32_t crc, uint64_t len) {
        if (fake_crc_ > 0 && st_.segments + 1 >= (uint64_t)fake_crc_ && len) crc ^= 1u;  // test knob: a wrong decode only the CRC sees
        run_crc_ = crc32_join(run_crc_, crc, len);  // (crc of nothing is 0, and joining to it changes nothing)
        run_len_ += len;
        return true;
    }
    bool member_end(uint32_t crc, uint32_t isize) {
        {
            std::lock_guard<std::mutex> lk(st_mu_);
            st_.members++;
        }
        if (crc != run_crc_ || isize != (uint32_t)run_len_) {
            static const bool go_on = getenv("NOHUMAN_GZDEV_NOCRC") != nullptr;  // debugging aid: write the text anyway, say where
            if (go_on) {
                fprintf(stderr, "[gzdev] member %llu: crc %08x / %08x, length %u / %u (text so far %llu)\n", (unsigned long long)st_.members, run_crc_, crc,
                        (uint32_t)run_len_, isize, (unsigned long long)st_.text_bytes);
            } else {
                fail(crc != run_crc_ ? "crc error" : "length error", true);
                return false;
            }
        }
        run_crc_ = 0;
        run_len_ = 0;
        return true;
    }

    // ---- a piece in two phases.  A: upload, block search, decode to symbols, chain check (with the re-decodes it asks for) --
    // everything that needs neither the window nor the stream's state, so a piece AHEAD of the stream can run it (first_bit
    // NONE: no known start; the reader over several devices decodes the pieces of the grid concurrently this way).  B:
    // windows, text, CRCs, the window behind the piece, the member bookkeeping -- in stream order.
    // BGZF: the member whose header starts at byte h of the file: its whole size (0: no BGZF member there) and where its data begin
    uint64_t bgzf_block_size(uint64_t h, uint64_t *data_at = nullptr) const {
        if (h + 18 > size_) return 0;
        const uint8_t *q = base_ + h;
        if (q[0] != 0x1f || q[1] != 0x8b || q[2] != 8 || !(q[3] & 4) || (q[3] & 0xE0)) return 0;
        const uint32_t xlen = q[10] | ((uint32_t)q[11] << 8);
        if (h + 12 + xlen > size_) return 0;
        uint64_t bsize = 0;
        for (uint32_t o = 0; o + 4 <= xlen;) {  // the subfields: SI1 SI2 LEN(2) data
            const uint8_t *f = q + 12 + o;
            const uint32_t len = f[2] | ((uint32_t)f[3] << 8);
            if (f[0] == 'B' && f[1] == 'C' && len == 2 && o + 6 <= xlen) bsize = (uint64_t)(f[4] | ((uint32_t)f[5] << 8)) + 1;
            o += 4 + len;
        }
        if (q[3] & (8 | 16 | 2)) return 0;  // (a name, a comment or a header CRC: bgzip writes none; such a file takes the search's way)
        if (bsize < 12 + xlen + 8 || h + bsize > size_) return 0;
        if (data_at) *data_at = h + 12 + xlen;
        return bsize;
    }
    // the chunks' starts of a piece from the members' headers: chunk c starts at the first member whose data begin in stretch c
    bool bgzf_starts(DevSet &d, const PieceJob &j, uint32_t n_str) {
        for (uint32_t c = 0; c < n_str; c++) d.h_start_[c] = NONE;
        d.h_start_[0] = j.first_bit;
        // the member the stream stands at: walk on from the last one known
        uint64_t h = bgzf_hdr_;
        const uint64_t pos_byte = j.a_byte + (j.first_bit >> 3);
        for (;;) {
            uint64_t at = 0;
            const uint64_t bs = bgzf_block_size(h, &at);
            if (!bs) return false;
            if (at == pos_byte) break;
            if (at > pos_byte) return false;
            h += bs;
        }
        bgzf_hdr_ = h;
        const uint64_t end_byte = j.a_byte + (uint64_t)n_str * stretch_;
        for (;;) {
            uint64_t at = 0;
            const uint64_t bs = bgzf_block_size(h, &at);
            if (!bs || at >= end_byte) break;
            const uint64_t bit = 8 * (at - j.a_byte);
            const uint32_t c = (uint32_t)(bit / (8 * (uint64_t)stretch_));
            if (c > 0 && c < n_str && d.h_start_[c] == NONE && bit > j.first_bit) d.h_start_[c] = bit;
            h += bs;
        }
        return true;
    }

    long phase_a(DevSet &d, PieceJob &j, hipStream_t stream, bool spec) {
        auto bad = [&](const std::string &m) -> long { return spec ? -1 : fail(m); };
#define GZA_TRY(x)                                                                      \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) return bad(std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)
        j.valid = false;
        dev_check(d.device_, "gzip reader, phase A of a piece");
        dev_check_ptr(d.d_in_, d.device_, "gzip reader, phase A (input buffer)");
        const size_t want = (size_t)j.n * stretch_ + look_;
        const size_t avail = (size_t)std::min<uint64_t>(want, size_ - j.a_byte);
        j.at_eof = j.a_byte + avail == size_;
        const uint32_t n_str = (uint32_t)std::min<uint64_t>(j.n, (avail + stretch_ - 1) / stretch_);
        j.n_str = n_str;
        j.valid_bits = 8ull * avail;
        const uint64_t valid_bits = j.valid_bits;
        // the bytes: page cache -> page-locked staging -> device
        const auto c0 = std::chrono::steady_clock::now();
        const uint8_t *src = nullptr;
        uint8_t *stage = d.h_in_;  // the buffer this piece's bytes are uploaded from
        if (!spec && pf_.valid) {
            if (pf_.th.joinable()) pf_.th.join();
            pf_.valid = false;
            if (pf_.buf && j.a_byte >= pf_.start && j.a_byte + avail <= pf_.start + pf_.len) {
                src = pf_.buf + (j.a_byte - pf_.start);
                stage = pf_.buf;
                pf_hits_++;
            } else {
                pf_misses_++;  // (a piece cut down, the host decoder's turn, a block longer than the look-ahead: copied now)
                stage = (pf_.buf == d.h_in_ || !d.h_in2_) ? (d.h_in2_ ? d.h_in2_ : d.h_in_) : d.h_in_;  // (the other one where there are two)
            }
        }
        if (!src) {
            memcpy(stage, base_ + j.a_byte, avail);
            src = stage;
        }
        const double copy_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
        GZA_TRY(hipMemcpyAsync(d.d_in_, src, avail, hipMemcpyHostToDevice, stream));
        GZA_TRY(hipMemsetAsync(d.d_in_ + avail, 0, 1024, stream));  // (what a kernel reads beyond the valid bits is zeros)
        // (not with a host lane: there the ONE set also decodes cells ahead, on its lane's thread, and stages their bytes through h_in_
        //  -- the very buffer this helper may be filling; found by the soak as "invalid deflate data" in the piece behind a host piece)
        if (!spec && sets_.size() == 1 && stage_ahead_ && !host_cell_ && !j.at_eof) {
            // where the next piece is expected: at the end of this one's last stretch (it starts at the first block boundary at
            // or behind it: up to a block further on, which the buffer's slack covers)
            pf_.start = (j.a_byte + (uint64_t)n_str * stretch_) / ALIGN * ALIGN;
            pf_.len = pf_.start < size_ ? (size_t)std::min<uint64_t>(stage_bytes_, size_ - pf_.start) : 0;
            if (pf_.len) {
                const uint8_t *from = base_ + pf_.start;
                const size_t len = pf_.len;
                const bool into_second = stage == d.h_in_;
                DevSet *dp = &d;
                const size_t bytes = stage_bytes_;
                pf_.buf = nullptr;  // (the helper says: the second buffer is made by its first use)
                pf_.th = std::thread([this, dp, from, len, into_second, bytes] {
                    if (into_second && !dp->h_in2_) dp->h_in2_ = (uint8_t *)cache_alloc(dp->device_, bytes, true);
                    uint8_t *to = into_second ? dp->h_in2_ : dp->h_in_;
                    if (to) memcpy(to, from, len);
                    pf_.buf = to;
                });
                pf_.valid = true;
            }
        }
        uint64_t end_bit = std::min<uint64_t>((uint64_t)n_str * stretch_ * 8, valid_bits);
        if (j.limit_bits && j.limit_bits < end_bit) end_bit = j.limit_bits;
        j.end_bit = end_bit;
        if (trace_) (void)hipEventRecord(d.ev_[0], stream);
        bool by_headers = false;
        if (bgzf_ && !spec && d.h_start_ && j.first_bit != NONE) {
            by_headers = bgzf_starts(d, j, n_str);
            j.by_headers = by_headers;
            if (by_headers) GZA_TRY(hipMemcpyAsync(d.d_start_, d.h_start_, (size_t)n_str * 8, hipMemcpyHostToDevice, stream));
            else bgzf_ = false;  // (the headers do not go on as they began: an ordinary gzip stream from here on)
        }
        if (!by_headers)
            hipLaunchKernelGGL(k_search3, dim3(n_str), dim3(64), SEARCH3_LDS, stream, (const uint32_t *)d.d_in_, valid_bits,
                               (uint64_t)stretch_ * 8, j.first_bit, d.d_start_);
        long fake_end = -1;
        if (!spec && fake_start_ > 0 && (uint32_t)fake_start_ < n_str) {  // test knob: pretend the search found a start that is none
            const uint64_t bogus = (uint64_t)fake_start_ * stretch_ * 8 + 13;
            GZA_TRY(hipMemcpyAsync(d.d_start_ + fake_start_, &bogus, 8, hipMemcpyHostToDevice, stream));
            GZA_TRY(hipStreamSynchronize(stream));
            if (getenv("NOHUMAN_GZDEV_FAKE_END")) fake_end = fake_start_;  // ... whose garbage even looks like the end of the stream
            fake_start_ = -1;
        }
        if (spec && fake_spec_ > 0) {  // test knob: the first start of a piece decoded ahead is no block start (the piece is refused)
            if (--fake_spec_ == 0) {
                const uint64_t bogus = 13;
                GZA_TRY(hipMemcpyAsync(d.d_start_, &bogus, 8, hipMemcpyHostToDevice, stream));
                GZA_TRY(hipStreamSynchronize(stream));
            }
        }
        hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, stream, (const uint64_t *)d.d_start_, n_str, end_bit, slot_syms_, d.d_desc_);
        if (trace_) (void)hipEventRecord(d.ev_[1], stream);
        const uint32_t kflags = (j.at_eof ? 1u : 0u) | (by_headers ? 2u : 0u);
        hipLaunchKernelGGL(inflate_kernel(), dim3(n_str), dim3(64), inflate_lds(), stream, (const uint32_t *)d.d_in_, valid_bits,
                           kflags, d.d_desc_, d.d_sym_, slot_syms_, NOIDX);
        if (fake_end >= 0) hipLaunchKernelGGL(k_test_mark_end, dim3(1), dim3(1), 0, stream, d.d_desc_, (uint32_t)fake_end);
        if (trace_) (void)hipEventRecord(d.ev_[2], stream);
        uint32_t redo = 0;
        double wait_s = 0;
        for (;;) {
            hipLaunchKernelGGL(k_finish, dim3(1), dim3(1024), 0, stream, d.d_desc_, n_str, d.d_toff_, d.d_res_);
            GZA_TRY(hipMemcpyAsync(d.h_res_, d.d_res_, sizeof(SegResult), hipMemcpyDeviceToHost, stream));
            const auto w0 = std::chrono::steady_clock::now();
            GZA_TRY(hipStreamSynchronize(stream));
            wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
            if (d.h_res_->bad_chunk != NOIDX && (d.h_res_->broken == NOIDX || d.h_res_->bad_chunk < d.h_res_->broken)) break;
            if (d.h_res_->broken == NOIDX) break;
            // A chunk does not start where its predecessor ended: what the search found there was no block start.
            // It is struck from the plan and its predecessor decoded again, now to the start after it (the host
            // reader's "searching on"; costs time, never correctness).
            const uint32_t badc = d.h_res_->broken;
            GZA_TRY(hipMemcpyAsync(d.d_start_ + badc, &NONE, 8, hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, stream, (const uint64_t *)d.d_start_, n_str, end_bit, slot_syms_, d.d_desc_);
            GZA_TRY(hipMemcpyAsync(d.h_desc_, d.d_desc_, (size_t)n_str * sizeof(ChunkDesc), hipMemcpyDeviceToHost, stream));
            GZA_TRY(hipStreamSynchronize(stream));
            uint32_t prev = badc;
            while (prev > 0 && d.h_desc_[--prev].bit_start == NONE) {
            }
            if (d.h_desc_[prev].bit_start == NONE) return bad("the chunks of a piece do not chain");  // (the struck one was the piece's first)
            hipLaunchKernelGGL(inflate_kernel(), dim3(1), dim3(64), inflate_lds(), stream, (const uint32_t *)d.d_in_, valid_bits,
                               kflags, d.d_desc_, d.d_sym_, slot_syms_, prev);
            redo++;
            if (redo > n_str) return bad("the chunks of a piece do not chain");
        }
        j.r = *d.h_res_;
        j.redo = redo;
        {
            std::lock_guard<std::mutex> lk(st_mu_);
            st_.s_host_copy += copy_s;
            st_.s_wait += wait_s;
            st_.redecoded += redo;
            if (trace_) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, d.ev_[0], d.ev_[1]) == hipSuccess) st_.ms_search += ms;
                if (hipEventElapsedTime(&ms, d.ev_[1], d.ev_[2]) == hipSuccess) st_.ms_decode += ms;
            }
        }
        j.valid = true;
        return 0;
#undef GZA_TRY
    }

    // phase B of a piece whose phase A went through (no bad chunk, the text fits the room)
    long phase_b(DevSet &d, const PieceJob &j, uint8_t *d_dst, size_t room, hipStream_t stream) {
        const SegResult &r = j.r;
        const uint32_t n_str = j.n_str;
        const uint64_t a_byte = j.a_byte;
        dev_check(d.device_, "gzip reader, phase B of a piece");
        dev_check_ptr(d_dst, d.device_, "gzip reader, phase B (text buffer)");
        dev_check_ptr(d.d_sym_, d.device_, "gzip reader, phase B (symbols)");
        // windows by the prefix scan, text, CRCs, the window behind the piece
        if (trace_) (void)hipEventRecord(d.ev_[3], stream);
        const uint32_t gy = 4;
        if (j.by_headers) {
            // (every chunk starts at a member's first block: it has no window and leaves no markers)
        } else {
            const uint32_t ng = (n_str + SCAN_GROUP - 1) / SCAN_GROUP;
            hipLaunchKernelGGL(k_scan_local, dim3(ng), dim3(1024), 2 * WSIZE, stream, (const ChunkDesc *)d.d_desc_, (const uint16_t *)d.d_sym_, slot_syms_,
                               n_str, d.d_maps_[0]);
            hipLaunchKernelGGL(k_scan_groups, dim3(1), dim3(1024), 2 * WSIZE, stream, (const uint16_t *)d.d_maps_[0], n_str, d.d_maps_[1]);
            hipLaunchKernelGGL(k_scan_windows, dim3(n_str, gy), dim3(256), 0, stream, (const uint16_t *)d.d_maps_[0], (const uint16_t *)d.d_maps_[1],
                               (const uint8_t *)d.d_win_[d.win_], d.d_windows_);
        }
        if (trace_) (void)hipEventRecord(d.ev_[4], stream);
        hipLaunchKernelGGL(k_resolve, dim3(n_str, 8), dim3(256), 0, stream, (const ChunkDesc *)d.d_desc_, (const uint16_t *)d.d_sym_, slot_syms_,
                           (const uint8_t *)d.d_windows_, (const uint64_t *)d.d_toff_, d_dst);
        if (trace_) (void)hipEventRecord(d.ev_[5], stream);
        hipLaunchKernelGGL(k_crc, dim3(n_str), dim3(256), 0, stream, d.d_desc_, (const uint64_t *)d.d_toff_, (const uint8_t *)d_dst);
        hipLaunchKernelGGL(k_carry_window, dim3(1), dim3(1024), 0, stream, (const uint8_t *)d_dst, (uint64_t)room, (const SegResult *)d.d_res_,
                           (const uint8_t *)d.d_win_[d.win_], d.d_win_[d.win_ ^ 1]);
        if (trace_) (void)hipEventRecord(d.ev_[6], stream);
        GZ_TRY(hipMemcpyAsync(d.h_desc_, d.d_desc_, (size_t)n_str * sizeof(ChunkDesc), hipMemcpyDeviceToHost, stream));
        const auto w0 = std::chrono::steady_clock::now();
        GZ_TRY(hipStreamSynchronize(stream));
        {
            std::lock_guard<std::mutex> lk(st_mu_);
            st_.s_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
            if (trace_) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, d.ev_[3], d.ev_[4]) == hipSuccess) st_.ms_scan += ms;
                if (hipEventElapsedTime(&ms, d.ev_[4], d.ev_[5]) == hipSuccess) st_.ms_resolve += ms;
                if (hipEventElapsedTime(&ms, d.ev_[5], d.ev_[6]) == hipSuccess) st_.ms_crc += ms;
            }
        }
        d.win_ ^= 1;
        // members: CRC-32 and ISIZE like gzip checks them
        for (uint32_t c = 0; c <= r.end_chunk; c++) {
            const ChunkDesc &cd = d.h_desc_[c];
            if (cd.bit_start == NONE) continue;
#ifdef NH_GZ_PROF
            for (int i = 0; i < 12; i++) prof_[i] += cd.prof[i];
#endif
            uint32_t a = 0;
            for (uint32_t p = 0; p <= cd.n_members; p++) {
                const uint32_t b = p < cd.n_members ? cd.m_off[p] : cd.out_len;
                account(cd.piece_crc[p], b - a);
                if (p < cd.n_members && !member_end(cd.m_crc[p], cd.m_isize[p])) return -1;
                a = b;
            }
        }
        const uint64_t first_bit = pos_bit_ - 8 * a_byte;
        if (debug_)
            fprintf(stderr, "[gzdev] piece %llu: file byte %llu + bit %llu, %u stretches, %u chunks (last %u), %u decoded again, text %llu at %llu, ends at bit %llu%s%s\n",
                    (unsigned long long)st_.segments, (unsigned long long)a_byte, (unsigned long long)first_bit, n_str, r.n_chunks, r.end_chunk, j.redo,
                    (unsigned long long)r.total, (unsigned long long)st_.text_bytes, (unsigned long long)r.end_bit, r.stream_end ? " (end of stream)" : "",
                    j.first_bit == NONE ? " (decoded ahead)" : "");
        {
            std::lock_guard<std::mutex> lk(st_mu_);
            st_.segments++;
            d.pieces++;
            st_.chunks += r.n_chunks;
            st_.text_bytes += r.total;
            st_.gzip_bytes += (a_byte * 8 + r.end_bit - pos_bit_) / 8;
        }
        if (r.total) ratio_ = std::max(ratio_ * 0.9, (double)r.total / std::max<double>(1.0, (double)(r.end_bit - first_bit) / 8));
        pos_bit_ = a_byte * 8 + r.end_bit;
        if (r.stream_end) {
            if (run_len_) return fail("unexpected end of file", true);
            ended_ = true;
        } else if ((pos_bit_ >> 3) >= size_) {
            return fail("unexpected end of file", true);
        }
        return (long)r.total;
    }

    // one piece of the stream, both phases on the current set: up to n_slots_ stretches from the position the stream goes on
    // at (limit_byte: but no further than the first block boundary at or behind that byte of the file -- a boundary of the
    // piece grid, see DevFastqReader over several devices)
    long piece(uint8_t *d_dst, size_t room, hipStream_t stream, uint64_t limit_byte = 0) {
        if (dev_set(s_->device_) != hipSuccess) return fail("hipSetDevice failed");
        const uint64_t a_byte = (pos_bit_ >> 3) / ALIGN * ALIGN;  // the piece's buffer starts here in the file
        const uint64_t first_bit = pos_bit_ - 8 * a_byte;
        if (a_byte >= size_) return fail("unexpected end of file");
        if (host_mode_) {
            // the device gave up on this file twice in a row (text beyond 16 : 1, many small members, giant blocks): the host
            // decoder reads on -- for a stint of 16 MiB steps, then the device is tried again (one more refusal doubles the
            // next stint, up to 64 steps = 1 GiB of gzip): a stretch of odd data does not put the rest of the file on one core
            SegResult r{};
            r.bad_status = last_bad_;
            if (host_left_ == 0) {
                host_mode_ = false;
                failed_in_a_row_ = 1;  // (one refusal is enough to come back here)
            } else {
                host_left_--;
                return host_piece(d_dst, room, stream, a_byte, first_bit, 8ull * (size_ - a_byte), r);
            }
        }
        // stretches this piece decodes: what the room holds at the ratio seen so far (a piece that does not fit is cut down).
        // (Smaller first pieces, to give the pipeline its first batches sooner, were measured and dropped twice: round 4, an
        // eighth and a quarter of a piece first cost 0.1 s of a 1.7 s run -- small pieces leave the chip's wave slots empty;
        // round 5, an eighth / a quarter / a half first, 20 M pairs: gzip out 1.389 -> 1.409 s, plain out 1.753 -> 1.797,
        // nothing kept 1.170 -> 1.264 -- the run is bound by the GPU's codec kernels, not by when its first batch flows.)
        uint32_t n = n_slots_;
        {
            const double per_stretch = ratio_ * 1.3 * (double)stretch_ + 4096;
            const uint64_t fit = (uint64_t)((double)room / per_stretch);
            if (fit < n) n = fit < 1 ? 1u : (uint32_t)fit;
        }
        for (;;) {
            PieceJob j;
            j.a_byte = a_byte;
            j.first_bit = first_bit;
            j.n = n;
            if (limit_byte > a_byte) {
                j.limit_bits = 8 * (limit_byte - a_byte);
                if (j.limit_bits <= first_bit) j.limit_bits = 0;  // (cannot be: the limit is a boundary behind the position)
                const uint64_t need = (limit_byte - a_byte + stretch_ - 1) / stretch_;
                if (need < j.n) j.n = (uint32_t)need;
            }
            if (phase_a(*s_, j, stream, false) != 0) return -1;
            const SegResult &r = j.r;
            if (r.bad_chunk != NOIDX || r.end_chunk == NOIDX) {
                last_bad_ = r.bad_status;
                if (++failed_in_a_row_ >= 2) {
                    host_mode_ = true;
                    host_left_ = host_stint_;
                    host_stint_ = std::min<uint32_t>(64, host_stint_ * 2);
                }
                return host_piece(d_dst, room, stream, a_byte, first_bit, j.end_bit, r);
            }
            if (failed_in_a_row_ == 0) host_stint_ = 8;  // (two clean pieces in a row: the odd stretch is behind)
            failed_in_a_row_ = 0;
            if (r.total > room) {
                if (j.n_str <= 1) return fail("a chunk's text does not fit the batch buffer");
                n = std::max<uint32_t>(1u, j.n_str / 4);  // cut the piece down and decode again
                ratio_ = std::max(ratio_, (double)r.total / ((double)j.n_str * stretch_));
                continue;
            }
            return phase_b(*s_, j, d_dst, room, stream);
        }
    }

    // ---- pieces decoded ahead of the stream (several devices).  The file is a grid of cells of n_slots_ stretches from the
    // first member's first block; cell k's piece starts at the first block start the search finds in it and ends at the first
    // block boundary behind it.  spec_decode() runs phase A of a cell on set k -- any thread, while the stream is elsewhere;
    // take() is the stream's next piece ON set k, in stream order: the set's decoded cell if it starts exactly where the
    // stream stands (the normal case), else a piece decoded now, from the stream's position to the end of ITS cell -- which
    // puts the stream back on the grid.
    // (With a host lane -- the hybrid reader, below -- the grid ALTERNATES: even cells of a piece's size for the devices, odd cells
    //  of host_cell_ bytes, a quarter of that, for the host's workers: a host cell must cost less time than the device cell
    //  before it, or the chip waits for it.  host_cell_ == 0: the uniform grid of rounds 4-5.)
    uint64_t cell_bytes() const { return (uint64_t)n_slots_ * stretch_; }
    uint64_t pair_bytes() const { return cell_bytes() + host_cell_; }
    uint64_t cell_lo(uint64_t c) const {
        return host_cell_ ? grid0_ + (c / 2) * pair_bytes() + ((c & 1) ? cell_bytes() : 0) : grid0_ + c * cell_bytes();
    }
    uint64_t cell_hi(uint64_t c) const { return cell_lo(c) + (host_cell_ && (c & 1) ? host_cell_ : cell_bytes()); }
    uint64_t cell_of_pos() const {
        const uint64_t off = (pos_bit_ >> 3) - grid0_;
        if (!host_cell_) return off / cell_bytes();
        return 2 * (off / pair_bytes()) + (off % pair_bytes() >= cell_bytes() ? 1 : 0);
    }
    uint64_t cells() const {
        const uint64_t total = size_ - grid0_;
        if (!host_cell_) return (total + cell_bytes() - 1) / cell_bytes();
        const uint64_t r = total % pair_bytes();
        return 2 * (total / pair_bytes()) + (r > 0 ? 1 : 0) + (r > cell_bytes() ? 1 : 0);
    }
    bool spec_ok() const { return !bgzf_ && stretch_ % ALIGN == 0 && !host_mode_; }
    void spec_decode(int k, uint64_t cell, hipStream_t stream) {
        DevSet &d = *sets_[(size_t)k];
        d.spec = PieceJob();
        d.spec_cell = cell;
        if (dev_set(d.device_) != hipSuccess) return;
        d.spec.a_byte = cell_lo(cell);
        if (d.spec.a_byte >= size_) return;
        d.spec.first_bit = NONE;
        d.spec.n = (uint32_t)((cell_hi(cell) - cell_lo(cell)) / stretch_);
        if (phase_a(d, d.spec, stream, true) != 0) d.spec.valid = false;
    }
    long take(int k, bool use_spec, void *d_dst, size_t room, hipStream_t stream) {
        if (!error_.empty()) return -1;
        if (ended_) return 0;
        if (k < 0 || k >= (int)sets_.size()) return fail("no such device set");
        if (!select_set(k, stream)) return fail("moving the window between devices failed");
        DevSet &d = *s_;
        const PieceJob &j = d.spec;
        if (use_spec && j.valid && !host_mode_ && j.r.bad_chunk == NOIDX && j.r.end_chunk != NOIDX && j.r.first_start != NONE &&
            j.a_byte * 8 + j.r.first_start == pos_bit_ && j.r.total <= room) {
            failed_in_a_row_ = 0;
            spec_used_++;
            const long n = phase_b(d, j, (uint8_t *)d_dst, room, stream);
            d.spec.valid = false;
            if (n != 0) return n;
            // (a piece without text -- an empty member --: the stream goes on below)
        } else if (use_spec) {
            spec_refused_++;
            if (debug_)
                fprintf(stderr, "[gzdev] cell %llu decoded ahead is refused: valid %d, bad chunk %u (status %u), last chunk %u, first start bit %llu of the file, the stream stands at %llu, text %llu of %zu\n",
                        (unsigned long long)d.spec_cell, (int)j.valid, j.r.bad_chunk, j.r.bad_status, j.r.end_chunk,
                        (unsigned long long)(j.a_byte * 8 + j.r.first_start), (unsigned long long)pos_bit_, (unsigned long long)j.r.total, room);
        }
        d.spec.valid = false;
        for (;;) {
            if (ended_) return 0;
            const uint64_t limit = cell_hi(cell_of_pos());
            const long n = piece((uint8_t *)d_dst, room, stream, limit < size_ ? limit : 0);
            if (n != 0) return n;
        }
    }

    // ---- the HYBRID reader's host lane (round 6).  With the text re-encoded on the GPU the chip's codec kernels are the run's
    // bottleneck and the host's cores idle: some cells of the piece grid are inflated there.  host_decode_ahead(cell): every
    // chunk of the cell decoded speculatively by `host_threads_` workers (RangeGunzip, nh_inflate.cpp: the host decoder's own
    // chunk machinery on a byte range) -- any thread, needs neither the stream's position nor its window.  take_host(): the
    // stream's next piece, in stream order: the chunks stitched from the stream's position with the window behind the piece
    // before (fetched from the device), markers replaced and CRCs taken on the workers, the text uploaded to where the
    // GPU's resolve kernel would have put it, the window behind it left on the device -- the piece ends at the first block
    // boundary at or behind its cell that passes the host's block-header test, which is where the GPU's search starts the next
    // cell (a cell decoded ahead on the GPU whose first start is elsewhere is refused and decoded in order: correctness
    // never depends on the two searches agreeing).
    bool host_ok(size_t room) const { return host_threads_ > 0 && host_cell_ > 0 && spec_ok() && ratio_ * 1.1 * (double)host_cell_ <= (double)room; }
    // (before the first piece: the grid's shape is fixed for the file)
    void set_host(unsigned threads) {
        host_threads_ = threads;
        host_cell_ = 0;
        if (!threads || !spec_ok()) return;
        uint64_t hc = cell_bytes() / 4;
        if (const char *e = getenv("NOHUMAN_GZ_HYBRID_CELL")) hc = (uint64_t)atoll(e);  // tuning / test knob: bytes of gzip a host cell
        hc = hc / stretch_ * stretch_;
        if (hc < stretch_) hc = stretch_;
        if (hc > cell_bytes()) hc = cell_bytes();
        host_cell_ = hc;
    }
    void host_decode_ahead(uint64_t cell) {
        hc_valid_ = false;
        if (!hc_) hc_.reset(new RangeGunzip());
        const uint64_t lo = cell_lo(cell), hi = cell_hi(cell);
        if (lo >= size_) return;
        // (chunks of 4 MiB like the host reader's; the tests' tiny cells are cut into eight)
        const size_t chunk = (size_t)std::min<uint64_t>((uint64_t)4u << 20, std::max<uint64_t>((hi - lo) / 8, 4096));
        if (hc_->start(base_, size_, lo, hi, host_threads_, chunk) != 0) return;
        hc_->wait_speculated();
        hc_cell_ = cell;
        hc_valid_ = true;
    }
    long take_host(bool use_spec, void *d_dst, size_t room, hipStream_t stream) {
        if (!error_.empty()) return -1;
        if (sets_.empty() || !select_set(0, stream)) return fail("moving the window between devices failed");
        dev_check(s_->device_, "gzip reader, a piece by the host's cores");
        for (;;) {
            if (ended_) return 0;
            const uint64_t cell = cell_of_pos();
            if (!(use_spec && hc_valid_ && hc_cell_ == cell)) host_decode_ahead(cell);  // (nobody decoded it ahead: now, in order)
            use_spec = false;
            if (!hc_valid_) return fail("the host decoder could not be started");
            hc_valid_ = false;
            // (page-locked room for the cell's text: eight times its gzip bytes, more where the file has been seen to inflate further)
            const size_t want = (size_t)std::min<double>((double)room, std::max(8.0, ratio_ * 2.0) * (double)(cell_hi(cell) - cell_lo(cell)) + (double)(1u << 20));
            if (!h_text_ || h_text_cap_ < want) {
                if (h_text_) cache_free(s_->device_, h_text_cap_, h_text_, true);
                h_text_cap_ = 0;
                h_text_ = (uint8_t *)cache_alloc(s_->device_, want, true);
                if (!h_text_) return fail("the host lane's page-locked text buffer cannot be had");
                h_text_cap_ = want;
            }
            uint8_t window[WSIZE], wafter[WSIZE];
            GZ_TRY(hipMemcpyAsync(window, s_->d_win_[s_->win_], WSIZE, hipMemcpyDeviceToHost, stream));
            GZ_TRY(hipStreamSynchronize(stream));
            uint64_t eb = 0;
            bool stream_end = false;
            std::vector<GzSeg> segs;
            const auto t0 = std::chrono::steady_clock::now();
            const long n = hc_->finish(pos_bit_, window, h_text_, h_text_cap_, &eb, &stream_end, wafter, segs);
            if (n < 0) {
                error_ = hc_->error() + " (" + path_ + ")";
                hc_->close();
                return -1;
            }
            if (debug_) {
                uint64_t sa = 0, sr = 0, sg = 0;
                hc_->stats(&sa, &sr, &sg);
                fprintf(stderr, "[gzdev] host piece: cell %llu [%llu, %llu), the stream at bit %llu, text %ld at %llu, ends at bit %llu%s; chunks accepted %llu rejected %llu, %llu bytes decoded in order\n",
                        (unsigned long long)cell, (unsigned long long)cell_lo(cell), (unsigned long long)cell_hi(cell), (unsigned long long)pos_bit_, n,
                        (unsigned long long)st_.text_bytes, (unsigned long long)eb, stream_end ? " (end of stream)" : "", (unsigned long long)sa, (unsigned long long)sr,
                        (unsigned long long)sg);
            }
            hc_->close();
            if (n > 0) GZ_TRY(hipMemcpyAsync(d_dst, h_text_, (size_t)n, hipMemcpyHostToDevice, stream));
            GZ_TRY(hipMemcpyAsync(s_->d_win_[s_->win_ ^ 1], wafter, WSIZE, hipMemcpyHostToDevice, stream));
            GZ_TRY(hipStreamSynchronize(stream));
            s_->win_ ^= 1;
            for (const GzSeg &sg : segs) {
                account(sg.crc, sg.len);
                if (sg.member_end && !member_end(sg.want_crc, sg.want_isize)) return -1;
            }
            {
                std::lock_guard<std::mutex> lk(st_mu_);
                st_.segments++;
                host_pieces_++;
                st_.text_bytes += (uint64_t)n;
                st_.gzip_bytes += (eb - pos_bit_) / 8;
                host_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            if (n > 0) ratio_ = std::max(ratio_ * 0.9, (double)n / std::max<double>(1.0, (double)(eb - pos_bit_) / 8));
            if (!stream_end && eb <= pos_bit_) return fail("the host lane did not advance");
            pos_bit_ = eb;
            if (stream_end) {
                if (run_len_) return fail("unexpected end of file", true);
                ended_ = true;
            } else if ((pos_bit_ >> 3) >= size_) {
                return fail("unexpected end of file", true);
            }
            if (n != 0) return n;
        }
    }

    // The host decoder takes the piece over: from the same bit position, with the same window, to the first block
    // boundary behind the piece's first stretch (it is one core: the pieces behind go back to the device).
    long host_piece(uint8_t *d_dst, size_t room, hipStream_t stream, uint64_t a_byte, uint64_t first_bit, uint64_t end_bit, const SegResult &r) {
        if (!warned_) {
            fprintf(stderr,
                    "nohuman: %s: a piece of the gzip stream is decoded on the host (chunk status %u: %s); the GPU reader goes on behind it\n",
                    path_.c_str(), r.bad_status,
                    r.bad_status == ST_ROOM      ? "text beyond 16:1"
                    : r.bad_status == ST_INPUT   ? "a block longer than the look-ahead, or a truncated file"
                    : r.bad_status == ST_MEMBERS ? "many small members"
                                                 : "invalid deflate data?");
            warned_ = true;
        }
        std::vector<uint8_t> window(WSIZE);
        dev_check(s_->device_, "gzip reader, a piece by the host decoder");
        GZ_TRY(hipMemcpyAsync(window.data(), s_->d_win_[s_->win_], WSIZE, hipMemcpyDeviceToHost, stream));
        GZ_TRY(hipStreamSynchronize(stream));
        // to the first block boundary behind a megabyte of input (in host mode: sixteen), or with half the room full
        std::vector<uint8_t> out;
        std::vector<GzMemberEnd> members;
        const uint64_t from = a_byte * 8 + first_bit;
        const uint64_t stop = std::min<uint64_t>(a_byte * 8 + end_bit, from + 8 * (uint64_t)((host_mode_ ? 16u : 1u) << 20));
        uint64_t eb = 0;
        bool stream_end = false;
        std::string err;
        // (a member that started before this piece keeps its window; one that starts here has none -- the window only
        //  matters for valid references, which never reach before a member's start)
        if (inflate_from(base_, base_ + size_, from, stop, window.data(), WSIZE, out, members, &eb, &stream_end, err, room / 2) != 0) {
            error_ = err + " (" + path_ + ")";
            return -1;
        }
        if (out.size() > room) return fail("a block's text does not fit the batch buffer");
        if (!out.empty()) GZ_TRY(hipMemcpyAsync(d_dst, out.data(), out.size(), hipMemcpyHostToDevice, stream));
        // the window behind it
        std::vector<uint8_t> nw(WSIZE);
        if (out.size() >= WSIZE) memcpy(nw.data(), out.data() + out.size() - WSIZE, WSIZE);
        else {
            memcpy(nw.data(), window.data() + out.size(), WSIZE - out.size());
            memcpy(nw.data() + WSIZE - out.size(), out.data(), out.size());
        }
        GZ_TRY(hipMemcpyAsync(s_->d_win_[s_->win_ ^ 1], nw.data(), WSIZE, hipMemcpyHostToDevice, stream));
        GZ_TRY(hipStreamSynchronize(stream));
        s_->win_ ^= 1;
        uint64_t a = 0;
        for (const GzMemberEnd &m : members) {
            account(crc32_fast(0, out.data() + a, (size_t)(m.out_pos - a)), m.out_pos - a);
            if (!member_end(m.crc, m.isize)) return -1;
            a = m.out_pos;
        }
        account(crc32_fast(0, out.data() + a, out.size() - (size_t)a), out.size() - a);
        {
            std::lock_guard<std::mutex> lk(st_mu_);  // (phase A of a piece decoded ahead updates the statistics on another thread)
            st_.segments++;
            s_->pieces++;
            st_.fallback_segments++;
            st_.text_bytes += out.size();
            st_.gzip_bytes += (eb - pos_bit_) / 8;
        }
        pos_bit_ = eb;
        if (stream_end) {
            if (run_len_) return fail("unexpected end of file", true);
            ended_ = true;
        }
        return (long)out.size();
    }

    std::string path_;
    int fd_ = -1;
    size_t size_ = 0;
    const uint8_t *base_ = nullptr;
    uint64_t pos_bit_ = 0;  // where the stream goes on (a block boundary), bits from the start of the file
    size_t stretch_ = 0, seg_ = 0, look_ = 0;
    uint32_t n_slots_ = 0, slot_syms_ = 0;
    double ratio_ = 5.0;  // text per compressed byte seen lately (before anything was seen: FASTQ's 4-5 : 1; a piece that does not fit is cut down)
    uint32_t run_crc_ = 0;
    uint64_t run_len_ = 0;
    bool trace_ = false, warned_ = false, host_mode_ = false;
    bool bgzf_ = false;          // a BGZF file: members of one final block each, their sizes in the headers -- no search needed
    uint64_t bgzf_hdr_ = 0;      // header of the member whose data the stream stands at (byte offset in the file)
    std::mutex st_mu_;           // the statistics: phase A of a piece decoded ahead runs on another thread
    uint64_t grid0_ = 0;         // the piece grid starts at the first member's first block (aligned down)
    uint64_t spec_used_ = 0, spec_refused_ = 0;
    long fake_spec_ = 0;         // test knob NOHUMAN_GZDEV_FAKE_SPEC=k: the k-th piece decoded ahead gets a false first start
    bool debug_ = getenv("NOHUMAN_GZDEV_NOCRC") != nullptr;  // debugging aid: one line per piece, CRC failures reported and passed over
    uint32_t failed_in_a_row_ = 0, last_bad_ = 0;
    uint32_t host_left_ = 0, host_stint_ = 8;  // host mode: steps left of this stint; the next stint's length
    long fake_start_ = -1;
    long fake_crc_ = 0;
    // the host lane of the hybrid reader (take_host)
    unsigned host_threads_ = 0;
    uint64_t host_cell_ = 0;  // bytes of gzip of the grid's odd cells (0: no host lane, a uniform grid)
    std::unique_ptr<RangeGunzip> hc_;
    uint64_t hc_cell_ = 0;
    bool hc_valid_ = false;
    uint8_t *h_text_ = nullptr;  // page-locked: a host cell's text on its way to the device
    size_t h_text_cap_ = 0;
    uint64_t host_pieces_ = 0;
    double host_s_ = 0;
    bool integrity_ = false;  // the error is one of the END-TO-END checks on the decode: a member's CRC-32 / ISIZE, the stream's end
    double open_s_ = 0;
    size_t in_bytes_ = 0, sym_bytes_ = 0, maps_bytes_ = 0, stage_bytes_ = 0;
    // The bytes of the stream's NEXT piece on their way into the other staging buffer (a helper thread, while this piece's
    // kernels run): page cache -> page-locked memory runs at ~10 GB/s on one core, 50 ms of a 512 MiB piece's 250 that the
    // GPU used to spend idle.  One reader with one buffer set only (several sets decode cells ahead on threads of their own).
    struct Staged {
        std::thread th;
        uint64_t start = 0;
        size_t len = 0;
        uint8_t *buf = nullptr;
        bool valid = false;
    } pf_;
    uint64_t pf_hits_ = 0, pf_misses_ = 0;
    bool stage_ahead_ = false;
    uint64_t prof_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};

DevGunzip::DevGunzip() : impl_(new DevGunzipImpl()) {}
DevGunzip::~DevGunzip() { delete impl_; }
int DevGunzip::open(const char *path, int device, size_t seg_bytes, size_t stretch_bytes, std::string &err) {
    return impl_->open(path, device, seg_bytes, stretch_bytes, err);
}
long DevGunzip::next(void *d_dst, size_t room, hipStream_t stream) { return impl_->next(d_dst, room, stream); }
int DevGunzip::add_device(int device, std::string &err) { return impl_->add_device(device, err); }
long DevGunzip::next_on(int set, void *d_dst, size_t room, hipStream_t stream) { return impl_->next_on(set, d_dst, room, stream); }
uint64_t DevGunzip::pieces_of(int set) const { return impl_->pieces_of(set); }
bool DevGunzip::ahead_ok() const { return impl_->spec_ok(); }
uint64_t DevGunzip::cell_of_position() const { return impl_->cell_of_pos(); }
uint64_t DevGunzip::cells() const { return impl_->cells(); }
void DevGunzip::decode_ahead(int set, uint64_t cell, hipStream_t stream) { impl_->spec_decode(set, cell, stream); }
long DevGunzip::take(int set, bool use_ahead, void *d_dst, size_t room, hipStream_t stream) { return impl_->take(set, use_ahead, d_dst, room, stream); }
bool DevGunzip::ended() const { return impl_->ended_; }
const std::string &DevGunzip::error() const { return impl_->error_; }
bool DevGunzip::integrity_failure() const { return impl_->integrity_; }
void DevGunzip::set_host_threads(unsigned n) { impl_->set_host(n); }
bool DevGunzip::host_ok(size_t room) const { return impl_->host_ok(room); }
void DevGunzip::decode_ahead_host(uint64_t cell) { impl_->host_decode_ahead(cell); }
long DevGunzip::take_host(bool use_ahead, void *d_dst, size_t room, hipStream_t stream) { return impl_->take_host(use_ahead, d_dst, room, stream); }
uint64_t DevGunzip::host_pieces() const { return impl_->host_pieces_; }
const DevGunzipStats &DevGunzip::stats() const { return impl_->st_; }
void DevGunzip::close() { impl_->close(); }

bool dev_gunzip_wants(const char *path) {
    int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    uint8_t h[18];
    bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 18 && ::read(fd, h, 18) == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8;
    // (BGZF -- FEXTRA with the 'B' 'C' subfield -- is taken as well: its chunks' starts come from the members' headers)
    ::close(fd);
    return ok;
}

}  // namespace nh

extern "C" uint64_t nh_cache_bytes(int32_t device, int32_t page_locked) { return (uint64_t)nh::cache_bytes(device, page_locked != 0); }

extern "C" int nh_gunzip_device_file(const char *in, const char *out, int32_t device, uint64_t seg_bytes, uint64_t stretch_bytes,
                                     uint64_t *stats8) {
    if (!in || !out) return nh::set_error(NH_EINVAL, "null argument");
    nh::DevGunzip gz;
    std::string err;
    if (gz.open(in, device, (size_t)seg_bytes, (size_t)stretch_bytes, err) != 0) return nh::set_error(NH_EIO, "%s", err.c_str());
    size_t room = (size_t)3584u << 20;
    {
        struct stat st;
        if (stat(in, &st) == 0 && (uint64_t)st.st_size * 16 + ((size_t)64u << 20) < room) room = (size_t)st.st_size * 16 + ((size_t)64u << 20);
    }
    if (const char *e = getenv("NOHUMAN_GZDEV_ROOM")) room = std::max<size_t>((size_t)atoll(e), (size_t)1u << 20);  // tool knob: text per piece
    uint8_t *d_text = nullptr;
    hipStream_t stream = nullptr;
    if (nh::dev_set(device) != hipSuccess || nh::dev_malloc((void **)&d_text, room + 64) != hipSuccess ||
        hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) {
        if (d_text) (void)hipFree(d_text);
        return nh::set_error(NH_EOOM, "cannot allocate the text buffer on device %d", device);
    }
    int rc = NH_OK;
    FILE *f = fopen(out, "wb");
    if (!f) rc = nh::set_error(NH_EIO, "cannot create %s", out);
    std::vector<uint8_t> host;
    while (rc == NH_OK) {
        const long n = gz.next(d_text, room, stream);
        if (n < 0) {
            rc = nh::set_error(NH_EIO, "%s", gz.error().c_str());
            break;
        }
        if (n == 0) break;
        host.resize((size_t)n);
        if (hipMemcpy(host.data(), d_text, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = nh::set_error(NH_EDEVICE, "D2H of the text failed");
        else if (fwrite(host.data(), 1, (size_t)n, f) != (size_t)n) rc = nh::set_error(NH_EIO, "write error on %s", out);
    }
    if (f && fclose(f) != 0 && rc == NH_OK) rc = nh::set_error(NH_EIO, "write error on %s", out);
    if (stats8) {
        const nh::DevGunzipStats &s = gz.stats();
        stats8[0] = s.segments;
        stats8[1] = s.chunks;
        stats8[2] = s.redecoded;
        stats8[3] = s.fallback_segments;
        stats8[4] = s.members;
        stats8[5] = s.text_bytes;
        stats8[6] = s.gzip_bytes;
        stats8[7] = (uint64_t)((s.ms_search + s.ms_decode + s.ms_scan + s.ms_resolve + s.ms_crc) * 1000.0);
    }
    gz.close();
    (void)hipStreamDestroy(stream);
    (void)hipFree(d_text);
    return rc;
}

// =====================================================================================================================
// DevFastqReader: the record index on the device (nh_gunzip.h)
// =====================================================================================================================
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "nh_fastx.h"

namespace nh {
namespace fq {

constexpr uint32_t TILE = 16384;  // bytes of text a workgroup counts / lists newlines of

__global__ __launch_bounds__(256) void k_nl_count(const uint8_t *text, uint64_t n, uint32_t *tile_cnt) {
    __shared__ uint32_t s_sum[4];
    const uint64_t t0 = (uint64_t)blockIdx.x * TILE;
    uint32_t c = 0;
    for (uint32_t i = threadIdx.x * 16; i < TILE; i += 256 * 16) {
        const uint64_t p = t0 + i;
        if (p + 16 <= n) {
            struct __attribute__((packed)) U16 {
                uint32_t v[4];
            };
            const U16 v = *(const U16 *)(text + p);  // (any alignment: a piece's text starts where the part carried over begins)
            const uint32_t w[4] = {v.v[0], v.v[1], v.v[2], v.v[3]};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t x = w[k] ^ 0x0A0A0A0Au;  // zero bytes where the text has '\n'
                c += (uint32_t)__popc(~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu));
            }
        } else {
            for (uint64_t q = p; q < n && q < p + 16; q++) c += text[q] == '\n';
        }
    }
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// exclusive prefix sum of the tiles' counts (one workgroup), total behind the last entry
__global__ __launch_bounds__(1024) void k_nl_scan(uint32_t *tile_cnt, uint32_t n_tiles) {
    __shared__ unsigned long long s[1024];
    const uint32_t t = threadIdx.x, per = (n_tiles + 1023) / 1024;
    unsigned long long sum = 0;
    for (uint32_t i = 0; i < per; i++) {
        const uint32_t k = t * per + i;
        if (k < n_tiles) sum += tile_cnt[k];
    }
    s[t] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const unsigned long long v = t >= o ? s[t - o] : 0;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    unsigned long long run = s[t] - sum;
    for (uint32_t i = 0; i < per; i++) {
        const uint32_t k = t * per + i;
        if (k < n_tiles) {
            const uint32_t c = tile_cnt[k];
            tile_cnt[k] = (uint32_t)run;
            run += c;
        }
    }
    if (t == 1023) tile_cnt[n_tiles] = (uint32_t)s[1023];
}

// positions of the newlines, in order
__global__ __launch_bounds__(256) void k_nl_list(const uint8_t *text, uint64_t n, const uint32_t *tile_off, uint32_t *nl) {
    __shared__ uint32_t s_cnt[256];
    const uint64_t t0 = (uint64_t)blockIdx.x * TILE;
    const uint32_t per = TILE / 256;  // 64 consecutive bytes a thread
    const uint64_t p0 = t0 + (uint64_t)threadIdx.x * per;
    uint64_t m = 0;  // bit i: byte p0 + i is a newline
    for (uint32_t i = 0; i < per; i++)
        if (p0 + i < n && text[p0 + i] == '\n') m |= 1ull << i;
    s_cnt[threadIdx.x] = (uint32_t)__popcll(m);
    __syncthreads();
    for (uint32_t o = 1; o < 256; o <<= 1) {
        const uint32_t v = threadIdx.x >= o ? s_cnt[threadIdx.x - o] : 0;
        __syncthreads();
        s_cnt[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t at = tile_off[blockIdx.x] + s_cnt[threadIdx.x] - (uint32_t)__popcll(m);
    while (m) {
        nl[at++] = (uint32_t)(p0 + (uint64_t)__builtin_ctzll(m));
        m &= m - 1;
    }
}

__device__ __forceinline__ bool is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }

// One record a thread: lines 4r .. 4r+3.  Offsets are relative to the first record of the record's batch (batches of
// `bf` records from record 0 on); bstart[b] = absolute offset of batch b's first record, bstart[nb] = end of the last record.
// bad: the smallest (record << 2 | kind) of a record that is none -- kind 1: empty header line or "@" alone (kraken2 ends
// the input there), 2: no '@' (malformed).
__global__ __launch_bounds__(256) void k_records(const uint8_t *text, const uint32_t *nl, uint32_t n_rec, uint32_t bf, RecRef *recs,
                                                 uint32_t *bstart, unsigned long long *bad) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rec) return;
    const uint32_t l0 = 4 * r;
    const uint32_t b0 = l0 ? nl[l0 - 1] + 1 : 0, e0 = nl[l0];
    const uint32_t b1 = e0 + 1, e1 = nl[l0 + 1], b2 = e1 + 1, e2 = nl[l0 + 2], b3 = e2 + 1, e3 = nl[l0 + 3];
    uint32_t he = e0, se = e1, qe = e3;
    while (he > b0 && is_space(text[he - 1])) he--;
    while (se > b1 && is_space(text[se - 1])) se--;
    while (qe > b3 && is_space(text[qe - 1])) qe--;
    // (the host parser, nh_fastx.cpp parse_one: an empty line ends the input, anything else without '@' is malformed --
    //  a one-character line "X" too --, "@" alone ends the input)
    if (he == b0 || (he - b0 == 1 && text[b0] == '@')) atomicMin(bad, ((unsigned long long)r << 2) | 1ull);
    else if (text[b0] != '@') atomicMin(bad, ((unsigned long long)r << 2) | 2ull);
    uint32_t ie = b0 + 1;
    while (ie < he && text[ie] != ' ' && text[ie] != '\t' && text[ie] != '\r') ie++;
    const uint32_t rb = r / bf * bf;  // first record of the batch
    const uint32_t base = rb ? nl[4 * rb - 1] + 1 : 0;
    const bool canon = he == e0 && se == e1 && qe == e3 && e2 - b2 == 1 && text[b2] == '+';
    RecRef x;
    x.h = b0 - base;
    x.hlen = he - b0;
    x.idlen = ie > b0 ? ie - b0 - 1 : 0;
    x.s = b1 - base;
    x.slen = se - b1;
    x.q = b3 - base;
    x.qlen = qe - b3;
    x.raw_end = canon ? e3 + 1 - base : 0;
    recs[r] = x;
    if (r == rb) bstart[r / bf] = b0;
    if (r == n_rec - 1) bstart[(n_rec + bf - 1) / bf] = e3 + 1;
}

}  // namespace fq

class DevFastqImpl {
public:
    ~DevFastqImpl() { close(); }
    int open(const char *path, const int *devices, int n_devices, std::string &err, unsigned host_threads) {
        if (!dev_gunzip_wants(path)) return 1;
        if (n_devices < 1) return 1;
        path_ = path;
        const auto t_open = std::chrono::steady_clock::now();
        // A lane = a device of the run with a buffer set of the decoder, two streams and two text buffers.  One lane: piece
        // after piece, each decoded while the batches of the one before go out.  Several lanes: the pieces of the stream are
        // decoded AHEAD on all lanes at once (DevGunzip::decode_ahead: search and decode need neither the window nor the
        // stream's state) and taken in stream order (windows, text, CRC, record index) -- piece i's batches are classified on
        // the device it was decoded on.
        lanes_.resize((size_t)n_devices);
        for (int g = 0; g < n_devices; g++) lanes_[(size_t)g].device = devices[g];
        if (nh::dev_set(devices[0]) != hipSuccess) {
            err = "hipSetDevice failed";
            return -1;
        }
        if (gz_.open(path, devices[0], 0, 0, err) != 0) return -1;
        const double t_a = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_open).count();
        for (int g = 1; g < n_devices; g++)
            if (gz_.add_device(devices[g], err) != g) return -1;
        n_dev_lanes_ = lanes_.size();
        // The hybrid reader (round 6): one more lane whose cells are inflated by the HOST's cores (DevGunzip::decode_ahead_host /
        // take_host) -- its pieces' text is uploaded to the first device and indexed and classified there like any other's.
        // What it is for: with gzip outputs encoded on the GPU the chip's codec kernels are the run's bottleneck while the
        // host's cores idle (profiles/r05_e2e_busy_gzip.txt).  Needs the piece grid (no BGZF, which has no search to share).
        if (host_threads > 0 && gz_.ahead_ok() && !(getenv("NOHUMAN_GZ_AHEAD") && getenv("NOHUMAN_GZ_AHEAD")[0] == '0')) {
            Lane h;
            h.device = devices[0];
            h.host = true;
            lanes_.push_back(std::move(h));
            gz_.set_host_threads(host_threads);
        }
        const double t_b = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_open).count();
        // text of a piece: 512 MiB of gzip at FASTQ's 4-5 : 1 and the partial batch carried in front of it (the record index
        // addresses a piece's text with 32 bits: below 4 GiB)
        room_ = (size_t)3584u << 20;
        struct stat st;
        if (stat(path, &st) == 0 && (uint64_t)st.st_size * 16 + ((size_t)64u << 20) < room_) room_ = (size_t)st.st_size * 16 + ((size_t)64u << 20);
        if (const char *e = getenv("NOHUMAN_GZDEV_ROOM")) room_ = std::max<size_t>((size_t)atoll(e), (size_t)1u << 20);
        if (room_ > ((size_t)4092u << 20)) room_ = (size_t)4092u << 20;  // (32-bit positions in a piece's text)
        // The decoder writes a piece's text at d_text + head_; what the piece before could not hand out as whole batches (less
        // than a batch of records and an incomplete one) is copied in FRONT of it afterwards, by the index stage -- so the
        // decode of piece i + 1 does not wait for the index of piece i.
        head_ = room_ >= ((size_t)2u << 30) ? (size_t)768u << 20 : (room_ / 2 + 4095) & ~(size_t)4095;
        buf_.resize(2 * lanes_.size());
        for (size_t i = 0; i < buf_.size(); i++) buf_[i].lane = (int)(i / 2);
        for (;;) {
            bool ok = true;
            for (Piece &b : buf_) {
                (void)nh::dev_set(lanes_[(size_t)b.lane].device);
                b.d_text = (uint8_t *)cache_alloc(lanes_[(size_t)b.lane].device, room_ + 4096, false);
                ok = ok && b.d_text;
            }
            if (ok) break;
            for (Piece &b : buf_) {
                cache_free(lanes_[(size_t)b.lane].device, room_ + 4096, b.d_text, false);
                b.d_text = nullptr;
            }
            if (room_ <= ((size_t)128u << 20)) {
                err = "the gzip reader's text buffers cannot be had";
                return -1;
            }
            room_ /= 2;
        }
        const double t_c = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_open).count();
        for (Lane &l : lanes_) {
            if (nh::dev_set(l.device) != hipSuccess || hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) != hipSuccess ||
                hipStreamCreateWithFlags(&l.stream_a, hipStreamNonBlocking) != hipSuccess ||
                hipStreamCreateWithFlags(&l.stream_i, hipStreamNonBlocking) != hipSuccess || dev_malloc((void **)&l.d_bad, 8) != hipSuccess ||
                host_malloc((void **)&l.h_bad, 8, hipHostMallocDefault) != hipSuccess) {
                err = "the gzip reader's buffers cannot be had";
                return -1;
            }
        }
        ahead_ = lanes_.size() > 1 && gz_.ahead_ok() && !(getenv("NOHUMAN_GZ_AHEAD") && getenv("NOHUMAN_GZ_AHEAD")[0] == '0');
        trace_ = getenv("NOHUMAN_TRACE") != nullptr;
        if (const char *e = getenv("NOHUMAN_GZDEV_FAIL_AT")) fail_at_ = atol(e);
        open_s_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_open).count();
        if (trace_ && getenv("NOHUMAN_GZDEV_NOCRC"))
            fprintf(stderr, "[gzdev] reader set-up: decoder %.3f s, further lanes %.3f, text buffers %.3f, streams %.3f\n", t_a, t_b - t_a, t_c - t_b, open_s_ - t_c);
        return 0;
    }

    int next_batch(HalfBatch &hb, size_t max_recs, size_t max_text) {
        hb.reset();
        hb.format = FMT_FASTQ;
        if (bf_ == 0) {
            bf_ = max_recs ? max_recs : 1;
            max_text_ = std::min<size_t>(max_text, (size_t)3u << 30);  // (0: batches of exactly bf_ records -- paired inputs)
            th_ = std::thread([this] { produce(); });  // pieces are decoded and indexed ahead of the batches handed out
        }
        if (max_recs != bf_) {
            hb.error = "DevFastqReader: the batch size changed";
            return 0;
        }
        for (;;) {
            Piece *pp = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return !order_.empty() || done_; });
                if (order_.empty()) {  // the producer has ended: the end of the input, an error, or "not ours"
                    if (fallback_) return 1;
                    bool hard = false;
                    const std::string e = error_text(&hard);
                    if (!e.empty()) {
                        if (!hard) {  // the host reader goes on from records_handed() (handover_reason())
                            reason_ = e;
                            return 1;
                        }
                        hb.error = e;
                    }
                    hb.eof = true;
                    return 0;
                }
                pp = &buf_[order_.front()];
            }
            Piece &p = *pp;
            // whole batches, and at the end of the input what is left
            const size_t full = p.last || max_text_ ? p.n_rec : p.n_rec / bf_ * bf_;
            if (p.next_rec < full) return emit(hb, p, full);
            const bool last = p.last;
            {   // this piece has handed out what it had: on to the next
                std::lock_guard<std::mutex> lk(mu_);
                p.loaded = false;
                order_.pop_front();
            }
            cv_.notify_all();
            if (last) {
                hb.eof = true;
                return 0;
            }
        }
    }

    uint64_t records_handed() const { return handed_recs_; }
    const std::string &handover_reason() const { return reason_; }  // (set by the next_batch() call that returned 1)

    void close() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
        for (Lane &l : lanes_)
            if (l.worker.joinable()) l.worker.join();
        // every batch handed out points into the text buffers: wait until the pipeline has let go of them
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] {
                for (const Piece &b : buf_)
                    if (b.outstanding) return false;
                return true;
            });
        }
        if (trace_ && pieces_)
            fprintf(stderr, "[nohuman trace] record index on GPU %d (%zu lane%s%s), %s: %llu pieces, %llu records, index kernels + table D2H %.3f s, carried %.2f GB, buffers set up in %.3f s\n",
                    lanes_.empty() ? -1 : lanes_[0].device, lanes_.size(), lanes_.size() == 1 ? "" : "s", ahead_ ? ", pieces decoded ahead" : "", path_.c_str(),
                    (unsigned long long)pieces_, (unsigned long long)records_, index_s_, carried_ / 1e9, open_s_);
        pieces_ = 0;
        gz_.close();
        for (Piece &b : buf_) {
            const int dev = lanes_[(size_t)b.lane].device;
            (void)nh::dev_set(dev);
            cache_free(dev, room_ + 4096, b.d_text, false);
            cache_free(dev, b.nl_cap * 4, b.d_nl, false);
            cache_free(dev, b.tile_cap * 4, b.d_tiles, false);
            cache_free(dev, b.rec_cap * sizeof(RecRef), b.d_recs, false);
            cache_free(dev, b.bs_cap * 4, b.d_bstart, false);
            cache_free(dev, b.rec_cap * sizeof(RecRef), b.h_recs, true);
            cache_free(dev, b.bs_cap * 4, b.h_bstart, true);
        }
        buf_.clear();
        order_.clear();
        for (Lane &l : lanes_) {
            (void)nh::dev_set(l.device);
            if (l.d_bad) (void)hipFree(l.d_bad);
            if (l.h_bad) (void)hipHostFree(l.h_bad);
            if (l.stream) (void)hipStreamDestroy(l.stream);
            if (l.stream_a) (void)hipStreamDestroy(l.stream_a);
            if (l.stream_i) (void)hipStreamDestroy(l.stream_i);
        }
        lanes_.clear();
    }

private:
    struct Piece {
        uint8_t *d_text = nullptr;
        uint32_t *d_nl = nullptr, *d_tiles = nullptr, *d_bstart = nullptr, *h_bstart = nullptr;
        RecRef *d_recs = nullptr, *h_recs = nullptr;
        size_t nl_cap = 0, tile_cap = 0, rec_cap = 0, bs_cap = 0;
        size_t text_len = 0, used_len = 0;  // bytes of text; bytes up to the end of the last complete record
        size_t n_rec = 0, next_rec = 0;
        bool loaded = false, last = false, indexed = false;  // loaded: handed to the consumer; indexed: its record table is valid
        int outstanding = 0;  // batches handed out and not yet released
        int lane = 0;
        bool prepared = false;  // prealloc() is through with it
        uint8_t *text0 = nullptr;  // where the piece's text begins: d_text + head_ - (what was carried over from the piece before)
        size_t body_len = 0;       // bytes the decoder wrote at d_text + head_
        bool in_pipe = false;      // between the decode stage and its publication
        bool tail_needed = false;  // published, and the piece after it has not taken its tail yet
    };
    struct Lane {
        int device = -1;
        bool host = false;  // the hybrid reader's host lane: cells inflated by the host's cores, text uploaded to `device`
        hipStream_t stream = nullptr, stream_a = nullptr, stream_i = nullptr;  // the stream's own work; a piece decoded ahead; the index stage
        unsigned long long *d_bad = nullptr, *h_bad = nullptr;
        std::thread worker;
        int state = 0;      // 0 idle, 1 a cell is being decoded ahead, 2 decoded
        uint64_t cell = 0;  // which
    };

    // What stops this reader does not stop the run: the host reader takes the file over from the records handed out so far
    // (next_batch() == 1) and says what it finds.  Two things are final (hard): malformed FASTQ in text whose CRC-32 was right
    // -- the input's fault on any reader --, and a failed CRC-32 / ISIZE / stream end once records have been handed out -- the
    // check that fails may be about those very records (decode_stage).
    int fail(const std::string &m, bool hard = false) {
        std::lock_guard<std::mutex> lk(err_mu_);  // (the decode stage, the index stage and the lanes' workers may all end here)
        if (error_.empty()) {
            error_ = m;
            hard_ = hard;
        }
        return -1;
    }
    std::string error_text(bool *hard = nullptr) {
        std::lock_guard<std::mutex> lk(err_mu_);
        if (hard) *hard = hard_;
        return error_;
    }
    // the record index's buffers (a piece of 2.3 GB of text: 9 M newline positions, a 220 MB record table on the device and
    // page-locked on the host) come from the process-wide store like the decoder's: page-locking the table took 0.1 s a buffer
    template <class T>
    bool grow_dev(int device, T *&p, size_t &cap, size_t need, bool host = false) {
        if (need <= cap) return true;
        cache_free(device, cap * sizeof(T), p, host);
        p = nullptr;
        cap = need + need / 4 + 1024;
        p = (T *)cache_alloc(device, cap * sizeof(T), host);
        if (!p) cap = 0;
        return p != nullptr;
    }
    // Before the first piece is indexed: the buffers at the size this file's pieces will need (a guess: 4.5 : 1, records of
    // 300 bytes), made by a helper while the first piece is still being decoded
    void prealloc() {
        struct stat st;
        memset(&st, 0, sizeof st);
        size_t text = room_;
        if (stat(path_.c_str(), &st) == 0 && (uint64_t)st.st_size * 5 < text) text = (size_t)st.st_size * 5;
        const size_t recs = text / 300 + 4096, lines = 4 * recs + 16, tiles = text / fq::TILE + 2, batches = recs / (bf_ ? bf_ : 1) + 8;
        // (no more buffers than the file has pieces: page-locking a record table is 0.1 s, and a run over three lanes has twelve)
        size_t want = buf_.size();
        if (st.st_size > 0) want = std::min<size_t>(want, (size_t)((uint64_t)st.st_size / ((uint64_t)512u << 20)) + 2);
        // the decode stage takes lane 0's buffers first, then the other lanes' in turn
        std::vector<size_t> turn;
        for (size_t k = 0; k < 2; k++)
            for (size_t g = 0; g < lanes_.size(); g++) turn.push_back(2 * g + k);
        size_t made = 0;
        for (size_t bi : turn) {
            Piece &p = buf_[bi];
            if (made++ >= want) break;
            const int dev = lanes_[(size_t)p.lane].device;
            if (nh::dev_set(dev) != hipSuccess) break;
            (void)grow_dev(dev, p.d_tiles, p.tile_cap, tiles);
            (void)grow_dev(dev, p.d_nl, p.nl_cap, lines);
            if (grow_dev(dev, p.d_recs, p.rec_cap, recs)) {
                size_t hc = 0;
                if (!grow_dev(dev, p.h_recs, hc, p.rec_cap, true)) p.rec_cap = 0;  // (the two tables grow together)
                else if (hc < p.rec_cap) p.rec_cap = hc;
            }
            if (grow_dev(dev, p.d_bstart, p.bs_cap, batches)) {
                size_t hc = 0;
                if (!grow_dev(dev, p.h_bstart, hc, p.bs_cap, true)) p.bs_cap = 0;
                else if (hc < p.bs_cap) p.bs_cap = hc;
            }
            std::lock_guard<std::mutex> lk(mu_);
            p.prepared = true;
            cv_.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu_);
        for (Piece &p : buf_) p.prepared = true;
        cv_.notify_all();
    }

    int emit(HalfBatch &hb, Piece &p, size_t full) {
        // the record table is laid out in batches of bf_ records from the piece's first (k_records: offsets relative to the
        // batch's first record).  A batch whose text exceeds max_text_ goes out in parts: cut behind the record that reaches
        // the budget (BlockReader::next_batch's rule), the part's offsets moved to its own first record
        const size_t r0 = p.next_rec, b = r0 / bf_, g0 = b * bf_, gend = std::min(full, g0 + bf_);
        const size_t base = p.h_bstart[b];
        const size_t t0 = r0 == g0 ? base : base + p.h_recs[r0].h;
        size_t r1 = gend, t1 = p.h_bstart[b + 1];
        if (max_text_ && t1 - t0 > max_text_) {
            r1 = r0;
            size_t pos = 0;
            while (r1 < gend && pos < max_text_) {
                r1++;
                pos = (r1 < gend ? base + p.h_recs[r1].h : (size_t)p.h_bstart[b + 1]) - t0;
            }
            t1 = t0 + pos;
        }
        hb.recs.assign(p.h_recs + r0, p.h_recs + r1);
        if (const uint32_t delta = (uint32_t)(t0 - base))
            for (RecRef &x : hb.recs) {
                x.h -= delta;
                x.s -= delta;
                x.q -= delta;
                if (x.raw_end) x.raw_end -= delta;
            }
        handed_recs_ += r1 - r0;
        const size_t len = t1 - t0;
        // the host text buffer stays a token (its address is the batch's handle in the writer's span lists); whoever needs the
        // bytes on the host reserves the real thing (nh_run: stage_text, the writer's fetch)
        hb.text.clear();
        if (!hb.text.reserve(64)) {
            hb.error = "out of memory";
            return 0;
        }
        hb.text.set_size(len);
        hb.dev_text = p.text0 + t0;
        hb.dev_device = lanes_[(size_t)p.lane].device;
        hb.host_text_valid = false;
        p.next_rec = r1;
        hb.eof = p.last && r1 == full;
        {
            std::lock_guard<std::mutex> lk(mu_);
            p.outstanding++;
        }
        Piece *pp = &p;
        hb.release = [this, pp] {
            {
                std::lock_guard<std::mutex> lk(mu_);
                pp->outstanding--;
            }
            cv_.notify_all();
        };
        return 0;
    }

    // the lanes' workers: a cell of the piece grid decoded ahead of the stream
    void work(size_t g) {
        Lane &ln = lanes_[g];
        for (;;) {
            uint64_t cell;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return ln.state == 1 || stop_ || done_; });
                if (ln.state != 1) return;
                cell = ln.cell;
            }
            if (ln.host) gz_.decode_ahead_host(cell);
            else gz_.decode_ahead((int)g, cell, ln.stream_a);
            {
                std::lock_guard<std::mutex> lk(mu_);
                ln.state = 2;
            }
            cv_.notify_all();
        }
    }
    // idle lanes get the next cells of the grid (never the cell the stream stands in, nor one behind it)
    void assign_ahead() {
        const uint64_t k = gz_.cell_of_position(), n = gz_.cells();
        std::lock_guard<std::mutex> lk(mu_);
        if (next_cell_ <= k) next_cell_ = k + 1;
        for (Lane &l : lanes_)
            if (l.state == 2 && l.cell < k) l.state = 0;  // (the stream is past it: a block longer than a cell -- tiny cells of the tests)
        if (lanes_.size() == n_dev_lanes_) {
            for (Lane &l : lanes_)
                if (l.state == 0 && next_cell_ < n) {
                    l.cell = next_cell_++;
                    l.state = 1;
                }
            cv_.notify_all();
            return;
        }
        // The hybrid reader: the grid alternates (DevGunzip: even cells of a piece's size, odd cells a quarter of that): the odd
        // cells are the host lane's, the even ones the devices' -- which take the odd ones too while the host lane is off (a file
        // that inflates further than its buffer holds).  While nobody holds the cell the stream stands in, one idle DEVICE lane is
        // kept back for it: load_next takes it in order at once.  Nobody runs more than eight cells ahead.
        auto taken = [&](uint64_t c) {
            for (const Lane &l : lanes_)
                if (l.state != 0 && l.cell == c) return true;
            return false;
        };
        const bool host_on = gz_.host_ok(room_ - head_);
        const bool owned = taken(k);
        bool kept = false;
        for (Lane &l : lanes_) {
            if (l.state != 0 || (l.host && !host_on)) continue;
            if (!owned && !l.host && !kept) {
                kept = true;
                continue;
            }
            uint64_t c = k + 1;
            for (; c < n && c <= k + 8; c++) {
                if (taken(c)) continue;
                const bool odd = (c & 1) != 0;
                if (l.host ? odd : (!odd || !host_on)) break;
            }
            if (c >= n || c > k + 8) continue;
            l.cell = c;
            l.state = 1;
        }
        cv_.notify_all();
    }

    // Two stages.  DECODE (this thread): the stream's pieces in order -- which lane, a free text buffer there, the decoder
    // (DevGunzip::take / next_on) writes the text at d_text + head_.  INDEX (a thread of its own, in the same order): what
    // the piece before could not hand out as whole batches goes in front of the text, newline scan, record table, and the
    // piece is published to next_batch().  The decode of piece i + 1 runs while piece i is indexed.
    void produce() {
        std::thread pre([this] { prealloc(); });
        std::thread ix([this] { index_stage(); });
        struct Join {
            std::thread &a, &b;
            ~Join() {
                if (a.joinable()) a.join();
                if (b.joinable()) b.join();
            }
        } join_them{pre, ix};
        if (ahead_)
            for (size_t g = 0; g < lanes_.size(); g++) lanes_[g].worker = std::thread([this, g] { work(g); });
        for (;;) {
            bool last = false;
            const int rc = load_next(last);
            std::lock_guard<std::mutex> lk(mu_);
            if (rc != 0 || last || stop_ || done_) {
                if (rc == 1) fallback_ = true;
                if (rc != 0) done_ = true;  // (an error: nothing more comes; the end of the input: the index stage says so)
                decode_done_ = true;
                cv_.notify_all();
                return;
            }
        }
    }

    int load_next(bool &last) {
        // ---- which lane takes the stream's next piece
        size_t g = piece_no_ % n_dev_lanes_;  // (in turn over the DEVICE lanes where pieces are not decoded ahead)
        bool use_ahead = false;
        if (ahead_ && gz_.ahead_ok()) {
            assign_ahead();
            const uint64_t k = gz_.cell_of_position();
            std::unique_lock<std::mutex> lk(mu_);
            size_t pick = lanes_.size();
            for (size_t i = 0; i < lanes_.size(); i++)
                if (lanes_[i].state != 0 && lanes_[i].cell == k) pick = i;
            if (pick < lanes_.size()) {
                use_ahead = true;
            } else {
                // nobody decoded this cell ahead (the stream's first piece; a piece that was refused, or ended early): an idle
                // lane decodes it in order -- or, all being busy with cells further on, the one furthest ahead gives its cell up
                // (device lanes only: an in-order piece on the host's cores would be the slowest way to decode it)
                for (size_t i = 0; i < n_dev_lanes_; i++)
                    if (lanes_[i].state == 0) pick = i;
                if (pick == lanes_.size()) {
                    pick = 0;
                    for (size_t i = 1; i < n_dev_lanes_; i++)
                        if (lanes_[i].cell > lanes_[pick].cell) pick = i;
                    if (next_cell_ > lanes_[pick].cell) next_cell_ = lanes_[pick].cell;  // (it is dealt out again)
                }
            }
            g = pick;
            cv_.wait(lk, [&] { return lanes_[g].state != 1 || stop_ || done_; });
            if (stop_ || done_) return -2;
            lanes_[g].state = 3;  // the stream's own (not idle: no new cell until the piece is through)
            lanes_[g].cell = k;
        }
        Lane &ln = lanes_[g];
        if (fail_at_ > 0 && piece_no_ == (uint64_t)fail_at_) return fail("test knob NOHUMAN_GZDEV_FAIL_AT");  // (the handover, provoked)
        // ---- a free text buffer of that lane
        Piece *pp = nullptr;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] {
                for (size_t i = 2 * g; i < 2 * g + 2; i++)
                    if (!buf_[i].loaded && buf_[i].outstanding == 0 && !buf_[i].in_pipe && !buf_[i].tail_needed) {
                        pp = &buf_[i];
                        return true;
                    }
                return stop_ || done_;
            });
            if (!pp) return -2;
            pp->in_pipe = true;
        }
        Piece &p = *pp;
        if (nh::dev_set(ln.device) != hipSuccess) return fail("hipSetDevice failed");
        const long n = ln.host  ? gz_.take_host(use_ahead, p.d_text + head_, room_ - head_, ln.stream)
                       : ahead_ ? gz_.take((int)g, use_ahead, p.d_text + head_, room_ - head_, ln.stream)
                                : gz_.next_on((int)g, p.d_text + head_, room_ - head_, ln.stream);
        if (ahead_) {
            std::lock_guard<std::mutex> lk(mu_);
            ln.state = 0;
        }
        // The members' CRC-32 / ISIZE (and the stream's end) are the only end-to-end check on the device decode, and a member that
        // spans pieces has had its earlier pieces handed out, classified and written by the time its trailer is read: once records
        // are out, such a failure ends the RUN (the host reader would decode the file again, pass its own check and
        // the run would return NH_OK with whatever the device had decoded wrongly already in the outputs).  Nothing handed out
        // yet: the host reader starts from the top and says what it finds.
        if (n < 0) return fail(gz_.error(), gz_.integrity_failure() && handed_recs_.load() > 0);
        if (ahead_ && gz_.ahead_ok() && !gz_.ended()) assign_ahead();  // (this lane's next cell is decoded while its piece is indexed)
        p.body_len = (size_t)n;
        p.last = gz_.ended();
        last = p.last;
        piece_no_++;
        {
            std::lock_guard<std::mutex> lk(mu_);
            indexq_.push_back((size_t)(pp - &buf_[0]));
        }
        cv_.notify_all();
        return 0;
    }

    void index_stage() {
        for (;;) {
            Piece *pp = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return !indexq_.empty() || decode_done_ || stop_ || done_; });
                if (stop_ || done_ || indexq_.empty()) {
                    if (!done_) {  // (the decode stage ended without an error and everything it made is published)
                        done_ = true;
                        cv_.notify_all();
                    }
                    return;
                }
                pp = &buf_[indexq_.front()];
                indexq_.pop_front();
            }
            const int rc = index_piece(*pp);
            std::lock_guard<std::mutex> lk(mu_);
            if (rc != 0 || pp->last) {
                if (rc == 1) fallback_ = true;
                done_ = true;
                cv_.notify_all();
                return;
            }
        }
    }

    int index_piece(Piece &p) {
        Lane &ln = lanes_[(size_t)p.lane];
        if (nh::dev_set(ln.device) != hipSuccess) return fail("hipSetDevice failed");
        size_t carry = 0;
        if (last_ && last_->indexed) {
            // the records behind the last whole batch and the incomplete record behind them (from the lane before: over xGMI)
            Piece &old = *last_;
            const size_t full = max_text_ ? old.n_rec : old.n_rec / bf_ * bf_;  // (single-end: every complete record went out)
            const size_t from = full < old.n_rec ? old.h_bstart[full / bf_] : old.used_len;
            carry = old.text_len - from;
            if (carry > head_)  // (paired batches of very long records, or one record of hundreds of megabytes: the host reader's)
                return fail("a batch of " + std::to_string(bf_) + " records holds more text than the " + std::to_string(head_ >> 20) +
                            " MiB the reader keeps in front of a piece");
            const int odev = lanes_[(size_t)old.lane].device;
            uint8_t *const dst = p.d_text + head_ - carry;
            if (carry && (dev_copy_between(dst, ln.device, old.text0 + from, odev, carry, ln.stream_i) != hipSuccess ||
                          hipStreamSynchronize(ln.stream_i) != hipSuccess))
                return fail("D2D of the carried text failed");
            carried_ += carry;
            std::lock_guard<std::mutex> lk(mu_);
            old.tail_needed = false;
            cv_.notify_all();
        }
        p.text0 = p.d_text + head_ - carry;
        p.text_len = carry + p.body_len;
        p.n_rec = p.next_rec = 0;
        p.used_len = 0;
        p.indexed = false;
        pieces_++;
        const int irc = p.text_len ? index(p) : 0;
        if (irc != 0) return irc;
        p.indexed = true;
        last_ = &p;
        {
            std::lock_guard<std::mutex> lk(mu_);
            p.loaded = true;
            p.in_pipe = false;
            p.tail_needed = !p.last;
            order_.push_back((size_t)(&p - &buf_[0]));
        }
        cv_.notify_all();
        return 0;
    }

    // newlines -> lines -> records of the piece's text
    int index(Piece &p) {
        using namespace fq;
        Lane &ln = lanes_[(size_t)p.lane];
        hipStream_t const stream_ = ln.stream_i;
        unsigned long long *const d_bad_ = ln.d_bad, *const h_bad_ = ln.h_bad;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return p.prepared; });
        }
        const int dev = ln.device;
        dev_check(dev, "record index of a piece");
        dev_check_ptr(p.text0, dev, "record index (text)");
        const auto t0 = std::chrono::steady_clock::now();
        if (p.last) {  // the input's last line may lack its newline
            uint8_t lastc = 0;
            if (hipMemcpyAsync(&lastc, p.text0 + p.text_len - 1, 1, hipMemcpyDeviceToHost, stream_) != hipSuccess || hipStreamSynchronize(stream_) != hipSuccess)
                return fail("reading the text's last byte failed");
            if (lastc != '\n') {
                if (hipMemsetAsync(p.text0 + p.text_len, '\n', 1, stream_) != hipSuccess) return fail("hipMemsetAsync failed");
                p.text_len++;
            }
        }
        // newlines -> lines -> records
        const uint32_t n_tiles = (uint32_t)((p.text_len + TILE - 1) / TILE);
        if (!grow_dev(dev, p.d_tiles, p.tile_cap, (size_t)n_tiles + 1)) return fail("the record index's buffers cannot be had");
        hipLaunchKernelGGL(k_nl_count, dim3(n_tiles), dim3(256), 0, stream_, (const uint8_t *)p.text0, (uint64_t)p.text_len, p.d_tiles);
        hipLaunchKernelGGL(k_nl_scan, dim3(1), dim3(1024), 0, stream_, p.d_tiles, n_tiles);
        uint32_t n_lines = 0;
        if (hipMemcpyAsync(&n_lines, p.d_tiles + n_tiles, 4, hipMemcpyDeviceToHost, stream_) != hipSuccess || hipStreamSynchronize(stream_) != hipSuccess)
            return fail("the newline count could not be read");
        if (!grow_dev(dev, p.d_nl, p.nl_cap, (size_t)n_lines + 4)) return fail("the record index's buffers cannot be had");
        if (n_lines) hipLaunchKernelGGL(k_nl_list, dim3(n_tiles), dim3(256), 0, stream_, (const uint8_t *)p.text0, (uint64_t)p.text_len, (const uint32_t *)p.d_tiles, p.d_nl);
        size_t n_rec = n_lines / 4;
        if (n_rec) {
            const size_t nb = (n_rec + bf_ - 1) / bf_;
            if (n_rec > p.rec_cap) {
                size_t hc = p.rec_cap;
                if (!grow_dev(dev, p.d_recs, p.rec_cap, n_rec) || !grow_dev(dev, p.h_recs, hc, p.rec_cap, true))
                    return fail("the record table's buffers cannot be had");
                p.rec_cap = std::min(p.rec_cap, hc);
            }
            if (nb + 1 > p.bs_cap) {
                size_t hc = p.bs_cap;
                if (!grow_dev(dev, p.d_bstart, p.bs_cap, nb + 1) || !grow_dev(dev, p.h_bstart, hc, p.bs_cap, true))
                    return fail("the record table's buffers cannot be had");
                p.bs_cap = std::min(p.bs_cap, hc);
            }
            *h_bad_ = ~0ull;
            if (hipMemcpyAsync(d_bad_, h_bad_, 8, hipMemcpyHostToDevice, stream_) != hipSuccess) return fail("H2D failed");
            hipLaunchKernelGGL(k_records, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, stream_, (const uint8_t *)p.text0, (const uint32_t *)p.d_nl,
                               (uint32_t)n_rec, (uint32_t)bf_, p.d_recs, p.d_bstart, d_bad_);
            if (hipMemcpyAsync(h_bad_, d_bad_, 8, hipMemcpyDeviceToHost, stream_) != hipSuccess ||
                hipMemcpyAsync(p.h_recs, p.d_recs, n_rec * sizeof(RecRef), hipMemcpyDeviceToHost, stream_) != hipSuccess ||
                hipMemcpyAsync(p.h_bstart, p.d_bstart, (nb + 1) * 4, hipMemcpyDeviceToHost, stream_) != hipSuccess ||
                hipStreamSynchronize(stream_) != hipSuccess)
                return fail("the record table could not be read");
            if (*h_bad_ != ~0ull) {
                const size_t br = (size_t)(*h_bad_ >> 2);
                const unsigned kind = (unsigned)(*h_bad_ & 3);
                if (kind == 2) {
                    if (!handed_out_ && pieces_ == 1 && br == 0) return 1;  // no FASTQ at all (FASTA, ...): the host reader's business
                    // kraken2's message, with the line it saw
                    const size_t b = br / bf_;
                    const size_t at = (size_t)p.h_bstart[b] + p.h_recs[br].h;
                    std::vector<char> line(std::min<size_t>(p.h_recs[br].hlen, 200));
                    (void)hipMemcpy(line.data(), p.text0 + at, line.size(), hipMemcpyDeviceToHost);
                    return fail("malformed FASTQ file (exp. '@', saw \"" + std::string(line.begin(), line.end()) + "\"), aborting", true);
                }
                // an empty header line (or "@" alone): kraken2 stops reading there
                n_rec = br;
                p.last = true;
                if (n_rec) {
                    const size_t nb2 = (n_rec + bf_ - 1) / bf_;
                    const RecRef &lr = p.h_recs[n_rec - 1];
                    const size_t lb = (n_rec - 1) / bf_;
                    // the end of the last record that counts: its quality line's end (+ newline)
                    uint32_t endq = p.h_bstart[lb] + lr.q + lr.qlen;
                    p.h_bstart[nb2] = lr.raw_end ? p.h_bstart[lb] + lr.raw_end : endq + 1;
                }
            }
        }
        p.n_rec = n_rec;
        p.used_len = n_rec ? p.h_bstart[(n_rec + bf_ - 1) / bf_] : 0;
        if (p.last && p.used_len > p.text_len) p.used_len = p.text_len;
        records_ += n_rec;
        if (n_rec) handed_out_ = true;
        // FASTA, or nothing recognisable, in the very first piece: not ours
        if (!handed_out_ && pieces_ == 1 && p.text_len) {
            uint8_t c0 = 0;
            (void)hipMemcpy(&c0, p.text0, 1, hipMemcpyDeviceToHost);
            if (c0 != '@') return 1;
        }
        index_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    }

    DevGunzip gz_;
    std::string path_, error_;
    bool hard_ = false;            // error_ is the input's fault (malformed FASTQ): no handover to the host reader
    std::mutex err_mu_;            // error_ / hard_
    std::string reason_;           // why next_batch() handed the file over (a copy: the producers may still be ending)
    size_t max_text_ = 0;          // single-end: a batch is cut behind the record that reaches this many bytes (0: by records only)
    std::atomic<uint64_t> handed_recs_{0};  // records in the batches handed out (read by the decode stage when it fails)
    long fail_at_ = 0;             // test knob NOHUMAN_GZDEV_FAIL_AT=k: the reader gives up before the stream's k-th piece (k >= 1)
    std::vector<Lane> lanes_;
    std::vector<Piece> buf_;
    std::deque<size_t> order_;   // the pieces handed to the consumer, in stream order (indices into buf_)
    std::deque<size_t> indexq_;  // decoded, waiting for the index stage
    size_t head_ = 0;            // room in front of a piece's text for what is carried over from the piece before
    bool decode_done_ = false;
    Piece *last_ = nullptr;      // the piece produced last: what it could not hand out as whole batches goes in front of the next
    uint64_t piece_no_ = 0, next_cell_ = 0;
    size_t n_dev_lanes_ = 1;  // lanes_[0, n_dev_lanes_) are devices; one more behind them: the host lane (hybrid reader)
    bool ahead_ = false;
    size_t room_ = 0, bf_ = 0;
    std::thread th_;
    bool stop_ = false, done_ = false, fallback_ = false;
    bool handed_out_ = false, trace_ = false;
    uint64_t pieces_ = 0, records_ = 0, carried_ = 0;
    double index_s_ = 0, open_s_ = 0;
    std::mutex mu_;
    std::condition_variable cv_;
};

DevFastqReader::DevFastqReader() : impl_(new DevFastqImpl()) {}
DevFastqReader::~DevFastqReader() { delete impl_; }
int DevFastqReader::open(const char *path, const int *devices, int n_devices, std::string &err, unsigned host_threads) {
    return impl_->open(path, devices, n_devices, err, host_threads);
}
int DevFastqReader::next_batch(HalfBatch &hb, size_t max_recs, size_t max_text) { return impl_->next_batch(hb, max_recs, max_text); }
uint64_t DevFastqReader::records_handed() const { return impl_->records_handed(); }
const std::string &DevFastqReader::handover_reason() const { return impl_->handover_reason(); }
void DevFastqReader::close() { impl_->close(); }

}  // namespace nh
