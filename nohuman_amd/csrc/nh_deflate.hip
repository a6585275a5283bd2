// nh_deflate.hip -- gzip output on the GPU.  The reference compresses the kept reads with
// gzp's block-parallel gzip at the default level (/root/reference/src/compression.rs:214-233, called from
// src/main.rs:342-368) -- on the host cores, at some 15 MB/s per core for FASTQ text, which is what a
// `nohuman reads.fq.gz` run spends most of its time on once the classifier is a GPU kernel.  Here the text is cut
// into regions of 64 KiB and ONE WAVE compresses a region:
//
//   match finding   64 consecutive positions per step, a lane each: hash of 5 bytes -> a 4-way bucket (6 / 8 by
//                   NOHUMAN_GZIP_WAYS) of earlier positions (16 bits each, in LDS), plus distance 1 (runs); compared in
//                   lockstep rounds of 16 bytes, the longest is taken if it saves bits under the PREVIOUS block's code
//                   lengths (match_probe + match_finish, nh_deflate_core.h: the bucket read and the first round of
//                   gathers of step s + 1 are issued before step s's parse).
//   parse           lazy rule by a lane shift, then the chain of tokens through the step with v_readlane; a match
//                   that reached the scan cap is extended by the whole wave at once (8 bytes a lane).
//   block           every 32 KiB of input: symbol counts (LDS atomics) -> rank sort by the wave -> code lengths
//                   (Moffat-Katajainen, length-limited), canonical codes, the run-length coded header; the tokens
//                   (kept in HBM, 4 bytes each) are then coded 64 at a time: bit lengths -> wave prefix sum ->
//                   ds_or into a staging row -> coalesced dword stores.  A block that would not shrink is stored.
//   region end      an empty stored block byte-aligns the stream (what pigz / gzp do between their blocks), so the
//                   regions' streams concatenate; a small kernel packs them for one D2H copy.
//
// The host side (GpuGzipEncoder, a StreamEncoder of nh_codec.h) stages the writer's spans in page-locked chunks,
// keeps two chunks in flight, joins the regions' CRC-32s (computed by the same waves, a byte-table look-up a byte) and writes header, streams and
// trailer: one ordinary gzip member.  Parity target is the decompressed content (compression.rs:282-288), checked
// by zlib and by this repo's own reader in tests/test_gpu_deflate.py.
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "nh_codec.h"
#include "nh_deflate_core.h"
#include "nh_inflate.h"
#include "nh_internal.h"
#include "nohuman_engine.h"

namespace nh {
namespace dfl {

constexpr uint32_t BLOCK_IN = 32768;           // input bytes after which a block ends
constexpr uint32_t TOK_CAP = BLOCK_IN + 512;   // tokens a block can hold (one per byte at worst, plus the last step)
constexpr uint32_t OB_WORDS = 136;             // staging row of the bit writer
constexpr uint32_t PRIOR_BYTES = NLIT + NDIST; // code lengths handed from chunk to chunk: the match finder's prices
// The phase timers of NOHUMAN_GZIP_PROF keep a dozen scalars alive across the step loop, in a kernel that spills scalar registers
// already: they are compiled in by -DNH_DFL_PROF=1 only (make ab-dflprof -> tools/ab_engine_dflprof.so, NOHUMAN_ENGINE_LIB)
#ifndef NH_DFL_PROF
#define NH_DFL_PROF 0
#endif

struct DeflateArgs {
    const uint8_t *in;      // the text; readable up to n + 64
    uint64_t n;
    uint32_t region;        // bytes per region (<= MAX_REGION)
    uint32_t n_regions;
    uint8_t *slots;         // n_regions output slots, slot_stride bytes apart (4-byte aligned)
    uint32_t slot_stride;
    uint32_t *sizes;        // bytes written per region
    uint32_t *crcs;         // CRC-32 of every region's text
    uint32_t *tok;          // n_regions x TOK_CAP
    const uint8_t *prior;   // PRIOR_BYTES: literal/length lengths then distance lengths; the starting prices
    uint8_t *prior_out;     // region 0 leaves its last block's lengths here
    unsigned long long *prof;  // NULL, or 8 counters the waves add their phase times to (NOHUMAN_GZIP_PROF)
};

// Inclusive prefix sum over the wave by data-parallel primitives (DPP): four shifts inside the rows of sixteen lanes, then row 0's
// and rows 0-1's totals broadcast into the rows above (row_bcast:15 / :31, gfx9) -- six vector instructions and no LDS traffic,
// where the __shfl_up form was six ds_bpermute round trips (the bit writer runs one of these per 64 tokens).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v, 0), 63);
}
__device__ __forceinline__ uint32_t readlane_u(uint32_t v, uint32_t l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)l));
}

struct BitOut {
    uint32_t *ob;     // LDS staging row, OB_WORDS
    uint32_t *out;    // the region's slot
    uint32_t out_dw;  // dwords of the slot already written
    uint32_t obits;   // bits waiting in ob[0]
};
__device__ __forceinline__ void or_bits(uint32_t *ob, uint32_t off, uint32_t v, uint32_t n) {
    const uint32_t w = off >> 5, sh = off & 31u;
    const uint64_t vv = (uint64_t)(n >= 32u ? v : (v & ((1u << n) - 1u))) << sh;
    atomicOr(&ob[w], (uint32_t)vv);
    if (vv >> 32) atomicOr(&ob[w + 1], (uint32_t)(vv >> 32));
}
// every lane appends piece a (na bits) then piece b (nb bits), lanes in order; na, nb <= 32
__device__ void wave_put2(BitOut &bo, int lane, uint32_t a, uint32_t na, uint32_t b, uint32_t nb) {
    const uint32_t tot = na + nb;
    const uint32_t incl = wave_incl_scan(tot, lane);
    const uint32_t total = readlane_u(incl, 63) + bo.obits;
    const uint32_t off = bo.obits + incl - tot;
    if (na) or_bits(bo.ob, off, a, na);
    if (nb) or_bits(bo.ob, off + na, b, nb);
    __syncthreads();
    const uint32_t nfull = total >> 5;
    for (uint32_t j = (uint32_t)lane; j < nfull; j += 64) bo.out[bo.out_dw + j] = bo.ob[j];
    const uint32_t rem = bo.ob[nfull];
    __syncthreads();
    for (uint32_t j = (uint32_t)lane; j <= nfull; j += 64) bo.ob[j] = j == 0 ? rem : 0u;
    __syncthreads();
    bo.out_dw += nfull;
    bo.obits = total & 31u;
}

// symbol counts of a block: two 16-bit counters a word (a block has fewer than 65536 tokens), bumped by LDS atomics
struct PackedCounts {
    uint32_t *w;
    __device__ __forceinline__ uint32_t get(int s) const { return (w[s >> 1] >> (16 * (s & 1))) & 0xFFFFu; }
    __device__ __forceinline__ void bump(int s) const { atomicAdd(&w[s >> 1], 1u << (16 * (s & 1))); }
};
struct PlainCounts {
    uint32_t *w;
    __device__ __forceinline__ uint32_t get(int s) const { return w[s]; }
    __device__ __forceinline__ void bump(int s) const { atomicAdd(&w[s], 1u); }
};
struct TreeLds {  // scratch of the code construction: the used symbols in ascending order of their counts
    uint16_t a[NLIT];
    uint16_t ssym[NLIT];
};

// Code lengths and codes of one alphabet from its counts (all in LDS).  The wave rank-sorts the used symbols; lane 0
// runs the three passes of the minimum-redundancy construction over them (the only serial part: a few dozen symbols
// on FASTQ text); the histogram of depths, the Kraft repair, the lengths' way back to the symbols and the canonical
// codes are wave-parallel, with lane l holding what belongs to length l.
template <typename Counts>
__device__ void build_tree_wave(const Counts &freq, int nsym, int maxbits, uint8_t *lens, uint16_t *codes, uint16_t *ta,
                                uint16_t *tsym, int lane) {
    // a code needs two symbols to be complete
    uint32_t used = 0;
    for (int s = lane; s < nsym; s += 64) used += freq.get(s) != 0;
    used = wave_sum(used);
    if (used < 2 && lane == 0)
        for (int s = 0; used < 2 && s < nsym; s++)
            if (!freq.get(s)) {
                freq.bump(s);
                used++;
            }
    used = used < 2 ? 2 : used;
    __syncthreads();
    // rank sort by (count, symbol): a lane holds the keys count << 9 | symbol of its (up to five) symbols in registers; every
    // USED symbol's key goes round by v_readlane and each lane counts the smaller ones.  (The first version read all nsym counts
    // from LDS for every lane: 1430 dependent-latency iterations for the literal code, a fifth of what a block's end cost;
    // FASTQ text uses about a hundred of the 286 symbols.)
    constexpr int SLOTS = (NLIT + 63) / 64;
    uint32_t key[SLOTS], rank[SLOTS];
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
        const int s = lane + 64 * k;
        if (s < nsym) lens[s] = 0;
        const uint32_t f = s < nsym ? freq.get(s) : 0u;
        key[k] = f ? (f << 9) | (uint32_t)s : 0xFFFFFFFFu;
        rank[k] = 0;
    }
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
        if (64 * k >= nsym) break;
        uint64_t m = __ballot(key[k] != 0xFFFFFFFFu);
        while (m) {
            const uint32_t l = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            const uint32_t g = readlane_u(key[k], l);
#pragma unroll
            for (int q = 0; q < SLOTS; q++) rank[q] += g < key[q];
        }
    }
#pragma unroll
    for (int k = 0; k < SLOTS; k++)
        if (key[k] != 0xFFFFFFFFu) {
            ta[rank[k]] = (uint16_t)(key[k] >> 9);
            tsym[rank[k]] = (uint16_t)(key[k] & 511u);
        }
    __syncthreads();
    const int n = (int)used;
    if (lane == 0) huff_depths_sorted(ta, n);  // ta[i] = depth of the i-th rarest symbol
    __syncthreads();
    // lane l: how many symbols have depth l (deeper ones counted at maxbits)
    uint32_t my_cnt = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const uint32_t dep = i < n ? (ta[i] > (uint32_t)maxbits ? (uint32_t)maxbits : (uint32_t)ta[i]) : 0u;
#pragma unroll
        for (int l = 1; l <= MAXBITS; l++) {
            const uint64_t b = __ballot(dep == (uint32_t)l);
            if (lane == l) my_cnt += (uint32_t)__popcll(b);
        }
    }
    // Kraft sum in units of 2^-maxbits; while it is above one: a code of the longest length is given up and the
    // longest shorter code is lengthened by one bit to take its sibling's place -- one unit less each time
    uint32_t total = wave_sum((lane >= 1 && lane <= maxbits) ? my_cnt << (maxbits - lane) : 0u);
    while (total > (1u << maxbits)) {
        if (lane == maxbits) my_cnt--;
        const uint64_t have = __ballot(my_cnt != 0u && lane >= 1 && lane < maxbits);
        const int l = 63 - __builtin_clzll(have);
        if (lane == l) my_cnt--;
        if (lane == l + 1) my_cnt += 2;
        total--;
    }
    // the rarest symbols take the longest codes: sorted index i lies in the stretch of its length
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        uint32_t cum = 0, mine = 0;
#pragma unroll
        for (int l = MAXBITS; l >= 1; l--) {
            const uint32_t c = readlane_u(my_cnt, (uint32_t)l);
            if ((uint32_t)i >= cum && (uint32_t)i < cum + c) mine = (uint32_t)l;
            cum += c;
        }
        if (i < n) lens[tsym[i]] = (uint8_t)mine;
    }
    __syncthreads();
    // canonical codes (RFC 1951 3.2.2): lane l holds the first code of length l and the symbols seen so far with it
    uint32_t my_next = 0, my_run = 0;
    {
        uint32_t code = 0, prev = 0;
#pragma unroll
        for (int l = 1; l <= MAXBITS; l++) {
            code = (code + prev) << 1;
            if (lane == l) my_next = code;
            prev = readlane_u(my_cnt, (uint32_t)l);
        }
    }
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int base = 0; base < nsym; base += 64) {
        const int s = base + lane;
        const uint32_t ln = s < nsym ? lens[s] : 0u;
        const uint32_t first = (uint32_t)__shfl((int)my_next, (int)ln), before = (uint32_t)__shfl((int)my_run, (int)ln);
        uint32_t rank = 0;
#pragma unroll
        for (int l = 1; l <= MAXBITS; l++) {
            const uint64_t b = __ballot(ln == (uint32_t)l);
            if (ln == (uint32_t)l) rank = (uint32_t)__popcll(b & lt);
            if (lane == l) my_run += (uint32_t)__popcll(b);
        }
        if (s < nsym) codes[s] = ln ? (uint16_t)bitrev(first + before + rank, ln) : (uint16_t)0;
    }
    __syncthreads();
}

template <int WAYS>
struct __attribute__((aligned(16))) RegionLds {
    uint16_t bucket[WAYS << BUCKET_BITS];
    uint32_t lfreq2[NLIT / 2];  // PackedCounts
    uint32_t dfreq2[NDIST / 2];
    uint32_t clfreq[32];
    uint8_t llen[NLIT];   // lengths of the last block built: the codes' lengths while coding
    uint8_t dlen[NDIST];
    uint8_t lprice[NLIT]; // what the match finder prices with: the same lengths, an absent symbol at its penalty (CostsT<true>)
    uint8_t dprice[NDIST];
    uint8_t cllen[32];
    uint16_t lcode[NLIT];
    uint16_t dcode[NDIST];
    uint16_t clcode[32];
    union {
        TreeLds tree;                     // while a code is built
        struct {
            uint8_t all[NLIT + NDIST];     // afterwards: hlit + hdist lengths in a row
            uint16_t items[NLIT + NDIST];  // and their run-length form
        } hdr;
        uint32_t crct[256];                // before the first block: the byte table of the region's CRC-32
    } u;
    uint16_t cl_a[32], cl_sym[32];        // the code-length code is built while hdr is live
    uint32_t ob[OB_WORDS];
    int misc[8];
};

// Ends a block: builds the codes from the counts and writes header + tokens (or the bytes, stored).
template <int WAYS>
__device__ void finish_block(RegionLds<WAYS> &S, BitOut &bo, const uint32_t *tok, uint32_t ntok, const uint8_t *src,
                             uint32_t from, uint32_t to, int lane, unsigned long long *prof) {
    if (!NH_DFL_PROF) prof = nullptr;
    const unsigned long long f0 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
    const PackedCounts lfreq{S.lfreq2}, dfreq{S.dfreq2};
    if (lane == 0) lfreq.bump(256);
    __syncthreads();
    build_tree_wave(lfreq, NLIT_USED, MAXBITS, S.llen, S.lcode, S.u.tree.a, S.u.tree.ssym, lane);
    const unsigned long long f1 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
    build_tree_wave(dfreq, NDIST_USED, MAXBITS, S.dlen, S.dcode, S.u.tree.a, S.u.tree.ssym, lane);
    const unsigned long long f2 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
    {
        // hlit / hdist: one past the last used symbol; the two rows of lengths in a row, then their run-length form
        uint32_t top_l = 0, top_d = 0;
        for (int s0 = lane; s0 < NLIT_USED; s0 += 64)
            if (S.llen[s0]) top_l = (uint32_t)s0 + 1u;
        if (lane < NDIST_USED && S.dlen[lane]) top_d = (uint32_t)lane + 1u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t a1 = (uint32_t)__shfl_xor((int)top_l, o), a2 = (uint32_t)__shfl_xor((int)top_d, o);
            top_l = a1 > top_l ? a1 : top_l;
            top_d = a2 > top_d ? a2 : top_d;
        }
        const int hlit_ = (int)(top_l < 257u ? 257u : top_l), hdist_ = (int)(top_d < 1u ? 1u : top_d);
        for (int i0 = lane; i0 < hlit_; i0 += 64) S.u.hdr.all[i0] = S.llen[i0];
        if (lane < hdist_) S.u.hdr.all[hlit_ + lane] = S.dlen[lane];
        __syncthreads();
        if (lane == 0) {
            S.misc[0] = hlit_;
            S.misc[1] = hdist_;
            S.misc[2] = rle_lengths(S.u.hdr.all, hlit_ + hdist_, S.u.hdr.items, S.clfreq);
        }
    }
    __syncthreads();
    const unsigned long long f3 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
    build_tree_wave(PlainCounts{S.clfreq}, NCL, MAXBITS_CL, S.cllen, S.clcode, S.cl_a, S.cl_sym, lane);
    if (prof && lane == 0) {
        const unsigned long long f4 = __builtin_amdgcn_s_memtime();
        atomicAdd(&prof[5], f4 - f0);
        atomicAdd(&prof[6], f1 - f0);
        atomicAdd(&prof[7], f2 - f1);
        atomicAdd(&prof[8], f3 - f2);
        atomicAdd(&prof[9], f4 - f3);
    }
    const int hlit = S.misc[0], hdist = S.misc[1], ni = S.misc[2];
    int hclen = NCL;
    while (hclen > 4 && S.cllen[cl_order(hclen - 1)] == 0) hclen--;
    // size of the block with these codes
    uint32_t bits = 0;
    for (int i = lane; i < ni; i += 64) {
        const uint32_t s = S.u.hdr.items[i] & 31u;
        bits += S.cllen[s] + cl_extra_bits(s);
    }
    for (int s = lane; s < NLIT_USED; s += 64) bits += lfreq.get(s) * (S.llen[s] + (s > 256 ? len_extra_bits((uint32_t)s) : 0u));
    for (int s = lane; s < NDIST_USED; s += 64) bits += dfreq.get(s) * (S.dlen[s] + dist_extra_bits((uint32_t)s));
    bits = wave_sum(bits) + 17u + 3u * (uint32_t)hclen;
    const uint32_t nbytes = to - from;
    if (bits >= 8u * nbytes + 40u) {
        // stored: 3 header bits, padding to a byte, LEN, NLEN, the bytes
        const uint32_t pos = (bo.out_dw * 32u + bo.obits + 3u) & 7u;
        const uint32_t pad = pos ? 8u - pos : 0u;
        wave_put2(bo, lane, 0u, lane == 0 ? 3u + pad : 0u, (nbytes & 0xFFFFu) | ((~nbytes & 0xFFFFu) << 16), lane == 0 ? 32u : 0u);
        for (uint32_t i = 0; i < nbytes; i += 64) {
            const bool in = i + (uint32_t)lane < nbytes;
            wave_put2(bo, lane, in ? src[from + i + lane] : 0u, in ? 8u : 0u, 0u, 0u);
        }
        return;
    }
    // dynamic block: BFINAL 0, BTYPE 2, hlit, hdist, hclen | the code-length code's lengths | the items
    {
        const uint32_t head = 4u | ((uint32_t)(hlit - 257) << 3) | ((uint32_t)(hdist - 1) << 8) | ((uint32_t)(hclen - 4) << 13);
        wave_put2(bo, lane, head, lane == 0 ? 17u : 0u, 0u, 0u);
        wave_put2(bo, lane, lane < hclen ? S.cllen[cl_order(lane < NCL ? lane : 0)] : 0u, lane < hclen ? 3u : 0u, 0u, 0u);
        for (int i = 0; i < ni; i += 64) {
            const bool in = i + lane < ni;
            const uint32_t it = in ? S.u.hdr.items[i + lane] : 0u;
            const uint32_t s = it & 31u;
            wave_put2(bo, lane, S.clcode[s], in ? S.cllen[s] : 0u, it >> 5, in ? cl_extra_bits(s) : 0u);
        }
    }
    // (the tokens were written by this wave's own stores: read past the L1; the next round's are fetched a round early)
    uint32_t t_next = (uint32_t)lane < ntok ? __hip_atomic_load(&tok[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    for (uint32_t i = 0; i < ntok; i += 64) {
        const bool in = i + (uint32_t)lane < ntok;
        const uint32_t t = t_next;
        if (i + 64u + (uint32_t)lane < ntok)
            t_next = __hip_atomic_load(&tok[i + 64u + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t a = 0, na = 0, b = 0, nb = 0;
        if (in) {
            if (t & 0x80000000u) {
                uint32_t sym, eb, ev, dsym, deb, dev;
                len_symbol((t >> 16) & 0xFFu, sym, eb, ev);
                dist_symbol(t & 0x7FFFu, dsym, deb, dev);
                na = S.llen[sym];
                a = S.lcode[sym] | (ev << na);
                na += eb;
                nb = S.dlen[dsym];
                b = S.dcode[dsym] | (dev << nb);
                nb += deb;
            } else {
                a = S.lcode[t];
                na = S.llen[t];
            }
        }
        wave_put2(bo, lane, a, na, b, nb);
    }
    wave_put2(bo, lane, S.lcode[256], lane == 0 ? S.llen[256] : 0u, 0u, 0u);
}

template <int WAYS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_deflate(DeflateArgs a) {
    __shared__ RegionLds<WAYS> S;
    const int lane = (int)threadIdx.x;
    const uint32_t r = blockIdx.x;
    const uint64_t base = (uint64_t)r * a.region;
    const uint32_t n = (uint32_t)((a.n - base) < (uint64_t)a.region ? (a.n - base) : (uint64_t)a.region);
    const uint8_t *src = a.in + base;
    uint32_t *tok = a.tok + (size_t)r * TOK_CAP;
    for (uint32_t i = (uint32_t)lane; i < (uint32_t)(WAYS << BUCKET_BITS); i += 64) S.bucket[i] = (uint16_t)EMPTY_ENTRY;
    for (int i = lane; i < NLIT; i += 64) {
        if (i < NLIT / 2) S.lfreq2[i] = 0;
        S.llen[i] = a.prior[i];
        S.lprice[i] = price_of_litlen((uint32_t)i, a.prior[i]);
    }
    if (lane < NDIST) {
        if (lane < NDIST / 2) S.dfreq2[lane] = 0;
        S.dlen[lane] = a.prior[NLIT + lane];
        S.dprice[lane] = price_of_dist(a.prior[NLIT + lane]);
    }
    // the CRC's table: entry b = eight shifts of b (RFC 1952's polynomial), four entries a lane
    for (uint32_t i = (uint32_t)lane; i < 256u; i += 64u) {
        uint32_t c = i;
#pragma unroll
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (CRC_POLY & (0u - (c & 1u)));
        S.u.crct[i] = c;
    }
    const PackedCounts lfreq{S.lfreq2}, dfreq{S.dfreq2};
    for (uint32_t i = (uint32_t)lane; i < OB_WORDS; i += 64) S.ob[i] = 0;
    __syncthreads();
    // CRC-32 of the region's text: a slice per lane, a byte a look-up in the table above (bit by bit it was 40 vector
    // instructions per 64 bytes of text, 7 % of the kernel's; it also brings the text into the L2 before the match finder
    // asks for it), the slices joined by one multiplication each
    {
        const uint32_t slice = (((n + 63u) >> 6) + 15u) & ~15u;
        const uint32_t start = (uint32_t)lane * slice;
        const uint32_t len = start >= n ? 0u : (n - start < slice ? n - start : slice);
        uint32_t crc = 0xFFFFFFFFu;
        const uint8_t *sp = src + start;
        const uint32_t *T = S.u.crct;
        uint32_t i = 0;
        for (; i + 16u <= len; i += 16u) {
            const uint4 v = *(const uint4 *)(sp + i);  // (region bases and slices are multiples of 16)
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                crc ^= w[q];
#pragma unroll
                for (int k = 0; k < 4; k++) crc = T[crc & 0xFFu] ^ (crc >> 8);
            }
        }
        for (; i < len; i++) crc = T[(crc ^ sp[i]) & 0xFFu] ^ (crc >> 8);
        crc = len ? ~crc : 0u;
        uint32_t t = len ? gf2_mul(crc, gf2_xpow8((uint64_t)(n - (start + len)))) : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t ^= (uint32_t)__shfl_xor((int)t, o);
        if (lane == 0) a.crcs[r] = t;
    }
    BitOut bo{S.ob, (uint32_t *)(a.slots + (size_t)r * a.slot_stride), 0u, 0u};
    uint32_t carry = 0, ntok = 0, blk_from = 0;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    Bytes16 next16 = load16(src + lane);
    // the candidates of the step to come: the bucket read and the first round's gathers of step s + 1 are issued before step
    // s's parse (below), the compare half runs at the top of step s + 1
    MatchProbe<1 + WAYS> pr;
    match_probe<WAYS>(src, (uint32_t)lane, n, &S.bucket[WAYS * hash_bucket(hash_at(next16.lo))], pr);
    unsigned long long *const prof = NH_DFL_PROF ? a.prof : nullptr;
    unsigned long long t_match = 0, t_parse = 0, t_block = 0, t_all = prof ? __builtin_amdgcn_s_memtime() : 0ull;
    for (uint32_t s = 0; s < n; s += 64) {
        const unsigned long long c0 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
        const uint32_t p = s + (uint32_t)lane;
        const bool inside = p < n;
        const bool any = carry < s + 64u;
        // the sixteen bytes at this lane's position were loaded during the previous step
        const Bytes16 cur16 = next16;
        if (s + 64u < n) next16 = load16(src + p + 64u);
        const uint32_t lit = (uint32_t)cur16.lo & 0xFFu;  // the byte at this lane's position
        const bool hashed = p + HASH_BYTES <= n;  // positions with a full hash context enter the buckets
        const uint32_t h = hash_at(cur16.lo);
        uint32_t L = 0, D = 0;
        if (any && inside && p >= carry) {
            const CostsT<true> costs{S.lprice, S.dprice};
            int gain = 0;
            L = match_finish<WAYS>(src, p, n, cur16, pr, costs, D, gain);
        }
        __syncthreads();
        const unsigned long long c1 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
        if (hashed) S.bucket[WAYS * hash_bucket(h) + ((p >> 6) % WAYS)] = make_entry(p);
        // (LDS operations of one wave execute in order: the next step's bucket reads see this step's entries, as they did
        //  when they were issued at the top of that step)
        if (s + 64u < n) match_probe<WAYS>(src, p + 64u, n, &S.bucket[WAYS * hash_bucket(hash_at(next16.lo))], pr);
        if (any) {
            // lazy rule: a longer match one position on wins over a short one here
            const uint32_t nx = (uint32_t)__shfl_down((int)L, 1);
            if (L && L < 16u && lane < 63 && nx > L) L = 0;
            uint64_t sel = 0;
            uint32_t q = carry > s ? carry - s : 0u;
            const uint32_t qend = n - s < 64u ? n - s : 64u;
            const uint64_t mm = __ballot(L != 0);
            while (q < qend) {
                // literals up to the next position with a match are taken in one go
                const uint64_t ahead = mm >> q;
                const uint32_t m = ahead ? q + (uint32_t)__builtin_ctzll(ahead) : 64u;
                const uint32_t stop = m < qend ? m : qend;
                if (stop > q) {
                    sel |= (stop >= 64u ? ~0ull : (1ull << stop) - 1ull) & ~((1ull << q) - 1ull);
                    q = stop;
                    continue;
                }
                uint32_t l = readlane_u(L, q);
                if (l == SCAN_CAP) {  // the wave measures the rest of the match: 8 bytes a lane
                    const uint32_t d = readlane_u(D, q), cpos = s + q;
                    const uint32_t maxl = n - cpos < MAX_MATCH ? n - cpos : MAX_MATCH;
                    const uint32_t off = SCAN_CAP + 8u * (uint32_t)lane;
                    const bool within = off < maxl;
                    const uint64_t x = within ? (load8(src + cpos + off) ^ load8(src + cpos - d + off)) : 1ull;
                    const uint64_t stopm = __ballot(x != 0);
                    const uint32_t f = (uint32_t)__builtin_ctzll(stopm);
                    uint32_t tl = off + (within ? (uint32_t)__builtin_ctzll(x) >> 3 : 0u);
                    tl = readlane_u(tl, f);
                    l = tl < maxl ? tl : maxl;
                    if ((uint32_t)lane == q) L = l;
                }
                sel |= 1ull << q;
                q += l;
            }
            carry = s + q;
            if ((sel >> lane) & 1ull) {
                const uint32_t idx = ntok + (uint32_t)__popcll(sel & lt_mask);
                if (L) {
                    uint32_t sym, eb, ev, dsym, deb, dev;
                    len_symbol(L - 3u, sym, eb, ev);
                    dist_symbol(D - 1u, dsym, deb, dev);
                    tok[idx] = tok_match(L, D);
                    lfreq.bump((int)sym);
                    dfreq.bump((int)dsym);
                } else {
                    tok[idx] = lit;
                    lfreq.bump((int)lit);
                }
            }
            ntok += (uint32_t)__popcll(sel);
        }
        __syncthreads();
        const unsigned long long c2 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
        t_match += c1 - c0;
        t_parse += c2 - c1;
        if (carry - blk_from >= BLOCK_IN || s + 64u >= n) {
            const uint32_t to = carry < n ? carry : n;
            __threadfence_block();
            finish_block<WAYS>(S, bo, tok, ntok, src, blk_from, to, lane, prof);
            for (int i = lane; i < NLIT / 2; i += 64) S.lfreq2[i] = 0;
            if (lane < NDIST / 2) S.dfreq2[lane] = 0;
            for (int i = lane; i < NLIT; i += 64) S.lprice[i] = price_of_litlen((uint32_t)i, S.llen[i]);  // (built even when the block went out stored)
            if (lane < NDIST) S.dprice[lane] = price_of_dist(S.dlen[lane]);
            __syncthreads();
            blk_from = to;
            ntok = 0;
            if (prof) t_block += __builtin_amdgcn_s_memtime() - c2;
        }
    }
    if (prof && lane == 0) {
        atomicAdd(&prof[0], t_match);
        atomicAdd(&prof[1], t_parse);
        atomicAdd(&prof[2], t_block);
        atomicAdd(&prof[3], __builtin_amdgcn_s_memtime() - t_all);
        atomicAdd(&prof[4], 1ull);
    }
    // the region ends on a byte boundary: an empty stored block (BFINAL 0, BTYPE 0, padding, LEN 0, NLEN 0xFFFF)
    {
        const uint32_t pos = (bo.out_dw * 32u + bo.obits + 3u) & 7u;
        const uint32_t pad = pos ? 8u - pos : 0u;
        wave_put2(bo, lane, 0u, lane == 0 ? 3u + pad : 0u, 0xFFFF0000u, lane == 0 ? 32u : 0u);
    }
    if (lane == 0) {
        if (bo.obits) bo.out[bo.out_dw] = S.ob[0];
        a.sizes[r] = bo.out_dw * 4u + (bo.obits >> 3);
    }
    if (r == 0 && a.prior_out) {
        for (int i = lane; i < NLIT; i += 64) a.prior_out[i] = S.llen[i];
        if (lane < NDIST) a.prior_out[NLIT + lane] = S.dlen[lane];
    }
}

// offsets[i] = bytes of the regions before i; offsets[n] = all.  One wave: a stretch of regions per lane, the
// stretches' sums scanned across the lanes.
__global__ __launch_bounds__(64) void k_deflate_offsets(const uint32_t *sizes, uint32_t n, uint64_t *offsets) {
    if (blockIdx.x != 0) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t per = (n + 63u) / 64u;
    const uint32_t lo = lane * per < n ? lane * per : n, hi = lo + per < n ? lo + per : n;
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += sizes[i];
    uint64_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t t = (uint64_t)__shfl_up((long long)incl, o);
        if ((int)lane >= o) incl += t;
    }
    uint64_t acc = incl - sum;
    for (uint32_t i = lo; i < hi; i++) {
        offsets[i] = acc;
        acc += sizes[i];
    }
    if (lane == 63) offsets[n] = incl;
}
__global__ __launch_bounds__(256) void k_deflate_pack(const uint8_t *slots, uint32_t slot_stride, const uint32_t *sizes,
                                                       const uint64_t *offsets, uint8_t *out) {
    const uint32_t r = blockIdx.x;
    const uint8_t *s = slots + (size_t)r * slot_stride;
    uint8_t *d = out + offsets[r];
    const uint32_t n = sizes[r];
    for (uint32_t i = threadIdx.x; i < n; i += 256) d[i] = s[i];
}

}  // namespace dfl

// ---- host side --------------------------------------------------------------------------------------------------
namespace {

using dfl::DeflateArgs;
using dfl::crc32_join;

int gzip_ways() {
    static const int w = [] {
        const char *e = getenv("NOHUMAN_GZIP_WAYS");
        // four ways since round 4: 12 KB of LDS a wave = three waves a SIMD, 27.2 GB/s against 20.3 with eight ways (two waves),
        // the run of 50 M gzip pairs to gzip 15 % shorter, the files 1.4 % larger (4.054 : 1 against 4.112; zlib -6: 4.299);
        // NOHUMAN_GZIP_WAYS=6 | 8 for the smaller files (profiles/r04_e2e.txt)
        const int v = e ? atoi(e) : 4;
        return v == 8 ? 8 : v == 6 ? 6 : 4;
    }();
    return w;
}

// page-locked host memory; a host that will not lock that much gets pageable memory (the copies accept it and stage
// it themselves).  A 64-byte header says which kind a block is.  NOHUMAN_NO_PINNED exercises the fallback.
void *host_alloc(size_t n) {
    void *p = nullptr;
    static const bool no_pin = getenv("NOHUMAN_NO_PINNED") != nullptr;
    if (!no_pin && host_malloc(&p, n + 64, hipHostMallocDefault) == hipSuccess) {
        *(uint64_t *)p = 1;
        return (char *)p + 64;
    }
    (void)hipGetLastError();
    if (posix_memalign(&p, 64, n + 64) != 0) return nullptr;
    *(uint64_t *)p = 2;
    return (char *)p + 64;
}
void host_free(void *q) {
    if (!q) return;
    void *p = (char *)q - 64;
    if (*(uint64_t *)p == 1)
        (void)hipHostFree(p);
    else
        free(p);
}

struct DeflateDev {  // device side of one encoder: buffers for chunks in flight
    // Text per kernel launch: 128 MiB = 2048 regions, a wave each.  The chip holds 3072 of this kernel's waves, and 192 MiB chunks do
    // make the encoder ALONE faster (34.2 -> 39.1 GB/s of kernels) -- but a run with two output files has two encoders' chunks in flight
    // already and pays for the coarser pipeline (20 M pairs gzip -> gzip: 1.395 s at 128, 1.475 at 192, 1.512 at 256), and the one-file
    // long-read run gains 2-4 %, inside its noise (profiles/r06_gzip_summary.txt).  NOHUMAN_GZIP_CHUNK_MB: 64 .. 512.
    static size_t chunk_bytes_now() {  // (read when an encoder is set up: a test may ask for small chunks)
        const char *e = getenv("NOHUMAN_GZIP_CHUNK_MB");
        const long mb = e ? atol(e) : 128;
        return (size_t)(mb < 64 ? 64 : mb > 512 ? 512 : mb) << 20;
    }
    size_t chunk = 0;  // this encoder's
    static uint32_t region_bytes() {  // NOHUMAN_GZIP_REGION: tuning (bytes of text a wave compresses)
        static const uint32_t r = [] {
            const char *e = getenv("NOHUMAN_GZIP_REGION");
            const long v = e ? atol(e) : 65536;
            return (uint32_t)(v < 4096 ? 4096 : v > (long)dfl::MAX_REGION ? (long)dfl::MAX_REGION : (v & ~63L));
        }();
        return r;
    }
    const uint32_t REGION = region_bytes();
    static constexpr int NBUF = 2;
    static constexpr uint32_t PILOT_REGIONS = 16;
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // brings the text into d_in while the kernels of earlier chunks run
    hipStream_t out_stream = nullptr;   // brings a finished chunk's stream to the host (never behind a later chunk's kernels)
    struct Buf {
        uint8_t *h_in = nullptr;   // page-locked staging of the text
        uint8_t *d_in = nullptr;
        uint8_t *d_slots = nullptr;
        uint32_t *d_sizes = nullptr;
        uint32_t *d_crcs = nullptr;
        uint32_t *h_crcs = nullptr;  // page-locked
        uint64_t *d_offsets = nullptr;
        uint8_t *d_out = nullptr;
        uint8_t *h_out = nullptr;  // page-locked
        uint64_t *h_total = nullptr;
        hipEvent_t done = nullptr, k0 = nullptr, k1 = nullptr, filled = nullptr;
        size_t fill = 0;
        size_t staged_from = 0;    // [staged_from, fill) of h_in is not on the device yet
        size_t submitted = 0;      // bytes of text in the chunk that is in flight
        uint32_t n_regions = 0;
        bool in_flight = false;
    } buf[NBUF];
    uint32_t *d_tok = nullptr;
    uint8_t *d_prior = nullptr;  // two rows, alternating
    unsigned long long *d_prof = nullptr;
    uint32_t slot_stride = 0, max_regions = 0;
    uint64_t chunks = 0;
    double kernel_ms = 0;

    int fail(hipError_t e, const char *what) { return set_error(NH_EDEVICE, "gzip encoder: %s: %s", what, hipGetErrorString(e)); }

    int init(int dev) {
        device = dev;
        hipError_t e = dev_set(device);
        if (e != hipSuccess) return fail(e, "hipSetDevice");
        if ((e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "stream");
        if ((e = hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "stream");
        if ((e = hipStreamCreateWithFlags(&out_stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "stream");
        chunk = chunk_bytes_now();
        max_regions = (uint32_t)((chunk + REGION - 1) / REGION);
        slot_stride = REGION + 256;
        for (Buf &b : buf) {
            if (!(b.h_in = (uint8_t *)host_alloc(chunk + 64))) return set_error(NH_EOOM, "gzip encoder: no memory for the staging buffers");
            if ((e = dev_malloc((void **)&b.d_in, chunk + 256)) != hipSuccess) return fail(e, "device input");
            if ((e = hipMemset(b.d_in, 0, chunk + 256)) != hipSuccess) return fail(e, "memset");
            if ((e = dev_malloc((void **)&b.d_slots, (size_t)max_regions * slot_stride)) != hipSuccess) return fail(e, "slots");
            if ((e = dev_malloc((void **)&b.d_sizes, max_regions * sizeof(uint32_t))) != hipSuccess) return fail(e, "sizes");
            if ((e = dev_malloc((void **)&b.d_crcs, max_regions * sizeof(uint32_t))) != hipSuccess) return fail(e, "crcs");
            if (!(b.h_crcs = (uint32_t *)host_alloc(max_regions * sizeof(uint32_t)))) return set_error(NH_EOOM, "gzip encoder: no memory for the staging buffers");
            if ((e = dev_malloc((void **)&b.d_offsets, (max_regions + 1) * sizeof(uint64_t))) != hipSuccess) return fail(e, "offsets");
            if ((e = dev_malloc((void **)&b.d_out, (size_t)max_regions * slot_stride)) != hipSuccess) return fail(e, "packed output");
            if (!(b.h_out = (uint8_t *)host_alloc((size_t)max_regions * slot_stride)) || !(b.h_total = (uint64_t *)host_alloc(64)))
                return set_error(NH_EOOM, "gzip encoder: no memory for the staging buffers");
            if ((e = hipEventCreate(&b.done)) != hipSuccess) return fail(e, "event");
            if ((e = hipEventCreate(&b.k0)) != hipSuccess) return fail(e, "event");
            if ((e = hipEventCreate(&b.k1)) != hipSuccess) return fail(e, "event");
            if ((e = hipEventCreateWithFlags(&b.filled, hipEventDisableTiming)) != hipSuccess) return fail(e, "event");
        }
        if ((e = dev_malloc((void **)&d_tok, (size_t)max_regions * dfl::TOK_CAP * sizeof(uint32_t))) != hipSuccess) return fail(e, "tokens");
        // d_prior: two rows that alternate from chunk to chunk, then the two candidates for the first prices of a stream
        if ((e = dev_malloc((void **)&d_prior, 4 * dfl::PRIOR_BYTES)) != hipSuccess) return fail(e, "prior");
        // The price feedback from block to block has two stable states on FASTQ text: bases matched wherever four of
        // them repeat and their literals dear (where zlib's rules lead), or bases literal and cheap with only long
        // repeats matched.  Which one is smaller depends on the qualities (few distinct values: the second, by 7 %;
        // forty values or a long-read spread: the first, by 1-5 %), and neither is left once entered.  So a stream
        // starts from both -- literals 6 bits, lengths 7, distances 5, with and without 2 bits for A C G T N -- on
        // its first regions and goes on with the one that came out smaller (submit(), chunk 0).
        uint8_t prior[2][dfl::PRIOR_BYTES];
        for (int c = 0; c < 2; c++) {
            for (int s = 0; s < dfl::NLIT; s++) prior[c][s] = s < 256 ? 6 : 7;
            for (int s = 0; s < dfl::NDIST; s++) prior[c][dfl::NLIT + s] = 5;
        }
        for (const char *c = "ACGTN"; *c; c++) prior[1][(int)*c] = 2;
        if ((e = hipMemcpy(d_prior + 2 * dfl::PRIOR_BYTES, prior, sizeof prior, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "prior copy");
        if ((e = hipMemcpy(d_prior, prior[1], dfl::PRIOR_BYTES, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "prior copy");
        if (getenv("NOHUMAN_GZIP_PROF") && !NH_DFL_PROF)
            fprintf(stderr, "[gzip prof] this build has no phase timers: make -C nohuman_amd/csrc ab-dflprof, NOHUMAN_ENGINE_LIB=tools/ab_engine_dflprof.so\n");
        if (getenv("NOHUMAN_GZIP_PROF") && NH_DFL_PROF) {
            if ((e = dev_malloc((void **)&d_prof, 128)) != hipSuccess) return fail(e, "prof");
            (void)hipMemset(d_prof, 0, 128);
        }
        return NH_OK;
    }
    void destroy() {
        if (device < 0) return;
        (void)dev_set(device);
        if (copy_stream) (void)hipStreamSynchronize(copy_stream);
        if (stream) (void)hipStreamSynchronize(stream);
        for (Buf &b : buf) {
            if (b.filled) (void)hipEventDestroy(b.filled);
            host_free(b.h_in);
            if (b.d_in) (void)hipFree(b.d_in);
            if (b.d_slots) (void)hipFree(b.d_slots);
            if (b.d_sizes) (void)hipFree(b.d_sizes);
            if (b.d_crcs) (void)hipFree(b.d_crcs);
            host_free(b.h_crcs);
            if (b.d_offsets) (void)hipFree(b.d_offsets);
            if (b.d_out) (void)hipFree(b.d_out);
            host_free(b.h_out);
            host_free(b.h_total);
            if (b.done) (void)hipEventDestroy(b.done);
            if (b.k0) (void)hipEventDestroy(b.k0);
            if (b.k1) (void)hipEventDestroy(b.k1);
        }
        if (d_tok) (void)hipFree(d_tok);
        if (d_prior) (void)hipFree(d_prior);
        if (d_prof) {
            unsigned long long h[16] = {};
            (void)hipMemcpy(h, d_prof, 128, hipMemcpyDeviceToHost);
            if (h[4])
                fprintf(stderr, "[gzip prof] %llu regions: per region (cycles) match %.0f  insert+parse+tokens %.0f  blocks %.0f (codes %.0f: literal %.0f distance %.0f "
                                "header rows + run lengths %.0f code-length code %.0f)  all %.0f\n",
                        h[4], (double)h[0] / h[4], (double)h[1] / h[4], (double)h[2] / h[4], (double)h[5] / h[4], (double)h[6] / h[4], (double)h[7] / h[4],
                        (double)h[8] / h[4], (double)h[9] / h[4], (double)h[3] / h[4]);
            (void)hipFree(d_prof);
        }
        if (stream) (void)hipStreamDestroy(stream);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        if (out_stream) (void)hipStreamDestroy(out_stream);
        device = -1;
    }
    // the staged bytes of buf[i] that are not on the device yet go there (one copy)
    int upload_staged(int i) {
        Buf &b = buf[i];
        if (b.staged_from == b.fill) return NH_OK;
        const hipError_t e = hipMemcpyAsync(b.d_in + b.staged_from, b.h_in + b.staged_from, b.fill - b.staged_from,
                                            hipMemcpyHostToDevice, copy_stream);
        if (e != hipSuccess) return fail(e, "H2D");
        b.staged_from = b.fill;
        return NH_OK;
    }
    // `take` bytes that already lie in this GPU's memory are appended to buf[i]
    int append_device(int i, const void *dev, size_t take) {
        Buf &b = buf[i];
        int rc = upload_staged(i);
        if (rc != NH_OK) return rc;
        const hipError_t e = hipMemcpyAsync(b.d_in + b.fill, dev, take, hipMemcpyDeviceToDevice, copy_stream);
        if (e != hipSuccess) return fail(e, "D2D");
        b.fill += take;
        b.staged_from = b.fill;
        return NH_OK;
    }
    int settle() {
        const hipError_t e = hipStreamSynchronize(copy_stream);
        return e == hipSuccess ? NH_OK : fail(e, "copy");
    }
    // the first regions of a stream under both candidate price sets; the smaller result's prices start the stream
    int pilot(Buf &b) {
        static const char *force = getenv("NOHUMAN_GZIP_PRICES");  // "0" / "1": no pilot, that candidate (tuning)
        int pick = 1;
        if (force && (force[0] == '0' || force[0] == '1')) {
            pick = force[0] - '0';
        } else {
            const uint32_t nr = b.n_regions < PILOT_REGIONS ? b.n_regions : PILOT_REGIONS;
            const uint64_t n = b.fill < (uint64_t)nr * REGION ? b.fill : (uint64_t)nr * REGION;
            for (int c = 0; c < 2; c++) {
                DeflateArgs a{};
                a.in = b.d_in;
                a.n = n;
                a.region = REGION;
                a.n_regions = nr;
                a.slots = b.d_slots;
                a.slot_stride = slot_stride;
                a.sizes = b.d_sizes + c * PILOT_REGIONS;
                a.crcs = b.d_crcs;
                a.tok = d_tok;
                a.prior = d_prior + (2 + c) * dfl::PRIOR_BYTES;
                a.prior_out = nullptr;
                a.prof = nullptr;
                if (gzip_ways() == 4)
                    hipLaunchKernelGGL(dfl::k_deflate<4>, dim3(nr), dim3(64), 0, stream, a);
                else if (gzip_ways() == 6)
                    hipLaunchKernelGGL(dfl::k_deflate<6>, dim3(nr), dim3(64), 0, stream, a);
                else
                    hipLaunchKernelGGL(dfl::k_deflate<8>, dim3(nr), dim3(64), 0, stream, a);
            }
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(e, "launch");
            uint32_t sz[2 * PILOT_REGIONS];
            if ((e = hipMemcpyAsync(sz, b.d_sizes, sizeof sz, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail(e, "D2H");
            if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e, "pilot");
            uint64_t tot[2] = {0, 0};
            for (int c = 0; c < 2; c++)
                for (uint32_t r = 0; r < nr; r++) tot[c] += sz[c * PILOT_REGIONS + r];
            pick = tot[1] <= tot[0] ? 1 : 0;
            if (getenv("NOHUMAN_TRACE"))
                fprintf(stderr, "[nohuman trace] gzip encoder: first %u regions under the two starting price sets: %llu / %llu bytes -> %s\n",
                        nr, (unsigned long long)tot[0], (unsigned long long)tot[1], pick ? "bases literal" : "zlib-like");
        }
        const hipError_t e = hipMemcpyAsync(d_prior, d_prior + (2 + pick) * dfl::PRIOR_BYTES, dfl::PRIOR_BYTES, hipMemcpyDeviceToDevice, stream);
        return e == hipSuccess ? NH_OK : fail(e, "prior copy");
    }
    // queues the compression of buf[i]
    int submit(int i) {
        Buf &b = buf[i];
        hipError_t e = dev_set(device);
        if (e != hipSuccess) return fail(e, "hipSetDevice");
        b.n_regions = (uint32_t)((b.fill + REGION - 1) / REGION);
        const int urc = upload_staged(i);
        if (urc != NH_OK) return urc;
        if ((e = hipEventRecord(b.filled, copy_stream)) != hipSuccess) return fail(e, "event");
        dev_check(device, "gzip encoder, a chunk's kernels");
        dev_check_ptr(b.d_in, device, "gzip encoder (input chunk)");
        if ((e = hipStreamWaitEvent(stream, b.filled, 0)) != hipSuccess) return fail(e, "wait");
        if (chunks == 0) {
            const int prc = pilot(b);
            if (prc != NH_OK) return prc;
        }
        DeflateArgs a{};
        a.in = b.d_in;
        a.n = b.fill;
        a.region = REGION;
        a.n_regions = b.n_regions;
        a.slots = b.d_slots;
        a.slot_stride = slot_stride;
        a.sizes = b.d_sizes;
        a.crcs = b.d_crcs;
        a.tok = d_tok;
        a.prior = d_prior + (chunks & 1) * dfl::PRIOR_BYTES;
        a.prior_out = d_prior + ((chunks + 1) & 1) * dfl::PRIOR_BYTES;
        a.prof = d_prof;
        (void)hipEventRecord(b.k0, stream);
        if (gzip_ways() == 4)
            hipLaunchKernelGGL(dfl::k_deflate<4>, dim3(b.n_regions), dim3(64), 0, stream, a);
        else if (gzip_ways() == 6)
            hipLaunchKernelGGL(dfl::k_deflate<6>, dim3(b.n_regions), dim3(64), 0, stream, a);
        else
            hipLaunchKernelGGL(dfl::k_deflate<8>, dim3(b.n_regions), dim3(64), 0, stream, a);
        hipLaunchKernelGGL(dfl::k_deflate_offsets, dim3(1), dim3(64), 0, stream, b.d_sizes, b.n_regions, b.d_offsets);
        hipLaunchKernelGGL(dfl::k_deflate_pack, dim3(b.n_regions), dim3(256), 0, stream, b.d_slots, slot_stride, b.d_sizes,
                           b.d_offsets, b.d_out);
        (void)hipEventRecord(b.k1, stream);
        if ((e = hipGetLastError()) != hipSuccess) return fail(e, "launch");
        if ((e = hipMemcpyAsync(b.h_total, b.d_offsets + b.n_regions, sizeof(uint64_t), hipMemcpyDeviceToHost, stream)) != hipSuccess)
            return fail(e, "D2H");
        if ((e = hipMemcpyAsync(b.h_crcs, b.d_crcs, b.n_regions * sizeof(uint32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess)
            return fail(e, "D2H");
        if ((e = hipEventRecord(b.done, stream)) != hipSuccess) return fail(e, "event");
        b.submitted = b.fill;
        b.in_flight = true;
        chunks++;
        return NH_OK;
    }
    // waits for buf[i] and brings its stream to h_out; *len = its bytes
    int collect(int i, size_t *len, uint32_t *crc) {
        Buf &b = buf[i];
        hipError_t e = dev_set(device);
        if (e != hipSuccess) return fail(e, "hipSetDevice");
        if ((e = hipEventSynchronize(b.done)) != hipSuccess) return fail(e, "kernel");
        float ms = 0;
        if (hipEventElapsedTime(&ms, b.k0, b.k1) == hipSuccess) kernel_ms += ms;
        const uint64_t total = *b.h_total;
        if (total > (uint64_t)max_regions * slot_stride) return set_error(NH_EDEVICE, "gzip encoder: impossible stream size");
        if ((e = hipMemcpyAsync(b.h_out, b.d_out, total, hipMemcpyDeviceToHost, out_stream)) != hipSuccess) return fail(e, "D2H");
        if ((e = hipStreamSynchronize(out_stream)) != hipSuccess) return fail(e, "D2H");
        // NOHUMAN_GZIP_VERIFY=1: the chunk's stream is inflated again on the host (zlib, raw deflate) and its length and
        // CRC-32 are compared with the text's -- a paranoid mode, one core at ~0.5 GB/s of text
        static const bool verify = getenv("NOHUMAN_GZIP_VERIFY") != nullptr;
        uint32_t crc_before = *crc;
        {
            static const uint32_t full = dfl::gf2_xpow8(REGION);
            uint32_t c = *crc;
            size_t left = b.submitted;
            for (uint32_t r = 0; r < b.n_regions; r++) {
                const size_t rl = left < REGION ? left : REGION;
                c = dfl::gf2_mul(c, rl == REGION ? full : dfl::gf2_xpow8(rl)) ^ b.h_crcs[r];
                left -= rl;
            }
            *crc = c;
        }
        if (verify) {
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) return set_error(NH_EIO, "gzip encoder: verify: inflateInit2 failed");
            std::vector<unsigned char> back(1u << 20);
            uint32_t vcrc = 0;
            uint64_t got = 0;
            zs.next_in = b.h_out;
            zs.avail_in = (uInt)total;
            int zrc = Z_OK;
            while (zrc == Z_OK || zrc == Z_BUF_ERROR) {
                zs.next_out = back.data();
                zs.avail_out = (uInt)back.size();
                zrc = inflate(&zs, Z_NO_FLUSH);
                const size_t have = back.size() - zs.avail_out;
                vcrc = crc32_fast(vcrc, back.data(), have);
                got += have;
                if (have == 0 && zs.avail_in == 0) break;
            }
            inflateEnd(&zs);
            const uint32_t want = crc32_join(crc_before, vcrc, got);
            if ((zrc != Z_OK && zrc != Z_BUF_ERROR) || zs.avail_in != 0 || got != b.submitted || want != *crc)
                return set_error(NH_EDEVICE, "gzip encoder: verification failed (zlib %d, %llu of %llu bytes, CRC %08x / %08x)",
                                 zrc, (unsigned long long)got, (unsigned long long)b.submitted, want, *crc);
        }
        b.in_flight = false;
        b.fill = 0;
        b.staged_from = 0;
        *len = (size_t)total;
        return NH_OK;
    }
};

bool write_fd(int fd, const void *p, size_t n) {
    const char *c = (const char *)p;
    while (n) {
        ssize_t w = ::write(fd, c, n);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        c += w;
        n -= (size_t)w;
    }
    return true;
}

class GpuGzipEncoder : public StreamEncoder {
public:
    GpuGzipEncoder(int fd, const char *name) : fd_(fd), name_(name) {}
    ~GpuGzipEncoder() override {
        stop_writer();
        dev_.destroy();
    }
    int init(int device) {
        const int rc = dev_.init(device);
        if (rc != NH_OK) return rc;
        static const unsigned char header[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};  // no name, no mtime, unix
        if (!write_fd(fd_, header, sizeof header)) return rc_ = set_error(NH_EIO, "write error on %s", name_.c_str());
        return NH_OK;
    }
    void map_device(const void *host, size_t len, const void *dev, int device, bool host_valid) override {
        map_host_ = (const uint8_t *)host;
        dev_check_ptr(dev, device, "gzip encoder, a batch's text in HBM");
        map_len_ = device == dev_.device ? len : 0;
        map_dev_ = (const uint8_t *)dev;
        map_host_valid_ = host_valid;
        if (!host_valid && device != dev_.device && rc_ == NH_OK)
            rc_ = set_error(NH_EDEVICE, "gzip encoder on GPU %d was handed text that lies only on GPU %d", dev_.device, device);
    }
    bool takes_device_spans() const override { return true; }
    int settle() override {
        const uint64_t t0 = now_ns();
        if (rc_ == NH_OK && dev_set(dev_.device) == hipSuccess) rc_ = dev_.settle();
        map_len_ = 0;
        t_settle_ += now_ns() - t0;
        return rc_;
    }
    int write(const void *p, size_t n) override {
        const uint64_t tw0 = now_ns();
        struct Acc {
            uint64_t &t, t0;
            ~Acc() { t += now_ns() - t0; }
        } acc{t_write_, tw0};
        const uint8_t *c = (const uint8_t *)p;
        if (n && dev_set(dev_.device) != hipSuccess) rc_ = set_error(NH_EDEVICE, "gzip encoder: hipSetDevice failed");
        while (n && rc_ == NH_OK) {
            DeflateDev::Buf &b = dev_.buf[cur_];
            const size_t room = dev_.chunk - b.fill;
            const size_t take = n < room ? n : room;
            // a long span of text that is on this GPU already (the batch the classifier worked on) is copied
            // there; everything else is staged in page-locked memory and uploaded in one piece
            const bool mapped = c >= map_host_ && c + take <= map_host_ + map_len_;
            if (mapped && (take >= DEVICE_SPAN_MIN || !map_host_valid_)) {
                rc_ = dev_.append_device(cur_, map_dev_ + (c - map_host_), take);
            } else {
                memcpy(b.h_in + b.fill, c, take);
                b.fill += take;
            }
            total_ += take;
            c += take;
            n -= take;
            if (rc_ == NH_OK && b.fill == dev_.chunk) rotate();
        }
        return rc_;
    }
    int finish() override {
        if (rc_ == NH_OK && dev_.buf[cur_].fill) rotate();
        for (int k = 0; k < DeflateDev::NBUF && rc_ == NH_OK; k++) retire((cur_ + k) % DeflateDev::NBUF);
        if (rc_ != NH_OK) return rc_;
        wait_writer();
        if (rc_ != NH_OK) return rc_;
        unsigned char tail[10] = {0x03, 0x00};  // final block: fixed codes, end-of-block only
        for (int i = 0; i < 4; i++) {
            tail[2 + i] = (unsigned char)(crc_ >> (8 * i));
            tail[6 + i] = (unsigned char)((uint32_t)total_ >> (8 * i));
        }
        if (!write_fd(fd_, tail, sizeof tail)) rc_ = set_error(NH_EIO, "write error on %s", name_.c_str());
        if (getenv("NOHUMAN_TRACE"))
            fprintf(stderr, "[nohuman trace] gzip encoder %s: write() %.3f s (of it: waiting for chunks + D2H %.3f, file %.3f), settle %.3f s, "
                            "kernels %.3f s, %.2f GB in, %.2f GB out\n",
                    name_.c_str(), t_write_ / 1e9, t_collect_ / 1e9, t_file_ / 1e9, t_settle_ / 1e9, dev_.kernel_ms / 1e3, total_ / 1e9,
                    out_bytes_ / 1e9);
        return rc_;
    }
    double kernel_ms() const { return dev_.kernel_ms; }
    uint64_t bytes_out() const { return out_bytes_; }

private:
    // the current chunk goes to the GPU; the next one is filled while it is compressed -- after the chunk that
    // used that buffer before has been written
    void rotate() {
        if ((rc_ = dev_.submit(cur_)) != NH_OK) return;
        cur_ = (cur_ + 1) % DeflateDev::NBUF;
        retire(cur_);
    }
    // A finished chunk's bytes go to the file on a thread of their own, one chunk behind: the caller (nh_run's flusher) waits
    // for the GPU on the next chunk meanwhile instead of for the file (4.4 GB a mate file: 0.7 s of a 2.8 s run).  The chunk's
    // page-locked output buffer is taken again only NBUF rotations later, long after its bytes are written.
    void retire(int i) {
        if (!dev_.buf[i].in_flight || rc_ != NH_OK) return;
        size_t len = 0;
        const uint64_t t0 = now_ns();
        if ((rc_ = dev_.collect(i, &len, &crc_)) != NH_OK) return;
        const uint64_t t1 = now_ns();
        wait_writer();
        if (rc_ == NH_OK && len) post_write(dev_.buf[i].h_out, len);
        t_collect_ += t1 - t0;
        t_file_ += now_ns() - t1;
        out_bytes_ += len;
    }
    void post_write(const void *p, size_t n) {
        if (!wt_.joinable()) wt_ = std::thread([this] { writer_main(); });
        std::lock_guard<std::mutex> lk(wmu_);
        wjob_ = p;
        wlen_ = n;
        wbusy_ = true;
        wcv_.notify_all();
    }
    void wait_writer() {
        std::unique_lock<std::mutex> lk(wmu_);
        wcv_.wait(lk, [&] { return !wbusy_; });
        if (werr_ && rc_ == NH_OK) rc_ = set_error(NH_EIO, "write error on %s", name_.c_str());
    }
    void stop_writer() {
        {
            std::lock_guard<std::mutex> lk(wmu_);
            wquit_ = true;
            wcv_.notify_all();
        }
        if (wt_.joinable()) wt_.join();
    }
    void writer_main() {
        for (;;) {
            const void *p;
            size_t n;
            {
                std::unique_lock<std::mutex> lk(wmu_);
                wcv_.wait(lk, [&] { return wbusy_ || wquit_; });
                if (!wbusy_) return;
                p = wjob_;
                n = wlen_;
            }
            const bool ok = write_fd(fd_, p, n);
            std::lock_guard<std::mutex> lk(wmu_);
            if (!ok) werr_ = true;
            wbusy_ = false;
            wcv_.notify_all();
        }
    }
    std::thread wt_;
    std::mutex wmu_;
    std::condition_variable wcv_;
    const void *wjob_ = nullptr;
    size_t wlen_ = 0;
    bool wbusy_ = false, wquit_ = false, werr_ = false;
    static uint64_t now_ns() {
        return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    uint64_t t_settle_ = 0, t_collect_ = 0, t_file_ = 0, t_write_ = 0;
    static constexpr size_t DEVICE_SPAN_MIN = 32u << 10;
    int fd_;
    std::string name_;
    DeflateDev dev_;
    const uint8_t *map_host_ = nullptr, *map_dev_ = nullptr;
    size_t map_len_ = 0;
    bool map_host_valid_ = true;
    int cur_ = 0;
    uint32_t crc_ = 0;
    uint64_t total_ = 0, out_bytes_ = 0;
    int rc_ = NH_OK;
};

}  // namespace

StreamEncoder *make_gpu_gzip_encoder(int fd, int device, const char *name) {
    GpuGzipEncoder *e = new GpuGzipEncoder(fd, name);
    if (e->init(device) != NH_OK) {
        delete e;
        return nullptr;
    }
    return e;
}

}  // namespace nh

// One gzip member of a host buffer through the GPU encoder, to a file: what tests and bench.py drive.
// stats: [0] bytes written, [1] kernel microseconds (HIP events around the three kernels of every chunk).
extern "C" int nh_gzip_gpu_file(int32_t device, const void *in, uint64_t n, const char *out_path, uint64_t *stats) {
    if ((!in && n) || !out_path) return nh::set_error(NH_EINVAL, "nh_gzip_gpu_file: null argument");
    int fd = ::open(out_path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fd < 0) return nh::set_error(NH_EIO, "cannot create %s", out_path);
    int rc = NH_OK;
    {
        nh::GpuGzipEncoder enc(fd, out_path);
        rc = enc.init(device);
        if (rc == NH_OK) rc = enc.write(in, (size_t)n);
        if (rc == NH_OK) rc = enc.finish();
        if (stats) {
            stats[0] = enc.bytes_out() + 20;
            stats[1] = (uint64_t)(enc.kernel_ms() * 1000.0);
        }
    }
    if (::close(fd) != 0 && rc == NH_OK) rc = nh::set_error(NH_EIO, "write error on %s", out_path);
    return rc;
}
