// nh_kernels.hip -- gfx950 (CDNA4) kernels of the read-classification hot path.
//
// One 64-lane wavefront classifies one fragment (read or read pair) end to end; it replaces the
// body of kraken2's ClassifySequence loop (minimizer scan -> compact-hash probe -> ResolveTree;
// SURVEY.md section 8a rows a5-a9, Appendix A.2-A.5) that nohuman reaches through
// /root/reference/src/lib.rs:22-48.  Integer / byte work bound by random HBM line fetches:
// no MFMA by design.
//
// Data flow of one tile (128 l-mers = 2 per lane, i.e. 128-(k-l) k-mers):
//   global bases (one coalesced dword per lane, prefetched one tile ahead)
//   -> 2-bit packed stream in LDS (1 byte per lane)
//   -> per lane two l-mers by one 64-bit funnel read; ONE 32-base reverse complement serves both
//   -> canonical/spaced/toggled candidates in LDS
//   -> per lane two k-mer minimizers = min over a (k-l+1)-wide candidate window
//   -> run starts (minimizer != previous non-ambiguous minimizer) compacted into an LDS queue
//   -> one lane per queued minimizer: fmix64, exact hc % capacity, 16-byte-chunk linear probe
//   -> taxa back through LDS -> per-taxon hit counts, hit groups -> ResolveTree on the wave.
// The kernel is instruction-issue sensitive (see profiles/): tile-local arithmetic is 32-bit,
// control flow is wave-uniform wherever possible, and kraken2's default k=35/l=31 geometry is a
// compile-time specialisation (STD) next to the fully general variant.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "nh_device.h"

namespace nh {

#define NH_FULL 0xFFFFFFFFFFFFFFFFull
#ifndef NH_PROBE_CHUNKS
#define NH_PROBE_CHUNKS 2
#endif

constexpr int CAND_PAD = 66;  // window reads of idle lanes stay inside the array (k-l <= 64)

struct WaveLds {
    uint32_t pk[24];  // 2-bit packed bases: base i' of the tile frame at bit 2*(255-i'); 64 B + zero pad
    uint32_t pa[24];  // same layout, value 1 where the base is ambiguous
    uint64_t cand[TL + CAND_PAD];
    uint64_t runmin[TL];
    uint32_t runtax[TL];
    uint32_t list_tax[LIST_CAP];
    uint32_t list_cnt[LIST_CAP];
};

__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wave are issued and serviced in order; only the compiler must be
    // kept from moving LDS accesses across the hand-off points.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

// exact hc % capacity: q' = floor(hc * floor((2^64-1)/cap) / 2^64) is q, q-1 or q-2
__device__ __forceinline__ uint64_t mod_capacity(uint64_t hc, uint64_t cap, uint64_t magic) {
    uint64_t q = __umul64hi(hc, magic);
    uint64_t r = hc - q * cap;
    if (r >= cap) r -= cap;
    if (r >= cap) r -= cap;
    return r;
}

// reverse the 32 two-bit groups of x and complement every base
__device__ __forceinline__ uint64_t revcomp_word(uint64_t x) {
    const uint64_t br = __builtin_bitreverse64(x);
    return ~(((br & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((br & 0x5555555555555555ull) << 1));
}

__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int src) {
    uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src);
    uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// CompactHashTable::Get (A.4).  Linear probing walks aligned 16-byte chunks (4 cells per load)
// and decides each chunk with selects only; the loop branch is the single divergent one.
template <bool LINEAR>
__device__ __forceinline__ uint32_t table_get(const DevDB &db, uint64_t hc, bool active) {
    const uint32_t vbits = db.value_bits;
    const uint32_t vmask = db.vmask;
    const uint32_t compacted = (uint32_t)(hc >> (32 + vbits));
    const uint64_t cap = db.capacity;
    uint64_t pos = mod_capacity(hc, cap, db.cap_magic);
    uint32_t result = 0;
    if (LINEAR) {
        // Each round fetches up to PROBE_CHUNKS 16-byte chunks, never past the end of the 128-byte
        // line that holds `pos`: the line is the unit HBM delivers (profiles/: 1.16 fabric requests
        // per lookup), L1/L2 are too small to keep it until a later round, and the mean number of
        // dependent rounds a wave waits for drops from ~10 to ~3.3 (39 lookups, load factor 0.7).
        constexpr int PROBE_CHUNKS = NH_PROBE_CHUNKS;
        const uint32_t ckey = compacted << vbits;
        uint32_t chunks_left = db.max_chunks;
        bool done = !active;
        while (!done) {
            const uint64_t base = pos & ~3ull;
            const uint32_t first = (uint32_t)pos & 3u;
            const uint64_t room = cap - base;                              // existing cells from base
            const uint32_t in_line = (32u - ((uint32_t)base & 31u)) >> 2;  // chunks to the line end
            uint32_t nch = in_line < (uint32_t)PROBE_CHUNKS ? in_line : (uint32_t)PROBE_CHUNKS;
            const uint32_t room_chunks = room >= 4 * PROBE_CHUNKS ? PROBE_CHUNKS : (uint32_t)((room + 3) >> 2);
            nch = nch < room_chunks ? nch : room_chunks;
            const uint32_t nvalid = room < 4 * nch ? (uint32_t)room : 4 * nch;
            uint4 c[PROBE_CHUNKS];
            const uint4 *src = reinterpret_cast<const uint4 *>(db.table + base);
#pragma unroll
            for (int q = 0; q < PROBE_CHUNKS; q++)  // idle slots re-read the last useful chunk
                c[q] = src[(uint32_t)q < nch ? (uint32_t)q : nch - 1];
            bool found = false;
            uint32_t res = 0;
#pragma unroll
            for (int j = 4 * PROBE_CHUNKS - 1; j >= 0; j--) {  // lowest eligible stopping cell wins
                const uint4 &cq = c[j >> 2];
                const uint32_t cell = (j & 3) == 0 ? cq.x : (j & 3) == 1 ? cq.y : (j & 3) == 2 ? cq.z : cq.w;
                const uint32_t x = cell ^ ckey;  // key bits vanish on a match
                const bool elig = ((uint32_t)j >= first) & ((uint32_t)j < nvalid);
                const bool stop = elig & ((x <= vmask) | ((cell & vmask) == 0));
                found = stop ? true : found;
                res = stop ? (x <= vmask ? x : 0u) : res;
            }
            result = res;
            const uint64_t nxt = base + 4 * nch;
            pos = nxt >= cap ? 0 : nxt;
            chunks_left = chunks_left > nch ? chunks_left - nch : 0;
            done = found | (chunks_left == 0);
            if (!found) result = 0;
        }
    } else {
        const uint64_t first_idx = pos;
        const uint64_t step = mod_capacity((hc >> 8) | 1, cap, db.cap_magic);
        bool done = !active;
        while (!done) {
            const uint32_t cell = db.table[pos];
            if ((cell & vmask) == 0) {
                done = true;
            } else if ((cell >> vbits) == compacted) {
                result = cell & vmask;
                done = true;
            } else {
                pos += step;
                if (pos >= cap) pos -= cap;
                if (pos == first_idx) done = true;
            }
        }
    }
    return result;
}

__device__ __forceinline__ bool is_a_ancestor_of_b(const uint32_t *parent, uint32_t a, uint32_t b) {
    if (!a || !b) return false;
    while (b > a) b = parent[b];
    return a == b;
}

__device__ __forceinline__ uint32_t lowest_common_ancestor(const uint32_t *parent, uint32_t a,
                                                           uint32_t b) {
    if (!a || !b) return a ? a : b;
    while (a != b) {
        if (a > b)
            a = parent[a];
        else
            b = parent[b];
    }
    return a;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

// 4 ASCII bytes -> one byte of packed 2-bit codes (first base in bits 7:6); `suspect` is set when
// any of the 4 bytes is not one of ACGTacgt (SWAR, no per-byte work on the common path)
__device__ __forceinline__ uint32_t encode4(uint32_t w, bool &suspect) {
    const uint32_t up = w & 0xDFDFDFDFu;          // fold case
    const uint32_t x = (up >> 1) & 0x03030303u;   // A0 C1 G3 T2
    const uint32_t code = x ^ ((x >> 1) & 0x01010101u);  // A0 C1 G2 T3
    const uint32_t tbit = (x >> 1) & ~x & 0x01010101u;   // 1 where the byte decodes as T
    const uint32_t recon = (0x41414141u | (x << 1)) ^ (tbit * 0x11u);  // canonical letter of code
    suspect = recon != up;
    return (code * 0x40100401u) >> 24;
}

// exact per-base flags "real base of this read and not ACGTacgt", same packing as the codes
__device__ __forceinline__ uint32_t ambig4(uint32_t w, uint32_t p0, uint32_t lo, uint32_t hi) {
    uint32_t bad = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const uint32_t ch = (w >> (8 * b)) & 0xDFu;
        const bool ok = (ch == 0x41u) | (ch == 0x43u) | (ch == 0x47u) | (ch == 0x54u);
        const uint32_t p = p0 + b;
        const bool real = (p >= lo) & (p < hi);
        bad |= ((real & !ok) ? 1u : 0u) << (6 - 2 * b);
    }
    return bad;
}

__device__ __forceinline__ uint64_t funnel_read(const uint32_t *pkd, uint32_t s) {
    const uint32_t d = s >> 5, r = s & 31u;
    const uint64_t lo = (uint64_t)pkd[d] | ((uint64_t)pkd[d + 1] << 32);
    const uint64_t hi = pkd[d + 2];
    return r ? ((lo >> r) | (hi << (64 - r))) : lo;
}

// per-fragment accumulation state of one wave (wave-uniform scalars)
struct FragState {
    uint32_t nlist;       // distinct taxa in the LDS list
    uint32_t hit_groups;  // minimizer_hit_groups
    uint64_t carry_min;   // kraken2 last_minimizer (NH_FULL = none)
    uint32_t carry_tax;   // kraken2 last_taxon
    bool overflow;
};

// One tile: l-mers [q0, q0+nlt) / k-mers [q0, q0+nqt) of a sequence whose tile frame starts `sh`
// bytes into the dword stream `w` (4 bases per lane).  kt = index in kmer_taxa of k-mer q0.
#define NH_STAMP(i)                                        \
    do {                                                   \
        if (PROF) {                                        \
            const uint64_t _t = __builtin_readcyclecounter(); \
            prof[i] += _t - tprev;                         \
            tprev = _t;                                    \
        }                                                  \
    } while (0)

template <bool LINEAR, bool STD, bool PROF>
__device__ __forceinline__ void process_tile(const DevDB &db, WaveLds &S, const int lane,
                                             const uint64_t lane_lt, const uint32_t w,
                                             const uint32_t sh, const uint32_t nlt,
                                             const uint32_t nqt, FragState &st,
                                             uint32_t *__restrict__ kmer_taxa, const uint64_t kt,
                                             uint32_t &acc_lookups, uint64_t (&prof)[8],
                                             uint64_t &tprev) {
    const uint32_t L = STD ? 31u : db.l;
    const uint32_t W = STD ? 4u : db.window;
    const uint64_t LMASK = STD ? ((1ull << 62) - 1) : db.lmer_mask;
    const int RV = STD ? 1 : db.revcom_version;
    const uint64_t MIN_HASH = STD ? 0ull : db.min_hash;

    // ---- 1. bases -> packed 2-bit stream -------------------------------------------------------
    bool suspect;
    const uint32_t codes = encode4(w, suspect);
    reinterpret_cast<uint8_t *>(S.pk)[63 - lane] = (uint8_t)codes;
    bool has_amb = __ballot(suspect) != 0;
    if (has_amb) {  // exact flags, restricted to the bases of this tile
        const uint32_t bad = ambig4(w, 4u * lane, sh, sh + nlt + L - 1);
        has_amb = __ballot(bad != 0) != 0;
        reinterpret_cast<uint8_t *>(S.pa)[63 - lane] = (uint8_t)bad;
    }
    wave_sync();
    NH_STAMP(1);

    // ---- 2. two l-mers per lane -> candidates --------------------------------------------------
    {
        const uint32_t j1 = sh + 2u * lane + L;  // frame index of the last base of l-mer 2t+1
        const uint32_t s = 2u * (255u - j1);
        const uint64_t wv = funnel_read(S.pk, s);
        const uint64_t lm1 = wv & LMASK;
        const uint64_t lm0 = (wv >> 2) & LMASK;
        uint64_t rc0, rc1;
        if (RV != 0) {
            const uint64_t R = revcomp_word(wv);  // one reverse complement serves both l-mers
            rc1 = R >> (64 - 2 * L);
            rc0 = (R >> (62 - 2 * L)) & LMASK;
        } else {  // legacy databases: complement of the un-shifted reversed word
            rc1 = revcomp_word(lm1) & LMASK;
            rc0 = revcomp_word(lm0) & LMASK;
        }
        const uint64_t c0 = (umin64(lm0, rc0) & db.spaced_mask) ^ db.toggle;
        const uint64_t c1 = (umin64(lm1, rc1) & db.spaced_mask) ^ db.toggle;
        bool dead0 = 2u * lane >= nlt, dead1 = 2u * lane + 1 >= nlt;
        if (has_amb) {
            const uint64_t wa = funnel_read(S.pa, s);
            dead1 |= (wa & LMASK) != 0;
            dead0 |= ((wa >> 2) & LMASK) != 0;
        }
        ulonglong2 cc;
        cc.x = dead0 ? NH_FULL : c0;
        cc.y = dead1 ? NH_FULL : c1;
        *reinterpret_cast<ulonglong2 *>(&S.cand[2 * lane]) = cc;
    }
    wave_sync();

    // ---- 3. two k-mer minimizers per lane (window min) -----------------------------------------
    const uint32_t qi0 = 2u * lane, qi1 = 2u * lane + 1;
    uint64_t mz0, mz1;
    bool v0, v1;  // valid and non-ambiguous
    {
        uint64_t first, mid, last0, last1;
        if (STD) {
            const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&S.cand[qi0]);
            const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(&S.cand[qi0 + 2]);
            const ulonglong2 c = *reinterpret_cast<const ulonglong2 *>(&S.cand[qi0 + 4]);
            first = a.x;
            mid = umin64(umin64(a.y, b.x), umin64(b.y, c.x));
            last0 = c.x;
            last1 = c.y;
        } else if (W == 0) {
            first = S.cand[qi0];
            last1 = S.cand[qi1];
            last0 = first;
            mid = NH_FULL;
        } else {
            first = S.cand[qi0];
            mid = S.cand[qi0 + 1];
            for (uint32_t i = 2; i <= W; i++) mid = umin64(mid, S.cand[qi0 + i]);
            last0 = S.cand[qi0 + W];
            last1 = S.cand[qi1 + W];
        }
        const uint64_t m0 = umin64(first, mid);
        const uint64_t m1 = (!STD && W == 0) ? last1 : umin64(mid, last1);
        v0 = (qi0 < nqt) & (last0 != NH_FULL);
        v1 = (qi1 < nqt) & (last1 != NH_FULL);
        mz0 = m0 ^ db.toggle;
        mz1 = m1 ^ db.toggle;
    }

    NH_STAMP(2);
    // ---- 4. run starts: minimizer differs from the previous non-ambiguous one -----------------
    uint64_t prev_in;
    if (!has_amb) {
        prev_in = __shfl_up(mz1, 1, 64);
        if (lane == 0) prev_in = st.carry_min;
    } else {
        // inclusive scan of "rightmost lane that holds a non-ambiguous k-mer"
        bool has = v0 | v1;
        uint64_t val = v1 ? mz1 : mz0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const bool h2 = __shfl_up((int)has, d, 64) != 0;
            const uint64_t x2 = __shfl_up(val, d, 64);
            if (lane >= d && !has) {
                has = h2;
                val = x2;
            }
        }
        const bool hx = __shfl_up((int)has, 1, 64) != 0;
        const uint64_t vx = __shfl_up(val, 1, 64);
        prev_in = (lane > 0 && hx) ? vx : st.carry_min;
    }
    const uint64_t prev1 = v0 ? mz0 : prev_in;
    const bool new0 = v0 & (mz0 != prev_in);
    const bool new1 = v1 & (mz1 != prev1);

    // ---- 5. compact run starts into the LDS queue ----------------------------------------------
    const uint64_t b0 = __ballot(new0), b1 = __ballot(new1);
    const uint32_t ex = __popcll(b0 & lane_lt) + __popcll(b1 & lane_lt);
    const uint32_t nruns = __popcll(b0) + __popcll(b1);
    if (new0) S.runmin[ex] = mz0;
    if (new1) S.runmin[ex + (new0 ? 1u : 0u)] = mz1;
    const int ri0 = (int)(ex + (new0 ? 1u : 0u)) - 1;
    const int ri1 = ri0 + (new1 ? 1 : 0);
    wave_sync();
    NH_STAMP(3);

    // ---- 6. one lane per queued minimizer: hash + probe ----------------------------------------
    uint64_t hit_mask_any = 0;
    for (uint32_t r0 = 0; r0 < nruns; r0 += 64) {
        const uint32_t r = r0 + lane;
        const bool act = r < nruns;
        const uint64_t hc = fmix64(S.runmin[r & (TL - 1)]);
        const bool look = act & !(MIN_HASH != 0 && hc < MIN_HASH);
        const uint32_t taxon = table_get<LINEAR>(db, hc, look);
        if (act) S.runtax[r] = taxon;
        const uint64_t hm = __ballot(taxon != 0);
        hit_mask_any |= hm;
        st.hit_groups += __popcll(hm);
        acc_lookups += __popcll(__ballot(look));
    }
    wave_sync();
    NH_STAMP(4);

    // ---- 7. per-k-mer taxa, hit counts, carry --------------------------------------------------
    uint32_t t0 = 0, t1 = 0;
    const bool any_hit = (hit_mask_any != 0) | (st.carry_tax != 0);
    if (any_hit || kmer_taxa) {
        if (v0) t0 = ri0 >= 0 ? S.runtax[ri0] : st.carry_tax;
        if (v1) t1 = ri1 >= 0 ? S.runtax[ri1] : st.carry_tax;
    }
    if (kmer_taxa) {
        if (qi0 < nqt) kmer_taxa[kt + qi0] = v0 ? t0 : TAXON_AMBIGUOUS;
        if (qi1 < nqt) kmer_taxa[kt + qi1] = v1 ? t1 : TAXON_AMBIGUOUS;
    }
    {
        const uint64_t m1 = __ballot(v1), m0 = __ballot(v0);
        if (m0 | m1) {
            const int l1 = m1 ? 63 - __builtin_clzll(m1) : -1;
            const int l0 = m0 ? 63 - __builtin_clzll(m0) : -1;
            if (l1 >= l0) {
                st.carry_min = readlane64(mz1, l1);
                st.carry_tax = __builtin_amdgcn_readlane(t1, l1);
            } else {
                st.carry_min = readlane64(mz0, l0);
                st.carry_tax = __builtin_amdgcn_readlane(t0, l0);
            }
        }
    }
    // distinct non-zero taxa of this tile -> (taxon, count) list
    if (any_hit) {
        for (;;) {
            const uint64_t pend0 = __ballot(t0 != 0), pend1 = __ballot(t1 != 0);
            if (!(pend0 | pend1)) break;
            uint32_t T;
            if (pend0)
                T = __builtin_amdgcn_readlane(t0, __builtin_ctzll(pend0));
            else
                T = __builtin_amdgcn_readlane(t1, __builtin_ctzll(pend1));
            const uint32_t cnt = __popcll(__ballot(t0 == T)) + __popcll(__ballot(t1 == T));
            if (t0 == T) t0 = 0;
            if (t1 == T) t1 = 0;
            const bool match = (uint32_t)lane < st.nlist && S.list_tax[lane] == T;
            const uint64_t mb = __ballot(match);
            if (mb) {
                if (match) S.list_cnt[lane] += cnt;
            } else if (st.nlist < (uint32_t)LIST_CAP) {
                if (lane == 0) {
                    S.list_tax[st.nlist] = T;
                    S.list_cnt[st.nlist] = cnt;
                }
                st.nlist++;
            } else {
                st.overflow = true;
            }
            wave_sync();
        }
    }
    NH_STAMP(5);
}

// ResolveTree (A.5) on the wave: lane i owns list entry i.  Returns the call; sets clade_hits.
__device__ __forceinline__ uint32_t resolve_tree(const DevDB &db, WaveLds &S, const int lane,
                                                 const FragState &st, const uint32_t total_kmers,
                                                 const double confidence, uint32_t &clade_hits) {
    const uint32_t nlist = st.nlist;
    const uint32_t *parent = db.parent;
    const bool own = (uint32_t)lane < nlist;
    const uint32_t my_t = own ? S.list_tax[lane] : 0;
    const uint32_t my_c = own ? S.list_cnt[lane] : 0;
    uint32_t call = 0;
    if (nlist == 1) {
        call = S.list_tax[0];
    } else {
        uint32_t score = 0;
        for (uint32_t j = 0; j < nlist; j++) {
            const uint32_t tj = S.list_tax[j], cj = S.list_cnt[j];
            if (own && is_a_ancestor_of_b(parent, tj, my_t)) score += cj;
        }
        const uint32_t top = wave_max(score);
        uint64_t best_mask = __ballot(own && score == top);
        while (best_mask) {
            const int j = __builtin_ctzll(best_mask);
            best_mask &= best_mask - 1;
            call = lowest_common_ancestor(parent, call, S.list_tax[j]);
        }
    }
    const uint32_t required = (uint32_t)ceil(confidence * (double)total_kmers);
    uint32_t s = wave_sum((own && my_t == call) ? my_c : 0u);  // hits exactly at the call
    while (call && s < required) {
        s = wave_sum((own && is_a_ancestor_of_b(parent, call, my_t)) ? my_c : 0u);
        if (s >= required) break;
        call = parent[call];
    }
    if (call && st.hit_groups < db.min_hit_groups) call = 0;
    clade_hits = 0;
    if (call) clade_hits = wave_sum((own && is_a_ancestor_of_b(parent, call, my_t)) ? my_c : 0u);
    return call;
}

constexpr uint32_t PREF_LANES = 42;  // dwords a tile can need: (3 + 128 + 30 + 3) / 4 <= 41

template <bool LINEAR, bool STD, bool PROF>
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK) void k_classify(
    const DevDB db, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ seq_off,
    const uint64_t n_frag, const int mates, const double confidence, Result *__restrict__ out,
    uint32_t *__restrict__ kmer_taxa, const uint64_t *__restrict__ kmer_taxa_off,
    unsigned long long *__restrict__ counters, int *__restrict__ error_flag,
    unsigned long long *__restrict__ work, const uint32_t frag_chunk) {
    __shared__ WaveLds lds_all[WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds &S = lds_all[wib];

    // one-time LDS init: zero pads of the packed streams, sentinel tail of the candidate array
    if (lane < 24) {
        S.pk[lane] = 0;
        S.pa[lane] = 0;
    }
    for (int i = lane; i < CAND_PAD; i += 64) S.cand[TL + i] = NH_FULL;
    wave_sync();

    const uint32_t K = STD ? 35u : db.k;
    const uint32_t L = STD ? 31u : db.l;
    const uint32_t TQ = TL - (STD ? 4u : db.window);  // k-mers per tile
    const uint64_t lane_lt = (lane == 0) ? 0ull : (NH_FULL >> (64 - lane));
    // dword index of the last dword the caller guarantees readable (8 bytes of slack, see ABI)
    const uint64_t last_dw = (seq_off[n_frag * (uint64_t)mates] + 4) >> 2;
    const uint32_t pl = (uint32_t)lane < PREF_LANES ? (uint32_t)lane : PREF_LANES - 1;

    // the dword stream of the tile that starts at byte g0: 4 bases per lane, coalesced
    auto load_tile = [&](uint64_t g0) -> uint32_t {
        uint64_t dw = (g0 >> 2) + pl;
        dw = dw < last_dw ? dw : last_dw;
        return reinterpret_cast<const uint32_t *>(bases)[dw];
    };

    uint32_t acc_frag = 0, acc_class = 0, acc_lookups = 0;
    uint64_t acc_bases = 0;
    bool bad_input = false;

    // speculative one-tile-ahead prefetch: pref_g0 is the byte offset w_pref was loaded for
    uint64_t pref_g0 = ~0ull;
    uint32_t w_pref = 0;

    uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tprev = PROF ? __builtin_readcyclecounter() : 0;

    // Dynamic distribution: every wave pulls chunks of consecutive fragments from one counter, so
    // late-starting (non-resident) workgroups of the grid find no work instead of a static share.
    for (;;) {
    unsigned long long cbeg = 0;
    if (lane == 0) cbeg = atomicAdd(work, (unsigned long long)frag_chunk);
    cbeg = readlane64(cbeg, 0);
    if (cbeg >= n_frag) break;
    const uint64_t cend = cbeg + frag_chunk < n_frag ? cbeg + frag_chunk : n_frag;
    for (uint64_t f = cbeg; f < cend; f++) {
        const uint64_t s0 = f * (uint64_t)mates;
        const uint64_t o0 = seq_off[s0], o1 = seq_off[s0 + 1];
        const uint64_t o2 = mates == 2 ? seq_off[s0 + 2] : o1;
        if (((o1 - o0) | (o2 - o1)) >> 31) bad_input = true;  // sequences of 2 Gbases and more
        const uint32_t n0 = (uint32_t)(o1 - o0), n1 = (uint32_t)(o2 - o1);
        const uint32_t nk0 = n0 >= K ? n0 - K + 1 : 0;
        const uint32_t nk1 = n1 >= K ? n1 - K + 1 : 0;
        // the next fragment of this chunk starts where this one ends (prefetch across fragments)
        const uint64_t next_frag_g0 = f + 1 < cend ? o2 : ~0ull;
        const uint64_t kt_base = kmer_taxa ? kmer_taxa_off[f] : 0;

        FragState st;
        st.nlist = 0;
        st.hit_groups = 0;
        st.carry_min = NH_FULL;
        st.carry_tax = 0;
        st.overflow = false;

        for (int m = 0; m < mates; m++) {
            const uint32_t n = m ? n1 : n0;
            const uint32_t nk = m ? nk1 : nk0;
            const uint64_t sb = m ? o1 : o0;
            if (m == 1 && db.reset_per_mate) {
                st.carry_min = NH_FULL;
                st.carry_tax = 0;
            }
            for (uint32_t q0 = 0; q0 < nk; q0 += TQ) {
                const uint64_t g0 = sb + q0;
                NH_STAMP(0);
                const uint32_t w = (pref_g0 == g0) ? w_pref : load_tile(g0);
                // guess the tile after this one and start its load now
                uint64_t ng0;
                if (q0 + TQ < nk)
                    ng0 = g0 + TQ;
                else if (m == 0 && mates == 2)
                    ng0 = o1;
                else
                    ng0 = next_frag_g0;
                if (ng0 != ~0ull) w_pref = load_tile(ng0);
                pref_g0 = ng0;

                const uint32_t nl_left = (n - L + 1) - q0;
                const uint32_t nlt = nl_left < (uint32_t)TL ? nl_left : (uint32_t)TL;
                const uint32_t nq_left = nk - q0;
                const uint32_t nqt = nq_left < TQ ? nq_left : TQ;
                const uint64_t kt = kt_base + (m ? (uint64_t)nk0 + 1 : 0) + q0;
                process_tile<LINEAR, STD, PROF>(db, S, lane, lane_lt, w, (uint32_t)g0 & 3u, nlt, nqt,
                                                st, kmer_taxa, kt, acc_lookups, prof, tprev);
            }
        }

        // ---- end of fragment --------------------------------------------------------------------
        const uint32_t total_kmers = nk0 + nk1;
        uint32_t call = 0, clade_hits = 0;
        if (st.nlist > 0) call = resolve_tree(db, S, lane, st, total_kmers, confidence, clade_hits);
        if (lane == 0) {
            uint4 rec;
            rec.x = call;
            rec.y = total_kmers;
            rec.z = clade_hits;
            rec.w = st.hit_groups;
            *reinterpret_cast<uint4 *>(&out[f]) = rec;
            if (kmer_taxa && mates == 2) kmer_taxa[kt_base + nk0] = TAXON_MATE_BORDER;
            if (st.overflow) atomicMax(error_flag, 1);
        }
        acc_frag += 1;
        acc_class += call ? 1 : 0;
        acc_bases += (uint64_t)n0 + n1;
        NH_STAMP(6);
    }
    }

    if (lane == 0) {
        if (PROF && counters)
            for (int i = 0; i < 8; i++) atomicAdd(&counters[CNT_N + i], (unsigned long long)prof[i]);
        if (counters) {
            atomicAdd(&counters[CNT_FRAGMENTS], (unsigned long long)acc_frag);
            atomicAdd(&counters[CNT_CLASSIFIED], (unsigned long long)acc_class);
            atomicAdd(&counters[CNT_BASES], (unsigned long long)acc_bases);
            atomicAdd(&counters[CNT_LOOKUPS], (unsigned long long)acc_lookups);
        }
        if (bad_input) atomicMax(error_flag, 2);
    }
}

// ---- synthetic table generation (bench/test support; stands in for HPRC.r2) -----------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// kraken2 CompactHashTable::CompareAndSet with a constant value: claim the first empty cell of
// the probe sequence, or stop at a cell that already holds this compacted key.
__global__ void k_synth_insert(uint32_t *table, uint64_t capacity, uint64_t cap_magic,
                               uint32_t value_bits, uint32_t value, uint64_t n_keys, uint64_t seed,
                               uint64_t key_mask, unsigned long long *size_counter) {
    uint64_t inserted = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_keys; i += stride) {
        const uint64_t mz = splitmix64(seed + i) & key_mask;
        const uint64_t hc = fmix64(mz);
        const uint32_t compacted = (uint32_t)(hc >> (32 + value_bits));
        const uint32_t cell = (compacted << value_bits) | value;
        uint64_t idx = mod_capacity(hc, capacity, cap_magic);
        for (uint64_t tries = 0; tries < capacity; tries++) {
            const uint32_t old = atomicCAS(&table[idx], 0u, cell);
            if (old == 0) {
                inserted++;
                break;
            }
            if ((old >> value_bits) == compacted) break;
            idx++;
            if (idx >= capacity) idx = 0;
        }
    }
    // one atomic per wave
    for (int d = 32; d >= 1; d >>= 1) inserted += __shfl_xor(inserted, d, 64);
    if ((threadIdx.x & 63) == 0 && inserted) atomicAdd(size_counter, (unsigned long long)inserted);
}

// ---- host-side launchers ---------------------------------------------------------------------
static bool is_std(const DevDB &db) {
    return db.k == 35 && db.l == 31 && db.revcom_version != 0 && db.min_hash == 0;
}

template <bool LINEAR, bool STD, bool PROF = false>
static void launch_variant(const DevDB &db, dim3 g, dim3 b, hipStream_t stream, const void *d_bases,
                           const void *d_seq_off, uint64_t n_frag, int mates, double confidence,
                           void *d_out, void *d_kmer_taxa, const void *d_kmer_taxa_off,
                           void *d_counters, int *d_error, unsigned long long *d_work,
                           uint32_t frag_chunk) {
    hipLaunchKernelGGL((k_classify<LINEAR, STD, PROF>), g, b, 0, stream, db, (const uint8_t *)d_bases,
                       (const uint64_t *)d_seq_off, n_frag, mates, confidence, (Result *)d_out,
                       (uint32_t *)d_kmer_taxa, (const uint64_t *)d_kmer_taxa_off,
                       (unsigned long long *)d_counters, d_error, d_work, frag_chunk);
}

hipError_t launch_classify(const DevDB &db, const void *d_bases, const void *d_seq_off,
                           uint64_t n_frag, int mates, double confidence, void *d_out,
                           void *d_kmer_taxa, const void *d_kmer_taxa_off, void *d_counters,
                           int *d_error, unsigned long long *d_work, uint32_t frag_chunk,
                           int grid_blocks, hipStream_t stream) {
    if (n_frag == 0) return hipSuccess;
    if (frag_chunk == 0) frag_chunk = 1;
    hipError_t me = hipMemsetAsync(d_work, 0, sizeof(unsigned long long), stream);
    if (me != hipSuccess) return me;
    uint64_t need = (n_frag + (uint64_t)WAVES_PER_BLOCK * frag_chunk - 1) / ((uint64_t)WAVES_PER_BLOCK * frag_chunk);
    int grid = (int)(need < (uint64_t)grid_blocks ? need : (uint64_t)grid_blocks);
    dim3 g(grid), b(WAVE * WAVES_PER_BLOCK);
    const bool std_geom = is_std(db);
    if (db.linear_probing && std_geom && getenv("NH_PHASE_PROF")) {
        // phase profile build of the default variant: d_counters must hold CNT_N + 8 words
        launch_variant<true, true, true>(db, g, b, stream, d_bases, d_seq_off, n_frag, mates, confidence,
                                         d_out, d_kmer_taxa, d_kmer_taxa_off, d_counters, d_error, d_work, frag_chunk);
    } else if (db.linear_probing) {
        if (std_geom)
            launch_variant<true, true>(db, g, b, stream, d_bases, d_seq_off, n_frag, mates, confidence,
                                       d_out, d_kmer_taxa, d_kmer_taxa_off, d_counters, d_error, d_work, frag_chunk);
        else
            launch_variant<true, false>(db, g, b, stream, d_bases, d_seq_off, n_frag, mates,
                                        confidence, d_out, d_kmer_taxa, d_kmer_taxa_off, d_counters,
                                        d_error, d_work, frag_chunk);
    } else {
        launch_variant<false, false>(db, g, b, stream, d_bases, d_seq_off, n_frag, mates, confidence,
                                     d_out, d_kmer_taxa, d_kmer_taxa_off, d_counters, d_error, d_work, frag_chunk);
    }
    return hipGetLastError();
}

int classify_blocks_per_cu() {
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_classify<true, true, false>,
                                                                WAVE * WAVES_PER_BLOCK, 0);
    if (e != hipSuccess || nb < 1) nb = 4;
    return nb > 8 ? 8 : nb;
}

hipError_t launch_synth_insert(uint32_t *table, uint64_t capacity, uint64_t cap_magic,
                               uint32_t value_bits, uint32_t value, uint64_t n_keys, uint64_t seed,
                               uint64_t key_mask, unsigned long long *d_size, hipStream_t stream) {
    if (n_keys == 0) return hipSuccess;
    hipLaunchKernelGGL(k_synth_insert, dim3(256 * 16), dim3(256), 0, stream, table, capacity,
                       cap_magic, value_bits, value, n_keys, seed, key_mask, d_size);
    return hipGetLastError();
}

}  // namespace nh
