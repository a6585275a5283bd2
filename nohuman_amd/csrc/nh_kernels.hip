// nh_kernels.hip -- gfx950 (CDNA4) kernels of the read-classification hot path.
//
// One 64-lane wavefront classifies fragments (reads or read pairs) end to end; it replaces the
// body of kraken2's ClassifySequence loop (minimizer scan -> compact-hash probe -> ResolveTree;
// SURVEY.md section 8a rows a5-a9, Appendix A.2-A.5) that nohuman reaches through
// /root/reference/src/lib.rs:22-48.  Integer / byte work bound by random HBM line fetches:
// no MFMA by design.
//
// Work unit = a GROUP of up to NSLOT tiles (a tile = 128 l-mers = 2 per lane = 128-(k-l) k-mers;
// a 150-bp read is one tile, a read pair or two single reads one group):
//   SCAN each tile of the group
//     global bases (one coalesced dword per lane, prefetched one tile ahead)
//     -> 2-bit packed stream in LDS (1 byte per lane)
//     -> per lane two l-mers by one 64-bit funnel read; ONE 32-base reverse complement serves both
//     -> canonical/spaced/toggled candidates in LDS -> per lane two k-mer minimizers (window min)
//     -> run starts (minimizer != previous non-ambiguous minimizer) appended to the LDS queue
//   HASH the queue densely (fmix64, exact hc % capacity), then PROBE it with lane refill: a lane
//     owns one lookup at a time and pulls the next queued lookup the moment it resolves -- the wave
//     does not idle on the longest probe sequence (linear probing at load 0.7 is heavy-tailed).  On
//     the hot variants a round's bytes are fetched quad-cooperatively: four lanes read 64 contiguous
//     bytes (16 cells) of the owner's 128-byte line (probe_queue_quad)
//   POST each tile: taxa back from LDS -> per-taxon hit counts, hit groups; at the end of a
//     fragment ResolveTree runs on the wave.
// The kernel is instruction-issue and latency sensitive (profiles/): tile-local arithmetic is
// 32-bit, control flow is wave-uniform wherever possible, kraken2's default k=35/l=31 geometry is a
// compile-time specialisation (STD) next to the fully general variant, and fragments are handed
// out dynamically in chunks so that non-resident workgroups of the grid cost nothing.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "nh_device.h"

namespace nh {

#define NH_FULL 0xFFFFFFFFFFFFFFFFull
#ifndef NH_WIDE_CELLS
#define NH_WIDE_CELLS 16  // cells per round of an old lookup (8, 12 or 16)
#endif
#ifndef NH_WIDE_AFTER
#define NH_WIDE_AFTER 2  // rounds after which a lookup examines 16 cells per round instead of 4
#endif
#ifndef NH_R2_CELLS
#define NH_R2_CELLS 4    // cells of a lookup's second round
#endif

// window reads of idle lanes stay inside the candidate array: k-l+2 pad entries (k-l = 4 for
// kraken2's default geometry, <= 64 in general)
template <bool STD> struct CandPad { static constexpr int value = STD ? 6 : 66; };

#ifndef NH_MIN_WAVES
#define NH_MIN_WAVES 5  // waves per SIMD the hot variant is built for (LDS: 5 workgroups of 31.5 KB per CU)
#endif
#ifndef NH_NSLOT
#define NH_NSLOT 4
#endif
#ifndef NH_LPO
#define NH_LPO 4  // lanes that fetch an owner's probe round together (probe_queue_quad): 4 = 16 cells per round, 2 = 8
#endif
#ifndef NH_QCAP
#define NH_QCAP 256  // 5 waves per SIMD need <= 31 KB of LDS per workgroup (a 32 KB one fits only 4 times: profiles/r02_tuning.txt)
#endif
constexpr int NSLOT = NH_NSLOT;  // tiles scanned before one shared probe phase
static_assert(NSLOT <= 4, "carry_pack holds four 16-bit queue indices per group: one inherited-minimizer entry per tile");
constexpr int QCAP_SHORT = NH_QCAP;  // queue entries per group: a tile joins only if it is sure to fit (< 512)
#ifndef NH_QCAP_GENERIC
#define NH_QCAP_GENERIC 288  // the generic kernel keeps one packed stream instead of NSLOT: room for a longer queue
#endif
constexpr int QCAP_GENERIC = NH_QCAP_GENERIC;
constexpr uint32_t QTAX_SKIP = 0xFFFFFFFFu;  // queue entry dropped by the min-hash filter

constexpr int PKW = 18;  // words of a tile's packed stream: 16 of data + the 2 zero words a funnel read may touch

struct alignas(16) SlotLds {  // a scanned tile waiting for its probe results (written by lane 0)
    uint32_t f_lo, f_hi;    // fragment
    uint32_t koff, pad0;    // k-mer index of the tile within its fragment
    uint32_t nqt, qbase, nruns;
    uint32_t pad1;
    uint32_t last_lane;     // 2*lane+slot of the last unambiguous k-mer, 0xFFFFFFFF if none
    uint32_t flags;         // 1 = last tile of its fragment, 2 = last tile of mate 0, mate 1 follows
    uint32_t nk0, total_kmers;
};

// the generic kernel keeps the probe results apart from the queue entries; STD aliases them
template <bool STD, int QC> struct QTax { uint32_t v[2][QC]; };
template <int QC> struct QTax<true, QC> {};

// LDS slice of one wave.  NPK = tiles whose packed streams exist at a time (1: the generic kernel encodes
// and scans tile by tile; NSLOT: the short-read kernel encodes a batch first), QC = queue entries per
// group.  Both kernels are sized to 31 KB per workgroup: five workgroups per CU.
template <bool STD, int NPK, int QC>
struct WaveLdsT {
    static constexpr int QCAP = QC;
#if defined(NH_LDS_PAD) && NH_LDS_PAD > 0
    uint32_t occupancy_pad[NH_LDS_PAD / 4];  // tuning aid: lowers the number of resident workgroups
#endif
    unsigned long long acc[4];  // fragments, classified, bases, lookups of this wave (lane 0 adds)
    uint64_t last_dw;           // last readable dword of the bases buffer
    uint4 frag_state;        // FragState between post_group calls: nlist, hit_groups, carry_tax, overflow
    SlotLds slot[2][NSLOT];  // [parity of the group][tile]
    uint16_t ps[2][NSLOT][WAVE];  // per-lane packed k-mer state of the tiles in flight
    // 2-bit packed bases of a tile: base i' of the tile frame at bit 2*(255-i'); 64 B + zero pad.  One per
    // tile of a group: the short-read kernel encodes a whole batch of tiles before it scans them
    uint32_t pk[NPK][PKW];
    uint32_t pa[NPK][PKW];  // same layout, value 1 where the base is ambiguous
    // result records of the fragments the last post_group finished: stored to global memory by the next
    // turn, right before its probe phase (flush_records)
    uint4 stage_rec[NSLOT];
    uint64_t stage_f[NSLOT];
    uint32_t stage_n;
    uint64_t cand[TL + CandPad<STD>::value];
    uint64_t q[2][QC];  // [parity] queue: run-start minimizers, hashed in place (see probe_queue)
    QTax<STD, QC> qtax;       // taxon found for each queued run (generic kernel only, see tax_at)
    // (taxon, count) list of the fragment being post-processed: tiles are post-processed strictly
    // in input order, so one list (and one FragState, kept in registers) serves all fragments
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

// number of set bits of a wave mask below this lane
__device__ __forceinline__ uint32_t below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Kernel arguments are read on demand from the kernarg segment (constant address space, scalar
// loads).  Each phase launders the pointer first, which stops the compiler from hoisting every
// argument load to the kernel entry and pinning ~50 SGPRs for the whole kernel.
typedef const __attribute__((address_space(4))) KArgs *KArgsP;
__device__ __forceinline__ KArgsP launder(KArgsP p) {
    const uint64_t v = (uint64_t)p;
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (KArgsP)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int src) {
    uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src);
    uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
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

// 4 ASCII bytes -> one byte of packed 2-bit codes (first base in bits 7:6); `diff` is non-zero in
// every byte that is not one of ACGTacgt (SWAR, no per-byte work on the common path)
__device__ __forceinline__ uint32_t encode4(uint32_t w, uint32_t &diff) {
    const uint32_t up = w & 0xDFDFDFDFu;          // fold case
    const uint32_t x = (up >> 1) & 0x03030303u;   // A0 C1 G3 T2
    const uint32_t code = x ^ ((x >> 1) & 0x01010101u);  // A0 C1 G2 T3
    const uint32_t tbit = (x >> 1) & ~x & 0x01010101u;   // 1 where the byte decodes as T
    const uint32_t recon = (0x41414141u | (x << 1)) ^ (tbit * 0x11u);  // canonical letter of code
    diff = recon ^ up;
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

#define NH_STAMP(i)                                        \
    do {                                                   \
        if (PROF) {                                        \
            const uint64_t _t = __builtin_readcyclecounter(); \
            prof[i] += _t - tprev;                         \
            tprev = _t;                                    \
        }                                                  \
    } while (0)

// ENCODE one tile into the packed streams of `slot`: 4 bases per lane (dword stream `w`, tile frame
// starting `sh` bytes in), of which bytes [sh, sh + nbases) belong to this sequence.  What follows them
// is whatever lies behind the sequence in the caller's buffer (the next read, or -- when records are
// classified in place inside their FASTQ text -- a newline and the quality line): it must neither count
// as ambiguous nor send the tile down the slow path.  Returns "the tile has an ambiguous base".
template <bool STD, class WL>
__device__ __forceinline__ bool encode_tile(WL &S, const int lane, const uint32_t slot, const uint32_t w,
                                            const uint32_t sh, const uint32_t nbases) {
    uint32_t diff;
    const uint32_t codes = encode4(w, diff);
    reinterpret_cast<uint8_t *>(S.pk[slot])[63 - lane] = (uint8_t)codes;
    // bytes of this lane's dword that are bases of the sequence: frame positions [4 lane, 4 lane + 4) cut to [sh, hi)
    const uint32_t p0 = 4u * (uint32_t)lane, hi = sh + nbases;
    uint32_t m = 0xFFFFFFFFu;
    if (p0 < sh) m = sh - p0 >= 4u ? 0u : m << (8u * (sh - p0));
    if (p0 + 4u > hi) m = hi <= p0 ? 0u : m & (0xFFFFFFFFu >> (8u * (p0 + 4u - hi)));
    bool has_amb = __ballot((diff & m) != 0) != 0;
    if (has_amb) {  // exact flags, restricted to the bases of this tile
        const uint32_t bad = ambig4(w, p0, sh, hi);
        reinterpret_cast<uint8_t *>(S.pa[slot])[63 - lane] = (uint8_t)bad;
    }
    return has_amb;
}

// SCAN one encoded tile (streams of `slot`): l-mers [q0, q0+nlt) / k-mers [q0, q0+nqt) of a sequence
// whose tile frame starts `sh` bytes into its dword stream.  Appends the run-start minimizers to
// S.q[par][qbase ...], returns their number, and leaves in `ps` the lane's packed per-k-mer state
// (bit0/1 = k-mer 2t / 2t+1 is valid and unambiguous, bit 2 = k-mer 2t+1 starts a run, bits 3-10 = 1 + index of the run
// that covers k-mer 2t, 0 = continuation of the run that entered the tile).
template <bool STD, bool PROF, class WL>
__device__ __forceinline__ uint32_t scan_body(KArgsP ap, WL &S, const int lane, const uint32_t slot,
                                              const bool has_amb, const uint32_t sh, const uint32_t nlt,
                                              const uint32_t nqt, const uint32_t par,
                                              const uint32_t qbase, uint64_t &carry_min,
                                              uint32_t &ps, int &last_lane,
                                              uint64_t (&prof)[12], uint64_t &tprev) {
    ap = launder(ap);
    const uint32_t L = STD ? 31u : ap->db.l;
    const uint32_t W = STD ? 4u : ap->db.window;
    const uint64_t LMASK = STD ? ((1ull << 62) - 1) : ap->db.lmer_mask;
    const int RV = STD ? 1 : ap->db.revcom_version;
    const uint64_t SPACED = ap->db.spaced_mask, TOGGLE = ap->db.toggle;

    // ---- 2. two l-mers per lane -> candidates --------------------------------------------------
    {
        const uint32_t j1 = sh + 2u * lane + L;  // frame index of the last base of l-mer 2t+1
        const uint32_t s = 2u * (255u - j1);
        const uint64_t wv = funnel_read(S.pk[slot], s);
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
        const uint64_t c0 = (umin64(lm0, rc0) & SPACED) ^ TOGGLE;
        const uint64_t c1 = (umin64(lm1, rc1) & SPACED) ^ TOGGLE;
        bool dead0 = 2u * lane >= nlt, dead1 = 2u * lane + 1 >= nlt;
        if (has_amb) {
            const uint64_t wa = funnel_read(S.pa[slot], s);
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
        uint64_t m0 = umin64(first, mid);
        uint64_t m1 = (!STD && W == 0) ? last1 : umin64(mid, last1);
        if (!STD && has_amb && W > L) {
            // kraken2's scanner empties its queue at an ambiguous base, so a k-mer's window holds only
            // the l-mers that lie wholly after the last ambiguous base.  While k-l <= l
            // every l-mer of the window before that base contains it (is +inf) and the plain min is
            // the same thing; beyond that, walk back from the last l-mer and stop at the first dead one.
            uint64_t a0 = NH_FULL, a1 = NH_FULL;
            bool open0 = true, open1 = true;
            for (uint32_t i = W + 1; i-- > 0;) {
                const uint64_t c0 = S.cand[qi0 + i], c1 = S.cand[qi1 + i];
                open0 &= c0 != NH_FULL;
                open1 &= c1 != NH_FULL;
                if (open0) a0 = umin64(a0, c0);
                if (open1) a1 = umin64(a1, c1);
            }
            m0 = a0;
            m1 = a1;
        }
        v0 = (qi0 < nqt) & (last0 != NH_FULL);
        v1 = (qi1 < nqt) & (last1 != NH_FULL);
        if (has_amb && ap->db.ambig_rule != 0) {
            // nh_options.ambiguity_rule 1 (mmscanner.h is_ambiguous(): queue_pos < k-l || last_ambig): a k-mer counts
            // only when k-l l-mers have been queued since the last ambiguous base, i.e. when every l-mer of its window
            // but the first is alive (the first may be dead: the window of the first clean k-mer holds k-l l-mers;
            // +inf drops out of the min by itself).  Rule 0 asks for the last l-mer only.
            if (STD) {
                const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&S.cand[qi0]);
                const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(&S.cand[qi0 + 2]);
                const bool mid_alive = (b.x != NH_FULL) & (b.y != NH_FULL);
                v0 &= mid_alive & (a.y != NH_FULL);
                v1 &= mid_alive & (last0 != NH_FULL);
            } else {
                for (uint32_t i = 1; i < W; i++) {  // (i == W is last0 / last1)
                    v0 &= S.cand[qi0 + i] != NH_FULL;
                    v1 &= S.cand[qi1 + i] != NH_FULL;
                }
            }
        }
        mz0 = m0 ^ TOGGLE;
        mz1 = m1 ^ TOGGLE;
    }

    NH_STAMP(2);
    // ---- 4. run starts: minimizer differs from the previous non-ambiguous one -----------------
    uint64_t prev_in;
    if (!has_amb) {
        prev_in = __shfl_up(mz1, 1, 64);
        if (lane == 0) prev_in = carry_min;
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
        prev_in = (lane > 0 && hx) ? vx : carry_min;
    }
    const uint64_t prev1 = v0 ? mz0 : prev_in;
    const bool new0 = v0 & (mz0 != prev_in);
    const bool new1 = v1 & (mz1 != prev1);

    // ---- 5. append run starts to the LDS queue; remember the last minimizer --------------------
    const uint64_t b0 = __ballot(new0), b1 = __ballot(new1);
    const uint32_t ex = below(b0) + below(b1);
    const uint32_t nruns = __popcll(b0) + __popcll(b1);
    if (new0) S.q[par][qbase + ex] = mz0;
    if (new1) S.q[par][qbase + ex + (new0 ? 1u : 0u)] = mz1;
    const uint32_t r0p = ex + (new0 ? 1u : 0u);  // 1 + run index of k-mer 2t (0 = carried run)
    ps = (v0 ? 1u : 0u) | (v1 ? 2u : 0u) | (new1 ? 4u : 0u) | (r0p << 3);
    {
        const uint64_t m1 = __ballot(v1), m0 = __ballot(v0);
        last_lane = -1;
        if (m0 | m1) {
            const int l1 = m1 ? 63 - __builtin_clzll(m1) : -1;
            const int l0 = m0 ? 63 - __builtin_clzll(m0) : -1;
            if (l1 >= l0) {
                carry_min = readlane64(mz1, l1);
                last_lane = 2 * l1 + 1;
            } else {
                carry_min = readlane64(mz0, l0);
                last_lane = 2 * l0;
            }
        }
    }
    NH_STAMP(3);
    return nruns;
}


// One tile of the generic kernel: encode (slot 0), start the prefetch of a later tile, scan.
template <bool STD, bool PROF, class WL>
__device__ __forceinline__ uint32_t scan_tile(KArgsP ap, WL &S, const int lane,
                                              const uint32_t w,
                                              const uint32_t sh, const uint32_t nlt,
                                              const uint32_t nqt, const uint32_t par,
                                              const uint32_t qbase, uint64_t &carry_min,
                                              uint32_t &ps, int &last_lane,
                                              const uint32_t *pf_ptr, const bool pf_on,
                                              uint32_t &w_pref,
                                              uint64_t (&prof)[12], uint64_t &tprev) {
    const uint32_t L = STD ? 31u : launder(ap)->db.l;
    const bool has_amb = encode_tile<STD>(S, lane, 0u, w, sh, nlt + L - 1);
    wave_sync();
    // `w` has been consumed: start the load of the next tile's bases now, so that no wait for
    // `w` can be widened into a wait for the prefetch (vmcnt retires loads in issue order)
    if (pf_on) w_pref = *pf_ptr;
    NH_STAMP(1);
    return scan_body<STD, PROF>(ap, S, lane, 0u, has_amb, sh, nlt, nqt, par, qbase, carry_min, ps, last_lane, prof,
                                tprev);
}

// Where the taxon of queued run r is stored: the generic kernel has its own array (it also marks
// entries dropped by the min-hash filter there); the STD kernel reuses the low dword of the queue
// entry itself, which the owning lane has copied to registers before it writes the result.
template <bool STD, class WL>
__device__ __forceinline__ uint32_t &tax_at(WL &S, uint32_t par, uint32_t r) {
    if constexpr (STD)
        return reinterpret_cast<uint32_t *>(&S.q[par][r])[0];
    else
        return S.qtax.v[par][r];
}

// The lookup a lane currently owns.  It persists across probe_queue calls: a lane may carry an
// unresolved lookup of group g into the probe phase of group g+1 (software pipelining: tiles of
// group g are only post-processed after that phase), so no wave ever drains a probe tail alone.
struct LaneLookup {
    uint32_t busy;       // owns an unresolved lookup
    uint32_t r;          // queue index | parity << 9 | table copy << 10
    uint64_t pos;        // next cell to examine (32 bits used when CAP32)
    uint64_t first_pos;  // double hashing: home cell
    uint64_t step;       // double hashing: stride
    uint32_t ckey;       // compacted key << value_bits
    uint32_t budget;     // rounds left before the whole table was seen
};

// Which copy of the table a lookup probes (DevDB::copy_stride): the one where its HOME cell lies in the
// first 2^copy_shift cells of a 128-byte line.  It stays there: later lines are entered at their start.
__device__ __forceinline__ uint32_t pick_copy(uint32_t home, uint32_t copy_shift) {
    return (home & 31u) >> copy_shift;  // 0 when there is one copy (copy_shift 5)
}
// Where a probe round at cell p32 reads in copy j; in_line = cells from p32 to the end of its line.
__device__ __forceinline__ const uint32_t *probe_src(const uint32_t *table, uint64_t copy_stride,
                                                     uint32_t copy_shift, uint32_t j, uint32_t p32,
                                                     uint32_t &in_line) {
    in_line = 32u - ((p32 - (j << copy_shift)) & 31u);
    return table + (uint64_t)j * copy_stride + p32;
}

// Up to NSLOT entries of a group's queue are look-ups of an inherited minimizer (first tile of a segment
// of a split long read): 1 + queue index in 16-bit fields of a wave-uniform word, 0 = unused.
__device__ __forceinline__ bool is_carry_entry(const uint64_t pack, const uint32_t r) {
    const uint32_t k = r + 1u;
    return ((uint32_t)pack & 0xFFFFu) == k || ((uint32_t)(pack >> 16) & 0xFFFFu) == k ||
           ((uint32_t)(pack >> 32) & 0xFFFFu) == k || (uint32_t)(pack >> 48) == k;
}
__device__ __forceinline__ uint32_t carry_entries(const uint64_t pack) {
    return (((uint32_t)pack & 0xFFFFu) != 0) + (((uint32_t)(pack >> 16) & 0xFFFFu) != 0) +
           (((uint32_t)(pack >> 32) & 0xFFFFu) != 0) + ((uint32_t)(pack >> 48) != 0);
}

// The 16-byte load of a probe round.  -DNH_PROBE_NT=1 marks it non-temporal (`global_load_dwordx4 ... nt`): the gather
// microbenchmark measured 50.75 against 47.20 G probes/s for it (profiles/r02_gather_bench.txt; sc1 / sc0 variants: no
// difference from plain) -- whether the product kernels gain is the A/B of profiles/r06_cache_policy.txt.
#ifndef NH_PROBE_NT
#define NH_PROBE_NT 0
#endif
typedef uint32_t nh_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 probe_load16(const uint32_t *p) {
#if NH_PROBE_NT
    const nh_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nh_u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const uint4 *>(p);
#endif
}

// stopping cell among 4 loaded cells: the lowest j >= lo that is empty or holds the key
__device__ __forceinline__ void scan4(const uint4 &c, uint32_t ckey, uint32_t vmask, uint32_t lo, uint32_t &res,
                                      uint32_t &resj) {
    const uint32_t cells[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int j = 3; j >= 0; j--) {
        const uint32_t cell = cells[j];
        const uint32_t x = cell ^ ckey;  // key bits vanish on a match
        const bool stop = ((uint32_t)j >= lo) & ((x <= vmask) | ((cell & vmask) == 0));
        res = stop ? x : res;
        resj = stop ? (uint32_t)j : resj;
    }
}

// HASH the queue S.q[par][0, qn) and PROBE it: CompactHashTable::Get (A.4) for every entry, result
// in tax_at(par, r).  Returns when every entry has been handed to a lane AND every lookup of the
// other parity (the previous group) is resolved; lookups of this group may still be in flight.
// Queue entry after the hash pass (linear probing):
//   CAP32 (capacity < 2^32 - 256):  low dword = home cell, high dword = compacted key << value_bits
//   otherwise:                      home cell << key_bits | compacted key   (<= 63 bits, checked at open)
// Double hashing keeps the hash code itself.
// (Tried in round 2 and dropped: issuing the first round of EVERY queue entry from the hash pass, 2.4 loads
// in flight per lane.  Half the round trips per group, 10 % fewer instructions -- and 16 % slower: a fully
// divergent wave-load occupies the CU's L1 path for ~152 cycles (2.4 per distinct line,
// profiles/r02_mem_study.txt), and that path was the second wall next to HBM.  probe_queue_quad below is
// what came of it.)
template <bool LINEAR, bool STD, bool CAP32, bool PROF, class WL>
__device__ __forceinline__ void probe_queue(KArgsP ap, WL &S, const int lane,
                                            const uint32_t par,
                                            const uint32_t qn, LaneLookup &lk, const bool count_lookups,
                                            const uint64_t carry_pack, uint64_t (&prof)[12],
                                            uint64_t &tprev) {
    ap = launder(ap);
    const uint64_t MIN_HASH = STD ? 0ull : ap->db.min_hash;
    const uint32_t vbits = ap->db.value_bits;
    const uint32_t vmask = ap->db.vmask;
    const uint32_t kbits = 32 - vbits;
    const uint64_t cap = ap->db.capacity;
    const uint64_t magic = ap->db.cap_magic;
    const uint32_t max_rounds = ap->db.max_chunks;
    const uint32_t *const table = ap->db.table;
    const uint64_t copy_stride = ap->db.copy_stride;
    const uint32_t copy_shift = ap->db.copy_shift;

    uint32_t qhead = 0;  // next queue entry to hand out (uniform)
    const uint32_t qend = qn;
    {
        // ---- 6a. dense hash pass ------------------------------------------------------------------
        for (uint32_t r0 = 0; r0 < qn; r0 += 64) {
            const uint32_t r = r0 + lane;
            const bool act = r < qn;
            const uint64_t hc = fmix64(S.q[par][act ? r : 0u]);
            const bool look = act & !(MIN_HASH != 0 && hc < MIN_HASH);
            uint64_t e = hc;
            if (LINEAR) {
                const uint64_t home = mod_capacity(hc, cap, magic);
                const uint32_t compacted = (uint32_t)(hc >> (32 + vbits));
                e = CAP32 ? (((uint64_t)(compacted << vbits) << 32) | (uint32_t)home)
                          : ((home << kbits) | compacted);
            }
            if (act) {
                S.q[par][r] = e;
                if constexpr (!STD) S.qtax.v[par][r] = look ? 0u : QTAX_SKIP;
            }
            // (a segment's look-up of the minimizer it inherits is not one kraken2 makes: not counted)
            const uint32_t nlook = __popcll(__ballot(look && !is_carry_entry(carry_pack, r)));
            if (lane == 0 && count_lookups) S.acc[CNT_LOOKUPS] += nlook;
        }
        wave_sync();
        NH_STAMP(4);
    }

    // ---- 6b. probe with lane refill ---------------------------------------------------------------
    uint32_t busy = lk.busy, r = lk.r, ckey = lk.ckey, budget = lk.budget;
    uint64_t pos = lk.pos, first_pos = lk.first_pos, step = lk.step;
    for (;;) {
        if (qhead < qend) {
            const uint64_t idle_mask = __ballot(busy == 0);
            if (idle_mask) {
                const uint32_t my = qhead + below(idle_mask);
                if (busy == 0 && my < qend) {
                    {
                        const uint64_t e = S.q[par][my];
                        bool skip = false;
                        if constexpr (!STD) {
                            skip = S.qtax.v[par][my] == QTAX_SKIP;
                            if (skip) S.qtax.v[par][my] = 0;
                        }
                        if (!skip) {
                            r = my | (par << 9);
                            budget = max_rounds;
                            if (LINEAR) {
                                if (CAP32) {
                                    pos = (uint32_t)e;
                                    ckey = (uint32_t)(e >> 32);
                                    r |= pick_copy((uint32_t)e, copy_shift) << 10;
                                } else {
                                    pos = e >> kbits;
                                    ckey = (uint32_t)(e & ((1ull << kbits) - 1)) << vbits;
                                }
                            } else {
                                pos = mod_capacity(e, cap, magic);
                                first_pos = pos;
                                step = mod_capacity((e >> 8) | 1, cap, magic);
                                ckey = (uint32_t)(e >> (32 + vbits)) << vbits;
                            }
                            busy = 1;
                        }
                    }
                }
                const uint32_t taken = __popcll(idle_mask);
                qhead = qhead + taken < qend ? qhead + taken : qend;
            }
        }
        // done when everything is handed out and no lane still works for the previous group
        if (qhead >= qend && __ballot(busy != 0 && ((r >> 9) & 1u) != par) == 0) break;
        if (PROF) prof[11] += 1;  // probe rounds executed (not cycles)
        if (busy) {
            if (LINEAR) {
                // One round: cells from `pos` on (unaligned 16-byte loads), of which only those
                // before the end of the 128-byte line (the unit HBM delivers) and before the end of
                // the table count.  All loaded cells are scanned; the first stopping cell decides,
                // and it only counts if it is one of the nvalid eligible ones.
                uint32_t nvalid, in_line;
                const uint32_t *src;
                if (CAP32) {
                    const uint32_t p32 = (uint32_t)pos;
                    src = probe_src(table, copy_stride, copy_shift, (r >> 10) & 7u, p32, in_line);
                    const uint32_t room = (uint32_t)cap - p32;
                    nvalid = in_line < room ? in_line : room;
                } else {
                    in_line = 32u - ((uint32_t)pos & 31u);
                    const uint64_t room = cap - pos;
                    nvalid = room < in_line ? (uint32_t)room : in_line;
                    src = table + pos;
                }
                uint32_t res = 0, resj = 64, lo = 0;
                // A lookup's first NH_WIDE_AFTER rounds examine 4 cells with ONE 16-byte load that
                // must not leave the line: if fewer than 4 cells remain it starts up to 3 cells
                // early and those are skipped (a load across the line end would cost a second
                // fabric request for cells that do not count).  Most lookups end there.  An older
                // lookup is in a long probe run -- linear probing at load 0.7 is heavy-tailed, and
                // the slowest lookup of a group decides when the group can be post-processed -- so
                // it examines up to 16 cells per round from then on.
                const uint32_t age = max_rounds - budget;  // rounds this lookup has had
                const uint32_t lim = age >= (uint32_t)NH_WIDE_AFTER ? (uint32_t)NH_WIDE_CELLS
                                     : age == 1u                    ? (uint32_t)NH_R2_CELLS
                                                                    : 4u;
                const bool wide = lim > 4u;
                nvalid = nvalid < lim ? nvalid : lim;
                lo = (!wide && in_line < 4u) ? 4u - in_line : 0u;
                const uint4 c0 = probe_load16(src - lo);
                constexpr int WCH = NH_WIDE_CELLS / 4;  // 16-byte chunks of a wide round
                uint4 cw[WCH];
                if (wide) {  // only the chunks that hold eligible cells are loaded (every lane-load costs the L1 a cycle)
                    const uint32_t nch = (nvalid + 3u) >> 2;
#pragma unroll
                    for (int q = 1; q < WCH; q++) {
                        cw[q] = make_uint4(0, 0, 0, 0);  // (a stop found in an unloaded chunk lies beyond nvalid: ignored)
                        if ((uint32_t)q < nch) cw[q] = probe_load16(src + 4 * q);
                    }
                }
                if (wide) {
#pragma unroll
                    for (int j = 4 * WCH - 1; j >= 4; j--) {
                        const uint4 &cq = cw[j >> 2];
                        const uint32_t cell = (j & 3) == 0 ? cq.x : (j & 3) == 1 ? cq.y : (j & 3) == 2 ? cq.z : cq.w;
                        const uint32_t x = cell ^ ckey;
                        const bool stop = (x <= vmask) | ((cell & vmask) == 0);
                        res = stop ? x : res;
                        resj = stop ? (uint32_t)j : resj;
                    }
                }
                scan4(c0, ckey, vmask, lo, res, resj);
                // a re-read chunk repeats cells of an earlier one: its stops can only come after
                // an identical earlier stop, so resj < nvalid is exact
                const bool found = resj < lo + nvalid;
                if (CAP32) {
                    uint32_t np = (uint32_t)pos + nvalid;
                    pos = np >= (uint32_t)cap ? 0u : np;
                } else {
                    const uint64_t np = pos + nvalid;
                    pos = np >= cap ? 0 : np;
                }
                budget--;
                if (found | (budget == 0)) {
                    tax_at<STD>(S, (r >> 9) & 1u, r & 0x1FFu) = (found && res <= vmask) ? res : 0u;
                    busy = 0;
                }
            } else {
                const uint32_t cell = table[pos];
                const uint32_t x = cell ^ ckey;
                bool end = false;
                uint32_t val = 0;
                if ((cell & vmask) == 0) {
                    end = true;
                } else if (x <= vmask) {
                    val = x;
                    end = true;
                } else {
                    pos += step;
                    if (pos >= cap) pos -= cap;
                    if (pos == first_pos) end = true;
                }
                if (end) {
                    tax_at<STD>(S, (r >> 9) & 1u, r & 0x1FFu) = val;
                    busy = 0;
                }
            }
        }
    }
    lk.busy = busy;
    lk.r = r;
    lk.ckey = ckey;
    lk.budget = budget;
    lk.pos = pos;
    lk.first_pos = first_pos;
    lk.step = step;
    wave_sync();
    NH_STAMP(5);
}

// ---- quad-cooperative probing (default geometry, linear probing, 32-bit positions) -------------------
// What binds the probe phase is measured (profiles/r02_mem_study.txt, tools/gather_bench mode 2): every
// L2 miss moves a whole 128-byte line over the fabric, a lookup whose 2nd / 3rd round comes microseconds
// later fetches its line AGAIN, and the L1 takes a wave-load at ~2.4 cycles per DISTINCT LINE -- lanes
// that read neighbouring bytes of one line ride for free.  So a round fetches 64 contiguous bytes (16
// cells) of a lookup's line with FOUR lanes: one wave-load instruction serves the lookups of 16 owner
// lanes (lane l loads chunk l&3 of owner 16k + l/4), four such instructions serve all 64 owners and are
// in flight together (one line per owner, as before).  91 % of the lookups at load 0.7 end in their
// first round (62 % with 4 cells), the line is fetched once, and the L1 sees 1.1 line-visits per lookup
// instead of 2.4.  Owners keep their state in registers; addresses and keys travel by ds_bpermute, the
// verdict of a quad comes back through a ballot.
// WIDE: tables of 2^32 - 256 cells and more.  Cell positions are 64-bit in the owner; what travels to the
// loading lanes is the low dword, and the high dword rides in the spare bits of the round's meta word.
template <bool PROF, bool WIDE, class WL>
__device__ __forceinline__ void probe_queue_quad(KArgsP ap, WL &S, const int lane, const uint32_t par,
                                                 const uint32_t qn, LaneLookup &lk, const bool count_lookups,
                                                 const uint64_t carry_pack, uint64_t (&prof)[12], uint64_t &tprev) {
    constexpr bool STD = true;
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type Pos;
    ap = launder(ap);
    const uint32_t vbits = ap->db.value_bits;
    const uint32_t kbits = 32 - vbits;
    const uint32_t vmask = ap->db.vmask;
    const uint64_t cap = ap->db.capacity;
    const uint64_t magic = ap->db.cap_magic;
    const uint32_t max_rounds = ap->db.max_chunks;
    const uint32_t *const table = ap->db.table;
    const uint64_t copy_stride = ap->db.copy_stride;
    const uint32_t copy_shift = ap->db.copy_shift;

    // ---- dense hash pass -----------------------------------------------------------------------------
    for (uint32_t r0 = 0; r0 < qn; r0 += 64) {
        const uint32_t r = r0 + lane;
        const bool act = r < qn;
        const uint64_t hc = fmix64(S.q[par][act ? r : 0u]);
        const uint64_t home = mod_capacity(hc, cap, magic);
        const uint32_t compacted = (uint32_t)(hc >> (32 + vbits));
        // (entry formats as in probe_queue: home | key << 32, or home << key_bits | key for wide tables)
        if (act) S.q[par][r] = WIDE ? ((home << kbits) | compacted) : (((uint64_t)(compacted << vbits) << 32) | (uint32_t)home);
    }
    if (lane == 0 && count_lookups) S.acc[CNT_LOOKUPS] += qn - carry_entries(carry_pack);
    wave_sync();
    NH_STAMP(4);

    uint32_t qhead = 0;  // next queue entry to hand out (uniform)
    uint32_t busy = lk.busy, r = lk.r, ckey = lk.ckey, budget = lk.budget;
    Pos pos = (Pos)lk.pos;
    // LPO lanes fetch an owner's round together (4: a quad, 64 contiguous bytes = 16 cells; 2: a pair, 8 cells):
    // an instruction serves OPI = 64 / LPO owners, LPO instructions serve all 64
    constexpr uint32_t LPO = NH_LPO, OPI = 64 / LPO, RCELLS = 4 * LPO;
    const uint32_t q4 = ((uint32_t)lane % LPO) * 4u;         // first cell of the chunk this lane loads
    const uint32_t own_sub = (uint32_t)lane / LPO;           // which of an instruction's OPI owners it loads for
    for (;;) {
        if (qhead < qn) {
            const uint64_t idle_mask = __ballot(busy == 0);
            if (idle_mask) {
                const uint32_t my = qhead + below(idle_mask);
                if (busy == 0 && my < qn) {
                    const uint64_t e = S.q[par][my];
                    if (WIDE) {
                        pos = (Pos)(e >> kbits);
                        ckey = (uint32_t)(e & ((1ull << kbits) - 1)) << vbits;
                    } else {
                        pos = (Pos)(uint32_t)e;
                        ckey = (uint32_t)(e >> 32);
                    }
                    r = my | (par << 9) | (pick_copy((uint32_t)pos, copy_shift) << 10);
                    budget = max_rounds;
                    busy = 1;
                }
                const uint32_t taken = __popcll(idle_mask);
                qhead = qhead + taken < qn ? qhead + taken : qn;
            }
        }
        // done when everything is handed out and no lane still works for the previous group
        if (qhead >= qn && __ballot(busy != 0 && ((r >> 9) & 1u) != par) == 0) break;
        if (PROF) prof[11] += 1;
        // owner side: cells of this round = from pos to the end of its line / of the table, 16 at most
        const uint32_t cj = (r >> 10) & 7u;
        uint32_t nv = 0;
        if (busy) {
            const uint32_t in_line = 32u - (((uint32_t)pos - (cj << copy_shift)) & 31u);
            const Pos room = (Pos)cap - pos;
            nv = room < (Pos)in_line ? (uint32_t)room : in_line;
            nv = nv < RCELLS ? nv : RCELLS;
        }
        const uint32_t meta = nv | (cj << 8) | (WIDE ? (uint32_t)((uint64_t)pos >> 32) << 11 : 0u);
        uint4 c[LPO];
        uint32_t ck[LPO], nvk[LPO];
#pragma unroll
        for (int k = 0; k < (int)LPO; k++) {  // instruction k: the lookups of owner lanes OPI k .. OPI k + OPI - 1, LPO lanes each
            const int src = (int)(4u * (OPI * (uint32_t)k + own_sub));
            const uint32_t p = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)pos);
            ck[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)ckey);
            const uint32_t m = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)meta);
            nvk[k] = m & 0xFFu;
            c[k] = make_uint4(0, 0, 0, 0);
            if (q4 < nvk[k]) {  // (a chunk without eligible cells is not loaded: every line-visit costs the L1)
                const uint64_t cell = WIDE ? (((uint64_t)(m >> 11) << 32) | p) : (uint64_t)p;
                const uint32_t copy = WIDE ? ((m >> 8) & 7u) : (m >> 8);
                c[k] = probe_load16(table + (uint64_t)copy * copy_stride + cell + q4);
            }
        }
        bool found = false;
        uint32_t val = 0;
#pragma unroll
        for (int k = 0; k < (int)LPO; k++) {
            uint32_t res = 0, resj = 64;
            scan4(c[k], ck[k], vmask, 0u, res, resj);
            const bool hit = q4 + resj < nvk[k];  // (resj = 64: no stopping cell in this chunk)
            const uint64_t hm = __ballot(hit);
            // the owner reads its group's bits: lowest set bit = first chunk with an eligible stopping cell
            const uint32_t nib = (uint32_t)(hm >> (LPO * ((uint32_t)lane % OPI))) & ((1u << LPO) - 1u);
            const uint32_t win = LPO * ((uint32_t)lane % OPI) + (nib ? (uint32_t)__builtin_ctz(nib) : 0u);
            const uint32_t rv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * win), (int)res);
            if (((uint32_t)lane / OPI) == (uint32_t)k && nib != 0) {
                found = true;
                val = rv;
            }
        }
        if (busy) {
            const Pos np = pos + nv;
            pos = np >= (Pos)cap ? (Pos)0 : np;
            budget--;
            if (found | (budget == 0)) {
                tax_at<STD>(S, (r >> 9) & 1u, r & 0x1FFu) = (found && val <= vmask) ? val : 0u;
                busy = 0;
            }
        }
    }
    lk.busy = busy;
    lk.r = r;
    lk.ckey = ckey;
    lk.budget = budget;
    lk.pos = pos;
    wave_sync();
    NH_STAMP(5);
}

// The (taxon, count) list of the fragment being post-processed.  The hot kernel keeps 64 entries in
// the wave's LDS slice (enough whenever the taxonomy has <= 64 nodes, e.g. every human-only
// database); the BIG kernel variant re-runs the few fragments that overflowed it with 2048 entries.
constexpr uint32_t BIG_LIST_CAP = 2048;
constexpr uint32_t CALL_OVERFLOW = 0xFFFFFFFFu;  // result.call of a fragment left to the BIG variant
struct TaxList {
    uint32_t *tax, *cnt, *score;  // score: BIG only
    uint32_t cap;
};

// Adds `cnt` hits of taxon T to the (taxon, count) list of the fragment being post-processed.
template <bool BIG, class WL>
__device__ __forceinline__ void list_add(WL &S, const TaxList &TLI, const int lane, FragState &st, const uint32_t T,
                                         const uint32_t cnt) {
    if constexpr (!BIG) {  // hot variant: at most 64 entries, lane i looks at entry i
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
    } else {
        bool found = false;
        for (uint32_t base = 0; base < st.nlist; base += 64) {
            const uint32_t idx = base + lane;
            const bool match = idx < st.nlist && TLI.tax[idx] == T;
            if (__ballot(match)) {
                if (match) TLI.cnt[idx] += cnt;
                found = true;
                break;
            }
        }
        if (!found) {
            if (st.nlist < TLI.cap) {
                if (lane == 0) {
                    TLI.tax[st.nlist] = T;
                    TLI.cnt[st.nlist] = cnt;
                }
                st.nlist++;
            } else {
                st.overflow = true;
            }
        }
    }
    wave_sync();
}

// POST one tile: per-k-mer taxa from the probe results, hit groups, (taxon, count) list.
template <bool STD, bool BIG, bool PROF, class WL>
__device__ __forceinline__ void post_tile(WL &S, const TaxList &TLI, const int lane, const uint32_t ps,
                                          const uint32_t nqt, const uint32_t par,
                                          const uint32_t qbase, const uint32_t nruns,
                                          const int last_lane, FragState &st,
                                          uint32_t *__restrict__ kmer_taxa,
                                          const uint64_t kt, uint64_t (&prof)[12], uint64_t &tprev) {
    const uint32_t qi0 = 2u * lane, qi1 = 2u * lane + 1;
    const bool v0 = ps & 1u, v1 = (ps >> 1) & 1u;
    const uint32_t r0p = (ps >> 3) & 0xFFu, r1p = r0p + ((ps >> 2) & 1u);
    // hit groups = runs of this tile that found a taxon
    uint64_t hit_any = 0;
    for (uint32_t r0 = 0; r0 < nruns; r0 += 64) {
        const uint32_t r = r0 + lane;
        const uint64_t hm = __ballot(r < nruns && tax_at<STD>(S, par, qbase + (r & (TL - 1))) != 0);
        hit_any |= hm;
        st.hit_groups += __popcll(hm);
    }
    uint32_t t0 = 0, t1 = 0;
    const bool any_hit = (hit_any != 0) | (st.carry_tax != 0);
    if (any_hit || kmer_taxa) {
        if (v0) t0 = r0p ? tax_at<STD>(S, par, qbase + r0p - 1) : st.carry_tax;
        if (v1) t1 = r1p ? tax_at<STD>(S, par, qbase + r1p - 1) : st.carry_tax;
    }
    if (kmer_taxa) {
        if (qi0 < nqt) kmer_taxa[kt + qi0] = v0 ? t0 : TAXON_AMBIGUOUS;
        if (qi1 < nqt) kmer_taxa[kt + qi1] = v1 ? t1 : TAXON_AMBIGUOUS;
    }
    if (last_lane >= 0)
        st.carry_tax = __builtin_amdgcn_readlane((last_lane & 1) ? t1 : t0, last_lane >> 1);
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
            list_add<BIG>(S, TLI, lane, st, T, cnt);
        }
    }
    NH_STAMP(6);
}

// ResolveTree (A.5) for lists of any length (BIG variant): lane i owns entries i, i+64, ...
__device__ __forceinline__ uint32_t resolve_tree_big(KArgsP ap, const TaxList &TLI, const int lane,
                                                     const FragState &st, const uint32_t total_kmers,
                                                     uint32_t &clade_hits) {
    ap = launder(ap);
    const uint32_t n = st.nlist;
    const uint32_t *parent = ap->db.parent;
    const double confidence = ap->confidence;
    const uint32_t min_hit_groups = ap->db.min_hit_groups;
    uint32_t top = 0;
    for (uint32_t i = lane; i < n; i += 64) {  // LTR score of every entry
        const uint32_t t = TLI.tax[i];
        uint32_t score = 0;
        for (uint32_t j = 0; j < n; j++)
            if (is_a_ancestor_of_b(parent, TLI.tax[j], t)) score += TLI.cnt[j];
        TLI.score[i] = score;
        top = score > top ? score : top;
    }
    top = wave_max(top);
    uint32_t call = 0;  // LCA of all entries that reach the top score (order independent)
    for (uint32_t i = lane; i < n; i += 64)
        if (TLI.score[i] == top) call = lowest_common_ancestor(parent, call, TLI.tax[i]);
    for (int d = 32; d >= 1; d >>= 1)
        call = lowest_common_ancestor(parent, call, (uint32_t)__shfl_xor((int)call, d, 64));
    auto sum_if = [&](bool clade) {
        uint32_t s = 0;
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t t = TLI.tax[i];
            if (clade ? is_a_ancestor_of_b(parent, call, t) : (t == call)) s += TLI.cnt[i];
        }
        return wave_sum(s);
    };
    const uint32_t required = (uint32_t)ceil(confidence * (double)total_kmers);
    uint32_t s = sum_if(false);  // hits exactly at the call
    while (call && s < required) {
        s = sum_if(true);
        if (s >= required) break;
        call = parent[call];
    }
    if (call && st.hit_groups < min_hit_groups) call = 0;
    clade_hits = call ? sum_if(true) : 0;
    return call;
}

// ResolveTree (A.5) on the wave: lane i owns list entry i.  Returns the call; sets clade_hits.
template <bool STD, class WL>
__device__ __forceinline__ uint32_t resolve_tree(KArgsP ap, WL &S, const int lane,
                                                 const FragState &st, const uint32_t total_kmers,
                                                 uint32_t &clade_hits) {
    ap = launder(ap);
    const uint32_t nlist = st.nlist;
    const uint32_t *parent = ap->db.parent;
    const double confidence = ap->confidence;
    const uint32_t min_hit_groups = ap->db.min_hit_groups;
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
    if (call && st.hit_groups < min_hit_groups) call = 0;
    clade_hits = 0;
    if (call) clade_hits = wave_sum((own && is_a_ancestor_of_b(parent, call, my_t)) ? my_c : 0u);
    return call;
}

// one-time LDS init of a wave: zero pads of the packed streams, sentinel tail of the candidate array
template <bool STD, class WL>
__device__ __forceinline__ void init_wave_lds(WL &S, const int lane) {
    if (lane == 0) S.stage_n = 0;
    for (int i = lane; i < (int)(sizeof(S.pk) / 4); i += 64) {
        (&S.pk[0][0])[i] = 0;
        (&S.pa[0][0])[i] = 0;
    }
    for (int i = lane; i < CandPad<STD>::value; i += 64) S.cand[TL + i] = NH_FULL;
    wave_sync();
}

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// The fragments claim `i` of the launch's work counter stands for (Sched in nh_device.h): large chunks for
// the body of the launch, smaller ones for its tail.  False when the launch is handed out.  Everything
// here is wave-uniform (i comes from a readlane).
__device__ __forceinline__ bool claim_range(KArgsP ap, const uint64_t i, const uint64_t n_frag, uint64_t &beg,
                                            uint32_t &n) {
    ap = launder(ap);
    if (i >= ap->sched.total) return false;
    uint32_t c;
    if (i < ap->sched.n0) {
        c = ap->sched.c0;
        beg = i * c;
    } else if (i < ap->sched.n01) {
        c = ap->sched.c1;
        beg = ap->sched.base1 + (i - ap->sched.n0) * c;
    } else {
        c = ap->sched.c2;
        beg = ap->sched.base2 + (i - ap->sched.n01) * c;
    }
    n = beg + c <= n_frag ? c : (uint32_t)(n_frag - beg);
    return true;
}

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & (TAIL_SHARDS - 1);
}

// Claims beyond the static first ones (one per wave of the grid) come from the work counters (nh_device.h):
// word 0 deals the body, n_static + [0, ...) up to the first tail claim; words 1 .. TAIL_SHARDS deal the tail,
// claim tail0 + shard + TAIL_SHARDS * c.  draw_raw() bumps the counter the wave currently draws from (lane 0; the
// value may be looked at much later); settle() turns it into a claim index -- moving from the body to the wave's
// own tail shard, then on to the next shards as they run empty -- or ~0 when everything is handed out.
struct Draw {
    unsigned long long raw;  // lane 0: what the atomic returned
    uint32_t word;           // wave-uniform: 0 = body, 1 + shard = a tail shard
};
__device__ __forceinline__ void draw_raw(KArgsP ap, const int lane, Draw &d) {
    if (lane == 0) d.raw = atomicAdd(launder(ap)->work + (size_t)d.word * WORK_STRIDE, 1ull);
}
// tail0 = index of the first tail claim (= total_claims when the launch has no guided tail)
__device__ __forceinline__ uint64_t settle(KArgsP ap, const int lane, Draw &d, const uint64_t tail0,
                                           const uint64_t total_claims) {
    const uint64_t n_static = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    const uint64_t t0 = tail0 > n_static ? tail0 : n_static;  // (static claims may reach into the tail of a small launch)
    uint32_t tried = 0;
    for (;;) {
        const uint64_t c = readlane64(d.raw, 0);
        if (d.word == 0) {
            const uint64_t claim = n_static + c;
            if (claim < t0) return claim;
            if (t0 >= total_claims) return ~0ull;
            d.word = 1 + xcc_id();  // the body is handed out: on to the tail, own shard first
        } else {
            const uint64_t claim = t0 + (d.word - 1) + (uint64_t)TAIL_SHARDS * c;
            if (claim < total_claims) return claim;
            if (++tried == TAIL_SHARDS) return ~0ull;
            d.word = 1 + (d.word & (TAIL_SHARDS - 1));  // next shard (1 + ((word - 1 + 1) mod TAIL_SHARDS))
        }
        draw_raw(ap, lane, d);
    }
}

// the four counters of a wave go to the row of its workgroup (folded into the caller's behind the launch)
__device__ __forceinline__ void add_counters(KArgsP ap, const unsigned long long *acc) {
#ifdef NH_COUNTER_DIRECT  // tuning builds: the round-2 way, four atomics per wave on the caller's own words
    if (unsigned long long *const c = launder(ap)->counters) {
        atomicAdd(&c[CNT_FRAGMENTS], acc[CNT_FRAGMENTS]);
        atomicAdd(&c[CNT_CLASSIFIED], acc[CNT_CLASSIFIED]);
        atomicAdd(&c[CNT_BASES], acc[CNT_BASES]);
        atomicAdd(&c[CNT_LOOKUPS], acc[CNT_LOOKUPS]);
    }
    return;
#endif
    unsigned long long *const row = launder(ap)->cshard + (size_t)(blockIdx.x & (COUNTER_SHARDS - 1)) * COUNTER_STRIDE;
    atomicAdd(&row[CNT_FRAGMENTS], acc[CNT_FRAGMENTS]);
    atomicAdd(&row[CNT_CLASSIFIED], acc[CNT_CLASSIFIED]);
    atomicAdd(&row[CNT_BASES], acc[CNT_BASES]);
    atomicAdd(&row[CNT_LOOKUPS], acc[CNT_LOOKUPS]);
}

// Stores the records staged by the last post_group (lane i = record i).  Called right before a probe
// phase: the stores complete in the shadow of the first probe round trip.
template <bool STD, class WL>
__device__ __forceinline__ void flush_records(KArgsP ap, WL &S, const int lane) {
    const uint32_t n = uni(S.stage_n);
    if (n == 0) return;
    if ((uint32_t)lane < n) {
        const uint64_t f = S.stage_f[lane];
        *reinterpret_cast<uint4 *>(&launder(ap)->out[f]) = S.stage_rec[lane];
    }
    wave_sync();
    if (lane == 0) S.stage_n = 0;
}

// POST a group: finish the tiles of group `pp` (and each fragment whose last tile is among them).
// The accumulation state of the fragment being post-processed is parked in LDS between calls (it is
// wave-uniform and idle during scan and probe: four scalar registers less to keep there).
// SPLIT: the tiles may be segments of split long reads (generic kernel); compiled out of the short-read kernel
template <bool STD, bool BIG, bool PROF, bool SPLIT, class WL>
__device__ __forceinline__ void post_group(KArgsP ap, WL &S, const TaxList &TLI, const int lane,
                                           const int mates, const bool reset_per_mate, const uint32_t pp,
                                           const uint32_t nslot, uint64_t (&prof)[12], uint64_t &tprev) {
        KArgsP a2 = launder(ap);
        FragState st;
        {
            const uint4 fs = S.frag_state;
            st.nlist = uni(fs.x);
            st.hit_groups = uni(fs.y);
            st.carry_tax = uni(fs.z);
            st.overflow = uni(fs.w) != 0;
            st.carry_min = 0;  // (the scan keeps its own last-minimizer)
        }
        uint32_t *const kmer_taxa = a2->kmer_taxa;
        for (uint32_t s = 0; s < nslot; s++) {
            // descriptor: three 16-byte LDS reads (same address in every lane), then scalars
            const uint4 *dp = reinterpret_cast<const uint4 *>(&S.slot[pp][s]);
            const uint4 d0 = dp[0], d1 = dp[1], d2 = dp[2];
            const uint32_t d_last = uni(d2.x), flags = uni(d2.y);
            // f_lo, f_hi (a segment of a split read keeps the number of segments in f_hi: such launches have
            // fewer than 2^32 fragments)
            const uint64_t f = ((SPLIT && (flags & 16u)) ? 0ull : ((uint64_t)uni(d0.y) << 32)) | uni(d0.x);
            const uint64_t kt = kmer_taxa ? a2->kmer_taxa_off[f] + uni(d0.z) : 0;  // tile's first k-mer
            const uint32_t d_nqt = uni(d1.x), d_qbase = uni(d1.y), d_nruns = uni(d1.z);
            const uint32_t d_nk0 = uni(d2.z), d_total = uni(d2.w);
            if (flags & 4u) {  // first tile of its fragment (or segment): fresh accumulation state
                st.nlist = 0;
                st.hit_groups = 0;
                st.carry_tax = 0;
                st.overflow = false;
                // a segment of a split read inherits kraken2's last_taxon: the taxon of the minimizer it
                // inherited, looked up as entry pad0 - 1 of this group's queue
                if constexpr (SPLIT) {
                    const uint32_t cq = uni(d0.w);
                    if (cq) st.carry_tax = uni(tax_at<STD>(S, pp, cq - 1u));
                }
            }
            const uint32_t ps = S.ps[pp][s][lane];
            post_tile<STD, BIG, PROF>(S, TLI, lane, ps, d_nqt, pp, d_qbase, d_nruns, (int)d_last, st,
                                      kmer_taxa, kt, prof, tprev);
            if ((flags & 2u) && reset_per_mate) st.carry_tax = 0;  // mate 0 ended, mate 1 follows
            bool finish = (flags & 1u) != 0;  // fragment ended
            if (SPLIT && finish && (flags & 16u)) {
                // ... or rather one SEGMENT of a split read (descriptor: f_hi = segments, nk0 = this segment,
                // pad1 = first partial slot of the read).  Leave the partial; whoever finishes last adds up.
                const uint32_t nseg = uni(d0.y), seg = d_nk0, sb = uni(d1.w);
                KArgsP a4 = launder(ap);
                const bool over = st.overflow || st.nlist > PART_CAP;
                uint32_t v = 0;
                if (lane == 0) v = st.hit_groups;
                if (lane == 1) v = over ? 0u : st.nlist;
                if (lane == 2) v = over ? 1u : 0u;
                if (lane >= 4 && !over) {
                    const uint32_t idx = ((uint32_t)lane - 4u) >> 1;
                    if (idx < st.nlist) {
                        if constexpr (BIG) v = (lane & 1) ? TLI.cnt[idx] : TLI.tax[idx];
                        else v = (lane & 1) ? S.list_cnt[idx] : S.list_tax[idx];
                    }
                }
                // publish: the partial must be in memory before the counter says so -- another CU, possibly on
                // another XCD, reads it.  Write-through (sc1) stores, drained, then the agent-scope atomic; the reader
                // uses sc1 loads (past its L1; its XCD's L2 has never seen these lines).  NOT an agent-scope release /
                // acquire pair: a release writes back the XCD's whole L2 and an acquire empties the CU's L1, once per
                // segment on every wave of the chip -- that halved the long-read throughput when it was tried
                // (MI355X_MICROARCH.md, inter-workgroup visibility: the "sc1 payload, vmcnt(0), flag" form).
                __hip_atomic_store(&a4->split.part[(uint64_t)(sb + seg) * PART_DWORDS + (uint32_t)lane], v, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                uint32_t old = 0;
                if (lane == 0) old = atomicAdd(&a4->split.part_done[sb], 1u);
                old = uni(old);
                finish = old + 1u == nseg;
                // The partial loads below must not move above the counter's atomic: relaxed operations
                // carry no order of their own, the control dependency on `finish` is all the compiler is held by.  A
                // compiler-only fence pins the program order; the hardware side (in-order issue from one wave, sc1
                // loads that bypass the L1, lines this XCD's L2 has never held) is the gfx942 / gfx950 behaviour this
                // file is built for -- the Makefile's ARCH list is the assumption.
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                if (finish) {
                    for (uint32_t s2 = 0; s2 < nseg; s2++) {
                        if (s2 == seg) continue;
                        const uint32_t pv = __hip_atomic_load(&a4->split.part[(uint64_t)(sb + s2) * PART_DWORDS + (uint32_t)lane],
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        st.hit_groups += __builtin_amdgcn_readlane(pv, 0);
                        const uint32_t n2 = __builtin_amdgcn_readlane(pv, 1);
                        if (__builtin_amdgcn_readlane(pv, 2)) st.overflow = true;
                        for (uint32_t i = 0; i < n2; i++)
                            list_add<BIG>(S, TLI, lane, st, __builtin_amdgcn_readlane(pv, 4 + 2 * (int)i),
                                          __builtin_amdgcn_readlane(pv, 5 + 2 * (int)i));
                    }
                    if (over) st.overflow = true;
                }
            }
            if (finish) {
                const uint32_t total_kmers = d_total;
                uint32_t call = 0, clade_hits = 0;
                if (st.nlist > 0) {
                    if (BIG) {
                        call = resolve_tree_big(ap, TLI, lane, st, total_kmers, clade_hits);
                    } else if (st.nlist == 1) {
                        // one taxon only (nearly every fragment on a human-only database): its score is
                        // its count and climbing to its ancestors cannot add hits, so ResolveTree is a
                        // comparison -- no walk up the parent chain (a dependent global load per level)
                        const uint32_t cnt = S.list_cnt[0];
                        const uint32_t required = (uint32_t)ceil(a2->confidence * (double)total_kmers);
                        call = cnt >= required ? S.list_tax[0] : 0u;
                        if (call && st.hit_groups < a2->db.min_hit_groups) call = 0;
                        clade_hits = call ? cnt : 0u;
                    } else {
                        call = resolve_tree<STD>(ap, S, lane, st, total_kmers, clade_hits);
                    }
                }
                // hot variant: a fragment with more than 64 distinct taxa is left to the BIG variant
                const bool defer = !BIG && st.overflow;
                if (defer) call = 0;
                if (lane == 0) {
                    uint4 rec;
                    rec.x = defer ? CALL_OVERFLOW : call;
                    rec.y = total_kmers;
                    rec.z = clade_hits;
                    rec.w = st.hit_groups;
                    // staged: a store issued here would still be in flight when the next scan waits for
                    // its bases (vmcnt counts stores too) -- a whole memory round trip per group
                    const uint32_t sn = S.stage_n;
                    S.stage_rec[sn] = rec;
                    S.stage_f[sn] = f;
                    S.stage_n = sn + 1;
                    if (kmer_taxa && mates == 2)
                        kmer_taxa[a2->kmer_taxa_off[f] + d_nk0] = TAXON_MATE_BORDER;
                    if (defer) atomicMax(&a2->pending[0], 1);               // work for the BIG variant
                    if (BIG && st.overflow) atomicOr(&a2->error_flag[0], 1);  // beyond 2048 too: error
                }
                if (lane == 0 && call) S.acc[CNT_CLASSIFIED] += 1;
            }
            wave_sync();
        }
        if (lane == 0) S.frag_state = make_uint4(st.nlist, st.hit_groups, st.carry_tax, st.overflow ? 1u : 0u);
        NH_STAMP(7);
    }

constexpr uint32_t PREF_LANES = 42;  // dwords a tile can need: (3 + 128 + 30 + 3) / 4 <= 41
constexpr uint32_t SHORT_MAX = 158;  // TQ + K - 1 bases: at most one tile of 124 k-mers (k_classify_short)

// 64 sequences spread evenly over the launch (lane i: sequence i * n / 64): is every one of them longer than a tile of the
// short-read kernel?  Then that kernel has nothing to do but mark every chunk for the generic one -- 25 000 claims on one
// counter, 0.35-0.41 ms of a 2.8 ms launch of 2 x 250 bp pairs (profiles/r04_pe250_summary.txt) -- so it returns at once,
// and the generic kernel, which takes the same sample, classifies every chunk instead of the marked ones.  Only a
// scheduling decision: the generic kernel classifies sequences of any length, whatever the sample missed.
__device__ __forceinline__ bool sample_all_long(KArgsP ap, const int lane) {
    KArgsP a = launder(ap);
    const uint64_t ns = a->n_frag * (uint64_t)a->mates;
    const uint64_t i = ns * (uint64_t)lane / 64u;
    const uint32_t len = a->seq_len != nullptr ? a->seq_len[i] : (uint32_t)(a->seq_off[i + 1] - a->seq_off[i]);
    return __ballot(len <= SHORT_MAX) == 0;
}

template <bool LINEAR, bool STD, bool CAP32, bool PROF, bool BIG>
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK, BIG ? 1 : (STD ? NH_MIN_WAVES : 3)) void k_classify(const KArgs args_by_kernarg_pointer) {
    KArgsP ap = (KArgsP)__builtin_amdgcn_kernarg_segment_ptr();
    // BIG variant: only runs when some fragment overflowed the 64-entry list of the hot variant
    if (BIG && ap->pending[0] == 0) return;
    // after the short-read kernel: only the chunks it left behind (none: nothing to do)
    bool only_deferred = ap->only_deferred != 0;
    if (only_deferred && ap->pending_long[0] == 0) {
        // nothing marked: nothing to do -- unless the short-read kernel left EVERYTHING to this one (sample_all_long)
        if (BIG || !sample_all_long(ap, (int)(threadIdx.x & 63))) return;
        only_deferred = false;
    }
    typedef WaveLdsT<STD, 1, QCAP_GENERIC> WL;
    __shared__ WL lds_all[WAVES_PER_BLOCK];
    __shared__ uint32_t big_lists[BIG ? WAVES_PER_BLOCK * 3 * BIG_LIST_CAP : 1];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WL &S = lds_all[wib];
    TaxList TLI = {nullptr, nullptr, nullptr, 0};  // used by the BIG variant only
    if constexpr (BIG) {
        TLI.tax = &big_lists[(wib * 3 + 0) * BIG_LIST_CAP];
        TLI.cnt = &big_lists[(wib * 3 + 1) * BIG_LIST_CAP];
        TLI.score = &big_lists[(wib * 3 + 2) * BIG_LIST_CAP];
        TLI.cap = BIG_LIST_CAP;
    }

    init_wave_lds<STD>(S, lane);

    const uint32_t K = STD ? 35u : ap->db.k;
    const uint32_t L = STD ? 31u : ap->db.l;
    const uint32_t TQ = TL - (STD ? 4u : ap->db.window);  // k-mers per tile
    const int mates = ap->mates;
    const uint64_t n_frag = ap->n_frag;
    const bool reset_per_mate = ap->db.reset_per_mate != 0;
    // dword index of the last dword the caller guarantees readable (8 bytes of slack, see ABI),
    // parked in LDS: it is needed once per tile, not worth two SGPRs for the whole kernel
    const bool inplace = ap->seq_len != nullptr;
    if (lane == 0) S.last_dw = ((inplace ? ap->bases_end : ap->seq_off[n_frag * (uint64_t)mates]) + 4) >> 2;
    const uint32_t pl = (uint32_t)lane < PREF_LANES ? (uint32_t)lane : PREF_LANES - 1;

    // the dword stream of the tile that starts at byte g0: 4 bases per lane, coalesced
    auto tile_ptr = [&](uint64_t g0) -> const uint32_t * {
        KArgsP a3 = launder(ap);
        const uint64_t last_dw = S.last_dw;
        uint64_t dw = (g0 >> 2) + pl;
        dw = dw < last_dw ? dw : last_dw;
        return reinterpret_cast<const uint32_t *>(a3->bases) + dw;
    };

    if (lane < 4) S.acc[lane] = 0;
    bool bad_input = false;
    uint64_t prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tprev = PROF ? __builtin_readcyclecounter() : 0;

    // speculative prefetch of the bases of the next NH_PF_DEPTH tiles (byte offset tags, loaded dwords): a FIFO, slot 0 = the
    // next tile.  Depth 2 until round 5; a group is up to four tiles scanned back to back, so only a load issued four scans ahead
    // has a whole probe phase to arrive (what k_classify_short does with its batches): profiles/r06_prefetch_depth.txt.
#ifndef NH_PF_DEPTH
#define NH_PF_DEPTH 2
#endif
    constexpr int PFD = NH_PF_DEPTH;
    uint32_t tagf[PFD], wf[PFD];  // 1 + byte offset relative to the chunk's first base; 0 = nothing loaded
#pragma unroll
    for (int i = 0; i < PFD; i++) tagf[i] = 0, wf[i] = 0;

    // Two groups of tiles are in flight: the one being scanned / probed (parity `par`) and the
    // previous one, which is post-processed only after the probe phase of its successor.
    uint32_t par = 0;
    uint32_t nslot_new = 0;     // tiles scanned into the current group (descriptors in S.slot[par])
    uint32_t nslot_old = 0;     // tiles of the previous group still to be post-processed
    uint32_t qn = 0;            // queue entries of the current group
    LaneLookup lk;
    lk.busy = 0;
    lk.r = 0;
    lk.pos = lk.first_pos = lk.step = 0;
    lk.ckey = 0;
    lk.budget = 0;

    if (lane == 0) S.frag_state = make_uint4(0, 0, 0, 0);
    uint64_t carry_pack = 0;  // queue entries of the current group that are look-ups of an inherited minimizer
    // group complete (or input exhausted): hash + probe it -- which also resolves what is left of
    // the previous group -- then post-process the previous group and switch buffers
    auto turn = [&]() {
        flush_records<STD>(ap, S, lane);
#ifndef NH_NO_QUAD
        if constexpr (LINEAR && STD)
            probe_queue_quad<PROF, !CAP32>(ap, S, lane, par, qn, lk, !BIG, carry_pack, prof, tprev);
        else
#endif
            probe_queue<LINEAR, STD, CAP32, PROF>(ap, S, lane, par, qn, lk, !BIG, carry_pack, prof, tprev);
        carry_pack = 0;
        if (nslot_old) post_group<STD, BIG, PROF, true>(ap, S, TLI, lane, mates, reset_per_mate, par ^ 1u, nslot_old, prof, tprev);
        nslot_old = nslot_new;
        par ^= 1u;
        nslot_new = 0;
        qn = 0;
    };

    // Fragments are handed out dynamically: every wave pulls chunks of consecutive fragments from
    // one counter (claimed one chunk ahead), so late-starting (non-resident) workgroups of the grid
    // find no work instead of a static share.
    // (mates * sched.c0 + 1 <= 64: the offsets of a chunk fit the lanes of off_v)
    // Long-read launches (items != 0) hand out ITEMS instead: whole reads or segments of a split read,
    // largest first (SplitBufs in nh_device.h; the BIG variant always goes by fragments).
    const bool items = !BIG && ap->split.hdr != nullptr;
    // a wave's first claim is its index in the grid, later ones come from the work counter (k_classify_short
    // says why)
    Draw dr;
    dr.raw = 0;
    dr.word = 0;
    bool first_claim = true;
    auto take_claim = [&]() { draw_raw(ap, lane, dr); };
    // claims there are in all: items of a long-read launch (known on the device only), else the claim map
    uint64_t total_claims = launder(ap)->sched.total;
    uint64_t tail0 = launder(ap)->sched.n0;  // (first claim of a guided tail; == total without one)
    if (items) {
        KArgsP a5 = launder(ap);
        const SplitHdr *const hdr = a5->split.hdr;
        const uint32_t used = hdr->seg_used;
        total_claims = (uint64_t)(used < a5->split.seg_cap ? used : a5->split.seg_cap) + hdr->n_mid + hdr->n_small;
        tail0 = total_claims;
    }
    for (;;) {
        uint64_t claim;
        if (first_claim) {
            claim = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + (uint32_t)wib;
            first_claim = false;
            if (claim >= total_claims) claim = ~0ull;
        } else {
            claim = settle(ap, lane, dr, tail0, total_claims);
        }
        if (claim == ~0ull) break;
        uint64_t cbeg;
        uint32_t ncf;
        uint32_t seg = 0, nseg = 1, seg_slot = 0;  // the claimed segment of a split read (items only)
        if (items) {
            KArgsP a5 = launder(ap);
            const SplitHdr *const hdr = a5->split.hdr;
            const uint32_t used = hdr->seg_used, n_mid = hdr->n_mid;
            const uint32_t n_multi = used < a5->split.seg_cap ? used : a5->split.seg_cap;
            ncf = 1;
            bool hole = false;
            if (claim < n_multi) {
                const uint4 it = *reinterpret_cast<const uint4 *>(&a5->split.items_multi[claim]);
                cbeg = uni(it.x);
                seg = uni(it.y);
                nseg = uni(it.z);
                seg_slot = uni(it.w);
                hole = nseg == 0;
            } else {
                const uint64_t j = claim - n_multi;
                cbeg = uni(a5->split.items_single[j < n_mid ? j : n_frag - 1 - (j - n_mid)]);
            }
            if (hole) {  // (reserved by a read that is listed whole instead: the buffers were full)
                take_claim();
                continue;
            }
        } else if (!claim_range(ap, claim, n_frag, cbeg, ncf)) {
            break;
        }
        take_claim();
        if (only_deferred) {  // second pass: only what the short-read kernel left behind
            if (!((launder(ap)->defer_bits[claim >> 5] >> (claim & 31)) & 1u)) continue;
        }
        // all sequence offsets of the chunk with ONE coalesced load (lane i = offset i), kept
        // relative to the chunk's first byte so that everything per fragment is 32-bit
        const uint32_t nseq = ncf * (uint32_t)mates;
        const uint32_t nof = inplace ? nseq : nseq + 1;
        const uint64_t sidx = cbeg * (uint64_t)mates + ((uint32_t)lane < nof ? (uint32_t)lane : nof - 1);
        const uint64_t off64 = launder(ap)->seq_off[sidx];
        const uint64_t cbase = readlane64(off64, 0);
        const uint64_t crel = off64 - cbase;
        if (__ballot((crel >> 32) != 0)) bad_input = true;  // a chunk of 4 Gbases and more
        const uint32_t off_v = (uint32_t)crel;
        // lane i: length of sequence i of the chunk
        uint32_t len_v = inplace ? launder(ap)->seq_len[sidx] : (uint32_t)__shfl_down((int)off_v, 1, 64) - off_v;
        if ((uint32_t)lane >= nseq) len_v = 0;
        if (!BIG && seg == 0) {  // (a split read is counted by its first segment)
            const uint32_t cb = wave_sum(len_v);
            if (lane == 0) {
                S.acc[CNT_FRAGMENTS] += ncf;
                S.acc[CNT_BASES] += cb;
            }
        }
#pragma unroll
        for (int i = 0; i < PFD; i++) tagf[i] = 0;  // tags are relative to the chunk
        for (uint32_t fc = 0; fc < ncf; fc++) {
            NH_STAMP(8);
            const uint64_t f = cbeg + fc;
            const int oi = (int)fc * mates;
            const uint32_t o0 = __builtin_amdgcn_readlane(off_v, oi);
            const uint32_t o1 = __builtin_amdgcn_readlane(off_v, oi + 1);
            const uint32_t n0 = __builtin_amdgcn_readlane(len_v, oi);
            const uint32_t n1 = mates == 2 ? __builtin_amdgcn_readlane(len_v, oi + 1) : 0u;
            const uint32_t nk0 = n0 >= K ? n0 - K + 1 : 0;
            const uint32_t nk1 = n1 >= K ? n1 - K + 1 : 0;
            if (BIG) {  // only the fragments the hot variant gave up on
                const uint32_t prev_call = launder(ap)->out[f].call;
                if (uni(prev_call) != CALL_OVERFLOW) continue;
            }
            if (nk0 + nk1 == 0) {  // no k-mer at all: all-zero record, only the mate border
                if (lane == 0) {
                    KArgsP a2 = launder(ap);
                    uint4 rec = {0, 0, 0, 0};
                    *reinterpret_cast<uint4 *>(&a2->out[f]) = rec;
                    uint32_t *const kmer_taxa = a2->kmer_taxa;
                    if (kmer_taxa && mates == 2) kmer_taxa[a2->kmer_taxa_off[f]] = TAXON_MATE_BORDER;
                }
                continue;
            }
            uint64_t carry_min = NH_FULL;  // kraken2 last_minimizer of this fragment
            bool frag_first = true;
            NH_STAMP(9);
            // the k-mers this claim covers: all of them, or those of one segment of a split read (single-end)
            const bool split = nseg > 1;
            const uint32_t qbeg = split ? seg * SEG_TILES * TQ : 0u;
            uint32_t carry_q = 0;  // 1 + queue entry of the inherited minimizer's look-up (first tile of a segment)
            if (split && seg > 0) {
                // kraken2's last_minimizer at the start of the segment = the minimizer of the last unambiguous
                // k-mer before it: scan the tile that ends there (its run starts land beyond the queue's end and
                // are forgotten), further back while a tile holds none
                for (uint32_t hq = qbeg; hq > 0 && carry_min == NH_FULL;) {
                    hq -= TQ;
                    const uint64_t hg0 = cbase + o0 + hq;
                    const uint32_t *hp = tile_ptr(hg0);
                    const uint32_t hw = *hp;
                    uint32_t hps, hdummy = 0;
                    int hlast;
                    (void)scan_tile<STD, PROF>(ap, S, lane, hw, (uint32_t)hg0 & 3u, (uint32_t)TL, TQ, par, qn, carry_min, hps,
                                               hlast, hp, false, hdummy, prof, tprev);
                    wave_sync();
                }
                // ... and last_taxon = what that minimizer finds in the table: one more entry of this group's
                // queue, which must then be the group of the segment's first tile as well
                if (carry_min != NH_FULL) {
                    if (nslot_new == NSLOT || qn + TL + 1 > (uint32_t)QCAP_GENERIC) turn();
                    if (lane == 0) S.q[par][qn] = carry_min;
                    carry_q = qn + 1u;
                    carry_pack = (carry_pack << 16) | carry_q;
                    qn++;
                }
            }
            for (int m = 0; m < mates; m++) {
                const uint32_t n = m ? n1 : n0;
                const uint32_t nk_all = m ? nk1 : nk0;
                uint32_t nk = nk_all;  // end of the k-mers to scan
                if (split && qbeg + SEG_TILES * TQ < nk) nk = qbeg + SEG_TILES * TQ;
                if (m == 1 && reset_per_mate) carry_min = NH_FULL;
                for (uint32_t q0 = (m == 0 ? qbeg : 0u); q0 < nk; q0 += TQ) {
                    const uint64_t g0 = cbase + (m ? o1 : o0) + q0;
                    NH_STAMP(0);
                    // prefetch FIFO: (tagf[0], wf[0]) was loaded for the next tile, (tagf[1], wf[1]) for the one after it, ...
                    const uint32_t gtag = (uint32_t)(g0 - cbase) + 1u;
                    uint32_t w = 0;
                    {
                        bool have = false;
#pragma unroll
                        for (int i = 0; i < PFD; i++)
                            if (!have && tagf[i] == gtag) {
                                w = wf[i];
                                have = true;
                            }
                        if (!have) w = *tile_ptr(g0);
                    }
#pragma unroll
                    for (int i = 0; i + 1 < PFD; i++) tagf[i] = tagf[i + 1], wf[i] = wf[i + 1];
                    // the tile to be scanned two scans from now (a full group away, so its load has a whole probe phase
                    // to arrive): the scan order -- tiles of a sequence, then the next mate, then the next fragment of the
                    // chunk -- stepped twice from here.  (Round 5: the guess used to assume ONE tile per sequence once a
                    // sequence ended, so reads of two or three tiles -- 2 x 250 bp -- found every second tile's dword not
                    // loaded and fetched it synchronously.)  Lengths of the sequences ahead come from the lanes of len_v; a
                    // sequence without k-mers there makes the guess wrong, which costs a load, never a result: the tag decides.
                    const bool seq_end = q0 + TQ >= nk;
                    uint64_t ng0 = g0 + (uint32_t)PFD * TQ;
                    bool pf_on = true;
                    if (q0 + (uint32_t)PFD * TQ >= nk) {
                        uint32_t si = (uint32_t)oi + (uint32_t)m;  // sequence index within the chunk
                        uint32_t q2 = q0, nk2 = nk;
#pragma unroll
                        for (int hop = 0; hop < PFD; hop++) {
                            q2 += TQ;
                            if (q2 >= nk2) {
                                q2 = 0;
                                si++;
                                const uint32_t ln = __builtin_amdgcn_readlane(len_v, (int)(si < 63u ? si : 63u));
                                nk2 = ln >= K ? ln - K + 1 : 0u;
                            }
                        }
                        pf_on = si < nseq;
                        ng0 = cbase + __builtin_amdgcn_readlane(off_v, (int)(si < 63u ? si : 63u)) + q2;
                    }
                    const uint32_t *pf_ptr = tile_ptr(pf_on ? ng0 : g0);
                    tagf[PFD - 1] = pf_on ? (uint32_t)(ng0 - cbase) + 1u : 0u;

                    const uint32_t nl_left = (n - L + 1) - q0;
                    const uint32_t nlt = nl_left < (uint32_t)TL ? nl_left : (uint32_t)TL;
                    const uint32_t nq_left = nk - q0;
                    const uint32_t nqt = nq_left < TQ ? nq_left : TQ;
                    const bool frag_end = seq_end && (m == mates - 1 || nk1 == 0);
                    const bool mate_end = seq_end && m == 0 && mates == 2 && nk1 != 0;

                    uint32_t ps;
                    int last_lane;
                    const uint32_t nruns =
                        scan_tile<STD, PROF>(ap, S, lane, w, (uint32_t)g0 & 3u, nlt, nqt, par, qn,
                                             carry_min, ps, last_lane, pf_ptr, pf_on, wf[PFD - 1], prof, tprev);
                    if (lane == 0) {
                        uint4 d0, d1, d2;  // layout of SlotLds
                        d0.x = (uint32_t)f;
                        d0.y = (uint32_t)(f >> 32);
                        d0.z = (m ? nk0 + 1 : 0) + q0;  // k-mer index of the tile within its fragment
                        d0.w = frag_first ? carry_q : 0u;
                        d1.x = nqt;
                        d1.y = qn;
                        d1.z = nruns;
                        d1.w = seg_slot;
                        d2.x = (uint32_t)last_lane;
                        d2.y = (frag_end ? 1u : 0u) | (mate_end ? 2u : 0u) | (frag_first ? 4u : 0u) | (split ? 16u : 0u);
                        d2.z = split ? seg : nk0;
                        d2.w = nk0 + nk1;
                        if (split) d0.y = nseg;
                        uint4 *dp = reinterpret_cast<uint4 *>(&S.slot[par][nslot_new]);
                        dp[0] = d0;
                        dp[1] = d1;
                        dp[2] = d2;
                    }
                    S.ps[par][nslot_new][lane] = (uint16_t)ps;
                    NH_STAMP(10);
                    frag_first = false;
                    qn += nruns;
                    nslot_new++;
                    // another tile joins this group only if its run starts are sure to fit the queue
                    if (nslot_new == NSLOT || qn + TL > (uint32_t)QCAP_GENERIC) turn();
                }
            }
        }
    }
    // drain: probe what is left, post-process the last two groups
    turn();
    turn();
    flush_records<STD>(ap, S, lane);

    wave_sync();
    if (lane == 0) {
        unsigned long long *const counters = ap->counters;
        int *const error_flag = ap->error_flag;
        if (PROF && counters)
            for (int i = 0; i < 12; i++) atomicAdd(&counters[CNT_N + i], (unsigned long long)prof[i]);
        add_counters(ap, S.acc);
        if (bad_input) atomicOr(&error_flag[0], 2);
    }
}

// ---- short-read kernel --------------------------------------------------------------------------------
// The same path for the case nohuman meets most: every sequence of a chunk fits ONE tile (<= 158 bases
// at k=35/l=31: Illumina reads), default database geometry, linear probing, 32-bit cell positions.
// What that buys over the generic loop nest (fragment x mate x tile, any length):
//   * tile i of a chunk IS sequence i: geometry, descriptors and the prefetch address come straight from
//     the lanes that hold the chunk's offsets and lengths -- no iterator state in scalar registers;
//   * tiles are taken in batches of NSLOT: all of a batch are encoded first, and the moment a tile's
//     dword is consumed the load of the tile NSLOT further on is issued into the same register -- it has
//     a whole probe phase to arrive, and no scan ever waits for bases;
//   * one call site of the probe / post code (the drain of the pipeline runs through it as well).
// A chunk that holds a longer sequence is not touched: its bit is set in defer_bits and the generic
// kernel, launched right behind, classifies exactly those chunks (none: it returns at once).
// TLINE: the variant tools/timeline.py runs (per-wave timestamps, KArgs::timeline); compiled out of the others
template <bool PROF, bool WIDE, bool TLINE>
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK, PROF ? 4 : NH_MIN_WAVES) void k_classify_short(const KArgs args_by_kernarg_pointer) {
    constexpr bool STD = true;
    KArgsP ap = (KArgsP)__builtin_amdgcn_kernarg_segment_ptr();
    typedef WaveLdsT<STD, NSLOT, QCAP_SHORT> WL;
    __shared__ WL lds_all[WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WL &S = lds_all[wib];
    const TaxList TLI = {nullptr, nullptr, nullptr, 0};
    if (sample_all_long(ap, lane)) return;  // reads longer than a tile throughout: all of it is the generic kernel's
    init_wave_lds<STD>(S, lane);

    constexpr uint32_t K = 35, L = 31;
    const int mates = ap->mates;
    const bool reset_per_mate = ap->db.reset_per_mate != 0;
    // (the launch's size and the form of its input are read again where a chunk starts: two scalars less to keep)
    if (lane == 0)
        S.last_dw = ((ap->seq_len != nullptr ? ap->bases_end : ap->seq_off[ap->n_frag * (uint64_t)mates]) + 4) >> 2;
    const uint32_t pl = (uint32_t)lane < PREF_LANES ? (uint32_t)lane : PREF_LANES - 1;
    auto tile_ptr = [&](uint64_t g0) -> const uint32_t * {
        KArgsP a3 = launder(ap);
        const uint64_t last_dw = S.last_dw;
        uint64_t dw = (g0 >> 2) + pl;
        dw = dw < last_dw ? dw : last_dw;
        return reinterpret_cast<const uint32_t *>(a3->bases) + dw;
    };
    if (lane < 4) S.acc[lane] = 0;
    bool bad_input = false;
    uint64_t prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tprev = PROF ? __builtin_readcyclecounter() : 0;

    uint32_t par = 0, nslot_new = 0, nslot_old = 0, qn = 0;
    LaneLookup lk;
    lk.busy = 0;
    lk.r = 0;
    lk.pos = lk.first_pos = lk.step = 0;
    lk.ckey = 0;
    lk.budget = 0;
    if (lane == 0) S.frag_state = make_uint4(0, 0, 0, 0);

    // tuning aid (KArgs::timeline): looked up again at every stamp, so that nothing of it lives in registers
    auto tl_row = [&]() -> unsigned long long * {
        if constexpr (!TLINE) return nullptr;
        unsigned long long *const base = launder(ap)->timeline;
        return base ? base + 32ull * ((uint64_t)blockIdx.x * WAVES_PER_BLOCK + (uint32_t)wib) : nullptr;
    };
    if (TLINE && lane == 0) {
        if (unsigned long long *const tline = tl_row()) {
            tline[0] = wall_clock64();
            tline[2] = tline[3] = tline[6] = 0;
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            tline[7] = xcc;
        }
    }
    // Claims (claim_range: which fragments a claim stands for).  The FIRST claim of a wave is its own index in
    // the grid -- no atomic: 5120 waves bumping one counter at the same moment, and bumping it again for the
    // claim ahead, kept every wave of a launch waiting ~120 us for its first bases (vmcnt retires in order: the
    // offsets load sits behind the atomic) -- later ones are n_static + the work counter.  The claim ahead is
    // taken when the LAST batch of the current chunk has been encoded: early enough to be back when the chunk
    // ends, and late enough that the chunk it reserves is started soon (a claim taken a whole chunk ahead made
    // the launch end two chunks after the counter ran out, profiles/r03_timeline.txt).
    // ... and the later ones come from the work counters (Draw above).
    Draw dr;
    dr.raw = 0;
    dr.word = 0;
    bool first_claim = true;
    bool claim_ahead = true;  // a claim has been taken that has not been looked at yet
    auto take_claim = [&]() {
        draw_raw(ap, lane, dr);
        claim_ahead = true;
    };

    // the chunk being worked on
    uint64_t cbeg = 0, cbase = 0;
    uint32_t nseq = 0, t = 0;   // sequences of the chunk, first sequence of the next batch
    uint32_t off_v = 0, len_v = 0;  // lane i: offset (relative to cbase) and length of sequence i
    uint32_t w[NSLOT];          // dword streams of the next batch's tiles, loaded one batch ahead
#pragma unroll
    for (int j = 0; j < NSLOT; j++) w[j] = 0;
    uint64_t carry_min = NH_FULL;
    uint32_t drain = 0;         // input exhausted: two more turns empty the group pipeline
    for (;;) {
        bool batch = true;
        if (t >= nseq) {  // next chunk
            NH_STAMP(8);
            uint64_t claim = ~0ull;
            if (first_claim) {
                claim = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + (uint32_t)wib;
                first_claim = false;
            } else if (drain == 0) {
                claim = settle(ap, lane, dr, launder(ap)->sched.n0, launder(ap)->sched.total);
            }
            uint64_t c0 = 0;
            uint32_t ncf = 0;
            if (claim == ~0ull || !claim_range(ap, claim, launder(ap)->n_frag, c0, ncf)) {
                if (drain == 2) break;
                drain++;
                batch = false;
            } else {
                claim_ahead = false;
                if (TLINE && lane == 0) {
                    if (unsigned long long *const tline = tl_row()) {
                        tline[4] = wall_clock64();
                        if (tline[6] < 24) tline[8 + tline[6]] = (tline[4] << 8) | ncf;  // start and size of every chunk
                        tline[6] += 1;
                    }
                }
                cbeg = c0;
                const uint32_t ns = ncf * (uint32_t)mates;
                const bool inplace = launder(ap)->seq_len != nullptr;
                const uint32_t nof = inplace ? ns : ns + 1;
                const uint64_t sidx = cbeg * (uint64_t)mates + ((uint32_t)lane < nof ? (uint32_t)lane : nof - 1);
                const uint64_t off64 = launder(ap)->seq_off[sidx];
                cbase = readlane64(off64, 0);
                const uint64_t crel = off64 - cbase;
                if (__ballot((crel >> 32) != 0)) bad_input = true;
                off_v = (uint32_t)crel;
                len_v = inplace ? launder(ap)->seq_len[sidx] : (uint32_t)__shfl_down((int)off_v, 1, 64) - off_v;
                if ((uint32_t)lane >= ns) len_v = 0;
                if (__ballot(len_v > SHORT_MAX)) {  // a longer sequence: the whole chunk goes to the generic kernel
                    if (lane == 0) {
                        atomicOr(&launder(ap)->defer_bits[claim >> 5], 1u << (claim & 31));
                        // (one word for the whole launch: when every chunk is deferred -- 2 x 250 bp -- 25 000 atomics on it
                        //  were half of this kernel's 0.41 ms; it only ever goes from 0 to 1)
                        if (__hip_atomic_load(launder(ap)->pending_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                            atomicMax(launder(ap)->pending_long, 1);
                    }
                    nseq = 0;
                    t = 0;
                    take_claim();
                    continue;
                }
                const uint32_t cb = wave_sum(len_v);
                if (lane == 0) {
                    S.acc[CNT_FRAGMENTS] += ncf;
                    S.acc[CNT_BASES] += cb;
                }
                nseq = ns;
                t = 0;
                // the first batch of a chunk is the only one whose bases are not loaded a batch ahead
#pragma unroll
                for (int j = 0; j < NSLOT; j++)
                    if ((uint32_t)j < nseq) w[j] = *tile_ptr(cbase + __builtin_amdgcn_readlane(off_v, j));
            }
        }
        uint32_t ambmask = 0;
        if (batch) {
            NH_STAMP(0);
            // ---- batch prologue: encode tiles t .. t+NSLOT-1, start the loads of the batch after ----
#pragma unroll
            for (int j = 0; j < NSLOT; j++) {
                const uint32_t i = t + (uint32_t)j;
                if (i < nseq) {
                    const uint32_t o = __builtin_amdgcn_readlane(off_v, i), n = __builtin_amdgcn_readlane(len_v, i);
                    const bool amb = encode_tile<STD>(S, lane, (uint32_t)j, w[j], (uint32_t)(cbase + o) & 3u, n);
                    ambmask |= (amb ? 1u : 0u) << j;
                }
                const uint32_t i2 = i + NSLOT;
                if (i2 < nseq) w[j] = *tile_ptr(cbase + __builtin_amdgcn_readlane(off_v, i2));
            }
            wave_sync();
            NH_STAMP(1);
            if (TLINE && lane == 0) {
                unsigned long long *const tline = tl_row();
                if (tline && tline[2] == 0) tline[2] = wall_clock64();
            }
            if (!claim_ahead && t + NSLOT >= nseq) take_claim();  // the chunk's last batch: take the claim ahead now
        }
        // ---- the batch's tiles; j == NSLOT is the end-of-batch turn --------------------------------
        for (uint32_t j = 0; j <= (uint32_t)NSLOT; j++) {
            const uint32_t i = t + j;
            if (batch && j < (uint32_t)NSLOT && i < nseq) {
                const uint32_t o = __builtin_amdgcn_readlane(off_v, i), n = __builtin_amdgcn_readlane(len_v, i);
                const uint32_t nk = n >= K ? n - K + 1 : 0;
                const uint32_t m = mates == 2 ? (i & 1u) : 0u;
                const uint64_t f = cbeg + (mates == 2 ? i >> 1 : i);
                uint32_t nk_other = 0;
                if (mates == 2) {
                    const uint32_t no = __builtin_amdgcn_readlane(len_v, i ^ 1u);
                    nk_other = no >= K ? no - K + 1 : 0;
                }
                const uint32_t nk0 = m ? nk_other : nk, nk1 = mates == 2 ? (m ? nk : nk_other) : 0u;
                if (nk0 + nk1 == 0) {  // no k-mer at all: all-zero record, only the mate border
                    if (lane == 0 && m == (uint32_t)mates - 1) {
                        KArgsP a2 = launder(ap);
                        uint4 rec = {0, 0, 0, 0};
                        *reinterpret_cast<uint4 *>(&a2->out[f]) = rec;
                        uint32_t *const kmer_taxa = a2->kmer_taxa;
                        if (kmer_taxa && mates == 2) kmer_taxa[a2->kmer_taxa_off[f]] = TAXON_MATE_BORDER;
                    }
                } else if (nk != 0) {
                    // kraken2's last_minimizer: fresh per fragment, and per mate when the DB says so
                    if (m == 0 || nk0 == 0 || reset_per_mate) carry_min = NH_FULL;
                    const bool frag_first = m == 0 || nk0 == 0;
                    const bool frag_end = m == (uint32_t)mates - 1 || nk1 == 0;
                    const bool mate_end = m == 0 && mates == 2 && nk1 != 0;
                    uint32_t ps;
                    int last_lane;
                    const uint32_t nruns = scan_body<STD, PROF>(ap, S, lane, j, (ambmask >> j) & 1u,
                                                                (uint32_t)(cbase + o) & 3u, n - L + 1, nk, par, qn,
                                                                carry_min, ps, last_lane, prof, tprev);
                    if (lane == 0) {
                        uint4 d0, d1, d2;  // layout of SlotLds
                        d0.x = (uint32_t)f;
                        d0.y = (uint32_t)(f >> 32);
                        d0.z = m ? nk0 + 1 : 0;  // k-mer index of the tile within its fragment
                        d0.w = 0;
                        d1.x = nk;
                        d1.y = qn;
                        d1.z = nruns;
                        d1.w = 0;
                        d2.x = (uint32_t)last_lane;
                        d2.y = (frag_end ? 1u : 0u) | (mate_end ? 2u : 0u) | (frag_first ? 4u : 0u);
                        d2.z = nk0;
                        d2.w = nk0 + nk1;
                        uint4 *dp = reinterpret_cast<uint4 *>(&S.slot[par][nslot_new]);
                        dp[0] = d0;
                        dp[1] = d1;
                        dp[2] = d2;
                    }
                    S.ps[par][nslot_new][lane] = (uint16_t)ps;
                    NH_STAMP(10);
                    qn += nruns;
                    nslot_new++;
                }
            }
            // group complete -- no room for another tile's run starts, or the batch is over -- or the
            // pipeline is being drained: hash + probe it (which also resolves what is left of the previous
            // group), post-process the previous group, switch buffers
            const bool full = nslot_new == (uint32_t)NSLOT || qn + TL > (uint32_t)QCAP_SHORT;
            if ((nslot_new != 0 && (full || j == (uint32_t)NSLOT)) || (!batch && j == (uint32_t)NSLOT)) {
                flush_records<STD>(ap, S, lane);
#ifdef NH_NO_QUAD
                probe_queue<true, STD, !WIDE, PROF>(ap, S, lane, par, qn, lk, true, 0ull, prof, tprev);
#else
                probe_queue_quad<PROF, WIDE>(ap, S, lane, par, qn, lk, true, 0ull, prof, tprev);
#endif
                if (TLINE && lane == 0) {
                    unsigned long long *const tline = tl_row();
                    if (tline && tline[3] == 0) tline[3] = wall_clock64();
                }
                if (nslot_old)
                    post_group<STD, false, PROF, false>(ap, S, TLI, lane, mates, reset_per_mate, par ^ 1u, nslot_old, prof, tprev);
                nslot_old = nslot_new;
                par ^= 1u;
                nslot_new = 0;
                qn = 0;
            }
        }
        if (batch) t += NSLOT;
    }
    flush_records<STD>(ap, S, lane);

    wave_sync();
    if (lane == 0) {
        unsigned long long *const counters = ap->counters;
        int *const error_flag = ap->error_flag;
        if (PROF && counters)
            for (int i = 0; i < 12; i++) atomicAdd(&counters[CNT_N + i], (unsigned long long)prof[i]);
        add_counters(ap, S.acc);
        if (bad_input) atomicOr(&error_flag[0], 2);
        if constexpr (TLINE)
            if (unsigned long long *const tline = tl_row()) tline[5] = wall_clock64();
    }
}

// Behind every launch, one wave: folds the COUNTER_SHARDS rows the waves added to into the caller's four
// counters, and leaves the launch slot CLEAN for its next launch -- counter rows, the work counters of the
// three passes, the deferral bitmap words that were used, the "chunks / fragments left" words, the item
// header of a long-read launch.  (Round 2 cleared these with four hipMemsetAsync calls per launch, each a
// dependent dispatch of its own: ~20 us of a 1.1 ms launch of 1 M reads.)
struct FinishArgs {
    unsigned long long *cshard, *counters, *work;
    uint32_t *defer;
    uint32_t n_defer_words;
    int *pending_long, *pending;
    SplitHdr *split_hdr;
};
__global__ __launch_bounds__(64) void k_finish_launch(const FinishArgs a) {
    const int lane = threadIdx.x;
    unsigned long long v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        unsigned long long x = 0;
        for (uint32_t r = lane; r < COUNTER_SHARDS; r += 64) {
            x += a.cshard[(size_t)r * COUNTER_STRIDE + i];
            a.cshard[(size_t)r * COUNTER_STRIDE + i] = 0;
        }
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
        v[i] = x;
    }
    if (lane == 0 && a.counters)
        for (int i = 0; i < 4; i++)
            if (v[i]) atomicAdd(&a.counters[i], v[i]);
    if ((uint32_t)lane < WORK_PASSES * WORK_WORDS) a.work[(size_t)lane * WORK_STRIDE] = 0;
    for (uint32_t w = lane; w < a.n_defer_words; w += 64) a.defer[w] = 0;
    if (lane == 0) {
        *a.pending_long = 0;
        *a.pending = 0;
        if (a.split_hdr) *a.split_hdr = SplitHdr{0, 0, 0, 0};
    }
}

// ---- long-read prepass: the launch's work items (SplitBufs, nh_device.h) -----------------------------------
// One thread per read.  Reads of more than SPLIT_MIN_TILES tiles reserve one item = one partial slot per
// segment (one atomic per wave); whole reads are listed by size class.  A wave whose segments no longer fit
// the buffers lists its reads whole and marks what it had reserved as holes (nseg = 0).
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, const int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

__global__ __launch_bounds__(256) void k_prep_items(const KArgs args_by_kernarg_pointer) {
    KArgsP ap = (KArgsP)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63;
    const uint64_t n_frag = ap->n_frag;
    const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = f < n_frag;
    const uint32_t K = ap->db.k, TQ = TL - ap->db.window;
    uint64_t len = 0;
    if (act) len = ap->seq_len ? (uint64_t)ap->seq_len[f] : ap->seq_off[f + 1] - ap->seq_off[f];
    if (len > 0x7FFFFFFFull) len = 0x7FFFFFFFull;  // (the classify kernel reports such a read)
    const uint32_t nk = len >= K ? (uint32_t)len - K + 1 : 0;
    const uint32_t nt = (nk + TQ - 1) / TQ;
    uint32_t nseg = nt > SPLIT_MIN_TILES ? (nt + SEG_TILES - 1) / SEG_TILES : 1u;
    bool multi = act && nseg > 1;
    SplitHdr *const hdr = ap->split.hdr;
    const uint32_t cap = ap->split.seg_cap;
    {   // segments: item index = partial slot index
        const uint32_t incl = wave_scan_incl(multi ? nseg : 0u, lane);
        const uint32_t tot = (uint32_t)__shfl((int)incl, 63, 64);
        if (tot) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&hdr->seg_used, tot);
            base = (uint32_t)__shfl((int)base, 0, 64);
            const uint32_t mine = base + incl - (multi ? nseg : 0u);
            const bool fits = (uint64_t)base + tot <= cap;
            if (multi) {
                SplitItem *const it = ap->split.items_multi;
                for (uint32_t s = 0; s < nseg; s++) {
                    if (mine + s >= cap) break;
                    SplitItem v;
                    v.f = (uint32_t)f;
                    v.seg = s;
                    v.nseg = fits ? nseg : 0u;  // 0: a hole, the read is listed whole below
                    v.slot = mine;
                    it[mine + s] = v;
                }
                if (fits) ap->split.part_done[mine] = 0;
                else multi = false;
            }
        }
    }
    const bool whole = act && !multi;
    const bool mid = whole && nt >= MID_TILES;
    const bool small = whole && !mid;
    {
        const uint64_t bm = __ballot(mid), bs = __ballot(small);
        uint32_t base_m = 0, base_s = 0;
        if (lane == 0) {
            if (bm) base_m = atomicAdd(&hdr->n_mid, (uint32_t)__popcll(bm));
            if (bs) base_s = atomicAdd(&hdr->n_small, (uint32_t)__popcll(bs));
        }
        base_m = (uint32_t)__shfl((int)base_m, 0, 64);
        base_s = (uint32_t)__shfl((int)base_s, 0, 64);
        uint32_t *const single = ap->split.items_single;
        if (mid) single[base_m + below(bm)] = (uint32_t)f;
        if (small) single[n_frag - 1 - (base_s + below(bs))] = (uint32_t)f;
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

// Inserts every minimizer of the given sequences into the table with a constant value (kraken2
// build_db.cc ProcessSequence + CompareAndSet, linear probing): lets the bench put "human" reads
// into the synthetic table so that the hit path is measured too.  Default geometry (STD) only.
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK) void k_insert_sequences(const KArgs args_by_kernarg_pointer,
                                                                            const uint32_t value) {
    KArgsP ap = (KArgsP)__builtin_amdgcn_kernarg_segment_ptr();
    typedef WaveLdsT<true, 1, QCAP_GENERIC> WL;
    __shared__ WL lds_all[WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WL &S = lds_all[wib];
    init_wave_lds<true>(S, lane);
    const uint32_t TQ = TL - 4u;
    const uint64_t n_seq = ap->n_frag;
    const uint64_t *const seq_off = ap->seq_off;
    const uint64_t last_dw = (seq_off[n_seq] + 4) >> 2;
    uint32_t *const table = const_cast<uint32_t *>(ap->db.table);
    const uint64_t cap = ap->db.capacity, magic = ap->db.cap_magic;
    const uint32_t vbits = ap->db.value_bits;
    uint64_t prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tprev = 0;
    unsigned long long inserted = 0;
    const uint64_t n_waves = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    for (uint64_t s = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wib; s < n_seq; s += n_waves) {
        const uint64_t o0 = seq_off[s];
        const uint32_t n = (uint32_t)(seq_off[s + 1] - o0);
        const uint32_t nk = n >= 35u ? n - 34u : 0;
        uint64_t carry_min = NH_FULL;
        for (uint32_t q0 = 0; q0 < nk; q0 += TQ) {
            const uint64_t g0 = o0 + q0;
            uint64_t dw = (g0 >> 2) + ((uint32_t)lane < PREF_LANES ? (uint32_t)lane : PREF_LANES - 1);
            dw = dw < last_dw ? dw : last_dw;
            const uint32_t *wp = reinterpret_cast<const uint32_t *>(ap->bases) + dw;
            const uint32_t w = *wp;
            const uint32_t nl_left = (n - 31u + 1) - q0;
            const uint32_t nlt = nl_left < (uint32_t)TL ? nl_left : (uint32_t)TL;
            const uint32_t nq_left = nk - q0;
            const uint32_t nqt = nq_left < TQ ? nq_left : TQ;
            uint32_t ps, wdummy = 0;
            int last_lane;
            const uint32_t nruns = scan_tile<true, false>(ap, S, lane, w, (uint32_t)g0 & 3u, nlt, nqt, 0u, 0u,
                                                          carry_min, ps, last_lane, wp, false, wdummy, prof,
                                                          tprev);
            wave_sync();
            for (uint32_t r = lane; r < nruns; r += 64) {
                const uint64_t hc = fmix64(S.q[0][r]);
                const uint32_t compacted = (uint32_t)(hc >> (32 + vbits));
                const uint32_t cell = (compacted << vbits) | value;
                uint64_t idx = mod_capacity(hc, cap, magic);
                for (uint64_t tries = 0; tries < cap; tries++) {
                    const uint32_t old = atomicCAS(&table[idx], 0u, cell);
                    if (old == 0) {
                        inserted++;
                        break;
                    }
                    if ((old >> vbits) == compacted) break;
                    idx++;
                    if (idx >= cap) idx = 0;
                }
            }
            wave_sync();
        }
    }
    for (int d = 32; d >= 1; d >>= 1) inserted += __shfl_xor(inserted, d, 64);
    if (lane == 0 && inserted) atomicAdd(&ap->counters[0], inserted);
}

// ---- database check at open (nh_engine.hip: validate_table) ------------------------------------------------
// One streaming pass over the cells of copy 0: how many are non-empty (value field != 0: kraken2's CompactHashCell is
// empty when its value is 0) and the largest value.  Every kernel above indexes parent[] with a cell's value
// unchecked -- the check is made here, once, at HBM speed (16 bytes a lane a step, one atomic pair per workgroup).
typedef nh_u32x4 u32x4;
__global__ __launch_bounds__(256) void k_validate_table(const u32x4 *cells4, const uint64_t n4, const uint32_t vmask,
                                                        unsigned long long *out2) {
    unsigned long long cnt = 0;
    uint32_t mx = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const u32x4 c = __builtin_nontemporal_load(cells4 + i);  // (read once: no reason to keep the lines in the L2)
        const uint32_t a = c.x & vmask, b = c.y & vmask, d = c.z & vmask, e = c.w & vmask;
        cnt += (a != 0) + (b != 0) + (d != 0) + (e != 0);
        const uint32_t m1 = a > b ? a : b, m2 = d > e ? d : e;
        const uint32_t m = m1 > m2 ? m1 : m2;
        mx = m > mx ? m : mx;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        cnt += __shfl_xor(cnt, d, 64);
        const uint32_t o = (uint32_t)__shfl_xor((int)mx, d, 64);
        mx = o > mx ? o : mx;
    }
    __shared__ unsigned long long s_cnt[4];
    __shared__ uint32_t s_mx[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_cnt[w] = cnt;
        s_mx[w] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) {
            cnt += s_cnt[i];
            mx = s_mx[i] > mx ? s_mx[i] : mx;
        }
        if (cnt) atomicAdd(&out2[0], cnt);
        if (mx) atomicMax(&out2[1], (unsigned long long)mx);
    }
}

// n_cells: a multiple of 4 (the allocation's zero padding counts as empty cells); table: 16-byte aligned
hipError_t launch_validate_table(const uint32_t *table, uint64_t n_cells, uint32_t vmask, unsigned long long *d_out2,
                                 hipStream_t stream) {
    const uint64_t n4 = n_cells / 4;
    if (n4 == 0) return hipSuccess;
    const uint64_t want = (n4 + 255) / 256;
    const unsigned grid = (unsigned)(want < 256u * 16u ? want : 256u * 16u);
    hipLaunchKernelGGL(k_validate_table, dim3(grid), dim3(256), 0, stream, (const u32x4 *)table, n4, vmask, d_out2);
    return hipGetLastError();
}

// ---- host-side launchers ---------------------------------------------------------------------
static bool is_std(const DevDB &db) {
    return db.k == 35 && db.l == 31 && db.revcom_version != 0 && db.min_hash == 0;
}

template <bool LINEAR, bool STD, bool CAP32, bool PROF = false>
static void launch_variant(const KArgs &ka, dim3 g, dim3 b, hipStream_t stream, bool may_overflow,
                           unsigned long long *d_work) {
    hipLaunchKernelGGL((k_classify<LINEAR, STD, CAP32, PROF, false>), g, b, 0, stream, ka);
    if (may_overflow) {
        // second pass for fragments with more than 64 distinct taxa: exits at once if there are none
        KArgs kb = ka;
        kb.only_deferred = 0;
        kb.work = d_work + 2 * (size_t)WORK_WORDS * WORK_STRIDE;  // the third pass has work counters of its own
        hipLaunchKernelGGL((k_classify<LINEAR, STD, CAP32, false, true>), g, b, 0, stream, kb);
    }
}

// The claim map of a launch (Sched, nh_device.h): chunks of c0 fragments for the body; then, for about half a
// c0-chunk of every resident wave, chunks of c1 = c0 / 2; then, for half a c1-chunk per wave, chunks of c2 =
// three batches of four tiles -- the waves that finish early fill up on the small ones and the launch ends
// within ~80 us instead of ~200 (profiles/r03_sched.txt).  It pays little (1 M single reads 1.124 -> 1.107 ms,
// 2.5 M pairs 4.884 -> 4.858): while the tail of a flat map has idle waves, the others run faster, the kernel
// being memory-bound.  And it only pays since (a) the waves' counters go to rows of their own -- before, the
// end of a launch was an atomic storm that no claim map could see through -- and (b) the tail's claims come
// from TAIL_SHARDS counters: four times the claim rate in the last moments is more than one word retires.
// NOHUMAN_SCHED=off: flat; NOHUMAN_SCHED=c1,c2,p1,p2: sizes and tail lengths in percent (tuning knob).
// (the knob is parsed once, when the engine is opened: read_launch_knobs)
SchedKnobs parse_sched_knobs(const char *env) {
    SchedKnobs kn;
    if (!env) return kn;
    unsigned a = 0, b = 0, c = 0, d = 0;
    if (strcmp(env, "off") == 0) kn.set = kn.off = true;
    else if (sscanf(env, "%u,%u,%u,%u", &a, &b, &c, &d) == 4 && a >= 1 && b >= 1) {
        kn.set = true;
        kn.c1 = a;
        kn.c2 = b;
        kn.p1 = c;
        kn.p2 = d;
    }
    return kn;
}
Sched make_sched(uint64_t n_frag, uint32_t c0, int mates, uint64_t waves, const SchedKnobs &kn) {
    Sched sc;
    memset(&sc, 0, sizeof sc);
    if (c0 == 0) c0 = 1;
    const uint32_t step = mates == 2 ? 2u : 4u;  // one batch of four tiles
    uint32_t c1 = c0 / 2 / step * step, c2 = 3 * step;
    uint64_t p1 = 100, p2 = 100;
    bool guided = c0 >= 4 * step;
    if (kn.set) {
        guided = !kn.off;
        if (guided) {
            c1 = kn.c1;
            c2 = kn.c2;
            p1 = kn.p1;
            p2 = kn.p2;
        }
    }
    if (c1 > c0) c1 = c0;
    if (c2 > c1) c2 = c1;
    sc.c0 = c0;
    sc.c1 = guided ? c1 : c0;
    sc.c2 = guided ? c2 : c0;
    uint64_t t1 = guided ? waves * c0 / 2 * p1 / 100 : 0, t2 = guided ? waves * c1 / 2 * p2 / 100 : 0;
    if (t1 + t2 > n_frag / 2) {  // a small launch: the tail is at most half of it
        const uint64_t cut = n_frag / 2;
        const uint64_t s = t1 + t2;
        t1 = t1 * cut / s;
        t2 = t2 * cut / s;
    }
    sc.n0 = (n_frag - t1 - t2) / c0;
    sc.base1 = sc.n0 * c0;
    const uint64_t r1 = n_frag - sc.base1;               // fragments left to the smaller chunks
    const uint64_t n1 = r1 > t2 ? (r1 - t2) / sc.c1 : 0;  // ... of which these many chunks of c1
    sc.n01 = sc.n0 + n1;
    sc.base2 = sc.base1 + n1 * sc.c1;
    sc.total = sc.n01 + (n_frag - sc.base2 + sc.c2 - 1) / sc.c2;
    return sc;
}

hipError_t launch_classify(const DevDB &db, const LaunchIO &io, double confidence, const LaunchSlot &sl,
                           uint32_t frag_chunk, int grid_blocks, hipStream_t stream, const SchedKnobs &sched_knobs) {
    const uint64_t n_frag = io.n_frag;
    if (n_frag == 0) return hipSuccess;
    if (frag_chunk == 0) frag_chunk = 1;
    // (the launch slot is clean: its previous launch's k_finish_launch, or the engine's start, left it so)
    const Sched sched = make_sched(n_frag, frag_chunk, io.mates, (uint64_t)grid_blocks * WAVES_PER_BLOCK, sched_knobs);
    uint64_t need = (sched.total + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    int grid = (int)(need < (uint64_t)grid_blocks ? need : (uint64_t)grid_blocks);
    dim3 g(grid), b(WAVE * WAVES_PER_BLOCK);
    const bool std_geom = is_std(db);
    KArgs ka;
    memset(&ka, 0, sizeof ka);
    ka.sched = sched;
    ka.db = db;
    ka.bases = (const uint8_t *)io.d_bases;
    ka.seq_off = (const uint64_t *)io.d_seq_off;
    ka.seq_len = (const uint32_t *)io.d_seq_len;
    ka.bases_end = io.bases_end;
    ka.n_frag = n_frag;
    ka.mates = io.mates;
    ka.frag_chunk = frag_chunk;
    ka.confidence = confidence;
    ka.out = (Result *)io.d_out;
    ka.kmer_taxa = (uint32_t *)io.d_kmer_taxa;
    ka.kmer_taxa_off = (const uint64_t *)io.d_kmer_taxa_off;
    ka.counters = (unsigned long long *)io.d_counters;
    ka.cshard = sl.d_cshard;
    ka.error_flag = sl.d_error;
    ka.pending = sl.d_pending;
    ka.work = sl.d_work;
    ka.defer_bits = sl.d_defer;
    ka.pending_long = sl.d_pending_long;
#ifdef NH_TIMELINE  // tuning builds only (tools/timeline.py): a raw device pointer from the environment has no place in the product
    if (const char *tl = getenv("NH_TIMELINE_PTR")) ka.timeline = (unsigned long long *)strtoull(tl, nullptr, 0);
#endif
    // Long single-end reads: a prepass lists the launch's work items -- segments of the reads worth cutting,
    // then whole reads by size class -- and the generic kernel claims items (SplitBufs, nh_device.h)
    static const bool no_split = getenv("NOHUMAN_NO_SPLIT") != nullptr;  // tuning / test knob
    const bool use_items = io.long_reads && io.mates == 1 && sl.split.hdr != nullptr && !no_split &&
                           n_frag <= sl.split_single_cap && n_frag < 0xFFFFFFFFull;
    if (sl.split_fresh && sl.split.hdr) (void)hipMemsetAsync(sl.split.hdr, 0, sizeof(SplitHdr), stream);  // fresh buffers
    if (use_items) {
        ka.split = sl.split;
        hipLaunchKernelGGL(k_prep_items, dim3((unsigned)((n_frag + 255) / 256)), dim3(256), 0, stream, ka);
        g = dim3(grid_blocks);  // (the number of items is only known on the device)
    }
    // tables of 2^32 - 256 cells or more take the variant with 64-bit cell positions;
    // NOHUMAN_FORCE_WIDE=1 selects it for any table (tests: small tables through the wide path)
    static const bool force_wide = getenv("NOHUMAN_FORCE_WIDE") != nullptr;
    static const bool no_short = getenv("NOHUMAN_NO_SHORT") != nullptr;  // tuning / test knob
    static const bool phase_prof = getenv("NH_PHASE_PROF") != nullptr;
    const bool cap32 = db.capacity < 0xFFFFFF00ull && !force_wide;
    // <= 64 taxa can never overflow the list; a segment of a split read keeps at most PART_CAP
    const bool may_overflow = db.node_count > LIST_CAP || (use_items && db.node_count > PART_CAP);
    const bool hot = db.linear_probing && std_geom;  // quad probing, short-read kernel (32- or 64-bit cell positions)
    // Short reads first: the one-tile-per-sequence kernel takes every chunk whose sequences fit a tile and
    // marks the others for the generic kernel launched right behind it (which returns at once when
    // nothing was marked).  Skipped when the caller says the reads are long.
    const uint64_t n_chunks = sched.total;
    uint32_t n_defer_words = 0;
    if (hot && !io.long_reads && !no_short && n_chunks <= sl.defer_cap_bits) {
        n_defer_words = (uint32_t)((n_chunks + 31) / 32);
        if (phase_prof && cap32)
            hipLaunchKernelGGL((k_classify_short<true, false, false>), g, b, 0, stream, ka);
        else if (cap32 && ka.timeline)
            hipLaunchKernelGGL((k_classify_short<false, false, true>), g, b, 0, stream, ka);
        else if (cap32)
            hipLaunchKernelGGL((k_classify_short<false, false, false>), g, b, 0, stream, ka);
        else
            hipLaunchKernelGGL((k_classify_short<false, true, false>), g, b, 0, stream, ka);
        ka.only_deferred = 1;
        ka.work = sl.d_work + (size_t)WORK_WORDS * WORK_STRIDE;  // the second pass has work counters of its own
    }
    if (hot && cap32 && phase_prof)
        launch_variant<true, true, true, true>(ka, g, b, stream, may_overflow, sl.d_work);  // d_counters: CNT_N + 12 words
    else if (hot && cap32)
        launch_variant<true, true, true>(ka, g, b, stream, may_overflow, sl.d_work);
    else if (hot)
        launch_variant<true, true, false>(ka, g, b, stream, may_overflow, sl.d_work);
    else if (db.linear_probing)
        launch_variant<true, false, false>(ka, g, b, stream, may_overflow, sl.d_work);
    else
        launch_variant<false, false, false>(ka, g, b, stream, may_overflow, sl.d_work);
    FinishArgs fa;
    fa.cshard = sl.d_cshard;
    fa.counters = (unsigned long long *)io.d_counters;
    fa.work = sl.d_work;
    fa.defer = sl.d_defer;
    fa.n_defer_words = n_defer_words;
    fa.pending_long = sl.d_pending_long;
    fa.pending = sl.d_pending;
    fa.split_hdr = use_items ? sl.split.hdr : nullptr;
    hipLaunchKernelGGL(k_finish_launch, dim3(1), dim3(64), 0, stream, fa);
    return hipGetLastError();
}

hipError_t launch_insert_sequences(const DevDB &db, const void *d_bases, const void *d_seq_off,
                                   uint64_t n_seq, uint32_t value, unsigned long long *d_inserted,
                                   int grid_blocks, hipStream_t stream) {
    if (n_seq == 0) return hipSuccess;
    if (!is_std(db) || !db.linear_probing) return hipErrorInvalidValue;
    KArgs ka;
    memset(&ka, 0, sizeof ka);
    ka.db = db;
    ka.bases = (const uint8_t *)d_bases;
    ka.seq_off = (const uint64_t *)d_seq_off;
    ka.n_frag = n_seq;
    ka.counters = d_inserted;
    hipLaunchKernelGGL(k_insert_sequences, dim3(grid_blocks), dim3(WAVE * WAVES_PER_BLOCK), 0, stream, ka,
                       value);
    return hipGetLastError();
}

int classify_blocks_per_cu() {
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_classify_short<false, false, false>,
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
