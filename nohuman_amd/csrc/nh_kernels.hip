// nh_kernels.hip -- gfx950 (CDNA4) kernels of the read-classification hot path.
//
// One 64-lane wavefront classifies one fragment (read or read pair) end to end; it replaces the
// body of kraken2's ClassifySequence loop (minimizer scan -> compact-hash probe -> ResolveTree;
// SURVEY.md section 8a rows a5-a9, Appendix A.2-A.5) that nohuman reaches through
// /root/reference/src/lib.rs:22-48.  Integer / byte work bound by random HBM line fetches:
// no MFMA by design.
//
// Data flow of one tile (128 l-mers = 2 per lane, i.e. 128-(k-l) k-mers):
//   global bases (one coalesced dword per lane) -> 2-bit packed stream in LDS (1 byte per lane)
//   -> per lane two l-mers by one 64-bit funnel read -> canonical/spaced/toggled candidates in LDS
//   -> per lane two k-mer minimizers = min over a (k-l+1)-wide candidate window
//   -> run starts (minimizer != previous non-ambiguous minimizer) compacted into an LDS queue
//   -> one lane per queued minimizer: fmix64, exact hc % capacity, 16-byte-chunk linear probe
//   -> taxa back through LDS -> per-taxon hit counts, hit groups -> ResolveTree on the wave.
#include <hip/hip_runtime.h>

#include "nh_device.h"

namespace nh {

#define NH_FULL 0xFFFFFFFFFFFFFFFFull

struct WaveLds {
    uint32_t pk[24];  // 2-bit packed bases: base i' of the tile frame at bit 2*(255-i'); 64 B + zero pad
    uint32_t pa[24];  // same layout, value 1 where the base is ambiguous
    uint64_t cand[TL + 2];
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

__device__ __forceinline__ uint64_t revcomp(uint64_t x, uint32_t l, int revcom_version,
                                            uint64_t lmer_mask) {
    uint64_t br = __builtin_bitreverse64(x);
    uint64_t r2 = ((br & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((br & 0x5555555555555555ull) << 1);
    uint64_t c = ~r2;
    if (revcom_version != 0) c >>= (64 - 2 * l);
    return c & lmer_mask;
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int src) {
    uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src);
    uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// CompactHashTable::Get (A.4).  Linear probing walks 16-byte chunks (4 cells per load).
template <bool LINEAR>
__device__ __forceinline__ uint32_t table_get(const DevDB &db, uint64_t hc) {
    const uint32_t vbits = db.value_bits;
    const uint32_t vmask = db.vmask;
    const uint32_t compacted = (uint32_t)(hc >> (32 + vbits));
    const uint64_t cap = db.capacity;
    uint64_t idx = mod_capacity(hc, cap, db.cap_magic);
    if (LINEAR) {
        uint64_t pos = idx;
        uint64_t scanned = 0;
        for (;;) {
            const uint64_t base = pos & ~3ull;
            const uint4 c = *reinterpret_cast<const uint4 *>(db.table + base);
            const uint32_t cells[4] = {c.x, c.y, c.z, c.w};
            const uint32_t first = (uint32_t)(pos & 3);
            bool wrapped = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if ((uint32_t)j < first || wrapped) continue;
                if (base + j >= cap) {
                    wrapped = true;
                    continue;
                }
                const uint32_t cell = cells[j];
                if ((cell & vmask) == 0) return 0;
                if ((cell >> vbits) == compacted) return cell & vmask;
                scanned++;
            }
            if (scanned >= cap) return 0;
            pos = base + 4;
            if (wrapped || pos >= cap) pos = 0;
        }
    } else {
        const uint64_t first_idx = idx;
        const uint64_t step = mod_capacity((hc >> 8) | 1, cap, db.cap_magic);
        for (;;) {
            const uint32_t cell = db.table[idx];
            if ((cell & vmask) == 0) return 0;
            if ((cell >> vbits) == compacted) return cell & vmask;
            idx += step;
            if (idx >= cap) idx -= cap;
            if (idx == first_idx) return 0;
        }
    }
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

// 4 ASCII bytes -> packed 2-bit codes (first base in bits 7:6) and per-base "not ACGTacgt" flags
__device__ __forceinline__ void encode4(uint32_t w, uint32_t &codes, uint32_t &bad) {
    codes = 0;
    bad = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const uint32_t ch = (w >> (8 * b)) & 0xDFu;  // fold case
        const uint32_t x = (ch >> 1) & 3u;
        const uint32_t code = x ^ (x >> 1);  // A0 C1 G2 T3
        const bool ok = (ch == 0x41u) | (ch == 0x43u) | (ch == 0x47u) | (ch == 0x54u);
        codes |= code << (6 - 2 * b);
        bad |= (ok ? 0u : 1u) << (6 - 2 * b);
    }
}

__device__ __forceinline__ uint64_t funnel_read(const uint32_t *pkd, uint32_t s) {
    const uint32_t d = s >> 5, r = s & 31u;
    const uint64_t lo = (uint64_t)pkd[d] | ((uint64_t)pkd[d + 1] << 32);
    const uint64_t hi = pkd[d + 2];
    return r ? ((lo >> r) | (hi << (64 - r))) : lo;
}

template <bool LINEAR>
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK) void k_classify(
    const DevDB db, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ seq_off,
    const uint64_t n_frag, const int mates, const double confidence, Result *__restrict__ out,
    uint32_t *__restrict__ kmer_taxa, const uint64_t *__restrict__ kmer_taxa_off,
    unsigned long long *__restrict__ counters, int *__restrict__ error_flag) {
    __shared__ WaveLds lds_all[WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    WaveLds &S = lds_all[wib];

    // one-time LDS init: zero pads of the packed streams, sentinel tail of the candidate array
    if (lane < 24) {
        S.pk[lane] = 0;
        S.pa[lane] = 0;
    }
    if (lane < 2) S.cand[TL + lane] = NH_FULL;
    wave_sync();

    const uint32_t K = db.k, L = db.l, W = db.window;
    const uint32_t TQ = TL - W;  // k-mers per tile
    const uint64_t lane_lt = (lane == 0) ? 0ull : (NH_FULL >> (64 - lane));

    uint64_t acc_frag = 0, acc_class = 0, acc_bases = 0, acc_lookups = 0;
    const uint64_t n_waves = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    for (uint64_t f = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wib; f < n_frag; f += n_waves) {
        uint32_t nlist = 0;
        uint32_t hit_groups = 0;
        uint32_t total_kmers = 0;
        bool overflow = false;
        uint64_t carry_min = NH_FULL;
        uint32_t carry_tax = 0;
        uint64_t kt_pos = kmer_taxa ? kmer_taxa_off[f] : 0;

        for (int m = 0; m < mates; m++) {
            const uint64_t sidx = f * (uint64_t)mates + (uint64_t)m;
            const uint64_t sbeg = seq_off[sidx];
            const uint64_t n = seq_off[sidx + 1] - sbeg;
            acc_bases += n;
            if (db.reset_per_mate) {
                carry_min = NH_FULL;
                carry_tax = 0;
            }
            const uint64_t nk = n >= K ? n - K + 1 : 0;
            const uint64_t nl = n >= L ? n - L + 1 : 0;
            total_kmers += (uint32_t)nk;

            for (uint64_t q0 = 0; q0 < nk; q0 += TQ) {
                // ---- 1. bases -> packed 2-bit stream (one dword of 4 bases per lane) --------
                const uint64_t g0 = sbeg + q0;
                const uint64_t a0 = g0 & ~3ull;
                const uint32_t sh = (uint32_t)(g0 & 3);
                const uint32_t nlt = (uint32_t)((nl - q0) < (uint64_t)TL ? (nl - q0) : TL);
                const uint32_t nqt = (uint32_t)((nk - q0) < (uint64_t)TQ ? (nk - q0) : TQ);
                const uint32_t nb = nlt + L - 1;  // bases of this tile
                const uint32_t ndw = (sh + nb + 3) >> 2;
                uint32_t w = 0x41414141u;
                if ((uint32_t)lane < ndw)
                    w = *reinterpret_cast<const uint32_t *>(bases + a0 + 4ull * lane);
                uint32_t codes, bad;
                encode4(w, codes, bad);
                // keep only flags of real bases of this read: frame positions [sh, sh+nb)
                {
                    const int p0 = 4 * lane;
                    uint32_t keep = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const uint32_t p = (uint32_t)(p0 + b);
                        if (p >= sh && p < sh + nb) keep |= 1u << (6 - 2 * b);
                    }
                    bad &= keep;
                }
                reinterpret_cast<uint8_t *>(S.pk)[63 - lane] = (uint8_t)codes;
                const bool has_amb = __ballot(bad != 0) != 0;
                if (has_amb) reinterpret_cast<uint8_t *>(S.pa)[63 - lane] = (uint8_t)bad;
                wave_sync();

                // ---- 2. two l-mers per lane -> candidates ----------------------------------
                {
                    const uint32_t j1 = sh + 2 * lane + L;  // frame index of the last base of l-mer 2t+1
                    const uint32_t s = 2 * (255 - j1);
                    const uint64_t wv = funnel_read(S.pk, s);
                    const uint64_t lm1 = wv & db.lmer_mask;
                    const uint64_t lm0 = (wv >> 2) & db.lmer_mask;
                    bool amb0 = false, amb1 = false;
                    if (has_amb) {
                        const uint64_t wa = funnel_read(S.pa, s);
                        amb1 = (wa & db.lmer_mask) != 0;
                        amb0 = ((wa >> 2) & db.lmer_mask) != 0;
                    }
                    const uint64_t rc0 = revcomp(lm0, L, db.revcom_version, db.lmer_mask);
                    const uint64_t rc1 = revcomp(lm1, L, db.revcom_version, db.lmer_mask);
                    uint64_t c0 = ((lm0 < rc0 ? lm0 : rc0) & db.spaced_mask) ^ db.toggle;
                    uint64_t c1 = ((lm1 < rc1 ? lm1 : rc1) & db.spaced_mask) ^ db.toggle;
                    if (amb0 || (uint32_t)(2 * lane) >= nlt) c0 = NH_FULL;
                    if (amb1 || (uint32_t)(2 * lane + 1) >= nlt) c1 = NH_FULL;
                    ulonglong2 cc;
                    cc.x = c0;
                    cc.y = c1;
                    *reinterpret_cast<ulonglong2 *>(&S.cand[2 * lane]) = cc;
                }
                wave_sync();

                // ---- 3. two k-mer minimizers per lane (window min) -------------------------
                const uint32_t qi0 = 2 * lane, qi1 = 2 * lane + 1;
                const bool valid0 = qi0 < nqt, valid1 = qi1 < nqt;
                uint64_t mz0 = 0, mz1 = 0;
                bool v0 = false, v1 = false;  // valid and non-ambiguous
                if (valid0) {
                    const uint64_t first = S.cand[qi0];
                    const uint64_t last1 = S.cand[qi1 + W];
                    uint64_t last0, m0, m1;
                    if (W == 0) {
                        last0 = first;
                        m0 = first;
                        m1 = last1;
                    } else {
                        uint64_t mid = S.cand[qi0 + 1];
                        for (uint32_t i = 2; i <= W; i++) {
                            const uint64_t c = S.cand[qi0 + i];
                            mid = c < mid ? c : mid;
                        }
                        last0 = S.cand[qi0 + W];
                        m0 = first < mid ? first : mid;
                        m1 = last1 < mid ? last1 : mid;
                    }
                    v0 = last0 != NH_FULL;
                    v1 = valid1 && last1 != NH_FULL;
                    mz0 = m0 ^ db.toggle;
                    mz1 = m1 ^ db.toggle;
                }

                // ---- 4. run starts: minimizer differs from the previous non-ambiguous one ---
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
                const bool new0 = v0 && (mz0 != prev_in);
                const bool new1 = v1 && (mz1 != prev1);

                // ---- 5. compact run starts into the LDS queue ------------------------------
                const uint64_t b0 = __ballot(new0), b1 = __ballot(new1);
                const uint32_t ex = __popcll(b0 & lane_lt) + __popcll(b1 & lane_lt);
                const uint32_t nruns = __popcll(b0) + __popcll(b1);
                if (new0) S.runmin[ex] = mz0;
                if (new1) S.runmin[ex + (new0 ? 1u : 0u)] = mz1;
                const int ri0 = (int)(ex + (new0 ? 1u : 0u)) - 1;
                const int ri1 = ri0 + (new1 ? 1 : 0);
                wave_sync();

                // ---- 6. one lane per queued minimizer: hash + probe -------------------------
                for (uint32_t r = lane; r < ((nruns + 63u) & ~63u); r += 64) {
                    const bool act = r < nruns;
                    uint32_t taxon = 0;
                    bool looked = false;
                    if (act) {
                        const uint64_t hc = fmix64(S.runmin[r]);
                        if (!(db.min_hash && hc < db.min_hash)) {
                            taxon = table_get<LINEAR>(db, hc);
                            looked = true;
                        }
                        S.runtax[r] = taxon;
                    }
                    hit_groups += __popcll(__ballot(act && taxon != 0));
                    acc_lookups += __popcll(__ballot(looked));
                }
                wave_sync();

                // ---- 7. per-k-mer taxa, hit counts, carry ----------------------------------
                uint32_t t0 = 0, t1 = 0;
                if (v0) t0 = ri0 >= 0 ? S.runtax[ri0] : carry_tax;
                if (v1) t1 = ri1 >= 0 ? S.runtax[ri1] : carry_tax;
                if (kmer_taxa) {
                    if (valid0) kmer_taxa[kt_pos + q0 + qi0] = v0 ? t0 : TAXON_AMBIGUOUS;
                    if (valid1) kmer_taxa[kt_pos + q0 + qi1] = v1 ? t1 : TAXON_AMBIGUOUS;
                }
                {
                    const uint64_t m1 = __ballot(v1), m0 = __ballot(v0);
                    if (m0 | m1) {
                        const int l1 = m1 ? 63 - __builtin_clzll(m1) : -1;
                        const int l0 = m0 ? 63 - __builtin_clzll(m0) : -1;
                        if (l1 >= l0) {
                            carry_min = readlane64(mz1, l1);
                            carry_tax = __builtin_amdgcn_readlane(t1, l1);
                        } else {
                            carry_min = readlane64(mz0, l0);
                            carry_tax = __builtin_amdgcn_readlane(t0, l0);
                        }
                    }
                }
                // distinct non-zero taxa of this tile -> (taxon, count) list
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
                    const bool match = (uint32_t)lane < nlist && S.list_tax[lane] == T;
                    const uint64_t mb = __ballot(match);
                    if (mb) {
                        if (match) S.list_cnt[lane] += cnt;
                    } else if (nlist < (uint32_t)LIST_CAP) {
                        if (lane == 0) {
                            S.list_tax[nlist] = T;
                            S.list_cnt[nlist] = cnt;
                        }
                        nlist++;
                    } else {
                        overflow = true;
                    }
                    wave_sync();
                }
            }  // tiles
            if (kmer_taxa) {
                kt_pos += nk;
                if (mates == 2 && m == 0) {
                    if (lane == 0) kmer_taxa[kt_pos] = TAXON_MATE_BORDER;
                    kt_pos += 1;
                }
            }
        }  // mates

        // ---- ResolveTree (A.5) on the wave: lane i owns list entry i -------------------------
        wave_sync();
        const uint32_t *parent = db.parent;
        const bool own = (uint32_t)lane < nlist;
        const uint32_t my_t = own ? S.list_tax[lane] : 0;
        const uint32_t my_c = own ? S.list_cnt[lane] : 0;
        uint32_t call = 0, clade_hits = 0;
        if (nlist == 1) {
            call = S.list_tax[0];
        } else if (nlist > 1) {
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
        if (nlist > 0) {
            const uint32_t required = (uint32_t)ceil(confidence * (double)total_kmers);
            // hits exactly at `call`
            uint32_t s = wave_sum((own && my_t == call) ? my_c : 0u);
            while (call && s < required) {
                s = wave_sum((own && is_a_ancestor_of_b(parent, call, my_t)) ? my_c : 0u);
                if (s >= required) break;
                call = parent[call];
            }
            if (call && hit_groups < db.min_hit_groups) call = 0;
            if (call)
                clade_hits = wave_sum((own && is_a_ancestor_of_b(parent, call, my_t)) ? my_c : 0u);
        }
        if (lane == 0) {
            uint4 rec;
            rec.x = call;
            rec.y = total_kmers;
            rec.z = clade_hits;
            rec.w = hit_groups;
            *reinterpret_cast<uint4 *>(&out[f]) = rec;
            if (overflow) atomicMax(error_flag, 1);
        }
        acc_frag += 1;
        acc_class += call ? 1 : 0;
    }  // fragments

    if (counters && lane == 0) {
        atomicAdd(&counters[CNT_FRAGMENTS], (unsigned long long)acc_frag);
        atomicAdd(&counters[CNT_CLASSIFIED], (unsigned long long)acc_class);
        atomicAdd(&counters[CNT_BASES], (unsigned long long)acc_bases);
        atomicAdd(&counters[CNT_LOOKUPS], (unsigned long long)acc_lookups);
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
hipError_t launch_classify(const DevDB &db, const void *d_bases, const void *d_seq_off,
                           uint64_t n_frag, int mates, double confidence, void *d_out,
                           void *d_kmer_taxa, const void *d_kmer_taxa_off, void *d_counters,
                           int *d_error, int grid_blocks, hipStream_t stream) {
    if (n_frag == 0) return hipSuccess;
    uint64_t need = (n_frag + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    int grid = (int)(need < (uint64_t)grid_blocks ? need : (uint64_t)grid_blocks);
    dim3 g(grid), b(WAVE * WAVES_PER_BLOCK);
    if (db.linear_probing)
        hipLaunchKernelGGL(k_classify<true>, g, b, 0, stream, db, (const uint8_t *)d_bases,
                           (const uint64_t *)d_seq_off, n_frag, mates, confidence, (Result *)d_out,
                           (uint32_t *)d_kmer_taxa, (const uint64_t *)d_kmer_taxa_off,
                           (unsigned long long *)d_counters, d_error);
    else
        hipLaunchKernelGGL(k_classify<false>, g, b, 0, stream, db, (const uint8_t *)d_bases,
                           (const uint64_t *)d_seq_off, n_frag, mates, confidence, (Result *)d_out,
                           (uint32_t *)d_kmer_taxa, (const uint64_t *)d_kmer_taxa_off,
                           (unsigned long long *)d_counters, d_error);
    return hipGetLastError();
}

int classify_blocks_per_cu() {
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_classify<true>,
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
