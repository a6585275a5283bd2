// gpu_inflate_proto.hip -- prototype with a measured go / no-go for the next step of the input side (DESIGN.md section
// 6, "what comes next"): the end-to-end run is bound by inflating its gzip inputs on the host cores.  How fast does
// ONE WAVE inflate a chunk of a deflate stream on an MI355X, in the marker mode of nh_inflate.cpp's speculative scheme
// (16-bit symbols; what a match copies out of the unknown 32 KiB before the chunk is the marker 0x8000 | index)?
//
//   host     zlib with Z_BLOCK lists the block boundaries of the stream (bit position, output offset) -- the block
//            SEARCH of the real scheme is left out, the prototype starts every chunk at a known boundary -- and inflates
//            the text for the comparison
//   kernel   a wave per chunk of ~4 MiB of compressed bytes.  All decode state is wave-uniform: the input window lives
//            in two registers per lane (128 dwords, the next 64 prefetched), a token costs one 64-bit bit fetch by
//            v_readlane, a 10-bit root table in LDS (longer codes: canonical walk), and -- for a match -- a copy by all
//            64 lanes through a 32 Ki-symbol circular window in LDS that starts out as the markers themselves, so that a
//            copy out of the unknown window is a copy like any other.  Dynamic, fixed and stored blocks.
//   check    markers replaced from the true window on the host, every byte compared with zlib's text
//   full     (third argument "full") the whole reader on the device: k_search finds the block starts itself, k_inflate_open
//            decodes from each to the next, the chunks' windows come from a serial walk (k_chain) and -- the same bytes,
//            twenty times faster -- from a prefix scan over the chunks' index maps (k_maps, k_scan_round, k_windows),
//            k_resolve replaces the markers; the text is compared with zlib's.  profiles/r03_gpu_inflate_proto.txt
//
//   hipcc --offload-arch=gfx950 -O3 tools/gpu_inflate_proto.hip -o tools/gpu_inflate_proto -lz
//   ./gpu_inflate_proto file.gz [chunk KiB = 4096] [full]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));             \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

struct ChunkDesc {
    uint64_t bit_start, bit_end;  // of the deflate data, from the start of the file buffer
    uint64_t out_start, out_end;  // positions in the text
    uint32_t status, blocks;      // written by the kernel: 0 ok, else an error code; blocks decoded
    uint64_t cycles;
};

constexpr int ROOT = 10, DROOT = 8;
constexpr uint32_t WSIZE = 32768;
#ifndef NEAR
#define NEAR 8192  // symbols of the window kept in LDS (a power of two <= 32768); older ones are read back from the output in HBM
#endif
constexpr uint32_t NEARSZ = NEAR;

struct Lds {
    uint16_t window[NEARSZ];
    // root tables: 0 = code longer than the root; else code length (4 bits) | extra bits (4) | base or literal (16) |
    // kind << 28 (0 literal, 1 length, 2 end of block; distances: always 1)
    uint32_t lit[1 << ROOT];
    uint32_t dist[1 << DROOT];
    uint32_t clt[128];
    uint16_t lbase[32], dbase[32];
    uint8_t lext[32], dext[32];
    uint16_t lsym[288], dsym[32];
    uint16_t lcount[16], dcount[16];
    uint16_t code[320];
    uint8_t lens[320];
    uint16_t tmp[40];
};

__constant__ uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// one wave per workgroup: LDS operations of a wave execute in order, so lanes see each other's LDS writes without a
// barrier; what is needed is that the compiler keeps the order (and __syncthreads() would also wait for the global
// stores of the copy before it -- most of a token's time in the first version of this file)
#define LDS_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ __forceinline__ uint32_t uni(uint32_t v) {  // a value all lanes hold: to a scalar register, so that what is computed from it runs on the scalar unit
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)lane));
}

struct BitIn {  // wave-uniform reader over dwords held in the lanes
    const uint32_t *in;  // dwords of the file buffer
    uint32_t wbase;      // dword index of winA's lane 0
    uint32_t winA, winB; // this lane's dwords: wbase + lane, wbase + 64 + lane
    __device__ __forceinline__ void init(const uint32_t *p, uint64_t bitpos, int lane) {
        in = p;
        wbase = (uint32_t)(bitpos >> 5) & ~63u;
        winA = in[wbase + lane];
        winB = in[wbase + 64 + lane];
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __device__ __forceinline__ uint32_t dw(uint32_t d) const {  // d relative to wbase, < 128
        return d < 64u ? rl(winA, d) : rl(winB, d - 64u);
    }
    // 64 bits of the stream at bit position bitpos (positions move forward only)
    __device__ __forceinline__ uint64_t fetch(uint64_t bitpos, int lane) {
        uint32_t d = (uint32_t)(bitpos >> 5) - wbase;
        if (d >= 64u) {  // the first register is used up: the second takes its place and is loaded again
            winA = winB;
            wbase += 64u;
            winB = in[wbase + 64 + lane];
            // waited for HERE, once per 256 bytes of input: left to the compiler, every later use of the registers
            // waits for vmcnt(0) -- that is, for all the global stores of the copies in between
            __builtin_amdgcn_s_waitcnt(0x0F70);
            d -= 64u;
        }
        const uint32_t lo = dw(d), mid = dw(d + 1), hi = dw(d + 2);
        const uint32_t sh = (uint32_t)bitpos & 31u;
        const uint64_t lm = ((uint64_t)mid << 32) | lo;
        return sh ? (lm >> sh) | ((uint64_t)hi << (64u - sh)) : lm;
    }
};

// counts, canonical symbol list, codes and root table of one alphabet (lens[0..n) in LDS); mode 0: the code-length
// code, 1: literals / lengths, 2: distances
__device__ __forceinline__ uint32_t entry_of(const Lds &S, int mode, uint32_t s, uint32_t l) {
    if (mode == 0) return l | (s << 8);
    if (mode == 2) return l | ((uint32_t)S.dext[s] << 4) | ((uint32_t)S.dbase[s] << 8) | (1u << 28);
    if (s < 256u) return l | (s << 8);
    if (s == 256u) return l | (2u << 28);
    return l | ((uint32_t)S.lext[s - 257u] << 4) | ((uint32_t)S.lbase[s - 257u] << 8) | (1u << 28);
}
__device__ void build(Lds &S, const uint8_t *lens, int n, uint16_t *count, uint16_t *symlist, uint32_t *root, int rootbits, int mode,
                      int lane) {
    LDS_ORDER();
    if (lane == 0) {
        uint16_t *offs = S.tmp, *next = S.tmp + 16;
        for (int l = 0; l < 16; l++) count[l] = 0;
        for (int s = 0; s < n; s++) count[lens[s]]++;
        count[0] = 0;
        uint32_t o = 0, c = 0;
        for (int l = 1; l < 16; l++) {
            offs[l] = (uint16_t)o;
            o += count[l];
            c = (c + count[l - 1]) << 1;
            next[l] = (uint16_t)c;
        }
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (l) {
                symlist[offs[l]++] = (uint16_t)s;
                S.code[s] = next[l]++;
            }
        }
    }
    for (int k = lane; k < (1 << rootbits); k += 64) root[k] = 0;
    LDS_ORDER();
    for (int s = lane; s < n; s += 64) {
        const int l = lens[s];
        if (l && l <= rootbits) {
            const uint32_t r = __builtin_bitreverse32((uint32_t)S.code[s]) >> (32 - l);
            const uint32_t e = entry_of(S, mode, (uint32_t)s, (uint32_t)l);
            for (uint32_t k = r; k < (1u << rootbits); k += 1u << l) root[k] = e;
        }
    }
    LDS_ORDER();
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

// the header of a dynamic block behind its three type bits: the code lengths of both alphabets into S.lens[0 .. nlit + ndist)
__device__ uint32_t parse_dynamic(Lds &S, BitIn &bi, uint64_t &bitpos, int lane, int &nlit, int &ndist) {
    uint64_t w = bi.fetch(bitpos, lane);
    nlit = (int)(w & 31) + 257;
    ndist = (int)((w >> 5) & 31) + 1;
    const int ncl = (int)((w >> 10) & 15) + 4;
    if (nlit > 286 || ndist > 30) return 4;
    bitpos += 14;
    LDS_ORDER();
    if (lane < 19) S.lens[lane] = 0;
    LDS_ORDER();
    for (int i = 0; i < ncl; i++) {  // (uniform; 3 bits each)
        if ((i & 15) == 0) w = bi.fetch(bitpos, lane);
        if (lane == 0) S.lens[CLORDER[i]] = (uint8_t)(w & 7);
        w >>= 3;
        bitpos += 3;
    }
    // the code-length code: all codes fit the 7-bit root
    build(S, S.lens, 19, S.lcount, S.lsym, S.clt, 7, 0, lane);
    int i = 0;
    uint32_t prev = 0;
    while (i < nlit + ndist) {
        w = bi.fetch(bitpos, lane);
        const uint32_t e = uni(S.clt[w & 127]);
        const uint32_t l = e & 15u, sym = e >> 8;
        if (l == 0) return 4;
        bitpos += l;
        w >>= l;
        uint32_t rep = 1, val = sym;
        if (sym == 16) {
            if (i == 0) return 4;
            rep = 3 + ((uint32_t)w & 3);
            bitpos += 2;
            val = prev;
        } else if (sym == 17) {
            rep = 3 + ((uint32_t)w & 7);
            bitpos += 3;
            val = 0;
        } else if (sym == 18) {
            rep = 11 + ((uint32_t)w & 127);
            bitpos += 7;
            val = 0;
        }
        if (i + (int)rep > nlit + ndist) return 4;
        for (uint32_t k = (uint32_t)lane; k < rep; k += 64) S.code[i + k] = (uint16_t)val;
        i += (int)rep;
        prev = val;
    }
    LDS_ORDER();
    for (int s = lane; s < nlit + ndist; s += 64) S.lens[s] = (uint8_t)S.code[s];
    LDS_ORDER();
    return 0;
}

__global__ __launch_bounds__(64) void k_inflate(const uint32_t *in, ChunkDesc *desc, uint16_t *out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 70 KB: more than a static array may have
    Lds &S = *(Lds *)smem;
    const int lane = (int)threadIdx.x;
    ChunkDesc &cd = desc[blockIdx.x];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // the unknown 32 KiB before the chunk as markers: the newest NEARSZ of them in the LDS ring (position q of the
    // stream, counted from 32768 before the chunk, lives at q & (NEARSZ - 1))
    for (uint32_t i = (uint32_t)lane; i < NEARSZ; i += 64) S.window[(WSIZE - NEARSZ + i) & (NEARSZ - 1)] = (uint16_t)(0x8000u | (WSIZE - NEARSZ + i));
    if (lane < 29) {
        S.lbase[lane] = LBASE[lane];
        S.lext[lane] = LEXT[lane];
    }
    if (lane < 30) {
        S.dbase[lane] = DBASE[lane];
        S.dext[lane] = DEXT[lane];
    }
    LDS_ORDER();
    uint64_t bitpos = cd.bit_start;
    const uint64_t bit_end = cd.bit_end;
    uint16_t *o = out + cd.out_start;
    uint32_t op = 0;  // symbols written; window position = op & (WSIZE - 1)
    const uint32_t cap = (uint32_t)(cd.out_end - cd.out_start);
    BitIn bi;
    bi.init(in, bitpos, lane);
    uint32_t status = 0, blocks = 0;
    while (bitpos < bit_end && status == 0) {
        uint64_t w = bi.fetch(bitpos, lane);
        const uint32_t bfinal = (uint32_t)w & 1u, btype = (uint32_t)(w >> 1) & 3u;
        bitpos += 3;
        blocks++;
        if (btype == 0) {  // stored
            bitpos = (bitpos + 7) & ~7ull;
            w = bi.fetch(bitpos, lane);
            const uint32_t len = (uint32_t)w & 0xFFFFu, nlen = (uint32_t)(w >> 16) & 0xFFFFu;
            if ((len ^ nlen) != 0xFFFFu) {
                status = 2;
                break;
            }
            bitpos += 32;
            if (op + len > cap) {
                status = 5;
                break;
            }
            const uint8_t *bytes = (const uint8_t *)in + (bitpos >> 3);
            for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                const uint16_t v = bytes[i];
                S.window[(op + WSIZE + i) & (NEARSZ - 1)] = v;
                o[op + i] = v;
            }
            LDS_ORDER();
            op += len;
            bitpos += 8ull * len;
            bi.init(in, bitpos, lane);
            if (bfinal) break;
            continue;
        }
        if (btype == 3) {
            status = 3;
            break;
        }
        int nlit, ndist;
        if (btype == 1) {  // fixed codes
            for (int s = lane; s < 288; s += 64) S.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) S.lens[288 + lane] = 5;
            nlit = 288;
            ndist = 30;
        } else {
            status = parse_dynamic(S, bi, bitpos, lane, nlit, ndist);
            if (status) break;
        }
        build(S, S.lens, nlit, S.lcount, S.lsym, S.lit, ROOT, 1, lane);
        build(S, S.lens + nlit, ndist, S.dcount, S.dsym, S.dist, DROOT, 2, lane);
        // ---- the symbols of the block
        for (;;) {
            w = bi.fetch(bitpos, lane);
            uint32_t e = uni(S.lit[w & ((1u << ROOT) - 1u)]);
            if (e == 0) {  // a code longer than the root
                uint32_t l;
                const uint32_t sym = slow_symbol(w, S.lcount, S.lsym, l);
                if (l == 0 || sym >= 286u) {
                    status = 6;
                    break;
                }
                e = entry_of(S, 1, sym, l);
            }
            const uint32_t l = e & 15u;
            bitpos += l;
            w >>= l;
            const uint32_t kind = e >> 28;
            if (kind == 0) {
                if (op >= cap) {
                    status = 5;
                    break;
                }
                const uint16_t v = (uint16_t)(e >> 8);
                if (lane == 0) {
                    S.window[(op + WSIZE) & (NEARSZ - 1)] = v;
                    o[op] = v;
                }
                op++;
                continue;
            }
            if (kind == 2) break;
            const uint32_t lext = (e >> 4) & 15u;
            const uint32_t len = ((e >> 8) & 0xFFFFu) + ((uint32_t)w & ((1u << lext) - 1u));
            bitpos += lext;
            w >>= lext;
            uint32_t de = uni(S.dist[w & ((1u << DROOT) - 1u)]);
            if (de == 0) {
                uint32_t dl;
                const uint32_t dsymv = slow_symbol(w, S.dcount, S.dsym, dl);
                if (dl == 0 || dsymv >= 30u) {
                    status = 8;
                    break;
                }
                de = entry_of(S, 2, dsymv, dl);
            }
            const uint32_t dl = de & 15u;
            bitpos += dl;
            w >>= dl;
            const uint32_t dext = (de >> 4) & 15u;
            const uint32_t dist = ((de >> 8) & 0xFFFFu) + ((uint32_t)w & ((1u << dext) - 1u));
            bitpos += dext;
            if (op + len > cap) {
                status = 5;
                break;
            }
            // the copy: every lane a symbol; with dist < len the pattern repeats.  (LDS operations of one wave execute
            // in order: the literal lane 0 wrote, the symbols of the last copy are there for this one)
            asm volatile("" ::: "memory");
            for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                const uint32_t from = dist >= len ? i : i % dist;
                uint16_t v;
                // (a ring slot is overwritten by the position NEARSZ later: a source this copy could reach with its own
                //  writes -- up to 258 symbols ahead -- is not taken from the ring)
                if (dist - from + 320u <= NEARSZ) {
                    v = S.window[(op + WSIZE - dist + from) & (NEARSZ - 1)];
                } else if (op + from >= dist) {  // older than the ring, inside the chunk: from the output (written long ago)
                    v = __hip_atomic_load(&o[op + from - dist], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {                         // older than the ring, before the chunk: the marker itself
                    v = (uint16_t)(0x8000u | (op + WSIZE - dist + from));
                }
                o[op + i] = v;
                S.window[(op + WSIZE + i) & (NEARSZ - 1)] = v;
            }
            asm volatile("" ::: "memory");
            op += len;
        }
        if (bfinal) break;
    }
    if (lane == 0) {
        cd.status = status ? status : (op == cap ? 0u : 100u);
        cd.blocks = blocks;
        cd.cycles = __builtin_amdgcn_s_memtime() - t0;
    }
}

// ================================================================================================================
// The whole reader on the device ("full" mode): block search per chunk, decode, the window chain, marker replacement.
// ================================================================================================================
constexpr uint64_t NONE = ~0ull;

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

// Does a non-final dynamic block header start at bit `cand`?  The seam test of the host reader: the code lengths must
// decode, both codes must be complete (the distance code may have a single symbol), end-of-block must have a code.
__device__ bool plausible_block(Lds &S, const uint32_t *in, uint64_t cand, int lane) {
    BitIn bi;
    bi.init(in, cand, lane);
    uint64_t bp = cand + 3;
    int nlit, ndist;
    if (parse_dynamic(S, bi, bp, lane, nlit, ndist)) return false;
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
    if (!(S.lens[256] != 0 && sl == 32768u && (sd == 32768u || cd <= 1u))) return false;
    // a header that passes by chance decodes into nonsense soon: the first tokens of the block are walked as well
    // (the host reader gets there by running into the error and searching on)
    build(S, S.lens, nlit, S.lcount, S.lsym, S.lit, ROOT, 1, lane);
    build(S, S.lens + nlit, ndist, S.dcount, S.dsym, S.dist, DROOT, 2, lane);
    uint32_t produced = 0;
    for (int tok = 0; tok < 256; tok++) {
        uint64_t w = bi.fetch(bp, lane);
        uint32_t e = uni(S.lit[w & ((1u << ROOT) - 1u)]);
        if (e == 0) {
            uint32_t l;
            const uint32_t sym = slow_symbol(w, S.lcount, S.lsym, l);
            if (l == 0 || sym >= 286u) return false;
            e = entry_of(S, 1, sym, l);
        }
        const uint32_t l = e & 15u;
        bp += l;
        w >>= l;
        const uint32_t kind = e >> 28;
        if (kind == 2) break;
        if (kind == 0) {
            produced++;
            continue;
        }
        const uint32_t lext = (e >> 4) & 15u;
        const uint32_t len = ((e >> 8) & 0xFFFFu) + ((uint32_t)w & ((1u << lext) - 1u));
        bp += lext;
        w >>= lext;
        uint32_t de = uni(S.dist[w & ((1u << DROOT) - 1u)]);
        if (de == 0) {
            uint32_t dl;
            const uint32_t dsymv = slow_symbol(w, S.dcount, S.dsym, dl);
            if (dl == 0 || dsymv >= 30u) return false;
            de = entry_of(S, 2, dsymv, dl);
        }
        const uint32_t dext = (de >> 4) & 15u;
        const uint32_t dist = ((de >> 8) & 0xFFFFu) + ((uint32_t)(w >> (de & 15u)) & ((1u << dext) - 1u));
        if (dist > produced + WSIZE) return false;
        bp += (de & 15u) + dext;
        produced += len;
    }
    return true;
}

// start[c] = the first bit position in chunk c's stretch of the file that passes the seam test (NONE: none)
__global__ __launch_bounds__(64) void k_search(const uint32_t *in, uint64_t total_bits, uint64_t chunk_bits, uint64_t first_bit,
                                               uint64_t *start) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Lds &S = *(Lds *)smem;
    const int lane = (int)threadIdx.x;
    const uint32_t c = blockIdx.x;
    if (c == 0) {
        if (lane == 0) start[0] = first_bit;
        return;
    }
    if (lane < 29) {  // (the trial walk reads the base / extra-bit tables through entry_of)
        S.lbase[lane] = LBASE[lane];
        S.lext[lane] = LEXT[lane];
    }
    if (lane < 30) {
        S.dbase[lane] = DBASE[lane];
        S.dext[lane] = DEXT[lane];
    }
    LDS_ORDER();
    uint64_t from = (uint64_t)c * chunk_bits, to = from + chunk_bits;
    if (from <= first_bit) from = first_bit + 1;
    if (to + 160 > total_bits) to = total_bits > 160 ? total_bits - 160 : 0;
    const uint8_t *bytes = (const uint8_t *)in;
    uint64_t found = NONE;
    for (uint64_t b0 = from; b0 < to && found == NONE; b0 += 64) {
        const uint64_t b = b0 + (uint64_t)lane;
        const uint64_t w = bits_at(bytes, b), w2 = bits_at(bytes + 7, b);  // w2: bits 56.. of the window
        bool pre = b < to && (w & 7) == 4 && ((w >> 3) & 31) <= 29 && ((w >> 8) & 31) <= 29;
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
        pre = pre && any && left == 0;
        uint64_t m = __ballot(pre);
        while (m && found == NONE) {
            const uint64_t cand = b0 + (uint64_t)__builtin_ctzll(m);
            if (plausible_block(S, in, cand, lane)) found = cand;
            m &= m - 1;
        }
    }
    if (lane == 0) start[c] = found;
}

// the chunks in order: does c end where c + 1 starts?  the 32 KiB behind c, markers replaced, are c + 1's window
__global__ __launch_bounds__(1024) void k_chain(ChunkDesc *d, uint32_t n, const uint16_t *sym, uint8_t *windows, uint32_t *broken) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *w = smem, *nw = smem + WSIZE;  // this chunk's window and the next one's
    const uint32_t t = threadIdx.x;
    for (uint32_t j = t; j < WSIZE; j += 1024) w[j] = 0;
    __syncthreads();
    for (uint32_t c = 0; c < n; c++) {
        for (uint32_t j = t; j < WSIZE / 16; j += 1024) ((uint4 *)(windows + (size_t)c * WSIZE))[j] = ((const uint4 *)w)[j];
        if (c + 1 < n && (d[c].status != 0 || d[c].bit_end != d[c + 1].bit_start)) {
            if (t == 0) *broken = c + 1;
            return;
        }
        const uint32_t nsym = (uint32_t)(d[c].out_end - d[c].out_start);
        const uint16_t *src = sym + d[c].out_start;
        for (uint32_t j = t; j < WSIZE; j += 1024) {
            uint8_t v;
            if (nsym >= WSIZE) {
                const uint16_t x = src[nsym - WSIZE + j];
                v = x & 0x8000u ? w[x & 0x7FFFu] : (uint8_t)x;
            } else if (j < WSIZE - nsym) {
                v = w[j + nsym];
            } else {
                const uint16_t x = src[j - (WSIZE - nsym)];
                v = x & 0x8000u ? w[x & 0x7FFFu] : (uint8_t)x;
            }
            nw[j] = v;
        }
        __syncthreads();
        uint8_t *sw = w;
        w = nw;
        nw = sw;
    }
    if (t == 0) *broken = 0;
}

// ---- the same windows without the serial walk: a chunk's effect on the window is an index map (a byte of the next
// window is a literal, or the byte at some index of this one: 0x8000 | index, the symbols' own form), maps compose
// associatively, and a parallel prefix scan over the chunks' maps gives every chunk's window.
__global__ __launch_bounds__(256) void k_maps(const ChunkDesc *d, const uint16_t *sym, uint16_t *maps) {
    const uint32_t c = blockIdx.x;
    const uint32_t nsym = (uint32_t)(d[c].out_end - d[c].out_start);
    const uint16_t *src = sym + d[c].out_start;
    uint16_t *m = maps + (size_t)c * WSIZE;
    for (uint32_t j = blockIdx.y * 256 + threadIdx.x; j < WSIZE; j += gridDim.y * 256) {
        uint16_t v;
        if (nsym >= WSIZE) v = src[nsym - WSIZE + j];
        else if (j < WSIZE - nsym) v = (uint16_t)(0x8000u | (j + nsym));  // the old window moves up
        else v = src[j - (WSIZE - nsym)];
        m[j] = v;
    }
}
// one round of the scan: dst[c] = src[c] o src[c - stride] (first through the earlier map, then through c's)
__global__ __launch_bounds__(256) void k_scan_round(const uint16_t *src, uint16_t *dst, uint32_t n, uint32_t stride) {
    const uint32_t c = blockIdx.x;
    const uint16_t *b = src + (size_t)c * WSIZE;
    uint16_t *o = dst + (size_t)c * WSIZE;
    if (c < stride) {
        for (uint32_t j = blockIdx.y * 256 + threadIdx.x; j < WSIZE; j += gridDim.y * 256) o[j] = b[j];
        return;
    }
    const uint16_t *a = src + (size_t)(c - stride) * WSIZE;
    for (uint32_t j = blockIdx.y * 256 + threadIdx.x; j < WSIZE; j += gridDim.y * 256) {
        const uint16_t v = b[j];
        o[j] = v & 0x8000u ? a[v & 0x7FFFu] : v;
    }
}
// windows[c + 1] = the scanned map of chunk c applied to the window before chunk 0 (zeros); windows[0] = that window
__global__ __launch_bounds__(256) void k_windows(const uint16_t *scanned, uint32_t n, uint8_t *windows) {
    const uint32_t c = blockIdx.x;  // the window of chunk c
    uint8_t *w = windows + (size_t)c * WSIZE;
    for (uint32_t j = blockIdx.y * 256 + threadIdx.x; j < WSIZE; j += gridDim.y * 256) {
        uint8_t v = 0;
        if (c > 0) {
            const uint16_t x = scanned[(size_t)(c - 1) * WSIZE + j];
            v = x & 0x8000u ? 0 : (uint8_t)x;
        }
        w[j] = v;
    }
}
__global__ void k_seams(const ChunkDesc *d, uint32_t n, uint32_t *broken) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c + 1 < n && (d[c].status != 0 || d[c].bit_end != d[c + 1].bit_start)) atomicMin(broken, c + 1);
    if (c + 1 == n && d[c].status != 0) atomicMin(broken, c + 1);
}

// every chunk's symbols become bytes at their place in the text
__global__ __launch_bounds__(256) void k_resolve(const ChunkDesc *d, const uint16_t *sym, const uint8_t *windows, const uint64_t *text_off,
                                                 uint8_t *text) {
    const uint32_t c = blockIdx.x;
    const uint32_t nsym = (uint32_t)(d[c].out_end - d[c].out_start);
    const uint16_t *src = sym + d[c].out_start;
    const uint8_t *w = windows + (size_t)c * WSIZE;
    uint8_t *dst = text + text_off[c];
    for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < nsym; i += gridDim.y * 256) {
        const uint16_t x = src[i];
        dst[i] = x & 0x8000u ? w[x & 0x7FFFu] : (uint8_t)x;
    }
}

// decode kernel of the full mode: k_inflate, but the chunk's end and output length are results (bit_end on entry = where
// to stop: the first block boundary at or behind it)
__global__ __launch_bounds__(64) void k_inflate_open(const uint32_t *in, ChunkDesc *desc, uint16_t *out);

static int full_mode(const std::vector<uint8_t> &gz, const std::vector<uint8_t> &text, uint64_t first_bit, uint64_t end_bit, size_t chunk) {
    const uint64_t chunk_bits = 8 * (uint64_t)chunk;
    const uint32_t nchunk = (uint32_t)((end_bit + chunk_bits - 1) / chunk_bits);
    uint32_t *d_in;
    uint64_t *d_start;
    CK(hipMalloc(&d_in, gz.size() + 4096));
    CK(hipMemset(d_in, 0, gz.size() + 4096));
    CK(hipMemcpy(d_in, gz.data(), gz.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_start, nchunk * sizeof(uint64_t)));
    CK(hipFuncSetAttribute((const void *)k_search, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds)));
    CK(hipFuncSetAttribute((const void *)k_inflate_open, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds)));
    CK(hipFuncSetAttribute((const void *)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * WSIZE)));
    hipEvent_t ev[6];
    for (auto &evt : ev) CK(hipEventCreate(&evt));
    float t_search = 1e30f, t_decode = 1e30f, t_chain = 1e30f, t_resolve = 1e30f, t_scan = 0;
    std::vector<uint64_t> start(nchunk);
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(ev[0]));
        hipLaunchKernelGGL(k_search, dim3(nchunk), dim3(64), sizeof(Lds), 0, d_in, end_bit, chunk_bits, first_bit, d_start);
        CK(hipEventRecord(ev[1]));
        CK(hipEventSynchronize(ev[1]));
        float ms;
        CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
        if (ms < t_search) t_search = ms;
    }
    CK(hipMemcpy(start.data(), d_start, nchunk * sizeof(uint64_t), hipMemcpyDeviceToHost));
    // the chunks that have a start; each runs to the next one's start
    std::vector<ChunkDesc> cd;
    for (uint32_t c = 0; c < nchunk; c++)
        if (start[c] != NONE) {
            ChunkDesc x{};
            x.bit_start = start[c];
            cd.push_back(x);
        }
    uint64_t slots = 0;
    for (size_t k = 0; k < cd.size(); k++) {
        cd[k].bit_end = k + 1 < cd.size() ? cd[k + 1].bit_start : end_bit;
        const uint64_t cap = (cd[k].bit_end - cd[k].bit_start) / 8 * 16 + 65536;  // symbols this chunk may write
        cd[k].out_start = slots;
        cd[k].out_end = slots + cap;
        slots += cap;
    }
    printf("search: %u stretches of %zu KiB, %zu block starts found; %.1f M symbols of room\n", nchunk, chunk >> 10, cd.size(), slots / 1e6);
    ChunkDesc *d_desc;
    uint16_t *d_sym;
    uint8_t *d_windows, *d_text;
    uint64_t *d_toff;
    uint32_t *d_broken;
    CK(hipMalloc(&d_desc, cd.size() * sizeof(ChunkDesc)));
    CK(hipMalloc(&d_sym, (slots + 64) * 2));
    CK(hipMalloc(&d_windows, cd.size() * (size_t)WSIZE));
    CK(hipMalloc(&d_text, text.size() + 64));
    CK(hipMalloc(&d_toff, cd.size() * sizeof(uint64_t)));
    CK(hipMalloc(&d_broken, 4));
    std::vector<ChunkDesc> res(cd.size());
    std::vector<uint64_t> toff(cd.size());
    uint32_t broken = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemcpy(d_desc, cd.data(), cd.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice));
        CK(hipEventRecord(ev[0]));
        hipLaunchKernelGGL(k_inflate_open, dim3((unsigned)cd.size()), dim3(64), sizeof(Lds), 0, d_in, d_desc, d_sym);
        CK(hipEventRecord(ev[1]));
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(1024), 2 * WSIZE, 0, d_desc, (uint32_t)cd.size(), d_sym, d_windows, d_broken);
        CK(hipEventRecord(ev[2]));
        CK(hipEventSynchronize(ev[2]));
        if (rep == 1) {  // the windows once more by the scan over the chunks' index maps: same bytes, parallel
            const uint32_t n = (uint32_t)cd.size();
            uint16_t *d_m0, *d_m1;
            uint8_t *d_w2;
            uint32_t *d_b2;
            CK(hipMalloc(&d_m0, (size_t)n * WSIZE * 2));
            CK(hipMalloc(&d_m1, (size_t)n * WSIZE * 2));
            CK(hipMalloc(&d_w2, (size_t)n * WSIZE));
            CK(hipMalloc(&d_b2, 4));
            float best_scan = 1e30f;
            for (int r2 = 0; r2 < 2; r2++) {
                const uint32_t big = 0xFFFFFFFFu;
                CK(hipMemcpy(d_b2, &big, 4, hipMemcpyHostToDevice));
                CK(hipEventRecord(ev[3]));
                hipLaunchKernelGGL(k_seams, dim3((n + 255) / 256), dim3(256), 0, 0, d_desc, n, d_b2);
                hipLaunchKernelGGL(k_maps, dim3(n, 4), dim3(256), 0, 0, d_desc, d_sym, d_m0);
                uint16_t *a = d_m0, *b = d_m1;
                for (uint32_t stride = 1; stride < n; stride <<= 1) {
                    hipLaunchKernelGGL(k_scan_round, dim3(n, 4), dim3(256), 0, 0, a, b, n, stride);
                    uint16_t *t = a;
                    a = b;
                    b = t;
                }
                hipLaunchKernelGGL(k_windows, dim3(n, 4), dim3(256), 0, 0, a, n, d_w2);
                CK(hipEventRecord(ev[4]));
                CK(hipEventSynchronize(ev[4]));
                float ms;
                CK(hipEventElapsedTime(&ms, ev[3], ev[4]));
                if (ms < best_scan) best_scan = ms;
            }
            std::vector<uint8_t> w1((size_t)n * WSIZE), w2((size_t)n * WSIZE);
            CK(hipMemcpy(w1.data(), d_windows, w1.size(), hipMemcpyDeviceToHost));
            CK(hipMemcpy(w2.data(), d_w2, w2.size(), hipMemcpyDeviceToHost));
            uint32_t b2 = 0;
            CK(hipMemcpy(&b2, d_b2, 4, hipMemcpyDeviceToHost));
            printf("windows by a prefix scan over the chunks' index maps: %.2f ms (the serial walk: %.2f ms), %s, seams %s\n", best_scan, t_chain,
                   w1 == w2 ? "the same bytes" : "DIFFERENT", b2 == 0xFFFFFFFFu ? "whole" : "BROKEN");
            t_scan = best_scan;
            CK(hipFree(d_m0));
            CK(hipFree(d_m1));
            CK(hipFree(d_w2));
            CK(hipFree(d_b2));
        }
        // (the text offsets: a prefix sum over the chunks' lengths -- on the host here, a scan kernel in the real thing)
        CK(hipMemcpy(res.data(), d_desc, cd.size() * sizeof(ChunkDesc), hipMemcpyDeviceToHost));
        uint64_t acc = 0;
        for (size_t k = 0; k < res.size(); k++) {
            toff[k] = acc;
            acc += res[k].out_end - res[k].out_start;
        }
        if (acc != text.size()) fprintf(stderr, "lengths add up to %llu, the text has %zu bytes\n", (unsigned long long)acc, text.size());
        CK(hipMemcpy(d_toff, toff.data(), toff.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        CK(hipEventRecord(ev[3]));
        hipLaunchKernelGGL(k_resolve, dim3((unsigned)cd.size(), 8), dim3(256), 0, 0, d_desc, d_sym, d_windows, d_toff, d_text);
        CK(hipEventRecord(ev[4]));
        CK(hipEventSynchronize(ev[4]));
        float a, b, c2;
        CK(hipEventElapsedTime(&a, ev[0], ev[1]));
        CK(hipEventElapsedTime(&b, ev[1], ev[2]));
        if (rep == 1) b = t_chain;  // (ev[3] was reused by the scan above: keep the first round's chain time)
        CK(hipEventElapsedTime(&c2, ev[3], ev[4]));
        if (a < t_decode) t_decode = a;
        if (b < t_chain) t_chain = b;
        if (c2 < t_resolve) t_resolve = c2;
        CK(hipMemcpy(&broken, d_broken, 4, hipMemcpyDeviceToHost));
    }
    size_t bad_status = 0;
    for (size_t k = 0; k < res.size(); k++)
        if (res[k].status) {
            if (bad_status < 4)
                fprintf(stderr, "chunk %zu of %zu: status %u after %u blocks, bits %llu .. %llu (asked to stop at %llu), %llu symbols of %llu\n", k,
                        res.size(), res[k].status, res[k].blocks, (unsigned long long)cd[k].bit_start, (unsigned long long)res[k].bit_end,
                        (unsigned long long)cd[k].bit_end, (unsigned long long)(res[k].out_end - res[k].out_start),
                        (unsigned long long)(cd[k].out_end - cd[k].out_start));
            bad_status++;
        }
    std::vector<uint8_t> back(text.size());
    CK(hipMemcpy(back.data(), d_text, text.size(), hipMemcpyDeviceToHost));
    const bool same = broken == 0 && bad_status == 0 && memcmp(back.data(), text.data(), text.size()) == 0;
    const double tot = t_search + t_decode + t_chain + t_resolve;
    printf("search %.2f ms, decode %.2f ms, chain %.2f ms, resolve %.2f ms: %.2f ms = %.2f GB/s of text (%.2f GB/s of gzip)\n", t_search, t_decode,
           t_chain, t_resolve, tot, text.size() / (tot * 1e-3) / 1e9, gz.size() / (tot * 1e-3) / 1e9);
    const double tot2 = t_search + t_decode + t_scan + t_resolve;
    printf("with the scan instead of the walk: %.2f ms = %.2f GB/s of text\n", tot2, text.size() / (tot2 * 1e-3) / 1e9);
    printf("check: %zu chunks with an error status, chain %s, text %s\n", bad_status, broken ? "BROKEN" : "whole", same ? "identical to zlib's" : "DIFFERS");
    return same ? 0 : 1;
}

__global__ __launch_bounds__(64) void k_inflate_open(const uint32_t *in, ChunkDesc *desc, uint16_t *out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 70 KB: more than a static array may have
    Lds &S = *(Lds *)smem;
    const int lane = (int)threadIdx.x;
    ChunkDesc &cd = desc[blockIdx.x];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // the unknown 32 KiB before the chunk as markers: the newest NEARSZ of them in the LDS ring (position q of the
    // stream, counted from 32768 before the chunk, lives at q & (NEARSZ - 1))
    for (uint32_t i = (uint32_t)lane; i < NEARSZ; i += 64) S.window[(WSIZE - NEARSZ + i) & (NEARSZ - 1)] = (uint16_t)(0x8000u | (WSIZE - NEARSZ + i));
    if (lane < 29) {
        S.lbase[lane] = LBASE[lane];
        S.lext[lane] = LEXT[lane];
    }
    if (lane < 30) {
        S.dbase[lane] = DBASE[lane];
        S.dext[lane] = DEXT[lane];
    }
    LDS_ORDER();
    uint64_t bitpos = cd.bit_start;
    const uint64_t bit_end = cd.bit_end;
    uint16_t *o = out + cd.out_start;
    uint32_t op = 0;  // symbols written; window position = op & (WSIZE - 1)
    const uint32_t cap = (uint32_t)(cd.out_end - cd.out_start);
    BitIn bi;
    bi.init(in, bitpos, lane);
    uint32_t status = 0, blocks = 0;
    while (bitpos < bit_end && status == 0) {
        uint64_t w = bi.fetch(bitpos, lane);
        const uint32_t bfinal = (uint32_t)w & 1u, btype = (uint32_t)(w >> 1) & 3u;
        bitpos += 3;
        blocks++;
        if (btype == 0) {  // stored
            bitpos = (bitpos + 7) & ~7ull;
            w = bi.fetch(bitpos, lane);
            const uint32_t len = (uint32_t)w & 0xFFFFu, nlen = (uint32_t)(w >> 16) & 0xFFFFu;
            if ((len ^ nlen) != 0xFFFFu) {
                status = 2;
                break;
            }
            bitpos += 32;
            if (op + len > cap) {
                status = 5;
                break;
            }
            const uint8_t *bytes = (const uint8_t *)in + (bitpos >> 3);
            for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                const uint16_t v = bytes[i];
                S.window[(op + WSIZE + i) & (NEARSZ - 1)] = v;
                o[op + i] = v;
            }
            LDS_ORDER();
            op += len;
            bitpos += 8ull * len;
            bi.init(in, bitpos, lane);
            if (bfinal) break;
            continue;
        }
        if (btype == 3) {
            status = 3;
            break;
        }
        int nlit, ndist;
        if (btype == 1) {  // fixed codes
            for (int s = lane; s < 288; s += 64) S.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) S.lens[288 + lane] = 5;
            nlit = 288;
            ndist = 30;
        } else {
            status = parse_dynamic(S, bi, bitpos, lane, nlit, ndist);
            if (status) break;
        }
        build(S, S.lens, nlit, S.lcount, S.lsym, S.lit, ROOT, 1, lane);
        build(S, S.lens + nlit, ndist, S.dcount, S.dsym, S.dist, DROOT, 2, lane);
        // ---- the symbols of the block
        for (;;) {
            w = bi.fetch(bitpos, lane);
            uint32_t e = uni(S.lit[w & ((1u << ROOT) - 1u)]);
            if (e == 0) {  // a code longer than the root
                uint32_t l;
                const uint32_t sym = slow_symbol(w, S.lcount, S.lsym, l);
                if (l == 0 || sym >= 286u) {
                    status = 6;
                    break;
                }
                e = entry_of(S, 1, sym, l);
            }
            const uint32_t l = e & 15u;
            bitpos += l;
            w >>= l;
            const uint32_t kind = e >> 28;
            if (kind == 0) {
                if (op >= cap) {
                    status = 5;
                    break;
                }
                const uint16_t v = (uint16_t)(e >> 8);
                if (lane == 0) {
                    S.window[(op + WSIZE) & (NEARSZ - 1)] = v;
                    o[op] = v;
                }
                op++;
                continue;
            }
            if (kind == 2) break;
            const uint32_t lext = (e >> 4) & 15u;
            const uint32_t len = ((e >> 8) & 0xFFFFu) + ((uint32_t)w & ((1u << lext) - 1u));
            bitpos += lext;
            w >>= lext;
            uint32_t de = uni(S.dist[w & ((1u << DROOT) - 1u)]);
            if (de == 0) {
                uint32_t dl;
                const uint32_t dsymv = slow_symbol(w, S.dcount, S.dsym, dl);
                if (dl == 0 || dsymv >= 30u) {
                    status = 8;
                    break;
                }
                de = entry_of(S, 2, dsymv, dl);
            }
            const uint32_t dl = de & 15u;
            bitpos += dl;
            w >>= dl;
            const uint32_t dext = (de >> 4) & 15u;
            const uint32_t dist = ((de >> 8) & 0xFFFFu) + ((uint32_t)w & ((1u << dext) - 1u));
            bitpos += dext;
            if (op + len > cap) {
                status = 5;
                break;
            }
            // the copy: every lane a symbol; with dist < len the pattern repeats.  (LDS operations of one wave execute
            // in order: the literal lane 0 wrote, the symbols of the last copy are there for this one)
            asm volatile("" ::: "memory");
            for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                const uint32_t from = dist >= len ? i : i % dist;
                uint16_t v;
                // (a ring slot is overwritten by the position NEARSZ later: a source this copy could reach with its own
                //  writes -- up to 258 symbols ahead -- is not taken from the ring)
                if (dist - from + 320u <= NEARSZ) {
                    v = S.window[(op + WSIZE - dist + from) & (NEARSZ - 1)];
                } else if (op + from >= dist) {  // older than the ring, inside the chunk: from the output (written long ago)
                    v = __hip_atomic_load(&o[op + from - dist], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {                         // older than the ring, before the chunk: the marker itself
                    v = (uint16_t)(0x8000u | (op + WSIZE - dist + from));
                }
                o[op + i] = v;
                S.window[(op + WSIZE + i) & (NEARSZ - 1)] = v;
            }
            asm volatile("" ::: "memory");
            op += len;
        }
        if (bfinal) break;
    }
    if (lane == 0) {  // where the chunk ended and how much it wrote are results here
        cd.status = status;
        cd.blocks = blocks;
        cd.bit_end = bitpos;
        cd.out_end = cd.out_start + op;
        cd.cycles = __builtin_amdgcn_s_memtime() - t0;
    }
}


int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: gpu_inflate_proto file.gz [chunk KiB]\n");
        return 2;
    }
    const size_t chunk = (argc > 2 ? (size_t)atol(argv[2]) : 4096) << 10;
    const bool full = argc > 3 && !strcmp(argv[3], "full");
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<uint8_t> gz;
    {
        std::vector<uint8_t> buf(1 << 22);
        size_t r;
        while ((r = fread(buf.data(), 1, buf.size(), f)) > 0) gz.insert(gz.end(), buf.begin(), buf.begin() + r);
    }
    fclose(f);
    // ---- zlib: the text and the block boundaries
    std::vector<uint8_t> text;
    struct Boundary {
        uint64_t bit, out;
    };
    std::vector<Boundary> bounds;
    {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, 31) != Z_OK) return 1;
        text.resize(gz.size() * 8 + (1 << 20));
        zs.next_in = gz.data();
        zs.avail_in = (uInt)gz.size();
        zs.next_out = text.data();
        zs.avail_out = (uInt)text.size();
        for (;;) {
            if (zs.avail_out < (1u << 20)) {
                const size_t used = text.size() - zs.avail_out;
                text.resize(text.size() * 2);
                zs.next_out = text.data() + used;
                zs.avail_out = (uInt)(text.size() - used);
            }
            const int rc = inflate(&zs, Z_BLOCK);
            if (rc != Z_OK && rc != Z_STREAM_END) {
                fprintf(stderr, "zlib: %d\n", rc);
                return 1;
            }
            if ((zs.data_type & 128) && !(zs.data_type & 64) && rc == Z_OK)
                bounds.push_back({(uint64_t)zs.total_in * 8 - (uint64_t)(zs.data_type & 7), (uint64_t)zs.total_out});
            if (rc == Z_STREAM_END) break;
        }
        text.resize(zs.total_out);
        // the end of the deflate data: 8 bytes of trailer behind it
        bounds.push_back({((uint64_t)zs.total_in - 8) * 8, (uint64_t)zs.total_out});
        inflateEnd(&zs);
    }
    if (full) return full_mode(gz, text, bounds.front().bit, bounds.back().bit, chunk);
    // (a boundary recorded while the LAST block is being decoded is flagged 64 and skipped above, so the last pair is
    //  the start of the final block or of the block before it; the chunk that holds it runs to the end of the data)
    std::vector<ChunkDesc> chunks;
    {
        size_t i = 0;
        while (i + 1 < bounds.size()) {
            size_t j = i + 1;
            while (j + 1 < bounds.size() && bounds[j].bit < bounds[i].bit + 8 * (uint64_t)chunk) j++;
            ChunkDesc c{};
            c.bit_start = bounds[i].bit;
            c.bit_end = bounds[j].bit;
            c.out_start = bounds[i].out;
            c.out_end = bounds[j].out;
            chunks.push_back(c);
            i = j;
        }
    }
    printf("%zu bytes of gzip, %zu of text (%.2f : 1), %zu blocks, %zu chunks of ~%zu KiB\n", gz.size(), text.size(),
           (double)text.size() / gz.size(), bounds.size() - 1, chunks.size(), chunk >> 10);
    uint32_t *d_in;
    ChunkDesc *d_desc;
    uint16_t *d_out;
    CK(hipMalloc(&d_in, gz.size() + 4096));
    CK(hipMemset(d_in, 0, gz.size() + 4096));
    CK(hipMemcpy(d_in, gz.data(), gz.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_desc, chunks.size() * sizeof(ChunkDesc)));
    CK(hipMalloc(&d_out, (text.size() + 64) * 2));
    CK(hipFuncSetAttribute((const void *)k_inflate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemcpy(d_desc, chunks.data(), chunks.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_inflate, dim3((unsigned)chunks.size()), dim3(64), sizeof(Lds), 0, d_in, d_desc, d_out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<ChunkDesc> res(chunks.size());
    CK(hipMemcpy(res.data(), d_desc, chunks.size() * sizeof(ChunkDesc), hipMemcpyDeviceToHost));
    std::vector<uint16_t> sym(text.size());
    CK(hipMemcpy(sym.data(), d_out, text.size() * 2, hipMemcpyDeviceToHost));
    size_t bad_chunks = 0, bad_bytes = 0, markers = 0;
    double cyc = 0;
    for (size_t c = 0; c < res.size(); c++) {
        if (res[c].status) {
            if (bad_chunks < 5) fprintf(stderr, "chunk %zu: status %u after %u blocks\n", c, res[c].status, res[c].blocks);
            bad_chunks++;
        }
        cyc += (double)res[c].cycles;
        const uint64_t a = res[c].out_start, b = res[c].out_end;
        for (uint64_t p = a; p < b; p++) {
            const uint16_t v = sym[p];
            uint8_t got;
            if (v & 0x8000u) {
                markers++;
                const uint64_t idx = v & 0x7FFFu;  // position a + idx - 32768 of the text
                got = a + idx >= WSIZE ? text[a + idx - WSIZE] : 0;
            } else {
                got = (uint8_t)v;
            }
            bad_bytes += got != text[p];
        }
    }
    printf("kernel %.3f ms for %zu chunks = %.2f GB/s of text (%.2f GB/s of gzip); a chunk's wave %.0f M cycles on average\n", best,
           chunks.size(), text.size() / (best * 1e-3) / 1e9, gz.size() / (best * 1e-3) / 1e9, cyc / res.size() / 1e6);
    printf("check: %zu chunks with an error status, %zu of %zu bytes differ after the markers were replaced (%.1f %% of the symbols are markers)\n",
           bad_chunks, bad_bytes, text.size(), 100.0 * markers / (double)text.size());
    return bad_chunks || bad_bytes ? 1 : 0;
}
