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
//
//   hipcc --offload-arch=gfx950 -O3 tools/gpu_inflate_proto.hip -o tools/gpu_inflate_proto -lz
//   ./gpu_inflate_proto file.gz [chunk KiB = 4096]
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
            w = bi.fetch(bitpos, lane);
            nlit = (int)(w & 31) + 257;
            ndist = (int)((w >> 5) & 31) + 1;
            const int ncl = (int)((w >> 10) & 15) + 4;
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
                if (l == 0) {
                    status = 4;
                    break;
                }
                bitpos += l;
                w >>= l;
                uint32_t rep = 1, val = sym;
                if (sym == 16) {
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
                if (i + (int)rep > nlit + ndist) {
                    status = 4;
                    break;
                }
                // lengths go to lens[32 ...) while the code-length code's own lengths occupy lens[0..19)
                for (uint32_t k = (uint32_t)lane; k < rep; k += 64) S.code[i + k] = (uint16_t)val;
                i += (int)rep;
                prev = val;
            }
            if (status) break;
            LDS_ORDER();
            for (int s = lane; s < nlit + ndist; s += 64) S.lens[s] = (uint8_t)S.code[s];
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

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: gpu_inflate_proto file.gz [chunk KiB]\n");
        return 2;
    }
    const size_t chunk = (argc > 2 ? (size_t)atol(argv[2]) : 4096) << 10;
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
