// valu_bench.hip -- calibration microbenchmarks for the instruction-issue model of one MI355X CU:
//   (1) how fast do s_memtime (clock64) and s_memrealtime (wall_clock64) tick in wall time?
//   (2) how many independent 32-bit VALU / 64-bit-shift / 32-bit integer multiply wave-instructions
//       per second does the chip issue at 4..8 waves per SIMD?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_clocks(long long *out, int spin) {
    long long c0 = clock64(), w0 = wall_clock64();
    uint32_t x = threadIdx.x;
    for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = x; }
}

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(uint32_t *sink, int iters) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint64_t b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {  // 8 independent v_xor/v_add chains: 16 VALU per iteration
            a0 = (a0 ^ 0x9E3779B9u) + a1; a1 = (a1 ^ 0x7F4A7C15u) + a2; a2 = (a2 ^ 0x85EBCA6Bu) + a3; a3 = (a3 ^ 0xC2B2AE35u) + a4;
            a4 = (a4 ^ 0x27D4EB2Fu) + a5; a5 = (a5 ^ 0x165667B1u) + a6; a6 = (a6 ^ 0xD3A2646Cu) + a7; a7 = (a7 ^ 0xFD7046C5u) + a0;
        } else if (KIND == 1) {  // 8 v_mul_lo_u32
            a0 *= 0x9E3779B9u; a1 *= 0x7F4A7C15u; a2 *= 0x85EBCA6Bu; a3 *= 0xC2B2AE35u;
            a4 *= 0x27D4EB2Fu; a5 *= 0x165667B1u; a6 *= 0xD3A2646Cu; a7 *= 0xFD7046C5u;
        } else {  // 4 x (64-bit shift + 64-bit xor): v_lshrrev_b64 + 2 v_xor
            b0 ^= b0 >> 7; b1 ^= b1 >> 9; b2 ^= b2 >> 11; b3 ^= b3 >> 13;
        }
    }
    uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(b0 ^ b1 ^ b2 ^ b3);
    if (r == 0x12345) sink[0] = r;
}

template <int KIND>
static void run(const char *name, int per_iter, int blocks_per_cu, uint32_t *sink) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_valu<KIND>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, sink, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_valu<KIND>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, sink, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double winstr = (double)256 * blocks_per_cu * 4 * iters * per_iter;  // wave-instructions
    printf("%-22s %d waves/SIMD: %.3f ms, %.2f G wave-instr/s per CU\n", name, blocks_per_cu, ms, winstr / ms / 1e6 / 256);
}

int main() {
    long long *d, h[3];
    uint32_t *sink;
    CK(hipMalloc((void **)&d, 24)); CK(hipMalloc((void **)&sink, 64));
    for (int spin : {1000000, 4000000}) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_clocks, dim3(1), dim3(64), 0, 0, d, spin);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
        printf("spin %d: %.3f ms; clock64 %.1f MHz, wall_clock64 %.1f MHz (one idle wave)\n", spin, ms, h[0] / ms / 1e3, h[1] / ms / 1e3);
    }
    for (int b : {4, 6, 8}) {
        run<0>("v_xor+v_add (32-bit)", 16, b, sink);
        run<1>("v_mul_lo_u32", 8, b, sink);
        run<2>("lshr_b64 + 2 xor", 12, b, sink);
    }
    return 0;
}
