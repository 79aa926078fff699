// valu_rate_lab.hip — issue cost of the vector instructions the byte-parallel kernels are made of.  Every wave runs a
// dependent chain  x = op(x, y, c); y = op(y, x, c)  (inline asm, so the instruction is exactly the one named), 16 pairs per
// loop iteration; ticks (s_memtime) per instruction and wave, at 1 wave per SIMD (latency of a dependent issue) and at
// 4 waves per SIMD (the occupancy of the step / rollout kernels: 1024 x 256 threads on 256 CUs — throughput).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/valu_rate_lab tools/labs/valu_rate_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

#define OPS(X) \
    X(0,  "v_xor_b32 %0, %1, %2",                 2) \
    X(1,  "v_add_u32 %0, %1, %2",                 2) \
    X(2,  "v_sub_u32 %0, %1, %2",                 2) \
    X(3,  "v_and_b32 %0, %1, %2",                 2) \
    X(4,  "v_lshlrev_b32 %0, 7, %1",              1) \
    X(5,  "v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96", 3) \
    X(6,  "v_bfi_b32 %0, %1, %2, %3",             3) \
    X(7,  "v_perm_b32 %0, %1, %2, %3",            3) \
    X(8,  "v_perm_b32 %0, %1, %2, %4",            4) \
    X(9,  "v_or3_b32 %0, %1, %2, %3",             3) \
    X(10, "v_and_or_b32 %0, %1, %2, %3",          3) \
    X(11, "v_lshl_or_b32 %0, %1, 8, %2",          2) \
    X(12, "v_lshl_add_u32 %0, %1, 1, %2",         2) \
    X(13, "v_add3_u32 %0, %1, %2, %3",            3) \
    X(14, "v_pk_mad_u16 %0, %1, %2, %3",          3) \
    X(15, "v_pk_add_u16 %0, %1, %2",              2) \
    X(16, "v_alignbit_b32 %0, %1, %2, 16",        2) \
    X(17, "v_alignbyte_b32 %0, %1, %2, 1",        2) \
    X(18, "v_bcnt_u32_b32 %0, %1, %2",            2) \
    X(19, "v_mul_lo_u32 %0, %1, %2",              2) \
    X(20, "v_mul_hi_u32 %0, %1, %2",              2) \
    X(21, "v_mul_u32_u24 %0, %1, %2",             2) \
    X(22, "v_mad_u32_u24 %0, %1, %2, %3",         3) \
    X(23, "v_mov_b32 %0, %1",                     1) \
    X(24, "v_xad_u32 %0, %1, %2, %3",             3) \
    X(25, "v_sub_u32 %0, 0x80808080, %1",         1) \
    X(26, "v_and_b32 %0, 0x7070707, %1",          1) \
    X(27, "v_xor_b32 %0, %4, %1",                 4) \
    X(28, "v_bitop3_b32 %0, %1, %2, %4 bitop3:0x96", 4) \
    X(29, "v_cndmask_b32 %0, %1, %2, vcc",        2)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* cyc, uint32_t a0, uint32_t m, int iters) {
    uint32_t x = a0 + threadIdx.x, y = a0 ^ (threadIdx.x * 2654435761u), c = m + threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#define X(ID, TXT, N) if (OP == ID) { asm volatile(TXT : "=v"(x) : "v"(x), "v"(y), "v"(c), "s"(m)); asm volatile(TXT : "=v"(y) : "v"(y), "v"(x), "v"(c), "s"(m)); }
            OPS(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = x ^ y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// 64-bit results: v_mad_u64_u32 / v_lshl_add_u64 (address arithmetic)
template <int OP>
__global__ __launch_bounds__(256) void k64(uint32_t* out, unsigned long long* cyc, uint32_t a0, uint32_t m, int iters) {
    unsigned long long x = a0 + threadIdx.x, y = a0 ^ (threadIdx.x * 2654435761u), c = m + threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(x) : "v"((uint32_t)x), "s"(m) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(y) : "v"((uint32_t)y), "s"(m) : "vcc"); }
            else { asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(x) : "v"(x), "v"(c)); asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(y) : "v"(y), "v"(c)); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(x ^ y ^ (x >> 32));
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static double run(void (*kern)(uint32_t*, unsigned long long*, uint32_t, uint32_t, int), int blocks, uint32_t* out, unsigned long long* cyc, int iters) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, 1u, 0x03010200u, iters); CK(hipDeviceSynchronize()); }
    std::vector<unsigned long long> h(blocks);
    CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto v : h) s += (double)v;
    return s / blocks / (iters * 32.0);
}

int main() {
    const int iters = 256;
    uint32_t* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 4096 * 256 * 4)); CK(hipMalloc(&cyc, 4096 * 8));
    printf("%-46s %12s %12s   (ticks per instruction and wave)\n", "instruction", "1 wave/SIMD", "4 waves/SIMD");
#define X(ID, TXT, N) printf("%-46s %12.2f %12.2f\n", TXT, run(k<ID>, 256, out, cyc, iters), run(k<ID>, 1024, out, cyc, iters));
    OPS(X)
#undef X
    printf("%-46s %12.2f %12.2f\n", "v_mad_u64_u32 (x SGPR constant)", run(k64<0>, 256, out, cyc, iters), run(k64<0>, 1024, out, cyc, iters));
    printf("%-46s %12.2f %12.2f\n", "v_lshl_add_u64", run(k64<1>, 256, out, cyc, iters), run(k64<1>, 1024, out, cyc, iters));
    return 0;
}
