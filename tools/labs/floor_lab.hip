// floor_lab.hip — what bounds a single-step launch at 2^20 lanes?  Same grid (1024 x 256 threads, 4 lanes per thread), same
// streams and non-temporal dword accesses as step_kernel_swar, graph-replayed like bench.py:
//   empty      nothing                                    -> launch + dependent-kernel boundary
//   copy       8 loads, 10 stores, one xor per dword      -> + the 19 B/lane round trip
//   copy+rng   + the Philox block of the thread's 4 lanes -> + the vector work that no rule needs
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igym_soccer_littman94_amd/csrc -o build/floor_lab tools/labs/floor_lab.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "soccer_kernels.hpp"
using namespace soccer;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

struct Args { uint8_t* state; unsigned long long stride; const int8_t* aa; const int8_t* ab; uint16_t* obs; int8_t* rew; uint8_t* te; uint8_t* tr; unsigned long long tick; };

__global__ __launch_bounds__(256) void k_empty(const Args A) {}

template <bool RNG>
__global__ __launch_bounds__(256) void k_copy(const Args A) {
    const unsigned long long i0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) << 2;
    const uint8_t* sp = A.state + i0;
    uint32_t s[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(sp + k * A.stride));
    const uint32_t a = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(A.aa + i0));
    const uint32_t b = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(A.ab + i0));
    uint32_t x = a ^ b;
    if (RNG) { const Philox4 p = philox4x32_10((uint32_t)(i0 >> 2), 0u, (uint32_t)A.tick, 0u, 1u, 2u); x ^= p.w[0] ^ p.w[1] ^ p.w[2] ^ p.w[3]; }
    x &= 0x01010101u;
    uint8_t* sw = A.state + i0;
#pragma unroll
    for (int k = 0; k < 6; ++k) __builtin_nontemporal_store(s[k] ^ (k == 5 ? x : 0u), reinterpret_cast<uint32_t*>(sw + k * A.stride));
    __builtin_nontemporal_store((unsigned long long)s[0] | ((unsigned long long)s[1] << 32), reinterpret_cast<unsigned long long*>(A.obs + i0));
    __builtin_nontemporal_store(s[2] ^ x, reinterpret_cast<uint32_t*>(A.rew + i0));
    __builtin_nontemporal_store(s[3], reinterpret_cast<uint32_t*>(A.te + i0));
    __builtin_nontemporal_store(s[4], reinterpret_cast<uint32_t*>(A.tr + i0));
}

int main() {
    const size_t N = 1 << 20; const int K = 200, ROUNDS = 7, T = 64;
    uint8_t* st_; int8_t* act; uint16_t* obs; int8_t* rew; uint8_t* te; uint8_t* tr;
    CK(hipMalloc(&st_, 6 * N)); CK(hipMemset(st_, 1, 6 * N));
    CK(hipMalloc(&act, (size_t)T * 2 * N)); CK(hipMemset(act, 2, (size_t)T * 2 * N));
    CK(hipMalloc(&obs, (size_t)T * N * 2)); CK(hipMalloc(&rew, (size_t)T * N)); CK(hipMalloc(&te, (size_t)T * N)); CK(hipMalloc(&tr, (size_t)T * N));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto args = [&](int k) { const size_t r = (size_t)(k % T); return Args{st_, N, act + r * 2 * N, act + r * 2 * N + N, obs + r * N, rew + r * N, te + r * N, tr + r * N, (unsigned long long)k}; };
    const char* names[3] = {"empty", "copy (19 B/lane, nt dwords)", "copy + Philox block"};
    for (int v = 0; v < 3; ++v) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < K; ++k) {
            const Args a = args(k);
            if (v == 0) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, a);
            else if (v == 1) hipLaunchKernelGGL(k_copy<false>, dim3(1024), dim3(256), 0, s, a);
            else hipLaunchKernelGGL(k_copy<true>, dim3(1024), dim3(256), 0, s, a);
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        std::vector<float> ms;
        for (int r = 0; r < ROUNDS; ++r) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t * 1e3f / K);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-30s median %.2f  min %.2f us per launch (graph of %d, N = 2^20)\n", names[v], ms[ms.size() / 2], ms[0], K);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
