// pipeline_lab.hip — can consecutive batched_step launches overlap?  (diagnostic; not shipped)
// Consecutive steps depend on each other only lane by lane, i.e. workgroup j of step k+1 needs workgroup j
// of step k, nothing else.  Variant "chain": every workgroup waits on a per-workgroup counter its
// predecessor publishes, and the launches carry hipExtAnyOrderLaunch (no barrier between the dispatches),
// so step k+1's workgroups start while step k's tail is still running.  Every wait is bounded.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I gym_soccer_littman94_amd/csrc tools/labs/pipeline_lab.hip -o build/pipeline_lab
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "soccer_kernels.hpp"
#include "soccer_rules.hpp"
using namespace soccer;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <bool SLIP>
__global__ __launch_bounds__(kBlock) void k_plain(const KernelParams P, const StepIO IO, unsigned long long tick) {
    const unsigned long long g = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if ((g << 2) >= P.n) return;
    hot_group<SLIP>(P, IO, g, nullptr, tick);
}

// Kernel-argument preload experiment: the arguments the first loads depend on come first, as scalars, so that
// (built with -mllvm -amdgpu-kernarg-preload-count=14) they arrive in SGPRs at wave launch and the data loads
// can be issued without waiting for the scalar load of the kernarg segment.
template <bool SLIP, int UNROLL = 1>
__global__ __launch_bounds__(kBlock) void k_plain_pre(uint8_t* state, unsigned long long stride, const int8_t* act_a, const int8_t* act_b,
                                                      unsigned long long tick, unsigned long long n, unsigned long long first,
                                                      const KernelParams P, const StepIO IO) {
    const unsigned long long g = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if ((g << 2) >= n) return;
    KernelParams Q = P; Q.state = state; Q.state_stride = stride; Q.n = n; Q.first = first;
    StepIO J = IO; J.act_a = act_a; J.act_b = act_b;
    hot_group<SLIP, false, UNROLL>(Q, J, g, nullptr, tick);
}

template <bool SLIP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_plain_blk(uint8_t* state, unsigned long long stride, const int8_t* act_a, const int8_t* act_b,
                                                     unsigned long long tick, unsigned long long n, unsigned long long first,
                                                     const KernelParams P, const StepIO IO) {
    const unsigned long long g = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    if ((g << 2) >= n) return;
    KernelParams Q = P; Q.state = state; Q.state_stride = stride; Q.n = n; Q.first = first;
    StepIO J = IO; J.act_a = act_a; J.act_b = act_b;
    hot_group<SLIP, false, 4>(Q, J, g, nullptr, tick);
}

// Phase-structured variant of the slip-0 hot body: all eight move-table reads of the thread's four lanes are issued
// together, then the four observation reads, and auto-reset / frozen lanes are patched afterwards in one place
// (one group of four ISD reads) instead of a divergent branch per lane.
__device__ __forceinline__ void hot_group_phased(const KernelParams& P, const StepIO& IO, unsigned long long g, unsigned long long tick) {
    const unsigned long long i0 = P.first + (g << 2);
    Tables T; T.lut = P.lut; T.nc = P.next_cell; T.isd = P.isd;
    const uint8_t* sp = P.state;
    const uint32_t ra = *reinterpret_cast<const uint32_t*>(sp + i0);
    const uint32_t ca = *reinterpret_cast<const uint32_t*>(sp + P.state_stride + i0);
    const uint32_t rb = *reinterpret_cast<const uint32_t*>(sp + 2 * P.state_stride + i0);
    const uint32_t cb = *reinterpret_cast<const uint32_t*>(sp + 3 * P.state_stride + i0);
    const uint32_t ps = *reinterpret_cast<const uint32_t*>(sp + 4 * P.state_stride + i0);
    const uint32_t tt = *reinterpret_cast<const uint32_t*>(sp + 5 * P.state_stride + i0);
    const uint32_t aa = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(IO.act_a + i0));
    const uint32_t ab = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(IO.act_b + i0));
    const Philox4 blk = lane_block(P, (P.lane_offset + i0) >> 2, tick, 0u);
    const uint32_t Wm1 = (uint32_t)(P.W - 1);
    uint32_t A[4], B[4], p[4], fz[4], t[4], a[4], b[4], gA[4], gB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t sh = 8u * j, psj = (ps >> sh) & 0xffu;
        A[j] = make_pos((ra >> sh) & 0xffu, (ca >> sh) & 0xffu, P.W);
        B[j] = make_pos((rb >> sh) & 0xffu, (cb >> sh) & 0xffu, P.W);
        p[j] = psj & 1u; fz[j] = (psj >> 1) & 1u; t[j] = (tt >> sh) & 0xffu;
        a[j] = (aa >> sh) & 0xffu; b[j] = (ab >> sh) & 0xffu;
        gA[j] = moved(T, P, A[j], p[j] ^ 1u, a[j]); gB[j] = moved(T, P, B[j], p[j], b[j]);
    }
    uint32_t nA[4], nB[4], np[4], nt[4], nd[4], ob[4], tr[4], dn[4]; int32_t rw[4];
    uint32_t special = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t cc = col_of(p[j] ? B[j] : A[j]);
        const bool in_goal = (cc == 0u) | (cc == Wm1);
        const Resolved R = classify(A[j], B[j], gA[j], gB[j], a[j], b[j]);
        const uint32_t w = blk.w[j], top2 = w >> 30;
        const uint32_t k = R.kind == K_COIN ? (top2 >> 1) : top2;
        Outcome sel = pick(A[j], B[j], p[j], R, k);
        sel.A = in_goal ? A[j] : sel.A; sel.B = in_goal ? B[j] : sel.B; sel.p = in_goal ? p[j] : sel.p;
        const uint32_t ncc = col_of(sel.p ? sel.B : sel.A);
        const bool goal_now = (ncc == 0u) | (ncc == Wm1);
        rw[j] = (goal_now & !in_goal) ? (ncc == Wm1 ? 1 : -1) : 0;
        nt[j] = t[j] + 1u; tr[j] = nt[j] >= (uint32_t)P.max_steps ? 1u : 0u; dn[j] = goal_now ? 1u : 0u;
        nd[j] = dn[j] | tr[j];
        nA[j] = sel.A; nB[j] = sel.B; np[j] = sel.p;
        ob[j] = obs_of(T, P, sel.A, sel.B, sel.p);
        special |= (nd[j] & P.autoreset) | fz[j];
    }
    bool mis = false;
    if (special) {
        uint4 e[4]; uint32_t cur[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[j] = *reinterpret_cast<const uint4*>(T.isd + 4u * ((blk.w[j] & 3u) >> P.isd_shift));
            cur[j] = obs_of(T, P, A[j], B[j], p[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool rs = (nd[j] & P.autoreset) != 0u, frozen = fz[j] != 0u;
            nA[j] = rs ? e[j].x : nA[j]; nB[j] = rs ? e[j].y : nB[j]; np[j] = rs ? (e[j].z & 1u) : np[j];
            nt[j] = rs ? 0u : nt[j]; ob[j] = rs ? (e[j].z >> 16) : ob[j]; nd[j] = rs ? 0u : nd[j];
            const uint32_t cc = col_of(p[j] ? B[j] : A[j]);
            const bool in_goal = (cc == 0u) | (cc == Wm1);
            nA[j] = frozen ? A[j] : nA[j]; nB[j] = frozen ? B[j] : nB[j]; np[j] = frozen ? p[j] : np[j];
            tr[j] = frozen ? (t[j] >= (uint32_t)P.max_steps ? 1u : 0u) : tr[j]; dn[j] = frozen ? (in_goal ? 1u : 0u) : dn[j];
            nt[j] = frozen ? t[j] : nt[j]; nd[j] = frozen ? 1u : nd[j]; ob[j] = frozen ? cur[j] : ob[j]; rw[j] = frozen ? 0 : rw[j];
            mis |= frozen;
        }
    }
    uint32_t nra = 0, nca = 0, nrb = 0, ncb = 0, nps = 0, ntt = 0, o_rew = 0, o_term = 0, o_trunc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t sh = 8u * j;
        nra |= (nA[j] >> 24) << sh; nca |= ((nA[j] >> 16) & 0xffu) << sh; nrb |= (nB[j] >> 24) << sh; ncb |= ((nB[j] >> 16) & 0xffu) << sh;
        nps |= (np[j] | (nd[j] << 1)) << sh; ntt |= nt[j] << sh;
        o_rew |= ((uint32_t)rw[j] & 0xffu) << sh; o_term |= dn[j] << sh; o_trunc |= tr[j] << sh;
    }
    const uint32_t o_lo = ob[0] | (ob[1] << 16), o_hi = ob[2] | (ob[3] << 16);
    uint8_t* sw = P.state;
    *reinterpret_cast<uint32_t*>(sw + i0) = nra; *reinterpret_cast<uint32_t*>(sw + P.state_stride + i0) = nca;
    *reinterpret_cast<uint32_t*>(sw + 2 * P.state_stride + i0) = nrb; *reinterpret_cast<uint32_t*>(sw + 3 * P.state_stride + i0) = ncb;
    *reinterpret_cast<uint32_t*>(sw + 4 * P.state_stride + i0) = nps; *reinterpret_cast<uint32_t*>(sw + 5 * P.state_stride + i0) = ntt;
    if (IO.obs) __builtin_nontemporal_store((unsigned long long)o_lo | ((unsigned long long)o_hi << 32), reinterpret_cast<unsigned long long*>(IO.obs + i0));
    if (IO.reward) __builtin_nontemporal_store(o_rew, reinterpret_cast<uint32_t*>(IO.reward + i0));
    if (IO.terminated) __builtin_nontemporal_store(o_term, reinterpret_cast<uint32_t*>(IO.terminated + i0));
    if (IO.truncated) __builtin_nontemporal_store(o_trunc, reinterpret_cast<uint32_t*>(IO.truncated + i0));
    if (mis) *P.misuse = 1u;
}

__global__ __launch_bounds__(kBlock) void k_phased(uint8_t* state, unsigned long long stride, const int8_t* act_a, const int8_t* act_b,
                                                   unsigned long long tick, unsigned long long n, unsigned long long first,
                                                   const KernelParams P, const StepIO IO) {
    const unsigned long long g = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if ((g << 2) >= n) return;
    KernelParams Q = P; Q.state = state; Q.state_stride = stride; Q.n = n; Q.first = first;
    StepIO J = IO; J.act_a = act_a; J.act_b = act_b;
    hot_group_phased(Q, J, g, tick);
}

// STRIDE: uint32 words between consecutive workgroups' counters (1 = packed, 16 = one 64-byte line each)
template <bool SLIP, int STRIDE>
__global__ __launch_bounds__(kBlock) void k_chain(const KernelParams P, const StepIO IO, unsigned long long tick,
                                                  uint32_t* done, uint32_t seq, uint32_t* err) {
    uint32_t* mine = done + (size_t)blockIdx.x * STRIDE;
    if (threadIdx.x == 0) {
        uint32_t spins = 0;
        while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 17)) { *err = 1u; break; }               // bounded: never hang the GPU
        }
    }
    __syncthreads();
    const unsigned long long g = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if ((g << 2) < P.n) hot_group<SLIP>(P, IO, g, nullptr, tick);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(mine, seq + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

int main(int argc, char** argv) {
    const size_t N = 1 << 20;
    const int K = argc > 1 ? atoi(argv[1]) : 1000;
    Rules R; R.build(5, 4);
    KernelParams P{};
    uint8_t* d_state; CK(hipMalloc(&d_state, 6 * N));
    uint16_t* d_lut; uint32_t* d_nc; uint32_t* d_isd; unsigned int* d_mis;
    CK(hipMalloc(&d_lut, R.lut.size() * 2)); CK(hipMemcpy(d_lut, R.lut.data(), R.lut.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_nc, R.next_cell.size() * 4)); CK(hipMemcpy(d_nc, R.next_cell.data(), R.next_cell.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_isd, 64)); CK(hipMemcpy(d_isd, R.isd_words, 64, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_mis, 128)); CK(hipMemset(d_mis, 0, 128));
    P.state = d_state; P.state_stride = N; P.lut = d_lut; P.lut_len = (int)R.lut.size(); P.next_cell = d_nc; P.isd = d_isd;
    P.key0 = 1; P.key1 = 2; P.lane_offset = 0; P.misuse = d_mis; P.first = 0; P.n = N; P.W = R.W; P.HW = R.H * R.W; P.HW5 = 5 * R.H * R.W;
    P.nc_len = (int)R.next_cell.size(); P.max_steps = 100; P.autoreset = 1; P.step_stats = 0; P.isd_shift = 0;
    P.w[0] = 1; P.w[1] = P.w[2] = P.w[3] = 0; P.nb = 1; P.act_pack = 0; for (int c = 0; c < 9; ++c) P.B[c] = c ? __builtin_inf() : 1.0;
    const int T = 64;
    int8_t* d_act; CK(hipMalloc(&d_act, (size_t)T * 2 * N));
    { std::vector<int8_t> h((size_t)T * 2 * N); srand(7); for (auto& x : h) x = rand() % 5; CK(hipMemcpy(d_act, h.data(), h.size(), hipMemcpyHostToDevice)); }
    uint16_t* d_obs; int8_t* d_rew; uint8_t* d_term; uint8_t* d_trunc;
    CK(hipMalloc(&d_obs, N * 2)); CK(hipMalloc(&d_rew, N)); CK(hipMalloc(&d_term, N)); CK(hipMalloc(&d_trunc, N));
    uint32_t* d_done; CK(hipMalloc(&d_done, 1024 * 16 * 4)); uint32_t* d_err; CK(hipMalloc(&d_err, 4)); CK(hipMemset(d_err, 0, 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto io_for = [&](int k) { StepIO io{}; io.act_a = d_act + (size_t)(k % T) * 2 * N; io.act_b = io.act_a + N;
        io.obs = d_obs; io.reward = d_rew; io.terminated = d_term; io.truncated = d_trunc; return io; };
    auto init = [&] {
        CK(hipMemset(d_state, 1, N)); CK(hipMemset(d_state + N, 2, N)); CK(hipMemset(d_state + 2 * N, 2, N));
        CK(hipMemset(d_state + 3 * N, 4, N)); CK(hipMemset(d_state + 4 * N, 0, N)); CK(hipMemset(d_state + 5 * N, 0, N));
        CK(hipMemset(d_done, 0, 1024 * 16 * 4)); CK(hipDeviceSynchronize());
    };
    std::vector<uint8_t> ref(6 * N + 2 * N), got(6 * N + 2 * N);
    auto snapshot = [&](std::vector<uint8_t>& v) { CK(hipMemcpy(v.data(), d_state, 6 * N, hipMemcpyDeviceToHost)); CK(hipMemcpy(v.data() + 6 * N, d_obs, 2 * N, hipMemcpyDeviceToHost)); };
    auto timed = [&](const char* name, auto launch, std::vector<uint8_t>* out) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            init();
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < K; ++k) launch(k);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        if (out) snapshot(*out);
        uint32_t err; CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
        printf("%-46s %7.2f us/step%s\n", name, best * 1e3f / K, err ? "   [WAIT GAVE UP]" : "");
        CK(hipMemset(d_err, 0, 4));
    };
    timed("plain, normal launches", [&](int k) { hipLaunchKernelGGL(k_plain<false>, dim3(1024), dim3(256), 0, st, P, io_for(k), (unsigned long long)k); }, &ref);
    timed("plain, leading scalar args (kernarg preload)", [&](int k) { StepIO io = io_for(k); hipLaunchKernelGGL(k_plain_pre<false>, dim3(1024), dim3(256), 0, st, P.state, P.state_stride, io.act_a, io.act_b, (unsigned long long)k, P.n, P.first, P, io); }, &got);
    printf("   identical to plain: %s\n", ref == got ? "yes" : "NO");
    timed("preload + lane loop unrolled x2", [&](int k) { StepIO io = io_for(k); hipLaunchKernelGGL((k_plain_pre<false, 2>), dim3(1024), dim3(256), 0, st, P.state, P.state_stride, io.act_a, io.act_b, (unsigned long long)k, P.n, P.first, P, io); }, &got);
    printf("   identical to plain: %s\n", ref == got ? "yes" : "NO");
    timed("preload + lane loop unrolled x4", [&](int k) { StepIO io = io_for(k); hipLaunchKernelGGL((k_plain_pre<false, 4>), dim3(1024), dim3(256), 0, st, P.state, P.state_stride, io.act_a, io.act_b, (unsigned long long)k, P.n, P.first, P, io); }, &got);
    printf("   identical to plain: %s\n", ref == got ? "yes" : "NO");
    timed("preload, phased body (gathers grouped, resets patched once)", [&](int k) { StepIO io = io_for(k); hipLaunchKernelGGL(k_phased, dim3(1024), dim3(256), 0, st, P.state, P.state_stride, io.act_a, io.act_b, (unsigned long long)k, P.n, P.first, P, io); }, &got);
    printf("   identical to plain: %s\n", ref == got ? "yes" : "NO");
    timed("preload x4 (again)", [&](int k) { StepIO io = io_for(k); hipLaunchKernelGGL((k_plain_pre<false, 4>), dim3(1024), dim3(256), 0, st, P.state, P.state_stride, io.act_a, io.act_b, (unsigned long long)k, P.n, P.first, P, io); }, &got);
    printf("   identical to plain: %s\n", ref == got ? "yes" : "NO");
    timed("preload x4, 128-thread workgroups", [&](int k) { StepIO io = io_for(k); hipLaunchKernelGGL((k_plain_blk<false, 128>), dim3(2048), dim3(128), 0, st, P.state, P.state_stride, io.act_a, io.act_b, (unsigned long long)k, P.n, P.first, P, io); }, &got);
    printf("   identical to plain: %s\n", ref == got ? "yes" : "NO");
    timed("preload x4, 512-thread workgroups", [&](int k) { StepIO io = io_for(k); hipLaunchKernelGGL((k_plain_blk<false, 512>), dim3(512), dim3(512), 0, st, P.state, P.state_stride, io.act_a, io.act_b, (unsigned long long)k, P.n, P.first, P, io); }, &got);
    printf("   identical to plain: %s\n", ref == got ? "yes" : "NO");
    timed("preload x4, 1024-thread workgroups", [&](int k) { StepIO io = io_for(k); hipLaunchKernelGGL((k_plain_blk<false, 1024>), dim3(256), dim3(1024), 0, st, P.state, P.state_stride, io.act_a, io.act_b, (unsigned long long)k, P.n, P.first, P, io); }, &got);
    printf("   identical to plain: %s\n", ref == got ? "yes" : "NO");
    timed("preload (again)", [&](int k) { StepIO io = io_for(k); hipLaunchKernelGGL((k_plain_pre<false, 1>), dim3(1024), dim3(256), 0, st, P.state, P.state_stride, io.act_a, io.act_b, (unsigned long long)k, P.n, P.first, P, io); }, &got);
    printf("   identical to plain: %s\n", ref == got ? "yes" : "NO");
    if (argc > 2) return 0;
    timed("chain (packed flags), normal launches", [&](int k) { hipLaunchKernelGGL((k_chain<false, 1>), dim3(1024), dim3(256), 0, st, P, io_for(k), (unsigned long long)k, d_done, (uint32_t)k, d_err); }, &got);
    printf("   identical to plain: %s\n", ref == got ? "yes" : "NO");
    timed("chain (packed flags), any-order launches", [&](int k) { hipExtLaunchKernelGGL((k_chain<false, 1>), dim3(1024), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, P, io_for(k), (unsigned long long)k, d_done, (uint32_t)k, d_err); }, &got);
    printf("   identical to plain: %s\n", ref == got ? "yes" : "NO");
    timed("chain (64 B per flag), any-order launches", [&](int k) { hipExtLaunchKernelGGL((k_chain<false, 16>), dim3(1024), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, P, io_for(k), (unsigned long long)k, d_done, (uint32_t)k, d_err); }, &got);
    printf("   identical to plain: %s\n", ref == got ? "yes" : "NO");
    timed("plain, normal launches (again)", [&](int k) { hipLaunchKernelGGL(k_plain<false>, dim3(1024), dim3(256), 0, st, P, io_for(k), (unsigned long long)k); }, &got);
    printf("   identical to plain: %s\n", ref == got ? "yes" : "NO");
    return 0;
}
