// kernel_lab.hip — ablation harness for step_kernel at N = 2^20 (diagnostic; not shipped).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I gym_soccer_littman94_amd/csrc tools/labs/kernel_lab.hip -o build/kernel_lab
// Times interleaved variants in one process with HIP events (cdna_hip_programming.md §5.4 rule 24).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>
#include "soccer_kernels.hpp"
#include "soccer_rules.hpp"
using namespace soccer;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_empty(const KernelParams P, const StepIO IO) {}

__global__ __launch_bounds__(256) void k_tick(const KernelParams P, const StepIO IO) {
    const unsigned long long tick = *P.tick_in;
    publish_tick(P, tick, 1ull);
}

__global__ __launch_bounds__(256) void k_stage(const KernelParams P, const StepIO IO) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const Tables T = stage_tables<true>(P, smem);
    if (T.nc[threadIdx.x] == 0x1234 && IO.obs) IO.obs[0] = 1;   // keep it live
}

template <int E, bool STAGE>
__global__ __launch_bounds__(256) void k_copy(const KernelParams P, const StepIO IO) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long i0 = g * E;
    if (i0 + E > P.n) return;
    RawState<E> raw; PackB<E> aa, ab;
    raw.load(P, i0); aa.load(IO.act_a, i0); ab.load(IO.act_b, i0);
    if (STAGE) { const Tables T = stage_tables<true>(P, smem); if (T.nc[threadIdx.x] == 0x1234) aa.w[0] ^= 1; }
    LaneVec<E> S; S.unpack(P, raw);
    PackB<E> o1, o2, o3; PackH<E> oh; o1.clear(); o2.clear(); o3.clear(); oh.clear();
#pragma unroll
    for (int j = 0; j < E; ++j) {
        S.L[j].t = (S.L[j].t + aa.get(j)) & 63u;
        o1.put(j, aa.get(j)); o2.put(j, ab.get(j)); o3.put(j, S.L[j].p); oh.put(j, S.L[j].A & 0xffffu);
    }
    S.store(P, i0);
    oh.store(IO.obs, i0); o1.store(IO.reward, i0); o2.store(IO.terminated, i0); o3.store(IO.truncated, i0);
}


// ---- experimental step kernels -------------------------------------------------------------------
// rolled per-lane loop (small code footprint), E = 4, byte accumulators shifted in with alignbyte
template <bool USE_LDS, bool HIST>
__global__ __launch_bounds__(256) void k_step_rolled(const KernelParams P, const StepIO IO) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long i0 = g * 4;
    const bool active = i0 + 4 <= P.n;
    uint32_t ra = 0, ca = 0, rb = 0, cb = 0, ps = 0, tt = 0, aa = 0, ab = 0;
    if (active) {
        const uint8_t* s = P.state;
        ra = *reinterpret_cast<const uint32_t*>(s + i0); ca = *reinterpret_cast<const uint32_t*>(s + P.state_stride + i0);
        rb = *reinterpret_cast<const uint32_t*>(s + 2 * P.state_stride + i0); cb = *reinterpret_cast<const uint32_t*>(s + 3 * P.state_stride + i0);
        ps = *reinterpret_cast<const uint32_t*>(s + 4 * P.state_stride + i0); tt = *reinterpret_cast<const uint32_t*>(s + 5 * P.state_stride + i0);
        aa = *reinterpret_cast<const uint32_t*>(IO.act_a + i0); ab = *reinterpret_cast<const uint32_t*>(IO.act_b + i0);
    }
    const unsigned long long tick = *P.tick_in;
    HistAcc<true> hist; hist.init(P);
    Tables T;
    if (USE_LDS) T = stage_tables<true>(P, smem);
    else { T.lut = P.lut; T.nc = P.next_cell; T.isd = P.isd; if (HIST) __syncthreads(); }
    publish_tick(P, tick, 1ull);
    if (active) {
        const Philox4 blk = lane_block(P, (P.lane_offset + i0) >> 2, tick, 0u);
        uint32_t nra = 0, nca = 0, nrb = 0, ncb = 0, nps = 0, ntt = 0, o_rew = 0, o_term = 0, o_trunc = 0, o_lo = 0, o_hi = 0;
        bool mis = false;
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const uint32_t w = j & 2 ? (j & 1 ? blk.w[3] : blk.w[2]) : (j & 1 ? blk.w[1] : blk.w[0]);
            Lane L;
            L.A = make_pos(ra & 0xffu, ca & 0xffu, P.W); L.B = make_pos(rb & 0xffu, cb & 0xffu, P.W);
            L.p = ps & 1u; L.need = (ps >> 1) & 1u; L.t = tt & 0xffu;
            StepResult R;
            mis |= lane_step<false>(T, P, L, aa & 0xffu, ab & 0xffu, draw_from_word(w), R);
            ra >>= 8; ca >>= 8; rb >>= 8; cb >>= 8; ps >>= 8; tt >>= 8; aa >>= 8; ab >>= 8;
            nra = __builtin_amdgcn_alignbyte(L.A >> 24, nra, 1); nca = __builtin_amdgcn_alignbyte((L.A >> 16) & 0xffu, nca, 1);
            nrb = __builtin_amdgcn_alignbyte(L.B >> 24, nrb, 1); ncb = __builtin_amdgcn_alignbyte((L.B >> 16) & 0xffu, ncb, 1);
            nps = __builtin_amdgcn_alignbyte(L.p | (L.need << 1), nps, 1); ntt = __builtin_amdgcn_alignbyte(L.t, ntt, 1);
            o_rew = __builtin_amdgcn_alignbyte((uint32_t)R.reward & 0xffu, o_rew, 1);
            o_term = __builtin_amdgcn_alignbyte(R.term, o_term, 1); o_trunc = __builtin_amdgcn_alignbyte(R.trunc, o_trunc, 1);
            o_lo = __builtin_amdgcn_alignbit(o_hi, o_lo, 16); o_hi = (o_hi >> 16) | (R.obs << 16);
            if (HIST) hist.add(R.finished, R.reward);
        }
        uint8_t* s = P.state;
        *reinterpret_cast<uint32_t*>(s + i0) = nra; *reinterpret_cast<uint32_t*>(s + P.state_stride + i0) = nca;
        *reinterpret_cast<uint32_t*>(s + 2 * P.state_stride + i0) = nrb; *reinterpret_cast<uint32_t*>(s + 3 * P.state_stride + i0) = ncb;
        *reinterpret_cast<uint32_t*>(s + 4 * P.state_stride + i0) = nps; *reinterpret_cast<uint32_t*>(s + 5 * P.state_stride + i0) = ntt;
        *reinterpret_cast<uint2*>(IO.obs + i0) = make_uint2(o_lo, o_hi);
        *reinterpret_cast<uint32_t*>(IO.reward + i0) = o_rew; *reinterpret_cast<uint32_t*>(IO.terminated + i0) = o_term;
        *reinterpret_cast<uint32_t*>(IO.truncated + i0) = o_trunc;
        if (mis) *P.misuse = 1u;
    }
    if (HIST) hist.flush(P);
}


// LPT lanes per thread (1, 2 or 4) with the rolled lane loop, LEAN outputs: more waves for the same batch
// STAG: which workgroups start late (0 none, 1: (b>>8)&1, 2: b&1, 3: (b>>9)&1, 4: wave parity within the block);
// SLEEP: s_sleep argument (x64 cycles).  Late starters load while early ones compute, store while late compute.
template <int LPT, int STAG = 0, int SLEEP = 0>
__global__ __launch_bounds__(256) void k_step_lpt(const KernelParams P, const StepIO IO) {
    if (STAG) {
        const bool late = STAG == 1 ? ((blockIdx.x >> 8) & 1) : STAG == 2 ? (blockIdx.x & 1) : STAG == 3 ? ((blockIdx.x >> 9) & 1)
                                                                                           : ((threadIdx.x >> 6) & 1);
        if (late) __builtin_amdgcn_s_sleep(SLEEP);
    }
    const unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long i0 = g * LPT;
    if (i0 + LPT > P.n) return;
    const unsigned long long tick = *P.tick_in;
    publish_tick(P, tick, 1ull);
    Tables T; T.lut = P.lut; T.nc = P.next_cell; T.isd = P.isd;
    const uint8_t* s = P.state;
    auto ld = [&](const void* base) -> uint32_t {
        const uint8_t* p = static_cast<const uint8_t*>(base) + i0;
        if (LPT == 4) return *reinterpret_cast<const uint32_t*>(p);
        if (LPT == 2) return *reinterpret_cast<const uint16_t*>(p);
        return *p;
    };
    uint32_t ra = ld(s), ca = ld(s + P.state_stride), rb = ld(s + 2 * P.state_stride), cb = ld(s + 3 * P.state_stride);
    uint32_t ps = ld(s + 4 * P.state_stride), tt = ld(s + 5 * P.state_stride), aa = ld(IO.act_a), ab = ld(IO.act_b);
    const unsigned long long gl = P.lane_offset + i0;
    const Philox4 blk = lane_block(P, gl >> 2, tick, 0u);
    uint32_t nra = 0, nca = 0, nrb = 0, ncb = 0, nps = 0, ntt = 0, o_rew = 0, o_term = 0, o_trunc = 0, o_lo = 0, o_hi = 0;
    bool mis = false;
#pragma unroll 1
    for (int j = 0; j < LPT; ++j) {
        const uint32_t wi = ((uint32_t)gl + j) & 3u;
        const uint32_t w = wi & 2 ? (wi & 1 ? blk.w[3] : blk.w[2]) : (wi & 1 ? blk.w[1] : blk.w[0]);
        const uint32_t sh = 8u * j;
        const uint32_t psj = __builtin_amdgcn_ubfe(ps, sh, 8u);
        Lane L;
        L.A = make_pos(__builtin_amdgcn_ubfe(ra, sh, 8u), __builtin_amdgcn_ubfe(ca, sh, 8u), P.W);
        L.B = make_pos(__builtin_amdgcn_ubfe(rb, sh, 8u), __builtin_amdgcn_ubfe(cb, sh, 8u), P.W);
        L.p = psj & 1u; L.need = (psj >> 1) & 1u; L.t = __builtin_amdgcn_ubfe(tt, sh, 8u);
        StepResult R;
        mis |= lane_step<false>(T, P, L, __builtin_amdgcn_ubfe(aa, sh, 8u), __builtin_amdgcn_ubfe(ab, sh, 8u), draw_from_word(w), R);
        nra = __builtin_amdgcn_alignbyte(L.A >> 24, nra, 1); nca = __builtin_amdgcn_alignbyte((L.A >> 16) & 0xffu, nca, 1);
        nrb = __builtin_amdgcn_alignbyte(L.B >> 24, nrb, 1); ncb = __builtin_amdgcn_alignbyte((L.B >> 16) & 0xffu, ncb, 1);
        nps = __builtin_amdgcn_alignbyte(L.p | (L.need << 1), nps, 1); ntt = __builtin_amdgcn_alignbyte(L.t, ntt, 1);
        o_rew = __builtin_amdgcn_alignbyte((uint32_t)R.reward & 0xffu, o_rew, 1);
        o_term = __builtin_amdgcn_alignbyte(R.term, o_term, 1); o_trunc = __builtin_amdgcn_alignbyte(R.trunc, o_trunc, 1);
        o_lo = __builtin_amdgcn_alignbit(o_hi, o_lo, 16); o_hi = (o_hi >> 16) | (R.obs << 16);
    }
    const int shb = 8 * (4 - LPT);
    auto st8 = [&](void* base, uint32_t v) {
        uint8_t* p = static_cast<uint8_t*>(base) + i0; v >>= shb;
        if (LPT == 4) *reinterpret_cast<uint32_t*>(p) = v; else if (LPT == 2) *reinterpret_cast<uint16_t*>(p) = (uint16_t)v; else *p = (uint8_t)v;
    };
    uint8_t* sw = P.state;
    st8(sw, nra); st8(sw + P.state_stride, nca); st8(sw + 2 * P.state_stride, nrb); st8(sw + 3 * P.state_stride, ncb);
    st8(sw + 4 * P.state_stride, nps); st8(sw + 5 * P.state_stride, ntt);
    st8(IO.reward, o_rew); st8(IO.terminated, o_term); st8(IO.truncated, o_trunc);
    if (LPT == 4) *reinterpret_cast<uint2*>(IO.obs + i0) = make_uint2(o_lo, o_hi);
    else if (LPT == 2) *reinterpret_cast<uint32_t*>(IO.obs + i0) = o_hi;
    else IO.obs[i0] = (uint16_t)(o_hi >> 16);
    if (mis) *P.misuse = 1u;
}

// two groups per thread, software pipelined: all 16 loads up front, compute A, store A, compute B, store B
struct G4 { uint32_t ra, ca, rb, cb, ps, tt, aa, ab; };
__device__ __forceinline__ G4 load_g4(const KernelParams& P, const StepIO& IO, unsigned long long i0) {
    G4 g; const uint8_t* s = P.state;
    g.ra = *reinterpret_cast<const uint32_t*>(s + i0); g.ca = *reinterpret_cast<const uint32_t*>(s + P.state_stride + i0);
    g.rb = *reinterpret_cast<const uint32_t*>(s + 2 * P.state_stride + i0); g.cb = *reinterpret_cast<const uint32_t*>(s + 3 * P.state_stride + i0);
    g.ps = *reinterpret_cast<const uint32_t*>(s + 4 * P.state_stride + i0); g.tt = *reinterpret_cast<const uint32_t*>(s + 5 * P.state_stride + i0);
    g.aa = *reinterpret_cast<const uint32_t*>(IO.act_a + i0); g.ab = *reinterpret_cast<const uint32_t*>(IO.act_b + i0);
    return g;
}
template <bool HIST>
__device__ __forceinline__ void do_g4(const Tables& T, const KernelParams& P, const StepIO& IO, unsigned long long i0,
                                      unsigned long long tick, G4 g, HistAcc<true>& hist, bool& mis) {
    const Philox4 blk = lane_block(P, (P.lane_offset + i0) >> 2, tick, 0u);
    uint32_t nra = 0, nca = 0, nrb = 0, ncb = 0, nps = 0, ntt = 0, o_rew = 0, o_term = 0, o_trunc = 0, o_lo = 0, o_hi = 0;
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        const uint32_t w = j & 2 ? (j & 1 ? blk.w[3] : blk.w[2]) : (j & 1 ? blk.w[1] : blk.w[0]);
        Lane L;
        L.A = make_pos(g.ra & 0xffu, g.ca & 0xffu, P.W); L.B = make_pos(g.rb & 0xffu, g.cb & 0xffu, P.W);
        L.p = g.ps & 1u; L.need = (g.ps >> 1) & 1u; L.t = g.tt & 0xffu;
        StepResult R;
        mis |= lane_step<false>(T, P, L, g.aa & 0xffu, g.ab & 0xffu, draw_from_word(w), R);
        g.ra >>= 8; g.ca >>= 8; g.rb >>= 8; g.cb >>= 8; g.ps >>= 8; g.tt >>= 8; g.aa >>= 8; g.ab >>= 8;
        nra = __builtin_amdgcn_alignbyte(L.A >> 24, nra, 1); nca = __builtin_amdgcn_alignbyte((L.A >> 16) & 0xffu, nca, 1);
        nrb = __builtin_amdgcn_alignbyte(L.B >> 24, nrb, 1); ncb = __builtin_amdgcn_alignbyte((L.B >> 16) & 0xffu, ncb, 1);
        nps = __builtin_amdgcn_alignbyte(L.p | (L.need << 1), nps, 1); ntt = __builtin_amdgcn_alignbyte(L.t, ntt, 1);
        o_rew = __builtin_amdgcn_alignbyte((uint32_t)R.reward & 0xffu, o_rew, 1);
        o_term = __builtin_amdgcn_alignbyte(R.term, o_term, 1); o_trunc = __builtin_amdgcn_alignbyte(R.trunc, o_trunc, 1);
        o_lo = __builtin_amdgcn_alignbit(o_hi, o_lo, 16); o_hi = (o_hi >> 16) | (R.obs << 16);
        if (HIST) hist.add(R.finished, R.reward);
    }
    uint8_t* s = P.state;
    *reinterpret_cast<uint32_t*>(s + i0) = nra; *reinterpret_cast<uint32_t*>(s + P.state_stride + i0) = nca;
    *reinterpret_cast<uint32_t*>(s + 2 * P.state_stride + i0) = nrb; *reinterpret_cast<uint32_t*>(s + 3 * P.state_stride + i0) = ncb;
    *reinterpret_cast<uint32_t*>(s + 4 * P.state_stride + i0) = nps; *reinterpret_cast<uint32_t*>(s + 5 * P.state_stride + i0) = ntt;
    *reinterpret_cast<uint2*>(IO.obs + i0) = make_uint2(o_lo, o_hi);
    *reinterpret_cast<uint32_t*>(IO.reward + i0) = o_rew; *reinterpret_cast<uint32_t*>(IO.terminated + i0) = o_term;
    *reinterpret_cast<uint32_t*>(IO.truncated + i0) = o_trunc;
}
template <bool HIST, int NG>
__global__ __launch_bounds__(256) void k_step_pipe(const KernelParams P, const StepIO IO) {
    const unsigned long long groups = P.n >> 2;
    const unsigned long long per = groups / NG;            // assumes divisibility (lab only)
    const unsigned long long g0 = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (g0 >= per) return;
    const unsigned long long tick = *P.tick_in;
    publish_tick(P, tick, 1ull);
    HistAcc<true> hist; hist.init(P);
    Tables T; T.lut = P.lut; T.nc = P.next_cell; T.isd = P.isd;
    bool mis = false;
    G4 g[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) g[k] = load_g4(P, IO, (g0 + k * per) << 2);
#pragma unroll 1
    for (int k = 0; k < NG; ++k) {
        G4 cur = g[0];
#pragma unroll
        for (int m = 0; m + 1 < NG; ++m) g[m] = g[m + 1];
        do_g4<HIST>(T, P, IO, (g0 + k * per) << 2, tick, cur, hist, mis);
    }
    if (mis) *P.misuse = 1u;
    if (HIST) hist.flush(P);
}

struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; };
#include <functional>
#include <chrono>

int main(int argc, char** argv) {
    const size_t N = 1 << 20;
    const int K = 200, ROUNDS = 7;
    Rules R; R.build(5, 4);
    KernelParams P{};
    const size_t padded = N;
    uint8_t* d_state; CK(hipMalloc(&d_state, 6 * padded));
    CK(hipMemset(d_state, 1, padded)); CK(hipMemset(d_state + padded, 2, padded)); CK(hipMemset(d_state + 2 * padded, 2, padded));
    CK(hipMemset(d_state + 3 * padded, 4, padded)); CK(hipMemset(d_state + 4 * padded, 0, padded)); CK(hipMemset(d_state + 5 * padded, 0, padded));
    uint16_t* d_lut; uint32_t* d_nc; uint32_t* d_isd; unsigned long long* d_tick; unsigned long long* d_hist; unsigned int* d_mis;
    CK(hipMalloc(&d_lut, R.lut.size() * 2)); CK(hipMemcpy(d_lut, R.lut.data(), R.lut.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_nc, R.next_cell.size() * 4)); CK(hipMemcpy(d_nc, R.next_cell.data(), R.next_cell.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_isd, 64)); CK(hipMemcpy(d_isd, R.isd_words, 64, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_tick, 256)); CK(hipMemset(d_tick, 0, 256));
    CK(hipMalloc(&d_hist, 8 * kHistSlots * kHistStride)); CK(hipMemset(d_hist, 0, 8 * kHistSlots * kHistStride));
    CK(hipMalloc(&d_mis, 128)); CK(hipMemset(d_mis, 0, 128));
    P.state = d_state; P.state_stride = padded; P.lut = d_lut; P.lut_len = (int)R.lut.size(); P.next_cell = d_nc; P.isd = d_isd;
    P.tick_in = d_tick; P.tick_out = d_tick + 16; P.key0 = 1; P.key1 = 2; P.lane_offset = 0;
    P.hist = d_hist; P.misuse = d_mis; P.first = 0; P.n = N; P.W = R.W; P.HW = R.H * R.W; P.HW5 = 5 * R.H * R.W;
    P.nc_len = (int)R.next_cell.size(); P.max_steps = 100; P.autoreset = 1; P.step_stats = 0; P.isd_shift = 0;
    P.w[0] = 1; P.w[1] = P.w[2] = P.w[3] = 0; P.nb = 1; P.act_pack = 0; for (int c = 0; c < 9; ++c) P.B[c] = c ? __builtin_inf() : 1.0;
    const size_t smem = (kIsdWords + R.next_cell.size()) * 4 + R.lut.size() * 2;
    // actions: T rows cycled
    const int T = 64;
    int8_t* d_act; CK(hipMalloc(&d_act, (size_t)T * 2 * N));
    { std::vector<int8_t> h((size_t)T * 2 * N); for (auto& x : h) x = rand() % 5; CK(hipMemcpy(d_act, h.data(), h.size(), hipMemcpyHostToDevice)); }
    uint16_t* d_obs; int8_t* d_rew; uint8_t* d_term; uint8_t* d_trunc;
    CK(hipMalloc(&d_obs, (size_t)T * N * 2)); CK(hipMalloc(&d_rew, (size_t)T * N)); CK(hipMalloc(&d_term, (size_t)T * N)); CK(hipMalloc(&d_trunc, (size_t)T * N));
    int8_t* d_last; CK(hipMalloc(&d_last, N)); CK(hipMemset(d_last, 0, N));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int step = 0;
    auto io_for = [&](int k) { StepIO io{}; io.act_a = d_act + (size_t)(k % T) * 2 * N; io.act_b = io.act_a + N;
        io.obs = d_obs + (size_t)(k % T) * N; io.reward = d_rew + (size_t)(k % T) * N; io.terminated = d_term + (size_t)(k % T) * N;
        io.truncated = d_trunc + (size_t)(k % T) * N; return io; };
    std::vector<Variant> V;
    auto add = [&](const char* name, std::function<void()> f) { V.push_back(Variant{name, f, {}}); };
    add("empty            grid1024", [&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("tick             grid1024", [&] { hipLaunchKernelGGL(k_tick, dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("stage            grid1024", [&] { hipLaunchKernelGGL(k_stage, dim3(1024), dim3(256), smem, st, P, io_for(step)); });
    add("copy E4                  ", [&] { hipLaunchKernelGGL((k_copy<4, false>), dim3(1024), dim3(256), smem, st, P, io_for(step)); });
    add("copy E4 +stage           ", [&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(1024), dim3(256), smem, st, P, io_for(step)); });
    add("copy E8 +stage           ", [&] { hipLaunchKernelGGL((k_copy<8, true>), dim3(512), dim3(256), smem, st, P, io_for(step)); });
    add("rolled E4 LDS hist       ", [&] { hipLaunchKernelGGL((k_step_rolled<true, true>), dim3(1024), dim3(256), smem, st, P, io_for(step)); });
    add("rolled E4 LDS nohist     ", [&] { hipLaunchKernelGGL((k_step_rolled<true, false>), dim3(1024), dim3(256), smem, st, P, io_for(step)); });
    add("rolled E4 global hist    ", [&] { hipLaunchKernelGGL((k_step_rolled<false, true>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("PRODUCT step (no last_ret)", [&] { hipLaunchKernelGGL((step_kernel<false, false, true, true>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("PRODUCT step LEAN         ", [&] { hipLaunchKernelGGL((step_kernel<false, false, true, true, 1, 256, true>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("LPT4 (4096 waves)         ", [&] { hipLaunchKernelGGL((k_step_lpt<4>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("LPT4 stag b>>8 sleep20    ", [&] { hipLaunchKernelGGL((k_step_lpt<4, 1, 20>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("LPT4 stag b>>8 sleep40    ", [&] { hipLaunchKernelGGL((k_step_lpt<4, 1, 40>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("LPT4 stag b>>8 sleep70    ", [&] { hipLaunchKernelGGL((k_step_lpt<4, 1, 70>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("LPT4 stag b&1  sleep40    ", [&] { hipLaunchKernelGGL((k_step_lpt<4, 2, 40>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("LPT4 stag b>>9 sleep40    ", [&] { hipLaunchKernelGGL((k_step_lpt<4, 3, 40>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("LPT4 stag wave&1 sleep40  ", [&] { hipLaunchKernelGGL((k_step_lpt<4, 4, 40>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("LPT2 stag b>>9 sleep40    ", [&] { hipLaunchKernelGGL((k_step_lpt<2, 3, 40>), dim3(2048), dim3(256), 0, st, P, io_for(step)); });
    add("LPT2 stag b&1 sleep40     ", [&] { hipLaunchKernelGGL((k_step_lpt<2, 2, 40>), dim3(2048), dim3(256), 0, st, P, io_for(step)); });
    add("LPT2 (8192 waves)         ", [&] { hipLaunchKernelGGL((k_step_lpt<2>), dim3(2048), dim3(256), 0, st, P, io_for(step)); });
    add("LPT1 (16384 waves)        ", [&] { hipLaunchKernelGGL((k_step_lpt<1>), dim3(4096), dim3(256), 0, st, P, io_for(step)); });
    add("PRODUCT step + step stats ", [&] { KernelParams Q = P; Q.step_stats = 1; hipLaunchKernelGGL((step_kernel<false, false, true, true>), dim3(1024), dim3(256), 0, st, Q, io_for(step)); });
    add("PRODUCT step + last_return", [&] { StepIO io = io_for(step); io.last_return = d_last; hipLaunchKernelGGL((step_kernel<false, false, true, true>), dim3(1024), dim3(256), 0, st, P, io); });
    add("pipe2 hist               ", [&] { hipLaunchKernelGGL((k_step_pipe<true, 2>), dim3(512), dim3(256), 0, st, P, io_for(step)); });
    add("pipe2 nohist             ", [&] { hipLaunchKernelGGL((k_step_pipe<false, 2>), dim3(512), dim3(256), 0, st, P, io_for(step)); });
    add("pipe4 hist               ", [&] { hipLaunchKernelGGL((k_step_pipe<true, 4>), dim3(256), dim3(256), 0, st, P, io_for(step)); });
    add("pipe1 hist (ref)         ", [&] { hipLaunchKernelGGL((k_step_pipe<true, 1>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("PRODUCT step unroll 2     ", [&] { hipLaunchKernelGGL((step_kernel<false, false, true, true, 2>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("PRODUCT step unroll 4     ", [&] { hipLaunchKernelGGL((step_kernel<false, false, true, true, 4>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("PRODUCT step 2048 x 128   ", [&] { hipLaunchKernelGGL((step_kernel<false, false, true, true, 1, 128>), dim3(2048), dim3(128), 0, st, P, io_for(step)); });
    add("PRODUCT step  512 x 512   ", [&] { hipLaunchKernelGGL((step_kernel<false, false, true, true, 1, 512>), dim3(512), dim3(512), 0, st, P, io_for(step)); });
    add("PRODUCT step  256 x 1024  ", [&] { hipLaunchKernelGGL((step_kernel<false, false, true, true, 1, 1024>), dim3(256), dim3(1024), 0, st, P, io_for(step)); });
    add("PRODUCT step bytes (VEC=0)", [&] { hipLaunchKernelGGL((step_kernel<false, false, false, false>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    add("rolled E4 global nohist  ", [&] { hipLaunchKernelGGL((k_step_rolled<false, false>), dim3(1024), dim3(256), 0, st, P, io_for(step)); });
    // two (or four) independent lane ranges on separate streams: phases of different ranges can overlap
    hipStream_t st2[4]; for (auto& x : st2) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    auto split_launch = [&](int parts, int k) {
        for (int q = 0; q < parts; ++q) {
            KernelParams Q = P; Q.first = (unsigned long long)q * (N / parts); Q.n = N / parts;
            if (q) Q.tick_out = nullptr;
            hipLaunchKernelGGL((step_kernel<false, false, true, true, 1, 256, true>), dim3(1024 / parts), dim3(256), 0, st2[q], Q, io_for(k));
        }
    };
    // reset state so `step` variants act on valid tuples: run the real reset first
    { ResetIO rio{nullptr, nullptr, nullptr}; hipLaunchKernelGGL(reset_kernel<true>, dim3(1024), dim3(256), smem, st, P, rio); CK(hipStreamSynchronize(st)); }
    for (int r = 0; r < ROUNDS; ++r) for (auto& v : V) {
        for (int k = 0; k < 10; ++k) { v.launch(); ++step; }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < K; ++k) { v.launch(); ++step; }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms * 1000.f / K);
        CK(hipGetLastError());
    }
    for (int parts : {2, 4}) {
        std::vector<float> ms;
        for (int r = 0; r < ROUNDS; ++r) {
            for (int k = 0; k < 10; ++k) split_launch(parts, k);
            for (int q = 0; q < parts; ++q) CK(hipStreamSynchronize(st2[q]));
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int k = 0; k < K; ++k) split_launch(parts, k);
            for (int q = 0; q < parts; ++q) CK(hipStreamSynchronize(st2[q]));
            auto t1 = std::chrono::high_resolution_clock::now();
            ms.push_back(std::chrono::duration<float, std::micro>(t1 - t0).count() / K);
        }
        std::sort(ms.begin(), ms.end());
        printf("split in %d ranges on %d streams (host wall): median %.2f min %.2f us per full step\n", parts, parts, ms[ms.size() / 2], ms[0]);
    }
    // graph with a fork/join per step: the step's lane ranges run as parallel branches
    for (int parts : {1, 2, 4}) {
        hipGraph_t graph; hipGraphExec_t exec;
        hipEvent_t fork, join[4]; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        for (auto& e : join) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < K; ++k) {
            if (parts == 1) {
                hipLaunchKernelGGL((step_kernel<false, false, true, true, 1, 256, true>), dim3(1024), dim3(256), 0, st, P, io_for(k));
                continue;
            }
            CK(hipEventRecord(fork, st));
            for (int q = 0; q < parts; ++q) {
                hipStream_t s2 = q == 0 ? st : st2[q];
                if (q) CK(hipStreamWaitEvent(s2, fork, 0));
                KernelParams Q = P; Q.first = (unsigned long long)q * (N / parts); Q.n = N / parts;
                if (q) Q.tick_out = nullptr;
                hipLaunchKernelGGL((step_kernel<false, false, true, true, 1, 256, true>), dim3(1024 / parts), dim3(256), 0, s2, Q, io_for(k));
                if (q) { CK(hipEventRecord(join[q], s2)); CK(hipStreamWaitEvent(st, join[q], 0)); }
            }
        }
        CK(hipStreamEndCapture(st, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        std::vector<float> ms;
        for (int r = 0; r < ROUNDS; ++r) {
            CK(hipGraphLaunch(exec, st)); CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(exec, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t * 1000.f / K);
        }
        std::sort(ms.begin(), ms.end());
        printf("graph, %d parallel lane range(s) per step: median %.2f min %.2f us per full step\n", parts, ms[ms.size() / 2], ms[0]);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    {   // same measurement method for the single-stream kernel, for comparison
        std::vector<float> ms;
        for (int r = 0; r < ROUNDS; ++r) {
            CK(hipStreamSynchronize(st));
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int k = 0; k < K; ++k) hipLaunchKernelGGL((step_kernel<false, false, true, true, 1, 256, true>), dim3(1024), dim3(256), 0, st, P, io_for(k));
            CK(hipStreamSynchronize(st));
            auto t1 = std::chrono::high_resolution_clock::now();
            ms.push_back(std::chrono::duration<float, std::micro>(t1 - t0).count() / K);
        }
        std::sort(ms.begin(), ms.end());
        printf("single stream LEAN (host wall): median %.2f min %.2f us per full step\n", ms[ms.size() / 2], ms[0]);
    }
    printf("%-28s %8s %8s   (us per launch, N=2^20, %d launches x %d rounds)\n", "variant", "median", "min", K, ROUNDS);
    for (auto& v : V) { std::sort(v.ms.begin(), v.ms.end()); printf("%-28s %8.2f %8.2f\n", v.name.c_str(), v.ms[v.ms.size() / 2], v.ms[0]); }
    return 0;
}
