// soccer_hip.hip — C-ABI implementation of libsoccer_hip.so (see include/soccer_hip.h).
//
// Host side: validates arguments the way the reference's asserts do, builds the rule tables
// (soccer_rules.hpp), owns the resident SoA state, and enqueues the kernels of soccer_kernels.hpp
// on the handle's HIP stream.  There is no CPU execution path in this library: every batched_* call
// is a kernel launch, and a missing/failed device is an error, not a fallback.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/soccer_hip.h"
#include "soccer_kernels.hpp"
#include "soccer_rules.hpp"
#include "soccer_slip.hpp"

using namespace soccer;

struct soccer_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t ticks = 0;       // ticks consumed by one replay
    int start_slot = 0;       // tick slot the first captured launch reads
    bool stamped = false;     // soccer_timer_start / _mark were captured: a replay writes stamp slots 0 and 1
};

struct soccer_handle {
    soccer_config cfg{};
    Rules rules;
    KernelParams P{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // device buffers owned by the handle
    uint16_t* d_lut = nullptr; uint32_t* d_nc = nullptr; uint32_t* d_isd = nullptr;
    int8_t* d_policy[2] = {nullptr, nullptr};
    unsigned long long* d_tick = nullptr;   // two slots, 128 B apart
    unsigned long long* d_hist = nullptr;
    unsigned int* d_misuse = nullptr;       // device alias of misuse_host
    unsigned int* misuse_host = nullptr;    // pinned + mapped: kernels store to it only when a frozen lane is stepped (rare),
                                            // the host reads it without a copy
    uint8_t* d_state = nullptr;             // one allocation holding the six SoA streams back to back
    size_t state_stride = 0;                // bytes between consecutive streams
    uint8_t* stage_dev = nullptr;           // staging for the host-pointer entry points
    uint8_t* stage_host = nullptr;          // pinned
    bool mapped = false;                    // SOCCER_F_HOST_MAPPED: d_state and the staging block are pinned host memory
    size_t stage_bytes = 0;
    int tick_slot = 0;                      // slot the NEXT launch reads
    uint64_t tick = 0;                      // host mirror of the device tick
    bool slip = false, lut_lds = false;
    size_t smem_bytes = 0;
    int E = 4;
    int grid_cap = 2048;
    bool capturing = false;
    uint64_t capture_ticks = 0;
    int capture_calls = 0;
    int capture_start_slot = 0;
    int n_cu = 256;
    size_t lds_limit = 64 * 1024;           // hipDeviceProp_t::sharedMemPerBlockOptin: what a workgroup may be given (160 KB on gfx950)
    uint4* d_sub = nullptr;                 // integer slip thresholds (KernelParams::sub)
    uint4* rec_host = nullptr; uint4* rec_dev = nullptr; uint32_t rec_seq = 0;   // soccer_step_scalar's mapped result record
    // byte-parallel step (soccer_swar.hpp)
    swar::Consts swar_c{}; bool swar_ok = false;
    swar::SlipConsts slip_c{}; bool slip_swar_ok = false;   // integer slip selection usable by the byte-parallel kernels
    uint32_t* d_slip_lut = nullptr;         // SlipTables::lut + T for the table form of the selection (when lut_ok)
    uint32_t* d_slip_step_lut = nullptr;    // SlipTables::lut_step + T: the single step's table (when lut_step_ok)
    size_t hist_slots = kHistSlots;         // per-wave histogram slots (a power of two; see soccer_create)
    bool timer_stamped = false; int wall_clock_khz = 100000;   // captured timers: see stamp_kernel
    bool capture_stamped = false;           // THIS capture recorded soccer_timer_start / _mark (what soccer_graph::stamped is copied from)
    bool stamp_poll = false;                // soccer_timer_read may watch the closing stamp of the last soccer_graph_launch change ...
    unsigned long long stamp_prev = 0;      // ... from this value (what the slot held when the replay was enqueued)
    unsigned long long swar_launch_lanes = kSwarLaunchLanes;   // lanes per step_kernel_swar / rollout_swar_kernel launch (SOCCER_SWAR_LAUNCH_LANES: tests of the split)
    int rollout_pref = 0;                   // SOCCER_ROLLOUT=1 (A/B runs, tests of the fallback): never the byte-parallel rollout
    SlipF64* d_slip_f64 = nullptr;          // SLIPM == 3: nominal float64 slip thresholds (step_kernel_swar with caller-supplied uniforms)
    uint32_t* d_worklist = nullptr;         // ... and the groups it leaves to the exact walk: [n / 4] indices + the count behind them
    unsigned long long* d_traj_hist = nullptr;   // soccer_trajectory_returns: u64[3] the kernel adds into
    void* comm = nullptr; int comm_world = 0, comm_rank = 0;   // soccer_comm_init: the RCCL communicator of this handle's device
    unsigned long long* d_comm_scratch = nullptr;   // 64 B for the small reductions (barrier, histogram, clocks)
    PlanIO plan{};                          // cached planner lists (single-agent mode), see build_plan
    std::vector<void*> plan_bufs;
    bool plan_ready = false;
    std::string err;
};

// the handle's host-mapped block: dwords 0 / 1 the sticky misuse words, from byte 64 on SOCCER_STAMP_SLOTS u64 clock stamps,
// ONE PER 64-BYTE LINE: a line the host has written or is polling costs the device a coherence round trip to write, and a
// stamp kernel's store must complete before the next kernel starts — with the opening and the closing stamp of a captured
// timer in one line (and the host clearing the closing one before every replay) the opening stamp's kernel boundary took
// microseconds longer and inflated the region it opens
constexpr size_t kStampStride = 8;          // in u64
constexpr size_t kMappedBytes = 64 + 8 * kStampStride * SOCCER_STAMP_SLOTS;

static thread_local std::string g_err;

static int fail(soccer_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (h) h->err = buf; else g_err = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((h), e_ == hipErrorOutOfMemory ? SOCCER_E_NOMEM : SOCCER_E_HIP,          \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                          \
    } while (0)

template <typename T>
static bool aligned(const T* p, size_t a) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// ------------------------------------------------------------------------------------------------
extern "C" int soccer_abi_version(void) { return SOCCER_ABI_VERSION; }

extern "C" int soccer_device_count(int* count) {
    if (!count) return fail(nullptr, SOCCER_E_INVALID, "count is NULL");
    HIP_TRY(nullptr, hipGetDeviceCount(count));
    return SOCCER_OK;
}

extern "C" const char* soccer_last_error(const soccer_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

static void comm_release(soccer_handle* h);

static void free_handle(soccer_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    if (h->mapped) { if (h->d_state) (void)hipHostFree(h->d_state); h->d_state = nullptr; if (h->stage_host) (void)hipHostFree(h->stage_host); h->stage_host = nullptr; h->stage_dev = nullptr; }
    comm_release(h);
    void* bufs[] = {h->d_state, h->d_lut, h->d_nc, h->d_isd, h->d_policy[0], h->d_policy[1], h->d_tick, h->d_hist, h->stage_dev, h->d_sub, h->d_slip_lut, h->d_slip_step_lut,
                    h->d_traj_hist, h->d_comm_scratch, h->d_slip_f64, h->d_worklist};
    for (void* b : bufs) if (b) (void)hipFree(b);
    for (void* b : h->plan_bufs) if (b) (void)hipFree(b);
    if (h->stage_host) (void)hipHostFree(h->stage_host);
    if (h->rec_host) (void)hipHostFree(h->rec_host);
    if (h->misuse_host) (void)hipHostFree(h->misuse_host);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

static void set_key(soccer_handle* h, uint64_t seed) {
    h->cfg.seed = seed;
    h->P.key0 = static_cast<uint32_t>(seed);
    h->P.key1 = static_cast<uint32_t>(seed >> 32);
}

template <int E, bool SLIP, bool LUT_LDS>
static hipError_t raise_smem_limit(size_t bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_kernel<E, SLIP, LUT_LDS, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_kernel<E, SLIP, LUT_LDS, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

extern "C" int soccer_create(const soccer_config* cfg, soccer_handle** out) {
    if (!cfg || !out) return fail(nullptr, SOCCER_E_INVALID, "cfg/out is NULL");
    *out = nullptr;
    if (cfg->n_lanes < 1) return fail(nullptr, SOCCER_E_INVALID, "n_lanes must be >= 1");
    if (!(cfg->slip_prob >= 0.0 && cfg->slip_prob <= 1.0))
        return fail(nullptr, SOCCER_E_INVALID, "slip_prob must be in [0, 1]");
    if (cfg->max_steps < 1 || cfg->max_steps > 250)
        return fail(nullptr, SOCCER_E_INVALID, "max_steps must be in 1..250 (timestep is a uint8)");
    const uint32_t e = cfg->envs_per_thread;
    if (!(e == 0 || e == 1 || e == 4 || e == 8))
        return fail(nullptr, SOCCER_E_INVALID, "envs_per_thread must be 0, 1, 4 or 8");
    soccer_handle* h = new soccer_handle();
    h->cfg = *cfg;
    const std::string msg = h->rules.build(cfg->width, cfg->height);
    if (!msg.empty()) { delete h; return fail(nullptr, SOCCER_E_INVALID, "%s", msg.c_str()); }
    int ndev = 0;
    hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev < 1) {
        delete h;
        return fail(nullptr, SOCCER_E_HIP, "no HIP device available (%s); libsoccer_hip has no CPU path",
                    de == hipSuccess ? "device count is 0" : hipGetErrorString(de));
    }
    if (cfg->device < 0 || cfg->device >= ndev) {
        delete h; return fail(nullptr, SOCCER_E_INVALID, "device %d out of range (0..%d)", cfg->device, ndev - 1);
    }
#define CREATE_TRY(expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            int code_ = fail(nullptr, e_ == hipErrorOutOfMemory ? SOCCER_E_NOMEM : SOCCER_E_HIP,  \
                             "%s failed: %s", #expr, hipGetErrorString(e_));                      \
            free_handle(h); return code_;                                                         \
        }                                                                                         \
    } while (0)
    CREATE_TRY(hipSetDevice(cfg->device));
    if (cfg->flags & SOCCER_F_NULL_STREAM) { h->stream = nullptr; h->own_stream = false; }
    else if (cfg->stream) { h->stream = static_cast<hipStream_t>(cfg->stream); h->own_stream = false; }
    else { CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    CREATE_TRY(hipEventCreate(&h->ev0));
    CREATE_TRY(hipEventCreate(&h->ev1));

    const Rules& R = h->rules;
    KernelParams& P = h->P;
    const size_t n = cfg->n_lanes;
    const size_t padded = (n + 255) & ~size_t(255);
    h->state_stride = padded;
    h->mapped = (cfg->flags & SOCCER_F_HOST_MAPPED) != 0;
    if (h->mapped) {
        if (n > 4096) { free_handle(h); return fail(nullptr, SOCCER_E_INVALID, "SOCCER_F_HOST_MAPPED is for small handles (n_lanes <= 4096)"); }
        CREATE_TRY(hipHostMalloc(&h->d_state, 6 * padded, hipHostMallocMapped));
    } else {
        CREATE_TRY(hipMalloc(&h->d_state, 6 * padded));
    }
    P.state = h->d_state; P.state_stride = padded;
    // every lane starts needing a reset (:140), parked on the first ISD state so the tuple is valid
    if (h->mapped) {
        const uint8_t init[6] = {(uint8_t)R.isd[0][0], (uint8_t)R.isd[0][1], (uint8_t)R.isd[0][2], (uint8_t)R.isd[0][3], (uint8_t)(2 | R.isd[0][4]), 0};
        for (int k = 0; k < 6; ++k) std::memset(h->d_state + k * padded, init[k], padded);
    } else {
    CREATE_TRY(hipMemsetAsync(h->d_state, R.isd[0][0], padded, h->stream));
    CREATE_TRY(hipMemsetAsync(h->d_state + padded, R.isd[0][1], padded, h->stream));
    CREATE_TRY(hipMemsetAsync(h->d_state + 2 * padded, R.isd[0][2], padded, h->stream));
    CREATE_TRY(hipMemsetAsync(h->d_state + 3 * padded, R.isd[0][3], padded, h->stream));
    CREATE_TRY(hipMemsetAsync(h->d_state + 4 * padded, 2 | R.isd[0][4], padded, h->stream));
    CREATE_TRY(hipMemsetAsync(h->d_state + 5 * padded, 0, padded, h->stream));
    }

    CREATE_TRY(hipMalloc(&h->d_nc, R.next_cell.size() * sizeof(uint32_t)));
    CREATE_TRY(hipMalloc(&h->d_isd, sizeof(R.isd_words)));
    CREATE_TRY(hipMemcpy(h->d_isd, R.isd_words, sizeof(R.isd_words), hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(h->d_nc, R.next_cell.data(), R.next_cell.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    CREATE_TRY(hipMalloc(&h->d_tick, 256));
    CREATE_TRY(hipMemset(h->d_tick, 0, 256));
    // one private histogram slot per wave of the largest grid that counts episodes: the capped grids (rollout, per-lane
    // step) stay below kHistSlots waves; the byte-parallel step launches one wave per 256 lanes, uncapped
    while (h->hist_slots < (n + 255) / 256) h->hist_slots <<= 1;
    CREATE_TRY(hipMalloc(&h->d_hist, sizeof(unsigned long long) * h->hist_slots * kHistStride));
    CREATE_TRY(hipMemset(h->d_hist, 0, sizeof(unsigned long long) * h->hist_slots * kHistStride));
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->misuse_host), kMappedBytes, hipHostMallocMapped));
    std::memset(h->misuse_host, 0, kMappedBytes);
    CREATE_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_misuse), h->misuse_host, 0));

    P.next_cell = h->d_nc; P.isd = h->d_isd;
    P.hist = h->d_hist; P.hist_mask = (uint32_t)(h->hist_slots - 1); P.misuse = h->d_misuse;
    P.lane_offset = cfg->lane_offset;
    P.first = 0; P.n = n; P.W = R.W; P.HW = R.H * R.W; P.HW5 = 5 * R.H * R.W;
    P.nc_len = static_cast<int32_t>(R.next_cell.size());
    P.max_steps = cfg->max_steps;
    P.autoreset = (cfg->flags & SOCCER_F_AUTORESET) ? 1u : 0u;
    P.step_stats = (cfg->flags & SOCCER_F_STEP_STATS) ? 1u : 0u;
    P.isd_shift = R.n_isd == 4 ? 0u : 1u;
    // slip-combination weights, the nominal float64 thresholds of the slip fast path and their integer form
    // (soccer_slip.hpp: host-only, shared with the CPU test of the byte-parallel slip step)
    {
        const SlipTables ST = build_slip_tables(cfg->slip_prob);
        for (int i = 0; i < 4; ++i) P.w[i] = ST.w[i];
        for (int i = 0; i < 9; ++i) { P.B[i] = ST.B[i]; P.CB[i] = ST.CB[i]; }
        P.nb = ST.nb; P.act_pack = ST.act_pack; P.slip_int = ST.slip_int;
        static_assert(sizeof(swar::Quad) == sizeof(uint4), "threshold rows are 16 bytes");
        CREATE_TRY(hipMalloc(&h->d_sub, sizeof(ST.sub)));
        CREATE_TRY(hipMemcpy(h->d_sub, ST.sub, sizeof(ST.sub), hipMemcpyHostToDevice));
        P.sub = h->d_sub;
        h->slip_swar_ok = ST.swar_ok;
        static_assert(kSlipLutWords == kSlipLdsWords && kSlipBuckets == 16384 && kSlipThresholds == 40, "table layout shared with the kernels");
        if (ST.lut_ok) {
            std::vector<uint32_t> img(kSlipLdsWords, 0xFFFFFFFFu);
            std::memcpy(img.data(), ST.lut, kSlipBuckets);
            std::memcpy(img.data() + kSlipBuckets / 4, ST.T, sizeof(ST.T));
            CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_slip_lut), img.size() * sizeof(uint32_t)));
            CREATE_TRY(hipMemcpy(h->d_slip_lut, img.data(), img.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        if (ST.lut_step_ok && !std::getenv("SOCCER_STEP_SLIP_ONE_BY_ONE")) {      // (the variable: tests and A/B runs of the other form)
            std::vector<uint32_t> img(kSlipStepLdsWords, 0xFFFFFFFFu);
            std::memcpy(img.data(), ST.lut_step, kSlipStepBuckets);
            std::memcpy(img.data() + kSlipStepBuckets / 4, ST.T, sizeof(ST.T));
            CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_slip_step_lut), img.size() * sizeof(uint32_t)));
            CREATE_TRY(hipMemcpy(h->d_slip_step_lut, img.data(), img.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        {
            SlipF64 F{};
            for (int i = 0; i < 9; ++i) F.B[i] = ST.B[i];
            for (int i = 0; i < 4; ++i) F.w[i] = ST.w[i];
            F.act_pack = ST.act_pack; F.nb = ST.nb;
            CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_slip_f64), sizeof F));
            CREATE_TRY(hipMemcpy(h->d_slip_f64, &F, sizeof F, hipMemcpyHostToDevice));
        }
        h->slip_c = swar::SlipConsts{};
        for (int i = 0; i < 9; ++i) h->slip_c.CB[i] = ST.CB[i];
        h->slip_c.c_off = ST.c_off;
    }
    set_key(h, cfg->seed);
    h->swar_ok = swar::fits(R.H, R.W, cfg->max_steps);
    if (h->swar_ok) h->swar_c = swar::make_consts(R.H, R.W, R.goal_lo, R.goal_hi, cfg->max_steps, R.n_isd, R.isd, P.autoreset != 0u);
    h->slip = cfg->slip_prob != 0.0;
    if (const char* e2 = std::getenv("SOCCER_ROLLOUT")) h->rollout_pref = std::atoi(e2);
    if (const char* e3 = std::getenv("SOCCER_SWAR_LAUNCH_LANES")) {
        const unsigned long long v = std::strtoull(e3, nullptr, 10) & ~3ull;
        if (v >= 4ull && v <= kSwarLaunchLanes) h->swar_launch_lanes = v;
    }
    h->E = e ? static_cast<int>(e) : 4;

    // Observation table (uint16 index < 65535 bounds it to a few hundred KB): global for the step kernel,
    // staged into LDS by the rollout / reset kernels when it fits next to the move/bounds table.
    const size_t nc_bytes = (R.next_cell.size() + kIsdWords) * sizeof(uint32_t);
    const size_t lut_bytes = R.lut.size() * sizeof(uint16_t);
    if (nc_bytes > 150 * 1024) {
        free_handle(h);
        return fail(nullptr, SOCCER_E_INVALID, "pitch too large: the move/bounds table (%zu bytes) must fit the 160 KB LDS", nc_bytes);
    }
    CREATE_TRY(hipMalloc(&h->d_lut, lut_bytes));
    CREATE_TRY(hipMemcpy(h->d_lut, R.lut.data(), lut_bytes, hipMemcpyHostToDevice));
    P.lut = h->d_lut; P.lut_len = static_cast<int32_t>(R.lut.size());
    h->lut_lds = nc_bytes + lut_bytes <= 150 * 1024;
    h->smem_bytes = nc_bytes + (h->lut_lds ? lut_bytes : 0);
    if (h->smem_bytes > 48 * 1024) {
        hipError_t se = hipSuccess;
        const size_t b = h->smem_bytes;
#define RAISE(EV) if (se == hipSuccess) { se = h->slip ? (h->lut_lds ? raise_smem_limit<EV, true, true>(b) : raise_smem_limit<EV, true, false>(b)) \
                                                       : (h->lut_lds ? raise_smem_limit<EV, false, true>(b) : raise_smem_limit<EV, false, false>(b)); }
        RAISE(1) RAISE(4) RAISE(8)
#undef RAISE
        if (se == hipSuccess)
            se = hipFuncSetAttribute(h->slip ? reinterpret_cast<const void*>(h->lut_lds ? &reset_kernel<true, true> : &reset_kernel<false, true>)
                                             : reinterpret_cast<const void*>(h->lut_lds ? &reset_kernel<true, false> : &reset_kernel<false, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
        CREATE_TRY(se);
    }
    hipDeviceProp_t prop;
    CREATE_TRY(hipGetDeviceProperties(&prop, cfg->device));
    { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, cfg->device) == hipSuccess && khz > 0) h->wall_clock_khz = khz; }
    h->grid_cap = prop.multiProcessorCount * 8;
    h->n_cu = prop.multiProcessorCount;
    if (prop.sharedMemPerBlockOptin > 0) h->lds_limit = prop.sharedMemPerBlockOptin;
    else if (prop.sharedMemPerBlock > 0) h->lds_limit = prop.sharedMemPerBlock;
    CREATE_TRY(hipStreamSynchronize(h->stream));
    CREATE_TRY(hipDeviceSynchronize());      // the hipMemset / hipMemcpy calls above went to the null stream, which a non-blocking stream does not wait for
#undef CREATE_TRY
    *out = h;
    return SOCCER_OK;
}

extern "C" int soccer_destroy(soccer_handle* h) {
    if (!h) return SOCCER_OK;
    free_handle(h);
    return SOCCER_OK;
}

extern "C" int soccer_sync(soccer_handle* h) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_sync during graph capture");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SOCCER_OK;
}

extern "C" int soccer_seed(soccer_handle* h, uint64_t seed) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_seed during graph capture");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    set_key(h, seed);
    h->tick = 0;
    HIP_TRY(h, hipMemsetAsync(h->d_tick, 0, 256, h->stream));
    return SOCCER_OK;
}

extern "C" uint64_t soccer_tick(const soccer_handle* h) { return h ? h->tick : 0; }

// checkpoint / resume: (state streams, seed, tick) fully determine every later result
extern "C" int soccer_set_tick(soccer_handle* h, uint64_t tick) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_set_tick during graph capture");
    if (tick >> 63) return fail(h, SOCCER_E_INVALID, "tick must be below 2^63");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    unsigned long long slots[32] = {0};
    slots[0] = tick; slots[16] = tick;
    HIP_TRY(h, hipMemcpy(h->d_tick, slots, sizeof slots, hipMemcpyHostToDevice));
    h->tick = tick;
    return SOCCER_OK;
}
extern "C" uint64_t soccer_get_seed(const soccer_handle* h) { return h ? h->cfg.seed : 0; }

// the tick lives in device memory so that a captured graph advances it on every replay: launch j
// reads slot (j & 1) and writes slot ((j + 1) & 1)
static void bind_tick(soccer_handle* h, KernelParams& P, uint64_t ticks) {
    P.tick_in = h->d_tick + (h->tick_slot ? 16 : 0);
    P.tick_out = h->d_tick + (h->tick_slot ? 0 : 16);
    h->tick_slot ^= 1;
    if (h->capturing) { h->capture_ticks += ticks; h->capture_calls += 1; }
    else h->tick += ticks;
}

static int grid_for(const soccer_handle* h, uint64_t work_items) {
    uint64_t blocks = (work_items + kBlock - 1) / kBlock;
    if (blocks < 1) blocks = 1;
    if (blocks > (uint64_t)h->grid_cap) blocks = h->grid_cap;
    return static_cast<int>(blocks);
}

// ------------------------------------------------------------------------------------------------
extern "C" int batched_reset(soccer_handle* h, const uint8_t* mask, const double* u_reset, uint16_t* obs) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (!aligned(u_reset, 8) || !aligned(obs, 2))
        return fail(h, SOCCER_E_INVALID, "batched_reset: u_reset must be 8-byte and obs 2-byte aligned");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    KernelParams P = h->P;
    bind_tick(h, P, 1);
    ResetIO io{mask, u_reset, obs};
    // the byte-parallel kernel over the 4-aligned part (Philox draws, dword-aligned streams); whatever is left — a ragged
    // tail, or everything — through the per-lane kernel on the same tick
    unsigned long long n4 = 0;
    if (h->swar_ok && !u_reset && h->E != 1 && (P.lane_offset & 3ull) == 0ull && aligned(mask, 4) && aligned(obs, 8)) n4 = P.n & ~3ull;
    if (n4) {
        ResetSwar RS{h->swar_c, P.state, P.state_stride, n4, P.lane_offset, P.tick_in, P.tick_out, P.key0, P.key1, mask, obs};
        const dim3 g(static_cast<unsigned>(((n4 >> 2) + kBlock - 1) / kBlock)), b(kBlock);
        if (h->slip) { if (mask) hipLaunchKernelGGL((reset_kernel_swar<true, true>), g, b, 0, h->stream, RS);
                       else hipLaunchKernelGGL((reset_kernel_swar<false, true>), g, b, 0, h->stream, RS); }
        else { if (mask) hipLaunchKernelGGL((reset_kernel_swar<true, false>), g, b, 0, h->stream, RS);
               else hipLaunchKernelGGL((reset_kernel_swar<false, false>), g, b, 0, h->stream, RS); }
    }
    if (n4 < P.n) {
        KernelParams Q = P;
        Q.first = n4; Q.n = P.n - n4;
        if (n4) Q.tick_out = nullptr;           // the main launch publishes the tick
        const int grid = grid_for(h, Q.n);
        if (h->slip) { if (h->lut_lds) hipLaunchKernelGGL((reset_kernel<true, true>), dim3(grid), dim3(kBlock), h->smem_bytes, h->stream, Q, io);
                       else hipLaunchKernelGGL((reset_kernel<false, true>), dim3(grid), dim3(kBlock), h->smem_bytes, h->stream, Q, io); }
        else { if (h->lut_lds) hipLaunchKernelGGL((reset_kernel<true, false>), dim3(grid), dim3(kBlock), h->smem_bytes, h->stream, Q, io);
               else hipLaunchKernelGGL((reset_kernel<false, false>), dim3(grid), dim3(kBlock), h->smem_bytes, h->stream, Q, io); }
    }
    HIP_TRY(h, hipGetLastError());
    return SOCCER_OK;
}

template <class T> static inline T* off(T* p, unsigned long long lanes) { return p ? p + lanes : nullptr; }   // NULL stays NULL

// the work list of step_kernel_swar<.., SLIPM = 3, ..>: one index per 4-lane group of the handle, the count 16 bytes behind them
static bool ensure_worklist(soccer_handle* h) {
    if (h->d_worklist) return true;
    const size_t words = (size_t)(h->P.n >> 2) + 8;
    if (hipMalloc(reinterpret_cast<void**>(&h->d_worklist), words * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); h->d_worklist = nullptr; return false; }
    // (on the handle's own stream: a memset on the null stream is not ordered with a non-blocking stream's kernels)
    if (hipMemsetAsync(h->d_worklist, 0, words * sizeof(uint32_t), h->stream) != hipSuccess) { (void)hipGetLastError(); return false; }
    return true;
}

template <bool EXPLICIT_U, bool VEC, bool SHARED>
static void launch_step3(soccer_handle* h, const KernelParams& P, const StepIO& io) {
    const int grid = grid_for(h, (P.n + 3) / 4);
    const dim3 g(grid), b(kBlock);
    if (h->slip) hipLaunchKernelGGL((step_kernel<true, EXPLICIT_U, VEC, SHARED>), g, b, 0, h->stream, P, io);
    else hipLaunchKernelGGL((step_kernel<false, EXPLICIT_U, VEC, SHARED>), g, b, 0, h->stream, P, io);
}
static void launch_step(soccer_handle* h, const KernelParams& P, const StepIO& io, bool explicit_u, bool vec) {
    const bool shared = ((P.lane_offset + P.first) & 3ull) == 0ull;
    const bool policy_only = explicit_u && !io.u_step && !io.u_reset;       // fixed-policy handle, Philox draws
    const bool swar_fit = vec && shared && h->swar_ok && (!h->slip || h->slip_swar_ok) && aligned(io.last_return, 4);
    // caller-supplied uniforms at slip_prob == 0: floor(4u) is the reference's decision for any double (step_kernel_swar, EXPL)
    const bool expl = explicit_u && !policy_only && !h->slip && aligned(io.u_step, 16) && aligned(io.u_reset, 16);
    // ... and at slip_prob > 0 the float64 decision against the nominal thresholds (SLIPM = 3); the groups it cannot decide safely go
    // to the per-lane kernel's exact walk through a work list (one extra small launch per call)
    const bool expl_slip = explicit_u && !policy_only && h->slip && io.u_step && aligned(io.u_step, 16) && aligned(io.u_reset, 16) &&
                           vec && shared && h->swar_ok && aligned(io.last_return, 4) && (P.n >> 2) < 0xffffffffull &&
                           (h->d_worklist || (!h->capturing && ensure_worklist(h)));      // (no allocation inside a capture: the per-lane kernel then)
    if (((policy_only || !explicit_u || expl) && swar_fit) || expl_slip) {
        // the byte-parallel kernel (four lanes stay packed in their dwords, no rule-table reads)
        // which outputs the launch needs decides the instantiation: 0 the four result streams, 1 + the gym floats /
        // finished / last_return, 2 + final_obs / prob_code / episode histogram
        const int out = (io.prob_code || io.final_obs || P.step_stats) ? 2
                      : (io.reward_a_f32 || io.reward_b_f32 || io.finished || io.last_return) ? 1 : 0;
        const dim3 b(kBlock);
#define SWAR_ARGS P.state + c0, P.state_stride, off(io.act_a, c0), off(io.act_b, c0), (h->capturing ? P.tick_in : nullptr), cn, (unsigned long long)(h->tick - 1), Q
#define SWAR_GO(OV, SV, PV, XV) do { if (h->swar_c.small) hipLaunchKernelGGL((step_kernel_swar<OV, SV, PV, 1, XV>), gh, b, 0, h->stream, SWAR_ARGS); \
                                     else hipLaunchKernelGGL((step_kernel_swar<OV, SV, PV, 0, XV>), gh, b, 0, h->stream, SWAR_ARGS); } while (0)
#define SWAR_SLIP(OV, PV) do { if (expl_slip) SWAR_GO(OV, 3, PV, true); else if (expl) SWAR_GO(OV, 0, PV, true); else if (!h->slip) SWAR_GO(OV, 0, PV, false); \
                               else if (h->d_slip_step_lut) SWAR_GO(OV, 2, PV, false); else SWAR_GO(OV, 1, PV, false); } while (0)
#define SWAR_OUT(PV) do { if (out == 2) SWAR_SLIP(2, PV); else if (out == 1) SWAR_SLIP(1, PV); else SWAR_SLIP(0, PV); } while (0)
        // The kernel's byte offsets are 32-bit (soccer_kernels.hpp): a handle beyond kSwarLaunchLanes lanes is stepped by
        // several launches on the same tick, each handed its part of every stream; only the last one publishes the tick.
        for (unsigned long long c0 = P.first; c0 < P.first + P.n; c0 += h->swar_launch_lanes) {
            const unsigned long long cn = std::min<unsigned long long>(h->swar_launch_lanes, P.first + P.n - c0);
            const bool last = c0 + cn == P.first + P.n;
            const dim3 gh(static_cast<unsigned>(((cn >> 2) + kBlock - 1) / kBlock));
            SwarParams Q{h->swar_c, P.key0, P.key1, P.lane_offset + c0, 0ull, last ? P.tick_out : nullptr, P.misuse,
                         P.step_stats ? P.hist : nullptr, P.hist_mask,
                         h->slip_c, reinterpret_cast<const swar::Quad*>(P.sub), h->d_slip_step_lut,
                         (h->cfg.flags & SOCCER_F_STREAM_ACTIONS) ? 1u : 0u, P.policy_a, P.policy_b,
                         off(io.obs, c0), off(io.reward, c0), off(io.terminated, c0), off(io.truncated, c0), off(io.prob_code, c0),
                         off(io.final_obs, c0), off(io.reward_a_f32, c0), off(io.reward_b_f32, c0), off(io.finished, c0),
                         off(io.last_return, c0), off(io.u_step, c0), off(io.u_reset, c0),
                         h->d_slip_f64, h->d_worklist, h->d_worklist ? h->d_worklist + (h->P.n >> 2) + 4 : nullptr};
            if (P.policy_a || P.policy_b) SWAR_OUT(true); else SWAR_OUT(false);
            if (expl_slip) {
                // the groups of THIS part that were listed: the per-lane kernel, one workgroup, same tick (it publishes nothing)
                KernelParams R = P; R.first = c0; R.n = cn; R.tick_out = nullptr;
                StepIO jo = io; jo.worklist = h->d_worklist; jo.work_count = h->d_worklist + (h->P.n >> 2) + 4;
                hipLaunchKernelGGL((step_kernel<true, true, true, true>), dim3(1), dim3(kBlock), 0, h->stream, R, jo);
            }
        }
#undef SWAR_OUT
#undef SWAR_SLIP
#undef SWAR_GO
#undef SWAR_ARGS
    } else if (explicit_u) {    // caller-supplied uniforms (facade, tests) and fixed-policy handles beyond the byte arithmetic: generic kernel
        if (vec && shared) launch_step3<true, true, true>(h, P, io); else launch_step3<true, false, false>(h, P, io);
    } else if (vec && shared) {
        // the hot instantiations of the per-lane kernel (slip handles, pitches beyond the byte arithmetic);
        // LEAN drops the code for prob_code / final_obs / last_return / step stats
        const bool lean = !io.prob_code && !io.final_obs && !io.last_return && !io.reward_a_f32 && !io.reward_b_f32 && !io.finished && !P.step_stats;
        const int grid = grid_for(h, (P.n + 3) / 4);
        const dim3 g(grid), b(kBlock);
        if (lean) {                 // one 4-lane group per thread, as many workgroups as it takes
            const unsigned long long blocks = ((P.n >> 2) + kBlock - 1) / kBlock;
            const dim3 gh(static_cast<unsigned>(blocks));
#define HOT_ARGS P.state, P.state_stride, io.act_a, io.act_b, (h->capturing ? P.tick_in : nullptr), P.n, (unsigned long long)(h->tick - 1), P, io
            if (h->slip && P.slip_int == 1u) hipLaunchKernelGGL((step_kernel_hot<true, true>), gh, b, 0, h->stream, HOT_ARGS);
            else if (h->slip) hipLaunchKernelGGL(step_kernel_hot<true>, gh, b, 0, h->stream, HOT_ARGS);
            else hipLaunchKernelGGL(step_kernel_hot<false>, gh, b, 0, h->stream, HOT_ARGS);
#undef HOT_ARGS
        } else launch_step3<false, true, true>(h, P, io);
    }
    else if (vec) launch_step3<false, true, false>(h, P, io);
    else launch_step3<false, false, false>(h, P, io);
}

extern "C" int batched_step_ex(soccer_handle* h, const soccer_step_args* a) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (!a || (!a->act_a && !h->P.policy_a) || (!a->act_b && !h->P.policy_b))
        return fail(h, SOCCER_E_INVALID, "batched_step: an action stream is required for every player without a fixed policy");
    if (!aligned(a->u_step, 8) || !aligned(a->u_reset, 8) || !aligned(a->obs, 2) || !aligned(a->final_obs, 2) ||
        !aligned(a->reward_a_f32, 4) || !aligned(a->reward_b_f32, 4))
        return fail(h, SOCCER_E_INVALID, "batched_step: u_* must be 8-byte, reward_*_f32 4-byte and obs/final_obs 2-byte aligned");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    // dword I/O needs every byte stream 4-aligned and the uint16 streams 8-aligned; else byte I/O
    const bool vec = h->E != 1 && aligned(a->act_a, 4) && aligned(a->act_b, 4) && aligned(a->reward, 4) &&
                     aligned(a->terminated, 4) && aligned(a->truncated, 4) && aligned(a->prob_code, 4) &&
                     aligned(a->obs, 8) && aligned(a->final_obs, 8) && aligned(a->reward_a_f32, 16) && aligned(a->reward_b_f32, 16) &&
                     aligned(a->finished, 4);
    const bool explicit_u = a->u_step || a->u_reset || h->P.policy_a || h->P.policy_b;   // generic kernel
    KernelParams P = h->P;
    bind_tick(h, P, 1);
    StepIO io{a->act_a, a->act_b, a->u_step, a->u_reset, a->obs, a->reward, a->terminated, a->truncated,
              a->prob_code, a->final_obs, a->last_return, a->reward_a_f32, a->reward_b_f32, a->finished, nullptr, nullptr};
    const unsigned long long n = h->P.n, n4 = vec ? (n & ~3ull) : 0ull;
    if (n4) { P.first = 0; P.n = n4; launch_step(h, P, io, explicit_u, true); }
    if (n4 < n) {               // ragged tail (or everything, when the buffers are not dword-aligned)
        KernelParams Q = P;
        Q.first = n4; Q.n = n - n4;
        if (n4) Q.tick_out = nullptr;   // same tick as the main launch, which publishes it
        launch_step(h, Q, io, explicit_u, false);
    }
    HIP_TRY(h, hipGetLastError());
    return SOCCER_OK;
}

extern "C" int batched_step(soccer_handle* h, const int8_t* act_a, const int8_t* act_b, uint16_t* obs,
                            int8_t* reward, uint8_t* terminated, uint8_t* truncated, uint8_t* prob_code) {
    soccer_step_args a{};
    a.act_a = act_a; a.act_b = act_b; a.obs = obs; a.reward = reward;
    a.terminated = terminated; a.truncated = truncated; a.prob_code = prob_code;
    return batched_step_ex(h, &a);
}

template <int E, bool DYN>
static void launch_rollout2(soccer_handle* h, const KernelParams& P, const RolloutIO& io) {
    const int grid = grid_for(h, (P.n + E - 1) / E);
    const dim3 g(grid), b(kBlock);
    if (h->slip) {
        if (h->lut_lds) hipLaunchKernelGGL((rollout_kernel<E, true, true, DYN>), g, b, h->smem_bytes, h->stream, P, io);
        else hipLaunchKernelGGL((rollout_kernel<E, true, false, DYN>), g, b, h->smem_bytes, h->stream, P, io);
    } else {
        if (h->lut_lds) hipLaunchKernelGGL((rollout_kernel<E, false, true, DYN>), g, b, h->smem_bytes, h->stream, P, io);
        else hipLaunchKernelGGL((rollout_kernel<E, false, false, DYN>), g, b, h->smem_bytes, h->stream, P, io);
    }
}
template <int E>
static void launch_rollout(soccer_handle* h, const KernelParams& P, const RolloutIO& io) {
    // DYN: some action is produced in the kernel (sampling, mixed policy, fixed policy)
    const bool dyn = io.sample_actions || P.policy_a || P.policy_b;
    if (dyn) launch_rollout2<E, true>(h, P, io); else launch_rollout2<E, false>(h, P, io);
}

extern "C" int batched_rollout(soccer_handle* h, const soccer_rollout_args* a) { return batched_rollout_ex(h, a, nullptr); }

extern "C" int batched_rollout_ex(soccer_handle* h, const soccer_rollout_args* a, const soccer_rollout_extra* x) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (!a || a->n_steps < 1) return fail(h, SOCCER_E_INVALID, "batched_rollout: n_steps must be >= 1");
    if (!a->sample_actions && ((!a->act_a && !h->P.policy_a) || (!a->act_b && !h->P.policy_b)))
        return fail(h, SOCCER_E_INVALID, "batched_rollout: an action stream is required for every player without a fixed policy (or sample_actions)");
    if (!a->sample_actions && a->act_stride < (int64_t)h->P.n)
        return fail(h, SOCCER_E_INVALID, "batched_rollout: act_stride must be >= n_lanes");
    uint16_t* x_fin = x ? x->final_obs : nullptr; uint8_t* x_code = x ? x->prob_code : nullptr;
    const bool any_out = a->obs || a->reward || a->terminated || a->truncated || x_fin || x_code;
    if (!aligned(x_fin, 2)) return fail(h, SOCCER_E_INVALID, "batched_rollout_ex: final_obs must be 2-byte aligned");
    if (any_out && a->out_stride < (int64_t)h->P.n)
        return fail(h, SOCCER_E_INVALID, "batched_rollout: out_stride must be >= n_lanes");
    if (!aligned(a->obs, 2) || !aligned(a->return_sum, 4) || !aligned(a->episode_count, 4) ||
        !aligned(a->mix_a, 8) || !aligned(a->mix_b, 8))
        return fail(h, SOCCER_E_INVALID, "batched_rollout: misaligned obs/return_sum/episode_count/mix_*");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    int E = h->E;
    auto ok = [&](int e) {
        const bool strides = (a->sample_actions || a->act_stride % e == 0) && (!any_out || a->out_stride % e == 0);
        return strides && aligned(a->act_a, e) && aligned(a->act_b, e) && aligned(a->reward, e) &&
               aligned(a->terminated, e) && aligned(a->truncated, e) && aligned(a->obs, 2 * e) && aligned(x_code, e) && aligned(x_fin, 2 * e) &&
               aligned(a->return_sum, 4 * e) && aligned(a->episode_count, 4 * e);
    };
    while (E > 1 && !ok(E)) E = E == 4 ? 1 : E / 2;
    // one launch covers at most kChunk steps (per-thread episode counters are 16 bit wide); the tick
    // sequence of consecutive launches is contiguous, so chunking does not change any result
    constexpr int kChunk = 4096;
    const Rules& R0 = h->rules;
    for (int s0 = 0; s0 < a->n_steps; s0 += kChunk) {
        const int ns = a->n_steps - s0 < kChunk ? a->n_steps - s0 : kChunk;
        KernelParams P = h->P;
        bind_tick(h, P, (uint64_t)ns);
        const long long ao = (long long)s0 * a->act_stride, oo = (long long)s0 * a->out_stride;
        RolloutIO io{ns, a->sample_actions, a->mix_a, a->mix_b, a->act_a ? a->act_a + ao : nullptr, a->act_b ? a->act_b + ao : nullptr,
                     (long long)a->act_stride, a->obs ? a->obs + oo : nullptr, a->reward ? a->reward + oo : nullptr,
                     a->terminated ? a->terminated + oo : nullptr, a->truncated ? a->truncated + oo : nullptr,
                     (long long)a->out_stride, a->return_sum, a->episode_count, x_fin ? x_fin + oo : nullptr, x_code ? x_code + oo : nullptr};
        // the byte-parallel rollout: every pitch that fits the byte arithmetic, slip 0 or an exact integer slip decision
        // (a lane count that is not a multiple of 4: the byte-parallel kernel over the first n & ~3 lanes, the one to three
        // left over through the per-lane kernel on the same ticks, like batched_step's ragged tail)
        const bool swar_roll = h->swar_ok && (!h->slip || h->slip_swar_ok) && P.n >= 4ull && ((P.lane_offset + P.first) & 3ull) == 0ull &&
                               ok(4) && h->rollout_pref != 1;
        const unsigned long long n_all = P.n, n4 = swar_roll ? (P.n & ~3ull) : 0ull;
        if (swar_roll) {
            P.n = n4;
            const bool dyn = io.sample_actions || P.policy_a || P.policy_b;
            RolloutSwar RS{P.state, P.state_stride, P.first, P.n, P.lane_offset, P.tick_in, P.tick_out, P.hist, P.misuse,
                           P.policy_a, P.policy_b, P.key0, P.key1,
                           h->swar_c, h->slip_c, reinterpret_cast<const swar::Quad*>(P.sub), P.hist_mask, R0.nS, 0, 0u, 0u, h->d_slip_lut};
            const int sm = !h->slip ? 0 : (h->d_slip_lut ? 2 : 1);    // slip selection: none / threshold by threshold / by table
            size_t smem = 36 * sizeof(uint32_t);        // (the bucket table of sm == 2 is static LDS of the kernel)
            RS.tab_off = (uint32_t)(smem / sizeof(uint32_t));
            const bool fixed = P.policy_a || P.policy_b;
            // both sides sampled from mixed-policy tables whose 16-byte rows fit LDS: the shape of config 5
            // what the tables may take: the device's per-workgroup LDS limit (64 KB on CDNA3, 160 KB on gfx950 — never a literal)
            // minus the action staging area that is added below and the static bucket table of sm == 2
            const size_t staging = io.sample_actions ? 0 : 16 * kBlock * sizeof(uint32_t) + 16;
            const size_t lds_cap = h->lds_limit > staging + (sm == 2 ? kSlipLutWords * sizeof(uint32_t) : 0)
                                 ? h->lds_limit - staging - (sm == 2 ? kSlipLutWords * sizeof(uint32_t) : 0) : 0;
            const bool both_mix = dyn && !fixed && io.sample_actions && io.mix_a && io.mix_b &&
                                  smem + (size_t)R0.nS * sizeof(uint4) <= lds_cap;
            if (both_mix) { RS.lds_tables = 1; smem += (size_t)R0.nS * sizeof(uint4); }
            else if (dyn && (io.mix_a || io.mix_b || fixed)) {
                const size_t need = smem + 2 * (size_t)R0.nS * sizeof(uint2) + 2 * (((size_t)R0.nS + 15) & ~size_t(15));
                if (need <= lds_cap) { RS.lds_tables = 1; smem = need; }     // else: the tables stay in global memory
            }
            if (!io.sample_actions) {        // action streams are staged through LDS: 16 dwords per thread
                smem = (smem + 15) & ~size_t(15);
                RS.act_off = (uint32_t)(smem / sizeof(uint32_t));
                smem += 16 * kBlock * sizeof(uint32_t);
            }
            // The kernel's byte offsets are 32-bit: a handle beyond kSwarLaunchLanes lanes is rolled out part by part (lanes never
            // interact), every part over the same ticks, each handed its piece of every stream; the last one publishes the tick.
            const RolloutSwar RS0 = RS; const RolloutIO io0 = io;
            for (unsigned long long c0 = 0; c0 < n4; c0 += h->swar_launch_lanes) {
            const unsigned long long cn = std::min<unsigned long long>(h->swar_launch_lanes, n4 - c0);
            RS = RS0; io = io0;
            RS.state = RS0.state + P.first + c0; RS.first = 0ull; RS.n = cn; RS.lane_offset = RS0.lane_offset + P.first + c0;
            if (c0 + cn < n4) RS.tick_out = nullptr;
            const unsigned long long lane0 = P.first + c0;
            io.act_a = off(io0.act_a, lane0); io.act_b = off(io0.act_b, lane0); io.obs = off(io0.obs, lane0); io.reward = off(io0.reward, lane0);
            io.terminated = off(io0.terminated, lane0); io.truncated = off(io0.truncated, lane0);
            io.return_sum = off(io0.return_sum, lane0); io.episode_count = off(io0.episode_count, lane0);
            io.final_obs = off(io0.final_obs, lane0); io.prob_code = off(io0.prob_code, lane0);
            const uint64_t groups = cn >> 2;
            uint64_t blocks = (groups + kBlock - 1) / kBlock;
            if (blocks > (uint64_t)h->grid_cap) blocks = h->grid_cap;
            const dim3 g((unsigned)blocks), bl(kBlock);
#define LAUNCH_F(DV, SV, GV, FV) do { if (smem > 48 * 1024) HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_swar_kernel<DV, SV, GV, FV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
                                      hipLaunchKernelGGL((rollout_swar_kernel<DV, SV, GV, FV>), g, bl, smem, h->stream, RS, io); } while (0)
#define LAUNCH_G(DV, SV, GV) do { if (io.final_obs || io.prob_code) LAUNCH_F(DV, SV, GV, true); else LAUNCH_F(DV, SV, GV, false); } while (0)
#define LAUNCH_S(DV, SV) do { if (h->swar_c.small) LAUNCH_G(DV, SV, 1); else LAUNCH_G(DV, SV, 0); } while (0)
            // the action source as a compile-time shape (rollout_swar_group): streams / sampled uniformly / both sides from
            // mixed-policy tables / single-agent A or B / anything else
            const int dm = !dyn ? 0 : (!fixed && io.sample_actions && !io.mix_a && !io.mix_b) ? 1
                                : both_mix ? 2
                                : (!io.sample_actions && P.policy_a && !P.policy_b && io.act_b) ? 4
                                : (!io.sample_actions && P.policy_b && !P.policy_a && io.act_a) ? 5 : 3;
#define LAUNCH_D(SV) do { if (dm == 0) LAUNCH_S(0, SV); else if (dm == 1) LAUNCH_S(1, SV); else if (dm == 2) LAUNCH_S(2, SV); \
                          else if (dm == 4) LAUNCH_S(4, SV); else if (dm == 5) LAUNCH_S(5, SV); else LAUNCH_S(3, SV); } while (0)
            if (sm == 0) LAUNCH_D(0); else if (sm == 1) LAUNCH_D(1); else LAUNCH_D(2);
#undef LAUNCH_D
#undef LAUNCH_S
#undef LAUNCH_G
#undef LAUNCH_F
            }
            if (n4 < n_all) {
                KernelParams Q = h->P;
                Q.tick_in = P.tick_in; Q.tick_out = nullptr;      // the main launch publishes the tick
                Q.first = n4; Q.n = n_all - n4;
                launch_rollout<1>(h, Q, io0);           // (io0: the loop above left `io` offset to its last part; the per-lane kernel indexes by absolute lane)
            }
        } else switch (E) {
            case 8: launch_rollout<8>(h, P, io); break;
            case 4: launch_rollout<4>(h, P, io); break;
            default: launch_rollout<1>(h, P, io); break;
        }
        HIP_TRY(h, hipGetLastError());
    }
    return SOCCER_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int soccer_set_state(soccer_handle* h, const int8_t* row_a, const int8_t* col_a, const int8_t* row_b,
                                const int8_t* col_b, const uint8_t* poss, const uint8_t* t,
                                const uint8_t* needs_reset) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_set_state during graph capture");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t n = h->P.n, S = h->state_stride;
    const Rules& R = h->rules;
    // current device copy of whatever is not supplied, so the resulting tuple can be validated
    std::vector<uint8_t> img(6 * S);
    if (h->mapped) std::memcpy(img.data(), h->d_state, 6 * S);
    else HIP_TRY(h, hipMemcpy(img.data(), h->d_state, 6 * S, hipMemcpyDeviceToHost));
    int8_t* ra = reinterpret_cast<int8_t*>(img.data());
    int8_t* ca = ra + S; int8_t* rb = ra + 2 * S; int8_t* cb = ra + 3 * S;
    uint8_t* ps = img.data() + 4 * S; uint8_t* tt = img.data() + 5 * S;
    for (size_t i = 0; i < n; ++i) {
        if (row_a) ra[i] = row_a[i];
        if (col_a) ca[i] = col_a[i];
        if (row_b) rb[i] = row_b[i];
        if (col_b) cb[i] = col_b[i];
        uint8_t p = ps[i] & 1, nr = (ps[i] >> 1) & 1;
        if (poss) { if (poss[i] > 1) return fail(h, SOCCER_E_INVALID, "lane %zu: possession must be 0 or 1", i); p = poss[i]; }
        if (needs_reset) nr = needs_reset[i] ? 1 : 0;
        ps[i] = static_cast<uint8_t>(p | (nr << 1));
        if (t) { if (t[i] > h->cfg.max_steps) return fail(h, SOCCER_E_INVALID, "lane %zu: timestep %d > max_steps", i, (int)t[i]); tt[i] = t[i]; }
        const bool in_range = ra[i] >= 0 && ra[i] < R.H && rb[i] >= 0 && rb[i] < R.H &&
                              ca[i] >= 0 && ca[i] < R.W && cb[i] >= 0 && cb[i] < R.W;
        // the reference raises KeyError when stepping from a tuple it has no table entry for
        if (!in_range || R.kind[R.flat(ra[i], ca[i], rb[i], cb[i], p)] == 0)
            return fail(h, SOCCER_E_INVALID, "lane %zu: state (%d, %d, %d, %d, %d) is not a reachable state tuple",
                        i, (int)ra[i], (int)ca[i], (int)rb[i], (int)cb[i], (int)p);
    }
    if (h->mapped) std::memcpy(h->d_state, img.data(), 6 * S);
    else HIP_TRY(h, hipMemcpy(h->d_state, img.data(), 6 * S, hipMemcpyHostToDevice));
    return SOCCER_OK;
}

extern "C" int soccer_host_view(soccer_handle* h, uint8_t** state, uint64_t* stride) {
    if (!h || !state || !stride) return fail(h, SOCCER_E_INVALID, "handle/state/stride is NULL");
    if (!h->mapped) return fail(h, SOCCER_E_STATE, "soccer_host_view needs a SOCCER_F_HOST_MAPPED handle");
    *state = h->d_state; *stride = h->state_stride;
    return SOCCER_OK;
}

extern "C" int soccer_get_state(soccer_handle* h, int8_t* row_a, int8_t* col_a, int8_t* row_b, int8_t* col_b,
                                uint8_t* poss, uint8_t* t, uint8_t* needs_reset) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_get_state during graph capture");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t n = h->P.n, S = h->state_stride;
    std::vector<uint8_t> img(6 * S);
    if (h->mapped) std::memcpy(img.data(), h->d_state, 6 * S);
    else HIP_TRY(h, hipMemcpy(img.data(), h->d_state, 6 * S, hipMemcpyDeviceToHost));
    if (row_a) std::memcpy(row_a, img.data(), n);
    if (col_a) std::memcpy(col_a, img.data() + S, n);
    if (row_b) std::memcpy(row_b, img.data() + 2 * S, n);
    if (col_b) std::memcpy(col_b, img.data() + 3 * S, n);
    if (t) std::memcpy(t, img.data() + 5 * S, n);
    const uint8_t* ps = img.data() + 4 * S;
    for (size_t i = 0; i < n; ++i) {
        if (poss) poss[i] = ps[i] & 1;
        if (needs_reset) needs_reset[i] = (ps[i] >> 1) & 1;
    }
    return SOCCER_OK;
}

// ---- host-pointer entry points: stage through one pinned block, one copy each way ---------------
namespace {
struct StageLayout {
    size_t act_a, act_b, mask, u_step, u_reset;            // inputs (the action streams first: they always travel)
    size_t obs, final_obs, reward, term, trunc, code;      // outputs
    size_t out_begin, total;
};
StageLayout stage_layout(size_t n) {
    auto up = [](size_t x) { return (x + 63) & ~size_t(63); };
    StageLayout L{};
    size_t o = 0;
    L.act_a = o; o = up(o + n);
    L.act_b = o; o = up(o + n);
    L.mask = o; o = up(o + n);
    L.u_step = o; o = up(o + 8 * n);
    L.u_reset = o; o = up(o + 8 * n);
    L.out_begin = o;
    L.obs = o; o = up(o + 2 * n);
    L.final_obs = o; o = up(o + 2 * n);
    L.reward = o; o = up(o + n);
    L.term = o; o = up(o + n);
    L.trunc = o; o = up(o + n);
    L.code = o; o = up(o + n);
    L.total = o;
    return L;
}
int ensure_stage(soccer_handle* h, const StageLayout& L) {
    if (h->stage_bytes >= L.total) return SOCCER_OK;
    if (h->mapped) {            // the kernel reads inputs from / writes outputs to the pinned block in place
        HIP_TRY(h, hipHostMalloc(&h->stage_host, L.total, hipHostMallocMapped));
        h->stage_dev = h->stage_host;
    } else {
        HIP_TRY(h, hipMalloc(&h->stage_dev, L.total));
        HIP_TRY(h, hipHostMalloc(&h->stage_host, L.total, hipHostMallocDefault));
    }
    h->stage_bytes = L.total;
    return SOCCER_OK;
}
}  // namespace

extern "C" int soccer_staging(soccer_handle* h, soccer_staging_view* v) {
    if (!h || !v) return fail(h, SOCCER_E_INVALID, "handle/view is NULL");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const StageLayout L = stage_layout(h->P.n);
    if (int rc = ensure_stage(h, L)) return rc;
    uint8_t* H = h->stage_host;
    v->act_a = reinterpret_cast<int8_t*>(H + L.act_a); v->act_b = reinterpret_cast<int8_t*>(H + L.act_b);
    v->mask = H + L.mask;
    v->u_step = reinterpret_cast<double*>(H + L.u_step); v->u_reset = reinterpret_cast<double*>(H + L.u_reset);
    v->obs = reinterpret_cast<uint16_t*>(H + L.obs); v->final_obs = reinterpret_cast<uint16_t*>(H + L.final_obs);
    v->reward = reinterpret_cast<int8_t*>(H + L.reward); v->terminated = H + L.term; v->truncated = H + L.trunc;
    v->prob_code = H + L.code;
    return SOCCER_OK;
}

// every action byte of a host array must be 0..4 (the reference indexes ACTION_STRING with it: IndexError, :393)
static long first_bad_action(const int8_t* a, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {                    // eight bytes at a time: any bit above 2 set, or 5..7
        uint64_t v; std::memcpy(&v, a + i, 8);
        if ((v & 0xF8F8F8F8F8F8F8F8ull) | (((v & 0x0707070707070707ull) + 0x0303030303030303ull) & 0x0808080808080808ull)) break;
    }
    for (; i < n; ++i) if (a[i] < 0 || a[i] > 4) return (long)i;
    return -1;
}

extern "C" int batched_step_staged(soccer_handle* h, uint32_t use) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "batched_step_staged during graph capture");
    const bool has_a = use & SOCCER_STAGE_ACT_A, has_b = use & SOCCER_STAGE_ACT_B;
    if ((!has_a && !h->P.policy_a) || (!has_b && !h->P.policy_b))
        return fail(h, SOCCER_E_INVALID, "batched_step: an action stream is required for every player without a fixed policy");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t n = h->P.n;
    const StageLayout L = stage_layout(n);
    if (int rc = ensure_stage(h, L)) return rc;
    uint8_t* H = h->stage_host; uint8_t* D = h->stage_dev;
    for (int pl = 0; pl < 2; ++pl) {
        if (!(pl ? has_b : has_a)) continue;
        const long bad = first_bad_action(reinterpret_cast<const int8_t*>(H + (pl ? L.act_b : L.act_a)), n);
        if (bad >= 0) return fail(h, SOCCER_E_INVALID, "batched_step: action of player_%c in lane %ld is %d; actions must be in 0..4",
                                  pl ? 'b' : 'a', bad, (int)reinterpret_cast<const int8_t*>(H + (pl ? L.act_b : L.act_a))[bad]);
    }
    if (!h->mapped) {           // only what this call uses crosses the bus
        if (has_a && has_b) HIP_TRY(h, hipMemcpyAsync(D + L.act_a, H + L.act_a, L.act_b + n - L.act_a, hipMemcpyHostToDevice, h->stream));
        else if (has_a) HIP_TRY(h, hipMemcpyAsync(D + L.act_a, H + L.act_a, n, hipMemcpyHostToDevice, h->stream));
        else if (has_b) HIP_TRY(h, hipMemcpyAsync(D + L.act_b, H + L.act_b, n, hipMemcpyHostToDevice, h->stream));
        if (use & SOCCER_STAGE_U_STEP) HIP_TRY(h, hipMemcpyAsync(D + L.u_step, H + L.u_step, 8 * n, hipMemcpyHostToDevice, h->stream));
        if (use & SOCCER_STAGE_U_RESET) HIP_TRY(h, hipMemcpyAsync(D + L.u_reset, H + L.u_reset, 8 * n, hipMemcpyHostToDevice, h->stream));
    }
    soccer_step_args d{};
    d.act_a = has_a ? reinterpret_cast<const int8_t*>(D + L.act_a) : nullptr;
    d.act_b = has_b ? reinterpret_cast<const int8_t*>(D + L.act_b) : nullptr;
    d.u_step = (use & SOCCER_STAGE_U_STEP) ? reinterpret_cast<const double*>(D + L.u_step) : nullptr;
    d.u_reset = (use & SOCCER_STAGE_U_RESET) ? reinterpret_cast<const double*>(D + L.u_reset) : nullptr;
    d.obs = reinterpret_cast<uint16_t*>(D + L.obs); d.final_obs = reinterpret_cast<uint16_t*>(D + L.final_obs);
    d.reward = reinterpret_cast<int8_t*>(D + L.reward); d.terminated = D + L.term; d.truncated = D + L.trunc;
    d.prob_code = D + L.code;
    if (int rc = batched_step_ex(h, &d)) return rc;
    if (!h->mapped) HIP_TRY(h, hipMemcpyAsync(H + L.out_begin, D + L.out_begin, L.total - L.out_begin, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SOCCER_OK;
}

extern "C" int batched_reset_staged(soccer_handle* h, uint32_t use) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "batched_reset_staged during graph capture");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t n = h->P.n;
    const StageLayout L = stage_layout(n);
    if (int rc = ensure_stage(h, L)) return rc;
    uint8_t* H = h->stage_host; uint8_t* D = h->stage_dev;
    if (!h->mapped) {
        if (use & SOCCER_STAGE_MASK) HIP_TRY(h, hipMemcpyAsync(D + L.mask, H + L.mask, n, hipMemcpyHostToDevice, h->stream));
        if (use & SOCCER_STAGE_U_RESET) HIP_TRY(h, hipMemcpyAsync(D + L.u_reset, H + L.u_reset, 8 * n, hipMemcpyHostToDevice, h->stream));
    }
    if (int rc = batched_reset(h, (use & SOCCER_STAGE_MASK) ? D + L.mask : nullptr,
                               (use & SOCCER_STAGE_U_RESET) ? reinterpret_cast<const double*>(D + L.u_reset) : nullptr,
                               reinterpret_cast<uint16_t*>(D + L.obs))) return rc;
    if (!h->mapped) HIP_TRY(h, hipMemcpyAsync(H + L.obs, D + L.obs, 2 * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SOCCER_OK;
}

// ------------------------------------------------------------------------------------------------
// one environment, one call: inputs by value, result polled from a mapped record (see scalar_kernel)
static int scalar_call(soccer_handle* h, const char* what, uint32_t op, soccer_scalar_io* io) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (!io) return fail(h, SOCCER_E_INVALID, "%s: io is NULL", what);
    if (h->capturing) return fail(h, SOCCER_E_STATE, "%s during graph capture", what);
    if (h->P.n != 1) return fail(h, SOCCER_E_INVALID, "%s needs a handle with n_lanes == 1", what);
    const Rules& R = h->rules;
    ScalarIO k{};
    k.op = op; k.u_step = io->u_step; k.u_reset = io->u_reset;
    if (op == 0u) {
        if (io->needs_reset) return fail(h, SOCCER_E_INVALID, "Please reset the environment before taking a step");   // :376
        const int ra = io->row_a, ca = io->col_a, rb = io->row_b, cb = io->col_b;
        if (ra < 0 || ra >= R.H || rb < 0 || rb >= R.H || ca < 0 || ca >= R.W || cb < 0 || cb >= R.W || io->poss > 1)
            return fail(h, SOCCER_E_INVALID, "%s: tuple (%d, %d, %d, %d, %d) is outside the pitch", what, ra, ca, rb, cb, (int)io->poss);
        if (R.kind[((((size_t)ra * R.W + ca) * R.H + rb) * R.W + cb) * 2 + io->poss] == 0)
            return fail(h, SOCCER_E_INVALID, "%s: tuple (%d, %d, %d, %d, %d) is unreachable", what, ra, ca, rb, cb, (int)io->poss);
        if ((int)io->t > h->cfg.max_steps) return fail(h, SOCCER_E_INVALID, "%s: t = %d exceeds max_steps", what, (int)io->t);
        if ((!h->P.policy_a && (io->act_a < 0 || io->act_a > 4)) || (!h->P.policy_b && (io->act_b < 0 || io->act_b > 4)))
            return fail(h, SOCCER_E_INVALID, "%s: actions must be in 0..4", what);
        k.pos = (uint32_t)ra | ((uint32_t)ca << 8) | ((uint32_t)rb << 16) | ((uint32_t)cb << 24);
        k.misc = (uint32_t)io->poss | ((uint32_t)io->t << 8) | ((uint32_t)(uint8_t)io->act_a << 16) | ((uint32_t)(uint8_t)io->act_b << 24);
    }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!h->rec_host) {
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->rec_host), 64, hipHostMallocMapped));
        std::memset(h->rec_host, 0, 64);
        HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->rec_dev), h->rec_host, 0));
    }
    k.seq = ++h->rec_seq ? h->rec_seq : ++h->rec_seq;                    // never 0
    k.record = h->rec_dev;
    KernelParams P = h->P;
    bind_tick(h, P, 1);
    if (h->slip) hipLaunchKernelGGL(scalar_kernel<true>, dim3(1), dim3(64), 0, h->stream, P, k);
    else hipLaunchKernelGGL(scalar_kernel<false>, dim3(1), dim3(64), 0, h->stream, P, k);
    HIP_TRY(h, hipGetLastError());
    volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(h->rec_host);
    const auto t_start = std::chrono::steady_clock::now();
    // complete record: word 0 == seq and the top byte of word 3 == seq's low byte (both ends of the one store)
    // ... and the check byte in word 3 is the byte-sum of words 1 and 2, so a record of which only some dwords have
    // landed is never accepted (soccer_hip.h, soccer_step_scalar)
    auto byte_sum = [](uint32_t a, uint32_t b) { uint32_t s_ = 0; for (int q = 0; q < 4; ++q) s_ += ((a >> (8 * q)) & 0xffu) + ((b >> (8 * q)) & 0xffu); return s_ & 0xffu; };
    auto landed = [&]() {
        const uint32_t f0 = flag[0], f1 = flag[1], f2 = flag[2], f3 = flag[3];
        return f0 == k.seq && (f3 >> 24) == (k.seq & 0xffu) && ((f3 >> 16) & 0xffu) == byte_sum(f1, f2);
    };
    for (uint32_t spins = 0; !landed(); ++spins) {
        __builtin_ia32_pause();
        if ((spins & 0xfffffu) == 0xfffffu) {                            // every few ms: has the stream died?
            const hipError_t e = hipStreamQuery(h->stream);
            if (e != hipSuccess && e != hipErrorNotReady) return fail(h, SOCCER_E_HIP, "%s: %s", what, hipGetErrorString(e));
            if (e == hipSuccess && !landed()) return fail(h, SOCCER_E_HIP, "%s: the kernel finished without publishing its record", what);
            if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(30))
                return fail(h, SOCCER_E_HIP, "%s: no result after 30 s (is another stream hogging the device?)", what);
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    const volatile uint32_t* r = flag;
    const uint32_t res = r[1], npos = r[2], nm = r[3];
    io->obs = (uint16_t)(res & 0xffffu); io->reward = (int8_t)((res >> 16) & 0xffu);
    io->terminated = (res >> 24) & 1u; io->truncated = (res >> 25) & 1u; io->prob_code = (uint8_t)(res >> 26);
    io->row_a = (int8_t)(npos & 0xffu); io->col_a = (int8_t)((npos >> 8) & 0xffu);
    io->row_b = (int8_t)((npos >> 16) & 0xffu); io->col_b = (int8_t)(npos >> 24);
    io->poss = nm & 1u; io->needs_reset = (nm >> 1) & 1u; io->t = (uint8_t)((nm >> 8) & 0xffu);
    return SOCCER_OK;
}

extern "C" int soccer_step_scalar(soccer_handle* h, soccer_scalar_io* io) { return scalar_call(h, "soccer_step_scalar", 0u, io); }
extern "C" int soccer_reset_scalar(soccer_handle* h, soccer_scalar_io* io) { return scalar_call(h, "soccer_reset_scalar", 1u, io); }

// host arrays in, host arrays out: copies through the staging block around the *_staged calls
extern "C" int batched_step_host(soccer_handle* h, const soccer_step_args* a) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (!a) return fail(h, SOCCER_E_INVALID, "batched_step: arguments are NULL");
    if (a->last_return || a->reward_a_f32 || a->reward_b_f32 || a->finished)
        return fail(h, SOCCER_E_INVALID, "batched_step_host: last_return / reward_*_f32 / finished are device-only");
    soccer_staging_view v{};
    if (int rc = soccer_staging(h, &v)) return rc;
    const size_t n = h->P.n;
    uint32_t use = 0;
    if (a->act_a) { std::memcpy(v.act_a, a->act_a, n); use |= SOCCER_STAGE_ACT_A; }
    if (a->act_b) { std::memcpy(v.act_b, a->act_b, n); use |= SOCCER_STAGE_ACT_B; }
    if (a->u_step) { std::memcpy(v.u_step, a->u_step, 8 * n); use |= SOCCER_STAGE_U_STEP; }
    if (a->u_reset) { std::memcpy(v.u_reset, a->u_reset, 8 * n); use |= SOCCER_STAGE_U_RESET; }
    if (int rc = batched_step_staged(h, use)) return rc;
    if (a->obs) std::memcpy(a->obs, v.obs, 2 * n);
    if (a->final_obs) std::memcpy(a->final_obs, v.final_obs, 2 * n);
    if (a->reward) std::memcpy(a->reward, v.reward, n);
    if (a->terminated) std::memcpy(a->terminated, v.terminated, n);
    if (a->truncated) std::memcpy(a->truncated, v.truncated, n);
    if (a->prob_code) std::memcpy(a->prob_code, v.prob_code, n);
    return SOCCER_OK;
}

extern "C" int batched_reset_host(soccer_handle* h, const uint8_t* mask, const double* u_reset, uint16_t* obs) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    soccer_staging_view v{};
    if (int rc = soccer_staging(h, &v)) return rc;
    const size_t n = h->P.n;
    uint32_t use = 0;
    if (mask) { std::memcpy(v.mask, mask, n); use |= SOCCER_STAGE_MASK; }
    if (u_reset) { std::memcpy(v.u_reset, u_reset, 8 * n); use |= SOCCER_STAGE_U_RESET; }
    if (int rc = batched_reset_staged(h, use)) return rc;
    if (obs) std::memcpy(obs, v.obs, 2 * n);
    return SOCCER_OK;
}

// the reference's P_readable, computed on the device by the rule functions of the step kernels
extern "C" int soccer_enumerate_transitions(soccer_handle* h, int32_t* count, double* prob, int32_t* next_flat,
                                            int8_t* reward, uint8_t* done) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_enumerate_transitions during graph capture");
    if (!count || !prob || !next_flat || !reward || !done) return fail(h, SOCCER_E_INVALID, "all five outputs are required");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t keys = h->rules.lut.size() * 25, ent = keys * kMaxOutcomes;
    EnumIO io{};
    io.n_tuples = static_cast<int32_t>(h->rules.lut.size()); io.H = h->rules.H;
    void* bufs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t sizes[5] = {keys * sizeof(int32_t), ent * sizeof(double), ent * sizeof(int32_t), ent, ent};
    int rc = SOCCER_OK;
    for (int i = 0; i < 5 && rc == SOCCER_OK; ++i)
        if (hipMalloc(&bufs[i], sizes[i]) != hipSuccess) rc = fail(h, SOCCER_E_NOMEM, "out of device memory for the transition table");
    if (rc == SOCCER_OK) {
        io.count = static_cast<int32_t*>(bufs[0]); io.prob = static_cast<double*>(bufs[1]);
        io.next = static_cast<int32_t*>(bufs[2]); io.reward = static_cast<int8_t*>(bufs[3]); io.done = static_cast<uint8_t*>(bufs[4]);
        const unsigned grid = static_cast<unsigned>((keys + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(enumerate_kernel, dim3(grid), dim3(kBlock), 0, h->stream, h->P, io);
        void* dst[5] = {count, prob, next_flat, reward, done};
        hipError_t e = hipGetLastError();
        for (int i = 0; i < 5 && e == hipSuccess; ++i) e = hipMemcpyAsync(dst[i], bufs[i], sizes[i], hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(h, SOCCER_E_HIP, "transition table export failed: %s", hipGetErrorString(e));
    }
    for (void* b : bufs) if (b) (void)hipFree(b);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// planners (reference gym_soccer/utils/planners.py).  The (state, learner action) lists are assembled on
// the host from the device-enumerated transition relation exactly as the reference's constructor builds
// P[s][a], Pmat and Rmat (:167-293), cached on the handle until the policy changes, and one
// single-workgroup kernel runs the whole planner.
static void drop_plan(soccer_handle* h) {
    for (void* b : h->plan_bufs) if (b) (void)hipFree(b);
    h->plan_bufs.clear(); h->plan_ready = false;
}

template <class T>
static int plan_upload(soccer_handle* h, const std::vector<T>& v, const T** out) {
    void* d = nullptr;
    const size_t bytes = v.size() * sizeof(T);
    if (hipMalloc(&d, bytes ? bytes : 1) != hipSuccess) return fail(h, SOCCER_E_NOMEM, "out of device memory for the planner lists");
    h->plan_bufs.push_back(d);
    if (bytes && hipMemcpy(d, v.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(h, SOCCER_E_HIP, "planner list upload failed");
    *out = static_cast<const T*>(d);
    return SOCCER_OK;
}

static int build_plan(soccer_handle* h) {
    if (h->plan_ready) return SOCCER_OK;
    const bool fixed_a = h->P.policy_a != nullptr, fixed_b = h->P.policy_b != nullptr;
    if (fixed_a == fixed_b)
        return fail(h, SOCCER_E_INVALID, "planners need single-agent mode: exactly one side with a fixed policy (soccer_set_policy)");
    const Rules& R = h->rules;
    const int nS = R.nS;
    if ((size_t)nS * sizeof(double) > 150 * 1024) return fail(h, SOCCER_E_INVALID, "too many states (%d) for the single-workgroup planner", nS);
    const size_t T = R.lut.size(), keys = T * 25, ent = keys * kMaxOutcomes;
    std::vector<int32_t> count(keys), nxt(ent); std::vector<double> prob(ent); std::vector<int8_t> rew(ent); std::vector<uint8_t> done(ent);
    if (int rc = soccer_enumerate_transitions(h, count.data(), prob.data(), nxt.data(), rew.data(), done.data())) return rc;
    std::vector<int8_t> policy(nS);
    HIP_TRY(h, hipMemcpy(policy.data(), fixed_a ? h->P.policy_a : h->P.policy_b, (size_t)nS, hipMemcpyDeviceToHost));
    auto obs_of = [&](size_t f) { return R.kind[f] == 2 ? 0 : (int)R.lut[f]; };
    const bool flip = fixed_a;                                          // learner B sees -r (:243-244)
    // P[s][a]: goal tuples all write index 0 and overwrite each other (identical lists), live tuples own theirs
    std::vector<long> tuple_of(nS, -1);
    for (size_t f = 0; f < T; ++f) if (R.kind[f] != 0) tuple_of[obs_of(f)] = (long)f;
    const PlanEntry pad_entry{0.0, (int32_t)0x80000000, 0.0f};
    auto pad = [&](std::vector<PlanEntry>& v) { while (v.size() % kPlanPad) v.push_back(pad_entry); };
    std::vector<int32_t> off((size_t)nS * 5 + 1, 0); std::vector<PlanEntry> lists;
    for (int s = 0; s < nS; ++s) for (int a = 0; a < 5; ++a) {
        const long f = tuple_of[s];
        if (f < 0) return fail(h, SOCCER_E_INVALID, "internal error: observation index %d has no tuple", s);
        const size_t key = (size_t)f * 25 + (fixed_a ? policy[s] : a) * 5 + (fixed_b ? policy[s] : a);
        for (int k = 0; k < count[key]; ++k) {
            const size_t e = key * kMaxOutcomes + k;
            const double rr = flip ? -1.0 * (double)rew[e] : (double)rew[e];
            lists.push_back(PlanEntry{prob[e], obs_of((size_t)nxt[e]) | (done[e] ? (int32_t)0x80000000 : 0), (float)rr});
        }
        pad(lists);
        off[(size_t)s * 5 + a + 1] = (int32_t)lists.size();
    }
    // Pmat[s][ns][a] += p and Rmat[s][a] (= 0, then += p * r) in the constructor's tuple order (:280-291):
    // index 0 accumulates one unit of probability per goal tuple, its Rmat is the last goal tuple's (0)
    std::vector<std::vector<double>> row((size_t)nS * 5);               // dense rows only for touched (s, a)
    std::vector<double> Rm((size_t)nS * 5, 0.0);
    for (size_t f = 0; f < T; ++f) {
        if (R.kind[f] == 0) continue;
        const int s = obs_of(f);
        for (int a = 0; a < 5; ++a) {
            const size_t key = f * 25 + (fixed_a ? policy[s] : a) * 5 + (fixed_b ? policy[s] : a);
            std::vector<double>& r = row[(size_t)s * 5 + a];
            if (r.empty()) r.assign(nS, 0.0);
            double acc = 0.0;
            for (int k = 0; k < count[key]; ++k) {
                const size_t e = key * kMaxOutcomes + k;
                const double rr = flip ? -1.0 * (double)rew[e] : (double)rew[e];
                r[obs_of((size_t)nxt[e])] += prob[e];
                acc = acc + prob[e] * rr;
            }
            Rm[(size_t)s * 5 + a] = acc;
        }
    }
    std::vector<int32_t> m_off((size_t)nS * 5 + 1, 0); std::vector<PlanEntry> m_lists;
    const PlanEntry m_pad{0.0, 0, 0.0f};
    for (size_t q = 0; q < (size_t)nS * 5; ++q) {
        for (int ns = 0; ns < nS; ++ns) if (row[q][ns] != 0.0) m_lists.push_back(PlanEntry{row[q][ns], ns, 0.0f});
        while (m_lists.size() % kPlanPad) m_lists.push_back(m_pad);
        m_off[q + 1] = (int32_t)m_lists.size();
        std::vector<double>().swap(row[q]);
    }
    PlanIO& io = h->plan;
    io = PlanIO{};
    int rc = plan_upload(h, off, &io.offset);
    if (!rc) rc = plan_upload(h, lists, &io.list);
    if (!rc) rc = plan_upload(h, m_off, &io.m_offset);
    if (!rc) rc = plan_upload(h, m_lists, &io.m_list);
    if (!rc) rc = plan_upload(h, Rm, &io.m_R);
    const std::vector<double> zV(nS, 0.0), zQ((size_t)nS * 5, 0.0); const std::vector<int32_t> zpi(nS, 0), zc(16, 0);
    const double* cV = nullptr; const double* cN = nullptr; const double* cQ = nullptr; const int32_t* cpi = nullptr; const int32_t* cc = nullptr;
    if (!rc) rc = plan_upload(h, zV, &cV);
    if (!rc) rc = plan_upload(h, zV, &cN);
    if (!rc) rc = plan_upload(h, zQ, &cQ);
    if (!rc) rc = plan_upload(h, zpi, &cpi);
    if (!rc) rc = plan_upload(h, zc, &cc);
    if (rc) { drop_plan(h); return rc; }
    io.V = const_cast<double*>(cV); io.newV = const_cast<double*>(cN); io.Q = const_cast<double*>(cQ);
    io.pi = const_cast<int32_t*>(cpi); io.counters = const_cast<int32_t*>(cc);
    io.nS = nS;
    const size_t smem = (size_t)nS * sizeof(double);
    if (smem > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&planner_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    h->plan_ready = true;
    return SOCCER_OK;
}

// runs one planner; inputs pi_in / V_in and every output are HOST pointers (any output may be NULL)
static int run_plan(soccer_handle* h, const char* what, int mode, double theta, double gamma, int32_t max_sweeps, int32_t k,
                    const int32_t* pi_in, const double* V_in, const double* Q_in, double* V, double* Q, int32_t* pi, int32_t* iterations) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "%s during graph capture", what);
    if (max_sweeps < 1) return fail(h, SOCCER_E_INVALID, "max_sweeps must be >= 1");
    if (!(gamma >= 0.0 && gamma <= 1.0)) return fail(h, SOCCER_E_INVALID, "discount_factor must be in [0, 1]");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (int rc = build_plan(h)) return rc;
    PlanIO io = h->plan;
    const int nS = io.nS;
    if (pi_in) {
        for (int s = 0; s < nS; ++s) if (pi_in[s] < 0 || pi_in[s] > 4) return fail(h, SOCCER_E_INVALID, "pi[%d] = %d is not an action", s, pi_in[s]);
        HIP_TRY(h, hipMemcpyAsync(io.pi, pi_in, (size_t)nS * 4, hipMemcpyHostToDevice, h->stream));
    }
    if (V_in) HIP_TRY(h, hipMemcpyAsync(io.V, V_in, (size_t)nS * 8, hipMemcpyHostToDevice, h->stream));
    else if (mode == kPlanEvalDense) HIP_TRY(h, hipMemsetAsync(io.V, 0, (size_t)nS * 8, h->stream));
    if (Q_in) HIP_TRY(h, hipMemcpyAsync(io.Q, Q_in, (size_t)nS * 40, hipMemcpyHostToDevice, h->stream));
    io.mode = mode; io.theta = theta; io.gamma = gamma; io.max_sweeps = max_sweeps; io.k = k;
    io.threshold = (theta * (1 - gamma)) / (2 * gamma);                  // planners.py:75
    hipLaunchKernelGGL(planner_kernel, dim3(1), dim3(1024), (size_t)nS * sizeof(double), h->stream, io);
    HIP_TRY(h, hipGetLastError());
    int32_t counters[4] = {0, 0, 0, 0};
    if (V) HIP_TRY(h, hipMemcpyAsync(V, io.V, (size_t)nS * 8, hipMemcpyDeviceToHost, h->stream));
    if (Q) HIP_TRY(h, hipMemcpyAsync(Q, io.Q, (size_t)nS * 40, hipMemcpyDeviceToHost, h->stream));
    if (pi) HIP_TRY(h, hipMemcpyAsync(pi, io.pi, (size_t)nS * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(counters, io.counters, 12, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (iterations) *iterations = counters[0];
    if (counters[2]) return fail(h, SOCCER_E_STATE, "%s stopped after max_sweeps = %d sweeps without converging", what, max_sweeps);
    return SOCCER_OK;
}

extern "C" int soccer_value_iteration(soccer_handle* h, double theta, double discount_factor, int32_t max_sweeps,
                                      double* V, double* Q, int32_t* pi, int32_t* iterations) {
    return run_plan(h, "soccer_value_iteration", kPlanVI, theta, discount_factor, max_sweeps, 0, nullptr, nullptr, nullptr, V, Q, pi, iterations);
}

extern "C" int soccer_policy_evaluation(soccer_handle* h, const int32_t* pi, double theta, double discount_factor,
                                        int32_t max_sweeps, double* V, int32_t* sweeps) {
    if (h && !pi) return fail(h, SOCCER_E_INVALID, "pi is NULL");
    return run_plan(h, "soccer_policy_evaluation", kPlanEval, theta, discount_factor, max_sweeps, 0, pi, nullptr, nullptr, V, nullptr, nullptr, sweeps);
}

extern "C" int soccer_policy_improvement(soccer_handle* h, const double* V, double discount_factor, double* Q, int32_t* new_pi) {
    if (h && !V) return fail(h, SOCCER_E_INVALID, "V is NULL");
    return run_plan(h, "soccer_policy_improvement", kPlanImprove, 0.0, discount_factor, 1, 0, nullptr, V, nullptr, nullptr, Q, new_pi, nullptr);
}

extern "C" int soccer_policy_iteration(soccer_handle* h, const int32_t* pi0, double theta, double discount_factor,
                                       int32_t max_sweeps, double* V, double* Q, int32_t* pi, int32_t* iterations) {
    if (h && !pi0) return fail(h, SOCCER_E_INVALID, "pi0 (the initial policy) is NULL");
    return run_plan(h, "soccer_policy_iteration", kPlanPI, theta, discount_factor, max_sweeps, 0, pi0, nullptr, nullptr, V, Q, pi, iterations);
}

extern "C" int soccer_modified_policy_iteration(soccer_handle* h, int32_t k, double theta, double discount_factor,
                                                int32_t max_sweeps, double* V, double* Q, int32_t* pi, int32_t* iterations) {
    if (h && k < 1) return fail(h, SOCCER_E_INVALID, "k must be >= 1");
    if (h && !(discount_factor > 0.0)) return fail(h, SOCCER_E_INVALID, "discount_factor must be > 0 for the stopping threshold");
    return run_plan(h, "soccer_modified_policy_iteration", kPlanMPI, theta, discount_factor, max_sweeps, k, nullptr, nullptr, nullptr, V, Q, pi, iterations);
}

extern "C" int soccer_policy_eval_dense(soccer_handle* h, const double* policy, int32_t k, double theta, double discount_factor,
                                        int32_t max_sweeps, const double* init, double* v, int32_t* sweeps) {
    if (h && !policy) return fail(h, SOCCER_E_INVALID, "policy is NULL");
    if (h && k < 1) return fail(h, SOCCER_E_INVALID, "k must be >= 1");
    return run_plan(h, "soccer_policy_eval_dense", kPlanEvalDense, theta, discount_factor, max_sweeps, k, nullptr, init, policy, v, nullptr, nullptr, sweeps);
}

// single-agent mode: one side follows a fixed policy looked up by the current observation index
// (reference :54-56, :187-188).  policy_host NULL clears it.
extern "C" int soccer_set_policy(soccer_handle* h, int32_t player, const int8_t* policy_host, int32_t n_states) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_set_policy during graph capture");
    if (player != 0 && player != 1) return fail(h, SOCCER_E_INVALID, "player must be 0 (player_a) or 1 (player_b)");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int8_t** slot = player == 0 ? &h->P.policy_a : &h->P.policy_b;
    drop_plan(h);
    if (!policy_host) { *slot = nullptr; return SOCCER_OK; }
    if (n_states != h->rules.nS) return fail(h, SOCCER_E_INVALID, "policy must have one action per observation index (%d)", h->rules.nS);
    if ((player == 0 ? h->P.policy_b : h->P.policy_a) != nullptr)
        return fail(h, SOCCER_E_INVALID, "Both players cannot have a policy. At least one must be None.");   // :38
    for (int i = 0; i < n_states; ++i)
        if (policy_host[i] < 0 || policy_host[i] > 4) return fail(h, SOCCER_E_INVALID, "policy[%d] = %d is not an action", i, (int)policy_host[i]);
    if (!h->d_policy[player]) HIP_TRY(h, hipMalloc(&h->d_policy[player], (size_t)h->rules.nS));
    HIP_TRY(h, hipMemcpy(h->d_policy[player], policy_host, (size_t)n_states, hipMemcpyHostToDevice));
    *slot = h->d_policy[player];
    return SOCCER_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int soccer_dims(const soccer_handle* h, int32_t* n_states, int32_t* lut_len, int32_t* n_isd,
                           int32_t* internal_width) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (n_states) *n_states = h->rules.nS;
    if (lut_len) *lut_len = static_cast<int32_t>(h->rules.lut.size());
    if (n_isd) *n_isd = h->rules.n_isd;
    if (internal_width) *internal_width = h->rules.W;
    return SOCCER_OK;
}

extern "C" int soccer_get_tables(const soccer_handle* h, uint16_t* lut, int8_t* goal_value, int8_t* isd_states) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    const Rules& R = h->rules;
    if (lut) std::memcpy(lut, R.lut.data(), R.lut.size() * sizeof(uint16_t));
    if (goal_value) std::memcpy(goal_value, R.goal_value.data(), R.goal_value.size());
    if (isd_states) for (int i = 0; i < R.n_isd; ++i) for (int k = 0; k < 5; ++k) isd_states[i * 5 + k] = R.isd[i][k];
    return SOCCER_OK;
}

extern "C" int soccer_prob_table(const soccer_handle* h, double prob[12]) {
    if (!h || !prob) return fail(nullptr, SOCCER_E_INVALID, "handle/prob is NULL");
    static const double nsp[3] = {1.0, 0.5, 0.25};
    for (int c = 0; c < 4; ++c) for (int k = 0; k < 3; ++k) prob[c * 3 + k] = h->P.w[c] * nsp[k];   // :241
    return SOCCER_OK;
}

extern "C" int soccer_get_stats(soccer_handle* h, uint64_t hist[3], uint64_t* misuse) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_get_stats during graph capture");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (hist) {
        std::vector<unsigned long long> shards(h->hist_slots * kHistStride);
        HIP_TRY(h, hipMemcpy(shards.data(), h->d_hist, shards.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        hist[0] = hist[1] = hist[2] = 0;
        for (size_t s = 0; s < h->hist_slots; ++s) for (int b = 0; b < 3; ++b) hist[b] += shards[s * kHistStride + b];
    }
    if (misuse) {                                   // the stream is idle: no copy needed
        const volatile unsigned int* m = h->misuse_host;
        *misuse = (m[0] ? 1u : 0u) | (m[1] ? 2u : 0u);
    }
    return SOCCER_OK;
}

extern "C" uint32_t soccer_peek_misuse(const soccer_handle* h) {
    if (!h || !h->misuse_host) return 0u;
    const volatile unsigned int* m = h->misuse_host;
    return (m[0] ? SOCCER_MISUSE_FROZEN : 0u) | (m[1] ? SOCCER_MISUSE_ACTION : 0u);
}

extern "C" int soccer_reset_stats(soccer_handle* h) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMemsetAsync(h->d_hist, 0, sizeof(unsigned long long) * h->hist_slots * kHistStride, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_misuse, 0, 64, h->stream));
    return SOCCER_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int soccer_malloc(soccer_handle* h, size_t bytes, void** dptr) {
    if (!h || !dptr) return fail(h, SOCCER_E_INVALID, "handle/dptr is NULL");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMalloc(dptr, bytes ? bytes : 1));
    return SOCCER_OK;
}
extern "C" int soccer_free(soccer_handle* h, void* dptr) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (dptr) HIP_TRY(h, hipFree(dptr));
    return SOCCER_OK;
}
extern "C" int soccer_memcpy_h2d(soccer_handle* h, void* dst, const void* src, size_t bytes) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_memcpy_h2d during graph capture");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SOCCER_OK;
}
extern "C" int soccer_memcpy_d2h(soccer_handle* h, void* dst, const void* src, size_t bytes) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_memcpy_d2h during graph capture");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SOCCER_OK;
}
extern "C" int soccer_memset(soccer_handle* h, void* dst, int value, size_t bytes) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMemsetAsync(dst, value, bytes, h->stream));
    return SOCCER_OK;
}

// Inside a graph capture an event record becomes a node that carries no timestamp (hipEventElapsedTime refuses such
// events), so a captured timer_start / timer_mark is a one-thread kernel that stores the device's constant-rate wall clock
// into the handle's host-mapped block instead; soccer_timer_read converts the difference.
__global__ void stamp_kernel(unsigned long long* slot) {
    if (threadIdx.x == 0) *slot = wall_clock64();
}

static volatile unsigned long long* stamp_host(soccer_handle* h) {
    return reinterpret_cast<volatile unsigned long long*>(reinterpret_cast<uint8_t*>(h->misuse_host) + 64);
}
static unsigned long long* stamp_dev(soccer_handle* h) {
    return reinterpret_cast<unsigned long long*>(reinterpret_cast<uint8_t*>(h->d_misuse) + 64);
}

extern "C" int soccer_stamp(soccer_handle* h, int32_t slot) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (slot < 0 || slot >= SOCCER_STAMP_SLOTS) return fail(h, SOCCER_E_INVALID, "stamp slot %d out of range (0..%d)", slot, SOCCER_STAMP_SLOTS - 1);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, h->stream, stamp_dev(h) + kStampStride * slot);
    HIP_TRY(h, hipGetLastError());
    return SOCCER_OK;
}

extern "C" int soccer_stamps_clear(soccer_handle* h, int32_t first, int32_t count) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (first < 0 || count < 0 || first + count > SOCCER_STAMP_SLOTS) return fail(h, SOCCER_E_INVALID, "stamp range out of bounds");
    for (int32_t i = 0; i < count; ++i) stamp_host(h)[kStampStride * (first + i)] = 0ull;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    return SOCCER_OK;
}

extern "C" int soccer_stamps_read(soccer_handle* h, int32_t first, int32_t count, uint64_t* ticks, int32_t* khz) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (first < 0 || count < 0 || first + count > SOCCER_STAMP_SLOTS || (count && !ticks)) return fail(h, SOCCER_E_INVALID, "stamp range out of bounds");
    for (int32_t i = 0; i < count; ++i) ticks[i] = stamp_host(h)[kStampStride * (first + i)];
    if (khz) *khz = h->wall_clock_khz;
    return SOCCER_OK;
}

extern "C" int soccer_timer_start(soccer_handle* h) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    h->timer_stamped = h->capturing;
    if (h->capturing) { h->capture_stamped = true; return soccer_stamp(h, 0); }
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    return SOCCER_OK;
}
extern "C" int soccer_timer_mark(soccer_handle* h) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (h->capturing) {
        if (!h->capture_stamped) return fail(h, SOCCER_E_STATE, "soccer_timer_mark in a capture needs soccer_timer_start in the same capture");
        return soccer_stamp(h, 1);
    }
    h->timer_stamped = false;
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    return SOCCER_OK;
}
extern "C" int soccer_timer_read(soccer_handle* h, float* elapsed_ms) {
    if (!h || !elapsed_ms) return fail(h, SOCCER_E_INVALID, "handle/elapsed_ms is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_timer_read during graph capture");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (h->timer_stamped) {                 // the stamps of the last replay of a graph that captured start / mark
        volatile unsigned long long* st = stamp_host(h);
        if (h->stamp_poll) {
            // The closing stamp's kernel writes it to host-mapped memory and clock values only grow: watching that word
            // change from what it held when the (single) replay was enqueued costs no runtime call at all — the host
            // sees the end of the region ~1 us after the device reaches it — and the host never WRITES the stamp lines.  (The runtime's own completion signal of the replay arrives ~13 us after the last
            // kernel on an MI355X — the write-back of the dirty L2 lines and the signal path, tools/labs/sync_cost.py — whether
            // one waits for it in hipStreamSynchronize, in hipDeviceSynchronize or by polling an event recorded behind
            // the replay; a host that only needs the device time does not have to.)
            const auto t_start = std::chrono::steady_clock::now();
            for (uint32_t spins = 0; st[kStampStride] == h->stamp_prev; ++spins) {
                __builtin_ia32_pause();
                if ((spins & 0xfffffu) == 0xfffffu) {                    // every few ms: has the stream died?
                    const hipError_t e = hipStreamQuery(h->stream);
                    if (e != hipSuccess && e != hipErrorNotReady) return fail(h, SOCCER_E_HIP, "soccer_timer_read: %s", hipGetErrorString(e));
                    if (e == hipSuccess && st[kStampStride] == h->stamp_prev) return fail(h, SOCCER_E_STATE, "soccer_timer_read: the stream is idle but the closing stamp was never written");
                    if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(60))
                        return fail(h, SOCCER_E_HIP, "soccer_timer_read: no closing stamp after 60 s");
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        } else {
            for (uint32_t spins = 0;; ++spins) {
                const hipError_t q = hipStreamQuery(h->stream);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) return fail(h, SOCCER_E_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
                if (spins > 2000u) { (void)hipGetLastError(); HIP_TRY(h, hipStreamSynchronize(h->stream)); break; }
                __builtin_ia32_pause();
            }
            (void)hipGetLastError();
        }
        *elapsed_ms = (float)((double)(st[kStampStride] - st[0]) / (double)h->wall_clock_khz);
        return SOCCER_OK;
    }
    // poll instead of blocking: a blocked waiter is woken tens of microseconds after the event completes, which is
    // as long as the whole timed region of a short run
    for (uint32_t spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(h->ev1);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return fail(h, SOCCER_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        if (spins > 2000u) { (void)hipGetLastError(); HIP_TRY(h, hipEventSynchronize(h->ev1)); break; }   // long waits: block
        __builtin_ia32_pause();
    }
    (void)hipGetLastError();
    HIP_TRY(h, hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    return SOCCER_OK;
}
extern "C" int soccer_timer_stop(soccer_handle* h, float* elapsed_ms) {
    if (int rc = soccer_timer_mark(h)) return rc;
    return soccer_timer_read(h, elapsed_ms);
}

// ------------------------------------------------------------------------------------------------
__global__ void move_tick_kernel(unsigned long long* dst, const unsigned long long* src) {
    if (threadIdx.x == 0) *dst = *src;
}

extern "C" int soccer_graph_begin(soccer_handle* h) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "graph capture already in progress");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    h->capturing = true; h->capture_ticks = 0; h->capture_calls = 0; h->capture_stamped = false;
    h->capture_start_slot = h->tick_slot;

    return SOCCER_OK;
}

extern "C" int soccer_graph_end(soccer_handle* h, soccer_graph** out) {
    if (!h || !out) return fail(h, SOCCER_E_INVALID, "handle/out is NULL");
    if (!h->capturing) return fail(h, SOCCER_E_STATE, "no graph capture in progress");
    if (h->capture_calls % 2 != 0) {
        // An odd number of launches leaves the tick in the other slot, and the nodes' slot pointers are baked in: one more node
        // (a one-thread kernel, ~1.5 us per replay; an even count needs none) moves it back to where a replay starts reading.
        hipLaunchKernelGGL(move_tick_kernel, dim3(1), dim3(64), 0, h->stream,
                           h->d_tick + (h->capture_start_slot ? 16 : 0), h->d_tick + (h->tick_slot ? 16 : 0));
        h->tick_slot = h->capture_start_slot;
    }
    h->capturing = false;
    hipGraph_t graph = nullptr;
    HIP_TRY(h, hipStreamEndCapture(h->stream, &graph));
    soccer_graph* g = new soccer_graph();
    g->graph = graph; g->ticks = h->capture_ticks; g->start_slot = h->capture_start_slot; g->stamped = h->capture_stamped;
    hipError_t e = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(graph); delete g;
        return fail(h, SOCCER_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    // move the executable graph to the device now, so that the first soccer_graph_launch does not pay for it
    // (best effort: a runtime without hipGraphUpload support just uploads on first launch)
    (void)hipGraphUpload(g->exec, h->stream);
    (void)hipGetLastError();
    *out = g;
    return SOCCER_OK;
}

extern "C" int soccer_graph_launch(soccer_handle* h, soccer_graph* g, int32_t replays) {
    if (!h || !g) return fail(h, SOCCER_E_INVALID, "handle/graph is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_graph_launch during graph capture");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (replays > 0 && h->tick_slot != g->start_slot) {
        // launches issued since the capture left the tick in the other slot: move it across
        HIP_TRY(h, hipMemcpyAsync(h->d_tick + (g->start_slot ? 16 : 0), h->d_tick + (h->tick_slot ? 16 : 0),
                                  sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
        h->tick_slot = g->start_slot;
    }
    // one replay: remember what the closing stamp holds, so that soccer_timer_read can watch it change (several replays
    // would each write it)
    // (only with the stream idle: a replay still in flight would write the slot AFTER it was sampled, and soccer_timer_read would
    // return on that earlier replay's stamp; otherwise — and for graphs without timer nodes — soccer_timer_read waits for the stream)
    h->stamp_poll = replays == 1 && g->stamped && hipStreamQuery(h->stream) == hipSuccess;
    (void)hipGetLastError();
    h->timer_stamped = g->stamped;          // soccer_timer_read after this launch: the graph's stamps, or the eager events
    if (h->stamp_poll) h->stamp_prev = stamp_host(h)[kStampStride];
    for (int32_t r = 0; r < replays; ++r) HIP_TRY(h, hipGraphLaunch(g->exec, h->stream));
    h->tick += g->ticks * (uint64_t)(replays > 0 ? replays : 0);
    return SOCCER_OK;
}

extern "C" int soccer_graph_destroy(soccer_handle* h, soccer_graph* g) {
    if (!g) return SOCCER_OK;
    if (h) { (void)hipSetDevice(h->cfg.device); (void)hipStreamSynchronize(h->stream); }
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return SOCCER_OK;
}

// ------------------------------------------------------------------------------------------------
// episode returns from [T][n] result trajectories (trajectory_returns_kernel)
extern "C" int soccer_trajectory_returns(soccer_handle* h, int32_t n_steps, const int8_t* reward, const uint8_t* terminated,
                                         const uint8_t* truncated, int64_t stride, int8_t* last_return,
                                         int32_t* episode_count, uint64_t hist[3]) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_trajectory_returns during graph capture");
    if (n_steps < 1 || !reward || !terminated || !truncated)
        return fail(h, SOCCER_E_INVALID, "soccer_trajectory_returns: n_steps >= 1 and the reward / terminated / truncated trajectories are required");
    if (stride < (int64_t)h->P.n) return fail(h, SOCCER_E_INVALID, "soccer_trajectory_returns: stride must be >= n_lanes");
    if (!aligned(episode_count, 4)) return fail(h, SOCCER_E_INVALID, "soccer_trajectory_returns: episode_count must be 4-byte aligned");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t slots = (size_t)h->grid_cap * 2;       // one u64[4] per workgroup of the (at most two) launches
    if (!h->d_traj_hist) HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_traj_hist), slots * 4 * sizeof(unsigned long long)));
    const unsigned long long n = h->P.n;
    const bool vec = stride % 4 == 0 && aligned(reward, 4) && aligned(terminated, 4) && aligned(truncated, 4) &&
                     aligned(last_return, 4) && aligned(episode_count, 16);
    const unsigned long long n4 = vec ? (n & ~3ull) : 0ull;
    TrajIO io{reward, terminated, truncated, (long long)stride, n_steps, n4, last_return, episode_count, h->d_traj_hist, 0u};
    uint32_t used = 0;
    if (n4) { used = (uint32_t)grid_for(h, n4 >> 2); hipLaunchKernelGGL(trajectory_returns_kernel<true>, dim3(used), dim3(kBlock), 0, h->stream, io); }
    if (n4 < n) {               // ragged tail, or everything when a stream is not dword-aligned: a lane per thread
        TrajIO t = io;
        t.reward += n4; t.terminated += n4; t.truncated += n4; t.n = n - n4; t.slot0 = used;
        t.last_return = off(last_return, n4); t.episode_count = off(episode_count, n4);
        const uint32_t g2 = (uint32_t)grid_for(h, t.n);
        hipLaunchKernelGGL(trajectory_returns_kernel<false>, dim3(g2), dim3(kBlock), 0, h->stream, t);
        used += g2;
    }
    HIP_TRY(h, hipGetLastError());
    if (hist) {
        std::vector<unsigned long long> part((size_t)used * 4);
        HIP_TRY(h, hipMemcpyAsync(part.data(), h->d_traj_hist, part.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        hist[0] = hist[1] = hist[2] = 0;
        for (uint32_t b = 0; b < used; ++b) for (int k = 0; k < 3; ++k) hist[k] += part[(size_t)b * 4 + k];
    }
    return SOCCER_OK;
}

// ------------------------------------------------------------------------------------------------
// RCCL over xGMI: the job's only exchange (BASELINE configs[3]: gather of per-lane episode returns).
// librccl is resolved at run time — a process that never calls soccer_comm_* never loads it — first among the symbols
// already in the process (a host that brought its own copy), then as librccl.so.1 next to the HIP runtime.
struct IdByValue { char internal[SOCCER_COMM_ID_BYTES]; };      // ncclUniqueId: passed BY VALUE to ncclCommInitRank (rccl.h:43, :220)
namespace {
struct Rccl {
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, IdByValue, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false; std::string why;
};
}  // namespace

static Rccl& rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        void* lib = nullptr;
        if (!dlsym(RTLD_DEFAULT, "ncclGetUniqueId")) {
            for (const char* name : {"librccl.so.1", "librccl.so"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
            if (!lib) { const char* e = dlerror(); R.why = std::string("librccl not found: ") + (e ? e : "?"); return; }
        }
        auto sym = [&](const char* n) -> void* { void* p = lib ? dlsym(lib, n) : dlsym(RTLD_DEFAULT, n); if (!p) R.why = std::string("librccl lacks ") + n; return p; };
        R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
        R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
        R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
        R.AllGather = reinterpret_cast<decltype(R.AllGather)>(sym("ncclAllGather"));
        R.AllReduce = reinterpret_cast<decltype(R.AllReduce)>(sym("ncclAllReduce"));
        R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
        R.ok = R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.AllGather && R.AllReduce && R.GetErrorString;
    });
    return R;
}
#define RCCL_TRY(h, expr)                                                                        \
    do {                                                                                         \
        const int r_ = (expr);                                                                   \
        if (r_ != 0) return fail((h), SOCCER_E_HIP, "%s failed: %s", #expr, rccl().GetErrorString(r_)); \
    } while (0)

static void comm_release(soccer_handle* h) {
    if (h && h->comm) { if (rccl().ok) (void)rccl().CommDestroy(h->comm); h->comm = nullptr; h->comm_world = 0; }
}

extern "C" int soccer_comm_unique_id(uint8_t id[SOCCER_COMM_ID_BYTES]) {
    if (!id) return fail(nullptr, SOCCER_E_INVALID, "id is NULL");
    if (!rccl().ok) return fail(nullptr, SOCCER_E_HIP, "%s", rccl().why.c_str());
    static_assert(SOCCER_COMM_ID_BYTES == 128, "ncclUniqueId is 128 bytes (rccl.h: NCCL_UNIQUE_ID_BYTES)");
    RCCL_TRY(nullptr, rccl().GetUniqueId(id));
    return SOCCER_OK;
}

extern "C" int soccer_comm_init(soccer_handle* h, int32_t world, int32_t rank, const uint8_t id[SOCCER_COMM_ID_BYTES]) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_comm_init during graph capture");
    if (world < 1 || rank < 0 || rank >= world || !id) return fail(h, SOCCER_E_INVALID, "soccer_comm_init: need 0 <= rank < world and the unique id of rank 0");
    if (h->comm) return fail(h, SOCCER_E_STATE, "soccer_comm_init: this handle already has a communicator");
    if (!rccl().ok) return fail(h, SOCCER_E_HIP, "%s", rccl().why.c_str());
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!h->d_comm_scratch) HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_comm_scratch), 64));
    IdByValue v; std::memcpy(v.internal, id, sizeof v.internal);
    RCCL_TRY(h, rccl().CommInitRank(&h->comm, world, v, rank));
    h->comm_world = world; h->comm_rank = rank;
    return SOCCER_OK;
}

extern "C" int soccer_comm_destroy(soccer_handle* h) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    comm_release(h);
    return SOCCER_OK;
}

extern "C" int soccer_comm_all_gather(soccer_handle* h, const void* send, void* recv, uint64_t bytes_per_rank) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (!h->comm) return fail(h, SOCCER_E_STATE, "soccer_comm_all_gather: no communicator (soccer_comm_init)");
    if (h->capturing) return fail(h, SOCCER_E_STATE, "soccer_comm_all_gather during graph capture");
    if (!send || !recv) return fail(h, SOCCER_E_INVALID, "soccer_comm_all_gather: send/recv is NULL");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    RCCL_TRY(h, rccl().AllGather(send, recv, (size_t)bytes_per_rank, /*ncclInt8*/ 0, h->comm, h->stream));
    return SOCCER_OK;
}

// small host-value reductions through the handle's 64-byte device scratch: up to 8 values, SUM of uint64 or MAX of float64.
// Synchronises (the result is returned to the host) — which also makes it the job's barrier.
static int comm_reduce_small(soccer_handle* h, void* values, int32_t count, bool f64_max, const char* what) {
    if (!h) return fail(nullptr, SOCCER_E_INVALID, "handle is NULL");
    if (!h->comm) return fail(h, SOCCER_E_STATE, "%s: no communicator (soccer_comm_init)", what);
    if (h->capturing) return fail(h, SOCCER_E_STATE, "%s during graph capture", what);
    if (!values || count < 1 || count > 8) return fail(h, SOCCER_E_INVALID, "%s: 1..8 values", what);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMemcpyAsync(h->d_comm_scratch, values, 8 * (size_t)count, hipMemcpyHostToDevice, h->stream));
    RCCL_TRY(h, rccl().AllReduce(h->d_comm_scratch, h->d_comm_scratch, (size_t)count, f64_max ? /*ncclFloat64*/ 8 : /*ncclUint64*/ 5,
                                 f64_max ? /*ncclMax*/ 2 : /*ncclSum*/ 0, h->comm, h->stream));
    HIP_TRY(h, hipMemcpyAsync(values, h->d_comm_scratch, 8 * (size_t)count, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SOCCER_OK;
}
extern "C" int soccer_comm_sum_u64(soccer_handle* h, uint64_t* values, int32_t count) { return comm_reduce_small(h, values, count, false, "soccer_comm_sum_u64"); }
extern "C" int soccer_comm_max_f64(soccer_handle* h, double* values, int32_t count) { return comm_reduce_small(h, values, count, true, "soccer_comm_max_f64"); }
extern "C" int soccer_comm_barrier(soccer_handle* h) { uint64_t one = 1; return comm_reduce_small(h, &one, 1, false, "soccer_comm_barrier"); }
