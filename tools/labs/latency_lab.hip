// Where do the ~45 us of a 1-lane facade step go?  Completion mechanisms for a tiny kernel on host-mapped
// memory, measured from C++ (no Python): stream sync, event-query spin, stream write-value + poll, and a
// flag the kernel itself writes to mapped memory + poll.  Plus the C-ABI staged step on a mapped handle.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/labs/latency_lab.hip -o build/latency_lab \
//         -Lgym_soccer_littman94_amd -lsoccer_hip -Wl,-rpath,$PWD/gym_soccer_littman94_amd
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "soccer_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void tiny(const volatile int* in, volatile int* out, volatile unsigned* flag, unsigned seq) {
    if (threadIdx.x == 0) {
        out[0] = in[0] + 1;
        if (flag) { __threadfence_system(); *flag = seq; }
    }
}

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    CK(hipSetDevice(0));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* hm; CK(hipHostMalloc(&hm, 4096, hipHostMallocMapped));
    std::memset(hm, 0, 4096);
    int* dm; CK(hipHostGetDevicePointer((void**)&dm, hm, 0));
    volatile unsigned* hflag = (volatile unsigned*)(hm + 64);
    unsigned* dflag = (unsigned*)(dm + 64);
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int N = 20000;
    for (int i = 0; i < 200; ++i) { hipLaunchKernelGGL(tiny, 1, 64, 0, s, dm, dm + 1, nullptr, 0u); CK(hipStreamSynchronize(s)); }
    double t0 = now();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny, 1, 64, 0, s, dm, dm + 1, nullptr, 0u); CK(hipStreamSynchronize(s)); }
    printf("launch + hipStreamSynchronize          %6.2f us\n", (now() - t0) / N);
    t0 = now();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny, 1, 64, 0, s, dm, dm + 1, nullptr, 0u); CK(hipEventRecord(ev, s)); while (hipEventQuery(ev) == hipErrorNotReady) {} }
    printf("launch + event record + query spin     %6.2f us\n", (now() - t0) / N);
    t0 = now();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny, 1, 64, 0, s, dm, dm + 1, nullptr, 0u); while (hipStreamQuery(s) == hipErrorNotReady) {} }
    printf("launch + hipStreamQuery spin           %6.2f us\n", (now() - t0) / N);
    t0 = now();
    for (int i = 0; i < N; ++i) {
        const unsigned seq = (unsigned)i + 1;
        hipLaunchKernelGGL(tiny, 1, 64, 0, s, dm, dm + 1, dflag, seq);
        while (*hflag != seq) {}
    }
    printf("launch + kernel-written flag + poll    %6.2f us\n", (now() - t0) / N);
    CK(hipStreamSynchronize(s));
    {
        hipError_t e = hipStreamWriteValue32(s, (void*)dflag, 7u, 0);
        if (e != hipSuccess) printf("hipStreamWriteValue32 unsupported here: %s\n", hipGetErrorString(e));
        else {
            CK(hipStreamSynchronize(s));
            t0 = now();
            for (int i = 0; i < N; ++i) {
                const unsigned seq = 0x10000000u + (unsigned)i;
                hipLaunchKernelGGL(tiny, 1, 64, 0, s, dm, dm + 1, nullptr, 0u);
                CK(hipStreamWriteValue32(s, (void*)dflag, seq, 0));
                while (*hflag != seq) {}
            }
            printf("launch + stream write-value + poll     %6.2f us\n", (now() - t0) / N);
        }
    }
    // launch cost alone (no wait), then drain
    t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, 1, 64, 0, s, dm, dm + 1, nullptr, 0u);
    double t1 = now(); CK(hipStreamSynchronize(s));
    printf("launch only (host side, back to back)  %6.2f us  (drained after %.2f us per launch)\n", (t1 - t0) / N, (now() - t0) / N);

    // the C ABI: 1-lane host-mapped handle, staged step (what the facade calls)
    soccer_config cfg{}; cfg.n_lanes = 1; cfg.width = 5; cfg.height = 4; cfg.slip_prob = 0.0; cfg.max_steps = 100; cfg.flags |= SOCCER_F_AUTORESET;
    cfg.flags |= SOCCER_F_HOST_MAPPED;
    soccer_handle* h = nullptr;
    if (soccer_create(&cfg, &h)) { printf("create: %s\n", soccer_last_error(nullptr)); return 1; }
    soccer_staging_view v{};
    if (soccer_staging(h, &v)) { printf("staging: %s\n", soccer_last_error(h)); return 1; }
    v.u_reset[0] = 0.3; batched_reset_staged(h, SOCCER_STAGE_U_RESET);
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now();
        for (int i = 0; i < N; ++i) {
            v.act_a[0] = 0; v.act_b[0] = 0; v.u_step[0] = 0.5;
            if (batched_step_staged(h, SOCCER_STAGE_ACT_A | SOCCER_STAGE_ACT_B | SOCCER_STAGE_U_STEP)) { printf("step: %s\n", soccer_last_error(h)); return 1; }
        }
        printf("C ABI batched_step_staged, mapped 1 lane %6.2f us\n", (now() - t0) / N);
    }
    soccer_destroy(h);
    // the scalar entry point: inputs as kernel arguments, result polled from a mapped record
    for (double slip : {0.0, 0.2}) {
        soccer_config c2{}; c2.n_lanes = 1; c2.width = 5; c2.height = 4; c2.slip_prob = slip; c2.max_steps = 100;
        soccer_handle* h2 = nullptr;
        if (soccer_create(&c2, &h2)) { printf("create: %s\n", soccer_last_error(nullptr)); return 1; }
        soccer_scalar_io io{}; io.u_reset = 0.3;
        if (soccer_reset_scalar(h2, &io)) { printf("reset: %s\n", soccer_last_error(h2)); return 1; }
        for (int rep = 0; rep < 2; ++rep) {
            t0 = now();
            for (int i = 0; i < N; ++i) {
                io.act_a = (int8_t)(i % 5); io.act_b = (int8_t)((i / 5) % 5); io.u_step = 0.37; io.t = 0;
                if (io.needs_reset) { soccer_reset_scalar(h2, &io); continue; }
                if (soccer_step_scalar(h2, &io)) { printf("scalar: %s\n", soccer_last_error(h2)); return 1; }
            }
            printf("C ABI soccer_step_scalar, slip %.1f          %6.2f us\n", slip, (now() - t0) / N);
        }
        soccer_destroy(h2);
    }
    return 0;
}
