// What does a timed region of K dependent batched_step launches cost the host, by how the K launches are enqueued?
// bench.py's region is "device idle -> K steps -> hipDeviceSynchronize"; at the driver's K = 20 a graph replay spends 15 us
// before the device starts and the closing synchronisation another 13 - 18 us after it has finished (profiles/r03_summary.md
// section 2).  Variants, each through the C ABI only (2^20 lanes, 5x4, slip 0, the 8-argument batched_step):
//   graph      one replay of a K-step captured graph                       (what bench.py times)
//   eager      K batched_step calls from a C loop
//   head+graph `h` eager calls, then a replay of a (K - h)-step graph       (the replay's start-up hides behind the head)
//   +tail      ... and the last `t` steps eager again                       (the region then ends on an ordinary dispatch)
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/labs/region_lab.hip -o build/region_lab \
//         -Lgym_soccer_littman94_amd -lsoccer_hip -Wl,-rpath,'$ORIGIN/../gym_soccer_littman94_amd'
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "soccer_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define SK(x) do { int e_ = (x); if (e_) { printf("%s -> %d %s\n", #x, e_, soccer_last_error(h)); exit(1); } } while (0)

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int K = argc > 1 ? std::atoi(argv[1]) : 20;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 40;
    const uint64_t n = 1ull << 20;
    soccer_config cfg{}; cfg.n_lanes = n; cfg.width = 5; cfg.height = 4; cfg.slip_prob = 0.0; cfg.max_steps = 100; cfg.seed = 1;
    cfg.flags |= SOCCER_F_AUTORESET;
    soccer_handle* h = nullptr;
    if (soccer_create(&cfg, &h)) { printf("create: %s\n", soccer_last_error(nullptr)); return 1; }
    int8_t *A, *B; uint16_t* obs; int8_t* rew; uint8_t *te, *tr;
    SK(soccer_malloc(h, n * K, (void**)&A)); SK(soccer_malloc(h, n * K, (void**)&B));
    SK(soccer_malloc(h, n * 2 * K, (void**)&obs)); SK(soccer_malloc(h, n * K, (void**)&rew));     // [K, n] trajectories, like bench.py
    SK(soccer_malloc(h, n * K, (void**)&te)); SK(soccer_malloc(h, n * K, (void**)&tr));
    { std::vector<int8_t> a(n * K); for (size_t i = 0; i < a.size(); ++i) a[i] = (int8_t)((i * 2654435761u >> 13) % 5u);
      SK(soccer_memcpy_h2d(h, A, a.data(), a.size())); for (auto& v : a) v = (int8_t)((v + 2) % 5); SK(soccer_memcpy_h2d(h, B, a.data(), a.size())); }
    SK(batched_reset(h, nullptr, nullptr, nullptr));
    auto step = [&](int k) { const size_t o = (size_t)k * n; SK(batched_step(h, A + o, B + o, obs + o, rew + o, te + o, tr + o, nullptr)); };
    auto capture = [&](int k0, int k1) { soccer_graph* g = nullptr; SK(soccer_graph_begin(h)); for (int k = k0; k < k1; ++k) step(k); SK(soccer_graph_end(h, &g)); return g; };
    struct Variant { const char* name; int head, tail; soccer_graph* g; };
    std::vector<Variant> V;
    V.push_back({"graph", 0, 0, capture(0, K)});
    V.push_back({"eager", K, 0, nullptr});
    for (int hd : {2, 4, 6}) { if (K - hd < 2) continue; char* nm = new char[32]; snprintf(nm, 32, "head %d + graph", hd); V.push_back({nm, hd, 0, capture(hd, K)}); }
    for (int hd : {0, 2, 4}) { if (K - hd - 2 < 2) continue; char* nm = new char[40]; snprintf(nm, 40, "head %d + graph + tail 2", hd); V.push_back({nm, hd, 2, capture(hd, K - 2)}); }
    for (int w = 0; w < 3; ++w) for (auto& v : V) { for (int k = 0; k < v.head; ++k) step(k); if (v.g) SK(soccer_graph_launch(h, v.g, 1)); for (int k = K - v.tail; k < K; ++k) step(k); }
    CK(hipDeviceSynchronize());
    for (int pass = 0; pass < 2; ++pass)
    for (auto& v : V) {
        std::vector<double> wall, enq;
        for (int r = 0; r < reps; ++r) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (int k = 0; k < v.head; ++k) step(k);
            if (v.g) SK(soccer_graph_launch(h, v.g, 1));
            for (int k = K - v.tail; k < K; ++k) step(k);
            const double t1 = now();
            CK(hipDeviceSynchronize());
            const double t2 = now();
            wall.push_back(t2 - t0); enq.push_back(t1 - t0);
        }
        std::sort(wall.begin(), wall.end()); std::sort(enq.begin(), enq.end());
        printf("K=%d %-26s wall median %7.1f us  min %7.1f  p90 %7.1f   enqueue median %6.1f us   => %.2f us per step\n", K, v.name,
               wall[reps / 2], wall[0], wall[reps * 9 / 10], enq[reps / 2], wall[reps / 2] / K);
    }
    soccer_destroy(h);
    return 0;
}
