// hipGraph with two captured branches against two eager streams (why did the step's graph with the
// weight-gradient stream forked inside the capture replay slower than eager streams?).
// Two independent chains of NK kernels; each kernel is `blocks` workgroups spinning for ~`us` microseconds (half the chip each by
// default, so that two chains can run side by side).  Timed, wall clock over REP repetitions each:
//   serial     both chains on one stream
//   streams    chain A on stream 0, chain B on stream 1 (fork / join with events), eager launches
//   graph1     the serial form captured into one hipGraph
//   graph2     the two-stream form captured (fork / join inside the capture) -> a graph with two branches
//   graph2x    the same graph launched with explicit dependencies built by hand (hipGraphAddKernelNode), no capture
//   2graphs    chain A and chain B as two linear graphs launched on two streams;  g+eager: graph A on stream 0 beside eager B on stream 1
//   ladder / ladder-g   the weight-gradient pattern: B_i waits for A_i only (one event per rung), eager / captured
// and, from in-kernel timestamps (s_memrealtime, 100 MHz) of the graph2 replay: how much of chain B ran while chain A was running.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/graph_branch tools/probe/graph_branch.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void spin(unsigned long long* stamps, int slot, int ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0 && blockIdx.x == 0 && stamps != nullptr) { stamps[2 * slot] = t0; stamps[2 * slot + 1] = __builtin_amdgcn_s_memrealtime(); }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int NK = argc > 1 ? atoi(argv[1]) : 200;          // kernels per chain
    const int us = argc > 2 ? atoi(argv[2]) : 20;           // duration of each kernel
    const int blocks = argc > 3 ? atoi(argv[3]) : 128;      // workgroups per kernel (256 CUs on the chip)
    const int REP = 20, ticks = us * 100;
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, sizeof(unsigned long long) * 4 * NK));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    auto serial = [&](hipStream_t s) {
        for (int i = 0; i < NK; ++i) { hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, stamps, i, ticks); hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, stamps, NK + i, ticks); }
    };
    auto two = [&]() {
        CK(hipEventRecord(fork, s0));
        CK(hipStreamWaitEvent(s1, fork, 0));
        for (int i = 0; i < NK; ++i) { hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s0, stamps, i, ticks); hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, stamps, NK + i, ticks); }
        CK(hipEventRecord(join, s1));
        CK(hipStreamWaitEvent(s0, join, 0));
    };
    // the weight-gradient pattern of the train step: kernel B_i (side stream) may start once A_i (main chain) is done; joined at the end only
    std::vector<hipEvent_t> evs(NK);
    for (auto& e : evs) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    auto ladder = [&]() {
        for (int i = 0; i < NK; ++i) {
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s0, stamps, i, ticks);
            CK(hipEventRecord(evs[i], s0));
            CK(hipStreamWaitEvent(s1, evs[i], 0));
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, stamps, NK + i, ticks);
        }
        CK(hipEventRecord(join, s1));
        CK(hipStreamWaitEvent(s0, join, 0));
    };
    auto timeit = [&](const char* name, auto&& fn) {
        fn(); CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int r = 0; r < REP; ++r) fn();
        const double t1 = now();
        CK(hipDeviceSynchronize());
        const double t2 = now();
        printf("%-8s %8.1f us per repetition (host issue %7.1f us)   ideal serial %d us, ideal overlapped %d us\n", name, (t2 - t0) / REP * 1e6, (t1 - t0) / REP * 1e6, 2 * NK * us, NK * us);
    };
    printf("chains of %d kernels x %d us x %d workgroups\n", NK, us, blocks);
    timeit("serial", [&] { serial(s0); });
    timeit("streams", two);
    hipGraph_t g1, g2, g3;
    hipGraphExec_t e1, e2, e3;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal)); serial(s0); CK(hipStreamEndCapture(s0, &g1));
    CK(hipGraphInstantiate(&e1, g1, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal)); two(); CK(hipStreamEndCapture(s0, &g2));
    CK(hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0));
    // hand-built: two chains, explicit edges
    CK(hipGraphCreate(&g3, 0));
    {
        hipGraphNode_t prev[2] = {nullptr, nullptr};
        for (int i = 0; i < NK; ++i)
            for (int c = 0; c < 2; ++c) {
                int slot = c * NK + i, tk = ticks;
                void* args[3] = {&stamps, &slot, &tk};
                hipKernelNodeParams p = {};
                p.func = (void*)spin; p.gridDim = dim3(blocks); p.blockDim = dim3(256); p.kernelParams = args;
                hipGraphNode_t n;
                CK(hipGraphAddKernelNode(&n, g3, prev[c] ? &prev[c] : nullptr, prev[c] ? 1 : 0, &p));
                prev[c] = n;
            }
    }
    CK(hipGraphInstantiate(&e3, g3, nullptr, nullptr, 0));
    hipGraph_t g4; hipGraphExec_t e4;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal)); ladder(); CK(hipStreamEndCapture(s0, &g4));
    CK(hipGraphInstantiate(&e4, g4, nullptr, nullptr, 0));
    // two LINEAR graphs (chain A, chain B) launched on two streams
    hipGraph_t ga, gb; hipGraphExec_t ea, eb;
    auto chain = [&](hipStream_t s, int base) { for (int i = 0; i < NK; ++i) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, stamps, base + i, ticks); };
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal)); chain(s0, 0); CK(hipStreamEndCapture(s0, &ga));
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal)); chain(s1, NK); CK(hipStreamEndCapture(s1, &gb));
    CK(hipGraphInstantiate(&ea, ga, nullptr, nullptr, 0));
    CK(hipGraphInstantiate(&eb, gb, nullptr, nullptr, 0));
    auto twographs = [&]() {
        CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0));
        CK(hipGraphLaunch(ea, s0)); CK(hipGraphLaunch(eb, s1));
        CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0));
    };
    // a linear graph on stream 0 beside EAGER launches on stream 1
    auto graph_eager = [&]() {
        CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0));
        CK(hipGraphLaunch(ea, s0)); chain(s1, NK);
        CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0));
    };
    timeit("2graphs", twographs);
    timeit("g+eager", graph_eager);
    timeit("ladder", ladder);
    timeit("ladder-g", [&] { CK(hipGraphLaunch(e4, s0)); });
    timeit("graph1", [&] { CK(hipGraphLaunch(e1, s0)); });
    timeit("graph2", [&] { CK(hipGraphLaunch(e2, s0)); });
    timeit("graph2x", [&] { CK(hipGraphLaunch(e3, s0)); });
    // overlap inside one graph2 / graph2x / streams replay, from the in-kernel stamps
    std::vector<unsigned long long> h(4 * NK);
    auto overlap = [&](const char* name, auto&& fn) {
        CK(hipMemset(stamps, 0, sizeof(unsigned long long) * 4 * NK));
        fn(); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 4 * NK, hipMemcpyDeviceToHost));
        unsigned long long a0 = h[0], a1 = h[2 * (NK - 1) + 1], b0 = h[2 * NK], b1 = h[2 * (2 * NK - 1) + 1];
        unsigned long long lo = a0 < b0 ? a0 : b0, hi = a1 > b1 ? a1 : b1;
        double gapA = 0, gapB = 0;
        for (int i = 1; i < NK; ++i) { gapA += (double)(h[2 * i] - h[2 * (i - 1) + 1]); gapB += (double)(h[2 * (NK + i)] - h[2 * (NK + i - 1) + 1]); }
        printf("%-8s chain A %7.1f us, chain B %7.1f us, both %7.1f us; mean gap between consecutive kernels of a chain: A %.2f us, B %.2f us\n", name,
               (a1 - a0) / 100.0, (b1 - b0) / 100.0, (hi - lo) / 100.0, gapA / (NK - 1) / 100.0, gapB / (NK - 1) / 100.0);
    };
    overlap("2graphs", twographs);
    overlap("g+eager", graph_eager);
    overlap("ladder", ladder);
    overlap("ladder-g", [&] { CK(hipGraphLaunch(e4, s0)); });
    overlap("serial", [&] { serial(s0); });
    overlap("streams", two);
    overlap("graph1", [&] { CK(hipGraphLaunch(e1, s0)); });
    overlap("graph2", [&] { CK(hipGraphLaunch(e2, s0)); });
    overlap("graph2x", [&] { CK(hipGraphLaunch(e3, s0)); });
    return 0;
}
