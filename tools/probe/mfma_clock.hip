// What fp32-MFMA rate does the chip actually hold under load?  Bare MFMA loops on random register operands, the two f32
// shapes (32x32x2: 64 cycles, 16x16x4: 32 cycles -- the same 64 FLOP/cycle/SIMD), 1-3 waves per SIMD, launched back to back
// for >= 2 s; reports wall TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime, MI355X_MICROARCH.md DVFS give-back
// item 6).  Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/mfma_clock tools/probe/mfma_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void loop(const float* __restrict__ in, float* __restrict__ out, unsigned long long* stamps, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[tid * 8 + i]; b[i] = in[tid * 8 + 4 + i]; }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (SHAPE == 32) {                                       // 4 accumulators of 32x32 = the 64x64 wave tile of the GEMM kernels
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[(kk + 1) & 3], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(kk + 1) & 3], b[kk], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(kk + 2) & 3], b[(kk + 3) & 3], acc[3], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {                                                 // 16 accumulators of 16x16: the same 64x64 wave tile
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)                   // 2 x 16 MFMAs of 2048 FLOP = the 4 x 4 MFMAs of 4096 FLOP above... x2
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + kk) & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[tid] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = c1 - c0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(int blocks_per_cu, const float* in, float* out, unsigned long long* stamps, int iters) {
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(loop<SHAPE>, dim3(blocks), dim3(256), 0, 0, in, out, stamps, iters);
    hipDeviceSynchronize();
    float ms = 0.f; int launches = 0; double total = 0.0;
    while (total < 2500.0) {                                 // >= 2 s back to back, time the last batch
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(loop<SHAPE>, dim3(blocks), dim3(256), 0, 0, in, out, stamps, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        total += ms; launches = 20;
    }
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * blocks * 2, hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (int b = 0; b < blocks; ++b) if (h[2 * b + 1]) clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
    std::sort(clk.begin(), clk.end());
    const double flop = (double)blocks * 4 /*waves*/ * iters * 16.0 * 4096.0;    // both shapes: 65536 FLOP per wave-iteration
    printf("shape %2d  %d wave(s)/SIMD  %7.1f TFLOP/s  in-kernel clock median %.0f MHz (min %.0f max %.0f)  %.3f ms/launch\n", SHAPE, blocks_per_cu,
           flop * launches / (ms * 1e-3) / 1e12, clk.empty() ? 0.0 : clk[clk.size() / 2], clk.empty() ? 0.0 : clk.front(), clk.empty() ? 0.0 : clk.back(), ms / launches);
}

int main() {
    const int maxblocks = 256 * 3;
    float* in; float* out; unsigned long long* stamps;
    hipMalloc(&in, sizeof(float) * maxblocks * 256 * 8); hipMalloc(&out, sizeof(float) * maxblocks * 256); hipMalloc(&stamps, 16 * maxblocks);
    std::vector<float> h(maxblocks * 256 * 8);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice);
    const int iters = 4000;
    for (int w = 1; w <= 3; ++w) { run<32>(w, in, out, stamps, iters); run<16>(w, in, out, stamps, iters); }
    // zero operands: the clock the chip holds when the multipliers toggle nothing
    hipMemset(in, 0, sizeof(float) * h.size());
    run<32>(1, in, out, stamps, iters); run<16>(1, in, out, stamps, iters);
    return 0;
}
