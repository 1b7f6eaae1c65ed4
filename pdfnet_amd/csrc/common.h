// Shared helpers for the gfx950 kernels of libpdfnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PDF_API extern "C" __attribute__((visibility("default")))

// Every entry point returns 0 on success or a negative PDF_E_* / positive hipError_t code.
#define PDF_E_BADARG (-1)
#define PDF_E_WORKSPACE (-2)

#define PDF_LAUNCH_CHECK()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// memory-bound launches: cap the grid and grid-stride the rest (guide: Guideline 11)
static inline int grid_for(long n, int block = 256, int cap = 256 * 8) {
    long g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// stateless counter RNG for dropout masks (same mask regenerated in backward from seed+index)
__device__ __forceinline__ uint32_t pdf_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float pdf_uniform(uint64_t seed, uint64_t idx) {
    uint32_t h = pdf_hash32((uint32_t)idx ^ pdf_hash32((uint32_t)(idx >> 32) + (uint32_t)seed) ^ (uint32_t)(seed >> 32) * 0x9E3779B9U);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
