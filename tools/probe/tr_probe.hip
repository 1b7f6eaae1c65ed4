// Probe of ds_read_b64_tr_b16 lane mapping: LDS image sm[k][n] (16-bit), value = k*100 + n.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int S = 160;   // row stride in elements
__global__ void k(s16x4* out) {
    __shared__ __attribute__((aligned(16))) short sm[64 * S];
    for (int i = threadIdx.x; i < 64 * S; i += 64) sm[i] = (short)((i / S) * 100 + (i % S));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int cb = 16 * (g & 1), kb = 8 * (g >> 1);
    out[lane] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm + (kb + q) * S + cb + 4 * p));
}
int main() {
    s16x4* d; hipMalloc(&d, 64 * sizeof(s16x4));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    s16x4 h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %d %d %d %d\n", l, h[l][0], h[l][1], h[l][2], h[l][3]);
    return 0;
}
