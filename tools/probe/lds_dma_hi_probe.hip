// Probe: does LDS-DMA reach LDS addresses above 64 KiB (M0 wider than 16 bits)?  Three forms: buffer_load ... lds from inline assembly
// (the gemm_dma.hip / gemm_x3.hip form), the raw_ptr_buffer_load_lds builtin, and the global_load_lds builtin.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/lds_dma_hi_probe.hip -o tools/probe/lds_dma_hi_probe && tools/probe/lds_dma_hi_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int NF = 36864;                                  // 147,456 bytes of LDS
__global__ __launch_bounds__(256) void k(const float* p, float* out, int form, int base) {
    __shared__ __attribute__((aligned(16))) float sm[NF];
    for (int i = threadIdx.x; i < NF; i += 256) sm[i] = -7.f;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* dst = sm + base + wave * 256;                   // 1 KiB per wave
    const unsigned off = threadIdx.x * 16u;
    if (form == 0) {
        const unsigned long long a = (unsigned long long)p;
        i32x4 r = {(int)(unsigned)(a & 0xffffffffu), (int)(unsigned)((a >> 32) & 0xffffu), 4096, 0x00020000};
        const unsigned la = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const float*)dst;
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(la), "v"(off), "s"(r) : "memory");
    } else if (form == 1) {
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 4096, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
    } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + threadIdx.x * 4), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < NF; i += 256) out[i] = sm[i];
}
int main() {
    float *p, *o, *h = new float[NF], src[1024];
    for (int i = 0; i < 1024; ++i) src[i] = 1.f + i;
    hipMalloc(&p, 4096); hipMalloc(&o, NF * 4);
    hipMemcpy(p, src, 4096, hipMemcpyHostToDevice);
    const int bases[] = {0, 8192, 16384 - 1024, 16384, 20000 * 1, 32768, 35840};
    for (int form = 0; form < 3; ++form)
        for (int base : bases) {
            hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, p, o, form, base);
            hipMemcpy(h, o, NF * 4, hipMemcpyDeviceToHost);
            int good = 0, first = -1, touched = 0;
            for (int i = 0; i < 1024; ++i) good += h[base + i] == src[i];
            for (int i = 0; i < NF; ++i) if (h[i] != -7.f) { ++touched; if (first < 0) first = i; }
            printf("form %d, LDS byte base %6d: %4d/1024 correct at the target; %d floats touched, first at byte %d\n", form, base * 4, good, touched, first * 4);
        }
    return 0;
}
