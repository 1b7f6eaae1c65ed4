// Probe: does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer descriptor) WRITE ZEROS to
// its LDS slot, or leave the slot untouched?  (wgemm_tn_dma masks padded taps / rows past the split by an out-of-range offset.)
//   hipcc --offload-arch=gfx950 -O3 tools/probe/buf_lds_probe.hip -o tools/probe/buf_lds_probe && tools/probe/buf_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* p, float* out, int n) {
    __shared__ __attribute__((aligned(16))) float sm[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = -7.f;          // sentinel
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, n * 4, 0x00020000);
    const unsigned off = (threadIdx.x & 1) ? 0xffffffffu : threadIdx.x * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(sm + wave * 256), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = sm[i];
}
int main() {
    float *p, *o, h[1024], src[1024];
    for (int i = 0; i < 1024; ++i) src[i] = 1.f + i;
    hipMalloc(&p, 4096); hipMalloc(&o, 4096);
    hipMemcpy(p, src, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, p, o, 1024);
    hipMemcpy(h, o, 4096, hipMemcpyDeviceToHost);
    int zeros = 0, stale = 0, good = 0, other = 0;
    for (int t = 0; t < 256; ++t)
        for (int e = 0; e < 4; ++e) {
            const float v = h[t * 4 + e];
            if (t & 1) { if (v == 0.f) ++zeros; else if (v == -7.f) ++stale; else ++other; }
            else { if (v == src[t * 4 + e]) ++good; else ++other; }
        }
    printf("in-range lanes correct: %d/512; out-of-range lanes: %d zeros, %d untouched (sentinel), %d other\n", good, zeros, stale, other);
    return 0;
}
