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

// ---------------------------------------------------------------------------------------------------------------
// "Last block finishes the job": the blocks of one launch that contribute partials to the same output each take a ticket
// from a device counter after publishing their partials; the block that draws the last ticket sees everybody's partials and
// reduces them in a FIXED order (deterministic), which removes the separate reduce launch and its kernel boundary.
// Protocol = the guide's split-K recipe (cdna_hip_programming.md section 5, "In-launch split-K reduction"; Guideline 16):
// producer: every storing wave drains its stores (s_waitcnt vmcnt(0)), workgroup barrier, lane 0 agent-scope RELEASE fence,
// explicit s_waitcnt vmcnt(0) (ROCm 7.2 may drop the fence's own wait), relaxed agent-scope fetch_add;
// consumer (the last arriver): lane 0 agent-scope ACQUIRE fence, s_waitcnt vmcnt(0), workgroup barrier, plain loads.
// Correct for any placement of the blocks on XCDs / CUs.  The last arriver resets the counter, so counters only need to be
// zero once (hipMemset at allocation).  `flag` is a word of LDS.
// `release`: true = the partials were written with plain stores (agent-scope release fence: writes back the XCD's dirty L2
// lines -- expensive when the L2 holds megabytes of a previous kernel's output, so use it for a few blocks per launch only);
// false = the partials were written with pdf_store_wt (write-through stores, no fence needed: Guideline 16 R1).
__device__ __forceinline__ void pdf_store_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool pdf_last_block_arrives(int* counter, int expected, int* flag, bool release = true) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (release) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const int t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == expected - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *flag = last;
    }
    __syncthreads();
    return *flag != 0;
}

// Ticket counters: a zero-initialised device ring shared by every launch of the library; each launch takes a fresh region, and
// a region is only handed out again after PDF_COUNTER_RING / (counters per step) steps -- long after the launch that used it
// (and reset it) has finished.  Defined in elementwise.hip.
#define PDF_COUNTER_RING (1 << 22)
int* pdf_ticket_counters(int n);
// Scratch for the split-K partial sums of the small-M GEMMs (entry points without a workspace argument): a 256 MiB ring
// allocated once (pdf_init), handed out in launch order.  One use takes at most 16 MiB and a train step a few tens of MiB, so
// a region comes round again only several steps later -- far beyond what the launch queue can hold in flight.
#define PDF_SCRATCH_RING (1L << 26)
#define PDF_SCRATCH_MAX (1L << 22)
float* pdf_scratch(long floats);
// bf16 shadows handed to the NEXT entry-point call of this thread (pdf_set_bf16_operands / pdf_set_bf16_output, elementwise.hip):
// every GEMM-family entry point takes (and clears) the operand pair first thing, the BatchNorm / pyramid entry points the output.
typedef __bf16 pdf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pdf_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pdf_pk_bf16(float a, float b) {      // two floats -> packed bf16 pair, round-to-nearest-even
    pdf_f32x2 f = {a, b};
    pdf_bf16x2 v = __builtin_convertvector(f, pdf_bf16x2);
    return *reinterpret_cast<unsigned int*>(&v);
}
// x3 arithmetic (gemm_x3.hip): a fp32 pair -> its three bf16 components, packed pairs h / m / l with a = h + m + l exactly
__device__ __forceinline__ void pdf_x3_split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = pdf_pk_bf16(a, b);
    float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = pdf_pk_bf16(ra, rb);
    ra -= __uint_as_float(m << 16); rb -= __uint_as_float(m & 0xffff0000u);
    l = pdf_pk_bf16(ra, rb);
}
// ... stored at element offsets o, o + cs, o + 2 cs of a bf16 tensor (o even)
__device__ __forceinline__ void pdf_x3_store2(unsigned short* base, long o, long cs, float a, float b) {
    unsigned h, m, l;
    pdf_x3_split2(a, b, h, m, l);
    *reinterpret_cast<unsigned*>(base + o) = h;
    *reinterpret_cast<unsigned*>(base + o + cs) = m;
    *reinterpret_cast<unsigned*>(base + o + 2 * cs) = l;
}
// Everything a call may take beyond its positional arguments -- must match `PdfCallOpts` of include/pdfnet_hip.h field for field
// (tests/c_abi/c_client.c compares sizeof with pdf_debug_callopts_size()).  The `_x` form of an entry point takes it explicitly; the
// plain form takes (and clears) the thread's hand-over slots FIRST THING, so no return path leaves a slot armed.
struct PdfCallOpts {
    const void* op0_bf16; const void* op1_bf16;     // bf16 shadows of the call's two operands (GEMM family)
    void* out_bf16;                                  // bf16 shadow of the output, or the bf16-only output (storage mode)
    const void* bn_x_bf16;                           // BatchNorm: x as bf16 instead of the fp32 pointer
    float* stats_out; long stats_cap;                // conv / linear forward: BatchNorm statistics of the output from the epilogue
    long stats_tiles, stats_rows;                    // OUT: row blocks / rows per block written to stats_out (0: none)
    const float* tile_stats; long tile_n, tile_rows; // BatchNorm: such partials instead of its own statistics pass
    const float* in_scale; const float* in_shift;    // x read as relu(x * scale[k] + shift[k]) (linear fwd / weight gradient)
    const void* op1_bf16_t;                          // backward-data: the weight's TRANSPOSED bf16 shadow wt[c][tap][r] (pdf_cast_bf16_transposed)
    float* ws; long ws_floats;                       // conv2d forward / backward-data / weight gradient: workspace of pdf_conv2d_winograd_workspace_floats -> Winograd path
    const float* wino_v;                             // conv2d weight gradient: the forward's transformed input (pdf_conv2d_winograd_v_offset into ITS ws)
};
PdfCallOpts pdf_tls_take_all();                      // the thread's armed slots, cleared
void pdf_tls_publish(const PdfCallOpts& o);          // stats_tiles / stats_rows -> pdf_stats_result_*

