// IntagHand-style mesh decoder kernels for gfx950: Chebyshev (K=2) graph convolution as a fixed-width
// ELL SpMM (the reference densifies the Laplacian and calls torch.mm, model_attn/gcn.py:54,79-86: 409-1546
// non-zeros in a 63^2..252^2 matrix) and small-V multi-head attention with K/V of one (sample, head)
// resident in LDS (V <= 252, V*dh = 4032 floats: 16 KB each).
#include "common.h"

#define GRID_STRIDE(i, total) for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (total); i += (long)gridDim.x * blockDim.x)

// out[b][v][2f] = x[b][v][f];  out[b][v][2f+1] = sum_w val[v][w] * x[b][col[v][w]][f]
// (feature order fin*K + k, k fastest: gcn.py:61-63)
// samples b >= Bsplit use the second Laplacian (col1/val1): left and right hand graphs in one launch (DualGraph.py:83-84)
__global__ void cheby2_fwd_kernel(const float* __restrict__ x, int ldx, int V, int F, const int* __restrict__ col0, const float* __restrict__ val0,
                                  const int* __restrict__ col1, const float* __restrict__ val1, long Bsplit, int Wd,
                                  float* __restrict__ out, int ldo, long total) {
    GRID_STRIDE(i, total) {
        int f = (int)(i % F); long p = i / F;
        int v = (int)(p % V); long b = p / V;
        const int* __restrict__ col = b < Bsplit ? col0 : col1;
        const float* __restrict__ val = b < Bsplit ? val0 : val1;
        const float* xb = x + b * V * ldx;
        float acc = 0.f;
        for (int w = 0; w < Wd; ++w) acc += val[v * Wd + w] * xb[(long)col[v * Wd + w] * ldx + f];
        float* o = out + (b * V + v) * ldo + 2 * f;
        o[0] = xb[(long)v * ldx + f];
        o[1] = acc;
    }
}
PDF_API int pdf_cheby2_fwd(const float* x, int ldx, int B, int V, int F, const int* col, const float* val, int Wd,
                           float* out, int ldo, hipStream_t s) {
    long total = (long)B * V * F;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(cheby2_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, ldx, V, F, col, val, col, val, (long)B, Wd, out, ldo, total);
    PDF_LAUNCH_CHECK();
    return 0;
}
// paired: samples [0, B) with (col0, val0), samples [B, 2B) with (col1, val1)
PDF_API int pdf_cheby2_fwd_pair(const float* x, int ldx, int B, int V, int F, const int* col0, const float* val0,
                                const int* col1, const float* val1, int Wd, float* out, int ldo, hipStream_t s) {
    long total = (long)2 * B * V * F;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(cheby2_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, ldx, V, F, col0, val0, col1, val1, (long)B, Wd, out, ldo, total);
    PDF_LAUNCH_CHECK();
    return 0;
}
// dx[b][v][f] = d[b][v][2f] + sum_w valT[v][w] * d[b][colT[v][w]][2f+1]     (colT/valT = ELL of L^T)
__global__ void cheby2_bwd_kernel(const float* __restrict__ d, int ldd, int V, int F, const int* __restrict__ colT0, const float* __restrict__ valT0,
                                  const int* __restrict__ colT1, const float* __restrict__ valT1, long Bsplit, int Wd,
                                  float* __restrict__ dx, int lddx, long total) {
    GRID_STRIDE(i, total) {
        int f = (int)(i % F); long p = i / F;
        int v = (int)(p % V); long b = p / V;
        const int* __restrict__ colT = b < Bsplit ? colT0 : colT1;
        const float* __restrict__ valT = b < Bsplit ? valT0 : valT1;
        const float* db = d + b * V * ldd;
        float acc = db[(long)v * ldd + 2 * f];
        for (int w = 0; w < Wd; ++w) acc += valT[v * Wd + w] * db[(long)colT[v * Wd + w] * ldd + 2 * f + 1];
        dx[(b * V + v) * lddx + f] = acc;
    }
}
PDF_API int pdf_cheby2_bwd(const float* d, int ldd, int B, int V, int F, const int* colT, const float* valT, int Wd,
                           float* dx, int lddx, hipStream_t s) {
    long total = (long)B * V * F;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(cheby2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, d, ldd, V, F, colT, valT, colT, valT, (long)B, Wd, dx, lddx, total);
    PDF_LAUNCH_CHECK();
    return 0;
}
PDF_API int pdf_cheby2_bwd_pair(const float* d, int ldd, int B, int V, int F, const int* colT0, const float* valT0,
                                const int* colT1, const float* valT1, int Wd, float* dx, int lddx, hipStream_t s) {
    long total = (long)2 * B * V * F;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(cheby2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, d, ldd, V, F, colT0, valT0, colT1, valT1, (long)B, Wd, dx, lddx, total);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// softmax(q k^T / sqrt(dh)) v per (sample, head); heads are contiguous dh-slices of the feature axis
// (self_attn.py:63-76, inter_attn.py:82-105).  Dropout on the attention matrix (dropout1) uses the stateless mask of common.h.
// kv_shift: queries of sample b attend to keys / values of sample (b + kv_shift) % B -- with the two hands stacked along
// the batch axis, kv_shift = B/2 is the cross-hand attention of inter_attn.py:82-105 in one launch.
//
// Work split (round 2): a row (query row in fwd / bwd_q, key row in bwd_kv) belongs to G adjacent lanes; each owns a DH / G slice
// of the feature axis -- of the row's operand and of its accumulators (a handful of registers) -- and the dot products over
// the other index are completed with a shuffle reduction inside the group.  G is chosen so that a (sample, head) pair is ~1,024
// threads = 4 blocks, each staging its own copy of the two [V][DH] operands in LDS.  (Round 1 ran one row per thread in one
// block per pair: 63-252 busy threads on a CU, 128-256 accumulator registers per thread, ~100-250 us per launch.)
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = 1; o < G; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
static int attn_group(int V, int dh) {                 // lanes per row: power of two, V * G ~ 1024, at most 16 and at most dh
    int G = 1;
    while (G < 16 && G < dh && V * G * 2 <= 1024) G <<= 1;
    return G;
}

template <int DH, int G>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld,
                                                       int B, int V, int H, int kv_shift, float inv_norm, float pdrop, unsigned long long seed, const unsigned long long* __restrict__ step,
                                                       float* __restrict__ out, int ldo, float* __restrict__ stat /*[B][H][V][2] = max, sumexp*/) {
    constexpr int SL = DH / G, RPB = 256 / G;
    if (step != nullptr) seed += step[0] * 0x9E3779B97F4A7C15ull;
    extern __shared__ float sm[];
    float* sk = sm;              // [V][DH]
    float* sv = sm + V * DH;
    const int NT = (V + RPB - 1) / RPB;
    const int tile = blockIdx.x % NT, bh = blockIdx.x / NT;
    const int b = bh / H, h = bh % H;
    const long base = (long)b * V * ld + h * DH;
    const long kbase = (long)((b + kv_shift) % B) * V * ld + h * DH;
    for (int i = threadIdx.x; i < V * DH; i += 256) {
        int r = i / DH, d = i - r * DH;
        sk[i] = k[kbase + (long)r * ld + d];
        sv[i] = v[kbase + (long)r * ld + d];
    }
    __syncthreads();
    const float keep_scale = 1.f / (1.f - pdrop);
    const int row = tile * RPB + threadIdx.x / G, d0 = (threadIdx.x % G) * SL;
    const bool live = row < V;
    const int i = live ? row : V - 1;                    // dead rows compute along (the shuffles want whole groups) and write nothing
    float qr[SL], o[SL];
#pragma unroll
    for (int d = 0; d < SL; ++d) { qr[d] = q[base + (long)i * ld + d0 + d] * inv_norm; o[d] = 0.f; }
    // one pass, running maximum: o and l are rescaled whenever the maximum moves
    float m = -INFINITY, l = 0.f;
    const unsigned long long rowid = ((unsigned long long)(b * H + h) * V + i) * V;
    for (int j = 0; j < V; ++j) {
        float sdot = 0.f;
#pragma unroll
        for (int d = 0; d < SL; ++d) sdot += qr[d] * sk[j * DH + d0 + d];
        sdot = group_sum<G>(sdot);
        const float mn = fmaxf(m, sdot);
        const float sc = expf(m - mn);                   // 0 on the first key (m = -inf)
        float p = expf(sdot - mn);
        l = l * sc + p;
        if (pdrop > 0.f) p = pdf_uniform(seed, rowid + j) >= pdrop ? p * keep_scale : 0.f;
#pragma unroll
        for (int d = 0; d < SL; ++d) o[d] = o[d] * sc + p * sv[j * DH + d0 + d];
        m = mn;
    }
    if (live) {
        const float il = 1.f / l;
#pragma unroll
        for (int d = 0; d < SL; ++d) out[(long)b * V * ldo + (long)i * ldo + h * DH + d0 + d] = o[d] * il;
        if (d0 == 0) {
            stat[(((long)b * H + h) * V + i) * 2 + 0] = m;
            stat[(((long)b * H + h) * V + i) * 2 + 1] = l;
        }
    }
}

// ---- the same three kernels with the G lanes of a row striding over the OTHER index (keys in fwd / bwd_q, queries in bwd_kv)
// and a final shuffle reduction of the accumulators: every exp / dropout hash is evaluated once instead of G times, at the
// price of full-width accumulators per lane -- the better split for DH <= 32 (V = 126, 252: 45-90 us against 78-150 us),
// while DH = 64 wants the feature-axis split above (46-58 us against 70-220 us).  LDS rows are DH + 1 floats: the lanes of a
// group read G different rows at the same d.
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = 1; o < G; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int DH, int G>
__global__ __launch_bounds__(256) void attn_fwd_keys_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld,
                                                            int B, int V, int H, int kv_shift, float inv_norm, float pdrop, unsigned long long seed, const unsigned long long* __restrict__ step,
                                                            float* __restrict__ out, int ldo, float* __restrict__ stat) {
    constexpr int LD = DH + 1, RPB = 256 / G;
    if (step != nullptr) seed += step[0] * 0x9E3779B97F4A7C15ull;
    extern __shared__ float sm[];
    float* sk = sm;              // [V][LD]
    float* sv = sm + V * LD;
    float* sq = sm + 2 * V * LD; // [RPB][LD]: this block's query rows
    const int NT = (V + RPB - 1) / RPB;
    const int tile = blockIdx.x % NT, bh = blockIdx.x / NT;
    const int b = bh / H, h = bh % H;
    const long base = (long)b * V * ld + h * DH;
    const long kbase = (long)((b + kv_shift) % B) * V * ld + h * DH;
    for (int i = threadIdx.x; i < V * DH; i += 256) {
        int r = i / DH, d = i - r * DH;
        sk[r * LD + d] = k[kbase + (long)r * ld + d];
        sv[r * LD + d] = v[kbase + (long)r * ld + d];
    }
    for (int i = threadIdx.x; i < RPB * DH; i += 256) {
        int r = i / DH, d = i - r * DH;
        sq[r * LD + d] = q[base + (long)min(tile * RPB + r, V - 1) * ld + d] * inv_norm;
    }
    __syncthreads();
    const float keep_scale = 1.f / (1.f - pdrop);
    const int lr = threadIdx.x / G, row = tile * RPB + lr, gp = threadIdx.x % G;
    const bool live = row < V;
    const int i = live ? row : V - 1;
    float qr[DH], o[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) { qr[d] = sq[lr * LD + d]; o[d] = 0.f; }
    float m = -INFINITY;
    for (int j = gp; j < V; j += G) {
        float sdot = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) sdot += qr[d] * sk[j * LD + d];
        m = fmaxf(m, sdot);
    }
    m = group_max<G>(m);
    float l = 0.f;
    const unsigned long long rowid = ((unsigned long long)(b * H + h) * V + i) * V;
    for (int j = gp; j < V; j += G) {
        float sdot = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) sdot += qr[d] * sk[j * LD + d];
        float p = expf(sdot - m);
        l += p;
        if (pdrop > 0.f) p = pdf_uniform(seed, rowid + j) >= pdrop ? p * keep_scale : 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] += p * sv[j * LD + d];
    }
    l = group_sum<G>(l);
    const float il = 1.f / l;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        const float od = group_sum<G>(o[d]);
        if (live && gp == (d % G)) out[(long)b * V * ldo + (long)i * ldo + h * DH + d] = od * il;
    }
    if (live && gp == 0) {
        stat[(((long)b * H + h) * V + i) * 2 + 0] = m;
        stat[(((long)b * H + h) * V + i) * 2 + 1] = l;
    }
}
template <int DH, int G>
__global__ __launch_bounds__(256) void attn_bwd_q_keys_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld,
                                                              const float* __restrict__ o, const float* __restrict__ dout, int ldo,
                                                              const float* __restrict__ stat, int B, int V, int H, int kv_shift, float inv_norm, float pdrop, unsigned long long seed, const unsigned long long* __restrict__ step,
                                                              float* __restrict__ dq, int lddq, float* __restrict__ dvec) {
    constexpr int LD = DH + 1, RPB = 256 / G;
    if (step != nullptr) seed += step[0] * 0x9E3779B97F4A7C15ull;
    extern __shared__ float sm[];
    float* sk = sm;
    float* sv = sm + V * LD;
    float* sq = sm + 2 * V * LD; // [RPB][LD] x 3: this block's rows of q, dO, O
    float* sgo = sq + RPB * LD;
    float* so = sgo + RPB * LD;
    const int NT = (V + RPB - 1) / RPB;
    const int tile = blockIdx.x % NT, bh = blockIdx.x / NT;
    const int b = bh / H, h = bh % H;
    const long base = (long)b * V * ld + h * DH;
    const long kbase = (long)((b + kv_shift) % B) * V * ld + h * DH;
    for (int i = threadIdx.x; i < V * DH; i += 256) {
        int r = i / DH, d = i - r * DH;
        sk[r * LD + d] = k[kbase + (long)r * ld + d];
        sv[r * LD + d] = v[kbase + (long)r * ld + d];
    }
    for (int i = threadIdx.x; i < RPB * DH; i += 256) {
        int r = i / DH, d = i - r * DH;
        const long rr = min(tile * RPB + r, V - 1);
        sq[r * LD + d] = q[base + rr * ld + d] * inv_norm;
        sgo[r * LD + d] = dout[(long)b * V * ldo + rr * ldo + h * DH + d];
        so[r * LD + d] = o[(long)b * V * ldo + rr * ldo + h * DH + d];
    }
    __syncthreads();
    const float keep_scale = 1.f / (1.f - pdrop);
    const int lr = threadIdx.x / G, row = tile * RPB + lr, gp = threadIdx.x % G;
    const bool live = row < V;
    const int i = live ? row : V - 1;
    float qr[DH], go[DH], acc[DH];
    float D = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) { qr[d] = sq[lr * LD + d]; go[d] = sgo[lr * LD + d]; D += go[d] * so[lr * LD + d]; acc[d] = 0.f; }
    const long si = ((long)b * H + h) * V + i;
    const float m = stat[si * 2], il = 1.f / stat[si * 2 + 1];
    if (live && gp == 0) dvec[si] = D;
    const unsigned long long rowid = (unsigned long long)si * V;
    for (int j = gp; j < V; j += G) {
        float sdot = 0.f, gv = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) { sdot += qr[d] * sk[j * LD + d]; gv += go[d] * sv[j * LD + d]; }
        const float a = expf(sdot - m) * il;
        if (pdrop > 0.f) gv = pdf_uniform(seed, rowid + j) >= pdrop ? gv * keep_scale : 0.f;
        const float ds = a * (gv - D) * inv_norm;
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] += ds * sk[j * LD + d];
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        const float ad = group_sum<G>(acc[d]);
        if (live && gp == (d % G)) dq[(long)b * V * lddq + (long)i * lddq + h * DH + d] = ad;
    }
}
template <int DH, int G>
__global__ __launch_bounds__(256) void attn_bwd_kv_keys_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld,
                                                               const float* __restrict__ dout, int ldo, const float* __restrict__ stat,
                                                               const float* __restrict__ dvec, int B, int V, int H, int kv_shift, float inv_norm, float pdrop, unsigned long long seed, const unsigned long long* __restrict__ step,
                                                               float* __restrict__ dk, float* __restrict__ dv, int lddk) {
    constexpr int LD = DH + 1, RPB = 256 / G;
    if (step != nullptr) seed += step[0] * 0x9E3779B97F4A7C15ull;
    extern __shared__ float sm[];
    float* sq = sm;              // [V][LD]
    float* sg = sm + V * LD;     // dO
    float* ss = sm + 2 * V * LD; // [V][3]: max, 1 / sumexp, D
    float* skt = ss + 3 * V;     // [RPB][LD] x 2: this block's key / value rows
    float* svt = skt + RPB * LD;
    const int NT = (V + RPB - 1) / RPB;
    const int tile = blockIdx.x % NT, bh = blockIdx.x / NT;
    const int b = bh / H, h = bh % H;
    const long base = (long)b * V * ld + h * DH;
    const int bkv = (b + kv_shift) % B;
    const long kbase = (long)bkv * V * ld + h * DH;
    const long sb = ((long)b * H + h) * V;
    for (int i = threadIdx.x; i < V * DH; i += 256) {
        int r = i / DH, d = i - r * DH;
        sq[r * LD + d] = q[base + (long)r * ld + d];
        sg[r * LD + d] = dout[(long)b * V * ldo + (long)r * ldo + h * DH + d];
    }
    for (int i = threadIdx.x; i < RPB * DH; i += 256) {
        int r = i / DH, d = i - r * DH;
        const long rr = min(tile * RPB + r, V - 1);
        skt[r * LD + d] = k[kbase + rr * ld + d] * inv_norm;
        svt[r * LD + d] = v[kbase + rr * ld + d];
    }
    for (int i = threadIdx.x; i < V; i += 256) {
        ss[i * 3] = stat[(sb + i) * 2]; ss[i * 3 + 1] = 1.f / stat[(sb + i) * 2 + 1]; ss[i * 3 + 2] = dvec[sb + i];
    }
    __syncthreads();
    const float keep_scale = 1.f / (1.f - pdrop);
    const int lr = threadIdx.x / G, row = tile * RPB + lr, gp = threadIdx.x % G;
    const bool live = row < V;
    const int j = live ? row : V - 1;
    float kr[DH], vr[DH], ak[DH], av[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) { kr[d] = skt[lr * LD + d]; vr[d] = svt[lr * LD + d]; ak[d] = 0.f; av[d] = 0.f; }
    for (int i = gp; i < V; i += G) {
        float sdot = 0.f, gv = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) { sdot += sq[i * LD + d] * kr[d]; gv += sg[i * LD + d] * vr[d]; }
        const float a = expf(sdot - ss[i * 3]) * ss[i * 3 + 1];
        float at = a;
        if (pdrop > 0.f) {
            bool keep = pdf_uniform(seed, (unsigned long long)(sb + i) * V + j) >= pdrop;
            at = keep ? a * keep_scale : 0.f;
            gv = keep ? gv * keep_scale : 0.f;
        }
        const float ds = a * (gv - ss[i * 3 + 2]) * inv_norm;
#pragma unroll
        for (int d = 0; d < DH; ++d) { av[d] += at * sg[i * LD + d]; ak[d] += ds * sq[i * LD + d]; }
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        const float kd = group_sum<G>(ak[d]), vd = group_sum<G>(av[d]);
        if (live && gp == (d % G)) {
            dk[(long)bkv * V * lddk + (long)j * lddk + h * DH + d] = kd;
            dv[(long)bkv * V * lddk + (long)j * lddk + h * DH + d] = vd;
        }
    }
}

#define ATT_DISPATCH(KERNEL_CALL)                                                                          \
    switch (dh * 100 + G) {                                                                                \
        ATT_CASE(4, 1) ATT_CASE(4, 2) ATT_CASE(4, 4)                                                       \
        ATT_CASE(16, 1) ATT_CASE(16, 2) ATT_CASE(16, 4) ATT_CASE(16, 8) ATT_CASE(16, 16)                   \
        ATT_CASE(32, 1) ATT_CASE(32, 2) ATT_CASE(32, 4) ATT_CASE(32, 8) ATT_CASE(32, 16)                   \
        ATT_CASE(64, 1) ATT_CASE(64, 2) ATT_CASE(64, 4) ATT_CASE(64, 8) ATT_CASE(64, 16)                   \
        default: return PDF_E_BADARG;                                                                      \
    }

PDF_API int pdf_attn_fwd(const float* q, const float* k, const float* v, int ld, int B, int V, int H, int dh, int kv_shift,
                         float pdrop, unsigned long long seed, const unsigned long long* step, float* out, int ldo, float* stat, hipStream_t s) {
    if (V <= 0) return PDF_E_BADARG;
    const int G = attn_group(V, dh), NT = cdiv(V, 256 / G);
    const bool keys = dh <= 32 && G > 1;                   // which split (see the *_keys kernels)
    size_t smem = keys ? (size_t)(2 * V + 256 / G) * (dh + 1) * sizeof(float) : (size_t)2 * V * dh * sizeof(float);
    if (smem > 64 * 1024) return PDF_E_BADARG;
    float inv_norm = 1.f / sqrtf((float)dh);
    dim3 grid(B * H * NT);
#define ATT_CASE(D, GG) case D * 100 + GG: \
        if (keys && D <= 32) hipLaunchKernelGGL((attn_fwd_keys_kernel<(D <= 32 ? D : 4), GG>), grid, dim3(256), smem, s, q, k, v, ld, B, V, H, kv_shift, inv_norm, pdrop, seed, step, out, ldo, stat); \
        else hipLaunchKernelGGL((attn_fwd_kernel<D, GG>), grid, dim3(256), smem, s, q, k, v, ld, B, V, H, kv_shift, inv_norm, pdrop, seed, step, out, ldo, stat); \
        break;
    ATT_DISPATCH()
#undef ATT_CASE
    PDF_LAUNCH_CHECK();
    return 0;
}

// backward, query side: dq_i = sum_j dS_ij k_j * inv_norm,  dS_ij = A_ij (dA_ij - D_i),
//   A_ij = exp(s_ij - m_i)/l_i,  dA_ij = mask_ij/(1-p) * (dO_i . v_j),  D_i = dO_i . O_i
template <int DH, int G>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld,
                                                         const float* __restrict__ o, const float* __restrict__ dout, int ldo,
                                                         const float* __restrict__ stat, int B, int V, int H, int kv_shift, float inv_norm, float pdrop, unsigned long long seed, const unsigned long long* __restrict__ step,
                                                         float* __restrict__ dq, int lddq, float* __restrict__ dvec /*[B][H][V] D_i*/) {
    constexpr int SL = DH / G, RPB = 256 / G;
    if (step != nullptr) seed += step[0] * 0x9E3779B97F4A7C15ull;
    extern __shared__ float sm[];
    float* sk = sm;
    float* sv = sm + V * DH;
    const int NT = (V + RPB - 1) / RPB;
    const int tile = blockIdx.x % NT, bh = blockIdx.x / NT;
    const int b = bh / H, h = bh % H;
    const long base = (long)b * V * ld + h * DH;
    const long kbase = (long)((b + kv_shift) % B) * V * ld + h * DH;
    for (int i = threadIdx.x; i < V * DH; i += 256) {
        int r = i / DH, d = i - r * DH;
        sk[i] = k[kbase + (long)r * ld + d];
        sv[i] = v[kbase + (long)r * ld + d];
    }
    __syncthreads();
    const float keep_scale = 1.f / (1.f - pdrop);
    const int row = tile * RPB + threadIdx.x / G, d0 = (threadIdx.x % G) * SL;
    const bool live = row < V;
    const int i = live ? row : V - 1;
    float qr[SL], go[SL], acc[SL];
    float D = 0.f;
#pragma unroll
    for (int d = 0; d < SL; ++d) {
        qr[d] = q[base + (long)i * ld + d0 + d] * inv_norm;
        go[d] = dout[(long)b * V * ldo + (long)i * ldo + h * DH + d0 + d];
        D += go[d] * o[(long)b * V * ldo + (long)i * ldo + h * DH + d0 + d];
        acc[d] = 0.f;
    }
    D = group_sum<G>(D);
    const long si = ((long)b * H + h) * V + i;
    const float m = stat[si * 2], il = 1.f / stat[si * 2 + 1];
    if (live && d0 == 0) dvec[si] = D;
    const unsigned long long rowid = (unsigned long long)si * V;
    for (int j = 0; j < V; ++j) {
        float sdot = 0.f, gv = 0.f;
#pragma unroll
        for (int d = 0; d < SL; ++d) { sdot += qr[d] * sk[j * DH + d0 + d]; gv += go[d] * sv[j * DH + d0 + d]; }
        sdot = group_sum<G>(sdot);
        gv = group_sum<G>(gv);
        const float a = expf(sdot - m) * il;
        if (pdrop > 0.f) gv = pdf_uniform(seed, rowid + j) >= pdrop ? gv * keep_scale : 0.f;
        const float ds = a * (gv - D) * inv_norm;
#pragma unroll
        for (int d = 0; d < SL; ++d) acc[d] += ds * sk[j * DH + d0 + d];
    }
    if (live) {
#pragma unroll
        for (int d = 0; d < SL; ++d) dq[(long)b * V * lddq + (long)i * lddq + h * DH + d0 + d] = acc[d];
    }
}

// backward, key side: dv_j = sum_i Atilde_ij dO_i ; dk_j = sum_i dS_ij q_i * inv_norm
template <int DH, int G>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld,
                                                          const float* __restrict__ dout, int ldo, const float* __restrict__ stat,
                                                          const float* __restrict__ dvec, int B, int V, int H, int kv_shift, float inv_norm, float pdrop, unsigned long long seed, const unsigned long long* __restrict__ step,
                                                          float* __restrict__ dk, float* __restrict__ dv, int lddk) {
    constexpr int SL = DH / G, RPB = 256 / G;
    if (step != nullptr) seed += step[0] * 0x9E3779B97F4A7C15ull;
    extern __shared__ float sm[];
    float* sq = sm;              // [V][DH]
    float* sg = sm + V * DH;     // dO
    float* ss = sm + 2 * V * DH; // [V][3]: max, 1 / sumexp, D
    const int NT = (V + RPB - 1) / RPB;
    const int tile = blockIdx.x % NT, bh = blockIdx.x / NT;
    const int b = bh / H, h = bh % H;
    const long base = (long)b * V * ld + h * DH;
    const int bkv = (b + kv_shift) % B;
    const long kbase = (long)bkv * V * ld + h * DH;
    const long sb = ((long)b * H + h) * V;
    for (int i = threadIdx.x; i < V * DH; i += 256) {
        int r = i / DH, d = i - r * DH;
        sq[i] = q[base + (long)r * ld + d];
        sg[i] = dout[(long)b * V * ldo + (long)r * ldo + h * DH + d];
    }
    for (int i = threadIdx.x; i < V; i += 256) {
        ss[i * 3] = stat[(sb + i) * 2]; ss[i * 3 + 1] = 1.f / stat[(sb + i) * 2 + 1]; ss[i * 3 + 2] = dvec[sb + i];
    }
    __syncthreads();
    const float keep_scale = 1.f / (1.f - pdrop);
    const int row = tile * RPB + threadIdx.x / G, d0 = (threadIdx.x % G) * SL;
    const bool live = row < V;
    const int j = live ? row : V - 1;
    float kr[SL], vr[SL], ak[SL], av[SL];
#pragma unroll
    for (int d = 0; d < SL; ++d) { kr[d] = k[kbase + (long)j * ld + d0 + d] * inv_norm; vr[d] = v[kbase + (long)j * ld + d0 + d]; ak[d] = 0.f; av[d] = 0.f; }
    for (int i = 0; i < V; ++i) {
        float sdot = 0.f, gv = 0.f;
#pragma unroll
        for (int d = 0; d < SL; ++d) { sdot += sq[i * DH + d0 + d] * kr[d]; gv += sg[i * DH + d0 + d] * vr[d]; }
        sdot = group_sum<G>(sdot);
        gv = group_sum<G>(gv);
        const float a = expf(sdot - ss[i * 3]) * ss[i * 3 + 1];
        float at = a;
        if (pdrop > 0.f) {
            bool keep = pdf_uniform(seed, (unsigned long long)(sb + i) * V + j) >= pdrop;
            at = keep ? a * keep_scale : 0.f;
            gv = keep ? gv * keep_scale : 0.f;
        }
        const float ds = a * (gv - ss[i * 3 + 2]) * inv_norm;
#pragma unroll
        for (int d = 0; d < SL; ++d) { av[d] += at * sg[i * DH + d0 + d]; ak[d] += ds * sq[i * DH + d0 + d]; }
    }
    if (live) {
#pragma unroll
        for (int d = 0; d < SL; ++d) {
            dk[(long)bkv * V * lddk + (long)j * lddk + h * DH + d0 + d] = ak[d];
            dv[(long)bkv * V * lddk + (long)j * lddk + h * DH + d0 + d] = av[d];
        }
    }
}

PDF_API int pdf_attn_bwd(const float* q, const float* k, const float* v, int ld, const float* o, const float* dout, int ldo,
                         const float* stat, int B, int V, int H, int dh, int kv_shift, float pdrop, unsigned long long seed, const unsigned long long* step,
                         float* dq, float* dk, float* dv, int lddq, float* dvec, hipStream_t s) {
    if (V <= 0) return PDF_E_BADARG;
    const int G = attn_group(V, dh), NT = cdiv(V, 256 / G), RPB = 256 / G;
    const bool keys = dh <= 32 && G > 1;
    const size_t smem_q = keys ? (size_t)(2 * V + 3 * RPB) * (dh + 1) * sizeof(float) : (size_t)2 * V * dh * sizeof(float);
    const size_t smem_kv = keys ? ((size_t)(2 * V + 2 * RPB) * (dh + 1) + 3 * V) * sizeof(float) : (size_t)(2 * V * dh + 3 * V) * sizeof(float);
    if (smem_q > 64 * 1024 || smem_kv > 64 * 1024) return PDF_E_BADARG;
    float inv_norm = 1.f / sqrtf((float)dh);
    dim3 grid(B * H * NT);
#define ATT_CASE(D, GG) case D * 100 + GG: \
        if (keys && D <= 32) { \
            hipLaunchKernelGGL((attn_bwd_q_keys_kernel<(D <= 32 ? D : 4), GG>), grid, dim3(256), smem_q, s, q, k, v, ld, o, dout, ldo, stat, B, V, H, kv_shift, inv_norm, pdrop, seed, step, dq, lddq, dvec); \
            hipLaunchKernelGGL((attn_bwd_kv_keys_kernel<(D <= 32 ? D : 4), GG>), grid, dim3(256), smem_kv, s, q, k, v, ld, dout, ldo, stat, dvec, B, V, H, kv_shift, inv_norm, pdrop, seed, step, dk, dv, lddq); \
        } else { \
            hipLaunchKernelGGL((attn_bwd_q_kernel<D, GG>), grid, dim3(256), smem_q, s, q, k, v, ld, o, dout, ldo, stat, B, V, H, kv_shift, inv_norm, pdrop, seed, step, dq, lddq, dvec); \
            hipLaunchKernelGGL((attn_bwd_kv_kernel<D, GG>), grid, dim3(256), smem_kv, s, q, k, v, ld, dout, ldo, stat, dvec, B, V, H, kv_shift, inv_norm, pdrop, seed, step, dk, dv, lddq); \
        } break;
    ATT_DISPATCH()
#undef ATT_CASE
    PDF_LAUNCH_CHECK();
    return 0;
}
