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
// (self_attn.py:63-76, inter_attn.py:82-105).  One block per (b, h); one query row per thread.
// Dropout on the attention matrix (dropout1) uses the stateless mask of common.h.
// kv_shift: queries of sample b attend to keys / values of sample (b + kv_shift) % B -- with the two hands stacked along
// the batch axis, kv_shift = B/2 is the cross-hand attention of inter_attn.py:82-105 in one launch.
template <int DH>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld,
                                                       int V, int H, int kv_shift, float inv_norm, float pdrop, unsigned long long seed, const unsigned long long* __restrict__ step,
                                                       float* __restrict__ out, int ldo, float* __restrict__ stat /*[B][H][V][2] = max, sumexp*/) {
    if (step != nullptr) seed += step[0] * 0x9E3779B97F4A7C15ull;
    extern __shared__ float sm[];
    float* sk = sm;              // [V][DH]
    float* sv = sm + V * DH;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const long base = (long)b * V * ld + h * DH;
    const long kbase = (long)((b + kv_shift) % (int)(gridDim.x / H)) * V * ld + h * DH;
    for (int i = threadIdx.x; i < V * DH; i += 256) {
        int r = i / DH, d = i - r * DH;
        sk[i] = k[kbase + (long)r * ld + d];
        sv[i] = v[kbase + (long)r * ld + d];
    }
    __syncthreads();
    const float keep_scale = 1.f / (1.f - pdrop);
    for (int i = threadIdx.x; i < V; i += 256) {
        float qr[DH], o[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) { qr[d] = q[base + (long)i * ld + d]; o[d] = 0.f; }
        float m = -INFINITY;
        for (int j = 0; j < V; ++j) {
            float sdot = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) sdot += qr[d] * sk[j * DH + d];
            m = fmaxf(m, sdot * inv_norm);
        }
        float l = 0.f;
        const unsigned long long rowid = ((unsigned long long)(b * H + h) * V + i) * V;
        for (int j = 0; j < V; ++j) {
            float sdot = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) sdot += qr[d] * sk[j * DH + d];
            float p = expf(sdot * inv_norm - m);
            l += p;
            if (pdrop > 0.f) p = pdf_uniform(seed, rowid + j) >= pdrop ? p * keep_scale : 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) o[d] += p * sv[j * DH + d];
        }
        const float il = 1.f / l;
#pragma unroll
        for (int d = 0; d < DH; ++d) out[(long)b * V * ldo + (long)i * ldo + h * DH + d] = o[d] * il;
        stat[(((long)b * H + h) * V + i) * 2 + 0] = m;
        stat[(((long)b * H + h) * V + i) * 2 + 1] = l;
    }
}

PDF_API int pdf_attn_fwd(const float* q, const float* k, const float* v, int ld, int B, int V, int H, int dh, int kv_shift,
                         float pdrop, unsigned long long seed, const unsigned long long* step, float* out, int ldo, float* stat, hipStream_t s) {
    size_t smem = (size_t)2 * V * dh * sizeof(float);
    if (smem > 64 * 1024) return PDF_E_BADARG;
    float inv_norm = 1.f / sqrtf((float)dh);
    dim3 grid(B * H);
#define ATT_CASE(D) case D: hipLaunchKernelGGL(attn_fwd_kernel<D>, grid, dim3(256), smem, s, q, k, v, ld, V, H, kv_shift, inv_norm, pdrop, seed, step, out, ldo, stat); break;
    switch (dh) { ATT_CASE(4) ATT_CASE(16) ATT_CASE(32) ATT_CASE(64) default: return PDF_E_BADARG; }
#undef ATT_CASE
    PDF_LAUNCH_CHECK();
    return 0;
}

// backward, query side: dq_i = sum_j dS_ij k_j * inv_norm,  dS_ij = A_ij (dA_ij - D_i),
//   A_ij = exp(s_ij - m_i)/l_i,  dA_ij = mask_ij/(1-p) * (dO_i . v_j),  D_i = dO_i . O_i
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld,
                                                         const float* __restrict__ o, const float* __restrict__ dout, int ldo,
                                                         const float* __restrict__ stat, int V, int H, int kv_shift, float inv_norm, float pdrop, unsigned long long seed, const unsigned long long* __restrict__ step,
                                                         float* __restrict__ dq, int lddq, float* __restrict__ dvec /*[B][H][V] D_i*/) {
    if (step != nullptr) seed += step[0] * 0x9E3779B97F4A7C15ull;
    extern __shared__ float sm[];
    float* sk = sm;
    float* sv = sm + V * DH;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const long base = (long)b * V * ld + h * DH;
    const long kbase = (long)((b + kv_shift) % (int)(gridDim.x / H)) * V * ld + h * DH;
    for (int i = threadIdx.x; i < V * DH; i += 256) {
        int r = i / DH, d = i - r * DH;
        sk[i] = k[kbase + (long)r * ld + d];
        sv[i] = v[kbase + (long)r * ld + d];
    }
    __syncthreads();
    const float keep_scale = 1.f / (1.f - pdrop);
    for (int i = threadIdx.x; i < V; i += 256) {
        float qr[DH], go[DH], acc[DH];
        float D = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            qr[d] = q[base + (long)i * ld + d];
            go[d] = dout[(long)b * V * ldo + (long)i * ldo + h * DH + d];
            D += go[d] * o[(long)b * V * ldo + (long)i * ldo + h * DH + d];
            acc[d] = 0.f;
        }
        const long si = ((long)b * H + h) * V + i;
        const float m = stat[si * 2], il = 1.f / stat[si * 2 + 1];
        dvec[si] = D;
        const unsigned long long rowid = (unsigned long long)si * V;
        for (int j = 0; j < V; ++j) {
            float sdot = 0.f, gv = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) { sdot += qr[d] * sk[j * DH + d]; gv += go[d] * sv[j * DH + d]; }
            float a = expf(sdot * inv_norm - m) * il;
            if (pdrop > 0.f) gv = pdf_uniform(seed, rowid + j) >= pdrop ? gv * keep_scale : 0.f;
            float ds = a * (gv - D) * inv_norm;
#pragma unroll
            for (int d = 0; d < DH; ++d) acc[d] += ds * sk[j * DH + d];
        }
#pragma unroll
        for (int d = 0; d < DH; ++d) dq[(long)b * V * lddq + (long)i * lddq + h * DH + d] = acc[d];
    }
}

// backward, key side: dv_j = sum_i Atilde_ij dO_i ; dk_j = sum_i dS_ij q_i * inv_norm
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld,
                                                          const float* __restrict__ dout, int ldo, const float* __restrict__ stat,
                                                          const float* __restrict__ dvec, int V, int H, int kv_shift, float inv_norm, float pdrop, unsigned long long seed, const unsigned long long* __restrict__ step,
                                                          float* __restrict__ dk, float* __restrict__ dv, int lddk) {
    if (step != nullptr) seed += step[0] * 0x9E3779B97F4A7C15ull;
    extern __shared__ float sm[];
    float* sq = sm;              // [V][DH]
    float* sg = sm + V * DH;     // dO
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const long base = (long)b * V * ld + h * DH;
    const int bkv = (b + kv_shift) % (int)(gridDim.x / H);
    const long kbase = (long)bkv * V * ld + h * DH;
    for (int i = threadIdx.x; i < V * DH; i += 256) {
        int r = i / DH, d = i - r * DH;
        sq[i] = q[base + (long)r * ld + d];
        sg[i] = dout[(long)b * V * ldo + (long)r * ldo + h * DH + d];
    }
    __syncthreads();
    const float keep_scale = 1.f / (1.f - pdrop);
    const long sb = ((long)b * H + h) * V;
    for (int j = threadIdx.x; j < V; j += 256) {
        float kr[DH], vr[DH], ak[DH], av[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) { kr[d] = k[kbase + (long)j * ld + d]; vr[d] = v[kbase + (long)j * ld + d]; ak[d] = 0.f; av[d] = 0.f; }
        for (int i = 0; i < V; ++i) {
            float sdot = 0.f, gv = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) { sdot += sq[i * DH + d] * kr[d]; gv += sg[i * DH + d] * vr[d]; }
            const float a = expf(sdot * inv_norm - stat[(sb + i) * 2]) / stat[(sb + i) * 2 + 1];
            float at = a;
            if (pdrop > 0.f) {
                bool keep = pdf_uniform(seed, (unsigned long long)(sb + i) * V + j) >= pdrop;
                at = keep ? a * keep_scale : 0.f;
                gv = keep ? gv * keep_scale : 0.f;
            }
            const float ds = a * (gv - dvec[sb + i]) * inv_norm;
#pragma unroll
            for (int d = 0; d < DH; ++d) { av[d] += at * sg[i * DH + d]; ak[d] += ds * sq[i * DH + d]; }
        }
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            dk[(long)bkv * V * lddk + (long)j * lddk + h * DH + d] = ak[d];
            dv[(long)bkv * V * lddk + (long)j * lddk + h * DH + d] = av[d];
        }
    }
}

PDF_API int pdf_attn_bwd(const float* q, const float* k, const float* v, int ld, const float* o, const float* dout, int ldo,
                         const float* stat, int B, int V, int H, int dh, int kv_shift, float pdrop, unsigned long long seed, const unsigned long long* step,
                         float* dq, float* dk, float* dv, int lddq, float* dvec, hipStream_t s) {
    size_t smem = (size_t)2 * V * dh * sizeof(float);
    if (smem > 64 * 1024) return PDF_E_BADARG;
    float inv_norm = 1.f / sqrtf((float)dh);
    dim3 grid(B * H);
#define ATT_CASE(D) case D: \
        hipLaunchKernelGGL(attn_bwd_q_kernel<D>, grid, dim3(256), smem, s, q, k, v, ld, o, dout, ldo, stat, V, H, kv_shift, inv_norm, pdrop, seed, step, dq, lddq, dvec); \
        hipLaunchKernelGGL(attn_bwd_kv_kernel<D>, grid, dim3(256), smem, s, q, k, v, ld, dout, ldo, stat, dvec, V, H, kv_shift, inv_norm, pdrop, seed, step, dk, dv, lddq); break;
    switch (dh) { ATT_CASE(4) ATT_CASE(16) ATT_CASE(32) ATT_CASE(64) default: return PDF_E_BADARG; }
#undef ATT_CASE
    PDF_LAUNCH_CHECK();
    return 0;
}
