// Normalisation kernels on row-major [rows][channels] (NHWC) activations: BatchNorm (train/eval,
// fused ReLU / residual), LayerNorm, L2Norm.  All HBM-bound: each is a column- or row-reduction plus
// one streaming pass; channels are the contiguous axis so every access is coalesced.
#include "common.h"
#include <initializer_list>
#include <cstdlib>

// ---------------------------------------------------------------------------------------------
// BatchNorm2d / BatchNorm1d over rows.  Statistics are accumulated as shifted sums
// sum(x - x0), sum((x - x0)^2) (x0 = first row) in fp32 per thread, combined in fp64.
#define BN_CT 64   // channels per block
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int ldx, int C, long R, long rows_per_chunk,
                                                         float* __restrict__ part /*[chunks][C][2]*/) {
    __shared__ float s1[4][BN_CT], s2[4][BN_CT];
    const int tx = threadIdx.x & (BN_CT - 1), ty = threadIdx.x / BN_CT;
    const int c = blockIdx.x * BN_CT + tx;
    const long r0 = blockIdx.y * rows_per_chunk;
    const long r1 = min(R, r0 + rows_per_chunk);
    float a = 0.f, b = 0.f;
    if (c < C) {
        const float sh = x[c];
        for (long r = r0 + ty; r < r1; r += 4) {
            float v = x[r * ldx + c] - sh;
            a += v; b += v * v;
        }
    }
    s1[ty][tx] = a; s2[ty][tx] = b;
    __syncthreads();
    if (ty == 0 && c < C) {
        a = s1[0][tx] + s1[1][tx] + s1[2][tx] + s1[3][tx];
        b = s2[0][tx] + s2[1][tx] + s2[2][tx] + s2[3][tx];
        part[((long)blockIdx.y * C + c) * 2 + 0] = a;
        part[((long)blockIdx.y * C + c) * 2 + 1] = b;
    }
}


// sum the per-chunk partials part[chunk][C][2] for channel c: 64 channels x 16 chunk-lanes per block
// (a serial 512-iteration loop per channel was latency-bound: ~50 us for a kernel that moves a few KB)
#define FIN_TX 64
#define FIN_TY 16
__device__ __forceinline__ void reduce_chunks(const float* __restrict__ part, int chunks, int C, int c, int ty,
                                              double (&s1)[FIN_TY][FIN_TX], double (&s2)[FIN_TY][FIN_TX], double& a, double& b) {
    a = 0.0; b = 0.0;
    if (c < C) {
        const float2* p2 = reinterpret_cast<const float2*>(part) + c;
#pragma unroll 4
        for (int k = ty; k < chunks; k += FIN_TY) { const float2 v = p2[(long)k * C]; a += v.x; b += v.y; }
    }
    const int tx = threadIdx.x;
    s1[ty][tx] = a; s2[ty][tx] = b;
    __syncthreads();
    a = 0.0; b = 0.0;
    for (int j = 0; j < FIN_TY; ++j) { a += s1[j][tx]; b += s2[j][tx]; }
}

static bool bn_inlaunch(int C, long R) {
    static long maxb = -2;
    if (maxb == -2) { const char* e = getenv("PDF_BN_INLAUNCH_MAXMB"); maxb = e ? atol(e) : 0; }
    return maxb < 0 || (long)C * R * 4 <= maxb * (1L << 20);
}

// ---- finalize inside the partial kernels (pdf_last_block_arrives): the last block of a 64-channel tile sums the chunk
// partials part[chunk][C][2] in a fixed order -- 64 channels x 4 chunk lanes, doubles -- and writes the per-channel results.
// Saves one launch per BatchNorm call (164 per step) but every block then waits for its write-through stores and the ticket
// before it retires, and the tile's last block adds the serial sum: measured on MI355X the partial kernels run 1.3-1.4 ms
// per step longer than the finalize launches they replace (fp32 B=32 393 vs 399 img/s, bf16 652 vs 664), so it is opt-in:
// PDF_BN_INLAUNCH_MAXMB=<n> enables it for tensors up to n MiB (-1: always).
__device__ __forceinline__ bool tile_sums(const float* __restrict__ part, int chunks, int C, int c_tile, double* red /*[2][4][64]*/, double& a, double& b) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = c_tile * BN_CT + tx;
    a = 0.0; b = 0.0;
    if (c < C) {
        const float2* p2 = reinterpret_cast<const float2*>(part) + c;
        for (int k = ty; k < chunks; k += 4) { const float2 v = p2[(long)k * C]; a += v.x; b += v.y; }
    }
    red[ty * 64 + tx] = a; red[256 + ty * 64 + tx] = b;
    __syncthreads();
    a = (red[tx] + red[64 + tx]) + (red[128 + tx] + red[192 + tx]);
    b = (red[256 + tx] + red[320 + tx]) + (red[384 + tx] + red[448 + tx]);
    return ty == 0 && c < C;
}
struct BnFin {                                              // forward finalize arguments
    const float* gamma; const float* beta; float* running_mean; float* running_var; float momentum, eps;
    float* save_mean; float* save_rstd; float* scale; float* shift; int* counters;
    float* shift0;                                          // X16 input: row 0 of x as floats (written by the partial kernel)
};
__device__ __forceinline__ void bn_finalize_tile(const float* __restrict__ part, int chunks, const float* __restrict__ x, int C, long R, const BnFin& f,
                                                 int c_tile, double* red) {
    double a, b;
    if (!tile_sums(part, chunks, C, c_tile, red, a, b)) return;
    const int c = c_tile * BN_CT + (threadIdx.x & 63);
    const double n = (double)R;
    const double dm = a / n;
    double var = b / n - dm * dm;
    if (var < 0.0) var = 0.0;
    const double mean = dm + (double)x[c];
    const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
    f.save_mean[c] = (float)mean;
    f.save_rstd[c] = rstd;
    if (f.running_mean != nullptr) {
        f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)mean;
        const double unb = R > 1 ? var * n / (n - 1.0) : var;
        f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unb;
    }
    const float sc = f.gamma[c] * rstd;
    f.scale[c] = sc;
    f.shift[c] = f.beta[c] - (float)mean * sc;
}
struct BnBwdFin { const float* gamma; const float* rstd; float* dgamma; float* dbeta; int accumulate; float* coef; int* counters; };
__device__ __forceinline__ void bn_bwd_finalize_tile(const float* __restrict__ part, int chunks, int C, long R, const BnBwdFin& f, int c_tile, double* red) {
    double a, b;
    if (!tile_sums(part, chunks, C, c_tile, red, a, b)) return;
    const int c = c_tile * BN_CT + (threadIdx.x & 63);
    if (f.accumulate) { f.dbeta[c] += (float)a; f.dgamma[c] += (float)b; }
    else { f.dbeta[c] = (float)a; f.dgamma[c] = (float)b; }
    f.coef[c] = f.gamma[c] * f.rstd[c];
    f.coef[C + c] = (float)(a / (double)R);
    f.coef[2 * C + c] = (float)(b / (double)R);
}

__global__ __launch_bounds__(FIN_TX * FIN_TY) void bn_finalize_kernel(const float* __restrict__ part, int chunks, const float* __restrict__ x, int C, long R,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
                                   float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                   float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double s1[FIN_TY][FIN_TX], s2[FIN_TY][FIN_TX];
    const int c = blockIdx.x * FIN_TX + threadIdx.x;
    double a, b;
    reduce_chunks(part, chunks, C, c, threadIdx.y, s1, s2, a, b);
    if (c >= C || threadIdx.y != 0) return;
    const double n = (double)R;
    const double dm = a / n;
    double var = b / n - dm * dm;
    if (var < 0.0) var = 0.0;
    const double mean = dm + (double)x[c];
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = (float)mean;
    save_rstd[c] = rstd;
    if (running_mean != nullptr) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        const double unb = R > 1 ? var * n / (n - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
}

// Statistics that arrive as per-row-block (mean, M2) pairs out of the producing GEMM's epilogue (IGemm::stat, gemm_common.h):
// part[(t * C + c) * 2 + {0, 1}], block t covers rows [t rpt, min(R, (t + 1) rpt)).  Combined in fp64 in block order
// (deterministic): mean = sum n_t mean_t / R, M2 = sum (M2_t + n_t mean_t^2) - R mean^2.
#define FT_C 16                                              // channels per block
#define FT_L 64                                              // tile lanes per channel
__global__ __launch_bounds__(FT_C * FT_L) void bn_finalize_tiles_kernel(const float* __restrict__ part, int tiles, long rpt, int C, long R,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
                                   float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                   float* __restrict__ scale, float* __restrict__ shift) {
    // 16 channels x 64 tile lanes per block (C / 16 blocks): with 64 channels per block a 64-channel layer was ONE block walking
    // 1,024 row blocks 16 at a time -- 15 us per call of pure latency, 78 calls per step
    __shared__ double s1[FT_L][FT_C], s2[FT_L][FT_C];
    const int tx = threadIdx.x & (FT_C - 1), ty = threadIdx.x / FT_C;
    const int c = blockIdx.x * FT_C + tx;
    double a = 0.0, b = 0.0;
    if (c < C) {
        const float2* p2 = reinterpret_cast<const float2*>(part) + c;
#pragma unroll 4
        for (int t = ty; t < tiles; t += FT_L) {
            const float2 v = p2[(long)t * C];
            const double n = (double)min(rpt, R - (long)t * rpt), m = (double)v.x;
            a += n * m; b += (double)v.y + n * m * m;
        }
    }
    s1[ty][tx] = a; s2[ty][tx] = b;
    __syncthreads();
    for (int h = FT_L / 2; h >= 1; h >>= 1) {                // fixed tree: deterministic
        if (ty < h) { s1[ty][tx] += s1[ty + h][tx]; s2[ty][tx] += s2[ty + h][tx]; }
        __syncthreads();
    }
    if (c >= C || ty != 0) return;
    a = s1[0][tx]; b = s2[0][tx];
    const double n = (double)R;
    const double mean = a / n;
    double var = b / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = (float)mean;
    save_rstd[c] = rstd;
    if (running_mean != nullptr) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        const double unb = R > 1 ? var * n / (n - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
}
struct TileStats { const float* part; long tiles, rows; };

__global__ void bn_eval_coeff_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                     float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// y = act(scale*x + shift (+ res))
__global__ __launch_bounds__(256) void affine_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ res, int ldr,
                                                           float* __restrict__ y, int ldy, int C, long total, int relu) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        float v = fmaf(x[r * ldx + c], scale[c], shift[c]);
        if (res != nullptr) v += res[r * ldr + c];
        if (relu) v = fmaxf(v, 0.f);
        y[r * ldy + c] = v;
    }
}


// ---- float4 variants of the column reductions: 16 channel-quads (64 channels) x 16 row-lanes per block, 4 rows in
// flight per thread.  The scalar forms above moved 1.1-1.6 TB/s (one 4-byte load in flight per thread).
#define V4_TX 16
#define V4_TY 16
// rows a block of the float4 kernels walks: rows_per_chunk > 0: its own contiguous chunk [blockIdx.y * rpc, ...); < 0: the chunks taken
// from the END of the tensor (bn_second_pass_reverse); == 0: INTERLEAVED -- block y takes the 16-row groups y, y + gridDim.y, ... so the
// whole grid reads one moving window of the tensor instead of gridDim.y separate streams (bn_interleave)
__device__ __forceinline__ void v4_rows(long rows_per_chunk, long R, int ty, long& rs, long& re, long& rstep) {
    if (rows_per_chunk == 0) { rs = (long)blockIdx.y * V4_TY + ty; re = R; rstep = (long)gridDim.y * V4_TY; return; }
    const long rpc = rows_per_chunk < 0 ? -rows_per_chunk : rows_per_chunk;
    const long r0 = (rows_per_chunk < 0 ? (long)(gridDim.y - 1 - blockIdx.y) : (long)blockIdx.y) * rpc;
    rs = r0 + ty; re = min(R, r0 + rpc); rstep = V4_TY;
}
__device__ __forceinline__ void v4_block_reduce(float4 a, float4 b, float4 (&sa)[V4_TY][V4_TX], float4 (&sb)[V4_TY][V4_TX],
                                                float* __restrict__ part, int C, int c0, bool two, bool wt = false) {
    const int tx = threadIdx.x & (V4_TX - 1), ty = threadIdx.x / V4_TX;
    sa[ty][tx] = a; sb[ty][tx] = b;
    __syncthreads();
    if (ty == 0 && c0 < C) {
        float4 x = sa[0][tx], y = sb[0][tx];
        for (int j = 1; j < V4_TY; ++j) {
            float4 u = sa[j][tx], v = sb[j][tx];
            x.x += u.x; x.y += u.y; x.z += u.z; x.w += u.w;
            y.x += v.x; y.y += v.y; y.z += v.z; y.w += v.w;
        }
        float* o = part + ((long)blockIdx.y * C + c0) * 2;
        const float v[8] = {x.x, two ? y.x : 0.f, x.y, two ? y.y : 0.f, x.z, two ? y.z : 0.f, x.w, two ? y.w : 0.f};
        if (wt) {                                            // consumed by another block of this launch: write-through stores
#pragma unroll
            for (int e = 0; e < 8; ++e) pdf_store_wt(o + e, v[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = v[e];
        }
    }
}


// bf16 storage mode: the BatchNorm input x (a conv output nobody else reads) may come as bf16 (X16): 4 channels = one 8-byte load
template <bool X16>
__device__ __forceinline__ float4 ld_x4(const float* __restrict__ x, long idx) {
    if constexpr (X16) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(x) + idx);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    } else {
        return *reinterpret_cast<const float4*>(x + idx);
    }
}

template <bool X16 = false>
__global__ __launch_bounds__(256) void bn_partial_v4_kernel(const float* __restrict__ x, int ldx, int C, long R, long rows_per_chunk,
                                                            float* __restrict__ part, const BnFin fin) {
    __shared__ float4 sa[V4_TY][V4_TX], sb[V4_TY][V4_TX];
    const int tx = threadIdx.x & (V4_TX - 1), ty = threadIdx.x / V4_TX;
    const int c0 = blockIdx.x * BN_CT + tx * 4;
    long rs, re, rstep;
    v4_rows(rows_per_chunk, R, ty, rs, re, rstep);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (c0 < C) {
        const float4 sh = ld_x4<X16>(x, c0);
        if (X16 && blockIdx.y == 0 && ty == 0) *reinterpret_cast<float4*>(fin.shift0 + c0) = sh;      // (the finaliser's `x[c]`: row 0 as floats)
#pragma unroll 4
        for (long r = rs; r < re; r += rstep) {
            float4 v = ld_x4<X16>(x, r * ldx + c0);
            v.x -= sh.x; v.y -= sh.y; v.z -= sh.z; v.w -= sh.w;
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            b.x += v.x * v.x; b.y += v.y * v.y; b.z += v.z * v.z; b.w += v.w * v.w;
        }
    }
    v4_block_reduce(a, b, sa, sb, part, C, c0, true, fin.counters != nullptr);
    if (fin.counters != nullptr) {
        __shared__ double red[512];
        __shared__ int flag;
        if (pdf_last_block_arrives(fin.counters + blockIdx.x, gridDim.y, &flag, false)) bn_finalize_tile(part, gridDim.y, X16 ? fin.shift0 : x, C, R, fin, blockIdx.x, red);
    }
}

// relu == 2: no residual went into the ReLU, so its mask is recomputed from x as fmaf(x, scale, shift) > 0 -- the exact
// expression of the forward -- and y is not read at all
template <bool X16 = false>
__global__ __launch_bounds__(256) void bn_bwd_partial_v4_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy, int relu,
                                                                const float* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int C, long R, long rows_per_chunk,
                                                                float* __restrict__ part, const BnBwdFin fin) {
    __shared__ float4 sa[V4_TY][V4_TX], sb[V4_TY][V4_TX];
    const long RT = R;
    const int tx = threadIdx.x & (V4_TX - 1), ty = threadIdx.x / V4_TX;
    const int c0 = blockIdx.x * BN_CT + tx * 4;
    long row_s, row_e, row_step;
    v4_rows(rows_per_chunk, R, ty, row_s, row_e, row_step);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (c0 < C) {
        const float4 m = *reinterpret_cast<const float4*>(mean + c0), rs = *reinterpret_cast<const float4*>(rstd + c0);
        float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
        if (relu == 2) { sc = *reinterpret_cast<const float4*>(scale + c0); sh = *reinterpret_cast<const float4*>(shift + c0); }
#pragma unroll 4
        for (long r = row_s; r < row_e; r += row_step) {
            float4 g = *reinterpret_cast<const float4*>(dy + r * lddy + c0);
            const float4 xv = ld_x4<X16>(x, r * ldx + c0);
            if (relu) {
                const float4 yv = relu == 2 ? make_float4(fmaf(xv.x, sc.x, sh.x), fmaf(xv.y, sc.y, sh.y), fmaf(xv.z, sc.z, sh.z), fmaf(xv.w, sc.w, sh.w))
                                            : *reinterpret_cast<const float4*>(y + r * ldy + c0);
                if (!(yv.x > 0.f)) g.x = 0.f;
                if (!(yv.y > 0.f)) g.y = 0.f;
                if (!(yv.z > 0.f)) g.z = 0.f;
                if (!(yv.w > 0.f)) g.w = 0.f;
            }
            a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
            b.x += g.x * (xv.x - m.x) * rs.x; b.y += g.y * (xv.y - m.y) * rs.y;
            b.z += g.z * (xv.z - m.z) * rs.z; b.w += g.w * (xv.w - m.w) * rs.w;
        }
    }
    v4_block_reduce(a, b, sa, sb, part, C, c0, true, fin.counters != nullptr);
    if (fin.counters != nullptr) {
        __shared__ double red[512];
        __shared__ int flag;
        if (pdf_last_block_arrives(fin.counters + blockIdx.x, gridDim.y, &flag, false)) bn_bwd_finalize_tile(part, gridDim.y, C, RT, fin, blockIdx.x, red);
    }
}

// (colsum kernels: blockIdx.z == 1 is group 1 of a paired call -- rows [R, 2R), partials after group 0's)
__global__ __launch_bounds__(256) void colsum_partial_v4_kernel(const float* __restrict__ g, int ldg, int C, long R, long rows_per_chunk,
                                                                float* __restrict__ part) {
    __shared__ float4 sa[V4_TY][V4_TX], sb[V4_TY][V4_TX];
    if (blockIdx.z) { g += R * ldg; part += (long)gridDim.y * C * 2; }
    const int tx = threadIdx.x & (V4_TX - 1), ty = threadIdx.x / V4_TX;
    const int c0 = blockIdx.x * BN_CT + tx * 4;
    const long r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c0 < C) {
#pragma unroll 4
        for (long r = r0 + ty; r < r1; r += V4_TY) {
            const float4 v = *reinterpret_cast<const float4*>(g + r * ldg + c0);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    v4_block_reduce(a, a, sa, sb, part, C, c0, false);
}

// float4 streaming passes on the same (16 channel-quads x 16 row-lanes) block shape: per-channel coefficients live in
// registers, no index division, 4 independent 16-byte loads per operand in flight per thread.
template <bool X16 = false>
__global__ __launch_bounds__(256) void affine_apply_v4_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ res, int ldr,
                                                              float* __restrict__ y, int ldy, int C, long R, long rows_per_chunk, int relu,
                                                              unsigned short* __restrict__ y16 = nullptr) {
    // y16 (bf16 mode, optional): the same values rounded to bf16 beside y, same leading dimension -- the shadow the bf16 GEMMs read
    const int tx = threadIdx.x & (V4_TX - 1), ty = threadIdx.x / V4_TX;
    const int c0 = blockIdx.x * BN_CT + tx * 4;
    if (c0 >= C) return;
    long row_s, row_e, row_step;
    v4_rows(rows_per_chunk, R, ty, row_s, row_e, row_step);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c0), sh = *reinterpret_cast<const float4*>(shift + c0);
#pragma unroll 4
    for (long r = row_s; r < row_e; r += row_step) {
        const float4 xv = ld_x4<X16>(x, r * ldx + c0);
        float4 v = make_float4(fmaf(xv.x, sc.x, sh.x), fmaf(xv.y, sc.y, sh.y), fmaf(xv.z, sc.z, sh.z), fmaf(xv.w, sc.w, sh.w));
        if (res != nullptr) {
            const float4 rv = *reinterpret_cast<const float4*>(res + r * ldr + c0);
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(y + r * ldy + c0) = v;
        if (y16 != nullptr) *reinterpret_cast<uint2*>(y16 + r * ldy + c0) = uint2{pdf_pk_bf16(v.x, v.y), pdf_pk_bf16(v.z, v.w)};
    }
}

template <bool X16 = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_v4_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy, int relu,
                                                              const float* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ coef,
                                                              const float* __restrict__ scale, const float* __restrict__ shift, int C, long R,
                                                              long rows_per_chunk, float* __restrict__ dx, int lddx, float* __restrict__ dres, int lddr,
                                                              unsigned short* __restrict__ dx16 = nullptr) {
    const int tx = threadIdx.x & (V4_TX - 1), ty = threadIdx.x / V4_TX;
    const int c0 = blockIdx.x * BN_CT + tx * 4;
    if (c0 >= C) return;
    long row_s, row_e, row_step;
    v4_rows(rows_per_chunk, R, ty, row_s, row_e, row_step);
    const float4 m = *reinterpret_cast<const float4*>(mean + c0), rs = *reinterpret_cast<const float4*>(rstd + c0);
    const float4 ka = *reinterpret_cast<const float4*>(coef + c0), k1 = *reinterpret_cast<const float4*>(coef + C + c0),
                 k2 = *reinterpret_cast<const float4*>(coef + 2 * C + c0);
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
    if (relu == 2) { sc = *reinterpret_cast<const float4*>(scale + c0); sh = *reinterpret_cast<const float4*>(shift + c0); }
#pragma unroll 4
    for (long r = row_s; r < row_e; r += row_step) {
        float4 g = *reinterpret_cast<const float4*>(dy + r * lddy + c0);
        const float4 xv = ld_x4<X16>(x, r * ldx + c0);
        if (relu) {
            const float4 yv = relu == 2 ? make_float4(fmaf(xv.x, sc.x, sh.x), fmaf(xv.y, sc.y, sh.y), fmaf(xv.z, sc.z, sh.z), fmaf(xv.w, sc.w, sh.w))
                                        : *reinterpret_cast<const float4*>(y + r * ldy + c0);
            if (!(yv.x > 0.f)) g.x = 0.f;
            if (!(yv.y > 0.f)) g.y = 0.f;
            if (!(yv.z > 0.f)) g.z = 0.f;
            if (!(yv.w > 0.f)) g.w = 0.f;
        }
        if (dres != nullptr) *reinterpret_cast<float4*>(dres + r * lddr + c0) = g;
        float4 o;
        o.x = ka.x * (g.x - k1.x - (xv.x - m.x) * rs.x * k2.x);
        o.y = ka.y * (g.y - k1.y - (xv.y - m.y) * rs.y * k2.y);
        o.z = ka.z * (g.z - k1.z - (xv.z - m.z) * rs.z * k2.z);
        o.w = ka.w * (g.w - k1.w - (xv.w - m.w) * rs.w * k2.w);
        if (dx != nullptr) *reinterpret_cast<float4*>(dx + r * lddx + c0) = o;
        if (dx16 != nullptr) *reinterpret_cast<uint2*>(dx16 + r * lddx + c0) = uint2{pdf_pk_bf16(o.x, o.y), pdf_pk_bf16(o.z, o.w)};
    }
}

// row chunks for the streaming passes: ~4096 blocks, >= 64 rows each
static long apply_rows_per_chunk(int C, long R) {
    static const long blocks = getenv("PDF_BN_APPLY_BLOCKS") ? atol(getenv("PDF_BN_APPLY_BLOCKS")) : 4096;
    long want = blocks / ((C + BN_CT - 1) / BN_CT);
    if (want < 1) want = 1;
    long rpc = (R + want - 1) / want;
    if (rpc < 64) rpc = 64;
    return (rpc + V4_TY - 1) / V4_TY * V4_TY;
}

// The second pass of a two-pass operator (BatchNorm forward: statistics, then apply; backward: sums, then apply) re-reads what the first
// pass has just streamed.  The 256 MiB Infinity Cache is memory-side and keeps what was touched most recently (MI355X_MICROARCH.md
// "Infinity Cache"): walking the tensor in the SAME order evicts a line just before it is wanted again once the operands exceed the
// cache; walking it BACKWARDS meets the most recently read rows first.  sign(rows_per_chunk) carries the order into the kernels.
static int bn_interleave() {                                  // 1: partial passes, 2: apply passes, 3: both (PDF_BN_INTERLEAVE)
    static const int v = getenv("PDF_BN_INTERLEAVE") ? atoi(getenv("PDF_BN_INTERLEAVE")) : 0;
    return v;
}
static int bn_second_pass_reverse() {
    static const int v = getenv("PDF_BN_REVERSE") ? atoi(getenv("PDF_BN_REVERSE")) : 0;
    return v;
}
static bool v4_ok(int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
    if (C % 4) return false;
    for (int l : lds) if (l % 4) return false;
    for (const void* p : ptrs) if (p != nullptr && (reinterpret_cast<uintptr_t>(p) & 15)) return false;
    return true;
}

static void launch_affine_apply(const float* x, int ldx, const float* scale, const float* shift, const float* res, int ldr,
                                float* y, int ldy, int C, long R, int relu, hipStream_t s, void* y16 = nullptr, const void* x16 = nullptr) {
    if (x16 != nullptr) {                                    // (the caller checked v4_ok)
        const long rpc = apply_rows_per_chunk(C, R);
        hipLaunchKernelGGL((affine_apply_v4_kernel<true>), dim3(cdiv(C, BN_CT), (unsigned)((R + rpc - 1) / rpc)), dim3(256), 0, s,
                           reinterpret_cast<const float*>(x16), ldx, scale, shift, res, ldr, y, ldy, C, R, (bn_interleave() & 2) ? 0 : bn_second_pass_reverse() ? -rpc : rpc, relu, reinterpret_cast<unsigned short*>(y16));
    } else if (v4_ok(C, {ldx, ldy, res ? ldr : 0}, {x, y, res, scale, shift})) {
        const long rpc = apply_rows_per_chunk(C, R);
        hipLaunchKernelGGL((affine_apply_v4_kernel<false>), dim3(cdiv(C, BN_CT), (unsigned)((R + rpc - 1) / rpc)), dim3(256), 0, s,
                           x, ldx, scale, shift, res, ldr, y, ldy, C, R, (bn_interleave() & 2) ? 0 : bn_second_pass_reverse() ? -rpc : rpc, relu, reinterpret_cast<unsigned short*>(y16));
    } else
        hipLaunchKernelGGL(affine_apply_kernel, dim3(grid_for(R * C)), dim3(256), 0, s, x, ldx, scale, shift, res, ldr, y, ldy, C, R * C, relu);
}

// Training forward.  ws: >= pdf_bn_workspace_floats(C, R) floats.  scale/shift [C] are outputs the
// caller keeps for backward-free reuse; save_mean/save_rstd [C] feed the backward.
static long bn_chunks(int C, long R) {
    static const long blocks = getenv("PDF_BN_PARTIAL_BLOCKS") ? atol(getenv("PDF_BN_PARTIAL_BLOCKS")) : 1024;
    static const long cap_env = getenv("PDF_BN_PARTIAL_CAP") ? atol(getenv("PDF_BN_PARTIAL_CAP")) : 0;
    const long ctiles = (C + BN_CT - 1) / BN_CT;
    long want = blocks / ctiles;                            // ~1024 blocks over (channel tiles x chunks)
    // at most 256 chunks -- but at least 512 blocks: a 64-channel tensor (one channel tile: the stem's 1M-row BatchNorm) ran its
    // statistics passes on 256 blocks, one per CU, at 4.3 TB/s; 512 chunks: 5.2 (profiles/r04_bn_chunks_ab.txt; 1024+ lose again)
    const long cap = cap_env > 0 ? cap_env : (512 / ctiles > 256 ? 512 / ctiles : 256);
    if (want > cap) want = cap;
    long by_rows = (R + 63) / 64;                           // at least 64 rows per chunk
    if (want > by_rows) want = by_rows;
    if (want < 1) want = 1;
    return want;
}
PDF_API long pdf_bn_workspace_floats(int C, long R) {
    long chunks = bn_chunks(C, R);
    return chunks * C * 2 + C;                              // (+ C: row 0 of a bf16 input as floats, BnFin::shift0)
}
// bf16 storage mode (PdfCallOpts::bn_x_bf16): pdf_bn_train_fwd / pdf_bn_train_bwd read their input x from a bf16 tensor (same shape and
// leading dimension in elements) instead of the fp32 pointer they are given

static int pdf_bn_train_fwd_impl(const float* x, int ldx, int C, long R, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps,
                             const float* res, int ldr, int relu, float* y, int ldy,
                             float* save_mean, float* save_rstd, float* scale, float* shift, float* ws, hipStream_t s, PdfCallOpts& co) {
    void* y16 = co.out_bf16;                                 // bf16 shadow of y, vectorised path only
    const TileStats ts = {co.tile_stats, co.tile_n, co.tile_rows};
    const void* x16 = co.bn_x_bf16;
    if (R <= 0 || C <= 0) return 0;
    if (x16 != nullptr && !v4_ok(C, {ldx, ldy, res ? ldr : 0}, {x16, y, res, scale, shift})) return PDF_E_BADARG;
    long chunks = bn_chunks(C, R);
    long rpc = (R + chunks - 1) / chunks;
    chunks = (R + rpc - 1) / rpc;
    BnFin fin = {gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_rstd, scale, shift, nullptr, ws + bn_chunks(C, R) * C * 2};
    if (ts.part != nullptr) {                                // statistics came out of the producing GEMM's epilogue: no pass over x
        if (ts.tiles * ts.rows < R || (ts.tiles - 1) * ts.rows >= R) return PDF_E_BADARG;
        hipLaunchKernelGGL(bn_finalize_tiles_kernel, dim3(cdiv(C, FT_C)), dim3(FT_C * FT_L), 0, s, ts.part, (int)ts.tiles, ts.rows, C, R, gamma, beta,
                           running_mean, running_var, momentum, eps, save_mean, save_rstd, scale, shift);
        fin.counters = reinterpret_cast<int*>(1);            // (marks "finalised" for the branch below)
    } else if (x16 != nullptr) {
        fin.counters = bn_inlaunch(C, R) ? pdf_ticket_counters(cdiv(C, BN_CT)) : nullptr;
        hipLaunchKernelGGL((bn_partial_v4_kernel<true>), dim3(cdiv(C, BN_CT), (unsigned)chunks), dim3(256), 0, s, reinterpret_cast<const float*>(x16), ldx, C, R, (bn_interleave() & 1) ? 0 : rpc, ws, fin);
    } else if (v4_ok(C, {ldx}, {x})) {
        fin.counters = bn_inlaunch(C, R) ? pdf_ticket_counters(cdiv(C, BN_CT)) : nullptr;          // finalize in the last block of each channel tile
        hipLaunchKernelGGL((bn_partial_v4_kernel<false>), dim3(cdiv(C, BN_CT), (unsigned)chunks), dim3(256), 0, s, x, ldx, C, R, (bn_interleave() & 1) ? 0 : rpc, ws, fin);
    } else
        hipLaunchKernelGGL(bn_partial_kernel, dim3(cdiv(C, BN_CT), (unsigned)chunks), dim3(256), 0, s, x, ldx, C, R, rpc, ws);
    PDF_LAUNCH_CHECK();
    if (fin.counters == nullptr) {
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, FIN_TX)), dim3(FIN_TX, FIN_TY), 0, s, ws, (int)chunks, x16 != nullptr ? fin.shift0 : x, C, R, gamma, beta,
                           running_mean, running_var, momentum, eps, save_mean, save_rstd, scale, shift);
        PDF_LAUNCH_CHECK();
    }
    if (y == nullptr) return res == nullptr && y16 == nullptr ? 0 : PDF_E_BADARG;      // statistics and coefficients only: the consumer applies them (pdf_set_input_affine_relu)
    if (x16 == nullptr && y16 != nullptr && !v4_ok(C, {ldx, ldy, res ? ldr : 0}, {x, y, res, scale, shift, y16})) return PDF_E_BADARG;
    launch_affine_apply(x, ldx, scale, shift, res, ldr, y, ldy, C, R, relu, s, y16, x16);
    PDF_LAUNCH_CHECK();
    return 0;
}
PDF_API int pdf_bn_train_fwd_x(const float* x, int ldx, int C, long R, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps,
                             const float* res, int ldr, int relu, float* y, int ldy,
                             float* save_mean, float* save_rstd, float* scale, float* shift, float* ws, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_bn_train_fwd_impl(x, ldx, C, R, gamma, beta, running_mean, running_var, momentum, eps, res, ldr, relu, y, ldy, save_mean, save_rstd, scale, shift, ws, s, co); }
PDF_API int pdf_bn_train_fwd(const float* x, int ldx, int C, long R, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps,
                             const float* res, int ldr, int relu, float* y, int ldy,
                             float* save_mean, float* save_rstd, float* scale, float* shift, float* ws, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_bn_train_fwd_impl(x, ldx, C, R, gamma, beta, running_mean, running_var, momentum, eps, res, ldr, relu, y, ldy, save_mean, save_rstd, scale, shift, ws, s, co); pdf_tls_publish(co); return rc; }


PDF_API int pdf_bn_eval_fwd(const float* x, int ldx, int C, long R, const float* gamma, const float* beta,
                            const float* running_mean, const float* running_var, float eps,
                            const float* res, int ldr, int relu, float* y, int ldy, float* scale, float* shift, hipStream_t s) {
    if (R <= 0 || C <= 0) return 0;
    hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3(cdiv(C, 128)), dim3(128), 0, s, C, gamma, beta, running_mean, running_var, eps, scale, shift);
    PDF_LAUNCH_CHECK();
    launch_affine_apply(x, ldx, scale, shift, res, ldr, y, ldy, C, R, relu, s);
    PDF_LAUNCH_CHECK();
    return 0;
}

// backward partials: sum(g), sum(g * xhat) with g = dy * (y > 0 if relu)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy, int relu,
                                                             const float* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int C, long R, long rows_per_chunk,
                                                             float* __restrict__ part) {
    __shared__ float s1[4][BN_CT], s2[4][BN_CT];
    const int tx = threadIdx.x & (BN_CT - 1), ty = threadIdx.x / BN_CT;
    const int c = blockIdx.x * BN_CT + tx;
    const long r0 = blockIdx.y * rows_per_chunk;
    const long r1 = min(R, r0 + rows_per_chunk);
    float a = 0.f, b = 0.f;
    if (c < C) {
        const float m = mean[c], rs = rstd[c];
        for (long r = r0 + ty; r < r1; r += 4) {
            float g = dy[r * lddy + c];
            if (relu && !((relu == 2 ? fmaf(x[r * ldx + c], scale[c], shift[c]) : y[r * ldy + c]) > 0.f)) g = 0.f;
            a += g; b += g * (x[r * ldx + c] - m) * rs;
        }
    }
    s1[ty][tx] = a; s2[ty][tx] = b;
    __syncthreads();
    if (ty == 0 && c < C) {
        part[((long)blockIdx.y * C + c) * 2 + 0] = s1[0][tx] + s1[1][tx] + s1[2][tx] + s1[3][tx];
        part[((long)blockIdx.y * C + c) * 2 + 1] = s2[0][tx] + s2[1][tx] + s2[2][tx] + s2[3][tx];
    }
}

__global__ __launch_bounds__(FIN_TX * FIN_TY) void bn_bwd_finalize_kernel(const float* __restrict__ part, int chunks, int C, long R, const float* __restrict__ gamma,
                                       const float* __restrict__ rstd, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                       float* __restrict__ coef /*[3][C]: a, c1, c2*/) {
    __shared__ double s1[FIN_TY][FIN_TX], s2[FIN_TY][FIN_TX];
    const int c = blockIdx.x * FIN_TX + threadIdx.x;
    double a, b;
    reduce_chunks(part, chunks, C, c, threadIdx.y, s1, s2, a, b);
    if (c >= C || threadIdx.y != 0) return;
    if (accumulate) { dbeta[c] += (float)a; dgamma[c] += (float)b; }
    else { dbeta[c] = (float)a; dgamma[c] = (float)b; }
    coef[c] = gamma[c] * rstd[c];
    coef[C + c] = (float)(a / (double)R);
    coef[2 * C + c] = (float)(b / (double)R);
}

// dx = a * (g - c1 - xhat * c2);  dres = g (optional)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy, int relu,
                                                           const float* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ coef,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, int C, long total,
                                                           float* __restrict__ dx, int lddx, float* __restrict__ dres, int lddr) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        float g = dy[r * lddy + c];
        if (relu && !((relu == 2 ? fmaf(x[r * ldx + c], scale[c], shift[c]) : y[r * ldy + c]) > 0.f)) g = 0.f;
        if (dres != nullptr) dres[r * lddr + c] = g;
        const float xh = (x[r * ldx + c] - mean[c]) * rstd[c];
        dx[r * lddx + c] = coef[c] * (g - coef[C + c] - xh * coef[2 * C + c]);
    }
}

// ws: pdf_bn_workspace_floats(C,R) + 3*C floats
// relu: 0 none; 1 ReLU, mask from the saved output y; 2 ReLU without residual, mask recomputed from x with the forward's
// scale / shift (y may be NULL: two fewer full-tensor reads)
static int pdf_bn_train_bwd_impl(const float* dy, int lddy, const float* y, int ldy, int relu, const float* x, int ldx,
                             const float* save_mean, const float* save_rstd, const float* gamma,
                             const float* scale, const float* shift, int C, long R,
                             float* dx, int lddx, float* dres, int lddr, float* dgamma, float* dbeta, int accumulate,
                             float* ws, hipStream_t s, PdfCallOpts& co) {
    void* dx16 = co.out_bf16;                                  // bf16 shadow of dx
    const void* x16 = co.bn_x_bf16;                         // bf16 storage mode: x comes as bf16; dx == NULL: only the bf16 gradient is written
    if (R <= 0 || C <= 0) return 0;
    if ((relu == 1 && y == nullptr) || (relu == 2 && (scale == nullptr || shift == nullptr || dres != nullptr))) return PDF_E_BADARG;
    if (dx == nullptr && dx16 == nullptr) return PDF_E_BADARG;
    long chunks = bn_chunks(C, R);
    long rpc = (R + chunks - 1) / chunks;
    chunks = (R + rpc - 1) / rpc;
    float* coef = ws + pdf_bn_workspace_floats(C, R);
    const float* xin = x16 != nullptr ? reinterpret_cast<const float*>(x16) : x;
    const bool vec = v4_ok(C, {lddy, ldx, lddx, relu == 1 ? ldy : 0, dres ? lddr : 0},
                           {dy, xin, dx, dres, relu == 1 ? y : nullptr, save_mean, save_rstd, coef, relu == 2 ? scale : nullptr, relu == 2 ? shift : nullptr});
    if ((x16 != nullptr || dx == nullptr) && !vec) return PDF_E_BADARG;
    BnBwdFin fin = {gamma, save_rstd, dgamma, dbeta, accumulate, coef, nullptr};
    if (vec) {
        fin.counters = bn_inlaunch(C, R) ? pdf_ticket_counters(cdiv(C, BN_CT)) : nullptr;
        if (x16 != nullptr)
            hipLaunchKernelGGL((bn_bwd_partial_v4_kernel<true>), dim3(cdiv(C, BN_CT), (unsigned)chunks), dim3(256), 0, s, dy, lddy, y, ldy, relu, xin, ldx,
                               save_mean, save_rstd, scale, shift, C, R, (bn_interleave() & 1) ? 0 : rpc, ws, fin);
        else
            hipLaunchKernelGGL((bn_bwd_partial_v4_kernel<false>), dim3(cdiv(C, BN_CT), (unsigned)chunks), dim3(256), 0, s, dy, lddy, y, ldy, relu, x, ldx,
                               save_mean, save_rstd, scale, shift, C, R, (bn_interleave() & 1) ? 0 : rpc, ws, fin);
    } else
        hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(cdiv(C, BN_CT), (unsigned)chunks), dim3(256), 0, s, dy, lddy, y, ldy, relu, x, ldx,
                           save_mean, save_rstd, scale, shift, C, R, rpc, ws);
    PDF_LAUNCH_CHECK();
    if (fin.counters == nullptr) {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, FIN_TX)), dim3(FIN_TX, FIN_TY), 0, s, ws, (int)chunks, C, R, gamma, save_rstd, dgamma, dbeta, accumulate, coef);
        PDF_LAUNCH_CHECK();
    }
    if (vec) {
        const long arpc = apply_rows_per_chunk(C, R);
        const dim3 grid(cdiv(C, BN_CT), (unsigned)((R + arpc - 1) / arpc));
        if (x16 != nullptr)
            hipLaunchKernelGGL((bn_bwd_apply_v4_kernel<true>), grid, dim3(256), 0, s, dy, lddy, y, ldy, relu,
                               xin, ldx, save_mean, save_rstd, coef, scale, shift, C, R, (bn_interleave() & 2) ? 0 : bn_second_pass_reverse() ? -arpc : arpc, dx, lddx, dres, lddr, reinterpret_cast<unsigned short*>(dx16));
        else
            hipLaunchKernelGGL((bn_bwd_apply_v4_kernel<false>), grid, dim3(256), 0, s, dy, lddy, y, ldy, relu,
                               x, ldx, save_mean, save_rstd, coef, scale, shift, C, R, (bn_interleave() & 2) ? 0 : bn_second_pass_reverse() ? -arpc : arpc, dx, lddx, dres, lddr, reinterpret_cast<unsigned short*>(dx16));
    } else if (dx16 != nullptr) return PDF_E_BADARG;
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(R * C)), dim3(256), 0, s, dy, lddy, y, ldy, relu, x, ldx, save_mean, save_rstd, coef,
                           scale, shift, C, R * C, dx, lddx, dres, lddr);
    PDF_LAUNCH_CHECK();
    return 0;
}
PDF_API int pdf_bn_train_bwd_x(const float* dy, int lddy, const float* y, int ldy, int relu, const float* x, int ldx,
                             const float* save_mean, const float* save_rstd, const float* gamma,
                             const float* scale, const float* shift, int C, long R,
                             float* dx, int lddx, float* dres, int lddr, float* dgamma, float* dbeta, int accumulate,
                             float* ws, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_bn_train_bwd_impl(dy, lddy, y, ldy, relu, x, ldx, save_mean, save_rstd, gamma, scale, shift, C, R, dx, lddx, dres, lddr, dgamma, dbeta, accumulate, ws, s, co); }
PDF_API int pdf_bn_train_bwd(const float* dy, int lddy, const float* y, int ldy, int relu, const float* x, int ldx,
                             const float* save_mean, const float* save_rstd, const float* gamma,
                             const float* scale, const float* shift, int C, long R,
                             float* dx, int lddx, float* dres, int lddr, float* dgamma, float* dbeta, int accumulate,
                             float* ws, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_bn_train_bwd_impl(dy, lddy, y, ldy, relu, x, ldx, save_mean, save_rstd, gamma, scale, shift, C, R, dx, lddx, dres, lddr, dgamma, dbeta, accumulate, ws, s, co); pdf_tls_publish(co); return rc; }


// ---------------------------------------------------------------------------------------------
// Set-abstraction tail (intaghand_encoder.py:59-62,79-82,97-100: BatchNorm2d -> ReLU -> MaxPool2d over the K neighbours) in
// one pass over the convolution output y [R][K][C]: out[r][c] = max_k relu(fma(y, scale, shift)), arg = first k attaining it.
// The normalised tensor (537 MB per hand at level 1) is never written; the backward rebuilds the gradient of y from
// (dout, arg, y): its two BatchNorm sums only involve the R*C selected elements.
__global__ __launch_bounds__(256) void bn_relu_maxk_fwd_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int C, int K, float* __restrict__ out, int ldo,
                                                               int* __restrict__ arg, long total /* R * C/4 */) {
    const int cq = C / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cq;
        const int c0 = (int)(i - r * cq) * 4;
        const float4 sc = *reinterpret_cast<const float4*>(scale + c0), sh = *reinterpret_cast<const float4*>(shift + c0);
        const float* p = y + r * K * ldy + c0;
        float4 best = make_float4(-1.f, -1.f, -1.f, -1.f);            // relu output is >= 0: the first k always wins against -1
        int4 bi = make_int4(0, 0, 0, 0);
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(p + (long)k * ldy);
            const float zx = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), zy = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
            const float zz = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f), zw = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
            if (zx > best.x) { best.x = zx; bi.x = k; }
            if (zy > best.y) { best.y = zy; bi.y = k; }
            if (zz > best.z) { best.z = zz; bi.z = k; }
            if (zw > best.w) { best.w = zw; bi.w = k; }
        }
        *reinterpret_cast<float4*>(out + r * ldo + c0) = best;
        *reinterpret_cast<int4*>(arg + r * C + c0) = bi;
    }
}
// partial sums over the rows r of g = dout * [z > 0] and g * xhat at the selected neighbour (same [chunks][C][2] layout as
// bn_bwd_partial_v4_kernel, so bn_bwd_finalize_kernel finishes them)
__global__ __launch_bounds__(256) void bn_maxk_bwd_partial_kernel(const float* __restrict__ dm, int lddm, const int* __restrict__ arg,
                                                                  const float* __restrict__ y, int ldy, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, int C, int K, long R, long rows_per_chunk,
                                                                  float* __restrict__ part, const BnBwdFin fin) {
    const long RT = R * K;                                   // the statistics were taken over R*K rows: so are the backward's means
    __shared__ float4 sa[V4_TY][V4_TX], sb[V4_TY][V4_TX];
    const int tx = threadIdx.x & (V4_TX - 1), ty = threadIdx.x / V4_TX;
    const int c0 = blockIdx.x * BN_CT + tx * 4;
    const long r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (c0 < C) {
        const float4 m = *reinterpret_cast<const float4*>(mean + c0), rs = *reinterpret_cast<const float4*>(rstd + c0);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c0), sh = *reinterpret_cast<const float4*>(shift + c0);
        for (long r = r0 + ty; r < r1; r += V4_TY) {
            const float4 g4 = *reinterpret_cast<const float4*>(dm + r * lddm + c0);
            const int4 k4 = *reinterpret_cast<const int4*>(arg + r * C + c0);
            const float* base = y + r * K * ldy + c0;
            const float vx = base[(long)k4.x * ldy], vy = base[(long)k4.y * ldy + 1], vz = base[(long)k4.z * ldy + 2], vw = base[(long)k4.w * ldy + 3];
            const float gx = fmaf(vx, sc.x, sh.x) > 0.f ? g4.x : 0.f, gy = fmaf(vy, sc.y, sh.y) > 0.f ? g4.y : 0.f;
            const float gz = fmaf(vz, sc.z, sh.z) > 0.f ? g4.z : 0.f, gw = fmaf(vw, sc.w, sh.w) > 0.f ? g4.w : 0.f;
            a.x += gx; a.y += gy; a.z += gz; a.w += gw;
            b.x += gx * (vx - m.x) * rs.x; b.y += gy * (vy - m.y) * rs.y; b.z += gz * (vz - m.z) * rs.z; b.w += gw * (vw - m.w) * rs.w;
        }
    }
    v4_block_reduce(a, b, sa, sb, part, C, c0, true, fin.counters != nullptr);
    if (fin.counters != nullptr) {
        __shared__ double red[512];
        __shared__ int flag;
        if (pdf_last_block_arrives(fin.counters + blockIdx.x, gridDim.y, &flag, false)) bn_bwd_finalize_tile(part, gridDim.y, C, RT, fin, blockIdx.x, red);
    }
}
// dy[r][k][c] = a * (g - c1 - xhat * c2), g = dout[r][c] where k == arg[r][c] and the ReLU was active, else 0
__global__ __launch_bounds__(256) void bn_maxk_bwd_apply_kernel(const float* __restrict__ dm, int lddm, const int* __restrict__ arg,
                                                                const float* __restrict__ y, int ldy, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ coef,
                                                                const float* __restrict__ scale, const float* __restrict__ shift, int C, int K,
                                                                float* __restrict__ dy, int lddy, long total /* R * C/4 */) {
    const int cq = C / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cq;
        const int c0 = (int)(i - r * cq) * 4;
        const float4 m = *reinterpret_cast<const float4*>(mean + c0), rs = *reinterpret_cast<const float4*>(rstd + c0);
        const float4 ka = *reinterpret_cast<const float4*>(coef + c0), k1 = *reinterpret_cast<const float4*>(coef + C + c0),
                     k2 = *reinterpret_cast<const float4*>(coef + 2 * C + c0);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c0), sh = *reinterpret_cast<const float4*>(shift + c0);
        const float4 g4 = *reinterpret_cast<const float4*>(dm + r * lddm + c0);
        const int4 k4 = *reinterpret_cast<const int4*>(arg + r * C + c0);
        const float* p = y + r * K * ldy + c0;
        float* q = dy + r * K * lddy + c0;
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(p + (long)k * ldy);
            const float gx = (k == k4.x && fmaf(v.x, sc.x, sh.x) > 0.f) ? g4.x : 0.f, gy = (k == k4.y && fmaf(v.y, sc.y, sh.y) > 0.f) ? g4.y : 0.f;
            const float gz = (k == k4.z && fmaf(v.z, sc.z, sh.z) > 0.f) ? g4.z : 0.f, gw = (k == k4.w && fmaf(v.w, sc.w, sh.w) > 0.f) ? g4.w : 0.f;
            float4 o;
            o.x = ka.x * (gx - k1.x - (v.x - m.x) * rs.x * k2.x);
            o.y = ka.y * (gy - k1.y - (v.y - m.y) * rs.y * k2.y);
            o.z = ka.z * (gz - k1.z - (v.z - m.z) * rs.z * k2.z);
            o.w = ka.w * (gw - k1.w - (v.w - m.w) * rs.w * k2.w);
            *reinterpret_cast<float4*>(q + (long)k * lddy) = o;
        }
    }
}
// training != 0: batch statistics over all R*K rows (running statistics updated with `momentum`); else running statistics.
// C % 4 == 0, 16-byte aligned rows.  ws: pdf_bn_workspace_floats(C, R*K) floats.
static int pdf_bn_relu_maxk_fwd_impl(const float* y, int ldy, int C, long R, int K, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps, int training,
                                 float* out, int ldo, int* arg, float* save_mean, float* save_rstd, float* scale, float* shift,
                                 float* ws, hipStream_t s, PdfCallOpts& co) {
    if (R <= 0 || C <= 0 || K <= 0) return 0;
    if (!v4_ok(C, {ldy, ldo}, {y, out, arg, scale, shift})) return PDF_E_BADARG;
    const long rows = R * K;
    const TileStats ts = {co.tile_stats, co.tile_n, co.tile_rows};
    if (training && ts.part != nullptr) {
        if (ts.tiles * ts.rows < rows || (ts.tiles - 1) * ts.rows >= rows) return PDF_E_BADARG;
        hipLaunchKernelGGL(bn_finalize_tiles_kernel, dim3(cdiv(C, FT_C)), dim3(FT_C * FT_L), 0, s, ts.part, (int)ts.tiles, ts.rows, C, rows, gamma, beta,
                           running_mean, running_var, momentum, eps, save_mean, save_rstd, scale, shift);
    } else if (training) {
        long chunks = bn_chunks(C, rows);
        long rpc = (rows + chunks - 1) / chunks;
        chunks = (rows + rpc - 1) / rpc;
        BnFin fin = {gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_rstd, scale, shift, bn_inlaunch(C, rows) ? pdf_ticket_counters(cdiv(C, BN_CT)) : nullptr};
        hipLaunchKernelGGL((bn_partial_v4_kernel<false>), dim3(cdiv(C, BN_CT), (unsigned)chunks), dim3(256), 0, s, y, ldy, C, rows, rpc, ws, fin);
        if (fin.counters == nullptr)
            hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, FIN_TX)), dim3(FIN_TX, FIN_TY), 0, s, ws, (int)chunks, y, C, rows, gamma, beta,
                               running_mean, running_var, momentum, eps, save_mean, save_rstd, scale, shift);
    } else {
        hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3(cdiv(C, 128)), dim3(128), 0, s, C, gamma, beta, running_mean, running_var, eps, scale, shift);
    }
    PDF_LAUNCH_CHECK();
    const long total = R * (C / 4);
    hipLaunchKernelGGL(bn_relu_maxk_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, y, ldy, scale, shift, C, K, out, ldo, arg, total);
    PDF_LAUNCH_CHECK();
    return 0;
}
PDF_API int pdf_bn_relu_maxk_fwd_x(const float* y, int ldy, int C, long R, int K, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps, int training,
                                 float* out, int ldo, int* arg, float* save_mean, float* save_rstd, float* scale, float* shift,
                                 float* ws, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_bn_relu_maxk_fwd_impl(y, ldy, C, R, K, gamma, beta, running_mean, running_var, momentum, eps, training, out, ldo, arg, save_mean, save_rstd, scale, shift, ws, s, co); }
PDF_API int pdf_bn_relu_maxk_fwd(const float* y, int ldy, int C, long R, int K, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps, int training,
                                 float* out, int ldo, int* arg, float* save_mean, float* save_rstd, float* scale, float* shift,
                                 float* ws, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_bn_relu_maxk_fwd_impl(y, ldy, C, R, K, gamma, beta, running_mean, running_var, momentum, eps, training, out, ldo, arg, save_mean, save_rstd, scale, shift, ws, s, co); pdf_tls_publish(co); return rc; }

// ws: pdf_bn_workspace_floats(C, R) + 3*C floats
PDF_API int pdf_bn_relu_maxk_bwd(const float* dout, int lddo, const int* arg, const float* y, int ldy, const float* save_mean, const float* save_rstd,
                                 const float* gamma, const float* scale, const float* shift, int C, long R, int K,
                                 float* dy, int lddy, float* dgamma, float* dbeta, int accumulate, float* ws, hipStream_t s) {
    if (R <= 0 || C <= 0 || K <= 0) return 0;
    if (!v4_ok(C, {ldy, lddo, lddy}, {y, dout, dy, arg, save_mean, save_rstd, scale, shift, ws})) return PDF_E_BADARG;
    long chunks = bn_chunks(C, R);
    long rpc = (R + chunks - 1) / chunks;
    chunks = (R + rpc - 1) / rpc;
    float* coef = ws + pdf_bn_workspace_floats(C, R);
    BnBwdFin fin = {gamma, save_rstd, dgamma, dbeta, accumulate, coef, bn_inlaunch(C, R) ? pdf_ticket_counters(cdiv(C, BN_CT)) : nullptr};
    hipLaunchKernelGGL(bn_maxk_bwd_partial_kernel, dim3(cdiv(C, BN_CT), (unsigned)chunks), dim3(256), 0, s, dout, lddo, arg, y, ldy, save_mean, save_rstd,
                       scale, shift, C, K, R, rpc, ws, fin);
    if (fin.counters == nullptr)      // the statistics were taken over R*K rows: the means of the backward are over R*K as well
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, FIN_TX)), dim3(FIN_TX, FIN_TY), 0, s, ws, (int)chunks, C, R * K, gamma, save_rstd, dgamma, dbeta, accumulate, coef);
    const long total = R * (C / 4);
    hipLaunchKernelGGL(bn_maxk_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, dout, lddo, arg, y, ldy, save_mean, save_rstd, coef,
                       scale, shift, C, K, dy, lddy, total);
    PDF_LAUNCH_CHECK();
    return 0;
}

// column sums: out[c] (+)= sum_r g[r][c]   (conv / linear bias gradients), optional relu mask by y
__global__ __launch_bounds__(FIN_TX * FIN_TY) void colsum_finalize_kernel(const float* __restrict__ part, int chunks, int C, float* __restrict__ out,
                                                                          float* __restrict__ out1, int accumulate) {
    __shared__ double s1[FIN_TY][FIN_TX], s2[FIN_TY][FIN_TX];
    if (blockIdx.y) { part += (long)chunks * C * 2; out = out1; }
    const int c = blockIdx.x * FIN_TX + threadIdx.x;
    double a, b;
    reduce_chunks(part, chunks, C, c, threadIdx.y, s1, s2, a, b);
    if (c >= C || threadIdx.y != 0) return;
    out[c] = (accumulate ? out[c] : 0.f) + (float)a;
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ g, int ldg, int C, long R, long rows_per_chunk, float* __restrict__ part) {
    __shared__ float s1[4][BN_CT];
    if (blockIdx.z) { g += R * ldg; part += (long)gridDim.y * C * 2; }
    const int tx = threadIdx.x & (BN_CT - 1), ty = threadIdx.x / BN_CT;
    const int c = blockIdx.x * BN_CT + tx;
    const long r0 = blockIdx.y * rows_per_chunk;
    const long r1 = min(R, r0 + rows_per_chunk);
    float a = 0.f;
    if (c < C) for (long r = r0 + ty; r < r1; r += 4) a += g[r * ldg + c];
    s1[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && c < C) {
        part[((long)blockIdx.y * C + c) * 2] = s1[0][tx] + s1[1][tx] + s1[2][tx] + s1[3][tx];
        part[((long)blockIdx.y * C + c) * 2 + 1] = 0.f;
    }
}

static int colsum_launch(const float* g, int ldg, int C, long R, float* out, float* out1, int accumulate, float* ws, hipStream_t s) {
    if (R <= 0 || C <= 0) return 0;
    const unsigned groups = out1 ? 2 : 1;
    long chunks = bn_chunks(C, R);
    long rpc = (R + chunks - 1) / chunks;
    chunks = (R + rpc - 1) / rpc;
    if (v4_ok(C, {ldg, (int)((R * ldg) % 4)}, {g}))
        hipLaunchKernelGGL(colsum_partial_v4_kernel, dim3(cdiv(C, BN_CT), (unsigned)chunks, groups), dim3(256), 0, s, g, ldg, C, R, rpc, ws);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(C, BN_CT), (unsigned)chunks, groups), dim3(256), 0, s, g, ldg, C, R, rpc, ws);
    PDF_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(C, FIN_TX), groups), dim3(FIN_TX, FIN_TY), 0, s, ws, (int)chunks, C, out, out1, accumulate);
    PDF_LAUNCH_CHECK();
    return 0;
}
int pdf_internal_colsum(const float* g, int ldg, int C, long R, float* out, int accumulate, float* ws, hipStream_t s) {
    return colsum_launch(g, ldg, C, R, out, nullptr, accumulate, ws, s);
}
long pdf_internal_colsum_ws(int C, long R) { return pdf_bn_workspace_floats(C, R); }
PDF_API int pdf_colsum(const float* g, int ldg, int C, long R, float* out, int accumulate, float* ws, hipStream_t s) {
    return colsum_launch(g, ldg, C, R, out, nullptr, accumulate, ws, s);
}
// paired bias gradients: rows [0, R) -> out0, rows [R, 2R) -> out1; ws >= 2 * pdf_bn_workspace_floats(C, R)
PDF_API int pdf_colsum_pair(const float* g, int ldg, int C, long R, float* out0, float* out1, int accumulate, float* ws, hipStream_t s) {
    return colsum_launch(g, ldg, C, R, out0, out1, accumulate, ws, s);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last axis (F <= 1024), one wave per row (nn.LayerNorm(eps=1e-6) in model_attn/*).
#define LN_MAXV 16
// One kernel family serves the plain op and the decoder's fused forms (gcn.py:100-110, self_attn.py:24-33,78-84):
//   z = x + dropout(add)      (optional; z is written out because it is also the residual the next block adds)
//   y = act( LayerNorm(z) * gamma_g + beta_g ),  g = 0 for rows < R_split, 1 otherwise (left / right hand parameters)
// blockIdx.y = group, so a block's dgamma / dbeta partial sums belong to one parameter set.
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ add, int ldadd,
                                                            float p, unsigned long long seed, const unsigned long long* __restrict__ step,
                                                            int F, long R, long R_split, const float* __restrict__ gamma0, const float* __restrict__ beta0,
                                                            const float* __restrict__ gamma1, const float* __restrict__ beta1, float eps, int act,
                                                            float* __restrict__ z, int ldz, float* __restrict__ y, int ldy,
                                                            float* __restrict__ mean, float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const long rb = blockIdx.y ? R_split : 0, re = blockIdx.y ? R : min(R, R_split);
    const float* __restrict__ gamma = blockIdx.y ? gamma1 : gamma0;
    const float* __restrict__ beta = blockIdx.y ? beta1 : beta0;
    const long w0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    const float sc = 1.f / (1.f - p);
    if (add != nullptr && p > 0.f && step != nullptr) seed += step[0] * 0x9E3779B97F4A7C15ull;
    for (long r = rb + w0; r < re; r += nw) {
        float v[LN_MAXV];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            int c = lane + 64 * i;
            float t = 0.f;
            if (c < F) {
                t = x[r * ldx + c];
                if (add != nullptr) {
                    float a = add[r * ldadd + c];
                    if (p > 0.f) a = pdf_uniform(seed, (unsigned long long)(r * F + c)) >= p ? a * sc : 0.f;
                    t += a;
                    z[r * ldz + c] = t;
                }
            }
            v[i] = t;
            sum += t;
        }
        const float mu = wave_sum(sum) / (float)F;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            int c = lane + 64 * i;
            float d = c < F ? v[i] - mu : 0.f;
            sq += d * d;
        }
        const float rs = 1.0f / sqrtf(wave_sum(sq) / (float)F + eps);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            int c = lane + 64 * i;
            if (c < F) {
                float o = (v[i] - mu) * rs * gamma[c] + beta[c];
                if (act == 1) o = fmaxf(o, 0.f);
                y[r * ldy + c] = o;
            }
        }
        if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
    }
}

static int ln_fwd_launch(const float* x, int ldx, const float* add, int ldadd, float p, unsigned long long seed, const unsigned long long* step,
                         int F, long R, long R_split, const float* g0, const float* b0, const float* g1, const float* b1, float eps, int act,
                         float* z, int ldz, float* y, int ldy, float* mean, float* rstd, hipStream_t s) {
    if (F > 64 * LN_MAXV || (add != nullptr && z == nullptr)) return PDF_E_BADARG;
    if (R <= 0) return 0;
    const unsigned groups = R_split < R ? 2 : 1;
    const long rows = groups == 2 ? max(R_split, R - R_split) : R;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid_for(rows * 64), groups), dim3(256), 0, s, x, ldx, add, ldadd, p, seed, step, F, R, R_split,
                       g0, b0, g1, b1, eps, act, z, ldz, y, ldy, mean, rstd);
    PDF_LAUNCH_CHECK();
    return 0;
}
PDF_API int pdf_layernorm_fwd(const float* x, int ldx, int F, long R, const float* gamma, const float* beta, float eps,
                              float* y, int ldy, float* mean, float* rstd, hipStream_t s) {
    return ln_fwd_launch(x, ldx, nullptr, 0, 0.f, 0, nullptr, F, R, R, gamma, beta, gamma, beta, eps, 0, nullptr, 0, y, ldy, mean, rstd, s);
}
// fused / paired form, see the kernel comment.  add == NULL: plain LayerNorm input (z unused).  R_split >= R: one parameter set.
PDF_API int pdf_layernorm_fused_fwd(const float* x, int ldx, const float* add, int ldadd, float p, unsigned long long seed,
                                    const unsigned long long* step, int F, long R, long R_split,
                                    const float* gamma0, const float* beta0, const float* gamma1, const float* beta1, float eps, int act,
                                    float* z, int ldz, float* y, int ldy, float* mean, float* rstd, hipStream_t s) {
    return ln_fwd_launch(x, ldx, add, ldadd, p, seed, step, F, R, R_split, gamma0, beta0, gamma1, beta1, eps, act, z, ldz, y, ldy, mean, rstd, s);
}

// dz = rstd * (g - mean(g) - xhat * mean(g*xhat)) [+ dz_in], g = dy * act'(y) * gamma; dgamma += sum dy' * xhat; dbeta += sum dy'
// dadd = dropout-mask(dz) (the gradient of the dropped operand); dz itself is the gradient of x (the residual operand).
// (dgamma/dbeta accumulated with one atomic per channel per block; caller zero-fills or accumulates)
// DZ: write dz / dadd (the data gradient, on the dependent chain); PARAMS: accumulate dgamma / dbeta.  The two halves can be
// launched separately: the parameter half then runs on the weight-gradient side stream, and the chain's half has no atomics
// (512 blocks adding into the same 2 F addresses were most of the one-shot kernel's 28 us).
template <bool DZ, bool PARAMS>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy, int act,
                                                            const float* __restrict__ zin, int ldz, int F, long R, long R_split,
                                                            const float* __restrict__ gamma0, const float* __restrict__ gamma1,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ dz_in, int lddzin, float* __restrict__ dz, int lddz,
                                                            float* __restrict__ dadd, int lddadd, float p, unsigned long long seed,
                                                            const unsigned long long* __restrict__ step,
                                                            float* __restrict__ dgamma0, float* __restrict__ dbeta0,
                                                            float* __restrict__ dgamma1, float* __restrict__ dbeta1) {
    __shared__ float sg[PARAMS ? 64 * LN_MAXV : 1], sb[PARAMS ? 64 * LN_MAXV : 1];
    if constexpr (PARAMS) {
        for (int i = threadIdx.x; i < F; i += 256) { sg[i] = 0.f; sb[i] = 0.f; }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const long rb = blockIdx.y ? R_split : 0, re = blockIdx.y ? R : min(R, R_split);
    const float* __restrict__ gamma = blockIdx.y ? gamma1 : gamma0;
    float* __restrict__ dgamma = blockIdx.y ? dgamma1 : dgamma0;
    float* __restrict__ dbeta = blockIdx.y ? dbeta1 : dbeta0;
    const long w0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    const float sc = 1.f / (1.f - p);
    if (dadd != nullptr && p > 0.f && step != nullptr) seed += step[0] * 0x9E3779B97F4A7C15ull;
    float ag[LN_MAXV], ab[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) { ag[i] = 0.f; ab[i] = 0.f; }
    for (long r = rb + w0; r < re; r += nw) {
        const float mu = mean[r], rs = rstd[r];
        float g[LN_MAXV], xh[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            int c = lane + 64 * i;
            float d = 0.f, h = 0.f;
            if (c < F) {
                d = dy[r * lddy + c];
                if (act == 1 && !(y[r * ldy + c] > 0.f)) d = 0.f;
                h = (zin[r * ldz + c] - mu) * rs;
                if constexpr (PARAMS) { ag[i] += d * h; ab[i] += d; }
                d *= gamma[c];
            }
            g[i] = d; xh[i] = h;
            s1 += d; s2 += d * h;
        }
        if constexpr (!DZ) continue;
        s1 = wave_sum(s1) / (float)F;
        s2 = wave_sum(s2) / (float)F;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            int c = lane + 64 * i;
            if (c < F) {
                float o = rs * (g[i] - s1 - xh[i] * s2);
                if (dz_in != nullptr) o += dz_in[r * lddzin + c];
                dz[r * lddz + c] = o;
                if (dadd != nullptr) {
                    if (p > 0.f) o = pdf_uniform(seed, (unsigned long long)(r * F + c)) >= p ? o * sc : 0.f;
                    dadd[r * lddadd + c] = o;
                }
            }
        }
    }
    if constexpr (PARAMS) {
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            int c = lane + 64 * i;
            if (c < F) { atomicAdd(&sg[c], ag[i]); atomicAdd(&sb[c], ab[i]); }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < F; i += 256) {
            if (dgamma != nullptr) atomicAdd(&dgamma[i], sg[i]);
            if (dbeta != nullptr) atomicAdd(&dbeta[i], sb[i]);
        }
    }
}

static int ln_bwd_launch(const float* dy, int lddy, const float* y, int ldy, int act, const float* z, int ldz, int F, long R, long R_split,
                         const float* g0, const float* g1, const float* mean, const float* rstd, const float* dz_in, int lddzin,
                         float* dz, int lddz, float* dadd, int lddadd, float p, unsigned long long seed, const unsigned long long* step,
                         float* dg0, float* db0, float* dg1, float* db1, hipStream_t s) {
    if (F > 64 * LN_MAXV || (act == 1 && y == nullptr)) return PDF_E_BADARG;
    if (R <= 0) return 0;
    const unsigned groups = R_split < R ? 2 : 1;
    const long rows = groups == 2 ? max(R_split, R - R_split) : R;
    const bool params = dg0 != nullptr || db0 != nullptr || dg1 != nullptr || db1 != nullptr;
    if (dz == nullptr && !params) return 0;
    if (dz == nullptr && (dadd != nullptr || dz_in != nullptr)) return PDF_E_BADARG;
    // one-shot / parameter half: few blocks (every block ends with 2 F same-address atomics); data half: one row per wave
    const int grid = grid_for(rows * 64, 256, dz == nullptr ? 128 : params ? 512 : 4096);
#define LN_BWD(DZ_, PA_) hipLaunchKernelGGL((layernorm_bwd_kernel<DZ_, PA_>), dim3(grid, groups), dim3(256), 0, s, dy, lddy, y, ldy, act, z, ldz, F, R, \
                       R_split, g0, g1, mean, rstd, dz_in, lddzin, dz, lddz, dadd, lddadd, p, seed, step, dg0, db0, dg1, db1)
    if (dz == nullptr) LN_BWD(false, true); else if (params) LN_BWD(true, true); else LN_BWD(true, false);
#undef LN_BWD
    PDF_LAUNCH_CHECK();
    return 0;
}
PDF_API int pdf_layernorm_bwd(const float* dy, int lddy, const float* x, int ldx, int F, long R, const float* gamma,
                              const float* mean, const float* rstd, float* dx, int lddx, float* dgamma, float* dbeta, hipStream_t s) {
    return ln_bwd_launch(dy, lddy, nullptr, 0, 0, x, ldx, F, R, R, gamma, gamma, mean, rstd, nullptr, 0, dx, lddx, nullptr, 0, 0.f, 0, nullptr,
                         dgamma, dbeta, dgamma, dbeta, s);
}
PDF_API int pdf_layernorm_fused_bwd(const float* dy, int lddy, const float* y, int ldy, int act, const float* z, int ldz, int F, long R, long R_split,
                                    const float* gamma0, const float* gamma1, const float* mean, const float* rstd,
                                    const float* dz_in, int lddzin, float* dz, int lddz, float* dadd, int lddadd,
                                    float p, unsigned long long seed, const unsigned long long* step,
                                    float* dgamma0, float* dbeta0, float* dgamma1, float* dbeta1, hipStream_t s) {
    return ln_bwd_launch(dy, lddy, y, ldy, act, z, ldz, F, R, R_split, gamma0, gamma1, mean, rstd, dz_in, lddzin, dz, lddz, dadd, lddadd, p, seed, step,
                         dgamma0, dbeta0, dgamma1, dbeta1, s);
}

// ---------------------------------------------------------------------------------------------
// L2Norm (intaghand_encoder.py:318-334): y = w[c] * x / (sqrt(sum_c x^2) + 1e-10), one wave per pixel.
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, int ldx, int C, long R, const float* __restrict__ w,
                                                         float eps, float* __restrict__ y, int ldy, float* __restrict__ norm) {
    const int lane = threadIdx.x & 63;
    const long w0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long r = w0; r < R; r += nw) {
        float sq = 0.f;
        for (int c = lane; c < C; c += 64) { float v = x[r * ldx + c]; sq += v * v; }
        const float n = sqrtf(wave_sum(sq)) + eps;
        for (int c = lane; c < C; c += 64) y[r * ldy + c] = w[c] * (x[r * ldx + c] / n);
        if (lane == 0) norm[r] = n;
    }
}

PDF_API int pdf_l2norm_fwd(const float* x, int ldx, int C, long R, const float* w, float eps, float* y, int ldy, float* norm, hipStream_t s) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(grid_for(R * 64)), dim3(256), 0, s, x, ldx, C, R, w, eps, y, ldy, norm);
    PDF_LAUNCH_CHECK();
    return 0;
}

// dx_j = w_j g_j / n - x_j * (sum_c w_c g_c x_c) / (n^2 * (n - eps));  dw_c += sum_r g_c x_c / n
#define L2_MAXV 8     // channels per lane: C <= 512
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx, int C, long R,
                                                         const float* __restrict__ w, float eps, const float* __restrict__ norm,
                                                         float* __restrict__ dx, int lddx, float* __restrict__ dw) {
    extern __shared__ float sw[];
    for (int i = threadIdx.x; i < C; i += 256) sw[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long w0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    float acc[L2_MAXV], wv[L2_MAXV];
#pragma unroll
    for (int i = 0; i < L2_MAXV; ++i) { acc[i] = 0.f; int c = lane + 64 * i; wv[i] = c < C ? w[c] : 0.f; }
    for (long r = w0; r < R; r += nw) {
        const float n = norm[r];
        float g[L2_MAXV], xv[L2_MAXV];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < L2_MAXV; ++i) {
            int c = lane + 64 * i;
            g[i] = c < C ? dy[r * lddy + c] : 0.f;
            xv[i] = c < C ? x[r * ldx + c] : 0.f;
            dot += wv[i] * g[i] * xv[i];
        }
        dot = wave_sum(dot);
        const float nn = n - eps;
        const float k = nn > 0.f ? dot / (n * n * nn) : 0.f;
        const float in = 1.f / n;
#pragma unroll
        for (int i = 0; i < L2_MAXV; ++i) {
            int c = lane + 64 * i;
            if (c < C) dx[r * lddx + c] = wv[i] * g[i] * in - xv[i] * k;
            acc[i] += g[i] * xv[i] * in;                     // dw partial stays in registers across this wave's rows
        }
    }
#pragma unroll
    for (int i = 0; i < L2_MAXV; ++i) { int c = lane + 64 * i; if (c < C) atomicAdd(&sw[c], acc[i]); }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) atomicAdd(&dw[i], sw[i]);
}

PDF_API int pdf_l2norm_bwd(const float* dy, int lddy, const float* x, int ldx, int C, long R, const float* w, float eps,
                           const float* norm, float* dx, int lddx, float* dw, hipStream_t s) {
    if (R <= 0) return 0;
    if (C > 64 * L2_MAXV) return PDF_E_BADARG;
    int grid = grid_for(R * 64, 256, 1024);
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(grid), dim3(256), C * sizeof(float), s, dy, lddy, x, ldx, C, R, w, eps, norm, dx, lddx, dw);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---- the pyramid form (intaghand_encoder.py:724-739): up to 4 maps L2-normalised into / out of ONE concatenated NHWC buffer
// in ONE launch (blockIdx.y = part).  One launch per part reads a 1 KB slice of every 4 KB row of the concatenated gradient,
// i.e. a quarter of the HBM channels at a time: measured 529 us per part against 87 us on contiguous rows.
#define L2_MAXPARTS 4
struct L2Parts {
    const float* x[L2_MAXPARTS]; const float* w[L2_MAXPARTS]; float* norm[L2_MAXPARTS]; float* dx[L2_MAXPARTS]; float* dw[L2_MAXPARTS];
    unsigned short* dx16[L2_MAXPARTS];                   // optional bf16 shadows of dx (bf16 mode)
    int C[L2_MAXPARTS], off[L2_MAXPARTS];
};
__global__ __launch_bounds__(256) void l2norm_cat_fwd_kernel(const L2Parts p, long R, float eps, float* __restrict__ y, int ldy,
                                                             unsigned short* __restrict__ y16) {
    const int part = blockIdx.y, C = p.C[part], lane = threadIdx.x & 63;
    const float* __restrict__ x = p.x[part]; const float* __restrict__ w = p.w[part];
    float* __restrict__ yo = y + p.off[part];
    const long w0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long r = w0; r < R; r += nw) {
        float sq = 0.f;
        for (int c = lane; c < C; c += 64) { float v = x[r * C + c]; sq += v * v; }
        const float n = sqrtf(wave_sum(sq)) + eps;
        for (int c = lane; c < C; c += 64) {
            const float v = w[c] * (x[r * C + c] / n);
            yo[r * ldy + c] = v;
            if (y16 != nullptr) {                            // bf16 shadow: lanes pair up so that 4-byte words are written
                const float vn = __shfl_down(v, 1, 64);
                if ((lane & 1) == 0) *reinterpret_cast<unsigned int*>(y16 + p.off[part] + r * ldy + c) = pdf_pk_bf16(v, vn);
            }
        }
        if (lane == 0) p.norm[part][r] = n;
    }
}
// 256-channel parts (the pyramid): a lane owns 4 consecutive channels -- one 16-byte load per row, the row stays in registers for
// the second pass, two rows per wave in flight (the generic form above re-reads the row with 4-byte accesses: 2.8 TB/s by counters)
__global__ __launch_bounds__(256) void l2norm_cat_fwd_c256_kernel(const L2Parts p, long R, float eps, float* __restrict__ y, int ldy,
                                                                  unsigned short* __restrict__ y16) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    const int part = blockIdx.y, lane = threadIdx.x & 63;
    const float* __restrict__ x = p.x[part];
    float* __restrict__ yo = y + p.off[part];
    const v4f wv = reinterpret_cast<const v4f*>(p.w[part])[lane];
    const long w0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long r = w0; r < R; r += 2 * nw) {
        const long r1 = r + nw;
        const bool two = r1 < R;
        const v4f a = reinterpret_cast<const v4f*>(x + r * 256)[lane];
        v4f b = {0.f, 0.f, 0.f, 0.f};
        if (two) b = reinterpret_cast<const v4f*>(x + r1 * 256)[lane];
        const float na = sqrtf(wave_sum(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w)) + eps;
        const float nb = sqrtf(wave_sum(b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w)) + eps;
        const v4f oa = {wv.x * (a.x / na), wv.y * (a.y / na), wv.z * (a.z / na), wv.w * (a.w / na)};
        *reinterpret_cast<v4f*>(yo + r * ldy + 4 * lane) = oa;
        if (y16 != nullptr) *reinterpret_cast<v2u*>(y16 + p.off[part] + r * ldy + 4 * lane) = v2u{pdf_pk_bf16(oa.x, oa.y), pdf_pk_bf16(oa.z, oa.w)};
        if (lane == 0) p.norm[part][r] = na;
        if (two) {
            const v4f ob = {wv.x * (b.x / nb), wv.y * (b.y / nb), wv.z * (b.z / nb), wv.w * (b.w / nb)};
            *reinterpret_cast<v4f*>(yo + r1 * ldy + 4 * lane) = ob;
            if (y16 != nullptr) *reinterpret_cast<v2u*>(y16 + p.off[part] + r1 * ldy + 4 * lane) = v2u{pdf_pk_bf16(ob.x, ob.y), pdf_pk_bf16(ob.z, ob.w)};
            if (lane == 0) p.norm[part][r1] = nb;
        }
    }
}
template <int MAXV>
__global__ __launch_bounds__(256) void l2norm_cat_bwd_kernel(const L2Parts p, long R, float eps, const float* __restrict__ dy, int lddy) {
    // blockIdx.y = part, one wave per pixel.  Kept light on registers (MAXV = channels per lane: 4 for the 256-channel pyramid):
    // this kernel runs beside the `feat` weight gradient, whose three resident blocks per CU leave ~80 VGPRs per SIMD lane free -- a
    // walk over all four parts per wave (96 VGPRs) could not co-reside at all and took 4.9 ms there (0.3 ms alone); this form 2.5.
    // (The same trimming of the BatchNorm backward kernels -- 100 -> 54-80 VGPRs via unroll 2-3 -- was measured and is neutral in fp32,
    // -2 % in bf16 where they run alone more often.)
    __shared__ float sw[64 * MAXV];
    const int q = blockIdx.y, C = p.C[q], lane = threadIdx.x & 63;
    const float* __restrict__ x = p.x[q]; const float* __restrict__ w = p.w[q]; const float* __restrict__ norm = p.norm[q];
    const float* __restrict__ g0 = dy + p.off[q];
    float* __restrict__ dx = p.dx[q];
    unsigned short* __restrict__ dx16 = p.dx16[q];
    for (int i = threadIdx.x; i < C; i += 256) sw[i] = 0.f;
    __syncthreads();
    const long w0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    float acc[MAXV], wv[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { acc[i] = 0.f; const int c = lane + 64 * i; wv[i] = c < C ? w[c] : 0.f; }
    for (long r = w0; r < R; r += nw) {
        const float n = norm[r];
        float g[MAXV], xv[MAXV];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            g[i] = c < C ? g0[r * lddy + c] : 0.f;
            xv[i] = c < C ? x[r * C + c] : 0.f;
            dot += wv[i] * g[i] * xv[i];
        }
        dot = wave_sum(dot);
        const float nn = n - eps;
        const float k = nn > 0.f ? dot / (n * n * nn) : 0.f;
        const float in = 1.f / n;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            const float o = wv[i] * g[i] * in - xv[i] * k;
            if (c < C) dx[r * C + c] = o;
            if (dx16 != nullptr) {                            // (C % 64 == 0 then: whole waves, lanes pair up for 4-byte words)
                const float on = __shfl_down(o, 1, 64);
                if (c < C && (lane & 1) == 0) *reinterpret_cast<unsigned int*>(dx16 + r * C + c) = pdf_pk_bf16(o, on);
            }
            acc[i] += g[i] * xv[i] * in;
        }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { const int c = lane + 64 * i; if (c < C) atomicAdd(&sw[c], acc[i]); }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) atomicAdd(&p.dw[q][i], sw[i]);
}
static int l2_parts(L2Parts& p, int nparts, const float* const* x, const float* const* w, float* const* norm, float* const* dx, float* const* dw,
                    const int* C) {
    if (nparts < 1 || nparts > L2_MAXPARTS) return PDF_E_BADARG;
    int off = 0;
    for (int i = 0; i < nparts; ++i) {
        if (C[i] <= 0 || C[i] > 64 * L2_MAXV) return PDF_E_BADARG;
        p.x[i] = x[i]; p.w[i] = w[i]; p.norm[i] = norm[i]; p.dx[i] = dx ? dx[i] : nullptr; p.dw[i] = dw ? dw[i] : nullptr;
        p.C[i] = C[i]; p.off[i] = off;
        off += C[i];
    }
    return 0;
}
// x[i]: [R][C[i]] contiguous rows; y: [R][ldy] with part i at channel offset C[0] + ... + C[i-1]; norm[i]: [R]
static int pdf_l2norm_cat_fwd_impl(int nparts, const float* const* x, const int* C, const float* const* w, float eps, long R,
                               float* y, int ldy, float* const* norm, hipStream_t s, PdfCallOpts& co) {
    void* y16 = co.out_bf16;                                 // bf16 shadow of y (needs even channel counts)
    if (R <= 0) return 0;
    L2Parts p = {};
    if (int rc = l2_parts(p, nparts, x, w, norm, nullptr, nullptr, C)) return rc;
    for (int i = 0; y16 != nullptr && i < nparts; ++i) if (C[i] % 64) return PDF_E_BADARG;
    bool c256 = ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (y16 == nullptr || (reinterpret_cast<uintptr_t>(y16) & 7) == 0);
    for (int i = 0; i < nparts; ++i)
        c256 = c256 && C[i] == 256 && p.off[i] % 4 == 0 && (reinterpret_cast<uintptr_t>(x[i]) & 15) == 0 && (reinterpret_cast<uintptr_t>(w[i]) & 15) == 0;
    if (c256) hipLaunchKernelGGL(l2norm_cat_fwd_c256_kernel, dim3(grid_for(R * 32, 256, 4096), nparts), dim3(256), 0, s, p, R, eps, y, ldy,
                                 reinterpret_cast<unsigned short*>(y16));
    else hipLaunchKernelGGL(l2norm_cat_fwd_kernel, dim3(grid_for(R * 64, 256, 2048), nparts), dim3(256), 0, s, p, R, eps, y, ldy,
                            reinterpret_cast<unsigned short*>(y16));
    PDF_LAUNCH_CHECK();
    return 0;
}
PDF_API int pdf_l2norm_cat_fwd_x(int nparts, const float* const* x, const int* C, const float* const* w, float eps, long R,
                               float* y, int ldy, float* const* norm, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_l2norm_cat_fwd_impl(nparts, x, C, w, eps, R, y, ldy, norm, s, co); }
PDF_API int pdf_l2norm_cat_fwd(int nparts, const float* const* x, const int* C, const float* const* w, float eps, long R,
                               float* y, int ldy, float* const* norm, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_l2norm_cat_fwd_impl(nparts, x, C, w, eps, R, y, ldy, norm, s, co); pdf_tls_publish(co); return rc; }

// dw[i] must be zero-filled (atomically accumulated)
PDF_API int pdf_l2norm_cat_bwd(int nparts, const float* dy, int lddy, const float* const* x, const int* C, const float* const* w, float eps, long R,
                               float* const* norm, float* const* dx, float* const* dw, void* const* dx16, hipStream_t s) {
    if (R <= 0) return 0;
    L2Parts p = {};
    if (int rc = l2_parts(p, nparts, x, w, norm, dx, dw, C)) return rc;
    for (int i = 0; dx16 != nullptr && i < nparts; ++i) {
        if (dx16[i] != nullptr && C[i] % 64) return PDF_E_BADARG;
        p.dx16[i] = reinterpret_cast<unsigned short*>(dx16[i]);
    }
    int cmax = 0;
    for (int i = 0; i < nparts; ++i) cmax = max(cmax, C[i]);
    const dim3 grid(grid_for(R * 64, 256, 1024), nparts);
    if (cmax <= 256) hipLaunchKernelGGL(l2norm_cat_bwd_kernel<4>, grid, dim3(256), 0, s, p, R, eps, dy, lddy);
    else hipLaunchKernelGGL(l2norm_cat_bwd_kernel<L2_MAXV>, grid, dim3(256), 0, s, p, R, eps, dy, lddy);
    PDF_LAUNCH_CHECK();
    return 0;
}
