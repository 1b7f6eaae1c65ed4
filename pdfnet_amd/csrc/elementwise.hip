// Streaming elementwise / spatial kernels (all HBM-bound, channel axis contiguous):
// activations, SFT modulation, dropout, 3x3/s2 max-pool, bilinear x2 (align_corners), fused Adam.
#include "common.h"

#define GRID_STRIDE(i, total) for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (total); i += (long)gridDim.x * blockDim.x)

// act: 1 relu, 2 leaky-relu(0.1)
__global__ void act_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int C, long total, int act) {
    GRID_STRIDE(i, total) {
        long r = i / C; int c = (int)(i - r * C);
        float v = x[r * ldx + c];
        y[r * ldy + c] = act == 1 ? fmaxf(v, 0.f) : (v > 0.f ? v : 0.1f * v);
    }
}
PDF_API int pdf_act_fwd(const float* x, int ldx, float* y, int ldy, int C, long R, int act, hipStream_t s) {
    if (R * C <= 0) return 0;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(R * C)), dim3(256), 0, s, x, ldx, y, ldy, C, R * C, act);
    PDF_LAUNCH_CHECK();
    return 0;
}

// dx = dy * act'(.) using the activation OUTPUT y (sign(y) == sign(pre-activation) for both)
__global__ void act_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                               float* __restrict__ dx, int lddx, int C, long total, int act) {
    GRID_STRIDE(i, total) {
        long r = i / C; int c = (int)(i - r * C);
        float g = dy[r * lddy + c];
        bool pos = y[r * ldy + c] > 0.f;
        dx[r * lddx + c] = pos ? g : (act == 1 ? 0.f : 0.1f * g);
    }
}
PDF_API int pdf_act_bwd(const float* dy, int lddy, const float* y, int ldy, float* dx, int lddx, int C, long R, int act, hipStream_t s) {
    if (R * C <= 0) return 0;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(R * C)), dim3(256), 0, s, dy, lddy, y, ldy, dx, lddx, C, R * C, act);
    PDF_LAUNCH_CHECK();
    return 0;
}

// out = fea * (scale + 1) + shift      (SFTLayer.forward, intaghand_encoder.py:213-219)
__global__ void sft_fwd_kernel(const float* __restrict__ fea, int ldf, const float* __restrict__ scale, int lds_, const float* __restrict__ shift, int ldh,
                               float* __restrict__ out, int ldo, int C, long total) {
    GRID_STRIDE(i, total) {
        long r = i / C; int c = (int)(i - r * C);
        out[r * ldo + c] = fea[r * ldf + c] * (scale[r * lds_ + c] + 1.f) + shift[r * ldh + c];
    }
}
PDF_API int pdf_sft_fwd(const float* fea, int ldf, const float* scale, int lds_, const float* shift, int ldh,
                        float* out, int ldo, int C, long R, hipStream_t s) {
    if (R * C <= 0) return 0;
    hipLaunchKernelGGL(sft_fwd_kernel, dim3(grid_for(R * C)), dim3(256), 0, s, fea, ldf, scale, lds_, shift, ldh, out, ldo, C, R * C);
    PDF_LAUNCH_CHECK();
    return 0;
}
// dfea = g*(scale+1); dscale = g*fea; (dshift = g: caller aliases)
__global__ void sft_bwd_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ fea, int ldf, const float* __restrict__ scale, int lds_,
                               float* __restrict__ dfea, int lddf, float* __restrict__ dscale, int ldds, int C, long total) {
    GRID_STRIDE(i, total) {
        long r = i / C; int c = (int)(i - r * C);
        float gv = g[r * ldg + c];
        dfea[r * lddf + c] = gv * (scale[r * lds_ + c] + 1.f);
        dscale[r * ldds + c] = gv * fea[r * ldf + c];
    }
}
PDF_API int pdf_sft_bwd(const float* g, int ldg, const float* fea, int ldf, const float* scale, int lds_,
                        float* dfea, int lddf, float* dscale, int ldds, int C, long R, hipStream_t s) {
    if (R * C <= 0) return 0;
    hipLaunchKernelGGL(sft_bwd_kernel, dim3(grid_for(R * C)), dim3(256), 0, s, g, ldg, fea, ldf, scale, lds_, dfea, lddf, dscale, ldds, C, R * C);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---- SFTLayer(3, 3) as ONE kernel (sft0, intaghand_encoder.py:120-122 with :205-219): the four 3 -> 3 point convolutions,
// the two LeakyReLU(0.1) and the modulation for one point per thread.  As separate launches the layer is 10 forward and
// ~30 backward launches over 32 K x 3 floats (the 3 x 3 weight gradients alone took 47 us each).
struct Sft3Params { const float* w[4]; const float* b[4]; };      // scale_conv0, scale_conv1, shift_conv0, shift_conv1: [3][3], [3]
__device__ __forceinline__ void mat3(const float* __restrict__ w, const float* __restrict__ b, const float (&x)[3], float (&y)[3]) {
#pragma unroll
    for (int o = 0; o < 3; ++o) y[o] = fmaf(w[o * 3 + 2], x[2], fmaf(w[o * 3 + 1], x[1], fmaf(w[o * 3], x[0], b[o])));
}
__global__ __launch_bounds__(256) void sft3_fwd_kernel(const float* __restrict__ fea, int ldf, const float* __restrict__ cond, int ldc, const Sft3Params p,
                                                        float* __restrict__ out, int ldo, long R) {
    GRID_STRIDE(r, R) {
        const float c[3] = {cond[r * ldc], cond[r * ldc + 1], cond[r * ldc + 2]};
        float zs[3], zh[3], sc[3], sh[3];
        mat3(p.w[0], p.b[0], c, zs);
        mat3(p.w[2], p.b[2], c, zh);
#pragma unroll
        for (int i = 0; i < 3; ++i) { zs[i] = zs[i] > 0.f ? zs[i] : 0.1f * zs[i]; zh[i] = zh[i] > 0.f ? zh[i] : 0.1f * zh[i]; }
        mat3(p.w[1], p.b[1], zs, sc);
        mat3(p.w[3], p.b[3], zh, sh);
#pragma unroll
        for (int i = 0; i < 3; ++i) out[r * ldo + i] = fea[r * ldf + i] * (sc[i] + 1.f) + sh[i];
    }
}
PDF_API int pdf_sft3_fwd(const float* fea, int ldf, const float* cond, int ldc, const float* w_scale0, const float* b_scale0,
                         const float* w_scale1, const float* b_scale1, const float* w_shift0, const float* b_shift0,
                         const float* w_shift1, const float* b_shift1, float* out, int ldo, long R, hipStream_t s) {
    if (R <= 0) return 0;
    Sft3Params p = {{w_scale0, w_scale1, w_shift0, w_shift1}, {b_scale0, b_scale1, b_shift0, b_shift1}};
    hipLaunchKernelGGL(sft3_fwd_kernel, dim3(grid_for(R)), dim3(256), 0, s, fea, ldf, cond, ldc, p, out, ldo, R);
    PDF_LAUNCH_CHECK();
    return 0;
}
// backward: dfea, dcond per point; the 48 parameter-gradient sums (per conv: dW [3][3] then db [3], in Sft3Params order) as
// per-block partials part[block][48] (fixed order: lane tree, 4 waves), summed by sft3_bwd_finish in block order.
#define SFT3_BLOCKS 256
__global__ __launch_bounds__(256) void sft3_bwd_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ fea, int ldf,
                                                        const float* __restrict__ cond, int ldc, const Sft3Params p,
                                                        float* __restrict__ dfea, int lddf, float* __restrict__ dcond, int lddc,
                                                        float* __restrict__ part, long R) {
    __shared__ float red[4][48];
    float acc[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) acc[i] = 0.f;
    GRID_STRIDE(r, R) {
        const float c[3] = {cond[r * ldc], cond[r * ldc + 1], cond[r * ldc + 2]};
        const float gv[3] = {g[r * ldg], g[r * ldg + 1], g[r * ldg + 2]};
        const float f[3] = {fea[r * ldf], fea[r * ldf + 1], fea[r * ldf + 2]};
        float z[2][3], h[2][3], sc[3], dc[3] = {0.f, 0.f, 0.f};
        mat3(p.w[0], p.b[0], c, z[0]);
        mat3(p.w[2], p.b[2], c, z[1]);
#pragma unroll
        for (int i = 0; i < 3; ++i) { h[0][i] = z[0][i] > 0.f ? z[0][i] : 0.1f * z[0][i]; h[1][i] = z[1][i] > 0.f ? z[1][i] : 0.1f * z[1][i]; }
        mat3(p.w[1], p.b[1], h[0], sc);
#pragma unroll
        for (int i = 0; i < 3; ++i) dfea[r * lddf + i] = gv[i] * (sc[i] + 1.f);
#pragma unroll
        for (int br = 0; br < 2; ++br) {                     // 0: scale branch (d scale = g * fea), 1: shift branch (d shift = g)
            float d1[3], dz[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) d1[i] = br == 0 ? gv[i] * f[i] : gv[i];
            float* a1 = acc + (br * 2 + 1) * 12;             // conv1 of this branch
#pragma unroll
            for (int o = 0; o < 3; ++o) {
#pragma unroll
                for (int k = 0; k < 3; ++k) a1[o * 3 + k] += d1[o] * h[br][k];
                a1[9 + o] += d1[o];
            }
            const float* w1 = p.w[br * 2 + 1];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float dh = w1[k] * d1[0] + w1[3 + k] * d1[1] + w1[6 + k] * d1[2];
                dz[k] = z[br][k] > 0.f ? dh : 0.1f * dh;
            }
            float* a0 = acc + (br * 2) * 12;                 // conv0 of this branch
#pragma unroll
            for (int o = 0; o < 3; ++o) {
#pragma unroll
                for (int k = 0; k < 3; ++k) a0[o * 3 + k] += dz[o] * c[k];
                a0[9 + o] += dz[o];
            }
            const float* w0 = p.w[br * 2];
#pragma unroll
            for (int k = 0; k < 3; ++k) dc[k] += w0[k] * dz[0] + w0[3 + k] * dz[1] + w0[6 + k] * dz[2];
        }
        if (dcond != nullptr) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dcond[r * lddc + k] = dc[k];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 48; ++i) {
        float v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 48) part[blockIdx.x * 48 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
struct Sft3Grads { float* w[4]; float* b[4]; };
__global__ void sft3_bwd_finish(const float* __restrict__ part, int nblk, const Sft3Grads d, int accumulate) {
    const int i = threadIdx.x;
    if (i >= 48) return;
    float v = 0.f;
    for (int b = 0; b < nblk; ++b) v += part[b * 48 + i];
    const int conv = i / 12, e = i - conv * 12;
    float* base = e < 9 ? d.w[conv] : d.b[conv];
    // accumulate: atomic, so two calls of the layer on different streams (sft0 serves both hands) cannot lose an update
    if (base != nullptr) { float* o = base + (e < 9 ? e : e - 9); if (accumulate) atomicAdd(o, v); else *o = v; }
}
// ws: 48 * 256 floats.  d*: the eight parameter gradients (any may be NULL), += when accumulate.  dcond may be NULL.
PDF_API int pdf_sft3_bwd(const float* g, int ldg, const float* fea, int ldf, const float* cond, int ldc,
                         const float* w_scale0, const float* b_scale0, const float* w_scale1, const float* b_scale1,
                         const float* w_shift0, const float* b_shift0, const float* w_shift1, const float* b_shift1,
                         float* dfea, int lddf, float* dcond, int lddc,
                         float* dw_scale0, float* db_scale0, float* dw_scale1, float* db_scale1,
                         float* dw_shift0, float* db_shift0, float* dw_shift1, float* db_shift1, int accumulate,
                         float* ws, long R, hipStream_t s) {
    if (R <= 0) return 0;
    Sft3Params p = {{w_scale0, w_scale1, w_shift0, w_shift1}, {b_scale0, b_scale1, b_shift0, b_shift1}};
    Sft3Grads d = {{dw_scale0, dw_scale1, dw_shift0, dw_shift1}, {db_scale0, db_scale1, db_shift0, db_shift1}};
    const int nblk = (int)min((long)SFT3_BLOCKS, (R + 255) / 256);
    hipLaunchKernelGGL(sft3_bwd_kernel, dim3(nblk), dim3(256), 0, s, g, ldg, fea, ldf, cond, ldc, p, dfea, lddf, dcond, lddc, ws, R);
    hipLaunchKernelGGL(sft3_bwd_finish, dim3(1), dim3(64), 0, s, ws, nblk, d, accumulate);
    PDF_LAUNCH_CHECK();
    return 0;
}

// dropout with a stateless mask: keep iff u(seed, i) >= p ; y = x * keep / (1-p).  Same call serves
// forward and backward (the mask is a pure function of seed and element index).
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float p, unsigned long long seed,
                               const unsigned long long* __restrict__ step) {
    const float sc = 1.f / (1.f - p);
    if (step != nullptr) seed += step[0] * 0x9E3779B97F4A7C15ull;      // per-step stream under hipGraph replay
    GRID_STRIDE(i, n) y[i] = pdf_uniform(seed, (unsigned long long)i) >= p ? x[i] * sc : 0.f;
}
PDF_API int pdf_dropout(const float* x, float* y, long n, float p, unsigned long long seed, const unsigned long long* step, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n, p, seed, step);
    PDF_LAUNCH_CHECK();
    return 0;
}
// y = res + dropout(x): the residual tail of MLP_res_block / SelfAttn (self_attn.py:31-33, 80-84).  Backward: d res = dy,
// dx = pdf_dropout(dy) with the same seed.
__global__ void dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ y, long n, float p,
                                   unsigned long long seed, const unsigned long long* __restrict__ step) {
    const float sc = 1.f / (1.f - p);
    if (step != nullptr) seed += step[0] * 0x9E3779B97F4A7C15ull;
    GRID_STRIDE(i, n) y[i] = res[i] + ((p <= 0.f || pdf_uniform(seed, (unsigned long long)i) >= p) ? x[i] * sc : 0.f);
}
PDF_API int pdf_dropout_add(const float* x, const float* res, float* y, long n, float p, unsigned long long seed,
                            const unsigned long long* step, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(dropout_add_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, res, y, n, p, seed, step);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(3, stride 2, pad 1) on NHWC (resnet.maxpool, intaghand_encoder.py:716)
__global__ void maxpool3s2_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C, int OH, int OW,
                                      float* __restrict__ y, unsigned char* __restrict__ arg, long total) {
    GRID_STRIDE(i, total) {
        int c = (int)(i % C); long p = i / C;
        int ox = (int)(p % OW); p /= OW;
        int oy = (int)(p % OH); int n = (int)(p / OH);
        float best = -INFINITY; int bi = 0;
        for (int ky = 0; ky < 3; ++ky) {
            int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                float v = x[(((long)n * H + iy) * W + ix) * C + c];
                if (v > best) { best = v; bi = ky * 3 + kx; }
            }
        }
        y[i] = best; arg[i] = (unsigned char)bi;
    }
}
// four channels per thread (C % 4 == 0): 9 float4 loads instead of 36 scalar ones per output quad
__global__ __launch_bounds__(256) void maxpool3s2_fwd_v4_kernel(const float* __restrict__ x, int N, int H, int W, int C, int OH, int OW,
                                                                 float* __restrict__ y, unsigned char* __restrict__ arg, long total4) {
    const int C4 = C >> 2;
    GRID_STRIDE(i, total4) {
        const int c = (int)(i % C4) * 4; long p = i / C4;
        const int ox = (int)(p % OW); p /= OW;
        const int oy = (int)(p % OH); const int n = (int)(p / OH);
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + (((long)n * H + iy) * W + ix) * C + c);
                const int t = ky * 3 + kx;
                if (v.x > best.x) { best.x = v.x; b0 = t; }
                if (v.y > best.y) { best.y = v.y; b1 = t; }
                if (v.z > best.z) { best.z = v.z; b2 = t; }
                if (v.w > best.w) { best.w = v.w; b3 = t; }
            }
        }
        *reinterpret_cast<float4*>(y + i * 4) = best;
        *reinterpret_cast<uchar4*>(arg + i * 4) = make_uchar4((unsigned char)b0, (unsigned char)b1, (unsigned char)b2, (unsigned char)b3);
    }
}
PDF_API int pdf_maxpool3s2_fwd(const float* x, int N, int H, int W, int C, float* y, unsigned char* arg, hipStream_t s) {
    int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    long total = (long)N * OH * OW * C;
    if (total <= 0) return 0;
    if (C % 4 == 0 && !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) && !(reinterpret_cast<uintptr_t>(arg) & 3))
        hipLaunchKernelGGL(maxpool3s2_fwd_v4_kernel, dim3(grid_for(total / 4)), dim3(256), 0, s, x, N, H, W, C, OH, OW, y, arg, total / 4);
    else
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, N, H, W, C, OH, OW, y, arg, total);
    PDF_LAUNCH_CHECK();
    return 0;
}
// dx must be zero-filled; windows overlap so scatter with atomics
__global__ void maxpool3s2_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg, int N, int H, int W, int C,
                                      int OH, int OW, float* __restrict__ dx, long total) {
    GRID_STRIDE(i, total) {
        int c = (int)(i % C); long p = i / C;
        int ox = (int)(p % OW); p /= OW;
        int oy = (int)(p % OH); int n = (int)(p / OH);
        int a = arg[i];
        int iy = oy * 2 - 1 + a / 3, ix = ox * 2 - 1 + a % 3;
        atomicAdd(&dx[(((long)n * H + iy) * W + ix) * C + c], dy[i]);
    }
}
// gather form: every INPUT element looks at the <= 4 windows that contain it and takes the gradient of those whose arg-max it
// is -- writes every dx element exactly once (no zero fill, no atomics: 134 MB of memset + 8.4 M atomic adds at B = 32 before)
__global__ __launch_bounds__(256) void maxpool3s2_bwd_gather_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg, int N, int H, int W, int C,
                                                                     int OH, int OW, float* __restrict__ dx, long total4, int accumulate) {
    const int C4 = C >> 2;
    GRID_STRIDE(i, total4) {
        const int c = (int)(i % C4) * 4; long p = i / C4;
        const int ix = (int)(p % W); p /= W;
        const int iy = (int)(p % H); const int n = (int)(p / H);
        float4 g = accumulate ? *reinterpret_cast<const float4*>(dx + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        // windows with oy * 2 - 1 + ky = iy, ky in 0..2:  oy in [ceil((iy - 1) / 2), floor((iy + 1) / 2)] = [iy / 2, (iy + 1) / 2]
        for (int oy = iy / 2; oy <= min(OH - 1, (iy + 1) / 2); ++oy) {
            const int ky = iy + 1 - 2 * oy;
            if (ky < 0 || ky > 2) continue;
            for (int ox = ix / 2; ox <= min(OW - 1, (ix + 1) / 2); ++ox) {
                const int kx = ix + 1 - 2 * ox;
                if (kx < 0 || kx > 2) continue;
                const long o = (((long)n * OH + oy) * OW + ox) * C + c;
                const uchar4 a = *reinterpret_cast<const uchar4*>(arg + o);
                const float4 d = *reinterpret_cast<const float4*>(dy + o);
                const unsigned char t = (unsigned char)(ky * 3 + kx);
                if (a.x == t) g.x += d.x;
                if (a.y == t) g.y += d.y;
                if (a.z == t) g.z += d.z;
                if (a.w == t) g.w += d.w;
            }
        }
        *reinterpret_cast<float4*>(dx + i * 4) = g;
    }
}
static int maxpool3s2_bwd(const float* dy, const unsigned char* arg, int N, int H, int W, int C, float* dx, int accumulate, hipStream_t s) {
    int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    long total = (long)N * OH * OW * C;
    if (total <= 0) return 0;
    if (C % 4 == 0 && !((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) && !(reinterpret_cast<uintptr_t>(arg) & 3)) {
        const long tin4 = (long)N * H * W * C / 4;          // (dx need not be zero-filled on this path; the caller's fill is harmless)
        hipLaunchKernelGGL(maxpool3s2_bwd_gather_kernel, dim3(grid_for(tin4)), dim3(256), 0, s, dy, arg, N, H, W, C, OH, OW, dx, tin4, accumulate);
        PDF_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, dy, arg, N, H, W, C, OH, OW, dx, total);
    PDF_LAUNCH_CHECK();
    return 0;
}
PDF_API int pdf_maxpool3s2_bwd(const float* dy, const unsigned char* arg, int N, int H, int W, int C, float* dx, hipStream_t s) {
    return maxpool3s2_bwd(dy, arg, N, H, W, C, dx, 0, s);
}
// dx += the same: dx already holds the gradient of the input's other consumers (the stem output also feeds the PointNet++ row
// gathers) -- one pass instead of a separate add over both tensors.  (The atomic fallback accumulates by construction.)
PDF_API int pdf_maxpool3s2_bwd_add(const float* dy, const unsigned char* arg, int N, int H, int W, int C, float* dx, hipStream_t s) {
    return maxpool3s2_bwd(dy, arg, N, H, W, C, dx, 1, s);
}

// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) on NHWC (intaghand_encoder.py:281-302)
__device__ __forceinline__ void up2_coords(int o, int in_sz, int out_sz, int& i0, int& i1, float& w1) {
    float src = out_sz > 1 ? (float)o * ((float)(in_sz - 1) / (float)(out_sz - 1)) : 0.f;
    i0 = (int)src;
    if (i0 > in_sz - 1) i0 = in_sz - 1;
    i1 = i0 + 1 < in_sz ? i0 + 1 : i0;
    w1 = src - (float)i0;
}
__global__ void up2_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C, float* __restrict__ y, long total) {
    const int OH = 2 * H, OW = 2 * W;
    GRID_STRIDE(i, total) {
        int c = (int)(i % C); long p = i / C;
        int ox = (int)(p % OW); p /= OW;
        int oy = (int)(p % OH); int n = (int)(p / OH);
        int y0, y1, x0, x1; float wy, wx;
        up2_coords(oy, H, OH, y0, y1, wy);
        up2_coords(ox, W, OW, x0, x1, wx);
        const float* b = x + (long)n * H * W * C + c;
        float v00 = b[((long)y0 * W + x0) * C], v01 = b[((long)y0 * W + x1) * C];
        float v10 = b[((long)y1 * W + x0) * C], v11 = b[((long)y1 * W + x1) * C];
        // same association as ATen's upsample_bilinear2d: h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)
        y[i] = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
    }
}
// float4 form: one thread per (output pixel, channel quad); grid.y = output row, so the only division is by C/4
__global__ __launch_bounds__(256) void up2_fwd_v4_kernel(const float* __restrict__ x, int H, int W, int C4, float* __restrict__ y) {
    const int OH = 2 * H, OW = 2 * W;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= OW * C4) return;
    const int ox = j / C4, c4 = j - ox * C4;
    const int n = blockIdx.y / OH, oy = blockIdx.y - n * OH;
    int y0, y1, x0, x1; float wy, wx;
    up2_coords(oy, H, OH, y0, y1, wy);
    up2_coords(ox, W, OW, x0, x1, wx);
    const float4* b = reinterpret_cast<const float4*>(x) + (long)n * H * W * C4 + c4;
    const float4 v00 = b[((long)y0 * W + x0) * C4], v01 = b[((long)y0 * W + x1) * C4];
    const float4 v10 = b[((long)y1 * W + x0) * C4], v11 = b[((long)y1 * W + x1) * C4];
    const float hy = 1.f - wy, hx = 1.f - wx;
    float4 o;
    o.x = hy * (hx * v00.x + wx * v01.x) + wy * (hx * v10.x + wx * v11.x);
    o.y = hy * (hx * v00.y + wx * v01.y) + wy * (hx * v10.y + wx * v11.y);
    o.z = hy * (hx * v00.z + wx * v01.z) + wy * (hx * v10.z + wx * v11.z);
    o.w = hy * (hx * v00.w + wx * v01.w) + wy * (hx * v10.w + wx * v11.w);
    reinterpret_cast<float4*>(y)[((long)blockIdx.y * OW + ox) * C4 + c4] = o;
}
static bool up2_v4_ok(const void* a, const void* b, int N, int H, int W, int C) {
    return C % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0 && (long)N * 2 * H < 65536 &&
           (long)2 * W * (C / 4) < (1L << 30);
}
PDF_API int pdf_upsample2x_fwd(const float* x, int N, int H, int W, int C, float* y, hipStream_t s) {
    long total = (long)N * 4 * H * W * C;
    if (total <= 0) return 0;
    if (up2_v4_ok(x, y, N, H, W, C)) {
        hipLaunchKernelGGL(up2_fwd_v4_kernel, dim3(cdiv(2 * W * (C / 4), 256), N * 2 * H), dim3(256), 0, s, x, H, W, C / 4, y);
        PDF_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(up2_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, N, H, W, C, y, total);
    PDF_LAUNCH_CHECK();
    return 0;
}
// scatter form (any C / alignment): dx is zeroed by the entry point
__global__ void up2_bwd_kernel(const float* __restrict__ dy, int N, int H, int W, int C, float* __restrict__ dx, long total) {
    const int OH = 2 * H, OW = 2 * W;
    GRID_STRIDE(i, total) {
        int c = (int)(i % C); long p = i / C;
        int ox = (int)(p % OW); p /= OW;
        int oy = (int)(p % OH); int n = (int)(p / OH);
        int y0, y1, x0, x1; float wy, wx;
        up2_coords(oy, H, OH, y0, y1, wy);
        up2_coords(ox, W, OW, x0, x1, wx);
        float g = dy[i];
        float* b = dx + (long)n * H * W * C + c;
        atomicAdd(&b[((long)y0 * W + x0) * C], g * (1.f - wy) * (1.f - wx));
        atomicAdd(&b[((long)y0 * W + x1) * C], g * (1.f - wy) * wx);
        atomicAdd(&b[((long)y1 * W + x0) * C], g * wy * (1.f - wx));
        atomicAdd(&b[((long)y1 * W + x1) * C], g * wy * wx);
    }
}
// Gather form of the backward (deterministic, no atomics, dx need not be zeroed): one thread per (input pixel, channel
// quad) sums the <= 5x5 output pixels whose bilinear footprint touches it, with the weights up2_coords gives the forward.
__device__ __forceinline__ int up2_touching(int i, int in_sz, int out_sz, int (&o_of)[6], float (&w_of)[6]) {
    int cnt = 0;
    const int lo = max(0, 2 * i - 2), hi = min(out_sz - 1, 2 * i + 4);
    for (int o = lo; o <= hi; ++o) {
        int i0, i1; float w1;
        up2_coords(o, in_sz, out_sz, i0, i1, w1);
        float w = 0.f;
        if (i0 == i) w += 1.f - w1;
        if (i1 == i) w += w1;
        if (w != 0.f && cnt < 6) { o_of[cnt] = o; w_of[cnt] = w; ++cnt; }
    }
    return cnt;
}
__global__ __launch_bounds__(256) void up2_bwd_v4_kernel(const float* __restrict__ dy, int H, int W, int C4, float* __restrict__ dx) {
    const int OH = 2 * H, OW = 2 * W;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= W * C4) return;
    const int ix = j / C4, c4 = j - ix * C4;
    const int n = blockIdx.y / H, iy = blockIdx.y - n * H;
    int oys[6], oxs[6]; float wys[6], wxs[6];
    const int ny = up2_touching(iy, H, OH, oys, wys), nx = up2_touching(ix, W, OW, oxs, wxs);
    const float4* g = reinterpret_cast<const float4*>(dy) + (long)n * OH * OW * C4 + c4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
            const float4 v = g[((long)oys[a] * OW + oxs[b]) * C4];
            const float w = wys[a] * wxs[b];
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        }
    reinterpret_cast<float4*>(dx)[((long)blockIdx.y * W + ix) * C4 + c4] = acc;
}
// the same for any channel count (the 2-channel mask head: the scatter form spent 190 us on 21 MB in atomics on 2 x 16 K addresses per image)
__global__ __launch_bounds__(256) void up2_bwd_gather_kernel(const float* __restrict__ dy, int H, int W, int C, float* __restrict__ dx) {
    const int OH = 2 * H, OW = 2 * W;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= W * C) return;
    const int ix = j / C, c = j - ix * C;
    const int n = blockIdx.y / H, iy = blockIdx.y - n * H;
    int oys[6], oxs[6]; float wys[6], wxs[6];
    const int ny = up2_touching(iy, H, OH, oys, wys), nx = up2_touching(ix, W, OW, oxs, wxs);
    const float* g = dy + (long)n * OH * OW * C + c;
    float acc = 0.f;
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) acc += wys[a] * wxs[b] * g[((long)oys[a] * OW + oxs[b]) * C];
    dx[((long)blockIdx.y * W + ix) * C + c] = acc;
}
PDF_API int pdf_upsample2x_bwd(const float* dy, int N, int H, int W, int C, float* dx, hipStream_t s) {
    long total = (long)N * 4 * H * W * C;
    if (total <= 0) return 0;
    if (up2_v4_ok(dy, dx, N, H, W, C)) {
        hipLaunchKernelGGL(up2_bwd_v4_kernel, dim3(cdiv(W * (C / 4), 256), N * H), dim3(256), 0, s, dy, H, W, C / 4, dx);
        PDF_LAUNCH_CHECK();
        return 0;
    }
    if ((long)N * H < 65536) {                               // (grid.y limit; beyond it the scatter form below)
        hipLaunchKernelGGL(up2_bwd_gather_kernel, dim3(cdiv(W * C, 256), N * H), dim3(256), 0, s, dy, H, W, C, dx);
        PDF_LAUNCH_CHECK();
        return 0;
    }
    { hipError_t e = hipMemsetAsync(dx, 0, (size_t)N * H * W * C * sizeof(float), s); if (e != hipSuccess) return (int)e; }
    hipLaunchKernelGGL(up2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, dy, N, H, W, C, dx, total);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// dst [R2][C2] (ldd) = src [R][C] (lds) in its top-left corner, zeros elsewhere -- and the same launch CROPS when R2 <= R, C2 <= C.  The
// PointNet++ 1x1 layers keep the reference's parameter shapes (131 / 259 input channels, `intaghand_encoder.py:48-103`) and hand the GEMMs
// matrices padded to 16-float rows: torch's constant_pad_nd is a fill plus a strided copy per pad and direction (56 pads = 112 launches per
// step); here a padded matrix is one launch forward and one (the crop of the gradient) backward.
__global__ void pad2d_kernel(const float* __restrict__ src, int lds, long R, int C, float* __restrict__ dst, int ldd, long R2, int C2) {
    const long total = R2 * C2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C2;
        const int c = (int)(i - r * C2);
        dst[r * ldd + c] = (r < R && c < C) ? src[r * lds + c] : 0.f;
    }
}
PDF_API int pdf_pad2d(const float* src, int lds, long R, int C, float* dst, int ldd, long R2, int C2, hipStream_t s) {
    if (R < 0 || C < 0 || R2 < 0 || C2 < 0 || lds < C || ldd < C2) return PDF_E_BADARG;
    if (R2 == 0 || C2 == 0) return 0;
    hipLaunchKernelGGL(pad2d_kernel, dim3(grid_for(R2 * C2)), dim3(256), 0, s, src, lds, R, C, dst, ldd, R2, C2);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam semantics, main.py:63) over one flat fp32 buffer.
// step_size / bias corrections are read from a 2-float device buffer so the launch is graph-replayable.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                            float lr, float b1, float b2, float eps, const float* __restrict__ corr /*[bc1, bc2]*/, float grad_scale) {
    const float bc1 = corr[0], bc2 = corr[1];
    const float step = lr / bc1;
    const float rs2 = 1.f / sqrtf(bc2);
    GRID_STRIDE(i, n) {
        float gv = g[i] * grad_scale;
        float mv = b1 * m[i] + (1.f - b1) * gv;
        float vv = b2 * v[i] + (1.f - b2) * gv * gv;
        m[i] = mv; v[i] = vv;
        p[i] -= step * mv / (sqrtf(vv) * rs2 + eps);
    }
}
PDF_API int pdf_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                          const float* corr, float grad_scale, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, corr, grad_scale);
    PDF_LAUNCH_CHECK();
    return 0;
}


// ---------------------------------------------------------------------------------------------
// ticket-counter ring of common.h (pdf_last_block_arrives)
#include <atomic>
#include <mutex>
int* pdf_ticket_counters(int n) {
    static int* base = nullptr;
    static std::once_flag once;
    static std::atomic<long> next{0};
    std::call_once(once, [] {
        if (hipMalloc(&base, sizeof(int) * PDF_COUNTER_RING) != hipSuccess) { base = nullptr; return; }
        if (hipMemset(base, 0, sizeof(int) * PDF_COUNTER_RING) != hipSuccess || hipDeviceSynchronize() != hipSuccess) base = nullptr;
    });
    if (base == nullptr || n <= 0 || n > PDF_COUNTER_RING) return nullptr;
    long o = next.fetch_add(n);
    long start = o % PDF_COUNTER_RING;
    if (start + n > PDF_COUNTER_RING) {                      // do not straddle the end: take the region at the start of the ring instead
        o = next.fetch_add(n + (PDF_COUNTER_RING - start)) + (PDF_COUNTER_RING - start);
        start = o % PDF_COUNTER_RING;
        if (start + n > PDF_COUNTER_RING) start = 0;
    }
    return base + start;
}
// split-K scratch ring of common.h (pdf_scratch)
float* pdf_scratch(long floats) {
    static float* base = nullptr;
    static std::once_flag once;
    static std::atomic<long> next{0};
    std::call_once(once, [] { if (hipMalloc(&base, sizeof(float) * PDF_SCRATCH_RING) != hipSuccess) base = nullptr; });
    floats = (floats + 63) & ~63L;                            // 256-byte granules
    if (base == nullptr || floats <= 0 || floats > PDF_SCRATCH_MAX) return nullptr;
    long o = next.fetch_add(floats);
    long start = o % PDF_SCRATCH_RING;
    if (start + floats > PDF_SCRATCH_RING) {
        o = next.fetch_add(floats + (PDF_SCRATCH_RING - start)) + (PDF_SCRATCH_RING - start);
        start = o % PDF_SCRATCH_RING;
        if (start + floats > PDF_SCRATCH_RING) start = 0;
    }
    return base + start;
}
// ---------------------------------------------------------------------------------------------
// bf16 shadows (bf16 mode): a producer that writes an fp32 tensor the GEMMs will read can write the same values rounded to bf16
// (RNE) beside it; the consumer passes that copy along and the bf16 GEMM kernels stage 2-byte operands -- half the L2 -> LDS bytes
// of the kernels that are bound by exactly those, results bit-identical to rounding while staging.  Carried through
// the explicit PdfCallOpts of the `_x` entry points; the pdf_set_* functions below arm thread-local slots for the plain entry points
// (set, then call, on the same thread), which take and clear ALL of them first thing.
static thread_local PdfCallOpts tl_opts = {};
static thread_local long tl_res_tiles = 0, tl_res_rows = 0;
PDF_API int pdf_set_bf16_operands(const void* op0_bf16, const void* op1_bf16) { tl_opts.op0_bf16 = op0_bf16; tl_opts.op1_bf16 = op1_bf16; return 0; }
PDF_API int pdf_set_bf16_output(void* out_bf16) { tl_opts.out_bf16 = out_bf16; return 0; }
PDF_API int pdf_set_bn_input_bf16(const void* x16) { tl_opts.bn_x_bf16 = x16; return 0; }
PDF_API int pdf_set_stats_output(float* part, long cap_floats) { tl_opts.stats_out = part; tl_opts.stats_cap = cap_floats; return 0; }
PDF_API int pdf_set_bn_tile_stats(const float* part, long tiles, long rows_per_tile) { tl_opts.tile_stats = part; tl_opts.tile_n = tiles; tl_opts.tile_rows = rows_per_tile; return 0; }
// BatchNorm + ReLU of an input tensor applied by the NEXT pdf_linear_fwd (its x) / pdf_linear_bwd_weight (its x) of this thread
PDF_API int pdf_set_input_affine_relu(const float* scale, const float* shift) { tl_opts.in_scale = scale; tl_opts.in_shift = shift; return 0; }
PDF_API long pdf_stats_result_tiles(void) { return tl_res_tiles; }
PDF_API long pdf_stats_result_rows(void) { return tl_res_rows; }
PdfCallOpts pdf_tls_take_all() { PdfCallOpts o = tl_opts; tl_opts = PdfCallOpts{}; o.stats_tiles = o.stats_rows = 0; tl_res_tiles = tl_res_rows = 0; return o; }
void pdf_tls_publish(const PdfCallOpts& o) { tl_res_tiles = o.stats_tiles; tl_res_rows = o.stats_rows; }
// how many hand-over slots of the calling thread are armed (tests: a rejected call must leave none)
PDF_API int pdf_debug_armed_slots(void) {
    const PdfCallOpts& o = tl_opts;
    return (o.op0_bf16 != nullptr) + (o.op1_bf16 != nullptr) + (o.out_bf16 != nullptr) + (o.bn_x_bf16 != nullptr) + (o.stats_out != nullptr) +
           (o.tile_stats != nullptr) + (o.in_scale != nullptr) + (o.in_shift != nullptr) + (o.op1_bf16_t != nullptr);
}
PDF_API int pdf_debug_callopts_size(void) { return (int)sizeof(PdfCallOpts); }
// dst[i] = bf16(src[i]) (RNE): the weight shadows, refreshed from the flat fp32 master buffer once per step
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        reinterpret_cast<uint2*>(dst)[i] = uint2{pdf_pk_bf16(v.x, v.y), pdf_pk_bf16(v.z, v.w)};
    }
}
PDF_API int pdf_cast_bf16(const float* src, void* dst, long n, hipStream_t s) {
    if (n <= 0) return 0;
    if (n % 4 != 0) return PDF_E_BADARG;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, src, reinterpret_cast<unsigned short*>(dst), n / 4);
    PDF_LAUNCH_CHECK();
    return 0;
}

// Transposed bf16 weight shadows (bf16 mode, round 4).  A backward-data contraction reads the weight as [K][N] rows -- an operand whose
// K runs ACROSS rows, which the LDS-DMA kernel (gemm_dma.hip) cannot take: its lanes fetch 16-byte chunks of K-contiguous rows.  So the
// trainer keeps, beside the bf16 copy of every large conv / linear weight, the same values TRANSPOSED: weight w[r][t][c] (r = output
// channel, t = tap, c = input channel; a linear layer has one tap) -> wt[c][t][r].  For the backward-data GEMM (rows = input channels,
// reduction over (tap, output channel)) that is an ordinary [N][K] row operand.  One launch transposes every listed tensor:
// table[l] = {src offset (floats), dst offset (elements), R, T, C, first tile}; tiles are 32 x 32 over (r, c) per tap.
struct PdfTDesc { long src, dst; int R, T, C, tile0; };
__global__ __launch_bounds__(256) void cast_bf16_t_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                          const PdfTDesc* __restrict__ tab, int nl) {
    __shared__ float tile[32][33];
    const int bid = blockIdx.x;
    int lo = 0, hi = nl - 1;                                 // the layer whose tile range holds this block (tile0 ascending)
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].tile0 <= bid) lo = mid; else hi = mid - 1; }
    const PdfTDesc d = tab[lo];
    const int tr = (d.R + 31) / 32, tc = (d.C + 31) / 32;
    int t = bid - d.tile0;
    const int tap = t / (tr * tc);
    t -= tap * tr * tc;
    const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* sp = src + d.src;
    unsigned short* dp = dst + d.dst;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < d.R && c < d.C) ? sp[((long)r * d.T + tap) * d.C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < d.R && c < d.C) dp[((long)c * d.T + tap) * d.R + r] = (unsigned short)(pdf_pk_bf16(tile[tx][ty + 8 * i], 0.f) & 0xffffu);
    }
}
PDF_API int pdf_cast_bf16_transposed(const float* src, void* dst, const void* table, int nlayers, long total_tiles, hipStream_t s) {
    if (nlayers <= 0 || total_tiles <= 0) return 0;
    if (total_tiles > 0x7fffffffL) return PDF_E_BADARG;
    hipLaunchKernelGGL(cast_bf16_t_kernel, dim3((unsigned)total_tiles), dim3(256), 0, s, src, reinterpret_cast<unsigned short*>(dst),
                       reinterpret_cast<const PdfTDesc*>(table), nlayers);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---- stream fork / join without Python objects (the host layer switches to a side stream ~180 times a step)
#define PDF_EVENT_RING 1024
static hipEvent_t g_events[PDF_EVENT_RING];
static std::atomic<unsigned> g_event_next{0};
static std::atomic<int> g_events_ready{0};
static int pdf_event_ring(void) {
    if (g_events_ready.load(std::memory_order_acquire)) return 0;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (g_events_ready.load(std::memory_order_relaxed)) return 0;
    for (int i = 0; i < PDF_EVENT_RING; ++i)
        if (hipEventCreateWithFlags(&g_events[i], hipEventDisableTiming) != hipSuccess) return PDF_E_WORKSPACE;
    g_events_ready.store(1, std::memory_order_release);
    return 0;
}
// `waiter` waits for everything `signaler` has been given so far.  A wait captures the event's latest record when it is
// issued, so re-recording a ring slot later does not disturb it.
PDF_API int pdf_stream_wait(hipStream_t waiter, hipStream_t signaler) {
    if (waiter == signaler) return 0;
    if (int rc = pdf_event_ring()) return rc;
    hipEvent_t e = g_events[g_event_next.fetch_add(1, std::memory_order_relaxed) % PDF_EVENT_RING];
    if (hipError_t rc = hipEventRecord(e, signaler)) return (int)rc;
    if (hipError_t rc = hipStreamWaitEvent(waiter, e, 0)) return (int)rc;
    return 0;
}

// A stream of the given HIP priority (hipDeviceGetStreamPriorityRange: -1 high ... 1 low on this runtime; torch.cuda.Stream offers only -1 / 0): the
// weight-gradient side streams can be created BELOW the main chain's priority (round 6 experiment, profiles/r06_wgrad_priority.txt).
// *lo / *hi (optional) receive the range.  The caller owns the stream (torch.cuda.ExternalStream) for the life of the process.
PDF_API int pdf_stream_create(void** out, int priority, int* lo, int* hi) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return PDF_E_WORKSPACE;
    if (lo) *lo = least;
    if (hi) *hi = greatest;
    if (out == nullptr) return 0;
    if (priority > least) priority = least;
    if (priority < greatest) priority = greatest;
    hipStream_t s = nullptr;
    if (hipError_t rc = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority)) return (int)rc;
    *out = (void*)s;
    return 0;
}

// The rings (ticket counters, split-K scratch, events) are process-wide and live on the device that was current at the first
// call: one process drives one GPU (one rank per GPU).  A later call from another current device is refused instead of handing
// kernels on GPU n the memory and events of GPU m.
static std::atomic<int> g_init_device{-1};
PDF_API int pdf_debug_init_device(void) { return g_init_device.load(); }
PDF_API int pdf_init(void) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return PDF_E_WORKSPACE;
    int expected = -1;
    if (!g_init_device.compare_exchange_strong(expected, dev) && expected != dev) return PDF_E_WORKSPACE;
    if (int rc = pdf_event_ring()) return rc;
    return (pdf_ticket_counters(1) != nullptr && pdf_scratch(64) != nullptr) ? 0 : PDF_E_WORKSPACE;
}
