// Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions with many channels (fp32 mode): `feat` (1024 -> 256 at 64x64, a fifth of the
// step's multiply-adds), p2 and the first convolution of the hm / wh / params heads (256 -> 256) -- forward and backward-data.
//
// Why.  Every dense contraction of the step already runs on the matrix pipe at 75-83 % of its fp32 peak; what is left
// is to do fewer multiplications.  A 2x2 output tile of a 3x3 convolution costs 36 multiply-adds per (input, output) channel pair
// directly and 16 in the Winograd domain: 2.25x fewer MFMA instructions for the price of three streaming passes,
//     V[xi][t][c] = (B^T d B)[xi]      input transform of every 4x4 patch d (stride 2) -- 16 planes, 4x the input's bytes
//     M[xi][t][n] = sum_c V[xi][t][c] U[xi][n][c]      16 plain GEMMs [T x C] x [N x C]^T in ONE batched launch (gemm.hip)
//     y tile      = A^T M A + bias, activation          output transform
// with U = G g G^T the transformed weights (16 planes of [N][C], rebuilt per call: the weights change every step).  The transform
// matrices of F(2x2, 3x3) hold only 0, +-1 and 1/2, so the result differs from the direct sum by a few fp32 roundings (the parity bars
// of the tests are unchanged).  reference layers: intaghand_encoder.py:617 (feat), :602 (p2), :675-693 (heads).
//
// Backward-data is the same convolution of dy with the taps mirrored and the channel roles swapped: only the weight transform differs
// (U'[xi][ci][co] from w[co][2 - ky][2 - kx][ci]).
#include "common.h"
#include <cstdio>

int pdf_internal_batched_gemm(const float* A, const float* B, float* C, int batch, long gsA, long gsB, long gsC, int M, int N, int K, hipStream_t s);

// ---- weights: w [Cout][3][3][Cin] -> U [16][N][K]; forward: N = Cout, K = Cin; backward-data (FLIP): N = Cin, K = Cout, taps mirrored
template <bool FLIP>
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin) {
    const int N = FLIP ? Cin : Cout, K = FLIP ? Cout : Cin;
    const long total = (long)N * K;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i / K), k = (int)(i - (long)n * K);
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                g[a][b] = FLIP ? w[(((long)k * 3 + (2 - a)) * 3 + (2 - b)) * Cin + n] : w[(((long)n * 3 + a) * 3 + b) * Cin + k];
        float t[4][3];                                       // G g
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t[0][b] = g[0][b];
            t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            t[3][b] = g[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {                        // (G g) G^T
            const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
            U[(long)(a * 4 + 0) * total + i] = u0;
            U[(long)(a * 4 + 1) * total + i] = u1;
            U[(long)(a * 4 + 2) * total + i] = u2;
            U[(long)(a * 4 + 3) * total + i] = u3;
        }
    }
}

// ---- input: x [N][H][W][C] (row stride ldx) -> V [16][T][C], T = N (H/2) (W/2); one thread = 4 channels of one tile
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int ldx, float* __restrict__ V, int N, int H, int W, int C) {
    const int C4 = C >> 2, TW = W >> 1, TH = H >> 1;
    const long T = (long)N * TH * TW, total = T * C4;
    const long plane = T * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long t = i / C4;
        const int c = (int)(i - t * C4) * 4;
        const int n = (int)(t / (TH * TW)), r = (int)(t - (long)n * TH * TW), ty = r / TW, tx = r - ty * TW;
        float4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int iy = 2 * ty - 1 + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ix = 2 * tx - 1 + b;
                d[a][b] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? *reinterpret_cast<const float4*>(x + (((long)n * H + iy) * W + ix) * ldx + c)
                                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // B^T d: rows (d0 - d2), (d1 + d2), (d2 - d1), (d1 - d3); then the same on the columns
        float4 s[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#define WSUB(p, q) make_float4(p.x - q.x, p.y - q.y, p.z - q.z, p.w - q.w)
#define WADD(p, q) make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w)
            s[0][b] = WSUB(d[0][b], d[2][b]);
            s[1][b] = WADD(d[1][b], d[2][b]);
            s[2][b] = WSUB(d[2][b], d[1][b]);
            s[3][b] = WSUB(d[1][b], d[3][b]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 v0 = WSUB(s[a][0], s[a][2]), v1 = WADD(s[a][1], s[a][2]), v2 = WSUB(s[a][2], s[a][1]), v3 = WSUB(s[a][1], s[a][3]);
            float* o = V + t * C + c;
            *reinterpret_cast<float4*>(o + (long)(a * 4 + 0) * plane) = v0;
            *reinterpret_cast<float4*>(o + (long)(a * 4 + 1) * plane) = v1;
            *reinterpret_cast<float4*>(o + (long)(a * 4 + 2) * plane) = v2;
            *reinterpret_cast<float4*>(o + (long)(a * 4 + 3) * plane) = v3;
        }
    }
}

// ---- output: M [16][T][Co] -> y [N][H][W][Co] (row stride ldy): y tile = A^T m A (+ bias, activation; += y when accum)
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mx, const float* __restrict__ bias, float* __restrict__ y, int ldy,
                                                          int N, int H, int W, int Co, int act, int accum) {
    const int C4 = Co >> 2, TW = W >> 1, TH = H >> 1;
    const long T = (long)N * TH * TW, total = T * C4;
    const long plane = T * Co;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long t = i / C4;
        const int c = (int)(i - t * C4) * 4;
        const int n = (int)(t / (TH * TW)), r = (int)(t - (long)n * TH * TW), ty = r / TW, tx = r - ty * TW;
        float4 m[4][4];
        const float* p = Mx + t * Co + c;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) m[a][b] = *reinterpret_cast<const float4*>(p + (long)(a * 4 + b) * plane);
        float4 s[2][4];                                      // A^T m: rows (m0 + m1 + m2), (m1 - m2 - m3)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            s[0][b] = WADD(WADD(m[0][b], m[1][b]), m[2][b]);
            s[1][b] = WSUB(WSUB(m[1][b], m[2][b]), m[3][b]);
        }
        const float4 bv = bias != nullptr ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float4 o[2];
            o[0] = WADD(WADD(s[a][0], s[a][1]), s[a][2]);
            o[1] = WSUB(WSUB(s[a][1], s[a][2]), s[a][3]);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float4 v = WADD(o[b], bv);
                if (act == 1) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                else if (act == 2) v = make_float4(v.x > 0.f ? v.x : 0.1f * v.x, v.y > 0.f ? v.y : 0.1f * v.y, v.z > 0.f ? v.z : 0.1f * v.z, v.w > 0.f ? v.w : 0.1f * v.w);
                float* q = y + (((long)n * H + 2 * ty + a) * W + 2 * tx + b) * ldy + c;
                if (accum) { const float4 e = *reinterpret_cast<const float4*>(q); v = WADD(v, e); }
                *reinterpret_cast<float4*>(q) = v;
            }
        }
    }
}
#undef WSUB
#undef WADD

// floats of workspace a call needs: U + V + M
long pdf_internal_wino_workspace(int N, int H, int W, int Ck, int Cn) {
    const long T = (long)N * (H / 2) * (W / 2);
    return 16L * Cn * Ck + 16L * T * Ck + 16L * T * Cn;
}
// Is this convolution taken by the Winograd path?  (3x3, stride 1, pad 1, even map, channel counts the fast GEMM tiles like, enough work)
int pdf_internal_wino_eligible(int N, int H, int W, int Ck, int Cn, int KH, int KW, int stride, int pad) {
    static const int on = getenv("PDF_WINOGRAD") ? atoi(getenv("PDF_WINOGRAD")) : 1;
    static const int minc = getenv("PDF_WINOGRAD_MINC") ? atoi(getenv("PDF_WINOGRAD_MINC")) : 128;
    if (!on || KH != 3 || KW != 3 || stride != 1 || pad != 1 || (H & 1) || (W & 1)) return 0;
    if (Ck % 16 != 0 || Cn % 16 != 0 || Ck < minc || Cn < 64) return 0;
    const long T = (long)N * (H / 2) * (W / 2);
    if (T < 4096) return 0;                                  // (the 16-batch GEMM must fill the chip)
    if (16.0 * T * (Ck > Cn ? Ck : Cn) * 4.0 > 4.0e9) return 0;      // a transform-domain plane set beyond 4 GB: leave it to the direct kernel
    return 1;
}
// x [N][H][W][Ck] (ldx) * w -> y [N][H][W][Cn] (ldy).  flip = 0: forward, w = [Cn][3][3][Ck]; flip = 1: backward-data, w = [Ck][3][3][Cn]
// (x = dy, y = dx).  ws: pdf_internal_wino_workspace(N, H, W, Ck, Cn) floats.
int pdf_internal_conv3x3_winograd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, float* ws,
                                  int N, int H, int W, int Ck, int Cn, int act, int accum, int flip, hipStream_t s) {
    const long T = (long)N * (H / 2) * (W / 2);
    float* U = ws;
    float* V = U + 16L * Cn * Ck;
    float* Mx = V + 16L * T * Ck;
    const int gw = grid_for((long)Cn * Ck);
    if (flip) hipLaunchKernelGGL((wino_weight_kernel<true>), dim3(gw), dim3(256), 0, s, w, U, Ck, Cn);        // w [Cout = Ck][3][3][Cin = Cn]
    else hipLaunchKernelGGL((wino_weight_kernel<false>), dim3(gw), dim3(256), 0, s, w, U, Cn, Ck);
    hipLaunchKernelGGL(wino_input_kernel, dim3(grid_for(T * (Ck / 4), 256, 256 * 32)), dim3(256), 0, s, x, ldx, V, N, H, W, Ck);
    PDF_LAUNCH_CHECK();
    if (int rc = pdf_internal_batched_gemm(V, U, Mx, 16, T * Ck, (long)Cn * Ck, T * Cn, (int)T, Cn, Ck, s)) return rc;
    hipLaunchKernelGGL(wino_output_kernel, dim3(grid_for(T * (Cn / 4), 256, 256 * 32)), dim3(256), 0, s, Mx, bias, y, ldy, N, H, W, Cn, act, accum);
    PDF_LAUNCH_CHECK();
    return 0;
}
