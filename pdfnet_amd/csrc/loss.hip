// Loss kernels of the mesh supervision (lib/trains/simplified.py:66-115,425-525): the reference builds every term from
// dozens of tiny elementwise launches per hand; here each term is one forward and one backward launch over BOTH hands.
#include "common.h"

__device__ __forceinline__ float block_sum_256(float v, float* sm /*[4]*/) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// out[r] = mean_i f(pred[r][i] - tgt[r][i]),  f = |.| (mode 0: F.l1_loss(reduction='none').mean(-1)) or (.)^2 (mode 1: F.mse_loss)
__global__ __launch_bounds__(256) void rowloss_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, long n, int mode,
                                                          float* __restrict__ out) {
    __shared__ float sm[4];
    const float* p = pred + blockIdx.x * n;
    const float* t = tgt + blockIdx.x * n;
    float a = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float d = p[i] - t[i];
        a += mode == 0 ? fabsf(d) : d * d;
    }
    a = block_sum_256(a, sm);
    if (threadIdx.x == 0) out[blockIdx.x] = a / (float)n;
}
PDF_API int pdf_rowloss_fwd(const float* pred, const float* tgt, long rows, long n, int mode, float* out, hipStream_t s) {
    if (rows <= 0 || n <= 0) return 0;
    hipLaunchKernelGGL(rowloss_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, s, pred, tgt, n, mode, out);
    PDF_LAUNCH_CHECK();
    return 0;
}
// dpred[r][i] = gout[r] / n * f'(pred - tgt),  f' = sign (0 at 0, like torch) or 2 (.)
__global__ void rowloss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ gout, long n, int mode,
                                   float* __restrict__ dpred, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float d = pred[i] - tgt[i];
        const float g = gout[i / n] / (float)n;
        dpred[i] = mode == 0 ? (d > 0.f ? g : (d < 0.f ? -g : 0.f)) : 2.f * d * g;
    }
}
PDF_API int pdf_rowloss_bwd(const float* pred, const float* tgt, const float* gout, long rows, long n, int mode, float* dpred, hipStream_t s) {
    if (rows <= 0 || n <= 0) return 0;
    hipLaunchKernelGGL(rowloss_bwd_kernel, dim3(grid_for(rows * n)), dim3(256), 0, s, pred, tgt, gout, n, mode, dpred, rows * n);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Face terms.  For face (i0,i1,i2) of sample (g, b), u(.) = x / max(|x|, 1e-12) (F.normalize):
//   normal (simplified.py:66-91): n = u(u(g1-g0) x u(g2-g0)) from the ground truth;
//                                 sum_k |u(v_k) . n|, v = (p1-p0, p2-p0, p2-p1)
//   edge   (simplified.py:94-115): sum over (0,1),(0,2),(1,2) of | |p_i - p_j| - |g_i - g_j| |
// part[(g*B + b)*2 + {0,1}] = the two sums over the faces of one sample (the caller divides by B*3F for the means).
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 ld3(const float* p, long i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float len3(V3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ V3 unit3(V3 a, float& l) { l = fmaxf(len3(a), 1e-12f); return {a.x / l, a.y / l, a.z / l}; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

__global__ __launch_bounds__(256) void face_loss_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                            const long long* __restrict__ faces, int B, int V, int Fc,
                                                            float* __restrict__ part) {
    __shared__ float sm[4];
    const int g = blockIdx.x / B;
    const float* p = pred + (long)blockIdx.x * V * 3;
    const float* q = gt + (long)blockIdx.x * V * 3;
    const long long* fc = faces + (long)g * Fc * 3;
    float an = 0.f, ae = 0.f;
    for (int f = threadIdx.x; f < Fc; f += 256) {
        const long i0 = fc[3 * f], i1 = fc[3 * f + 1], i2 = fc[3 * f + 2];
        const V3 p0 = ld3(p, i0), p1 = ld3(p, i1), p2 = ld3(p, i2);
        const V3 g0 = ld3(q, i0), g1 = ld3(q, i1), g2 = ld3(q, i2);
        float l;
        const V3 n = unit3(cross3(unit3(sub3(g1, g0), l), unit3(sub3(g2, g0), l)), l);
        const V3 v1 = sub3(p1, p0), v2 = sub3(p2, p0), v3 = sub3(p2, p1);
        an += fabsf(dot3(unit3(v1, l), n)) + fabsf(dot3(unit3(v2, l), n)) + fabsf(dot3(unit3(v3, l), n));
        ae += fabsf(len3(v1) - len3(sub3(g0, g1))) + fabsf(len3(v2) - len3(sub3(g0, g2))) + fabsf(len3(v3) - len3(sub3(g1, g2)));
    }
    an = block_sum_256(an, sm);
    ae = block_sum_256(ae, sm);
    if (threadIdx.x == 0) { part[blockIdx.x * 2] = an; part[blockIdx.x * 2 + 1] = ae; }
}
PDF_API int pdf_face_loss_fwd(const float* pred, const float* gt, const long long* faces, int G, int B, int V, int Fc, float* part, hipStream_t s) {
    if (G * B <= 0) return 0;
    hipLaunchKernelGGL(face_loss_fwd_kernel, dim3(G * B), dim3(256), 0, s, pred, gt, faces, B, V, Fc, part);
    PDF_LAUNCH_CHECK();
    return 0;
}

// dpred of  sum_g ( wn[g] * normal_sum(g) + we[g] * edge_sum(g) )  (wn / we already hold the upstream gradients and the 1/(B*3F)
// of the means).  One block owns one sample: the vertex gradients are accumulated in LDS and written once.
#define FACE_MAXV 1024
__device__ __forceinline__ void acc3(float* s, long i, V3 v, float w) {
    atomicAdd(&s[3 * i], w * v.x); atomicAdd(&s[3 * i + 1], w * v.y); atomicAdd(&s[3 * i + 2], w * v.z);
}
__global__ __launch_bounds__(256) void face_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                            const long long* __restrict__ faces, int B, int V, int Fc,
                                                            const float* __restrict__ wn, const float* __restrict__ we,
                                                            float* __restrict__ dpred) {
    __shared__ float acc[FACE_MAXV * 3];
    const int g = blockIdx.x / B;
    const float* p = pred + (long)blockIdx.x * V * 3;
    const float* q = gt + (long)blockIdx.x * V * 3;
    const long long* fc = faces + (long)g * Fc * 3;
    for (int i = threadIdx.x; i < V * 3; i += 256) acc[i] = 0.f;
    __syncthreads();
    const float cn = wn[g], ce = we != nullptr ? we[g] : 0.f;
    for (int f = threadIdx.x; f < Fc; f += 256) {
        const long idx[3] = {(long)fc[3 * f], (long)fc[3 * f + 1], (long)fc[3 * f + 2]};
        const V3 pv[3] = {ld3(p, idx[0]), ld3(p, idx[1]), ld3(p, idx[2])};
        const V3 g0 = ld3(q, idx[0]), g1 = ld3(q, idx[1]), g2 = ld3(q, idx[2]);
        float l;
        const V3 n = unit3(cross3(unit3(sub3(g1, g0), l), unit3(sub3(g2, g0), l)), l);
        const V3 gv[3] = {g0, g1, g2};
        const int ea[3] = {0, 0, 1}, eb[3] = {1, 2, 2};          // v_k = p[eb] - p[ea]; edge pairs (0,1),(0,2),(1,2)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const V3 v = sub3(pv[eb[k]], pv[ea[k]]);
            float lv;
            const V3 u = unit3(v, lv);
            const float c = dot3(u, n);
            // d|u.n|/dv = sign(u.n) (n - u (u.n)) / |v|
            const float sg = c > 0.f ? 1.f : (c < 0.f ? -1.f : 0.f);
            const float kn = cn * sg / lv;
            V3 d = {kn * (n.x - u.x * c), kn * (n.y - u.y * c), kn * (n.z - u.z * c)};
            if (ce != 0.f) {
                // d| |v| - |g| |/dv = sign(|v| - |g|) v / |v|
                const float lp = len3(v), lg = len3(sub3(gv[ea[k]], gv[eb[k]]));
                const float ke = lp > 0.f ? ce * (lp > lg ? 1.f : (lp < lg ? -1.f : 0.f)) / lp : 0.f;
                d.x += ke * v.x; d.y += ke * v.y; d.z += ke * v.z;
            }
            acc3(acc, idx[eb[k]], d, 1.f);
            acc3(acc, idx[ea[k]], d, -1.f);
        }
    }
    __syncthreads();
    float* o = dpred + (long)blockIdx.x * V * 3;
    for (int i = threadIdx.x; i < V * 3; i += 256) o[i] = acc[i];
}
PDF_API int pdf_face_loss_bwd(const float* pred, const float* gt, const long long* faces, int G, int B, int V, int Fc,
                              const float* wn, const float* we, float* dpred, hipStream_t s) {
    if (V > FACE_MAXV) return PDF_E_BADARG;
    if (G * B <= 0) return 0;
    hipLaunchKernelGGL(face_loss_bwd_kernel, dim3(G * B), dim3(256), 0, s, pred, gt, faces, B, V, Fc, wn, we, dpred);
    PDF_LAUNCH_CHECK();
    return 0;
}
