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

// ---------------------------------------------------------------------------------------------
// Dense-map terms of CtdetLoss (lib/trains/simplified.py:368,374,376,391): SmoothL1 on the hand masks, MSE on the joint
// heat-maps and the CornerNet focal loss (lib/models/losses.py:138-165) on the clamped sigmoid (lib/models/utils.py:8-10) of
// the centre heat-map -- ~40 aten launches forward + backward in the reference, here two launches forward (partials +
// finalize) and one backward for all three terms.  Predictions are NHWC (the model's layout), targets NCHW (the
// dataset's): element (n, c, p) of a prediction sits at ((n*HW + p)*C + c), of a target at ((n*C + c)*HW + p).
struct DenseTerm { const float* pred; const float* tgt; float* dpred; int C, HW; };
struct DenseLoss { DenseTerm t[3]; int B, nblk; };            // t[0] mask (SmoothL1), t[1] hms (MSE), t[2] hm (focal)

__device__ __forceinline__ float clamp_sigmoid(float x) { return fminf(fmaxf(1.f / (1.f + expf(-x)), 1e-4f), 1.f - 1e-4f); }

// part[((term*B + b)*nblk + blk)*3 + {0,1,2}]: SmoothL1 / MSE: {sum, 0, 0}; focal: {pos_sum, neg_sum, num_pos}
__global__ __launch_bounds__(256) void dense_loss_partial_kernel(const DenseLoss a, float* __restrict__ part) {
    __shared__ float sm[4];
    const int term = blockIdx.z, b = blockIdx.y, blk = blockIdx.x;
    const DenseTerm t = a.t[term];
    const long per = (long)t.C * t.HW;
    const float* pt = t.tgt + b * per;
    const float* pp = t.pred + b * per;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (long i = blk * 256L + threadIdx.x; i < per; i += a.nblk * 256L) {
        const int c = (int)(i / t.HW), p = (int)(i - (long)c * t.HW);
        const float x = pp[(long)p * t.C + c], g = pt[i];
        if (term == 0) { const float d = fabsf(x - g); s0 += d < 1.f ? 0.5f * d * d : d - 0.5f; }
        else if (term == 1) { const float d = x - g; s0 += d * d; }
        else {
            const float q = clamp_sigmoid(x);
            if (g == 1.f) { s0 += logf(q) * (1.f - q) * (1.f - q); s2 += 1.f; }
            else if (g < 1.f) { const float w = (1.f - g) * (1.f - g); s1 += logf(1.f - q) * q * q * w * w; }
        }
    }
    s0 = block_sum_256(s0, sm);
    s1 = block_sum_256(s1, sm);
    s2 = block_sum_256(s2, sm);
    if (threadIdx.x == 0) {
        float* o = part + (((long)term * a.B + b) * a.nblk + blk) * 3;
        o[0] = s0; o[1] = s1; o[2] = s2;
    }
}
// out[0] = SmoothL1 mean, out[1] = MSE mean, out[2 + b] = focal loss of sample b, out[2 + B + b] = num_pos[b],
// out[2 + 2B] = 1 if the batch holds no positive at all (the reference's `if num_pos.sum() == 0` branch, losses.py:161)
__global__ __launch_bounds__(64) void dense_loss_finalize_kernel(const DenseLoss a, const float* __restrict__ part, float* __restrict__ out) {
    const int lane = threadIdx.x;
    for (int term = 0; term < 2; ++term) {
        float s = 0.f;
        for (int i = lane; i < a.B * a.nblk; i += 64) s += part[((long)term * a.B * a.nblk + i) * 3];
        s = wave_sum(s);
        if (lane == 0) out[term] = s / ((float)a.B * a.t[term].C * a.t[term].HW);
    }
    float tot = 0.f;
    for (int b = lane; b < a.B; b += 64)
        for (int k = 0; k < a.nblk; ++k) tot += part[(((long)2 * a.B + b) * a.nblk + k) * 3 + 2];
    tot = wave_sum(tot);
    for (int b = lane; b < a.B; b += 64) {
        float ps = 0.f, ns = 0.f, np = 0.f;
        for (int k = 0; k < a.nblk; ++k) {
            const float* p = part + (((long)2 * a.B + b) * a.nblk + k) * 3;
            ps += p[0]; ns += p[1]; np += p[2];
        }
        out[2 + b] = tot == 0.f ? -ns : -(ps + ns) / (np + 1e-3f);
        out[2 + a.B + b] = np;
    }
    if (lane == 0) out[2 + 2 * a.B] = tot == 0.f ? 1.f : 0.f;
}
PDF_API long pdf_dense_loss_workspace_floats(int B) { return 3L * B * 8 * 3; }
PDF_API int pdf_dense_loss_fwd(const float* mask, const float* mask_gt, int mask_c, int mask_hw,
                               const float* hms, const float* hms_gt, int hms_c, int hms_hw,
                               const float* hm, const float* hm_gt, int hm_c, int hm_hw, int B, float* ws, float* out, hipStream_t s) {
    if (B <= 0) return 0;
    DenseLoss a = {{{mask, mask_gt, nullptr, mask_c, mask_hw}, {hms, hms_gt, nullptr, hms_c, hms_hw}, {hm, hm_gt, nullptr, hm_c, hm_hw}}, B, 8};
    hipLaunchKernelGGL(dense_loss_partial_kernel, dim3(a.nblk, B, 3), dim3(256), 0, s, a, ws);
    hipLaunchKernelGGL(dense_loss_finalize_kernel, dim3(1), dim3(64), 0, s, a, ws, out);
    PDF_LAUNCH_CHECK();
    return 0;
}
// g0, g1: upstream gradients (one float each) of the two means; g2[b]: of the focal loss of sample b; stat = the forward's `out`
__global__ __launch_bounds__(256) void dense_loss_bwd_kernel(const DenseLoss a, const float* __restrict__ g0, const float* __restrict__ g1,
                                                             const float* __restrict__ g2, const float* __restrict__ stat) {
    const int term = blockIdx.z, b = blockIdx.y;
    const DenseTerm t = a.t[term];
    if (t.dpred == nullptr) return;
    const long per = (long)t.C * t.HW;
    const float* pt = t.tgt + b * per;
    const float* pp = t.pred + b * per;
    float* dp = t.dpred + b * per;
    float k;
    if (term < 2) k = (term == 0 ? g0[0] : g1[0]) / ((float)a.B * per);
    else k = stat[2 + 2 * a.B] != 0.f ? -g2[b] : -g2[b] / (stat[2 + a.B + b] + 1e-3f);
    const bool no_pos = term == 2 && stat[2 + 2 * a.B] != 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < per; i += gridDim.x * 256L) {
        const int c = (int)(i / t.HW), p = (int)(i - (long)c * t.HW);
        const long o = (long)p * t.C + c;
        const float x = pp[o], gt = pt[i];
        float d;
        if (term == 0) { const float e = x - gt; d = k * (fabsf(e) < 1.f ? e : (e > 0.f ? 1.f : -1.f)); }
        else if (term == 1) d = k * 2.f * (x - gt);
        else {
            const float sgm = 1.f / (1.f + expf(-x));
            const float q = fminf(fmaxf(sgm, 1e-4f), 1.f - 1e-4f);
            const float dq = (sgm >= 1e-4f && sgm <= 1.f - 1e-4f) ? sgm * (1.f - sgm) : 0.f;      // clamp passes gradient inside [min, max]
            float dl = 0.f;
            if (gt == 1.f) dl = no_pos ? 0.f : (1.f - q) * (1.f - q) / q - 2.f * (1.f - q) * logf(q);
            else if (gt < 1.f) { const float w = (1.f - gt) * (1.f - gt); dl = w * w * (2.f * q * logf(1.f - q) - q * q / (1.f - q)); }
            d = k * dl * dq;
        }
        dp[o] = d;
    }
}
PDF_API int pdf_dense_loss_bwd(const float* mask, const float* mask_gt, float* dmask, int mask_c, int mask_hw,
                               const float* hms, const float* hms_gt, float* dhms, int hms_c, int hms_hw,
                               const float* hm, const float* hm_gt, float* dhm, int hm_c, int hm_hw, int B,
                               const float* g_mask, const float* g_hms, const float* g_hm, const float* stat, hipStream_t s) {
    if (B <= 0) return 0;
    DenseLoss a = {{{mask, mask_gt, g_mask ? dmask : nullptr, mask_c, mask_hw}, {hms, hms_gt, g_hms ? dhms : nullptr, hms_c, hms_hw},
                    {hm, hm_gt, g_hm ? dhm : nullptr, hm_c, hm_hw}}, B, 8};
    hipLaunchKernelGGL(dense_loss_bwd_kernel, dim3(32, B, 3), dim3(256), 0, s, a, g_mask, g_hms, g_hm, stat);
    PDF_LAUNCH_CHECK();
    return 0;
}

// Evaluation metric (lib/trains/base_trainer.py:263-323): out[r] = sum over the n points of row r of ||pred - gt||_2
// (dim = 3: joints / vertices in metres, dim = 2: landmarks in pixels).  One block per row = (sample, hand).
__global__ __launch_bounds__(256) void point_dist_sum_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int n, int dim,
                                                             float* __restrict__ out) {
    __shared__ float sm[4];
    const float* p = pred + (long)blockIdx.x * n * dim;
    const float* q = gt + (long)blockIdx.x * n * dim;
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        float s = 0.f;
        for (int k = 0; k < dim; ++k) { const float d = p[i * dim + k] - q[i * dim + k]; s += d * d; }
        a += sqrtf(s);
    }
    a = block_sum_256(a, sm);
    if (threadIdx.x == 0) out[blockIdx.x] = a;
}
PDF_API int pdf_point_dist_sum(const float* pred, const float* gt, int rows, int n, int dim, float* out, hipStream_t s) {
    if (rows <= 0 || n <= 0) return 0;
    if (dim < 1 || dim > 4) return PDF_E_BADARG;
    hipLaunchKernelGGL(point_dist_sum_kernel, dim3(rows), dim3(256), 0, s, pred, gt, n, dim, out);
    PDF_LAUNCH_CHECK();
    return 0;
}
