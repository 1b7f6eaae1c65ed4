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

// the same for a block of ML_T threads (sm [ML_T / 64]): the wave partials summed in a fixed order
template <int NW>
__device__ __forceinline__ float block_sum_waves(float v, float* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += sm[w];
    return s;
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
    // index order (pixel tile of 64, channel, pixel in tile): a wave reads 64 consecutive target pixels of one channel (NCHW) and the 64 prediction
    // rows (NHWC) its neighbours on the channel axis read next -- with the plain (channel, pixel) order every prediction line was fetched
    // for ONE element and again, from L2, for each of the other channels (62 + 85 us for 20 MB of maps)
    for (long i = blk * 256L + threadIdx.x; i < per; i += a.nblk * 256L) {
        int c, p;
        if ((t.HW & 63) == 0) { const long tile = i / (64L * t.C); const int rem = (int)(i - tile * 64L * t.C); c = rem >> 6; p = (int)(tile * 64) + (rem & 63); }
        else { c = (int)(i / t.HW); p = (int)(i - (long)c * t.HW); }
        const float x = pp[(long)p * t.C + c], g = pt[(long)c * t.HW + p];
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
        int c, p;
        if ((t.HW & 63) == 0) { const long tile = i / (64L * t.C); const int rem = (int)(i - tile * 64L * t.C); c = rem >> 6; p = (int)(tile * 64) + (rem & 63); }
        else { c = (int)(i / t.HW); p = (int)(i - (long)c * t.HW); }
        const long o = (long)p * t.C + c;
        const float x = pp[o], gt = pt[(long)c * t.HW + p];
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

// ---------------------------------------------------------------------------------------------
// Round 5: ALL mesh terms of CtdetLoss.forward's train branch (lib/trains/simplified.py:425-525, lib/models/losses.py:26-94) in one forward
// launch pair and one backward launch.  One block per (hand, sample) computes every per-row quantity of that hand: the L1 / MSE terms on the
// 778- and 252-vertex meshes, the 21 regressed joints (Mano_model.py:309-323 full regressor, dense [21][V]), normal + edge-length terms over
// the 1538 faces (:66-115), the root un-projection (Mano_render.py:211-223), the pinhole projection of the joints (:203-209), the 2-D joint
// term and the 20-bone direction term, and the GCN-level supervision against the ground truth pooled 1008 -> 252 in GCN order (:461-482,
// incl. the reference's use of the LEFT ground truth and valid[:, 0] for both hands).  The unfused path spent 1.15 ms forward and ~1.8 ms
// backward of the step's critical chain on ~250 launches of 2-4 us (rowloss / face_loss kernels plus aten index bookkeeping).
// part[(g * B + b) * MLP + k]: k = 0 sum (v2p - v2gt)^2, 1 mean |vp - vgt_off|, 2 mean |jp_off - jg_off|, 3 normal sum, 4 edge sum,
// 5 mean |hd3 - g3|, 6 sum (hd2 - g2)^2, 7 mean |root_pred - root_gt|, 8 mean |jp - jgt|, 9 mean |vpred - vgt|, 10 sum (lms - lmsgt)^2, 11 bone term
#define MLP 12
// threads per (hand, sample) block: 1,024 -- the 126 length-778 dot products of the joint regression take one wave-turn each, a dependent chain per
// turn: 16 waves make 8 turns of what 4 waves made 32 (round 6: forward 133 -> 88 us, backward 140 -> 76 us; profiles/r06_mesh_loss_threads.txt)
#define ML_T 1024
#define ML_V 778
#define ML_VG 252
#define ML_J 21
struct MeshLoss {
    const float* vp; const float* v2p; const float* hd3; const float* hd2; const float* r;              // predictions [2][B][...]
    const float* vgt[2]; const float* jgt[2]; const float* v2gt[2]; const float* lmsgt[2];              // ground truth per hand [B][...] (the batch's own tensors)
    const long long* ind; const float* K; const float* valid;                                          // [B][2] i64, [B][3][3], [B][2]
    const float* reg[2]; const long long* faces; const long long* perm[2];                              // [21][778] x 2, [2][F][3], [1008] x 2
    int B, Fc, size, down;
    float* part;                                                                                        // [2][B][MLP]
    float* out;                                                                                         // forward: see mesh_loss_finalize_kernel
    float coef[12];                                                                                     // weights of the reference's sum (:610-640), order of `out`
    // backward
    const float* gmp;                                                                                   // [B] upstream gradient of the weighted sum out[4 + 8 B ..]
    int edge_grad;
    float* dvp; float* dv2p; float* dhd3; float* dhd2; float* dr;
};
__constant__ int ml_bone_a[20] = {0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 0, 13, 14, 15, 0, 17, 18, 19};
__constant__ int ml_bone_c[20] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20};

// shared per-block geometry: regressed joints (offset frame) of prediction and ground truth, root, absolute joints, landmarks
struct MlShared {
    float jp[ML_J * 3], jg[ML_J * 3], lms[ML_J * 2], root_pred[3], root_gt[3], ax, ay, z;
};
__device__ __forceinline__ void ml_common(const MeshLoss& a, int g, int b, MlShared& S) {
    const long gb = (long)g * a.B + b;
    const float* vp = a.vp + gb * ML_V * 3;
    const float* vg = a.vgt[g] + (long)b * ML_V * 3;
    const float* jgt = a.jgt[g] + (long)b * ML_J * 3;
    if (threadIdx.x < 3) S.root_gt[threadIdx.x] = jgt[9 * 3 + threadIdx.x];
    __syncthreads();
    // joints = reg . verts: (21 x 3) x 2 dot products of length 778, one per wave-turn
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = wave; o < ML_J * 3 * 2; o += ML_T / 64) {
        const int which = o / (ML_J * 3), jk = o - which * ML_J * 3, j = jk / 3, k = jk - 3 * j;
        const float* rg = a.reg[g] + (long)j * ML_V;
        const float* src = which ? vg : vp;
        float s = 0.f;
        for (int v = lane; v < ML_V; v += 64) s += rg[v] * (which ? src[v * 3 + k] - S.root_gt[k] : src[v * 3 + k]);
        s = wave_sum(s);
        if (lane == 0) (which ? S.jg : S.jp)[jk] = s;
    }
    if (threadIdx.x == 0) {
        // root_pred (Mano_render.py:211-223)
        const float* r = a.r + gb * 3;
        const float* K = a.K + (long)b * 9;
        const int gsz = a.size / a.down;
        const long long idx = a.ind[(long)b * 2 + g];
        const float cx = (float)((idx % gsz) * a.down), cy = (float)((idx / gsz) * a.down);
        const float z = 0.4f + r[0] / 100.f;
        S.ax = (r[1] / 100.f + cx - K[2]) / (K[0] + 1e-7f);
        S.ay = (r[2] / 100.f + cy - K[5]) / (K[4] + 1e-7f);
        S.z = z;
        S.root_pred[0] = z * S.ax; S.root_pred[1] = z * S.ay; S.root_pred[2] = z;
    }
    __syncthreads();
    if (threadIdx.x < ML_J) {
        // jp (train mode: + root_gt), landmarks = perspective(jp, K)
        const int j = threadIdx.x;
        const float* K = a.K + (long)b * 9;
        const float x = S.jp[j * 3] + S.root_gt[0], y = S.jp[j * 3 + 1] + S.root_gt[1], zz = S.jp[j * 3 + 2] + S.root_gt[2];
        const float px = K[0] * x + K[1] * y + K[2] * zz, py = K[3] * x + K[4] * y + K[5] * zz, pz = K[6] * x + K[7] * y + K[8] * zz;
        S.lms[j * 2] = px / (pz + 1e-7f); S.lms[j * 2 + 1] = py / (pz + 1e-7f);
    }
    __syncthreads();
}
// ground truth of the GCN-level terms: node i of the 252 = mean of mean of GCN nodes 4i .. 4i + 3 (two pair-averagings, :117-122)
__device__ __forceinline__ float ml_pool4(float x0, float x1, float x2, float x3) { return ((x0 + x1) * 0.5f + (x2 + x3) * 0.5f) * 0.5f; }

__global__ __launch_bounds__(ML_T) void mesh_loss_fwd_kernel(const MeshLoss a) {
    __shared__ MlShared S;
    __shared__ float sm[ML_T / 64];
    const int g = blockIdx.x / a.B, b = blockIdx.x - g * a.B;
    const long gb = blockIdx.x;
    ml_common(a, g, b, S);
    const float* vp = a.vp + gb * ML_V * 3;
    const float* vg = a.vgt[g] + (long)b * ML_V * 3;
    float s0 = 0.f, s1 = 0.f, s9 = 0.f;
    {
        const float* v2p = a.v2p + gb * ML_V * 2;
        const float* v2g = a.v2gt[g] + (long)b * ML_V * 2;
        for (int i = threadIdx.x; i < ML_V * 2; i += ML_T) { const float d = v2p[i] - v2g[i]; s0 += d * d; }
        for (int i = threadIdx.x; i < ML_V * 3; i += ML_T) {
            const int k = i % 3;
            s1 += fabsf(vp[i] - (vg[i] - S.root_gt[k]));
            s9 += fabsf((vp[i] + S.root_pred[k]) - vg[i]);
        }
    }
    float s2 = 0.f, s8 = 0.f, s10 = 0.f, s11 = 0.f, s7 = 0.f;
    if (threadIdx.x < ML_J * 3) {
        const int k = threadIdx.x % 3;
        s2 = fabsf(S.jp[threadIdx.x] - S.jg[threadIdx.x]);
        s8 = fabsf((S.jp[threadIdx.x] + S.root_gt[k]) - a.jgt[g][(long)b * ML_J * 3 + threadIdx.x]);
    }
    if (threadIdx.x < ML_J * 2) { const float d = S.lms[threadIdx.x] - a.lmsgt[g][(long)b * ML_J * 2 + threadIdx.x]; s10 = d * d; }
    if (threadIdx.x < 3) s7 = fabsf(S.root_pred[threadIdx.x] - S.root_gt[threadIdx.x]);
    if (threadIdx.x < 20) {
        const float* lg = a.lmsgt[g] + (long)b * ML_J * 2;
        const int pa = ml_bone_a[threadIdx.x], pc = ml_bone_c[threadIdx.x];
        const float vx = S.lms[pc * 2] - S.lms[pa * 2], vy = S.lms[pc * 2 + 1] - S.lms[pa * 2 + 1];
        const float gx = lg[pc * 2] - lg[pa * 2], gy = lg[pc * 2 + 1] - lg[pa * 2 + 1];
        const float nv = sqrtf(vx * vx + vy * vy + 1e-4f), ng = sqrtf(gx * gx + gy * gy + 1e-4f);
        const float dx = vx / nv - gx / ng, dy = vy / nv - gy / ng;
        s11 = dx * dx + dy * dy;
    }
    // faces (same arithmetic as face_loss_fwd_kernel; ground truth in the offset frame: differences only, so the frame cancels)
    float s3 = 0.f, s4 = 0.f;
    {
        const long long* fc = a.faces + (long)g * a.Fc * 3;
        for (int f = threadIdx.x; f < a.Fc; f += ML_T) {
            const long i0 = fc[3 * f], i1 = fc[3 * f + 1], i2 = fc[3 * f + 2];
            const V3 p0 = ld3(vp, i0), p1 = ld3(vp, i1), p2 = ld3(vp, i2);
            const V3 r = {S.root_gt[0], S.root_gt[1], S.root_gt[2]};
            const V3 g0 = sub3(ld3(vg, i0), r), g1 = sub3(ld3(vg, i1), r), g2 = sub3(ld3(vg, i2), r);
            float l;
            const V3 n = unit3(cross3(unit3(sub3(g1, g0), l), unit3(sub3(g2, g0), l)), l);
            const V3 v1 = sub3(p1, p0), v2 = sub3(p2, p0), v3 = sub3(p2, p1);
            s3 += fabsf(dot3(unit3(v1, l), n)) + fabsf(dot3(unit3(v2, l), n)) + fabsf(dot3(unit3(v3, l), n));
            s4 += fabsf(len3(v1) - len3(sub3(g0, g1))) + fabsf(len3(v2) - len3(sub3(g0, g2))) + fabsf(len3(v3) - len3(sub3(g1, g2)));
        }
    }
    // GCN-level terms: 3-D against the LEFT hand's ground truth (offset frame) in this hand's GCN order, 2-D against this hand's own
    float s5 = 0.f, s6 = 0.f;
    {
        const float* hd3 = a.hd3 + gb * ML_VG * 3;
        const float* hd2 = a.hd2 + gb * ML_VG * 2;
        const float* vl = a.vgt[0] + (long)b * ML_V * 3;                 // hand 0 of this sample
        const float* rl = a.jgt[0] + (long)b * ML_J * 3 + 9 * 3;
        const float* v2g = a.v2gt[g] + (long)b * ML_V * 2;
        const long long* pm = a.perm[g];
        for (int i = threadIdx.x; i < ML_VG * 3; i += ML_T) {
            const int n = i / 3, k = i - 3 * n;
            const float t = ml_pool4(vl[pm[4 * n] * 3 + k] - rl[k], vl[pm[4 * n + 1] * 3 + k] - rl[k], vl[pm[4 * n + 2] * 3 + k] - rl[k], vl[pm[4 * n + 3] * 3 + k] - rl[k]);
            s5 += fabsf(hd3[i] - t);
        }
        for (int i = threadIdx.x; i < ML_VG * 2; i += ML_T) {
            const int n = i / 2, k = i - 2 * n;
            const float t = ml_pool4(v2g[pm[4 * n] * 2 + k], v2g[pm[4 * n + 1] * 2 + k], v2g[pm[4 * n + 2] * 2 + k], v2g[pm[4 * n + 3] * 2 + k]);
            const float d = hd2[i] - t;
            s6 += d * d;
        }
    }
    float v[MLP] = {s0, s1 / (ML_V * 3), s2 / (ML_J * 3), s3, s4, s5 / (ML_VG * 3), s6, s7 / 3.f, s8 / (ML_J * 3), s9 / (ML_V * 3), s10, s11 / 20.f};
#pragma unroll
    for (int k = 0; k < MLP; ++k) {
        const float t = block_sum_waves<ML_T / 64>(v[k], sm);
        if (threadIdx.x == 0) a.part[gb * MLP + k] = t;
    }
}
// out layout (floats): [0] verts2d, [1] norm, [2] edge, [3] gcn_2d (scalars), then 8 vectors of B: root, verts, abs_verts, gcn, abs_joints,
// joints2d, joints, bone  (the weights of `valid` and the x1000 of the absolute terms as in the reference, :506-525), then the weighted sum
// sum_k coef[k] term_k per sample [B] (what CtdetLoss.total adds up for these twelve terms)
__global__ __launch_bounds__(64) void mesh_loss_finalize_kernel(const MeshLoss a) {
    const int lane = threadIdx.x, B = a.B;
    const float k2 = (2.f / a.size) * (2.f / a.size);
    float t0 = 0.f, t3 = 0.f, t4 = 0.f, t6 = 0.f, m10[2] = {0.f, 0.f};
    for (int i = lane; i < 2 * B; i += 64) {
        const float* p = a.part + (long)i * MLP;
        t0 += p[0]; t3 += p[3]; t4 += p[4]; t6 += p[6];
        m10[i / B] += p[10];
    }
    t0 = wave_sum(t0); t3 = wave_sum(t3); t4 = wave_sum(t4); t6 = wave_sum(t6);
    m10[0] = wave_sum(m10[0]) / (float)(B * ML_J * 2); m10[1] = wave_sum(m10[1]) / (float)(B * ML_J * 2);
    if (lane == 0) {
        a.out[0] = t0 / (float)(B * ML_V * 2) * k2;
        a.out[1] = t3 / (float)(B * 3 * a.Fc);
        a.out[2] = t4 / (float)(B * 3 * a.Fc);
        a.out[3] = t6 / (float)(B * ML_VG * 2) * k2;
    }
    for (int b = lane; b < B; b += 64) {
        const float* p0 = a.part + (long)b * MLP;
        const float* p1 = a.part + (long)(B + b) * MLP;
        const float v0 = a.valid[b * 2], v1 = a.valid[b * 2 + 1];
        float* o = a.out + 4 + b;
        o[0 * B] = (p0[7] * v0 + p1[7] * v1) * 1000.f;                   // root
        o[1 * B] = p0[1] * v0 + p1[1] * v1;                              // verts
        o[2 * B] = (p0[9] * v0 + p1[9] * v1) * 1000.f;                   // abs_verts
        o[3 * B] = (p0[5] + p1[5]) * v0;                                 // gcn: valid[:, 0] for both hands (:481-482)
        o[4 * B] = (p0[8] * v0 + p1[8] * v1) * 1000.f;                   // abs_joints
        o[5 * B] = (m10[0] * v0 + m10[1] * v1) * k2;                     // joints2d: the hand's batch mean, weighted per sample (:499-500)
        o[6 * B] = p0[2] * v0 + p1[2] * v1;                              // joints
        o[7 * B] = p0[11] * v0 + p1[11] * v1;                            // bone direction
    }
    __syncthreads();                                                     // (one wave: orders the scalar terms' stores above before the reads below)
    for (int b = lane; b < B; b += 64) {
        float t = a.coef[0] * a.out[0] + a.coef[1] * a.out[1] + a.coef[2] * a.out[2] + a.coef[3] * a.out[3];
#pragma unroll
        for (int k = 0; k < 8; ++k) t += a.coef[4 + k] * a.out[4 + k * B + b];
        a.out[4 + 8 * B + b] = t;
    }
}
PDF_API int pdf_mesh_loss_fwd(const MeshLoss* a, hipStream_t s) {
    if (a == nullptr || a->B < 1 || a->part == nullptr || a->out == nullptr || a->Fc < 1) return PDF_E_BADARG;
    hipLaunchKernelGGL(mesh_loss_fwd_kernel, dim3(2 * a->B), dim3(ML_T), 0, s, *a);
    hipLaunchKernelGGL(mesh_loss_finalize_kernel, dim3(1), dim3(64), 0, s, *a);
    PDF_LAUNCH_CHECK();
    return 0;
}

// backward: gradients of sum_k gout[k] * out[k] with respect to vp, v2p, hd3, hd2, r
__global__ __launch_bounds__(ML_T) void mesh_loss_bwd_kernel(const MeshLoss a) {
    __shared__ MlShared S;
    __shared__ float acc[ML_V * 3];                                       // d vp of this (hand, sample)
    __shared__ float djp[ML_J * 3], dlms[ML_J * 2], droot[3], gsum[2];
    const int g = blockIdx.x / a.B, b = blockIdx.x - g * a.B, B = a.B;
    const long gb = blockIdx.x;
    ml_common(a, g, b, S);
    const float k2 = (2.f / a.size) * (2.f / a.size);
    // upstream gradient of term k: coef[k] gmp[b] for the per-sample terms, coef[k] sum_b gmp[b] for the scalar ones
    if (threadIdx.x < 64) {
        float t = 0.f, tj = 0.f;
        for (int i = threadIdx.x; i < B; i += 64) { t += a.gmp[i]; tj += a.gmp[i] * a.valid[i * 2 + g]; }
        t = wave_sum(t); tj = wave_sum(tj);
        // joints2d: d (hand's batch mean) = sum over samples of coef gmp[b'] valid[b'][g] k2
        if (threadIdx.x == 0) { gsum[1] = t; gsum[0] = a.coef[9] * tj * k2 / (float)(B * ML_J * 2); }
    }
    __syncthreads();
    const float gs = gsum[1], gb_ = a.gmp[b];
    const float vld = a.valid[b * 2 + g], vld0 = a.valid[b * 2];
    const float c0 = a.coef[0] * gs * k2 / (float)(B * ML_V * 2);
    const float c1 = a.coef[5] * gb_ * vld / (float)(ML_V * 3);
    const float c2 = a.coef[10] * gb_ * vld / (float)(ML_J * 3);
    const float c3 = a.coef[1] * gs / (float)(B * 3 * a.Fc), c4 = a.edge_grad ? a.coef[2] * gs / (float)(B * 3 * a.Fc) : 0.f;
    const float c5 = a.coef[7] * gb_ * vld0 / (float)(ML_VG * 3);
    const float c6 = a.coef[3] * gs * k2 / (float)(B * ML_VG * 2);
    const float c7 = a.coef[4] * gb_ * 1000.f * vld / 3.f;
    const float c8 = a.coef[8] * gb_ * 1000.f * vld / (float)(ML_J * 3);
    const float c9 = a.coef[6] * gb_ * 1000.f * vld / (float)(ML_V * 3);
    const float c11 = a.coef[11] * gb_ * vld / 20.f;
    if (threadIdx.x < 3) droot[threadIdx.x] = 0.f;
    if (threadIdx.x < ML_J * 3) djp[threadIdx.x] = 0.f;
    if (threadIdx.x < ML_J * 2) dlms[threadIdx.x] = 0.f;
    __syncthreads();
    const float c10 = gsum[0];
    const float* vp = a.vp + gb * ML_V * 3;
    const float* vg = a.vgt[g] + (long)b * ML_V * 3;
    // 2-D mesh term
    {
        const float* v2p = a.v2p + gb * ML_V * 2;
        const float* v2g = a.v2gt[g] + (long)b * ML_V * 2;
        float* d = a.dv2p + gb * ML_V * 2;
        for (int i = threadIdx.x; i < ML_V * 2; i += ML_T) d[i] = c0 * 2.f * (v2p[i] - v2g[i]);
    }
    // vertex L1 terms -> acc; the absolute term also feeds the root
    float dr0 = 0.f, dr1 = 0.f, dr2 = 0.f;
    for (int i = threadIdx.x; i < ML_V * 3; i += ML_T) {
        const int k = i % 3;
        const float e1 = vp[i] - (vg[i] - S.root_gt[k]), e9 = (vp[i] + S.root_pred[k]) - vg[i];
        const float s1 = e1 > 0.f ? 1.f : (e1 < 0.f ? -1.f : 0.f), s9 = e9 > 0.f ? 1.f : (e9 < 0.f ? -1.f : 0.f);
        acc[i] = c1 * s1 + c9 * s9;
        const float t = c9 * s9;
        if (k == 0) dr0 += t; else if (k == 1) dr1 += t; else dr2 += t;
    }
    dr0 = wave_sum(dr0); dr1 = wave_sum(dr1); dr2 = wave_sum(dr2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&droot[0], dr0); atomicAdd(&droot[1], dr1); atomicAdd(&droot[2], dr2); }
    // joint-level terms
    if (threadIdx.x < 20) {                                               // bone direction -> d lms
        const float* lg = a.lmsgt[g] + (long)b * ML_J * 2;
        const int pa = ml_bone_a[threadIdx.x], pc = ml_bone_c[threadIdx.x];
        const float vx = S.lms[pc * 2] - S.lms[pa * 2], vy = S.lms[pc * 2 + 1] - S.lms[pa * 2 + 1];
        const float gx = lg[pc * 2] - lg[pa * 2], gy = lg[pc * 2 + 1] - lg[pa * 2 + 1];
        const float nv = sqrtf(vx * vx + vy * vy + 1e-4f), ng = sqrtf(gx * gx + gy * gy + 1e-4f);
        const float ux = vx / nv, uy = vy / nv;
        const float ex = 2.f * c11 * (ux - gx / ng), ey = 2.f * c11 * (uy - gy / ng);     // d / d u
        // u = v / n(v): d u / d v = (I - u u^T) / n
        const float dot = ex * ux + ey * uy;
        const float dvx = (ex - ux * dot) / nv, dvy = (ey - uy * dot) / nv;
        atomicAdd(&dlms[pc * 2], dvx); atomicAdd(&dlms[pc * 2 + 1], dvy);
        atomicAdd(&dlms[pa * 2], -dvx); atomicAdd(&dlms[pa * 2 + 1], -dvy);
    }
    __syncthreads();
    if (threadIdx.x < ML_J) {                                             // 2-D joint term + projection backward -> d jp
        const int j = threadIdx.x;
        const float* K = a.K + (long)b * 9;
        const float* lg = a.lmsgt[g] + (long)b * ML_J * 2;
        const float dlx = dlms[j * 2] + c10 * 2.f * (S.lms[j * 2] - lg[j * 2]);
        const float dly = dlms[j * 2 + 1] + c10 * 2.f * (S.lms[j * 2 + 1] - lg[j * 2 + 1]);
        const float x = S.jp[j * 3] + S.root_gt[0], y = S.jp[j * 3 + 1] + S.root_gt[1], zz = S.jp[j * 3 + 2] + S.root_gt[2];
        const float pz = K[6] * x + K[7] * y + K[8] * zz + 1e-7f;
        const float dpx = dlx / pz, dpy = dly / pz, dpz = -(dlx * S.lms[j * 2] + dly * S.lms[j * 2 + 1]) / pz;
        djp[j * 3] += K[0] * dpx + K[3] * dpy + K[6] * dpz;
        djp[j * 3 + 1] += K[1] * dpx + K[4] * dpy + K[7] * dpz;
        djp[j * 3 + 2] += K[2] * dpx + K[5] * dpy + K[8] * dpz;
    }
    __syncthreads();
    if (threadIdx.x < ML_J * 3) {
        const int k = threadIdx.x % 3;
        const float e2 = S.jp[threadIdx.x] - S.jg[threadIdx.x];
        const float e8 = (S.jp[threadIdx.x] + S.root_gt[k]) - a.jgt[g][(long)b * ML_J * 3 + threadIdx.x];
        djp[threadIdx.x] += c2 * (e2 > 0.f ? 1.f : (e2 < 0.f ? -1.f : 0.f)) + c8 * (e8 > 0.f ? 1.f : (e8 < 0.f ? -1.f : 0.f));
    }
    if (threadIdx.x < 3) {                                                // root term
        const float e = S.root_pred[threadIdx.x] - S.root_gt[threadIdx.x];
        droot[threadIdx.x] += c7 * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
    }
    __syncthreads();
    // d vp += reg^T d jp_off
    {
        const float* rg = a.reg[g];
        for (int v = threadIdx.x; v < ML_V; v += ML_T) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < ML_J; ++j) { const float w = rg[(long)j * ML_V + v]; s0 += w * djp[j * 3]; s1 += w * djp[j * 3 + 1]; s2 += w * djp[j * 3 + 2]; }
            acc[v * 3] += s0; acc[v * 3 + 1] += s1; acc[v * 3 + 2] += s2;
        }
    }
    __syncthreads();
    // faces (face_loss_bwd_kernel's arithmetic)
    if (c3 != 0.f || c4 != 0.f) {
        const long long* fc = a.faces + (long)g * a.Fc * 3;
        for (int f = threadIdx.x; f < a.Fc; f += ML_T) {
            const long idx[3] = {(long)fc[3 * f], (long)fc[3 * f + 1], (long)fc[3 * f + 2]};
            const V3 pv[3] = {ld3(vp, idx[0]), ld3(vp, idx[1]), ld3(vp, idx[2])};
            const V3 g0 = ld3(vg, idx[0]), g1 = ld3(vg, idx[1]), g2 = ld3(vg, idx[2]);
            float l;
            const V3 n = unit3(cross3(unit3(sub3(g1, g0), l), unit3(sub3(g2, g0), l)), l);
            const V3 gvv[3] = {g0, g1, g2};
            const int ea[3] = {0, 0, 1}, eb[3] = {1, 2, 2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const V3 v = sub3(pv[eb[k]], pv[ea[k]]);
                float lv;
                const V3 u = unit3(v, lv);
                const float c = dot3(u, n);
                const float sg = c > 0.f ? 1.f : (c < 0.f ? -1.f : 0.f);
                const float kn = c3 * sg / lv;
                V3 d = {kn * (n.x - u.x * c), kn * (n.y - u.y * c), kn * (n.z - u.z * c)};
                if (c4 != 0.f) {
                    const float lp = len3(v), lg = len3(sub3(gvv[ea[k]], gvv[eb[k]]));
                    const float ke = lp > 0.f ? c4 * (lp > lg ? 1.f : (lp < lg ? -1.f : 0.f)) / lp : 0.f;
                    d.x += ke * v.x; d.y += ke * v.y; d.z += ke * v.z;
                }
                acc3(acc, idx[eb[k]], d, 1.f);
                acc3(acc, idx[ea[k]], d, -1.f);
            }
        }
    }
    // GCN-level terms
    {
        const float* hd3 = a.hd3 + gb * ML_VG * 3;
        const float* hd2 = a.hd2 + gb * ML_VG * 2;
        const float* vl = a.vgt[0] + (long)b * ML_V * 3;
        const float* rl = a.jgt[0] + (long)b * ML_J * 3 + 9 * 3;
        const float* v2g = a.v2gt[g] + (long)b * ML_V * 2;
        const long long* pm = a.perm[g];
        float* d3 = a.dhd3 + gb * ML_VG * 3;
        float* d2 = a.dhd2 + gb * ML_VG * 2;
        for (int i = threadIdx.x; i < ML_VG * 3; i += ML_T) {
            const int n = i / 3, k = i - 3 * n;
            const float t = ml_pool4(vl[pm[4 * n] * 3 + k] - rl[k], vl[pm[4 * n + 1] * 3 + k] - rl[k], vl[pm[4 * n + 2] * 3 + k] - rl[k], vl[pm[4 * n + 3] * 3 + k] - rl[k]);
            const float e = hd3[i] - t;
            d3[i] = c5 * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
        }
        for (int i = threadIdx.x; i < ML_VG * 2; i += ML_T) {
            const int n = i / 2, k = i - 2 * n;
            const float t = ml_pool4(v2g[pm[4 * n] * 2 + k], v2g[pm[4 * n + 1] * 2 + k], v2g[pm[4 * n + 2] * 2 + k], v2g[pm[4 * n + 3] * 2 + k]);
            d2[i] = c6 * 2.f * (hd2[i] - t);
        }
    }
    __syncthreads();
    {
        float* o = a.dvp + gb * ML_V * 3;
        for (int i = threadIdx.x; i < ML_V * 3; i += ML_T) o[i] = acc[i];
    }
    if (threadIdx.x == 0) {
        // root_pred = (z ax, z ay, z), z = 0.4 + r0 / 100, ax = (r1 / 100 + cx - K02) / (K00 + eps)
        const float* K = a.K + (long)b * 9;
        float* d = a.dr + gb * 3;
        d[0] = (droot[2] + droot[0] * S.ax + droot[1] * S.ay) / 100.f;
        d[1] = droot[0] * S.z / (100.f * (K[0] + 1e-7f));
        d[2] = droot[1] * S.z / (100.f * (K[4] + 1e-7f));
    }
}
PDF_API int pdf_mesh_loss_bwd(const MeshLoss* a, hipStream_t s) {
    if (a == nullptr || a->B < 1 || a->gmp == nullptr || a->dvp == nullptr || a->dv2p == nullptr || a->dhd3 == nullptr || a->dhd2 == nullptr || a->dr == nullptr) return PDF_E_BADARG;
    hipLaunchKernelGGL(mesh_loss_bwd_kernel, dim3(2 * a->B), dim3(ML_T), 0, s, *a);
    PDF_LAUNCH_CHECK();
    return 0;
}
PDF_API int pdf_debug_mesh_loss_size() { return (int)sizeof(MeshLoss); }
