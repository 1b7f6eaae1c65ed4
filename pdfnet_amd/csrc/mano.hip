// MANO linear-blend-skinning forward (reference lib/models/networks/manolayer.py:257-334 with
// use_pca=False) and small strided batched matmuls (joint regressor Mano_model.py:309-323, vertex
// up-sampling / avg_head of intaghand_decoder.py:205,224).  One workgroup per sample: everything a
// hand needs (778x3 shaped vertices, 16 SE(3) transforms) lives in LDS; the ~150 tiny launches of the
// reference become one.
#include "common.h"

__device__ __constant__ int c_mano_parent[16] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14};
__device__ __constant__ int c_mano_order[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};

__global__ __launch_bounds__(256) void mano_lbs_kernel(
    const float* __restrict__ root_aa, const float* __restrict__ pose_aa, const float* __restrict__ shape, const float* __restrict__ trans,
    const float* __restrict__ v_template, const float* __restrict__ shapedirs, const float* __restrict__ posedirs,
    const float* __restrict__ J_reg, const float* __restrict__ weights, int left_side, int center_idx,
    float* __restrict__ verts, float* __restrict__ joints) {
    __shared__ float vs[778 * 3];      // shaped, then posed vertices
    __shared__ float Rm[16][9];
    __shared__ float pf[135];
    __shared__ float jt[16][3];
    __shared__ float G[16][12];
    __shared__ float jo[21][3];
    __shared__ float sh[10];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 10) sh[tid] = shape[b * 10 + tid];
    if (tid < 16) {
        // rodrigues_batch (manolayer.py:32-48): angle = |axis| + 1e-8
        const float* ax = tid == 0 ? root_aa + b * 3 : pose_aa + b * 45 + (tid - 1) * 3;
        float x = ax[0], y = ax[1], z = ax[2];
        float ang = sqrtf(x * x + y * y + z * z) + 1e-8f;
        x /= ang; y /= ang; z /= ang;
        float sn = sinf(ang), cs = cosf(ang), oc = 1.f - cs;
        // L = [[0,-z,y],[z,0,-x],[-y,x,0]];  R = I + sn L + oc L^2
        float L[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                float l2 = L[r * 3 + 0] * L[0 * 3 + c] + L[r * 3 + 1] * L[1 * 3 + c] + L[r * 3 + 2] * L[2 * 3 + c];
                Rm[tid][r * 3 + c] = (r == c ? 1.f : 0.f) + sn * L[r * 3 + c] + oc * l2;
            }
    }
    __syncthreads();
    for (int i = tid; i < 778 * 3; i += 256) {
        float a = v_template[i];
        for (int k = 0; k < 10; ++k) a += shapedirs[i * 10 + k] * sh[k];
        vs[i] = a;
    }
    if (tid < 135) {
        int jnt = tid / 9, e = tid % 9;
        pf[tid] = Rm[jnt + 1][e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
    }
    __syncthreads();
    for (int o = wave; o < 48; o += 4) {                     // j_tpose = J_regressor @ v_shaped
        int j = o / 3, c = o % 3;
        float a = 0.f;
        for (int v = lane; v < 778; v += 64) a += J_reg[j * 778 + v] * vs[v * 3 + c];
        a = wave_sum(a);
        if (lane == 0) jt[j][c] = a;
    }
    __syncthreads();
    float posed[10];
    {
        int n = 0;
        for (int i = tid; i < 778 * 3; i += 256, ++n) {
            float a = vs[i];
            const float* pd = posedirs + (long)i * 135;
            for (int k = 0; k < 135; ++k) a += pd[k] * pf[k];
            posed[n] = a;
        }
    }
    if (tid == 0) {
        for (int i = 0; i < 16; ++i) {
            float loc[12];
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) loc[r * 4 + c] = Rm[i][r * 3 + c];
                float t = 0.f;                                   // (I - R) j
                for (int c = 0; c < 3; ++c) t += ((r == c ? 1.f : 0.f) - Rm[i][r * 3 + c]) * jt[i][c];
                loc[r * 4 + 3] = t;
            }
            int p = c_mano_parent[i];
            if (p < 0) { for (int e = 0; e < 12; ++e) G[i][e] = loc[e]; }
            else {
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 4; ++c) {
                        float a = G[p][r * 4 + 0] * loc[0 * 4 + c] + G[p][r * 4 + 1] * loc[1 * 4 + c] + G[p][r * 4 + 2] * loc[2 * 4 + c];
                        if (c == 3) a += G[p][r * 4 + 3];
                        G[i][r * 4 + c] = a;
                    }
            }
        }
    }
    __syncthreads();
    {
        int n = 0;
        for (int i = tid; i < 778 * 3; i += 256, ++n) vs[i] = posed[n];
    }
    __shared__ float jun[21][3];      // unordered: 0..15 joints, 16..20 finger tips
    if (tid < 16) {
        int p = c_mano_parent[tid];
        for (int r = 0; r < 3; ++r) {
            float a;
            if (p < 0) a = jt[0][r];
            else a = G[p][r * 4 + 0] * jt[tid][0] + G[p][r * 4 + 1] * jt[tid][1] + G[p][r * 4 + 2] * jt[tid][2] + G[p][r * 4 + 3];
            jun[tid][r] = a;
        }
    }
    for (int v = tid; v < 778; v += 256) {
        float T[12];
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        for (int j = 0; j < 16; ++j) {
            float w = weights[v * 16 + j];
            if (w != 0.f) for (int e = 0; e < 12; ++e) T[e] += w * G[j][e];
        }
        float x = vs[v * 3], y = vs[v * 3 + 1], z = vs[v * 3 + 2];
        float ox = T[0] * x + T[1] * y + T[2] * z + T[3];
        float oy = T[4] * x + T[5] * y + T[6] * z + T[7];
        float oz = T[8] * x + T[9] * y + T[10] * z + T[11];
        vs[v * 3] = ox; vs[v * 3 + 1] = oy; vs[v * 3 + 2] = oz;      // each thread only touches its own vertex
        // tips (manolayer.py:305-308): left uses vertex 445, right 444
        int tip = -1;
        if (v == 745) tip = 0; else if (v == 317) tip = 1; else if (v == (left_side ? 445 : 444)) tip = 2;
        else if (v == 556) tip = 3; else if (v == 673) tip = 4;
        if (tip >= 0) { jun[16 + tip][0] = ox; jun[16 + tip][1] = oy; jun[16 + tip][2] = oz; }
    }
    __syncthreads();
    if (tid < 63) { int j = tid / 3, c = tid % 3; jo[j][c] = jun[c_mano_order[j]][c]; }
    __syncthreads();
    float off[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < 3; ++c) {
        if (center_idx >= 0) off[c] -= jo[center_idx][c];
        if (trans != nullptr) off[c] += trans[b * 3 + c];
    }
    float* vout = verts + (long)b * 778 * 3;
    for (int i = tid; i < 778 * 3; i += 256) vout[i] = vs[i] + off[i % 3];
    if (tid < 63) joints[(long)b * 63 + tid] = jo[tid / 3][tid % 3] + off[tid % 3];
}

PDF_API int pdf_mano_lbs_fwd(const float* root_aa, const float* pose_aa, const float* shape, const float* trans,
                             const float* v_template, const float* shapedirs, const float* posedirs, const float* J_reg,
                             const float* weights, int B, int left_side, int center_idx, float* verts, float* joints, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(mano_lbs_kernel, dim3(B), dim3(256), 0, s, root_aa, pose_aa, shape, trans, v_template, shapedirs, posedirs,
                       J_reg, weights, left_side, center_idx, verts, joints);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Backward of mano_lbs_kernel: gradients of (verts, joints) with respect to the axis-angle root / pose, the shape
// coefficients and the translation.  One workgroup per sample re-runs the forward quantities it needs into LDS and walks the
// graph in reverse: offsets -> joint order / tips -> skinning (dT_v = dout_v (x) [vp_v; 1], dG_j = sum_v w_vj dT_v) ->
// joint positions -> kinematic chain in reverse order (G_i = G_parent o [R_i | (I - R_i) j_i]) -> pose blend shapes and joint
// regressor -> shape blend shapes -> Rodrigues (R = I + sin(t) L + (1 - cos(t)) L^2, t = |a| + 1e-8, L = skew(a / t)).
__device__ __forceinline__ void rodrigues_dev(const float* ax, float* R, float* Lout, float& ang, float& nrm) {
    float x = ax[0], y = ax[1], z = ax[2];
    nrm = sqrtf(x * x + y * y + z * z);
    ang = nrm + 1e-8f;
    x /= ang; y /= ang; z /= ang;
    const float sn = sinf(ang), oc = 1.f - cosf(ang);
    const float L[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            const float l2 = L[r * 3 + 0] * L[0 * 3 + c] + L[r * 3 + 1] * L[1 * 3 + c] + L[r * 3 + 2] * L[2 * 3 + c];
            R[r * 3 + c] = (r == c ? 1.f : 0.f) + sn * L[r * 3 + c] + oc * l2;
            if (Lout) Lout[r * 3 + c] = L[r * 3 + c];
        }
}

__global__ __launch_bounds__(256) void mano_lbs_bwd_kernel(
    const float* __restrict__ root_aa, const float* __restrict__ pose_aa, const float* __restrict__ shape,
    const float* __restrict__ v_template, const float* __restrict__ shapedirs, const float* __restrict__ posedirs,
    const float* __restrict__ J_reg, const float* __restrict__ weights, int left_side, int center_idx,
    const float* __restrict__ dverts, const float* __restrict__ djoints,
    float* __restrict__ droot, float* __restrict__ dpose, float* __restrict__ dshape, float* __restrict__ dtrans) {
    __shared__ float vs[778 * 3];      // v_shaped, later d v_shaped
    __shared__ float vp[778 * 3];      // posed (pre-skinning) vertices, later d vp
    __shared__ float dout[778 * 3];    // gradient of the skinned vertices (tips and offset terms folded in)
    __shared__ float Rm[16][9], Lm[16][9], jt[16][3], G[16][12], dG[16][12], dR[16][9], djt[16][3], djun[21][3];
    __shared__ float pf[135], dpf[135], sh[10], ang[16], nrm[16], tot[3];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---------------- forward quantities
    if (tid < 10) sh[tid] = shape[b * 10 + tid];
    if (tid < 16) rodrigues_dev(tid == 0 ? root_aa + b * 3 : pose_aa + b * 45 + (tid - 1) * 3, Rm[tid], Lm[tid], ang[tid], nrm[tid]);
    __syncthreads();
    for (int i = tid; i < 778 * 3; i += 256) {
        float a = v_template[i];
        for (int k = 0; k < 10; ++k) a += shapedirs[i * 10 + k] * sh[k];
        vs[i] = a;
    }
    if (tid < 135) { const int j = tid / 9, e = tid % 9; pf[tid] = Rm[j + 1][e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f); }
    __syncthreads();
    for (int o = wave; o < 48; o += 4) {
        const int j = o / 3, c = o % 3;
        float a = 0.f;
        for (int v = lane; v < 778; v += 64) a += J_reg[j * 778 + v] * vs[v * 3 + c];
        a = wave_sum(a);
        if (lane == 0) jt[j][c] = a;
    }
    for (int i = tid; i < 778 * 3; i += 256) {
        float a = vs[i];
        const float* pd = posedirs + (long)i * 135;
        for (int k = 0; k < 135; ++k) a += pd[k] * pf[k];
        vp[i] = a;
    }
    __syncthreads();
    if (tid == 0) {
        for (int i = 0; i < 16; ++i) {
            float loc[12];
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) loc[r * 4 + c] = Rm[i][r * 3 + c];
                float t = 0.f;
                for (int c = 0; c < 3; ++c) t += ((r == c ? 1.f : 0.f) - Rm[i][r * 3 + c]) * jt[i][c];
                loc[r * 4 + 3] = t;
            }
            const int p = c_mano_parent[i];
            if (p < 0) { for (int e = 0; e < 12; ++e) G[i][e] = loc[e]; }
            else
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 4; ++c) {
                        float a = G[p][r * 4 + 0] * loc[0 * 4 + c] + G[p][r * 4 + 1] * loc[1 * 4 + c] + G[p][r * 4 + 2] * loc[2 * 4 + c];
                        if (c == 3) a += G[p][r * 4 + 3];
                        G[i][r * 4 + c] = a;
                    }
        }
    }
    // ---------------- backward: offsets, joint order, tips
    const float* dv = dverts + (long)b * 778 * 3;
    const float* dj = djoints + (long)b * 63;
    for (int i = tid; i < 778 * 3; i += 256) dout[i] = dv != nullptr && dverts != nullptr ? dv[i] : 0.f;
    if (tid < 63) djun[c_mano_order[tid / 3]][tid % 3] = djoints != nullptr ? dj[tid] : 0.f;
    if (tid < 48) { (&dG[0][0])[tid] = 0.f; (&dG[0][0])[tid + 48] = 0.f; (&dG[0][0])[tid + 96] = 0.f; (&dG[0][0])[tid + 144] = 0.f; (&djt[0][0])[tid] = 0.f; }
    if (tid < 144) (&dR[0][0])[tid] = 0.f;
    __syncthreads();
    if (wave < 3) {                                          // total incoming gradient per coordinate: d trans, and -d jo[center]
        float a = 0.f;
        for (int v = lane; v < 778; v += 64) a += dout[v * 3 + wave];
        if (lane < 21) a += djoints != nullptr ? dj[lane * 3 + wave] : 0.f;
        a = wave_sum(a);
        if (lane == 0) tot[wave] = a;
    }
    __syncthreads();
    if (tid < 3) {
        if (dtrans != nullptr) dtrans[b * 3 + tid] = tot[tid];
        if (center_idx >= 0) djun[c_mano_order[center_idx]][tid] -= tot[tid];
    }
    __syncthreads();
    if (tid < 5) {                                           // finger tips are skinned vertices
        const int tipv[5] = {745, 317, left_side ? 445 : 444, 556, 673};
        for (int c = 0; c < 3; ++c) dout[tipv[tid] * 3 + c] += djun[16 + tid][c];
    }
    __syncthreads();
    // ---------------- skinning: out_v = T_v [vp_v; 1], T_v = sum_j w_vj G_j
    {
        float dvp_loc[10][3];                                 // 778*3/256 < 10 vertices-coordinates per thread; vertices: 778/256 < 4
        int n = 0;
        for (int v = tid; v < 778; v += 256, ++n) {
            float T[12];
            for (int e = 0; e < 12; ++e) T[e] = 0.f;
            for (int j = 0; j < 16; ++j) { const float w = weights[v * 16 + j]; if (w != 0.f) for (int e = 0; e < 12; ++e) T[e] += w * G[j][e]; }
            const float gx = dout[v * 3], gy = dout[v * 3 + 1], gz = dout[v * 3 + 2];
            dvp_loc[n][0] = T[0] * gx + T[4] * gy + T[8] * gz;
            dvp_loc[n][1] = T[1] * gx + T[5] * gy + T[9] * gz;
            dvp_loc[n][2] = T[2] * gx + T[6] * gy + T[10] * gz;
        }
        __syncthreads();
        // dG_j[r][c] = sum_v w_vj dout_v[r] * [vp_v; 1][c]: 192 outputs, one thread each
        if (tid < 192) {
            const int j = tid / 12, e = tid % 12, r = e / 4, c = e % 4;
            float a = 0.f;
            for (int v = 0; v < 778; ++v) {
                const float w = weights[v * 16 + j];
                if (w != 0.f) a += w * dout[v * 3 + r] * (c < 3 ? vp[v * 3 + c] : 1.f);
            }
            dG[j][e] = a;
        }
        __syncthreads();
        n = 0;
        for (int v = tid; v < 778; v += 256, ++n) { vp[v * 3] = dvp_loc[n][0]; vp[v * 3 + 1] = dvp_loc[n][1]; vp[v * 3 + 2] = dvp_loc[n][2]; }   // vp := d vp
    }
    __syncthreads();
    // ---------------- joints and the kinematic chain, in reverse (serial: 16 small steps)
    if (tid == 0) {
        for (int i = 15; i >= 0; --i) {
            const int p = c_mano_parent[i];
            // joint position: jun_i = A_p jt_i + t_p (i > 0), jt_0 (root)
            if (p < 0) { for (int c = 0; c < 3; ++c) djt[0][c] += djun[0][c]; }
            else
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) { dG[p][r * 4 + c] += djun[i][r] * jt[i][c]; djt[i][c] += G[p][r * 4 + c] * djun[i][r]; }
                    dG[p][r * 4 + 3] += djun[i][r];
                }
        }
        for (int i = 15; i >= 0; --i) {
            const int p = c_mano_parent[i];
            float cvec[3], dc[3];
            for (int r = 0; r < 3; ++r) { cvec[r] = 0.f; for (int c = 0; c < 3; ++c) cvec[r] += ((r == c ? 1.f : 0.f) - Rm[i][r * 3 + c]) * jt[i][c]; }
            if (p < 0) {                                      // G_0 = [R_0 | c_0]
                for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) dR[0][r * 3 + c] += dG[0][r * 4 + c]; dc[r] = dG[0][r * 4 + 3]; }
            } else {                                          // A_i = A_p R_i, t_i = A_p c_i + t_p
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) {
                        float a = 0.f, q = 0.f;
                        for (int k = 0; k < 3; ++k) { a += dG[i][r * 4 + k] * Rm[i][c * 3 + k]; q += G[p][k * 4 + r] * dG[i][k * 4 + c]; }
                        dG[p][r * 4 + c] += a + dG[i][r * 4 + 3] * cvec[c];
                        dR[i][r * 3 + c] += q;
                    }
                    dG[p][r * 4 + 3] += dG[i][r * 4 + 3];
                    dc[r] = G[p][0 * 4 + r] * dG[i][0 * 4 + 3] + G[p][1 * 4 + r] * dG[i][1 * 4 + 3] + G[p][2 * 4 + r] * dG[i][2 * 4 + 3];
                }
            }
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) { dR[i][r * 3 + c] -= dc[r] * jt[i][c]; djt[i][c] += ((r == c ? 1.f : 0.f) - Rm[i][r * 3 + c]) * dc[r]; }
        }
    }
    __syncthreads();
    // ---------------- pose blend shapes: vp = vs + posedirs pf  ->  d pf, d vs;  joint regressor: jt = J_reg vs
    if (tid < 135) {
        float a = 0.f;
        for (int i = 0; i < 778 * 3; ++i) a += posedirs[(long)i * 135 + tid] * vp[i];
        dpf[tid] = a;
    }
    for (int i = tid; i < 778 * 3; i += 256) {
        const int v = i / 3, c = i % 3;
        float a = vp[i];
        for (int j = 0; j < 16; ++j) a += J_reg[j * 778 + v] * djt[j][c];
        vs[i] = a;                                            // vs := d v_shaped
    }
    __syncthreads();
    if (tid < 135) dR[tid / 9 + 1][tid % 9] += dpf[tid];
    if (dshape != nullptr)
        for (int k = wave; k < 10; k += 4) {
            float a = 0.f;
            for (int i = lane; i < 778 * 3; i += 64) a += shapedirs[i * 10 + k] * vs[i];
            a = wave_sum(a);
            if (lane == 0) dshape[b * 10 + k] = a;
        }
    __syncthreads();
    // ---------------- Rodrigues
    if (tid < 16) {
        const float* ax = tid == 0 ? root_aa + b * 3 : pose_aa + b * 45 + (tid - 1) * 3;
        const float th = ang[tid], sn = sinf(th), cs = cosf(th), oc = 1.f - cs;
        const float* L = Lm[tid];
        const float* g = dR[tid];
        float L2[9], dL[9], dsn = 0.f, doc = 0.f;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) { L2[r * 3 + c] = L[r * 3] * L[c] + L[r * 3 + 1] * L[3 + c] + L[r * 3 + 2] * L[6 + c]; }
        for (int e = 0; e < 9; ++e) { dsn += g[e] * L[e]; doc += g[e] * L2[e]; }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                float a = 0.f;                                // (g L^T + L^T g)[r][c]
                for (int k = 0; k < 3; ++k) a += g[r * 3 + k] * L[c * 3 + k] + L[k * 3 + r] * g[k * 3 + c];
                dL[r * 3 + c] = sn * g[r * 3 + c] + oc * a;
            }
        const float dth = cs * dsn + sn * doc;
        const float dn[3] = {dL[7] - dL[5], dL[2] - dL[6], dL[3] - dL[1]};
        const float n[3] = {ax[0] / th, ax[1] / th, ax[2] / th};
        const float ndn = n[0] * dn[0] + n[1] * dn[1] + n[2] * dn[2];
        float* o = tid == 0 ? droot + b * 3 : dpose + b * 45 + (tid - 1) * 3;
        for (int k = 0; k < 3; ++k) {
            const float dth_da = nrm[tid] > 0.f ? ax[k] / nrm[tid] : 0.f;     // torch.norm's subgradient at 0 is 0
            if ((tid == 0 ? droot : dpose) != nullptr) o[k] = dn[k] / th + (dth - ndn / th) * dth_da;
        }
    }
}

PDF_API int pdf_mano_lbs_bwd(const float* root_aa, const float* pose_aa, const float* shape,
                             const float* v_template, const float* shapedirs, const float* posedirs, const float* J_reg,
                             const float* weights, int B, int left_side, int center_idx, const float* dverts, const float* djoints,
                             float* droot, float* dpose, float* dshape, float* dtrans, hipStream_t s) {
    if (B <= 0) return 0;
    if (droot == nullptr || dpose == nullptr) return PDF_E_BADARG;
    hipLaunchKernelGGL(mano_lbs_bwd_kernel, dim3(B), dim3(256), 0, s, root_aa, pose_aa, shape, v_template, shapedirs, posedirs, J_reg, weights,
                       left_side, center_idx, dverts, djoints, droot, dpose, dshape, dtrans);
    PDF_LAUNCH_CHECK();
    return 0;
}

// Decode of the 122-channel `params` head (ManoRender.Split_coeff, Mano_render.py:145-194, the contract a MANO layer fed by
// network output follows -- simplified.py:730-736): per hand h (0 left, 1 right) the 61 channels at the hand's centre pixel
// are [orient 3 | pose 45 | shape 10 (multiplied by 0) | trans 3]; trans z += 0.6, x / y un-projected from the centre pixel:
// t_x = t_z (t_x + cx - K02) / K00, t_y = t_z (t_y + cy - K12) / K11 with (cx, cy) = (ind % g, ind / g) * down.
// params NHWC [B][HW][ldp] (122 channels), ind int64 [B][2].  Outputs [2][B][3|45|10|3].  bwd: scatter into dparams (zero-filled).
__global__ void mano_split_coeff_kernel(const float* __restrict__ params, float* __restrict__ dparams, int ldp, long HW, const long* __restrict__ ind,
                                        const float* __restrict__ Kc, int B, int g, int down,
                                        float* __restrict__ orient, float* __restrict__ pose, float* __restrict__ shp, float* __restrict__ trans, int bwd) {
    const int b = blockIdx.x, h = blockIdx.y, t = threadIdx.x;       // 64 threads: 61 channels
    const long i = ind[b * 2 + h];
    const long base = ((long)b * HW + i) * ldp + 61 * h;
    const float* K = Kc + b * 9;
    const float cx = (float)((i % g) * down), cy = (float)((i / g) * down);
    const long o = (long)h * B + b;
    if (!bwd) {
        if (t < 3) orient[o * 3 + t] = params[base + t];
        else if (t < 48) pose[o * 45 + t - 3] = params[base + t];
        else if (t < 58) shp[o * 10 + t - 48] = params[base + t] * 0.f;
        else if (t == 58) {
            const float tz = params[base + 60] + 0.6f;
            trans[o * 3 + 0] = tz * (params[base + 58] + cx - K[2]) / K[0];
            trans[o * 3 + 1] = tz * (params[base + 59] + cy - K[5]) / K[4];
            trans[o * 3 + 2] = tz;
        }
    } else {
        if (t < 3) dparams[base + t] = orient[o * 3 + t];
        else if (t < 48) dparams[base + t] = pose[o * 45 + t - 3];
        else if (t < 58) dparams[base + t] = 0.f;
        else if (t == 58) {
            const float tz = params[base + 60] + 0.6f;
            const float ax = (params[base + 58] + cx - K[2]) / K[0], ay = (params[base + 59] + cy - K[5]) / K[4];
            const float gx = trans[o * 3], gy = trans[o * 3 + 1], gz = trans[o * 3 + 2];
            dparams[base + 58] = gx * tz / K[0];
            dparams[base + 59] = gy * tz / K[4];
            dparams[base + 60] = gz + gx * ax + gy * ay;
        }
    }
}
PDF_API int pdf_mano_split_coeff(const float* params, int ldp, long HW, const long* ind, const float* K, int B, int input_res, int down,
                                 float* orient, float* pose, float* shape, float* trans, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(mano_split_coeff_kernel, dim3(B, 2), dim3(64), 0, s, params, nullptr, ldp, HW, ind, K, B, input_res / down, down, orient, pose, shape, trans, 0);
    PDF_LAUNCH_CHECK();
    return 0;
}
PDF_API int pdf_mano_split_coeff_bwd(const float* params, float* dparams, int ldp, long HW, const long* ind, const float* K, int B, int input_res, int down,
                                     const float* dorient, const float* dpose, const float* dtrans, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(mano_split_coeff_kernel, dim3(B, 2), dim3(64), 0, s, params, dparams, ldp, HW, ind, K, B, input_res / down, down,
                       const_cast<float*>(dorient), const_cast<float*>(dpose), nullptr, const_cast<float*>(dtrans), 1);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// C[b][m][n] = (beta ? C : 0) + sum_{r < RB} sum_k A[b, r][m][k] * B[b, r][k][n] with full strides
// (element strides; a batch stride of 0 broadcasts; RB > 1 folds a second batch axis into the
// reduction, which is how shared-weight gradients are summed over the batch).
__global__ void bmm_strided_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C,
                                   int M, int N, int K, int RB,
                                   long sab, long sar, long sam, long sak, long sbb, long sbr, long sbk, long sbn,
                                   long scb, long scm, long scn, int beta, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int n = (int)(i % N); long p = i / N;
        int m = (int)(p % M); long b = p / M;
        float acc = 0.f;
        for (int r = 0; r < RB; ++r) {
            const float* a = A + b * sab + r * sar + m * sam;
            const float* bb = Bm + b * sbb + r * sbr + n * sbn;
            for (int k = 0; k < K; ++k) acc += a[k * sak] * bb[k * sbk];
        }
        float* c = C + b * scb + m * scm + n * scn;
        *c = beta ? *c + acc : acc;
    }
}

// long reductions over few outputs (the 21 x 778 joint regressor applied to B x 2 meshes: 4,032 outputs, K = 778): one WAVE per
// output element, lanes stride over k, shuffle reduction -- the thread-per-output form above walks K serially (100 us per call)
__global__ __launch_bounds__(256) void bmm_strided_wave_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C,
                                                                int M, int N, int K, int RB,
                                                                long sab, long sar, long sam, long sak, long sbb, long sbr, long sbk, long sbn,
                                                                long scb, long scm, long scn, int beta, long total) {
    const int lane = threadIdx.x & 63;
    for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6; i < total; i += ((long)gridDim.x * blockDim.x) >> 6) {
        const int n = (int)(i % N); const long p = i / N;
        const int m = (int)(p % M); const long b = p / M;
        float acc = 0.f;
        for (int r = 0; r < RB; ++r) {
            const float* a = A + b * sab + r * sar + m * sam;
            const float* bb = Bm + b * sbb + r * sbr + n * sbn;
            for (int k = lane; k < K; k += 64) acc += a[k * sak] * bb[k * sbk];
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            float* c = C + b * scb + m * scm + n * scn;
            *c = beta ? *c + acc : acc;
        }
    }
}

PDF_API int pdf_bmm_strided(const float* A, const float* Bm, float* C, int batch, int M, int N, int K, int RB,
                            long sab, long sar, long sam, long sak, long sbb, long sbr, long sbk, long sbn,
                            long scb, long scm, long scn, int beta, hipStream_t s) {
    long total = (long)batch * M * N;
    if (total <= 0) return 0;
    if ((long)K * RB >= 128 && total <= (1L << 16))
        hipLaunchKernelGGL(bmm_strided_wave_kernel, dim3(grid_for(total * 64)), dim3(256), 0, s, A, Bm, C, M, N, K, RB,
                           sab, sar, sam, sak, sbb, sbr, sbk, sbn, scb, scm, scn, beta, total);
    else
        hipLaunchKernelGGL(bmm_strided_kernel, dim3(grid_for(total)), dim3(256), 0, s, A, Bm, C, M, N, K, RB,
                           sab, sar, sam, sak, sbb, sbr, sbk, sbn, scb, scm, scn, beta, total);
    PDF_LAUNCH_CHECK();
    return 0;
}
