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

PDF_API int pdf_bmm_strided(const float* A, const float* Bm, float* C, int batch, int M, int N, int K, int RB,
                            long sab, long sar, long sam, long sak, long sbb, long sbr, long sbk, long sbn,
                            long scb, long scm, long scn, int beta, hipStream_t s) {
    long total = (long)batch * M * N;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(bmm_strided_kernel, dim3(grid_for(total)), dim3(256), 0, s, A, Bm, C, M, N, K, RB,
                       sab, sar, sam, sak, sbb, sbr, sbk, sbn, scb, scm, scn, beta, total);
    PDF_LAUNCH_CHECK();
    return 0;
}
