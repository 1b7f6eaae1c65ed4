// Depth -> per-hand point cloud on the GPU.  Replaces the reference's CPU/numpy `depth2pcl`
// (lib/models/networks/intaghand_encoder.py:369-491, lib/utils/utils.py:251-275 get_points_coordinate / get_normal with
// with_normal=False), which forces a device->host->device round trip and batch size 1 in test/demo mode:
//   d = depth * (0.2 < depth < 2.5) * (mask > 0.5);  xyz = K^-1 [x, y, 1]^T * d
//   z-window: mean z over non-zero pixels +- 0.08 m (clamped to [0.2, 2.5]);  candidates = pixels strictly inside
//   n < 10 -> all-zero indices;  n > 1024 -> uniformly random subset of 1024;  else wrap-pad to 1024;  random order
//   cloud[i] = xyz[choose[i]].
// One 1024-thread workgroup per (sample, hand).  The random subset is the 1024 smallest of a stateless per-pixel hash
// (bitwise K-th-smallest search with workgroup counts, as in knn_ball_group), the random order a bitonic sort by hash in
// LDS -- deterministic for a given seed, statistically equivalent to numpy's shuffle (which no device code can replay).
#include "common.h"

#define FE_T 1024
#define FE_N 1024      // points per cloud (opt.SAMPLE_NUM)

__device__ __forceinline__ int block_sum_i(int v, int* red) {
    v = wave_sum_i(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    int t = 0;
    for (int i = 0; i < FE_T / 64; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_sum_f(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < FE_T / 64; ++i) t += red[i];
    return t;
}

// sort 1024 (key, val) pairs ascending by (key, val) -- bitonic network in LDS, one element per thread
__device__ __forceinline__ void bitonic1024(unsigned long long* kv) {
    const int t = threadIdx.x;
    for (int k = 2; k <= FE_N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            const int o = t ^ j;
            if (o > t) {
                const unsigned long long a = kv[t], b = kv[o];
                const bool up = (t & k) == 0;
                if ((a > b) == up) { kv[t] = b; kv[o] = a; }
            }
        }
    __syncthreads();
}

__global__ __launch_bounds__(FE_T) void depth2pcl_kernel(const float* __restrict__ depth, const float* __restrict__ mask,
                                                        const float* __restrict__ Kmat, const float* __restrict__ valid,
                                                        int H, int W, unsigned long long seed,
                                                        long* __restrict__ choose, float* __restrict__ cloud, int* __restrict__ count) {
    __shared__ unsigned long long kv[FE_N];
    __shared__ int redi[FE_T / 64];
    __shared__ float redf[FE_T / 64];
    __shared__ float Ki[9];
    __shared__ int s_cnt;
    const int b = blockIdx.x >> 1, hand = blockIdx.x & 1;           // hand 0 = left, 1 = right
    const int HW = H * W, t = threadIdx.x;
    const float* dep = depth + (long)b * HW;
    const float* msk = mask + ((long)b * 2 + (hand == 0 ? 1 : 0)) * HW;   // mask channel order = (right, left)  (:376-377)
    if (t == 0) {                                                      // K^-1 by cofactors (np.linalg.inv in the reference)
        const float* k = Kmat + b * 9;
        const float a = k[0], bb = k[1], c = k[2], d = k[3], e = k[4], f = k[5], g = k[6], h = k[7], i = k[8];
        const float det = a * (e * i - f * h) - bb * (d * i - f * g) + c * (d * h - e * g);
        const float id = 1.f / det;
        Ki[0] = (e * i - f * h) * id; Ki[1] = (c * h - bb * i) * id; Ki[2] = (bb * f - c * e) * id;
        Ki[3] = (f * g - d * i) * id; Ki[4] = (a * i - c * g) * id;  Ki[5] = (c * d - a * f) * id;
        Ki[6] = (d * h - e * g) * id; Ki[7] = (bb * g - a * h) * id; Ki[8] = (a * e - bb * d) * id;
        s_cnt = 0;
    }
    __syncthreads();
    auto zof = [&](int p) -> float {
        const float dv = dep[p];
        const float d = (dv > 0.2f && dv < 2.5f && msk[p] > 0.5f) ? dv : 0.f;
        const int y = p / W, x = p - y * W;
        return (Ki[6] * (float)x + Ki[7] * (float)y + Ki[8]) * d;
    };
    const bool is_valid = valid[b * 2 + hand] == 1.f;
    // mean depth over the non-zero pixels
    float zs = 0.f; int zc = 0;
    for (int p = t; p < HW; p += FE_T) { const float z = zof(p); if (z != 0.f) { zs += z; ++zc; } }
    const float zsum = block_sum_f(zs, redf);
    const int zcnt = block_sum_i(zc, redi);
    const float mean = zcnt > 0 ? zsum / (float)zcnt : 0.f;
    const float lo = fmaxf(0.2f, mean - 0.08f), hi = fminf(2.5f, mean + 0.08f);
    int nc = 0;
    for (int p = t; p < HW; p += FE_T) { const float z = zof(p); nc += (z > lo && z < hi) ? 1 : 0; }
    const int n = (is_valid && zcnt > 0) ? block_sum_i(nc, redi) : 0;
    if (t == 0 && count != nullptr) count[blockIdx.x] = n;
    long* cho = choose + (long)blockIdx.x * FE_N;
    float* cl = cloud + (long)blockIdx.x * FE_N * 3;
    int mine = 0;                                                       // this thread's output pixel index
    if (n >= 10) {
        const unsigned long long salt = seed * 0x9E3779B97F4A7C15ull + (unsigned long long)blockIdx.x * 0x632BE59BD9B4E019ull;
        if (n > FE_N) {
            // random subset: the FE_N candidates with the smallest hash.  31-bit keys, K-th smallest by bit search.
            unsigned int thr = 0;
            for (int bit = 30; bit >= 0; --bit) {
                const unsigned int cand = thr | (1u << bit);
                int c = 0;
                for (int p = t; p < HW; p += FE_T) {
                    const float z = zof(p);
                    if (z > lo && z < hi) c += ((pdf_hash32((unsigned int)p ^ (unsigned int)salt) ^ (unsigned int)(salt >> 32)) >> 1) < cand ? 1 : 0;
                }
                if (block_sum_i(c, redi) < FE_N) thr = cand;
            }
            for (int p = t; p < HW; p += FE_T) {                       // below the threshold: all in
                const float z = zof(p);
                if (z > lo && z < hi) {
                    const unsigned int key = (pdf_hash32((unsigned int)p ^ (unsigned int)salt) ^ (unsigned int)(salt >> 32)) >> 1;
                    if (key < thr) kv[atomicAdd(&s_cnt, 1)] = ((unsigned long long)key << 32) | (unsigned int)p;
                }
            }
            __syncthreads();
            for (int p = t; p < HW; p += FE_T) {                       // ties at the threshold fill the rest
                const float z = zof(p);
                if (z > lo && z < hi) {
                    const unsigned int key = (pdf_hash32((unsigned int)p ^ (unsigned int)salt) ^ (unsigned int)(salt >> 32)) >> 1;
                    if (key == thr) { const int pos = atomicAdd(&s_cnt, 1); if (pos < FE_N) kv[pos] = ((unsigned long long)key << 32) | (unsigned int)p; }
                }
            }
            __syncthreads();
            bitonic1024(kv);                                            // random order = order of the hash
            mine = (int)(kv[t] & 0xffffffffu);
        } else {
            // wrap-pad (np.pad(..., 'wrap')): candidates in index order, repeated
            kv[t] = 0xffffffffffffffffull;
            __syncthreads();
            for (int p = t; p < HW; p += FE_T) {
                const float z = zof(p);
                if (z > lo && z < hi) kv[atomicAdd(&s_cnt, 1)] = (unsigned long long)(unsigned int)p;
            }
            __syncthreads();
            bitonic1024(kv);                                            // ascending pixel index, padding last
            const int src = (int)(kv[t % n] & 0xffffffffu);
            __syncthreads();
            // shuffle: sort the padded list by a per-slot hash
            const unsigned int key = pdf_hash32((unsigned int)t ^ (unsigned int)(salt >> 13)) ^ (unsigned int)salt;
            kv[t] = ((unsigned long long)key << 32) | (unsigned int)src;
            bitonic1024(kv);
            mine = (int)(kv[t] & 0xffffffffu);
        }
    }
    cho[t] = mine;
    float X = 0.f, Y = 0.f, Z = 0.f;
    if (is_valid) {                                                     // invalid hand: zero cloud (:449-454)
        const float dv = dep[mine];
        const float d = (dv > 0.2f && dv < 2.5f && msk[mine] > 0.5f) ? dv : 0.f;
        const int y = mine / W, x = mine - y * W;
        X = (Ki[0] * (float)x + Ki[1] * (float)y + Ki[2]) * d;
        Y = (Ki[3] * (float)x + Ki[4] * (float)y + Ki[5]) * d;
        Z = (Ki[6] * (float)x + Ki[7] * (float)y + Ki[8]) * d;
    }
    cl[t * 3 + 0] = X; cl[t * 3 + 1] = Y; cl[t * 3 + 2] = Z;
}

PDF_API int pdf_depth2pcl(const float* depth, const float* mask, const float* K, const float* valid, int B, int H, int W,
                          unsigned long long seed, long* choose, float* cloud, int* count, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(depth2pcl_kernel, dim3(2 * B), dim3(FE_T), 0, s, depth, mask, K, valid, H, W, seed, choose, cloud, count);
    PDF_LAUNCH_CHECK();
    return 0;
}
