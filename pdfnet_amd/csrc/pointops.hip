// PointNet++ set-abstraction data movement for gfx950: kNN(K) + ball mask + grouping in ONE launch,
// row gathers/scatters on NHWC maps, and the max-over-neighbours pool.  All HBM-bound: coalesced
// row reads, the cloud staged once per 16 centroids in LDS, wavefront ballots for the selection.
#include "common.h"
#include <cstdlib>
#include <mutex>

// ---------------------------------------------------------------------------------------------
// knn_ball_group: replaces group_points / group_points_2 (reference lib/utils/utils.py:134-188):
//   centroids = first S points; K smallest squared distances; neighbours with d2 > r2 replaced by the
//   centroid's own index (utils.py:149-151,176-179); gather all C features, subtract the centre from
//   channels 0:3 (utils.py:160,186).  Distances use the reference's operation order
//   ((dx*dx + dy*dy) + dz*dz) with no fma contraction so the selected index SETS are bit-exact.
// One wave per centroid: each lane keeps N/64 keys in registers; the K-th smallest key is found by a
// 31-step bitwise search with wave-wide counts (no sort, no [S][N] distance matrix in memory).
template <int NPL>   // points per lane = N / 64
__global__ __launch_bounds__(256) void knn_ball_group_kernel(
    const float* __restrict__ pts, int ldp, int C, int N, int S, int K, float r2,
    int* __restrict__ idx_out, float* __restrict__ grouped, int ldg) {
    extern __shared__ float smem[];
    float* sx = smem;            // [N]
    float* sy = sx + N;
    float* sz = sy + N;
    int* sidx = reinterpret_cast<int*>(sz + N);   // [4 waves][K]
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* cloud = pts + (long)b * N * ldp;
    for (int j = tid; j < N; j += 256) {
        sx[j] = cloud[(long)j * ldp + 0];
        sy[j] = cloud[(long)j * ldp + 1];
        sz[j] = cloud[(long)j * ldp + 2];
    }
    __syncthreads();
    int* widx = sidx + wave * K;
    const uint32_t r2bits = __float_as_uint(r2);
    for (int cc = 0; cc < 4; ++cc) {
        const int s = blockIdx.x * 16 + wave * 4 + cc;      // wave-uniform
        if (s >= S) break;
        const float cx = sx[s], cy = sy[s], cz = sz[s];
        uint32_t key[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            int j = lane + 64 * i;
            const int jj = j < N ? j : 0;
            float dx = __fsub_rn(sx[jj], cx), dy = __fsub_rn(sy[jj], cy), dz = __fsub_rn(sz[jj], cz);
            float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            // d >= 0: uint order == float order.  Slots past the cloud's end (N not a multiple of 64 * NPL: the reference takes any
            // N, lib/utils/utils.py:134-163) hold a key above every distance and every search candidate: never counted, never selected
            key[i] = j < N ? __float_as_uint(d) : 0xFFFFFFFFu;
        }
        uint32_t thr = 0;
        for (int bit = 30; bit >= 0; --bit) {
            uint32_t cand = thr | (1u << bit);
            int c = 0;
#pragma unroll
            for (int i = 0; i < NPL; ++i) c += key[i] < cand ? 1 : 0;
            c = wave_sum_i(c);
            if (c < K) thr = cand;                          // K-th smallest key is >= cand
        }
        // compaction: everything below the threshold, then ties in increasing point index
        int base = 0;
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            bool sel = key[i] < thr;
            unsigned long long m = __ballot(sel);
            if (sel) {
                int pos = base + __popcll(m & lt_mask);
                widx[pos] = key[i] > r2bits ? s : lane + 64 * i;
            }
            base += __popcll(m);
        }
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            bool sel = key[i] == thr;
            unsigned long long m = __ballot(sel);
            if (sel) {
                int pos = base + __popcll(m & lt_mask);
                if (pos < K) widx[pos] = key[i] > r2bits ? s : lane + 64 * i;
            }
            base += __popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        // NB: widx is wave-private LDS; LDS ops of one wave complete in order
        const long orow = ((long)b * S + s) * K;
        for (int k = lane; k < K; k += 64) idx_out[orow + k] = widx[k];
        if (grouped != nullptr) {
            if (ldg <= 16) {
                // few channels: one neighbour per lane, whole padded row from one lane
                for (int k = lane; k < K; k += 64) {
                    int j = widx[k];
                    float* o = grouped + (orow + k) * ldg;
                    const float* p = cloud + (long)j * ldp;
                    for (int c = 0; c < ldg; ++c) {
                        float v = 0.f;
                        if (c < C) v = p[c];
                        if (c == 0) v = __fsub_rn(v, cx); else if (c == 1) v = __fsub_rn(v, cy); else if (c == 2) v = __fsub_rn(v, cz);
                        if (c >= C) v = 0.f;
                        o[c] = v;
                    }
                }
            } else {
                for (int k = 0; k < K; ++k) {
                    int j = widx[k];
                    const float* p = cloud + (long)j * ldp;
                    float* o = grouped + (orow + k) * ldg;
                    for (int c = lane; c < ldg; c += 64) {
                        float v = c < C ? p[c] : 0.f;
                        if (c == 0) v = __fsub_rn(v, cx); else if (c == 1) v = __fsub_rn(v, cy); else if (c == 2) v = __fsub_rn(v, cz);
                        o[c] = v;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

PDF_API int pdf_knn_ball_group(const float* pts, int ldp, int C, int Bc, int N, int S, int K, float r2,
                               int* idx, float* grouped, int ldg, hipStream_t s) {
    if (N < 1 || N > 1024 || K > N || K < 1 || S > N || C < 3 || ldp < C || (grouped && ldg < C)) return PDF_E_BADARG;
    dim3 grid(cdiv(S, 16), Bc);
    size_t smem = (size_t)3 * N * sizeof(float) + (size_t)4 * K * sizeof(int);
    int npl = 1;                                                // points per lane: the power of two that covers N (padded slots: see the kernel)
    while (npl * 64 < N) npl <<= 1;
#define KNN_CASE(NPL_) case NPL_: hipLaunchKernelGGL(knn_ball_group_kernel<NPL_>, grid, dim3(256), smem, s, pts, ldp, C, N, S, K, r2, idx, grouped, ldg); break;
    switch (npl) {
        KNN_CASE(1) KNN_CASE(2) KNN_CASE(4) KNN_CASE(8) KNN_CASE(16)
        default: return PDF_E_BADARG;
    }
#undef KNN_CASE
    PDF_LAUNCH_CHECK();
    return 0;
}

// backward of the grouping gather: dpts (zero-filled by caller) [Bc][N][ldd]
//   dpts[b][idx[b,s,k]][c] += dg[b,s,k,c];  dpts[b][s][c<3] -= sum_k dg[b,s,k,c]
__global__ __launch_bounds__(256) void group_bwd_kernel(const float* __restrict__ dg, int ldg, const int* __restrict__ idx,
                                                        float* __restrict__ dpts, int ldd, int C, int N, int S, int K, long total_rows) {
    const int lane = threadIdx.x & 63;
    const long w0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long row = w0; row < total_rows; row += nw) {       // row = b*S + s
        const long b = row / S;
        const int s = (int)(row - b * S);
        float* base = dpts + b * N * ldd;
        float csum = 0.f;                                    // lanes 0..2: centre gradient
        for (int k = 0; k < K; ++k) {
            const int j = idx[row * K + k];
            const float* g = dg + (row * K + k) * ldg;
            for (int c = lane; c < C; c += 64) {
                float v = g[c];
                atomicAdd(base + (long)j * ldd + c, v);
                if (c < 3) csum += v;
            }
        }
        if (lane < 3) atomicAdd(base + (long)s * ldd + lane, -csum);
    }
}

PDF_API int pdf_group_bwd(const float* dg, int ldg, const int* idx, float* dpts, int ldd, int C,
                          int Bc, int N, int S, int K, hipStream_t s) {
    long rows = (long)Bc * S;
    hipLaunchKernelGGL(group_bwd_kernel, dim3(grid_for(rows * 64)), dim3(256), 0, s, dg, ldg, idx, dpts, ldd, C, N, S, K, rows);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// First layer of a set-abstraction MLP without the grouped tensor.  The reference gathers K neighbours per centroid, subtracts
// the centre from xyz (utils.py:153-160,181-186) and applies a 1x1 convolution to the [C, S, K] block; a 1x1 convolution is
// linear, so  W (p_idx - c_pad) + b = (W p + b)[idx] - W c_pad:  the convolution runs ONCE PER POINT (N rows instead of
// S*K = 32x / 16x as many) and this kernel gathers its rows:  y[b,s,k,:] = u[b, idx[b,s,k], :] - v[b,s,:].
// The [Bc,S,K,C_in] grouped tensor (2.1 / 4.7 MB per cloud) is never written or read.
__global__ __launch_bounds__(256) void gather_sub_fwd_kernel(const float* __restrict__ u, int ldu, const float* __restrict__ v, int ldv,
                                                             const int* __restrict__ idx, int N, int S, int K, int C,
                                                             float* __restrict__ y, int ldy, long total /* rows * C/4 */) {
    const int cq = C / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / cq;                             // (b*S + s)*K + k
        const int c0 = (int)(i - row * cq) * 4;
        const long bs = row / K;
        const long b = bs / S;
        const float4 a = *reinterpret_cast<const float4*>(u + (b * N + idx[row]) * ldu + c0);
        const float4 c = *reinterpret_cast<const float4*>(v + bs * ldv + c0);
        *reinterpret_cast<float4*>(y + row * ldy + c0) = make_float4(a.x - c.x, a.y - c.y, a.z - c.z, a.w - c.w);
    }
}
PDF_API int pdf_gather_sub_fwd(const float* u, int ldu, const float* v, int ldv, const int* idx, int Bc, int N, int S, int K, int C,
                               float* y, int ldy, hipStream_t s) {
    if (C % 4 != 0 || ldu % 4 != 0 || ldv % 4 != 0 || ldy % 4 != 0) return PDF_E_BADARG;
    const long total = (long)Bc * S * K * (C / 4);
    if (total == 0) return 0;
    hipLaunchKernelGGL(gather_sub_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, u, ldu, v, ldv, idx, N, S, K, C, y, ldy, total);
    PDF_LAUNCH_CHECK();
    return 0;
}
// backward: du (zero-filled by the caller) [Bc][N][ldu] += dy rows scattered by idx (float atomics at the memory side);
// dv[b,s,:] = -sum_k dy[b,s,k,:] (one wave owns a centroid: plain store)
__global__ __launch_bounds__(256) void gather_sub_bwd_kernel(const float* __restrict__ dy, int lddy, const int* __restrict__ idx,
                                                             float* __restrict__ du, int ldu, float* __restrict__ dv, int ldv,
                                                             int N, int S, int K, int C, long total_rows) {
    const int lane = threadIdx.x & 63;
    const long w0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long row = w0; row < total_rows; row += nw) {       // row = b*S + s
        const long b = row / S;
        float* base = du + b * N * ldu;
        float cs[4] = {0.f, 0.f, 0.f, 0.f};                  // up to 256 channels
        for (int k = 0; k < K; ++k) {
            const int j = idx[row * K + k];
            const float* g = dy + (row * K + k) * lddy;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = lane + 64 * t;
                if (c < C) {
                    const float x = g[c];
                    atomicAdd(base + (long)j * ldu + c, x);
                    cs[t] += x;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = lane + 64 * t;
            if (c < C) dv[row * ldv + c] = -cs[t];
        }
    }
}
// ---- deterministic backward of the gather: the index is inverted ONCE per forward (per cloud: for every point the list of
// (centroid, neighbour) slots e = s*K + k that picked it, ascending), and the backward sums each point's rows of dy in list order
// -- no float atomics, bit-reproducible, every du row written exactly once (no zero fill).
//   start [Bc][N + 1] : list segment of point n = [start[n], start[n + 1])      list [Bc][E] : slots, ascending within a segment
// One block per cloud, a STABLE counting sort with no sorting pass: the slots are cut into one contiguous range per wave; every wave
// histograms its range (LDS integer atomics: order-free), a scan turns the [wave][point] counts into cursors
// (segment start + what the earlier waves hold of that point), and every wave then walks its range IN ORDER, 64 slots per round:
// a lane's position is its cursor plus the number of LOWER lanes of the round with the same point (ballots over the key's bits), so the
// segments come out ascending without atomics in the fill and without the rank sort this kernel used to end with -- that sort was
// O(L^2) per segment and, on clouds where a few points own hundreds of slots (an invalid hand's all-zero cloud, ball-query padding),
// made the launch 110-175 us for 2 MB of indices (profiles/r03, r04_kernel_stats_exclusive.csv).  Same output, bit for bit.
#define INV_NT 1024
#define INV_KR 32                                            // rounds of 64 slots whose keys a lane keeps in registers (16 waves x 32 x 64 = 32,768 slots: level 1's S * K)
__global__ __launch_bounds__(INV_NT) void invert_index_kernel(const int* __restrict__ idx, int N, int E, int nw, int* __restrict__ start,
                                                              int* __restrict__ list) {
    extern __shared__ int sm[];                              // [nw][N] counts -> cursors, [N + 1] starts
    int* cnt = sm;
    int* st = sm + (long)nw * N;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int* id = idx + (long)b * E;
    int* lst = list + (long)b * E;
    for (int i = tid; i < nw * N; i += INV_NT) cnt[i] = 0;
    __syncthreads();
    const int per = ((E + nw - 1) / nw + 63) & ~63;          // slots per wave (whole rounds)
    const int e0 = wave * per, e1 = min(E, e0 + per);
    // this lane's keys of the wave's first INV_KR rounds, loaded ONCE (one memory latency for both passes instead of one per round);
    // (indices are clamped into [0, N): an index outside the cloud must not reach past the LDS arrays -- ADVICE r3)
    int keys[INV_KR];
#pragma unroll
    for (int r = 0; r < INV_KR; ++r) {
        const int e = e0 + r * 64 + lane;
        keys[r] = (wave < nw && e < e1) ? min(max(id[e], 0), N - 1) : -1;
    }
    auto key_of = [&](int r, int e) { return e < e1 ? min(max(id[e], 0), N - 1) : -1; };      // rounds past INV_KR (E > nw * 64 * INV_KR)
    if (wave < nw) {
#pragma unroll
        for (int r = 0; r < INV_KR; ++r) if (keys[r] >= 0) atomicAdd(&cnt[wave * N + keys[r]], 1);
        for (int eb = e0 + INV_KR * 64; eb < e1; eb += 64) { const int k = key_of(0, eb + lane); if (k >= 0) atomicAdd(&cnt[wave * N + k], 1); }
    }
    __syncthreads();
    for (int n = tid; n < N; n += INV_NT) {                  // per point: counts of the waves -> offsets inside the segment; total -> st (scanned below)
        int run = 0;
        for (int w = 0; w < nw; ++w) { const int c = cnt[w * N + n]; cnt[w * N + n] = run; run += c; }
        st[n] = run;
    }
    __syncthreads();
    if (wave == 0) {                                         // exclusive scan of the totals by one wave, 64 points per round
        int run = 0;
        for (int n0 = 0; n0 < N; n0 += 64) {
            const int n = n0 + lane;
            const int c = n < N ? st[n] : 0;
            int inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
            if (n < N) st[n] = run + inc - c;
            run += __shfl(inc, 63, 64);
        }
        if (lane == 0) st[N] = run;
    }
    __syncthreads();
    for (int n = tid; n <= N; n += INV_NT) start[(long)b * (N + 1) + n] = st[n];
    if (wave >= nw) return;
    int* cur = cnt + wave * N;                               // this wave's cursors, relative to the segment starts
    const unsigned long long lt = (1ull << lane) - 1ull;    // lanes below this one
    const int kbits = 32 - __clz(max(N - 1, 1));
    auto place = [&](int key, int e) {
        // the lanes of the round holding the same point: intersect, bit by bit of the key, the ballots of "my bit" (<= 15 steps instead of
        // 64 v_readlane compares); a lane's position = cursor + the number of LOWER such lanes; the lowest of them advances the cursor
        unsigned long long same = __ballot(key >= 0);
        for (int i = 0; i < kbits; ++i) {
            const bool bit = (key >> i) & 1;
            const unsigned long long m = __ballot(bit);
            same &= bit ? m : ~m;
        }
        if (key >= 0) {
            const int below = __popcll(same & lt);
            lst[st[key] + cur[key] + below] = e;
            if (below == 0) cur[key] += __popcll(same);      // (one lane per point of the round; the wave's LDS accesses execute in order)
        }
    };
#pragma unroll
    for (int r = 0; r < INV_KR; ++r) {
        if (e0 + r * 64 >= e1) break;                        // (wave-uniform)
        place(keys[r], e0 + r * 64 + lane);
    }
    for (int eb = e0 + INV_KR * 64; eb < e1; eb += 64) place(key_of(0, eb + lane), eb + lane);
}
PDF_API int pdf_invert_index(const int* idx, int Bc, int N, int E, int* start, int* list, int* tmp, hipStream_t s) {
    (void)tmp;
    if (Bc <= 0 || E <= 0) return 0;
    if (N <= 0) return PDF_E_BADARG;
    const long room = 160 * 1024 / 4 - (N + 1);              // LDS words left for the per-wave histograms
    int nw = (int)(room / N);
    if (nw > INV_NT / 64) nw = INV_NT / 64;
    if (nw < 1) return PDF_E_BADARG;
    const size_t lds = ((size_t)nw * N + N + 1) * 4;
    static std::once_flag once;
    std::call_once(once, [] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(invert_index_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL(invert_index_kernel, dim3(Bc), dim3(INV_NT), lds, s, idx, N, E, nw, start, list);
    PDF_LAUNCH_CHECK();
    return 0;
}
// du[b, n, :] = sum over the slots of point n, in list order, of dy[b, slot, :].  CL lanes (one float4 each) cover a point's C <= 4 CL channels and a
// wave takes 64 / CL points at once: with one wave per point the 64-channel level ran 16 of its 64 lanes (2.0 TB/s).  The
// summation order per element is unchanged (list order, four rows in flight), so the results are bit-identical to the one-wave-per-point form.
template <int CL>
__global__ __launch_bounds__(256) void gather_sub_bwd_du_kernel(const float* __restrict__ dy, int lddy, const int* __restrict__ start,
                                                                const int* __restrict__ list, float* __restrict__ du, int ldu,
                                                                int N, int E, int C, long total_points) {
    constexpr int PPW = 64 / CL;
    const int lane = threadIdx.x & 63, sub = lane / CL;
    const long w0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    const int c4 = (lane % CL) * 4;
    for (long p0 = w0 * PPW; p0 < total_points; p0 += nw * PPW) {
        const long pt = p0 + sub;
        if (pt >= total_points || c4 >= C) continue;
        const long b = pt / N;
        const int n = (int)(pt - b * N);
        const int* st = start + b * (N + 1);
        const int s0 = st[n], s1 = st[n + 1];
        const int* lst = list + b * E;
        const float* g = dy + b * E * lddy;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int j = s0;
        for (; j + 4 <= s1; j += 4) {                        // four rows in flight; summed in list order
            const float4 x0 = *reinterpret_cast<const float4*>(g + (long)lst[j] * lddy + c4);
            const float4 x1 = *reinterpret_cast<const float4*>(g + (long)lst[j + 1] * lddy + c4);
            const float4 x2 = *reinterpret_cast<const float4*>(g + (long)lst[j + 2] * lddy + c4);
            const float4 x3 = *reinterpret_cast<const float4*>(g + (long)lst[j + 3] * lddy + c4);
            acc.x = ((((acc.x + x0.x) + x1.x) + x2.x) + x3.x); acc.y = ((((acc.y + x0.y) + x1.y) + x2.y) + x3.y);
            acc.z = ((((acc.z + x0.z) + x1.z) + x2.z) + x3.z); acc.w = ((((acc.w + x0.w) + x1.w) + x2.w) + x3.w);
        }
        for (; j < s1; ++j) {
            const float4 x = *reinterpret_cast<const float4*>(g + (long)lst[j] * lddy + c4);
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        *reinterpret_cast<float4*>(du + pt * ldu + c4) = acc;
    }
}
// dv[b, s, :] = -sum_k dy[b, s, k, :]   (CL lanes per centroid, 64 / CL centroids per wave; k in order)
template <int CL>
__global__ __launch_bounds__(256) void gather_sub_bwd_dv_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ dv, int ldv,
                                                                int K, int C, long total_rows) {
    constexpr int PPW = 64 / CL;
    const int lane = threadIdx.x & 63, sub = lane / CL;
    const long w0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    const int c4 = (lane % CL) * 4;
    for (long r0 = w0 * PPW; r0 < total_rows; r0 += nw * PPW) {
        const long row = r0 + sub;
        if (row >= total_rows || c4 >= C) continue;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* g = dy + row * K * lddy + c4;
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            const float4 x = *reinterpret_cast<const float4*>(g + (long)k * lddy);
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        *reinterpret_cast<float4*>(dv + row * ldv + c4) = make_float4(-acc.x, -acc.y, -acc.z, -acc.w);
    }
}
PDF_API int pdf_gather_sub_bwd_sorted(const float* dy, int lddy, const int* start, const int* list, float* du, int ldu, float* dv, int ldv,
                                      int Bc, int N, int S, int K, int C, hipStream_t s) {
    if (C > 256 || C % 4 != 0 || lddy % 4 != 0 || ldu % 4 != 0 || ldv % 4 != 0) return PDF_E_BADARG;
    const long pts = (long)Bc * N, rows = (long)Bc * S;
    if (pts == 0 || rows == 0) return 0;
    if (C <= 64) {
        hipLaunchKernelGGL((gather_sub_bwd_du_kernel<16>), dim3(grid_for(pts * 16)), dim3(256), 0, s, dy, lddy, start, list, du, ldu, N, S * K, C, pts);
        hipLaunchKernelGGL((gather_sub_bwd_dv_kernel<16>), dim3(grid_for(rows * 16)), dim3(256), 0, s, dy, lddy, dv, ldv, K, C, rows);
    } else if (C <= 128) {
        hipLaunchKernelGGL((gather_sub_bwd_du_kernel<32>), dim3(grid_for(pts * 32)), dim3(256), 0, s, dy, lddy, start, list, du, ldu, N, S * K, C, pts);
        hipLaunchKernelGGL((gather_sub_bwd_dv_kernel<32>), dim3(grid_for(rows * 32)), dim3(256), 0, s, dy, lddy, dv, ldv, K, C, rows);
    } else {
        hipLaunchKernelGGL((gather_sub_bwd_du_kernel<64>), dim3(grid_for(pts * 64)), dim3(256), 0, s, dy, lddy, start, list, du, ldu, N, S * K, C, pts);
        hipLaunchKernelGGL((gather_sub_bwd_dv_kernel<64>), dim3(grid_for(rows * 64)), dim3(256), 0, s, dy, lddy, dv, ldv, K, C, rows);
    }
    PDF_LAUNCH_CHECK();
    return 0;
}

PDF_API int pdf_gather_sub_bwd(const float* dy, int lddy, const int* idx, float* du, int ldu, float* dv, int ldv,
                               int Bc, int N, int S, int K, int C, hipStream_t s) {
    if (C > 256) return PDF_E_BADARG;
    const long rows = (long)Bc * S;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(gather_sub_bwd_kernel, dim3(grid_for(rows * 64)), dim3(256), 0, s, dy, lddy, idx, du, ldu, dv, ldv, N, S, K, C, rows);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// gather_rows: replaces _tranpose_and_gather_feat (lib/models/utils.py:22-26) without the full-map
// permute: feat is already NHWC, so a gathered pixel is one contiguous row.
// shift > 0 applies the pyramid index math of intaghand_encoder.py:125-126:
//   ind' = (ind / R >> shift) * (R >> shift) + (ind % R >> shift)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ feat, int ldf, int C, long HW,
                                                          const long* __restrict__ ind, long ind_bstride, int M, int R, int shift,
                                                          float* __restrict__ out, int ldo, long total) {
    const int lane = threadIdx.x & 63;
    const long w0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long r = w0; r < total; r += nw) {                  // r = b*M + m
        const long b = r / M;
        long i = ind[b * ind_bstride + (r - b * M)];
        if (shift > 0) {
            long y = i / R, x = i - y * R;
            i = (y >> shift) * (R >> shift) + (x >> shift);
        }
        const float* p = feat + (b * HW + i) * ldf;
        float* o = out + r * ldo;
        for (int c = lane; c < ldo; c += 64) o[c] = c < C ? p[c] : 0.f;
    }
}

PDF_API int pdf_gather_rows(const float* feat, int ldf, int C, long HW, const long* ind, long ind_bstride,
                            int B, int M, int R, int shift, float* out, int ldo, hipStream_t s) {
    long total = (long)B * M;
    if (total == 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(total * 64)), dim3(256), 0, s, feat, ldf, C, HW, ind, ind_bstride, M, R, shift, out, ldo, total);
    PDF_LAUNCH_CHECK();
    return 0;
}

// dfeat (zero-filled by caller) [B][HW][ldf] += rows of dout
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ dout, int ldo, int C, long HW,
                                                           const long* __restrict__ ind, long ind_bstride, int M, int R, int shift,
                                                           float* __restrict__ dfeat, int ldf, long total) {
    const int lane = threadIdx.x & 63;
    const long w0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long r = w0; r < total; r += nw) {
        const long b = r / M;
        long i = ind[b * ind_bstride + (r - b * M)];
        if (shift > 0) {
            long y = i / R, x = i - y * R;
            i = (y >> shift) * (R >> shift) + (x >> shift);
        }
        float* p = dfeat + (b * HW + i) * ldf;
        const float* o = dout + r * ldo;
        for (int c = lane; c < C; c += 64) atomicAdd(p + c, o[c]);
    }
}

PDF_API int pdf_scatter_rows_add(const float* dout, int ldo, int C, long HW, const long* ind, long ind_bstride,
                                 int B, int M, int R, int shift, float* dfeat, int ldf, hipStream_t s) {
    long total = (long)B * M;
    if (total == 0) return 0;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(total * 64)), dim3(256), 0, s, dout, ldo, C, HW, ind, ind_bstride, M, R, shift, dfeat, ldf, total);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// max over the K axis of [R][K][ld] rows (nn.MaxPool2d((1,K)) / ((S2,1)), intaghand_encoder.py:62,82,100)
__global__ __launch_bounds__(256) void maxk_fwd_kernel(const float* __restrict__ x, int ldx, int C, int K,
                                                       float* __restrict__ y, int ldy, int* __restrict__ arg, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        const float* p = x + r * K * ldx + c;
        float best = p[0];
        int bi = 0;
        for (int k = 1; k < K; ++k) {
            float v = p[(long)k * ldx];
            if (v > best) { best = v; bi = k; }
        }
        y[r * ldy + c] = best;
        arg[r * C + c] = bi;
    }
}

PDF_API int pdf_maxk_fwd(const float* x, int ldx, int C, long R, int K, float* y, int ldy, int* arg, hipStream_t s) {
    long total = R * C;
    if (total == 0) return 0;
    hipLaunchKernelGGL(maxk_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, ldx, C, K, y, ldy, arg, total);
    PDF_LAUNCH_CHECK();
    return 0;
}

// dx[r][k][c] = (k == arg[r][c]) ? dy[r][c] : 0   (writes every element of dx's C columns)
__global__ __launch_bounds__(256) void maxk_bwd_kernel(const float* __restrict__ dy, int ldy, const int* __restrict__ arg,
                                                       int C, int K, float* __restrict__ dx, int ldx, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long rk = i / C;
        const int c = (int)(i - rk * C);
        const long r = rk / K;
        const int k = (int)(rk - r * K);
        dx[rk * ldx + c] = (arg[r * C + c] == k) ? dy[r * ldy + c] : 0.f;
    }
}

PDF_API int pdf_maxk_bwd(const float* dy, int ldy, const int* arg, int C, long R, int K, float* dx, int ldx, hipStream_t s) {
    long total = R * K * C;
    if (total == 0) return 0;
    hipLaunchKernelGGL(maxk_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, dy, ldy, arg, C, K, dx, ldx, total);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Window gathers for the exact sparse evaluation of center_feat_up0 -> center_feat_up1 -> gather
// (reference intaghand_encoder.py:790-792 computes two full 3x3 convolutions, 48.3 GFLOP/img, and keeps
// 2 pixels per sample).  out[(b*M+m)][wy][wx][:] = feat[b][cy-r+wy][cx-r+wx][:] (zero outside the map),
// (cy, cx) = divmod(ind[b][m], W).  mode 0: gather; mode 1: scatter-add of `buf` into feat (backward);
// mode 2: in-place-safe mask: y = inside ? x : 0 on a [.., win, win, C] patch tensor (the 3x3 up0 patch must
// be ZERO where it falls outside the image -- conv padding pads up0, it does not convolve the padding).
__global__ __launch_bounds__(256) void window_kernel(float* __restrict__ feat, int ldf, int C, int H, int W,
                                                     const long* __restrict__ ind, long ind_bstride, int M, int r,
                                                     float* __restrict__ buf, const float* __restrict__ src, int mode, long total) {
    const int lane = threadIdx.x & 63;
    const int win = 2 * r + 1;
    const long w0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long p = w0; p < total; p += nw) {                  // p = ((b*M + m)*win + wy)*win + wx
        long q = p;
        const int wx = (int)(q % win); q /= win;
        const int wy = (int)(q % win); q /= win;
        const long b = q / M;
        const long i = ind[b * ind_bstride + (q - b * M)];
        const int y = (int)(i / W) - r + wy, x = (int)(i % W) - r + wx;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        float* row = buf + p * C;
        if (mode == 2) {
            const float* s = src + p * C;
            for (int c = lane; c < C; c += 64) row[c] = in ? s[c] : 0.f;
        } else {
            float* f = feat + ((b * H + y) * (long)W + x) * ldf;
            if (mode == 0) { for (int c = lane; c < C; c += 64) row[c] = in ? f[c] : 0.f; }
            else if (in) { for (int c = lane; c < C; c += 64) atomicAdd(f + c, row[c]); }
        }
    }
}

PDF_API int pdf_window_op(float* feat, int ldf, int C, int H, int W, const long* ind, long ind_bstride,
                          int B, int M, int r, float* buf, const float* src, int mode, hipStream_t s) {
    long total = (long)B * M * (2 * r + 1) * (2 * r + 1);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(window_kernel, dim3(grid_for(total * 64)), dim3(256), 0, s, feat, ldf, C, H, W, ind, ind_bstride, M, r, buf, src, mode, total);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Centre decode (test path): 5x5 max-pool NMS on the centre heat map + top-1 per (sample, channel).
// Replaces _nms + _topk (reference intaghand_encoder.py:349-367,750-758; lib/trains/simplified.py:378-384).
//   keep = (maxpool5x5(hm) == hm);  score = hm * keep;  ind = argmax(score)  (lowest index on ties, like a stable top-1)
// One block per (b, c) map; hm is [B][C][H][W] (plain NCHW, tiny) .
__global__ __launch_bounds__(256) void nms_top1_kernel(const float* __restrict__ hm, int H, int W, long* __restrict__ ind, float* __restrict__ score) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const float* m = hm + (long)blockIdx.x * H * W;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int p = threadIdx.x; p < H * W; p += 256) {
        const int y = p / W, x = p - y * W;
        const float v = m[p];
        float mx = v;
        for (int dy = -2; dy <= 2; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -2; dx <= 2; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                mx = fmaxf(mx, m[yy * W + xx]);
            }
        }
        const float sc = (mx == v) ? v : v * 0.f;          // heat * keep (keeps the sign of zero / NaN behaviour of the product)
        if (sc > best || (sc == best && p < bi)) { best = sc; bi = p; }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float o = sv[threadIdx.x + s]; const int oi = si[threadIdx.x + s];
            if (o > sv[threadIdx.x] || (o == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = o; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { ind[blockIdx.x] = si[0]; if (score) score[blockIdx.x] = sv[0]; }
}

PDF_API int pdf_nms_top1(const float* hm, int BC, int H, int W, long* ind, float* score, hipStream_t s) {
    if (BC <= 0) return 0;
    hipLaunchKernelGGL(nms_top1_kernel, dim3(BC), dim3(256), 0, s, hm, H, W, ind, score);
    PDF_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Farthest point sampling (lib/datasets/interhand.py:147-178 `farthest_point_sampling_fast`, the NumPy helper the reference
// keeps for `--sample_strategy FPS`, opts.py:231): idx[0] = start; idx[i] = argmax of the running minimum squared distance
// (first index on ties); distances are frozen once they are <= 1e-8 (:171-173).  One 1024-thread block per cloud, the
// points and their running distances live in registers (<= 16 per thread), one block-wide arg-max per picked point.
#define FPS_THREADS 1024
#define FPS_MAXPT 16
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float* __restrict__ xyz, int ld, int N, int S,
                                                          const int* __restrict__ start, int* __restrict__ idx) {
    __shared__ float s_val[FPS_THREADS / 64];
    __shared__ int s_idx[FPS_THREADS / 64];
    __shared__ float s_pick[3];
    __shared__ int s_best;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = xyz + (long)blockIdx.x * N * ld;
    int* out = idx + (long)blockIdx.x * S;
    float px[FPS_MAXPT], py[FPS_MAXPT], pz[FPS_MAXPT], d[FPS_MAXPT];
#pragma unroll
    for (int k = 0; k < FPS_MAXPT; ++k) {
        const int i = tid + k * FPS_THREADS;
        const bool ok = i < N;
        px[k] = ok ? p[(long)i * ld] : 0.f;
        py[k] = ok ? p[(long)i * ld + 1] : 0.f;
        pz[k] = ok ? p[(long)i * ld + 2] : 0.f;
    }
    int cur = start != nullptr ? start[blockIdx.x] : 0;
    cur = min(max(cur, 0), N - 1);
    for (int it = 0; it < S; ++it) {
        if (tid == 0) {
            out[it] = cur;
            s_pick[0] = p[(long)cur * ld]; s_pick[1] = p[(long)cur * ld + 1]; s_pick[2] = p[(long)cur * ld + 2];
        }
        __syncthreads();
        const float cx = s_pick[0], cy = s_pick[1], cz = s_pick[2];
        float best = -1.f; int bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < FPS_MAXPT; ++k) {
            const int i = tid + k * FPS_THREADS;
            if (i < N) {
                const float dx = px[k] - cx, dy = py[k] - cy, dz = pz[k] - cz;
                // (dx*dx + dy*dy) + dz*dz without fused multiply-adds: the NumPy helper's float32 arithmetic
                const float nd = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                if (it == 0) d[k] = nd;
                else if (d[k] > 1e-8f) d[k] = fminf(d[k], nd);
                if (d[k] > best) { best = d[k]; bi = i; }       // k ascending => i ascending: the first maximum of this thread
            }
        }
        // block arg-max, lowest index among equal values
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; }
        __syncthreads();
        if (wave == 0) {
            float v = lane < FPS_THREADS / 64 ? s_val[lane] : -2.f;
            int vi = lane < FPS_THREADS / 64 ? s_idx[lane] : 0x7fffffff;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                const float ov = __shfl_xor(v, o, 64);
                const int oi = __shfl_xor(vi, o, 64);
                if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
            }
            if (lane == 0) s_best = vi;
        }
        __syncthreads();
        cur = s_best;
    }
}
// Single-wave form for clouds of at most 1,024 points (the reference's SAMPLE_NUM, opts.py:228): ONE wave per cloud, 16 points per
// lane in registers, no __syncthreads and no LDS round trip in the arg-max (round 3: 1,024 threads, two block barriers and an LDS
// exchange per pick -- 1.77 us per pick for a cloud that fits one wave's registers).  Per pick: 16 x (distance, frozen-minimum
// update, running arg-max) of straight-line VALU work, a 64-bit key (distance bits << 32 | ~index: the maximum is the largest
// distance, lowest index on ties, distances being non-negative) reduced with six DPP steps (quad swaps, half-mirror, mirror, two row
// broadcasts), v_readlane of lane 63 into scalar registers, and the winner's coordinates read back from an LDS copy of the cloud at a
// wave-uniform address.  A block is 64 threads: 16 times the clouds in flight per CU.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long v) {
    const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROW_MASK, 0xf, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROW_MASK, 0xf, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o > v ? o : v;
}
#define FPS1_MAXN 1024
__global__ __launch_bounds__(64) void fps_wave_kernel(const float* __restrict__ xyz, int ld, int N, int S,
                                                      const int* __restrict__ start, int* __restrict__ idx) {
    __shared__ float sp[FPS1_MAXN * 3];
    const int lane = threadIdx.x;
    const float* p = xyz + (long)blockIdx.x * N * ld;
    int* out = idx + (long)blockIdx.x * S;
    float px[16], py[16], pz[16], d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = min(lane + k * 64, N - 1);               // (slots past N repeat the last point: equal distance, higher index -- never picked first)
        px[k] = p[(long)i * ld]; py[k] = p[(long)i * ld + 1]; pz[k] = p[(long)i * ld + 2];
        const int j = lane + k * 64;
        if (j < N) { sp[j * 3] = px[k]; sp[j * 3 + 1] = py[k]; sp[j * 3 + 2] = pz[k]; }
        d[k] = __builtin_inff();                               // first round: min(inf, nd) = nd
    }
    __syncthreads();                                           // (one wave: orders the LDS writes before the first read)
    int cur = start != nullptr ? start[blockIdx.x] : 0;
    cur = __builtin_amdgcn_readfirstlane(min(max(cur, 0), N - 1));
    for (int it = 0; it < S; ++it) {
        if (lane == 0) out[it] = cur;
        const float cx = sp[cur * 3], cy = sp[cur * 3 + 1], cz = sp[cur * 3 + 2];
        float best = -1.f; int bk = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float dx = px[k] - cx, dy = py[k] - cy, dz = pz[k] - cz;
            // (dx*dx + dy*dy) + dz*dz without fused multiply-adds: the NumPy helper's float32 arithmetic
            const float nd = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            d[k] = d[k] > 1e-8f ? fminf(d[k], nd) : d[k];
            if (d[k] > best) { best = d[k]; bk = k; }          // k ascending => index ascending: this lane's first maximum
        }
        const unsigned bi = (unsigned)(lane + bk * 64);
        unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (0xffffffffu - (bi < (unsigned)N ? bi : 0x7fffffffu));
        key = dpp_max_u64<0xB1, 0xf>(key);                    // quad_perm [1,0,3,2]
        key = dpp_max_u64<0x4E, 0xf>(key);                    // quad_perm [2,3,0,1]
        key = dpp_max_u64<0x141, 0xf>(key);                   // row_half_mirror
        key = dpp_max_u64<0x140, 0xf>(key);                   // row_mirror: every lane of a row holds the row's maximum
        key = dpp_max_u64<0x142, 0xa>(key);                   // row_bcast:15 into rows 1 and 3
        key = dpp_max_u64<0x143, 0xc>(key);                   // row_bcast:31 into rows 2 and 3: lane 63 holds the cloud's maximum
        const unsigned wlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, 63);
        cur = (int)(0xffffffffu - wlo);
    }
}
PDF_API int pdf_fps(const float* xyz, int ld, int Bc, int N, int S, const int* start, int* idx, hipStream_t s) {
    if (N <= 0 || S <= 0 || N > FPS_THREADS * FPS_MAXPT || ld < 3) return PDF_E_BADARG;
    if (Bc <= 0) return 0;
    static const bool wave_form = getenv("PDF_FPS_WAVE") == nullptr || atoi(getenv("PDF_FPS_WAVE")) != 0;
    if (N <= FPS1_MAXN && wave_form) {
        hipLaunchKernelGGL(fps_wave_kernel, dim3(Bc), dim3(64), 0, s, xyz, ld, N, S, start, idx);
        PDF_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(fps_kernel, dim3(Bc), dim3(FPS_THREADS), 0, s, xyz, ld, N, S, start, idx);
    PDF_LAUNCH_CHECK();
    return 0;
}
