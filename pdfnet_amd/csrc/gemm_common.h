// Descriptors and helpers shared by the fp32 (gemm.hip) and bf16 (gemm_bf16.hip) MFMA GEMM kernels.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MAX_TAPS 64

struct IGemm {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, Cin;
    int lda, ldb;
    int H, W, QH, QW, sy, sx, T;
    int plain_in, plain_out;
    int OH, OW, osy, osx, ooy, oox, ldc;
    int ps_cout, ps_kw;
    int act;                                  // 0 none, 1 relu, 2 leaky-relu(0.1)
    // B layout: 0 = [N][K] rows (element (n, tap, ci) at n*ldb + wt[tap]*Cin + ci);  1 = [K][N] rows (element at
    // ci*ldb + wt[tap]*btap + n): the backward-data / transposed-conv contractions read the weight in its forward storage
    int b_kn, btap;
    // group 1 of a paired launch (blockIdx.y == 1): same shapes, own weight/bias, A and C advanced by gsA / gsC floats
    const float* B1; const float* bias1; long gsA, gsC;
    // split-K (blockIdx.z = split): K-steps [z*ksteps, (z+1)*ksteps) of this launch's BK, raw partial sums to
    // part[z][M][N] (no bias / activation -- splitk_finish applies them).  ksteps == 0: no split.
    int ksteps; float* part;
    // bf16 shadows of the operands (bf16 mode, optional): the same elements as A / B (B1) already rounded to bf16 (RNE) by their
    // producer, same layout with the same leading dimensions in ELEMENTS -- the bf16 kernels then stage 2-byte operands
    // straight into LDS (half the L2 -> LDS bytes, no conversion); results are bit-identical to rounding while staging
    const void* A16; const void* B16; const void* B116;
    int accum;                                // C += result (after bias / activation): a gradient accumulated into an existing one
    // BatchNorm statistics of the OUTPUT out of the accumulator registers (forward of a conv / linear layer that feeds a
    // BatchNorm): per output tile row-block t and column c, stat[(t * N + c) * 2 + {0, 1}] = (mean, sum of squared deviations)
    // of the stored values C[m][c] over the block's rows m in [t BM, min(M, (t+1) BM)).  Chan-combined per lane -> half-wave
    // pair -> waves in a fixed order (deterministic); pdf_bn_train_fwd combines the row-blocks in fp64 (norm.hip).
    float* stat;
    // byte extents of A / B from their (group-adjusted) base pointers for the buffer-descriptor form of the kernels (0: not used)
    unsigned int abytes, bbytes;
    unsigned int cbytes;                      // extent of a dense row-major C (plain_out, no pixel-shuffle) for buffer stores, else 0
    // bf16 storage mode (gemm_bf16.hip, whole tiles only): the output goes to C16 (same shape, ldc in ELEMENTS) rounded to bf16
    // (RNE) INSTEAD of C -- a conv output that only a training BatchNorm reads; statistics are taken from the rounded values
    unsigned short* C16;
    // A = relu(A * a_scale[k] + a_shift[k]) while the tile is staged (plain GEMMs, fp32 kernels): the BatchNorm + ReLU in front of a
    // PointNet++ linear layer applied by its consumer -- the normalised tensor is never written (functional._BatchNorm lazy=True)
    const float* a_scale; const float* a_shift;
    // b_kn launches in bf16 mode: the same B as a [N][K] ROW operand -- the weight's transposed bf16 shadow, element (n, tap, c) at
    // n * ldbT + wt[tap] * Cin + c (PdfCallOpts::op1_bf16_t) -- so that the LDS-DMA kernel can take a backward-data launch
    const void* B16T; int ldbT;
    // batched plain GEMMs in one launch (winograd.hip: the 16 transform-domain products): blockIdx.y = batch index b, operands
    // advanced by b * gsA / gsB / gsC floats (batch > 0 replaces the two-group meaning of blockIdx.y)
    int batch; long gsB;
    int gm;                                          // tile order: row-tiles per group (xcd_tile); 0 / 1 = rows of tiles one after the other
    int dy[MAX_TAPS], dx[MAX_TAPS], wt[MAX_TAPS];    // int, not short: a uniform index then compiles to s_load_dword; 16-bit entries become vector loads whose vmcnt(0) wait drains the prefetch
};

// the word masked lanes read instead of branching around their load
static __device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// gm > 1 (round 5, IGemm::gm): the tile list is walked in groups of gm row-tiles -- within a group the row index runs fastest -- so that the ~32
// blocks an XCD has in flight form a gm x (32 / gm) patch of the output instead of a 1 x 32 strip and share gm A panels AND 32 / gm B panels
// through its L2 (the "group-M" order of the GEMM literature).
__device__ __forceinline__ void xcd_tile(int bid, int nblk, int ntn, int& tm, int& tn, int gm = 1) {
    // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous chunk of the
    // tile list so the n-tiles that re-read one A panel hit the same L2 (guide T1, bijective form)
    int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    if (gm > 1) {
        const int ntm = nblk / ntn, per = gm * ntn, grp = lin / per, first = grp * gm, rem = lin - grp * per;
        const int gsz = min(ntm - first, gm);
        tm = first + rem % gsz;
        tn = rem / gsz;
        return;
    }
    tn = lin % ntn;
    tm = lin / ntn;
}

// Statistics epilogue shared by igemm_nt / igemm_halo3x3 (see IGemm::stat).  A lane holds, per column tile j, TM*16 values of
// ONE column; `StatAcc` collects them as shifted sums around the lane's first value.
struct StatAcc { float n, s, a, b; };
__device__ __forceinline__ void stat_add(StatAcc& t, float v) {
    if (t.n == 0.f) t.s = v;
    const float d = v - t.s;
    t.n += 1.f; t.a += d; t.b = fmaf(d, d, t.b);
}
// (n, mean, M2) of two disjoint row sets of one column -> of their union
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float n2, float mean2, float m22) {
    const float nt = n + n2;
    if (nt > 0.f) {
        const float d = mean2 - mean, f = n2 / nt;
        mean = fmaf(d, f, mean);
        m2 = m2 + m22 + d * d * n * f;
    }
    n = nt;
}
// sm: >= WM * BN * 3 floats of LDS nobody reads any more; every thread of the block must call this (it has a barrier)
template <int TN, int WM, int WN, int BN>
__device__ __forceinline__ void stat_finish(const StatAcc (&acc)[TN], float* sm, float* __restrict__ stat, int tile_row, int n0, int N,
                                            int wm, int wn, int lane, int tid) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        float n = acc[j].n;
        float mean = n > 0.f ? acc[j].s + acc[j].a / n : 0.f;
        float m2 = n > 0.f ? fmaxf(acc[j].b - acc[j].a * acc[j].a / n, 0.f) : 0.f;
        chan_merge(n, mean, m2, __shfl_xor(n, 32, 64), __shfl_xor(mean, 32, 64), __shfl_xor(m2, 32, 64));     // the other 16*TM rows of this column
        if (lane < 32) {
            float* o = sm + ((wm * BN) + wn * TN * 32 + j * 32 + lane) * 3;
            o[0] = n; o[1] = mean; o[2] = m2;
        }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < N) {
        float n = sm[tid * 3], mean = sm[tid * 3 + 1], m2 = sm[tid * 3 + 2];
#pragma unroll
        for (int w = 1; w < WM; ++w) chan_merge(n, mean, m2, sm[(w * BN + tid) * 3], sm[(w * BN + tid) * 3 + 1], sm[(w * BN + tid) * 3 + 2]);
        float* o = stat + ((long)tile_row * N + n0 + tid) * 2;
        o[0] = mean; o[1] = m2;
    }
}

// Epilogue of a WHOLE BM x BN tile of a dense row-major C (IGemm::cbytes != 0, every row of the tile < M): bias, activation,
// optional accumulate, store and the BatchNorm statistics.  Stores go through a buffer descriptor -- one per-lane byte offset per
// 32x32 block, the 16 row offsets as scalar operands, columns past N dropped by the range check -- instead of a 64-bit address,
// two compares and an exec-mask branch per element.  Measured (round 3): the per-tile instruction overhead, not memory, bounded
// the short-reduction layers -- with all loads or all stores removed a K = 64 layer kept 80-90 % of its time; this form took the
// ResNet 1x1 / PointNet++ linear layers from 52-70 to 70-88 TFLOP/s (fp32).  Lane layout of a 32x32 MFMA block: column lane & 31,
// rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  sm: LDS for stat_finish (only touched when g.stat != nullptr).
template <int TM, int TN, int WM, int WN, int BN>
__device__ __forceinline__ void lean_epilogue(const f32x16 (&acc)[TM][TN], const IGemm& g, float* Cp, const float* __restrict__ biasp,
                                              int m0, int n0, int tile_row, int wm, int wn, int lane, int tid, float* sm) {
    const auto rsC = __builtin_amdgcn_make_buffer_rsrc((void*)Cp, 0, g.cbytes, 0x00020000);
    const unsigned ldc4 = (unsigned)g.ldc * 4u;
    const bool do_stat = g.stat != nullptr;
    StatAcc st[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) st[j] = StatAcc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + (lane & 31);
        const bool cok = col < g.N;
        const float bv = (biasp != nullptr && cok) ? biasp[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned vo = cok ? (unsigned)((m0 + wm * TM * 32 + i * 32 + 4 * (lane >> 5)) * g.ldc + col) * 4u : 0xffffffffu;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bv;
            if (g.act == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (g.act == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : 0.1f * v[r];
            }
            if (g.accum) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    v[r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsC, vo, ((r & 3) + 8 * (r >> 2)) * ldc4, 0));
            }
            if (g.C16 != nullptr) {
                // bf16 output.  Lanes l and l ^ 1 hold adjacent columns: they swap one value per row pair so that each lane stores
                // ONE packed pair -- even lanes the even row of the pair, odd lanes the odd row -- 8 four-byte stores per lane.
                const auto rsH = __builtin_amdgcn_make_buffer_rsrc((void*)g.C16, 0, g.cbytes >> 1, 0x00020000);
                const bool odd = lane & 1;
                const unsigned vh = cok ? ((vo >> 1) - (odd ? 2u : 0u)) : 0xffffffffu;
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = __uint_as_float(pdf_pk_bf16(v[r], v[r]) << 16);        // the value BatchNorm will read
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r0 = 2 * q, r1 = r0 + 1;
                    const float other = __shfl_xor(odd ? v[r0] : v[r1], 1, 64);
                    const unsigned pk = odd ? ((__float_as_uint(other) >> 16) | (__float_as_uint(v[r1]) & 0xffff0000u))
                                            : ((__float_as_uint(v[r0]) >> 16) | (__float_as_uint(other) & 0xffff0000u));
                    const unsigned ro = (unsigned)(((odd ? r1 : r0) & 3) + 8 * ((odd ? r1 : r0) >> 2)) * (ldc4 >> 1);
                    __builtin_amdgcn_raw_buffer_store_b32(pk, rsH, cok ? vh + ro : 0xffffffffu, 0, 0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), rsC, vo, ((r & 3) + 8 * (r >> 2)) * ldc4, 0);
            }
            if (do_stat && cok) {                           // stat_add without the row test: same operations in the same order
                if (i == 0) st[j].s = v[0];
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = v[r] - st[j].s; st[j].a += d; st[j].b = fmaf(d, d, st[j].b); }
                st[j].n += 16.f;
            }
            __builtin_amdgcn_sched_barrier(0);              // one 32x32 block at a time: 16 live values, not TM * TN * 16 (occupancy of the 128x128 kernels)
        }
    }
    if (do_stat) stat_finish<TN, WM, WN, BN>(st, sm, g.stat, tile_row, n0, g.N, wm, wn, lane, tid);
}

struct WGemm {
    const float* P; const float* Q; float* slab;
    int M, NI, Cq, T;
    int ldp, ldq, ldw;
    int H, W, QH, QW, sy, sx;
    int plain_q;
    int rows_per_split;
    int tap_major;                            // tile order: channel-block major, taps inner (same XCD re-reads the same pixels)
    int beta;                                 // single-split launches write dW directly: dW = beta*dW + acc
    // group 1 of a paired launch (blockIdx.z == 1): P, Q advanced by gsP / gsQ floats, own slab (or output when one split)
    long gsP, gsQ; float* slab1;
    // optional bias gradient (column sums of P) riding along: per-split partials [split][NI] (or the output itself when one
    // split), accumulated by the j-tile-0 blocks from the P tiles they stage anyway
    float* bslab; float* bslab1;
    // in-launch reduction of the split slabs (pdf_last_block_arrives): the last block of every output tile sums the slabs in split
    // order into out / out1 (+= when accumulate) and the bias partials into bout / bout1; counters == NULL: reduce_slabs launch
    float* out; float* out1; float* bout; float* bout1; int accumulate; int* counters;
    // atomic != 0 (several splits, accumulate): every block adds its partial tile straight into out / out1 (bout / bout1) with
    // global_atomic_add_f32 -- no slabs, no reduction pass; the summation order then varies from run to run
    int atomic;
    const void* P16; const void* Q16;         // bf16 shadows of P / Q (see IGemm::A16)
    const float* q_scale; const float* q_shift;      // Q = relu(Q * q_scale[c] + q_shift[c]) on the fly (see IGemm::a_scale); plain_q only
    // byte extents of P / Q as seen from their (group-adjusted) base pointers, for the buffer descriptors of wgemm_tn_dma<.., true>
    // (0: an operand is >= 4 GiB - 32 and the flat-address form of the kernel is used)
    unsigned int pbytes, qbytes;
    unsigned int wbytes;                      // bytes of ONE split's result matrix (NI * ldw floats) for buffer stores, 0: flat stores
    // wgemm_tn_dma<.., true>: the 16 rows of a K-step are two runs of 8 consecutive pixels of ONE image row each (QW % 8 == 0)
    // and the tile's 128 columns belong to one tap (Cq % 128 == 0), or Q is plain rows: the pixel position, the tap and all
    // offsets except a per-thread constant are block-uniform, i.e. scalar-unit work
    int uniform;
    // batched launch (winograd.hip weight gradients: one transform-domain plane per blockIdx.z): P, Q advanced by z * gsP / gsQ floats,
    // the slabs of plane z start at slab + z * gsW (batch > 0 replaces the two-group meaning of blockIdx.z; no bias partials)
    int batch; long gsW;
    // bsplits > 0 (batched launch only): a ONE-dimensional grid whose linear id carries (tile, split, plane) so that the tiles of one
    // (split, plane) -- the blocks that read the same rows of P_b and Q_b -- are consecutive dispatches of ONE XCD and share its L2:
    // id & 7 = XCD, (id >> 3) = slot; group = (slot / tiles) * 8 + XCD = plane * bsplits + split, tile = slot % tiles
    int bsplits;
    int dy[MAX_TAPS], dx[MAX_TAPS], wt[MAX_TAPS];    // int, not short: a uniform index then compiles to s_load_dword; 16-bit entries become vector loads whose vmcnt(0) wait drains the prefetch
};


// Epilogue shared by the weight-gradient kernels: this block's partial tile -> its split's slab (or the output itself when there
// is one split), the bias partial, and -- in-launch -- the reduction over the splits by the tile's last-arriving block.
// bias_thread: this thread carries the bias partial `bval` of output row i0 + threadIdx.x.
template <int TM, int TN>
__device__ __forceinline__ void wgemm_finish(const WGemm& g, const f32x16 (&acc)[TM][TN], int i0, int j0, int wm, int wn, int lane,
                                             bool bias_thread, float bval, int tile_id, int ntiles, int* lds_flag, bool rows_whole = false,
                                             int split_ = -1, int plane_ = 0) {
    // (split_ >= 0: the caller decoded split / plane from a linear grid, WGemm::bsplits)
    const int grp = g.batch > 0 ? 0 : blockIdx.z, split = split_ >= 0 ? split_ : (int)blockIdx.y, splits = split_ >= 0 ? g.bsplits : (int)gridDim.y;
    float* slabp = g.batch > 0 ? g.slab + (long)(split_ >= 0 ? plane_ : (int)blockIdx.z) * g.gsW : (grp ? g.slab1 : g.slab);
    float* bslabp = grp ? g.bslab1 : g.bslab;
    const int NJ = g.T * g.Cq;
    if (g.atomic) {
        if (bias_thread) atomicAdd((grp ? g.bout1 : g.bout) + i0 + threadIdx.x, bval);
        float* dst = grp ? g.out1 : g.out;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = j0 + wn * TN * 32 + j * 32 + (lane & 31);
            if (col >= NJ) continue;
            const int t = col / g.Cq;
            const int wcol = g.wt[t] * g.Cq + (col - t * g.Cq);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < g.NI) atomicAdd(dst + (long)row * g.ldw + wcol, acc[i][j][r]);
                }
        }
        return;
    }
    if (bias_thread) {
        float* bo = bslabp + (long)split * g.NI + i0 + threadIdx.x;
        *bo = g.beta ? *bo + bval : bval;
    }
    float* out = slabp + (long)split * g.NI * g.ldw;
    if (g.wbytes != 0 && rows_whole) {
        // every row of the tile is inside the matrix: buffer stores (per-lane offset once per 32x32 block, row offsets as scalar
        // operands, columns past NJ dropped by the range check) -- see lean_epilogue
        const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, g.wbytes, 0x00020000);
        const unsigned ldw4 = (unsigned)g.ldw * 4u;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = j0 + wn * TN * 32 + j * 32 + (lane & 31);
            const bool cok = col < NJ;
            const int t = cok ? col / g.Cq : 0;
            const int wcol = g.wt[t] * g.Cq + (col - t * g.Cq);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned vo = cok ? (unsigned)((i0 + wm * TM * 32 + i * 32 + 4 * (lane >> 5)) * g.ldw + wcol) * 4u : 0xffffffffu;
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
                if (g.beta) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        v[r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsW, vo, ((r & 3) + 8 * (r >> 2)) * ldw4, 0));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), rsW, vo, ((r & 3) + 8 * (r >> 2)) * ldw4, 0);
            }
        }
    } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = j0 + wn * TN * 32 + j * 32 + (lane & 31);
        if (col >= NJ) continue;
        const int t = col / g.Cq;
        const int wcol = g.wt[t] * g.Cq + (col - t * g.Cq);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.NI) {
                    float* o = out + (long)row * g.ldw + wcol;
                    *o = g.beta ? *o + acc[i][j][r] : acc[i][j][r];
                }
            }
    }
    }
    if (g.counters == nullptr || splits == 1) return;
    if (!pdf_last_block_arrives(g.counters + grp * ntiles + tile_id, splits, lds_flag, true)) return;
    float* dst = grp ? g.out1 : g.out;
    const long per = (long)g.NI * g.ldw;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = j0 + wn * TN * 32 + j * 32 + (lane & 31);
        if (col >= NJ) continue;
        const int t = col / g.Cq;
        const int wcol = g.wt[t] * g.Cq + (col - t * g.Cq);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.NI) {
                    const long pos = (long)row * g.ldw + wcol;
                    float sum = g.accumulate ? dst[pos] : 0.f;
                    for (int z = 0; z < splits; ++z) sum += slabp[z * per + pos];
                    dst[pos] = sum;
                }
            }
    }
    if (bslabp != nullptr && j0 == 0 && threadIdx.x < TM * 64 && i0 + (int)threadIdx.x < g.NI) {      // TM * 64 = BI rows of this tile
        float* bdst = grp ? g.bout1 : g.bout;
        float sum = g.accumulate ? bdst[i0 + threadIdx.x] : 0.f;
        for (int z = 0; z < splits; ++z) sum += bslabp[(long)z * g.NI + i0 + threadIdx.x];
        bdst[i0 + threadIdx.x] = sum;
    }
}

// Per-kernel timing for bench.py's roofline (pdf_debug_kernel_timing, gemm.hip): while enabled, every GEMM-family kernel launch
// is bracketed by two events on ITS launch stream and recorded under its symbol name with the algorithmic FLOPs / bytes of
// that launch.  Disabled (the default) it costs one branch per launch.
struct KTimer {
    KTimer(const char* name, double flops, double bytes, hipStream_t s);
    ~KTimer();
    int slot; hipStream_t stream;
};

// bf16-input MFMA path (gemm_bf16.hip): same descriptors, operands rounded to bf16 while they are staged into LDS,
// fp32 accumulation.  Return 1 if the launch was issued, 0 if the shape is not supported there (the caller then uses the
// fp32 kernel), negative / hipError on failure.
int launch_igemm_bf16(const IGemm& g, hipStream_t s, int groups);
int igemm_bf16_tile_rows(const IGemm& g, int groups);
int launch_wgemm_bf16(const WGemm& g, int splits, int groups, int small, hipStream_t s);
// LDS-DMA form of the fp32 implicit GEMM (gemm_dma.hip).  tile: 64 (64x64) | 128 (128x128) | 12864 | 64128; variant: ring depth / K-step
// choice; splits > 0: split-K launch (g.ksteps / g.part set).  -> 1 launched, 0 shape not taken.
int launch_igemm_dma(const IGemm& g, int tile, int variant, int groups, int splits, hipStream_t s);
int launch_igemm_bf16_dma(const IGemm& g, int tile, int variant, int groups, hipStream_t s);
