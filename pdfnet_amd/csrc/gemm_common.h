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
    short dy[MAX_TAPS], dx[MAX_TAPS], wt[MAX_TAPS];
};

// the word masked lanes read instead of branching around their load
static __device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void xcd_tile(int bid, int nblk, int ntn, int& tm, int& tn) {
    // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous chunk of the
    // tile list so the n-tiles that re-read one A panel hit the same L2 (guide T1, bijective form)
    int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    tn = lin % ntn;
    tm = lin / ntn;
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
    short dy[MAX_TAPS], dx[MAX_TAPS], wt[MAX_TAPS];
};


// bf16-input MFMA path (gemm_bf16.hip): same descriptors, operands rounded to bf16 while they are staged into LDS,
// fp32 accumulation.  Return 1 if the launch was issued, 0 if the shape is not supported there (the caller then uses the
// fp32 kernel), negative / hipError on failure.
int launch_igemm_bf16(const IGemm& g, hipStream_t s, int groups);
int launch_wgemm_bf16(const WGemm& g, int splits, int groups, int small, hipStream_t s);
