// fp32 MFMA (v_mfma_f32_32x32x2_f32) implicit-GEMM kernels for gfx950.
//
// Every dense contraction of the PDFNet hot path is one of two kernels:
//   igemm_nt : C[m][n] = act( sum_{t,c} A[pos(m,t)][c] * B[n][wt[t]*Cin + c] + bias[n] )
//              A is an NHWC image gathered through a tap table (conv / transposed-conv parity
//              class / strided backward-data) or plain rows (Linear, 1x1 conv);  B is K-major.
//   wgemm_tn : dW[i][wt[t]*Cq + c] = sum_m P[m][i] * Q[pos(m,t)][c]      (weight gradients,
//              reduction over pixels, split over M into slabs that reduce_slabs() sums).
// Tiles are 64-wide-wavefront shaped: 256 threads = 4 waves, each wave owns (BM/WM)x(BN/WN) made of
// 32x32 MFMA tiles; K-step 16 staged through LDS with a register prefetch of the next tile.
// fp32-input MFMA is bit-for-bit an fmaf chain (guide: FP32-input MFMA), so results are exact fp32.
#include "common.h"

#include "gemm_common.h"

// winograd.hip
long pdf_internal_wino_workspace(int N, int H, int W, int Ck, int Cn, int flip);
int pdf_internal_wino_eligible(int N, int H, int W, int Ck, int Cn, int KH, int KW, int stride, int pad, int flip);
int pdf_internal_conv3x3_winograd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, float* ws,
                                  int N, int H, int W, int Ck, int Cn, int act, int accum, int flip, const float* v_shared, hipStream_t s);
int pdf_internal_wino_wgrad_eligible(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
long pdf_internal_wino_wgrad_workspace(int N, int H, int W, int Cin, int Cout);
long pdf_internal_wino_v_offset(int N, int H, int W, int Ck, int Cn);
int pdf_internal_conv3x3_winograd_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw, float* db, float* ws,
                                        int N, int H, int W, int Cin, int Cout, int accumulate, const float* v_cached, hipStream_t s);
// Workspace (floats) a stride-1 3x3 convolution [Cout][3][3][Cin] on N x H x W maps wants for its Winograd path -- backward = 0: the
// forward pass, 1: backward-data, 2: the weight gradient -- or 0 when the layer does not qualify (then no workspace is needed)
PDF_API long pdf_conv2d_winograd_v_offset(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (!pdf_internal_wino_eligible(N, H, W, Cin, Cout, KH, KW, stride, pad, 0) || !pdf_internal_wino_wgrad_eligible(N, H, W, Cin, Cout, KH, KW, stride, pad)) return -1;
    return pdf_internal_wino_v_offset(N, H, W, Cin, Cout);
}
PDF_API long pdf_conv2d_winograd_workspace_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int backward) {
    if (backward == 2) return pdf_internal_wino_wgrad_eligible(N, H, W, Cin, Cout, KH, KW, stride, pad) ? pdf_internal_wino_wgrad_workspace(N, H, W, Cin, Cout) : 0;
    const int Ck = backward ? Cout : Cin, Cn = backward ? Cin : Cout;
    return pdf_internal_wino_eligible(N, H, W, Ck, Cn, KH, KW, stride, pad, backward) ? pdf_internal_wino_workspace(N, H, W, Ck, Cn, backward) : 0;
}

// Read the LDS fragments of the next k-pair while the MFMAs of the current one run (see igemm_nt).  Compile-time switch for A/B runs.
#ifndef PDF_FRAG_PIPE
#define PDF_FRAG_PIPE 0
#endif
constexpr bool FRAG_PIPE = PDF_FRAG_PIPE != 0;
#ifndef PDF_WG_ISSUE_AT
#define PDF_WG_ISSUE_AT 1                       // wgemm_tn_dma, scalar-offset form: the k-pair after which the next tile's loads are issued
#endif
#ifndef PDF_IG_DEEP
#define PDF_IG_DEEP 1
#endif
constexpr bool IG_DEEP = PDF_IG_DEEP != 0;      // igemm_nt, buffer-load form: two K-steps of prefetch in flight
// Round 5: operand fragments as ONE ds_read_b128 per 32 rows and 8 k.  Lane l reads the four consecutive k of chunk (l >> 5) of row (l & 31);
// MFMA t of the group pairs k = 8 g + t (lanes 0-31) with k = 8 g + 4 + t (lanes 32-63) -- any pairing is a valid reduction order as long
// as A and B use the same.  Rows are BK + 4 floats apart (16-byte aligned, stride = 4 mod 8 floats: conflict-free for b128), so the staging
// store is one ds_write_b128 per thread and row too: 4x fewer LDS instructions on both sides than the [BK + 1] image with scalar accesses.
// MEASURED (profiles/r05_ig_b128.txt): no layer gains, the [K][N] backward-data forms lose 5-10 %, the step 50.7 vs 49.8 ms -- LDS instruction
// issue is not what bounds these loops (the what-if builds of round 3 said the same of the loads).  Off; kept as a compile-time switch.
#ifndef PDF_IG_B128
#define PDF_IG_B128 0
#endif
constexpr bool IG_B128 = PDF_IG_B128 != 0;
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 as_f4(const u32x4v& v) { return *reinterpret_cast<const float4*>(&v); }

// BUF (FAST only): both operands are fetched through buffer descriptors with 32-bit byte offsets; a masked element (row past M,
// padded tap, column past N) is the offset 0xffffffff, which the hardware's range check returns as zeros.  Per K-step and load
// that is one add, one select and the load.  The flat-address form costs ~25 instructions per load -- 64-bit multiply-adds under
// exec-mask branches, the zero word's address re-read from the GOT behind an s_waitcnt lgkmcnt(0) -- ~240 instructions per
// K-step of the 64x64 tile against 16 MFMAs: with four waves per SIMD the VALU, not the matrix pipe, was the bound.
// AFF (BUF, plain GEMMs): IGemm::a_scale / a_shift are applied to the A tile while it is staged -- a compile-time variant with its own
// kernel name (igemm_nt_aff), because even the unused run-time test cost the 64x64 tile 3 % (89 -> 86 TFLOP/s over the step's launches)
template <int BM, int BN, int WM, int WN, bool FAST, bool KN, int BKT, bool BUF, bool AFF>
__device__ __forceinline__ void igemm_nt_body(const IGemm& g) {    // (128x128: 3 waves per SIMD = 3 blocks per CU, as its LDS allows)
    constexpr int NT = WM * WN * 64;                     // threads: one wave per (BM/WM) x (BN/WN) sub-tile
    constexpr int BK = BKT, LD = IG_B128 ? BK + 4 : BK + 1;      // K-step: 16, or 32 for the small tile (half the barriers per flop)
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int TPR = BK / 4, RPP = NT / TPR;          // threads per staged row (a float4 each), rows per pass of the block
    constexpr int RA = BM / RPP, RB = BN / RPP;
    __shared__ __attribute__((aligned(16))) float As[2][BM * LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const float* __restrict__ Ap = g.A; const float* __restrict__ Bp = g.B; const float* __restrict__ biasp = g.bias;
    float* __restrict__ Cp = g.C;
    if (g.batch > 0) { Ap += (long)blockIdx.y * g.gsA; Bp += (long)blockIdx.y * g.gsB; Cp += (long)blockIdx.y * g.gsC; }
    else if (blockIdx.y) { Ap += g.gsA; Cp += g.gsC; Bp = g.B1; biasp = g.bias1; }
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    int tmi, tni;
    xcd_tile(blockIdx.x, ntm * ntn, ntn, tmi, tni, g.gm);
    const int m0 = tmi * BM, n0 = tni * BN;

    const int lrow = tid / TPR, kq = (tid % TPR) * 4;
    long abase[RA]; int iy0[RA], ix0[RA]; bool aval[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int r = m0 + lrow + i * RPP;
        aval[i] = r < g.M;
        if (g.plain_in) {
            abase[i] = (long)r * g.lda; iy0[i] = 0; ix0[i] = 0;
        } else {
            int hw = g.QH * g.QW;
            int ni = r / hw, rem = r - ni * hw;
            int qy = rem / g.QW, qx = rem - qy * g.QW;
            iy0[i] = qy * g.sy; ix0[i] = qx * g.sx;
            abase[i] = (long)ni * g.H * g.W * g.lda;
        }
    }
    // B staging.  [N][K] storage: thread -> (row n = lrow + 64 i, 4 consecutive k).  [K][N] storage (FAST): thread ->
    // (k row kb[i], 4 consecutive n) -- the float4 runs along n and is scattered into the same Bs[n][k] image.
    long bbase[RB]; bool bval[RB]; int kb[RB];
    const int nq4 = (tid % (BN / 4)) * 4;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        if (FAST && KN) {
            kb[i] = (tid + i * NT) / (BN / 4);
            bval[i] = n0 + nq4 < g.N;
            bbase[i] = (long)kb[i] * g.ldb + n0 + nq4;
        } else {
            int n = n0 + lrow + i * RPP;
            kb[i] = 0;
            bval[i] = n < g.N;
            bbase[i] = KN ? (long)n : (long)n * g.ldb;
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = (g.K + BK - 1) / BK;
    int kt0 = 0, kt1 = nk;                               // split-K launch: this block's K-steps
    if (g.ksteps > 0) { kt0 = blockIdx.z * g.ksteps; kt1 = min(nk, kt0 + g.ksteps); }

    float4 ra[RA], rb[RB];
    float4 ra1[RA], rb1[RB];                             // second register set of the two-steps-ahead prefetch (IG_DEEP)
    int nload = kt0;                                     // K-step the next gload fetches (BUF: a step past kt1 loads zeros, no traffic)
    // tap state of the NEXT tile to load (tiles are loaded strictly in order): no per-tile division and the
    // tap table (scalar loads that share lgkmcnt with the LDS traffic) is read only when the tap changes
    int nt_tap = 0, nt_ci = kt0 * BK;
    if (g.ksteps > 0 && g.T > 1) { nt_tap = nt_ci / g.Cin; nt_ci -= nt_tap * g.Cin; }      // split-K over taps: Cin % BK == 0 (launch_igemm)
    int ddy = g.dy[nt_tap], ddx = g.dx[nt_tap], wbase = g.wt[nt_tap] * (KN ? g.btap : g.Cin);
    // ---- BUF state: per-row byte offsets (fixed for the whole kernel), per-row validity under the current tap
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, BUF ? g.abytes : 0, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, BUF ? g.bbytes : 0, 0x00020000);
    unsigned aoffB[RA], boffB[RB]; bool aok[RA];
    int tapoffB = 0;
    auto tap_valid = [&]() {
        tapoffB = g.plain_in ? 0 : (ddy * g.W + ddx) * g.lda * 4;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int iy = iy0[i] + ddy, ix = ix0[i] + ddx;
            aok[i] = aval[i] && (g.plain_in || (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W));
        }
    };
    if constexpr (BUF) {
#pragma unroll
        for (int i = 0; i < RA; ++i) aoffB[i] = (unsigned)(abase[i] + ((long)iy0[i] * g.W + ix0[i]) * g.lda + kq) * 4u;
#pragma unroll
        for (int i = 0; i < RB; ++i) boffB[i] = (unsigned)(bbase[i] + (KN ? 0 : kq)) * 4u;
        tap_valid();
    }
    auto gload = [&](int kt, float4 (&ra)[RA], float4 (&rb)[RB]) {
        if constexpr (FAST && BUF) {
            const bool live = nload < kt1;
            ++nload;
            const unsigned sa = (unsigned)(tapoffB + nt_ci * 4);
            const unsigned sb = KN ? (unsigned)((nt_ci * g.ldb + wbase) * 4) : (unsigned)((wbase + nt_ci) * 4);
#pragma unroll
            for (int i = 0; i < RA; ++i) ra[i] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsA, (aok[i] && live) ? aoffB[i] + sa : 0xffffffffu, 0, 0));
#pragma unroll
            for (int i = 0; i < RB; ++i) rb[i] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsB, (bval[i] && live) ? boffB[i] + sb : 0xffffffffu, 0, 0));
            if constexpr (AFF) {                             // BatchNorm + ReLU of the producer applied here (plain GEMM: k = channel)
                const int ch = min(nt_ci + kq, g.K - 4);
                const float4 sc = *reinterpret_cast<const float4*>(g.a_scale + ch), sh = *reinterpret_cast<const float4*>(g.a_shift + ch);
#pragma unroll
                for (int i = 0; i < RA; ++i) {               // (rows past M become relu(shift): they only reach output rows nobody stores)
                    ra[i].x = fmaxf(fmaf(ra[i].x, sc.x, sh.x), 0.f); ra[i].y = fmaxf(fmaf(ra[i].y, sc.y, sh.y), 0.f);
                    ra[i].z = fmaxf(fmaf(ra[i].z, sc.z, sh.z), 0.f); ra[i].w = fmaxf(fmaf(ra[i].w, sc.w, sh.w), 0.f);
                }
            }
            nt_ci += BK;
            if (nt_ci >= g.Cin && nt_tap + 1 < g.T) {
                ++nt_tap; nt_ci = 0;
                ddy = g.dy[nt_tap]; ddx = g.dx[nt_tap]; wbase = g.wt[nt_tap] * (KN ? g.btap : g.Cin);
                tap_valid();
            }
        } else if (FAST) {
            const int ci0 = nt_ci + kq;
            const long wofs = KN ? (long)nt_ci * g.ldb + wbase : (long)(wbase + ci0);
            nt_ci += BK;
            const bool tap_done = nt_ci >= g.Cin;
#pragma unroll
            // branch-free: a masked element reads a 16-byte zero word through a SELECTED ADDRESS.  (`if (ok) v = load`
            // compiles to exec-mask branches with an s_waitcnt vmcnt(0) per load, which serialises the tile's loads.)
            for (int i = 0; i < RA; ++i) {
                const float* src;
                if (g.plain_in) src = aval[i] ? Ap + abase[i] + kt * BK + kq : g_zero16;
                else {
                    const int iy = iy0[i] + ddy, ix = ix0[i] + ddx;
                    const bool ok = aval[i] && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                    src = ok ? Ap + abase[i] + ((long)iy * g.W + ix) * g.lda + ci0 : g_zero16;
                }
                ra[i] = *reinterpret_cast<const float4*>(src);
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const float* src = bval[i] ? Bp + bbase[i] + wofs : g_zero16;
                rb[i] = *reinterpret_cast<const float4*>(src);
            }
            if (tap_done && nt_tap + 1 < g.T) {
                ++nt_tap; nt_ci = 0;
                ddy = g.dy[nt_tap]; ddx = g.dx[nt_tap]; wbase = g.wt[nt_tap] * (KN ? g.btap : g.Cin);
            }
        } else {
            float tmpa[RA][4], tmpb[RB][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int k = kt * BK + kq + j;
                bool kin = k < g.K;
                int t = kin ? k / g.Cin : 0;
                int ci = k - t * g.Cin;
                const int gdy = g.dy[t], gdx = g.dx[t];
                const long wofs = KN ? (long)ci * g.ldb + g.wt[t] * g.btap : (long)(g.wt[t] * g.Cin + ci);
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    float v = 0.f;
                    if (kin && aval[i]) {
                        if (g.plain_in) v = Ap[abase[i] + k];
                        else {
                            int iy = iy0[i] + gdy, ix = ix0[i] + gdx;
                            if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                                v = Ap[abase[i] + ((long)iy * g.W + ix) * g.lda + ci];
                        }
                    }
                    tmpa[i][j] = v;
                }
#pragma unroll
                for (int i = 0; i < RB; ++i) tmpb[i][j] = (kin && bval[i]) ? Bp[bbase[i] + wofs] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < RA; ++i) ra[i] = make_float4(tmpa[i][0], tmpa[i][1], tmpa[i][2], tmpa[i][3]);
#pragma unroll
            for (int i = 0; i < RB; ++i) rb[i] = make_float4(tmpb[i][0], tmpb[i][1], tmpb[i][2], tmpb[i][3]);
        }
    };
    auto lstore = [&](int buf, const float4 (&ra)[RA], const float4 (&rb)[RB]) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            float* p = &As[buf][(lrow + i * RPP) * LD + kq];
            if constexpr (IG_B128) *reinterpret_cast<float4*>(p) = ra[i];
            else { p[0] = ra[i].x; p[1] = ra[i].y; p[2] = ra[i].z; p[3] = ra[i].w; }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (FAST && KN) {
                float* p = &Bs[buf][nq4 * LD + kb[i]];
                p[0] = rb[i].x; p[LD] = rb[i].y; p[2 * LD] = rb[i].z; p[3 * LD] = rb[i].w;
            } else {
                float* p = &Bs[buf][(lrow + i * RPP) * LD + kq];
                if constexpr (IG_B128) *reinterpret_cast<float4*>(p) = rb[i];
                else { p[0] = rb[i].x; p[1] = rb[i].y; p[2] = rb[i].z; p[3] = rb[i].w; }
            }
        }
    };

    const int arow = (wm * TM * 32 + (lane & 31)) * LD + (lane >> 5) * (IG_B128 ? 4 : 1);
    const int brow = (wn * TN * 32 + (lane & 31)) * LD + (lane >> 5) * (IG_B128 ? 4 : 1);
    auto compute = [&](int cur) {
        const float* as = As[cur];
        const float* bs = Bs[cur];
        if constexpr (IG_B128) {
#pragma unroll
            for (int g8 = 0; g8 < BK / 8; ++g8) {
                float4 a4[TM], b4[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a4[i] = *reinterpret_cast<const float4*>(as + arow + i * 32 * LD + g8 * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) b4[j] = *reinterpret_cast<const float4*>(bs + brow + j * 32 * LD + g8 * 8);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float av = t == 0 ? a4[i].x : t == 1 ? a4[i].y : t == 2 ? a4[i].z : a4[i].w;
                            const float bv = t == 0 ? b4[j].x : t == 1 ? b4[j].y : t == 2 ? b4[j].z : b4[j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                        }
            }
            return;
        }
        // fragments of k-pair kk + 1 are read while the MFMAs of pair kk run (two register sets, FRAG_PIPE)
        float a[2][TM], b[2][TN];
        auto frag = [&](int set, int kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[set][i] = as[arow + i * 32 * LD + kk * 2];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[set][j] = bs[brow + j * 32 * LD + kk * 2];
        };
        if (FRAG_PIPE) frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int set = FRAG_PIPE ? (kk & 1) : 0;
            if (FRAG_PIPE) { if (kk + 1 < BK / 2) frag(set ^ 1, kk + 1); }
            else frag(0, kk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][i], b[set][j], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (FAST && BUF && IG_DEEP && BM * BN <= 64 * 64) {      // (wider tiles: the compiler doubles the accumulator registers across the two phases)
        // two K-steps in flight: while step kt is computed from LDS, step kt + 1 sits in one register set and the loads of step
        // kt + 2 have just been issued into the other -- short reductions (1x1 layers, K = 64 ... 256) expose one memory latency
        // per tile instead of one per step.  All loads are issued unconditionally (a step past the end reads zeros through the
        // descriptor's range check) so that the waits before an LDS store cover the older register set only.
        gload(kt0, ra, rb);
        gload(kt0 + 1, ra1, rb1);
        lstore(0, ra, rb);
        __syncthreads();
        for (int kt = kt0; kt < kt1; kt += 2) {
            gload(kt + 2, ra, rb);
            compute(0);
            if (kt + 1 >= kt1) break;
            lstore(1, ra1, rb1);
            __syncthreads();
            gload(kt + 3, ra1, rb1);
            compute(1);
            if (kt + 2 < kt1) lstore(0, ra, rb);
            __syncthreads();
        }
        if (g.stat != nullptr) __syncthreads();          // (the statistics epilogue re-uses As)
    } else {
        gload(kt0, ra, rb);
        lstore(0, ra, rb);
        __syncthreads();
        int cur = 0;
        for (int kt = kt0; kt < kt1; ++kt) {
            if (kt + 1 < kt1) gload(kt + 1, ra, rb);
            compute(cur);
            if (kt + 1 < kt1) lstore(cur ^ 1, ra, rb);
            __syncthreads();
            cur ^= 1;
        }
    }

    if (g.ksteps > 0) {                                  // split-K: raw partial tile -> part[split][M][N]
        float* pp = g.part + (long)blockIdx.z * g.M * g.N;
        if (m0 + BM <= g.M && (long)g.M * g.N < (1L << 29)) {      // whole rows: buffer stores, columns past N dropped by the range check
            const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void*)pp, 0, (unsigned)g.M * (unsigned)g.N * 4u, 0x00020000);
            const unsigned ldn4 = (unsigned)g.N * 4u;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * TN * 32 + j * 32 + (lane & 31);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const unsigned vo = col < g.N ? (unsigned)((m0 + wm * TM * 32 + i * 32 + 4 * (lane >> 5)) * g.N + col) * 4u : 0xffffffffu;
                    float v[16];                            // (through a VGPR copy: storing acc[i][j][r] directly wrote row 0's values to every row -- hipcc 7.2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), rsP, vo, ((r & 3) + 8 * (r >> 2)) * ldn4, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * TN * 32 + j * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (col < g.N && row < g.M) pp[(long)row * g.N + col] = acc[i][j][r];
                }
        }
        return;
    }
    // epilogue: lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    StatAcc st[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) st[j] = StatAcc{0.f, 0.f, 0.f, 0.f};
    const bool do_stat = g.stat != nullptr;
    if constexpr (FAST && BUF) {
        if (g.cbytes != 0 && m0 + BM <= g.M) {             // whole tile of a dense row-major output: the buffer-store epilogue
            lean_epilogue<TM, TN, WM, WN, BN>(acc, g, Cp, biasp, m0, n0, tmi, wm, wn, lane, tid, &As[0][0]);
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + (lane & 31);
        const bool cok = col < g.N;
        int co = col, padd_y = 0, padd_x = 0;
        if (g.ps_cout > 0) {
            int tap = col / g.ps_cout;
            co = col - tap * g.ps_cout;
            padd_y = tap / g.ps_kw;
            padd_x = tap - padd_y * g.ps_kw;
        }
        const float bv = (biasp != nullptr && cok) ? biasp[co] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (cok && row < g.M) {
                    float v = acc[i][j][r] + bv;
                    if (g.act == 1) v = fmaxf(v, 0.f);
                    else if (g.act == 2) v = v > 0.f ? v : 0.1f * v;
                    long o;
                    if (g.plain_out) o = (long)row * g.ldc + co;
                    else {
                        int hw = g.QH * g.QW;
                        int ni = row / hw, rem = row - ni * hw;
                        int qy = rem / g.QW, qx = rem - qy * g.QW;
                        int oy = qy * g.osy + g.ooy + padd_y, ox = qx * g.osx + g.oox + padd_x;
                        o = (((long)ni * g.OH + oy) * g.OW + ox) * g.ldc + co;
                    }
                    if (g.accum) v += Cp[o];
                    Cp[o] = v;
                    if (do_stat) stat_add(st[j], v);
                }
            }
        }
    }
    if (do_stat) stat_finish<TN, WM, WN, BN>(st, &As[0][0], g.stat, tmi, n0, g.N, wm, wn, lane, tid);     // (As: the loop's last barrier is behind us)
}

template <int BM, int BN, int WM, int WN, bool FAST, bool KN, int BKT = 16, bool BUF = false>
__global__ __launch_bounds__(WM * WN * 64, (BM * BN >= 128 * 128 ? 3 : 1)) void igemm_nt(const IGemm g) {
    igemm_nt_body<BM, BN, WM, WN, FAST, KN, BKT, BUF, false>(g);
}
// the same with IGemm::a_scale / a_shift applied to the A tile (buffer-load form, [N][K] weights)
template <int BM, int BN, int WM, int WN, int BKT>
__global__ __launch_bounds__(WM * WN * 64, (BM * BN >= 128 * 128 ? 3 : 1)) void igemm_nt_aff(const IGemm g) {
    igemm_nt_body<BM, BN, WM, WN, true, false, BKT, true, true>(g);
}

// ---------------------------------------------------------------------------------------------
// igemm_halo3x3: the 128x128 implicit GEMM for 3x3 stride-1 convolutions (forward, and backward-data, which is a 3x3
// stride-1 convolution of dy with the taps mirrored) that re-uses its input from LDS.  igemm_nt gathers the A tile from global
// memory once per tap: every input pixel is fetched 9 times, and because the 9 visits of a block are 1.5 MB of streaming
// apart (x 96 resident blocks per XCD) none of them hits L2 -- 5.7 GB of fetch for `feat` against 0.55 GB of input.  Here a
// block's 128 output pixels are R = 128 / W whole image rows; per 16-channel chunk the (R + 2) x (W + 2) halo of input pixels
// (zero-padded at the image border) is staged in LDS ONCE and the 9 taps read it at shifted row offsets, so the input is
// fetched (R + 2)(W + 2) / (R W) = 2.06x (W = 64) instead of 9x, with 4.4x fewer global-load instructions for A.
// K order: channel chunk outer, tap inner (B tile = 128 x 16 weights per (chunk, tap) as before).
// LDS: A halo [2][264][17] + B [2][128][17] floats = 53 KB -> 3 blocks / CU like igemm_nt<128,128>.
#define HALO_MAX_PIX 264
template <bool KN, bool BUF = false>
__global__ __launch_bounds__(256, 3) void igemm_halo3x3(const IGemm g) {
    constexpr int BM = 128, BN = 128, WN = 2, BK = 16, LD = 17, TM = 2, TN = 2;
    constexpr int NH = (HALO_MAX_PIX * 4 + 255) / 256;                    // float4 halo loads per thread and chunk (5)
    __shared__ float As[2][HALO_MAX_PIX * LD];
    __shared__ float Bs[2][BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const float* __restrict__ Ap = g.A; const float* __restrict__ Bp = g.B; const float* __restrict__ biasp = g.bias;
    float* __restrict__ Cp = g.C;
    const int ntm = g.M / BM, ntn = (g.N + BN - 1) / BN;
    int tmi, tni;
    xcd_tile(blockIdx.x, ntm * ntn, ntn, tmi, tni, g.gm);
    const int m0 = tmi * BM, n0 = tni * BN;
    const int W = g.W, H = g.H, HC = W + 2, R = BM / W, HP = (R + 2) * HC;
    const int img = m0 / (H * W), y0 = (m0 - img * H * W) / W;
    // halo staging plan of this thread: float4 q = tid + 256 i -> (halo pixel q / 4, channel quad q % 4)
    long hsrc[NH]; int hdst[NH]; bool hok[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int q = tid + i * 256, hp = q >> 2, part = q & 3;
        const int hy = hp / HC, hx = hp - hy * HC;
        const int iy = y0 - 1 + hy, ix = hx - 1;
        hok[i] = hp < HP && iy >= 0 && iy < H && ix >= 0 && ix < W;
        hsrc[i] = (((long)img * H + iy) * W + ix) * g.lda + part * 4;
        hdst[i] = hp < HP ? hp * LD + part * 4 : -1;
    }
    // B staging (as igemm_nt FAST): [N][K]: thread -> (row n, 4 consecutive k);  [K][N]: thread -> (k row, 4 consecutive n)
    const int lrow = tid >> 2, kq = (tid & 3) * 4;
    const int nq4 = (tid & 31) * 4;
    long bbase[2]; bool bval[2]; int kb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (KN) { kb[i] = (tid + i * 256) >> 5; bval[i] = n0 + nq4 < g.N; bbase[i] = (long)kb[i] * g.ldb + n0 + nq4; }
        else { const int n = n0 + lrow + i * 64; kb[i] = 0; bval[i] = n < g.N; bbase[i] = (long)n * g.ldb; }
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[NH], rb[2];
    // BUF: see igemm_nt -- 32-bit byte offsets through buffer descriptors, a masked element is the out-of-range offset 0xffffffff
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, BUF ? g.abytes : 0, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, BUF ? g.bbytes : 0, 0x00020000);
    unsigned hoffB[NH], boffB[2];
    if constexpr (BUF) {
#pragma unroll
        for (int i = 0; i < NH; ++i) hoffB[i] = (unsigned)hsrc[i] * 4u;
#pragma unroll
        for (int i = 0; i < 2; ++i) boffB[i] = (unsigned)(bbase[i] + (KN ? 0 : kq)) * 4u;
    }
    auto hload = [&](int ci0) {
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            if constexpr (BUF) ra[i] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsA, hok[i] ? hoffB[i] + (unsigned)ci0 * 4u : 0xffffffffu, 0, 0));
            else ra[i] = *reinterpret_cast<const float4*>(hok[i] ? Ap + hsrc[i] + ci0 : g_zero16);
        }
    };
    auto hstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NH; ++i)
            if (hdst[i] >= 0) { float* p = &As[buf][hdst[i]]; p[0] = ra[i].x; p[1] = ra[i].y; p[2] = ra[i].z; p[3] = ra[i].w; }
    };
    auto bload = [&](int ci0, int tap) {
        const int wb = g.wt[tap] * (KN ? g.btap : g.Cin);
        if constexpr (BUF) {
            const unsigned sb = KN ? (unsigned)(ci0 * g.ldb + wb) * 4u : (unsigned)(wb + ci0) * 4u;
#pragma unroll
            for (int i = 0; i < 2; ++i) rb[i] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsB, bval[i] ? boffB[i] + sb : 0xffffffffu, 0, 0));
        } else {
            const long wofs = KN ? (long)ci0 * g.ldb + wb : (long)(wb + ci0 + kq);
#pragma unroll
            for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const float4*>(bval[i] ? Bp + bbase[i] + wofs : g_zero16);
        }
    };
    auto bstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (KN) { float* p = &Bs[buf][nq4 * LD + kb[i]]; p[0] = rb[i].x; p[LD] = rb[i].y; p[2 * LD] = rb[i].z; p[3 * LD] = rb[i].w; }
            else { float* p = &Bs[buf][(lrow + i * 64) * LD + kq]; p[0] = rb[i].x; p[1] = rb[i].y; p[2] = rb[i].z; p[3] = rb[i].w; }
        }
    };
    // A fragment rows: output pixel p = (wm*2 + i)*32 + (lane & 31) -> image row r = p / W, column c = p % W -> halo pixel (r+1, c+1)
    int arow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = (wm * TM + i) * 32 + (lane & 31), r = p / W, c = p - r * W;
        arow[i] = ((r + 1) * HC + c + 1) * LD + (lane >> 5);
    }
    const int brow = (wn * TN * 32 + (lane & 31)) * LD + (lane >> 5);

    const int nchunks = g.Cin / BK, nsteps = nchunks * 9;
    hload(0); hstore(0);
    bload(0, 0); bstore(0);
    __syncthreads();
    int abuf = 0, bbuf = 0, chunk = 0, tap = 0;
    for (int st = 0; st < nsteps; ++st) {
        const bool more = st + 1 < nsteps;
        const int ntap = tap == 8 ? 0 : tap + 1, nchunk = tap == 8 ? chunk + 1 : chunk;
        if (more) bload(nchunk * BK, ntap);
        const bool stage_a = tap == 0 && chunk + 1 < nchunks;             // next chunk's halo: loaded during tap 0, written after its MFMAs
        if (stage_a) hload((chunk + 1) * BK);
        const int toff = (g.dy[tap] * HC + g.dx[tap]) * LD;
        const float* as = As[abuf] + toff;
        const float* bs = Bs[bbuf];
        float a[2][TM], b[2][TN];
        auto frag = [&](int set, int kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[set][i] = as[arow[i] + kk * 2];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[set][j] = bs[brow + j * 32 * LD + kk * 2];
        };
        if (FRAG_PIPE) frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int set = FRAG_PIPE ? (kk & 1) : 0;
            if (FRAG_PIPE) { if (kk + 1 < BK / 2) frag(set ^ 1, kk + 1); }
            else frag(0, kk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][i], b[set][j], acc[i][j], 0, 0, 0);
        }
        if (more) bstore(bbuf ^ 1);
        if (stage_a) hstore(abuf ^ 1);
        __syncthreads();
        bbuf ^= 1;
        if (tap == 8) abuf ^= 1;
        tap = ntap; chunk = nchunk;
    }
    if constexpr (BUF) {
        if (g.cbytes != 0) {                                 // (M % 128 == 0: every tile is whole)
            lean_epilogue<TM, TN, 2, WN, BN>(acc, g, Cp, biasp, m0, n0, tmi, wm, wn, lane, tid, &As[0][0]);
            return;
        }
    }
    StatAcc st[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) st[j] = StatAcc{0.f, 0.f, 0.f, 0.f};
    const bool do_stat = g.stat != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + (lane & 31);
        const bool cok = col < g.N;
        const float bv = (biasp != nullptr && cok) ? biasp[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (cok) {
                    float v = acc[i][j][r] + bv;
                    if (g.act == 1) v = fmaxf(v, 0.f);
                    else if (g.act == 2) v = v > 0.f ? v : 0.1f * v;
                    if (g.accum) v += Cp[(long)row * g.ldc + col];
                    Cp[(long)row * g.ldc + col] = v;
                    if (do_stat) stat_add(st[j], v);
                }
            }
    }
    if (do_stat) stat_finish<TN, 2, WN, BN>(st, &As[0][0], g.stat, tmi, n0, g.N, wm, wn, lane, tid);
}

// ---------------------------------------------------------------------------------------------
// BUF (FAST only): operands through buffer descriptors -- a masked element (row past the split, padded tap, column past the
// matrix) is an out-of-range 32-bit offset that the hardware returns as zeros; `if (ok) v = load` compiles to an exec-mask branch
// with an s_waitcnt vmcnt(0) per load, which serialises the tile's loads (see wgemm_tn_dma).
template <int BI, int BJ, int WM, int WN, bool FAST, int BKT, bool BUF, bool AFF>
__device__ __forceinline__ void wgemm_tn_body(const WGemm& g) {
    constexpr int BK = BKT;
    constexpr int TM = BI / WM / 32, TN = BJ / WN / 32;
    constexpr int TPR_P = BI / 4, TPR_Q = BJ / 4;        // threads per LDS row
    constexpr int RP = BK / (256 / TPR_P), RQ = BK / (256 / TPR_Q);
    __shared__ __attribute__((aligned(16))) float Ps[2][BK * BI];
    __shared__ __attribute__((aligned(16))) float Qs[2][BK * BJ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ Pp = g.P; const float* __restrict__ Qp = g.Q; float* slabp = g.slab; float* bslabp = g.bslab;
    if (blockIdx.z) { Pp += g.gsP; Qp += g.gsQ; slabp = g.slab1; bslabp = g.bslab1; }
    const int wm = wave / WN, wn = wave % WN;
    const int NJ = g.T * g.Cq;
    const int nti = (g.NI + BI - 1) / BI, ntj = (NJ + BJ - 1) / BJ;
    int ti, tj;
    if (g.tap_major) {
        // Each XCD gets a contiguous run of the tile list ordered (channel block, tap, i-tile): the j-tiles that gather the
        // SAME channels of the image through different taps (9 shifted views of the same pixels for a 3x3 conv) run on one
        // XCD, so the image is fetched into that L2 once instead of once per tap-owning XCD.
        const int nblk = nti * ntj, q = nblk >> 3, r = nblk & 7, x = blockIdx.x & 7;
        const int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
        const int nb = g.Cq / BJ;                         // channel blocks per tap (Cq % BJ == 0 guaranteed by the host)
        const int per_cb = g.T * nti;
        const int cb = lin / per_cb, rem = lin - cb * per_cb;
        const int tap = rem / nti;
        ti = rem - tap * nti;
        tj = tap * nb + cb;
    } else {
        xcd_tile(blockIdx.x, nti * ntj, ntj, ti, tj);
    }
    const int i0 = ti * BI, j0 = tj * BJ;
    const bool do_bias = bslabp != nullptr && tj == 0 && tid < BI;
    float bsum = 0.f;
    const int ms = blockIdx.y * g.rows_per_split;
    const int me = min(g.M, ms + g.rows_per_split);

    const int pr = tid / TPR_P, pc = (tid % TPR_P) * 4;
    const int qr = tid / TPR_Q, qc = (tid % TPR_Q) * 4;
    // this thread's Q columns j0+qc..+3 -> (tap, channel); fixed for the whole reduction
    int qt[4], qch[4]; bool qok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int j = j0 + qc + e;
        qok[e] = j < NJ;
        int t = qok[e] ? j / g.Cq : 0;
        qt[e] = t; qch[e] = j - t * g.Cq;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 rp[RP], rq[RQ];
    // (image, qy, qx) of each Q row this thread loads, advanced by BK rows per step -- no divisions in the loop
    int q_ni[RQ], q_y[RQ], q_x[RQ];
#pragma unroll
    for (int i = 0; i < RQ; ++i) {
        int m = ms + qr + i * (256 / TPR_Q);
        int hw = g.QH * g.QW;
        q_ni[i] = m / hw;
        int rem = m - q_ni[i] * hw;
        q_y[i] = rem / g.QW;
        q_x[i] = rem - q_y[i] * g.QW;
    }
    const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void*)Pp, 0, BUF ? g.pbytes : 0, 0x00020000);
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qp, 0, BUF ? g.qbytes : 0, 0x00020000);
    float4 qsc = make_float4(1.f, 1.f, 1.f, 1.f), qsh = make_float4(0.f, 0.f, 0.f, 0.f);      // WGemm::q_scale / q_shift of this thread's 4 columns
    if (AFF && qok[0]) { qsc = *reinterpret_cast<const float4*>(g.q_scale + qch[0]); qsh = *reinterpret_cast<const float4*>(g.q_shift + qch[0]); }
    auto gload = [&](int mb) {
        if constexpr (BUF) {
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                const int m = mb + pr + i * (256 / TPR_P);
                const unsigned op = (m < me && i0 + pc < g.NI) ? (unsigned)(m * g.ldp + i0 + pc) * 4u : 0xffffffffu;
                const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rsP, op, 0, 0);
                rp[i] = *reinterpret_cast<const float4*>(&v);
            }
#pragma unroll
            for (int i = 0; i < RQ; ++i) {
                const int m = mb + qr + i * (256 / TPR_Q);
                unsigned oq;
                if (g.plain_q) oq = (m < me && qok[0]) ? (unsigned)(m * g.ldq + qch[0]) * 4u : 0xffffffffu;
                else {
                    const int iy = q_y[i] * g.sy + g.dy[qt[0]], ix = q_x[i] * g.sx + g.dx[qt[0]];
                    const bool ok = m < me && qok[0] && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                    oq = ok ? (unsigned)(((q_ni[i] * g.H + iy) * g.W + ix) * g.ldq + qch[0]) * 4u : 0xffffffffu;
                }
                const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rsQ, oq, 0, 0);
                rq[i] = *reinterpret_cast<const float4*>(&v);
                if constexpr (AFF) {                         // BatchNorm + ReLU of x applied here; rows past the split's end must stay zero
                    if (mb + BK <= me && qok[0]) {           // (whole K-step: no per-row select)
                        rq[i].x = fmaxf(fmaf(rq[i].x, qsc.x, qsh.x), 0.f); rq[i].y = fmaxf(fmaf(rq[i].y, qsc.y, qsh.y), 0.f);
                        rq[i].z = fmaxf(fmaf(rq[i].z, qsc.z, qsh.z), 0.f); rq[i].w = fmaxf(fmaf(rq[i].w, qsc.w, qsh.w), 0.f);
                    } else {
                        const bool rok = oq != 0xffffffffu;
                        rq[i].x = rok ? fmaxf(fmaf(rq[i].x, qsc.x, qsh.x), 0.f) : 0.f; rq[i].y = rok ? fmaxf(fmaf(rq[i].y, qsc.y, qsh.y), 0.f) : 0.f;
                        rq[i].z = rok ? fmaxf(fmaf(rq[i].z, qsc.z, qsh.z), 0.f) : 0.f; rq[i].w = rok ? fmaxf(fmaf(rq[i].w, qsc.w, qsh.w), 0.f) : 0.f;
                    }
                }
                q_x[i] += BK;
                while (q_x[i] >= g.QW) {
                    q_x[i] -= g.QW;
                    if (++q_y[i] == g.QH) { q_y[i] = 0; ++q_ni[i]; }
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            int m = mb + pr + i * (256 / TPR_P);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < me) {
                const float* p = Pp + (long)m * g.ldp + i0 + pc;
                if (FAST) { if (i0 + pc < g.NI) v = *reinterpret_cast<const float4*>(p); }
                else {
                    if (i0 + pc + 0 < g.NI) v.x = p[0];
                    if (i0 + pc + 1 < g.NI) v.y = p[1];
                    if (i0 + pc + 2 < g.NI) v.z = p[2];
                    if (i0 + pc + 3 < g.NI) v.w = p[3];
                }
            }
            rp[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RQ; ++i) {
            int m = mb + qr + i * (256 / TPR_Q);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < me) {
                if (g.plain_q) {
                    const float* p = Qp + (long)m * g.ldq;
                    if (FAST) { if (qok[0]) v = *reinterpret_cast<const float4*>(p + qch[0]); }
                    else {
                        if (qok[0]) v.x = p[qch[0]];
                        if (qok[1]) v.y = p[qch[1]];
                        if (qok[2]) v.z = p[qch[2]];
                        if (qok[3]) v.w = p[qch[3]];
                    }
                } else {
                    const int qy = q_y[i], qx = q_x[i];
                    const float* img = Qp + (long)q_ni[i] * g.H * g.W * g.ldq;
                    if (FAST) {
                        int iy = qy * g.sy + g.dy[qt[0]], ix = qx * g.sx + g.dx[qt[0]];
                        if (qok[0] && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                            v = *reinterpret_cast<const float4*>(img + ((long)iy * g.W + ix) * g.ldq + qch[0]);
                    } else {
                        float e4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            int iy = qy * g.sy + g.dy[qt[e]], ix = qx * g.sx + g.dx[qt[e]];
                            e4[e] = (qok[e] && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                                        ? img[((long)iy * g.W + ix) * g.ldq + qch[e]] : 0.f;
                        }
                        v = make_float4(e4[0], e4[1], e4[2], e4[3]);
                    }
                }
            }
            rq[i] = v;
            q_x[i] += BK;
            while (q_x[i] >= g.QW) {
                q_x[i] -= g.QW;
                if (++q_y[i] == g.QH) { q_y[i] = 0; ++q_ni[i]; }
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RP; ++i)
            *reinterpret_cast<float4*>(&Ps[buf][(pr + i * (256 / TPR_P)) * BI + pc]) = rp[i];
#pragma unroll
        for (int i = 0; i < RQ; ++i)
            *reinterpret_cast<float4*>(&Qs[buf][(qr + i * (256 / TPR_Q)) * BJ + qc]) = rq[i];
    };

    if (ms < me) {
        gload(ms);
        lstore(0);
    }
    __syncthreads();
    int cur = 0;
    const int aoff = (lane >> 5) * BI + wm * TM * 32 + (lane & 31);
    const int boff = (lane >> 5) * BJ + wn * TN * 32 + (lane & 31);
    for (int mb = ms; mb < me; mb += BK) {
        const bool more = mb + BK < me;
        if (more) gload(mb + BK);
        const float* ps = Ps[cur];
        const float* qs = Qs[cur];
        if (do_bias) {
#pragma unroll
            for (int kb = 0; kb < BK; ++kb) bsum += ps[kb * BI + tid];
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = ps[aoff + kk * 2 * BI + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = qs[boff + kk * 2 * BJ + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // (the ticket flag lives in the staging array: a second __shared__ object can de-pipeline LDS-DMA kernels, guide section 5 item 4a)
    wgemm_finish<TM, TN>(g, acc, i0, j0, wm, wn, lane, do_bias && i0 + tid < g.NI, bsum, ti * ntj + tj, nti * ntj, reinterpret_cast<int*>(&Ps[0][0]), i0 + BI <= g.NI);
}


template <int BI, int BJ, int WM, int WN, bool FAST, int BKT = 16, bool BUF = false>
__global__ __launch_bounds__(256) void wgemm_tn(const WGemm g) {
    wgemm_tn_body<BI, BJ, WM, WN, FAST, BKT, BUF, false>(g);
}
__global__ __launch_bounds__(256) void wgemm_tn64_aff(const WGemm g) {      // WGemm::q_scale / q_shift applied to the Q rows
    wgemm_tn_body<64, 64, 2, 2, true, 32, true, true>(g);
}

// ---------------------------------------------------------------------------------------------
// wgemm_tn_dma: the 128x128 weight-gradient tile with LDS-DMA staging (global_load_lds, 16 B per lane).
// The register-staged wgemm_tn is bound by L2->CU latency: every K-step reads fresh rows (no L1 reuse) and only one
// tile is in flight per block.  Here the operand tiles go global -> LDS directly through a 3-stage ring, issued TWO
// K-steps ahead with counted waits (s_waitcnt vmcnt(4)) and one raw s_barrier per step (guide: T3+T4).  The [k][128]
// tile image is lane-linear (a wave's 64 x 16 B = two consecutive 128-float rows), exactly what LDS-DMA writes; rows past
// the split's end and padded / out-of-image taps read a 16-byte zero word instead (the source address is per lane).

#define GLDS16(src, dst) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                                          (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// BUF: both operands are addressed through buffer descriptors (buffer_load_dwordx4 ... offen lds): a 32-bit byte offset per lane,
// and a masked lane (row past the split, padded tap, column past the matrix) takes the offset 0xffffffff, which the hardware's
// range check turns into zeros -- no select between two 64-bit addresses.  The flat form computes each lane's address with
// 64-bit multiply-adds under exec-mask branches and re-loads the zero word's address from the GOT four times per K-step, each
// behind an s_waitcnt lgkmcnt(0) that also drains the wave's LDS reads (see the .s): ~150 scalar / vector instructions per K-step.
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
template <int ST, bool BUF, bool AFF>
__device__ __forceinline__ void wgemm_tn_dma_body(const WGemm& g) {
    constexpr int BI = 128, BJ = 128, BK = 16, WN = 2, TM = 2, TN = 2;
    __shared__ __attribute__((aligned(16))) float smem[ST * 2 * BK * 128];     // [stage][P|Q][k][128]  (48 KB, ONE array)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ Pp = g.P; const float* __restrict__ Qp = g.Q; float* slabp = g.slab; float* bslabp = g.bslab;
    const int wm = wave / WN, wn = wave % WN;
    const int NJ = g.T * g.Cq;
    const int nti = (g.NI + BI - 1) / BI, ntj = (NJ + BJ - 1) / BJ;
    int split = blockIdx.y, plane = blockIdx.z, lin_tile = -1;
    if (g.batch > 0 && g.bsplits > 0) {                     // linear grid: see WGemm::bsplits
        const int tiles = nti * ntj, slot = blockIdx.x >> 3;
        const int grp = (slot / tiles) * 8 + (blockIdx.x & 7);
        if (grp >= g.bsplits * g.batch) return;             // (grid padded to whole rounds of 8 groups)
        lin_tile = slot - (slot / tiles) * tiles;
        plane = grp / g.bsplits;
        split = grp - plane * g.bsplits;
    }
    if (g.batch > 0) { Pp += (long)plane * g.gsP; Qp += (long)plane * g.gsQ; bslabp = nullptr; }
    else if (blockIdx.z) { Pp += g.gsP; Qp += g.gsQ; slabp = g.slab1; bslabp = g.bslab1; }
    int ti, tj;
    if (lin_tile >= 0) { ti = lin_tile / ntj; tj = lin_tile - ti * ntj; }
    else if (g.tap_major) {
        const int nblk = nti * ntj, q = nblk >> 3, r = nblk & 7, x = blockIdx.x & 7;
        const int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
        const int nb = g.Cq / BJ, per_cb = g.T * nti;
        const int cb = lin / per_cb, rem = lin - cb * per_cb;
        const int tap = rem / nti;
        ti = rem - tap * nti;
        tj = tap * nb + cb;
    } else {
        xcd_tile(blockIdx.x, nti * ntj, ntj, ti, tj);
    }
    const int i0 = ti * BI, j0 = tj * BJ;
    const bool do_bias = bslabp != nullptr && tj == 0 && tid < BI;
    float bsum = 0.f;
    const int ms = split * g.rows_per_split;
    const int me = min(g.M, ms + g.rows_per_split);
    const int nt = (me - ms + BK - 1) / BK;

    const int pr = tid >> 5, pc = (tid & 31) * 4;           // this thread's row (0..7, +8 on the 2nd pass) and 4-float column
    const bool pcol_ok = i0 + pc < g.NI;
    const int jcol = j0 + pc;
    const bool qcol_ok = jcol < NJ;
    const int qtap = qcol_ok ? jcol / g.Cq : 0;
    const int qch = jcol - qtap * g.Cq;
    const int tdy = g.dy[qtap], tdx = g.dx[qtap];
    int q_ni[2], q_y[2], q_x[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = ms + pr + i * 8, hw = g.QH * g.QW;
        q_ni[i] = m / hw;
        const int rem = m - q_ni[i] * hw;
        q_y[i] = rem / g.QW;
        q_x[i] = rem - q_y[i] * g.QW;
    }
    const int wbase = __builtin_amdgcn_readfirstlane(wave) * 2 * 128;      // wave-uniform LDS row base (floats)
    const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void*)Pp, 0, BUF ? g.pbytes : 0, 0x00020000);
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qp, 0, BUF ? g.qbytes : 0, 0x00020000);

    auto issue = [&](int t, int st) {                       // tile t (rows ms + 16t ...) -> ring stage st
        float* sp = smem + (st * 2 + 0) * BK * 128 + wbase;
        float* sq = smem + (st * 2 + 1) * BK * 128 + wbase;
        const int mb = ms + t * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mb + pr + i * 8;
            const bool rowok = m < me;
            if constexpr (BUF) {
                const unsigned op = (rowok && pcol_ok) ? (unsigned)(m * g.ldp + i0 + pc) * 4u : 0xffffffffu;
                unsigned oq;
                if (g.plain_q) oq = (rowok && qcol_ok) ? (unsigned)(m * g.ldq + qch) * 4u : 0xffffffffu;
                else {
                    const int iy = q_y[i] * g.sy + tdy, ix = q_x[i] * g.sx + tdx;
                    const bool ok = rowok && qcol_ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                    oq = ok ? (unsigned)(((q_ni[i] * g.H + iy) * g.W + ix) * g.ldq + qch) * 4u : 0xffffffffu;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, LDS_PTR(sp + i * 8 * 128), 16, op, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, LDS_PTR(sq + i * 8 * 128), 16, oq, 0, 0, 0);
            } else {
                const float* srcp = (rowok && pcol_ok) ? Pp + (long)m * g.ldp + i0 + pc : g_zero16;
                const float* srcq;
                if (g.plain_q) srcq = (rowok && qcol_ok) ? Qp + (long)m * g.ldq + qch : g_zero16;
                else {
                    const int iy = q_y[i] * g.sy + tdy, ix = q_x[i] * g.sx + tdx;
                    const bool ok = rowok && qcol_ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                    srcq = ok ? Qp + ((long)q_ni[i] * g.H * g.W + (long)iy * g.W + ix) * g.ldq + qch : g_zero16;
                }
                GLDS16(srcp, sp + i * 8 * 128);
                GLDS16(srcq, sq + i * 8 * 128);
            }
        }
        if (g.QW * 2 >= BK) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                q_x[i] += BK;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const bool wrap = q_x[i] >= g.QW;
                    q_x[i] -= wrap ? g.QW : 0;
                    q_y[i] += wrap ? 1 : 0;
                    const bool wy = q_y[i] >= g.QH;
                    q_y[i] = wy ? 0 : q_y[i];
                    q_ni[i] += wy ? 1 : 0;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = mb + BK + pr + i * 8, hw = g.QH * g.QW;
                q_ni[i] = m / hw;
                const int rem = m - q_ni[i] * hw;
                q_y[i] = rem / g.QW;
                q_x[i] = rem - q_y[i] * g.QW;
            }
        }
    };

    // ---- uniform form of `issue` (WGemm::uniform; BUF only): positions of rows mb and mb + 8 in scalar registers
    const bool uni = BUF && g.uniform != 0;
    const int utap = __builtin_amdgcn_readfirstlane(j0 < NJ ? j0 / g.Cq : 0);
    const int udy = g.dy[utap], udx = g.dx[utap];
    int s_ni[2], s_y[2], s_x[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = __builtin_amdgcn_readfirstlane(ms + i * 8), hw = g.QH * g.QW;
        s_ni[i] = m / hw;
        const int rem = m - s_ni[i] * hw;
        s_y[i] = rem / g.QW;
        s_x[i] = rem - s_y[i] * g.QW;
    }
    const int cx = pr * g.sx + udx;                                                       // this thread's x = s_x * sx + cx
    const unsigned kP = (unsigned)(pr * g.ldp + i0 + pc) * 4u;                            // per-thread constants of the byte offsets
    const unsigned kQ = g.plain_q ? (unsigned)(pr * g.ldq + qch) * 4u : (unsigned)(cx * g.ldq + (jcol - utap * g.Cq)) * 4u;
    auto issue_uni = [&](int t, int st) {
        float* sp = smem + (st * 2 + 0) * BK * 128 + wbase;
        float* sq = smem + (st * 2 + 1) * BK * 128 + wbase;
        const int mb = ms + t * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m0 = mb + i * 8;                       // scalar
            const bool rowok = m0 + pr < me;
            unsigned sQ; bool yok = true, xok = true;
            if (g.plain_q) sQ = (unsigned)(m0 * g.ldq) * 4u;
            else {
                const int iy = s_y[i] * g.sy + udy, x0 = s_x[i] * g.sx;
                yok = iy >= 0 && iy < g.H;
                sQ = (unsigned)(((s_ni[i] * g.H + iy) * g.W + x0) * g.ldq) * 4u;
                xok = x0 + cx >= 0 && x0 + cx < g.W;
                s_x[i] += BK;                                // advance to the next K-step (uniform: scalar unit)
                while (s_x[i] >= g.QW) {
                    s_x[i] -= g.QW;
                    if (++s_y[i] == g.QH) { s_y[i] = 0; ++s_ni[i]; }
                }
            }
            const unsigned op = (rowok && pcol_ok) ? kP + (unsigned)(m0 * g.ldp) * 4u : 0xffffffffu;
            const unsigned oq = (rowok && qcol_ok && yok && xok) ? kQ + sQ : 0xffffffffu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, LDS_PTR(sp + i * 8 * 128), 16, op, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, LDS_PTR(sq + i * 8 * 128), 16, oq, 0, 0, 0);
        }
    };

    // ---- WGemm::uniform == 2: besides the above, a K-step's 16 rows are 16 consecutive pixels of ONE image row (map width and
    // rows per split multiples of 16) or plain rows.  One scalar position per step, advanced incrementally; the byte offset of a
    // load = per-thread constant (VGPR, fixed for the whole kernel) + scalar offset operand; the tap's shift sits in the
    // descriptor's base address, so the constants are non-negative.  Per step and wave: ~10 scalar and ~6 vector instructions
    // beside the 4 DMA loads (the general forms: 60-90 -- measured on `feat`: 5.55 ms with the loads' issue code, 4.64 without).
    const bool u16 = BUF && g.uniform == 2;
    const long qshift = g.plain_q ? 0 : ((long)udy * g.W + udx) * g.ldq;
    const auto rsQs = __builtin_amdgcn_make_buffer_rsrc((void*)(Qp + qshift), 0, BUF ? 0xfffffff0u : 0, 0x00020000);
    unsigned vP[2], vQ[2]; int cxl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        vP[i] = pcol_ok ? (unsigned)((pr + 8 * i) * g.ldp + i0 + pc) * 4u : 0xffffffffu;
        vQ[i] = !qcol_ok ? 0xffffffffu : g.plain_q ? (unsigned)((pr + 8 * i) * g.ldq + qch) * 4u : (unsigned)((pr + 8 * i) * g.sx * g.ldq + (jcol - utap * g.Cq)) * 4u;
        cxl[i] = (pr + 8 * i) * g.sx + udx;                 // this lane's image x = u_x0 * sx + cxl
    }
    int u_x0 = s_x[0], u_y = s_y[0];                        // (row ms: x is a multiple of 16)
    unsigned u_sp = (unsigned)__builtin_amdgcn_readfirstlane(ms) * (unsigned)g.ldp * 4u;
    unsigned u_sq = g.plain_q ? (unsigned)__builtin_amdgcn_readfirstlane(ms) * (unsigned)g.ldq * 4u
                              : (unsigned)(((s_ni[0] * g.H + s_y[0] * g.sy) * g.W + s_x[0] * g.sx) * g.ldq) * 4u;
    const unsigned stepP = 16u * (unsigned)g.ldp * 4u, stepQ = (g.plain_q ? 16u : 16u * (unsigned)g.sx) * (unsigned)g.ldq * 4u;
    const unsigned wrapX = (unsigned)((g.sy * g.W - g.QW * g.sx) * g.ldq) * 4u, wrapY = (unsigned)((g.H - g.QH * g.sy) * g.W * g.ldq) * 4u;
    auto issue_u16 = [&](int t, int st) {
        float* sp = smem + (st * 2 + 0) * BK * 128 + wbase;
        float* sq = smem + (st * 2 + 1) * BK * 128 + wbase;
        const bool live = t < nt;                           // (past the end: zeros, no traffic)
        const int iy = u_y * g.sy + udy;
        const bool rowq = live && (g.plain_q || (iy >= 0 && iy < g.H));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ix = u_x0 * g.sx + cxl[i];
            const bool xok = g.plain_q || (ix >= 0 && ix < g.W);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, LDS_PTR(sp + i * 8 * 128), 16, live ? vP[i] : 0xffffffffu, u_sp, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQs, LDS_PTR(sq + i * 8 * 128), 16, (rowq && xok) ? vQ[i] : 0xffffffffu, u_sq, 0, 0);
        }
        u_sp += stepP; u_sq += stepQ;
        if (!g.plain_q) {
            u_x0 += 16;
            if (u_x0 >= g.QW) {
                u_x0 = 0; u_sq += wrapX;
                if (++u_y == g.QH) { u_y = 0; u_sq += wrapY; }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int aoff = (lane >> 5) * 128 + wm * TM * 32 + (lane & 31);
    const int boff = (lane >> 5) * 128 + wn * TN * 32 + (lane & 31);
    constexpr bool qaff = BUF && AFF;                       // (host: only with uniform == 2 and plain rows -- no zero-filled partial K-step)
    float fsc[TN], fsh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = j0 + wn * TN * 32 + j * 32 + (lane & 31);
        fsc[j] = (qaff && c < NJ) ? g.q_scale[c] : 1.f;
        fsh[j] = (qaff && c < NJ) ? g.q_shift[c] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < ST - 1; ++p) { if (u16) issue_u16(p, p); else if (uni) issue_uni(p, p); else issue(p, p); }
    int st = 0, stn = ST - 1;
    for (int t = 0; t < nt; ++t) {
        // this wave's 4 DMAs of tile t have landed (the ST-2 younger tiles may still fly) ...
        if (ST == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // ... and everybody's; everybody also finished reading tile t-1
        // refill the stage tile t-1 used (rows past the end read zeros).  The scalar-offset form is issued from INSIDE the MFMA
        // sequence (after the second k-pair): its ~25 scalar / vector instructions and 4 DMA loads then overlap the matrix pipe
        // instead of standing between the barrier and the first MFMA
        if (!u16) { if (uni) issue_uni(t + ST - 1, stn); else issue(t + ST - 1, stn); }
        const float* ps = smem + (st * 2 + 0) * BK * 128;
        const float* qs = smem + (st * 2 + 1) * BK * 128;
        if (do_bias) {
#pragma unroll
            for (int kb = 0; kb < BK; ++kb) bsum += ps[kb * 128 + tid];
        }
        float a[2][TM], b[2][TN];
        auto frag = [&](int set, int kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[set][i] = ps[aoff + kk * 2 * 128 + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[set][j] = qs[boff + kk * 2 * 128 + j * 32];
            if constexpr (qaff) {                            // WGemm::q_scale: BatchNorm + ReLU of x at fragment-read time (every K-step is whole here)
#pragma unroll
                for (int j = 0; j < TN; ++j) b[set][j] = fmaxf(fmaf(b[set][j], fsc[j], fsh[j]), 0.f);
            }
        };
        if (FRAG_PIPE) frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int set = FRAG_PIPE ? (kk & 1) : 0;
            if (FRAG_PIPE) { if (kk + 1 < BK / 2) frag(set ^ 1, kk + 1); }
            else frag(0, kk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][i], b[set][j], acc[i][j], 0, 0, 0);
            if (kk == PDF_WG_ISSUE_AT && u16) issue_u16(t + ST - 1, stn);
        }
        st = st == ST - 1 ? 0 : st + 1;
        stn = stn == ST - 1 ? 0 : stn + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may outlive the workgroup

    wgemm_finish<TM, TN>(g, acc, i0, j0, wm, wn, lane, do_bias && i0 + tid < g.NI, bsum, ti * ntj + tj, nti * ntj, reinterpret_cast<int*>(smem), i0 + 128 <= g.NI,
                         lin_tile >= 0 ? split : -1, plane);
}

template <int ST, bool BUF>
__global__ __launch_bounds__(256) void wgemm_tn_dma(const WGemm g) {
    wgemm_tn_dma_body<ST, BUF, false>(g);
}
__global__ __launch_bounds__(256) void wgemm_tn_dma_aff(const WGemm g) {    // WGemm::q_scale / q_shift applied at fragment-read time
    wgemm_tn_dma_body<3, true, true>(g);
}

// Slab reducers.  blockIdx.y == 1 is group 1 of a paired launch (slabs after group 0's, own outputs).  A launch may carry a
// second, short segment -- the bias-gradient partials [splits][nb] behind the weight slabs -- handled by the trailing blocks.
struct Reduce {
    const float* slab; float* out; float* out1; long n;          // weight gradient: [groups][splits][n] -> out / out1
    const float* bslab; float* bout; float* bout1; int nb;        // bias gradient:   [groups][splits][nb] -> bout / bout1 (nb = 0: none)
    int splits, accumulate;
};
__global__ void reduce_slabs(const Reduce r, int main_blocks) {
    const bool bias = (int)blockIdx.x >= main_blocks;
    const float* slab = bias ? r.bslab : r.slab;
    float* out = bias ? (blockIdx.y ? r.bout1 : r.bout) : (blockIdx.y ? r.out1 : r.out);
    const long n = bias ? r.nb : r.n;
    if (blockIdx.y) slab += (long)r.splits * n;
    const long b0 = bias ? blockIdx.x - main_blocks : blockIdx.x, nblk = bias ? gridDim.x - main_blocks : main_blocks;
    for (long i = b0 * blockDim.x + threadIdx.x; i < n; i += nblk * blockDim.x) {
        float s = r.accumulate ? out[i] : 0.f;
        for (int z = 0; z < r.splits; ++z) s += slab[(long)z * n + i];
        out[i] = s;
    }
}

// many splits over a small matrix: 64 elements x 4 split-lanes per block, fixed summation tree (deterministic)
__global__ __launch_bounds__(256) void reduce_slabs_2d(const Reduce r, int main_blocks) {
    __shared__ float sm[4][64];
    const bool bias = (int)blockIdx.x >= main_blocks;
    const float* slab = bias ? r.bslab : r.slab;
    float* out = bias ? (blockIdx.y ? r.bout1 : r.bout) : (blockIdx.y ? r.out1 : r.out);
    const long n = bias ? r.nb : r.n;
    if (blockIdx.y) slab += (long)r.splits * n;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long i = (bias ? blockIdx.x - main_blocks : blockIdx.x) * 64L + tx;
    float s = 0.f;
    if (i < n) for (int z = ty; z < r.splits; z += 4) s += slab[(long)z * n + i];
    sm[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < n) out[i] = (r.accumulate ? out[i] : 0.f) + ((sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx]));
}

// ---------------------------------------------------------------------------------------------
// host side
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#include <cstdlib>
#include <climits>
#include <mutex>
// Tuning overrides for tools/gemm_bench.py sweeps.  The environment is read ONCE (std::call_once: the main thread and the
// autograd thread both launch GEMMs); an unset variable stays "unset", so every call site applies its OWN default --
// several sites pass shape-dependent defaults (round-1 bug: the first caller's default was cached for everybody).
enum { ENV_IG_T128, ENV_IG_BK32, ENV_WG_TARGET, ENV_WG_MINROWS, ENV_WG_TAPMAJOR, ENV_WG_BK32, ENV_WG_DMA, ENV_IG_HALO_MINC, ENV_IG_HALO, ENV_IG_T32, ENV_WG_INLAUNCH, ENV_WG_QUANT, ENV_IG_SPLITK, ENV_WG_LDSPAD, ENV_IG_SHORTK, ENV_WG_ATOMIC, ENV_WG_STEM, ENV_IG_SPLITK_MAXT, ENV_IG_SPLITK_TARGET, ENV_WG_SLOTS, ENV_WG_BUF, ENV_IG_BUF, ENV_WG_UNIFORM, ENV_IG_BUFSTORE, ENV_IG_BF16_STATS, ENV_IG_DMA, ENV_IG_DMA128, ENV_WG_BATCH_XCD, ENV_IG_GROUPM, ENV_COUNT };
static int env_int(int which, int dflt) {
    static const char* const names[ENV_COUNT] = {"PDF_IG_T128", "PDF_IG_BK32", "PDF_WG_TARGET", "PDF_WG_MINROWS", "PDF_WG_TAPMAJOR",
                                                 "PDF_WG_BK32", "PDF_WG_DMA", "PDF_IG_HALO_MINC", "PDF_IG_HALO", "PDF_IG_T32", "PDF_WG_INLAUNCH", "PDF_WG_QUANT", "PDF_IG_SPLITK", "PDF_WG_LDSPAD", "PDF_IG_SHORTK", "PDF_WG_ATOMIC", "PDF_WG_STEM", "PDF_IG_SPLITK_MAXT", "PDF_IG_SPLITK_TARGET", "PDF_WG_SLOTS", "PDF_WG_BUF", "PDF_IG_BUF", "PDF_WG_UNIFORM", "PDF_IG_BUFSTORE", "PDF_IG_BF16_STATS", "PDF_IG_DMA", "PDF_IG_DMA128", "PDF_WG_BATCH_XCD", "PDF_IG_GROUPM"};
    static int vals[ENV_COUNT];
    static std::once_flag once;
    std::call_once(once, [] {
        for (int i = 0; i < ENV_COUNT; ++i) {
            const char* e = getenv(names[i]);
            vals[i] = e ? atoi(e) : INT_MIN;
        }
    });
    return vals[which] == INT_MIN ? dflt : vals[which];
}

// ---- per-kernel timing (KTimer, gemm_common.h)
#include <string>
#include <vector>
#include <cstdio>
#include <cstring>
struct KRec { std::string name; hipEvent_t e0, e1; double flops, bytes; };
static std::vector<KRec> g_krecs;
static std::vector<hipEvent_t> g_kpool;                      // events are created when timing is switched on, not inside the timed step
static size_t g_kpool_next = 0;
static std::mutex g_krec_mu;
static int g_ktiming = 0;
#define PDF_KTIMER_POOL 8192
KTimer::KTimer(const char* name, double flops, double bytes, hipStream_t s) : slot(-1), stream(s) {
    if (!g_ktiming) return;
    std::lock_guard<std::mutex> lk(g_krec_mu);
    if (g_kpool_next + 2 > g_kpool.size()) return;           // pool exhausted: this launch goes unrecorded
    KRec r; r.name = name; r.flops = flops; r.bytes = bytes;
    r.e0 = g_kpool[g_kpool_next++]; r.e1 = g_kpool[g_kpool_next++];
    (void)hipEventRecord(r.e0, s);
    slot = (int)g_krecs.size();
    g_krecs.push_back(r);
}
KTimer::~KTimer() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_krec_mu);
    (void)hipEventRecord(g_krecs[slot].e1, stream);
}
// on != 0: drop the old records and start recording; 0: stop (the records stay readable)
PDF_API int pdf_debug_kernel_timing(int on) {
    std::lock_guard<std::mutex> lk(g_krec_mu);
    if (on) {
        g_krecs.clear();
        g_kpool_next = 0;
        while (g_kpool.size() < PDF_KTIMER_POOL) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return PDF_E_WORKSPACE;
            g_kpool.push_back(e);
        }
    }
    g_ktiming = on ? 1 : 0;
    return 0;
}
PDF_API int pdf_debug_kernel_record_count(void) { std::lock_guard<std::mutex> lk(g_krec_mu); return (int)g_krecs.size(); }
// record i -> kernel symbol, algorithmic FLOPs and bytes (operands once) of that launch, its duration (the device must be idle)
PDF_API int pdf_debug_kernel_record(int i, char* name, int cap, double* flops, double* bytes, float* ms) {
    std::lock_guard<std::mutex> lk(g_krec_mu);
    if (i < 0 || i >= (int)g_krecs.size() || cap < 2) return PDF_E_BADARG;
    const KRec& r = g_krecs[i];
    snprintf(name, cap, "%s", r.name.c_str());
    *flops = r.flops; *bytes = r.bytes;
    if (hipError_t e = hipEventElapsedTime(ms, r.e0, r.e1)) return (int)e;
    return 0;
}
static double igemm_bytes(const IGemm& g, int groups) {
    const double a = g.plain_in ? (double)g.M * g.Cin : (double)(g.M / max(1, g.QH * g.QW)) * g.H * g.W * g.Cin;
    return 4.0 * groups * (a + (double)g.N * g.K + (double)g.M * g.N * (g.accum ? 2 : 1));
}
static double wgemm_bytes(const WGemm& g, int groups) {
    const double q = g.plain_q ? (double)g.M * g.Cq : (double)(g.M / max(1, g.QH * g.QW)) * g.H * g.W * g.Cq;
    return 4.0 * groups * ((double)g.M * g.NI + q + (double)g.NI * g.T * g.Cq);
}

// C[m][n] = act(sum_k A[m][k] B[n][k] + bias[n]) for K <= 48, plain rows in / out: HBM-bound streaming (the backward-data
// of the 2- / 42-channel output convs: dy has 2 or 42 channels, dx 128-256).  B^T lives in LDS; one thread per (m, 4 n).
#define SMALLK_MAX 48
__global__ __launch_bounds__(256) void small_k_gemm(const IGemm g) {
    __shared__ float bt[SMALLK_MAX][260];                    // [k][n] for this block's <= 256 columns
    const int n0 = blockIdx.y * 256;
    const int nn = min(256, g.N - n0);
    if (g.b_kn) {                                            // [K][N] storage (backward-data of a 2- / 42-channel output conv)
        for (int i = threadIdx.x; i < nn * g.K; i += 256) {
            const int k = i / nn, n = i - k * nn;
            bt[k][n] = g.B[(long)k * g.ldb + n0 + n];
        }
    } else {
        for (int i = threadIdx.x; i < nn * g.K; i += 256) {
            const int n = i / g.K, k = i - n * g.K;
            bt[k][n] = g.B[(long)(n0 + n) * g.ldb + k];
        }
    }
    __syncthreads();
    const int nq = (nn + 3) / 4;
    const long total = (long)g.M * nq;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / nq;
        const int n4 = (int)(i - m * nq) * 4;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const float* ar = g.A + m * g.lda;
        for (int k = 0; k < g.K; ++k) {
            const float a = ar[k];
            a0 += a * bt[k][n4]; a1 += a * bt[k][n4 + 1]; a2 += a * bt[k][n4 + 2]; a3 += a * bt[k][n4 + 3];
        }
        float v[4] = {a0, a1, a2, a3};
        float* o = g.C + m * g.ldc + n0 + n4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (n4 + e < nn) {
                float x = v[e] + (g.bias ? g.bias[n0 + n4 + e] : 0.f);
                if (g.act == 1) x = fmaxf(x, 0.f); else if (g.act == 2) x = x > 0.f ? x : 0.1f * x;
                o[e] = x;
            }
        }
    }
}

// C[m][n] = act(bias[n] + sum over splits of part[z][m][n]) in split order (deterministic)
__global__ __launch_bounds__(256) void splitk_finish(const float* __restrict__ part, int splits, int M, int N, const float* __restrict__ bias,
                                                     int act, float* __restrict__ C, int ldc) {
    const long total = (long)M * N;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / N;
        const int n = (int)(i - m * N);
        float v = bias ? bias[n] : 0.f;
        for (int z = 0; z < splits; ++z) v += part[(long)z * total + i];
        if (act == 1) v = fmaxf(v, 0.f); else if (act == 2) v = v > 0.f ? v : 0.1f * v;
        C[m * ldc + n] = v;
    }
}

template <int BM, int BN, int WM, int WN, int BKF = 16>
static void launch_igemm_tile(const IGemm& g, bool fast, dim3 grid, hipStream_t s) {
    constexpr int NT = WM * WN * 64;
    char nm[96] = "";
    const bool buf = fast && g.abytes != 0 && env_int(ENV_IG_BUF, 1);
    if (g_ktiming && buf && !g.b_kn && g.a_scale != nullptr) snprintf(nm, sizeof nm, "igemm_nt_aff<%d, %d, %d, %d, %d>", BM, BN, WM, WN, BKF);
    else if (g_ktiming) snprintf(nm, sizeof nm, "igemm_nt<%d, %d, %d, %d, %s, %s, %d, %s>", BM, BN, WM, WN, fast ? "true" : "false", g.b_kn ? "true" : "false", fast ? BKF : 16, buf ? "true" : "false");
    KTimer kt(nm, 2.0 * g.M * g.N * g.K * grid.y, igemm_bytes(g, grid.y), s);
    if (buf && g.b_kn) hipLaunchKernelGGL((igemm_nt<BM, BN, WM, WN, true, true, BKF, true>), grid, dim3(NT), 0, s, g);
    else if (buf && g.a_scale != nullptr) hipLaunchKernelGGL((igemm_nt_aff<BM, BN, WM, WN, BKF>), grid, dim3(NT), 0, s, g);
    else if (buf) hipLaunchKernelGGL((igemm_nt<BM, BN, WM, WN, true, false, BKF, true>), grid, dim3(NT), 0, s, g);
    else if (fast && g.b_kn) hipLaunchKernelGGL((igemm_nt<BM, BN, WM, WN, true, true, BKF>), grid, dim3(NT), 0, s, g);
    else if (fast) hipLaunchKernelGGL((igemm_nt<BM, BN, WM, WN, true, false, BKF>), grid, dim3(NT), 0, s, g);
    else if (g.b_kn) hipLaunchKernelGGL((igemm_nt<BM, BN, WM, WN, false, true>), grid, dim3(NT), 0, s, g);
    else hipLaunchKernelGGL((igemm_nt<BM, BN, WM, WN, false, false>), grid, dim3(NT), 0, s, g);
}

// GEMM operand precision of the whole library: 0 = fp32 MFMA (exact fp32), 1 = bf16 MFMA with fp32 accumulation
// (gemm_bf16.hip; BASELINE configs 4 / 5).  Set once before training; shapes the bf16 kernels do not take (rows that are not
// 16-byte aligned, 3-channel stem) keep using the fp32 kernels.
static int g_gemm_bf16 = 0;
PDF_API int pdf_set_gemm_precision(int bf16) { g_gemm_bf16 = bf16 ? 1 : 0; return 0; }
PDF_API int pdf_debug_gemm_precision() { return g_gemm_bf16; }

static thread_local int g_last_tile = 0;            // BM * 1000 + BN of this thread's last implicit-GEMM launch (0: streaming small-K kernel)
static thread_local int g_igemm_launches = 0;
static int g_shadow_operands = 0;                   // bf16 shadow operands consumed by GEMM launches so far (tests)
PDF_API int pdf_debug_shadow_operands() { return g_shadow_operands; }       // implicit-GEMM kernel launches of this thread so far
PDF_API int pdf_debug_last_tile() { return g_last_tile; }
PDF_API int pdf_debug_igemm_launches() { return g_igemm_launches; }

// BatchNorm statistics out of a forward GEMM's epilogue (IGemm::stat): the request of the NEXT conv2d / linear forward call of
// this thread (pdf_set_stats_output) and what that call produced (pdf_stats_result_tiles / _rows; 0 tiles: the launch it chose
// has no statistics epilogue -- the caller then runs the ordinary statistics pass).
struct StatReq { float* part; long cap; };
static thread_local PdfCallOpts* tl_cur = nullptr;   // the call in progress on this thread (stat_plan publishes the layout it chose into it)
struct CurCall { PdfCallOpts* prev; explicit CurCall(PdfCallOpts& co) : prev(tl_cur) { tl_cur = &co; } ~CurCall() { tl_cur = prev; } };
static void stat_result(long tiles, long rows) { if (tl_cur != nullptr) { tl_cur->stats_tiles = tiles; tl_cur->stats_rows = rows; } }
// the launch about to be issued uses row blocks of BM rows: keep the request if the partials fit, and publish the layout
static void stat_plan(IGemm& g, long cap, int BM) {
    if (g.stat == nullptr) return;
    const long tiles = cdiv(g.M, BM);
    if (!g.plain_out || g.ps_cout > 0 || g.accum || tiles * g.N * 2 > cap) { g.stat = nullptr; return; }
    stat_result(tiles, BM);
}

// groups == 2: paired launch (see IGemm::B1), blockIdx.y selects the group
static int launch_igemm(IGemm& g, hipStream_t s, int groups = 1, long stat_cap = 0) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return 0;
    g.gm = env_int(ENV_IG_GROUPM, 4);               // (round 5: L2-miss reads of the 64x64 family 112 -> 81 MB per launch, of the transposed convolutions 1,539 -> 583; times unchanged -- profiles/r05_groupm.txt)
    ++g_igemm_launches;
    if (groups > 1) g.stat = nullptr;
    float* const stat_req = g.stat;
    g.stat = nullptr;                                   // (the streaming / split-K launches below have no statistics epilogue; bf16: whole tiles only)
    bool fast = (g.Cin % 16 == 0) && (g.lda % 4 == 0) && (g.ldb % 4 == 0) && aligned16(g.A) && aligned16(g.B);
    if (g.b_kn) fast = fast && (g.N % 4 == 0) && (g.btap % 4 == 0);
    if (groups > 1) fast = fast && aligned16(g.B1) && (g.gsA % 4 == 0);
    g.abytes = g.bbytes = 0;
    if (fast) {                                          // operand extents for the buffer-descriptor form (per group, from the base pointers)
        const double aext = 4.0 * (g.plain_in ? (double)g.M * g.lda : (double)cdiv(g.M, g.QH * g.QW) * g.H * g.W * g.lda);
        const double bext = 4.0 * (double)(g.b_kn ? g.Cin : g.N) * g.ldb;
        if (aext < 4294967000.0 && bext < 4294967000.0) { g.abytes = (unsigned)aext; g.bbytes = (unsigned)bext; }
    }
    g.cbytes = 0;
    if (fast && g.plain_out && g.ps_cout == 0 && 4.0 * g.M * g.ldc < 4294967000.0 && env_int(ENV_IG_BUFSTORE, 1)) g.cbytes = (unsigned)(4.0 * g.M * g.ldc);
    if (!fast && g.C16 == nullptr && groups == 1 && !g.accum && (!g.b_kn || (g.T == 1 && g.wt[0] == 0)) && g.plain_in && g.plain_out && g.K <= SMALLK_MAX && g.N >= 32 && g.ps_cout == 0 &&
        (long)g.M * g.N >= (1L << 20)) {
        dim3 grid(grid_for((long)g.M * ((min(g.N, 256) + 3) / 4)), cdiv(g.N, 256));
        KTimer kt("small_k_gemm", 2.0 * g.M * g.N * g.K, igemm_bytes(g, 1), s);
        hipLaunchKernelGGL(small_k_gemm, grid, dim3(256), 0, s, g);
        g_last_tile = 0;
        PDF_LAUNCH_CHECK();
        return 0;
    }
    if (!(g_gemm_bf16 && fast) || groups > 1) { g.A16 = nullptr; g.B16 = nullptr; g.B16T = nullptr; }
    if (g.B16T != nullptr && (!g.b_kn || g.ldbT % 8 != 0 || (reinterpret_cast<uintptr_t>(g.B16T) & 15))) g.B16T = nullptr;
    if (g.A16 != nullptr && (g.lda % 8 != 0 || (reinterpret_cast<uintptr_t>(g.A16) & 15))) g.A16 = nullptr;      // 16-byte chunks of 8 bf16
    if (g.B16 != nullptr && ((!g.b_kn && g.ldb % 8 != 0) || (reinterpret_cast<uintptr_t>(g.B16) & 15))) g.B16 = nullptr;
    if (g.A == nullptr && g.A16 == nullptr) return PDF_E_BADARG;          // bf16 storage mode: A exists only as bf16 -- it must be usable
    if (g.a_scale != nullptr && !(fast && !g_gemm_bf16 && groups == 1 && g.T == 1 && g.plain_in && !g.b_kn && g.abytes != 0 && g.K % 4 == 0 && env_int(ENV_IG_BUF, 1)))
        return PDF_E_BADARG;                             // the operand transform lives in the buffer-load form of igemm_nt only
    if (g_gemm_bf16 && fast) {
        g_shadow_operands += (g.A16 != nullptr) + (g.B16 != nullptr);
        // BatchNorm statistics out of the fp32 accumulators: the bf16 kernels take them in their whole-tile epilogue only
        const int bm16 = igemm_bf16_tile_rows(g, groups);
        if (g.C16 != nullptr && (groups != 1 || g.cbytes == 0 || g.M % bm16 != 0 || g.accum)) return PDF_E_BADARG;      // bf16 output: whole tiles only
        if (stat_req != nullptr && groups == 1 && g.cbytes != 0 && g.M % bm16 == 0 && env_int(ENV_IG_BF16_STATS, 1)) { g.stat = stat_req; stat_plan(g, stat_cap, bm16); }
        const int rc = launch_igemm_bf16(g, s, groups);
        if (rc < 0) return -rc;
        if (rc == 1) { g_last_tile = 16; return 0; }
        g.stat = nullptr;
        stat_result(0, 0);
    }
    if (g.C16 != nullptr) return PDF_E_BADARG;             // (only the bf16 kernels write a bf16 output)
    if (g.A == nullptr) return PDF_E_BADARG;               // (... and only they read a bf16-only operand)
    // Few output tiles under a long reduction (the M = 64 centre-window layers, 8x8-map 1x1 / 3x3 convs, the mesh decoder's vertex
    // up-projections): a handful of CUs would walk K step by step at load latency.  Split K over blockIdx.z so that ~512 blocks
    // run, partial tiles through the scratch ring, bias / activation in splitk_finish.  (round 3: up to 256 tiles instead of 128 and
    // ~512 blocks instead of ~320 -- ResNet layer-4 3x3: 79 -> 90 TFLOP/s forward and backward-data, its 1x1 backward-data 75 -> 83;
    // at 512 tiles the layer-3 1x1 layers lose 7 %; round 3)
    const long t64 = (long)cdiv(g.M, 64) * cdiv(g.N, 64);
    // (also the valid 3x3 convolutions on the 5x5 / 3x3 centre windows: taps are walked in K order, a split may start inside any tap)
    const bool sk_plain = g.T == 1 && g.plain_in, sk_taps = g.T > 1 && !g.plain_in && fast && g.Cin % 32 == 0 && g.K == g.T * g.Cin;
    if (groups == 1 && (sk_plain || sk_taps) && g.plain_out && g.ps_cout == 0 && !g.accum && t64 <= env_int(ENV_IG_SPLITK_MAXT, 256) && g.K >= 512 && env_int(ENV_IG_SPLITK, 1)) {
        const bool bk32 = fast && g.Cin % 32 == 0;
        const int bk = bk32 ? 32 : 16, nk = cdiv(g.K, bk);
        int splits = (int)min((long)cdiv(env_int(ENV_IG_SPLITK_TARGET, 512), (int)t64), (long)(g.K / 128));
        while (splits > 1 && (long)splits * g.M * g.N > PDF_SCRATCH_MAX) --splits;
        if (splits >= 2) {
            const int ksteps = cdiv(nk, splits);
            splits = cdiv(nk, ksteps);
            float* part = splits >= 2 ? pdf_scratch((long)splits * g.M * g.N) : nullptr;
            if (part != nullptr) {
                IGemm gs = g;
                gs.ksteps = ksteps; gs.part = part;
                const dim3 grid((unsigned)t64, 1, (unsigned)splits);
                const int dma = env_int(ENV_IG_DMA, 0);
                if (dma > 0 && fast && bk32 && launch_igemm_dma(gs, 64, dma - 1, 1, splits, s)) {}
                else if (bk32) launch_igemm_tile<64, 64, 2, 2, 32>(gs, fast, grid, s);
                else launch_igemm_tile<64, 64, 2, 2>(gs, fast, grid, s);
                KTimer kt("splitk_finish", 0.0, 4.0 * (splits + 1) * g.M * g.N, s);
                hipLaunchKernelGGL(splitk_finish, dim3(grid_for((long)g.M * g.N)), dim3(256), 0, s, part, splits, g.M, g.N, g.bias, g.act, g.C, g.ldc);
                g_last_tile = 64064;
                PDF_LAUNCH_CHECK();
                return 0;
            }
        }
    }
    // tile choice: wide tiles when there are enough of them to fill 256 CUs, else smaller ones
    long t128 = (long)cdiv(g.M, 128) * cdiv(g.N, 128) * groups;
    // (measured: below ~600 128x128 tiles the 64x64 kernel's 4x block count wins, e.g. ResNet layer2-4)
    // (a 256x128 tile -- 128 accumulator registers, one wave per SIMD -- was measured: 104 vs 123 TFLOP/s on the largest conv)
    // (also measured for this tile: 8 waves per block with K-step 32 -- <128,128,4,2,..,32>, 4 waves/SIMD, half the barriers
    // per flop: +1 % alone (125.5 vs 124.2 TFLOP/s on the largest conv), -0.3 % inside the step)
    // 1x1 layers with a short reduction (2-16 K-steps per tile) are all prologue and epilogue: the 64x64 kernel's 4x block count hides
    // them better (r02: ResNet 128->512 @32x32 49.7 -> 70.5 TFLOP/s, 64->256 @64x64 45.4 -> 53.5, 512->256 backward-data 61.6 -> 88.7);
    // the million-row PointNet++ linears keep the wide tile (54.5 vs 50.5)
    // (a batched launch -- the 16 transform-domain products of winograd.hip -- has its tile count multiplied by the batch: wide tiles)
    const bool short_k = g.T == 1 && g.batch == 0 && ((g.K <= 256 && g.M <= 262144) || (g.K <= 512 && g.M <= 32768)) && env_int(ENV_IG_SHORTK, 1);
    bool halo = fast && groups == 1 && g.T == 9 && !g.plain_in && g.plain_out && g.ps_cout == 0 && g.sy == 1 && g.sx == 1 && g.QW == g.W && g.QH == g.H &&
                (g.W == 64 || g.W == 32 || g.W == 16) && (g.H * g.W) % 128 == 0 && g.M % 128 == 0 && g.N > 64 && t128 >= env_int(ENV_IG_T128, 600) &&
                g.Cin >= env_int(ENV_IG_HALO_MINC, 256) &&      // measured: 128-channel layers lose (94 vs 107 TFLOP/s forward), 256+ gain 2-3 %
                env_int(ENV_IG_HALO, 1);
    for (int t = 0; halo && t < 9; ++t) halo = g.dy[t] >= -1 && g.dy[t] <= 1 && g.dx[t] >= -1 && g.dx[t] <= 1;
    g.stat = stat_req;
    if (halo) {
        stat_plan(g, stat_cap, 128);
        const dim3 grid((g.M / 128) * cdiv(g.N, 128));
        const bool buf = g.abytes != 0 && env_int(ENV_IG_BUF, 1);
        KTimer kt(g.b_kn ? (buf ? "igemm_halo3x3<true, true>" : "igemm_halo3x3<true, false>") : (buf ? "igemm_halo3x3<false, true>" : "igemm_halo3x3<false, false>"),
                  2.0 * g.M * g.N * g.K, igemm_bytes(g, 1), s);
        if (buf && g.b_kn) hipLaunchKernelGGL((igemm_halo3x3<true, true>), grid, dim3(256), 0, s, g);
        else if (buf) hipLaunchKernelGGL((igemm_halo3x3<false, true>), grid, dim3(256), 0, s, g);
        else if (g.b_kn) hipLaunchKernelGGL((igemm_halo3x3<true, false>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((igemm_halo3x3<false, false>), grid, dim3(256), 0, s, g);
        g_last_tile = 128128;
    }
    else if (g.N > 64 && t128 >= env_int(ENV_IG_T128, 600) && !short_k) {
        stat_plan(g, stat_cap, 128);
        const int dma = env_int(ENV_IG_DMA128, 0);
        if (!(dma > 0 && fast && g.batch == 0 && launch_igemm_dma(g, 128, dma - 1, groups, 0, s)))
            launch_igemm_tile<128, 128, 2, 2>(g, fast, dim3(cdiv(g.M, 128) * cdiv(g.N, 128), groups), s);
        g_last_tile = 128128;
    }
    else if (g.N <= 64 && (long)cdiv(g.M, 128) * groups >= env_int(ENV_IG_T128, 600))
        stat_plan(g, stat_cap, 128), launch_igemm_tile<128, 64, 4, 1>(g, fast, dim3(cdiv(g.M, 128) * cdiv(g.N, 64), groups), s), g_last_tile = 128064;   // (K-step 32: no gain here)
    else if (fast && (long)cdiv(g.M, 64) * cdiv(g.N, 64) * groups < 96 && g.K >= 512 && env_int(ENV_IG_T32, 1)) {
        // a handful of 64x64 tiles with a long reduction (M = 64 centre windows, the mesh decoder's 1024-wide layers): latency
        // bound on a few CUs -- 32x32 tiles put 4x as many blocks on the chip (one wave each)
        stat_plan(g, stat_cap, 32);
        launch_igemm_tile<32, 32, 1, 1>(g, fast, dim3(cdiv(g.M, 32) * cdiv(g.N, 32), groups), s), g_last_tile = 32032;
    }
    else
    {
        stat_plan(g, stat_cap, 64);
        const dim3 grid(cdiv(g.M, 64) * cdiv(g.N, 64), groups);
        const int dma = env_int(ENV_IG_DMA, 0);
        // K-step 32 for the small tile: its 8 MFMAs per wave and 16-wide step leave the barrier exposed (l4 3x3: 62 -> 72 TFLOP/s)
        // (a deep-ring form for the mesh decoder's latency-bound products -- the whole reduction in flight before the first MFMA, 4-8 stages
        // of the LDS-DMA kernel on <= 512 / 1024 tiles -- was measured: the pair entry points unchanged, the step 0.5-3 % slower;
        // profiles/r04_igemm_dma_ab.txt)
        if (dma > 0 && fast && g.batch == 0 && launch_igemm_dma(g, 64, dma - 1, groups, 0, s)) {}      // (batch == 0: igemm_dma reads blockIdx.y as the pair index, not as a plane of a batched launch -- it faulted on the Winograd products, round 6)
        else if (fast && g.Cin % 32 == 0 && env_int(ENV_IG_BK32, 1)) launch_igemm_tile<64, 64, 2, 2, 32>(g, fast, grid, s);
        else launch_igemm_tile<64, 64, 2, 2>(g, fast, grid, s);
        g_last_tile = 64064;
    }
    PDF_LAUNCH_CHECK();
    return 0;
}

// `batch` independent weight-gradient-shaped products in ONE launch of the LDS-DMA kernel: slab[b][y] [NI][NJ] = sum over rows m of split y
// of P_b[m][:]^T Q_b[m][:]  (P_b [M][NI], Q_b [M][NJ] plain rows; M % 16 == 0, rows_per_split % 16 == 0).  The caller sums the splits.
int pdf_internal_batched_wgemm(const float* P, const float* Q, float* slab, int batch, long gsP, long gsQ, int M, int NI, int NJ, int splits, hipStream_t s) {
    if (M % 16 != 0 || NI % 4 != 0 || NJ % 4 != 0 || splits < 1) return PDF_E_BADARG;
    WGemm g = {};
    g.P = P; g.Q = Q; g.M = M; g.NI = NI; g.Cq = NJ; g.T = 1; g.ldp = NI; g.ldq = NJ; g.ldw = NJ;
    g.H = 1; g.W = M; g.QH = 1; g.QW = M; g.sy = 1; g.sx = 1; g.plain_q = 1;
    g.dy[0] = 0; g.dx[0] = 0; g.wt[0] = 0;
    const int rps = cdiv(cdiv(M, splits), 16) * 16;
    splits = cdiv(M, rps);
    g.rows_per_split = rps;
    g.slab = slab; g.batch = batch; g.gsP = gsP; g.gsQ = gsQ; g.gsW = (long)splits * NI * NJ;
    g.beta = 0; g.wbytes = (unsigned)(4.0 * NI * NJ);
    g.pbytes = (unsigned)(4.0 * M * NI); g.qbytes = (unsigned)(4.0 * M * NJ);
    g.uniform = 2;
    dim3 grid((unsigned)(cdiv(NI, 128) * cdiv(NJ, 128)), (unsigned)splits, (unsigned)batch);
    if (env_int(ENV_WG_BATCH_XCD, 1)) {                     // the tiles of one (split, plane) on one XCD (WGemm::bsplits)
        g.bsplits = splits;
        grid = dim3((unsigned)(grid.x * 8 * cdiv(splits * batch, 8)));
    }
    KTimer kt("wgemm_tn_dma<3, true>", 2.0 * batch * M * NI * NJ, 4.0 * batch * ((double)M * (NI + NJ) + (double)splits * NI * NJ), s);
    hipLaunchKernelGGL((wgemm_tn_dma<3, true>), grid, dim3(256), 0, s, g);
    PDF_LAUNCH_CHECK();
    return splits;                                           // (> 0: the split count actually used)
}

// `batch` independent plain GEMMs C_b[M][N] = A_b[M][K] B_b[N][K]^T in ONE launch (fp32 MFMA kernels only): operand b at base + b * gs*.
// Used by the Winograd path (winograd.hip), whose 16 transform-domain products would each fill the chip only two thirds on their own.
int pdf_internal_batched_gemm(const float* A, const float* B, float* C, int batch, long gsA, long gsB, long gsC, int M, int N, int K, hipStream_t s) {
    IGemm g = {};
    g.A = A; g.B = B; g.C = C; g.bias = nullptr;
    g.M = M; g.N = N; g.K = K; g.Cin = K; g.lda = K; g.ldb = K; g.ldc = N;
    g.T = 1; g.plain_in = 1; g.plain_out = 1; g.act = 0;
    g.H = 1; g.W = M; g.QH = 1; g.QW = M; g.sy = 1; g.sx = 1;
    g.batch = batch; g.gsA = gsA; g.gsB = gsB; g.gsC = gsC;
    if (g_gemm_bf16) return PDF_E_BADARG;                   // (transform-domain products are fp32 only: the callers check the precision mode)
    return launch_igemm(g, s, batch);
}

long pdf_internal_x3_deconv_workspace(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int backward);
int pdf_internal_x3_deconv_fwd(const float* x, const float* w, const float* bias, float* y, float* ws, int N, int H, int W, int Cin, int Cout,
                               int KH, int KW, int stride, int OH, int OW, int ldy, hipStream_t s);
int pdf_internal_x3_deconv_bwd_data(const float* dy, const float* w, float* dx, float* ws, int N, int H, int W, int Cin, int lddx, int Cout,
                                    int KH, int KW, int stride, int OH, int OW, int lddy, hipStream_t s);
int pdf_internal_x3_deconv_general_fwd(const float* x, const float* w, const float* bias, float* y, float* ws, int N, int H, int W, int Cin, int Cout,
                                       int K, int stride, int pad, int OH, int OW, int ldy, hipStream_t s);
int pdf_internal_x3_deconv_general_bwd_data(const float* dy, const float* w, float* dx, float* ws, int N, int H, int W, int Cin, int lddx, int Cout,
                                            int K, int stride, int pad, int OH, int OW, int lddy, hipStream_t s);
int pdf_internal_x3_deconv_general_bwd_weight(const float* x, const float* dy, float* dw, float* ws, int N, int H, int W, int Cin, int Cout,
                                              int K, int stride, int pad, int OH, int OW, int lddy, int accumulate, hipStream_t s);
int pdf_internal_x3_deconv_bwd_weight(const float* x, const float* dy, float* dw, float* ws, int N, int H, int W, int Cin, int Cout,
                                      int KH, int KW, int stride, int OH, int OW, int lddy, int accumulate, hipStream_t s);
PDF_API long pdf_deconv2d_x3_workspace_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int backward) {
    return g_gemm_bf16 ? 0 : pdf_internal_x3_deconv_workspace(N, H, W, Cin, Cout, KH, KW, stride, pad, backward);
}
static void conv_taps(IGemm& g, int KH, int KW, int pad, int dil) {
    g.T = KH * KW;
    for (int ky = 0; ky < KH; ++ky)
        for (int kx = 0; kx < KW; ++kx) {
            int t = ky * KW + kx;
            g.dy[t] = (int)(ky * dil - pad); g.dx[t] = (int)(kx * dil - pad); g.wt[t] = t;
        }
}

struct Shadows { const void* op0; const void* op1; };

// Linear / 1x1: y[M][N] = act(x[M][K] w[N][K]^T + b).  Reference: nn.Linear / 1x1 nn.Conv2d call sites
// (e.g. model_attn/gcn.py:66, intaghand_encoder.py:48-103 netR_*, :205-219 SFT convs).
static IGemm linear_desc(const float* x, const float* w, const float* bias, float* y,
                         int M, int N, int K, int ldx, int ldw, int ldy, int act) {
    IGemm g = {};
    g.A = x; g.B = w; g.C = y; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.Cin = K; g.lda = ldx; g.ldb = ldw; g.ldc = ldy;
    g.T = 1; g.plain_in = 1; g.plain_out = 1; g.act = act;
    g.H = 1; g.W = M; g.QH = 1; g.QW = M; g.sy = 1; g.sx = 1;
    return g;
}
static int pdf_linear_fwd_impl(const float* x, const float* w, const float* bias, float* y,
                           int M, int N, int K, int ldx, int ldw, int ldy, int act, hipStream_t s, PdfCallOpts& co) {
    const Shadows sh = {co.op0_bf16, co.op1_bf16};
    const StatReq sr = {co.stats_out, co.stats_cap};
    const CurCall cur(co);
    IGemm g = linear_desc(x, w, bias, y, M, N, K, ldx, ldw, ldy, act);
    g.A16 = sh.op0; g.B16 = sh.op1;
    g.stat = sr.part;
    g.a_scale = co.in_scale; g.a_shift = co.in_shift;
    return launch_igemm(g, s, 1, sr.cap);
}
PDF_API int pdf_linear_fwd_x(const float* x, const float* w, const float* bias, float* y,
                           int M, int N, int K, int ldx, int ldw, int ldy, int act, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_linear_fwd_impl(x, w, bias, y, M, N, K, ldx, ldw, ldy, act, s, co); }
PDF_API int pdf_linear_fwd(const float* x, const float* w, const float* bias, float* y,
                           int M, int N, int K, int ldx, int ldw, int ldy, int act, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_linear_fwd_impl(x, w, bias, y, M, N, K, ldx, ldw, ldy, act, s, co); pdf_tls_publish(co); return rc; }

// Two same-shaped layers with their own parameters in ONE launch (the left / right hand branches of the mesh decoder,
// DualGraph.py:83-84, inter_attn.py:66-67): rows [0, M) of x / y belong to (w0, b0), rows [M, 2M) to (w1, b1).
static int pdf_linear_fwd_pair_impl(const float* x, const float* w0, const float* w1, const float* b0, const float* b1, float* y,
                                int M, int N, int K, int ldx, int ldw, int ldy, int act, hipStream_t s, PdfCallOpts& co) {
    IGemm g = linear_desc(x, w0, b0, y, M, N, K, ldx, ldw, ldy, act);
    g.B1 = w1; g.bias1 = b1; g.gsA = (long)M * ldx; g.gsC = (long)M * ldy;
    return launch_igemm(g, s, 2);
}
PDF_API int pdf_linear_fwd_pair_x(const float* x, const float* w0, const float* w1, const float* b0, const float* b1, float* y,
                                int M, int N, int K, int ldx, int ldw, int ldy, int act, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_linear_fwd_pair_impl(x, w0, w1, b0, b1, y, M, N, K, ldx, ldw, ldy, act, s, co); }
PDF_API int pdf_linear_fwd_pair(const float* x, const float* w0, const float* w1, const float* b0, const float* b1, float* y,
                                int M, int N, int K, int ldx, int ldw, int ldy, int act, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_linear_fwd_pair_impl(x, w0, w1, b0, b1, y, M, N, K, ldx, ldw, ldy, act, s, co); pdf_tls_publish(co); return rc; }

// dx[M][K] = dy[M][N] w[N][K]: the weight is read in its forward [N][K] storage (no transposed copy)
static int pdf_linear_bwd_data_impl(const float* dy, const float* w, float* dx, int M, int N, int K, int lddy, int ldw, int lddx,
                                hipStream_t s, PdfCallOpts& co) {
    const Shadows sh = {co.op0_bf16, co.op1_bf16};
    IGemm g = linear_desc(dy, w, nullptr, dx, M, K, N, lddy, ldw, lddx, 0);
    g.b_kn = 1; g.btap = 0;
    g.A16 = sh.op0; g.B16 = sh.op1;
    if (ldw == K) { g.B16T = co.op1_bf16_t; g.ldbT = N; }   // (the transposed shadow is of the dense [N][K] matrix)
    return launch_igemm(g, s);
}
PDF_API int pdf_linear_bwd_data_x(const float* dy, const float* w, float* dx, int M, int N, int K, int lddy, int ldw, int lddx,
                                hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_linear_bwd_data_impl(dy, w, dx, M, N, K, lddy, ldw, lddx, s, co); }
PDF_API int pdf_linear_bwd_data(const float* dy, const float* w, float* dx, int M, int N, int K, int lddy, int ldw, int lddx,
                                hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_linear_bwd_data_impl(dy, w, dx, M, N, K, lddy, ldw, lddx, s, co); pdf_tls_publish(co); return rc; }

static int pdf_linear_bwd_data_pair_impl(const float* dy, const float* w0, const float* w1, float* dx, int M, int N, int K,
                                     int lddy, int ldw, int lddx, hipStream_t s, PdfCallOpts& co) {
    IGemm g = linear_desc(dy, w0, nullptr, dx, M, K, N, lddy, ldw, lddx, 0);
    g.b_kn = 1; g.btap = 0;
    g.B1 = w1; g.bias1 = nullptr; g.gsA = (long)M * lddy; g.gsC = (long)M * lddx;
    return launch_igemm(g, s, 2);
}
PDF_API int pdf_linear_bwd_data_pair_x(const float* dy, const float* w0, const float* w1, float* dx, int M, int N, int K,
                                     int lddy, int ldw, int lddx, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_linear_bwd_data_pair_impl(dy, w0, w1, dx, M, N, K, lddy, ldw, lddx, s, co); }
PDF_API int pdf_linear_bwd_data_pair(const float* dy, const float* w0, const float* w1, float* dx, int M, int N, int K,
                                     int lddy, int ldw, int lddx, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_linear_bwd_data_pair_impl(dy, w0, w1, dx, M, N, K, lddy, ldw, lddx, s, co); pdf_tls_publish(co); return rc; }


// ---------------------------------------------------------------------------------------------
// Tiny-channel layers: HBM-bound streaming kernels instead of 128-wide MFMA tiles that would be > 95 % padding.
// e_conv1 (intaghand_encoder.py:711: Conv2d(3, 3, 3, padding=1) on the full-resolution image): one thread per output pixel.
template <int CI, int CO, int KH, int KW>
__global__ __launch_bounds__(256) void tiny_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ y, int N, int H, int W, int ldx, int pad, int OH, int OW, int ldy, int act) {
    __shared__ float ws[CO * KH * KW * CI];
    for (int i = threadIdx.x; i < CO * KH * KW * CI; i += 256) ws[i] = w[i];
    __syncthreads();
    const long total = (long)N * OH * OW;
    for (long p = blockIdx.x * 256L + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
        const int ox = (int)(p % OW), oy = (int)((p / OW) % OH);
        const long n = p / ((long)OW * OH);
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] = bias ? bias[c] : 0.f;
#pragma unroll
        for (int ky = 0; ky < KH; ++ky)
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const int iy = oy + ky - pad, ix = ox + kx - pad;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const float* xp = x + ((n * H + iy) * W + ix) * ldx;
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) {
                    const float v = xp[ci];
#pragma unroll
                    for (int c = 0; c < CO; ++c) acc[c] = fmaf(v, ws[((c * KH + ky) * KW + kx) * CI + ci], acc[c]);
                }
            }
        float* yp = y + p * ldy;
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            float v = acc[c];
            if (act == 1) v = fmaxf(v, 0.f); else if (act == 2) v = v > 0.f ? v : 0.1f * v;
            yp[c] = v;
        }
    }
}
// its weight gradient: every thread keeps all CO*KH*KW*CI sums in registers over a grid-stride loop of pixels; one partial
// row per block in the workspace, summed by reduce_slabs (fixed order: deterministic)
template <int CI, int CO, int KH, int KW>
__global__ __launch_bounds__(256) void tiny_conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                              int N, int H, int W, int ldx, int pad, int OH, int OW, int lddy) {
    constexpr int NW = CO * KH * KW * CI;
    __shared__ float red[4][NW];
    float acc[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) acc[i] = 0.f;
    const long total = (long)N * OH * OW;
    for (long p = blockIdx.x * 256L + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
        const int ox = (int)(p % OW), oy = (int)((p / OW) % OH);
        const long n = p / ((long)OW * OH);
        float g[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) g[c] = dy[p * lddy + c];
#pragma unroll
        for (int ky = 0; ky < KH; ++ky)
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const int iy = oy + ky - pad, ix = ox + kx - pad;
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                const float* xp = x + ((n * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * ldx;
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) {
                    const float v = ok ? xp[ci] : 0.f;
#pragma unroll
                    for (int c = 0; c < CO; ++c) acc[((c * KH + ky) * KW + kx) * CI + ci] = fmaf(g[c], v, acc[((c * KH + ky) * KW + kx) * CI + ci]);
                }
            }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NW; i += 256) part[(long)blockIdx.x * NW + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// Weight (and bias) gradient of a layer with <= 4 output channels on plain rows (the 2-channel hm / mask heads):
// dW[i][j] = sum_m P[m][i] Q[m][j], K = NJ <= 1024 columns, one float4 column group per thread, the block's row lanes reduced
// through LDS; per-block partials [blk][NI*NJ (+NI)] in the workspace, summed by reduce_slabs.
__global__ __launch_bounds__(256) void narrow_wgrad_kernel(const float* __restrict__ P, const float* __restrict__ Q, float* __restrict__ part, float* __restrict__ bpart,
                                                           int M, int NI, int NJ, int ldp, int ldq, int rows_per_block) {
    __shared__ float red[256 * 16];
    __shared__ float bred[256 * 4];
    const int cgs = NJ / 4, rl = 256 / cgs;                  // column groups, row lanes
    const int cg = threadIdx.x % cgs, lr = threadIdx.x / cgs;
    float acc[4][4] = {}, bs[4] = {};
    const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
    if (lr < rl) {
        for (int m = m0 + lr; m < m1; m += rl) {
            const float4 q = *reinterpret_cast<const float4*>(Q + (long)m * ldq + cg * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < NI) {
                    const float p = P[(long)m * ldp + i];
                    acc[i][0] = fmaf(p, q.x, acc[i][0]); acc[i][1] = fmaf(p, q.y, acc[i][1]);
                    acc[i][2] = fmaf(p, q.z, acc[i][2]); acc[i][3] = fmaf(p, q.w, acc[i][3]);
                    if (cg == 0) bs[i] += p;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[threadIdx.x * 16 + i * 4 + e] = acc[i][e];
#pragma unroll
    for (int i = 0; i < 4; ++i) bred[threadIdx.x * 4 + i] = bs[i];
    __syncthreads();
    if (lr == 0) {
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = 0.f;
                for (int r = 0; r < rl; ++r) t += red[(r * cgs + cg) * 16 + i * 4 + e];
                part[(long)blockIdx.x * NI * NJ + (long)i * NJ + cg * 4 + e] = t;
            }
        if (cg == 0 && bpart != nullptr)
            for (int i = 0; i < NI; ++i) {
                float t = 0.f;
                for (int r = 0; r < rl; ++r) t += bred[(r * cgs) * 4 + i];
                bpart[(long)blockIdx.x * NI + i] = t;
            }
    }
}

// Conv2d forward on NHWC.  w is [Cout][KH][KW][Cin] (the channels_last storage of an OIHW weight).
// Replaces nn.Conv2d.forward at intaghand_encoder.py:711-772,790-791 and resnet.py:202-218.

// Forward of the ResNet stem (7x7, stride 2, pad 3, 3 -> 64 channels) on the matrix pipe: the generic implicit GEMM gathers its
// 3-channel taps element by element (43 TFLOP/s, 0.22 ms at B = 32, the first kernel of every step).  A block keeps the weights
// in LDS as [j = (ky, kx, ci)][co] (148 rows, the last one zero), stages per chunk of 64 output pixels of one row the 7 x 133 x 3
// input patch, and each of its four waves computes one 32 pixel x 32 channel tile over the 74 K-steps: A = the im2col row
// gathered from the patch (offset of column j is a compile-time constant, selected per half-wave), B = a weight row.
__host__ __device__ constexpr int stem_off(int j) { return j < 147 ? ((j / 3 / 7) * 133 + (j / 3) % 7) * 3 + j % 3 : 2800; }
template <int KS>
__device__ __forceinline__ void stem_fwd_steps(const float* Xs, const float* Ws, int pxo, int hi, int bcol, f32x16& acc) {
    if constexpr (KS < 74) {
        const int off = hi ? stem_off(2 * KS + 1) : stem_off(2 * KS);
        const float a = Xs[off + pxo];
        const float b = Ws[(2 * KS) * 64 + bcol];                  // bcol already holds hi * 64 + column
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        stem_fwd_steps<KS + 1>(Xs, Ws, pxo, hi, bcol, acc);
    }
}
__global__ __launch_bounds__(256) void stem7x7_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                          int N, int H, int W, int OH, int OW, int ldy, int act, int cpb, float* __restrict__ stat) {
    constexpr int CH = 64, XW = 2 * CH + 5, RW = XW * 3, XF = 7 * RW, DYB = 3200;
    __shared__ __attribute__((aligned(16))) float Xs[DYB];          // [0, XF): patch, [XF, DYB): zeros
    __shared__ __attribute__((aligned(16))) float Ws[148 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = XF + tid; i < DYB; i += 256) Xs[i] = 0.f;
    for (int i = tid; i < 148 * 64; i += 256) {
        const int j = i >> 6, co = i & 63;
        Ws[i] = j < 147 ? w[co * 147 + j] : 0.f;
    }
    const int pt = wave >> 1, ct = wave & 1, hi = lane >> 5;
    const int pxo = (32 * pt + (lane & 31)) * 6, bcol = hi * 64 + 32 * ct + (lane & 31);
    const int cpr = OW / CH;
    const long total = (long)N * OH * cpr;
    const long c0 = (long)blockIdx.x * cpb, c1 = min(total, c0 + cpb);
    float xr[14];
    auto gload = [&](long c) {                                     // see stem7x7_wgrad_kernel
        const int n = (int)(c / ((long)OH * cpr));
        const int rem = (int)(c - (long)n * OH * cpr);
        const int oy = rem / cpr, ox0 = (rem - oy * cpr) * CH;
        const int lin0 = (2 * ox0 - 3) * 3;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int iy = 2 * oy - 3 + r;
            const float* row = x + ((long)(n * H + iy) * W) * 3;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = tid + 256 * h, lin = lin0 + e;
                const bool ok = iy >= 0 && iy < H && e < RW && lin >= 0 && lin < 3 * W;
                xr[r * 2 + h] = ok ? row[lin] : 0.f;
            }
        }
    };
    StatAcc st[1] = {StatAcc{0.f, 0.f, 0.f, 0.f}};             // BatchNorm statistics of this block's rows (IGemm::stat form: one row block per BLOCK)
    if (c0 < c1) gload(c0);
    for (long c = c0; c < c1; ++c) {
        const int n = (int)(c / ((long)OH * cpr));
        const int rem = (int)(c - (long)n * OH * cpr);
        const int oy = rem / cpr, ox0 = (rem - oy * cpr) * CH;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h) { const int e = tid + 256 * h; if (e < RW) Xs[r * RW + e] = xr[r * 2 + h]; }
        __syncthreads();
        if (c + 1 < c1) gload(c + 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        stem_fwd_steps<0>(Xs, Ws, pxo, hi, bcol, acc);
        float* yp = y + ((long)(n * OH + oy) * OW + ox0 + 32 * pt) * ldy + 32 * ct + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r];
            if (act == 1) v = fmaxf(v, 0.f); else if (act == 2) v = v > 0.f ? v : 0.1f * v;
            yp[(long)((r & 3) + 8 * (r >> 2) + 4 * hi) * ldy] = v;
            if (stat != nullptr) stat_add(st[0], v);
        }
    }
    if (stat != nullptr) {
        __syncthreads();                                            // the patch is no longer read: its LDS carries the wave partials
        stat_finish<1, 2, 2, 64>(st, Xs, stat, blockIdx.x, 0, 64, pt, ct, lane, tid);
    }
}

static int pdf_conv2d_fwd_impl(const float* x, const float* w, const float* bias, float* y,
                           int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                           int stride, int pad, int OH, int OW, int ldy, int act, hipStream_t s, PdfCallOpts& co) {
    const Shadows sh = {co.op0_bf16, co.op1_bf16};
    const StatReq sr = {co.stats_out, co.stats_cap};
    const CurCall cur(co);
    unsigned short* y16 = reinterpret_cast<unsigned short*>(co.out_bf16);      // bf16 storage mode: the output goes here INSTEAD of y
    if (KH * KW > MAX_TAPS) return PDF_E_BADARG;
    if (y16 != nullptr && !(g_gemm_bf16 && Cout % 2 == 0 && ldy % 2 == 0)) return PDF_E_BADARG;
    if (Cin == 3 && Cout == 3 && KH == 3 && KW == 3 && stride == 1 && (long)N * OH * OW >= (1L << 16)) {
        if (y16 != nullptr) return PDF_E_BADARG;
        KTimer kt("tiny_conv_fwd_kernel<3, 3, 3, 3>", 2.0 * N * OH * OW * 81, 4.0 * N * (H * W + OH * OW) * 3, s);
        hipLaunchKernelGGL((tiny_conv_fwd_kernel<3, 3, 3, 3>), dim3(grid_for((long)N * OH * OW)), dim3(256), 0, s, x, w, bias, y, N, H, W, ldx, pad, OH, OW, ldy, act);
        g_last_tile = 0;
        PDF_LAUNCH_CHECK();
        return 0;
    }
    if (Cin == 3 && ldx == 3 && Cout == 64 && KH == 7 && KW == 7 && stride == 2 && pad == 3 && bias == nullptr && OW % 64 == 0 && OH * 2 == H && OW * 2 == W &&
        env_int(ENV_WG_STEM, 1)) {
        if (y16 != nullptr) return PDF_E_BADARG;
        const long total = (long)N * OH * (OW / 64);
        int nblk = (int)min((long)512, total);
        const int cpb = (int)cdiv(total, nblk);
        nblk = (int)cdiv(total, cpb);
        KTimer kt("stem7x7_fwd_kernel", 2.0 * N * OH * OW * 64 * 147, 4.0 * N * ((double)H * W * 3 + (double)OH * OW * 64), s);
        float* stat = (sr.part != nullptr && (long)nblk * 64 * 2 <= sr.cap) ? sr.part : nullptr;
        if (stat != nullptr) stat_result(nblk, (long)cpb * 64);
        hipLaunchKernelGGL(stem7x7_fwd_kernel, dim3(nblk), dim3(256), 0, s, x, w, y, N, H, W, OH, OW, ldy, act, cpb, stat);
        g_last_tile = 0;
        PDF_LAUNCH_CHECK();
        return 0;
    }
    // Winograd F(4x4, 3x3) / F(2x2, 3x3) (winograd.hip: pdf_internal_wino_tile picks): fp32 mode, the caller handed a workspace (PdfCallOpts::ws), the layer qualifies
    if (!g_gemm_bf16 && co.ws != nullptr && y16 == nullptr && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(co.ws) &&
        pdf_internal_wino_eligible(N, H, W, Cin, Cout, KH, KW, stride, pad, 0) && co.ws_floats >= pdf_internal_wino_workspace(N, H, W, Cin, Cout, 0)) {
        g_last_tile = 128128;
        return pdf_internal_conv3x3_winograd(x, ldx, w, bias, y, ldy, co.ws, N, H, W, Cin, Cout, act, 0, 0,
                                             (co.wino_v != nullptr && aligned16(co.wino_v) && pdf_internal_wino_v_offset(N, H, W, Cin, Cout) >= 0) ? co.wino_v : nullptr, s);
    }
    IGemm g = {};
    g.A = x; g.B = w; g.C = y; g.bias = bias;
    g.M = N * OH * OW; g.N = Cout; g.K = KH * KW * Cin; g.Cin = Cin; g.lda = ldx; g.ldb = KH * KW * Cin; g.ldc = ldy;
    g.H = H; g.W = W; g.QH = OH; g.QW = OW; g.sy = stride; g.sx = stride;
    conv_taps(g, KH, KW, pad, 1);
    g.plain_in = (KH == 1 && KW == 1 && stride == 1 && pad == 0) ? 1 : 0;
    g.plain_out = 1; g.act = act;
    g.A16 = sh.op0; g.B16 = sh.op1;
    g.stat = sr.part;
    g.C16 = y16;
    return launch_igemm(g, s, 1, sr.cap);
}
PDF_API int pdf_conv2d_fwd_x(const float* x, const float* w, const float* bias, float* y,
                           int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                           int stride, int pad, int OH, int OW, int ldy, int act, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_conv2d_fwd_impl(x, w, bias, y, N, H, W, Cin, ldx, Cout, KH, KW, stride, pad, OH, OW, ldy, act, s, co); }
PDF_API int pdf_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                           int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                           int stride, int pad, int OH, int OW, int ldy, int act, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_conv2d_fwd_impl(x, w, bias, y, N, H, W, Cin, ldx, Cout, KH, KW, stride, pad, OH, OW, ldy, act, s, co); pdf_tls_publish(co); return rc; }


// Conv2d backward-data: dx[N,H,W,Cin] from dy[N,OH,OW,Cout] and the FORWARD weight w = [Cout][KH][KW][Cin], read as
// the [K = (tap, co)][N = ci] operand it is.  One launch per input-parity class so a stride-s conv
// never multiplies zeros.  dx must be zero-filled by the caller when stride > kernel (1x1 s2).
static int conv2d_bwd_data(const float* dy, const float* w, float* dx,
                           int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW,
                           int stride, int pad, int OH, int OW, int lddy, int accumulate, hipStream_t s, const PdfCallOpts& co) {
    const Shadows sh = {co.op0_bf16, co.op1_bf16};
    if (KH * KW > MAX_TAPS) return PDF_E_BADARG;
    if (accumulate && stride > 1) return PDF_E_BADARG;       // (every dx element must be written by exactly one launch)
    // Winograd: dx = the 3x3 convolution of dy (Cout channels) with the mirrored taps, Cin output channels
    if (!g_gemm_bf16 && co.ws != nullptr && lddx % 4 == 0 && lddy % 4 == 0 && aligned16(dy) && aligned16(dx) && aligned16(co.ws) && OH == H && OW == W &&
        pdf_internal_wino_eligible(N, H, W, Cout, Cin, KH, KW, stride, pad, 1) && co.ws_floats >= pdf_internal_wino_workspace(N, H, W, Cout, Cin, 1)) {
        g_last_tile = 128128;
        return pdf_internal_conv3x3_winograd(dy, lddy, w, nullptr, dx, lddx, co.ws, N, H, W, Cout, Cin, 0, accumulate, 1, nullptr, s);
    }
    for (int py = 0; py < stride; ++py)
        for (int px = 0; px < stride; ++px) {
            IGemm g = {};
            g.A = dy; g.B = w; g.C = dx; g.bias = nullptr;
            g.N = Cin; g.Cin = Cout; g.lda = lddy; g.ldb = KH * KW * Cin; g.b_kn = 1; g.btap = Cin; g.ldc = lddx;
            g.H = OH; g.W = OW;
            g.QH = (H - py + stride - 1) / stride; g.QW = (W - px + stride - 1) / stride;
            if (g.QH <= 0 || g.QW <= 0) continue;
            g.M = N * g.QH * g.QW;
            g.sy = 1; g.sx = 1; g.accum = accumulate;
            g.A16 = sh.op0; g.B16 = sh.op1;
            g.B16T = co.op1_bf16_t; g.ldbT = KH * KW * Cout;
            // input row iy = qy*stride + py; contributing taps: (iy + pad - ky) % stride == 0, oy = (iy+pad-ky)/stride
            int T = 0;
            for (int ky = 0; ky < KH; ++ky) {
                if ((py + pad - ky) % stride != 0) continue;
                for (int kx = 0; kx < KW; ++kx) {
                    if ((px + pad - kx) % stride != 0) continue;
                    // oy = qy + (py + pad - ky)/stride  (exact division, may be negative)
                    int ny = py + pad - ky, nx = px + pad - kx;
                    g.dy[T] = (int)(ny >= 0 ? ny / stride : -((-ny) / stride));
                    g.dx[T] = (int)(nx >= 0 ? nx / stride : -((-nx) / stride));
                    g.wt[T] = (int)(ky * KW + kx);
                    ++T;
                }
            }
            if (T == 0) continue;
            g.T = T; g.K = T * Cout;
            g.plain_in = 0; g.plain_out = 0;
            g.OH = H; g.OW = W; g.osy = stride; g.osx = stride; g.ooy = py; g.oox = px;
            if (stride == 1) { g.plain_out = 1; if (KH == 1 && KW == 1 && pad == 0) g.plain_in = 1; }
            int rc = launch_igemm(g, s);
            if (rc) return rc;
        }
    return 0;
}

static int pdf_conv2d_bwd_data_impl(const float* dy, const float* w, float* dx,
                                int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW,
                                int stride, int pad, int OH, int OW, int lddy, hipStream_t s, PdfCallOpts& co) {
    return conv2d_bwd_data(dy, w, dx, N, H, W, Cin, lddx, Cout, KH, KW, stride, pad, OH, OW, lddy, 0, s, co);
}
PDF_API int pdf_conv2d_bwd_data_x(const float* dy, const float* w, float* dx,
                                int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW,
                                int stride, int pad, int OH, int OW, int lddy, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_conv2d_bwd_data_impl(dy, w, dx, N, H, W, Cin, lddx, Cout, KH, KW, stride, pad, OH, OW, lddy, s, co); }
PDF_API int pdf_conv2d_bwd_data(const float* dy, const float* w, float* dx,
                                int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW,
                                int stride, int pad, int OH, int OW, int lddy, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_conv2d_bwd_data_impl(dy, w, dx, N, H, W, Cin, lddx, Cout, KH, KW, stride, pad, OH, OW, lddy, s, co); pdf_tls_publish(co); return rc; }

// dx += the same (stride 1): the gradient of a tensor with two consumers -- a ResNet block input feeds conv1 and the shortcut --
// is accumulated by the second producer's epilogue instead of a separate add pass over both gradients
static int pdf_conv2d_bwd_data_add_impl(const float* dy, const float* w, float* dx,
                                    int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW,
                                    int stride, int pad, int OH, int OW, int lddy, hipStream_t s, PdfCallOpts& co) {
    return conv2d_bwd_data(dy, w, dx, N, H, W, Cin, lddx, Cout, KH, KW, stride, pad, OH, OW, lddy, 1, s, co);
}
PDF_API int pdf_conv2d_bwd_data_add_x(const float* dy, const float* w, float* dx,
                                    int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW,
                                    int stride, int pad, int OH, int OW, int lddy, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_conv2d_bwd_data_add_impl(dy, w, dx, N, H, W, Cin, lddx, Cout, KH, KW, stride, pad, OH, OW, lddy, s, co); }
PDF_API int pdf_conv2d_bwd_data_add(const float* dy, const float* w, float* dx,
                                    int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW,
                                    int stride, int pad, int OH, int OW, int lddy, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_conv2d_bwd_data_add_impl(dy, w, dx, N, H, W, Cin, lddx, Cout, KH, KW, stride, pad, OH, OW, lddy, s, co); pdf_tls_publish(co); return rc; }


// Split count of a weight-gradient launch: the candidate with the smallest modelled time (see launch_wgemm).
//   tiles: output tiles x groups; occ: blocks a CU holds; cap: most splits allowed (rows, workspace); rq: row quantum of a split;
//   tile: 64 or 128; slab_bytes: bytes of one split's partial result.
static int wg_choose_splits(long tiles, int M, int occ, int cap, int rq, int tile, double slab_bytes) {
    // time for m co-resident blocks on one CU, in units of "one block alone at full pipe efficiency" (measured efficiencies of
    // 1 / 2 / 3 / 4 resident blocks of these kernels: ~0.5 / 0.72 / 0.83 / 0.85)
    static const double G[5] = {0.0, 2.0, 2.78, 3.61, 4.7};
    const double row_ns = (tile == 128 ? 32768.0 : 8192.0) / 614.5;       // one row of a tile at 157.3 TFLOP/s / 256 CUs
    const double t0_rows = tile == 128 ? 64.0 : 160.0;                    // prologue + epilogue of a block, in rows
    int best = 1; double best_t = 1e300;
    for (int sp = 1; sp <= cap; ++sp) {
        const int r = cdiv(cdiv(M, sp), rq) * rq;
        const int nsp = cdiv(M, r);
        if (nsp != sp) continue;                                           // (not a distinct candidate)
        const long blocks = tiles * nsp;
        const long busiest = (blocks + 255) / 256;
        if (busiest > occ && sp > 1) break;                               // stay within one round of `occ` blocks per CU
        const double rounds = (double)(busiest / occ) * G[occ] + G[busiest % occ];
        double t = rounds * (r + t0_rows) * row_ns;
        if (nsp > 1) t += 5000.0 + nsp * slab_bytes / 3000.0;              // the reduction pass: launch + slabs at ~3 TB/s (bytes / (B/ns))
        if (t < best_t) { best_t = t; best = sp; }
    }
    return best;
}

// out1 != nullptr: paired launch (see WGemm::gsP); ws then holds both groups' slabs
// db / db1 != nullptr: also the bias gradient (column sums of P), see WGemm::bslab
static int launch_wgemm(WGemm& g, float* out, float* ws, long ws_floats, int accumulate, hipStream_t s, float* out1 = nullptr,
                        float* db = nullptr, float* db1 = nullptr) {
    const int NJ = g.T * g.Cq;
    if (g.M <= 0 || g.NI <= 0 || NJ <= 0) return 0;
    const int groups = out1 ? 2 : 1;
    const bool fast = (g.NI % 4 == 0) && (g.Cq % 4 == 0) && (g.ldp % 4 == 0) && (g.ldq % 4 == 0) &&
                      aligned16(g.P) && aligned16(g.Q) && (g.gsP % 4 == 0) && (g.gsQ % 4 == 0);
    // 64x64 tiles when a side is narrow, and for the mesh decoder's small layers (measured: 44 vs 19 TFLOP/s at M = 2016, 256x1024;
    // the ResNet layer-3/4 convolutions, same M but larger matrices, stay on 128x128: 77 vs 56)
    const bool small = (g.NI <= 64 || NJ <= 64) || ((long)g.NI * NJ <= (1L << 18) && g.M <= 16384);
    const int BI = small ? 64 : 128, BJ = small ? 64 : 128;
    long tiles = (long)cdiv(g.NI, BI) * cdiv(NJ, BJ);
    // split policy (measured, tools/gemm_bench.py): big gradient matrices want ~1024 blocks; few-tile / huge-M
    // (HBM-bound) ones ~512 longer-running blocks; small M may go down to 128 rows per split to fill the chip
    // Split count (round 3).  Every block of the launch runs equally long and a CU holds `occ` of them at once (128x128 LDS-DMA
    // kernel: 48 KB of LDS, 144 VGPRs -> 3; 64x64 kernel: 4; bf16 kernels: 2 / 3), sharing the CU's matrix pipe -- so what counts
    // is the busiest CU: how many blocks it gets and in how many rounds of `occ` it runs them.  The r02 rule aimed at ~1,024 blocks
    // rounded to multiples of 256: `feat` ran 1,008 blocks, i.e. a fourth block on most CUs that ran alone at half the pipe
    // efficiency.  wg_choose_splits() prices every candidate with a small model (blocks on the busiest CU, efficiency of 1 / 2 /
    // 3 / 4 co-resident blocks, a fixed cost per block, the slab reduction) and stays within one round.  Measured with it
    // (tools/gemm_bench.py, TFLOP/s): feat 104.8 -> 112, p2 / hm 102.6 -> 116, decoder 3x3 88.9 -> 101, l3 3x3 70.8 -> 82.6,
    // l4 3x3 71.7 -> 82.3; PDF_WG_TARGET=<blocks> restores the r02 rule.
    const bool bf16_mode = g_gemm_bf16 && fast;
    const int occ = env_int(ENV_WG_SLOTS, 0) > 0 ? max(1, min(4, env_int(ENV_WG_SLOTS, 0) / 256)) : (bf16_mode ? (small ? 3 : 2) : (small ? 4 : 3));      // (the efficiency table of wg_choose_splits covers 1-4 co-resident blocks)
    const int target = env_int(ENV_WG_TARGET, 0);
    int splits = target > 0 ? (int)((target + tiles * groups - 1) / (tiles * groups)) : 0;
    int max_by_rows = cdiv(g.M, env_int(ENV_WG_MINROWS, g.M >= 16384 ? 512 : 128));
    if (splits > max_by_rows) splits = max_by_rows;
    long per = (long)g.NI * g.ldw;
    const long perb = db ? g.NI : 0;
    if (target > 0 && (long)splits * (per + perb) * groups > ws_floats) splits = (int)(ws_floats / ((per + perb) * groups));
    const bool bf16 = g_gemm_bf16 && fast;
    if (target <= 0) {
        int cap = max_by_rows;
        if ((long)cap * (per + perb) * groups > ws_floats) cap = (int)(ws_floats / ((per + perb) * groups));
        splits = wg_choose_splits(tiles * groups, g.M, occ, max(cap, 1), bf16 ? 64 : 16, small ? 64 : 128, (double)(per + perb) * 4.0);
    }
    if (splits < 1) splits = 1;
    if (!bf16 || groups > 1) { g.P16 = nullptr; g.Q16 = nullptr; }
    if (db != nullptr) g.P16 = nullptr;                     // the fused bias gradient sums the un-rounded rows of P
    // (16-byte groups of 8 bf16 along the operand's columns)
    if (g.P16 != nullptr && ((reinterpret_cast<uintptr_t>(g.P16) & 15) || g.ldp % 8 || g.NI % 8 || g.gsP % 8)) g.P16 = nullptr;
    if (g.Q16 != nullptr && ((reinterpret_cast<uintptr_t>(g.Q16) & 15) || g.ldq % 8 || g.Cq % 8 || g.gsQ % 8)) g.Q16 = nullptr;
    const int rq = bf16 ? 64 : 16;
    // The blocks all run equally long and the chip retires them CU by CU: 1044 blocks on 256 CUs leave most CUs idle for the
    // fifth pass (82 % busy).  Among split counts down to 3/4 of the target, take the one whose block count fills whole
    // passes best (never more splits than the workspace was sized for).
    if (splits > 1 && target > 0 && env_int(ENV_WG_QUANT, 1)) {
        int best = splits; double beste = -1.0;
        for (int sp = splits; sp >= 1 && sp * 4 >= splits * 3; --sp) {
            const int r = cdiv(cdiv(g.M, sp), rq) * rq;
            const long blocks = tiles * groups * cdiv(g.M, r);
            const double e = (double)blocks / (256.0 * (double)((blocks + 255) / 256));
            if (e > beste + 0.02) { beste = e; best = sp; }
        }
        splits = best;
    }
    int rps = cdiv(cdiv(g.M, splits), rq) * rq;
    splits = cdiv(g.M, rps);
    g.rows_per_split = rps;
    g.tap_major = (!g.plain_q && g.T > 1 && g.Cq % BJ == 0 && env_int(ENV_WG_TAPMAJOR, 1)) ? 1 : 0;
    g.slab = splits == 1 ? out : ws;          // one split: no slab round trip, no reduce launch
    g.slab1 = splits == 1 ? out1 : ws + (long)splits * per;
    float* bws = ws + (long)splits * per * groups;              // bias partials behind the weight slabs
    g.bslab = db ? (splits == 1 ? db : bws) : nullptr;
    g.bslab1 = db ? (splits == 1 ? db1 : bws + (long)splits * perb) : nullptr;
    g.beta = splits == 1 ? accumulate : 0;
    g.wbytes = (4.0 * g.NI * g.ldw < 4294967000.0 && env_int(ENV_IG_BUFSTORE, 1)) ? (unsigned)(4.0 * g.NI * g.ldw) : 0;
    g.out = out; g.out1 = out1; g.bout = db; g.bout1 = db1; g.accumulate = accumulate;
    // In-launch reduction (PDF_WG_INLAUNCH=1) is OFF by default: measured 245 vs 395 img/s.  Every block's agent-scope release
    // fence (buffer_wbl2) writes back the whole XCD L2, which holds megabytes of slabs and of the main stream's fresh outputs;
    // write-through slab stores would need 16-byte stores from a re-laid-out accumulator (4-byte sc1 stores are ~6x slower).
    // BatchNorm's in-launch finalisation (norm.hip) publishes 32 bytes per block and does use the write-through form.
    g.counters = (splits > 1 && env_int(ENV_WG_INLAUNCH, 0)) ? pdf_ticket_counters((int)tiles * groups) : nullptr;
    // Optional (PDF_WG_ATOMIC = minimum split count; default: never): accumulating launches add their partial tiles with fp32
    // atomics instead of writing slabs -- no slab round trip, no reduce_slabs launch (-133 launches, -2.1 ms of side-stream work
    // per step).  Measured: fp32 B=32 420.4 / 421.3 without vs 417.9 / 420.5 with, RGB-only encoder B=8 483 vs 473, bf16 B=64
    // 862 vs 865 -- the reductions run on the side stream beside MFMA-bound kernels and cost nothing there, the atomic
    // read-modify-writes cost the weight-gradient kernels' epilogues more than plain stores.
    g.atomic = (splits >= env_int(ENV_WG_ATOMIC, 1 << 30) && splits > 1 && accumulate && g.counters == nullptr) ? 1 : 0;
    dim3 grid((unsigned)tiles, (unsigned)splits, (unsigned)groups);
    int brc = 0;
    const double wflops = 2.0 * groups * g.M * g.NI * NJ, wbytes = wgemm_bytes(g, groups);
    // operand extents for the buffer-descriptor kernels (from the group's base pointer; every element the kernel may address)
    const double pext = 4.0 * g.M * g.ldp, qext = 4.0 * (g.plain_q ? (double)g.M * g.ldq : (double)cdiv(g.M, g.QH * g.QW) * g.H * g.W * g.ldq);
    if (bf16 && g.q_scale != nullptr) return PDF_E_BADARG;
    if (bf16) {
        g_shadow_operands += (g.P16 != nullptr) + (g.Q16 != nullptr);
        const bool fits = pext < 4294967000.0 && qext < 4294967000.0;
        g.pbytes = fits ? (unsigned)pext : 0; g.qbytes = fits ? (unsigned)qext : 0;
        brc = launch_wgemm_bf16(g, splits, groups, small ? 1 : 0, s);
        if (brc < 0) return -brc;
    }
    if (brc != 1 && (g.P == nullptr || g.Q == nullptr)) return PDF_E_BADARG;      // a bf16-only operand and no bf16 launch
    if (g.q_scale != nullptr && (brc == 1 || !g.plain_q || groups != 1)) return PDF_E_BADARG;      // operand transform: fp32 buffer-load kernels only
    if (brc == 1) {
    } else if (small) {
        const bool bk32 = fast && env_int(ENV_WG_BK32, 1);
        const bool buf = bk32 && pext < 4294967000.0 && qext < 4294967000.0 && env_int(ENV_WG_BUF, 1);
        g.pbytes = buf ? (unsigned)pext : 0; g.qbytes = buf ? (unsigned)qext : 0;
        if (g.q_scale != nullptr && !buf) return PDF_E_BADARG;
        KTimer kt(buf && g.q_scale != nullptr ? "wgemm_tn64_aff" : buf ? "wgemm_tn<64, 64, 2, 2, true, 32, true>" : bk32 ? "wgemm_tn<64, 64, 2, 2, true, 32, false>" : fast ? "wgemm_tn<64, 64, 2, 2, true, 16, false>" : "wgemm_tn<64, 64, 2, 2, false, 16, false>", wflops, wbytes, s);
        if (buf && g.q_scale != nullptr) hipLaunchKernelGGL(wgemm_tn64_aff, grid, dim3(256), 0, s, g);
        else if (buf) hipLaunchKernelGGL((wgemm_tn<64, 64, 2, 2, true, 32, true>), grid, dim3(256), 0, s, g);
        else if (bk32) hipLaunchKernelGGL((wgemm_tn<64, 64, 2, 2, true, 32>), grid, dim3(256), 0, s, g);
        else if (fast) hipLaunchKernelGGL((wgemm_tn<64, 64, 2, 2, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((wgemm_tn<64, 64, 2, 2, false>), grid, dim3(256), 0, s, g);
    } else {
        const int dma = env_int(ENV_WG_DMA, 3);
        const int pad = env_int(ENV_WG_LDSPAD, 0) * 1024;
        const bool buf = fast && dma == 3 && pext < 4294967000.0 && qext < 4294967000.0 && env_int(ENV_WG_BUF, 1);
        g.pbytes = buf ? (unsigned)pext : 0; g.qbytes = buf ? (unsigned)qext : 0;
        g.uniform = (buf && (g.plain_q || (g.QW % 8 == 0 && g.Cq % 128 == 0)) && env_int(ENV_WG_UNIFORM, 2)) ? 1 : 0;
        if (g.uniform && env_int(ENV_WG_UNIFORM, 2) >= 2 && g.rows_per_split % 16 == 0 && g.M % 16 == 0 && (g.plain_q || g.QW % 16 == 0)) g.uniform = 2;
        if (g.q_scale != nullptr && g.uniform != 2) return PDF_E_BADARG;      // (the fragment-read transform needs whole K-steps)
        KTimer kt(buf && g.q_scale != nullptr ? "wgemm_tn_dma_aff" : buf ? "wgemm_tn_dma<3, true>" : fast && dma == 4 ? "wgemm_tn_dma<4, false>" : fast && dma == 3 ? "wgemm_tn_dma<3, false>" : fast ? "wgemm_tn<128, 128, 2, 2, true, 16>" : "wgemm_tn<128, 128, 2, 2, false, 16>",
                  wflops, wbytes, s);
        if (buf && g.q_scale != nullptr) hipLaunchKernelGGL(wgemm_tn_dma_aff, grid, dim3(256), pad, s, g);
        else if (buf) hipLaunchKernelGGL((wgemm_tn_dma<3, true>), grid, dim3(256), pad, s, g);
        else if (fast && dma == 4) hipLaunchKernelGGL((wgemm_tn_dma<4, false>), grid, dim3(256), pad, s, g);
        else if (fast && dma == 3) hipLaunchKernelGGL((wgemm_tn_dma<3, false>), grid, dim3(256), pad, s, g);
        else if (fast) hipLaunchKernelGGL((wgemm_tn<128, 128, 2, 2, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((wgemm_tn<128, 128, 2, 2, false>), grid, dim3(256), 0, s, g);
    }
    PDF_LAUNCH_CHECK();
    if (splits > 1 && g.counters == nullptr && !g.atomic) {
        Reduce r = {ws, out, out1, per, bws, db, db1, (int)perb, splits, accumulate};
        const bool two_d = splits >= 16 && per <= (1L << 20);
        KTimer kt(two_d ? "reduce_slabs_2d" : "reduce_slabs", 0.0, 4.0 * groups * (splits + 1 + (accumulate ? 1 : 0)) * (per + perb), s);
        if (two_d) {
            const int mb = (int)((per + 63) / 64);
            hipLaunchKernelGGL(reduce_slabs_2d, dim3(mb + (int)((perb + 63) / 64), groups), dim3(256), 0, s, r, mb);
        } else {
            const int mb = grid_for(per);
            hipLaunchKernelGGL(reduce_slabs, dim3(mb + (perb ? cdiv(perb, 256) : 0), groups), dim3(256), 0, s, r, mb);
        }
        PDF_LAUNCH_CHECK();
    }
    return 0;
}

// Workspace (floats) the weight-gradient entry points need at most for an [NI][NJ] gradient over M rows.
PDF_API long pdf_wgrad_workspace_floats(int M, int NI, int NJ) {
    const bool small = (NI <= 64 || NJ <= 64);
    const int B = small ? 64 : 128;
    long tiles = (long)cdiv(NI, B) * cdiv(NJ, B);
    int splits = (int)((1024 + tiles - 1) / tiles);
    int max_by_rows = cdiv(M, 128);
    if (splits > max_by_rows) splits = max_by_rows;
    if (splits < 1) splits = 1;
    return (long)splits * ((long)NI * NJ + NI);           // weight slabs + bias partials
}

// dW[N][K] (+)= dy[M][N]^T x[M][K]   (Linear / 1x1 weight gradient)
static int pdf_linear_bwd_weight_impl(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats,
                                  int M, int N, int K, int ldx, int lddy, int accumulate, hipStream_t s, PdfCallOpts& co) {
    const Shadows sh = {co.op0_bf16, co.op1_bf16};                       // op0: x, op1: dy
    WGemm g = {};
    g.P = dy; g.Q = x; g.M = M; g.NI = N; g.Cq = K; g.T = 1; g.ldp = lddy; g.ldq = ldx; g.ldw = K;
    g.plain_q = 1; g.H = 1; g.W = M; g.QH = 1; g.QW = M; g.sy = 1; g.sx = 1;
    g.dy[0] = 0; g.dx[0] = 0; g.wt[0] = 0;
    g.Q16 = sh.op0; g.P16 = sh.op1;
    g.q_scale = co.in_scale; g.q_shift = co.in_shift;
    return launch_wgemm(g, dw, ws, ws_floats, accumulate, s, nullptr, db);
}
PDF_API int pdf_linear_bwd_weight_x(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats,
                                  int M, int N, int K, int ldx, int lddy, int accumulate, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_linear_bwd_weight_impl(x, dy, dw, db, ws, ws_floats, M, N, K, ldx, lddy, accumulate, s, co); }
PDF_API int pdf_linear_bwd_weight(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats,
                                  int M, int N, int K, int ldx, int lddy, int accumulate, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_linear_bwd_weight_impl(x, dy, dw, db, ws, ws_floats, M, N, K, ldx, lddy, accumulate, s, co); pdf_tls_publish(co); return rc; }


// paired form of the above: rows [0, M) -> dw0, rows [M, 2M) -> dw1; ws >= 2 * pdf_wgrad_workspace_floats(M, N, K)
static int pdf_linear_bwd_weight_pair_impl(const float* x, const float* dy, float* dw0, float* dw1, float* db0, float* db1,
                                       float* ws, long ws_floats, int M, int N, int K, int ldx, int lddy, int accumulate, hipStream_t s, PdfCallOpts& co) {
    WGemm g = {};
    g.P = dy; g.Q = x; g.M = M; g.NI = N; g.Cq = K; g.T = 1; g.ldp = lddy; g.ldq = ldx; g.ldw = K;
    g.plain_q = 1; g.H = 1; g.W = M; g.QH = 1; g.QW = M; g.sy = 1; g.sx = 1;
    g.dy[0] = 0; g.dx[0] = 0; g.wt[0] = 0;
    g.gsP = (long)M * lddy; g.gsQ = (long)M * ldx;
    return launch_wgemm(g, dw0, ws, ws_floats, accumulate, s, dw1, db0, db1);
}
PDF_API int pdf_linear_bwd_weight_pair_x(const float* x, const float* dy, float* dw0, float* dw1, float* db0, float* db1,
                                       float* ws, long ws_floats, int M, int N, int K, int ldx, int lddy, int accumulate, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_linear_bwd_weight_pair_impl(x, dy, dw0, dw1, db0, db1, ws, ws_floats, M, N, K, ldx, lddy, accumulate, s, co); }
PDF_API int pdf_linear_bwd_weight_pair(const float* x, const float* dy, float* dw0, float* dw1, float* db0, float* db1,
                                       float* ws, long ws_floats, int M, int N, int K, int ldx, int lddy, int accumulate, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_linear_bwd_weight_pair_impl(x, dy, dw0, dw1, db0, db1, ws, ws_floats, M, N, K, ldx, lddy, accumulate, s, co); pdf_tls_publish(co); return rc; }


// dW[Cout][KH][KW][Cin] (+)= sum over output pixels dy[m][co] * x[pos(m,tap)][ci]

// Weight gradient of the ResNet stem (7x7, stride 2, pad 3, 3 -> 64 channels): dW[co][ky][kx][ci] = sum over output pixels of
// dy[p][co] * x[2 oy - 3 + ky][2 ox - 3 + kx][ci].  The generic kernel walks its 3-channel taps element by element (18 TFLOP/s,
// 0.54 ms at B = 32, and it is the LAST weight gradient of the step -- the tail after the main chain).  Here a block stages, per
// chunk of 64 output pixels of one row, the 7 x 133 x 3 input patch and the 64 x 64 gradient tile in LDS and runs the 64 x 147
// (padded to 160) contraction on the matrix pipe: A = dy as it lies ([pixel][co]), B = the im2col row gathered from the patch
// through a per-lane offset; each wave takes half the channels and half the pixels, the partial tiles are summed in LDS at the end, one
// partial matrix per block goes to the workspace (reduce_slabs_2d sums them in a fixed order).
__global__ __launch_bounds__(256) void stem7x7_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                            int N, int H, int W, int OH, int OW, int lddy, int cpb) {
    constexpr int CH = 64, XW = 2 * CH + 5, RW = XW * 3, XF = 7 * RW, ZB = 2800, DYB = 3200, NJ = 147, NJP = 160;
    __shared__ __attribute__((aligned(16))) float sm[64 * NJP];     // [0, XF): patch, [XF, DYB): zeros, [DYB, DYB + 4096): dy tile; at the end [64][160]
    float* Xs = sm;
    float* dYs = sm + DYB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = XF + tid; i < DYB; i += 256) sm[i] = 0.f;
    int boff[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int j = 32 * t + (lane & 31);
        const int tap = j / 3, ci = j - tap * 3, ky = tap / 7, kx = tap - ky * 7;
        boff[t] = j < NJ ? (ky * XW + kx) * 3 + ci : ZB;           // columns 147..159 read zeros (ZB + 6 * 63 < DYB)
    }
    // wave = (channel half a, pixel half kh): 32 channels x 160 columns over 32 of the chunk's 64 pixels -- 80 accumulator registers
    // (all 64 channels per wave, 160 registers, spilled once the prefetch registers were added)
    const int wa = wave & 1, kh = wave >> 1;
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int cpr = OW / CH;
    const long total = (long)N * OH * cpr;
    const long c0 = (long)blockIdx.x * cpb, c1 = min(total, c0 + cpb);
    // the next chunk's patch rows (contiguous in NHWC with 3 channels: 399 floats each) and dy tile travel through registers while
    // the current chunk is on the matrix pipe
    typedef float v4f __attribute__((ext_vector_type(4)));          // (an array of HIP's float4 struct lands in scratch)
    float xr[14]; v4f dr[4];
    auto gload = [&](long c) {
        const int n = (int)(c / ((long)OH * cpr));
        const int rem = (int)(c - (long)n * OH * cpr);
        const int oy = rem / cpr, ox0 = (rem - oy * cpr) * CH;
        const int lin0 = (2 * ox0 - 3) * 3;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int iy = 2 * oy - 3 + r;
            const float* row = x + ((long)(n * H + iy) * W) * 3;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = tid + 256 * h, lin = lin0 + e;
                const bool ok = iy >= 0 && iy < H && e < RW && lin >= 0 && lin < 3 * W;
                xr[r * 2 + h] = ok ? row[lin] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, px = e >> 4, c4 = (e & 15) * 4;
            dr[q] = *reinterpret_cast<const v4f*>(&dy[((long)(n * OH + oy) * OW + ox0 + px) * lddy + c4]);
        }
    };
    if (c0 < c1) gload(c0);
    for (long c = c0; c < c1; ++c) {
        __syncthreads();                                            // the previous chunk's fragments have been read
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h) { const int e = tid + 256 * h; if (e < RW) Xs[r * RW + e] = xr[r * 2 + h]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e = tid + 256 * q; *reinterpret_cast<v4f*>(&dYs[(e >> 4) * 64 + (e & 15) * 4]) = dr[q]; }
        __syncthreads();
        if (c + 1 < c1) gload(c + 1);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int px = 32 * kh + 2 * ks + (lane >> 5);
            const float a0 = dYs[px * 64 + 32 * wa + (lane & 31)];
            float b[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) b[t] = Xs[boff[t] + px * 6];
#pragma unroll
            for (int t = 0; t < 5; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[t], acc[t], 0, 0, 0);
        }
    }
    __syncthreads();
    for (int i = tid; i < 64 * NJP; i += 256) sm[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 32 * wa + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = 32 * t + (lane & 31);
            atomicAdd(&sm[co * NJP + j], acc[t][r]);
        }
    __syncthreads();
    for (int i = tid; i < 64 * NJ; i += 256) {
        const int co = i / NJ, j = i - co * NJ;
        part[(long)blockIdx.x * (64 * NJ) + i] = sm[co * NJP + j];
    }
}

static int pdf_conv2d_bwd_weight_impl(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats,
                                  int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                                  int stride, int pad, int OH, int OW, int lddy, int accumulate, hipStream_t s, PdfCallOpts& co) {
    const Shadows sh = {co.op0_bf16, co.op1_bf16};                       // op0: x, op1: dy
    if (KH * KW > MAX_TAPS) return PDF_E_BADARG;
    // Winograd F(4x4, 3x3) weight gradient (winograd.hip): fp32 mode, the caller handed a workspace (PdfCallOpts::ws), the layer qualifies
    if (!g_gemm_bf16 && co.ws != nullptr && dy != nullptr && OH == H && OW == W && ldx % 4 == 0 && lddy % 4 == 0 && aligned16(x) && aligned16(dy) &&
        aligned16(co.ws) && pdf_internal_wino_wgrad_eligible(N, H, W, Cin, Cout, KH, KW, stride, pad) &&
        co.ws_floats >= pdf_internal_wino_wgrad_workspace(N, H, W, Cin, Cout))
        return pdf_internal_conv3x3_winograd_wgrad(x, ldx, dy, lddy, dw, db, co.ws, N, H, W, Cin, Cout, accumulate,
                                                   (co.wino_v != nullptr && aligned16(co.wino_v) && pdf_internal_wino_v_offset(N, H, W, Cin, Cout) >= 0) ? co.wino_v : nullptr, s);
    // bf16 storage mode: dy exists only as bf16 (dy == NULL) -- the launch must be one the bf16 kernel takes with a shadow operand
    if (dy == nullptr && (sh.op1 == nullptr || db != nullptr || !g_gemm_bf16 || Cin % 16 != 0 || Cout % 16 != 0 || lddy % 8 != 0)) return PDF_E_BADARG;
    if (Cin == 3 && Cout == 3 && KH == 3 && KW == 3 && stride == 1 && db == nullptr && (long)N * OH * OW >= (1L << 16) && ws_floats >= 81L * 64) {
        const int nblk = (int)min((long)1024, ws_floats / 81);
        KTimer kt("tiny_conv_wgrad_kernel<3, 3, 3, 3> + reduce_slabs_2d", 2.0 * N * OH * OW * 81, 4.0 * N * (H * W + OH * OW) * 3, s);
        hipLaunchKernelGGL((tiny_conv_wgrad_kernel<3, 3, 3, 3>), dim3(nblk), dim3(256), 0, s, x, dy, ws, N, H, W, ldx, pad, OH, OW, lddy);
        Reduce r = {ws, dw, nullptr, 81, nullptr, nullptr, nullptr, 0, nblk, accumulate};
        hipLaunchKernelGGL(reduce_slabs_2d, dim3(2, 1), dim3(256), 0, s, r, 2);
        PDF_LAUNCH_CHECK();
        return 0;
    }
    if (Cin == 3 && ldx == 3 && Cout == 64 && KH == 7 && KW == 7 && stride == 2 && pad == 3 && db == nullptr && OW % 64 == 0 && OH * 2 == H && OW * 2 == W &&
        lddy % 4 == 0 && aligned16(dy) && ws_floats >= 64L * 147 * 64 && env_int(ENV_WG_STEM, 1)) {
        const long total = (long)N * OH * (OW / 64);
        int nblk = (int)min(min((long)512, ws_floats / (64 * 147)), total);
        const int cpb = (int)cdiv(total, nblk);
        nblk = (int)cdiv(total, cpb);
        KTimer kt("stem7x7_wgrad_kernel + reduce_slabs_2d", 2.0 * N * OH * OW * 64 * 147, 4.0 * N * ((double)H * W * 3 + (double)OH * OW * 64), s);
        hipLaunchKernelGGL(stem7x7_wgrad_kernel, dim3(nblk), dim3(256), 0, s, x, dy, ws, N, H, W, OH, OW, lddy, cpb);
        Reduce r = {ws, dw, nullptr, 64L * 147, nullptr, nullptr, nullptr, 0, nblk, accumulate};
        const int mb = (64 * 147 + 63) / 64;
        hipLaunchKernelGGL(reduce_slabs_2d, dim3(mb, 1), dim3(256), 0, s, r, mb);
        PDF_LAUNCH_CHECK();
        return 0;
    }
    if (Cout <= 4 && KH == 1 && KW == 1 && stride == 1 && pad == 0 && Cin % 4 == 0 && Cin <= 1024 && 256 % (Cin / 4) == 0 && ldx % 4 == 0 && aligned16(x)) {
        const long per = (long)Cout * Cin, perb = db ? Cout : 0;
        const int M = N * OH * OW;
        int nblk = (int)min((long)1024, ws_floats / (per + perb));
        if (nblk >= 1) {
            const int rpb = cdiv(M, nblk);
            nblk = cdiv(M, rpb);
            float* bws = ws + (long)nblk * per;
            KTimer kt("narrow_wgrad_kernel + reduce_slabs_2d", 2.0 * M * Cout * Cin, 4.0 * M * (Cout + Cin), s);
            hipLaunchKernelGGL(narrow_wgrad_kernel, dim3(nblk), dim3(256), 0, s, dy, x, ws, db ? bws : nullptr, M, Cout, Cin, lddy, ldx, rpb);
            Reduce r = {ws, dw, nullptr, per, bws, db, nullptr, (int)perb, nblk, accumulate};
            const int mb = (int)((per + 63) / 64);
            hipLaunchKernelGGL(reduce_slabs_2d, dim3(mb + (int)((perb + 63) / 64), 1), dim3(256), 0, s, r, mb);
            PDF_LAUNCH_CHECK();
            return 0;
        }
    }
    WGemm g = {};
    g.P = dy; g.Q = x; g.M = N * OH * OW; g.NI = Cout; g.Cq = Cin; g.T = KH * KW;
    g.ldp = lddy; g.ldq = ldx; g.ldw = KH * KW * Cin;
    g.H = H; g.W = W; g.QH = OH; g.QW = OW; g.sy = stride; g.sx = stride;
    g.plain_q = (KH == 1 && KW == 1 && stride == 1 && pad == 0) ? 1 : 0;
    for (int ky = 0; ky < KH; ++ky)
        for (int kx = 0; kx < KW; ++kx) {
            int t = ky * KW + kx;
            g.dy[t] = (int)(ky - pad); g.dx[t] = (int)(kx - pad); g.wt[t] = t;
        }
    g.Q16 = sh.op0; g.P16 = sh.op1;
    return launch_wgemm(g, dw, ws, ws_floats, accumulate, s, nullptr, db);
}
PDF_API int pdf_conv2d_bwd_weight_x(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats,
                                  int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                                  int stride, int pad, int OH, int OW, int lddy, int accumulate, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_conv2d_bwd_weight_impl(x, dy, dw, db, ws, ws_floats, N, H, W, Cin, ldx, Cout, KH, KW, stride, pad, OH, OW, lddy, accumulate, s, co); }
PDF_API int pdf_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats,
                                  int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                                  int stride, int pad, int OH, int OW, int lddy, int accumulate, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_conv2d_bwd_weight_impl(x, dy, dw, db, ws, ws_floats, N, H, W, Cin, ldx, Cout, KH, KW, stride, pad, OH, OW, lddy, accumulate, s, co); pdf_tls_publish(co); return rc; }


// ConvTranspose2d forward on NHWC: y[n, iy*s - pad + ky, ix*s - pad + kx, co] += x[n,iy,ix,ci] w[ci][co][ky][kx].
// w = the weight in its natural [Cin][KH][KW][Cout] storage, read as the [K = (tap, ci)][N = co] operand it is.
// kernel == stride (p4/p5, intaghand_encoder.py:604-605): ONE plain GEMM + pixel-shuffle epilogue;
// otherwise (p3: k4 s2 p1, :603) one launch per output-parity class.
static int pdf_deconv2d_fwd_impl(const float* x, const float* w, const float* bias, float* y,
                             int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                             int stride, int pad, int OH, int OW, int ldy, hipStream_t s, PdfCallOpts& co) {
    const Shadows sh = {co.op0_bf16, co.op1_bf16};                       // op0: x, op1: w
    if (KH * KW > MAX_TAPS) return PDF_E_BADARG;
    if (!g_gemm_bf16 && co.ws != nullptr && ldx == Cin && ldy == Cout) {  // x3 form (gemm_x3.hip): kernel == stride, long reduction, workspace handed in
        const long need = pdf_internal_x3_deconv_workspace(N, H, W, Cin, Cout, KH, KW, stride, pad, 0);
        if (need > 0 && co.ws_floats >= need && ((uintptr_t)co.ws & 15) == 0) {
            const CurCall cur(co);
            if (KH != stride || pad != 0) return pdf_internal_x3_deconv_general_fwd(x, w, bias, y, co.ws, N, H, W, Cin, Cout, KH, stride, pad, OH, OW, ldy, s);
            return pdf_internal_x3_deconv_fwd(x, w, bias, y, co.ws, N, H, W, Cin, Cout, KH, KW, stride, OH, OW, ldy, s);
        }
    }
    if (KH == stride && KW == stride && pad == 0) {
        IGemm g = {};
        g.A = x; g.B = w; g.C = y; g.bias = bias; g.A16 = sh.op0; g.B16 = sh.op1;
        g.M = N * H * W; g.N = KH * KW * Cout; g.K = Cin; g.Cin = Cin; g.lda = ldx; g.ldc = ldy;
        // columns n = (tap, co): w[ci][tap][co] is exactly a [K = ci][N] matrix
        g.ldb = KH * KW * Cout; g.b_kn = 1; g.btap = 0;
        g.T = 1; g.plain_in = 1; g.plain_out = 0; g.dy[0] = 0; g.dx[0] = 0; g.wt[0] = 0;
        g.H = H; g.W = W; g.QH = H; g.QW = W; g.sy = 1; g.sx = 1;
        g.OH = OH; g.OW = OW; g.osy = stride; g.osx = stride; g.ooy = 0; g.oox = 0;
        g.ps_cout = Cout; g.ps_kw = KW;
        return launch_igemm(g, s);
    }
    for (int py = 0; py < stride; ++py)
        for (int px = 0; px < stride; ++px) {
            IGemm g = {};
            g.A = x; g.B = w; g.C = y; g.bias = bias; g.A16 = sh.op0; g.B16 = sh.op1;
            g.N = Cout; g.Cin = Cin; g.lda = ldx; g.ldb = KH * KW * Cout; g.b_kn = 1; g.btap = Cout; g.ldc = ldy;
            g.H = H; g.W = W;
            g.QH = (OH - py + stride - 1) / stride; g.QW = (OW - px + stride - 1) / stride;
            if (g.QH <= 0 || g.QW <= 0) continue;
            g.M = N * g.QH * g.QW;
            g.sy = 1; g.sx = 1;
            int T = 0;
            for (int ky = 0; ky < KH; ++ky) {
                if ((py + pad - ky) % stride != 0) continue;
                for (int kx = 0; kx < KW; ++kx) {
                    if ((px + pad - kx) % stride != 0) continue;
                    int ny = py + pad - ky, nx = px + pad - kx;      // iy = qy + ny/stride
                    g.dy[T] = (int)(ny >= 0 ? ny / stride : -((-ny) / stride));
                    g.dx[T] = (int)(nx >= 0 ? nx / stride : -((-nx) / stride));
                    g.wt[T] = (int)(ky * KW + kx);
                    ++T;
                }
            }
            if (T == 0) return PDF_E_BADARG;      // would need a bias-only fill; not used by PDFNet
            g.T = T; g.K = T * Cin;
            g.plain_in = 0; g.plain_out = 0;
            g.OH = OH; g.OW = OW; g.osy = stride; g.osx = stride; g.ooy = py; g.oox = px;
            int rc = launch_igemm(g, s);
            if (rc) return rc;
        }
    return 0;
}
PDF_API int pdf_deconv2d_fwd_x(const float* x, const float* w, const float* bias, float* y,
                             int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                             int stride, int pad, int OH, int OW, int ldy, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_deconv2d_fwd_impl(x, w, bias, y, N, H, W, Cin, ldx, Cout, KH, KW, stride, pad, OH, OW, ldy, s, co); }
PDF_API int pdf_deconv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                             int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                             int stride, int pad, int OH, int OW, int ldy, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_deconv2d_fwd_impl(x, w, bias, y, N, H, W, Cin, ldx, Cout, KH, KW, stride, pad, OH, OW, ldy, s, co); pdf_tls_publish(co); return rc; }


// ConvTranspose2d backward-data: dx[n,iy,ix,ci] = sum dy[n, iy*s-pad+ky, ix*s-pad+kx, co] w[ci][ky][kx][co]
// -- a plain strided conv over dy with the weight in its natural [Cin][KH][KW][Cout] storage.
static int pdf_deconv2d_bwd_data_impl(const float* dy, const float* w, float* dx,
                                  int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW,
                                  int stride, int pad, int OH, int OW, int lddy, hipStream_t s, PdfCallOpts& co) {
    const Shadows sh = {co.op0_bf16, co.op1_bf16};                       // op0: dy, op1: w
    if (KH * KW > MAX_TAPS) return PDF_E_BADARG;
    if (!g_gemm_bf16 && co.ws != nullptr && lddy == Cout && lddx >= Cin) {
        const long need = pdf_internal_x3_deconv_workspace(N, H, W, Cin, Cout, KH, KW, stride, pad, 1);
        if (need > 0 && co.ws_floats >= need && ((uintptr_t)co.ws & 15) == 0) {
            const CurCall cur(co);
            if (KH != stride || pad != 0) return pdf_internal_x3_deconv_general_bwd_data(dy, w, dx, co.ws, N, H, W, Cin, lddx, Cout, KH, stride, pad, OH, OW, lddy, s);
            return pdf_internal_x3_deconv_bwd_data(dy, w, dx, co.ws, N, H, W, Cin, lddx, Cout, KH, KW, stride, OH, OW, lddy, s);
        }
    }
    IGemm g = {};
    g.A = dy; g.B = w; g.C = dx; g.bias = nullptr; g.A16 = sh.op0; g.B16 = sh.op1;
    g.M = N * H * W; g.N = Cin; g.K = KH * KW * Cout; g.Cin = Cout; g.lda = lddy; g.ldb = KH * KW * Cout; g.ldc = lddx;
    g.H = OH; g.W = OW; g.QH = H; g.QW = W; g.sy = stride; g.sx = stride;
    conv_taps(g, KH, KW, pad, 1);
    g.plain_in = 0; g.plain_out = 1;
    return launch_igemm(g, s);
}
PDF_API int pdf_deconv2d_bwd_data_x(const float* dy, const float* w, float* dx,
                                  int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW,
                                  int stride, int pad, int OH, int OW, int lddy, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_deconv2d_bwd_data_impl(dy, w, dx, N, H, W, Cin, lddx, Cout, KH, KW, stride, pad, OH, OW, lddy, s, co); }
PDF_API int pdf_deconv2d_bwd_data(const float* dy, const float* w, float* dx,
                                  int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW,
                                  int stride, int pad, int OH, int OW, int lddy, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_deconv2d_bwd_data_impl(dy, w, dx, N, H, W, Cin, lddx, Cout, KH, KW, stride, pad, OH, OW, lddy, s, co); pdf_tls_publish(co); return rc; }


// ConvTranspose2d weight gradient in the natural [Cin][KH][KW][Cout] storage.
static int pdf_deconv2d_bwd_weight_impl(const float* x, const float* dy, float* dw, float* ws, long ws_floats,
                                    int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                                    int stride, int pad, int OH, int OW, int lddy, int accumulate, hipStream_t s, PdfCallOpts& co) {
    const Shadows sh = {co.op0_bf16, co.op1_bf16};                       // op0: x, op1: dy
    if (KH * KW > MAX_TAPS) return PDF_E_BADARG;
    if (!g_gemm_bf16 && co.ws != nullptr && ldx == Cin && lddy == Cout) {  // x3 form (gemm_x3.hip), workspace in PdfCallOpts::ws
        const long need = pdf_internal_x3_deconv_workspace(N, H, W, Cin, Cout, KH, KW, stride, pad, 2);
        if (need > 0 && co.ws_floats >= need && ((uintptr_t)co.ws & 15) == 0) {
            const CurCall cur(co);
            if (KH != stride || pad != 0) return pdf_internal_x3_deconv_general_bwd_weight(x, dy, dw, co.ws, N, H, W, Cin, Cout, KH, stride, pad, OH, OW, lddy, accumulate, s);
            return pdf_internal_x3_deconv_bwd_weight(x, dy, dw, co.ws, N, H, W, Cin, Cout, KH, KW, stride, OH, OW, lddy, accumulate, s);
        }
    }
    WGemm g = {};
    g.P = x; g.Q = dy; g.P16 = sh.op0; g.Q16 = sh.op1; g.M = N * H * W; g.NI = Cin; g.Cq = Cout; g.T = KH * KW;
    g.ldp = ldx; g.ldq = lddy; g.ldw = KH * KW * Cout;
    g.H = OH; g.W = OW; g.QH = H; g.QW = W; g.sy = stride; g.sx = stride; g.plain_q = 0;
    for (int ky = 0; ky < KH; ++ky)
        for (int kx = 0; kx < KW; ++kx) {
            int t = ky * KW + kx;
            g.dy[t] = (int)(ky - pad); g.dx[t] = (int)(kx - pad); g.wt[t] = t;
        }
    return launch_wgemm(g, dw, ws, ws_floats, accumulate, s);
}
PDF_API int pdf_deconv2d_bwd_weight_x(const float* x, const float* dy, float* dw, float* ws, long ws_floats,
                                    int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                                    int stride, int pad, int OH, int OW, int lddy, int accumulate, hipStream_t s, PdfCallOpts* opts) { PdfCallOpts z = {}; PdfCallOpts& co = opts ? *opts : z; co.stats_tiles = co.stats_rows = 0; return pdf_deconv2d_bwd_weight_impl(x, dy, dw, ws, ws_floats, N, H, W, Cin, ldx, Cout, KH, KW, stride, pad, OH, OW, lddy, accumulate, s, co); }
PDF_API int pdf_deconv2d_bwd_weight(const float* x, const float* dy, float* dw, float* ws, long ws_floats,
                                    int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                                    int stride, int pad, int OH, int OW, int lddy, int accumulate, hipStream_t s) { PdfCallOpts co = pdf_tls_take_all(); const int rc = pdf_deconv2d_bwd_weight_impl(x, dy, dw, ws, ws_floats, N, H, W, Cin, ldx, Cout, KH, KW, stride, pad, OH, OW, lddy, accumulate, s, co); pdf_tls_publish(co); return rc; }

