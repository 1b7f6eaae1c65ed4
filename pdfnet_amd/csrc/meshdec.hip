// Fused mesh decoder for gfx950 (round 5): one DualGraphLayer of the IntagHand-style dual-hand GCN / attention decoder
// (lib/models/networks/model_attn/DualGraph.py:62-92, gcn.py:34-69,99-110, self_attn.py:17-85, inter_attn.py:73-125) as
// THREE launches forward and ELEVEN backward (+ its 28 weight-gradient GEMMs on the side stream) instead of ~330 dependent 5-20 us launches:
//
//   forward   mesh_gcn_kernel        x -> the GCN_ResBlocks (LayerNorm, Chebyshev ELL product, linear, dropout, residual)
//             mesh_att_kernel<self>  SelfAttn (LN, q/k/v, attention, fc, residual, MLP block) -> LN + q/k/v of the cross-hand step
//             mesh_att_kernel<cross> attention over the keys / values of the OTHER hand of the same sample (the only dependence between
//                                    workgroups, hence the launch boundary) -> fc -> MLP block -> level output
//   backward  mesh_att_bwd1 / bwd2 per attention (the MLP block and fc back to the attention's output; then the attention itself in
//             key chunks -- the dq pass and the dk / dv pass in separate workgroups, their (head, tile) items shared by 2-4 workgroups each:
//             the only part of the chain with that much independent work, a quarter of the backward at level 2 -- then q/k/v linears and
//             LayerNorm in a second launch), mesh_gcn_bwd_kernel once per GCN block (one accumulator set at a time: the
//             register file, not the launch count, bounds these kernels); dY operands of every linear go to a gradient tape and the
//             weight gradients stay ordinary full-chip GEMM launches (pdf_linear_bwd_weight_pair) issued from here on the side stream.
//   (One launch per direction was the first form: it spilled -- 64-bit dropout index chains, tape offsets in scratch, three accumulator
//   sets live -- and ran slower than the split form; the splits are at points where nothing is live in registers.)
//
// One 512-thread workgroup per (hand, sample) -- eight waves, two per SIMD, two accumulator tiles each (round 6: the chain is latency-bound, and a second
// wave per SIMD hides what one cannot: every kernel 1.05-1.6x faster than with four waves of four tiles, forward 2.28 -> 1.80 ms at B = 32, the step
// +1.8 %; profiles/r06_mesh_w8.txt; -DMD_WAVES=4 builds the old form) -- keeps that hand's [V][C] features of the sample on ONE CU for a whole kernel:
// V x C = 63 x 256 = 126 x 128 = 252 x 64 = 16,128 floats at every level, so two [V][C + 4] staging buffers (<= 139 KB) fit the
// 160 KB LDS.  Every matrix product is "activation-stationary": the A operand (this hand's rows) is read from LDS by
// ds_read_b128 -- lane l takes row l & 31 and the four k of chunk l >> 5, which feeds four v_mfma_f32_32x32x2_f32 (fp32 in, exact
// fmaf chains) -- and the weights stream L2 -> registers once per workgroup, each lane eight consecutive floats of its output column
// per step (lanes l and l + 32 consume one 64-byte line), prefetched four K-steps ahead through a register ring in a fully unrolled
// loop fenced with sched_barrier (the compiler otherwise sinks the loads next to their use).  No barrier inside a K loop; LayerNorm,
// the Chebyshev ELL product (ELL rows staged in LDS), softmax, dropout and the residuals run on the LDS image between the products.
// Attention runs on the matrix pipe too: S^T = K Q^T per (head, 32-query tile) stays in accumulator registers (keys along rows), the
// softmax over keys is a register reduction plus ONE cross-half shuffle, and P goes straight back into the MFMA as the B operand of
// O^T = V^T P.  Dropout masks are a stateless hash of (seed, step, element index) -- the same function the per-op kernels use, so the
// fused and the per-op paths draw identical masks.
// Everything the backward (and the weight-gradient GEMMs) needs is written to a "tape" in HBM as the chain passes; in eval mode
// nothing but the level output and q / k / v is written.
#include "common.h"
#ifndef MD_BF16
#define MD_BF16 0
#endif
// MD_X3 = 1 (csrc/meshdec_x3.hip, a third compilation, with MD_BF16 = 1): the linear products as x3 arithmetic -- both operands split into three bf16
// values as they are fed (a = h + m + l exactly) and six bf16 MFMAs per product instead of one: fp32-grade results (gemm_x3.hip) in
// 6/16 of the fp32 MFMA's matrix-pipe time.  Round 6: with two waves per SIMD the level-0 products run at 80-90 % of the CU's fp32 MFMA rate
// (profiles/r06_mesh_stamps.txt), so the pipe is what bounds them.  Entry points pdf_mesh_level_fwd_x3 / _bwd_x3.
#ifndef MD_X3
#define MD_X3 0
#endif
#if MD_X3
namespace md_x3_build {
#elif MD_BF16
namespace md_bf16_build {                      // the second compilation of this file: its kernels need names of their own
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MD_WAVES waves per workgroup: 8 (two per SIMD, two accumulator tiles each) or 4 (the form of round 5: four tiles each).  Sixteen -- one tile each,
// 128 registers -- was measured too: the attention kernels spill hundreds of registers and run 0.4-0.8x, the rest gains less than with eight
// (profiles/r06_mesh_waves.txt).
#if defined(MD_WAVES) && MD_WAVES != 4 && MD_WAVES != 8
#error "meshdec.hip: MD_WAVES must be 4 or 8"
#endif
#ifndef MD_WAVES
#define MD_WAVES 8
#endif
#define MD_THREADS (64 * MD_WAVES)
// MD_BF16 = 1 (csrc/meshdec_bf16.hip compiles this file a second time): the LINEAR products run on the bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32
// accumulate) -- a lane holds 8 consecutive k of its activation row and of its weight row anyway, which is exactly one bf16 MFMA's operand pair
// instead of eight fp32 ones; operands are rounded (RNE) as they are fed, like the library's bf16 GEMM mode does.  Attention, LayerNorm, the graph
// product and all accumulation stay fp32.  Entry points pdf_mesh_level_fwd_bf16 / _bwd_bf16; everything else of this file is unchanged.
#ifndef MD_BF16
#define MD_BF16 0
#endif
#if MD_BF16
typedef __bf16 md_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ md_bf16x8 md_pack8(float a, float b, float c, float d, float e, float f, float g_, float h) {
    union { unsigned u[4]; md_bf16x8 v; } x;
    x.u[0] = pdf_pk_bf16(a, b); x.u[1] = pdf_pk_bf16(c, d); x.u[2] = pdf_pk_bf16(e, f); x.u[3] = pdf_pk_bf16(g_, h);
    return x.v;
}
#if MD_X3
// eight floats -> their three bf16 components (o[0] + o[1] + o[2] == the floats exactly), packed like md_pack8
struct md_x3x8 { md_bf16x8 c[3]; };
__device__ __forceinline__ md_x3x8 md_split8(float a, float b, float c, float d, float e, float f, float g_, float h) {
    union { unsigned u[4]; md_bf16x8 v; } x0, x1, x2;
    pdf_x3_split2(a, b, x0.u[0], x1.u[0], x2.u[0]); pdf_x3_split2(c, d, x0.u[1], x1.u[1], x2.u[1]);
    pdf_x3_split2(e, f, x0.u[2], x1.u[2], x2.u[2]); pdf_x3_split2(g_, h, x0.u[3], x1.u[3], x2.u[3]);
    md_x3x8 o;
    o.c[0] = x0.v; o.c[1] = x1.v; o.c[2] = x2.v;
    return o;
}
// acc += a . b as six bf16 MFMAs, smallest products first (l h', h l', m m', m h', h m', h h')
__device__ __forceinline__ f32x16 md_mma_x3(const md_x3x8& a, const md_x3x8& b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.c[2], b.c[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.c[0], b.c[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.c[1], b.c[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.c[1], b.c[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.c[0], b.c[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.c[0], b.c[0], acc, 0, 0, 0);
    return acc;
}
#endif
#endif
#ifndef MD_WHATIF
#define MD_WHATIF 0
#endif
// diagnostic build (-DMD_STAMPS=1): workgroup 0's thread 0 records the shader clock at stage boundaries (pdf_debug_mesh_stamps reads them back)
#ifndef MD_STAMPS
#define MD_STAMPS 0
#endif
#if MD_STAMPS
__device__ unsigned long long g_md_stamps[8 * 3 * 64];
#define MD_STAMP(KID, N) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_md_stamps[((KID) * 3 + LV) * 64 + (N)] = __builtin_readcyclecounter(); } while (0)
#else
#define MD_STAMP(KID, N) do { } while (0)
#endif

// ---- argument block of one level (mirrors PdfMeshLevel of include/pdfnet_hip.h field for field) --------------------------------
struct MdLin { const float* w[2]; const float* b[2]; };            // [hand]
struct MdLN { const float* g[2]; const float* b[2]; };
struct MdGcn { MdLin fc1, fc2, sc; MdLN n2, n3; unsigned long long seed; };
struct MdAttn {
    MdLN ln; MdLin q, k, v, fc; MdLN ffln; MdLin f1, f2;
    unsigned long long seed_att, seed_z, seed_t, seed_x;
};
// gradient outputs (accumulated, += ): same shapes as the parameters
struct MdLinG { float* w[2]; float* b[2]; };
struct MdLNG { float* g[2]; float* b[2]; };
struct MdGcnG { MdLinG fc1, fc2, sc; MdLNG n2, n3; };
struct MdAttnG { MdLNG ln; MdLinG q, k, v, fc; MdLNG ffln; MdLinG f1, f2; };

struct PdfMeshLevel {
    int level, B, training, cin0;                 // level 0..2 (V = 63 << level, C = 256 >> level); cin0 = width of x (= 2 C)
    float p;                                      // dropout probability (0 in eval mode)
    const unsigned long long* step;               // device step counter mixed into every dropout seed (hipGraph replays draw fresh masks)
    const float* x;                               // [2][B][V][cin0] level input (position embedding already added)
    float* out;                                   // [2][B][V][C] level output
    const int* ell_col[2]; const float* ell_val[2]; const int* ell_colT[2]; const float* ell_valT[2]; int ell_w;
    MdGcn gcn[4];
    MdAttn self_, cross;
    float* tape;                                  // training: pdf_mesh_tape_floats(level, B) floats (forward writes, backward reads)
    float* qkv;                                   // [3][2][B][V][C] cross-hand q / k / v (part 1 -> part 2; also part of the tape)
    // backward
    const float* dout;                            // [2][B][V][C]
    float* dx;                                    // [2][B][V][cin0]
    float* gtape;                                 // pdf_mesh_gtape_floats(level, B) floats: the dY operands of the weight-gradient GEMMs
    MdGcnG ggcn[4]; MdAttnG gself, gcross;
    float* wg_ws; long wg_ws_floats;              // workspace of the weight-gradient launches
};

template <int LV> struct Cfg {
    static constexpr int V = 63 << LV, VP = 64 << LV, C = 256 >> LV, LD = C + 4, DH = C / 4, H = 4;
    static constexpr int MT = VP / 32, NT = C / 32;
#if MD_WAVES == 8
    static constexpr int WMT = 2, WNT = 1;                                     // accumulator tiles per wave: always 2
#else
    static constexpr int WMT = LV == 0 ? 2 : 4, WNT = LV == 0 ? 2 : 1;         // accumulator tiles per wave: always 4
#endif
    static constexpr int BUF = VP * LD;                                        // floats per LDS staging buffer
};

// ---- tape layout (floats, per level; rows = 2 B V) ------------------------------------------------------------------------------
// GCN block i: x_i [R][cin_i] (i = 0: the level input itself, not copied), cat1 [R][2 cin_i], y [R][C], st2 [R][2], cat2 [R][2C], z [R][C], st3 [R][2]
// then out_i [R][C] = x_{i+1}.  Attention block a (0 = self, 1 = cross): h [R][C], st [R][2], q, k, v [R][C] (cross: in `qkv`), a [R][C],
// stat [R][H][2], z [R][C], stz [R][2], hn [R][C], t [R][C], xo [R][C] (cross: the level output itself)
struct TapeOff {
    long R, C;
    // GCN block i occupies R (2 cin + 5 C + 4) floats: [cat1 | y | st2 | cat2 | z | st3 | out]  (offsets by arithmetic: an indexed table of
    // offsets would live in scratch memory once the block loop is not unrolled)
    __host__ __device__ long gcn_base(int i) const { return i == 0 ? 0 : R * (9 * C + 4) + (long)(i - 1) * R * (7 * C + 4); }
    __host__ __device__ long cat1(int i) const { return gcn_base(i); }
    __host__ __device__ long y(int i) const { return gcn_base(i) + R * 2 * (i == 0 ? 2 * C : C); }
    __host__ __device__ long st2(int i) const { return y(i) + R * C; }
    __host__ __device__ long cat2(int i) const { return st2(i) + R * 2; }
    __host__ __device__ long z(int i) const { return cat2(i) + R * 2 * C; }
    __host__ __device__ long st3(int i) const { return z(i) + R * C; }
    __host__ __device__ long out(int i) const { return st3(i) + R * 2; }
    // attention block a (0 = self, 1 = cross) occupies R (10 C + 12) floats: [h | st | q | k | v | a | stat | z | stz | hn | t | xo]
    __host__ __device__ long att_base(int a) const { return gcn_base(4) + (long)a * R * (10 * C + 12); }
    __host__ __device__ long h(int a) const { return att_base(a); }
    __host__ __device__ long st(int a) const { return h(a) + R * C; }
    __host__ __device__ long q(int a) const { return st(a) + R * 2; }            // (a == 1: unused, the cross-hand q / k / v live in `qkv`)
    __host__ __device__ long k(int a) const { return q(a) + R * C; }
    __host__ __device__ long v(int a) const { return k(a) + R * C; }
    __host__ __device__ long a_(int a) const { return v(a) + R * C; }
    __host__ __device__ long stat(int a) const { return a_(a) + R * C; }
    __host__ __device__ long az(int a) const { return stat(a) + R * 8; }
    __host__ __device__ long stz(int a) const { return az(a) + R * C; }
    __host__ __device__ long hn(int a) const { return stz(a) + R * 2; }
    __host__ __device__ long t(int a) const { return hn(a) + R * C; }
    __host__ __device__ long xo(int a) const { return t(a) + R * C; }            // (a == 1: unused, the level output is `out`)
    __host__ __device__ long total() const { return att_base(2); }
};
__host__ __device__ inline TapeOff tape_offsets(int level, int B) {
    TapeOff o;
    o.C = 256L >> level;
    o.R = 2L * B * (63L << level);
    return o;
}
// gradient tape: the dY operands of the weight-gradient GEMMs
struct GTapeOff {
    long RC;
    __host__ __device__ long dz(int i) const { return (3L * i) * RC; }
    __host__ __device__ long dy2(int i) const { return (3L * i + 1) * RC; }
    __host__ __device__ long dy(int i) const { return (3L * i + 2) * RC; }
    // attention block a, slot k: 0 du, 1 dt_pre, 2 do, 3 dq, 4 dk, 5 dv, 6 dc (gradient of the attention output), 7 dxr (gradient reaching the
    // residual stream at z), 8 dxin (gradient of the block's input: what the kernel in front of it starts from)
    __host__ __device__ long att(int a, int k) const { return (12L + 9L * a + k) * RC; }
    __host__ __device__ long scr() const { return 30L * RC; }                               // [R][C] scratch of GCN block 0's backward
    __host__ __device__ long total() const { return 31L * RC; }
};
__host__ __device__ inline GTapeOff gtape_offsets(int level, int B) {
    GTapeOff o;
    o.RC = 2L * B * (63L << level) * (256L >> level);
    return o;
}
#if !MD_BF16
PDF_API long pdf_mesh_tape_floats(int level, int B) { return level < 0 || level > 2 || B < 1 ? 0 : tape_offsets(level, B).total(); }
PDF_API long pdf_mesh_gtape_floats(int level, int B) { return level < 0 || level > 2 || B < 1 ? 0 : gtape_offsets(level, B).total(); }
#endif

// ---- wave / lane geometry ---------------------------------------------------------------------------------------------------------
template <int LV> struct Geo {
    using G = Cfg<LV>;
    int lane, wave, half, l31;
    int mt0, nt0;                                 // first M-tile / N-tile of this wave
    __device__ Geo() {
        lane = threadIdx.x & 63; wave = threadIdx.x >> 6; half = lane >> 5; l31 = lane & 31;
#if MD_WAVES == 8
        mt0 = LV == 0 ? 0 : LV == 1 ? 2 * (wave >> 2) : 2 * (wave >> 1);
        nt0 = LV == 0 ? wave : LV == 1 ? (wave & 3) : (wave & 1);
#else
        mt0 = LV == 2 ? 4 * (wave >> 1) : 0;
        nt0 = LV == 0 ? 2 * wave : LV == 1 ? wave : (wave & 1);
#endif
    }
};

// waves of a workgroup that take (head, tile) items of an attention pass when `parts` workgroups share `items` of them (level 0: 8 items over 2
// workgroups -- four waves each, the other four only help with the staging)
__host__ __device__ inline int md_active_waves(int items, int parts) { const int w = items / (parts < 1 ? 1 : parts); return w < 1 ? 1 : (w < MD_WAVES ? w : MD_WAVES); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// acc[i][j] += A[rows of M-tile mt0 + i][0 .. C) . W[columns of N-tile nt0 + j][koff .. koff + C)^T          (C = the level's width)
// A: LDS image [VP][LD]; W: global, row n at W + n * ldw (nn.Linear storage [out][in]).  CHEBY: two A images (x, L x) against the
// interleaved columns of a graph_conv_cheby weight (feature index fin * 2 + k, gcn.py:61-63): W[n][koff2 + 2 kk + {0, 1}].
// A K-step covers 16 k of W's row (plain) / 8 k of each image (CHEBY): lane l fetches the 8 consecutive weights [16 s + 8 (l >> 5), + 8) of its
// row as two float4, so lanes l and l + 32 consume one whole 64-byte line per step.
// The K loop is fully unrolled and every step is fenced with sched_barrier: the weights of step s + PD are issued at the top of step s and stay
// there.  Left to itself hipcc sinks each load to a few MFMAs before its use (and, in a rolled loop, re-materialises the "prefetched" value at
// the top of the iteration that consumes it: vmcnt(15) ... vmcnt(0) in front of the MFMAs) -- the L2 latency then shows in every step
// (round 5: 82 % of the MFMA rate for this product, 50 % for the backward one).
template <int LV, bool CHEBY>
__device__ __forceinline__ void gemm_nt(f32x16 (&acc)[Cfg<LV>::WMT][Cfg<LV>::WNT], const Geo<LV>& g, const float* __restrict__ A0,
                                        const float* __restrict__ A1, const float* __restrict__ W, int ldw, int koff) {
    using G = Cfg<LV>;
    constexpr int WMT = G::WMT, WNT = G::WNT, NS = (MD_WHATIF == 3 ? 64 : G::C) / (CHEBY ? 8 : 16), PD = NS < 4 ? NS : 4;
    const float* wp[WNT];
#pragma unroll
    for (int j = 0; j < WNT; ++j) wp[j] = W + (long)((g.nt0 + j) * 32 + (MD_WHATIF == 5 ? 0 : g.l31)) * ldw + koff + 8 * g.half;
    // A columns of this lane in step s: plain 16 s + 8 half + {0..7}; CHEBY 8 s + 4 half + {0..3} of each image
    const float* ap0 = A0 + ((g.mt0 * 32 + g.l31) * G::LD + (CHEBY ? 4 : 8) * g.half);
    const float* ap1 = CHEBY ? A1 + ((g.mt0 * 32 + g.l31) * G::LD + 4 * g.half) : nullptr;
    f32x4 bq[PD][WNT][2];
#pragma unroll
    for (int u = 0; u < PD; ++u)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) bq[u][j][q] = ld4(wp[j] + 16 * u + 4 * q);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        f32x4 b[WNT][2];
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) b[j][q] = bq[s % PD][j][q];
        if (s + PD < NS) {
#pragma unroll
            for (int j = 0; j < WNT; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) bq[s % PD][j][q] = ld4(wp[j] + 16 * (s + PD) + 4 * q);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 a0[WMT], a1[WMT];
#pragma unroll
        for (int i = 0; i < WMT; ++i) {
            if constexpr (CHEBY) {
                a0[i] = ld4(ap0 + i * 32 * G::LD + 8 * s);
                a1[i] = ld4(ap1 + i * 32 * G::LD + 8 * s);
            } else {
                a0[i] = ld4(ap0 + i * 32 * G::LD + 16 * s);
                a1[i] = ld4(ap0 + i * 32 * G::LD + 16 * s + 4);
            }
        }
#if MD_BF16
        {
#if MD_X3
#define MD_PK8 md_split8
            md_x3x8 a8[WMT], b8[WNT];
#else
#define MD_PK8 md_pack8
            md_bf16x8 a8[WMT], b8[WNT];
#endif
#pragma unroll
            for (int i = 0; i < WMT; ++i)                        // CHEBY: (x, L x) pairs interleaved like the weight row's (even, odd) columns
                a8[i] = CHEBY ? MD_PK8(a0[i][0], a1[i][0], a0[i][1], a1[i][1], a0[i][2], a1[i][2], a0[i][3], a1[i][3])
                              : MD_PK8(a0[i][0], a0[i][1], a0[i][2], a0[i][3], a1[i][0], a1[i][1], a1[i][2], a1[i][3]);
#pragma unroll
            for (int j = 0; j < WNT; ++j) b8[j] = MD_PK8(b[j][0][0], b[j][0][1], b[j][0][2], b[j][0][3], b[j][1][0], b[j][1][1], b[j][1][2], b[j][1][3]);
#pragma unroll
            for (int i = 0; i < WMT; ++i)
#pragma unroll
                for (int j = 0; j < WNT; ++j) {
#if MD_X3
                    acc[i][j] = md_mma_x3(a8[i], b8[j], acc[i][j]);
#else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
#endif
                }
        }
        if constexpr (false) {
#else
        if constexpr (!CHEBY) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < WMT; ++i)
#pragma unroll
                    for (int j = 0; j < WNT; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][e], b[j][0][e], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][e], b[j][1][e], acc[i][j], 0, 0, 0);
                    }
#endif
        } else if constexpr (!MD_BF16) {
            // the lane's 8 consecutive weights = (even, odd) pairs of its 4 k: w[2e] multiplies x, w[2e + 1] multiplies L x
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int i = 0; i < WMT; ++i)
#pragma unroll
                    for (int j = 0; j < WNT; ++j) {
                        const float we = e < 2 ? b[j][0][2 * e] : b[j][1][2 * e - 4];
                        const float wo = e < 2 ? b[j][0][2 * e + 1] : b[j][1][2 * e - 3];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][e], we, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][e], wo, acc[i][j], 0, 0, 0);
                    }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// acc[i][j] += A[rows][0 .. C) . W[0 .. C)[columns of N-tile]   with W stored [n][ldw] and the OUTPUT column index running along W's rows'
// elements: out[m][c] = sum_n A[m][n] W[n][c0 + cs * c]   (backward-data of a Linear: dx = dy W; cs = 2 picks the even / odd interleaved columns of
// a graph_conv_cheby weight).  A step covers 8 rows of W: lane l fetches rows 8 s + 4 (l >> 5) + {0..3} at its column -- four dword loads whose
// 32 lanes read 128 contiguous (cs = 1) bytes.  Unrolled and fenced like gemm_nt.
template <int LV>
__device__ __forceinline__ void gemm_nn(f32x16 (&acc)[Cfg<LV>::WMT][Cfg<LV>::WNT], const Geo<LV>& g, const float* __restrict__ A0,
                                        const float* __restrict__ W, int ldw, int c0, int cs) {
    using G = Cfg<LV>;
    constexpr int WMT = G::WMT, WNT = G::WNT, NS = G::C / 8, PD = 4;
    // one running pointer per (column tile, row of the lane's four): advanced by 8 rows per prefetched step (offsets recomputed per step from a
    // run-time ldw cost the unrolled loop hundreds of address registers)
    const float* wr[WNT][4];
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) wr[j][e] = W + (long)(4 * g.half + e) * ldw + c0 + cs * ((g.nt0 + j) * 32 + g.l31);
    const float* ap0 = A0 + ((g.mt0 * 32 + g.l31) * G::LD + 4 * g.half);
    const long ldw8 = 8L * ldw;
    float bq[PD][WNT][4];
#pragma unroll
    for (int u = 0; u < PD; ++u)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { bq[u][j][e] = *wr[j][e]; wr[j][e] += ldw8; }
    __builtin_amdgcn_sched_barrier(0);
    // NS <= 16 steps: one straight line.  NS = 32 (level 0): two passes of 16 -- a kernel with five 32-step products in one basic block made the
    // register allocator spill hundreds of registers; the one drain at the pass boundary costs ~1 k cycles of 16 k
    constexpr int UNR = NS < 16 ? NS : 16;
#if MD_BF16
    f32x4 ae[WMT];
    float be[WNT][4];
#endif
#pragma unroll 1
    for (int s0 = 0; s0 < NS; s0 += UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float b[WNT][4];
#pragma unroll
            for (int j = 0; j < WNT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) b[j][e] = bq[u % PD][j][e];
            if (u + PD < UNR || s0 + UNR < NS) {                        // (the last PD steps of the LAST pass prefetch nothing)
#pragma unroll
                for (int j = 0; j < WNT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bq[u % PD][j][e] = *wr[j][e]; wr[j][e] += ldw8; }
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 a0[WMT];
#pragma unroll
            for (int i = 0; i < WMT; ++i) a0[i] = ld4(ap0 + i * 32 * G::LD + 8 * (s0 + u));
#if MD_BF16
            // two steps (2 x 4 k per half) make one bf16 MFMA: the even step's operands wait in registers for the odd step's
            if ((u & 1) == 0) {
#pragma unroll
                for (int i = 0; i < WMT; ++i) ae[i] = a0[i];
#pragma unroll
                for (int j = 0; j < WNT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) be[j][e] = b[j][e];
            } else {
#if MD_X3
                md_x3x8 a8[WMT], b8[WNT];
#pragma unroll
                for (int i = 0; i < WMT; ++i) a8[i] = md_split8(ae[i][0], ae[i][1], ae[i][2], ae[i][3], a0[i][0], a0[i][1], a0[i][2], a0[i][3]);
#pragma unroll
                for (int j = 0; j < WNT; ++j) b8[j] = md_split8(be[j][0], be[j][1], be[j][2], be[j][3], b[j][0], b[j][1], b[j][2], b[j][3]);
#pragma unroll
                for (int i = 0; i < WMT; ++i)
#pragma unroll
                    for (int j = 0; j < WNT; ++j) acc[i][j] = md_mma_x3(a8[i], b8[j], acc[i][j]);
#else
#pragma unroll
                for (int i = 0; i < WMT; ++i)
#pragma unroll
                    for (int j = 0; j < WNT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(md_pack8(ae[i][0], ae[i][1], ae[i][2], ae[i][3], a0[i][0], a0[i][1], a0[i][2], a0[i][3]),
                                                                            md_pack8(be[j][0], be[j][1], be[j][2], be[j][3], b[j][0], b[j][1], b[j][2], b[j][3]), acc[i][j], 0, 0, 0);
#endif
            }
#else
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < WMT; ++i)
#pragma unroll
                    for (int j = 0; j < WNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][e], b[j][e], acc[i][j], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int LV>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[Cfg<LV>::WMT][Cfg<LV>::WNT]) {
#pragma unroll
    for (int i = 0; i < Cfg<LV>::WMT; ++i)
#pragma unroll
        for (int j = 0; j < Cfg<LV>::WNT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// visit every accumulator element: f(row, col, value)   (32x32 block: column lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
template <int LV, typename Fn>
__device__ __forceinline__ void acc_each(f32x16 (&acc)[Cfg<LV>::WMT][Cfg<LV>::WNT], const Geo<LV>& g, Fn f) {
#pragma unroll
    for (int i = 0; i < Cfg<LV>::WMT; ++i)
#pragma unroll
        for (int j = 0; j < Cfg<LV>::WNT; ++j) {
            const int col = (g.nt0 + j) * 32 + g.l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (g.mt0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g.half;
                f(row, col, acc[i][j][r]);
            }
        }
}

// pdf_uniform(seed, idx) of common.h for idx < 2^32 (every index of a level is: mesh_check bounds B): the seed half of the hash is one
// value per site, so an element costs one 32-bit hash -- and the 64 independent elements of an unrolled epilogue no longer carry
// 64-bit index chains (round 5: those drove the kernel to 512 registers plus scratch)
__device__ __forceinline__ uint32_t md_key(unsigned long long seed) { return pdf_hash32((uint32_t)seed) ^ ((uint32_t)(seed >> 32) * 0x9E3779B9U); }
__device__ __forceinline__ bool md_keep(uint32_t key, uint32_t idx, float p) {
    return (float)(pdf_hash32(idx ^ key) >> 8) * (1.0f / 16777216.0f) >= p;
}
// the wave's tile positions of a global [V][ld] tensor -> registers (rows >= V read row V - 1)
template <int LV>
__device__ __forceinline__ void acc_load(f32x16 (&dst)[Cfg<LV>::WMT][Cfg<LV>::WNT], const Geo<LV>& g, const float* __restrict__ src, int ld) {
#pragma unroll
    for (int i = 0; i < Cfg<LV>::WMT; ++i)
#pragma unroll
        for (int j = 0; j < Cfg<LV>::WNT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = min((g.mt0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g.half, Cfg<LV>::V - 1);
                dst[i][j][r] = src[(long)row * ld + (g.nt0 + j) * 32 + g.l31];
            }
}

__device__ __forceinline__ float md_drop(float v, float p, float sc, uint32_t key, uint32_t idx) {
    return (p <= 0.f || md_keep(key, idx, p)) ? v * sc : 0.f;
}

// global [V][ld] (columns c0 .. c0 + C) * scale -> LDS image [VP][LD]; rows >= V zero-filled.  VP C / 4 = 4,096 float4 at every level = 16 per
// thread: all 16 loads are issued before the first LDS store (one load -> store pair at a time cost 16 dependent L2 round trips: 10 us per image)
template <int LV>
__device__ __forceinline__ void load_rows_scaled(float* __restrict__ dst, const float* __restrict__ src, int ld, int c0, float scale) {
    using G = Cfg<LV>;
    constexpr int C4 = G::C / 4, NI = G::VP * C4 / MD_THREADS, NB = NI < 8 ? NI : 8;
    if (MD_WHATIF == 6) return;
#pragma unroll
    for (int k0 = 0; k0 < NI; k0 += NB) {                               // two batches of 8 loads (32 registers)
        f32x4 t[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int i = threadIdx.x + (k0 + k) * MD_THREADS, v = i / C4, c = (i - v * C4) * 4;
            t[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (v < G::V) t[k] = ld4(src + (long)v * ld + c0 + c);
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int i = threadIdx.x + (k0 + k) * MD_THREADS, v = i / C4, c = (i - v * C4) * 4;
            st4(dst + v * G::LD + c, t[k] * scale);
        }
    }
}
template <int LV>
__device__ __forceinline__ void load_rows(float* __restrict__ dst, const float* __restrict__ src, int ld, int c0) {
    load_rows_scaled<LV>(dst, src, ld, c0, 1.f);
}
// LDS image rows < V -> global [V][ld] (columns c0 ..)
template <int LV>
__device__ __forceinline__ void store_rows(float* __restrict__ dst, int ld, int c0, const float* __restrict__ src) {
    using G = Cfg<LV>;
    constexpr int C4 = G::C / 4;
    for (int i = threadIdx.x; i < G::V * C4; i += MD_THREADS) {
        const int v = i / C4, c = (i - v * C4) * 4;
        st4(dst + (long)v * ld + c0 + c, ld4(src + v * G::LD + c));
    }
}

// ELL tables of this hand's Laplacian, staged in LDS once per kernel: ec[V][MD_ELLW] column indices, ev[V][MD_ELLW] values (Wd <= MD_ELLW, zero-padded)
#define MD_ELLW 11
template <int LV>
__device__ __forceinline__ void ell_stage(int* __restrict__ ec, float* __restrict__ ev, const int* __restrict__ col, const float* __restrict__ val, int Wd) {
    for (int i = threadIdx.x; i < Cfg<LV>::V * MD_ELLW; i += MD_THREADS) {      // rows padded to MD_ELLW entries (value 0, column 0): the products run unpredicated
        const int v = i / MD_ELLW, w = i - v * MD_ELLW;
        ec[i] = w < Wd ? col[v * Wd + w] : 0;
        ev[i] = w < Wd ? val[v * Wd + w] : 0.f;
    }
}
// LX = L X on the LDS images (fixed-width ELL, <= 11 non-zeros per row); optionally writes the interleaved [x | L x] rows the weight
// gradient of the following Linear reads (cat[v][2 (c0 + c) + {0, 1}], row stride ldcat).  All of a row's table entries are read first,
// then all its operand rows (round 5: one entry at a time from HBM was a chain of ~11 dependent L2 round trips per output: 19 us per product)
template <int LV>
__device__ __forceinline__ void spmm(float* __restrict__ LX, const float* __restrict__ X, const int* __restrict__ ec, const float* __restrict__ ev,
                                     int Wd, float* __restrict__ cat, int ldcat, int c0) {
    using G = Cfg<LV>;
    constexpr int C4 = G::C / 4;
    if (MD_WHATIF == 1) return;
    for (int i = threadIdx.x; i < G::V * C4; i += MD_THREADS) {
        const int v = i / C4, c = (i - v * C4) * 4;
        int cc[MD_ELLW]; float vv[MD_ELLW];
#pragma unroll
        for (int w = 0; w < MD_ELLW; ++w) { cc[w] = ec[v * MD_ELLW + w]; vv[w] = ev[v * MD_ELLW + w]; }
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < MD_ELLW; ++w) a += vv[w] * ld4(X + cc[w] * G::LD + c);
        st4(LX + v * G::LD + c, a);
        if (cat != nullptr) {
            const f32x4 x = ld4(X + v * G::LD + c);
            float* o = cat + (long)v * ldcat + 2 * (c0 + c);
            st4(o, f32x4{x[0], a[0], x[1], a[1]});
            st4(o + 4, f32x4{x[2], a[2], x[3], a[3]});
        }
    }
}
// X += L^T-type product of T (same ELL form with the transposed tables): the backward of the above
template <int LV>
__device__ __forceinline__ void spmm_add(float* __restrict__ X, const float* __restrict__ T, const int* __restrict__ ec, const float* __restrict__ ev, int Wd) {
    using G = Cfg<LV>;
    constexpr int C4 = G::C / 4;
    for (int i = threadIdx.x; i < G::V * C4; i += MD_THREADS) {
        const int v = i / C4, c = (i - v * C4) * 4;
        int cc[MD_ELLW]; float vv[MD_ELLW];
#pragma unroll
        for (int w = 0; w < MD_ELLW; ++w) { cc[w] = ec[v * MD_ELLW + w]; vv[w] = ev[v * MD_ELLW + w]; }
        f32x4 a = ld4(X + v * G::LD + c);
#pragma unroll
        for (int w = 0; w < MD_ELLW; ++w) a += vv[w] * ld4(T + cc[w] * G::LD + c);
        st4(X + v * G::LD + c, a);
    }
}

// LayerNorm of the LDS rows in place: buf = [relu](LN(buf) * gamma + beta); optional copies to HBM of the INPUT (zsave), of (mean, rstd) and of
// the OUTPUT.  A row belongs to GL = 256 / VP adjacent lanes (4, 2, 1), so every thread owns exactly 64 elements at every level -- float4
// chunks j GL + g of its row -- and the row sums close with at most two shuffles (round 5: one wave per row, 6-step reductions and one element
// per lane at C = 64 cost 13 / 22 / 40 us per call at the three levels).  Two passes (mean, then squared deviations) like the unfused kernel.
template <int LV>
__device__ __forceinline__ void ln_rows(float* __restrict__ buf, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, bool relu,
                                        float* __restrict__ zsave, float* __restrict__ st, float* __restrict__ ysave) {
    using G = Cfg<LV>;
    constexpr int GL = MD_THREADS / G::VP, NJ = G::C / (4 * GL);      // lanes per row; float4 chunks per thread (16)
    if (MD_WHATIF == 2) return;
    const int v = threadIdx.x / GL, gl = threadIdx.x % GL;
    const bool live = v < G::V;
    const int vr = live ? v : G::V - 1;
    f32x4 x[NJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { x[j] = ld4(buf + vr * G::LD + 4 * (j * GL + gl)); s += (x[j][0] + x[j][1]) + (x[j][2] + x[j][3]); }
#pragma unroll
    for (int o = 1; o < GL; o <<= 1) s += __shfl_xor(s, o, 64);
    const float mu = s / (float)G::C;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = x[j][e] - mu; sq += d * d; }
#pragma unroll
    for (int o = 1; o < GL; o <<= 1) sq += __shfl_xor(sq, o, 64);
    const float rs = 1.0f / sqrtf(sq / (float)G::C + eps);
    if (!live) return;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int cc = 4 * (j * GL + gl);
        const f32x4 gm = ld4(gamma + cc), bt = ld4(beta + cc);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = (x[j][e] - mu) * rs * gm[e] + bt[e];
            if (relu) o[e] = fmaxf(o[e], 0.f);
        }
        st4(buf + v * G::LD + cc, o);
        if (zsave != nullptr) st4(zsave + (long)v * G::C + cc, x[j]);
        if (ysave != nullptr) st4(ysave + (long)v * G::C + cc, o);
    }
    if (st != nullptr && gl == 0) { st[2 * v] = mu; st[2 * v + 1] = rs; }
}

// ---- attention on the matrix pipe ---------------------------------------------------------------------------------------------------
// K, V of the attended sample: LDS images XK, XV [VP][LD] (rows >= V zero).  One wave per (head, 32-query tile):
//   S^T[key][query] = sum_d K[key][d] (q[query][d] / sqrt(dh))        A = K rows (LDS b128), B = q rows (HBM float4)
//   P = exp(S - max) (keys >= V masked), l = sum P, P~ = dropout(P)    registers + one cross-half shuffle
//   O^T[d][query] = sum_key V[key][d] P~[key][query]                   A = V[key(lane)][d = lane & 31] (LDS b32), B = P~ straight from the S accumulators
// out[query][h DH + d] = O / l; stat = (max, l) per (head, query)
template <int LV>
__device__ __forceinline__ void attention_fwd(const Geo<LV>& g, const float* __restrict__ XK, const float* __restrict__ XV, const float* __restrict__ q,
                                              float* __restrict__ out, float* __restrict__ stat /*[H][V][2]*/, float p, unsigned long long seed,
                                              unsigned long long rowbase /* bb * H * V * V */) {
    using G = Cfg<LV>;
    constexpr int MT = G::MT, DH = G::DH, DT = DH >= 32 ? DH / 32 : 1;
    const float inv_norm = 1.f / sqrtf((float)DH), sc = 1.f / (1.f - p);
    const uint32_t dkey = md_key(seed);
    if (MD_WHATIF == 4) return;
    for (int item = g.wave; item < G::H * MT; item += MD_THREADS / 64) {
        const int h = item / MT, qt = item - h * MT;
        const int qrow = min(qt * 32 + g.l31, G::V - 1);            // (queries past V compute along and are never stored)
        f32x4 qf[DH / 8];
#pragma unroll
        for (int s = 0; s < DH / 8; ++s) qf[s] = ld4(q + (long)qrow * G::C + h * DH + 8 * s + 4 * g.half) * inv_norm;
        f32x16 S[MT];
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
#pragma unroll
            for (int s = 0; s < DH / 8; ++s) {
                const f32x4 kf = ld4(XK + (kt * 32 + g.l31) * G::LD + h * DH + 8 * s + 4 * g.half);
#pragma unroll
                for (int e = 0; e < 4; ++e) S[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[s][e], S[kt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // softmax over the keys of this lane's query: 16 MT values here + 16 MT in lane ^ 32
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < MT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g.half;
                if (kt == MT - 1 && key >= G::V) S[kt][r] = -INFINITY;
                m = fmaxf(m, S[kt][r]);
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
        const int qi = qt * 32 + g.l31;
        const uint32_t rid = (uint32_t)rowbase + (uint32_t)((h * G::V + qi) * G::V);
#pragma unroll
        for (int kt = 0; kt < MT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g.half;
                float pv = expf(S[kt][r] - m);
                l += pv;
                if (p > 0.f) pv = md_keep(dkey, rid + (uint32_t)key, p) ? pv * sc : 0.f;
                S[kt][r] = pv;
            }
        l += __shfl_xor(l, 32, 64);
        const float il = 1.f / l;
        f32x16 O[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
#pragma unroll
            for (int kt = 0; kt < MT; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g.half;
                    const float vf = XV[key * G::LD + h * DH + dt * 32 + g.l31];
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, S[kt][r], O[dt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);                  // (keeps the compiler from hoisting all MT x 16 LDS reads ahead of the first MFMA)
            }
        }
        if (qi < G::V) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int d = dt * 32 + 8 * r4 + 4 * g.half;          // rows d .. d + 3 of the O^T block
                    if (d < DH) st4(out + (long)qi * G::C + h * DH + d, f32x4{O[dt][4 * r4] * il, O[dt][4 * r4 + 1] * il, O[dt][4 * r4 + 2] * il, O[dt][4 * r4 + 3] * il});
                }
            if (g.half == 0 && stat != nullptr) { stat[((long)h * G::V + qi) * 2] = m; stat[((long)h * G::V + qi) * 2 + 1] = l; }
        }
    }
}

// ---- forward stages -------------------------------------------------------------------------------------------------------------------
template <int LV> struct Ctx {
    using G = Cfg<LV>;
    Geo<LV> g;
    float* XA; float* XB;
    int* ec; float* ev;                           // this hand's ELL tables in LDS (kernels with graph products: ell_stage)
    int hand, b, bb, B;                           // bb = hand * B + b: index of this (hand, sample) in the stacked [2 B] batch
    long row0;                                    // first row of this (hand, sample) in the stacked [2 B V] row space
    float p, sc;
    unsigned long long stepmix;
    bool train;
};

// y[V][C] (LDS dst, +bias) = A . W^T; every wave writes its tiles.  `dst` must not be an operand of the product.
template <int LV>
__device__ __forceinline__ void acc_to_lds(f32x16 (&acc)[Cfg<LV>::WMT][Cfg<LV>::WNT], const Geo<LV>& g, float* __restrict__ dst, const float* __restrict__ bias) {
    using G = Cfg<LV>;
#pragma unroll
    for (int j = 0; j < G::WNT; ++j) {
        const float bv = bias != nullptr ? bias[(g.nt0 + j) * 32 + g.l31] : 0.f;
#pragma unroll
        for (int i = 0; i < G::WMT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (g.mt0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g.half;
                dst[row * G::LD + (g.nt0 + j) * 32 + g.l31] = acc[i][j][r] + bv;
            }
    }
}

// One GCN_ResBlock (gcn.py:99-110).  In: x in HBM ([V][cin] rows of this (hand, sample)); when `x_in_lds` the block input (cin == C) already
// sits in XA.  Out: the block output in XA (and in HBM: `xout`).
template <int LV>
__device__ __forceinline__ void gcn_block(Ctx<LV>& c, const PdfMeshLevel& a, const MdGcn& P, int blk, const float* __restrict__ x, int cin, bool x_in_lds,
                                          const TapeOff& to, float* __restrict__ xout, bool relu_out) {
    using G = Cfg<LV>;
    constexpr int C = G::C;
    const int hd = c.hand;
    const int* col = c.ec; const float* val = c.ev;
    float* tape = a.tape;
    const bool tr = c.train;
    float* cat1 = tr ? tape + to.cat1(blk) + c.row0 * 2 * cin : nullptr;
    f32x16 accy[G::WMT][G::WNT], accs[G::WMT][G::WNT];
    acc_zero<LV>(accy);
    acc_zero<LV>(accs);
    MD_STAMP(0, 1 + blk * 8);
    // fc1 over cat(x, L x) and the shortcut over x, in column chunks of C (block 0: cin = 2 C)
    for (int c0 = 0; c0 < cin; c0 += C) {
        if (!(x_in_lds && c0 == 0)) {
            __syncthreads();                                            // (previous readers of XA are done)
            load_rows<LV>(c.XA, x, cin, c0);
        }
        __syncthreads();
        spmm<LV>(c.XB, c.XA, col, val, a.ell_w, cat1, 2 * cin, c0);
        __syncthreads();
        gemm_nt<LV, true>(accy, c.g, c.XA, c.XB, P.fc1.w[hd], 2 * cin, 2 * c0);
        gemm_nt<LV, false>(accs, c.g, c.XA, nullptr, P.sc.w[hd], cin, c0);
    }
    __syncthreads();
    MD_STAMP(0, 2 + blk * 8);                                           // fc1 + shortcut products (incl. loads and L x)
    // y = fc1 + b1 -> LN2 + ReLU -> h (XA)
    acc_to_lds<LV>(accy, c.g, c.XA, P.fc1.b[hd]);
    __syncthreads();
    MD_STAMP(0, 3 + blk * 8);                                           // epilogue
    ln_rows<LV>(c.XA, P.n2.g[hd], P.n2.b[hd], 1e-6f, true, tr ? tape + to.y(blk) + c.row0 * C : nullptr, tr ? tape + to.st2(blk) + c.row0 * 2 : nullptr, nullptr);
    __syncthreads();
    MD_STAMP(0, 4 + blk * 8);                                           // LN2
    spmm<LV>(c.XB, c.XA, col, val, a.ell_w, tr ? tape + to.cat2(blk) + c.row0 * 2 * C : nullptr, 2 * C, 0);
    __syncthreads();
    MD_STAMP(0, 5 + blk * 8);                                           // L h
    acc_zero<LV>(accy);
    gemm_nt<LV, true>(accy, c.g, c.XA, c.XB, P.fc2.w[hd], 2 * C, 0);
    __syncthreads();
    MD_STAMP(0, 6 + blk * 8);                                           // fc2 product
    // z = (s + bs) + dropout(y2 + b2) -> XA; LN3 (+ ReLU between blocks)
    {
        const uint32_t key = md_key(P.seed + c.stepmix);
        const uint32_t i0 = (uint32_t)(c.row0 * C);
        const float* b2 = P.fc2.b[hd]; const float* bs = P.sc.b[hd];
#pragma unroll
        for (int j = 0; j < G::WNT; ++j) {
            const int colj = (c.g.nt0 + j) * 32 + c.g.l31;
            const float b2v = b2[colj], bsv = bs[colj];
#pragma unroll
            for (int i = 0; i < G::WMT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (c.g.mt0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * c.g.half;
                    const float y2 = md_drop(accy[i][j][r] + b2v, c.p, c.sc, key, i0 + (uint32_t)(row * C + colj));
                    c.XA[row * G::LD + colj] = (accs[i][j][r] + bsv) + y2;
                }
        }
    }
    __syncthreads();
    MD_STAMP(0, 7 + blk * 8);                                           // z epilogue
    ln_rows<LV>(c.XA, P.n3.g[hd], P.n3.b[hd], 1e-6f, relu_out, tr ? tape + to.z(blk) + c.row0 * C : nullptr, tr ? tape + to.st3(blk) + c.row0 * 2 : nullptr, xout);
    MD_STAMP(0, 8 + blk * 8);                                           // LN3
}

// LN -> q / k / v projections.  In: x in XA.  Out: h (tape), q / k / v in HBM.  XA keeps h.
template <int LV>
__device__ __forceinline__ void qkv_stage(Ctx<LV>& c, const MdAttn& P, float* __restrict__ hsave, float* __restrict__ stsave,
                                          float* __restrict__ q, float* __restrict__ k, float* __restrict__ v) {
    using G = Cfg<LV>;
    constexpr int C = G::C;
    const int hd = c.hand;
    __syncthreads();
    ln_rows<LV>(c.XA, P.ln.g[hd], P.ln.b[hd], 1e-6f, false, nullptr, stsave, hsave);
    __syncthreads();
    const MdLin* lin[3] = {&P.q, &P.k, &P.v};
    float* dst[3] = {q, k, v};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        f32x16 acc[G::WMT][G::WNT];
        acc_zero<LV>(acc);
        gemm_nt<LV, false>(acc, c.g, c.XA, nullptr, lin[t]->w[hd], C, 0);
        const float* bias = lin[t]->b[hd];
        float* o = dst[t];
        acc_each<LV>(acc, c.g, [&](int row, int colj, float val) { if (row < G::V) o[(long)row * C + colj] = val + bias[colj]; });
    }
}

// attention (keys / values of stacked sample bkv) -> fc -> z = x + dropout(o) -> LN -> fc1 + ReLU + dropout -> fc2 -> x + dropout: the level's
// second half of SelfAttn.forward / inter_attn.forward (self_attn.py:63-85, 24-33; inter_attn.py:82-125)
template <int LV>
__device__ __forceinline__ void attn_tail_stage(Ctx<LV>& c, const MdAttn& P, const float* __restrict__ xres /* HBM [V][C]: the residual stream */,
                                                const float* __restrict__ q, const float* __restrict__ kk, const float* __restrict__ vv, int bkv,
                                                float* __restrict__ asave, float* __restrict__ statsave, float* __restrict__ zsave, float* __restrict__ stzsave,
                                                float* __restrict__ hnsave, float* __restrict__ tsave, float* __restrict__ xout) {
    using G = Cfg<LV>;
    constexpr int C = G::C;
    const int hd = c.hand;
    const long VC = (long)G::V * C;
    MD_STAMP(1, 0);
    __syncthreads();
    load_rows<LV>(c.XA, kk + (long)bkv * VC, C, 0);
    load_rows<LV>(c.XB, vv + (long)bkv * VC, C, 0);
    __syncthreads();
    MD_STAMP(1, 1);                                                     // K, V loads
    attention_fwd<LV>(c.g, c.XA, c.XB, q + (long)c.bb * VC, asave, statsave, c.p, P.seed_att + c.stepmix, (unsigned long long)c.bb * G::H * G::V * G::V);
    __syncthreads();                                                    // (asave written by this block: visible to it after the barrier)
    MD_STAMP(1, 2);                                                     // attention
    load_rows<LV>(c.XA, asave, C, 0);
    __syncthreads();
    f32x16 acc[G::WMT][G::WNT];
    acc_zero<LV>(acc);
    gemm_nt<LV, false>(acc, c.g, c.XA, nullptr, P.fc.w[hd], C, 0);
    {   // z = x + dropout(o + b) -> XB
        const uint32_t key = md_key(P.seed_z + c.stepmix), i0 = (uint32_t)(c.row0 * C);
        const float* bias = P.fc.b[hd];
        f32x16 res[G::WMT][G::WNT];                                     // the residual rows first: all 64 loads in flight (one at a time behind the
        acc_load<LV>(res, c.g, xres, C);                                // LDS stores of the loop below they cost ~30 k cycles per epilogue)
#pragma unroll
        for (int j = 0; j < G::WNT; ++j) {
            const int colj = (c.g.nt0 + j) * 32 + c.g.l31;
            const float bv = bias[colj];
#pragma unroll
            for (int i = 0; i < G::WMT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (c.g.mt0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * c.g.half;
                    const float o = md_drop(acc[i][j][r] + bv, c.p, c.sc, key, i0 + (uint32_t)(row * C + colj));
                    c.XB[row * G::LD + colj] = res[i][j][r] + o;
                }
        }
    }
    __syncthreads();
    MD_STAMP(1, 3);                                                     // fc product + z epilogue
    ln_rows<LV>(c.XB, P.ffln.g[hd], P.ffln.b[hd], 1e-6f, false, zsave, stzsave, hnsave);     // XB = hn; z saved to HBM (it is the residual of the tail)
    __syncthreads();
    MD_STAMP(1, 4);                                                     // LN
    acc_zero<LV>(acc);
    gemm_nt<LV, false>(acc, c.g, c.XB, nullptr, P.f1.w[hd], C, 0);
    {   // t = dropout(relu(fc1)) -> XA
        const uint32_t key = md_key(P.seed_t + c.stepmix), i0 = (uint32_t)(c.row0 * C);
        const float* bias = P.f1.b[hd];
        acc_each<LV>(acc, c.g, [&](int row, int colj, float val) {
            const float t = md_drop(fmaxf(val + bias[colj], 0.f), c.p, c.sc, key, i0 + (uint32_t)(row * C + colj));
            c.XA[row * G::LD + colj] = t;
            if (tsave != nullptr && row < G::V) tsave[(long)row * C + colj] = t;
        });
    }
    __syncthreads();
    acc_zero<LV>(acc);
    gemm_nt<LV, false>(acc, c.g, c.XA, nullptr, P.f2.w[hd], C, 0);
    __syncthreads();
    {   // x = z + dropout(u) -> XA and HBM.  z: in training from its HBM copy, else recomputed is not possible -> zsave is always given
        const uint32_t key = md_key(P.seed_x + c.stepmix), i0 = (uint32_t)(c.row0 * C);
        const float* bias = P.f2.b[hd];
        f32x16 res[G::WMT][G::WNT];
        acc_load<LV>(res, c.g, zsave, C);
#pragma unroll
        for (int j = 0; j < G::WNT; ++j) {
            const int colj = (c.g.nt0 + j) * 32 + c.g.l31;
            const float bv = bias[colj];
#pragma unroll
            for (int i = 0; i < G::WMT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (c.g.mt0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * c.g.half;
                    const float u = md_drop(acc[i][j][r] + bv, c.p, c.sc, key, i0 + (uint32_t)(row * C + colj));
                    const float xo = res[i][j][r] + u;
                    c.XA[row * G::LD + colj] = xo;
                    if (row < G::V) xout[(long)row * C + colj] = xo;
                }
        }
    }
    MD_STAMP(1, 5);                                                     // f1, f2 products + epilogues
}

template <int LV>
__device__ __forceinline__ void ctx_init(Ctx<LV>& c, const PdfMeshLevel& a, float* smem) {
    using G = Cfg<LV>;
    c.XA = smem; c.XB = smem + G::BUF;
    c.ec = reinterpret_cast<int*>(smem + 2 * G::BUF); c.ev = smem + 2 * G::BUF + G::V * MD_ELLW;
    c.B = a.B;
    c.bb = blockIdx.x; c.hand = c.bb / a.B; c.b = c.bb - c.hand * a.B;
    c.row0 = (long)c.bb * G::V;
    c.train = a.training != 0;
    c.p = c.train ? a.p : 0.f;
    c.sc = 1.f / (1.f - c.p);
    c.stepmix = (a.step != nullptr && c.p > 0.f) ? a.step[0] * 0x9E3779B97F4A7C15ull : 0ull;
}

// launch 1: the four GCN_ResBlocks of this (hand, sample) and the q / k / v projections of its SelfAttn
template <int LV>
__global__ __launch_bounds__(MD_THREADS, 1) void mesh_gcn_kernel(const PdfMeshLevel a) {
    using G = Cfg<LV>;
    constexpr int C = G::C;
    extern __shared__ float smem[];
    Ctx<LV> c;
    ctx_init<LV>(c, a, smem);
    const TapeOff to = tape_offsets(LV, a.B);
    float* tape = a.tape;
    // the per-(hand, sample) intermediates the chain hands from stage to stage through HBM: in training they ARE the tape; in eval mode the
    // same slots are used (the caller always provides the tape buffer)
    const float* x = a.x + c.row0 * a.cin0;
    MD_STAMP(0, 0);
    ell_stage<LV>(c.ec, c.ev, a.ell_col[c.hand], a.ell_val[c.hand], a.ell_w);       // (first read behind gcn_block's first barrier)
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        float* xo = tape + to.out(i) + c.row0 * C;
        gcn_block<LV>(c, a, a.gcn[i], i, x, i == 0 ? a.cin0 : C, i > 0, to, xo, i != 3);
        x = xo;
    }
    qkv_stage<LV>(c, a.self_, c.train ? tape + to.h(0) + c.row0 * C : nullptr, c.train ? tape + to.st(0) + c.row0 * 2 : nullptr,
                  tape + to.q(0) + c.row0 * C, tape + to.k(0) + c.row0 * C, tape + to.v(0) + c.row0 * C);
    MD_STAMP(0, 40);                                                    // LN + q / k / v
}

// launch 2 (CROSS = false): SelfAttn's attention + tail, then LN1 / LN2 and the shared q / k / v of the cross-hand step;
// launch 3 (CROSS = true): the cross-hand attention (keys / values of the other hand of the same sample, inter_attn.py:82-105) + tail
template <int LV, bool CROSS>
__global__ __launch_bounds__(MD_THREADS, 1) void mesh_att_kernel(const PdfMeshLevel a) {
    using G = Cfg<LV>;
    constexpr int C = G::C;
    extern __shared__ float smem[];
    Ctx<LV> c;
    ctx_init<LV>(c, a, smem);
    const TapeOff to = tape_offsets(LV, a.B);
    float* tape = a.tape;
    const long R = 2L * a.B * G::V;
    constexpr int A = CROSS ? 1 : 0;
    const MdAttn& P = CROSS ? a.cross : a.self_;
    const float* xres = CROSS ? tape + to.xo(0) + c.row0 * C : tape + to.out(3) + c.row0 * C;
    const float* q = CROSS ? a.qkv : tape + to.q(0);
    const float* k = CROSS ? a.qkv + R * C : tape + to.k(0);
    const float* v = CROSS ? a.qkv + 2 * R * C : tape + to.v(0);
    const int bkv = CROSS ? (c.bb + a.B) % (2 * a.B) : c.bb;
    float* xout = CROSS ? a.out + c.row0 * C : tape + to.xo(0) + c.row0 * C;
    attn_tail_stage<LV>(c, P, xres, q, k, v, bkv, tape + to.a_(A) + c.row0 * C, c.train ? tape + to.stat(A) + c.row0 * 8 : nullptr,
                        tape + to.az(A) + c.row0 * C, c.train ? tape + to.stz(A) + c.row0 * 2 : nullptr, c.train ? tape + to.hn(A) + c.row0 * C : nullptr,
                        c.train ? tape + to.t(A) + c.row0 * C : nullptr, xout);
    if constexpr (!CROSS)
        qkv_stage<LV>(c, a.cross, c.train ? tape + to.h(1) + c.row0 * C : nullptr, c.train ? tape + to.st(1) + c.row0 * 2 : nullptr,
                      a.qkv + c.row0 * C, a.qkv + R * C + c.row0 * C, a.qkv + 2 * R * C + c.row0 * C);
}

template <int LV>
static int mesh_fwd_launch(const PdfMeshLevel& a, hipStream_t s) {
    using G = Cfg<LV>;
    const size_t smem = (size_t)2 * G::BUF * sizeof(float), smem_g = smem + (size_t)2 * G::V * MD_ELLW * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_gcn_kernel<LV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_g)) return (int)e;
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_att_kernel<LV, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) return (int)e;
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_att_kernel<LV, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(mesh_gcn_kernel<LV>, dim3(2 * a.B), dim3(MD_THREADS), smem_g, s, a);
    PDF_LAUNCH_CHECK();
    hipLaunchKernelGGL((mesh_att_kernel<LV, false>), dim3(2 * a.B), dim3(MD_THREADS), smem, s, a);
    PDF_LAUNCH_CHECK();
    hipLaunchKernelGGL((mesh_att_kernel<LV, true>), dim3(2 * a.B), dim3(MD_THREADS), smem, s, a);
    PDF_LAUNCH_CHECK();
    return 0;
}

static int mesh_check(const PdfMeshLevel* a) {
    if (a == nullptr || a->level < 0 || a->level > 2 || a->B < 1 || a->x == nullptr || a->out == nullptr || a->tape == nullptr || a->qkv == nullptr) return PDF_E_BADARG;
    if (a->cin0 != 2 * (256 >> a->level) || a->ell_w < 1 || a->ell_w > MD_ELLW) return PDF_E_BADARG;
    if (a->training && (a->p < 0.f || a->p >= 1.f)) return PDF_E_BADARG;
    if (a->B > 2048) return PDF_E_BADARG;                              // 32-bit dropout indices: 2 B H V^2 and 2 B V C stay below 2^32
    return 0;
}

// One DualGraphLayer forward (position embedding already added to x).  Launches three kernels on `stream`; allocates nothing, synchronises nothing.
#if MD_X3
#define pdf_mesh_level_fwd pdf_mesh_level_fwd_x3
#define pdf_mesh_level_bwd pdf_mesh_level_bwd_x3
#elif MD_BF16
#define pdf_mesh_level_fwd pdf_mesh_level_fwd_bf16
#define pdf_mesh_level_bwd pdf_mesh_level_bwd_bf16
#endif
PDF_API int pdf_mesh_level_fwd(const PdfMeshLevel* a, hipStream_t stream) {
    if (int rc = mesh_check(a)) return rc;
    switch (a->level) {
        case 0: return mesh_fwd_launch<0>(*a, stream);
        case 1: return mesh_fwd_launch<1>(*a, stream);
        default: return mesh_fwd_launch<2>(*a, stream);
    }
}
#if !MD_BF16
PDF_API int pdf_debug_mesh_level_size() { return (int)sizeof(PdfMeshLevel); }
#endif

// =====================================================================================================================================
// Backward.  Five launches per level: cross tail | cross attention + projections | self tail | self attention + projections | 4 GCN blocks.
// The data gradient chain stays in LDS like the forward; every dY a weight-gradient GEMM needs goes to `gtape`, the LayerNorm parameter
// gradients are accumulated from here with atomics (per wave: its rows' column sums), and pdf_mesh_level_bwd then issues the library's
// weight-gradient launches on the side stream.

// LayerNorm backward of the LDS rows in place: buf holds dL/dy (before the ReLU mask); afterwards dL/dz (+ add).
//   y = [relu](xhat * gamma + beta), xhat = (z - mean) * rstd;  dz = rstd (g - mean(g) - xhat mean(g xhat)), g = dy' gamma
// z, (mean, rstd) from the tape.  Pass A, one thread per column (x 256 / C row groups): dgamma += sum dy' xhat, dbeta += sum dy' over this
// workgroup's rows, 256 / C atomics per column.  Pass B, GL lanes per row like the forward.  Contains two barriers (every thread must call it).
template <int LV>
__device__ __forceinline__ void ln_bwd_rows(float* __restrict__ buf, const float* __restrict__ zin, const float* __restrict__ st,
                                            const float* __restrict__ gamma, const float* __restrict__ beta, bool relu, const float* __restrict__ add,
                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ save) {
    using G = Cfg<LV>;
    constexpr int GL = MD_THREADS / G::VP, NJ = G::C / (4 * GL), NG = MD_THREADS / G::C;
    if (dgamma != nullptr || dbeta != nullptr) {
        const int cc = threadIdx.x % G::C, rg = threadIdx.x / G::C;
        const float gm = gamma[cc], bt = relu ? beta[cc] : 0.f;
        float ag = 0.f, ab = 0.f;
        constexpr int UB = 8;                                           // rows per batch: their loads are all in flight together (one L2 round trip per
        for (int v0 = rg; v0 < G::V; v0 += NG * UB) {                   // row made this pass 45-70 k cycles)
            float zz[UB], mu[UB], rr[UB], dd[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int v = v0 + u * NG, vv = v < G::V ? v : rg;
                zz[u] = zin[(long)vv * G::C + cc]; mu[u] = st[2 * vv]; rr[u] = st[2 * vv + 1];
                dd[u] = v < G::V ? buf[vv * G::LD + cc] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const float xh = (zz[u] - mu[u]) * rr[u];
                float d = dd[u];
                if (relu && !(xh * gm + bt > 0.f)) d = 0.f;
                ag += d * xh; ab += d;
            }
        }
        if (dgamma != nullptr) atomicAdd(dgamma + cc, ag);
        if (dbeta != nullptr) atomicAdd(dbeta + cc, ab);
    }
    __syncthreads();
    const int v = threadIdx.x / GL, gl = threadIdx.x % GL;
    const bool live = v < G::V;
    const int vr = live ? v : G::V - 1;
    const float mu = st[2 * vr], rs = st[2 * vr + 1];
    f32x4 g[NJ], xh[NJ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int cc = 4 * (j * GL + gl);
        const f32x4 d = ld4(buf + vr * G::LD + cc), z = ld4(zin + (long)vr * G::C + cc), gm = ld4(gamma + cc);
        f32x4 bt = {0.f, 0.f, 0.f, 0.f};
        if (relu) bt = ld4(beta + cc);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xh[j][e] = (z[e] - mu) * rs;
            float dd = d[e];
            if (relu && !(xh[j][e] * gm[e] + bt[e] > 0.f)) dd = 0.f;
            g[j][e] = dd * gm[e];
            s1 += g[j][e]; s2 += g[j][e] * xh[j][e];
        }
    }
#pragma unroll
    for (int o = 1; o < GL; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    s1 /= (float)G::C; s2 /= (float)G::C;
    if (live) {
#pragma unroll
        for (int j0 = 0; j0 < NJ; j0 += 8) {                            // (the residual's gradient in batches of 8 loads, ahead of the stores)
            f32x4 ad[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) ad[j] = add != nullptr ? ld4(add + (long)v * G::C + 4 * ((j0 + j) * GL + gl)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cc = 4 * ((j0 + j) * GL + gl);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[j0 + j][e] - s1 - xh[j0 + j][e] * s2) + ad[j][e];
                st4(buf + v * G::LD + cc, o);
                if (save != nullptr) st4(save + (long)v * G::C + cc, o);
            }
        }
    }
    __syncthreads();
}

// u = drop(t) for the V x C rows of `src` (LDS image, or HBM with ld = C when `from_global`): u -> LDS image `dstu` (+ HBM `save`); optionally the
// undropped t -> LDS image `dstt`.  16 float4 per thread, loads of a batch of 8 in flight before the stores; rows >= V of the images are zeroed.
template <int LV>
__device__ __forceinline__ void drop_rows(const float* __restrict__ src, bool from_global, float* __restrict__ dstt, float* __restrict__ dstu,
                                          float* __restrict__ save, float p, float sc, uint32_t key, uint32_t i0) {
    using G = Cfg<LV>;
    constexpr int C = G::C, C4 = C / 4, NI = G::VP * C4 / MD_THREADS, NB = NI < 8 ? NI : 8;
#pragma unroll
    for (int k0 = 0; k0 < NI; k0 += NB) {
        f32x4 t[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int i = threadIdx.x + (k0 + k) * MD_THREADS, v = i / C4, cc = (i - v * C4) * 4;
            t[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (v < G::V) t[k] = from_global ? ld4(src + (long)v * C + cc) : ld4(src + v * G::LD + cc);
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int i = threadIdx.x + (k0 + k) * MD_THREADS, v = i / C4, cc = (i - v * C4) * 4;
            f32x4 u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = md_drop(t[k][e], p, sc, key, i0 + (uint32_t)(v * C + cc + e));
            if (dstt != nullptr) st4(dstt + v * G::LD + cc, t[k]);
            st4(dstu + v * G::LD + cc, u);
            if (save != nullptr && v < G::V) st4(save + (long)v * C + cc, u);
        }
    }
}

// store the wave's accumulator tiles as rows of a global [V][ld] tensor (columns c0 + ...)
template <int LV>
__device__ __forceinline__ void acc_to_global(f32x16 (&acc)[Cfg<LV>::WMT][Cfg<LV>::WNT], const Geo<LV>& g, float* __restrict__ dst, int ld, int c0) {
    acc_each<LV>(acc, g, [&](int row, int colj, float val) { if (row < Cfg<LV>::V) dst[(long)row * ld + c0 + colj] = val; });
}

// ---- attention backward, query side: dq of this (hand, sample)'s queries against the keys / values in XK, XV -------------------------------
template <int LV>
__device__ __forceinline__ void attention_bwd_q(const Geo<LV>& g, const float* __restrict__ XK, const float* __restrict__ XV, const float* __restrict__ q,
                                                const float* __restrict__ o, const float* __restrict__ dO, const float* __restrict__ stat,
                                                float* __restrict__ dq, float p, unsigned long long seed, unsigned long long rowbase,
                                                int part = 0, int parts = 1) {
    using G = Cfg<LV>;
    constexpr int MT = G::MT, DH = G::DH, DT = DH >= 32 ? DH / 32 : 1, MTC = MT < 4 ? MT : 4;
    const float inv_norm = 1.f / sqrtf((float)DH), sc = 1.f / (1.f - p);
    const uint32_t dkey = md_key(seed);
    // (head, query tile) items: independent; `parts` workgroups share them (part = this one's share), `aw` waves of each take one at a time
    const int aw = md_active_waves(G::H * MT, parts);
    for (int item = g.wave + part * aw; g.wave < aw && item < G::H * MT; item += parts * aw) {
        const int h = item / MT, qt = item - h * MT;
        const int qi = qt * 32 + g.l31, qrow = min(qi, G::V - 1);
        f32x4 qf[DH / 8], gf[DH / 8];
        float D = 0.f;
#pragma unroll
        for (int s = 0; s < DH / 8; ++s) {
            const long off = (long)qrow * G::C + h * DH + 8 * s + 4 * g.half;
            qf[s] = ld4(q + off) * inv_norm;
            gf[s] = ld4(dO + off);
            const f32x4 of = ld4(o + off);
            D += gf[s][0] * of[0] + gf[s][1] * of[1] + gf[s][2] * of[2] + gf[s][3] * of[3];
        }
        D += __shfl_xor(D, 32, 64);
        const float m = stat[((long)h * G::V + qrow) * 2], il = 1.f / stat[((long)h * G::V + qrow) * 2 + 1];
        const uint32_t rid = (uint32_t)rowbase + (uint32_t)((h * G::V + qi) * G::V);
        f32x16 O[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
        // keys in chunks of MTC tiles: S and dP of a chunk live in 2 x MTC x 16 accumulator registers (MT = 8 at once would take 256)
#pragma unroll 1
        for (int kc = 0; kc < MT; kc += MTC) {
            f32x16 S[MTC], P[MTC];
#pragma unroll
            for (int i = 0; i < MTC; ++i) {
                const int kt = kc + i;
#pragma unroll
                for (int r = 0; r < 16; ++r) { S[i][r] = 0.f; P[i][r] = 0.f; }
#pragma unroll
                for (int s = 0; s < DH / 8; ++s) {
                    const f32x4 kf = ld4(XK + (kt * 32 + g.l31) * G::LD + h * DH + 8 * s + 4 * g.half);
                    const f32x4 vf = ld4(XV + (kt * 32 + g.l31) * G::LD + h * DH + 8 * s + 4 * g.half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        S[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[s][e], S[i], 0, 0, 0);
                        P[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[e], gf[s][e], P[i], 0, 0, 0);          // dP~^T[key][q] = V[key] . dO[q]
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < MTC; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = (kc + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g.half;
                    float a = expf(S[i][r] - m) * il;
                    if (key >= G::V) a = 0.f;
                    float gv = P[i][r];
                    if (p > 0.f) gv = md_keep(dkey, rid + (uint32_t)key, p) ? gv * sc : 0.f;
                    S[i][r] = a * (gv - D) * inv_norm;
                }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int i = 0; i < MTC; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = (kc + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g.half;
                        const float kf = XK[key * G::LD + h * DH + dt * 32 + g.l31];
                        O[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf, S[i][r], O[dt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (qi < G::V) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int d = dt * 32 + 8 * r4 + 4 * g.half;
                    if (d < DH) st4(dq + (long)qi * G::C + h * DH + d, f32x4{O[dt][4 * r4], O[dt][4 * r4 + 1], O[dt][4 * r4 + 2], O[dt][4 * r4 + 3]});
                }
        }
    }
}

// ---- attention backward, key side: dk, dv of this (hand, sample)'s keys from the queries that attended to them ------------------------------
// XQ = q / sqrt(dh) and XG = dO of the QUERYING sample (LDS images, rows >= V zero); sm = [3][H][VP]: max, 1 / sumexp, D = dO . O per (head, query)
template <int LV>
__device__ __forceinline__ void attention_bwd_kv(const Geo<LV>& g, const float* __restrict__ XQ, const float* __restrict__ XG, const float* __restrict__ sm,
                                                 const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ dk, float* __restrict__ dv,
                                                 float p, unsigned long long seed, unsigned long long rowbase, int part = 0, int parts = 1) {
    using G = Cfg<LV>;
    constexpr int MT = G::MT, DH = G::DH, DT = DH >= 32 ? DH / 32 : 1, HV = G::H * G::VP, MTC = MT < 4 ? MT : 4;
    const float sc = 1.f / (1.f - p);
    const uint32_t dkey = md_key(seed);
    const int aw = md_active_waves(G::H * MT, parts);
    for (int item = g.wave + part * aw; g.wave < aw && item < G::H * MT; item += parts * aw) {
        const int h = item / MT, kt = item - h * MT;
        const int key = kt * 32 + g.l31, krow = min(key, G::V - 1);
        f32x4 kf[DH / 8], vf[DH / 8];
#pragma unroll
        for (int s = 0; s < DH / 8; ++s) {
            const long off = (long)krow * G::C + h * DH + 8 * s + 4 * g.half;
            kf[s] = ld4(k + off);
            vf[s] = ld4(v + off);
        }
        f32x16 OK[DT], OV[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { OK[dt][r] = 0.f; OV[dt][r] = 0.f; }
#pragma unroll 1
        for (int mc = 0; mc < MT; mc += MTC) {                          // queries in chunks of MTC tiles (see attention_bwd_q)
            f32x16 S[MTC], P[MTC];
#pragma unroll
            for (int i = 0; i < MTC; ++i) {
                const int mt = mc + i;
#pragma unroll
                for (int r = 0; r < 16; ++r) { S[i][r] = 0.f; P[i][r] = 0.f; }
#pragma unroll
                for (int s = 0; s < DH / 8; ++s) {
                    const f32x4 qa = ld4(XQ + (mt * 32 + g.l31) * G::LD + h * DH + 8 * s + 4 * g.half);
                    const f32x4 ga = ld4(XG + (mt * 32 + g.l31) * G::LD + h * DH + 8 * s + 4 * g.half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        S[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[e], kf[s][e], S[i], 0, 0, 0);          // S[q][key]
                        P[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[e], vf[s][e], P[i], 0, 0, 0);          // dP~[q][key] = dO[q] . V[key]
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < MTC; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qi = (mc + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g.half;
                    const float m = sm[h * G::VP + qi], il = sm[HV + h * G::VP + qi], D = sm[2 * HV + h * G::VP + qi];
                    const float a = expf(S[i][r] - m) * il;                    // (queries >= V: il = 0)
                    float at = a, gv = P[i][r];
                    if (p > 0.f) {
                        const bool keep = md_keep(dkey, (uint32_t)rowbase + (uint32_t)((h * G::V + qi) * G::V + key), p);
                        at = keep ? a * sc : 0.f;
                        gv = keep ? gv * sc : 0.f;
                    }
                    S[i][r] = a * (gv - D);                                    // x (q / sqrt(dh)) below = dS q / sqrt(dh)
                    P[i][r] = at;
                }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int i = 0; i < MTC; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qi = (mc + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g.half;
                        const float qa = XQ[qi * G::LD + h * DH + dt * 32 + g.l31];
                        const float ga = XG[qi * G::LD + h * DH + dt * 32 + g.l31];
                        OK[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa, S[i][r], OK[dt], 0, 0, 0);
                        OV[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, P[i][r], OV[dt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (key < G::V) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int d = dt * 32 + 8 * r4 + 4 * g.half;
                    if (d < DH) {
                        st4(dk + (long)key * G::C + h * DH + d, f32x4{OK[dt][4 * r4], OK[dt][4 * r4 + 1], OK[dt][4 * r4 + 2], OK[dt][4 * r4 + 3]});
                        st4(dv + (long)key * G::C + h * DH + d, f32x4{OV[dt][4 * r4], OV[dt][4 * r4 + 1], OV[dt][4 * r4 + 2], OV[dt][4 * r4 + 3]});
                    }
                }
        }
    }
}

// tail of an attention block, backward: x_out = z + drop(f2(t)), t = drop(relu(f1(LN(z)))), z = x + drop(fc(a))
//   in: dX (HBM).  out (gtape): du, dt_pre, do, dc = d a, dxr = d z (all of it)
template <int LV, bool CROSS>
__global__ __launch_bounds__(MD_THREADS, 1) void mesh_att_bwd1_kernel(const PdfMeshLevel a) {
    using G = Cfg<LV>;
    constexpr int C = G::C, A = CROSS ? 1 : 0;
    extern __shared__ float smem[];
    Ctx<LV> c;
    ctx_init<LV>(c, a, smem);
    const int hd = c.hand;
    const TapeOff to = tape_offsets(LV, a.B);
    const GTapeOff go = gtape_offsets(LV, a.B);
    const MdAttn& P = CROSS ? a.cross : a.self_;
    const MdAttnG& GP = CROSS ? a.gcross : a.gself;
    const float* tape = a.tape;
    float* gt = a.gtape;
    const long ro = c.row0 * C;
    const uint32_t i0 = (uint32_t)ro;
    const float* dX = CROSS ? a.dout + ro : gt + go.att(1, 8) + ro;
    MD_STAMP(2, 0);
    // du = drop_x(dX) -> XB;  XA = dX
    drop_rows<LV>(dX, true, c.XA, c.XB, gt + go.att(A, 0) + ro, c.p, c.sc, md_key(P.seed_x + c.stepmix), i0);
    __syncthreads();
    MD_STAMP(2, 1);                                                     // du
    f32x16 acc[G::WMT][G::WNT];
    acc_zero<LV>(acc);
    gemm_nn<LV>(acc, c.g, c.XB, P.f2.w[hd], C, 0, 1);                // dt = du W2
    __syncthreads();
    MD_STAMP(2, 2);                                                     // product
    {   // dt_pre = dt * sc where t != 0 -> XB, gtape
        const float* t = tape + to.t(A) + ro;
        float* dtp = gt + go.att(A, 1) + ro;
        f32x16 tv[G::WMT][G::WNT];
        acc_load<LV>(tv, c.g, t, C);
#pragma unroll
        for (int j = 0; j < G::WNT; ++j) {
            const int colj = (c.g.nt0 + j) * 32 + c.g.l31;
#pragma unroll
            for (int i = 0; i < G::WMT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (c.g.mt0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * c.g.half;
                    const float o = tv[i][j][r] != 0.f ? acc[i][j][r] * c.sc : 0.f;
                    c.XB[row * G::LD + colj] = row < G::V ? o : 0.f;
                    if (row < G::V) dtp[(long)row * C + colj] = o;
                }
        }
    }
    __syncthreads();
    MD_STAMP(2, 3);                                                     // dt_pre epilogue
    acc_zero<LV>(acc);
    gemm_nn<LV>(acc, c.g, c.XB, P.f1.w[hd], C, 0, 1);                // d hn = dt_pre W1
    __syncthreads();
    MD_STAMP(2, 4);                                                     // product
    acc_to_lds<LV>(acc, c.g, c.XB, nullptr);
    __syncthreads();
    MD_STAMP(2, 5);                                                     // to LDS
    // LN_ff backward; dz = dX + (that) -> XB and gtape.dxr
    ln_bwd_rows<LV>(c.XB, tape + to.az(A) + ro, tape + to.stz(A) + c.row0 * 2, P.ffln.g[hd], nullptr, false, dX, GP.ffln.g[hd], GP.ffln.b[hd],
                    gt + go.att(A, 7) + ro);
    __syncthreads();
    MD_STAMP(2, 6);                                                     // LN backward
    // do = drop_z(dz) -> XA, gtape
    drop_rows<LV>(c.XB, false, nullptr, c.XA, gt + go.att(A, 2) + ro, c.p, c.sc, md_key(P.seed_z + c.stepmix), i0);
    __syncthreads();
    MD_STAMP(2, 7);                                                     // do
    acc_zero<LV>(acc);
    gemm_nn<LV>(acc, c.g, c.XA, P.fc.w[hd], C, 0, 1);                // dc = do Wfc
    acc_to_global<LV>(acc, c.g, gt + go.att(A, 6) + ro, C, 0);
    MD_STAMP(2, 8);                                                     // product + store
}

// attention + projections + LN of an attention block, backward: in dc (every workgroup's: kernel boundary in front), dxr; out dq, dk, dv, dxin
// PART 0: everything in one workgroup per (hand, sample).  PART 1 / 2 (round 5, late): the two attention passes -- (1) dq, (2) dk / dv: independent,
// together three quarters of this kernel at level 2 -- in TWO workgroups per (hand, sample) (blockIdx.y), then the projections and the LayerNorm
// in a second launch: the 64-workgroup chain becomes 128 + 64.
// PART 1 takes `parts`: the (head, tile) items of each pass are shared by that many workgroups -- grid (2 B, 2 * parts).
template <int LV, bool CROSS, int PART>
__global__ __launch_bounds__(MD_THREADS, 1) void mesh_att_bwd2_kernel(const PdfMeshLevel a, const int parts) {
    using G = Cfg<LV>;
    constexpr int C = G::C, A = CROSS ? 1 : 0, HV = G::H * G::VP;
    extern __shared__ float smem[];
    Ctx<LV> c;
    ctx_init<LV>(c, a, smem);
    float* sm = smem + 2 * G::BUF;                                      // [3][H][VP]
    const int hd = c.hand;
    const TapeOff to = tape_offsets(LV, a.B);
    const GTapeOff go = gtape_offsets(LV, a.B);
    const MdAttn& P = CROSS ? a.cross : a.self_;
    const MdAttnG& GP = CROSS ? a.gcross : a.gself;
    const float* tape = a.tape;
    float* gt = a.gtape;
    const long R = 2L * a.B * G::V, VC = (long)G::V * C, ro = c.row0 * C;
    const float* q = CROSS ? a.qkv : tape + to.q(0);
    const float* k = CROSS ? a.qkv + R * C : tape + to.k(0);
    const float* v = CROSS ? a.qkv + 2 * R * C : tape + to.v(0);
    const int bo = CROSS ? (c.bb + a.B) % (2 * a.B) : c.bb;             // the other party: whose keys my queries saw AND whose queries saw my keys
    const float* att_o = tape + to.a_(A);
    const float* stat = tape + to.stat(A);
    const float* dc = gt + go.att(A, 6);
    const float inv_norm = 1.f / sqrtf((float)G::DH);
    MD_STAMP(3, 0);
    const int np = PART == 1 ? parts : 1, part = PART == 1 ? (int)blockIdx.y % np : 0;
    const bool do1 = PART == 0 || (PART == 1 && (int)blockIdx.y < np), do2 = PART == 0 || (PART == 1 && (int)blockIdx.y >= np);
    if (do1) {
    // (1) dq of my queries against the other party's keys / values
    load_rows<LV>(c.XA, k + (long)bo * VC, C, 0);
    load_rows<LV>(c.XB, v + (long)bo * VC, C, 0);
    __syncthreads();
    attention_bwd_q<LV>(c.g, c.XA, c.XB, q + ro, att_o + ro, dc + ro, stat + c.row0 * 8, gt + go.att(A, 3) + ro, c.p, P.seed_att + c.stepmix,
                        (unsigned long long)c.bb * G::H * G::V * G::V, part, np);
    __syncthreads();
    MD_STAMP(3, 1);                                                     // loads + dq
    }
    if (do2) {
    // (2) dk, dv of my keys from the other party's queries
    load_rows_scaled<LV>(c.XA, q + (long)bo * VC, C, 0, inv_norm);
    load_rows<LV>(c.XB, dc + (long)bo * VC, C, 0);
    for (int i = threadIdx.x; i < HV; i += MD_THREADS) {
        const int h = i / G::VP, qi = i - h * G::VP;
        float m = 0.f, il = 0.f, D = 0.f;
        if (qi < G::V) {
            const float* st = stat + (long)bo * G::V * 8 + ((long)h * G::V + qi) * 2;
            m = st[0]; il = 1.f / st[1];
            const float* go_ = dc + (long)bo * VC + (long)qi * C + h * G::DH;
            const float* oo = att_o + (long)bo * VC + (long)qi * C + h * G::DH;
#pragma unroll
            for (int d = 0; d < G::DH; d += 4) {
                const f32x4 x = ld4(go_ + d), y = ld4(oo + d);
                D += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
            }
        }
        sm[i] = m; sm[HV + i] = il; sm[2 * HV + i] = D;
    }
    __syncthreads();
    MD_STAMP(3, 2);                                                     // loads + D
    attention_bwd_kv<LV>(c.g, c.XA, c.XB, sm, k + ro, v + ro, gt + go.att(A, 4) + ro, gt + go.att(A, 5) + ro, c.p, P.seed_att + c.stepmix,
                         (unsigned long long)bo * G::H * G::V * G::V, part, np);
    __syncthreads();
    MD_STAMP(3, 3);                                                     // dk, dv
    }
    if constexpr (PART == 1) return;
    // (3) d n = dq Wq + dk Wk + dv Wv  (n = LN(x): the projections' common input)
    f32x16 acc[G::WMT][G::WNT];
    acc_zero<LV>(acc);
    load_rows<LV>(c.XA, gt + go.att(A, 3) + ro, C, 0);
    load_rows<LV>(c.XB, gt + go.att(A, 4) + ro, C, 0);
    __syncthreads();
    gemm_nn<LV>(acc, c.g, c.XA, P.q.w[hd], C, 0, 1);
    gemm_nn<LV>(acc, c.g, c.XB, P.k.w[hd], C, 0, 1);
    __syncthreads();
    load_rows<LV>(c.XA, gt + go.att(A, 5) + ro, C, 0);
    __syncthreads();
    gemm_nn<LV>(acc, c.g, c.XA, P.v.w[hd], C, 0, 1);
    acc_to_lds<LV>(acc, c.g, c.XB, nullptr);
    __syncthreads();
    MD_STAMP(3, 4);                                                     // three products
    // (4) LN backward (input: the residual stream x of the block) + the residual's own gradient dxr -> dxin
    const float* xin = CROSS ? tape + to.xo(0) + ro : tape + to.out(3) + ro;
    ln_bwd_rows<LV>(c.XB, xin, tape + to.st(A) + c.row0 * 2, P.ln.g[hd], nullptr, false, gt + go.att(A, 7) + ro, GP.ln.g[hd], GP.ln.b[hd],
                    gt + go.att(A, 8) + ro);
    MD_STAMP(3, 5);                                                     // LN backward
}

// One GCN_ResBlock backward up to the gradients of its two graph convolutions' common input (gcn.py:99-110 reversed).
// In: XA = d out (LDS).  Out: XA = dz (gradient at the shortcut's output), XB = dy (gradient at fc1's output); dz, dy2, dy in gtape.
template <int LV>
__device__ __forceinline__ void gcn_bwd_head(Ctx<LV>& c, const PdfMeshLevel& a, int blk, const TapeOff& to, const GTapeOff& go) {
    using G = Cfg<LV>;
    constexpr int C = G::C;
    const int hd = c.hand;
    const MdGcn& P = a.gcn[blk];
    const MdGcnG& GP = a.ggcn[blk];
    const float* tape = a.tape;
    float* gt = a.gtape;
    const long ro = c.row0 * C;
    const uint32_t i0 = (uint32_t)ro;
    const int* colT = c.ec; const float* valT = c.ev;
    float* XA = c.XA; float* XB = c.XB;
    MD_STAMP(4, blk * 8);
    // LN3 (+ the ReLU GraphLayer puts between blocks) backward in place: XA = dz; saved for the shortcut's weight gradient
    ln_bwd_rows<LV>(XA, tape + to.z(blk) + ro, tape + to.st3(blk) + c.row0 * 2, P.n3.g[hd], P.n3.b[hd], blk != 3, nullptr, GP.n3.g[hd], GP.n3.b[hd],
                    gt + go.dz(blk) + ro);
    __syncthreads();
    MD_STAMP(4, blk * 8 + 1);                                           // LN3 backward
    // dy2 = drop(dz) -> XB, gtape
    drop_rows<LV>(XA, false, nullptr, XB, gt + go.dy2(blk) + ro, c.p, c.sc, md_key(P.seed + c.stepmix), i0);
    __syncthreads();
    MD_STAMP(4, blk * 8 + 2);                                           // dy2
    {   // fc2 backward: d cat2 = dy2 W2; dh = even columns + L^T (odd columns).  One accumulator set at a time: the odd part goes to XA (free:
        // dz is in gtape and comes back below), the even part replaces the operand once every wave is done with it
        f32x16 acc[G::WMT][G::WNT];
        acc_zero<LV>(acc);
        gemm_nn<LV>(acc, c.g, XB, P.fc2.w[hd], 2 * C, 1, 2);
        acc_to_lds<LV>(acc, c.g, XA, nullptr);
        acc_zero<LV>(acc);
        gemm_nn<LV>(acc, c.g, XB, P.fc2.w[hd], 2 * C, 0, 2);
        __syncthreads();
        acc_to_lds<LV>(acc, c.g, XB, nullptr);
    }
    __syncthreads();
    MD_STAMP(4, blk * 8 + 3);                                           // fc2 backward products
    spmm_add<LV>(XB, XA, colT, valT, a.ell_w);
    __syncthreads();
    MD_STAMP(4, blk * 8 + 4);                                           // L^T
    // LN2 + ReLU backward in place: XB = dy
    ln_bwd_rows<LV>(XB, tape + to.y(blk) + ro, tape + to.st2(blk) + c.row0 * 2, P.n2.g[hd], P.n2.b[hd], true, nullptr, GP.n2.g[hd], GP.n2.b[hd],
                    gt + go.dy(blk) + ro);
    load_rows<LV>(XA, gt + go.dz(blk) + ro, C, 0);                      // dz again (this workgroup wrote it: visible after the barrier)
    __syncthreads();
    MD_STAMP(4, blk * 8 + 5);                                           // LN2 backward + reload
}

// One GCN_ResBlock backward per launch (blk = 3, 2, 1, then 0 with BLK0): in d out_blk (self.dxin for block 3, else what the launch before left in
// gtape.dz(blk) -- the slot this block then overwrites with its own dz); out d x_blk -> gtape.dz(blk - 1), block 0: a.dx.
// (One kernel for all four blocks compiled to 500-900 spilled registers once the products were unrolled; four launches cost 3 x 2 us.)
template <int LV, bool BLK0>
__global__ __launch_bounds__(MD_THREADS, 1) void mesh_gcn_bwd_kernel(const PdfMeshLevel a, const int blk) {
    using G = Cfg<LV>;
    constexpr int C = G::C, CIN = 2 * C;
    extern __shared__ float smem[];
    Ctx<LV> c;
    ctx_init<LV>(c, a, smem);
    const int hd = c.hand;
    const TapeOff to = tape_offsets(LV, a.B);
    const GTapeOff go = gtape_offsets(LV, a.B);
    float* gt = a.gtape;
    const long ro = c.row0 * C;
    const int* colT = c.ec; const float* valT = c.ev;
    float* XA = c.XA; float* XB = c.XB;
    ell_stage<LV>(c.ec, c.ev, a.ell_colT[hd], a.ell_valT[hd], a.ell_w);
    load_rows<LV>(XA, gt + (blk == 3 ? go.att(0, 8) : go.dz(blk)) + ro, C, 0);
    __syncthreads();
    gcn_bwd_head<LV>(c, a, blk, to, go);
    const MdGcn& P = a.gcn[blk];
    if constexpr (!BLK0) {
        // fc1 + shortcut backward: d x = dy W1[even] + dz Ws + L^T (dy W1[odd])
        f32x16 acc[G::WMT][G::WNT];
        acc_zero<LV>(acc);
        gemm_nn<LV>(acc, c.g, XB, P.fc1.w[hd], 2 * C, 0, 2);
        gemm_nn<LV>(acc, c.g, XA, P.sc.w[hd], C, 0, 1);
        __syncthreads();
        acc_to_lds<LV>(acc, c.g, XA, nullptr);                          // direct part (dz is no longer needed)
        acc_zero<LV>(acc);
        gemm_nn<LV>(acc, c.g, XB, P.fc1.w[hd], 2 * C, 1, 2);
        __syncthreads();
        acc_to_lds<LV>(acc, c.g, XB, nullptr);
        __syncthreads();
        spmm_add<LV>(XA, XB, colT, valT, a.ell_w);                      // XA = d x of this block = d out of the one before
        __syncthreads();
        MD_STAMP(4, blk * 8 + 6);                                       // fc1 + shortcut backward
        store_rows<LV>(gt + go.dz(blk - 1) + ro, C, 0, XA);
    } else {
        // block 0 (input width 2 C): two column chunks share the operands in XA / XB -- the direct part goes to a.dx, the part L^T still has to
        // see to scratch (L2), and the ELL product finishes the chunk from there
        float* dxp = a.dx + c.row0 * CIN;
        float* T = gt + go.scr() + ro;
#pragma unroll 1
        for (int c0 = 0; c0 < CIN; c0 += C) {
            f32x16 acc[G::WMT][G::WNT];
            acc_zero<LV>(acc);
            gemm_nn<LV>(acc, c.g, XB, P.fc1.w[hd], 2 * CIN, 2 * c0, 2);
            gemm_nn<LV>(acc, c.g, XA, P.sc.w[hd], CIN, c0, 1);
            acc_to_global<LV>(acc, c.g, dxp, CIN, c0);
            acc_zero<LV>(acc);
            gemm_nn<LV>(acc, c.g, XB, P.fc1.w[hd], 2 * CIN, 2 * c0 + 1, 2);
            acc_to_global<LV>(acc, c.g, T, C, 0);
            __syncthreads();
            constexpr int C4 = C / 4;
            for (int i = threadIdx.x; i < G::V * C4; i += MD_THREADS) {
                const int v = i / C4, cc = (i - v * C4) * 4;
                int ci[MD_ELLW]; float vv[MD_ELLW];
#pragma unroll
                for (int w = 0; w < MD_ELLW; ++w) { ci[w] = colT[v * MD_ELLW + w]; vv[w] = valT[v * MD_ELLW + w]; }
                f32x4 s4 = ld4(dxp + (long)v * CIN + c0 + cc);
#pragma unroll
                for (int w = 0; w < MD_ELLW; ++w) s4 += vv[w] * ld4(T + (long)ci[w] * C + cc);
                st4(dxp + (long)v * CIN + c0 + cc, s4);
            }
            __syncthreads();
        }
        MD_STAMP(4, 6);                                                 // block 0: fc1 + shortcut backward
    }
}

// the library's own weight-gradient GEMMs (gemm.hip) and stream wait (elementwise.hip)
extern "C" int pdf_linear_bwd_weight(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats, int M, int N, int K, int ldx, int lddy,
                                     int accumulate, hipStream_t s);
extern "C" int pdf_linear_bwd_weight_pair(const float* x, const float* dy, float* dw0, float* dw1, float* db0, float* db1, float* ws, long ws_floats, int M, int N,
                                          int K, int ldx, int lddy, int accumulate, hipStream_t s);
extern "C" int pdf_stream_wait(hipStream_t waiter, hipStream_t signaler);

template <int LV>
static int mesh_bwd_launch(const PdfMeshLevel& a, hipStream_t s, hipStream_t side) {
    using G = Cfg<LV>;
    const size_t smem = (size_t)2 * G::BUF * sizeof(float), smem2 = smem + (size_t)3 * G::H * G::VP * sizeof(float);
    const size_t smem_g = smem + (size_t)2 * G::V * MD_ELLW * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_att_bwd1_kernel<LV, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) return (int)e;
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_att_bwd1_kernel<LV, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) return (int)e;
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_att_bwd2_kernel<LV, false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2)) return (int)e;
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_att_bwd2_kernel<LV, true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2)) return (int)e;
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_att_bwd2_kernel<LV, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2)) return (int)e;
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_att_bwd2_kernel<LV, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2)) return (int)e;
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_att_bwd2_kernel<LV, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2)) return (int)e;
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_att_bwd2_kernel<LV, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2)) return (int)e;
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_gcn_bwd_kernel<LV, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_g)) return (int)e;
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_gcn_bwd_kernel<LV, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_g)) return (int)e;
        attr_done = true;
    }
    const dim3 grid(2 * a.B), blk(MD_THREADS);
    hipLaunchKernelGGL((mesh_att_bwd1_kernel<LV, true>), grid, blk, smem, s, a);
    PDF_LAUNCH_CHECK();
    static const int att_split = getenv("PDF_MESH_ATT_SPLIT") ? atoi(getenv("PDF_MESH_ATT_SPLIT")) : 7;      // bit LV: the split form at that level
    const bool split = (att_split >> LV) & 1;
    // workgroups per attention pass: the (head, tile) items are 8 / 16 / 32 at level 0 / 1 / 2, a workgroup's waves take one each at a time (md_active_waves)
    static const int att_parts_env = getenv("PDF_MESH_ATT_PARTS") ? atoi(getenv("PDF_MESH_ATT_PARTS")) : 0;      // e.g. 124 = 1 / 2 / 4 at level 0 / 1 / 2
    const int parts_default[3] = {2, 2, 4};                 // (measured, profiles/r05_att_split.txt: 1-1-1 49.4, 1-2-2 49.4, 1-2-4 49.3, 2-2-4 49.05, 2-4-8 49.4 ms per step; one workgroup for everything: 50.0)
    int parts = att_parts_env > 0 ? (LV == 0 ? att_parts_env / 100 : LV == 1 ? (att_parts_env / 10) % 10 : att_parts_env % 10) : parts_default[LV];
    parts = max(1, min(parts, G::H * G::MT / 4));
    if (split) {
        hipLaunchKernelGGL((mesh_att_bwd2_kernel<LV, true, 1>), dim3(2 * a.B, 2 * parts), blk, smem2, s, a, parts);
        hipLaunchKernelGGL((mesh_att_bwd2_kernel<LV, true, 2>), grid, blk, smem2, s, a, 1);
    } else hipLaunchKernelGGL((mesh_att_bwd2_kernel<LV, true, 0>), grid, blk, smem2, s, a, 1);
    PDF_LAUNCH_CHECK();
    hipLaunchKernelGGL((mesh_att_bwd1_kernel<LV, false>), grid, blk, smem, s, a);
    PDF_LAUNCH_CHECK();
    if (split) {
        hipLaunchKernelGGL((mesh_att_bwd2_kernel<LV, false, 1>), dim3(2 * a.B, 2 * parts), blk, smem2, s, a, parts);
        hipLaunchKernelGGL((mesh_att_bwd2_kernel<LV, false, 2>), grid, blk, smem2, s, a, 1);
    } else hipLaunchKernelGGL((mesh_att_bwd2_kernel<LV, false, 0>), grid, blk, smem2, s, a, 1);
    PDF_LAUNCH_CHECK();
    for (int b = 3; b >= 1; --b) {
        hipLaunchKernelGGL((mesh_gcn_bwd_kernel<LV, false>), grid, blk, smem_g, s, a, b);
        PDF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((mesh_gcn_bwd_kernel<LV, true>), grid, blk, smem_g, s, a, 0);
    PDF_LAUNCH_CHECK();
    // ---- weight gradients: dW += dY^T A, db += colsum(dY) with the library's GEMMs, on the side stream (nothing on the chain reads them)
    if (side != s) { if (int rc = pdf_stream_wait(side, s)) return rc; }
    constexpr int C = G::C;
    const int M = a.B * G::V;                                           // rows per hand
    const TapeOff to = tape_offsets(LV, a.B);
    const GTapeOff go = gtape_offsets(LV, a.B);
    const float* tape = a.tape;
    const float* gt = a.gtape;
#define MD_WG_PAIR(X, LDX, DY, GL, K_)                                                                                                     \
    if ((GL).w[0] != nullptr) {                                                                                                            \
        if (int rc = pdf_linear_bwd_weight_pair((X), (DY), (GL).w[0], (GL).w[1], (GL).b[0], (GL).b[1], a.wg_ws, a.wg_ws_floats, M, C, (K_), (LDX), C, 1, side)) \
            return rc;                                                                                                                     \
    }
    for (int i = 0; i < 4; ++i) {
        const int cin = i == 0 ? a.cin0 : C;
        const float* xi = i == 0 ? a.x : tape + to.out(i - 1);
        MD_WG_PAIR(tape + to.cat1(i), 2 * cin, gt + go.dy(i), a.ggcn[i].fc1, 2 * cin)
        MD_WG_PAIR(tape + to.cat2(i), 2 * C, gt + go.dy2(i), a.ggcn[i].fc2, 2 * C)
        MD_WG_PAIR(xi, cin, gt + go.dz(i), a.ggcn[i].sc, cin)
    }
    for (int A = 0; A < 2; ++A) {
        const MdAttnG& GP = A ? a.gcross : a.gself;
        const float* hn = tape + to.h(A);
        if (A == 0) {
            MD_WG_PAIR(hn, C, gt + go.att(A, 3), GP.q, C)
            MD_WG_PAIR(hn, C, gt + go.att(A, 4), GP.k, C)
            MD_WG_PAIR(hn, C, gt + go.att(A, 5), GP.v, C)
            MD_WG_PAIR(tape + to.a_(A), C, gt + go.att(A, 2), GP.fc, C)
        } else {
            // the cross-hand projections are shared by both hands (inter_attn.py:82-108): one product over all 2 M rows
            const MdLinG* gl[4] = {&GP.q, &GP.k, &GP.v, &GP.fc};
            const float* xs[4] = {hn, hn, hn, tape + to.a_(A)};
            const int slot[4] = {3, 4, 5, 2};
            for (int t = 0; t < 4; ++t)
                if (gl[t]->w[0] != nullptr)
                    if (int rc = pdf_linear_bwd_weight(xs[t], gt + go.att(A, slot[t]), gl[t]->w[0], gl[t]->b[0], a.wg_ws, a.wg_ws_floats, 2 * M, C, C, C, C, 1, side)) return rc;
        }
        MD_WG_PAIR(tape + to.hn(A), C, gt + go.att(A, 1), GP.f1, C)
        MD_WG_PAIR(tape + to.t(A), C, gt + go.att(A, 0), GP.f2, C)
    }
#undef MD_WG_PAIR
    return 0;
}

// One DualGraphLayer backward: five launches on `stream`, then the layer's 28 weight-gradient GEMMs on `side_stream` (which first waits for `stream`).
// Every parameter gradient is ACCUMULATED into its g* pointer (NULL weight pointer: that Linear's gradient is skipped).
PDF_API int pdf_mesh_level_bwd(const PdfMeshLevel* a, hipStream_t stream, hipStream_t side_stream) {
    if (int rc = mesh_check(a)) return rc;
    if (a->dout == nullptr || a->dx == nullptr || a->gtape == nullptr || a->wg_ws == nullptr || !a->training) return PDF_E_BADARG;
    switch (a->level) {
        case 0: return mesh_bwd_launch<0>(*a, stream, side_stream);
        case 1: return mesh_bwd_launch<1>(*a, stream, side_stream);
        default: return mesh_bwd_launch<2>(*a, stream, side_stream);
    }
}

#if !MD_BF16
// diagnostic build only: copies the stage stamps of the last launches to out[8 * 3 * 64]; returns 0 when the library was built without them
PDF_API int pdf_debug_mesh_stamps(unsigned long long* out) {
#if MD_STAMPS
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_md_stamps), sizeof(unsigned long long) * 8 * 3 * 64) != hipSuccess) return 0;
    return 1;
#else
    (void)out;
    return 0;
#endif
}
#else
}   // namespace md_bf16_build / md_x3_build
#endif
