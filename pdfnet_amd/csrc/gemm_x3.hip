// gemm_x3: fp32 products on the bf16 matrix pipe ("x3" arithmetic).
//
// gfx950 has no TF32 and its fp32 MFMA runs at the fp32 VECTOR rate, 1/16 of the bf16 MFMA rate (MI355X_MICROARCH.md "Matrix cores").
// A fp32 value a is EXACTLY the sum of three bf16 values,  a = h + m + l  with  h = bf16(a), m = bf16(a - h), l = bf16(a - h - m)
// (8 + 8 + 8 significand bits; each residual is exact in fp32), so a fp32 product is
//     a b = hh' + (hm' + mh') + (mm' + hl' + lh') + (ml' + lm' + ll'),
// every partial product exact in the fp32 accumulator's input (8 x 8 bits), the last group below 2^-24 |a b|.  Six bf16 MFMAs (the first
// three groups) reproduce the fp32 product to ~2^-24 relative -- the same order as the one rounding the native v_mfma_f32_32x32x2_f32
// makes per product -- at 6 / 16 of its matrix-pipe time.  Measured error against float64 beside the native kernel: tools/x3_bench.py.
//
// Operands come PRE-SPLIT ("x3 planes"): three bf16 tensors of the operand's shape, component c at base + c * cs elements.  The producers
// that already stream the data write them (the Winograd transforms, winograd.hip); pdf_x3_split converts a plain fp32 tensor.
//
//   x3gemm_nt   C_b[M][N] = A_b[M][K] B_b[N][K]^T      both operands K-contiguous (transform-domain forward / backward-data products)
//   x3gemm_tn   W_b[I][J] = sum_m P_b[m][I] Q_b[m][J]   both operands reduction-major (transform-domain weight gradient), split over m
//
// Both kernels: LDS-DMA ring (buffer_load_dwordx4 ... lds, gemm_dma.hip), one [rows][4 chunks] (NT) or [k][columns] (TN) image per
// component, conflict-free fragment reads by XOR-swizzling WHICH global chunk a DMA lane fetches.
#include "gemm_common.h"
#include <cstdio>

typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef short s16x4v __attribute__((ext_vector_type(4)));
typedef short s16x8v __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 x3_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r = {(int)(unsigned)(a & 0xffffffffu), (int)(unsigned)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
    return r;
}
// (inline assembly: through the builtin the compiler drains the ring in front of every LDS read -- see gemm_dma.hip)
__device__ __forceinline__ void x3_dma16(const i32x4& rsrc, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}

// ---- fp32 -> x3 planes: 8 elements per thread, three 16-byte stores
__global__ __launch_bounds__(256) void x3_split_kernel(const float* __restrict__ x, unsigned short* __restrict__ o, long n8, long cs) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const float4 u = reinterpret_cast<const float4*>(x)[2 * i], v = reinterpret_cast<const float4*>(x)[2 * i + 1];
        uint4 h, m, l;
        pdf_x3_split2(u.x, u.y, h.x, m.x, l.x); pdf_x3_split2(u.z, u.w, h.y, m.y, l.y);
        pdf_x3_split2(v.x, v.y, h.z, m.z, l.z); pdf_x3_split2(v.z, v.w, h.w, m.w, l.w);
        *reinterpret_cast<uint4*>(o + 8 * i) = h;
        *reinterpret_cast<uint4*>(o + cs + 8 * i) = m;
        *reinterpret_cast<uint4*>(o + 2 * cs + 8 * i) = l;
    }
}
// the same for a matrix [R][C] (row stride ldx) into planes with their own row stride ldo (a padded stride keeps 2^k-byte rows off one memory channel)
__global__ __launch_bounds__(256) void x3_split_rows_kernel(const float* __restrict__ x, long ldx, unsigned short* __restrict__ o, long ldo, long cs, long R, int C8) {
    const long total = R * C8;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / C8;
        const int c = (int)(i - r * C8) * 8;
        const float4 u = *reinterpret_cast<const float4*>(x + r * ldx + c), v = *reinterpret_cast<const float4*>(x + r * ldx + c + 4);
        uint4 h, m, l;
        pdf_x3_split2(u.x, u.y, h.x, m.x, l.x); pdf_x3_split2(u.z, u.w, h.y, m.y, l.y);
        pdf_x3_split2(v.x, v.y, h.z, m.z, l.z); pdf_x3_split2(v.z, v.w, h.w, m.w, l.w);
        unsigned short* d = o + r * ldo + c;
        *reinterpret_cast<uint4*>(d) = h;
        *reinterpret_cast<uint4*>(d + cs) = m;
        *reinterpret_cast<uint4*>(d + 2 * cs) = l;
    }
}
PDF_API int pdf_x3_split(const float* x, void* out, long n, long cs, hipStream_t s) {
    if (n % 8 != 0 || cs % 8 != 0 || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return PDF_E_BADARG;
    hipLaunchKernelGGL(x3_split_kernel, dim3(grid_for(n / 8, 256, 256 * 16)), dim3(256), 0, s, x, (unsigned short*)out, n / 8, cs);
    PDF_LAUNCH_CHECK();
    return 0;
}

// diagnostic build (-DX3_STAMPS=1): wave 0 of block 0 sums the shader clocks it spends (a) parked at the top-of-loop wait + barrier, (b) from there to
// its first MFMA (fragment reads), (c) in the MFMA / DMA slots, and stamps s_memtime / s_memrealtime around the K loop (the clock the chip holds)
#ifndef X3_STAMPS
#define X3_STAMPS 0
#endif
#if X3_STAMPS
__device__ unsigned long long g_x3_stamps[16];
#define X3_T() __builtin_readcyclecounter()
#endif
struct X3Gemm {
    const unsigned short* A; const unsigned short* B; float* C;
    long csA, csB;                                      // component strides (elements)
    long gsA, gsB, gsC;                                 // batch strides (elements of A / B, floats of C)
    int M, N, K, batch, ldc;
    int lda, ldb;                                       // NT: row strides of A / B in elements (>= K, multiples of 8)
    int col_major_tiles;                                // NT: tile order (see the kernel)
    // NT epilogue of a transposed convolution with kernel == stride (pixel shuffle): row m = pixel (ni, qy, qx) of a QH x QW map, column n = (tap, co)
    // with ps_cout channels per tap and ps_kw taps per kernel row -> y[ni][qy * ps_s + ky][qx * ps_s + kx][co] (row stride ldc) + bias[co].  ps_cout == 0: plain C
    int ps_cout, ps_kw, ps_s, QH, QW, OH, OW;
    int ooy, oox;                                       // ... output pixel offset (one parity class of a strided transposed convolution)
    const float* bias;
    // NT, implicit A (template IM): row m = position (ni, qy, qx) of the QH x QW grid, K = T taps x Cc channels; element (m, t, c) is
    // A[ni][qy sy + tdy[t]][qx sx + tdx[t]][c] of an AH x AW map with pixel stride lda, zero outside the map (masked lanes read the zero row at
    // element offset zrow of every component plane, which the pre-pass appends).  Cc % 32 == 0.
    int AH, AW, sy, sx, T, Cc;
    long zrow;
    int tdy[16], tdx[16];
    // TN, implicit Q (template IQ): row m of the reduction = position (ni, iy, ix) of a QH x QW grid (both powers of two: lw = log2 QW, lhw = log2 QH QW);
    // column j = (tap, c): Q[m][j] = B[ni][iy sy + tdy[tap]][ix sx + tdx[tap]][c] of an AH x AW map with pixel stride ldb, zero outside (zero row at zrow)
    int lw, lhw;
    int rows_per_split, splits;                         // TN: rows of the reduction per block (multiple of 32)
    int accum;                                          // TN, one split: W += the product (a gradient accumulated into an existing one)
};

// the NPROD products of one (A fragment set, B fragment set) pair as (component of A, component of B), smallest terms first
// X3_SPLIT_ACC = 1: the leading product hh' goes to one accumulator set and the five small ones (2^-8 .. 2^-16 of it) to a second, added once after
// the K loop.  The fp32 accumulator that carries the full magnitude then takes ONE rounding per 16 k instead of six (the small set's roundings are
// 2^-8 of those): rms error against float64 0.8-0.9x -> ~0.4x the native fp32-MFMA kernel's (which rounds once per 2 k) -- tests/test_x3_gpu.py.
#ifndef X3_SPLIT_ACC
#define X3_SPLIT_ACC 1
#endif
template <int NPROD> struct X3Prod;
template <> struct X3Prod<3> { static constexpr int a[3] = {1, 0, 0}, b[3] = {0, 1, 0}; };
template <> struct X3Prod<6> { static constexpr int a[6] = {2, 0, 1, 1, 0, 0}, b[6] = {0, 2, 1, 0, 1, 0}; };
template <> struct X3Prod<9> { static constexpr int a[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, b[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0}; };

// the linear tile list of a batched launch, each XCD a contiguous chunk (blocks b and b + 8 share an XCD; bijective form, guide T1)
__device__ __forceinline__ int x3_xcd_lin(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
// one LDS-DMA instruction: lane l fetches 16 bytes at voff + soff (soff scalar: the K-step's advance costs no vector instruction)
__device__ __forceinline__ void x3_dma16s(const i32x4& rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// Schedule shared by both kernels (round 6, profiles/r06_x3_sq.txt).  A K-step of 32 is NG = 2 TM TN accumulator-tile visits of NPROD MFMAs
// each.  Issued as "all DMA, all LDS reads, then all MFMAs" the kernel ran at load time PLUS matrix time (the matrix pipe drains 32 cycles
// after the last MFMA while the wave issues ~60-cycle DMA instructions): 45 % MFMA busy.  So the stream is cut into SLOTS of two MFMAs and
// every slot carries at most one DMA instruction of the next tile and one fragment read of the second k-sub-step; the K advance of a DMA is
// a scalar offset and rows past the matrix edge are CLAMPED to the last row (their products land in rows / columns the epilogue never
// stores), so a DMA costs no vector instruction at all.

// ------------------------------------------------------------------------------------------------------------------------------
// x3gemm_nt.  Block = WM x WN waves, each TM x TN accumulator tiles of 32 x 32; K-step 32: a row of a component image is 64 bytes = 4
// chunks, chunk (r, c) at slot r * 4 + (c ^ ((r >> 2) & 3)) -- the 16 lanes of a ds_read_b128 group then hit 16 different 16-byte bank
// groups (as igemm_bf16_dma, CPR = 4).  Lane l of an MFMA operand holds 8 consecutive k of row l & 31 at k-offset 8 (l >> 5).
template <int WM, int WN, int TM, int TN, int ST, int NPROD, bool IM = false>
__global__ __launch_bounds__(WM * WN * 64) void x3gemm_nt(const X3Gemm g) {
    constexpr int NW = WM * WN, BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int AIMG = BM * 64, BIMG = BN * 64, STAGE = 3 * (AIMG + BIMG);
    constexpr int NIA = 3 * BM / 16 / NW, NIB = 3 * BN / 16 / NW;         // DMA instructions per wave and K-step
    static_assert(3 * BM / 16 % NW == 0 && 3 * BN / 16 % NW == 0, "tile / wave count");
    __shared__ __attribute__((aligned(16))) unsigned char smem[ST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN, per = ntm * ntn;
    const int lin = x3_xcd_lin(blockIdx.x, per * g.batch);
    const int b = lin / per, rem = lin - b * per;
    int tmi, tni;
    if (g.col_major_tiles) { tni = rem / ntm; tmi = rem - tni * ntm; }     // consecutive blocks share a B panel (few row tiles, many column tiles)
    else { tmi = rem / ntn; tni = rem - tmi * ntn; }
    const int m0 = tmi * BM, n0 = tni * BN;
    const unsigned short* Ab = g.A + (long)b * g.gsA;
    const unsigned short* Bb = g.B + (long)b * g.gsB;
    float* __restrict__ Cb = g.C + (long)b * g.gsC;
    // (one descriptor per operand spans its three components: the host checks 2 cs + rows K < 2^31 elements)
    const i32x4 rsA = x3_rsrc(Ab, 0xfffffff0u), rsB = x3_rsrc(Bb, 0xfffffff0u);
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const unsigned char*)smem;
    auto swz = [](int r) { return (r >> 2) & 3; };

    unsigned aoff[NIA], boff[NIB], alds[NIA], blds[NIB];
    int aiy0[IM ? NIA : 1], aix0[IM ? NIA : 1], anb[IM ? NIA : 1]; unsigned acomp[IM ? NIA : 1], azero[IM ? NIA : 1];
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        const int gi = wave * NIA + i, comp = gi / (BM / 16), rb = gi - comp * (BM / 16);
        const int row = rb * 16 + (lane >> 2);
        const int kc = (lane & 3) ^ swz(row);
        alds[i] = (unsigned)(comp * AIMG + rb * 1024);
        aoff[i] = (unsigned)((comp * g.csA + (long)min(m0 + row, g.M - 1) * g.lda + kc * 8) * 2);
        if constexpr (IM) {
            const int r = min(m0 + row, g.M - 1), hw = g.QH * g.QW, ni = r / hw, rem2 = r - ni * hw, qy = rem2 / g.QW, qx = rem2 - qy * g.QW;
            aiy0[i] = qy * g.sy; aix0[i] = qx * g.sx; anb[i] = ni * g.AH * g.AW;
            acomp[i] = (unsigned)((comp * g.csA + kc * 8) * 2);
            azero[i] = (unsigned)((comp * g.csA + g.zrow + kc * 8) * 2);
        }
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const int gi = wave * NIB + i, comp = gi / (BN / 16), rb = gi - comp * (BN / 16);
        const int row = rb * 16 + (lane >> 2);
        const int kc = (lane & 3) ^ swz(row);
        blds[i] = (unsigned)(3 * AIMG + comp * BIMG + rb * 1024);
        boff[i] = (unsigned)((comp * g.csB + (long)min(n0 + row, g.N - 1) * g.ldb + kc * 8) * 2);
    }
    const int nk = g.K / 32;
    int nissued = 0;
    unsigned ko = 0, koa = 0, sbase = lds0;
    int cur_tap = -1;
    auto piece = [&](int p) {
        if (p < NIA) x3_dma16s(rsA, sbase + alds[p], aoff[p], IM ? koa : ko);
        else x3_dma16s(rsB, sbase + blds[p - NIA], boff[p - NIA], ko);
    };
    auto begin_issue = [&](int stage) {
        const int kt = min(nissued, nk - 1);            // (a tile past the end re-reads the last one into a stage nobody consumes: uniform vmcnt counts)
        ko = (unsigned)kt * 64u;
        ++nissued;
        sbase = lds0 + (unsigned)(stage * STAGE);
        if constexpr (IM) {
            const int tap = kt * 32 / g.Cc;
            koa = (unsigned)(kt * 32 - tap * g.Cc) * 2u;
            if (tap != cur_tap) {                         // every Cc / 32 K-steps: this lane's pixel of the new tap, or the zero row
                cur_tap = tap;
                const int dy = g.tdy[tap], dx = g.tdx[tap];
#pragma unroll
                for (int i = 0; i < NIA; ++i) {
                    const int iy = aiy0[i] + dy, ix = aix0[i] + dx;
                    const bool ok = ((unsigned)iy < (unsigned)g.AH) & ((unsigned)ix < (unsigned)g.AW);
                    aoff[i] = ok ? acomp[i] + (unsigned)(anb[i] + iy * g.AW + ix) * (unsigned)(g.lda * 2) : azero[i];
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#if X3_SPLIT_ACC
    f32x16 acc2[TM][TN];                                     // the small products (see X3_SPLIT_ACC)
#endif
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int c = 0; c < TN; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[a][c][r] = 0.f;
#if X3_SPLIT_ACC
                acc2[a][c][r] = 0.f;
#endif
            }

    const int h = lane >> 5;
    int arow[TM], acx[TM], brow[TN], bcx[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int r = (wm * TM + i) * 32 + (lane & 31); arow[i] = r * 64; acx[i] = h ^ swz(r); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int r = (wn * TN + j) * 32 + (lane & 31); brow[j] = 3 * AIMG + r * 64; bcx[j] = h ^ swz(r); }

#pragma unroll
    for (int p = 0; p < ST - 1; ++p) {
        begin_issue(p);
#pragma unroll
        for (int q = 0; q < NIA + NIB; ++q) piece(q);
    }
    int st = 0, stn = ST - 1;
    constexpr int NG = 2 * TM * TN, NP = NIA + NIB, NR = (TM + TN) * 3;      // tile visits, DMA pieces, fragment reads of one k-sub-step
    constexpr int NSLOT = NG * NPROD / 2;
    constexpr int RSL = NSLOT / 2 - 1 > 0 ? NSLOT / 2 - 1 : 1;              // slots that carry the second sub-step's fragment reads (one-tile waves with three products: all in slot 0)
    using PR = X3Prod<NPROD>;
#if X3_STAMPS
    unsigned long long tw = 0, tr = 0, tm = 0, t0 = X3_T(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int kt = 0; kt < nk; ++kt) {
#if X3_STAMPS
        const unsigned long long s0 = X3_T();
#endif
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * (NIA + NIB)) : "memory");
        asm volatile("s_barrier" ::: "memory");
#if X3_STAMPS
        const unsigned long long s1 = X3_T();
#endif
        begin_issue(stn);
        const unsigned char* sp = smem + st * STAGE;
        bf16x8v af[2][TM][3], bf[2][TN][3];
        auto read1 = [&](int ks, int q) {                    // fragment read q < NR of k-sub-step ks
            if (q < TM * 3) { const int i = q / 3, c = q % 3; af[ks][i][c] = *reinterpret_cast<const bf16x8v*>(sp + c * AIMG + arow[i] + ((acx[i] ^ (2 * ks)) << 4)); }
            else { const int j = (q - TM * 3) / 3, c = (q - TM * 3) % 3; bf[ks][j][c] = *reinterpret_cast<const bf16x8v*>(sp + c * BIMG + brow[j] + ((bcx[j] ^ (2 * ks)) << 4)); }
        };
#pragma unroll
        for (int q = 0; q < NR; ++q) read1(0, q);
        __builtin_amdgcn_sched_barrier(0);
#if X3_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long s2 = X3_T();
        tw += s1 - s0; tr += s2 - s1;
#endif
#pragma unroll
        for (int q = 0; q < NG * NPROD; ++q) {
            const int gi = q / NPROD, pr = q % NPROD;
            const int ks = gi / (TM * TN), i = (gi / TN) % TM, j = gi % TN;
#if X3_SPLIT_ACC
            if (PR::a[pr] + PR::b[pr] != 0) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i][PR::a[pr]], bf[ks][j][PR::b[pr]], acc2[i][j], 0, 0, 0);
            else
#endif
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i][PR::a[pr]], bf[ks][j][PR::b[pr]], acc[i][j], 0, 0, 0);
            if (q % 2 == 1 || q == NG * NPROD - 1) {
                const int slot = q / 2;
                __builtin_amdgcn_sched_barrier(0);
                // second sub-step's reads in the first slots (they are needed from slot NSLOT / 2 on), the DMA pieces on every other slot
#pragma unroll
                for (int r = slot * NR / RSL; r < (slot + 1) * NR / RSL && r < NR; ++r)
                    if (slot < RSL) read1(1, r);
                if (slot % 2 == 0 && slot / 2 < NP) piece(slot / 2);
                if (slot == NSLOT - 1) {
#pragma unroll
                    for (int p = (NSLOT + 1) / 2; p < NP; ++p) piece(p);     // (more pieces than slots: the rest at the end)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#if X3_STAMPS
        tm += X3_T() - s2;
#endif
        st = st == ST - 1 ? 0 : st + 1;
        stn = stn == ST - 1 ? 0 : stn + 1;
    }
#if X3_STAMPS
    if (blockIdx.x == 0 && tid == 0) {
        g_x3_stamps[0] = tw; g_x3_stamps[1] = tr; g_x3_stamps[2] = tm; g_x3_stamps[3] = X3_T() - t0;
        g_x3_stamps[4] = __builtin_amdgcn_s_memrealtime() - rt0; g_x3_stamps[5] = nk;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#if X3_SPLIT_ACC
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int c = 0; c < TN; ++c) acc[a][c] += acc2[a][c];
#endif

    if (g.ps_cout > 0) {                                     // pixel-shuffled output (+ bias): 32 lanes of a store = 32 consecutive channels of one output pixel
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
            const bool cok = col < g.N;
            const int tap = col / g.ps_cout, co = col - tap * g.ps_cout;
            const int ky = tap / g.ps_kw, kx = tap - ky * g.ps_kw;
            const float bv = (g.bias != nullptr && cok) ? g.bias[co] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
                    if (cok && row < g.M) {
                        const int hw = g.QH * g.QW, ni = row / hw, rem2 = row - ni * hw, qy = rem2 / g.QW, qx = rem2 - qy * g.QW;
                        Cb[(((long)ni * g.OH + qy * g.ps_s + ky + g.ooy) * g.OW + qx * g.ps_s + kx + g.oox) * g.ldc + co] = acc[i][j][r] + bv;
                    }
                }
        }
        return;
    }
    const bool whole = m0 + BM <= g.M && (double)g.M * g.ldc * 4.0 < 4294967000.0;
    const auto rsC = __builtin_amdgcn_make_buffer_rsrc((void*)Cb, 0, whole ? (unsigned)g.M * (unsigned)g.ldc * 4u : 0u, 0x00020000);
    const unsigned ldc4 = (unsigned)g.ldc * 4u;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
        const bool cok = col < g.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r0 = m0 + (wm * TM + i) * 32 + 4 * h;
            if (whole) {
                const unsigned vo = cok ? (unsigned)(r0 * g.ldc + col) * 4u : 0xffffffffu;
                float v[16];                                   // (a copy first: __builtin_bit_cast on acc[i][j][r] itself stored element 0 sixteen times, hipcc 7.2)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), rsC, vo, ((r & 3) + 8 * (r >> 2)) * ldc4, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = r0 + (r & 3) + 8 * (r >> 2);
                    if (cok && row < g.M) Cb[(long)row * g.ldc + col] = acc[i][j][r];
                }
            }
        }
    }
}

// tile choice (profiles/r06_x3_kernel_ab.txt, ms with 256 x 128 / 8 waves : 128 x 128 / 8 waves): a narrow output over a long reduction runs best on
// the 128 x 128 tile with the 3-stage ring (`feat` forward, N = 256, K = 1,024: 0.95 : 0.84; p3 forward 0.80 : 0.78); wide outputs re-read less B
// per flop on the 256 x 128 tile (p5 forward, N = 16,384: 0.76 : 0.84; p3 backward-data 0.66 : 0.72) and short reductions are prologue-bound and
// prefer its fewer blocks (256 -> 256 at 64x64: 0.245 : 0.267) -- as long as those still fill the chip twice over
static int x3_nt_variant(long M, long N, long K, long batch) {
    static const int rule = getenv("PDF_X3_TILE_RULE") ? atoi(getenv("PDF_X3_TILE_RULE")) : 2;      // (A/B switch: 0 = round-6 first rule, 1 = K >= 512 -> 8-wave 128 x 128)
    const bool fills = (long)cdiv(M, 256) * cdiv(N, 128) * batch >= 512;
    if (rule == 0) return fills ? 0 : 1;
    if (rule == 1) return K >= 512 ? 5 : (fills ? 0 : 5);
    if (N <= 256 && K >= 1024) return 5;
    return fills ? 0 : 5;
}
template <int WM, int WN, int TM, int TN, int ST>
static int x3_launch_nt(const X3Gemm& g, int nprod, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const dim3 grid((unsigned)(cdiv(g.M, BM) * cdiv(g.N, BN) * g.batch));
    char nm[96];
    snprintf(nm, sizeof nm, g.T > 0 ? "x3gemm_nt<%d, %d, %d, %d, %d, %d, true>" : "x3gemm_nt<%d, %d, %d, %d, %d, %d>", WM, WN, TM, TN, ST, g.T > 0 ? 6 : nprod);
    KTimer kt(nm, 2.0 * g.batch * g.M * g.N * g.K, g.batch * (6.0 * ((double)g.M + g.N) * g.K + 4.0 * g.M * g.N), s);
    if (g.T > 0) hipLaunchKernelGGL((x3gemm_nt<WM, WN, TM, TN, ST, 6, true>), grid, dim3(WM * WN * 64), 0, s, g);       // implicit A (six products only)
    else if (nprod == 9) hipLaunchKernelGGL((x3gemm_nt<WM, WN, TM, TN, ST, 9>), grid, dim3(WM * WN * 64), 0, s, g);
    else if (nprod == 3) hipLaunchKernelGGL((x3gemm_nt<WM, WN, TM, TN, ST, 3>), grid, dim3(WM * WN * 64), 0, s, g);
    else hipLaunchKernelGGL((x3gemm_nt<WM, WN, TM, TN, ST, 6>), grid, dim3(WM * WN * 64), 0, s, g);
    PDF_LAUNCH_CHECK();
    return 0;
}

// C_b [M][N] (row stride N) = A_b [M][K] B_b [N][K]^T for b < batch, operands as x3 planes.  K % 32 == 0.
// variant: 0 = 256 x 128 tile, 8 waves;  1 = 128 x 128, 4 waves, 3-stage ring;  2 = 128 x 64, 4 waves;  nprod: 6 (default) | 9 | 3
int pdf_internal_x3_batched_gemm(const void* A3, long csA, const void* B3, long csB, float* C, int batch, long gsA, long gsB, long gsC,
                                 int M, int N, int K, int variant, int nprod, hipStream_t s) {
    if (K % 32 != 0 || K < 32 || M < 1 || N < 1 || batch < 1) return PDF_E_BADARG;
    if (2.0 * csA + (double)M * K >= 2147483000.0 || 2.0 * csB + (double)N * K >= 2147483000.0) return PDF_E_BADARG;      // 32-bit byte offsets
    X3Gemm g = {};
    g.A = (const unsigned short*)A3; g.B = (const unsigned short*)B3; g.C = C;
    g.csA = csA; g.csB = csB; g.gsA = gsA; g.gsB = gsB; g.gsC = gsC;
    g.M = M; g.N = N; g.K = K; g.batch = batch; g.ldc = N; g.lda = K; g.ldb = K;
    g.col_major_tiles = cdiv(N, 128) > 4 * cdiv(M, 128);
    if (variant < 0) variant = x3_nt_variant(M, N, K, batch);
    if (variant == 1) return x3_launch_nt<2, 2, 2, 2, 3>(g, nprod, s);
    if (variant == 2) return x3_launch_nt<2, 2, 2, 1, 3>(g, nprod, s);
    if (variant == 3) return x3_launch_nt<2, 2, 1, 1, 2>(g, nprod, s);     // 64 x 64, 48 KB of LDS
    if (variant == 4) return x3_launch_nt<4, 2, 1, 2, 3>(g, nprod, s);     // 128 x 128 on 8 waves (two per SIMD), 3-stage ring
    if (variant == 5) return x3_launch_nt<2, 4, 2, 1, 3>(g, nprod, s);     // the same, waves 2 x 4
    return x3_launch_nt<4, 2, 2, 2, 2>(g, nprod, s);
}
PDF_API int pdf_x3_batched_gemm_nt(const void* A3, long csA, const void* B3, long csB, float* C, int batch, long gsA, long gsB, long gsC,
                                   int M, int N, int K, int variant, int nprod, hipStream_t s) {
    return pdf_internal_x3_batched_gemm(A3, csA, B3, csB, C, batch, gsA, gsB, gsC, M, N, K, variant, nprod, s);
}
// diagnostic builds (-DX3_STAMPS=1) only: the six stamps of the last x3gemm_nt launch (see X3_STAMPS); returns 0 otherwise
PDF_API int pdf_debug_x3_stamps(unsigned long long* out) {
#if X3_STAMPS
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x3_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 16 : 0;
#else
    (void)out; return 0;
#endif
}
int pdf_internal_batched_gemm(const float* A, const float* B, float* C, int batch, long gsA, long gsB, long gsC, int M, int N, int K, hipStream_t s);
// the native fp32-MFMA batched product (gemm.hip) behind the same shape of call: the comparison arm of tools/x3_bench.py
PDF_API int pdf_batched_gemm_nt(const float* A, const float* B, float* C, int batch, long gsA, long gsB, long gsC, int M, int N, int K, hipStream_t s) {
    return pdf_internal_batched_gemm(A, B, C, batch, gsA, gsB, gsC, M, N, K, s);
}

// ------------------------------------------------------------------------------------------------------------------------------
// x3gemm_tn.  The reduction index m is the slow axis of both operands, so a component image is [32 k-rows][BI columns] exactly as it
// lies in HBM and the MFMA operand -- 8 consecutive k of ONE column per lane -- comes out of two transposing reads (ds_read_b64_tr_b16:
// the 16 lanes of a group read 4 k-rows x 16 columns and receive them column-major; layout as gemm_bf16.hip frag_col).  The 32 lanes of
// a tr-read service group touch 4 k-rows x 64 bytes: the 64-byte blocks of a row are XOR-swizzled by the row (k & 3; rows of 128 bytes:
// (k >> 1) & 1) so that they fall on 4 different quarters of the 256-byte bank line -- again by choosing which global chunk a DMA lane
// fetches.  Rows [split * rows_per_split, ...) of the reduction per block; the caller sums the splits.
template <int WM, int WN, int TM, int TN, int ST, int NPROD, bool IQ = false>
__global__ __launch_bounds__(WM * WN * 64) void x3gemm_tn(const X3Gemm g) {
    constexpr int NW = WM * WN, BI = WM * TM * 32, BJ = WN * TN * 32;
    constexpr int PIMG = 32 * BI * 2, QIMG = 32 * BJ * 2, STAGE = 3 * (PIMG + QIMG);
    constexpr int CP = BI / 8, CQ = BJ / 8;                               // 16-byte chunks per k-row
    constexpr int NIP = 3 * 32 * CP / 64 / NW, NIQ = 3 * 32 * CQ / 64 / NW;
    static_assert((3 * 32 * CP / 64) % NW == 0 && (3 * 32 * CQ / 64) % NW == 0, "tile / wave count");
    __shared__ __attribute__((aligned(16))) unsigned char smem[ST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int NI = g.M, NJ = g.N;                                         // (descriptor reuse: M = NI, N = NJ, K = rows of the reduction, % 32 == 0)
    const int ldp = g.lda > 0 ? g.lda : NI, ldq = g.ldb > 0 ? g.ldb : NJ;      // row strides of P / Q (elements)
    const int nti = (NI + BI - 1) / BI, ntj = (NJ + BJ - 1) / BJ, tiles = nti * ntj;
    const int lin = x3_xcd_lin(blockIdx.x, tiles * g.splits * g.batch);
    const int grp = lin / tiles, tile = lin - grp * tiles;
    const int b = grp / g.splits, split = grp - b * g.splits;
    const int ti = tile / ntj, tj = tile - ti * ntj;
    const int i0 = ti * BI, j0 = tj * BJ;
    const int ms = split * g.rows_per_split, me = min(g.K, ms + g.rows_per_split);
    const int nk = (me - ms) / 32;
    const unsigned short* Pb = g.A + (long)b * g.gsA;
    const unsigned short* Qb = g.B + (long)b * g.gsB;
    float* __restrict__ Wb = g.C + ((long)b * g.splits + split) * (long)NI * NJ;
    const i32x4 rsP = x3_rsrc(Pb, 0xfffffff0u), rsQ = x3_rsrc(Qb, 0xfffffff0u);
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const unsigned char*)smem;
    auto swzp = [](int k) { return BI >= 128 ? (k & 3) << 2 : ((k >> 1) & 1) << 2; };      // XOR on the chunk index (a 64-byte block = 4 chunks)
    auto swzq = [](int k) { return BJ >= 128 ? (k & 3) << 2 : ((k >> 1) & 1) << 2; };

    unsigned poff[NIP], qoff[NIQ], plds[NIP], qlds[NIQ];
    int qrowl[IQ ? NIQ : 1], qdy[IQ ? NIQ : 1], qdx[IQ ? NIQ : 1]; unsigned qcol[IQ ? NIQ : 1], qzero[IQ ? NIQ : 1];
#pragma unroll
    for (int i = 0; i < NIP; ++i) {
        constexpr int IPC = 32 * CP / 64, RPI = 64 / CP;                   // instructions per component image, k-rows per instruction
        const int gi = wave * NIP + i, comp = gi / IPC, sub = gi - comp * IPC;
        const int row = sub * RPI + lane / CP, ch = (lane % CP) ^ swzp(row);
        plds[i] = (unsigned)(comp * PIMG + sub * 1024);
        poff[i] = (unsigned)((comp * g.csA + (long)(ms + row) * ldp + min(i0 + ch * 8, NI - 8)) * 2);     // (columns past NI: clamped, never stored)
    }
#pragma unroll
    for (int i = 0; i < NIQ; ++i) {
        constexpr int IPC = 32 * CQ / 64, RPI = 64 / CQ;
        const int gi = wave * NIQ + i, comp = gi / IPC, sub = gi - comp * IPC;
        const int row = sub * RPI + lane / CQ, ch = (lane % CQ) ^ swzq(row);
        qlds[i] = (unsigned)(3 * PIMG + comp * QIMG + sub * 1024);
        qoff[i] = (unsigned)((comp * g.csB + (long)(ms + row) * ldq + min(j0 + ch * 8, NJ - 8)) * 2);
        if constexpr (IQ) {
            const int col = min(j0 + ch * 8, NJ - 8), tap = col / g.Cc;
            qrowl[i] = row;
            qcol[i] = (unsigned)((comp * g.csB + (col - tap * g.Cc)) * 2);
            qzero[i] = (unsigned)((comp * g.csB + g.zrow + (col - tap * g.Cc)) * 2);
            qdy[i] = g.tdy[tap]; qdx[i] = g.tdx[tap];
        }
    }
    int nissued = 0;
    unsigned kp = 0, kq = 0, sbase = lds0;
    auto piece = [&](int p) {
        if (p < NIP) x3_dma16s(rsP, sbase + plds[p], poff[p], kp);
        else x3_dma16s(rsQ, sbase + qlds[p - NIP], qoff[p - NIP], IQ ? 0u : kq);
    };
    auto begin_issue = [&](int stage) {
        const unsigned t = (unsigned)min(nissued, nk - 1);  // (past the end: the last tile again, into a stage nobody consumes)
        kp = t * 64u * (unsigned)ldp; kq = t * 64u * (unsigned)ldq;
        ++nissued;
        sbase = lds0 + (unsigned)(stage * STAGE);
        if constexpr (IQ) {                                  // this K-step's 32 rows of the reduction -> their pixels of the tile's tap (or the zero row)
#pragma unroll
            for (int i = 0; i < NIQ; ++i) {
                const int m = ms + (int)t * 32 + qrowl[i];
                const int ni = m >> g.lhw, iy = (m >> g.lw) & (g.QH - 1), ix = m & (g.QW - 1);
                const int yy = iy * g.sy + qdy[i], xx = ix * g.sx + qdx[i];
                const bool ok = ((unsigned)yy < (unsigned)g.AH) & ((unsigned)xx < (unsigned)g.AW);
                qoff[i] = ok ? qcol[i] + (unsigned)((ni * g.AH + yy) * g.AW + xx) * (unsigned)(ldq * 2) : qzero[i];
            }
        }
    };

    f32x16 acc[TM][TN];
#if X3_SPLIT_ACC
    f32x16 acc2[TM][TN];                                     // the small products (see X3_SPLIT_ACC)
#endif
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int c = 0; c < TN; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[a][c][r] = 0.f;
#if X3_SPLIT_ACC
                acc2[a][c][r] = 0.f;
#endif
            }

    // transposing fragment reads: lane (g, q, p) = (lane >> 4, (lane & 15) >> 2, lane & 3) reads k-row 8 (g >> 1) + q (and + 4), columns c0 + 16 (g & 1) + 4 p ...
    const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
    const int fk = 8 * (fg >> 1) + fq;                                     // (fk & 3 == fq, (fk + 4) & 3 == fq: one swizzle for both reads)
    int pfo[TM], qfo[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int col = (wm * TM + i) * 32 + 16 * (fg & 1) + 4 * fp;       // chunk col / 8, byte (col & 7) * 2 inside it
        pfo[i] = fk * BI * 2 + (((col >> 3) ^ swzp(fk)) << 4) + (col & 7) * 2;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = (wn * TN + j) * 32 + 16 * (fg & 1) + 4 * fp;
        qfo[j] = 3 * PIMG + fk * BJ * 2 + (((col >> 3) ^ swzq(fk)) << 4) + (col & 7) * 2;
    }
    typedef __attribute__((address_space(3))) s16x4v lds_s16x4;
    auto frag = [&](const unsigned char* p, int rowbytes) {
        const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
        const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * rowbytes));
        s16x8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return *reinterpret_cast<bf16x8v*>(&v);
    };

#pragma unroll
    for (int p = 0; p < ST - 1; ++p) {
        begin_issue(p);
#pragma unroll
        for (int q = 0; q < NIP + NIQ; ++q) piece(q);
    }
    int st = 0, stn = ST - 1;
    constexpr int NG = 2 * TM * TN, NP = NIP + NIQ, NR = (TM + TN) * 3;
    constexpr int NSLOT = NG * NPROD / 2;
    constexpr int RSL = NSLOT / 2 - 1 > 0 ? NSLOT / 2 - 1 : 1;              // slots that carry the second sub-step's fragment reads (one-tile waves with three products: all in slot 0)
    using PR = X3Prod<NPROD>;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * (NIP + NIQ)) : "memory");
        asm volatile("s_barrier" ::: "memory");
        begin_issue(stn);
        const unsigned char* sp = smem + st * STAGE;
        bf16x8v af[2][TM][3], bf[2][TN][3];
        auto read1 = [&](int ks, int q) {
            if (q < TM * 3) { const int i = q / 3, c = q % 3; af[ks][i][c] = frag(sp + c * PIMG + ks * 16 * BI * 2 + pfo[i], BI * 2); }
            else { const int j = (q - TM * 3) / 3, c = (q - TM * 3) % 3; bf[ks][j][c] = frag(sp + c * QIMG + ks * 16 * BJ * 2 + qfo[j], BJ * 2); }
        };
#pragma unroll
        for (int q = 0; q < NR; ++q) read1(0, q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NG * NPROD; ++q) {               // (schedule: see above x3gemm_nt)
            const int gi = q / NPROD, pr = q % NPROD;
            const int ks = gi / (TM * TN), i = (gi / TN) % TM, j = gi % TN;
#if X3_SPLIT_ACC
            if (PR::a[pr] + PR::b[pr] != 0) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i][PR::a[pr]], bf[ks][j][PR::b[pr]], acc2[i][j], 0, 0, 0);
            else
#endif
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i][PR::a[pr]], bf[ks][j][PR::b[pr]], acc[i][j], 0, 0, 0);
            if (q % 2 == 1 || q == NG * NPROD - 1) {
                const int slot = q / 2;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = slot * NR / RSL; r < (slot + 1) * NR / RSL && r < NR; ++r)
                    if (slot < RSL) read1(1, r);
                if (slot % 2 == 0 && slot / 2 < NP) piece(slot / 2);
                if (slot == NSLOT - 1) {
#pragma unroll
                    for (int p = (NSLOT + 1) / 2; p < NP; ++p) piece(p);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        st = st == ST - 1 ? 0 : st + 1;
        stn = stn == ST - 1 ? 0 : stn + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#if X3_SPLIT_ACC
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int c = 0; c < TN; ++c) acc[a][c] += acc2[a][c];
#endif

    const int h = lane >> 5;
    const bool whole = i0 + BI <= NI && (double)NI * NJ * 4.0 < 4294967000.0;
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, whole ? (unsigned)NI * (unsigned)NJ * 4u : 0u, 0x00020000);
    const unsigned ldw4 = (unsigned)NJ * 4u;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = j0 + (wn * TN + j) * 32 + (lane & 31);
        const bool cok = col < NJ;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r0 = i0 + (wm * TM + i) * 32 + 4 * h;
            if (whole) {
                const unsigned vo = cok ? (unsigned)(r0 * NJ + col) * 4u : 0xffffffffu;
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
                if (g.accum) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsW, vo, ((r & 3) + 8 * (r >> 2)) * ldw4, 0));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), rsW, vo, ((r & 3) + 8 * (r >> 2)) * ldw4, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = r0 + (r & 3) + 8 * (r >> 2);
                    if (cok && row < NI) Wb[(long)row * NJ + col] = acc[i][j][r] + (g.accum ? Wb[(long)row * NJ + col] : 0.f);
                }
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, int ST>
static int x3_launch_tn(const X3Gemm& g, int nprod, hipStream_t s) {
    constexpr int BI = WM * TM * 32, BJ = WN * TN * 32;
    const dim3 grid((unsigned)(cdiv(g.M, BI) * cdiv(g.N, BJ) * g.splits * g.batch));
    char nm[96];
    snprintf(nm, sizeof nm, g.T > 0 ? "x3gemm_tn<%d, %d, %d, %d, %d, %d, true>" : "x3gemm_tn<%d, %d, %d, %d, %d, %d>", WM, WN, TM, TN, ST, g.T > 0 ? 6 : nprod);
    KTimer kt(nm, 2.0 * g.batch * g.K * g.M * g.N, g.batch * (6.0 * g.K * ((double)g.M + g.N) + 4.0 * g.splits * g.M * g.N), s);
    if (g.T > 0) hipLaunchKernelGGL((x3gemm_tn<WM, WN, TM, TN, ST, 6, true>), grid, dim3(WM * WN * 64), 0, s, g);       // implicit Q (six products only)
    else if (nprod == 9) hipLaunchKernelGGL((x3gemm_tn<WM, WN, TM, TN, ST, 9>), grid, dim3(WM * WN * 64), 0, s, g);
    else if (nprod == 3) hipLaunchKernelGGL((x3gemm_tn<WM, WN, TM, TN, ST, 3>), grid, dim3(WM * WN * 64), 0, s, g);
    else hipLaunchKernelGGL((x3gemm_tn<WM, WN, TM, TN, ST, 6>), grid, dim3(WM * WN * 64), 0, s, g);
    PDF_LAUNCH_CHECK();
    return 0;
}

// the transposed convolutions' weight gradients: tile of the TN kernel (PDF_X3_TN_DECONV: 0 = 128 x 128 on 4 waves, 3 = on 8 waves, 2 = 256 x 128 on 8 waves;
// default: the wide tile when it still makes >= 256 blocks, else 128 x 128 on 8 waves -- isolated p5 0.77 -> 0.63 ms, p3 0.78 -> 0.70, p4 0.38 -> 0.37; the
// step does not notice, these run beside the main chain: profiles/r06_tn_deconv.txt)
static int x3_launch_tn_deconv(const X3Gemm& g, hipStream_t s) {
    static const int env = getenv("PDF_X3_TN_DECONV") ? atoi(getenv("PDF_X3_TN_DECONV")) : -1;
    int v = env;
    if (v < 0) v = (g.M % 256 == 0 && (long)(g.M / 256) * cdiv(g.N, 128) * g.splits * g.batch >= 256) ? 2 : 3;
    if (v == 3) return x3_launch_tn<2, 4, 2, 1, 3>(g, 6, s);
    if (v == 2 && g.M % 256 == 0) return x3_launch_tn<4, 2, 2, 2, 2>(g, 6, s);
    return x3_launch_tn<2, 2, 2, 2, 3>(g, 6, s);
}
// rows of the reduction per split and the split count actually used (rows per split a multiple of 32)
int pdf_internal_x3_tn_splits(int M, int splits) {
    const int rps = cdiv(cdiv(M, splits < 1 ? 1 : splits), 32) * 32;
    return cdiv(M, rps);
}
// slab_b [split][NI][NJ] = sum over the rows m of the split of P_b[m][:]^T Q_b[m][:], operands as x3 planes (P_b [M][NI], Q_b [M][NJ]).
// NI % 8 == 0, NJ % 8 == 0, M % 32 == 0.  -> the split count used (> 0) or a negative error.  variant: 0 = 128 x 128 tile; 1 = 128 x 64; 2 = 256 x 128 (8 waves)
int pdf_internal_x3_batched_wgemm(const void* P3, long csP, const void* Q3, long csQ, float* slab, int batch, long gsP, long gsQ,
                                  int M, int NI, int NJ, int splits, int variant, int nprod, hipStream_t s) {
    if (NI % 8 != 0 || NJ % 8 != 0 || M < 32 || M % 32 != 0 || batch < 1) return PDF_E_BADARG;
    if (2.0 * csP + (double)M * NI >= 2147483000.0 || 2.0 * csQ + (double)M * NJ >= 2147483000.0) return PDF_E_BADARG;
    X3Gemm g = {};
    g.A = (const unsigned short*)P3; g.B = (const unsigned short*)Q3; g.C = slab;
    g.csA = csP; g.csB = csQ; g.gsA = gsP; g.gsB = gsQ;
    g.M = NI; g.N = NJ; g.K = M; g.batch = batch;
    g.splits = pdf_internal_x3_tn_splits(M, splits);
    g.rows_per_split = cdiv(cdiv(M, splits < 1 ? 1 : splits), 32) * 32;
    int rc;
    if (variant < 0) variant = (NI % 256 == 0 && (long)(NI / 256) * cdiv(NJ, 128) * g.splits * batch >= 512) ? 2 : 0;
    if (variant == 3) rc = x3_launch_tn<2, 4, 2, 1, 3>(g, nprod, s);          // 128 x 128 on 8 waves
    else if (variant == 1) rc = x3_launch_tn<2, 2, 2, 1, 3>(g, nprod, s);
    else if (variant == 2) rc = x3_launch_tn<4, 2, 2, 2, 2>(g, nprod, s);
    else rc = x3_launch_tn<2, 2, 2, 2, 3>(g, nprod, s);
    return rc != 0 ? (rc < 0 ? rc : -rc) : g.splits;
}
PDF_API int pdf_x3_batched_gemm_tn(const void* P3, long csP, const void* Q3, long csQ, float* slab, int batch, long gsP, long gsQ,
                                   int M, int NI, int NJ, int splits, int variant, int nprod, hipStream_t s) {
    const int rc = pdf_internal_x3_batched_wgemm(P3, csP, Q3, csQ, slab, batch, gsP, gsQ, M, NI, NJ, splits, variant, nprod, s);
    return rc > 0 ? 0 : (rc == 0 ? PDF_E_BADARG : rc);
}
int pdf_internal_batched_wgemm(const float* P, const float* Q, float* slab, int batch, long gsP, long gsQ, int M, int NI, int NJ, int splits, hipStream_t s);
// the native fp32 weight-gradient-shaped batched product (gemm.hip; rows per split a multiple of 16)
PDF_API int pdf_batched_gemm_tn(const float* P, const float* Q, float* slab, int batch, long gsP, long gsQ, int M, int NI, int NJ, int splits, hipStream_t s) {
    const int rc = pdf_internal_batched_wgemm(P, Q, slab, batch, gsP, gsQ, M, NI, NJ, splits, s);
    return rc > 0 ? 0 : (rc == 0 ? PDF_E_BADARG : rc);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Transposed convolutions with kernel == stride (the pyramid's p4: 1024 -> 256, k = s = 4 and p5: 2048 -> 256, k = s = 8; reference
// intaghand_encoder.py:603-611) are plain GEMMs -- y_ps[m][(tap, co)] = x[m][:] . w[:][(tap, co)] with the output pixel-shuffled, and
// dx[m][ci] = dy_unshuffled[m][:] . w[ci][:] -- with 2,048 - 16,384 wide reductions over small activations: bound by the matrix pipe, the shape
// x3 is for (tools/probe/x3_stamps.py: 222 TFLOP/s-equivalent on the p5 shape against 116 of the native kernel).  The operands are split by
// pre-passes (the weights change every step): x -> x3 planes, w [Cin][taps * Cout] -> its transpose as x3 planes (forward) or as it lies
// (backward-data), dy -> un-shuffled x3 planes [m][(tap, co)].
// W [K][N] fp32 (row stride ldw) -> Wt3 [3][N][K] (row stride ldo): 64 x 64 tiles through LDS.  blockIdx.z = job: input and output advanced by
// in_off[z] / out_off[z] elements (the per-tap [Cin][Cout] slices of a transposed-convolution weight, gathered into per-parity operands)
struct X3TrJobs { long in_off[16], out_off[16]; };
__global__ __launch_bounds__(256) void x3_split_transpose_kernel(const float* __restrict__ w, long ldw, unsigned short* __restrict__ o, int K, int N, long ldo, long cs,
                                                                 const X3TrJobs jobs) {
    __shared__ float sm[64][65];
    w += jobs.in_off[blockIdx.z]; o += jobs.out_off[blockIdx.z];
    const int tk = blockIdx.y * 64, tn = blockIdx.x * 64, t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = t / 16 + 16 * i, c = (t % 16) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tk + r < K && tn + c < N) v = *reinterpret_cast<const float4*>(w + (long)(tk + r) * ldw + tn + c);
        sm[r][c] = v.x; sm[r][c + 1] = v.y; sm[r][c + 2] = v.z; sm[r][c + 3] = v.w;
    }
    __syncthreads();
    const int n = t / 4, k0 = (t % 4) * 16;
    if (tn + n < N && tk + k0 < K) {
        uint4 h[2], m[2], l[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            unsigned* hh = reinterpret_cast<unsigned*>(&h[q]); unsigned* mm = reinterpret_cast<unsigned*>(&m[q]); unsigned* ll = reinterpret_cast<unsigned*>(&l[q]);
#pragma unroll
            for (int e = 0; e < 4; ++e) pdf_x3_split2(sm[k0 + 8 * q + 2 * e][n], sm[k0 + 8 * q + 2 * e + 1][n], hh[e], mm[e], ll[e]);
        }
        unsigned short* d = o + (long)(tn + n) * ldo + tk + k0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<uint4*>(d + 8 * q) = h[q];
            *reinterpret_cast<uint4*>(d + cs + 8 * q) = m[q];
            *reinterpret_cast<uint4*>(d + 2 * cs + 8 * q) = l[q];
        }
    }
}
// dy [ni][OH][OW][C] (row stride lddy) -> x3 planes [3][M][taps * C], row m = input pixel (ni, qy, qx), column (ky, kx, c) = dy[ni][qy s + ky][qx s + kx][c]
__global__ __launch_bounds__(256) void x3_split_unshuffle_kernel(const float* __restrict__ dy, int lddy, unsigned short* __restrict__ o, long ldo, long cs,
                                                                 long M, int QH, int QW, int OH, int OW, int s, int C) {
    const int C8 = C / 8, T = s * s;
    const long total = M * T * C8;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        const long r = i / C8;
        const int tap = (int)(r % T);
        const long m = r / T;
        const int ky = tap / s, kx = tap - ky * s;
        const int hw = QH * QW, ni = (int)(m / hw), rem = (int)(m - (long)ni * hw), qy = rem / QW, qx = rem - qy * QW;
        const float* src = dy + (((long)ni * OH + qy * s + ky) * OW + qx * s + kx) * lddy + c8 * 8;
        const float4 u = *reinterpret_cast<const float4*>(src), v = *reinterpret_cast<const float4*>(src + 4);
        uint4 h, mm, l;
        pdf_x3_split2(u.x, u.y, h.x, mm.x, l.x); pdf_x3_split2(u.z, u.w, h.y, mm.y, l.y);
        pdf_x3_split2(v.x, v.y, h.z, mm.z, l.z); pdf_x3_split2(v.z, v.w, h.w, mm.w, l.w);
        unsigned short* d = o + m * ldo + (long)tap * C + c8 * 8;
        *reinterpret_cast<uint4*>(d) = h;
        *reinterpret_cast<uint4*>(d + cs) = mm;
        *reinterpret_cast<uint4*>(d + 2 * cs) = l;
    }
}
// Which launches take the x3 form: bit 0 = the wide Winograd-domain products (winograd.hip), bit 1 = the kernel == stride transposed convolutions.
// Default 7 (PDF_X3=0: none; PDF_X3_DECONV=0: not the transposed convolutions; PDF_X3_MESH=0: not the fused mesh decoder's linear products -- bit 2, the
// caller picks pdf_mesh_level_*_x3 by it); pdf_set_x3_mode changes it at run time (bench.py times the native
// fp32-MFMA step beside the shipped one).  Workspace sizes depend on it: a caller that caches them (functional.py) drops its cache on a change.
static int g_x3_mode = -1;
int pdf_internal_x3_mode() {
    if (g_x3_mode < 0) {
        int v = getenv("PDF_X3") ? (atoi(getenv("PDF_X3")) ? 7 : 0) : 7;
        if (getenv("PDF_X3_DECONV") && !atoi(getenv("PDF_X3_DECONV"))) v &= ~2;
        if (getenv("PDF_X3_MESH") && !atoi(getenv("PDF_X3_MESH"))) v &= ~4;
        g_x3_mode = v;
    }
    return g_x3_mode;
}
PDF_API int pdf_set_x3_mode(int mode) { g_x3_mode = mode < 0 ? -1 : (mode & 7); return 0; }
PDF_API int pdf_debug_x3_mode(void) { return pdf_internal_x3_mode(); }
static int x3_deconv_mode() { return (pdf_internal_x3_mode() & 2) != 0; }
// row stride (elements) of a pre-split operand with K columns: one 128-byte line of padding, so that rows of 2^k bytes (p5: 4 KB and 32 KB) do not
// all start on the same L2 / HBM channel (without it p5's backward-data ran 1.57 ms against 1.29 of the native kernel: profiles/r06_x3_deconv.txt)
static long x3_ld(long K) { return K + 64; }
// floats of workspace the x3 form of this transposed convolution wants (backward = 0: forward, 1: backward-data, 2: weight gradient), 0 when it does not qualify
static long x3_deconv_general_workspace(int N, int H, int W, int Cin, int Cout, int K, int stride, int pad, int backward);
long pdf_internal_x3_deconv_workspace(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int backward) {
    if (!x3_deconv_mode() || KH != KW || stride < 2) return 0;
    if (KH != stride || pad != 0) return x3_deconv_general_workspace(N, H, W, Cin, Cout, KH, stride, pad, backward);
    const long M = (long)N * H * W, NT = (long)KH * KW * Cout;
    if (Cin % 32 != 0 || Cout % 32 != 0 || M < 1024 || 2.0 * M * NT * Cin < 3.0e10) return 0;      // (a long reduction over a large product: p4, p5)
    if (backward == 2 && (M % 32 != 0 || 3.0 * Cin * NT * 4.0 >= 4294967000.0)) return 0;
    const long wel = backward == 2 ? M * x3_ld(Cin) : backward ? Cin * x3_ld(NT) : NT * x3_ld(Cin), ael = backward ? M * x3_ld(NT) : M * x3_ld(Cin);
    if (3.0 * wel >= 2147483000.0 || 3.0 * ael >= 2147483000.0) return 0;
    return ((3 * wel + 1) / 2 + 3) / 4 * 4 + ((3 * ael + 1) / 2 + 3) / 4 * 4 + 64;
}
static void x3_gemm_desc(X3Gemm& g, const void* A3, long csA, long lda, const void* B3, long csB, long ldb, float* C, int M, int N, int K, int ldc) {
    g = X3Gemm{};
    g.A = (const unsigned short*)A3; g.B = (const unsigned short*)B3; g.C = C;
    g.csA = csA; g.csB = csB; g.M = M; g.N = N; g.K = K; g.batch = 1; g.ldc = ldc; g.lda = (int)lda; g.ldb = (int)ldb;
    g.col_major_tiles = cdiv(N, 128) > 4 * cdiv(M, 128);
}
static int x3_launch_auto(const X3Gemm& g, hipStream_t s) {
    const int v = x3_nt_variant(g.M, g.N, g.K, g.batch);
    if (v == 0) return x3_launch_nt<4, 2, 2, 2, 2>(g, 6, s);
    if (v == 1) return x3_launch_nt<2, 2, 2, 2, 3>(g, 6, s);
    return x3_launch_nt<2, 4, 2, 1, 3>(g, 6, s);
}
// y [N][OH][OW][Cout] (row stride ldy) = ConvTranspose2d(x [N][H][W][Cin] dense, w [Cin][KH][KW][Cout]) + bias, kernel == stride, pad 0
int pdf_internal_x3_deconv_fwd(const float* x, const float* w, const float* bias, float* y, float* ws, int N, int H, int W, int Cin, int Cout,
                               int KH, int KW, int stride, int OH, int OW, int ldy, hipStream_t s) {
    const long M = (long)N * H * W, NT = (long)KH * KW * Cout, ld = x3_ld(Cin);
    unsigned short* w3 = reinterpret_cast<unsigned short*>(ws);
    unsigned short* x3 = reinterpret_cast<unsigned short*>(ws + ((3 * NT * ld + 1) / 2 + 3) / 4 * 4);
    {
        KTimer kt("x3_split_transpose_kernel", 0.0, 10.0 * NT * Cin, s);
        hipLaunchKernelGGL(x3_split_transpose_kernel, dim3((unsigned)cdiv(NT, 64), (unsigned)cdiv(Cin, 64)), dim3(256), 0, s, w, NT, w3, Cin, (int)NT, ld, NT * ld, X3TrJobs{});
    }
    hipLaunchKernelGGL(x3_split_rows_kernel, dim3(grid_for(M * Cin / 8, 256, 256 * 16)), dim3(256), 0, s, x, (long)Cin, x3, ld, M * ld, M, Cin / 8);
    PDF_LAUNCH_CHECK();
    X3Gemm g;
    x3_gemm_desc(g, x3, M * ld, ld, w3, NT * ld, ld, y, (int)M, (int)NT, Cin, ldy);
    g.ps_cout = Cout; g.ps_kw = KW; g.ps_s = stride; g.QH = H; g.QW = W; g.OH = OH; g.OW = OW; g.bias = bias;
    return x3_launch_auto(g, s);
}
// dx [N][H][W][Cin] (row stride lddx) = the input gradient of that layer from dy [N][OH][OW][Cout] (row stride lddy)
int pdf_internal_x3_deconv_bwd_data(const float* dy, const float* w, float* dx, float* ws, int N, int H, int W, int Cin, int lddx, int Cout,
                                    int KH, int KW, int stride, int OH, int OW, int lddy, hipStream_t s) {
    const long M = (long)N * H * W, NT = (long)KH * KW * Cout, ld = x3_ld(NT);
    unsigned short* w3 = reinterpret_cast<unsigned short*>(ws);
    unsigned short* d3 = reinterpret_cast<unsigned short*>(ws + ((3 * Cin * ld + 1) / 2 + 3) / 4 * 4);
    {
        KTimer kt("x3_split_rows_kernel", 0.0, 10.0 * NT * Cin, s);
        hipLaunchKernelGGL(x3_split_rows_kernel, dim3(grid_for(NT * Cin / 8, 256, 256 * 16)), dim3(256), 0, s, w, NT, w3, ld, Cin * ld, (long)Cin, (int)(NT / 8));
    }
    {
        KTimer kt("x3_split_unshuffle_kernel", 0.0, 10.0 * M * NT, s);
        hipLaunchKernelGGL(x3_split_unshuffle_kernel, dim3(grid_for(M * NT / 8, 256, 256 * 16)), dim3(256), 0, s, dy, lddy, d3, ld, M * ld, M, H, W, OH, OW, stride, Cout);
    }
    PDF_LAUNCH_CHECK();
    X3Gemm g;
    x3_gemm_desc(g, d3, M * ld, ld, w3, Cin * ld, ld, dx, (int)M, Cin, (int)NT, lddx);
    return x3_launch_auto(g, s);
}

// dw [Cin][KH][KW][Cout] (+)= the weight gradient of that layer: sum over the input pixels m of x[m][ci] dy_unshuffled[m][(tap, co)] -- one x3gemm_tn launch,
// no split of the reduction (p5: 2,048 rows against 2,048 output tiles), the accumulation into an existing gradient done by the epilogue
int pdf_internal_x3_deconv_bwd_weight(const float* x, const float* dy, float* dw, float* ws, int N, int H, int W, int Cin, int Cout,
                                      int KH, int KW, int stride, int OH, int OW, int lddy, int accumulate, hipStream_t s) {
    const long M = (long)N * H * W, NT = (long)KH * KW * Cout, ldx3 = x3_ld(Cin), ldd = x3_ld(NT);
    unsigned short* x3 = reinterpret_cast<unsigned short*>(ws);
    unsigned short* d3 = reinterpret_cast<unsigned short*>(ws + ((3 * M * ldx3 + 1) / 2 + 3) / 4 * 4);
    hipLaunchKernelGGL(x3_split_rows_kernel, dim3(grid_for(M * Cin / 8, 256, 256 * 16)), dim3(256), 0, s, x, (long)Cin, x3, ldx3, M * ldx3, M, Cin / 8);
    {
        KTimer kt("x3_split_unshuffle_kernel", 0.0, 10.0 * M * NT, s);
        hipLaunchKernelGGL(x3_split_unshuffle_kernel, dim3(grid_for(M * NT / 8, 256, 256 * 16)), dim3(256), 0, s, dy, lddy, d3, ldd, M * ldd, M, H, W, OH, OW, stride, Cout);
    }
    PDF_LAUNCH_CHECK();
    X3Gemm g = {};
    g.A = x3; g.B = d3; g.C = dw;
    g.csA = M * ldx3; g.csB = M * ldd; g.lda = (int)ldx3; g.ldb = (int)ldd;
    g.M = Cin; g.N = (int)NT; g.K = (int)M; g.batch = 1; g.splits = 1; g.rows_per_split = (int)M; g.accum = accumulate;
    return x3_launch_tn_deconv(g, s);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Transposed convolutions whose kernel is a multiple of the stride (the pyramid's p3: 512 -> 256, k = 4, s = 2, pad = 1; intaghand_encoder.py:602):
//   forward        s x s parity classes of output pixels, each an implicit GEMM over its (k / s)^2 taps: rows = the class's pixels (qy, qx), A = x at
//                  (qy + dy[t], qx + dx[t]) (zero outside the map), B = the class's taps of w gathered and transposed by the pre-pass, output scattered
//                  to (qy s + py, qx s + px);
//   backward-data  ONE implicit GEMM: rows = input pixels (iy, ix), A = dy at (iy s - pad + ky, ix s - pad + kx) over all k^2 taps, B = w as it lies.
// Same x3gemm_nt kernel (template IM): the tap's pixel offset is applied per lane when the tap changes, its channels advance by the scalar offset.
static bool x3_general_ok(int Cin, int Cout, int K, int stride, int pad) {
    return K % stride == 0 && K * K <= 16 && K > stride && pad >= 0 && pad < K && Cin % 32 == 0 && Cout % 32 == 0;
}
static int x3_general_wgrad_splits(long M, int NI, long NJ) {      // ~2 blocks per CU, at least 64 K-steps per block
    const long tiles = (long)cdiv(NI, 128) * cdiv(NJ, 128);
    long sp = (512 + tiles - 1) / tiles;
    if (sp > M / 2048) sp = M / 2048;
    return (int)(sp < 1 ? 1 : sp);
}
static long x3_deconv_general_workspace(int N, int H, int W, int Cin, int Cout, int K, int stride, int pad, int backward) {
    if (!x3_general_ok(Cin, Cout, K, stride, pad)) return 0;
    const int OH = (H - 1) * stride - 2 * pad + K, OW = (W - 1) * stride - 2 * pad + K;
    if (OH % stride != 0 || OW % stride != 0) return 0;
    const long Min = (long)N * H * W, Mout = (long)N * OH * OW, T = (long)(K / stride) * (K / stride);
    if (2.0 * Min * K * K * Cin * Cout < 3.0e10) return 0;
    long ael, bel;
    if (backward == 2) {                                     // P = x, Q = dy (+ zero row), slabs of the split reduction
        if ((H & (H - 1)) || (W & (W - 1)) || Min % 32 != 0 || Cout % 128 != 0) return 0;
        const long pel = Min * x3_ld(Cin), qel = (Mout + 1) * x3_ld(Cout);
        if (3.0 * pel >= 2147483000.0 || 3.0 * qel >= 2147483000.0) return 0;
        return ((3 * pel + 1) / 2 + 3) / 4 * 4 + ((3 * qel + 1) / 2 + 3) / 4 * 4 + (long)x3_general_wgrad_splits(Min, Cin, (long)K * K * Cout) * Cin * K * K * Cout + 64;
    }
    if (backward == 0) { ael = (Min + 1) * x3_ld(Cin); bel = (long)stride * stride * Cout * x3_ld(T * Cin); }
    else { ael = (Mout + 1) * x3_ld(Cout); bel = (long)Cin * x3_ld((long)K * K * Cout); }
    if (3.0 * ael >= 2147483000.0 || 3.0 * bel >= 2147483000.0) return 0;
    return ((3 * bel + 1) / 2 + 3) / 4 * 4 + ((3 * ael + 1) / 2 + 3) / 4 * 4 + 64;
}
static int x3_zero_rows(unsigned short* a3, long rows, long ld, hipStream_t s) {      // the zero row behind each component plane
    for (int c = 0; c < 3; ++c)
        if (hipMemsetAsync(a3 + c * (rows + 1) * ld + rows * ld, 0, (size_t)ld * 2, s) != hipSuccess) return PDF_E_BADARG;
    return 0;
}
int pdf_internal_x3_deconv_general_fwd(const float* x, const float* w, const float* bias, float* y, float* ws, int N, int H, int W, int Cin, int Cout,
                                       int K, int stride, int pad, int OH, int OW, int ldy, hipStream_t s) {
    const long Min = (long)N * H * W;
    const int kk = K / stride, T = kk * kk, P = stride * stride;
    const long ldA = x3_ld(Cin), ldB = x3_ld((long)T * Cin);
    unsigned short* b3 = reinterpret_cast<unsigned short*>(ws);
    unsigned short* a3 = reinterpret_cast<unsigned short*>(ws + ((3 * P * Cout * ldB + 1) / 2 + 3) / 4 * 4);
    // taps of parity class (py, px), in the order both operands use
    int tky[4][16], tkx[4][16];
    X3TrJobs jobs = {};
    if (P > 4 || P * T > 16) return PDF_E_BADARG;
    for (int py = 0; py < stride; ++py)
        for (int px = 0; px < stride; ++px) {
            int n = 0;
            for (int ky = 0; ky < K; ++ky) {
                if (((py + pad - ky) % stride + stride) % stride != 0) continue;
                for (int kx = 0; kx < K; ++kx) {
                    if (((px + pad - kx) % stride + stride) % stride != 0) continue;
                    const int pc = py * stride + px;
                    tky[pc][n] = ky; tkx[pc][n] = kx;
                    jobs.in_off[pc * T + n] = (long)(ky * K + kx) * Cout;
                    jobs.out_off[pc * T + n] = (long)pc * Cout * ldB + (long)n * Cin;
                    ++n;
                }
            }
            if (n != T) return PDF_E_BADARG;
        }
    {
        KTimer kt("x3_split_transpose_kernel", 0.0, 10.0 * K * K * Cout * Cin, s);
        hipLaunchKernelGGL(x3_split_transpose_kernel, dim3((unsigned)cdiv(Cout, 64), (unsigned)cdiv(Cin, 64), (unsigned)(P * T)), dim3(256), 0, s,
                           w, (long)K * K * Cout, b3, Cin, Cout, ldB, (long)P * Cout * ldB, jobs);
    }
    hipLaunchKernelGGL(x3_split_rows_kernel, dim3(grid_for(Min * Cin / 8, 256, 256 * 16)), dim3(256), 0, s, x, (long)Cin, a3, ldA, (Min + 1) * ldA, Min, Cin / 8);
    if (int rc = x3_zero_rows(a3, Min, ldA, s)) return rc;
    PDF_LAUNCH_CHECK();
    for (int py = 0; py < stride; ++py)
        for (int px = 0; px < stride; ++px) {
            const int pc = py * stride + px, QH = (OH - py + stride - 1) / stride, QW = (OW - px + stride - 1) / stride;
            X3Gemm g;
            x3_gemm_desc(g, a3, (Min + 1) * ldA, ldA, b3 + (long)pc * Cout * ldB, (long)P * Cout * ldB, ldB, y, N * QH * QW, Cout, T * Cin, ldy);
            g.ps_cout = Cout; g.ps_kw = 1; g.ps_s = stride; g.QH = QH; g.QW = QW; g.OH = OH; g.OW = OW; g.ooy = py; g.oox = px; g.bias = bias;
            g.AH = H; g.AW = W; g.sy = 1; g.sx = 1; g.T = T; g.Cc = Cin; g.zrow = Min * ldA;
            for (int n = 0; n < T; ++n) {                      // iy = qy + (py + pad - ky) / stride (exact division: the tap belongs to the class)
                g.tdy[n] = (py + pad - tky[pc][n]) / stride; g.tdx[n] = (px + pad - tkx[pc][n]) / stride;
            }
            if (int rc = x3_launch_auto(g, s)) return rc;
        }
    return 0;
}
int pdf_internal_x3_deconv_general_bwd_data(const float* dy, const float* w, float* dx, float* ws, int N, int H, int W, int Cin, int lddx, int Cout,
                                            int K, int stride, int pad, int OH, int OW, int lddy, hipStream_t s) {
    const long Mout = (long)N * OH * OW, M = (long)N * H * W, NT = (long)K * K * Cout;
    const long ldA = x3_ld(Cout), ldB = x3_ld(NT);
    unsigned short* b3 = reinterpret_cast<unsigned short*>(ws);
    unsigned short* a3 = reinterpret_cast<unsigned short*>(ws + ((3 * Cin * ldB + 1) / 2 + 3) / 4 * 4);
    {
        KTimer kt("x3_split_rows_kernel", 0.0, 10.0 * NT * Cin, s);
        hipLaunchKernelGGL(x3_split_rows_kernel, dim3(grid_for(NT * Cin / 8, 256, 256 * 16)), dim3(256), 0, s, w, NT, b3, ldB, Cin * ldB, (long)Cin, (int)(NT / 8));
    }
    {
        KTimer kt("x3_split_rows_kernel", 0.0, 10.0 * Mout * Cout, s);
        hipLaunchKernelGGL(x3_split_rows_kernel, dim3(grid_for(Mout * Cout / 8, 256, 256 * 16)), dim3(256), 0, s, dy, (long)lddy, a3, ldA, (Mout + 1) * ldA, Mout, Cout / 8);
    }
    if (int rc = x3_zero_rows(a3, Mout, ldA, s)) return rc;
    PDF_LAUNCH_CHECK();
    X3Gemm g;
    x3_gemm_desc(g, a3, (Mout + 1) * ldA, ldA, b3, Cin * ldB, ldB, dx, (int)M, Cin, (int)NT, lddx);
    g.QH = H; g.QW = W; g.AH = OH; g.AW = OW; g.sy = stride; g.sx = stride; g.T = K * K; g.Cc = Cout; g.zrow = Mout * ldA;
    for (int ky = 0; ky < K; ++ky)
        for (int kx = 0; kx < K; ++kx) { g.tdy[ky * K + kx] = ky - pad; g.tdx[ky * K + kx] = kx - pad; }
    return x3_launch_auto(g, s);
}

// slab [splits][n] summed in split order into dw [n] (+= when accumulate)
__global__ __launch_bounds__(256) void x3_slab_reduce_kernel(const float* __restrict__ slab, int splits, long n4, float* __restrict__ dw, int accumulate) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(slab)[i];
        for (int y = 1; y < splits; ++y) { const float4 u = reinterpret_cast<const float4*>(slab)[(long)y * n4 + i]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        if (accumulate) { const float4 u = reinterpret_cast<const float4*>(dw)[i]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        reinterpret_cast<float4*>(dw)[i] = v;
    }
}
// dw [Cin][K][K][Cout] (+)= the weight gradient of the general form: sum over input pixels m of x[m][ci] dy[pixel(m, tap)][co] -- x3gemm_tn with the Q operand
// gathered per tap (template IQ), the reduction split over the chip, the splits summed into dw
int pdf_internal_x3_deconv_general_bwd_weight(const float* x, const float* dy, float* dw, float* ws, int N, int H, int W, int Cin, int Cout,
                                              int K, int stride, int pad, int OH, int OW, int lddy, int accumulate, hipStream_t s) {
    const long Min = (long)N * H * W, Mout = (long)N * OH * OW, NT = (long)K * K * Cout;
    const long ldP = x3_ld(Cin), ldQ = x3_ld(Cout);
    unsigned short* p3 = reinterpret_cast<unsigned short*>(ws);
    float* wq = ws + ((3 * Min * ldP + 1) / 2 + 3) / 4 * 4;
    unsigned short* q3 = reinterpret_cast<unsigned short*>(wq);
    float* slab = wq + ((3 * (Mout + 1) * ldQ + 1) / 2 + 3) / 4 * 4;
    hipLaunchKernelGGL(x3_split_rows_kernel, dim3(grid_for(Min * Cin / 8, 256, 256 * 16)), dim3(256), 0, s, x, (long)Cin, p3, ldP, Min * ldP, Min, Cin / 8);
    {
        KTimer kt("x3_split_rows_kernel", 0.0, 10.0 * Mout * Cout, s);
        hipLaunchKernelGGL(x3_split_rows_kernel, dim3(grid_for(Mout * Cout / 8, 256, 256 * 16)), dim3(256), 0, s, dy, (long)lddy, q3, ldQ, (Mout + 1) * ldQ, Mout, Cout / 8);
    }
    if (int rc = x3_zero_rows(q3, Mout, ldQ, s)) return rc;
    PDF_LAUNCH_CHECK();
    const int splits = x3_general_wgrad_splits(Min, Cin, NT);
    X3Gemm g = {};
    g.A = p3; g.B = q3; g.C = splits > 1 ? slab : dw;
    g.csA = Min * ldP; g.csB = (Mout + 1) * ldQ; g.lda = (int)ldP; g.ldb = (int)ldQ;
    g.M = Cin; g.N = (int)NT; g.K = (int)Min; g.batch = 1;
    g.rows_per_split = cdiv(cdiv(Min, splits), 32) * 32; g.splits = cdiv(Min, g.rows_per_split);
    g.accum = (g.splits == 1) ? accumulate : 0;
    g.QH = H; g.QW = W; g.AH = OH; g.AW = OW; g.sy = stride; g.sx = stride; g.T = K * K; g.Cc = Cout; g.zrow = Mout * ldQ;
    g.lw = __builtin_ctz(W); g.lhw = __builtin_ctz(H * W);
    for (int ky = 0; ky < K; ++ky)
        for (int kx = 0; kx < K; ++kx) { g.tdy[ky * K + kx] = ky - pad; g.tdx[ky * K + kx] = kx - pad; }
    if (g.splits == 1) g.C = dw;
    if (int rc = x3_launch_tn_deconv(g, s)) return rc;
    if (g.splits > 1) {
        KTimer kt("x3_slab_reduce_kernel", 0.0, 4.0 * (g.splits + 1) * Cin * NT, s);
        hipLaunchKernelGGL(x3_slab_reduce_kernel, dim3(grid_for((long)Cin * NT / 4)), dim3(256), 0, s, slab, g.splits, (long)Cin * NT / 4, dw, accumulate);
        PDF_LAUNCH_CHECK();
    }
    return 0;
}
