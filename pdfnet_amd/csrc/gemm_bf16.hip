// bf16-input MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate) forms of the two GEMM kernels of gemm.hip -- BASELINE configs 4/5
// (mixed precision: fp32 master weights and fp32 activations in HBM, operands rounded to bf16 -- round-to-nearest-even,
// v_cvt_pk_bf16_f32 -- while they are staged into LDS; 16x the fp32 MFMA rate, so these kernels are L2 / HBM bound).
//
// LDS images (bf16), K-step 64:
//   ROW operand (contiguous along k in HBM: NHWC pixels, [N][K] weights): tile[row][k], row stride 144 B; a lane reads its
//       MFMA fragment (8 consecutive k of row lane&31, k-offset 8*(lane>>5)) with one ds_read_b128 -- the 16 lanes of a
//       read group hit 16 different 16-byte slots of the 256-byte bank row (conflict-free, guide section 2).
//       Staging: 2 x float4 -> 8 bf16 -> one ds_write_b128.
//   COL operand (contiguous along the tile-row index: [K][N] weights, both operands of a weight gradient): tile[k][row]
//       exactly as it lies in HBM, row stride 2*rows + 64 B (consecutive k rows start 16 banks apart); staging is one
//       float4 -> 4 bf16 -> one coalesced ds_write_b64, and the fragment is fetched with two ds_read_b64_tr_b16 (the
//       CDNA4 transposing LDS read, guide T10: lane l receives column l&31, k = 8*(l>>5) + 0..3; tools/probe/tr_probe.hip).
//       (A first version transposed in registers and wrote tile[row][k] with ds_write_b64: 16-way bank conflicts -- the
//       weight-gradient kernel ran at 54-80 TFLOP/s, below the fp32 kernel.)
#include "gemm_common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));      // raw bf16 bits of 8 / 4 elements (shadow operands)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

static __device__ __attribute__((aligned(32))) float g_zero32[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ float4 as_f4(const u32x4& v) { return *reinterpret_cast<const float4*>(&v); }
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    f32x2 f = {a, b};
    bf16x2 v = __builtin_convertvector(f, bf16x2);
    return *reinterpret_cast<uint32_t*>(&v);
}

constexpr int BK16 = 64;                         // K-step (bf16 elements)
constexpr int LROW = BK16 * 2 + 16;              // ROW image: row stride in bytes
constexpr int col_stride(int C) { return C * 2 + 64; }          // COL image [k][C]: row stride in bytes

// fragment of a ROW image: rows r0 .. r0+31, k-sub-step ks (16 k each)
__device__ __forceinline__ bf16x8 frag_row(const unsigned char* img, int r0, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + (r0 + (lane & 31)) * LROW + ks * 32 + (lane >> 5) * 16);
}
// fragment of a COL image with C columns: columns c0 .. c0+31, k-sub-step ks (two transposing reads of 4 k each)
template <int C>
__device__ __forceinline__ bf16x8 frag_col(const unsigned char* img, int c0, int ks, int lane) {
    constexpr int S = col_stride(C);
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const unsigned char* a = img + (ks * 16 + 8 * (g >> 1) + q) * S + (c0 + 16 * (g & 1) + 4 * p) * 2;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * S));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return *reinterpret_cast<bf16x8*>(&v);
}

__device__ __forceinline__ void st_row8(unsigned char* dst, const float4& lo, const float4& hi) {
    uint4 v = {pk_bf16(lo.x, lo.y), pk_bf16(lo.z, lo.w), pk_bf16(hi.x, hi.y), pk_bf16(hi.z, hi.w)};
    *reinterpret_cast<uint4*>(dst) = v;
}
__device__ __forceinline__ void st_col4(unsigned char* dst, const float4& v) {
    *reinterpret_cast<uint2*>(dst) = uint2{pk_bf16(v.x, v.y), pk_bf16(v.z, v.w)};
}

// ---------------------------------------------------------------------------------------------
// Implicit GEMM (see gemm.hip igemm_nt for the contraction and the descriptor).  Requires the FAST conditions of the fp32
// kernel (16-byte aligned rows, Cin % 8 == 0); a tap's channels are walked in steps of 64, chunks past Cin read zeros.
// A is always a ROW operand; B is ROW ([N][K] storage) or COL ([K][N] storage, KN).
// SA / SB: the A / B operand comes as a bf16 shadow (IGemm::A16 / B16) -- compile-time, because a run-time branch around the
// operand loads makes the compiler drain them one by one (measured: 672 -> 376 img/s with `if (A16 != nullptr)` in the loop)
template <int BM, int BN, int WM, int WN, bool KN, bool SHA, bool SHB>
__global__ __launch_bounds__(256, 2) void igemm_bf16_kernel(const IGemm g) {
    constexpr int BK = BK16;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int CPR = BK / 8;                          // 8-float chunks per ROW tile row
    constexpr int RA = BM * CPR / 256, RB = BN * CPR / 256;
    constexpr int CG = BN / 4;                           // COL: float4 column groups per k row
    constexpr int KPP = 256 / CG;                        // COL: k rows per pass of the block
    constexpr int NCB = BK / KPP;                        // COL: passes (= float4 loads per thread)
    constexpr int SB = col_stride(BN);
    constexpr int ABYTES = BM * LROW;
    constexpr int BBYTES = KN ? BK * SB : BN * LROW;
    constexpr int TILE = ABYTES + BBYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const float* __restrict__ Ap = g.A; const float* __restrict__ Bp = g.B; const float* __restrict__ biasp = g.bias;
    float* __restrict__ Cp = g.C;
    const unsigned short* __restrict__ A16 = reinterpret_cast<const unsigned short*>(g.A16);
    const unsigned short* __restrict__ B16 = reinterpret_cast<const unsigned short*>(g.B16);
    if (blockIdx.y) { Ap += g.gsA; Cp += g.gsC; Bp = g.B1; biasp = g.bias1; if (A16) A16 += g.gsA; B16 = reinterpret_cast<const unsigned short*>(g.B116); }
    const unsigned short* const z16 = reinterpret_cast<const unsigned short*>(g_zero32);
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    int tmi, tni;
    xcd_tile(blockIdx.x, ntm * ntn, ntn, tmi, tni, g.gm);
    const int m0 = tmi * BM, n0 = tni * BN;

    // ---- A: ROW staging.  chunk q = tid + 256 i -> (row q / CPR, chunk q % CPR)
    const int arow = tid / CPR, ach = (tid % CPR) * 8;
    long abase[RA]; int iy0[RA], ix0[RA]; bool aval[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int r = m0 + arow + i * (256 / CPR);
        aval[i] = r < g.M;
        if (g.plain_in) { abase[i] = (long)r * g.lda; iy0[i] = 0; ix0[i] = 0; }
        else {
            const int hw = g.QH * g.QW;
            const int ni = r / hw, rem = r - ni * hw;
            const int qy = rem / g.QW, qx = rem - qy * g.QW;
            iy0[i] = qy * g.sy; ix0[i] = qx * g.sx;
            abase[i] = (long)ni * g.H * g.W * g.lda;
        }
    }
    // ---- B
    long bbase[RB > 0 ? RB : 1]; bool bval[RB > 0 ? RB : 1];
    const int bcg = (tid % CG) * 4, bk0 = tid / CG;      // COL: 4 columns n, k rows bk0 + KPP * pass
    const bool bcol_ok = n0 + bcg < g.N;
    if (!KN) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int n = n0 + arow + i * (256 / CPR);
            bval[i] = n < g.N;
            bbase[i] = (long)n * g.ldb;
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int spt = (g.Cin + BK - 1) / BK;               // K-steps per tap
    const int nk = g.T * spt;
    // DEEP (both operands are bf16 shadows: 8 x 16-byte loads per thread and K-step): TWO register sets, tiles are loaded two
    // K-steps ahead.  A K-step is only 16 MFMAs (512 cycles) per wave and two waves share a SIMD, so a one-step prefetch leaves
    // most of the ~2-4k cycle L2 / HBM latency exposed (the kernel ran at 12 % of the bf16 MFMA peak).  fp32 sources need 16
    // float4 per set: two sets would not fit beside the accumulators.
    constexpr bool DEEP = SHA && SHB;
    constexpr int NS = DEEP ? 2 : 1;
    float4 ra[SHA ? 1 : RA][2];
    float4 rb[(KN || SHB) ? 1 : (RB > 0 ? RB : 1)][2];
    float4 rc[(KN && !SHB) ? NCB : 1];
    u32x4 ra16[NS][SHA ? RA : 1];                        // shadow operands: raw bf16 bits, 8 (ROW) / 4 (COL) elements per load
    u32x4 rb16[NS][(!KN && SHB) ? (RB > 0 ? RB : 1) : 1];
    u32x2 rc16[NS][(KN && SHB) ? NCB : 1];
    int nt_tap = 0, nt_ci = 0, ddy = g.dy[0], ddx = g.dx[0], wbase = g.wt[0] * (KN ? g.btap : g.Cin);
    // Operands through buffer descriptors (see gemm.hip igemm_nt BUF): per-row byte offsets fixed for the whole kernel, a masked
    // element = the out-of-range offset 0xffffffff (the hardware returns zeros); per K-step and load: one add, one select, the load.
    // (The flat form -- 64-bit multiply-adds under exec-mask branches, the zero word's address re-read from the GOT -- cost ~250
    // instructions per K-step against 16 MFMAs of 32 cycles: the kernels were issue-bound at ~12 % of the bf16 MFMA peak.)
    constexpr unsigned EA = SHA ? 2u : 4u, EB = SHB ? 2u : 4u;          // element sizes of the A / B sources
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(SHA ? (void*)A16 : (void*)Ap, 0, SHA ? g.abytes / 2 : g.abytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(SHB ? (void*)B16 : (void*)Bp, 0, SHB ? g.bbytes / 2 : g.bbytes, 0x00020000);
    unsigned aoffB[RA], boffB[RB > 0 ? RB : 1], coffB[NCB]; bool aok[RA];
    int tapoff = 0;
    auto tap_valid = [&]() {
        tapoff = g.plain_in ? 0 : (ddy * g.W + ddx) * g.lda;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int iy = iy0[i] + ddy, ix = ix0[i] + ddx;
            aok[i] = aval[i] && (g.plain_in || (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W));
        }
    };
#pragma unroll
    for (int i = 0; i < RA; ++i) aoffB[i] = (unsigned)(abase[i] + ((long)iy0[i] * g.W + ix0[i]) * g.lda + ach) * EA;
    if (!KN) {
#pragma unroll
        for (int i = 0; i < RB; ++i) boffB[i] = (unsigned)(bbase[i] + ach) * EB;
    } else {
#pragma unroll
        for (int u = 0; u < NCB; ++u) coffB[u] = (unsigned)((long)(bk0 + u * KPP) * g.ldb + n0 + bcg) * EB;
    }
    tap_valid();
    auto gload = [&](auto SET) {
        constexpr int S = decltype(SET)::value;
        const bool kin = nt_ci + ach < g.Cin;             // Cin % 8 == 0: a chunk is inside or outside as a whole
        const unsigned sa = (unsigned)(tapoff + nt_ci) * EA;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            // (masked: 0xffffffe0, so that the second half's offset + 16 is out of range as well -- extents are < 0xfffffed8)
            const unsigned vo = (aok[i] && kin) ? aoffB[i] + sa : 0xffffffe0u;
            if constexpr (SHA) ra16[S][i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, 0, 0);      // 8 bf16, stored to LDS as they are
            else {
                ra[i][0] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsA, vo, 0, 0));
                ra[i][1] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsA, vo + 16u, 0, 0));
            }
        }
        if (!KN) {
            const unsigned sb = (unsigned)(wbase + nt_ci) * EB;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const unsigned vo = (bval[i] && kin) ? boffB[i] + sb : 0xffffffe0u;
                if constexpr (SHB) rb16[S][i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, vo, 0, 0);
                else {
                    rb[i][0] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsB, vo, 0, 0));
                    rb[i][1] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsB, vo + 16u, 0, 0));
                }
            }
        } else {
            const unsigned sb = (unsigned)(nt_ci * g.ldb + wbase) * EB;
#pragma unroll
            for (int u = 0; u < NCB; ++u) {
                const bool ok = bcol_ok && nt_ci + bk0 + u * KPP < g.Cin;
                const unsigned vo = ok ? coffB[u] + sb : 0xffffffffu;
                if constexpr (SHB) rc16[S][u] = __builtin_amdgcn_raw_buffer_load_b64(rsB, vo, 0, 0);   // 4 bf16
                else rc[u] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsB, vo, 0, 0));
            }
        }
        nt_ci += BK;
        if (nt_ci >= g.Cin && nt_tap + 1 < g.T) {
            ++nt_tap; nt_ci = 0;
            ddy = g.dy[nt_tap]; ddx = g.dx[nt_tap]; wbase = g.wt[nt_tap] * (KN ? g.btap : g.Cin);
            tap_valid();
        }
    };
    auto lstore = [&](int buf, auto SET) {
        constexpr int S = decltype(SET)::value;
        unsigned char* as = smem + buf * TILE;
        unsigned char* bs = as + ABYTES;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            unsigned char* d = as + (arow + i * (256 / CPR)) * LROW + ach * 2;
            if constexpr (SHA) *reinterpret_cast<u32x4*>(d) = ra16[S][i]; else st_row8(d, ra[i][0], ra[i][1]);
        }
        if (!KN) {
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                unsigned char* d = bs + (arow + i * (256 / CPR)) * LROW + ach * 2;
                if constexpr (SHB) *reinterpret_cast<u32x4*>(d) = rb16[S][i]; else st_row8(d, rb[i][0], rb[i][1]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < NCB; ++u) {
                unsigned char* d = bs + (bk0 + u * KPP) * SB + bcg * 2;
                if constexpr (SHB) *reinterpret_cast<u32x2*>(d) = rc16[S][u];
                else st_col4(d, rc[u]);
            }
        }
    };
    auto compute = [&](int buf) {
        const unsigned char* as = smem + buf * TILE;
        const unsigned char* bs = as + ABYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = frag_row(as, (wm * TM + i) * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = KN ? frag_col<BN>(bs, (wn * TN + j) * 32, ks, lane) : frag_row(bs, (wn * TN + j) * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, NS - 1>;
    if constexpr (DEEP) {
        // tile t lives in register set t & 1 until it is written to LDS buffer t & 1 at the end of iteration t - 1;
        // iteration t issues the loads of tile t + 2 into the set tile t has just left
        // (the loads are issued UNCONDITIONALLY -- past the last tile every lane reads the zero word: a load under `if (more)`
        // makes the compiler's s_waitcnt pass assume it may not have been issued and wait vmcnt(7..0) for the older set, which
        // drains the set just issued as well)
        gload(S0{});
        gload(S1{});
        lstore(0, S0{});
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            gload(S0{});
            compute(0);
            lstore(1, S1{});
            __syncthreads();
            if (kt + 1 >= nk) break;
            gload(S1{});
            compute(1);
            lstore(0, S0{});
            __syncthreads();
        }
    } else {
        gload(S0{});
        lstore(0, S0{});
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) gload(S0{});
            compute(cur);
            if (kt + 1 < nk) lstore(cur ^ 1, S0{});
            __syncthreads();
            cur ^= 1;
        }
    }

    if (g.cbytes != 0 && m0 + BM <= g.M) {                 // whole tile of a dense row-major output: buffer stores + BatchNorm statistics
        lean_epilogue<TM, TN, WM, WN, BN>(acc, g, Cp, biasp, m0, n0, tmi, wm, wn, lane, tid, reinterpret_cast<float*>(smem));
        return;
    }
    // epilogue (same as igemm_nt): lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + (lane & 31);
        const bool cok = col < g.N;
        int co = col, padd_y = 0, padd_x = 0;
        if (g.ps_cout > 0) {
            const int tap = col / g.ps_cout;
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
                        const int hw = g.QH * g.QW;
                        const int ni = row / hw, rem = row - ni * hw;
                        const int qy = rem / g.QW, qx = rem - qy * g.QW;
                        const int oy = qy * g.osy + g.ooy + padd_y, ox = qx * g.osx + g.oox + padd_x;
                        o = (((long)ni * g.OH + oy) * g.OW + ox) * g.ldc + co;
                    }
                    if (g.accum) v += Cp[o];
                    Cp[o] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool SA_, bool SB_>
static void launch_tile_bf16_s(const IGemm& g, dim3 grid, hipStream_t s) {
    if (g.b_kn) hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, WM, WN, true, SA_, SB_>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, WM, WN, false, SA_, SB_>), grid, dim3(256), 0, s, g);
}
template <int BM, int BN, int WM, int WN>
static void launch_tile_bf16(const IGemm& g, dim3 grid, hipStream_t s) {
    const bool sa = g.A16 != nullptr, sb = g.B16 != nullptr;
    char nm[96];
    snprintf(nm, sizeof nm, "igemm_bf16_kernel<%d, %d, %d, %d, %s, %s, %s>", BM, BN, WM, WN, g.b_kn ? "true" : "false", sa ? "true" : "false", sb ? "true" : "false");
    const double a = g.plain_in ? (double)g.M * g.Cin : (double)(g.M / max(1, g.QH * g.QW)) * g.H * g.W * g.Cin;
    KTimer kt(nm, 2.0 * g.M * g.N * g.K * grid.y, grid.y * ((sa ? 2.0 : 4.0) * a + (sb ? 2.0 : 4.0) * g.N * g.K + 4.0 * g.M * g.N * (g.accum ? 2 : 1)), s);
    if (sa && sb) launch_tile_bf16_s<BM, BN, WM, WN, true, true>(g, grid, s);
    else if (sa) launch_tile_bf16_s<BM, BN, WM, WN, true, false>(g, grid, s);
    else if (sb) launch_tile_bf16_s<BM, BN, WM, WN, false, true>(g, grid, s);
    else launch_tile_bf16_s<BM, BN, WM, WN, false, false>(g, grid, s);
}

// rows of the tile launch_igemm_bf16 picks (the caller plans the statistics partials of IGemm::stat with it)
int igemm_bf16_tile_rows(const IGemm& g, int groups) {
    const long t128 = (long)cdiv(g.M, 128) * cdiv(g.N, 128) * groups;
    if (g.N > 64 && t128 >= 192) return 128;
    if (g.N <= 64 && (long)cdiv(g.M, 128) * groups >= 192) return 128;
    return 64;
}

int launch_igemm_bf16(const IGemm& g, hipStream_t s, int groups) {
    if (g.Cin % 8 != 0 || g.lda % 4 != 0 || g.ldb % 4 != 0) return 0;
    if (g.abytes == 0 || g.bbytes == 0) return 0;        // an operand of 4 GiB or more: the fp32 kernels' flat-address form takes it
    if (g.b_kn && (g.N % 4 != 0 || g.btap % 4 != 0)) return 0;
    const long t128 = (long)cdiv(g.M, 128) * cdiv(g.N, 128) * groups;
    static const int dma = getenv("PDF_BF16_DMA") ? atoi(getenv("PDF_BF16_DMA")) : 1;       // LDS-DMA form (gemm_dma.hip) when both operands are K-contiguous bf16 shadows: variant + 1, 0 = off
    static const int dma_tiles = getenv("PDF_BF16_DMA_TILES") ? atoi(getenv("PDF_BF16_DMA_TILES")) : 3;       // 1: 128x128 only, 2: + 128x64, 3: + 64x64
    if (dma > 0 && g.b_kn && g.B16T != nullptr && g.A16 != nullptr && groups == 1) {
        // backward-data with a transposed weight shadow: the same contraction as a [N][K] row operand -> the LDS-DMA kernel
        IGemm t = g;
        t.b_kn = 0; t.B16 = g.B16T; t.ldb = g.ldbT; t.btap = 0;
        const int tile = (g.N > 64 && t128 >= 192) ? 128 : (g.N <= 64 && (long)cdiv(g.M, 128) * groups >= 192) ? (dma_tiles >= 2 ? 12864 : 0) : (dma_tiles >= 3 ? 64 : 0);
        if (tile != 0 && launch_igemm_bf16_dma(t, tile, dma - 1, groups, s)) { hipError_t e = hipGetLastError(); return e == hipSuccess ? 1 : -(int)e; }
    }
    if (g.N > 64 && t128 >= 192) {
        if (!(dma > 0 && launch_igemm_bf16_dma(g, 128, dma - 1, groups, s)))
            launch_tile_bf16<128, 128, 2, 2>(g, dim3(cdiv(g.M, 128) * cdiv(g.N, 128), groups), s);
    } else if (g.N <= 64 && (long)cdiv(g.M, 128) * groups >= 192) {
        if (!(dma > 0 && dma_tiles >= 2 && launch_igemm_bf16_dma(g, 12864, dma - 1, groups, s)))
            launch_tile_bf16<128, 64, 4, 1>(g, dim3(cdiv(g.M, 128) * cdiv(g.N, 64), groups), s);
    } else if (!(dma > 0 && dma_tiles >= 3 && launch_igemm_bf16_dma(g, 64, dma - 1, groups, s)))
        launch_tile_bf16<64, 64, 2, 2>(g, dim3(cdiv(g.M, 64) * cdiv(g.N, 64), groups), s);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 1 : -(int)e;
}

// ---------------------------------------------------------------------------------------------
// Weight gradient (see gemm.hip wgemm_tn): dW[i][wt[t]*Cq + c] = sum_m P[m][i] * Q[pos(m,t)][c]; the reduction index m is
// the slow axis of both operands in HBM, so both are COL operands (tile[m][i], tile[m][j]) read with transposing loads.
// BI x BJ tile, K-step 64 pixels, M split over blockIdx.y into slabs (summed by reduce_slabs of gemm.hip).
template <int BI, int BJ, bool SP16, bool SQ16>
__global__ __launch_bounds__(256, 2) void wgemm_bf16_kernel(const WGemm g) {
    constexpr int BK = BK16;
    constexpr int WN = 2, TM = BI / 2 / 32, TN = BJ / 2 / 32;
    constexpr int GP = SP16 ? 8 : 4, GQ = SQ16 ? 8 : 4;  // columns per 16-byte load: 4 fp32 or 8 bf16 (shadow operand)
    constexpr int CGP = BI / GP, CGQ = BJ / GQ;          // column groups per k row
    constexpr int KPP_P = 256 / CGP, KPP_Q = 256 / CGQ;  // k rows per pass
    constexpr int NP = BK / KPP_P, NQ = BK / KPP_Q;      // 16-byte loads per thread and K-step (a shadow operand needs half as many)
    constexpr int SP = col_stride(BI), SQ = col_stride(BJ);
    constexpr int PBYTES = BK * SP, TILE = PBYTES + BK * SQ;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE];
    __shared__ float bred[KPP_P * BI];                   // bias partials: one row of BI columns per k row group

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ Pp = g.P; const float* __restrict__ Qp = g.Q; float* slabp = g.slab; float* bslabp = g.bslab;
    const unsigned short* __restrict__ P16 = reinterpret_cast<const unsigned short*>(g.P16);
    const unsigned short* __restrict__ Q16 = reinterpret_cast<const unsigned short*>(g.Q16);
    if (blockIdx.z) { Pp += g.gsP; Qp += g.gsQ; slabp = g.slab1; bslabp = g.bslab1; if (P16) P16 += g.gsP; if (Q16) Q16 += g.gsQ; }
    const unsigned short* const z16 = reinterpret_cast<const unsigned short*>(g_zero32);
    const int wm = wave / WN, wn = wave % WN;
    const int NJ = g.T * g.Cq;
    const int nti = (g.NI + BI - 1) / BI, ntj = (NJ + BJ - 1) / BJ;
    int ti, tj;
    if (g.tap_major) {
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
    const bool do_bias = !SP16 && bslabp != nullptr && tj == 0;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const int ms = blockIdx.y * g.rows_per_split;
    const int me = min(g.M, ms + g.rows_per_split);

    const int pcg = (tid % CGP) * GP, pk0 = tid / CGP;
    const int qcg = (tid % CGQ) * GQ, qk0 = tid / CGQ;
    const bool pcol_ok = i0 + pcg < g.NI;
    const int jcol = j0 + qcg;
    const bool qcol_ok = jcol < NJ;
    const int qtap = qcol_ok ? jcol / g.Cq : 0;
    const int qch = jcol - qtap * g.Cq;
    const int tdy = g.dy[qtap], tdx = g.dx[qtap];
    // (image, qy, qx) of each Q row this thread loads, advanced by BK rows per step -- no divisions in the loop
    int q_ni[NQ], q_y[NQ], q_x[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int m = ms + qk0 + u * KPP_Q, hw = g.QH * g.QW;
        q_ni[u] = m / hw;
        const int rem = m - q_ni[u] * hw;
        q_y[u] = rem / g.QW;
        q_x[u] = rem - q_y[u] * g.QW;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    constexpr bool DEEP = SP16 && SQ16;                    // two register sets, loads two K-steps ahead (see igemm_bf16_kernel)
    constexpr int NS = DEEP ? 2 : 1;
    float4 rp[SP16 ? 1 : NP], rq[SQ16 ? 1 : NQ];
    u32x4 rp16[NS][SP16 ? NP : 1], rq16[NS][SQ16 ? NQ : 1];        // 8 bf16 each
    // operands through buffer descriptors: 32-bit byte offsets, a masked element = the out-of-range offset (see igemm_bf16_kernel)
    constexpr unsigned EP = SP16 ? 2u : 4u, EQ = SQ16 ? 2u : 4u;
    const auto rsP = __builtin_amdgcn_make_buffer_rsrc(SP16 ? (void*)P16 : (void*)Pp, 0, SP16 ? g.pbytes / 2 : g.pbytes, 0x00020000);
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc(SQ16 ? (void*)Q16 : (void*)Qp, 0, SQ16 ? g.qbytes / 2 : g.qbytes, 0x00020000);
    auto gload = [&](int mb, auto SET) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int m = mb + pk0 + u * KPP_P;
            const unsigned vo = (m < me && pcol_ok) ? (unsigned)(m * g.ldp + i0 + pcg) * EP : 0xffffffffu;
            if constexpr (SP16) rp16[S][u] = __builtin_amdgcn_raw_buffer_load_b128(rsP, vo, 0, 0);       // 8 bf16
            else rp[u] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsP, vo, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int m = mb + qk0 + u * KPP_Q;
            unsigned vo;
            if (g.plain_q) vo = (m < me && qcol_ok) ? (unsigned)(m * g.ldq + qch) * EQ : 0xffffffffu;
            else {
                const int iy = q_y[u] * g.sy + tdy, ix = q_x[u] * g.sx + tdx;
                const bool ok = m < me && qcol_ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                vo = ok ? (unsigned)(((q_ni[u] * g.H + iy) * g.W + ix) * g.ldq + qch) * EQ : 0xffffffffu;
                q_x[u] += BK;
                while (q_x[u] >= g.QW) {
                    q_x[u] -= g.QW;
                    if (++q_y[u] == g.QH) { q_y[u] = 0; ++q_ni[u]; }
                }
            }
            if constexpr (SQ16) rq16[S][u] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, vo, 0, 0);
            else rq[u] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsQ, vo, 0, 0));
        }
    };
    auto lstore = [&](int buf, auto SET) {
        constexpr int S = decltype(SET)::value;
        unsigned char* ps = smem + buf * TILE;
        unsigned char* qs = ps + PBYTES;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            unsigned char* d = ps + (pk0 + u * KPP_P) * SP + pcg * 2;
            if constexpr (SP16) {
                *reinterpret_cast<u32x4*>(d) = rp16[S][u];   // (never together with the bias partials: launch_wgemm drops P16 then)
            } else {
                st_col4(d, rp[u]);
                if (do_bias) { bsum[0] += rp[u].x; bsum[1] += rp[u].y; bsum[2] += rp[u].z; bsum[3] += rp[u].w; }
            }
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            unsigned char* d = qs + (qk0 + u * KPP_Q) * SQ + qcg * 2;
            if constexpr (SQ16) *reinterpret_cast<u32x4*>(d) = rq16[S][u];
            else st_col4(d, rq[u]);
        }
    };

    auto compute = [&](int buf) {
        const unsigned char* ps = smem + buf * TILE;
        const unsigned char* qs = ps + PBYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = frag_col<BI>(ps, (wm * TM + i) * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = frag_col<BJ>(qs, (wn * TN + j) * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, NS - 1>;
    if constexpr (DEEP) {
        gload(ms, S0{});                                    // (unconditional: rows past `me` read the zero word; see igemm_bf16_kernel)
        gload(ms + BK, S1{});
        lstore(0, S0{});
        __syncthreads();
        for (int mb = ms; mb < me; mb += 2 * BK) {
            gload(mb + 2 * BK, S0{});
            compute(0);
            lstore(1, S1{});
            __syncthreads();
            if (mb + BK >= me) break;
            gload(mb + 3 * BK, S1{});
            compute(1);
            lstore(0, S0{});
            __syncthreads();
        }
    } else {
        if (ms < me) { gload(ms, S0{}); lstore(0, S0{}); }
        __syncthreads();
        int cur = 0;
        for (int mb = ms; mb < me; mb += BK) {
            const bool more = mb + BK < me;
            if (more) gload(mb + BK, S0{});
            compute(cur);
            if (more) lstore(cur ^ 1, S0{});
            __syncthreads();
            cur ^= 1;
        }
    }

    float bval = 0.f;
    if (do_bias) {                                        // column sums of P (fp32, un-rounded): reduce the k-row groups' partials
#pragma unroll
        for (int e = 0; e < 4; ++e) bred[pk0 * BI + pcg + e] = bsum[e];
        __syncthreads();
        if (tid < BI) for (int k = 0; k < KPP_P; ++k) bval += bred[k * BI + tid];
    }
    wgemm_finish<TM, TN>(g, acc, i0, j0, wm, wn, lane, do_bias && tid < BI && i0 + tid < g.NI, bval, ti * ntj + tj, nti * ntj, reinterpret_cast<int*>(smem), i0 + BI <= g.NI);
}

// g carries the split / slab plan of launch_wgemm (gemm.hip); `small` = its 64 x 64 tile choice
int launch_wgemm_bf16(const WGemm& g, int splits, int groups, int small, hipStream_t s) {
    const int NJ = g.T * g.Cq;
    if (g.NI % 4 != 0 || g.Cq % 4 != 0 || g.ldp % 4 != 0 || g.ldq % 4 != 0 || g.rows_per_split % BK16 != 0) return 0;
    if (g.pbytes == 0 || g.qbytes == 0) return 0;        // an operand of 4 GiB or more: the fp32 kernels take it
    const bool sp = g.P16 != nullptr, sq = g.Q16 != nullptr;
    char nm[96];
    snprintf(nm, sizeof nm, "wgemm_bf16_kernel<%d, %d, %s, %s>", small ? 64 : 128, small ? 64 : 128, sp ? "true" : "false", sq ? "true" : "false");
    const double qel = g.plain_q ? (double)g.M * g.Cq : (double)(g.M / max(1, g.QH * g.QW)) * g.H * g.W * g.Cq;
    KTimer kt(nm, 2.0 * groups * g.M * g.NI * NJ, groups * ((sp ? 2.0 : 4.0) * g.M * g.NI + (sq ? 2.0 : 4.0) * qel + 4.0 * g.NI * NJ), s);
    if (small) {
        dim3 grid((unsigned)(cdiv(g.NI, 64) * cdiv(NJ, 64)), (unsigned)splits, (unsigned)groups);
        if (sp && sq) hipLaunchKernelGGL((wgemm_bf16_kernel<64, 64, true, true>), grid, dim3(256), 0, s, g);
        else if (sp) hipLaunchKernelGGL((wgemm_bf16_kernel<64, 64, true, false>), grid, dim3(256), 0, s, g);
        else if (sq) hipLaunchKernelGGL((wgemm_bf16_kernel<64, 64, false, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((wgemm_bf16_kernel<64, 64, false, false>), grid, dim3(256), 0, s, g);
    } else {
        dim3 grid((unsigned)(cdiv(g.NI, 128) * cdiv(NJ, 128)), (unsigned)splits, (unsigned)groups);
        if (sp && sq) hipLaunchKernelGGL((wgemm_bf16_kernel<128, 128, true, true>), grid, dim3(256), 0, s, g);
        else if (sp) hipLaunchKernelGGL((wgemm_bf16_kernel<128, 128, true, false>), grid, dim3(256), 0, s, g);
        else if (sq) hipLaunchKernelGGL((wgemm_bf16_kernel<128, 128, false, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((wgemm_bf16_kernel<128, 128, false, false>), grid, dim3(256), 0, s, g);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 1 : -(int)e;
}
