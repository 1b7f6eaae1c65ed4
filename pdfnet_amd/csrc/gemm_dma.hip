// igemm_dma: the fp32 implicit GEMM of gemm.hip (same IGemm descriptor, same results up to fp32 summation order) with its operand
// tiles moved global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds) instead of through registers.
//
// Why (round 4).  The register-staged 64x64 tile spends, per K-step of 32 and thread, 4 buffer loads, 16 ds_write_b32 (the padded
// [row][33] image cannot take 16-byte stores) and 32 ds_read_b32 against 16 MFMAs; with four blocks per CU the LDS store path
// (64 B/clk/CU for ds_write_b32, MI355X_MICROARCH.md "LDS") and the read issue are half busy, and the what-if build without loads
// and LDS stores ran 115-123 TFLOP/s against 93-103 complete (profiles/r03_whatif.txt).  Here:
//   * staging is 4 DMA instructions per thread and K-step, no VGPR round trip and NO LDS store instructions;
//   * the LDS image of a K-contiguous operand (A pixels, [N][K] weights) is [row][BK] UNPADDED -- what a DMA wave-instruction
//     writes is 64 consecutive 16-byte chunks -- with the chunk index XOR-swizzled by the row, chunk (r, c) at slot
//     r * CPR + (c ^ f(r)), so that the 16 lanes of a ds_read_b128 group ({0-3,12-15,20-27}, ...) hit 16 different 16-byte bank
//     groups;  the swizzle costs nothing: every DMA lane picks WHICH global chunk it fetches;
//   * one ds_read_b128 feeds FOUR MFMAs: v_mfma_f32_32x32x2_f32 takes one k per lane (lanes 0-31: k0, lanes 32-63: k1), and any
//     pairing of the K-step's k values works as long as A and B agree, so lanes 0-31 read chunk 2q and lanes 32-63 chunk 2q+1 and
//     register s of both serves MFMA (q, s): 8 reads per K-step instead of 32;
//   * a [K][N] operand (backward-data / transposed-conv weights) is staged as it lies, [k][BN], and read with ds_read_b32
//     (consecutive lanes, consecutive addresses) under the same k pairing;
//   * a ring of ST stages, tiles issued ST-1 K-steps ahead with counted s_waitcnt vmcnt, one s_barrier per K-step.
// Masking is the descriptor's range check (offset 0xffffffff -> the DMA writes zeros; tools/probe/buf_lds_probe.hip).
#include "gemm_common.h"
#include <cstdio>

// The DMA is issued from inline assembly.  Through the builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc 7.2 knows that the load
// writes LDS and puts an s_waitcnt vmcnt(0) in front of the next LDS read it cannot prove disjoint -- i.e. it drains the tile just
// issued and the ring stops prefetching (seen in the -S listing of the first version of this kernel).  From assembly the compiler sees
// neither the LDS write nor the VMEM operation; the counted s_waitcnt vmcnt below are the only waits, as intended.  M0 = LDS byte
// address of the wave's 1 KiB destination (lane l lands at M0 + 16 l).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 dma_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r = {(int)(unsigned)(a & 0xffffffffu), (int)(unsigned)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
    return r;
}
__device__ __forceinline__ void dma16(const i32x4& rsrc, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const float* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const float*)p;
}

template <int TM, int TN, int BK, int ST, bool KN>
__global__ __launch_bounds__(256) void igemm_dma(const IGemm g) {
    constexpr int WM = 2, WN = 2, BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int CPR = BK / 4;                          // 16-byte chunks per row of a K-contiguous image
    constexpr int NIA = BM * CPR / 256, NIB = BN * CPR / 256;       // DMA instructions per wave and K-step (A, B)
    constexpr int ASZ = BM * BK, BSZ = BN * BK;         // floats per stage
    __shared__ __attribute__((aligned(16))) float smem[ST * (ASZ + BSZ)];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const float* __restrict__ Ap = g.A; const float* __restrict__ Bp = g.B; const float* __restrict__ biasp = g.bias;
    float* __restrict__ Cp = g.C;
    if (blockIdx.y) { Ap += g.gsA; Cp += g.gsC; Bp = g.B1; biasp = g.bias1; }
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    int tmi, tni;
    xcd_tile(blockIdx.x, ntm * ntn, ntn, tmi, tni, g.gm);
    const int m0 = tmi * BM, n0 = tni * BN;

    const i32x4 rsA = dma_rsrc(Ap, g.abytes), rsB = dma_rsrc(Bp, g.bbytes);
    const unsigned lds0 = lds_addr_of(smem);

    // ---- A: this lane's rows (one per DMA instruction) and the chunk it fetches for them
    constexpr int RPI = 64 / CPR;                        // rows covered by one wave-instruction
    auto swz = [](int r) { return CPR == 8 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
    unsigned aoff[NIA]; int iy0[NIA], ix0[NIA]; bool aval[NIA], aok[NIA];
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        const int row = (wave * NIA + i) * RPI + lane / CPR;               // row of the tile
        const int kc = (lane % CPR) ^ swz(row);                            // global chunk that lands in slot row * CPR + lane % CPR
        const int r = m0 + row;
        aval[i] = r < g.M;
        long base;
        if (g.plain_in) { base = (long)r * g.lda; iy0[i] = 0; ix0[i] = 0; }
        else {
            const int hw = g.QH * g.QW;
            const int ni = r / hw, rem = r - ni * hw;
            const int qy = rem / g.QW, qx = rem - qy * g.QW;
            iy0[i] = qy * g.sy; ix0[i] = qx * g.sx;
            base = (long)ni * g.H * g.W * g.lda + ((long)iy0[i] * g.W + ix0[i]) * g.lda;
        }
        aoff[i] = (unsigned)(base + kc * 4) * 4u;
    }
    // ---- B
    unsigned boff[NIB]; bool bval[NIB];
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        if (KN) {                                                           // image [k][BN]: a wave-instruction = 64 / (BN/4) k-rows
            constexpr int CPK = BN / 4;
            const int slot = (wave * NIB + i) * 64 + lane;
            const int k = slot / CPK, n4 = (slot % CPK) * 4;
            bval[i] = n0 + n4 < g.N;
            boff[i] = (unsigned)((long)k * g.ldb + n0 + n4) * 4u;
        } else {
            const int row = (wave * NIB + i) * RPI + lane / CPR;
            const int kc = (lane % CPR) ^ swz(row);
            bval[i] = n0 + row < g.N;
            boff[i] = (unsigned)((long)(n0 + row) * g.ldb + kc * 4) * 4u;
        }
    }

    const int nk = g.K / BK;                              // (host: Cin % BK == 0, K == T * Cin)
    int kt0 = 0, kt1 = nk;
    if (g.ksteps > 0) { const int ks = g.ksteps * (32 / BK); kt0 = blockIdx.z * ks; kt1 = min(nk, kt0 + ks); }      // (host plans split-K in 32-wide steps)
    // tap state of the next tile to issue (tiles are issued strictly in order)
    int nt_tap = 0, nt_ci = kt0 * BK;
    if (g.T > 1) { nt_tap = nt_ci / g.Cin; nt_ci -= nt_tap * g.Cin; }
    int ddy = g.dy[nt_tap], ddx = g.dx[nt_tap], wbase = g.wt[nt_tap] * (KN ? g.btap : g.Cin);
    int tapoff = 0;
    auto tap_valid = [&]() {
        tapoff = g.plain_in ? 0 : (ddy * g.W + ddx) * g.lda * 4;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int iy = iy0[i] + ddy, ix = ix0[i] + ddx;
            aok[i] = aval[i] & (g.plain_in | (((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W)));       // (no short-circuit: branch-free)
        }
    };
    tap_valid();
    int nissued = kt0;
    auto issue = [&](int stage) {
        const bool live = nissued < kt1;                 // a tile past the end: zeros, no traffic (keeps the vmcnt bookkeeping uniform)
        ++nissued;
        const unsigned sa = lds0 + (unsigned)(stage * (ASZ + BSZ) + wave * NIA * 256) * 4u;
        const unsigned sb = lds0 + (unsigned)(stage * (ASZ + BSZ) + ASZ + wave * NIB * 256) * 4u;
        const unsigned soa = (unsigned)(tapoff + nt_ci * 4);
        const unsigned sob = KN ? (unsigned)((nt_ci * g.ldb + wbase) * 4) : (unsigned)((wbase + nt_ci) * 4);
#pragma unroll
        for (int i = 0; i < NIA; ++i)
            dma16(rsA, sa + i * 1024, (aok[i] && live) ? aoff[i] + soa : 0xffffffffu);
#pragma unroll
        for (int i = 0; i < NIB; ++i)
            dma16(rsB, sb + i * 1024, (bval[i] && live) ? boff[i] + sob : 0xffffffffu);
        nt_ci += BK;
        if (nt_ci >= g.Cin && nt_tap + 1 < g.T) {
            ++nt_tap; nt_ci = 0;
            ddy = g.dy[nt_tap]; ddx = g.dx[nt_tap]; wbase = g.wt[nt_tap] * (KN ? g.btap : g.Cin);
            tap_valid();
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment addresses (floats): row image: row * BK + ((2q + h) ^ f(row)) * 4, h = lane >> 5
    const int h = lane >> 5;
    int arow[TM], acx[TM], brow[TN], bcx[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int r = (wm * TM + i) * 32 + (lane & 31); arow[i] = r * BK; acx[i] = h ^ swz(r); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int r = (wn * TN + j) * 32 + (lane & 31); brow[j] = KN ? r : r * BK; bcx[j] = h ^ swz(r); }

#pragma unroll
    for (int p = 0; p < ST - 1; ++p) issue(p);
    int st = 0, stn = ST - 1;
    for (int kt = kt0; kt < kt1; ++kt) {
        // this wave's DMAs of tile kt have landed (the ST-2 younger tiles may still be in flight) ...
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * (NIA + NIB)) : "memory");
        asm volatile("s_barrier" ::: "memory");          // ... and everybody's; everybody has also finished reading tile kt - 1
        issue(stn);                                      // refill the stage tile kt - 1 used
        const float* as = smem + st * (ASZ + BSZ);
        const float* bs = as + ASZ;
        // every fragment of the K-step is requested up front (CPR/2 x (TM + TN) reads, 4 registers each): the reads of pair q + 1 ...
        // are in flight while the MFMAs of pair q run; the compiler places counted lgkmcnt waits (LDS returns in order)
        float4 a4[CPR / 2][TM], b4[CPR / 2][TN];
#pragma unroll
        for (int q = 0; q < CPR / 2; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a4[q][i] = *reinterpret_cast<const float4*>(as + arow[i] + ((acx[i] ^ (2 * q)) << 2));
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (KN) {
                    const float* p = bs + (8 * q + 4 * h) * BN + brow[j];
                    b4[q][j] = make_float4(p[0], p[BN], p[2 * BN], p[3 * BN]);
                } else b4[q][j] = *reinterpret_cast<const float4*>(bs + brow[j] + ((bcx[j] ^ (2 * q)) << 2));
            }
        }
        __builtin_amdgcn_sched_barrier(0);               // (the scheduler otherwise sinks each read pair back in front of its MFMAs)
#pragma unroll
        for (int q = 0; q < CPR / 2; ++q) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float av = s == 0 ? a4[q][i].x : s == 1 ? a4[q][i].y : s == 2 ? a4[q][i].z : a4[q][i].w;
                        const float bv = s == 0 ? b4[q][j].x : s == 1 ? b4[q][j].y : s == 2 ? b4[q][j].z : b4[q][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        }
        st = st == ST - 1 ? 0 : st + 1;
        stn = stn == ST - 1 ? 0 : stn + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");    // no LDS-DMA may outlive the workgroup (nor land in LDS the epilogue re-uses)

    if (g.ksteps > 0) {                                  // split-K: raw partial tile -> part[split][M][N] (splitk_finish adds bias / activation)
        float* pp = g.part + (long)blockIdx.z * g.M * g.N;
        const bool whole = m0 + BM <= g.M && (long)g.M * g.N < (1L << 29);
        const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void*)pp, 0, whole ? (unsigned)g.M * (unsigned)g.N * 4u : 0u, 0x00020000);
        const unsigned ldn4 = (unsigned)g.N * 4u;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
                if (whole) {
                    const unsigned vo = col < g.N ? (unsigned)((m0 + (wm * TM + i) * 32 + 4 * h) * g.N + col) * 4u : 0xffffffffu;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), rsP, vo, ((r & 3) + 8 * (r >> 2)) * ldn4, 0);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (col < g.N && row < g.M) pp[(long)row * g.N + col] = v[r];
                    }
                }
            }
        }
        return;
    }
    if (g.cbytes != 0 && m0 + BM <= g.M) {               // whole tile of a dense row-major output: the buffer-store epilogue (+ statistics)
        lean_epilogue<TM, TN, WM, WN, BN>(acc, g, Cp, biasp, m0, n0, tmi, wm, wn, lane, tid, smem);
        return;
    }
    // general epilogue (ragged last row block, strided / pixel-shuffled outputs): as igemm_nt
    StatAcc sacc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) sacc[j] = StatAcc{0.f, 0.f, 0.f, 0.f};
    const bool do_stat = g.stat != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
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
                const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
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
                    if (do_stat) stat_add(sacc[j], v);
                }
            }
        }
    }
    if (do_stat) stat_finish<TN, WM, WN, BN>(sacc, smem, g.stat, tmi, n0, g.N, wm, wn, lane, tid);
}


// ------------------------------------------------------------------------------------------------------------------------------
// igemm_bf16_dma: the bf16-input implicit GEMM (gemm_bf16.hip igemm_bf16_kernel) for launches whose operands BOTH come as bf16 shadows
// with K contiguous (forward convolutions / linears: A = activation rows, B = [N][K] weights).  The register-staged kernel is bound by
// its LDS store path -- 8 ds_write_b128 per thread and K-step (13 cycles each through a path the CU's SIMD pairs share) against 16 MFMAs of
// 32 cycles -- and by a barrier per 16 MFMAs with two blocks per CU (LDS index-active 0.15-0.22, MFMA busy 9-13 of 32).
// Here the 16-byte chunks (8 bf16 of one row) go global -> LDS by DMA, the chunk order inside a row XOR-swizzled by the row exactly as in
// igemm_dma above (a row of the K-step is CPR = 8 chunks = 64 elements, or 4 = 32), and a lane's MFMA operand -- 8 consecutive k of row
// lane & 31, k-offset 8 (lane >> 5) -- is ONE conflict-free ds_read_b128: chunk 2 ks + (lane >> 5) of its row.
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
template <int TM, int TN, int CPR, int ST>
__global__ __launch_bounds__(256) void igemm_bf16_dma(const IGemm g) {
    constexpr int WM = 2, WN = 2, BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int BK = CPR * 8;                          // bf16 elements per K-step
    constexpr int NIA = BM * CPR / 256, NIB = BN * CPR / 256;
    constexpr int ASZ = BM * CPR * 16, BSZ = BN * CPR * 16;        // bytes per stage
    __shared__ __attribute__((aligned(16))) unsigned char smem[ST * (ASZ + BSZ)];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const unsigned short* A16 = reinterpret_cast<const unsigned short*>(g.A16);
    const unsigned short* B16 = reinterpret_cast<const unsigned short*>(g.B16);
    const float* __restrict__ biasp = g.bias;
    float* __restrict__ Cp = g.C;
    if (blockIdx.y) { A16 += g.gsA; Cp += g.gsC; B16 = reinterpret_cast<const unsigned short*>(g.B116); biasp = g.bias1; }
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    int tmi, tni;
    xcd_tile(blockIdx.x, ntm * ntn, ntn, tmi, tni, g.gm);
    const int m0 = tmi * BM, n0 = tni * BN;
    const i32x4 rsA = dma_rsrc(A16, g.abytes / 2), rsB = dma_rsrc(B16, g.bbytes / 2);
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const unsigned char*)smem;

    constexpr int RPI = 64 / CPR;
    auto swz = [](int r) { return CPR == 8 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
    unsigned aoff[NIA]; int iy0[NIA], ix0[NIA], akc[NIA]; bool aval[NIA], aok[NIA];
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        const int row = (wave * NIA + i) * RPI + lane / CPR;
        akc[i] = ((lane % CPR) ^ swz(row)) * 8;                            // first element of the chunk this lane fetches
        const int r = m0 + row;
        aval[i] = r < g.M;
        long base;
        if (g.plain_in) { base = (long)r * g.lda; iy0[i] = 0; ix0[i] = 0; }
        else {
            const int hw = g.QH * g.QW;
            const int ni = r / hw, rem = r - ni * hw;
            const int qy = rem / g.QW, qx = rem - qy * g.QW;
            iy0[i] = qy * g.sy; ix0[i] = qx * g.sx;
            base = (long)ni * g.H * g.W * g.lda + ((long)iy0[i] * g.W + ix0[i]) * g.lda;
        }
        aoff[i] = (unsigned)(base + akc[i]) * 2u;
    }
    unsigned boff[NIB]; int bkc[NIB]; bool bval[NIB];
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const int row = (wave * NIB + i) * RPI + lane / CPR;
        bkc[i] = ((lane % CPR) ^ swz(row)) * 8;
        bval[i] = n0 + row < g.N;
        boff[i] = (unsigned)((long)(n0 + row) * g.ldb + bkc[i]) * 2u;
    }
    const int spt = (g.Cin + BK - 1) / BK;               // K-steps per tap (chunks past Cin read zeros; Cin % 8 == 0)
    const int nk = g.T * spt;
    int nt_tap = 0, nt_ci = 0;
    int ddy = g.dy[0], ddx = g.dx[0], wbase = g.wt[0] * g.Cin;
    int tapoff = 0;
    auto tap_valid = [&]() {
        tapoff = g.plain_in ? 0 : (ddy * g.W + ddx) * g.lda;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int iy = iy0[i] + ddy, ix = ix0[i] + ddx;
            aok[i] = aval[i] & (g.plain_in | (((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W)));
        }
    };
    tap_valid();
    int nissued = 0;
    auto issue = [&](int stage) {
        const bool live = nissued < nk;
        ++nissued;
        const unsigned sa = lds0 + (unsigned)(stage * (ASZ + BSZ) + wave * NIA * 1024);
        const unsigned sb = lds0 + (unsigned)(stage * (ASZ + BSZ) + ASZ + wave * NIB * 1024);
        const unsigned soa = (unsigned)(tapoff + nt_ci) * 2u, sob = (unsigned)(wbase + nt_ci) * 2u;
#pragma unroll
        for (int i = 0; i < NIA; ++i)
            dma16(rsA, sa + i * 1024, (aok[i] && live && nt_ci + akc[i] < g.Cin) ? aoff[i] + soa : 0xffffffffu);
#pragma unroll
        for (int i = 0; i < NIB; ++i)
            dma16(rsB, sb + i * 1024, (bval[i] && live && nt_ci + bkc[i] < g.Cin) ? boff[i] + sob : 0xffffffffu);
        nt_ci += BK;
        if (nt_ci >= g.Cin && nt_tap + 1 < g.T) {
            ++nt_tap; nt_ci = 0;
            ddy = g.dy[nt_tap]; ddx = g.dx[nt_tap]; wbase = g.wt[nt_tap] * g.Cin;
            tap_valid();
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int h = lane >> 5;
    int arow[TM], acx[TM], brow[TN], bcx[TN];            // byte offsets of the rows, swizzle ^ h
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int r = (wm * TM + i) * 32 + (lane & 31); arow[i] = r * CPR * 16; acx[i] = h ^ swz(r); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int r = (wn * TN + j) * 32 + (lane & 31); brow[j] = r * CPR * 16; bcx[j] = h ^ swz(r); }

#pragma unroll
    for (int p = 0; p < ST - 1; ++p) issue(p);
    int st = 0, stn = ST - 1;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * (NIA + NIB)) : "memory");
        asm volatile("s_barrier" ::: "memory");
        issue(stn);
        const unsigned char* as = smem + st * (ASZ + BSZ);
        const unsigned char* bs = as + ASZ;
        bf16x8v af[CPR / 2][TM], bfr[CPR / 2][TN];
#pragma unroll
        for (int ks = 0; ks < CPR / 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[ks][i] = *reinterpret_cast<const bf16x8v*>(as + arow[i] + ((acx[i] ^ (2 * ks)) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[ks][j] = *reinterpret_cast<const bf16x8v*>(bs + brow[j] + ((bcx[j] ^ (2 * ks)) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < CPR / 2; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
        st = st == ST - 1 ? 0 : st + 1;
        stn = stn == ST - 1 ? 0 : stn + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    if (g.cbytes != 0 && m0 + BM <= g.M) {
        lean_epilogue<TM, TN, WM, WN, BN>(acc, g, Cp, biasp, m0, n0, tmi, wm, wn, lane, tid, reinterpret_cast<float*>(smem));
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
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
                const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
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

template <int TM, int TN, int CPR, int ST>
static void launch_bf16_one(const IGemm& g, dim3 grid, hipStream_t s) {
    char nm[96] = "";
    snprintf(nm, sizeof nm, "igemm_bf16_dma<%d, %d, %d, %d>", TM, TN, CPR, ST);
    const double a = g.plain_in ? (double)g.M * g.Cin : (double)(g.M / max(1, g.QH * g.QW)) * g.H * g.W * g.Cin;
    KTimer kt(nm, 2.0 * g.M * g.N * g.K * grid.y, grid.y * (2.0 * a + 2.0 * g.N * g.K + 4.0 * g.M * g.N * (g.accum ? 2 : 1)), s);
    hipLaunchKernelGGL((igemm_bf16_dma<TM, TN, CPR, ST>), grid, dim3(256), 0, s, g);
}
// bf16 shadows of BOTH operands, [N][K] weights.  tile: 128 (128x128) | 64 (64x64) | 12864 (128x64).  -> 1 launched, 0 not taken
int launch_igemm_bf16_dma(const IGemm& g, int tile, int variant, int groups, hipStream_t s) {
    if (g.A16 == nullptr || g.B16 == nullptr || g.b_kn || g.abytes == 0 || g.bbytes == 0 || g.Cin % 8 != 0 || g.lda % 8 != 0 || g.ldb % 8 != 0) return 0;
    // (a bf16 output -- IGemm::C16, storage mode -- is written by lean_epilogue: the caller has checked whole tiles for the tile height this launcher uses too)
    if (groups > 1 && (g.B116 == nullptr || g.gsA % 8 != 0)) return 0;
    const int bm = tile == 64 ? 64 : 128, bn = tile == 128 ? 128 : 64;
    const dim3 grid((unsigned)(cdiv(g.M, bm) * cdiv(g.N, bn)), (unsigned)groups);
    if (tile == 128) {
        if (variant == 0) { launch_bf16_one<2, 2, 8, 2>(g, grid, s); return 1; }
        if (variant == 1) { launch_bf16_one<2, 2, 4, 4>(g, grid, s); return 1; }
        if (variant == 2) { launch_bf16_one<2, 2, 8, 3>(g, grid, s); return 1; }
        if (variant == 3) { launch_bf16_one<2, 2, 4, 3>(g, grid, s); return 1; }
    } else if (tile == 12864) {
        if (variant == 0 || variant == 2) { launch_bf16_one<2, 1, 8, 2>(g, grid, s); return 1; }
        if (variant == 1 || variant == 3) { launch_bf16_one<2, 1, 4, 4>(g, grid, s); return 1; }
    } else if (tile == 64) {
        if (variant == 0 || variant == 2) { launch_bf16_one<1, 1, 8, 3>(g, grid, s); return 1; }
        if (variant == 1 || variant == 3) { launch_bf16_one<1, 1, 4, 4>(g, grid, s); return 1; }
    }
    return 0;
}

// -> 1: launched, 0: shape not taken (the caller uses igemm_nt).  `tile`: 64 (64x64) | 128 (128x128) | 12864 (128x64) | 64128 (64x128)
int igemm_dma_tile_ok(const IGemm& g, int bk) {
    return g.abytes != 0 && g.bbytes != 0 && g.a_scale == nullptr && g.Cin % bk == 0 && g.K == g.T * g.Cin && g.lda % 4 == 0 && g.ldb % 4 == 0 &&
           (!g.b_kn || g.N % 4 == 0);
}

template <int TM, int TN, int BK, int ST>
static void launch_one(const IGemm& g, dim3 grid, hipStream_t s) {
    char nm[96] = "";
    snprintf(nm, sizeof nm, "igemm_dma<%d, %d, %d, %d, %s>", TM, TN, BK, ST, g.b_kn ? "true" : "false");
    KTimer kt(nm, 2.0 * g.M * g.N * g.K * grid.y, 4.0 * grid.y * ((double)g.M * g.K / (g.plain_in ? 1 : g.T) + (double)g.N * g.K + (double)g.M * g.N), s);
    if (g.b_kn) hipLaunchKernelGGL((igemm_dma<TM, TN, BK, ST, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((igemm_dma<TM, TN, BK, ST, false>), grid, dim3(256), 0, s, g);
}

int launch_igemm_dma(const IGemm& g, int tile, int variant, int groups, int splits, hipStream_t s) {
    const int bm = tile == 128 || tile == 12864 ? 128 : 64, bn = tile == 128 || tile == 64128 ? 128 : 64;
    const dim3 grid((unsigned)(cdiv(g.M, bm) * cdiv(g.N, bn)), (unsigned)groups, (unsigned)(splits > 1 ? splits : 1));
    if (tile == 64) {
        if (variant == 0 && igemm_dma_tile_ok(g, 32)) { launch_one<1, 1, 32, 2>(g, grid, s); return 1; }
        if (variant == 1 && igemm_dma_tile_ok(g, 32)) { launch_one<1, 1, 32, 3>(g, grid, s); return 1; }
        if (variant == 2 && igemm_dma_tile_ok(g, 16)) { launch_one<1, 1, 16, 4>(g, grid, s); return 1; }
        if (variant == 3 && igemm_dma_tile_ok(g, 16)) { launch_one<1, 1, 16, 3>(g, grid, s); return 1; }
    } else if (tile == 64128) {
        if (variant == 0 && igemm_dma_tile_ok(g, 32)) { launch_one<1, 2, 32, 2>(g, grid, s); return 1; }
        if (variant == 1 && igemm_dma_tile_ok(g, 16)) { launch_one<1, 2, 16, 3>(g, grid, s); return 1; }
    } else if (tile == 12864) {
        if (variant == 0 && igemm_dma_tile_ok(g, 32)) { launch_one<2, 1, 32, 2>(g, grid, s); return 1; }
        if (variant == 1 && igemm_dma_tile_ok(g, 16)) { launch_one<2, 1, 16, 3>(g, grid, s); return 1; }
    } else if (tile == 128) {
        if (variant == 0 && igemm_dma_tile_ok(g, 16)) { launch_one<2, 2, 16, 3>(g, grid, s); return 1; }
        if (variant == 1 && igemm_dma_tile_ok(g, 32)) { launch_one<2, 2, 32, 2>(g, grid, s); return 1; }
    }
    return 0;
}
