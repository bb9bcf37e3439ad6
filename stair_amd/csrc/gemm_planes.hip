// Plane GEMMs: bf16 MFMA contractions whose operands ARRIVE as bf16 planes in HBM and are staged global -> LDS by
// LDS-DMA (global_load_lds_dwordx4), with no register round trip.
//
// Round 1's split kernels (csrc/gemm_bf16x3.hip) load fp32 rows into VGPRs, split them into bf16 hi + lo on the VALU and
// push them into LDS with ds_write_b128; ablations showed that skeleton costing as much as the MFMAs and not overlapping
// with them.  Here the split happens ONCE, outside the contraction:
//   * clip features are stored as bf16 (BASELINE.json configs[1]: "I3D rgb+flow [T=64,2048] feats, bf16"), so the A operand
//     of the LSTM input projection (/root/reference/video_nmn/module_net.py:39-42,160-163) is a single exact plane;
//   * weights are split into hi/lo bf16 planes once per plan run (stair_split_planes), W = hi + lo + O(2^-17 |W|);
// and a product is  A * Whi + A * Wlo  (two v_mfma_f32_32x32x16_bf16, fp32 accumulate) when A is exact bf16, or
// Ahi * Whi + Ahi * Wlo + Alo * Whi when A carries a lo plane as well.
//
// Kernel: 256 x 256 output tile, 8 waves as 2 (M) x 4 (N) of 128 x 64, K in steps of 32, ring of LDS stages
// (3 x 48 KB for one A plane, 2 x 64 KB for two); one raw s_barrier per K step; the LDS-DMA of the stages ahead stays in
// flight across the barrier behind a counted s_waitcnt vmcnt(N) (cdna_hip_programming.md section 5, "Pipelining across
// barriers").  LDS image of a plane: [row 0..255][4 chunks of 8 bf16], chunk c of row R stored at slot c ^ ((R >> 2) & 3):
// a 64-lane LDS-DMA instruction fills 16 rows x 64 B linearly (the swizzle is applied to the per-lane SOURCE address) and
// the ds_read_b128 fragment reads (32 consecutive rows of one chunk) are bank-conflict free.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.h"

namespace stair {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using v4f = __attribute__((ext_vector_type(4))) float;
using v4u = __attribute__((ext_vector_type(4))) unsigned;

namespace {

constexpr int PBK = 32;                    // k per stage
constexpr int PLANE_BYTES = 256 * 64;      // one plane of one stage: 256 rows x 32 bf16

struct PlParams {
    const __bf16 *A[2];      // hi, lo (lo unused when NPA == 1); [M, lda]
    const __bf16 *W[2];      // hi, lo; [N, ldw]
    int64_t lda, ldw;
    const float *bias;
    float *C;
    int64_t ldc;
    int M, N, K, tilesM, tilesN;
    int nt_store;             // 1: the output tile is stored with the non-temporal hint (it is a stream; the A panels and W planes are re-read from L2)
    int stagger;              // 1: waves 4..7 issue their LDS-DMA half a stage after waves 0..3 (MF == 0 form)
};

__device__ __forceinline__ void glds16(const __bf16 *src, __attribute__((address_space(3))) char *dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, dst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N < 0, "add the count");
}

}  // namespace

// swizzle of the LDS plane image: chunk c (8 bf16) of row R lives in slot c ^ swz(R) of the row's four 16-byte slots.
// swz maps (R >> 2) & 3 = 0,1,2,3 to 0,2,3,1: any bijection makes the 32-row fragment reads of v_mfma_f32_32x32x16_bf16
// conflict free; this one also does it for the 16-row x 4-chunk reads of v_mfma_f32_16x16x32_bf16.
__device__ __forceinline__ int pl_swz(int R) {
    const int q0 = (R >> 2) & 1, q1 = (R >> 3) & 1;
    return ((q0 ^ q1) << 1) | q1;
}

// NPA: A planes (1: exact bf16 A, two products per pair; 2: hi + lo, three products).  ACT as stair_gemm_args.act.
// WT: W planes in the tiled layout [K/32][N][32] written by stair_split_planes_tiled: one LDS-DMA instruction then reads
// 1 KB of contiguous memory (16 rows x 64 B) instead of 16 half lines 2*ldw bytes apart.
// MF: 0 = v_mfma_f32_32x32x16_bf16 (wave tile 4 x 2 blocks of 32 x 32, two k steps per stage, fragments of the next k
//     step prefetched across the barrier); 1 = v_mfma_f32_16x16x32_bf16 (8 x 4 blocks of 16 x 16, one k step per stage).
// The grid is at most one workgroup per CU and every workgroup walks its tiles itself, treating (tile, k step) as ONE
// stream of stages: the LDS-DMA of the next tile's first stages is issued while the current tile's last stages are
// multiplied, and the epilogue's stores drain behind the next tile's loads.
// ABL: ablation builds for measurements (1: no LDS-DMA in the loop, 2: no fragment reads / MFMAs); 0 in the product.
template <int NPA, int ACT, bool WT, int MF, int ABL = 0>
__global__ __launch_bounds__(512, 1) void gemm_planes_kernel(PlParams p) {
    extern __shared__ __attribute__((aligned(16))) char plds[];
    constexpr int NPL = NPA + 2;                         // planes per stage: A hi [, A lo], W hi, W lo
    constexpr int NST = NPA == 1 ? 3 : 2;                // ring depth (144 KB / 128 KB)
    constexpr int STAGE_BYTES = NPL * PLANE_BYTES;
    constexpr int LPS = NPL * 2;                         // LDS-DMA instructions per wave and stage
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (uniform: the role branches below are scalar)
    const int wm = wave >> 2, wn = wave & 3;
    const int K = p.K;
    const bool stagger = p.stagger != 0;

    // XCD-aware renumbering: workgroups with equal blockIdx % 8 share an XCD (speed only), and at any moment they hold
    // consecutive tiles, i.e. all tilesN column tiles of the same A panels: the panel leaves HBM once, W stays in L2.
    const int nb = p.tilesM * p.tilesN, G = gridDim.x;
    const int bid = blockIdx.x;
    const int qd = G >> 3, rm = G & 7, xcd = bid & 7, loc = bid >> 3;
    const int first = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;     // bijective on [0, G)
    const int my_tiles = (nb - first + G - 1) / G;      // tiles first, first + G, ... (>= 1: G <= nb)

    // staging: instruction i (0, 1) of this wave fills units [(2 wave + i) * 64, +64) of a plane; unit u = row * 4 + slot
    const int u0 = 2 * wave * 64 + lane;
    const int srow[2] = {u0 >> 2, (u0 + 64) >> 2};
    const int scol = ((u0 & 3) ^ pl_swz(u0 >> 2)) * 8;    // source chunk of this lane's slot (rows 16 apart share swz)
    const int dst_off[2] = {2 * wave * 1024, (2 * wave + 1) * 1024};      // wave-uniform LDS byte offsets inside a plane
    const int64_t wstep = WT ? p.N : 1;                   // W elements per unit of k0: tiled planes advance by N * 32 per stage
    const __bf16 *srcA[NPA][2], *srcW[2][2];
    auto set_src = [&](int tile) {
        const int tm = tile / p.tilesN, tn = tile - tm * p.tilesN;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t ra = min(tm * 256 + srow[i], p.M - 1), rw = min(tn * 256 + srow[i], p.N - 1);
#pragma unroll
            for (int q = 0; q < NPA; ++q) srcA[q][i] = p.A[q] + ra * p.lda + scol;
#pragma unroll
            for (int q = 0; q < 2; ++q) srcW[q][i] = p.W[q] + rw * (WT ? PBK : p.ldw) + scol;
        }
    };
    __attribute__((address_space(3))) char *lbase = (__attribute__((address_space(3))) char *)plds;

#define PL_STAGE(buf, k0)                                                                                   \
    {                                                                                                       \
        __attribute__((address_space(3))) char *sb_ = lbase + (buf);                                        \
        _Pragma("unroll") for (int q_ = 0; q_ < NPA; ++q_)                                                  \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                \
                glds16(srcA[q_][i_] + (k0), sb_ + q_ * PLANE_BYTES + dst_off[i_]);                          \
        _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_)                                                    \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                \
                glds16(srcW[q_][i_] + (k0) * wstep, sb_ + (NPA + q_) * PLANE_BYTES + dst_off[i_]);          \
    }

    // The issue cursor (i_ord, i_k) runs NST-1 stages ahead of the multiply cursor and crosses into the next tile on its
    // own; past the last stage it re-reads the last one into a buffer nobody multiplies, so the counted waits stay exact.
    int i_ord = 0, i_k = 0;
    set_src(first);
    auto advance = [&]() {
        i_k += PBK;
        if (i_k == K) {
            if (i_ord + 1 < my_tiles) { ++i_ord; i_k = 0; set_src(first + i_ord * G); }
            else i_k = K - PBK;
        }
    };
    const int nt = K / PBK;             // K % 32 == 0 (launcher)

    if constexpr (MF == 0) {
        const int r = lane & 31, h = lane >> 5;
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        // fragment byte offsets inside a plane for chunk 0; chunk c flips the slot bits: ^ (c << 4)
        int offA[4], offW[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int R = wm * 128 + 32 * i + r;
            offA[i] = (R * 4 + pl_swz(R)) * 16;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int R = wn * 64 + 32 * j + r;
            offW[j] = (R * 4 + pl_swz(R)) * 16;
        }
        // fragment sets: set 0 = k step 0 of a stage (k = 0..15), set 1 = k step 1 (k = 16..31)
        bf16x8 fa[2][4], fal[2][4], fwh[2][2], fwl[2][2];
#define PL_READ(buf, s_)                                                                                             \
    {                                                                                                                \
        const char *sb_ = plds + (buf);                                                                              \
        const int cx_ = (2 * (s_) + h) << 4;                                                                         \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                           \
            fwh[s_][j_] = *reinterpret_cast<const bf16x8 *>(sb_ + NPA * PLANE_BYTES + (offW[j_] ^ cx_));             \
            fwl[s_][j_] = *reinterpret_cast<const bf16x8 *>(sb_ + (NPA + 1) * PLANE_BYTES + (offW[j_] ^ cx_));       \
        }                                                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                           \
            fa[s_][i_] = *reinterpret_cast<const bf16x8 *>(sb_ + (offA[i_] ^ cx_));                                  \
            if (NPA == 2) fal[s_][i_] = *reinterpret_cast<const bf16x8 *>(sb_ + PLANE_BYTES + (offA[i_] ^ cx_));     \
        }                                                                                                            \
    }
#define PL_MUL(s_)                                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                           \
            if (NPA == 2) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[s_][i_], fwh[s_][j_], acc[i_][j_], 0, 0, 0);   \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][i_], fwl[s_][j_], acc[i_][j_], 0, 0, 0);    \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][i_], fwh[s_][j_], acc[i_][j_], 0, 0, 0);    \
        }
        // one (i, j) block of PL_MUL, and the rest: the first half of a step multiplies ONE block before it issues its reads -- the
        // fragments of k step 0 come out of the previous iteration, and hipcc's wait-count pass, which cannot count across the loop's
        // back edge, puts an lgkmcnt(0) in front of their first use: placed after the reads it waited for all eight of them
#define PL_MUL_ONE(s_)                                                                                               \
    {                                                                                                                \
        if (NPA == 2) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[s_][0], fwh[s_][0], acc[0][0], 0, 0, 0); \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][0], fwl[s_][0], acc[0][0], 0, 0, 0);              \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][0], fwh[s_][0], acc[0][0], 0, 0, 0);              \
    }
#define PL_MUL_REST(s_)                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                           \
            if (i_ == 0 && j_ == 0) continue;                                                                        \
            if (NPA == 2) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[s_][i_], fwh[s_][j_], acc[i_][j_], 0, 0, 0);   \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][i_], fwl[s_][j_], acc[i_][j_], 0, 0, 0);    \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][i_], fwh[s_][j_], acc[i_][j_], 0, 0, 0);    \
        }
        // Ring: at the top of a step, its stage is in LDS and visible to every wave, its k-step-0 fragments are already in
        // registers, and the next NST-2 stages are in flight.  The step
        //   issues the stage NST-1 ahead (into the buffer whose last reader was the previous step: every wave finished
        //   those reads -- lgkmcnt(0) -- before it joined that step's barrier),
        //   reads its k-step-1 fragments and multiplies k step 0 (the MFMAs cover the read latency),
        //   waits for its reads and for its own share of the next stage's LDS-DMA (counted vmcnt: the stage after that
        //   stays in flight across the barrier), joins the barrier,
        //   reads the NEXT stage's k-step-0 fragments and multiplies k step 1.
#pragma unroll
        for (int q = 0; q < NST - 1; ++q) { PL_STAGE(q * STAGE_BYTES, i_k); advance(); }
        wait_vm<(NST - 2) * LPS>();
        __builtin_amdgcn_s_barrier();
        int sbuf = 0, ibuf = (NST - 1) * STAGE_BYTES;
        if (ABL != 2) PL_READ(sbuf, 0);         // (the MFMA-only ablations multiply these fragments over and over)
        // STAGGER: the two waves of a SIMD (w and w + 4) run this same program in lockstep, so they used to issue their LDS-DMA -- six
        // instructions of ~100 cycles of issue each, as long as half a wave's MFMAs of the stage -- at the same moment, with the SIMD's
        // matrix pipe idle meanwhile.  Waves 4..7 now issue theirs HALF a stage later (after the barrier, before k step 1), beside their
        // partners' MFMAs, and wait for all of it (vmcnt(0)) before the next barrier: it has had a whole stage of MFMAs to land.
        const bool late = stagger && wave >= 4;
        for (int ord = 0; ord < my_tiles; ++ord) {
            for (int t = 0; t < nt; ++t) {
                // ABL (measurements): 1 no LDS-DMA; 2 no reads, no MFMAs; 3 MFMAs only (no DMA, reads, barrier); 4 MFMAs + barrier;
                // 5 MFMAs + reads, no barrier (and no DMA); 6 everything but the epilogue's stores
                constexpr bool PIN = ABL != 7;          // (7: the reads unpinned, as before)
                constexpr bool DMA = ABL == 0 || ABL == 2 || ABL == 6 || ABL == 7, RD = ABL == 0 || ABL == 1 || ABL == 5 || ABL == 6 || ABL == 7, MM = ABL != 2,
                               BAR = ABL != 3 && ABL != 5;
                if (DMA && !late) PL_STAGE(ibuf, i_k);
                if (MM && PIN) { PL_MUL_ONE(0); __builtin_amdgcn_sched_barrier(0); }
                if (RD) PL_READ(sbuf, 1);
                // the reads stay ABOVE the MFMAs they hide behind: hipcc otherwise sinks all eight to the END of the block (shorter live
                // ranges), right in front of the lgkmcnt(0) below, and every half stage waits out an LDS round trip with the matrix pipe idle
                if (PIN) __builtin_amdgcn_sched_barrier(0);
                if (MM) { if (PIN) { PL_MUL_REST(0); } else { PL_MUL(0); } }
                __builtin_amdgcn_sched_barrier(0);
                if (BAR) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (late) wait_vm<0>(); else wait_vm<(NST - 2) * LPS>();
                    __builtin_amdgcn_s_barrier();
                }
                __builtin_amdgcn_sched_barrier(0);
                if (DMA && late) PL_STAGE(ibuf, i_k);
                sbuf = sbuf + STAGE_BYTES == NST * STAGE_BYTES ? 0 : sbuf + STAGE_BYTES;
                ibuf = ibuf + STAGE_BYTES == NST * STAGE_BYTES ? 0 : ibuf + STAGE_BYTES;
                if (MM && PIN) { PL_MUL_ONE(1); __builtin_amdgcn_sched_barrier(0); }     // (as above: the compiler's own wait for k step 1's fragments lands here, where it is free)
                if (RD) PL_READ(sbuf, 0);
                if (PIN) __builtin_amdgcn_sched_barrier(0);
                if (MM) { if (PIN) { PL_MUL_REST(1); } else { PL_MUL(1); } }
                advance();                  // (the branches of the cursor sit at the end: the blocks above stay straight-line)
            }
            const int tile = first + ord * G;
            const int tm = tile / p.tilesN, tn = tile - tm * p.tilesN;
            const int m0 = tm * 256, n0 = tn * 256;
            // uniform tile base + 32-bit per-lane offset: the stores take the SGPR-base form and need no 64-bit VGPR math
            float *ctile = p.C + (int64_t)m0 * p.ldc + n0;
            const unsigned loff = (unsigned)((wm * 128 + 4 * h) * (int)p.ldc + wn * 64 + r);
            const bool full = m0 + 256 <= p.M && n0 + 256 <= p.N;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + r;
                const float b = (p.bias && n < p.N) ? p.bias[n] : 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rl = i * 32 + (e & 3) + 8 * (e >> 2);            // + wm * 128 + 4 h inside loff
                        float v = acc[i][j][e] + b;
                        if (ACT == 1) v = fmaxf(v, 0.0f);
                        if (ACT == 2) v = sigmoid_acc(v);
                        __attribute__((address_space(1))) float *rowp =
                            (__attribute__((address_space(1))) float *)(ctile + (int64_t)rl * p.ldc + j * 32);
                        if (ABL != 6 && (full || (n < p.N && m0 + wm * 128 + 4 * h + rl < p.M))) {
                            if (p.nt_store) __builtin_nontemporal_store(v, rowp + loff);
                            else rowp[loff] = v;
                        }
                        acc[i][j][e] = 0.0f;
                    }
                }
            }
        }
#undef PL_READ
#undef PL_MUL
#undef PL_MUL_ONE
#undef PL_MUL_REST
    } else {
        // ---- 16 x 16 x 32: lane l holds A[row l & 15][k = 8 (l >> 4) + j], the whole 32-wide stage is ONE k step ----
        const int r16 = lane & 15, c4 = lane >> 4;
        v4f acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
        int offA[8], offW[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int R = wm * 128 + 16 * i + r16;
            offA[i] = (R * 4 + (c4 ^ pl_swz(R))) * 16;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int R = wn * 64 + 16 * j + r16;
            offW[j] = (R * 4 + (c4 ^ pl_swz(R))) * 16;
        }
#define PL_STEP(buf)                                                                                                 \
    {                                                                                                                \
        const char *sb_ = plds + (buf);                                                                              \
        bf16x8 wh_[4], wl_[4];                                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                           \
            wh_[j_] = *reinterpret_cast<const bf16x8 *>(sb_ + NPA * PLANE_BYTES + offW[j_]);                         \
            wl_[j_] = *reinterpret_cast<const bf16x8 *>(sb_ + (NPA + 1) * PLANE_BYTES + offW[j_]);                   \
        }                                                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                           \
            const bf16x8 ah_ = *reinterpret_cast<const bf16x8 *>(sb_ + offA[i_]);                                    \
            bf16x8 al_ = ah_;                                                                                        \
            if (NPA == 2) al_ = *reinterpret_cast<const bf16x8 *>(sb_ + PLANE_BYTES + offA[i_]);                     \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                       \
                if (NPA == 2) acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al_, wh_[j_], acc[i_][j_], 0, 0, 0);   \
                acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah_, wl_[j_], acc[i_][j_], 0, 0, 0);           \
                acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah_, wh_[j_], acc[i_][j_], 0, 0, 0);           \
            }                                                                                                        \
        }                                                                                                            \
    }
#pragma unroll
        for (int q = 0; q < NST - 1; ++q) { PL_STAGE(q * STAGE_BYTES, i_k); advance(); }
        wait_vm<(NST - 2) * LPS>();
        __builtin_amdgcn_s_barrier();
        int sbuf = 0, ibuf = (NST - 1) * STAGE_BYTES;
        for (int ord = 0; ord < my_tiles; ++ord) {
            for (int t = 0; t < nt; ++t) {
                if (ABL != 1) PL_STAGE(ibuf, i_k);
                if (ABL != 2) PL_STEP(sbuf);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every read of this stage done before its buffer is refilled
                wait_vm<(NST - 2) * LPS>();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                sbuf = sbuf + STAGE_BYTES == NST * STAGE_BYTES ? 0 : sbuf + STAGE_BYTES;
                ibuf = ibuf + STAGE_BYTES == NST * STAGE_BYTES ? 0 : ibuf + STAGE_BYTES;
                advance();
            }
            const int tile = first + ord * G;
            const int tm = tile / p.tilesN, tn = tile - tm * p.tilesN;
            const int m0 = tm * 256, n0 = tn * 256;
            float *ctile = p.C + (int64_t)m0 * p.ldc + n0;
            const unsigned loff = (unsigned)((wm * 128 + 4 * c4) * (int)p.ldc + wn * 64 + r16);
            const bool full = m0 + 256 <= p.M && n0 + 256 <= p.N;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + r16;
                const float b = (p.bias && n < p.N) ? p.bias[n] : 0.0f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int rl = i * 16 + e;                                  // + wm * 128 + 4 c4 inside loff
                        float v = acc[i][j][e] + b;
                        if (ACT == 1) v = fmaxf(v, 0.0f);
                        if (ACT == 2) v = sigmoid_acc(v);
                        __attribute__((address_space(1))) float *rowp =
                            (__attribute__((address_space(1))) float *)(ctile + (int64_t)rl * p.ldc + j * 16);
                        if (full || (n < p.N && m0 + wm * 128 + 4 * c4 + rl < p.M)) rowp[loff] = v;
                        acc[i][j][e] = 0.0f;
                    }
                }
            }
        }
#undef PL_STEP
    }
    wait_vm<0>();                       // the trailing (unused) stages must land before the LDS is given back
#undef PL_STAGE
}

// ---- W from registers (round 4 experiment -> product where it wins; w_tiled == 2) ------------------------------------------------
// The ablations of the kernel above say it is co-limited by what it STAGES: 48 KB per 256 x 256 x 32 stage through LDS-DMA (A 16 KB,
// W hi + lo 32 KB), 12.9 GB per launch, and by the fragment reads that bring all of it back out of LDS.  Two thirds of those bytes are W,
// and a wave needs only ITS 64 columns of W: here W never enters LDS.  It is stored in MFMA fragment order (stair_pack_wfrag:
// [N/32][K/16][hi, lo][64 lanes][8 bf16], 1 KB per fragment) and each wave loads its own fragments global -> VGPR one stage ahead (the two
// waves that share a column block load the same lines: L1 hits); LDS-DMA stages A alone (16 KB per stage, ring of 4).  NPA == 1, act 0.
template <int ABL = 0>
__global__ __launch_bounds__(512, 1) void gemm_planes_wr_kernel(PlParams p) {
    extern __shared__ __attribute__((aligned(16))) char plds[];
    constexpr int NSTA = 4;                              // A stages in the ring (64 KB)
    constexpr int STAGE_BYTES = PLANE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int K = p.K, KS = K / 16;
    const int nb = p.tilesM * p.tilesN, G = gridDim.x;
    const int bid = blockIdx.x;
    const int qd = G >> 3, rm = G & 7, xcd = bid & 7, loc = bid >> 3;
    const int first = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    const int my_tiles = (nb - first + G - 1) / G;

    const int u0 = 2 * wave * 64 + lane;
    const int srow[2] = {u0 >> 2, (u0 + 64) >> 2};
    const int scol = ((u0 & 3) ^ pl_swz(u0 >> 2)) * 8;
    const int dst_off[2] = {2 * wave * 1024, (2 * wave + 1) * 1024};
    const __bf16 *srcA[2];
    auto set_src = [&](int tile) {
        const int tm = tile / p.tilesN;
#pragma unroll
        for (int i = 0; i < 2; ++i) srcA[i] = p.A[0] + (int64_t)min(tm * 256 + srow[i], p.M - 1) * p.lda + scol;
    };
    __attribute__((address_space(3))) char *lbase = (__attribute__((address_space(3))) char *)plds;
#define WR_STAGE(buf, k0)                                                                                   \
    {                                                                                                       \
        __attribute__((address_space(3))) char *sb_ = lbase + (buf);                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) glds16(srcA[i_] + (k0), sb_ + dst_off[i_]);        \
    }
    // A cursor (NSTA - 1 stages ahead of the multiply) and W cursor (one stage ahead): both cross into the next tile on their own
    int i_ord = 0, i_k = 0;
    set_src(first);
    auto advance = [&]() {
        i_k += PBK;
        if (i_k == K) {
            if (i_ord + 1 < my_tiles) { ++i_ord; i_k = 0; set_src(first + i_ord * G); }
            else i_k = K - PBK;
        }
    };
    int w_ord = 0, w_s = 0;                              // the stage the W cursor points at
    const __bf16 *wq0;                                   // fragment image of this wave's first column block of tile w_ord (uniform)
    auto set_w = [&](int tile) {
        const int tn = tile - (tile / p.tilesN) * p.tilesN;
        const int nt0 = min((tn * 256 + wn * 64) >> 5, (p.N >> 5) - 2);
        wq0 = p.W[0] + (int64_t)nt0 * KS * 2 * 512;
    };
    set_w(first);
    auto advance_w = [&]() {
        ++w_s;
        if (w_s == K / PBK) {
            if (w_ord + 1 < my_tiles) { ++w_ord; w_s = 0; set_w(first + w_ord * G); }
            else w_s = K / PBK - 1;
        }
    };
    const int nt = K / PBK;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    int offA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = wm * 128 + 32 * i + r;
        offA[i] = (R * 4 + pl_swz(R)) * 16;
    }
    bf16x8 fa[2][4];
    bf16x8 wf[2][2][2][2];                               // [ring slot][k step][column block][hi, lo]
    // The loads are inline asm: a load hipcc can see gets a wait-count it chooses -- across this loop's back edge a vmcnt(0) right
    // behind the issue, i.e. no prefetch at all.  Base in SGPRs (uniform: block nt0 + j, stage w_s), lane offset in one VGPR, the
    // four fragments of a column block (k step 0 / 1 x hi / lo) 1 KB apart through the immediate offset.
    const unsigned lane16 = (unsigned)lane * 16u;
#define WR_LOAD1(dst_, base_, off_) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #off_ : "=v"(dst_) : "v"(lane16), "s"(base_) : "memory")
#define WR_LOADW(slot_)                                                                                      \
    {                                                                                                        \
        const char *b0_ = reinterpret_cast<const char *>(wq0) + (int64_t)w_s * 4096;                         \
        const char *b1_ = b0_ + (int64_t)KS * 2048;                                                          \
        WR_LOAD1(wf[slot_][0][0][0], b0_, 0); WR_LOAD1(wf[slot_][0][0][1], b0_, 1024);                       \
        WR_LOAD1(wf[slot_][1][0][0], b0_, 2048); WR_LOAD1(wf[slot_][1][0][1], b0_, 3072);                    \
        WR_LOAD1(wf[slot_][0][1][0], b1_, 0); WR_LOAD1(wf[slot_][0][1][1], b1_, 1024);                       \
        WR_LOAD1(wf[slot_][1][1][0], b1_, 2048); WR_LOAD1(wf[slot_][1][1][1], b1_, 3072);                    \
    }
    // every fragment of ring slot `slot_` has landed (all but the N newest vector-memory operations are done); the registers are
    // operands of the statement, so nothing that reads them can be scheduled above it
#define WR_WAITW(slot_, N_)                                                                                  \
    asm volatile("s_waitcnt vmcnt(" #N_ ")"                                                                  \
                 : "+v"(wf[slot_][0][0][0]), "+v"(wf[slot_][0][0][1]), "+v"(wf[slot_][1][0][0]), "+v"(wf[slot_][1][0][1]),   \
                   "+v"(wf[slot_][0][1][0]), "+v"(wf[slot_][0][1][1]), "+v"(wf[slot_][1][1][0]), "+v"(wf[slot_][1][1][1]) :: "memory")
#define WR_READ(buf, s_)                                                                                     \
    {                                                                                                        \
        const char *sb_ = plds + (buf);                                                                      \
        const int cx_ = (2 * (s_) + h) << 4;                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) fa[s_][i_] = *reinterpret_cast<const bf16x8 *>(sb_ + (offA[i_] ^ cx_));   \
    }
#define WR_MUL_ONE(s_, slot_)                                                                                \
    {                                                                                                        \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][0], wf[slot_][s_][0][1], acc[0][0], 0, 0, 0);   \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][0], wf[slot_][s_][0][0], acc[0][0], 0, 0, 0);   \
    }
#define WR_MUL_REST(s_, slot_)                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                         \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                   \
            if (i_ == 0 && j_ == 0) continue;                                                                \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][i_], wf[slot_][s_][j_][1], acc[i_][j_], 0, 0, 0);   \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][i_], wf[slot_][s_][j_][0], acc[i_][j_], 0, 0, 0);   \
        }
    // prologue: A stages 0 .. NSTA-2 and the W fragments of stage 0
#pragma unroll
    for (int q = 0; q < NSTA - 1; ++q) { WR_STAGE(q * STAGE_BYTES, i_k); advance(); }
    WR_LOADW(0)
    advance_w();
    WR_WAITW(0, 0);                                      // A stages 0 .. 2 and the fragments of stage 0 have landed
    __builtin_amdgcn_s_barrier();
    int sbuf = 0, ibuf = (NSTA - 1) * STAGE_BYTES;
    WR_READ(sbuf, 0);
    int done = 0;                                        // stages multiplied since the start (the first two wait differently)
#define WR_STEP(slot_)                                                                                       \
    {                                                                                                        \
        /* this stage's fragments were requested one step ago; behind them only that step's two LDS-DMA: vmcnt(2) (first step: 0) */ \
        if (done > 0) { WR_WAITW(slot_, 2); } else { WR_WAITW(slot_, 0); }                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        WR_LOADW(1 - (slot_))                            /* W of the NEXT stage: 8 ops */                     \
        if (ABL != 1) WR_STAGE(ibuf, i_k);               /* then A of the stage NSTA-1 ahead: 2 ops */       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        WR_MUL_ONE(0, slot_); __builtin_amdgcn_sched_barrier(0);                                             \
        WR_READ(sbuf, 1); __builtin_amdgcn_sched_barrier(0);                                                 \
        WR_MUL_REST(0, slot_);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        /* (the NEXT stage's A was requested two steps ago: the wait at the top of THIS step has already seen it land) */ \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        sbuf = sbuf + STAGE_BYTES == NSTA * STAGE_BYTES ? 0 : sbuf + STAGE_BYTES;                            \
        ibuf = ibuf + STAGE_BYTES == NSTA * STAGE_BYTES ? 0 : ibuf + STAGE_BYTES;                            \
        WR_MUL_ONE(1, slot_); __builtin_amdgcn_sched_barrier(0);                                             \
        WR_READ(sbuf, 0); __builtin_amdgcn_sched_barrier(0);                                                 \
        WR_MUL_REST(1, slot_);                                                                               \
        advance(); advance_w(); ++done;                                                                      \
    }
    for (int ord = 0; ord < my_tiles; ++ord) {
        for (int t = 0; t < nt; t += 2) {                // nt is even (K % 64 == 0, launcher): the ring slot of a step is static
            WR_STEP(0)
            WR_STEP(1)
        }
        const int tile = first + ord * G;
        const int tm = tile / p.tilesN, tn = tile - tm * p.tilesN;
        const int m0 = tm * 256, n0 = tn * 256;
        float *ctile = p.C + (int64_t)m0 * p.ldc + n0;
        const unsigned loff = (unsigned)((wm * 128 + 4 * h) * (int)p.ldc + wn * 64 + r);
        const bool full = m0 + 256 <= p.M && n0 + 256 <= p.N;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + r;
            const float b = (p.bias && n < p.N) ? p.bias[n] : 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rl = i * 32 + (e & 3) + 8 * (e >> 2);
                    const float v = acc[i][j][e] + b;
                    __attribute__((address_space(1))) float *rowp =
                        (__attribute__((address_space(1))) float *)(ctile + (int64_t)rl * p.ldc + j * 32);
                    if (full || (n < p.N && m0 + wm * 128 + 4 * h + rl < p.M)) {
                        if (p.nt_store) __builtin_nontemporal_store(v, rowp + loff);
                        else rowp[loff] = v;
                    }
                    acc[i][j][e] = 0.0f;
                }
            }
        }
    }
    wait_vm<0>();
#undef WR_STEP
#undef WR_MUL_REST
#undef WR_MUL_ONE
#undef WR_READ
#undef WR_LOADW
#undef WR_LOAD1
#undef WR_WAITW
#undef WR_STAGE
}

// x -> hi = bf16(x), lo = bf16(x - hi); 8 elements per thread (two 16-byte loads, two 16-byte stores)
__global__ void split_planes_kernel(const float *x, __bf16 *hi, __bf16 *lo, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const v4f a = *reinterpret_cast<const v4f *>(x + 8 * i), b = *reinterpret_cast<const v4f *>(x + 8 * i + 4);
        bf16x8 vh, vl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vh[j] = (__bf16)a[j]; vl[j] = (__bf16)(a[j] - (float)vh[j]);
            vh[4 + j] = (__bf16)b[j]; vl[4 + j] = (__bf16)(b[j] - (float)vh[4 + j]);
        }
        *reinterpret_cast<bf16x8 *>(hi + 8 * i) = vh;
        if (lo) *reinterpret_cast<bf16x8 *>(lo + 8 * i) = vl;
    }
}

// tiled form for W operands: src [rows, cols] row-major fp32 -> planes [cols/32][rows][32]; one thread per 8 elements
__global__ void split_planes_tiled_kernel(const float *x, __bf16 *hi, __bf16 *lo, int rows, int cols, int row_off, int total_rows) {
    const int64_t n8 = (int64_t)rows * cols / 8;
    const int c8 = cols / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / c8), k8 = (int)(i - (int64_t)row * c8);          // source: row, columns 8 k8 .. 8 k8 + 7
        const v4f a = *reinterpret_cast<const v4f *>(x + 8 * i), b = *reinterpret_cast<const v4f *>(x + 8 * i + 4);
        bf16x8 vh, vl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vh[j] = (__bf16)a[j]; vl[j] = (__bf16)(a[j] - (float)vh[j]);
            vh[4 + j] = (__bf16)b[j]; vl[4 + j] = (__bf16)(b[j] - (float)vh[4 + j]);
        }
        const int64_t o = ((int64_t)(k8 >> 2) * total_rows + row_off + row) * PBK + (k8 & 3) * 8;
        *reinterpret_cast<bf16x8 *>(hi + o) = vh;
        *reinterpret_cast<bf16x8 *>(lo + o) = vl;
    }
}

// fp32 [rows, cols] (row stride ldx, cols % 4 == 0) -> hi / lo planes with the columns zero-padded to Kp (a multiple of 32): row-major
// [rows, Kp] (an A operand of the plane GEMM) or tiled [Kp/32][total_rows][32] (a W operand).  One thread per 8 output elements.
template <bool TILED>
__global__ void split_planes_pad_kernel(const float *x, int64_t ldx, __bf16 *hi, __bf16 *lo, int rows, int cols, int Kp, int row_off, int total_rows) {
    const int c8 = Kp / 8;
    const int64_t n8 = (int64_t)rows * c8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / c8), k8 = (int)(i - (int64_t)row * c8);
        const float *src = x + (int64_t)row * ldx + 8 * k8;
        const v4f z = {0.f, 0.f, 0.f, 0.f};
        const v4f a = 8 * k8 < cols ? *reinterpret_cast<const v4f *>(src) : z, b = 8 * k8 + 4 < cols ? *reinterpret_cast<const v4f *>(src + 4) : z;
        bf16x8 vh, vl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vh[j] = (__bf16)a[j]; vl[j] = (__bf16)(a[j] - (float)vh[j]);
            vh[4 + j] = (__bf16)b[j]; vl[4 + j] = (__bf16)(b[j] - (float)vh[4 + j]);
        }
        const int64_t o = TILED ? ((int64_t)(k8 >> 2) * total_rows + row_off + row) * PBK + (k8 & 3) * 8 : (int64_t)row * Kp + 8 * k8;
        *reinterpret_cast<bf16x8 *>(hi + o) = vh;
        *reinterpret_cast<bf16x8 *>(lo + o) = vl;
    }
}

int launch_split_planes_pad(const float *x, int64_t ldx, void *hi, void *lo, int rows, int cols, int Kp, bool tiled, hipStream_t s, int row_off, int total_rows) {
    if (total_rows <= 0) total_rows = rows;
    STAIR_CHECK(x && hi && lo && rows > 0, "null argument");
    STAIR_CHECK(cols > 0 && cols % 4 == 0 && ldx % 4 == 0 && Kp % PBK == 0 && Kp >= cols, "cols, ldx multiples of 4; Kp a multiple of 32, >= cols");
    STAIR_CHECK(row_off >= 0 && row_off + rows <= total_rows, "row block outside the plane");
    STAIR_CHECK(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 15) == 0, "pointers must be 16-byte aligned");
    const int64_t n8 = (int64_t)rows * (Kp / 8);
    const int blocks = (int)std::min<int64_t>((n8 + 255) / 256, 8192);
    if (tiled) hipLaunchKernelGGL(split_planes_pad_kernel<true>, dim3(blocks), dim3(256), 0, s, x, ldx, static_cast<__bf16 *>(hi), static_cast<__bf16 *>(lo), rows, cols, Kp, row_off, total_rows);
    else hipLaunchKernelGGL(split_planes_pad_kernel<false>, dim3(blocks), dim3(256), 0, s, x, ldx, static_cast<__bf16 *>(hi), static_cast<__bf16 *>(lo), rows, cols, Kp, row_off, total_rows);
    STAIR_LAUNCH_CHECK();
    return 0;
}

int launch_split_planes_tiled(const float *x, void *hi, void *lo, int rows, int cols, hipStream_t s, int row_off, int total_rows) {
    if (total_rows <= 0) total_rows = rows;
    STAIR_CHECK(row_off >= 0 && row_off + rows <= total_rows, "row block outside the plane");
    STAIR_CHECK(x && hi && lo, "null argument");
    STAIR_CHECK(rows > 0 && cols > 0 && cols % PBK == 0, "cols must be a positive multiple of 32");
    STAIR_CHECK(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 15) == 0,
                "pointers must be 16-byte aligned");
    const int64_t n8 = (int64_t)rows * cols / 8;
    const int blocks = (int)std::min<int64_t>((n8 + 255) / 256, 4096);
    hipLaunchKernelGGL(split_planes_tiled_kernel, dim3(blocks), dim3(256), 0, s, x, static_cast<__bf16 *>(hi), static_cast<__bf16 *>(lo), rows, cols, row_off, total_rows);
    STAIR_LAUNCH_CHECK();
    return 0;
}

int launch_split_planes(const float *x, void *hi, void *lo, int64_t n, hipStream_t s) {
    STAIR_CHECK(x && hi, "null argument");
    STAIR_CHECK(n % 8 == 0, "element count must be a multiple of 8");
    STAIR_CHECK(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 15) == 0,
                "pointers must be 16-byte aligned");
    if (n == 0) return 0;
    const int64_t n8 = n / 8;
    const int blocks = (int)std::min<int64_t>((n8 + 255) / 256, 4096);
    hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, s, x, static_cast<__bf16 *>(hi), static_cast<__bf16 *>(lo), n8);
    STAIR_LAUNCH_CHECK();
    return 0;
}

bool gemm_planes_supported(int64_t M, int N, int K) {
    return M >= 256 && N >= 256 && K >= 64 && K % PBK == 0;
}

int launch_gemm_planes(const stair_gemm_planes_args &a, hipStream_t s) {
    STAIR_CHECK(a.A_hi && a.W_hi && a.W_lo && a.C, "null operand");
    STAIR_CHECK(a.M > 0 && a.N > 0 && a.K > 0 && a.K % PBK == 0, "K must be a positive multiple of 32");
    STAIR_CHECK(a.lda % 8 == 0 && (a.w_tiled || a.ldw % 8 == 0), "lda / ldw must be multiples of 8 bf16 (16-byte rows)");
    STAIR_CHECK(((reinterpret_cast<uintptr_t>(a.A_hi) | reinterpret_cast<uintptr_t>(a.A_lo) | reinterpret_cast<uintptr_t>(a.W_hi) |
                  reinterpret_cast<uintptr_t>(a.W_lo)) & 15) == 0, "plane pointers must be 16-byte aligned");
    if (gemm_trace_on())
        fprintf(stderr, "STAIR_GEMM planes M=%d N=%d K=%d act=%d acc=0 gather=0 scale=0\n", a.M, a.N, a.K, a.act);
    PlParams p;
    p.A[0] = static_cast<const __bf16 *>(a.A_hi); p.A[1] = static_cast<const __bf16 *>(a.A_lo);
    p.W[0] = static_cast<const __bf16 *>(a.W_hi); p.W[1] = static_cast<const __bf16 *>(a.W_lo);
    p.lda = a.lda; p.ldw = a.ldw; p.bias = a.bias; p.C = a.C; p.ldc = a.ldc;
    {   // measured: 1.985 -> 1.969 ms per launch at the bench shape (two A/B pairs on one box); STAIR_PLANES_NT_STORE=0 switches it off
        static const int nts = [] { const char *e = getenv("STAIR_PLANES_NT_STORE"); return (e && e[0] == '0') ? 0 : 1; }();
        p.nt_store = nts;
        static const int stg = [] { const char *e = getenv("STAIR_PLANES_STAGGER"); return (e && e[0] == '0') ? 0 : 1; }();
        p.stagger = stg;
    }
    p.M = a.M; p.N = a.N; p.K = a.K;
    p.tilesM = (a.M + 255) / 256; p.tilesN = (a.N + 255) / 256;
    const int nb = p.tilesM * p.tilesN;
    STAIR_ACCT_MFMA("gemm_planes", (int64_t)a.M * a.K * (a.A_lo ? 4 : 2) + (int64_t)a.N * a.K * 4 + (int64_t)a.M * a.N * 4, 2ll * a.M * a.N * a.K);
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n > 0 ? n : 256;
    }();
    static const int ablate = [] { const char *e = getenv("STAIR_PLANES_ABLATE"); return e ? atoi(e) : 0; }();   // measurements only
    const dim3 grid(std::min(nb, ncu)), block(512);
    if (a.w_tiled == 2) {               // W in fragment order, loaded global -> VGPR by the wave that multiplies it (gemm_planes_wr_kernel)
        STAIR_CHECK(!a.A_lo && a.act == 0, "fragment-order W: exact-bf16 A, no activation");
        STAIR_CHECK(a.N % 32 == 0 && a.N >= 64 && a.K % 64 == 0, "fragment-order W: N % 32 == 0, N >= 64, K % 64 == 0");
        const size_t shw = 4 * PLANE_BYTES;
        if (ablate == 1) hipLaunchKernelGGL((gemm_planes_wr_kernel<1>), grid, block, shw, s, p);
        else hipLaunchKernelGGL((gemm_planes_wr_kernel<0>), grid, block, shw, s, p);
        STAIR_LAUNCH_CHECK();
        return 0;
    }
    const size_t sh1 = 3 * 3 * PLANE_BYTES, sh2 = 2 * 4 * PLANE_BYTES;
    static bool attr_set = false;       // one device per process (one ctx per GPU / process, see stair_hip.h)
    // MFMA shape: measured equal for two products (0.95 ms either way on the dominant shape); with three the 16x16x32 form
    // is 4 % faster (1.32 vs 1.37 ms) -- profiles/r02_a_planes_bench.txt.  STAIR_PLANES_MFMA = 0 / 1 forces one.
    static const int mf_force = [] { const char *e = getenv("STAIR_PLANES_MFMA"); return e ? atoi(e) : -1; }();
    const int mf_env = mf_force >= 0 ? mf_force : (a.A_lo ? 1 : 0);
#define P_FOREACH(X) X(0, false, 0) X(1, false, 0) X(2, false, 0) X(0, true, 0) X(1, true, 0) X(2, true, 0) \
                     X(0, false, 1) X(1, false, 1) X(2, false, 1) X(0, true, 1) X(1, true, 1) X(2, true, 1)
    if (!attr_set) {
#define P_ATTR(ACT_, WT_, MF_)                                                                                          \
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_planes_kernel<1, ACT_, WT_, MF_>),            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh1));                            \
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_planes_kernel<2, ACT_, WT_, MF_>),            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh2));
        P_FOREACH(P_ATTR)
#undef P_ATTR
#define P_ATTR_A(MF_, ABL_) STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_planes_kernel<1, 0, true, MF_, ABL_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh1));
        P_ATTR_A(0, 1) P_ATTR_A(0, 2) P_ATTR_A(1, 1) P_ATTR_A(1, 2) P_ATTR_A(0, 3) P_ATTR_A(0, 4) P_ATTR_A(0, 5) P_ATTR_A(0, 6) P_ATTR_A(0, 7)
#undef P_ATTR_A
        attr_set = true;
    }
#define P_LAUNCH2(ACT_, WT_, MF_)                                                                                       \
    {                                                                                                                   \
        if (a.A_lo) hipLaunchKernelGGL((gemm_planes_kernel<2, ACT_, WT_, MF_>), grid, block, sh2, s, p);                 \
        else hipLaunchKernelGGL((gemm_planes_kernel<1, ACT_, WT_, MF_>), grid, block, sh1, s, p);                        \
    }
#define P_LAUNCH(ACT_, WT_) { if (mf_env) P_LAUNCH2(ACT_, WT_, 1) else P_LAUNCH2(ACT_, WT_, 0) }
    if (ablate && a.act == 0 && a.w_tiled && !a.A_lo) {
        if (mf_env == 0 && ablate == 3) hipLaunchKernelGGL((gemm_planes_kernel<1, 0, true, 0, 3>), grid, block, sh1, s, p);
        else if (mf_env == 0 && ablate == 4) hipLaunchKernelGGL((gemm_planes_kernel<1, 0, true, 0, 4>), grid, block, sh1, s, p);
        else if (mf_env == 0 && ablate == 5) hipLaunchKernelGGL((gemm_planes_kernel<1, 0, true, 0, 5>), grid, block, sh1, s, p);
        else if (mf_env == 0 && ablate == 6) hipLaunchKernelGGL((gemm_planes_kernel<1, 0, true, 0, 6>), grid, block, sh1, s, p);
        else if (mf_env == 0 && ablate == 7) hipLaunchKernelGGL((gemm_planes_kernel<1, 0, true, 0, 7>), grid, block, sh1, s, p);
        else if (mf_env == 0 && ablate == 1) hipLaunchKernelGGL((gemm_planes_kernel<1, 0, true, 0, 1>), grid, block, sh1, s, p);
        else if (mf_env == 0) hipLaunchKernelGGL((gemm_planes_kernel<1, 0, true, 0, 2>), grid, block, sh1, s, p);
        else if (ablate == 1) hipLaunchKernelGGL((gemm_planes_kernel<1, 0, true, 1, 1>), grid, block, sh1, s, p);
        else hipLaunchKernelGGL((gemm_planes_kernel<1, 0, true, 1, 2>), grid, block, sh1, s, p);
    } else if (a.w_tiled) {
        switch (a.act) {
            case 0: P_LAUNCH(0, true) break;
            case 1: P_LAUNCH(1, true) break;
            default: P_LAUNCH(2, true) break;
        }
    } else {
        switch (a.act) {
            case 0: P_LAUNCH(0, false) break;
            case 1: P_LAUNCH(1, false) break;
            default: P_LAUNCH(2, false) break;
        }
    }
#undef P_LAUNCH
#undef P_LAUNCH2
#undef P_FOREACH
    STAIR_LAUNCH_CHECK();
    return 0;
}

// Measurement aid (bench.py's roofline): what the matrix pipes of THIS device sustain on random operands.  Every wave of a full grid
// (one 512-thread workgroup per CU, two waves per SIMD -- the residency of the plane GEMMs) issues v_mfma_f32_32x32x16_bf16 back to
// back from registers: no memory, no LDS, no barrier in the loop.  The chip lowers its clock under such a load
// (MI355X_MICROARCH.md "DVFS give-back"), so the rate is well under the 2.5 PFLOP/s the 2.4 GHz figure gives: this is the ceiling any
// bf16 MFMA kernel can reach here, and the number to read a kernel's EXECUTED flop rate against.
__global__ __launch_bounds__(512, 1) void mfma_probe_kernel(const __bf16 *seed, float *sink, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(seed + ((wave * 6 + i) * 64 + lane) * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8 *>(seed + ((wave * 6 + 4 + j) * 64 + lane) * 8);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) t += acc[i][j][e];
    if (t == 123.456f) sink[blockIdx.x * 512 + threadIdx.x] = t;      // never true for this data: keeps the loop alive
}

}  // namespace stair

extern "C" int stair_mfma_probe(int32_t iters, int32_t repeats, double *tflops, stair_stream stream) {
    using namespace stair;
    STAIR_CHECK(iters > 0 && repeats > 0 && tflops, "iters, repeats > 0 and a result pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const size_t seed_elems = (size_t)8 * 6 * 64 * 8;
    __bf16 *seed = nullptr;
    float *sink = nullptr;
    STAIR_HIP(hipMalloc(&seed, seed_elems * sizeof(__bf16)));
    STAIR_HIP(hipMalloc(&sink, (size_t)cus * 512 * sizeof(float)));
    std::vector<unsigned short> host(seed_elems);
    unsigned x = 12345u;
    for (size_t i = 0; i < seed_elems; ++i) {              // random bf16 values in (-2, 2): sign, exponent 125..127, 7 random mantissa bits
        x = x * 1664525u + 1013904223u;
        host[i] = (unsigned short)(((x >> 31) << 15) | ((125u + ((x >> 8) % 3u)) << 7) | ((x >> 16) & 0x7f));
    }
    STAIR_HIP(hipMemcpy(seed, host.data(), seed_elems * sizeof(__bf16), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    STAIR_HIP(hipEventCreate(&e0)); STAIR_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(cus), dim3(512), 0, s, seed, sink, iters);       // warm-up (clocks settle under load)
    STAIR_HIP(hipEventRecord(e0, s));
    for (int r = 0; r < repeats; ++r) hipLaunchKernelGGL(mfma_probe_kernel, dim3(cus), dim3(512), 0, s, seed, sink, iters);
    STAIR_HIP(hipEventRecord(e1, s));
    STAIR_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    STAIR_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(seed); (void)hipFree(sink);
    const double flops = (double)repeats * cus * 8.0 * iters * 8.0 * 2.0 * 32 * 32 * 16;
    *tflops = flops / ((double)ms * 1e-3) / 1e12;
    return 0;
}

namespace stair {
}  // namespace stair

extern "C" int stair_split_planes(const float *x, void *hi, void *lo, int64_t n, stair_stream stream) {
    return stair::launch_split_planes(x, hi, lo, n, static_cast<hipStream_t>(stream));
}

extern "C" int stair_split_planes_tiled(const float *x, void *hi, void *lo, int32_t rows, int32_t cols, stair_stream stream) {
    return stair::launch_split_planes_tiled(x, hi, lo, rows, cols, static_cast<hipStream_t>(stream), 0, rows);
}

extern "C" int stair_gemm_planes(const stair_gemm_planes_args *args, stair_stream stream) {
    if (!args) {
        stair::set_error("stair_gemm_planes: null args");
        return 1;
    }
    return stair::launch_gemm_planes(*args, static_cast<hipStream_t>(stream));
}
