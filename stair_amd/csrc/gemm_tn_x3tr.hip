// Weight-gradient GEMM of the module-level Linear layers, one long reduction per WEIGHT:
//     dW[n][k] += sum_m dZ[m][n] * (rs[m] * X[m][k]),   db[n] += sum_m dZ[m][n]
// (autograd of every nn.Linear of /root/reference/video_nmn/modules.py applied to [T, H] maps, train_module.py:408; M = all
// frames of all instances that used the weight in this step: 16 k ... 110 k rows at 2048 questions, N = K = H = 512).
// Both operands are fp32 rows whose ROW index is the reduction index.  csrc/gemm_bf16x3.hip's TN kernel transposes 8 x 4
// register blocks on the way into LDS and adds its M-slabs with fp32 atomics; this one follows csrc/gemm_tn_tr.hip instead:
//   * the tiles keep their natural row-major layout in LDS -- each element is split once into bf16 hi + lo on its way through
//     registers -- and the MFMA fragments are read TRANSPOSED by the hardware (ds_read_b64_tr_b16);
//   * a product is hi*hi + lo*hi + hi*lo: three v_mfma_f32_32x32x16_bf16 per pair, fp32 accumulate (the split mode's arithmetic);
//   * 256 (n) x 128 (k) output tile, 8 waves as 4 (n) x 2 (k) of 64 x 64; 32 reduction rows per stage, ring of two LDS stages
//     (48 KB each), ONE barrier per stage, the loads of two stages ahead in flight in registers;
//   * M is cut into <= 32 slabs; a slab's partial tile is STORED to scratch and a second kernel adds the slabs in fixed order:
//     no atomics, so the weight gradients are bit-reproducible run to run, and the epilogue is plain 128-byte stores.
// X rows are gathered per group of rows_per_group (= T) rows through b_gidx (the instances' input tiles) and may carry a
// per-row scale (Temporal's r_t); the bias gradient rides along in the staging of dZ.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace stair {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using v4f = __attribute__((ext_vector_type(4))) float;

constexpr int XT_ROWS = 32;                   // reduction rows per stage
constexpr int XT_HALF = XT_ROWS * 256;        // [32 rows][128 columns] of bf16: 8 KB
constexpr int XT_APLANE = 2 * XT_HALF;        // dZ: 256 columns
constexpr int XT_BPLANE = XT_HALF;            // X: 128 columns
constexpr int XT_STAGE = 2 * XT_APLANE + 2 * XT_BPLANE;     // 48 KB
constexpr int XT_MAXGRP = 2048;               // row groups per slab whose gather indices are staged in LDS
constexpr int XT_LDS = 2 * XT_STAGE + 2 * XT_MAXGRP * 4;     // 96 KB of stages + 16 KB of indices
constexpr int XT_MAXSLAB = 32;

struct XtParams {
    const float *A; int64_t lda;
    const float *B; int64_t ldb, b_gstride; const int32_t *b_gidx;
    const float *rs; int64_t rs_gstride; const int32_t *rs_gidx;
    float *P;                                 // [nslab][N][K] partial products
    float *Pc;                                // [nslab][N] partial column sums of dZ, or null
    int N, K, rpg, stages, nslab, tilesK;
};

// the 16-byte chunk ch of row r of half h sits at chunk ch ^ f(r) ^ h (cdna_hip_programming.md T10, image (b); same image as gemm_tn_tr.hip)
__device__ __forceinline__ int xt_chunk(int row, int ch, int h) { return ch ^ (((row & 3) << 2) | ((row >> 2) & 3)) ^ h; }

struct XtRegs { v4f a[4]; v4f b[2]; float rs; };

__device__ __forceinline__ bf16x4 tr_read(const char *base, int off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(base + off));
}
__device__ __forceinline__ bf16x8 join8(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

}  // namespace

template <bool GIDX, bool RS, bool KTAIL>
__global__ __launch_bounds__(512, 1) void gemm_tn_x3tr_kernel(XtParams p) {
    extern __shared__ __attribute__((aligned(16))) char xl[];       // [2 stages][dZ hi | dZ lo | X hi | X lo]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the tiles of one slab share its rows: blocks slab, slab + nslab, ... sit on one XCD when nslab % 8 == 0 (speed only)
    const int slab = blockIdx.x % p.nslab, tile = blockIdx.x / p.nslab;
    const int tn = tile / p.tilesK, tk = tile - tn * p.tilesK;
    const int n0 = tn * 256, k0 = tk * 128;
    const int st_beg = (int)((int64_t)p.stages * slab / p.nslab), st_end = (int)((int64_t)p.stages * (slab + 1) / p.nslab);
    const int S = st_end - st_beg;

    const int row = tid >> 4, seg = tid & 15;
    // dZ: (row, 16 consecutive columns): four float4 loads, two 16-byte chunks per plane
    const int a_h = seg >> 3, a_ch = 2 * (seg & 7);
    const int a_dst0 = a_h * XT_HALF + 256 * row + 16 * xt_chunk(row, a_ch, a_h);
    const int a_dst1 = a_h * XT_HALF + 256 * row + 16 * xt_chunk(row, a_ch + 1, a_h);
    // X: (row, 8 consecutive columns): two float4 loads, one chunk per plane
    const int b_dst = 2 * XT_APLANE + 256 * row + 16 * xt_chunk(row, seg, 0);
    const float cs_on = (p.Pc != nullptr && tk == 0) ? 1.0f : 0.0f;  // the bias gradient is summed by the k-tile-0 blocks (branch-free: a flag)
    float cs[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) cs[e] = 0.0f;

    // fragment addresses (see gemm_tn_tr.hip): lane l of a 32x32x16 operand = column l & 31 of its 32-column tile, reduction rows
    // 8 (l >> 5) .. +7 of the 16-row k-step; a 16-lane group reads a 4-row x 16-column block transposed
    const int wn = wave >> 1, wk = wave & 1;
    const int kg = lane >> 5, gi = (lane >> 4) & 1, q = (lane >> 2) & 3, pq = lane & 3;
    auto frag_off = [&](int col, int rd) {
        const int c = col + 16 * gi + 4 * pq;
        const int h = c >> 7, ch = (c & 127) >> 3;
        const int r = 8 * kg + 4 * rd + q;
        return h * XT_HALF + 256 * r + 16 * xt_chunk(r, ch, h) + 8 * (pq & 1);
    };
    int offA[2][2], offB[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            offA[t][rd] = frag_off(64 * wn + 32 * t, rd);
            offB[t][rd] = 2 * XT_APLANE + frag_off(64 * wk + 32 * t, rd);
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // loads walk the slab's stages in order: (gl, r0) = group (relative to the slab's first) and first row inside it of the next
    // stage to load.  The slab's gather indices are staged in LDS first: an index fetched from global memory inside the loop is
    // a dependent load in front of every stage's X loads (measured: the whole kernel then runs at the latency of that chain).
    // The steady-state loop must stay free of branches and of waits hipcc cannot count, see gemm_tn_tr.hip.
    const int64_t m_first = (int64_t)st_beg * XT_ROWS;
    const int grp_first = (int)(m_first / p.rpg);
    int gl = 0, r0 = (int)(m_first - (int64_t)grp_first * p.rpg);
    int32_t *gxs = reinterpret_cast<int32_t *>(xl + 2 * XT_STAGE), *grs = gxs + XT_MAXGRP;
    {
        const int ngrp = (int)(((int64_t)st_end * XT_ROWS - 1) / p.rpg) - grp_first + 1;
        for (int i = tid; i < ngrp; i += 512) {
            gxs[i] = GIDX ? p.b_gidx[grp_first + i] : grp_first + i;
            if (RS) grs[i] = p.rs_gidx ? p.rs_gidx[grp_first + i] : grp_first + i;
        }
        __syncthreads();
    }
    const float *a_src = p.A + (m_first + row) * p.lda + n0 + 16 * seg;
    const int64_t a_step = (int64_t)XT_ROWS * p.lda;
    // K % 128 != 0 (KTAIL; the text encoder's 300 input columns): the two float4 of a thread that lie past K are read from
    // column 0 instead and multiplied by zero, and the partial tile's columns past K are not stored
    const int kc = k0 + 8 * seg;
    const float bm0 = (!KTAIL || kc < p.K) ? 1.0f : 0.0f, bm1 = (!KTAIL || kc + 4 < p.K) ? 1.0f : 0.0f;
    const int b_off0 = row * (int)p.ldb + ((!KTAIL || kc < p.K) ? kc : 0), b_off1 = row * (int)p.ldb + ((!KTAIL || kc + 4 < p.K) ? kc + 4 : 0);
    auto load = [&](XtRegs &g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g.a[i] = *(const v4f *)(a_src + 4 * i);
        a_src += a_step;
        const float *bs = p.B + (int64_t)gxs[gl] * p.b_gstride + (int64_t)r0 * p.ldb;
        g.b[0] = *(const v4f *)(bs + b_off0);
        g.b[1] = *(const v4f *)(bs + b_off1);
        if (RS) g.rs = p.rs[(int64_t)grs[gl] * p.rs_gstride + r0 + row];
        const bool wrap = r0 + XT_ROWS >= p.rpg;
        gl += wrap ? 1 : 0;
        r0 = wrap ? 0 : r0 + XT_ROWS;
    };
    auto write = [&](int s, const XtRegs &g) {
        char *st = xl + (s & 1) * XT_STAGE;
        bf16x8 hi[2], lo[2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = g.a[i][e];
                const __bf16 hv = (__bf16)v;
                hi[i >> 1][4 * (i & 1) + e] = hv;
                lo[i >> 1][4 * (i & 1) + e] = (__bf16)(v - (float)hv);
                cs[4 * i + e] += cs_on * v;
            }
        *(bf16x8 *)(st + a_dst0) = hi[0];
        *(bf16x8 *)(st + a_dst1) = hi[1];
        *(bf16x8 *)(st + XT_APLANE + a_dst0) = lo[0];
        *(bf16x8 *)(st + XT_APLANE + a_dst1) = lo[1];
        bf16x8 bh, bl;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = RS ? g.b[i][e] * g.rs : g.b[i][e];
                if (KTAIL) v *= i ? bm1 : bm0;
                const __bf16 hv = (__bf16)v;
                bh[4 * i + e] = hv;
                bl[4 * i + e] = (__bf16)(v - (float)hv);
            }
        *(bf16x8 *)(st + b_dst) = bh;
        *(bf16x8 *)(st + XT_BPLANE + b_dst) = bl;
    };
    auto compute = [&](int s) {
        const char *st = xl + (s & 1) * XT_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const char *sk = st + ks * 4096;                          // 16 rows further
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bh[t] = join8(tr_read(sk, offB[t][0]), tr_read(sk, offB[t][1]));
                bl[t] = join8(tr_read(sk + XT_BPLANE, offB[t][0]), tr_read(sk + XT_BPLANE, offB[t][1]));
                ah[t] = join8(tr_read(sk, offA[t][0]), tr_read(sk, offA[t][1]));
                al[t] = join8(tr_read(sk + XT_APLANE, offA[t][0]), tr_read(sk + XT_APLANE, offA[t][1]));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[t], acc[i][t], 0, 0, 0);
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[t], acc[i][t], 0, 0, 0);
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[t], acc[i][t], 0, 0, 0);
                }
        }
    };

    // Stage s: barrier (every thread wrote its share of stage s; every wave finished reading stage s - 1, whose slot stage s + 1
    // is written into after this barrier); issue the loads of stage s + 2 into the register set stage s came from; multiply
    // stage s; split and write stage s + 1.
    // Invariant at the top of an iteration: stage s is written (slot s & 1), g1 holds the loaded stage s + 1, g0 is free.
    XtRegs g0, g1;
    g0.rs = g1.rs = 1.0f;
    load(g0);                                                        // S >= 2 (launcher)
    load(g1);
    write(0, g0);
    int s = 0;
    // (Tried: the two waves of a SIMD running a stage's two phases in opposite order -- 5 % slower; s_setprio(1) around the MFMA
    // clusters -- 3-5 % slower; __builtin_amdgcn_iglp_opt(0) in the multiply phase -- within noise, (1) -- 6 % slower
    // (tools/scratch/build_variant.sh, one box each).  Parts compiled out at
    // M = 109 056: loads + split + writes alone 0.52 of the full time, loads + multiplies alone 0.76: the phases add up.)
    for (; s + 4 <= S; s += 2) {                                     // steady state: both halves have a stage to load, no branches
        __syncthreads();
        load(g0);                                                    // stage s + 2
        __builtin_amdgcn_sched_barrier(0);                           // the loads stay above the multiplies (hipcc sinks them to the end of the stage otherwise: no lookahead)
        compute(s);
        write(s + 1, g1);
        __syncthreads();
        load(g1);                                                    // stage s + 3
        __builtin_amdgcn_sched_barrier(0);
        compute(s + 1);
        write(s + 2, g0);
    }
    const bool three = S - s == 3;                                   // 2 or 3 stages left
    __syncthreads();
    if (three) load(g0);
    compute(s);
    write(s + 1, g1);
    __syncthreads();
    compute(s + 1);
    if (three) {
        write(s + 2, g0);
        __syncthreads();
        compute(s + 2);
    }

    // ---- this slab's partial tile: plain stores (lanes 0..31 = 32 consecutive k = 128 bytes) ----
    float *P = p.P + (int64_t)slab * p.N * p.K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k = k0 + 64 * wk + 32 * t + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + 64 * wn + 32 * i + 8 * (e >> 2) + 4 * kg + (e & 3);
                if (!KTAIL || k < p.K) P[(int64_t)n * p.K + k] = acc[i][t][e];
            }
        }
    if (cs_on != 0.0f) {                                             // uniform per block
        __syncthreads();
        float *red = reinterpret_cast<float *>(xl);                  // [32 rows][256 columns]
#pragma unroll
        for (int e = 0; e < 16; ++e) red[row * 256 + 16 * seg + e] = cs[e];
        __syncthreads();
        if (tid < 256) {
            float sum = 0.0f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) sum += red[r * 256 + tid];
            p.Pc[(int64_t)slab * p.N + n0 + tid] = sum;
        }
    }
}

// dst[i] += sum over slabs, in slab order (deterministic); one launch for every weight of a step
struct XtReduceEntry { const float *P; float *dst; int nslab; int count4; };     // count4 float4 elements per slab
struct XtReduceBatch { XtReduceEntry e[40]; int n; };

__global__ __launch_bounds__(256) void tn_slab_reduce_kernel(XtReduceBatch b) {
    const XtReduceEntry e = b.e[blockIdx.y];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < e.count4; i += gridDim.x * 256) {
        v4f sum = *(const v4f *)(e.dst + 4 * (int64_t)i);
        const float *src = e.P + 4 * (int64_t)i;
        int s = 0;
        for (; s + 4 <= e.nslab; s += 4) {
            const v4f v0 = *(const v4f *)(src + (int64_t)(s + 0) * 4 * e.count4);
            const v4f v1 = *(const v4f *)(src + (int64_t)(s + 1) * 4 * e.count4);
            const v4f v2 = *(const v4f *)(src + (int64_t)(s + 2) * 4 * e.count4);
            const v4f v3 = *(const v4f *)(src + (int64_t)(s + 3) * 4 * e.count4);
            sum += v0; sum += v1; sum += v2; sum += v3;
        }
        for (; s < e.nslab; ++s) sum += *(const v4f *)(src + (int64_t)s * 4 * e.count4);
        *(v4f *)(e.dst + 4 * (int64_t)i) = sum;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
int tn_x3tr_slabs(int64_t M) {                                       // slabs of >= 2 stages: 8, 16, 24 or 32 (>= 8 stages each) where M allows
    const int64_t stages = M / XT_ROWS;
    if (stages < 16) return (int)std::max<int64_t>(1, stages / 2);
    return (int)std::min<int64_t>(XT_MAXSLAB, std::max<int64_t>(8, (stages / 8) & ~7ll));
}
int64_t tn_x3tr_scratch_floats(int64_t M, int64_t N, int64_t K) { return (int64_t)tn_x3tr_slabs(M) * (N * K + N); }

bool tn_x3tr_takes(const stair_gemm_tn_args &a) {
    static const bool on = [] { const char *e = getenv("STAIR_GEMM_TN_X3TR"); return !(e && e[0] == '0'); }();
    if (!on || a.b_is_bf16 || matmul_mode() != STAIR_MATMUL_BF16X3) return false;
    if (a.N % 256 || a.K % 4 || a.K < 4 || a.M % XT_ROWS || a.M < 2 * XT_ROWS || a.rows_per_group < 1) return false;
    if (a.rows_per_group % XT_ROWS && (a.rows_per_group != 1 || a.b_gidx || a.row_scale || a.b_gstride != a.ldb)) return false;   // groups of whole stages, or a plain row matrix
    if (a.K % 128 && (a.b_gidx || a.row_scale)) return false;
    if (a.colsum2 && !a.colsum) return false;
    if (a.lda % 4 || a.ldb % 4 || a.ldc != a.K || a.b_gstride % 4) return false;
    const int64_t rpg = a.rows_per_group % XT_ROWS ? a.M : a.rows_per_group;
    if (a.M / tn_x3tr_slabs(a.M) / rpg + 3 > XT_MAXGRP) return false;               // a slab's gather indices are staged in LDS
    if ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B) | reinterpret_cast<uintptr_t>(a.C)) & 15) return false;
    if (a.colsum && (reinterpret_cast<uintptr_t>(a.colsum) & 15)) return false;
    if (a.colsum2 && (reinterpret_cast<uintptr_t>(a.colsum2) & 15)) return false;
    return true;
}

static thread_local XtReduceBatch g_pending;                                      // entries queued by launch_gemm_tn_x3tr, flushed by tn_x3tr_flush (one thread per plan pass)

// Partial products of `a` into scratch (tn_x3tr_scratch_floats floats); the sums reach a.C / a.colsum at the next tn_x3tr_flush on the same stream.
int launch_gemm_tn_x3tr(const stair_gemm_tn_args &a, float *scratch, hipStream_t s) {
    XtParams p;
    p.A = a.A; p.lda = a.lda; p.B = a.B; p.ldb = a.ldb; p.b_gstride = a.b_gstride; p.b_gidx = a.b_gidx;
    p.rs = a.row_scale; p.rs_gstride = a.rs_gstride; p.rs_gidx = a.rs_gidx;
    p.N = a.N; p.K = a.K; p.rpg = a.rows_per_group % XT_ROWS ? a.M : a.rows_per_group;       // a plain row matrix is one group of M rows
    if (a.rows_per_group % XT_ROWS) p.b_gstride = 0; p.stages = a.M / XT_ROWS; p.nslab = tn_x3tr_slabs(a.M);
    p.tilesK = (a.K + 127) / 128;
    p.P = scratch; p.Pc = a.colsum ? scratch + (int64_t)p.nslab * a.N * a.K : nullptr;
    // the dynamic-LDS limit is an attribute per function AND device: set once for each device this process launches on
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_x3tr_kernel<false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, XT_LDS));
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_x3tr_kernel<true, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, XT_LDS));
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_x3tr_kernel<false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, XT_LDS));
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_x3tr_kernel<true, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, XT_LDS));
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_x3tr_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, XT_LDS));
        attr_set[dev] = true;
    }
    if (gemm_trace_on())
        fprintf(stderr, "STAIR_GEMM tn M=%d N=%d K=%d act=0 acc=1 gather=%d scale=%d\n", a.M, a.N, a.K, a.b_gidx ? 1 : 0, a.row_scale ? 1 : 0);
    STAIR_ACCT_MFMA("gemm_tn_x3tr", ((int64_t)a.M * a.N + (int64_t)a.M * a.K + (int64_t)a.N * a.K) * 4, 2ll * a.M * a.N * a.K);
    const dim3 grid(p.nslab * (a.N / 256) * p.tilesK);
    if (a.K % 128) hipLaunchKernelGGL((gemm_tn_x3tr_kernel<false, false, true>), grid, dim3(512), XT_LDS, s, p);
    else if (a.b_gidx && a.row_scale) hipLaunchKernelGGL((gemm_tn_x3tr_kernel<true, true, false>), grid, dim3(512), XT_LDS, s, p);
    else if (a.b_gidx) hipLaunchKernelGGL((gemm_tn_x3tr_kernel<true, false, false>), grid, dim3(512), XT_LDS, s, p);
    else if (a.row_scale) hipLaunchKernelGGL((gemm_tn_x3tr_kernel<false, true, false>), grid, dim3(512), XT_LDS, s, p);
    else hipLaunchKernelGGL((gemm_tn_x3tr_kernel<false, false, false>), grid, dim3(512), XT_LDS, s, p);
    STAIR_LAUNCH_CHECK();
    // one reduction launch adds every queued product to its destination in parallel: two products into the SAME buffer (a weight
    // used by two buckets outside the per-weight regions) must not share it -- the earlier one is added first
    bool same_dst = false;
    for (int i = 0; i < g_pending.n; ++i)
        same_dst |= g_pending.e[i].dst == a.C || (a.colsum && g_pending.e[i].dst == a.colsum) || (a.colsum2 && g_pending.e[i].dst == a.colsum2);
    if (same_dst || g_pending.n + 3 > 40) STAIR_CHECK(tn_x3tr_flush(s) == 0, "weight-gradient reduction failed");
    g_pending.e[g_pending.n++] = {p.P, a.C, p.nslab, (int)((int64_t)a.N * a.K / 4)};
    if (p.Pc) g_pending.e[g_pending.n++] = {p.Pc, a.colsum, p.nslab, a.N / 4};
    if (p.Pc && a.colsum2) g_pending.e[g_pending.n++] = {p.Pc, a.colsum2, p.nslab, a.N / 4};
    return 0;
}

void tn_x3tr_discard() { g_pending.n = 0; }

// another kernel's slab partials (csrc/gemm_tn_tr.hip) join the same fixed-order reduction
int tn_x3tr_queue(const float *P, float *dst, int nslab, int count4, hipStream_t s) {
    for (int i = 0; i < g_pending.n; ++i)
        if (g_pending.e[i].dst == dst) { STAIR_CHECK(tn_x3tr_flush(s) == 0, "weight-gradient reduction failed"); break; }
    if (g_pending.n + 1 > 40) STAIR_CHECK(tn_x3tr_flush(s) == 0, "weight-gradient reduction failed");
    g_pending.e[g_pending.n++] = {P, dst, nslab, count4};
    return 0;
}

int tn_x3tr_flush(hipStream_t s) {
    if (g_pending.n == 0) return 0;
    XtReduceBatch b = g_pending;
    g_pending.n = 0;
    int most = 0;
    for (int i = 0; i < b.n; ++i) most = std::max(most, b.e[i].count4);
    hipLaunchKernelGGL(tn_slab_reduce_kernel, dim3((most + 255) / 256, b.n), dim3(256), 0, s, b);
    STAIR_LAUNCH_CHECK();
    return 0;
}

}  // namespace stair

extern "C" int64_t stair_gemm_tn_slabs_scratch(int64_t M, int64_t N, int64_t K) { return stair::tn_x3tr_scratch_floats(M, N, K); }

extern "C" int stair_gemm_tn_slabs(const stair_gemm_tn_args *args, float *scratch, int64_t scratch_floats, stair_stream stream) {
    STAIR_CHECK(args && scratch, "null argument");
    STAIR_CHECK(stair::tn_x3tr_takes(*args), "shape not supported: needs the split matmul mode, fp32 operands, N % 256 == 0, K % 128 == 0, ldc == K, "
                                             "M and rows_per_group multiples of 32, 16-byte aligned rows");
    STAIR_CHECK(scratch_floats >= stair::tn_x3tr_scratch_floats(args->M, args->N, args->K), "scratch too small (stair_gemm_tn_slabs_scratch)");
    const hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = stair::launch_gemm_tn_x3tr(*args, scratch, s)) return rc;
    return stair::tn_x3tr_flush(s);
}
