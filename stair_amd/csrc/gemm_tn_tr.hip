// Weight-gradient GEMM of the video encoder's input projection on stored-bf16 clip features:
//     dW_ih[n][k] += sum_m dG[m][n] * X[m][k]        (autograd of nn.LSTM's x W_ih^T, /root/reference/video_nmn/module_net.py:39-42,
//                                                      train_module.py:408; M = clips x frames = 131 072, N = 4 Hh = 1024, K = V = 2048)
// dG (gate gradients, fp32, written by the BPTT kernel) and X (clip features, exact bf16) are both stored with the reduction
// index m as the ROW index, i.e. transposed with respect to what an MFMA operand wants (8 consecutive reduction elements per
// lane).  csrc/gemm_bf16x3.hip's TN kernels transpose 8 x 4 blocks in registers on the way into LDS; here the tiles keep
// their natural row-major layout in LDS and the fragments are read TRANSPOSED by the hardware (gfx950 ds_read_b64_tr_b16:
// a 16-lane group reads a 4-row x 16-column block of 16-bit elements and gets it column-major), so
//   * X needs no register round trip at all: global -> LDS by LDS-DMA (global_load_lds_dwordx4), swizzle on the SOURCE address;
//   * dG is split once per element into bf16 hi + lo on its way through registers (no transposes) and written row-major;
//   * a product is dGhi * X + dGlo * X (X is exact): two v_mfma_f32_32x32x16_bf16 per pair, fp32 accumulate.
// 256 (n) x 256 (k) output tile per workgroup, 8 waves as 4 (n) x 2 (k) of 64 x 128, reduction in stages of 32 rows through a
// ring of 3 LDS stages (3 planes of 16 KB each: dG hi, dG lo, X), one raw s_barrier per stage, the DMA of two stages ahead
// and the dG loads of two stages ahead in flight across it behind a counted s_waitcnt vmcnt.  M is cut into 8 slabs (one per
// XCD, so that a slab's rows leave HBM once and are shared through that XCD's L2 by the 32 tiles); slabs add up with fp32 atomics.
//
// LDS image of a plane of one stage: two half images (columns 0..127, 128..255) of [32 rows][256 B]; the 16-byte chunk ch of
// row r of half h sits at chunk  ch ^ (((r & 3) << 2) | ((r >> 2) & 3)) ^ h  (cdna_hip_programming.md T10, image (b)): the
// transposed reads of a 32-lane half touch all 64 banks once, and so do the 16-byte staging writes of a row.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace stair {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using v4f = __attribute__((ext_vector_type(4))) float;

constexpr int TR_ROWS = 32;                   // reduction rows per stage
constexpr int TR_HALF = TR_ROWS * 256;        // one 128-column half image: 8 KB
constexpr int TR_PLANE = 2 * TR_HALF;         // 256 columns: 16 KB
constexpr int TR_STAGE = 3 * TR_PLANE;        // dG hi, dG lo, X: 48 KB
constexpr int TR_NST = 3;

struct TrParams {
    const float *A; int64_t lda;              // dG [M, lda], columns n
    const __bf16 *B; int64_t ldb;             // X  [M, ldb], columns k
    float *C; int64_t ldc;                    // dW [N, ldc]
    int M, N, K, mslab, tilesN, tilesK;
    long long *C64;                           // fixed-point shadow of C (det_shadow): the 8 slabs add in any order, the sum is the same
    float *P;                                 // or: slab partials [8][N][K], STORED; tn_slab_reduce_kernel adds them to C in slab order
};

__device__ __forceinline__ int tr_chunk(int row, int ch, int h) { return ch ^ (((row & 3) << 2) | ((row >> 2) & 3)) ^ h; }

// LDS accesses of the main loop are inline asm: hipcc makes every LDS access it can see wait (vmcnt) for ALL LDS-DMA issued
// before it -- with the builtin form of the transposed read each stage drained the DMA and the dG loads of two stages ahead
// before its first fragment read, i.e. there was no lookahead at all (1.36 ms; loads alone 0.77, multiplies alone ~1.07).
// What hipcc cannot see it does not wait for; the waits that are needed are written out (lgkmcnt for the fragments, the
// counted vmcnt + barrier for the stages).
using v2u = __attribute__((ext_vector_type(2))) unsigned;
using v4u = __attribute__((ext_vector_type(4))) unsigned;
template <int OFF>
__device__ __forceinline__ v2u tr_read(unsigned addr) {
    v2u r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int OFF>
__device__ __forceinline__ void lds_write16(unsigned addr, v4u v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ v4f gload16(const float *p) {           // asynchronous: the caller waits (vmcnt) with the value as an operand
    v4f r;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
    return r;
}
__device__ __forceinline__ bf16x8 join8(v2u a, v2u b) {
    const v4u v = v4u{a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(bf16x8, v);
}

}  // namespace

// ABL (diagnostic builds only, STAIR_TN_TR_ABLATE): 1 = no dG loads / splits / writes, 2 = no fragment reads / MFMAs, 3 = no DMA of X
template <int ABL>
__global__ __launch_bounds__(512, 1) void gemm_tn_tr_kernel(TrParams p) {
    extern __shared__ __attribute__((aligned(16))) char tl[];      // [3 stages][3 planes][2 halves][32 rows][256 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // one slab per XCD (blocks with equal blockIdx % 8 share an XCD: speed only)
    const int slab = blockIdx.x & 7, tile = blockIdx.x >> 3;
    const int tn = tile / p.tilesK, tk = tile - tn * p.tilesK;
    const int n0 = tn * 256, k0 = tk * 256;
    const int m_beg = slab * p.mslab, m_end = min(p.M, m_beg + p.mslab);
    const int S = (m_end - m_beg) / TR_ROWS;                       // whole stages, an even number >= 2 of them (the launcher checks)
    if (S < 2) return;

    // ---- staging roles ----
    // dG: thread = (row 0..31, 16 consecutive columns): 4 float4 loads, split, two 16-byte writes per plane
    const int g_row = tid >> 4, g_seg = tid & 15;
    const int g_h = g_seg >> 3, g_ch = 2 * (g_seg & 7);
    const int g_src = g_row * (int)p.lda + 16 * g_seg;             // float offset inside the slab's column block (32-bit: < 2^31 checked by the launcher)
    const float *g_base = p.A + (int64_t)m_beg * p.lda + n0;
    const int g_dst0 = g_h * TR_HALF + 256 * g_row + 16 * tr_chunk(g_row, g_ch, g_h);
    const int g_dst1 = g_h * TR_HALF + 256 * g_row + 16 * tr_chunk(g_row, g_ch + 1, g_h);
    // X: wave w issues DMA instructions 2w, 2w+1 of the 16 that fill a stage's plane; instruction idx fills rows
    // 4 (idx & 7) .. +3 of half idx >> 3 linearly (1 KB), lane = (row in group, physical chunk)
    int x_dst[2], x_src[2];
    const __bf16 *x_base = p.B + (int64_t)m_beg * p.ldb + k0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = 2 * wave + i, h = idx >> 3, rg = idx & 7;
        const int row = 4 * rg + (lane >> 4);
        const int ch = tr_chunk(row, lane & 15, h);                // logical chunk stored at physical slot lane & 15 (the XOR is an involution)
        x_dst[i] = __builtin_amdgcn_readfirstlane(2 * TR_PLANE + h * TR_HALF + rg * 1024);
        x_src[i] = row * (int)p.ldb + 128 * h + 8 * ch;
    }
    // ---- fragment addresses (byte offsets inside a plane, stage-relative) ----
    // lane l of a 32x32x16 operand: column l & 31 of its 32-column tile, reduction rows 8 (l >> 5) .. +7 of the 16-row k-step.
    // transposed read: 16-lane group gi = (l >> 4) & 1 covers columns 16 gi .. +15; lane 4 q + pq of the group supplies the
    // address of row (base + q), columns 4 pq .. +3
    const int wn = wave >> 1, wk = wave & 1;                       // wave tile: n 64 wn .. +63, k 128 wk .. +127
    const int kg = lane >> 5, gi = (lane >> 4) & 1, q = (lane >> 2) & 3, pq = lane & 3;
    auto frag_off = [&](int col, int rd) {                         // col: first column of the 32-column tile inside the 256-column tile
        const int c = col + 16 * gi + 4 * pq;
        const int h = c >> 7, ch = (c & 127) >> 3;
        const int row = 8 * kg + 4 * rd + q;                        // + 16 per k-step: leaves (row & 3) and ((row >> 2) & 3) unchanged
        return h * TR_HALF + 256 * row + 16 * tr_chunk(row, ch, h) + 8 * (pq & 1);
    };
    int offA[2][2], offB[4][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) offA[t][rd] = frag_off(64 * wn + 32 * t, rd);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) offB[t][rd] = frag_off(128 * wk + 32 * t, rd);

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)tl;
    auto issue_x = [&](int s) {                                    // stage s of this slab -> ring slot s % 3
        if (ABL == 3) return;
        const unsigned st = lds0 + (s % TR_NST) * TR_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const __bf16 *src = x_base + (int64_t)s * TR_ROWS * p.ldb + x_src[i];
            const unsigned dst = st + x_dst[i];
            unsigned keep;
            // M0 = LDS destination base of the DMA; it is compiler-reserved, so it is saved, set and restored in one statement
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };
    // dG goes through ordinary loads: hipcc counts them and waits for exactly the set it is about to split.  (X's DMA is asm
    // and therefore NOT counted by hipcc: its counted waits for dG may leave fewer operations in flight than they say, never more.)
    auto load_g = [&](int s, v4f (&g)[4]) {
        if (ABL == 1) return;
        const float *src = g_base + (int64_t)s * TR_ROWS * p.lda + g_src;
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = *(const __attribute__((address_space(1))) v4f *)(src + 4 * i);
    };
    auto write_g = [&](int s, const v4f (&g)[4]) {
        if (ABL == 1) return;
        const unsigned st = lds0 + (s % TR_NST) * TR_STAGE;
        bf16x8 hi[2], lo[2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const __bf16 hv = (__bf16)g[i][e];
                hi[i >> 1][4 * (i & 1) + e] = hv;
                lo[i >> 1][4 * (i & 1) + e] = (__bf16)(g[i][e] - (float)hv);
            }
        lds_write16<0>(st + g_dst0, __builtin_bit_cast(v4u, hi[0]));
        lds_write16<0>(st + g_dst1, __builtin_bit_cast(v4u, hi[1]));
        lds_write16<TR_PLANE>(st + g_dst0, __builtin_bit_cast(v4u, lo[0]));
        lds_write16<TR_PLANE>(st + g_dst1, __builtin_bit_cast(v4u, lo[1]));
    };
    auto compute = [&](int s) {
        if (ABL == 2) return;
        const unsigned st = lds0 + (s % TR_NST) * TR_STAGE;
#define TR_KSTEP(KS_)                                                                                                     \
        {                                                                                                                 \
            v2u rb[4][2], rh[2][2], rl[2][2];                                                                             \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                               \
                rb[t][0] = tr_read<2 * TR_PLANE + (KS_) * 4096>(st + offB[t][0]);                                         \
                rb[t][1] = tr_read<2 * TR_PLANE + (KS_) * 4096>(st + offB[t][1]);                                         \
            }                                                                                                             \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                               \
                rh[i][0] = tr_read<(KS_) * 4096>(st + offA[i][0]);                                                        \
                rh[i][1] = tr_read<(KS_) * 4096>(st + offA[i][1]);                                                        \
                rl[i][0] = tr_read<TR_PLANE + (KS_) * 4096>(st + offA[i][0]);                                             \
                rl[i][1] = tr_read<TR_PLANE + (KS_) * 4096>(st + offA[i][1]);                                             \
            }                                                                                                             \
            /* the fragments are in flight: nothing may touch them before this wait (the operands tie every later use to it) */ \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                           \
                         : "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[2][0]), "+v"(rb[2][1]),   \
                           "+v"(rb[3][0]), "+v"(rb[3][1]), "+v"(rh[0][0]), "+v"(rh[0][1]), "+v"(rh[1][0]), "+v"(rh[1][1]),   \
                           "+v"(rl[0][0]), "+v"(rl[0][1]), "+v"(rl[1][0]), "+v"(rl[1][1])                                    \
                         :: "memory");                                                                                    \
            bf16x8 b[4];                                                                                                  \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) b[t] = join8(rb[t][0], rb[t][1]);                               \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                               \
                const bf16x8 ah = join8(rh[i][0], rh[i][1]), al = join8(rl[i][0], rl[i][1]);                              \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                           \
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b[t], acc[i][t], 0, 0, 0);                    \
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[t], acc[i][t], 0, 0, 0);                    \
                }                                                                                                         \
            }                                                                                                             \
        }
        TR_KSTEP(0)
        TR_KSTEP(1)
#undef TR_KSTEP
    };

    // ---- prologue: stage 0 complete in LDS, stage 1's loads in flight ----
    // Vector-memory operations per stage, in issue order: dG (4 loads), then X's DMA (2 instructions).
    v4f ga[4], gb[4];
    load_g(0, ga);
    issue_x(0);
    load_g(1, gb);
    issue_x(1);
    write_g(0, ga);

    // Stage s: barrier (stage s is in LDS: every wave wrote its share of dG(s) and retired its share of X(s) before arriving);
    // issue stage s + 2 (its ring slot was last read in stage s - 1); multiply stage s; split dG(s + 1) and write it (its slot
    // was last read in stage s - 2).  The steady-state loop is branch-free -- on a path that skips a wait hipcc has to assume
    // the loads still pending and drains everything (vmcnt(0)) at the next use -- and the last two stages are peeled.
    auto stage = [&](int s, v4f (&g_next)[4], v4f (&g_next2)[4]) {
        // g_next holds dG(s+1) (in flight or landed); g_next2 is free and receives dG(s+2)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's ds_writes of dG(s) are done
        // X(s) has landed once at most the 6 operations issued after its DMA -- dG(s+1) and X(s+1) -- are outstanding
        if (ABL == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        load_g(s + 2, g_next2);
        issue_x(s + 2);
        compute(s);
        write_g(s + 1, g_next);
    };
    int s = 0;
    for (; s + 2 < S; s += 2) {                                    // S is even (launcher): both stages of an iteration have a stage s + 2
        stage(s, gb, ga);
        stage(s + 1, ga, gb);
    }
    // stage S - 2: nothing left to issue
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ABL == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    compute(s);
    write_g(s + 1, gb);
    // stage S - 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    compute(s + 1);

    // ---- epilogue: this slab's partial sums ----
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = k0 + 128 * wk + 32 * t + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + 64 * wn + 32 * i + 8 * (e >> 2) + 4 * kg + (e & 3);
                if (p.P) p.P[((int64_t)slab * p.N + n) * p.K + k] = acc[i][t][e];
                else grad_add(p.C, p.C64, (int64_t)n * p.ldc + k, acc[i][t][e]);
            }
        }
}

// Returns -1 when the shape is not this kernel's (the caller falls back to csrc/gemm_bf16x3.hip), 0 on launch, 1 on error.
int launch_gemm_tn_tr(const stair_gemm_tn_args &a, hipStream_t s) { return launch_gemm_tn_tr_slabs(a, nullptr, s); }

// scratch (8 * N * K floats, needs ldc == K): the slabs' partial products are stored there and queued for the fixed-order reduction
// of csrc/gemm_tn_x3tr.hip (tn_x3tr_flush on the same stream adds them to C): no atomics, the same sum every run
int launch_gemm_tn_tr_slabs(const stair_gemm_tn_args &a, float *scratch, hipStream_t s) {
    static const bool on = [] { const char *e = getenv("STAIR_GEMM_TN_TR"); return !(e && e[0] == '0'); }();
    if (!on || !a.b_is_bf16 || a.row_scale || a.b_gidx || a.colsum || a.colsum2) return -1;
    if (a.rows_per_group != 1 && a.b_gstride != (int64_t)a.rows_per_group * a.ldb) return -1;
    if (a.N % 256 || a.K % 256 || a.M % (8 * 64) || a.M < 2048) return -1;        // >= 4 stages per slab; measured down to M = 2048 (32 questions)
    if (32 * a.lda + 256 >= (1ll << 31) || 32 * a.ldb + 256 >= (1ll << 31)) return -1;
    if (a.lda % 4 || a.ldb % 8 || (reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.B) & 15)) return -1;
    if (matmul_mode() != STAIR_MATMUL_BF16X3) return -1;
    TrParams p;
    p.A = a.A; p.lda = a.lda; p.B = reinterpret_cast<const __bf16 *>(a.B); p.ldb = a.ldb; p.C = a.C; p.ldc = a.ldc;
    p.M = a.M; p.N = a.N; p.K = a.K;
    p.P = scratch && a.ldc == a.K ? scratch : nullptr;
    p.C64 = p.P ? nullptr : det_shadow(a.C);
    p.tilesN = a.N / 256; p.tilesK = a.K / 256;
    p.mslab = a.M / 8;                                             // a multiple of 64: whole stages, an even number of them
    const size_t shmem = (size_t)TR_NST * TR_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_tr_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_tr_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_tr_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_tr_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        attr_set = true;
    }
    static const int abl = [] { const char *e = getenv("STAIR_TN_TR_ABLATE"); return e ? atoi(e) : 0; }();     // timing experiments: wrong results
    const dim3 grid(8 * p.tilesN * p.tilesK);
    STAIR_ACCT_MFMA("gemm_tn_tr", (int64_t)a.M * a.N * 4 + (int64_t)a.M * a.K * 2 + (int64_t)a.N * a.K * 4, 2ll * a.M * a.N * a.K);
    if (abl == 1) hipLaunchKernelGGL(gemm_tn_tr_kernel<1>, grid, dim3(512), shmem, s, p);
    else if (abl == 2) hipLaunchKernelGGL(gemm_tn_tr_kernel<2>, grid, dim3(512), shmem, s, p);
    else if (abl == 3) hipLaunchKernelGGL(gemm_tn_tr_kernel<3>, grid, dim3(512), shmem, s, p);
    else hipLaunchKernelGGL(gemm_tn_tr_kernel<0>, grid, dim3(512), shmem, s, p);
    STAIR_LAUNCH_CHECK();
    if (p.P) return tn_x3tr_queue(p.P, a.C, 8, (int)((int64_t)a.N * a.K / 4), s);
    return 0;
}

}  // namespace stair
