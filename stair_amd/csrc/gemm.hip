// fp32 MFMA GEMM with group gather/scatter, row scaling and fused bias/activation epilogue.
//
// Stands in for every nn.Linear on STAIR's NMN path (/root/reference/video_nmn/modules.py) once
// program nodes of one kind are packed into a launch: C = act((rs * A) W^T + b) with
// A [M,K] gathered in groups of `rows_per_group` rows (a [T,H] tile of one program node, or one
// [H] vector), W [N,K] exactly as nn.Linear stores it.
//
// gfx950 mapping: 128x128x32 block tile, 4 waves as 2x2, each wave 2x2 v_mfma_f32_32x32x2_f32
// tiles (exact fp32: one rounding per product, k-ordered -- MI355X_MICROARCH "FP32-input MFMA").
// Both operands are K-contiguous, so each lane stages float4s along K and the k index inside an
// 8-wide k group is permuted identically for A and B (half-wave h owns k = 8q+4h..8q+4h+3):
// fragments are then single ds_read_b128.  LDS image [kq][row ^ kq][4] is conflict-free for the
// ds_write_b128 of the staging pass and the ds_read_b128 of the fragment pass.  LDS is double
// buffered with global->register prefetch one chunk ahead (one barrier per chunk).  Blocks are
// renumbered so that the column tiles sharing one A row-panel run on the same XCD (private L2).
#include <algorithm>
#include <cstdlib>
#include <string>

#include "common.h"

namespace stair {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int KQ = BK / 4;                    // float4 columns per chunk
constexpr int LDS_FLOATS = 2 * 2 * KQ * 128 * 4;  // [buf][A|B][kq][row][4]

struct GemmParams {
    stair_gemm_args a;
    int M, tilesM, tilesN;
};

__device__ __forceinline__ int lds_off(int buf, int which, int kq, int row) {
    return ((((buf * 2 + which) * KQ + kq) * 128) + (row ^ kq)) * 4;
}

using v4f = __attribute__((ext_vector_type(4))) float;
// explicit global address space: pointers that arrive inside a by-value struct are otherwise treated
// as generic and lowered to flat_load, which counts on lgkmcnt as well and so drains the prefetch at
// every LDS wait of the fragment loop.
using gv4p = const __attribute__((address_space(1))) v4f *;

template <int ACT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const stair_gemm_args &a = p.a;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware bijective renumbering: blocks with equal blockIdx%8 share an XCD (speed only)
    const int nb = p.tilesM * p.tilesN;
    const int bid = blockIdx.x;
    const int qd = nb >> 3, rm = nb & 7, xcd = bid & 7, loc = bid >> 3;
    const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    const int tm = logical / p.tilesN, tn = logical - tm * p.tilesN;
    const int m0 = tm * BM, n0 = tn * BN;
    const int R = a.rows_per_group;
    const int K = a.K;

    // staging assignment: thread -> float4 column kq, rows r0 + 32 i.  Rows past M / N are clamped
    // to the last valid row: their products are computed and never stored (the epilogue masks them),
    // so the loads need no predicate.  Columns past K are clamped too and zeroed through the scale.
    const int kq = tid & 7, r0 = tid >> 3;
    const float *aptr0, *aptr1, *aptr2, *aptr3, *wptr0, *wptr1, *wptr2, *wptr3;
    float rs0, rs1, rs2, rs3;
    {
        const float *ap[4];
        const float *wp[4];
        float rs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = min(m0 + r0 + 32 * i, p.M - 1);
            const int g = m / R, rr = m - g * R;
            const int64_t gi = a.a_gidx ? a.a_gidx[g] : g;
            ap[i] = a.A + gi * a.a_gstride + (int64_t)rr * a.lda;
            rs[i] = 1.0f;
            if (a.row_scale) {
                const int64_t si = a.rs_gidx ? a.rs_gidx[g] : g;
                rs[i] = a.row_scale[si * a.rs_gstride + rr];
            }
            const int n = min(n0 + r0 + 32 * i, a.N - 1);
            wp[i] = a.W + (int64_t)n * a.ldw;
        }
        aptr0 = ap[0]; aptr1 = ap[1]; aptr2 = ap[2]; aptr3 = ap[3];
        wptr0 = wp[0]; wptr1 = wp[1]; wptr2 = wp[2]; wptr3 = wp[3];
        rs0 = rs[0]; rs1 = rs[1]; rs2 = rs[2]; rs3 = rs[3];
    }

    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc00[e] = acc01[e] = acc10[e] = acc11[e] = 0.0f;

    v4f ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    float kmask;
#define STAIR_GLOAD(k0)                                                       \
    {                                                                         \
        const int kraw = (k0) + 4 * kq;                                       \
        const int k = min(kraw, K - 4);                                       \
        kmask = kraw < K ? 1.0f : 0.0f;                                       \
        ra0 = *(gv4p)(aptr0 + k); ra1 = *(gv4p)(aptr1 + k);                   \
        ra2 = *(gv4p)(aptr2 + k); ra3 = *(gv4p)(aptr3 + k);                   \
        rb0 = *(gv4p)(wptr0 + k); rb1 = *(gv4p)(wptr1 + k);                   \
        rb2 = *(gv4p)(wptr2 + k); rb3 = *(gv4p)(wptr3 + k);                   \
    }
#define STAIR_LSTORE(buf)                                                                     \
    {                                                                                         \
        *reinterpret_cast<v4f *>(&lds[lds_off(buf, 0, kq, r0)]) = ra0 * (rs0 * kmask);        \
        *reinterpret_cast<v4f *>(&lds[lds_off(buf, 0, kq, r0 + 32)]) = ra1 * (rs1 * kmask);   \
        *reinterpret_cast<v4f *>(&lds[lds_off(buf, 0, kq, r0 + 64)]) = ra2 * (rs2 * kmask);   \
        *reinterpret_cast<v4f *>(&lds[lds_off(buf, 0, kq, r0 + 96)]) = ra3 * (rs3 * kmask);   \
        *reinterpret_cast<v4f *>(&lds[lds_off(buf, 1, kq, r0)]) = rb0;                        \
        *reinterpret_cast<v4f *>(&lds[lds_off(buf, 1, kq, r0 + 32)]) = rb1;                   \
        *reinterpret_cast<v4f *>(&lds[lds_off(buf, 1, kq, r0 + 64)]) = rb2;                   \
        *reinterpret_cast<v4f *>(&lds[lds_off(buf, 1, kq, r0 + 96)]) = rb3;                   \
    }

    const int nchunks = (K + BK - 1) / BK;
    STAIR_GLOAD(0);
    STAIR_LSTORE(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        // prefetch the next chunk into registers (the last iteration re-reads its own chunk: the
        // loads stay unconditional and the data is simply not used)
        STAIR_GLOAD(min(c + 1, nchunks - 1) * BK);
        __builtin_amdgcn_sched_barrier(0);   // keep the loads ABOVE the MFMA phase (hipcc sinks them otherwise)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int fq = 2 * q + h;
            const v4f a0 = *reinterpret_cast<const v4f *>(&lds[lds_off(buf, 0, fq, wm * 64 + r)]);
            const v4f a1 = *reinterpret_cast<const v4f *>(&lds[lds_off(buf, 0, fq, wm * 64 + 32 + r)]);
            const v4f b0 = *reinterpret_cast<const v4f *>(&lds[lds_off(buf, 1, fq, wn * 64 + r)]);
            const v4f b1 = *reinterpret_cast<const v4f *>(&lds[lds_off(buf, 1, fq, wn * 64 + 32 + r)]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc11, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        STAIR_LSTORE(buf ^ 1);
        __syncthreads();
    }
#undef STAIR_GLOAD
#undef STAIR_LSTORE

    // epilogue: per-row output offsets through LDS (the group gather needs a division per row)
    long long *rowoff = reinterpret_cast<long long *>(lds);
    if (tid < BM) {
        const int m = m0 + tid;
        long long off = -1;
        if (m < p.M) {
            const int g = m / R, rr = m - g * R;
            const int64_t gi = a.c_gidx ? a.c_gidx[g] : g;
            off = gi * a.c_gstride + (int64_t)rr * a.ldc;
        }
        rowoff[tid] = off;
    }
    __syncthreads();
    __attribute__((address_space(1))) float *Cg = (__attribute__((address_space(1))) float *)a.C;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + wn * 64 + nt * 32 + r;
        if (n >= a.N) continue;
        const float b = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const f32x16 &acc = mt == 0 ? (nt == 0 ? acc00 : acc01) : (nt == 0 ? acc10 : acc11);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rowl = wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const long long off = rowoff[rowl];
                if (off < 0) continue;
                float v = acc[e] + b;
                if (ACT == 1) v = fmaxf(v, 0.0f);
                if (ACT == 2) v = sigmoid_acc(v);
                if (a.accumulate) unsafeAtomicAdd(a.C + off + n, v);   // several groups may share an output slot
                else Cg[off + n] = v;
            }
        }
    }
}

// ---- contraction arithmetic: exact fp32 MFMA or split bf16x3 (csrc/gemm_bf16x3.hip) --------------------
static int g_matmul_mode = -1;
int matmul_mode() {
    if (g_matmul_mode < 0) {
        const char *e = getenv("STAIR_MATMUL");
        g_matmul_mode = (e && std::string(e) == "f32") ? STAIR_MATMUL_F32 : (e && std::string(e) == "bf16") ? STAIR_MATMUL_BF16 : STAIR_MATMUL_BF16X3;
    }
    return policy_or(STAIR_OPT_MATMUL_MODE, g_matmul_mode);
}
void set_matmul_mode(int m) { g_matmul_mode = m; }
// Row threshold of the split kernels.  Default 1 = every GEMM: a question's result must not depend on how many
// other questions share its launch (tests/test_gpu_parity.py::test_batch_composition...), and the split kernel's
// chunk (24 bf16 MFMAs + conversions) is also shorter than the fp32 kernel's (64 fp32 MFMAs) for small, latency
// bound launches.
static int kSplitMinRows = 1;
void set_split_min_rows(int r) { kSplitMinRows = r; }

int launch_gemm(const stair_gemm_args &a, hipStream_t s) {
    STAIR_CHECK(a.groups >= 0 && a.rows_per_group > 0 && a.N > 0 && a.K > 0, "bad shape");
    STAIR_CHECK(a.K % 4 == 0 && a.lda % 4 == 0 && a.ldw % 4 == 0 && a.a_gstride % 4 == 0,
                "K, lda, ldw and a_gstride must be multiples of 4 floats");
    STAIR_CHECK((reinterpret_cast<uintptr_t>(a.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.W) & 15) == 0,
                "A and W must be 16-byte aligned");
    STAIR_CHECK(a.act >= 0 && a.act <= 2, "act must be 0, 1 or 2");
    STAIR_CHECK(!(a.accumulate && a.act), "accumulate is only defined for act == 0");
    GemmParams p;
    p.a = a;
    const int64_t M = (int64_t)a.groups * a.rows_per_group;
    if (M == 0) return 0;
    STAIR_CHECK(M < (1ll << 31), "M too large");
    if (gemm_trace_on())
        fprintf(stderr, "STAIR_GEMM nt M=%lld N=%d K=%d act=%d acc=%d gather=%d scale=%d\n", (long long)M, a.N, a.K, a.act, a.accumulate,
                a.a_gidx || a.c_gidx ? 1 : 0, a.row_scale ? 1 : 0);
    if (matmul_mode() != STAIR_MATMUL_F32 && M >= kSplitMinRows) return launch_gemm_bf16x3(a, s);
    p.M = (int)M;
    p.tilesM = (p.M + BM - 1) / BM;
    p.tilesN = (a.N + BN - 1) / BN;
    const dim3 grid(p.tilesM * p.tilesN), block(256);
    const size_t shmem = LDS_FLOATS * sizeof(float);
    switch (a.act) {
        case 0: hipLaunchKernelGGL(gemm_f32_kernel<0>, grid, block, shmem, s, p); break;
        case 1: hipLaunchKernelGGL(gemm_f32_kernel<1>, grid, block, shmem, s, p); break;
        default: hipLaunchKernelGGL(gemm_f32_kernel<2>, grid, block, shmem, s, p); break;
    }
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Weight-gradient GEMM:  C[n][k] += sum_m A[m][n] * B[m][k]   (dW = dZ^T X of every nn.Linear).
// The contraction runs over ROWS of both operands, so a 32-row slab of A and of B is staged as is
// (float4 along n / k, LDS images [m][128+4]) and the MFMA fragments are single ds_read_b32 with the
// lane on the n / k axis (consecutive banks).  M is split over blockIdx.z; partial tiles are summed
// with fp32 atomics (one 128x128 tile per block: 64 KB of adds per ~2048-row slab, far below the
// chip-wide atomic rate).  B rows can be gathered in groups exactly like the forward kernel's A.
struct GemmTnParams {
    const float *A; int64_t lda;                    // [M, N]
    const float *B; int64_t ldb, b_gstride; const int32_t *b_gidx; int R;   // [M, K] in groups of R rows
    const float *row_scale; int64_t rs_gstride; const int32_t *rs_gidx;     // optional scale on B rows
    float *C; int64_t ldc;                          // [N, K]
    int M, N, K, mslab, tilesN, tilesK;
};

constexpr int TN_LD = 128 + 4;

__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(GemmTnParams p) {
    __shared__ __attribute__((aligned(16))) float As[32 * TN_LD];
    __shared__ __attribute__((aligned(16))) float Bs[32 * TN_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    // one M-slab per XCD (see gemm_tn_bf16x3_kernel): blocks with equal blockIdx % 8 walk the same rows
    const int tiles = p.tilesN * p.tilesK;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile = jb % tiles, slab = (jb / tiles) * 8 + xcd;
    const int n0 = (tile % p.tilesN) * 128, k0 = (tile / p.tilesN) * 128;
    const int mbeg = min(slab * p.mslab, p.M), mend = min(p.M, mbeg + p.mslab);

    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc00[e] = acc01[e] = acc10[e] = acc11[e] = 0.0f;

    // staging: thread -> column quad c4 (0..31) and rows mr + 8 i of the 32-row slab
    const int c4 = tid & 31, mr = tid >> 5;
    const int an = min(n0 + 4 * c4, p.N - 4), bk = min(k0 + 4 * c4, p.K - 4);   // clamped (N, K multiples of 4)
    const float amask = n0 + 4 * c4 < p.N ? 1.0f : 0.0f, bmask = k0 + 4 * c4 < p.K ? 1.0f : 0.0f;
    for (int m0 = mbeg; m0 < mend; m0 += 32) {
        v4f ra[4], rb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mraw = m0 + mr + 8 * i;
            const int m = min(mraw, p.M - 1);
            const float live = mraw < mend ? 1.0f : 0.0f;
            ra[i] = *(gv4p)(p.A + (int64_t)m * p.lda + an) * (live * amask);
            const int g = m / p.R, rr = m - g * p.R;
            const int64_t gi = p.b_gidx ? p.b_gidx[g] : g;
            float sc = bmask;
            if (p.row_scale) sc *= p.row_scale[(p.rs_gidx ? p.rs_gidx[g] : g) * p.rs_gstride + rr];
            rb[i] = *(gv4p)(p.B + gi * p.b_gstride + (int64_t)rr * p.ldb + bk) * sc;
        }
        __syncthreads();      // previous slab fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<v4f *>(&As[(mr + 8 * i) * TN_LD + 4 * c4]) = ra[i];
            *reinterpret_cast<v4f *>(&Bs[(mr + 8 * i) * TN_LD + 4 * c4]) = rb[i];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int mm = 2 * s + h;
            const float a0 = As[mm * TN_LD + wm * 64 + r], a1 = As[mm * TN_LD + wm * 64 + 32 + r];
            const float b0 = Bs[mm * TN_LD + wn * 64 + r], b1 = Bs[mm * TN_LD + wn * 64 + 32 + r];
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int k = k0 + wn * 64 + nt * 32 + r;
        if (k >= p.K) continue;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const f32x16 &acc = mt == 0 ? (nt == 0 ? acc00 : acc01) : (nt == 0 ? acc10 : acc11);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (n < p.N) unsafeAtomicAdd(p.C + (int64_t)n * p.ldc + k, acc[e]);
            }
        }
    }
}

int launch_colsum(const float *A, int64_t lda, float *out, int M, int N, hipStream_t s, float *out2);

int launch_gemm_tn(const stair_gemm_tn_args &a, hipStream_t s) {
    STAIR_CHECK(a.M >= 0 && a.N > 0 && a.K > 0 && a.rows_per_group > 0, "bad shape");
    STAIR_CHECK(a.N % 4 == 0 && a.K % 4 == 0 && a.lda % 4 == 0 && a.ldb % 4 == 0 && a.b_gstride % 4 == 0,
                "N, K, lda, ldb, b_gstride must be multiples of 4 floats");
    if (a.M == 0) return 0;
    if (gemm_trace_on())
        fprintf(stderr, "STAIR_GEMM tn M=%d N=%d K=%d act=0 acc=1 gather=%d scale=%d\n", a.M, a.N, a.K, a.b_gidx ? 1 : 0, a.row_scale ? 1 : 0);
    STAIR_CHECK(!a.b_is_bf16 || matmul_mode() == STAIR_MATMUL_BF16X3, "bf16 B rows (stored clip features) need the bf16x3 matmul mode");
    if (a.b_is_bf16) return launch_gemm_tn_bf16x3(a, s);
    if (matmul_mode() != STAIR_MATMUL_F32 && a.M >= kSplitMinRows) return launch_gemm_tn_bf16x3(a, s);
    GemmTnParams p;
    p.A = a.A; p.lda = a.lda; p.B = a.B; p.ldb = a.ldb; p.b_gstride = a.b_gstride; p.b_gidx = a.b_gidx;
    p.R = a.rows_per_group; p.row_scale = a.row_scale; p.rs_gstride = a.rs_gstride; p.rs_gidx = a.rs_gidx;
    p.C = a.C; p.ldc = a.ldc; p.M = a.M; p.N = a.N; p.K = a.K;
    p.tilesN = (a.N + 127) / 128; p.tilesK = (a.K + 127) / 128;
    const int tiles = p.tilesN * p.tilesK;
    // enough slabs to fill the chip (>= ~1024 blocks) but at least 256 rows each; a multiple of the XCD count
    int slabs = std::max(1, std::min((a.M + 255) / 256, (1024 + tiles - 1) / tiles));
    slabs = (slabs + 7) / 8 * 8;
    p.mslab = ((a.M + slabs - 1) / slabs + 31) / 32 * 32;
    hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3(tiles * slabs), dim3(256), 0, s, p);
    STAIR_LAUNCH_CHECK();
    if (a.colsum) return launch_colsum(a.A, a.lda, a.colsum, a.M, a.N, s, a.colsum2);   // fused only in the split kernel
    return 0;
}

// out[n] += sum_m A[m][n]  (bias gradients; out2, if given, receives the same sums: b_ih and b_hh of an LSTM)
__global__ void colsum_kernel(const float *A, int64_t lda, float *out, float *out2, int M, int N, int mslab, long long *out64, long long *out2_64) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int mbeg = blockIdx.y * mslab, mend = min(M, mbeg + mslab);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    const float *p = A + (int64_t)mbeg * lda + n;
    int m = mbeg;
    for (; m + 8 <= mend; m += 8, p += 8 * lda) {      // 8 independent loads in flight per thread
        a0 += p[0]; a1 += p[lda]; a2 += p[2 * lda]; a3 += p[3 * lda];
        a4 += p[4 * lda]; a5 += p[5 * lda]; a6 += p[6 * lda]; a7 += p[7 * lda];
    }
    for (; m < mend; ++m, p += lda) a0 += *p;
    const float acc = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    grad_add(out, out64, n, acc);
    if (out2) grad_add(out2, out2_64, n, acc);
}
int launch_colsum(const float *A, int64_t lda, float *out, int M, int N, hipStream_t s, float *out2) {
    if (M == 0) return 0;
    const int mslab = std::max(64, (M + 511) / 512);
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, (M + mslab - 1) / mslab), dim3(256), 0, s, A, lda, out, out2, M, N, mslab, det_shadow(out), out2 ? det_shadow(out2) : nullptr);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// out[c][r] = in[r][c]   (W^T images for the dX products; 32x32 LDS tile)
__global__ void transpose_kernel(const float *in, float *out, int rows, int cols) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int rr = r0 + i, cc = c0 + threadIdx.x;
        if (rr < rows && cc < cols) t[i][threadIdx.x] = in[(int64_t)rr * cols + cc];
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int cc = c0 + i, rr = r0 + threadIdx.x;
        if (rr < rows && cc < cols) out[(int64_t)cc * rows + rr] = t[threadIdx.x][i];
    }
}
int launch_transpose(const float *in, float *out, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, s, in, out, rows, cols);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// The same for up to 32 matrices in ONE launch (the backward pass transposes its 31 weights every step: 31 launches of ~4.7 us
// were 0.15 ms of a 20 ms step and most of it was launch floor).  The descriptors travel in the kernel argument.
__global__ void transpose_many_kernel(TransposeBatch tb) {
    __shared__ float t[32][33];
    int m = 0;
    while (m + 1 < tb.count && (int)blockIdx.x >= tb.first_tile[m + 1]) ++m;
    const int rows = tb.rows[m], cols = tb.cols[m];
    const int tile = blockIdx.x - tb.first_tile[m], tc = (cols + 31) / 32;
    const int c0 = (tile % tc) * 32, r0 = (tile / tc) * 32;
    const float *in = tb.in[m];
    float *out = tb.out[m];
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int rr = r0 + i, cc = c0 + threadIdx.x;
        if (rr < rows && cc < cols) t[i][threadIdx.x] = in[(int64_t)rr * cols + cc];
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int cc = c0 + i, rr = r0 + threadIdx.x;
        if (rr < rows && cc < cols) out[(int64_t)cc * rows + rr] = t[threadIdx.x][i];
    }
}
int launch_transpose_many(const TransposeBatch &tb, int total_tiles, hipStream_t s) {
    if (tb.count == 0 || total_tiles == 0) return 0;
    hipLaunchKernelGGL(transpose_many_kernel, dim3(total_tiles), dim3(32, 8), 0, s, tb);
    STAIR_LAUNCH_CHECK();
    return 0;
}

}  // namespace stair

extern "C" int stair_gemm_f32(const stair_gemm_args *args, stair_stream stream) {
    if (!args) {
        stair::set_error("stair_gemm_f32: null args");
        return 1;
    }
    return stair::launch_gemm(*args, static_cast<hipStream_t>(stream));
}

extern "C" int stair_gemm_tn_f32(const stair_gemm_tn_args *args, stair_stream stream) {
    if (!args) {
        stair::set_error("stair_gemm_tn_f32: null args");
        return 1;
    }
    return stair::launch_gemm_tn(*args, static_cast<hipStream_t>(stream));
}

extern "C" int stair_set_matmul_mode(int32_t mode) {
    if (mode != STAIR_MATMUL_F32 && mode != STAIR_MATMUL_BF16X3 && mode != STAIR_MATMUL_BF16) {
        stair::set_error("stair_set_matmul_mode: mode must be STAIR_MATMUL_F32, STAIR_MATMUL_BF16X3 or STAIR_MATMUL_BF16");
        return 1;
    }
    stair::set_matmul_mode(mode);
    return 0;
}
extern "C" int stair_get_matmul_mode(void) { return stair::matmul_mode(); }
extern "C" int stair_set_split_min_rows(int32_t rows) {
    if (rows < 1) {
        stair::set_error("stair_set_split_min_rows: rows must be >= 1");
        return 1;
    }
    stair::set_split_min_rows(rows);
    return 0;
}
