// Split-precision ("bf16 x 3") MFMA GEMMs with fp32 inputs and outputs.
//
// gfx950's fp32-input MFMA runs at the vector rate (157 TFLOP/s, 1/16 of the bf16 MFMA rate), and once
// questions are batched the NMN path is bound by exactly those contractions.  Here
// every fp32 operand x is split on the fly into two bf16 numbers, x = hi + lo + O(2^-17 |x|) with
// hi = bf16(x), lo = bf16(x - hi), and each product is evaluated as hi*hi + hi*lo + lo*hi with three
// v_mfma_f32_32x32x16_bf16 accumulating in fp32 (the dropped lo*lo term is O(2^-16) relative).  Relative
// error of a dot product is ~4e-6 (random signs), against ~1e-7 for the exact fp32 kernel -- two orders of
// magnitude inside the 1e-4 logit budget of BASELINE.json -- at 3/16 of the fp32 MFMA's matrix-pipe time.
//
// Structure mirrors csrc/gemm.hip (128x128x32 tile, 4 waves 2x2, LDS double buffer, register prefetch pinned
// above the MFMA phase, group gather / row scale / bias / activation / accumulate epilogue, XCD renumbering),
// with bf16 LDS images [kq][row ^ 2kq][8 bf16] (conflict-free ds_write_b128 staging and ds_read_b128
// fragments).  The TN variant (dW = dZ^T X) transposes 8x4 blocks in registers while staging, so its MFMA
// phase is the same code.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace stair {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using v4f = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using gv4p = const __attribute__((address_space(1))) v4f *;

namespace {

constexpr int XBK = 32;                 // k (or m, for TN) per chunk
constexpr int XKQ = XBK / 8;            // 8-element groups per chunk
constexpr int IMG = XKQ * 128 * 8;      // bf16 elements of one image (8 KB)
// LDS: [buf 2][operand 2][hi/lo 2][IMG] bf16 = 64 KB

// slot = row ^ ((row>>3)&3) ^ 2kq: conflict-free for (a) the NT staging writes (8-lane group = rows {r, r+1} x kq 0..3),
// (b) the TN staging writes (8-lane group = rows c, 4+c, .., 28+c of one kq) and (c) the fragment reads (32 consecutive
// rows of one kq; the XORs permute inside aligned groups of 4 and 8).
__device__ __forceinline__ int img_off(int buf, int operand, int part, int kq, int row) {
    return (((buf * 2 + operand) * 2 + part) * IMG) + (kq * 128 + ((row ^ ((row >> 3) & 3)) ^ (2 * kq))) * 8;
}

__device__ __forceinline__ void split8(const v4f &x0, const v4f &x1, float scale, bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x0[j] * scale, b = x1[j] * scale;
        hi[j] = (__bf16)a;
        hi[4 + j] = (__bf16)b;
        lo[j] = (__bf16)(a - (float)hi[j]);
        lo[4 + j] = (__bf16)(b - (float)hi[4 + j]);
    }
}

__device__ __forceinline__ void split8p(const v4f &x0, const v4f &x1, bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = (__bf16)x0[j];
        hi[4 + j] = (__bf16)x1[j];
        lo[j] = (__bf16)(x0[j] - (float)hi[j]);
        lo[4 + j] = (__bf16)(x1[j] - (float)hi[4 + j]);
    }
}

// one chunk of MFMAs from LDS buffer `buf`: 2 k-steps x (2x2 tiles) x 3 products
template <int NP>
__device__ __forceinline__ void mfma_chunk(const __bf16 *lds, int buf, int wm, int wn, int r, int h, f32x16 &acc00,
                                           f32x16 &acc01, f32x16 &acc10, f32x16 &acc11) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int kq = 2 * s + h;
        const bf16x8 ah0 = *reinterpret_cast<const bf16x8 *>(lds + img_off(buf, 0, 0, kq, wm * 64 + r));
        const bf16x8 ah1 = *reinterpret_cast<const bf16x8 *>(lds + img_off(buf, 0, 0, kq, wm * 64 + 32 + r));
        const bf16x8 bh0 = *reinterpret_cast<const bf16x8 *>(lds + img_off(buf, 1, 0, kq, wn * 64 + r));
        const bf16x8 bh1 = *reinterpret_cast<const bf16x8 *>(lds + img_off(buf, 1, 0, kq, wn * 64 + 32 + r));
        if constexpr (NP >= 2) {                     // the cross terms (NP == 1: plain bf16 products, hi * hi only; NP == 2: B is exact in bf16, no B lo image)
            const bf16x8 al0 = *reinterpret_cast<const bf16x8 *>(lds + img_off(buf, 0, 1, kq, wm * 64 + r));
            const bf16x8 al1 = *reinterpret_cast<const bf16x8 *>(lds + img_off(buf, 0, 1, kq, wm * 64 + 32 + r));
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh0, acc00, 0, 0, 0);      // small terms first
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh1, acc11, 0, 0, 0);
        }
        if constexpr (NP == 3) {
            const bf16x8 bl0 = *reinterpret_cast<const bf16x8 *>(lds + img_off(buf, 1, 1, kq, wn * 64 + r));
            const bf16x8 bl1 = *reinterpret_cast<const bf16x8 *>(lds + img_off(buf, 1, 1, kq, wn * 64 + 32 + r));
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl1, acc11, 0, 0, 0);
        }
        acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh1, acc11, 0, 0, 0);
    }
}

// the eight-wave kernel's chunk: wave tile 64 x 32 = two 32 x 32 tiles sharing the B fragments
template <int NP>
__device__ __forceinline__ void mfma_chunk_w8(const __bf16 *lds, int buf, int wm, int wn, int r, int h, f32x16 &acc00,
                                              f32x16 &acc10) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int kq = 2 * s + h;
        const bf16x8 ah0 = *reinterpret_cast<const bf16x8 *>(lds + img_off(buf, 0, 0, kq, wm * 64 + r));
        const bf16x8 ah1 = *reinterpret_cast<const bf16x8 *>(lds + img_off(buf, 0, 0, kq, wm * 64 + 32 + r));
        const bf16x8 bh0 = *reinterpret_cast<const bf16x8 *>(lds + img_off(buf, 1, 0, kq, wn * 32 + r));
        if constexpr (NP == 3) {
            const bf16x8 al0 = *reinterpret_cast<const bf16x8 *>(lds + img_off(buf, 0, 1, kq, wm * 64 + r));
            const bf16x8 al1 = *reinterpret_cast<const bf16x8 *>(lds + img_off(buf, 0, 1, kq, wm * 64 + 32 + r));
            const bf16x8 bl0 = *reinterpret_cast<const bf16x8 *>(lds + img_off(buf, 1, 1, kq, wn * 32 + r));
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh0, acc00, 0, 0, 0);      // small terms first
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh0, acc10, 0, 0, 0);
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl0, acc00, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl0, acc10, 0, 0, 0);
        }
        acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh0, acc00, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh0, acc10, 0, 0, 0);
    }
}

// four consecutive bf16 values (8 bytes) widened to fp32 (exact)
__device__ __forceinline__ v4f load4_bf16(const float *base, int64_t elem) {
    using v2u = __attribute__((ext_vector_type(2))) unsigned;
    const v2u w = *(const __attribute__((address_space(1))) v2u *)(reinterpret_cast<const char *>(base) + elem * 2);
    v4f r;
    r[0] = __uint_as_float(w[0] << 16); r[1] = __uint_as_float(w[0] & 0xffff0000u);
    r[2] = __uint_as_float(w[1] << 16); r[3] = __uint_as_float(w[1] & 0xffff0000u);
    return r;
}

struct XParams {
    stair_gemm_args a;
    int M, tilesM, tilesN;
    int ksplit = 1, kchunk = 0;      // split-K of small launches (4-wave kernel): blockIdx.y takes K range [y*kchunk, +kchunk)
    float *part = nullptr;           // [ksplit][M][N] partial sums (deterministic reduction); nullptr: atomic adds into C
};

}  // namespace

// ---------------------------------------------------------------------------------------------
// NT: C[m][n] = act(sum_k rs[m] A[m][k] W[n][k] + b[n])
// ---------------------------------------------------------------------------------------------
// PLAIN: K is a multiple of 64 and there is no row scale, so no staged element needs the K-tail mask or the scale
// (a quarter of the staging VALU work otherwise); chunks past K are then staged but never multiplied.
template <int ACT, bool PLAIN, int NP>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(XParams p) {
    extern __shared__ __attribute__((aligned(16))) __bf16 xlds[];
    const stair_gemm_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    const int nb = p.tilesM * p.tilesN;
    const int bid = blockIdx.x;
    const int qd = nb >> 3, rm = nb & 7, xcd = bid & 7, loc = bid >> 3;
    const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    const int tm = logical / p.tilesN, tn = logical - tm * p.tilesN;
    const int m0 = tm * 128, n0 = tn * 128;
    const int R = a.rows_per_group, K = a.K;
    // split-K: this workgroup multiplies k in [kbeg, Kend) and ADDS its raw partial sums (bias / activation are applied by
    // a later pass); Kend also bounds the tail masks below, so an odd trailing chunk of the range is zeroed, not borrowed
    const int kbeg = p.ksplit > 1 ? (int)blockIdx.y * p.kchunk : 0;
    const int Kend = p.ksplit > 1 ? min(K, kbeg + p.kchunk) : K;

    // staging units: (row, kq) with row = u >> 2, kq = u & 3, u = tid + 256 i  ->  rows tid/4 and 64 + tid/4
    const int kq = tid & 3, ra_ = tid >> 2;
    const float *aptr[2];
    const float *wptr[2];
    float rs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = min(m0 + ra_ + 64 * i, p.M - 1);
        const int g = m / R, rr = m - g * R;
        const int64_t gi = a.a_gidx ? a.a_gidx[g] : g;
        aptr[i] = a.A + gi * a.a_gstride + (int64_t)rr * a.lda;
        rs[i] = 1.0f;
        if (a.row_scale) rs[i] = a.row_scale[(a.rs_gidx ? a.rs_gidx[g] : g) * a.rs_gstride + rr];
        const int n = min(n0 + ra_ + 64 * i, a.N - 1);
        wptr[i] = a.W + (int64_t)n * a.ldw;
    }

    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc00[e] = acc01[e] = acc10[e] = acc11[e] = 0.0f;

    // Two staging register sets: the loads of chunks c+1 and c+2 are in flight while chunk c is multiplied
    // (one chunk in flight left the kernel latency bound at ~2 us per chunk against ~0.7 us of MFMA work).
    v4f va[2][2][2], vb[2][2][2];       // [set][row half][k half]
    float km[2][2];
#define X_GLOAD(set, k0)                                                       \
    {                                                                          \
        const int kraw = (k0) + 8 * kq;                                        \
        const int ka = max(0, min(kraw, K - 4)), kb = max(0, min(kraw + 4, K - 4));   \
        km[set][0] = kraw < Kend ? 1.0f : 0.0f;                                \
        km[set][1] = kraw + 4 < Kend ? 1.0f : 0.0f;                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                     \
            va[set][i_][0] = *(gv4p)(aptr[i_] + ka); va[set][i_][1] = *(gv4p)(aptr[i_] + kb);   \
            vb[set][i_][0] = *(gv4p)(wptr[i_] + ka); vb[set][i_][1] = *(gv4p)(wptr[i_] + kb);   \
        }                                                                      \
    }
#define X_LSTORE(set, buf)                                                                              \
    {                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                              \
            bf16x8 hi_, lo_;                                                                            \
            if (PLAIN) split8p(va[set][i_][0], va[set][i_][1], hi_, lo_);                               \
            else split8(va[set][i_][0] * km[set][0], va[set][i_][1] * km[set][1], rs[i_], hi_, lo_);    \
            *reinterpret_cast<bf16x8 *>(xlds + img_off(buf, 0, 0, kq, ra_ + 64 * i_)) = hi_;            \
            if (NP == 3) *reinterpret_cast<bf16x8 *>(xlds + img_off(buf, 0, 1, kq, ra_ + 64 * i_)) = lo_;            \
            split8p(vb[set][i_][0], vb[set][i_][1], hi_, lo_);                                          \
            *reinterpret_cast<bf16x8 *>(xlds + img_off(buf, 1, 0, kq, ra_ + 64 * i_)) = hi_;            \
            if (NP == 3) *reinterpret_cast<bf16x8 *>(xlds + img_off(buf, 1, 1, kq, ra_ + 64 * i_)) = lo_;            \
        }                                                                                               \
    }

    // Chunks past K are not skipped but zeroed (X_GLOAD clamps the address and sets the A mask to 0): every
    // load stays unconditional, which lets hipcc keep counted vmcnt waits (a load inside a branch forces vmcnt(0)
    // at the join and drains the second register set).
    const int nchunks = (Kend - kbeg + XBK - 1) / XBK;
    X_GLOAD(0, kbeg);
    X_GLOAD(1, kbeg + XBK);
    X_LSTORE(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
        // chunk c lives in LDS buffer 0, chunk c+1 in register set 1
        X_GLOAD(0, kbeg + (c + 2) * XBK);
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk<NP>(xlds, 0, wm, wn, r, h, acc00, acc01, acc10, acc11);
        __builtin_amdgcn_sched_barrier(0);
        X_LSTORE(1, 1);
        __syncthreads();
        X_GLOAD(1, kbeg + (c + 3) * XBK);
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk<NP>(xlds, 1, wm, wn, r, h, acc00, acc01, acc10, acc11);
        __builtin_amdgcn_sched_barrier(0);
        X_LSTORE(0, 0);
        __syncthreads();
    }
#undef X_GLOAD
#undef X_LSTORE

    long long *rowoff = reinterpret_cast<long long *>(xlds);
    if (tid < 128) {
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
                if (p.ksplit > 1) {
                    if (p.part) p.part[((int64_t)blockIdx.y * p.M + m0 + rowl) * a.N + n] = acc[e];
                    else unsafeAtomicAdd(a.C + off + n, acc[e]);
                    continue;
                }
                float v = acc[e] + b;
                if (ACT == 1) v = fmaxf(v, 0.0f);
                if (ACT == 2) v = sigmoid_acc(v);
                if (a.accumulate) unsafeAtomicAdd(a.C + off + n, v);
                else Cg[off + n] = v;
            }
        }
    }
}

// Eight-wave form of the same tile for large launches: waves 2 (M) x 4 (N), each 64 x 32 of the 128 x 128 tile,
// so a wave carries 32 accumulator and 16 staging registers per set instead of 64 and 32.  At <= 128 VGPRs two
// workgroups (16 waves, 4 per SIMD) share a CU, twice the four-wave kernel's, which is what hides the staging
// stalls: rocprof counted the MFMA pipe 43 % busy with 2 waves per SIMD (profiles/r01_e_pmc_gemm.json).
template <int ACT, bool PLAIN, int NP>
__global__ __launch_bounds__(512, 2) void gemm_bf16x3_w8_kernel(XParams p) {
    extern __shared__ __attribute__((aligned(16))) __bf16 xlds[];
    const stair_gemm_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;

    const int nb = p.tilesM * p.tilesN;
    const int bid = blockIdx.x;
    const int qd = nb >> 3, rm = nb & 7, xcd = bid & 7, loc = bid >> 3;
    const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    const int tm = logical / p.tilesN, tn = logical - tm * p.tilesN;
    const int m0 = tm * 128, n0 = tn * 128;
    const int R = a.rows_per_group, K = a.K;

    // staging units: (row, kq) with row = tid >> 2 (0..127), kq = tid & 3
    const int kq = tid & 3, ra_ = tid >> 2;
    const float *aptr[1];
    const float *wptr[1];
    float rs[1];
#pragma unroll
    for (int i = 0; i < 1; ++i) {
        const int m = min(m0 + ra_ + 64 * i, p.M - 1);
        const int g = m / R, rr = m - g * R;
        const int64_t gi = a.a_gidx ? a.a_gidx[g] : g;
        aptr[i] = a.A + gi * a.a_gstride + (int64_t)rr * a.lda;
        rs[i] = 1.0f;
        if (a.row_scale) rs[i] = a.row_scale[(a.rs_gidx ? a.rs_gidx[g] : g) * a.rs_gstride + rr];
        const int n = min(n0 + ra_ + 64 * i, a.N - 1);
        wptr[i] = a.W + (int64_t)n * a.ldw;
    }

    f32x16 acc00, acc10;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc00[e] = acc10[e] = 0.0f;

    // Two staging register sets: the loads of chunks c+1 and c+2 are in flight while chunk c is multiplied
    // (one chunk in flight left the kernel latency bound at ~2 us per chunk against ~0.7 us of MFMA work).
    v4f va[2][1][2], vb[2][1][2];       // [set][row][k half]
    float km[2][2];
#define X_GLOAD(set, k0)                                                       \
    {                                                                          \
        const int kraw = (k0) + 8 * kq;                                        \
        const int ka = max(0, min(kraw, K - 4)), kb = max(0, min(kraw + 4, K - 4));   \
        km[set][0] = kraw < K ? 1.0f : 0.0f;                                   \
        km[set][1] = kraw + 4 < K ? 1.0f : 0.0f;                               \
        _Pragma("unroll") for (int i_ = 0; i_ < 1; ++i_) {                     \
            va[set][i_][0] = *(gv4p)(aptr[i_] + ka); va[set][i_][1] = *(gv4p)(aptr[i_] + kb);   \
            vb[set][i_][0] = *(gv4p)(wptr[i_] + ka); vb[set][i_][1] = *(gv4p)(wptr[i_] + kb);   \
        }                                                                      \
    }
#define X_LSTORE(set, buf)                                                                              \
    {                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 1; ++i_) {                                              \
            bf16x8 hi_, lo_;                                                                            \
            if (PLAIN) split8p(va[set][i_][0], va[set][i_][1], hi_, lo_);                               \
            else split8(va[set][i_][0] * km[set][0], va[set][i_][1] * km[set][1], rs[i_], hi_, lo_);    \
            *reinterpret_cast<bf16x8 *>(xlds + img_off(buf, 0, 0, kq, ra_ + 64 * i_)) = hi_;            \
            if (NP == 3) *reinterpret_cast<bf16x8 *>(xlds + img_off(buf, 0, 1, kq, ra_ + 64 * i_)) = lo_;            \
            split8p(vb[set][i_][0], vb[set][i_][1], hi_, lo_);                                          \
            *reinterpret_cast<bf16x8 *>(xlds + img_off(buf, 1, 0, kq, ra_ + 64 * i_)) = hi_;            \
            if (NP == 3) *reinterpret_cast<bf16x8 *>(xlds + img_off(buf, 1, 1, kq, ra_ + 64 * i_)) = lo_;            \
        }                                                                                               \
    }

    // Chunks past K are not skipped but zeroed (X_GLOAD clamps the address and sets the A mask to 0): every
    // load stays unconditional, which lets hipcc keep counted vmcnt waits (a load inside a branch forces vmcnt(0)
    // at the join and drains the second register set).
    const int nchunks = (K + XBK - 1) / XBK;
    X_GLOAD(0, 0);
    X_GLOAD(1, XBK);
    X_LSTORE(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
        // chunk c lives in LDS buffer 0, chunk c+1 in register set 1
        X_GLOAD(0, (c + 2) * XBK);
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk_w8<NP>(xlds, 0, wm, wn, r, h, acc00, acc10);
        __builtin_amdgcn_sched_barrier(0);
        X_LSTORE(1, 1);
        __syncthreads();
        X_GLOAD(1, (c + 3) * XBK);
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk_w8<NP>(xlds, 1, wm, wn, r, h, acc00, acc10);
        __builtin_amdgcn_sched_barrier(0);
        X_LSTORE(0, 0);
        __syncthreads();
    }
#undef X_GLOAD
#undef X_LSTORE

    long long *rowoff = reinterpret_cast<long long *>(xlds);
    if (tid < 128) {
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
    for (int nt = 0; nt < 1; ++nt) {
        const int n = n0 + wn * 32 + r;
        if (n >= a.N) continue;
        const float b = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const f32x16 &acc = mt == 0 ? acc00 : acc10;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rowl = wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const long long off = rowoff[rowl];
                if (off < 0) continue;
                float v = acc[e] + b;
                if (ACT == 1) v = fmaxf(v, 0.0f);
                if (ACT == 2) v = sigmoid_acc(v);
                if (a.accumulate) unsafeAtomicAdd(a.C + off + n, v);
                else Cg[off + n] = v;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// 256 x 256 tile for the large PLAIN launches.  Ablations of the 128 x 128 kernels showed the
// data-movement skeleton -- fp32 rows pulled through the vector L1 into registers, then pushed into LDS by
// ds_write_b128 -- costing more than the MFMAs and NOT overlapping with them.  That cost is bytes per flop: a
// 256 x 256 tile stages half as many bytes per MFMA (and its 128 x 64 wave tile reads half as many LDS fragments per
// MFMA).  8 waves 2 (M) x 4 (N); 128 accumulator registers per lane; ONE staging register set (the 48 MFMAs of a
// chunk, 1536 cycles, cover the load latency on their own); LDS 2 buffers x 64 KB; one workgroup per CU.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int IMG2 = XKQ * 256 * 8;      // bf16 elements of one 256-row image (16 KB)
__device__ __forceinline__ int img2_off(int buf, int operand, int part, int kq, int row) {
    return (((buf * 2 + operand) * 2 + part) * IMG2) + (kq * 256 + ((row ^ ((row >> 3) & 3)) ^ (2 * kq))) * 8;
}
}  // namespace

template <int ACT, int NP>
__global__ __launch_bounds__(512, 1) void gemm_bf16x3_t256_kernel(XParams p) {
    extern __shared__ __attribute__((aligned(16))) __bf16 xlds[];
    const stair_gemm_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;

    const int nb = p.tilesM * p.tilesN;
    const int bid = blockIdx.x;
    const int qd = nb >> 3, rm = nb & 7, xcd = bid & 7, loc = bid >> 3;
    const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    const int tm = logical / p.tilesN, tn = logical - tm * p.tilesN;
    const int m0 = tm * 256, n0 = tn * 256;
    const int R = a.rows_per_group, K = a.K;

    // staging units: (row, kq), rows tid/4 and 128 + tid/4 of each operand
    const int kq = tid & 3, ra_ = tid >> 2;
    const float *aptr[2];
    const float *wptr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = min(m0 + ra_ + 128 * i, p.M - 1);
        const int g = m / R, rr = m - g * R;
        const int64_t gi = a.a_gidx ? a.a_gidx[g] : g;
        aptr[i] = a.A + gi * a.a_gstride + (int64_t)rr * a.lda;
        const int n = min(n0 + ra_ + 128 * i, a.N - 1);
        wptr[i] = a.W + (int64_t)n * a.ldw;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    v4f va[2][2], vb[2][2];       // [row half][k half]
#define Y_GLOAD(k0)                                                            \
    {                                                                          \
        const int ka = min((k0) + 8 * kq, K - 8);                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                     \
            va[i_][0] = *(gv4p)(aptr[i_] + ka); va[i_][1] = *(gv4p)(aptr[i_] + ka + 4);   \
            vb[i_][0] = *(gv4p)(wptr[i_] + ka); vb[i_][1] = *(gv4p)(wptr[i_] + ka + 4);   \
        }                                                                      \
    }
#define Y_LSTORE(buf)                                                                                   \
    {                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                              \
            bf16x8 hi_, lo_;                                                                            \
            split8p(va[i_][0], va[i_][1], hi_, lo_);                                                    \
            *reinterpret_cast<bf16x8 *>(xlds + img2_off(buf, 0, 0, kq, ra_ + 128 * i_)) = hi_;          \
            if (NP == 3) *reinterpret_cast<bf16x8 *>(xlds + img2_off(buf, 0, 1, kq, ra_ + 128 * i_)) = lo_;          \
            split8p(vb[i_][0], vb[i_][1], hi_, lo_);                                                    \
            *reinterpret_cast<bf16x8 *>(xlds + img2_off(buf, 1, 0, kq, ra_ + 128 * i_)) = hi_;          \
            if (NP == 3) *reinterpret_cast<bf16x8 *>(xlds + img2_off(buf, 1, 1, kq, ra_ + 128 * i_)) = lo_;          \
        }                                                                                               \
    }
#define Y_MFMA(buf)                                                                                                  \
    {                                                                                                                \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                           \
            const int kq_ = 2 * s_ + h;                                                                              \
            bf16x8 bh_[2], bl_[2];                                                                                   \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                       \
                bh_[j_] = *reinterpret_cast<const bf16x8 *>(xlds + img2_off(buf, 1, 0, kq_, wn * 64 + 32 * j_ + r)); \
                if (NP == 3) bl_[j_] = *reinterpret_cast<const bf16x8 *>(xlds + img2_off(buf, 1, 1, kq_, wn * 64 + 32 * j_ + r)); \
            }                                                                                                        \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                       \
                const bf16x8 ah_ = *reinterpret_cast<const bf16x8 *>(xlds + img2_off(buf, 0, 0, kq_, wm * 128 + 32 * i_ + r)); \
                bf16x8 al_ = ah_;                                                                                    \
                if (NP == 3) al_ = *reinterpret_cast<const bf16x8 *>(xlds + img2_off(buf, 0, 1, kq_, wm * 128 + 32 * i_ + r)); \
                _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                   \
                    if (NP == 3) {                                                                                   \
                        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_, bh_[j_], acc[i_][j_], 0, 0, 0);   \
                        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bl_[j_], acc[i_][j_], 0, 0, 0);   \
                    }                                                                                                \
                    acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bh_[j_], acc[i_][j_], 0, 0, 0);       \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
    }

    // K % 64 == 0 (PLAIN launches only): chunks come in pairs, no tail; the load past K re-reads the last chunk into
    // a buffer nobody multiplies.  (Running the two waves of a SIMD in opposite phase order -- one multiplying while the
    // other splits and stores -- and raising the MFMA phase's priority were both measured: no change.)
    const int nchunks = K / XBK;
    Y_GLOAD(0);
    Y_LSTORE(0);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
        Y_GLOAD((c + 1) * XBK);
        __builtin_amdgcn_sched_barrier(0);
        Y_MFMA(0);
        __builtin_amdgcn_sched_barrier(0);
        Y_LSTORE(1);
        __syncthreads();
        Y_GLOAD((c + 2) * XBK);
        __builtin_amdgcn_sched_barrier(0);
        Y_MFMA(1);
        __builtin_amdgcn_sched_barrier(0);
        Y_LSTORE(0);
        __syncthreads();
    }
#undef Y_GLOAD
#undef Y_LSTORE
#undef Y_MFMA

    long long *rowoff = reinterpret_cast<long long *>(xlds);
    if (tid < 256) {
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
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + r;
        if (n >= a.N) continue;
        const float b = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rowl = wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const long long off = rowoff[rowl];
                if (off < 0) continue;
                float v = acc[i][j][e] + b;
                if (ACT == 1) v = fmaxf(v, 0.0f);
                if (ACT == 2) v = sigmoid_acc(v);
                if (a.accumulate) unsafeAtomicAdd(a.C + off + n, v);
                else Cg[off + n] = v;
            }
        }
    }
}

// split-K companion: C[row mapping] = act(sum over the splits, in order, of part[y][m][n] + bias[n])
__global__ void c_rows_reduce_kernel(float *C, int64_t ldc, int64_t gs, const int32_t *gidx, int M, int R, int N, const float *part,
                                     int ksplit, const float *bias, int act) {
    const int64_t total = (int64_t)M * N;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(e / N), n = (int)(e - (int64_t)m * N);
        const int g = m / R, rr = m - g * R;
        float v = 0.0f;
        for (int y = 0; y < ksplit; ++y) v += part[(int64_t)y * total + e];
        if (bias) v += bias[n];
        if (act == 1) v = fmaxf(v, 0.0f);
        if (act == 2) v = sigmoid_acc(v);
        C[(gidx ? (int64_t)gidx[g] : (int64_t)g) * gs + (int64_t)rr * ldc + n] = v;
    }
}

static bool gemm_w8_enabled() {
    static const bool on = [] { const char *e = getenv("STAIR_GEMM_W8"); return !(e && e[0] == '0'); }();
    return on;
}

int launch_gemm_bf16x3(const stair_gemm_args &a, hipStream_t s) {
    XParams p;
    p.a = a;
    const int64_t M = (int64_t)a.groups * a.rows_per_group;
    if (M == 0) return 0;
    p.M = (int)M;
    p.tilesM = (p.M + 127) / 128;
    p.tilesN = (a.N + 127) / 128;
    const dim3 grid(p.tilesM * p.tilesN), block(256);
    const size_t shmem = 2 * 2 * 2 * IMG * sizeof(__bf16);
    const bool plain = a.K % 64 == 0 && !a.row_scale;
    const bool one = matmul_mode() == STAIR_MATMUL_BF16;        // single product per operand pair (top-1 identity only)
    static const int t256_min = [] { const char *e = getenv("STAIR_GEMM_T256"); return e ? atoi(e) : 256; }();   // 0 = off
    const int t2m = (p.M + 255) / 256, t2n = (a.N + 255) / 256;
    // one 256 x 256 workgroup per CU: the launch runs in rounds of 256 tiles, so it only pays when the last round is
    // (nearly) full -- 266 tiles took 1.4x the time of the 128 x 128 kernel's 1064, 2048 tiles 0.85x.
    const int t2 = t2m * t2n, rounds = (t2 + 255) / 256;
    if (plain && t256_min > 0 && t2 >= t256_min && a.K >= 64 && (t256_min == 1 || rounds * 256 * 100 <= t2 * 108)) {
        XParams q = p;
        q.tilesM = t2m; q.tilesN = t2n;
        const size_t shmem2 = 2 * 2 * 2 * IMG2 * sizeof(__bf16);       // 128 KB
        const dim3 grid2(t2m * t2n);
        STAIR_ACCT_MFMA("gemm_bf16x3_t256", (M * a.K + (int64_t)a.N * a.K + M * a.N) * 4, 2 * M * a.N * a.K);
        static bool attr_set = false;
        if (!attr_set) {
            const void *fns[6] = {reinterpret_cast<const void *>(&gemm_bf16x3_t256_kernel<0, 3>), reinterpret_cast<const void *>(&gemm_bf16x3_t256_kernel<1, 3>),
                                  reinterpret_cast<const void *>(&gemm_bf16x3_t256_kernel<2, 3>), reinterpret_cast<const void *>(&gemm_bf16x3_t256_kernel<0, 1>),
                                  reinterpret_cast<const void *>(&gemm_bf16x3_t256_kernel<1, 1>), reinterpret_cast<const void *>(&gemm_bf16x3_t256_kernel<2, 1>)};
            for (const void *f : fns) STAIR_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem2));
            attr_set = true;
        }
#define T_LAUNCH(ACT_)                                                                                              \
    if (one) hipLaunchKernelGGL((gemm_bf16x3_t256_kernel<ACT_, 1>), grid2, dim3(512), shmem2, s, q);                 \
    else hipLaunchKernelGGL((gemm_bf16x3_t256_kernel<ACT_, 3>), grid2, dim3(512), shmem2, s, q);
        switch (a.act) {
            case 0: T_LAUNCH(0) break;
            case 1: T_LAUNCH(1) break;
            default: T_LAUNCH(2) break;
        }
#undef T_LAUNCH
        STAIR_LAUNCH_CHECK();
        return 0;
    }
    // Small launches (the vector-level MLPs: 8..64 tiles, K up to 1536) leave most CUs idle and are bound by the latency
    // of their serial K loop, so they are split over K into ~256 workgroups.  Products that already accumulate without
    // bias or activation (the dX products of training) add their partial sums with the atomics they use anyway; forward
    // products write partials to the caller's scratch (stair_gemm_args.splitk_ws) and a second kernel reduces them in
    // a fixed order and applies bias + activation -- results stay deterministic and independent of the batch.
    static const bool splitk_on = [] { const char *e = getenv("STAIR_GEMM_SPLITK"); return !(e && e[0] == '0'); }();
    const int tiles128 = p.tilesM * p.tilesN;
    if (splitk_on && tiles128 <= 64 && a.K >= 256 && a.K <= 128 * 16) {
        const int kchunk = 128;                         // fixed, so the order of the partial sums does not depend on the batch
        const int ksplit = (a.K + kchunk - 1) / kchunk;
        // (K pieces that add into the target with atomics land in any order: only when run-to-run reproducibility is switched off)
        static const bool det = [] { const char *e = getenv("STAIR_DETERMINISTIC"); return !(e && e[0] == '0'); }();
        const bool direct = a.accumulate && a.act == 0 && !a.bias && !det;
        const bool staged = !a.accumulate && a.splitk_ws && a.splitk_ws_floats >= (int64_t)ksplit * M * a.N;
        if (ksplit > 1 && (direct || staged)) {
            p.ksplit = ksplit; p.kchunk = kchunk;
            p.part = staged ? a.splitk_ws : nullptr;
            const dim3 gridk(tiles128, ksplit);
            STAIR_ACCT_MFMA("gemm_bf16x3_splitk", (M * a.K + (int64_t)a.N * a.K + M * a.N) * 4, 2 * M * a.N * a.K);
            if (plain && one) hipLaunchKernelGGL((gemm_bf16x3_kernel<0, true, 1>), gridk, block, shmem, s, p);
            else if (plain) hipLaunchKernelGGL((gemm_bf16x3_kernel<0, true, 3>), gridk, block, shmem, s, p);
            else if (one) hipLaunchKernelGGL((gemm_bf16x3_kernel<0, false, 1>), gridk, block, shmem, s, p);
            else hipLaunchKernelGGL((gemm_bf16x3_kernel<0, false, 3>), gridk, block, shmem, s, p);
            STAIR_LAUNCH_CHECK();
            if (staged) {
                const int zb = (int)std::min<int64_t>((M * a.N + 255) / 256, 1024);
                hipLaunchKernelGGL(c_rows_reduce_kernel, dim3(zb), dim3(256), 0, s, a.C, a.ldc, a.c_gstride, a.c_gidx, p.M,
                                   a.rows_per_group, a.N, a.splitk_ws, ksplit, a.bias, a.act);
                STAIR_LAUNCH_CHECK();
            }
            return 0;
        }
    }
    const bool w8 = gemm_w8_enabled() && p.tilesM * p.tilesN >= 512;   // enough tiles for two 8-wave workgroups on every CU
    STAIR_ACCT_MFMA(w8 ? "gemm_bf16x3_w8" : "gemm_bf16x3", (M * a.K + (int64_t)a.N * a.K + M * a.N) * 4, 2 * M * a.N * a.K);
#define X_LAUNCH1(ACT_, NP_)                                                                                       \
    if (w8 && plain) hipLaunchKernelGGL((gemm_bf16x3_w8_kernel<ACT_, true, NP_>), grid, dim3(512), shmem, s, p);    \
    else if (w8) hipLaunchKernelGGL((gemm_bf16x3_w8_kernel<ACT_, false, NP_>), grid, dim3(512), shmem, s, p);       \
    else if (plain) hipLaunchKernelGGL((gemm_bf16x3_kernel<ACT_, true, NP_>), grid, block, shmem, s, p);            \
    else hipLaunchKernelGGL((gemm_bf16x3_kernel<ACT_, false, NP_>), grid, block, shmem, s, p);
#define X_LAUNCH(ACT_) if (one) { X_LAUNCH1(ACT_, 1) } else { X_LAUNCH1(ACT_, 3) }
    switch (a.act) {
        case 0: X_LAUNCH(0) break;
        case 1: X_LAUNCH(1) break;
        default: X_LAUNCH(2) break;
    }
#undef X_LAUNCH
#undef X_LAUNCH1
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// TN: C[n][k] += sum_m A[m][n] * (rs[m] B[m][k])      (weight gradients)
// waves 0,1 stage A, waves 2,3 stage B: a thread owns one 8(m) x 4(column) block, loads it as 8 float4
// (each wave-load covers 512 contiguous bytes of one row), splits it and writes the four columns as
// [mq][column][8 m] bf16 vectors -- the same LDS image the NT kernel reads.
// ---------------------------------------------------------------------------------------------
struct XTnParams {
    const float *A; int64_t lda;
    const float *B; int64_t ldb, b_gstride; const int32_t *b_gidx; int R;
    const float *row_scale; int64_t rs_gstride; const int32_t *rs_gidx;
    float *C; int64_t ldc;
    float *colsum, *colsum2;
    int M, N, K, mslab, tilesN, tilesK, fast8;
    long long *C64, *colsum64, *colsum2_64;        // fixed-point shadows of C / colsum / colsum2 (det_shadow), or NULL
};

// PLAIN: no row scale and every slab holds a multiple of 64 rows, so no element needs a mask or a scale (columns past
// N or K are computed from clamped addresses and never written).
// BX: B holds bf16 values (stored clip features, exact): rows are read as 8-byte quads, no lo image of B is written and the
// ah * bl product is dropped (NP == 3 -> two products per pair); plain row matrices only (fast8 == 1 or 2).
// RS: a row scale is present (its 8 values per register set are loaded with the rows and carry the row mask); without it the mask of a
// row is recomputed from the chunk index when the set is split, so no scale registers are live across the MFMA phase (with them the
// non-PLAIN kernels spilled 12-32 bytes per lane).
template <bool PLAIN, int NP, bool BX = false, bool RS = true>
__device__ __forceinline__ void gemm_tn_bf16x3_body(const XTnParams &p, const int bid) {
    extern __shared__ __attribute__((aligned(16))) __bf16 xlds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    // One M-slab per XCD: blocks with equal blockIdx % 8 share an XCD (speed only), so all (n,k) tiles of slab
    // 8*(j / tiles) + xcd run on one L2 and walk the slab's rows together: A/B rows leave HBM ~once per slab
    // instead of once per tile (the plain (n,k,slab) grid re-fetched them 8-16x: 16 GB per dW_ih launch).
    const int tiles = p.tilesN * p.tilesK;
    const int xcd = bid & 7, j = bid >> 3;
    const int tile = j % tiles, slab = (j / tiles) * 8 + xcd;
    const int n0 = (tile % p.tilesN) * 128, k0 = (tile / p.tilesN) * 128;
    const int mbeg = min(slab * p.mslab, p.M), mend = min(p.M, mbeg + p.mslab);
    if (mbeg >= mend) return;                      // a slab without rows (the slab count is a multiple of 8 whatever M is): nothing to add

    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc00[e] = acc01[e] = acc10[e] = acc11[e] = 0.0f;

    const bool isB = wave >= 2;                    // wave-uniform role
    const int u = tid & 127;
    const int cq = u & 31, mq = u >> 5;            // column quad (4 columns), m group (8 rows)
    const int col0 = isB ? k0 : n0, ncols = isB ? p.K : p.N;
    const int cc = min(col0 + 4 * cq, ncols - 4);
    const float cmask = col0 + 4 * cq < ncols ? 1.0f : 0.0f;

    v4f v[3][8];              // three register sets: the loads of chunks c+1, c+2, c+3 are in flight while chunk c multiplies
    float sc[RS ? 3 : 1][8];
    // bias gradient riding along: the workgroups of the first K tile already stage every dZ element of their (slab, N tile)
    // once, so waves 0,1 add them up per column while splitting (colsum_kernel re-read all of dZ for this)
    const bool do_colsum = p.colsum != nullptr && !isB && k0 == 0;
    float csum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // A thread's 8 rows (mfirst .. mfirst+7) lie inside ONE group whenever groups are multiples of 8 rows (R = T for
    // [T,H] tiles, or plain matrices passed as one group per 8 rows by the launcher): one group lookup and one 64-bit
    // base per chunk, then constant strides.  Otherwise (vectors gathered one row per group) every row is looked up.
#define T_GLOAD(set, m0_)                                                                                   \
    {                                                                                                       \
        const int mfirst = (m0_) + 8 * mq;                                                                  \
        if (!isB) {                                                                                         \
            const bool full_ = mfirst + 7 < p.M;          /* all 8 rows exist: constant stride from one base */ \
            const float *base_ = p.A + (int64_t)(full_ ? mfirst : 0) * p.lda + cc;                          \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                              \
                if (RS) sc[set][j_] = mfirst + j_ < mend ? cmask : 0.0f;                                    \
                const int64_t ro_ = full_ ? (int64_t)j_ * p.lda : (int64_t)max(0, min(mfirst + j_, p.M - 1)) * p.lda;   \
                v[set][j_] = *(gv4p)(base_ + ro_);                                                          \
            }                                                                                               \
        } else if (p.fast8 == 1) {                                                                          \
            const int mc_ = max(0, min(mfirst, p.M - 8));                                                   \
            const int g_ = mc_ / p.R, rr_ = mc_ - g_ * p.R;                                                 \
            const int64_t boff_ = (p.b_gidx ? (int64_t)p.b_gidx[g_] : (int64_t)g_) * p.b_gstride + (int64_t)rr_ * p.ldb + cc;   \
            const float *rsb_ = p.row_scale ? p.row_scale + (p.rs_gidx ? (int64_t)p.rs_gidx[g_] : (int64_t)g_) * p.rs_gstride + rr_ : nullptr;   \
            const bool in_ = mfirst <= p.M - 8;                                                             \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                              \
                if (RS) sc[set][j_] = (in_ && mfirst + j_ < mend) ? cmask : 0.0f;                           \
                if (RS && rsb_) sc[set][j_] *= rsb_[j_];                                                    \
                if (BX) v[set][j_] = load4_bf16(p.B, boff_ + (int64_t)j_ * p.ldb);                          \
                else v[set][j_] = *(gv4p)(p.B + boff_ + (int64_t)j_ * p.ldb);                               \
            }                                                                                               \
        } else if (p.fast8 == 2) {          /* plain matrix, any M: row m sits at B + m*ldb, no group arithmetic */   \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                              \
                const int mraw_ = mfirst + j_;                                                              \
                const int m_ = max(0, min(mraw_, p.M - 1));                                                 \
                if (RS) sc[set][j_] = mraw_ < mend ? cmask : 0.0f;                                          \
                if (RS && p.row_scale) sc[set][j_] *= p.row_scale[m_];                                      \
                if (BX) v[set][j_] = load4_bf16(p.B, (int64_t)m_ * p.ldb + cc);                             \
                else v[set][j_] = *(gv4p)(p.B + (int64_t)m_ * p.ldb + cc);                                  \
            }                                                                                               \
        } else {                                                                                            \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                              \
                const int mraw_ = mfirst + j_;                                                              \
                const int m_ = max(0, min(mraw_, p.M - 1));                                                 \
                const int g_ = m_ / p.R, rr_ = m_ - g_ * p.R;                                               \
                if (RS) sc[set][j_] = mraw_ < mend ? cmask : 0.0f;                                          \
                if (RS && p.row_scale) sc[set][j_] *= p.row_scale[(p.rs_gidx ? p.rs_gidx[g_] : g_) * p.rs_gstride + rr_];   \
                v[set][j_] = *(gv4p)(p.B + (p.b_gidx ? (int64_t)p.b_gidx[g_] : (int64_t)g_) * p.b_gstride + (int64_t)rr_ * p.ldb + cc);   \
            }                                                                                               \
        }                                                                                                   \
    }
#define T_LSTORE(set, buf, chunk_)                                                                          \
    {                                                                                                       \
        const int operand_ = isB ? 1 : 0;                                                                   \
        const bool sum_ = do_colsum && (chunk_) < nchunks;   /* a chunk past the slab is staged but belongs to nobody */ \
        const int mrow_ = mbeg + (chunk_) * XBK + 8 * mq;    /* first of this thread's 8 rows of the chunk */ \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                                                  \
            bf16x8 hi_, lo_;                                                                                \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                              \
                const float x_ = PLAIN ? v[set][j_][c_] : v[set][j_][c_] * (RS ? sc[RS ? set : 0][j_] : (mrow_ + j_ < mend ? cmask : 0.0f));   \
                if (sum_) csum[c_] += x_;                                                                   \
                hi_[j_] = (__bf16)x_;                                                                       \
                lo_[j_] = (__bf16)(x_ - (float)hi_[j_]);                                                    \
            }                                                                                               \
            *reinterpret_cast<bf16x8 *>(xlds + img_off(buf, operand_, 0, mq, 4 * cq + c_)) = hi_;           \
            if (NP == 3 && !(BX && isB)) *reinterpret_cast<bf16x8 *>(xlds + img_off(buf, operand_, 1, mq, 4 * cq + c_)) = lo_;           \
        }                                                                                                   \
    }

    const int nchunks = (mend - mbeg + XBK - 1) / XBK;
    if (nchunks > 0) {      // block-uniform; rows past mend are zeroed by sc, so every load below is unconditional
        // Chunk c multiplies out of LDS buffer c % 2 while register set (c+1) % 3 (loaded two steps ago) is split into the other
        // buffer and set c % 3 takes the loads of chunk c + 3: at M ~ 17 k rows a workgroup has 17 chunks of ~1 us and a load
        // takes 2-3 us, so with two sets (one step of slack less) every chunk waited for memory -- these launches move
        // 1.3 TB/s, nowhere near a bandwidth limit.
        T_GLOAD(0, mbeg);
        T_GLOAD(1, mbeg + XBK);
        T_GLOAD(2, mbeg + 2 * XBK);
        T_LSTORE(0, 0, 0);
        __syncthreads();
#define T_STEP(cur_, nxt_, buf_, c_)                                                                        \
        if ((c_) < nchunks) {                                                                               \
            T_GLOAD(cur_, mbeg + ((c_) + 3) * XBK);                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            mfma_chunk<(BX && NP == 3) ? 2 : NP>(xlds, buf_, wm, wn, r, h, acc00, acc01, acc10, acc11);     \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            T_LSTORE(nxt_, 1 - (buf_), (c_) + 1);                                                           \
            __syncthreads();                                                                                \
        }
        for (int c = 0; c < nchunks; c += 6) {
            T_STEP(0, 1, 0, c)
            T_STEP(1, 2, 1, c + 1)
            T_STEP(2, 0, 0, c + 2)
            T_STEP(0, 1, 1, c + 3)
            T_STEP(1, 2, 0, c + 4)
            T_STEP(2, 0, 1, c + 5)
        }
#undef T_STEP
    }
#undef T_GLOAD
#undef T_LSTORE
    if (do_colsum) {
#pragma unroll
        for (int c_ = 0; c_ < 4; ++c_) {
            const int n = n0 + 4 * cq + c_;
            if (n < p.N && csum[c_] != 0.0f) {
                grad_add(p.colsum, p.colsum64, n, csum[c_]);
                if (p.colsum2) grad_add(p.colsum2, p.colsum2_64, n, csum[c_]);
            }
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
                if (n < p.N) grad_add(p.C, p.C64, (int64_t)n * p.ldc + k, acc[e]);
            }
        }
    }
}


// (the row-scale forms are built for one workgroup per CU: with two per CU they need 12-32 bytes of scratch per lane, and a kernel that
// spills while two of its workgroups share a CU has returned wrong sums on MI355X -- DESIGN.md "scratch and co-resident workgroups")
template <bool PLAIN, int NP, bool BX = false, bool RS = true>
__global__ __launch_bounds__(256, (RS && !PLAIN) ? 1 : 2) void gemm_tn_bf16x3_kernel(XTnParams p) {
    gemm_tn_bf16x3_body<PLAIN, NP, BX, RS>(p, (int)blockIdx.x);
}
// Several small weight-gradient products in one launch of the same body over a problem table (STAIR_TN_BATCH=0: one launch each).  The body is
// the form without scale registers: the form with them spilled, and a spilling kernel with two workgroups on a CU is what made this launch
// irreproducible when it was first tried ("scratch and co-resident workgroups").
constexpr int XTN_BATCH = 20;
struct XTnBatch { XTnParams p[XTN_BATCH]; int first[XTN_BATCH + 1]; int n; };
static_assert(sizeof(XTnBatch) <= 4096, "kernarg limit");
__global__ __launch_bounds__(256, 2) void gemm_tn_bf16x3_batch_kernel(XTnBatch b) {
    int q = 0;
#pragma unroll
    for (int i = 1; i < XTN_BATCH; ++i) q += (i < b.n && (int)blockIdx.x >= b.first[i]) ? 1 : 0;
    q = __builtin_amdgcn_readfirstlane(q);
    const XTnParams p = b.p[q];
    gemm_tn_bf16x3_body<false, 3, false, false>(p, (int)blockIdx.x - b.first[q]);
}

// 256 x 256 tile form of the TN kernel for the largest weight gradient (dW_ih of the video encoder: N = 1024, K = 2048,
// 32 tiles): PLAIN launches whose B rows come in groups of 8 (fast8 == 1).  Waves 0-3 stage A (256 columns of dZ),
// waves 4-7 stage B; same LDS images and MFMA phase as gemm_bf16x3_t256_kernel; one staging register set.
template <int NP, bool BX = false>
__global__ __launch_bounds__(512, 1) void gemm_tn_bf16x3_t256_kernel(XTnParams p) {
    extern __shared__ __attribute__((aligned(16))) __bf16 xlds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const int tiles = p.tilesN * p.tilesK;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile = jb % tiles, slab = (jb / tiles) * 8 + xcd;
    const int n0 = (tile % p.tilesN) * 256, k0 = (tile / p.tilesN) * 256;
    const int mbeg = min(slab * p.mslab, p.M), mend = min(p.M, mbeg + p.mslab);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const bool isB = wave >= 4;
    const int u = tid & 255;
    const int cq = u & 63, mq = u >> 6;            // column quad (4 columns), m group (8 rows)
    const int col0 = isB ? k0 : n0, ncols = isB ? p.K : p.N;
    const int cc = min(col0 + 4 * cq, ncols - 4);
    const bool do_colsum = p.colsum != nullptr && !isB && k0 == 0;
    float csum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    v4f v[8];
#define Z_GLOAD(m0_)                                                                                        \
    {                                                                                                       \
        const int mc_ = max(0, min((m0_) + 8 * mq, p.M - 8));                                               \
        if (!isB) {                                                                                         \
            const float *base_ = p.A + (int64_t)mc_ * p.lda + cc;                                           \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) v[j_] = *(gv4p)(base_ + j_ * p.lda);           \
        } else {                                                                                            \
            const int g_ = mc_ / p.R, rr_ = mc_ - g_ * p.R;                                                 \
            const int64_t boff_ = (p.b_gidx ? (int64_t)p.b_gidx[g_] : (int64_t)g_) * p.b_gstride + (int64_t)rr_ * p.ldb + cc;   \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                              \
                if (BX) v[j_] = load4_bf16(p.B, boff_ + j_ * p.ldb);                                        \
                else v[j_] = *(gv4p)(p.B + boff_ + j_ * p.ldb);                                             \
            }                                                                                               \
        }                                                                                                   \
    }
#define Z_LSTORE(buf, chunk_)                                                                               \
    {                                                                                                       \
        const int operand_ = isB ? 1 : 0;                                                                   \
        const bool sum_ = do_colsum && (chunk_) < nchunks;                                                  \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                                                  \
            bf16x8 hi_, lo_;                                                                                \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                              \
                const float x_ = v[j_][c_];                                                                 \
                if (sum_) csum[c_] += x_;                                                                   \
                hi_[j_] = (__bf16)x_;                                                                       \
                lo_[j_] = (__bf16)(x_ - (float)hi_[j_]);                                                    \
            }                                                                                               \
            *reinterpret_cast<bf16x8 *>(xlds + img2_off(buf, operand_, 0, mq, 4 * cq + c_)) = hi_;          \
            if (NP == 3 && !(BX && isB)) *reinterpret_cast<bf16x8 *>(xlds + img2_off(buf, operand_, 1, mq, 4 * cq + c_)) = lo_;          \
        }                                                                                                   \
    }
#define Z_MFMA(buf)                                                                                                  \
    {                                                                                                                \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                           \
            const int kq_ = 2 * s_ + h;                                                                              \
            bf16x8 bh_[2], bl_[2];                                                                                   \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                       \
                bh_[j_] = *reinterpret_cast<const bf16x8 *>(xlds + img2_off(buf, 1, 0, kq_, wn * 64 + 32 * j_ + r)); \
                if (NP == 3 && !BX) bl_[j_] = *reinterpret_cast<const bf16x8 *>(xlds + img2_off(buf, 1, 1, kq_, wn * 64 + 32 * j_ + r)); \
            }                                                                                                        \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                       \
                const bf16x8 ah_ = *reinterpret_cast<const bf16x8 *>(xlds + img2_off(buf, 0, 0, kq_, wm * 128 + 32 * i_ + r)); \
                bf16x8 al_ = ah_;                                                                                    \
                if (NP == 3) al_ = *reinterpret_cast<const bf16x8 *>(xlds + img2_off(buf, 0, 1, kq_, wm * 128 + 32 * i_ + r)); \
                _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                   \
                    if (NP == 3) {                                                                                   \
                        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_, bh_[j_], acc[i_][j_], 0, 0, 0);   \
                        if (!BX) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bl_[j_], acc[i_][j_], 0, 0, 0);   \
                    }                                                                                                \
                    acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bh_[j_], acc[i_][j_], 0, 0, 0);       \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
    }
    const int nchunks = (mend - mbeg) / XBK;           // even: every slab holds a multiple of 64 rows (PLAIN)
    if (nchunks > 0) {
        Z_GLOAD(mbeg);
        Z_LSTORE(0, 0);
        __syncthreads();
        for (int c = 0; c < nchunks; c += 2) {
            Z_GLOAD(mbeg + (c + 1) * XBK);
            __builtin_amdgcn_sched_barrier(0);
            Z_MFMA(0);
            __builtin_amdgcn_sched_barrier(0);
            Z_LSTORE(1, c + 1);
            __syncthreads();
            Z_GLOAD(mbeg + (c + 2) * XBK);
            __builtin_amdgcn_sched_barrier(0);
            Z_MFMA(1);
            __builtin_amdgcn_sched_barrier(0);
            Z_LSTORE(0, c + 2);
            __syncthreads();
        }
    }
#undef Z_GLOAD
#undef Z_LSTORE
#undef Z_MFMA
    if (do_colsum) {
#pragma unroll
        for (int c_ = 0; c_ < 4; ++c_) {
            const int n = n0 + 4 * cq + c_;
            if (n < p.N && csum[c_] != 0.0f) {
                grad_add(p.colsum, p.colsum64, n, csum[c_]);
                if (p.colsum2) grad_add(p.colsum2, p.colsum2_64, n, csum[c_]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k0 + wn * 64 + j * 32 + r;
        if (k >= p.K) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (n < p.N) grad_add(p.C, p.C64, (int64_t)n * p.ldc + k, acc[i][j][e]);
            }
        }
    }
}

int launch_gemm_tn_bf16x3(const stair_gemm_tn_args &a, hipStream_t s) {
    if (a.M == 0) return 0;
    const bool one = matmul_mode() == STAIR_MATMUL_BF16;
    XTnParams p;
    p.A = a.A; p.lda = a.lda; p.B = a.B; p.ldb = a.ldb; p.b_gstride = a.b_gstride; p.b_gidx = a.b_gidx;
    p.R = a.rows_per_group; p.row_scale = a.row_scale; p.rs_gstride = a.rs_gstride; p.rs_gidx = a.rs_gidx;
    p.C = a.C; p.ldc = a.ldc; p.M = a.M; p.N = a.N; p.K = a.K;
    p.colsum = a.colsum; p.colsum2 = a.colsum2;
    p.C64 = det_shadow(a.C); p.colsum64 = a.colsum ? det_shadow(a.colsum) : nullptr; p.colsum2_64 = a.colsum2 ? det_shadow(a.colsum2) : nullptr;
    p.tilesN = (a.N + 127) / 128; p.tilesK = (a.K + 127) / 128;
    p.fast8 = 0;
    const bool bx = a.b_is_bf16 != 0;
    if (bx) {
        STAIR_CHECK(!one, "bf16 B rows need the bf16x3 matmul mode");
        STAIR_CHECK(a.rows_per_group == 1 && !a.b_gidx && !a.row_scale && a.b_gstride == a.ldb && a.K % 8 == 0 && a.ldb % 4 == 0,
                    "bf16 B rows: plain row matrix only (rows_per_group 1, no index, no row scale), K % 8 == 0");
        STAIR_CHECK((reinterpret_cast<uintptr_t>(a.B) & 7) == 0, "bf16 B rows must be 8-byte aligned");
    }
    const bool plain_matrix = !p.b_gidx && !p.rs_gidx && p.b_gstride == (int64_t)p.R * p.ldb && (!p.row_scale || p.rs_gstride == p.R);
    if (a.M % 8 == 0 && a.M >= 8) {
        if (p.R % 8 == 0) p.fast8 = 1;
        else if (plain_matrix) {
            // a plain contiguous matrix: regroup it as groups of 8 rows
            p.R = 8; p.b_gstride = 8 * p.ldb; p.rs_gstride = 8; p.fast8 = 1;
        }
    } else if (plain_matrix) {
        p.fast8 = 2;                  // ragged row count (e.g. the text encoder's sum of question lengths)
    }
    if (bx) {                     // the long dW_ih shapes: row-major LDS tiles read transposed (csrc/gemm_tn_tr.hip)
        const int rc = launch_gemm_tn_tr(a, s);
        if (rc >= 0) return rc;
    }
    {   // 256 x 256 tiles where the output is large enough (>= 16 of them) and the staging is the simple case
        static const int tn256 = [] { const char *e = getenv("STAIR_GEMM_TN256"); return e ? atoi(e) : 16; }();   // 0 = off
        const int t2n = (a.N + 255) / 256, t2k = (a.K + 255) / 256, t2 = t2n * t2k;
        // (a short reduction dimension gives each of the 512 workgroups a handful of chunks and a 256 x 256 atomic epilogue:
        // the decoder's dW at M = 2048 took 122 us here against ~40 us on the 128 x 128 kernel, so M must be long)
        if (tn256 > 0 && t2 >= tn256 && p.fast8 == 1 && !p.row_scale && a.M % 64 == 0 && (a.M >= 16384 || tn256 == 1)) {
            XTnParams q = p;
            q.tilesN = t2n; q.tilesK = t2k;
            int slabs = std::max(1, std::min(a.M / 64, (512 + t2 - 1) / t2));
            slabs = (slabs + 7) / 8 * 8;
            q.mslab = ((a.M + slabs - 1) / slabs + 63) / 64 * 64;
            const size_t shmem2 = 2 * 2 * 2 * IMG2 * sizeof(__bf16);
            STAIR_ACCT_MFMA("gemm_tn_bf16x3_t256", ((int64_t)a.M * a.N * 4 + (int64_t)a.M * a.K * (bx ? 2 : 4) + (int64_t)a.N * a.K * 4), 2ll * a.M * a.N * a.K);
            static bool attr_set = false;
            if (!attr_set) {
                STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_bf16x3_t256_kernel<3>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem2));
                STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_bf16x3_t256_kernel<1>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem2));
                STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_bf16x3_t256_kernel<3, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem2));
                attr_set = true;
            }
            if (bx) hipLaunchKernelGGL((gemm_tn_bf16x3_t256_kernel<3, true>), dim3(t2 * slabs), dim3(512), shmem2, s, q);
            else if (one) hipLaunchKernelGGL(gemm_tn_bf16x3_t256_kernel<1>, dim3(t2 * slabs), dim3(512), shmem2, s, q);
            else hipLaunchKernelGGL(gemm_tn_bf16x3_t256_kernel<3>, dim3(t2 * slabs), dim3(512), shmem2, s, q);
            STAIR_LAUNCH_CHECK();
            return 0;
        }
    }
    const int tiles = p.tilesN * p.tilesK;
    // M is split into slabs so that ~512 workgroups (2 per CU) exist; more slabs only add fp32 atomics (each slab adds
    // its whole N x K tile set: measured 172 -> 205 TFLOP/s at M=32768, N=K=512 going from 1024 to 512 workgroups)
    static const int target = [] { const char *e = getenv("STAIR_TN_BLOCKS"); return e ? std::max(8, atoi(e)) : 512; }();
    static const int minrows = [] { const char *e = getenv("STAIR_TN_MINROWS"); return e ? std::max(32, atoi(e)) : 256; }();
    int slabs = std::max(1, std::min((a.M + minrows - 1) / minrows, (target + tiles - 1) / tiles));
    slabs = (slabs + 7) / 8 * 8;                                   // a multiple of the XCD count
    p.mslab = ((a.M + slabs - 1) / slabs + 63) / 64 * 64;
    const size_t shmem = 2 * 2 * 2 * IMG * sizeof(__bf16);
    if (a.M <= p.mslab) p.C64 = nullptr;           // ONE slab holds every row: one add per element, the float atomic is reproducible as it is
                                                   // (the bias sums still meet from several threads of a workgroup: they keep their shadow)
    const bool plain = !p.row_scale && a.M % 64 == 0 && p.mslab % 64 == 0;
    STAIR_ACCT_MFMA("gemm_tn_bf16x3", ((int64_t)a.M * a.N * 4 + (int64_t)a.M * a.K * (bx ? 2 : 4) + (int64_t)a.N * a.K * 4), 2ll * a.M * a.N * a.K);
    // non-PLAIN launches without a row scale take the kernels that keep no scale registers (RS = false)
    static const bool rs_always = [] { const char *e = getenv("STAIR_TN_RS_ALWAYS"); return e && e[0] == '1'; }();   // diagnostics
    static const size_t lds_force = [] { const char *e = getenv("STAIR_TN_LDS"); return e ? (size_t)atol(e) : (size_t)0; }();   // diagnostics: 98304 = one workgroup per CU
    const bool rs = p.row_scale != nullptr || rs_always;
    size_t lds = shmem;
    if (lds_force > shmem) {
        lds = lds_force;
        static bool attr_set = false;
        if (!attr_set) {
#define TN_ATTR(...) STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_bf16x3_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
            TN_ATTR(true, 3, true); TN_ATTR(false, 3, true, false); TN_ATTR(false, 3, true, true); TN_ATTR(true, 1); TN_ATTR(true, 3);
            TN_ATTR(false, 1, false, true); TN_ATTR(false, 1, false, false); TN_ATTR(false, 3, false, true); TN_ATTR(false, 3, false, false);
#undef TN_ATTR
            attr_set = true;
        }
    }
    const dim3 grid(tiles * slabs), block(256);
    if (bx) {
        STAIR_CHECK(p.fast8 != 0, "internal: bf16 B rows need a plain row matrix");
        if (plain) hipLaunchKernelGGL((gemm_tn_bf16x3_kernel<true, 3, true>), grid, block, lds, s, p);
        else if (rs) hipLaunchKernelGGL((gemm_tn_bf16x3_kernel<false, 3, true, true>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((gemm_tn_bf16x3_kernel<false, 3, true, false>), grid, block, lds, s, p);
    } else if (plain && one) hipLaunchKernelGGL((gemm_tn_bf16x3_kernel<true, 1>), grid, block, lds, s, p);
    else if (plain) hipLaunchKernelGGL((gemm_tn_bf16x3_kernel<true, 3>), grid, block, lds, s, p);
    else if (one && rs) hipLaunchKernelGGL((gemm_tn_bf16x3_kernel<false, 1, false, true>), grid, block, lds, s, p);
    else if (one) hipLaunchKernelGGL((gemm_tn_bf16x3_kernel<false, 1, false, false>), grid, block, lds, s, p);
    else if (rs) hipLaunchKernelGGL((gemm_tn_bf16x3_kernel<false, 3, false, true>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((gemm_tn_bf16x3_kernel<false, 3, false, false>), grid, block, lds, s, p);
    STAIR_LAUNCH_CHECK();
    return 0;
}

int launch_gemm_tn_batch(const stair_gemm_tn_args *a, int n, hipStream_t s) {
    static const bool on = [] { const char *e = getenv("STAIR_TN_BATCH"); return !(e && e[0] == '0'); }();
    bool ok = on && n >= 2 && matmul_mode() == STAIR_MATMUL_BF16X3;
    for (int i = 0; i < n && ok; ++i)
        ok = a[i].M > 0 && a[i].M < 16384 && !a[i].b_is_bf16 && !a[i].row_scale && a[i].N % 4 == 0 && a[i].K % 4 == 0 && a[i].lda % 4 == 0 && a[i].ldb % 4 == 0 &&
             a[i].b_gstride % 4 == 0 && a[i].rows_per_group > 0;
    if (!ok) {
        for (int i = 0; i < n; ++i)
            if (a[i].M > 0)
                if (int rc = launch_gemm_tn(a[i], s)) return rc;
        return 0;
    }
    static const int lds_req = [] { const char *e = getenv("STAIR_TN_BATCH_LDS"); return e ? atoi(e) : 0; }();
    const size_t shmem = lds_req > 0 ? (size_t)lds_req : 2 * 2 * 2 * IMG * sizeof(__bf16);
    if (lds_req > 65536) {
        static bool set_ = false;
        if (!set_) { STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_bf16x3_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_req)); set_ = true; }
    }
    static const int bmax = [] { const char *e = getenv("STAIR_TN_BATCH_MAX"); return e ? std::max(1, std::min(XTN_BATCH, atoi(e))) : XTN_BATCH; }();
    for (int i0 = 0; i0 < n; i0 += bmax) {
        XTnBatch b;
        b.n = std::min(bmax, n - i0);
        b.first[0] = 0;
        for (int i = 0; i < XTN_BATCH; ++i) {
            if (i >= b.n) { b.p[i] = b.p[0]; b.first[i + 1] = b.first[b.n]; continue; }
            const stair_gemm_tn_args &g = a[i0 + i];
            XTnParams &p = b.p[i];
            p.A = g.A; p.lda = g.lda; p.B = g.B; p.ldb = g.ldb; p.b_gstride = g.b_gstride; p.b_gidx = g.b_gidx;
            p.R = g.rows_per_group; p.row_scale = g.row_scale; p.rs_gstride = g.rs_gstride; p.rs_gidx = g.rs_gidx;
            p.C = g.C; p.ldc = g.ldc; p.M = g.M; p.N = g.N; p.K = g.K;
            p.colsum = g.colsum; p.colsum2 = g.colsum2;
            p.C64 = det_shadow(g.C); p.colsum64 = g.colsum ? det_shadow(g.colsum) : nullptr; p.colsum2_64 = g.colsum2 ? det_shadow(g.colsum2) : nullptr;
            p.tilesN = (g.N + 127) / 128; p.tilesK = (g.K + 127) / 128;
            p.fast8 = 0;
            const bool plain_matrix = !p.b_gidx && !p.rs_gidx && p.b_gstride == (int64_t)p.R * p.ldb && (!p.row_scale || p.rs_gstride == p.R);
            if (g.M % 8 == 0 && g.M >= 8) {
                if (p.R % 8 == 0) p.fast8 = 1;
                else if (plain_matrix) { p.R = 8; p.b_gstride = 8 * p.ldb; p.rs_gstride = 8; p.fast8 = 1; }
            } else if (plain_matrix) p.fast8 = 2;
            const int tiles = p.tilesN * p.tilesK;
            int slabs = std::max(1, std::min((g.M + 255) / 256, (512 + tiles - 1) / tiles));
            slabs = (slabs + 7) / 8 * 8;
            p.mslab = ((g.M + slabs - 1) / slabs + 63) / 64 * 64;
            if (g.M <= p.mslab) p.C64 = nullptr;
            b.first[i + 1] = b.first[i] + tiles * slabs;
        }
        hipLaunchKernelGGL(gemm_tn_bf16x3_batch_kernel, dim3(b.first[b.n]), dim3(256), shmem, s, b);
        STAIR_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace stair
