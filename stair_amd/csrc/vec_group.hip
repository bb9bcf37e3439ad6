// Grouped vector-level products: every small row-wise Linear of a program level in ONE launch.
//
// The vector-level modules of /root/reference/video_nmn/modules.py -- Compare (:15-21), Equals (:24-37), Xor (:59-72), ToAction
// (:102-120), Exists (:141-159) -- Filter's dense layer on the pooled rows (:376-378), Localize's keyword projection (:199-203) and
// the decoder (module_net.py:49-53, 136-138) are Linear layers on ONE [H] row per instance.  A level of a 128-question batch holds a
// handful of such products of 13 .. 130 rows each; as separate launches (pack -> split-K GEMM -> reduction, per module) a training
// step spent ~150 of its ~290 launches on them.  Here a launch carries a list of PROBLEMS; its work items are (problem, 64-row tile,
// 64-column block) and a workgroup computes its [64 x 64] output block over the FULL reduction length (<= 1536), so there is no
// split-K scratch and no reduction launch:
//   * 8 waves = 2 row halves x 4 quarters of the reduction dimension; operands go global -> registers -> MFMA directly (the A / B
//     fragment of v_mfma_f32_32x32x16_bf16 for lane (r, h) is 8 consecutive floats of row r: both the instance rows and the
//     row-major weight rows have that shape), split into bf16 hi / lo on the way (three products, fp32 accumulate: the arithmetic
//     of csrc/gemm_bf16x3.hip);
//   * the concatenated inputs of the modules ([a, b], [|a - b|, a, b], [a, b, a * b]) are formed in registers from the two operand
//     rows -- never materialised for the product; a training plan keeps them (in_save) as the weight-gradient operand;
//   * the four quarters meet in LDS (fixed order: deterministic, and a row's result does not depend on the other rows of the
//     launch); bias / ReLU / relu' mask, then 256-byte row segments go out: plain stores, or float atomics (gradient rows that
//     several instances share);
//   * backward (kind ADJ): dX = dZ W through the transposed weight image, all 2 - 3 H-wide blocks of dX for one 64-column slice in
//     one work item, so that the adjoint of the concatenation (CAT2 / EXISTS / XOR, rowops_bwd.hip pack_bwd_kernel) is applied in
//     the epilogue and added straight into the operands' gradient rows; the relu' mask of the incoming gradient is applied on load
//     (IN_MASK) and the masked rows are kept (in_save) as the dZ operand of the weight-gradient product.
#include <algorithm>
#include <vector>

#include "common.h"
#include "ops.h"

namespace stair {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using v4f = __attribute__((ext_vector_type(4))) float;

namespace {

constexpr int VG_SEG = 512;                  // width of one input segment / one adjoint output block (the reference's hidden size)
constexpr int VG_PLD = 68;                   // row stride (floats) of the partial tiles in LDS
constexpr int VG_LDS = 4 * 64 * VG_PLD * 4;  // four reduction quarters x [64 x 64] partial tile
constexpr int VG_MAXP = 12;                  // problems per launch (kernel-argument block < 4 KB)

struct VgParams {
    VgProblem p[VG_MAXP];
    int first[VG_MAXP + 1];                  // work items of problem i: first[i] .. first[i + 1] - 1
    int np;
};
static_assert(sizeof(VgParams) <= 4096, "the argument block of a launch must stay under the 4 KB kernarg limit");

__device__ __forceinline__ void vg_split8(const v4f a, const v4f b, bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = (__bf16)a[j]; lo[j] = (__bf16)(a[j] - (float)hi[j]);
        hi[4 + j] = (__bf16)b[j]; lo[4 + j] = (__bf16)(b[j] - (float)hi[4 + j]);
    }
}

// 8 consecutive floats at p[k .. k + 7], zero beyond `lim` (the reduction tail of a row shorter than a whole step)
__device__ __forceinline__ void vg_load8(const float *p, int k, int lim, v4f &x0, v4f &x1) {
    if (k + 8 <= lim) {
        x0 = *reinterpret_cast<const v4f *>(p + k);
        x1 = *reinterpret_cast<const v4f *>(p + k + 4);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { x0[j] = k + j < lim ? p[k + j] : 0.f; x1[j] = k + 4 + j < lim ? p[k + 4 + j] : 0.f; }
    }
}

// input segment `s` of the row from its two operand pieces (8 floats each)
__device__ __forceinline__ void vg_form(int pack, int s, float in_scale, const v4f a0, const v4f a1, const v4f b0, const v4f b1, v4f &x0, v4f &x1) {
    switch (pack) {
        case VG_IN_A: x0 = a0; x1 = a1; break;
        case VG_IN_CAT2: x0 = s == 0 ? a0 : b0; x1 = s == 0 ? a1 : b1; break;
        case VG_IN_XOR:
            if (s == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { x0[e] = fabsf(a0[e] - b0[e]); x1[e] = fabsf(a1[e] - b1[e]); }
            } else { x0 = s == 1 ? a0 : b0; x1 = s == 1 ? a1 : b1; }
            break;
        case VG_IN_EXISTS:
            if (s == 2) { x0 = a0 * b0; x1 = a1 * b1; }
            else { x0 = s == 0 ? a0 : b0; x1 = s == 0 ? a1 : b1; }
            break;
        default:        // VG_IN_MASK: the incoming gradient times relu'(forward output) (x in_scale)
#pragma unroll
            for (int e = 0; e < 4; ++e) { x0[e] = b0[e] > 0.f ? a0[e] * in_scale : 0.f; x1[e] = b1[e] > 0.f ? a1[e] * in_scale : 0.f; }
            break;
    }
}

// One [64 x 64] output block of a work item: this wave's quarter of the reduction for its 32 rows x 64 columns.
// Micro-step m = (16-wide step ks, input segment s): the weight pieces of micro-step m + 1 and the row pieces of step ks + 1 are in
// flight while m's MFMAs run (two register buffers each; the loops are unrolled so that the buffers are registers).
template <int NIN>
__device__ __forceinline__ void vg_tile(const VgProblem &p, const float *ap, const float *bp, const float *w0, const float *w1, const int kq,
                                        const int h, float *save_row, f32x16 (&acc)[2]) {
    const int kred = p.kred;
    const bool two = p.pack != VG_IN_A;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
    const int nsteps = min(8, max(0, (kred - 128 * kq + 15) / 16));       // kred = 512: 8 steps per quarter
    if (nsteps == 0) return;
    v4f xa[2][2], xb[2][2], wv[2][2][2];
    auto load_x = [&](const int buf, const int ks) {
        const int k = 128 * kq + 16 * ks + 8 * h;
        vg_load8(ap, k, kred, xa[buf][0], xa[buf][1]);
        if (two) vg_load8(bp, k, kred, xb[buf][0], xb[buf][1]);
    };
    auto load_w = [&](const int buf, const int ks, const int s) {
        const int k = 128 * kq + 16 * ks + 8 * h;
        vg_load8(w0 + s * VG_SEG, k, kred, wv[buf][0][0], wv[buf][0][1]);
        vg_load8(w1 + s * VG_SEG, k, kred, wv[buf][1][0], wv[buf][1][1]);
    };
    load_x(0, 0);
    load_w(0, 0, 0);
#pragma unroll
    for (int m = 0; m < 8 * NIN; ++m) {
        const int ks = m / NIN, s = m - ks * NIN;
        if (ks >= nsteps) break;
        if (s == 0 && ks + 1 < nsteps) load_x((ks + 1) & 1, ks + 1);
        if (s + 1 < NIN) load_w((m + 1) & 1, ks, s + 1);
        else if (ks + 1 < nsteps) load_w((m + 1) & 1, ks + 1, 0);
        __builtin_amdgcn_sched_barrier(0);          // the prefetches stay above this micro-step's MFMAs
        v4f x0, x1;
        vg_form(p.pack, s, p.in_scale, xa[ks & 1][0], xa[ks & 1][1], xb[ks & 1][0], xb[ks & 1][1], x0, x1);
        const int k = 128 * kq + 16 * ks + 8 * h;
        if (save_row && k + 8 <= kred) {
            float *d = save_row + s * VG_SEG + k;
            *reinterpret_cast<v4f *>(d) = x0; *reinterpret_cast<v4f *>(d + 4) = x1;
        }
        bf16x8 xh, xl;
        vg_split8(x0, x1, xh, xl);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            bf16x8 wh, wl;
            vg_split8(wv[m & 1][nt][0], wv[m & 1][nt][1], wh, wl);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc[nt], 0, 0, 0);
        }
    }
}

// One work item: rows [64 rt, 64 rt + 64) x output columns of block cb.  NIN input segments; NOUT output blocks (1: a forward-shaped
// product; 2 / 3: the H-wide blocks of a concatenation's gradient at the same 64 columns, combined by the adjoint epilogue).
template <int NIN, int NOUT>
__device__ __forceinline__ void vg_item(const VgProblem &p, const int rt, const int cb, float *P) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tt = wave >> 2, kq = wave & 3;
    const int rows = p.rows;
    const int row = rt * 64 + 32 * tt + r, rowc = min(row, rows - 1);
    const float *ap = p.a + (int64_t)(p.ia ? p.ia[rowc] : rowc) * p.lda;
    const float *bp = p.pack != VG_IN_A ? p.b + (int64_t)(p.ib ? p.ib[rowc] : rowc) * p.ldb : ap;
    float *save_row = p.in_save != nullptr && cb == 0 && row < rows ? p.in_save + (int64_t)row * p.ld_save : nullptr;

    // ---- per output block: the product, then the four quarters meet in LDS; thread (wave, lane) owns column `lane` of rows
    //      wave, wave + 8, ... of the block ----
    float d[NOUT][8];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        const int n0 = (p.kind == VG_ADJ ? j * VG_SEG : 0) + cb * 64 + r;
        f32x16 acc[2];
        vg_tile<NIN>(p, ap, bp, p.W + (int64_t)min(n0, p.N - 1) * p.ldw, p.W + (int64_t)min(n0 + 32, p.N - 1) * p.ldw, kq, h,
                     j == 0 ? save_row : nullptr, acc);
        if (j > 0) __syncthreads();               // the previous block's partials have been read
        // lane (r, h) holds row t = 32 tt + r, columns 32 nt + 8 q + 4 h + i for e = 4 q + i
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4f z;
#pragma unroll
                for (int i = 0; i < 4; ++i) z[i] = acc[nt][4 * q + i];
                *reinterpret_cast<v4f *>(P + (kq * 64 + 32 * tt + r) * VG_PLD + 32 * nt + 8 * q + 4 * h) = z;
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = wave + 8 * i;
            d[j][i] = ((P[(0 * 64 + t) * VG_PLD + lane] + P[(1 * 64 + t) * VG_PLD + lane]) + P[(2 * 64 + t) * VG_PLD + lane]) + P[(3 * 64 + t) * VG_PLD + lane];
        }
    }

    const int col = cb * 64 + lane;
    if (p.kind == VG_FWD) {
        if (col >= p.N) return;
        const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rw = rt * 64 + wave + 8 * i;
            if (rw >= rows) break;
            float v = d[0][i] + bias;
            if (p.act == 1) v = fmaxf(v, 0.f);
            else if (p.act == 2) v = p.emask[(int64_t)rw * p.ldm + col] > 0.f ? v * p.escale : 0.f;
            float *dst = p.out + (int64_t)(p.io ? p.io[rw] : rw) * p.ldo + col;
            if (p.accumulate) unsafeAtomicAdd(dst, v);
            else *dst = v;
        }
    } else if (NOUT > 1) {
        // adjoint of the concatenation: d[0], d[1] (, d[2]) are the gradient blocks at column `col` of the H-wide operand rows
        constexpr int J1 = NOUT > 1 ? 1 : 0, J2 = NOUT > 2 ? 2 : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rw = rt * 64 + wave + 8 * i;
            if (rw >= rows) break;
            const int64_t ra = (int64_t)(p.fia ? p.fia[rw] : rw) * p.ldfa + col, rb = (int64_t)(p.fib ? p.fib[rw] : rw) * p.ldfb + col;
            float da, db;
            if (NOUT == 2) { da = d[0][i]; db = d[J1][i]; }
            else if (p.adj == VG_IN_EXISTS) {       // [a, b, a * b]
                const float a = p.fa[ra], b = p.fb[rb];
                da = d[0][i] + d[J2][i] * b; db = d[J1][i] + d[J2][i] * a;
            } else {                                // [|a - b|, a, b]
                const float df = p.fa[ra] - p.fb[rb];
                const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
                da = d[0][i] * sg + d[J1][i]; db = -d[0][i] * sg + d[J2][i];
            }
            unsafeAtomicAdd(p.ga + ra, da);
            unsafeAtomicAdd(p.gb + rb, db);
        }
    }
}

__global__ __launch_bounds__(512, 1) void vec_group_kernel(VgParams pp) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float *P = reinterpret_cast<float *>(lds);
    const int w = blockIdx.x;
    int sel = 0;
#pragma unroll
    for (int j = 1; j < VG_MAXP; ++j) sel += (j < pp.np && w >= pp.first[j]) ? 1 : 0;
    const VgProblem &p = pp.p[sel];
    const int item = w - pp.first[sel];
    const int nblk = p.kind == VG_ADJ ? VG_SEG / 64 : (p.N + 63) / 64;
    const int rt = item / nblk, cb = item - rt * nblk;
    const int nin = p.pack == VG_IN_A || p.pack == VG_IN_MASK ? 1 : (p.pack == VG_IN_CAT2 ? 2 : 3);
    if (p.kind == VG_FWD) {
        if (nin == 1) vg_item<1, 1>(p, rt, cb, P);
        else if (nin == 2) vg_item<2, 1>(p, rt, cb, P);
        else vg_item<3, 1>(p, rt, cb, P);
    } else {
        const int nout = p.N / VG_SEG;
        if (nin == 2) vg_item<2, 2>(p, rt, cb, P);
        else if (nout == 2) vg_item<1, 2>(p, rt, cb, P);
        else vg_item<1, 3>(p, rt, cb, P);
    }
}

int vg_check(const VgProblem &p) {
    STAIR_CHECK(p.rows >= 0 && p.a && p.W && p.N >= 1, "null operand / weight");
    STAIR_CHECK(p.pack >= VG_IN_A && p.pack <= VG_IN_MASK, "unknown input form");
    STAIR_CHECK(p.pack == VG_IN_A || p.b, "the input form needs a second operand row");
    STAIR_CHECK(p.kred >= 1 && p.kred <= VG_SEG && (p.kred == VG_SEG || p.pack == VG_IN_A), "reduction length per segment: 512, or shorter for a plain row");
    STAIR_CHECK(p.lda % 4 == 0 && p.ldb % 4 == 0 && p.ldw % 4 == 0 && p.ld_save % 4 == 0, "row strides must be multiples of 4 floats");
    STAIR_CHECK(((reinterpret_cast<uintptr_t>(p.a) | reinterpret_cast<uintptr_t>(p.b) | reinterpret_cast<uintptr_t>(p.W) | reinterpret_cast<uintptr_t>(p.in_save)) & 15) == 0,
                "operands must be 16-byte aligned");
    if (p.kind == VG_FWD) {
        STAIR_CHECK(p.out, "null output");
        STAIR_CHECK(p.act != 2 || p.emask, "act 2 multiplies by relu'(emask)");
    } else {
        STAIR_CHECK(p.kind == VG_ADJ, "unknown problem kind");
        STAIR_CHECK(p.N == 2 * VG_SEG || p.N == 3 * VG_SEG, "the adjoint form produces 2 or 3 H-wide gradient blocks");
        STAIR_CHECK(p.ga && p.gb, "null gradient rows");
        STAIR_CHECK(p.N == 2 * VG_SEG ? p.adj == VG_IN_CAT2 : ((p.adj == VG_IN_EXISTS || p.adj == VG_IN_XOR) && p.fa && p.fb), "adjoint form / forward operand rows");
        STAIR_CHECK(p.pack == VG_IN_A || p.pack == VG_IN_MASK || (p.pack == VG_IN_CAT2 && p.N == 2 * VG_SEG), "input form of an adjoint problem");
    }
    return 0;
}

}  // namespace

bool vec_group_usable(int H) { return H == VG_SEG && matmul_mode() == STAIR_MATMUL_BF16X3; }

int launch_vec_group(const VgProblem *probs, int n, hipStream_t s) {
    STAIR_CHECK(matmul_mode() == STAIR_MATMUL_BF16X3, "the grouped vector-level products compute split-bf16 products (STAIR_MATMUL_BF16X3)");
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&vec_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, VG_LDS));
        attr_set[dev] = true;
    }
    for (int at = 0; at < n;) {
        VgParams pp;
        pp.np = 0; pp.first[0] = 0;
        while (at < n && pp.np < VG_MAXP) {
            const VgProblem &p = probs[at++];
            if (p.rows == 0) continue;
            if (int rc = vg_check(p)) return rc;
            const int nblk = p.kind == VG_ADJ ? VG_SEG / 64 : (p.N + 63) / 64;
            const int nin = p.pack == VG_IN_A || p.pack == VG_IN_MASK ? 1 : (p.pack == VG_IN_CAT2 ? 2 : 3);
            pp.p[pp.np] = p;
            pp.first[pp.np + 1] = pp.first[pp.np] + ((p.rows + 63) / 64) * nblk;
            ++pp.np;
            const int64_t K = (int64_t)nin * p.kred;
            STAIR_ACCT_MFMA("vec_group", ((int64_t)p.rows * (K + p.N) + (int64_t)p.N * K) * 4, 2 * (int64_t)p.rows * p.N * K);
        }
        if (pp.np == 0) continue;
        for (int j = pp.np; j < VG_MAXP; ++j) { pp.p[j] = pp.p[0]; pp.first[j + 1] = pp.first[pp.np]; }
        hipLaunchKernelGGL(vec_group_kernel, dim3(pp.first[pp.np]), dim3(512), VG_LDS, s, pp);
        STAIR_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace stair

extern "C" int stair_vec_group(const stair_vec_problem *problems, int32_t count, stair_stream stream) {
    if (!problems || count < 0) {
        stair::set_error("stair_vec_group: null problems");
        return 1;
    }
    return stair::launch_vec_group(problems, count, static_cast<hipStream_t>(stream));
}
