// Grouped vector-level products: every small row-wise Linear of a program level in ONE launch.
//
// The vector-level modules of /root/reference/video_nmn/modules.py -- Compare (:15-21), Equals (:24-37), Xor (:59-72), ToAction
// (:102-120), Exists (:141-159) -- Filter's dense layer on the pooled rows (:376-378), Localize's keyword projection (:199-203) and
// the decoder (module_net.py:49-53, 136-138) are Linear layers on ONE [H] row per instance.  A level of a 128-question batch holds a
// handful of such products of 13 .. 130 rows each; as separate launches (pack -> split-K GEMM -> reduction, per module) a training
// step spent ~150 of its ~290 launches on them.  Here a launch carries a list of PROBLEMS; its work items are (problem, 32-row tile,
// 32-column block) and a workgroup computes its [32 x 32] output block over the FULL reduction length (<= 1536), so there is no
// split-K scratch and no reduction launch:
//   * 8 waves = 8 slices of the reduction dimension (64 columns of every input segment each).  A wave requests ALL its weight
//     fragments first (fragment-order bf16 hi / lo planes, 1 KB per fragment: stair_pack_wfrag), stages its slice of the operand rows
//     through a private LDS area with row-contiguous loads, and multiplies (hi.hi + lo.hi + hi.lo on v_mfma_f32_32x32x16_bf16, fp32
//     accumulate: the arithmetic of csrc/gemm_bf16x3.hip);
//   * the concatenated inputs of the modules ([a, b], [|a - b|, a, b], [a, b, a * b]) are formed in registers from the two operand
//     rows -- never materialised for the product; a training plan keeps them (in_save) as the weight-gradient operand;
//   * the eight slices meet in LDS (fixed order: deterministic, and a row's result does not depend on the other rows of the
//     launch); bias / ReLU / relu' mask, then 128-byte row segments go out: plain stores, or float atomics (gradient rows that
//     several instances share);
//   * backward (kind ADJ): dX = dZ W through the transposed weight image, all 2 - 3 H-wide blocks of dX for one 32-column slice in
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
constexpr int VG_PLD = 36;                   // row stride (floats) of the [64 x 32] partial tiles in LDS
constexpr int VG_ROWS = 32;                  // rows per work item
constexpr int VG_XLD = 68;                   // row stride (floats) of a wave's operand staging
constexpr int VG_LDS = 8 * 2 * 32 * VG_XLD * 4;      // eight waves x (a, b) x [32 rows x 64 columns] staging; the eight [32 x 32] partial
                                                     // tiles of the reduction slices (8 * 32 * VG_PLD floats) overlay it afterwards
static_assert(VG_LDS >= 8 * 32 * VG_PLD * 4, "the partial tiles fit over the staging");
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

// 8 consecutive floats at p[k .. k + 7]; TAIL: zero beyond `lim` (the reduction tail of a row shorter than a whole step)
template <bool TAIL>
__device__ __forceinline__ void vg_load8(const float *p, int k, int lim, v4f &x0, v4f &x1) {
    if (!TAIL || k + 8 <= lim) {
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

// One work item: rows [32 rt, 32 rt + 32) x the 32 output columns of block cb (x NOUT H-wide blocks for an adjoint problem).
// Wave w owns columns [64 w, 64 w + 64) of EVERY input segment (an eighth of the reduction).  A work item is a short chain of memory
// round trips, so everything is requested as early as it can be:
//   * weights: the wave's fragments of every output block -- NOUT x NIN segments x 4 steps x (hi, lo), 1 KB each, fragment-order
//     planes (stair_pack_wfrag; one [512 x 512] image per (block of 512 output columns, input segment)) -- are ALL requested first: the
//     whole weight stream of the item is one L2 round trip.  (A ring of a few steps made a launch take 9 us per 512 of reduction
//     length whatever it computed.)  wplanes == NULL: fp32 rows, split on the way (column / reduction tails);
//   * operand rows: the wave fetches its [32 rows x 64 columns] pieces of a and b with ROW-CONTIGUOUS 16-byte loads (16 lanes per row)
//     into a wave-private LDS staging area, once, and reads the MFMA fragments back from there -- the fragment shape itself (lane (r, h)
//     = 8 floats of row r) touches 32 cache lines per load instruction, and the first version of this kernel was bound by the L1's
//     line rate.  The formed, split fragments stay in registers for all output blocks.
template <int NIN, int NOUT, bool TAIL, bool PLANES>
__device__ __forceinline__ void vg_item(const VgProblem &p, const int rt, const int cb, float *P) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int rows = p.rows, kred = p.kred;
    const bool two = p.pack != VG_IN_A;
    const int k0 = 64 * wave;
    const int nsteps = min(4, max(0, (kred - k0 + 15) / 16));             // kred = 512: 4 steps per wave and segment
    const int blk0 = p.kind == VG_ADJ ? 0 : cb / 16, nt = cb % 16;

    // ---- 1. every weight fragment of the item ----
    bf16x8 wpl[PLANES ? NOUT : 1][PLANES ? NIN : 1][4][2];
    v4f wv[PLANES ? 1 : NIN][4][2];                                        // (fp32 rows: forward-shaped problems only, NOUT = 1)
    if (PLANES) {
        const bf16x8 *wq = static_cast<const bf16x8 *>(p.wplanes);
#pragma unroll
        for (int j = 0; j < NOUT; ++j)
#pragma unroll
            for (int s = 0; s < NIN; ++s) {
                const bf16x8 *img = wq + (int64_t)((blk0 + j) * NIN + s) * (VG_SEG * VG_SEG * 2 / 8) + lane;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) wpl[PLANES ? j : 0][PLANES ? s : 0][ks][pl] = img[((nt * (VG_SEG / 16) + 4 * wave + ks) * 2 + pl) * 64];
            }
    } else {
        const float *wrow = p.W + (int64_t)min(cb * 32 + r, p.N - 1) * p.ldw;
#pragma unroll
        for (int s = 0; s < NIN; ++s)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                if (ks < nsteps) vg_load8<TAIL>(wrow + s * VG_SEG, k0 + 16 * ks + 8 * h, kred, wv[PLANES ? 0 : s][ks][0], wv[PLANES ? 0 : s][ks][1]);
    }

    // ---- 2. the wave's [32 rows x 64 columns] of a and b: row-contiguous loads -> wave-private LDS -> fragments in registers ----
    float *Xa = P + wave * (2 * 32 * VG_XLD), *Xb = Xa + 32 * VG_XLD;
    bf16x8 xh[NIN][4], xl[NIN][4];
    if (nsteps > 0) {
        const int rsub = lane >> 4, c4 = (lane & 15) * 4;    // lane = (row 4 i + rsub, columns c4 .. c4 + 3 of the slice)
        const int kc = k0 + c4;
        const bool live = !TAIL || kc < kred;
        float *save_base = p.in_save != nullptr && cb == 0 ? p.in_save : nullptr;
        int aoff[8], boff[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rowc = min(rt * 32 + 4 * i + rsub, rows - 1);
            aoff[i] = (int)((p.ia ? p.ia[rowc] : rowc) * p.lda);
            boff[i] = two ? (int)((p.ib ? p.ib[rowc] : rowc) * p.ldb) : aoff[i];
        }
        v4f va[8], vb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            va[i] = live ? *reinterpret_cast<const v4f *>(p.a + aoff[i] + kc) : v4f{0.f, 0.f, 0.f, 0.f};
            if (two) vb[i] = live ? *reinterpret_cast<const v4f *>(p.b + boff[i] + kc) : v4f{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + rsub;
            *reinterpret_cast<v4f *>(Xa + row * VG_XLD + c4) = va[i];
            if (two) *reinterpret_cast<v4f *>(Xb + row * VG_XLD + c4) = vb[i];
            if (save_base && rt * 32 + row < rows && live) {       // the formed input rows (the weight-gradient operand), 256 contiguous bytes per 16 lanes
#pragma unroll
                for (int s = 0; s < NIN; ++s) {
                    v4f x0, x1;
                    vg_form(p.pack, s, p.in_scale, va[i], va[i], two ? vb[i] : va[i], two ? vb[i] : va[i], x0, x1);
                    *reinterpret_cast<v4f *>(save_base + (int64_t)(rt * 32 + row) * p.ld_save + s * VG_SEG + kc) = x0;
                }
            }
        }
        // (wave-private staging: the wave's own LDS accesses complete in order, no workgroup barrier)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float *xr = Xa + r * VG_XLD + 16 * ks + 8 * h, *yr = Xb + r * VG_XLD + 16 * ks + 8 * h;
            const v4f a0 = *reinterpret_cast<const v4f *>(xr), a1 = *reinterpret_cast<const v4f *>(xr + 4);
            v4f b0 = a0, b1 = a1;
            if (two) { b0 = *reinterpret_cast<const v4f *>(yr); b1 = *reinterpret_cast<const v4f *>(yr + 4); }
#pragma unroll
            for (int s = 0; s < NIN; ++s) {
                v4f x0, x1;
                vg_form(p.pack, s, p.in_scale, a0, a1, b0, b1, x0, x1);
                vg_split8(x0, x1, xh[s][ks], xl[s][ks]);
            }
        }
    }

    // ---- 3. per output block: the products, then the eight slices of the reduction meet in LDS (over the staging); thread (wave, lane)
    //         owns column lane % 32 of rows g and g + 16, g = 2 wave + lane / 32 ----
    float d[NOUT][2];
    const int ecol = lane & 31, erow = 2 * wave + (lane >> 5);
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks >= nsteps) break;
#pragma unroll
            for (int s = 0; s < NIN; ++s) {
                bf16x8 wh, wl;
                if (PLANES) { wh = wpl[PLANES ? j : 0][PLANES ? s : 0][ks][0]; wl = wpl[PLANES ? j : 0][PLANES ? s : 0][ks][1]; }
                else vg_split8(wv[PLANES ? 0 : s][ks][0], wv[PLANES ? 0 : s][ks][1], wh, wl);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[s][ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[s][ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[s][ks], acc, 0, 0, 0);
            }
        }
        __syncthreads();                          // every wave has its fragments in registers (j = 0) / has read the previous block's partials
        // lane (r, h) holds row r, columns 8 q + 4 h + i for e = 4 q + i
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v4f z;
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = acc[4 * q + i];
            *reinterpret_cast<v4f *>(P + (wave * 32 + r) * VG_PLD + 8 * q + 4 * h) = z;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = erow + 16 * i;
            float v = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) v += P[(w8 * 32 + t) * VG_PLD + ecol];        // fixed order: deterministic
            d[j][i] = v;
        }
    }

    const int col = cb * 32 + ecol;
    if (p.kind == VG_FWD) {
        if (col >= p.N) return;
        const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rw = rt * 32 + erow + 16 * i;
            if (rw >= rows) break;
            float v = d[0][i] + bias;
            if (p.act == 1) {
                v = fmaxf(v, 0.f);
                if (p.drop_site) {             // nn.Dropout behind the ReLU: stair_dropout_fwd's bits for element rw * N + col of the [rows, N] matrix
                    const unsigned long long e = (unsigned long long)rw * p.N + col;
                    v = drop_keep(drop_hash4(p.drop_seed, p.drop_site - 1u, e >> 2), (int)(e & 3), (unsigned)(p.drop_p * 65536.0f)) ? v * (1.0f / (1.0f - p.drop_p)) : 0.f;
                }
            }
            else if (p.act == 2) v = p.emask[(int64_t)rw * p.ldm + col] > 0.f ? v * p.escale : 0.f;
            float *dst = p.out + (int64_t)(p.io ? p.io[rw] : rw) * p.ldo + col;
            if (p.accumulate) unsafeAtomicAdd(dst, v);
            else *dst = v;
        }
    } else if (NOUT > 1) {
        // adjoint of the concatenation: d[0], d[1] (, d[2]) are the gradient blocks at column `col` of the H-wide operand rows
        constexpr int J1 = NOUT > 1 ? 1 : 0, J2 = NOUT > 2 ? 2 : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rw = rt * 32 + erow + 16 * i;
            if (rw >= rows) break;
            const int64_t ra = (int64_t)(p.fia ? p.fia[rw] : rw) * p.ldfa + col, rb = (int64_t)(p.fib ? p.fib[rw] : rw) * p.ldfb + col;
            // the operands' GRADIENT rows: the same rows unless the plan sends this reader to a staging row (deterministic fan-in)
            const int64_t gra = p.gia ? (int64_t)p.gia[rw] * p.ldfa + col : ra, grb = p.gib ? (int64_t)p.gib[rw] * p.ldfb + col : rb;
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
            unsafeAtomicAdd(p.ga + gra, da);
            unsafeAtomicAdd(p.gb + grb, db);
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
    const VgProblem p = pp.p[sel];            // a copy in scalar registers: a reference into the argument block made the compiler
                                              // copy the whole block to scratch
    const int item = w - pp.first[sel];
    const int nblk = p.kind == VG_ADJ ? VG_SEG / 32 : (p.N + 31) / 32;
    const int rt = item / nblk, cb = item - rt * nblk;
    const int nin = p.pack == VG_IN_A || p.pack == VG_IN_MASK ? 1 : (p.pack == VG_IN_CAT2 ? 2 : 3);
    // Variants: weights as fragment-order planes for every form; fp32 weight rows only where planes cannot be (a reduction tail: the
    // decoder's last layer seen from behind; a column tail: that layer itself) -- each variant is unrolled code of its own, and a
    // dispatcher holding all combinations spilled although no single variant does
    const bool planes = p.wplanes != nullptr;
    if (p.kind == VG_FWD) {
        if (nin == 1 && !planes) vg_item<1, 1, true, false>(p, rt, cb, P);
        else if (nin == 2 && !planes) vg_item<2, 1, false, false>(p, rt, cb, P);
        else if (nin == 1) vg_item<1, 1, false, true>(p, rt, cb, P);
        else if (nin == 2) vg_item<2, 1, false, true>(p, rt, cb, P);
        else vg_item<3, 1, false, true>(p, rt, cb, P);
    } else {
        const int nout = p.N / VG_SEG;
        if (nin == 2) vg_item<2, 2, false, true>(p, rt, cb, P);
        else if (nout == 2) vg_item<1, 2, false, true>(p, rt, cb, P);
        else vg_item<1, 3, false, true>(p, rt, cb, P);
    }
}

int vg_check(const VgProblem &p) {
    STAIR_CHECK(p.rows >= 0 && p.a && p.W && p.N >= 1, "null operand / weight");
    STAIR_CHECK(p.pack >= VG_IN_A && p.pack <= VG_IN_MASK, "unknown input form");
    STAIR_CHECK(p.pack == VG_IN_A || p.b, "the input form needs a second operand row");
    STAIR_CHECK(p.kred >= 1 && p.kred <= VG_SEG && (p.kred == VG_SEG || p.pack == VG_IN_A), "reduction length per segment: 512, or shorter for a plain row");
    STAIR_CHECK(p.kred % 4 == 0, "reduction length must be a multiple of 4");
    STAIR_CHECK(p.wplanes || (p.kind == VG_FWD && (p.pack == VG_IN_A || p.pack == VG_IN_CAT2)),
                "fp32 weight rows are read for plain and two-segment forward problems only; every other form needs weight planes");
    STAIR_CHECK(!p.wplanes || (p.kred == VG_SEG && (p.kind == VG_ADJ || p.N % 32 == 0) && (reinterpret_cast<uintptr_t>(p.wplanes) & 15) == 0),
                "weight planes need whole 32-column blocks and 512-wide segments");
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
            const int nblk = p.kind == VG_ADJ ? VG_SEG / 32 : (p.N + 31) / 32;
            const int nin = p.pack == VG_IN_A || p.pack == VG_IN_MASK ? 1 : (p.pack == VG_IN_CAT2 ? 2 : 3);
            pp.p[pp.np] = p;
            pp.first[pp.np + 1] = pp.first[pp.np] + ((p.rows + VG_ROWS - 1) / VG_ROWS) * nblk;
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
