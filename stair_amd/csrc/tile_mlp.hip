// Fused per-clip tile operators: one workgroup carries one [T, H] tile (T <= 64 frames of one module instance, H = 512)
// through up to three Linear layers and the module's tile-local tail WITHOUT the intermediates leaving the CU.
//
// Replaces, per module of /root/reference/video_nmn/modules.py, the launch sequences GEMM -> HBM -> GEMM -> HBM -> row kernel:
//   Localize   (:199-217)  Lin . ReLU . Lin on the tile, then (cos(f_t, k_j) + 1) * 0.49 against the K keyword rows
//   Filter     (:363-378)  Lin . ReLU . Lin . ReLU, then sum over the frames (its attention is identically 1)
//   FilterFrame(:399-414)  Lin . ReLU . Lin . ReLU = f, a_t = sigmoid(w[:H] . f_t + w[H:] . kw + b), ReLU(Lin(a_t f_t))
//   HasItem    (:123-138)  Lin . ReLU, then sigmoid(w . row + b)
//   Temporal   (:310-327)  ReLU(Lin(r_t feat_t)), then LayerNorm over H
//   Superlative(:220-248)  Localize's two layers (the scores against Ka = T action rows stay in cosine_attn_grouped_kernel)
//
// Mapping to the CU (8 waves, one workgroup per CU, 129 KB of LDS):
//   * the tile lives in LDS as bf16 hi + lo planes (x = hi + lo), laid out per 32-wide k stage as [64 rows][4 slots of 16 B]
//     with the slot swizzle of csrc/gemm_planes.hip (conflict-free ds_read_b128 fragment reads);
//   * a layer is computed TRANSPOSED, Z^T = W . tile^T: the weight rows are the MFMA A operand.  Wave w owns output columns
//     [64 w, 64 w + 64) and nobody else needs those weight rows, so W never goes through LDS: its bf16 hi / lo planes are
//     stored in HBM in FRAGMENT ORDER (stair_pack_wfrag: 1 KB per (32-row tile, 16-wide k step, plane)) and each wave streams
//     its own fragments global -> VGPR with three k steps in flight; the tile (the B operand) is the only thing read from
//     LDS, 4 KB per wave and k step for 12 v_mfma_f32_32x32x16_bf16 (hi.hi + lo.hi + hi.lo, fp32 accumulate);
//   * the k loop has NO barrier (the tile is read-only during a layer, W is private to the wave); the workgroup meets only
//     between layers: every layer's output is staged as fp32 [T][H] rows in LDS (rows 516 floats apart, over the dead image),
//     and everything that touches HBM there is ROW-wise -- a wave per row, a lane per 8 consecutive columns, 2 KB contiguous
//     per access: the saved activation a backward pass needs, relu'(saved activation) of a backward chain, FilterFrame's
//     attention -- before the rows are split into the next layer's bf16 image;
//   * every tail -- coalesced row stores, the sum over frames, cosine / dot products per row, LayerNorm, atomic accumulation
//     in 256-byte wave-instructions -- reads the staged rows of the last layer.
#include <algorithm>
#include <cstdlib>
#include <utility>
#include <vector>

#include "common.h"
#include "ops.h"

namespace stair {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using v4f = __attribute__((ext_vector_type(4))) float;

namespace {

constexpr int TM_H = 512;                       // hidden size the kernel is built for (the reference's, args.py:27)
constexpr int TM_ROWS = 64;                     // frames per tile
constexpr int TM_KS = TM_H / 16;                // 16-wide k steps per layer
constexpr int TM_STAGE = 8192;                  // one 32-wide k stage of the tile image: hi plane 4 KB, lo plane 4 KB
constexpr int TM_IMG = (TM_H / 32) * TM_STAGE;  // 128 KB
constexpr int TM_FLD = TM_H + 4;                // row stride of the fp32 staging (floats): 16-byte rows, conflict-free float4 writes
constexpr int TM_F_BYTES = TM_ROWS * TM_FLD * 4;
constexpr int TM_LDS = TM_F_BYTES;              // 129 KB: the fp32 staging overlays the bf16 image
static_assert(TM_F_BYTES >= TM_IMG, "the fp32 staging covers the bf16 image");

// slot swizzle of the tile image (as pl_swz of csrc/gemm_planes.hip)
__device__ __forceinline__ int tm_swz(int R) {
    const int q0 = (R >> 2) & 1, q1 = (R >> 3) & 1;
    return ((q0 ^ q1) << 1) | q1;
}

constexpr int TM_MAXB = 8;           // module buckets one launch can carry (the argument block stays under the 4 KB kernarg limit)
// One bucket's arguments as the kernel sees them: stair_tile_mlp_args with the fields that only ONE kernel form reads overlaid
// (a launch carries buckets of one form), so that eight buckets still fit the 4 KB argument block.  Same names as the public struct.
struct TmArg {
    const float *X; int64_t x_gstride; const int32_t *x_idx;
    const float *row_scale; const int32_t *rs_idx;
    const void *W[3]; const float *bias[3]; int32_t act[3]; int32_t n_layers;
    float *save[3];
    int32_t mid_rowdot; const float *vw, *vb, *extra; float *rs_out;
    int32_t tail;
    float *out; int64_t out_gstride; const int32_t *out_idx;
    const float *gamma, *beta; float ln_eps;
    const float *kb; const int32_t *pair_first, *pair_cnt, *att_idx; float *att;
    const int32_t *len;
    int32_t cnt, T, H;
    const float *act_mask[3]; float act_scale;
    const float *in_mask; int64_t in_mask_gstride; const int32_t *in_mask_idx; float in_scale; int32_t x_broadcast;
    float *save_in;
    int32_t acc_exclusive;
    union {
        struct {            // KIND 1: vector-level tiles
            int32_t vec_pack, vec_cnt; const float *pk_a, *pk_b; const int32_t *pk_a_idx, *pk_b_idx; float *cat_save; const int32_t *out_row_idx;
        };
        struct {            // KIND 2: Temporal's backward chain
            float *dgamma, *dbeta; const float *adj_feat; int64_t adj_feat_gstride; const int32_t *adj_feat_idx;
            const float *adj_rs; const int32_t *adj_rs_idx; float *adj_drs;
        };
        struct {            // KIND 0: relu' bit masks of the map-level operators and their chains
            unsigned long long *save_bits[3]; const unsigned long long *act_bits[3]; const unsigned long long *in_bits;
            uint16_t drop_site[3];       // forward launches: 1 + dropout site behind layer l's activation (0: none)
        };
    };
};
static TmArg tm_arg(const stair_tile_mlp_args &a, int kind) {
    TmArg t = {};
    t.X = a.X; t.x_gstride = a.x_gstride; t.x_idx = a.x_idx; t.row_scale = a.row_scale; t.rs_idx = a.rs_idx;
    for (int l = 0; l < 3; ++l) { t.W[l] = a.W[l]; t.bias[l] = a.bias[l]; t.act[l] = a.act[l]; t.save[l] = a.save[l]; t.act_mask[l] = a.act_mask[l]; }
    t.n_layers = a.n_layers; t.mid_rowdot = a.mid_rowdot; t.vw = a.vw; t.vb = a.vb; t.extra = a.extra; t.rs_out = a.rs_out;
    t.tail = a.tail; t.out = a.out; t.out_gstride = a.out_gstride; t.out_idx = a.out_idx;
    t.gamma = a.gamma; t.beta = a.beta; t.ln_eps = a.ln_eps;
    t.kb = a.kb; t.pair_first = a.pair_first; t.pair_cnt = a.pair_cnt; t.att_idx = a.att_idx; t.att = a.att; t.len = a.len;
    t.cnt = a.cnt; t.T = a.T; t.H = a.H; t.act_scale = a.act_scale;
    t.in_mask = a.in_mask; t.in_mask_gstride = a.in_mask_gstride; t.in_mask_idx = a.in_mask_idx; t.in_scale = a.in_scale;
    t.x_broadcast = a.x_broadcast; t.save_in = a.save_in; t.acc_exclusive = a.acc_exclusive;
    if (kind == 1) {
        t.vec_pack = a.vec_pack; t.vec_cnt = a.vec_cnt; t.pk_a = a.pk_a; t.pk_b = a.pk_b; t.pk_a_idx = a.pk_a_idx; t.pk_b_idx = a.pk_b_idx;
        t.cat_save = a.cat_save; t.out_row_idx = a.out_row_idx;
    } else if (kind == 2) {
        t.dgamma = a.dgamma; t.dbeta = a.dbeta; t.adj_feat = a.adj_feat; t.adj_feat_gstride = a.adj_feat_gstride; t.adj_feat_idx = a.adj_feat_idx;
        t.adj_rs = a.adj_rs; t.adj_rs_idx = a.adj_rs_idx; t.adj_drs = a.adj_drs;
    } else {
        for (int l = 0; l < 3; ++l) { t.save_bits[l] = a.save_bits[l]; t.act_bits[l] = a.act_bits[l]; }
        t.in_bits = a.in_bits;
        for (int l = 0; l < 3; ++l) t.drop_site[l] = (uint16_t)a.drop_site[l];
    }
    return t;
}
struct TmParams {
    TmArg a[TM_MAXB];                // the buckets of one program level: independent of each other, so their tiles share one launch
    int first[TM_MAXB + 1];          // work item w belongs to bucket b with first[b] <= w < first[b + 1]; its tile is w - first[b]
    int nb;
    long long *fx_g, *fx_b;          // KIND 2: fixed-point shadows of d gamma / d beta (NULL: float atomics into a[0].dgamma / dbeta)
    unsigned drop_thresh; float drop_inv_keep; unsigned long long drop_seed;    // forward launches under dropout (drop_thresh 0: none)
    unsigned *counter;               // work queue: counter[0] = head, counter[1] = workgroups that have left the queue.  Both words are
                                     // zero between launches (the last workgroup to leave puts them back); NULL: tiles are dealt out round robin
};

static_assert(sizeof(TmParams) <= 4096, "the argument block of a launch must stay under the 4 KB kernarg limit");

}  // namespace

// hi = bf16(x), lo = bf16(x - hi) of 8 consecutive floats
__device__ __forceinline__ void tm_split8(const v4f a, const v4f b, bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = (__bf16)a[j]; lo[j] = (__bf16)(a[j] - (float)hi[j]);
        hi[4 + j] = (__bf16)b[j]; lo[4 + j] = (__bf16)(b[j] - (float)hi[4 + j]);
    }
}

// NT: the tile's own traffic (input rows, masks, saved activations, output rows) is streamed with the non-temporal hint, so that
// it does not push the weight planes -- which every workgroup of the XCD re-reads for every tile -- out of the 4 MB L2.
template <bool NT>
__device__ __forceinline__ v4f tm_ld(const float *p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return *reinterpret_cast<const v4f *>(p);
}
template <bool NT>
__device__ __forceinline__ void tm_st(float *p, const v4f v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(p));
    else *reinterpret_cast<v4f *>(p) = v;
}

// a stage boundary: the workgroup barrier, and nothing scheduled across it (hipcc otherwise hoists the next stage's loads over the
// barrier into the previous stage, where the accumulators are still live: spills)
// hide a uniform pointer's provenance from the optimiser: what is derived from the result cannot be hoisted above this point
// (row addresses precomputed outside the tile / layer loops are what ran the kernel out of registers)
template <typename P>
__device__ __forceinline__ P *tm_fresh(P *p) {
    asm volatile("" : "+s"(p));
    return p;
}

__device__ __forceinline__ int tm_fresh_v(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

#define TM_SYNC() do { __builtin_amdgcn_sched_barrier(0); __syncthreads(); __builtin_amdgcn_sched_barrier(0); } while (0)

// KIND 0: map-level forward operators; 1: vector-level tiles (64 instances per tile); 2: Temporal's backward chain (the LayerNorm
// adjoint on the way in, the row-scale adjoint on the way out); 3: the map-level backward chains (relu' masks on the way in and
// between the layers, broadcast input, dZ saves, accumulation).  Kernels of their own: what one form needs in registers and code
// the others do not pay for (one kernel for everything kept 250+ registers live across the tile loop and spilled).
template <bool NT, int KIND>
__global__ __launch_bounds__(512, 1) void tile_mlp_kernel(TmParams pp) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float *F = reinterpret_cast<float *>(lds);
    // KIND 2: d gamma / d beta of the workgroup's tiles, column tid, as 64-bit fixed point (common.h): integer sums do not depend on
    // which workgroup the queue gave which tile, so the parameter gradients are bit-identical from run to run
    __shared__ long long tb_acc[KIND == 2 ? 2 * TM_H : 2];
    if (KIND == 2) { tb_acc[threadIdx.x] = 0; tb_acc[TM_H + threadIdx.x] = 0; }
    // the lane id comes from mbcnt wherever it is needed (EXEC is all ones there): threadIdx.x kept alive across the persistent
    // loop, beside 250+ live registers, went to scratch, and a kernel with ANY scratch pays for its setup at every launch
#define TM_LANE() ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
    // the wave index is uniform over the wave: in an SGPR every row address (row = wave + 8 j) is scalar arithmetic and the row-wise
    // accesses take the base-in-SGPR + one shared lane offset form -- as a VGPR it made every row of every array a 64-bit VGPR
    // pointer of its own (hoisted out of the loops: 150 spilled registers once the loads were batched)
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#define TM_TID() (64 * wave + TM_LANE())
    const int total = pp.first[pp.nb];
    __shared__ int next_work;
    // Work items = tiles of all buckets of the launch, handed out through one atomic counter: a workgroup that finishes a
    // one-layer tile takes the next item while another is still in a three-layer one, and the last partial round of one
    // bucket is filled with the next bucket's tiles (the buckets are listed by decreasing layer count).
    for (int it = 0;; ++it) {
        int w;
        TM_SYNC();                          // the previous tile's tail has finished reading the staging (and next_work)
        if (pp.counter) {
            if (TM_TID() == 0) next_work = (int)atomicAdd(pp.counter, 1u);
            TM_SYNC();
            w = __builtin_amdgcn_readfirstlane(next_work);       // uniform BY CONSTRUCTION: says so, so that everything derived from it
                                                                 // (argument block, tile pointers) lives in SGPRs, not in 64-bit VGPR pairs
        } else {
            w = blockIdx.x + it * gridDim.x;
        }
        // UNSIGNED: whatever the head word held (a ticket >= 2^31 read as a negative int passed the signed test and indexed the
        // bucket table and the tile pointers out of bounds), a workgroup only ever touches tiles 0 .. total - 1
        if ((unsigned)w >= (unsigned)total) break;
        int bsel = 0;
#pragma unroll
        for (int j = 1; j < TM_MAXB; ++j) bsel += (j < pp.nb && w >= pp.first[j]) ? 1 : 0;
        const TmArg &p = pp.a[bsel];
        const int inst = w - pp.first[bsel];
        // the lane id, re-derived per tile from an opaque copy: the dozens of lane-dependent LDS / row offsets below are then computed
        // where they are used instead of being hoisted out of this (persistent) loop and kept alive -- or spilled -- across it
        const int lane = tm_fresh_v(TM_LANE());
        const int r = lane & 31, h = lane >> 5;
        // B-operand fragment offsets (bytes inside a stage plane, chunk 0) of this lane's two frame tiles
        int offT[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int R = 32 * tt + r;
            offT[tt] = (R * 4 + tm_swz(R)) * 16;
        }


        // rows of this tile: the T frames of a module instance, or (vector-level modules) up to 64 INSTANCES of one row each
        const int vpack = KIND == 1 ? p.vec_pack : 0;    // the vector-level form is a kernel of its own (registers)
        const int T = vpack ? min(TM_ROWS, p.vec_cnt - TM_ROWS * inst) : p.T;
        const int Ts = vpack ? TM_ROWS : p.T;            // rows between two tiles in the [cnt, T, H] save / mask buffers
        const int nseg = vpack == 0 ? 1 : (vpack == 1 ? 2 : 3);
        // ---- the input tile: fp32 rows -> (row scale) -> bf16 hi / lo image ------------------------------------------
        const float *x = tm_fresh(p.X + (int64_t)(p.x_idx ? __builtin_amdgcn_readfirstlane(p.x_idx[inst]) : inst) * p.x_gstride);
        const float *rsrow = KIND == 0 && p.row_scale ? p.row_scale + (int64_t)(p.rs_idx ? __builtin_amdgcn_readfirstlane(p.rs_idx[inst]) : inst) * T : nullptr;
        // (a NULL must stay a visible NULL: through tm_fresh the forward kernel kept the whole mask path, and its registers)
        const float *imask = (KIND == 2 || KIND == 3) && p.in_mask ? tm_fresh(p.in_mask + (int64_t)(p.in_mask_idx ? __builtin_amdgcn_readfirstlane(p.in_mask_idx[inst]) : inst) * p.in_mask_gstride) : nullptr;
        const bool xbc = KIND == 3 && p.x_broadcast;
        const int Lrows = xbc ? (p.len ? __builtin_amdgcn_readfirstlane(p.len[inst]) : T) : T;     // a broadcast row fills the clip's own frames only
        // vector-level modules: row t of the tile is H-wide block `seg` of the concatenation built from the two operand rows
        // of instance 64 inst + t (never materialised for the GEMM; cat_save keeps it for the weight gradient)
        auto build_vec_image = [&](const int seg) {
            const int c8 = tm_fresh_v(lane);          // (row wave + 8 j, columns 8 c8 .. +7); offsets computed here, see above
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                v4f a0[4], a1[4], b0[4], b1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {         // all operand rows of four instances first, then the arithmetic
                    const int t = wave + 8 * (4 * half + j);
                    const int64_t i = (int64_t)TM_ROWS * inst + (t < T ? t : 0);
                    const float *ar = p.pk_a + (int64_t)(p.pk_a_idx ? __builtin_amdgcn_readfirstlane(p.pk_a_idx[i]) : i) * TM_H + 8 * c8;
                    const float *br = p.pk_b + (int64_t)(p.pk_b_idx ? __builtin_amdgcn_readfirstlane(p.pk_b_idx[i]) : i) * TM_H + 8 * c8;
                    a0[j] = *reinterpret_cast<const v4f *>(ar); a1[j] = *reinterpret_cast<const v4f *>(ar + 4);
                    b0[j] = *reinterpret_cast<const v4f *>(br); b1[j] = *reinterpret_cast<const v4f *>(br + 4);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = wave + 8 * (4 * half + j);
                    v4f a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
                    if (t < T) {
                        if (vpack == 1) { a = seg == 0 ? a0[j] : b0[j]; b = seg == 0 ? a1[j] : b1[j]; }
                        else if (vpack == 2) {
                            if (seg == 0) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) { a[e] = fabsf(a0[j][e] - b0[j][e]); b[e] = fabsf(a1[j][e] - b1[j][e]); }
                            } else { a = seg == 1 ? a0[j] : b0[j]; b = seg == 1 ? a1[j] : b1[j]; }
                        } else {
                            if (seg == 2) { a = a0[j] * b0[j]; b = a1[j] * b1[j]; }
                            else { a = seg == 0 ? a0[j] : b0[j]; b = seg == 0 ? a1[j] : b1[j]; }
                        }
                        if (p.cat_save) {
                            float *d = p.cat_save + (((int64_t)TM_ROWS * inst + t) * nseg + seg) * TM_H + 8 * c8;
                            *reinterpret_cast<v4f *>(d) = a; *reinterpret_cast<v4f *>(d + 4) = b;
                        }
                    }
                    bf16x8 hi, lo;
                    tm_split8(a, b, hi, lo);
                    const int off = (c8 >> 2) * TM_STAGE + (t * 4 + ((c8 & 3) ^ tm_swz(t))) * 16;
                    *reinterpret_cast<bf16x8 *>(lds + off) = hi;
                    *reinterpret_cast<bf16x8 *>(lds + off + 4096) = lo;
                }
            }
        };
        if (vpack) build_vec_image(0);
        else if (KIND == 2) {
            // Temporal's backward (autograd of modules.py:283,327: y = LayerNorm(a), a = ReLU(Lin(r_t feat_t))): the incoming gradient
            // rows dY = X go through the adjoint of LayerNorm at the saved rows a = in_mask and through relu'(a) on their way into the
            // image; what is written there (and kept: save_in) is dZ of the dense layer.  Thread (wave, lane) carries columns
            // 8 lane .. +7 of rows wave + 8 j, all loads of four rows first (as below).
            const int c8 = lane;
            const float *ysv = tm_fresh(imask);
            const v4f gm0 = *reinterpret_cast<const v4f *>(p.gamma + 8 * c8), gm1 = *reinterpret_cast<const v4f *>(p.gamma + 8 * c8 + 4);
            v4f ag0 = {0.f, 0.f, 0.f, 0.f}, ag1 = ag0, ab0 = ag0, ab1 = ag0;
            v4f dz[8][2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                v4f xa[4], xb[4], ya[4], yb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = wave + 8 * (4 * half + j);
                    const int ts = t < T ? t : 0;
                    xa[j] = tm_ld<NT>(x + (int64_t)ts * TM_H + 8 * c8);
                    xb[j] = tm_ld<NT>(x + (int64_t)ts * TM_H + 8 * c8 + 4);
                    ya[j] = tm_ld<NT>(ysv + (int64_t)ts * TM_H + 8 * c8);
                    yb[j] = tm_ld<NT>(ysv + (int64_t)ts * TM_H + 8 * c8 + 4);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = wave + 8 * (4 * half + j);
                    float sum = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) sum += ya[j][i] + yb[j][i];
                    const float mean = wave_sum(sum) / (float)TM_H;
                    float sq = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float a = ya[j][i] - mean, b = yb[j][i] - mean; sq += a * a + b * b; }
                    const float rstd = rsqrtf(wave_sum(sq) / (float)TM_H + p.ln_eps);
                    float s1 = 0.f, s2 = 0.f;
                    v4f ha, hb;                 // the normalised rows
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ha[i] = (ya[j][i] - mean) * rstd; hb[i] = (yb[j][i] - mean) * rstd;
                        const float da = xa[j][i] * gm0[i], db = xb[j][i] * gm1[i];
                        s1 += da + db; s2 += da * ha[i] + db * hb[i];
                    }
                    s1 = wave_sum(s1) / (float)TM_H; s2 = wave_sum(s2) / (float)TM_H;
                    v4f oa = {0.f, 0.f, 0.f, 0.f}, ob = oa;
                    if (t < T) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            oa[i] = ya[j][i] > 0.f ? rstd * (xa[j][i] * gm0[i] - s1 - ha[i] * s2) * p.in_scale : 0.f;
                            ob[i] = yb[j][i] > 0.f ? rstd * (xb[j][i] * gm1[i] - s1 - hb[i] * s2) * p.in_scale : 0.f;
                            ag0[i] += xa[j][i] * ha[i]; ag1[i] += xb[j][i] * hb[i];
                            ab0[i] += xa[j][i]; ab1[i] += xb[j][i];
                        }
                    }
                    dz[4 * half + j][0] = oa; dz[4 * half + j][1] = ob;
                }
            }
            if (p.save_in) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = wave + 8 * j;
                    if (t < T) {
                        float *d = p.save_in + ((int64_t)inst * Ts + t) * TM_H + 8 * c8;
                        tm_st<NT>(d, dz[j][0]); tm_st<NT>(d + 4, dz[j][1]);
                    }
                }
            }
            // d gamma, d beta: the eight waves' column sums meet in the (not yet written) image area, are added in wave order, and
            // join the workgroup's fixed-point accumulators
            float *red = F + (2 * wave) * TM_H + 8 * c8;
            *reinterpret_cast<v4f *>(red) = ag0; *reinterpret_cast<v4f *>(red + 4) = ag1;
            *reinterpret_cast<v4f *>(red + TM_H) = ab0; *reinterpret_cast<v4f *>(red + TM_H + 4) = ab1;
            TM_SYNC();
            {
                float g = 0.f, b = 0.f;
                const int tid = TM_TID();
#pragma unroll
                for (int q = 0; q < 8; ++q) { g += F[(2 * q) * TM_H + tid]; b += F[(2 * q + 1) * TM_H + tid]; }
                tb_acc[tid] += __float2ll_rn(g * kFxScale);
                tb_acc[TM_H + tid] += __float2ll_rn(b * kFxScale);
            }
            TM_SYNC();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = wave + 8 * j;
                bf16x8 hi, lo;
                tm_split8(dz[j][0], dz[j][1], hi, lo);
                const int off = (c8 >> 2) * TM_STAGE + (t * 4 + ((c8 & 3) ^ tm_swz(t))) * 16;
                *reinterpret_cast<bf16x8 *>(lds + off) = hi;
                *reinterpret_cast<bf16x8 *>(lds + off + 4096) = lo;
            }
        } else {
            // thread (wave, lane) carries columns 8 lane .. +7 of rows wave + 8 j.  All global loads of four rows are issued before
            // anything uses them (a load -> use -> store -> load loop runs one memory round trip per row: 8 x ~2 us per tile);
            // rows past the tile's length read row 0 and are zeroed afterwards, so that no load hides behind a branch.
            const int c8 = lane;
            // relu' of a saved activation as ONE bit per element (8 bytes per thread and tile instead of 256: the forward pass wrote
            // them in this thread's own (row, column) order, save_bits)
            const unsigned long long ibits = (KIND == 3 && p.in_bits) ? p.in_bits[(int64_t)inst * TM_H + 64 * wave + c8] : 0ull;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                v4f xa[4], xb[4], ma[4], mb[4];
                float sc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = wave + 8 * (4 * half + j);
                    const int ts = t < Lrows ? t : 0;
                    const float *xr = x + (xbc ? 0 : (int64_t)ts * TM_H) + 8 * c8;
                    xa[j] = tm_ld<NT>(xr);
                    xb[j] = tm_ld<NT>(xr + 4);
                    sc[j] = rsrow ? rsrow[ts] : 1.0f;
                    if (imask) {
                        ma[j] = tm_ld<NT>(imask + (int64_t)ts * TM_H + 8 * c8);
                        mb[j] = tm_ld<NT>(imask + (int64_t)ts * TM_H + 8 * c8 + 4);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = wave + 8 * (4 * half + j);
                    v4f a = xa[j], b = xb[j];
                    if (rsrow) { a *= sc[j]; b *= sc[j]; }
                    if (imask) {              // backward chains: the incoming gradient times relu'(saved activation) (x in_scale)
#pragma unroll
                        for (int i = 0; i < 4; ++i) { a[i] = ma[j][i] > 0.f ? a[i] * p.in_scale : 0.f; b[i] = mb[j][i] > 0.f ? b[i] * p.in_scale : 0.f; }
                    } else if (KIND == 3 && p.in_bits) {
                        const unsigned m = (unsigned)(ibits >> (8 * (4 * half + j))) & 0xffu;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { a[i] = (m >> i) & 1u ? a[i] * p.in_scale : 0.f; b[i] = (m >> (4 + i)) & 1u ? b[i] * p.in_scale : 0.f; }
                    }
                    if (t >= Lrows) { a = v4f{0.f, 0.f, 0.f, 0.f}; b = a; }
                    xa[j] = a; xb[j] = b;
                    bf16x8 hi, lo;
                    tm_split8(a, b, hi, lo);
                    const int off = (c8 >> 2) * TM_STAGE + (t * 4 + ((c8 & 3) ^ tm_swz(t))) * 16;
                    *reinterpret_cast<bf16x8 *>(lds + off) = hi;
                    *reinterpret_cast<bf16x8 *>(lds + off + 4096) = lo;
                }
                if (KIND == 3 && p.save_in) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int t = wave + 8 * (4 * half + j);
                        if (t < T) {
                            float *d = p.save_in + ((int64_t)inst * Ts + t) * TM_H + 8 * c8;
                            tm_st<NT>(d, xa[j]); tm_st<NT>(d + 4, xb[j]);
                        }
                    }
                }
            }
        }
        TM_SYNC();

        f32x16 acc[2][2];
        for (int ph = 0; ph < p.n_layers; ++ph) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
            // ---- Z^T[n][t] += W[n][k] tile[t][k]: this wave's 64 weight rows straight from HBM / L2 in fragment order ----
            // (a vector-level module's first layer: once per H-wide block of the concatenated input, into the same accumulators)
            for (int seg = 0; seg < (ph == 0 ? nseg : 1); ++seg) {
                if (seg > 0) {
                    TM_SYNC();              // every wave has finished reading the previous block's image
                    build_vec_image(seg);
                    TM_SYNC();
                }
                const bf16x8 *wq = static_cast<const bf16x8 *>(p.W[ph]) + (int64_t)seg * (TM_H * TM_H * 2 / 8) + (int64_t)(2 * wave) * TM_KS * 2 * 64 + lane;
                bf16x8 wf[4][2][2];
#define TM_LOADW(slot_, ks_)                                                                                  \
    _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_)                                                       \
        _Pragma("unroll") for (int pl_ = 0; pl_ < 2; ++pl_)                                                   \
            wf[slot_][nt_][pl_] = wq[((nt_ * TM_KS + (ks_)) * 2 + pl_) * 64];
                TM_LOADW(0, 0) TM_LOADW(1, 1) TM_LOADW(2, 2)
                // tile fragments are read one k step ahead as well (the LDS round trip hides behind the previous step's MFMAs)
                bf16x8 zh[2][2], zl[2][2];
#define TM_LOADZ(set_, ks_)                                                                                   \
    {                                                                                                         \
        const char *sb_ = lds + ((ks_) >> 1) * TM_STAGE;                                                      \
        const int cx_ = (2 * ((ks_) & 1) + h) << 4;                                                           \
        _Pragma("unroll") for (int tt_ = 0; tt_ < 2; ++tt_) {                                                 \
            zh[set_][tt_] = *reinterpret_cast<const bf16x8 *>(sb_ + (offT[tt_] ^ cx_));                       \
            zl[set_][tt_] = *reinterpret_cast<const bf16x8 *>(sb_ + 4096 + (offT[tt_] ^ cx_));                \
        }                                                                                                     \
    }
                TM_LOADZ(0, 0)
                for (int ks0 = 0; ks0 < TM_KS; ks0 += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ks = ks0 + u;
                        if (ks + 3 < TM_KS) { TM_LOADW((u + 3) & 3, ks + 3) }
                        if (ks + 1 < TM_KS) { TM_LOADZ((u + 1) & 1, ks + 1) }
                        // the loads stay ABOVE this step's MFMAs: hipcc otherwise sinks each weight load to just before its use three
                        // steps later (fewer live registers), and every k step then waits out an L2 round trip
                        __builtin_amdgcn_sched_barrier(0);
                        // product-major order: four independent accumulators between two MFMAs on the same one
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int tt = 0; tt < 2; ++tt)
                                acc[nt][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u][nt][1], zh[u & 1][tt], acc[nt][tt], 0, 0, 0);
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int tt = 0; tt < 2; ++tt)
                                acc[nt][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u][nt][0], zl[u & 1][tt], acc[nt][tt], 0, 0, 0);
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int tt = 0; tt < 2; ++tt)
                                acc[nt][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u][nt][0], zh[u & 1][tt], acc[nt][tt], 0, 0, 0);
                    }
                }
#undef TM_LOADZ
#undef TM_LOADW
            }
            // ---- bias + ReLU in the accumulator layout: lane (r, h) holds frame t = 32 tt + r, columns
            //      n = 64 wave + 32 nt + 8 q + 4 h + i for e = 4 q + i -- then the tile goes to LDS as fp32 rows ---------------
            const int lane_e = tm_fresh_v(lane), r_e = lane_e & 31, h_e = lane_e >> 5;     // offsets of this stage are computed HERE, not before the k loop
            const float *bias = p.bias[ph];
            const int act = (KIND == 3 || p.act[ph] != 3) ? p.act[ph] : 0;          // act 3 (relu' of a saved activation) is the chains' (KIND 3)
            const bool last = ph + 1 == p.n_layers;
            const unsigned dsite = (KIND == 0 && pp.drop_thresh) ? p.drop_site[ph] : 0u;
            // between two layers of an inference plan nothing touches HBM: the accumulators go straight into the next image
            const bool direct = !last && !p.save[ph] && !(KIND == 0 && p.save_bits[ph]) && act != 3 && !(KIND == 0 && p.mid_rowdot && ph == 1);
            v4f bvs[2][4];                        // the lane's 32 bias values, loaded together (one round trip, behind the barrier)
            __builtin_amdgcn_sched_barrier(0);    // ... and not earlier: inside the k loop they would cost 32 registers
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    bvs[nt][q] = bias ? *reinterpret_cast<const v4f *>(bias + 64 * wave + 32 * nt + 8 * q + 4 * h_e) : v4f{0.f, 0.f, 0.f, 0.f};
            TM_SYNC();                      // every wave has finished reading the tile image of this layer
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4f bv = bvs[nt][q];
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int t = 32 * tt + r_e;
                        v4f z;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            z[i] = acc[nt][tt][4 * q + i] + bv[i];
                            if (act == 1) z[i] = fmaxf(z[i], 0.0f);
                        }
                        if (KIND == 0 && dsite) {       // nn.Dropout behind this activation: the bits stair_dropout_fwd draws for the [cnt, T, H] rows
                            const unsigned long long e0 = ((unsigned long long)inst * p.T + t) * p.H + (64 * wave + 32 * nt + 8 * q + 4 * h_e);
                            const unsigned long long bits4 = drop_hash4(pp.drop_seed, dsite - 1u, e0 >> 2);      // e0 % 4 == 0 (H % 4 == 0)
#pragma unroll
                            for (int i = 0; i < 4; ++i) z[i] = drop_keep(bits4, i, pp.drop_thresh) ? z[i] * pp.drop_inv_keep : 0.0f;
                        }
                        if (direct) {             // 4 consecutive columns of frame t = 8 bytes of the next operand row
                            bf16x4 zh4, zl4;
#pragma unroll
                            for (int i = 0; i < 4; ++i) { zh4[i] = (__bf16)z[i]; zl4[i] = (__bf16)(z[i] - (float)zh4[i]); }
                            const int off = (2 * wave + nt) * TM_STAGE + (t * 4 + (q ^ tm_swz(t))) * 16 + 8 * h_e;
                            *reinterpret_cast<bf16x4 *>(lds + off) = zh4;
                            *reinterpret_cast<bf16x4 *>(lds + off + 4096) = zl4;
                        } else {
                            *reinterpret_cast<v4f *>(F + t * TM_FLD + 64 * wave + 32 * nt + 8 * q + 4 * h_e) = z;
                        }
                    }
                }
            TM_SYNC();
            if (!last && !direct) {
                // ---- between two layers, ROW-wise (a wave per row, a lane_e per 8 consecutive columns: every global access is a
                //      contiguous 2 KB row): relu'(saved activation) of a backward chain, the activation a backward pass will
                //      need, FilterFrame's attention; then the rows become the next layer's bf16 hi / lo operand image.
                //      The rows are read into registers first: the image overlays the staging. -----------------------------------
                float *sv = tm_fresh(p.save[ph]);
                const float *amask = KIND == 3 && act == 3 && p.act_mask[ph] ? tm_fresh(p.act_mask[ph] + (int64_t)inst * Ts * TM_H) : nullptr;
                const unsigned long long *abits_p = KIND == 3 && act == 3 && !amask ? p.act_bits[ph] : nullptr;
                const unsigned long long abits = abits_p ? abits_p[(int64_t)inst * TM_H + 64 * wave + lane_e] : 0ull;
                unsigned long long *svbits = KIND == 0 ? p.save_bits[ph] : nullptr;
                unsigned long long obits = 0ull;
                const bool rowdot = KIND == 0 && p.mid_rowdot && ph == 1;
                v4f rowv[8][2];
                // every global load of the stage first (the relu' masks of all eight rows, the row-dot weights): interleaved with
                // the saves they ran one memory round trip per row
                v4f w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
                float rd_b = 0.f;
                if (rowdot) {
                    w0 = *reinterpret_cast<const v4f *>(p.vw + 8 * lane_e); w1 = *reinterpret_cast<const v4f *>(p.vw + 8 * lane_e + 4);
                    rd_b = p.vb[0] + (p.extra ? p.extra[inst] : 0.f);
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    v4f mk[4][2];
                    __builtin_amdgcn_sched_barrier(0);            // the loads stay here: hoisted above the epilogue they cost registers there
                    if (amask) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int t = wave + 8 * (4 * half + jj), ts = t < T ? t : 0;
                            mk[jj][0] = tm_ld<NT>(amask + (int64_t)ts * TM_H + 8 * lane_e);
                            mk[jj][1] = tm_ld<NT>(amask + (int64_t)ts * TM_H + 8 * lane_e + 4);
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = 4 * half + jj, t = wave + 8 * j;
                        v4f a = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 8 * lane_e), b = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 8 * lane_e + 4);
                        if (t < T) {
                            if (amask) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) { a[i] = mk[jj][0][i] > 0.f ? a[i] * p.act_scale : 0.f; b[i] = mk[jj][1][i] > 0.f ? b[i] * p.act_scale : 0.f; }
                            } else if (abits_p) {
                                const unsigned m = (unsigned)(abits >> (8 * j)) & 0xffu;
#pragma unroll
                                for (int i = 0; i < 4; ++i) { a[i] = (m >> i) & 1u ? a[i] * p.act_scale : 0.f; b[i] = (m >> (4 + i)) & 1u ? b[i] * p.act_scale : 0.f; }
                            }
                            if (svbits) {         // relu' of this activation for the backward chain: one bit per element
                                unsigned m = 0;
#pragma unroll
                                for (int i = 0; i < 4; ++i) m |= (a[i] > 0.f ? 1u << i : 0u) | (b[i] > 0.f ? 1u << (4 + i) : 0u);
                                obits |= (unsigned long long)m << (8 * j);
                            }
                            if (sv) {
                                float *d = sv + ((int64_t)inst * Ts + t) * TM_H + 8 * lane_e;
                                tm_st<NT>(d, a); tm_st<NT>(d + 4, b);
                            }
                            if (rowdot) {         // FilterFrame: a_t = sigmoid(w[:H] . f_t + extra + b); the next layer runs on a_t f_t
                                float d = 0.f;
#pragma unroll
                                for (int i = 0; i < 4; ++i) d += a[i] * w0[i] + b[i] * w1[i];
                                const float at = sigmoid_acc(wave_sum(d) + rd_b);
                                if (p.rs_out && lane_e == 0) p.rs_out[(int64_t)inst * T + t] = at;
                                a *= at; b *= at;
                            }
                        }
                        rowv[j][0] = a; rowv[j][1] = b;
                    }
                }
                if (svbits) svbits[(int64_t)inst * TM_H + 64 * wave + lane_e] = obits;
                TM_SYNC();                  // every row is in registers: the staging may be overwritten
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = wave + 8 * j;
                    bf16x8 hi, lo;
                    tm_split8(rowv[j][0], rowv[j][1], hi, lo);
                    const int off = (lane_e >> 2) * TM_STAGE + (t * 4 + ((lane_e & 3) ^ tm_swz(t))) * 16;
                    *reinterpret_cast<bf16x8 *>(lds + off) = hi;
                    *reinterpret_cast<bf16x8 *>(lds + off + 4096) = lo;
                }
                TM_SYNC();
            }
        }

        // ---- tails, from the staged rows ---------------------------------------------------------------------------------
        float *svl = p.save[p.n_layers - 1];
        if (svl)                                      // the last layer's rows for the backward pass (coalesced 2 KB rows)
            for (int t = wave; t < T; t += 8) {
                float *dst = svl + ((int64_t)inst * Ts + t) * TM_H;
                tm_st<NT>(dst + 4 * lane, *reinterpret_cast<const v4f *>(F + t * TM_FLD + 4 * lane));
                tm_st<NT>(dst + 256 + 4 * lane, *reinterpret_cast<const v4f *>(F + t * TM_FLD + 256 + 4 * lane));
            }
        if (KIND == 0 && p.save_bits[p.n_layers - 1]) {     // relu' of the last layer's rows, in the order the chain's input stage reads them
            unsigned long long obits = 0ull;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = wave + 8 * j;
                const v4f a = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 8 * lane), b = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 8 * lane + 4);
                unsigned m = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) m |= (a[i] > 0.f ? 1u << i : 0u) | (b[i] > 0.f ? 1u << (4 + i) : 0u);
                obits |= (unsigned long long)m << (8 * j);
            }
            p.save_bits[p.n_layers - 1][(int64_t)inst * TM_H + 64 * wave + lane] = obits;
        }
        const int64_t oslot = p.out_idx ? __builtin_amdgcn_readfirstlane(p.out_idx[inst]) : inst;
        if (KIND == 2) {
            // tail ROWSCALE_ADJ: the staged rows are G = dZ W, the gradient of the SCALED input r_t feat_t of the dense layer:
            // d feat_t += r_t G_t (one dword per lane, 256 contiguous bytes per wave-instruction), d r_t += G_t . feat_t
            const float *feat = tm_fresh(p.adj_feat + (int64_t)(p.adj_feat_idx ? __builtin_amdgcn_readfirstlane(p.adj_feat_idx[inst]) : inst) * p.adj_feat_gstride);
            const int64_t rslot = p.adj_rs_idx ? __builtin_amdgcn_readfirstlane(p.adj_rs_idx[inst]) : inst;
            const float *rs = p.adj_rs + rslot * T;
            float *drs = p.adj_drs + rslot * T;
            float *dst = tm_fresh(p.out + oslot * p.out_gstride);
            __builtin_amdgcn_sched_barrier(0);      // the loads stay below the epilogue (hoisted, they cost registers there)
            v4f f0[8], f1[8];
            float rr[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {           // every global load of the tail first
                const int t = wave + 8 * j, ts = t < T ? t : 0;
                f0[j] = tm_ld<NT>(feat + (int64_t)ts * TM_H + 4 * lane);
                f1[j] = tm_ld<NT>(feat + (int64_t)ts * TM_H + 256 + 4 * lane);
                rr[j] = rs[ts];
            }
            if (p.acc_exclusive) {                  // read - add - write instead of atomics (see ACCUMULATE); a branch of its own:
                v4f o0[8], o1[8];                   // a condition inside the unrolled loops made every load wait and spill
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = wave + 8 * j, ts = t < T ? t : 0;
                    o0[j] = *reinterpret_cast<const v4f *>(dst + (int64_t)ts * TM_H + 4 * lane);
                    o1[j] = *reinterpret_cast<const v4f *>(dst + (int64_t)ts * TM_H + 256 + 4 * lane);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = wave + 8 * j;
                    if (t < T) {
                        const v4f g0 = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 4 * lane), g1 = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 256 + 4 * lane);
                        float d = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) d += g0[i] * f0[j][i] + g1[i] * f1[j][i];
                        d = wave_sum(d);
                        if (lane == 0) unsafeAtomicAdd(drs + t, d);
                        *reinterpret_cast<v4f *>(dst + (int64_t)t * TM_H + 4 * lane) = o0[j] + rr[j] * g0;
                        *reinterpret_cast<v4f *>(dst + (int64_t)t * TM_H + 256 + 4 * lane) = o1[j] + rr[j] * g1;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = wave + 8 * j;
                    if (t < T) {
                        const v4f g0 = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 4 * lane), g1 = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 256 + 4 * lane);
                        float d = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) d += g0[i] * f0[j][i] + g1[i] * f1[j][i];
                        d = wave_sum(d);
                        if (lane == 0) unsafeAtomicAdd(drs + t, d);
#pragma unroll
                        for (int c = 0; c < TM_H / 64; ++c) unsafeAtomicAdd(dst + (int64_t)t * TM_H + 64 * c + lane, rr[j] * F[t * TM_FLD + 64 * c + lane]);
                    }
                }
            }
        } else
        switch (KIND == 3 && p.tail != STAIR_TILE_ACCUMULATE ? (p.tail == STAIR_TILE_STORE ? STAIR_TILE_STORE : STAIR_TILE_NONE)
                                                             : (KIND != 3 && p.tail == STAIR_TILE_ACCUMULATE ? STAIR_TILE_NONE : p.tail)) {
            case STAIR_TILE_STORE:
                for (int t = wave; t < T; t += 8) {
                    float *dst = p.out + oslot * p.out_gstride + (int64_t)t * TM_H;
                    tm_st<NT>(dst + 4 * lane, *reinterpret_cast<const v4f *>(F + t * TM_FLD + 4 * lane));
                    tm_st<NT>(dst + 256 + 4 * lane, *reinterpret_cast<const v4f *>(F + t * TM_FLD + 256 + 4 * lane));
                }
                break;
            case STAIR_TILE_STORE_ROWS:               // vector-level modules: row t = instance 64 inst + t goes to its own slot
                if (KIND != 1) break;
                for (int t = wave; t < T; t += 8) {
                    float *dst = p.out + (int64_t)p.out_row_idx[(int64_t)TM_ROWS * inst + t] * p.out_gstride;
                    *reinterpret_cast<v4f *>(dst + 4 * lane) = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 4 * lane);
                    *reinterpret_cast<v4f *>(dst + 256 + 4 * lane) = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 256 + 4 * lane);
                }
                break;
            case STAIR_TILE_ACCUMULATE:               // backward chains: dX added into a gradient tile several instances may share
                if (KIND != 3) break;
                // one dword per lane, 256 contiguous bytes per wave-instruction: the shape float atomics run at full rate in
                // (MI355X_MICROARCH.md "Global float atomics"; 16-byte-strided lanes spread an instruction over 1 KB)
                if (p.acc_exclusive) {
                    // no other instance of this launch adds into this tile (the caller's promise; stair_plan_backward's fan-in staging
                    // gives every same-level reader of a slot a target of its own): whole rows read, added and written back, every
                    // load of the tile's rows in flight at once -- float atomics retire 256 B per ~50 ns and CU (1.3 TB/s chip-wide,
                    // MI355X_MICROARCH.md "Global float atomics"): 25 us per tile, as long as two layers of the chain
                    float *dst = tm_fresh(p.out + oslot * p.out_gstride);
                    __builtin_amdgcn_sched_barrier(0);          // the loads stay below the epilogue (hoisted, they cost registers there)
                    v4f o0[8], o1[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int t = wave + 8 * j, ts = t < T ? t : 0;
                        o0[j] = *reinterpret_cast<const v4f *>(dst + (int64_t)ts * TM_H + 4 * lane);
                        o1[j] = *reinterpret_cast<const v4f *>(dst + (int64_t)ts * TM_H + 256 + 4 * lane);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int t = wave + 8 * j;
                        if (t < T) {
                            *reinterpret_cast<v4f *>(dst + (int64_t)t * TM_H + 4 * lane) = o0[j] + *reinterpret_cast<const v4f *>(F + t * TM_FLD + 4 * lane);
                            *reinterpret_cast<v4f *>(dst + (int64_t)t * TM_H + 256 + 4 * lane) = o1[j] + *reinterpret_cast<const v4f *>(F + t * TM_FLD + 256 + 4 * lane);
                        }
                    }
                    break;
                }
                for (int t = wave; t < T; t += 8) {
                    float *dst = p.out + oslot * p.out_gstride + (int64_t)t * TM_H;
#pragma unroll
                    for (int c = 0; c < TM_H / 64; ++c) unsafeAtomicAdd(dst + 64 * c + lane, F[t * TM_FLD + 64 * c + lane]);
                }
                break;
            case STAIR_TILE_SUM_ROWS: {               // Filter: sum over the clip's own frames (modules.py:374,376)
                if (KIND == 3) break;
                const int L = p.len ? __builtin_amdgcn_readfirstlane(p.len[inst]) : T;
                const int col = tm_fresh_v(TM_TID());       // (not hoisted out of the tile loop: three values kept that way went to scratch)
                float s = 0.f;
                for (int t = 0; t < L; ++t) s += F[t * TM_FLD + col];
                p.out[oslot * p.out_gstride + col] = s;
                break;
            }
            case STAIR_TILE_COSINE: {                 // Localize: (cos(f_t, k_j) + 1) * 0.49, nn.CosineSimilarity eps 1e-8
                if (KIND == 3) break;
                const int first = __builtin_amdgcn_readfirstlane(p.pair_first[inst]), cn = __builtin_amdgcn_readfirstlane(p.pair_cnt[inst]);
                // pair-major: a keyword row (and its norm, and its attention slot) is fetched once, not once per frame
                for (int j = 0; j < cn; ++j) {
                    const float *k = p.kb + (int64_t)(first + j) * TM_H;
                    const v4f k0 = *reinterpret_cast<const v4f *>(k + 4 * lane), k1 = *reinterpret_cast<const v4f *>(k + 256 + 4 * lane);
                    float *arow = p.att + (int64_t)p.att_idx[first + j] * T;
                    float nk = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) nk += k0[i] * k0[i] + k1[i] * k1[i];
                    nk = wave_sum(nk);
                    for (int t = wave; t < T; t += 8) {
                        const v4f f0 = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 4 * lane), f1 = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 256 + 4 * lane);
                        float nf = 0.f, d = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { nf += f0[i] * f0[i] + f1[i] * f1[i]; d += f0[i] * k0[i] + f1[i] * k1[i]; }
                        nf = wave_sum(nf); d = wave_sum(d);
                        if (lane == 0) {
                            const float eps = 1e-8f;
                            const float c = d / (fmaxf(sqrtf(nf), eps) * fmaxf(sqrtf(nk), eps));
                            arow[t] = (c + 1.0f) * 0.49f;
                        }
                    }
                }
                break;
            }
            case STAIR_TILE_ROWDOT_SIGMOID: {         // HasItem: sigmoid(w . row + b) (modules.py:131-137)
                if (KIND == 3) break;
                const v4f w0 = *reinterpret_cast<const v4f *>(p.vw + 4 * lane), w1 = *reinterpret_cast<const v4f *>(p.vw + 256 + 4 * lane);
                const float off = p.vb[0] + (p.extra ? p.extra[inst] : 0.f);
                for (int t = wave; t < T; t += 8) {
                    const v4f f0 = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 4 * lane), f1 = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 256 + 4 * lane);
                    float d = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) d += f0[i] * w0[i] + f1[i] * w1[i];
                    d = wave_sum(d);
                    if (lane == 0) p.out[oslot * p.out_gstride + t] = sigmoid_acc(d + off);
                }
                break;
            }
            case STAIR_TILE_LAYERNORM: {              // Temporal: LayerNorm over H, eps 1e-5, biased variance (modules.py:283,327)
                if (KIND == 3) break;
                const v4f g0 = *reinterpret_cast<const v4f *>(p.gamma + 4 * lane), g1 = *reinterpret_cast<const v4f *>(p.gamma + 256 + 4 * lane);
                const v4f b0 = *reinterpret_cast<const v4f *>(p.beta + 4 * lane), b1 = *reinterpret_cast<const v4f *>(p.beta + 256 + 4 * lane);
                for (int t = wave; t < T; t += 8) {
                    v4f f0 = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 4 * lane), f1 = *reinterpret_cast<const v4f *>(F + t * TM_FLD + 256 + 4 * lane);
                    float sum = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) sum += f0[i] + f1[i];
                    const float mean = wave_sum(sum) / (float)TM_H;
                    float sq = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float a = f0[i] - mean, b = f1[i] - mean; sq += a * a + b * b; }
                    const float rstd = rsqrtf(wave_sum(sq) / (float)TM_H + p.ln_eps);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { f0[i] = (f0[i] - mean) * rstd * g0[i] + b0[i]; f1[i] = (f1[i] - mean) * rstd * g1[i] + b1[i]; }
                    float *dst = p.out + oslot * p.out_gstride + (int64_t)t * TM_H;
                    *reinterpret_cast<v4f *>(dst + 4 * lane) = f0;
                    *reinterpret_cast<v4f *>(dst + 256 + 4 * lane) = f1;
                }
                break;
            }
            default: break;
        }
    }
    // The queue resets ITSELF: every workgroup takes exactly one ticket >= total (the one that made it leave) and then signs off;
    // the last one to sign off knows that nobody will touch the head again and zeroes both words, so the next launch on the stream
    // -- eager, or the same kernel node of a replayed hipGraph -- finds them at 0 without any memset in between.
    // (thread 0's last head ticket has RETURNED before it signs off -- it branched on the value -- so no fence is needed between the two)
    const int tid = TM_TID();
    if (KIND == 2) {        // column tid of d gamma / d beta: written and read by this thread only, no barrier needed
        const long long g = tb_acc[tid], b = tb_acc[TM_H + tid];
        if (pp.fx_g) { if (g) atomicAdd(reinterpret_cast<unsigned long long *>(pp.fx_g + tid), (unsigned long long)g); }
        else if (g) unsafeAtomicAdd(pp.a[0].dgamma + tid, __ll2float_rn(g) * (1.0f / kFxScale));
        if (pp.fx_b) { if (b) atomicAdd(reinterpret_cast<unsigned long long *>(pp.fx_b + tid), (unsigned long long)b); }
        else if (b) unsafeAtomicAdd(pp.a[0].dbeta + tid, __ll2float_rn(b) * (1.0f / kFxScale));
    }
    if (pp.counter && tid == 0) {
        if (atomicAdd(pp.counter + 1, 1u) == gridDim.x - 1) {
            atomicExch(pp.counter, 0u);
            atomicExch(pp.counter + 1, 0u);
        }
    }
}

#undef TM_TID
#undef TM_LANE

// W [N, K] fp32 row-major -> bf16 hi / lo planes in MFMA fragment order: the A operand of v_mfma_f32_32x32x16_bf16 for the
// 32-row tile nt and the 16-wide k step ks is 64 lanes x 8 bf16, lane (r, h) = W[32 nt + r][16 ks + 8 h + 0..7]; the image is
// [N/32][K/16][hi, lo][64 lanes][8].  One thread per (row, 8 columns).
struct WfragBatch { const float *w[32]; __bf16 *o[32]; int ld[32]; int count, blocks_per; };     // ld: row stride of matrix m (floats)
// TRANSPOSE: the planes of W^T (the "weight" of a backward chain, dX = dZ W): image row n <-> column n of the stored W [K, N]
template <bool TRANSPOSE>
__global__ void pack_wfrag_kernel(WfragBatch tb, int N, int K) {
    const int m = blockIdx.x / tb.blocks_per;
    const int64_t i = (int64_t)(blockIdx.x - m * tb.blocks_per) * blockDim.x + threadIdx.x;
    const int k8n = K / 8;
    if (i >= (int64_t)N * k8n) return;
    int n, k8;
    v4f a, b;
    if (!TRANSPOSE) {
        n = (int)(i / k8n); k8 = (int)(i - (int64_t)n * k8n);
        const float *src = tb.w[m] + (int64_t)n * tb.ld[m] + 8 * k8;
        a = *reinterpret_cast<const v4f *>(src); b = *reinterpret_cast<const v4f *>(src + 4);
    } else {                            // consecutive threads walk n: the strided reads of a column run coalesce across the wave
        k8 = (int)(i / N); n = (int)(i - (int64_t)k8 * N);
        const float *src = tb.w[m] + (int64_t)(8 * k8) * tb.ld[m] + n;      // stored matrix is [K, N] row-major
#define TM_TLD tb.ld[m]
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = src[(int64_t)j * TM_TLD]; b[j] = src[(int64_t)(4 + j) * TM_TLD]; }
#undef TM_TLD
    }
    bf16x8 hi, lo;
    tm_split8(a, b, hi, lo);
    const int nt = n >> 5, r = n & 31, ks = k8 >> 1, h = k8 & 1;
    __bf16 *dst = tb.o[m] + ((int64_t)(nt * (K / 16) + ks) * 2 * 64 + (r + 32 * h)) * 8;
    *reinterpret_cast<bf16x8 *>(dst) = hi;
    *reinterpret_cast<bf16x8 *>(dst + 64 * 8) = lo;
}

// up to 32 matrices of one shape in ONE launch (the tables travel as kernel arguments: no copy, no host wait)
int launch_pack_wfrag_many(const float *const *W, void *const *out, int count, int N, int K, hipStream_t s, bool transpose, const int *ld) {
    STAIR_CHECK(count >= 1 && count <= 32, "1..32 matrices per launch");
    STAIR_CHECK(N % 32 == 0 && K % 16 == 0, "N % 32 == 0 and K % 16 == 0");
    WfragBatch t;
    t.count = count;
    t.blocks_per = (int)(((int64_t)N * (K / 8) + 255) / 256);
    for (int m = 0; m < 32; ++m) {
        t.w[m] = m < count ? W[m] : nullptr;
        t.o[m] = m < count ? static_cast<__bf16 *>(out[m]) : nullptr;
        t.ld[m] = (ld && m < count) ? ld[m] : (transpose ? N : K);
        STAIR_CHECK(t.ld[m] % 4 == 0 && t.ld[m] >= (transpose ? N : K), "row stride");
        STAIR_CHECK(m >= count || (t.w[m] && t.o[m] && ((reinterpret_cast<uintptr_t>(t.w[m]) | reinterpret_cast<uintptr_t>(t.o[m])) & 15) == 0),
                    "null or unaligned matrix");
    }
    if (transpose) hipLaunchKernelGGL(pack_wfrag_kernel<true>, dim3(t.blocks_per * count), dim3(256), 0, s, t, N, K);
    else hipLaunchKernelGGL(pack_wfrag_kernel<false>, dim3(t.blocks_per * count), dim3(256), 0, s, t, N, K);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// Measurement aid (bench.py's roofline_tile_operator): HIP events around every tile launch while switched on; the sum of the
// elapsed times is the kernel's device time on its own stream, without a profiler.
static bool g_tile_timing = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_tile_events;

static int g_tile_on = -1;          // stair_set_tile_mlp; -1 = the environment's STAIR_TILE_MLP (default on)
bool tile_mlp_usable(int H, int T) {
    static const bool env_on = [] { const char *e = getenv("STAIR_TILE_MLP"); return !(e && e[0] == '0'); }();
    const bool on = policy_or(STAIR_OPT_TILE_MLP, g_tile_on >= 0 ? g_tile_on : (env_on ? 1 : 0)) != 0;
    return on && H == TM_H && T >= 1 && T <= TM_ROWS && matmul_mode() == STAIR_MATMUL_BF16X3;
}

static int tile_mlp_check(const stair_tile_mlp_args &a) {
    STAIR_CHECK(a.H == TM_H, "the fused tile operators are built for hidden_size 512");
    STAIR_CHECK(a.n_layers >= 1 && a.n_layers <= 3, "1..3 layers");
    if (a.vec_pack) {                     // vector-level modules: 64 instances per tile
        STAIR_CHECK(a.vec_pack >= 1 && a.vec_pack <= 3, "vec_pack: 1 [a, b], 2 [|a - b|, a, b], 3 [a, b, a * b]");
        STAIR_CHECK(a.pk_a && a.pk_b && a.vec_cnt >= 0 && a.cnt == (a.vec_cnt + TM_ROWS - 1) / TM_ROWS, "vector operands / cnt = ceil(vec_cnt / 64)");
        STAIR_CHECK(((reinterpret_cast<uintptr_t>(a.pk_a) | reinterpret_cast<uintptr_t>(a.pk_b) | reinterpret_cast<uintptr_t>(a.cat_save)) & 15) == 0, "vector operands must be 16-byte aligned");
        STAIR_CHECK(!a.row_scale && !a.in_mask && !a.x_broadcast && !a.save_in && !a.mid_rowdot, "vector-level tiles take no map-level input options");
        STAIR_CHECK(a.tail == STAIR_TILE_STORE_ROWS || a.tail == STAIR_TILE_NONE, "vector-level tiles end in STORE_ROWS");
        for (int l = 0; l < a.n_layers; ++l) STAIR_CHECK(a.act[l] != 3, "no backward chains on vector-level tiles");
    } else {
        STAIR_CHECK(a.T >= 1 && a.T <= TM_ROWS, "a tile holds 1..64 frames");
        STAIR_CHECK(a.X && a.cnt >= 0, "null input");
        STAIR_CHECK(a.tail != STAIR_TILE_STORE_ROWS, "STORE_ROWS is the vector-level tail");
    }
    STAIR_CHECK((a.ln_bwd != 0) == (a.tail == STAIR_TILE_ROWSCALE_ADJ), "ln_bwd and the ROWSCALE_ADJ tail come together (Temporal's backward chain)");
    if (a.ln_bwd) {
        STAIR_CHECK(!a.vec_pack && a.n_layers == 1 && a.act[0] == 0 && !a.save[0] && !a.bias[0], "Temporal's backward chain is ONE transposed layer, no bias, no activation");
        STAIR_CHECK(a.in_mask && a.gamma && a.dgamma && a.dbeta, "ln_bwd: saved pre-LayerNorm rows (in_mask), gamma, dgamma, dbeta");
        STAIR_CHECK(!a.row_scale && !a.x_broadcast && !a.mid_rowdot, "ln_bwd takes no other input option");
        STAIR_CHECK(a.adj_feat && a.adj_rs && a.adj_drs && a.out && a.adj_feat_gstride % 4 == 0 && a.out_gstride % 4 == 0 && a.in_mask_gstride % 4 == 0, "ROWSCALE_ADJ: adj_feat, adj_rs, adj_drs, out");
        STAIR_CHECK(((reinterpret_cast<uintptr_t>(a.adj_feat) | reinterpret_cast<uintptr_t>(a.in_mask) | reinterpret_cast<uintptr_t>(a.gamma) | reinterpret_cast<uintptr_t>(a.save_in)) & 15) == 0, "ln_bwd operands must be 16-byte aligned");
    }
    for (int l = 0; l < a.n_layers; ++l)
        STAIR_CHECK(a.W[l] && (reinterpret_cast<uintptr_t>(a.W[l]) & 15) == 0, "weight planes (stair_pack_wfrag) missing or unaligned");
    for (int l = 0; l < a.n_layers; ++l) STAIR_CHECK(a.act[l] != 3 || a.act_mask[l] || a.act_bits[l], "act 3 multiplies by relu'(act_mask[l]) (or its bits, act_bits[l])");
    STAIR_CHECK(!(a.in_mask && a.in_bits), "in_mask or in_bits, not both");
    {   // the backward chains are a kernel of their own: chain options do not mix with the forward operators' options
        bool chain = a.tail == STAIR_TILE_ACCUMULATE || (!a.ln_bwd && a.in_mask) || a.in_bits || a.x_broadcast || (!a.ln_bwd && a.save_in);
        for (int l = 0; l < a.n_layers; ++l) chain = chain || a.act[l] == 3;
        if (chain) {
            STAIR_CHECK(!a.vec_pack && !a.ln_bwd, "chain options (act 3, in_mask, in_bits, x_broadcast, save_in, ACCUMULATE) are map-level");
            STAIR_CHECK(a.tail == STAIR_TILE_ACCUMULATE || a.tail == STAIR_TILE_STORE || a.tail == STAIR_TILE_NONE, "a backward chain ends in ACCUMULATE, STORE or NONE");
            STAIR_CHECK(!a.row_scale && !a.mid_rowdot && !a.save_bits[0] && !a.save_bits[1] && !a.save_bits[2], "a backward chain takes no row_scale / mid_rowdot / save_bits");
        } else {
            STAIR_CHECK(!a.act_bits[0] && !a.act_bits[1] && !a.act_bits[2], "act_bits belong to act 3");
        }
    }
    STAIR_CHECK(!((a.vec_pack || a.ln_bwd) && (a.in_bits || a.save_bits[0] || a.save_bits[1] || a.save_bits[2] || a.act_bits[0] || a.act_bits[1] || a.act_bits[2])),
                "bit masks are a map-level option");
    STAIR_CHECK(!a.x_broadcast || a.x_gstride == a.H, "a broadcast input is one [H] row per instance");
    STAIR_CHECK(!a.mid_rowdot || (a.n_layers == 3 && a.vw && a.vb), "mid_rowdot is FilterFrame's attention between layers 2 and 3");
    STAIR_CHECK(a.vec_pack || (a.x_gstride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0), "input tiles must be 16-byte aligned");
    switch (a.tail) {
        case STAIR_TILE_STORE: case STAIR_TILE_SUM_ROWS: case STAIR_TILE_ACCUMULATE:
            STAIR_CHECK(a.out && a.out_gstride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0, "out missing or unaligned"); break;
        case STAIR_TILE_COSINE: STAIR_CHECK(a.kb && a.pair_first && a.pair_cnt && a.att_idx && a.att, "cosine tail: keyword rows / pair tables / att"); break;
        case STAIR_TILE_ROWDOT_SIGMOID: STAIR_CHECK(a.vw && a.vb && a.out, "row-dot tail: vw, vb, out"); break;
        case STAIR_TILE_LAYERNORM: STAIR_CHECK(a.gamma && a.beta && a.out && a.out_gstride % 4 == 0, "LayerNorm tail: gamma, beta, out"); break;
        case STAIR_TILE_STORE_ROWS: STAIR_CHECK(a.out && a.out_row_idx && a.out_gstride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0, "row-scatter tail: out, out_row_idx"); break;
        case STAIR_TILE_ROWSCALE_ADJ: break;     // checked with ln_bwd above
        case STAIR_TILE_NONE: break;
        default: STAIR_FAIL("unknown tail");
    }
    return 0;
}

// n buckets (same T) in one launch; counter: TWO zeroed device words (the dynamic work queue: head, sign-off count; the kernel
// leaves them zeroed, so launches that follow each other on one stream may share them) or NULL (static round robin)
int launch_tile_mlp_batch(const stair_tile_mlp_args *args, int n, unsigned *counter, hipStream_t s) {
    STAIR_CHECK(n >= 0 && n <= TM_MAXB, "at most 8 buckets per launch");
    STAIR_CHECK(matmul_mode() == STAIR_MATMUL_BF16X3, "the fused tile operators compute split-bf16 products (STAIR_MATMUL_BF16X3)");
    for (int i = 0; i < n; ++i)
        if (int rc = tile_mlp_check(args[i])) return rc;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    static int cus[64] = {};
    if (!attr_set[dev]) {
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_mlp_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, TM_LDS));
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_mlp_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, TM_LDS));
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_mlp_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, TM_LDS));
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_mlp_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, TM_LDS));
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_mlp_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, TM_LDS));
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_mlp_kernel<true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, TM_LDS));
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_mlp_kernel<false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, TM_LDS));
        int v = 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = v;
        attr_set[dev] = true;
    }
    // non-temporal hints on the tile's own traffic (inputs, masks, saves): with the first version of the kernel plain accesses were
    // faster (saves 273 -> 290 us per 1024 tiles); now that a layer is bound by the weight stream from L2, keeping the streamed
    // tiles from displacing the weight planes pays: 17.02 -> 16.89 ms per 2048-question step (three A/B pairs on one box),
    // neutral at 128 questions and in inference.  STAIR_TILE_NT=0 switches them off.
    static const bool nt = [] { const char *e = getenv("STAIR_TILE_NT"); return !(e && e[0] == '0'); }();
    auto rounds = [&](int x) { return args[x].n_layers + (args[x].vec_pack ? (args[x].vec_pack == 1 ? 1 : 2) : 0); };   // k loops per tile
    // map-level tiles, vector-level tiles and Temporal's backward chains are three kernels (what one form needs in registers the
    // others do not pay for): one launch per form a level has; stream-ordered, so they share the (self-resetting) queue words
    auto kind_of = [](const stair_tile_mlp_args &a) {
        if (a.vec_pack) return 1;
        if (a.ln_bwd) return 2;
        bool chain = a.tail == STAIR_TILE_ACCUMULATE || a.in_mask || a.in_bits || a.x_broadcast || a.save_in;
        for (int l = 0; l < a.n_layers && l < 3; ++l) chain = chain || a.act[l] == 3;
        return chain ? 3 : 0;
    };
    for (int kind = 0; kind < 4; ++kind) {
        const int vec = kind == 1;
        TmParams pp;
        pp.nb = 0; pp.counter = counter; pp.first[0] = 0; pp.fx_g = pp.fx_b = nullptr;
        pp.drop_thresh = 0; pp.drop_inv_keep = 1.0f; pp.drop_seed = 0;
        int order[TM_MAXB], m = 0;
        for (int i = 0; i < n; ++i)
            if (args[i].cnt > 0 && kind_of(args[i]) == kind) order[m++] = i;
        if (m == 0) continue;
        if (kind == 2) {            // one accumulator pair per workgroup: the chains of a launch share their LayerNorm
            for (int j = 1; j < m; ++j)
                STAIR_CHECK(args[order[j]].dgamma == args[order[0]].dgamma && args[order[j]].dbeta == args[order[0]].dbeta,
                            "the Temporal chains of one launch must share dgamma / dbeta");
            pp.fx_g = det_shadow(args[order[0]].dgamma); pp.fx_b = det_shadow(args[order[0]].dbeta);
        }
        if (kind == 0) {            // dropout behind the activations of a forward launch: one probability and seed per launch
            for (int j = 0; j < m; ++j) {
                const stair_tile_mlp_args &a = args[order[j]];
                if (!(a.drop_p > 0.0f) || !(a.drop_site[0] | a.drop_site[1] | a.drop_site[2])) continue;
                STAIR_CHECK(a.drop_p < 1.0f, "dropout probability must be below 1");
                STAIR_CHECK(a.drop_site[0] < 65536u && a.drop_site[1] < 65536u && a.drop_site[2] < 65536u, "dropout site above 65534");
                const unsigned th = (unsigned)(a.drop_p * 65536.0f);
                STAIR_CHECK(pp.drop_thresh == 0 || (pp.drop_thresh == th && pp.drop_seed == a.drop_seed), "the buckets of one launch must share dropout probability and seed");
                pp.drop_thresh = th; pp.drop_inv_keep = 1.0f / (1.0f - a.drop_p); pp.drop_seed = a.drop_seed;
            }
        }
        std::stable_sort(order, order + m, [&](int x, int y) { return rounds(x) > rounds(y); });    // long tiles first
        for (int j = 0; j < m; ++j) {
            const stair_tile_mlp_args &a = args[order[j]];
            pp.a[j] = tm_arg(a, kind);
            pp.first[j + 1] = pp.first[j] + a.cnt;
            const int64_t M = a.vec_pack ? a.vec_cnt : (int64_t)a.cnt * a.T;
            const int kl = rounds(order[j]);
            STAIR_ACCT_MFMA("tile_mlp", (M * TM_H * 2 + (int64_t)kl * TM_H * TM_H) * 4, 2 * M * TM_H * TM_H * kl);
        }
        for (int j = m; j < TM_MAXB; ++j) { pp.a[j] = pp.a[0]; pp.first[j + 1] = pp.first[m]; }
        pp.nb = m;
        const int grid = std::min(pp.first[m], cus[dev]);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (g_tile_timing) {
            STAIR_HIP(hipEventCreate(&e0)); STAIR_HIP(hipEventCreate(&e1));
            STAIR_HIP(hipEventRecord(e0, s));
        }
        if (vec) hipLaunchKernelGGL((tile_mlp_kernel<false, 1>), dim3(grid), dim3(512), TM_LDS, s, pp);
        else if (kind == 2 && nt) hipLaunchKernelGGL((tile_mlp_kernel<true, 2>), dim3(grid), dim3(512), TM_LDS, s, pp);
        else if (kind == 2) hipLaunchKernelGGL((tile_mlp_kernel<false, 2>), dim3(grid), dim3(512), TM_LDS, s, pp);
        else if (kind == 3 && nt) hipLaunchKernelGGL((tile_mlp_kernel<true, 3>), dim3(grid), dim3(512), TM_LDS, s, pp);
        else if (kind == 3) hipLaunchKernelGGL((tile_mlp_kernel<false, 3>), dim3(grid), dim3(512), TM_LDS, s, pp);
        else if (nt) hipLaunchKernelGGL((tile_mlp_kernel<true, 0>), dim3(grid), dim3(512), TM_LDS, s, pp);
        else hipLaunchKernelGGL((tile_mlp_kernel<false, 0>), dim3(grid), dim3(512), TM_LDS, s, pp);
        STAIR_LAUNCH_CHECK();
        if (g_tile_timing) {
            STAIR_HIP(hipEventRecord(e1, s));
            g_tile_events.emplace_back(e0, e1);
        }
    }
    return 0;
}

int launch_tile_mlp(const stair_tile_mlp_args &a, hipStream_t s) { return launch_tile_mlp_batch(&a, 1, nullptr, s); }

}  // namespace stair

extern "C" int stair_set_tile_mlp(int32_t on) { stair::g_tile_on = on; return 0; }

extern "C" int stair_tile_mlp_fwd(const stair_tile_mlp_args *args, stair_stream stream) {
    if (!args) {
        stair::set_error("stair_tile_mlp_fwd: null args");
        return 1;
    }
    return stair::launch_tile_mlp(*args, static_cast<hipStream_t>(stream));
}

extern "C" int stair_pack_wfrag(const float *W, void *planes, int32_t N, int32_t K, int32_t transpose, stair_stream stream) {
    if (!W || !planes) {
        stair::set_error("stair_pack_wfrag: null argument");
        return 1;
    }
    const float *w[1] = {W};
    void *o[1] = {planes};
    return stair::launch_pack_wfrag_many(w, o, 1, N, K, static_cast<hipStream_t>(stream), transpose != 0);
}

extern "C" int stair_pack_wfrag_ld(const float *W, int64_t ld, void *planes, int32_t N, int32_t K, stair_stream stream) {
    if (!W || !planes) {
        stair::set_error("stair_pack_wfrag_ld: null argument");
        return 1;
    }
    const float *w[1] = {W};
    void *o[1] = {planes};
    const int l[1] = {(int)ld};
    return stair::launch_pack_wfrag_many(w, o, 1, N, K, static_cast<hipStream_t>(stream), false, l);
}

extern "C" int stair_tile_timing(int32_t on) {
    for (auto &e : stair::g_tile_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    stair::g_tile_events.clear();
    stair::g_tile_timing = on != 0;
    return 0;
}

extern "C" int stair_tile_timing_read(double *ms, int32_t *launches) {
    double total = 0.0;
    for (auto &e : stair::g_tile_events) {
        STAIR_HIP(hipEventSynchronize(e.second));
        float t = 0.0f;
        STAIR_HIP(hipEventElapsedTime(&t, e.first, e.second));
        total += t;
    }
    if (ms) *ms = total;
    if (launches) *launches = (int32_t)stair::g_tile_events.size();
    return 0;
}

