// Backward (reverse-mode) counterparts of csrc/rowops.hip: what autograd does for the reference's
// small operators when train_module.py:408 calls batch_loss.backward().  Same mapping rules as the
// forward kernels (one wave per row for row reductions, one block per instance where an instance
// needs LDS).  Gradient arenas mirror the value arenas slot for slot; a slot can have several
// consumers (the encoded video feeds every module of a question), so contributions into arenas are
// fp32 atomic adds (order-dependent in the last bits, as any parallel reduction is).
#include "ops.h"

namespace stair {

using v4f = __attribute__((ext_vector_type(4))) float;

namespace {
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;
__device__ __forceinline__ int idx_or_id(const int32_t *idx, int i) { return idx ? idx[i] : i; }
__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
}  // namespace

// dst[g][:] = G[gi][:] * (Y[yi][:] > 0)      (ReLU backward with optional gathers; rowlen floats per group)
__global__ void mask_relu_kernel(float *dst, const float *G, int64_t g_gs, const int32_t *g_idx, const float *Y,
                                 int64_t y_gs, const int32_t *y_idx, int groups, int rowlen, float scale) {
    const int64_t total = (int64_t)groups * rowlen;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(e / rowlen);
        const int64_t r = e - (int64_t)g * rowlen;
        const float gv = G[(int64_t)idx_or_id(g_idx, g) * g_gs + r];
        const float yv = Y[(int64_t)idx_or_id(y_idx, g) * y_gs + r];
        dst[e] = yv > 0.f ? gv * scale : 0.f;      // scale = 1/(1-p) where a dropout follows the ReLU (Y is the dropped tensor)
    }
}
// float4 form: blockIdx.y = group, threads walk the row in 16-byte pieces (no per-element division, 16-byte accesses)
__global__ void mask_relu_v4_kernel(float *dst, const float *G, int64_t g_gs, const int32_t *g_idx, const float *Y,
                                    int64_t y_gs, const int32_t *y_idx, int rowlen4, float scale) {
    const int g = blockIdx.y;
    const v4f *gp = reinterpret_cast<const v4f *>(G + (int64_t)idx_or_id(g_idx, g) * g_gs);
    const v4f *yp = reinterpret_cast<const v4f *>(Y + (int64_t)idx_or_id(y_idx, g) * y_gs);
    v4f *dp = reinterpret_cast<v4f *>(dst + (int64_t)g * rowlen4 * 4);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < rowlen4; e += gridDim.x * blockDim.x) {
        const v4f gv = gp[e], yv = yp[e];
        v4f o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = yv[j] > 0.f ? gv[j] * scale : 0.f;
        dp[e] = o;
    }
}
int launch_mask_relu(float *dst, const float *G, int64_t g_gs, const int32_t *g_idx, const float *Y, int64_t y_gs,
                     const int32_t *y_idx, int groups, int rowlen, hipStream_t s, float scale) {
    if (groups == 0) return 0;
    STAIR_ACCT("mask_relu_kernel", 3ll * groups * rowlen * 4);
    const bool al = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(Y)) & 15) == 0;
    if (al && rowlen % 4 == 0 && g_gs % 4 == 0 && y_gs % 4 == 0 && groups <= 65535) {
        const int r4 = rowlen / 4;
        const int bx = std::max(1, std::min((r4 + kBlock - 1) / kBlock, std::max(1, 4096 / groups)));
        hipLaunchKernelGGL(mask_relu_v4_kernel, dim3(bx, groups), dim3(kBlock), 0, s, dst, G, g_gs, g_idx, Y, y_gs, y_idx, r4, scale);
        STAIR_LAUNCH_CHECK();
        return 0;
    }
    const int64_t total = (int64_t)groups * rowlen;
    hipLaunchKernelGGL(mask_relu_kernel, dim3((unsigned)std::min<int64_t>((total + kBlock - 1) / kBlock, 4096)), dim3(kBlock), 0,
                       s, dst, G, g_gs, g_idx, Y, y_gs, y_idx, groups, rowlen, scale);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// Filter's sum over frames, backward + ReLU mask: dst[g][t][:] = dsum[g][:] * (Y[g][t][:] > 0)
__global__ void bcast_mask_relu_kernel(float *dst, const float *dsum, const float *Y, int groups, int T, int H, float scale,
                                       const int32_t *len) {
    const int64_t total = (int64_t)groups * T * H;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(e / ((int64_t)T * H));
        const int c = (int)(e % H);
        const int t = (int)((e / H) % T);
        const bool in = !len || t < len[g];            // frames past the clip's length never entered the sum
        dst[e] = (in && Y[e] > 0.f) ? dsum[(int64_t)g * H + c] * scale : 0.f;
    }
}
__global__ void bcast_mask_relu_v4_kernel(float *dst, const float *dsum, const float *Y, int T, int H4, float scale, const int32_t *len) {
    const int g = blockIdx.y;
    const int L = len ? len[g] : T;
    const v4f *yp = reinterpret_cast<const v4f *>(Y) + (int64_t)g * T * H4;
    const v4f *sp = reinterpret_cast<const v4f *>(dsum) + (int64_t)g * H4;
    v4f *dp = reinterpret_cast<v4f *>(dst) + (int64_t)g * T * H4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < T * H4; e += gridDim.x * blockDim.x) {
        const int t = e / H4, c = e - t * H4;
        const v4f yv = yp[e], sv = sp[c];
        v4f o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (t < L && yv[j] > 0.f) ? sv[j] * scale : 0.f;
        dp[e] = o;
    }
}
int launch_bcast_mask_relu(float *dst, const float *dsum, const float *Y, int groups, int T, int H, hipStream_t s, float scale,
                           const int32_t *len) {
    if (groups == 0) return 0;
    if (H % 4 == 0 && groups <= 65535 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(dsum) | reinterpret_cast<uintptr_t>(Y)) & 15) == 0) {
        STAIR_ACCT("bcast_mask_relu_kernel", (2ll * groups * T * H + (int64_t)groups * H) * 4);
        const int e4 = T * (H / 4);
        const int bx = std::max(1, std::min((e4 + kBlock - 1) / kBlock, std::max(1, 4096 / groups)));
        hipLaunchKernelGGL(bcast_mask_relu_v4_kernel, dim3(bx, groups), dim3(kBlock), 0, s, dst, dsum, Y, T, H / 4, scale, len);
        STAIR_LAUNCH_CHECK();
        return 0;
    }
    STAIR_ACCT("bcast_mask_relu_kernel", (2ll * groups * T * H + (int64_t)groups * H) * 4);
    const int64_t total = (int64_t)groups * T * H;
    hipLaunchKernelGGL(bcast_mask_relu_kernel, dim3((unsigned)std::min<int64_t>((total + kBlock - 1) / kBlock, 4096)), dim3(kBlock),
                       0, s, dst, dsum, Y, groups, T, H, scale, len);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// Deterministic fan-in of the gradient arenas (csrc/plan.hip build_grad_fanin): entry = (kind, destination slot, first staging slot,
// count, rows per slot); destination += staging slot 0 + staging slot 1 + ... in that order.  kind 0: vec rows [H], 1: map tiles [T, H],
// 2: att rows [rows, T].  One block column per entry, 4096 floats per block.
__global__ void grad_fanin_kernel(const int32_t *tab, float *gvec, float *gmap, float *gatt, int H, int T) {
    const int32_t *e = tab + 5 * blockIdx.x;
    const int kind = e[0], cnt = e[3], width = e[4];
    float *base = kind == 0 ? gvec : (kind == 1 ? gmap : gatt);
    const int64_t unit = kind == 0 ? H : (kind == 1 ? (int64_t)T * H : T);
    const int64_t len = kind == 2 ? (int64_t)width * T : unit;
    float *dst = base + (int64_t)e[1] * unit;
    const float *src = base + (int64_t)e[2] * unit;
    const int64_t step = kind == 2 ? (int64_t)width * T : unit;          // distance between two staging slots
    for (int64_t i = (int64_t)blockIdx.y * 4096 + threadIdx.x; i < std::min<int64_t>(len, ((int64_t)blockIdx.y + 1) * 4096); i += blockDim.x) {
        float acc = dst[i];
        for (int j = 0; j < cnt; ++j) acc += src[(int64_t)j * step + i];
        dst[i] = acc;
    }
}
int launch_grad_fanin(const int32_t *tab, int n, float *gvec, float *gmap, float *gatt, int H, int T, hipStream_t s) {
    if (n == 0) return 0;
    STAIR_ACCT("grad_fanin_kernel", (int64_t)n * T * H * 4);
    hipLaunchKernelGGL(grad_fanin_kernel, dim3(n, (unsigned)(((int64_t)T * H + 4095) / 4096)), dim3(256), 0, s, tab, gvec, gmap, gatt, H, T);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// dst[dst_idx[i]][:] += src[i][:] * scale   (rows of `len` floats, atomic)
__global__ void scatter_add_rows_kernel(float *dst, const int32_t *dst_idx, const float *src, int n, int len, float scale) {
    const int i = blockIdx.x;
    float *d = dst + (int64_t)idx_or_id(dst_idx, i) * len;
    const float *sr = src + (int64_t)i * len;
    for (int c = threadIdx.x; c < len; c += blockDim.x) unsafeAtomicAdd(d + c, sr[c] * scale);
}
int launch_scatter_add_rows(float *dst, const int32_t *dst_idx, const float *src, int n, int len, float scale, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(n), dim3(128), 0, s, dst, dst_idx, src, n, len, scale);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// cat backward (modules.py cat inputs of Exists/Xor/Equals/Compare/ToAction, decoder module_net.py:137)
//   CAT2: dA[ia] += g0, dB[ib] += g1;  EXISTS [a,b,a*b]: dA += g0 + g2*b, dB += g1 + g2*a
//   XOR [|a-b|,a,b]: dA += g0*sign(a-b) + g1, dB += -g0*sign(a-b) + g2
__global__ void pack_bwd_kernel(int mode, const float *A, const int32_t *ia, const float *B, const int32_t *ib,
                                const float *g, float *dA, float *dB, int n, int H) {
    const int i = blockIdx.x;
    const int64_t ra = (int64_t)idx_or_id(ia, i) * H, rb = (int64_t)idx_or_id(ib, i) * H;
    const int segs = mode == PACK_CAT2 ? 2 : 3;
    const float *gi = g + (int64_t)i * segs * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        float da, db;
        if (mode == PACK_CAT2) {
            da = gi[c]; db = gi[H + c];
        } else if (mode == PACK_EXISTS) {
            const float a = A[ra + c], b = B[rb + c];
            da = gi[c] + gi[2 * H + c] * b; db = gi[H + c] + gi[2 * H + c] * a;
        } else {
            const float sg = sgn(A[ra + c] - B[rb + c]);
            da = gi[c] * sg + gi[H + c]; db = -gi[c] * sg + gi[2 * H + c];
        }
        unsafeAtomicAdd(dA + ra + c, da);
        unsafeAtomicAdd(dB + rb + c, db);
    }
}
int launch_pack_bwd(int mode, const float *A, const int32_t *ia, const float *B, const int32_t *ib, const float *g,
                    float *dA, float *dB, int n, int H, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_bwd_kernel, dim3(n), dim3(128), 0, s, mode, A, ia, B, ib, g, dA, dB, n, H);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// span mean backward: dtok[start+r][:] += dvec[out][:] / count
__global__ void span_mean_bwd_kernel(float *dtok, int64_t ld, const int32_t *start, const int32_t *count, const float *dvec,
                                     const int32_t *out_idx, int n, int H) {
    const int i = blockIdx.x;
    const int s = start[i], c = count[i];
    const float *g = dvec + (int64_t)out_idx[i] * H;
    const float inv = 1.0f / (float)c;
    for (int col = threadIdx.x; col < H; col += blockDim.x) {
        const float v = g[col] * inv;
        for (int r = 0; r < c; ++r) unsafeAtomicAdd(dtok + (int64_t)(s + r) * ld + col, v);
    }
}
int launch_span_mean_bwd(float *dtok, int64_t ld, const int32_t *start, const int32_t *count, const float *dvec,
                         const int32_t *out_idx, int n, int H, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(span_mean_bwd_kernel, dim3(n), dim3(128), 0, s, dtok, ld, start, count, dvec, out_idx, n, H);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// The same as a gather per token row (training plans carry the row -> spans lists): dtok[row][:] += sum over the spans that contain
// the row, in span order, of dvec[out[span]][:] / count[span] -- one writer per element, no atomics, the same sum every run.
__global__ void span_mean_bwd_rows_kernel(float *dtok, int64_t ld, const int32_t *row_ptr, const int32_t *row_span, const int32_t *count,
                                          const float *dvec, const int32_t *out_idx, int H) {
    const int row = blockIdx.x;
    const int b = row_ptr[row], e = row_ptr[row + 1];
    if (b == e) return;
    for (int col = threadIdx.x; col < H; col += blockDim.x) {
        float acc = 0.f;
        for (int j = b; j < e; ++j) { const int sp = row_span[j]; acc += dvec[(int64_t)out_idx[sp] * H + col] / (float)count[sp]; }
        dtok[(int64_t)row * ld + col] += acc;
    }
}
int launch_span_mean_bwd_rows(float *dtok, int64_t ld, int rows, const int32_t *row_ptr, const int32_t *row_span, const int32_t *count,
                              const float *dvec, const int32_t *out_idx, int H, hipStream_t s) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(span_mean_bwd_rows_kernel, dim3(rows), dim3(128), 0, s, dtok, ld, row_ptr, row_span, count, dvec, out_idx, H);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// cosine attention backward.  out = (cos+1)*0.49, cos = f.k / (nf * nk), nf = max(|f|,eps), nk likewise.
//   dF[fi][t][:] += dcos * (k/(nf nk) - cos f/nf^2)        (atomic: pairs of one Localize share the tile)
//   dK[p][:]     += sum_t dcos_t * (f_t/(nf_t nk) - cos_t k/nk^2)
// One block per pair: phase 1 computes per-frame scalars into LDS, phase 2 is column-parallel.
__global__ void cosine_attn_bwd_kernel(const float *F, int64_t f_gs, const int32_t *f_idx, const float *Kmat,
                                       const int32_t *k_idx, const float *datt, const int32_t *out_idx, float *dF,
                                       float *dK, int npairs, int T, int H, const int32_t *gf_idx, const int32_t *gk_idx) {
    extern __shared__ float sm[];       // [3][T]: a_t = dcos/(nf nk), b_t = dcos*cos/nf^2, c_t = dcos*cos
    float *sa = sm, *sb = sm + T, *sc = sm + 2 * T;
    __shared__ float s_nk2;
    const int p = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fi = idx_or_id(f_idx, p), ki = idx_or_id(k_idx, p);
    const float *f = F + (int64_t)fi * f_gs;
    const float *k = Kmat + (int64_t)ki * H;
    const float *g = datt + (int64_t)idx_or_id(out_idx, p) * T;
    float nk2 = 0.f;
    for (int c = lane; c < H; c += 64) nk2 += k[c] * k[c];
    nk2 = wave_sum(nk2);
    const float nk = fmaxf(sqrtf(nk2), 1e-8f);
    if (threadIdx.x == 0) s_nk2 = nk * nk;
    for (int t = wave; t < T; t += kWavesPerBlock) {
        const float *ft = f + (int64_t)t * H;
        float d = 0.f, nf2 = 0.f;
        for (int c = lane; c < H; c += 64) { d += ft[c] * k[c]; nf2 += ft[c] * ft[c]; }
        d = wave_sum(d); nf2 = wave_sum(nf2);
        const float nf = fmaxf(sqrtf(nf2), 1e-8f);
        const float cosv = d / (nf * nk);
        const float dcos = 0.49f * g[t];
        if (lane == 0) { sa[t] = dcos / (nf * nk); sb[t] = dcos * cosv / (nf * nf); sc[t] = dcos * cosv; }
    }
    __syncthreads();
    const float inv_nk2 = 1.0f / s_nk2;
    const int gfi = gf_idx ? gf_idx[p] : fi, gki = gk_idx ? gk_idx[p] : ki;        // gradient slots (fan-in staging)
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float kc = k[c];
        float dk = 0.f;
        for (int t = 0; t < T; ++t) {
            const float fc = f[(int64_t)t * H + c];
            unsafeAtomicAdd(dF + (int64_t)gfi * f_gs + (int64_t)t * H + c, sa[t] * kc - sb[t] * fc);
            dk += sa[t] * fc - sc[t] * kc * inv_nk2;
        }
        unsafeAtomicAdd(dK + (int64_t)gki * H + c, dk);
    }
}
int launch_cosine_attn_bwd(const float *F, int64_t f_gs, const int32_t *f_idx, const float *Kmat, const int32_t *k_idx,
                           const float *datt, const int32_t *out_idx, float *dF, float *dK, int npairs, int T, int H,
                           hipStream_t s, const int32_t *gf_idx, const int32_t *gk_idx) {
    if (npairs == 0) return 0;
    hipLaunchKernelGGL(cosine_attn_bwd_kernel, dim3(npairs), dim3(kBlock), 3 * T * sizeof(float), s, F, f_gs, f_idx, Kmat,
                       k_idx, datt, out_idx, dF, dK, npairs, T, H, gf_idx, gk_idx);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// Grouped form for Localize / Superlative, where ALL pairs of an instance share one private [T,H] tile: one block
// per instance, no atomics.  The forward scores are saved (out = (cos+1)*0.49), so cos is recovered from them and
// only the row norms are recomputed:
//   dF[t][:] = sum_a sa[a][t] k_a  -  (sum_a sb[a][t]) f_t ;   dK[a][:] = sum_t sa[a][t] f_t  -  (sum_t sc[a][t]) k_a / nk_a^2
// with sa = dcos/(nf nk), sb = dcos*cos/nf^2, sc = dcos*cos, dcos = 0.49 * dscore.  (The per-pair kernel above
// issues Ka*T*H atomic adds per instance -- half a billion for a batch of Superlatives with Ka = T = 64.)
__global__ void rownorm_kernel(const float *X, int64_t rows, int H, float *out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4 *x = reinterpret_cast<const float4 *>(X + row * H);
    float ss = 0.f;
    for (int c = lane; c < H / 4; c += 64) { const float4 v = x[c]; ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    ss = wave_sum(ss);
    if (lane == 0) out[row] = fmaxf(sqrtf(ss), 1e-8f);
}

// grid (instance, 64-column chunk); the chunk's K [Ka][64] and F [T][64] slices and the pair scalars live in LDS.
// MF: the two products of a block -- dF[T x 64] = sa^T k and dK[Ka x 64] = sa f -- on v_mfma_f32_32x32x2_f32 (exact fp32
// products, fp32 accumulate) over LDS images zero-padded to multiples of 32 (sa rows TP + 1 floats apart: conflict-free for both
// operand orders).  The scalar loops did two LDS reads per FMA and were bound by LDS bandwidth (Superlative, Ka = T = 64:
// 247 us for 254 instances).
using f32x16_rb = __attribute__((ext_vector_type(16))) float;
template <bool MF>
__global__ __launch_bounds__(256) void cosine_attn_bwd_grouped_kernel(const float *F, const float *Kmat, const float *nf_all, const float *nk_all,
                                               const float *score, int64_t s_stride, const int32_t *score_idx,
                                               const float *dscore, const int32_t *dscore_idx, const int32_t *pair_start,
                                               const int32_t *pair_cnt, float *dF, float *dK, int n, int T, int H, int ka_max) {
    extern __shared__ float sm[];          // sa [KP][TS] | kt [KP][64] | ft [TP][64] | sbsum [TP] | scsum [KP]
    const int i = blockIdx.x, c0 = blockIdx.y * 64;
    const int p0 = pair_start[i], Ka = pair_cnt[i];
    const int TP = MF ? (T + 31) / 32 * 32 : T, KP = MF ? (ka_max + 31) / 32 * 32 : ka_max, TS = MF ? TP + 1 : T;
    float *sa = sm, *kt = sa + KP * TS, *ft = kt + KP * 64, *sbsum = ft + TP * 64, *scsum = sbsum + TP;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;      // 4 waves: lane = column, part = row quarter
    const float *f = F + (int64_t)i * T * H;
    const float *nf = nf_all + (int64_t)i * T, *nk = nk_all + p0;
    for (int t = threadIdx.x; t < TP; t += blockDim.x) sbsum[t] = 0.f;
    for (int a = threadIdx.x; a < KP; a += blockDim.x) scsum[a] = 0.f;
    if (MF) for (int e = threadIdx.x; e < KP * TS; e += blockDim.x) sa[e] = 0.f;
    // Loads first, in batches of 8 independent ones per thread, then the LDS stores: written as `load; store` per iteration
    // every global load waited for the previous store (a block took ~40 us whatever it computed: 16 dependent round trips)
    for (int t0 = part; t0 < TP; t0 += 32) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int t = t0 + 4 * j; v[j] = t < T ? f[(int64_t)t * H + c0 + lane] : 0.f; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int t = t0 + 4 * j; if (t < TP) ft[t * 64 + lane] = v[j]; }
    }
    for (int a0 = part; a0 < KP; a0 += 32) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int a = a0 + 4 * j; v[j] = a < Ka ? Kmat[(int64_t)(p0 + a) * H + c0 + lane] : 0.f; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int a = a0 + 4 * j; if (a < KP) kt[a * 64 + lane] = v[j]; }
    }
    float *su = scsum + KP;                 // u[a][t] = dcos * cos, same [KP][TS] layout as sa
    int *rowi = reinterpret_cast<int *>(su + KP * TS);                  // [2][ka_max]: score / dscore row of pair a
    for (int a = threadIdx.x; a < Ka; a += blockDim.x) {
        rowi[a] = score_idx ? score_idx[p0 + a] : p0 + a;
        rowi[ka_max + a] = dscore_idx ? dscore_idx[p0 + a] : p0 + a;
    }
    __syncthreads();
    for (int e0 = threadIdx.x; e0 < Ka * T; e0 += 8 * blockDim.x) {
        float sv[8], dv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = e0 + j * blockDim.x;
            sv[j] = dv[j] = 0.f;
            if (e < Ka * T) {
                const int a = e / T, t = e - a * T;
                sv[j] = score[(int64_t)rowi[a] * s_stride + t];
                dv[j] = dscore[(int64_t)rowi[ka_max + a] * s_stride + t];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = e0 + j * blockDim.x;
            if (e >= Ka * T) continue;
            const int a = e / T, t = e - a * T;
            const float cosv = sv[j] / 0.49f - 1.0f;
            const float dcos = 0.49f * dv[j];
            sa[a * TS + t] = dcos / (nf[t] * nk[a]);
            su[a * TS + t] = dcos * cosv;
        }
    }
    __syncthreads();
    {               // the two marginal sums as column / row walks: no LDS atomics (Ka * T adds onto T + Ka addresses serialised, and float
                    // atomics land in any order: the sums would differ from run to run)
        for (int t = threadIdx.x; t < T; t += blockDim.x) {
            float acc = 0.f;
            for (int a = 0; a < Ka; ++a) acc += su[a * TS + t];
            sbsum[t] = acc / (nf[t] * nf[t]);
        }
        for (int a = threadIdx.x; a < Ka; a += blockDim.x) {
            float acc = 0.f;
            for (int t = 0; t < T; ++t) acc += su[a * TS + t];
            scsum[a] = acc;
        }
        __syncthreads();
    }
    if (MF) {
        const int r = lane & 31, kk = lane >> 5;
        const int tilesT = TP / 32, tilesK = (Ka + 31) / 32;
        // dF tiles: (row tile of t) x (2 column tiles); contraction over a
        for (int tile = part; tile < tilesT * 2; tile += 4) {
            const int tt = tile >> 1, cc = tile & 1;
            f32x16_rb acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            for (int a = 0; a < tilesK * 32; a += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[(a + kk) * TS + 32 * tt + r], kt[(a + kk) * 64 + 32 * cc + r], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int t = 32 * tt + 8 * (e >> 2) + 4 * kk + (e & 3), col = 32 * cc + r;
                if (t < T) dF[((int64_t)i * T + t) * H + c0 + col] = acc[e] - sbsum[t] * ft[t * 64 + col];
            }
        }
        // dK tiles: (row tile of a) x (2 column tiles); contraction over t
        for (int tile = part; tile < tilesK * 2; tile += 4) {
            const int at = tile >> 1, cc = tile & 1;
            f32x16_rb acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            for (int t = 0; t < TP; t += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[(32 * at + r) * TS + t + kk], ft[(t + kk) * 64 + 32 * cc + r], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int a = 32 * at + 8 * (e >> 2) + 4 * kk + (e & 3), col = 32 * cc + r;
                if (a < Ka) dK[(int64_t)(p0 + a) * H + c0 + col] = acc[e] - scsum[a] * kt[a * 64 + col] / (nk[a] * nk[a]);
            }
        }
        return;
    }
    for (int t = part; t < T; t += 4) {
        float acc = 0.f;
        for (int a = 0; a < Ka; ++a) acc += sa[a * T + t] * kt[a * 64 + lane];
        dF[((int64_t)i * T + t) * H + c0 + lane] = acc - sbsum[t] * ft[t * 64 + lane];
    }
    for (int a = part; a < Ka; a += 4) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc += sa[a * T + t] * ft[t * 64 + lane];
        dK[(int64_t)(p0 + a) * H + c0 + lane] = acc - scsum[a] * kt[a * 64 + lane] / (nk[a] * nk[a]);
    }
}
int launch_cosine_attn_bwd_grouped(const float *F, const float *Kmat, const float *score, const int32_t *score_idx,
                                   const float *dscore, const int32_t *dscore_idx, const int32_t *pair_start,
                                   const int32_t *pair_cnt, float *dF, float *dK, float *nf_ws, float *nk_ws, int n, int npairs,
                                   int T, int H, int ka_max, hipStream_t s) {
    if (n == 0) return 0;
    STAIR_ACCT("cosine_attn_bwd_grouped_kernel", (2ll * n * T * H + 2ll * npairs * H + 2ll * npairs * T) * 4);
    STAIR_CHECK(H % 64 == 0, "H must be a multiple of 64");
    const int64_t frows = (int64_t)n * T;
    hipLaunchKernelGGL(rownorm_kernel, dim3((unsigned)((frows + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, s, F, frows, H, nf_ws);
    hipLaunchKernelGGL(rownorm_kernel, dim3((npairs + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, s, Kmat, (int64_t)npairs, H, nk_ws);
    STAIR_LAUNCH_CHECK();
    const int TP = (T + 31) / 32 * 32, KP = (ka_max + 31) / 32 * 32;
    const size_t shmem_mf = (2 * (size_t)KP * (TP + 1) + (size_t)KP * 64 + (size_t)TP * 64 + TP + KP + 2 * ka_max) * sizeof(float);
    const size_t shmem_sc = (2 * (size_t)ka_max * T + (size_t)ka_max * 64 + (size_t)T * 64 + T + ka_max + 2 * ka_max) * sizeof(float);
    const bool mf = ka_max > 8 && shmem_mf <= 80 * 1024;     // many pairs per instance (Superlative); Localize (1-2 pairs) is bound by its F and dF rows, not by the products; at least two blocks per CU
    const size_t shmem = mf ? shmem_mf : shmem_sc;
    STAIR_CHECK(shmem <= 160 * 1024, "cosine backward: Ka*T too large for LDS");
    if (shmem > 48 * 1024) {   // e.g. Superlative at T = 64: 16 + 16 + 16 KB; at max_video_length = 150 (args.py:29): 165 KB
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cosine_attn_bwd_grouped_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cosine_attn_bwd_grouped_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    }
    if (mf) hipLaunchKernelGGL(cosine_attn_bwd_grouped_kernel<true>, dim3(n, H / 64), dim3(256), shmem, s, F, Kmat, nf_ws, nk_ws, score, (int64_t)T,
                               score_idx, dscore, dscore_idx, pair_start, pair_cnt, dF, dK, n, T, H, ka_max);
    else hipLaunchKernelGGL(cosine_attn_bwd_grouped_kernel<false>, dim3(n, H / 64), dim3(256), shmem, s, F, Kmat, nf_ws, nk_ws, score, (int64_t)T,
                            score_idx, dscore, dscore_idx, pair_start, pair_cnt, dF, dK, n, T, H, ka_max);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Temporal relate nets backward (recomputes the three layers in LDS).  dw[6] accumulate with atomics.
struct RelateWB { const float *w[6]; float *dw[6]; long long *dw64[6]; };      // dw64: fixed-point shadows (det_shadow) or NULL
constexpr int kRelateTapStride = 72;      // LDS floats per staged Conv1d filter: up to 71 taps (2 k + 1 with k <= 35) + the bias
__global__ void temporal_relate_bwd_kernel(const float *att, const int32_t *att_idx, const int32_t *att_k,
                                           const float *drel, const int32_t *rel_idx, float *datt, int n, int T, int mode,
                                           int conv, int ksize, RelateWB W, const int32_t *len, const int32_t *gatt_idx) {
    extern __shared__ float sm[];   // x0, y1, y2, y3 (post-activation), g (ping), g2 (pong): 6 rows of T; then the conv filters
    float *x0 = sm, *y1 = sm + T, *y2 = sm + 2 * T, *y3 = sm + 3 * T, *ga = sm + 4 * T, *gb = sm + 5 * T;
    // Conv1d nets: the three filters (k, k, 2k + 1 taps) and biases staged in LDS once -- read from global memory inside the tap
    // loops below, every tap was a dependent scalar load (31 us per block whatever it computed)
    float *wl = sm + 6 * T;         // [3][kRelateTapStride]: taps, then the bias at [kRelateTapStride - 1]
    if (mode != 0 && conv)
        for (int e = threadIdx.x; e < 3 * kRelateTapStride; e += blockDim.x) {
            const int layer = e / kRelateTapStride, j = e - layer * kRelateTapStride;
            const int k = layer < 2 ? ksize : 2 * ksize + 1;
            wl[e] = j < k ? W.w[2 * layer][j] : (j == kRelateTapStride - 1 ? W.w[2 * layer + 1][0] : 0.f);
        }
    const int i = blockIdx.x;
    const int K = att_k[i];
    const int L = len ? len[i] : T;                 // the clip's own frame count (see temporal_relate_kernel); Linear nets: L == T
    const float *a = att + (int64_t)att_idx[i] * T;
    const float *dr = drel + (int64_t)rel_idx[i] * T;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += a[(int64_t)k * T + t];
        x0[t] = t < L ? acc / (float)K : 0.f;
        ga[t] = t < L ? dr[t] : 0.f;
        y1[t] = y2[t] = y3[t] = gb[t] = 0.f;
    }
    __syncthreads();
    if (mode != 0) {
        const float *xin[3] = {x0, y1, y2};
        float *yout[3] = {y1, y2, y3};
        for (int layer = 0; layer < 3; ++layer) {           // forward recompute
            const float *w = W.w[2 * layer], *b = W.w[2 * layer + 1];
            const float *wc = wl + layer * kRelateTapStride;
            const float *x = xin[layer];
            float *y = yout[layer];
            const int k = layer < 2 ? ksize : 2 * ksize + 1, left = (k - 1) / 2;
            for (int t = threadIdx.x; t < L; t += blockDim.x) {
                float acc;
                if (conv) {
                    acc = wc[kRelateTapStride - 1];
                    for (int j = 0; j < k; ++j) { const int u = t + j - left; if (u >= 0 && u < L) acc += wc[j] * x[u]; }
                } else {
                    acc = b[t];
                    for (int u = 0; u < T; ++u) acc += w[(int64_t)t * T + u] * x[u];
                }
                y[t] = layer < 2 ? fmaxf(acc, 0.f) : sigmoid_acc(acc);
            }
            __syncthreads();
        }
        float *gin = ga, *gout = gb;
        for (int layer = 2; layer >= 0; --layer) {          // backward
            const float *w = W.w[2 * layer];
            const float *wc = wl + layer * kRelateTapStride;
            const float *x = xin[layer], *y = yout[layer];
            const int k = layer < 2 ? ksize : 2 * ksize + 1, left = (k - 1) / 2;
            for (int t = threadIdx.x; t < L; t += blockDim.x)      // through the activation
                gin[t] = layer < 2 ? (y[t] > 0.f ? gin[t] : 0.f) : gin[t] * y[t] * (1.f - y[t]);
            __syncthreads();
            if (conv) {
                for (int j = threadIdx.x; j < k; j += blockDim.x) {        // dw[j] = sum_t dz[t] x[t+j-left]
                    float acc = 0.f;
                    for (int t = 0; t < L; ++t) { const int u = t + j - left; if (u >= 0 && u < L) acc += gin[t] * x[u]; }
                    grad_add(W.dw[2 * layer], W.dw64[2 * layer], j, acc);
                }
                if (threadIdx.x == 0) {
                    float acc = 0.f;
                    for (int t = 0; t < L; ++t) acc += gin[t];
                    grad_add(W.dw[2 * layer + 1], W.dw64[2 * layer + 1], 0, acc);
                }
                for (int u = threadIdx.x; u < L; u += blockDim.x) {       // dx[u] = sum_j w[j] dz[u-j+left]
                    float acc = 0.f;
                    for (int j = 0; j < k; ++j) { const int t = u - j + left; if (t >= 0 && t < L) acc += wc[j] * gin[t]; }
                    gout[u] = acc;
                }
            } else {
                for (int e = threadIdx.x; e < T * T; e += blockDim.x) {
                    const int t = e / T, u = e - t * T;
                    grad_add(W.dw[2 * layer], W.dw64[2 * layer], e, gin[t] * x[u]);
                }
                for (int t = threadIdx.x; t < T; t += blockDim.x) grad_add(W.dw[2 * layer + 1], W.dw64[2 * layer + 1], t, gin[t]);
                for (int u = threadIdx.x; u < T; u += blockDim.x) {
                    float acc = 0.f;
                    for (int t = 0; t < T; ++t) acc += w[(int64_t)t * T + u] * gin[t];
                    gout[u] = acc;
                }
            }
            __syncthreads();
            float *tmp = gin; gin = gout; gout = tmp;
        }
        ga = gin;
    }
    // mean over K rows backward
    for (int t = threadIdx.x; t < L; t += blockDim.x) {
        const float v = ga[t] / (float)K;
        for (int k = 0; k < K; ++k) unsafeAtomicAdd(datt + ((int64_t)(gatt_idx ? gatt_idx[i] : att_idx[i]) + k) * T + t, v);
    }
}
int launch_temporal_relate_bwd(const float *att, const int32_t *att_idx, const int32_t *att_k, const float *drel,
                               const int32_t *rel_idx, float *datt, int n, int T, int mode, int conv, int ksize,
                               const float *const w[6], float *const dw[6], hipStream_t s, const int32_t *len, const int32_t *gatt_idx) {
    if (n == 0) return 0;
    // reads: K attention rows in (the mean over K; K <= 2, counted as 2), the relate output's gradient; writes: K attention-row
    // gradients; the three layers' filters / matrices and their gradients are a few hundred floats per launch
    STAIR_ACCT("temporal_relate_bwd_kernel", (int64_t)n * T * 4 * (2 + 1 + 2));
    RelateWB W;
    for (int i = 0; i < 6; ++i) {
        W.w[i] = (mode && w) ? w[i] : nullptr; W.dw[i] = (mode && dw) ? dw[i] : nullptr;
        W.dw64[i] = W.dw[i] ? det_shadow(W.dw[i]) : nullptr;
    }
    STAIR_CHECK(!(mode && conv) || 2 * ksize + 1 < kRelateTapStride, "Conv1d relate nets: kernel size <= 35");
    hipLaunchKernelGGL(temporal_relate_bwd_kernel, dim3(n), dim3(64), (6 * T + 3 * kRelateTapStride) * sizeof(float), s, att, att_idx, att_k, drel,
                       rel_idx, datt, n, T, mode, conv, ksize, W, len, gatt_idx);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm backward fused with the ReLU that precedes it in TemporalModule (modules.py:326-327):
//   y = relu(z) saved; out = LN(y).  dz = relu'(y) * rstd * (dxh - mean(dxh) - xh * mean(dxh*xh)), dxh = dout*gamma
// stats[row] = (mean, rstd) for the parameter-gradient pass.
__global__ void layernorm_bwd_kernel(const float *dOut, int64_t g_gs, const int32_t *g_idx, const float *Y, int n, int T,
                                     int H, const float *gamma, float eps, float *dZ, float *stats, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= (int64_t)n * T) return;
    const int g = (int)(row / T), t = (int)(row - (int64_t)g * T);
    const float *y = Y + row * H;
    const float *go = dOut + (int64_t)idx_or_id(g_idx, g) * g_gs + (int64_t)t * H;
    float sum = 0.f;
    for (int c = lane; c < H; c += 64) sum += y[c];
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
    for (int c = lane; c < H; c += 64) { const float d = y[c] - mean; sq += d * d; }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < H; c += 64) {
        const float dxh = go[c] * gamma[c], xh = (y[c] - mean) * rstd;
        s1 += dxh; s2 += dxh * xh;
    }
    s1 = wave_sum(s1) / (float)H; s2 = wave_sum(s2) / (float)H;
    for (int c = lane; c < H; c += 64) {
        const float dxh = go[c] * gamma[c], xh = (y[c] - mean) * rstd;
        const float dy = rstd * (dxh - s1 - xh * s2);
        dZ[row * H + c] = y[c] > 0.f ? dy * scale : 0.f;
    }
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
__global__ void layernorm_param_grad_kernel(const float *dOut, int64_t g_gs, const int32_t *g_idx, const float *Y,
                                            const float *stats, int n, int T, int H, float *dgamma, float *dbeta, int slab, long long *dgamma64, long long *dbeta64) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    const int64_t rbeg = (int64_t)blockIdx.y * slab, rend = min((int64_t)n * T, rbeg + slab);
    float ag = 0.f, ab = 0.f;
    for (int64_t row = rbeg; row < rend; ++row) {
        const int g = (int)(row / T), t = (int)(row - (int64_t)g * T);
        const float go = dOut[(int64_t)idx_or_id(g_idx, g) * g_gs + (int64_t)t * H + c];
        ag += go * (Y[row * H + c] - stats[2 * row]) * stats[2 * row + 1];
        ab += go;
    }
    grad_add(dgamma, dgamma64, c, ag);
    grad_add(dbeta, dbeta64, c, ab);
}
// One pass for H a multiple of 256: a wave owns a row at a time (16-byte pieces in registers: y and dOut are read ONCE),
// walks rows grid-stride and keeps the dgamma / dbeta sums of its columns in registers until the end (the two-kernel form
// above reads Y four times and dOut three times and runs at 1.4 TB/s: profiles/r02_row_kernels.json).
template <int NV>
__global__ void layernorm_bwd_fused_kernel(const float *dOut, int64_t g_gs, const int32_t *g_idx, const float *Y, int n, int T,
                                           const float *gamma, float eps, float *dZ, float *dgamma, float *dbeta, float scale, long long *dgamma64, long long *dbeta64) {
    constexpr int H = NV * 256;
    const int lane = threadIdx.x & 63;
    const int64_t rows = (int64_t)n * T, stride = (int64_t)gridDim.x * kWavesPerBlock;
    v4f gm[NV], ag[NV], ab[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        gm[j] = *reinterpret_cast<const v4f *>(gamma + 256 * j + 4 * lane);
        ag[j] = v4f{0.f, 0.f, 0.f, 0.f}; ab[j] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); row < rows; row += stride) {
        const int g = (int)(row / T), t = (int)(row - (int64_t)g * T);
        const float *yr = Y + row * H;
        const float *gr = dOut + (int64_t)idx_or_id(g_idx, g) * g_gs + (int64_t)t * H;
        v4f y[NV], go[NV];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            y[j] = *reinterpret_cast<const v4f *>(yr + 256 * j + 4 * lane);
            go[j] = *reinterpret_cast<const v4f *>(gr + 256 * j + 4 * lane);
            sum += y[j][0] + y[j][1] + y[j][2] + y[j][3];
        }
        const float mean = wave_sum(sum) / (float)H;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = y[j][e] - mean; sq += d * d; }
        const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dxh = go[j][e] * gm[j][e], xh = (y[j][e] - mean) * rstd;
                s1 += dxh; s2 += dxh * xh;
                ag[j][e] += go[j][e] * xh; ab[j][e] += go[j][e];
            }
        s1 = wave_sum(s1) / (float)H; s2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v4f o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dxh = go[j][e] * gm[j][e], xh = (y[j][e] - mean) * rstd;
                o[e] = y[j][e] > 0.f ? rstd * (dxh - s1 - xh * s2) * scale : 0.f;
            }
            *reinterpret_cast<v4f *>(dZ + row * H + 256 * j + 4 * lane) = o;
        }
    }
    // the four waves of the workgroup add up through LDS, then ONE set of atomics per workgroup (at most 256 workgroups:
    // every workgroup adds into the same 2 x H addresses, where atomics run an order of magnitude below their streaming rate)
    __shared__ float red[2][kWavesPerBlock][H];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        *reinterpret_cast<v4f *>(&red[0][wv][256 * j + 4 * lane]) = ag[j];
        *reinterpret_cast<v4f *>(&red[1][wv][256 * j + 4 * lane]) = ab[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * H; c += kBlock) {
        const int which = c / H, col = c - which * H;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < kWavesPerBlock; ++q) v += red[which][q][col];
        grad_add(which ? dbeta : dgamma, which ? dbeta64 : dgamma64, col, v);
    }
}
int launch_layernorm_bwd(const float *dOut, int64_t g_gs, const int32_t *g_idx, const float *Y, int n, int T, int H,
                         const float *gamma, float eps, float *dZ, float *stats, float *dgamma, float *dbeta, hipStream_t s,
                         float scale) {
    if (n == 0) return 0;
    STAIR_ACCT("layernorm_bwd_kernel+param_grad", (3ll * n * T * H + 2ll * n * T) * 4);
    const int64_t rows = (int64_t)n * T;
    const bool al = ((reinterpret_cast<uintptr_t>(dOut) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(dZ) | reinterpret_cast<uintptr_t>(gamma)) & 15) == 0;
    if (al && (H == 256 || H == 512) && g_gs % 4 == 0) {
        const unsigned blocks = (unsigned)std::min<int64_t>((rows + kWavesPerBlock - 1) / kWavesPerBlock, 256 * 4);
        if (H == 512) hipLaunchKernelGGL(layernorm_bwd_fused_kernel<2>, dim3(blocks), dim3(kBlock), 0, s, dOut, g_gs, g_idx, Y, n, T, gamma, eps, dZ, dgamma, dbeta, scale, det_shadow(dgamma), det_shadow(dbeta));
        else hipLaunchKernelGGL(layernorm_bwd_fused_kernel<1>, dim3(blocks), dim3(kBlock), 0, s, dOut, g_gs, g_idx, Y, n, T, gamma, eps, dZ, dgamma, dbeta, scale, det_shadow(dgamma), det_shadow(dbeta));
        STAIR_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0,
                       s, dOut, g_gs, g_idx, Y, n, T, H, gamma, eps, dZ, stats, scale);
    STAIR_LAUNCH_CHECK();
    const int slab = (int)std::max<int64_t>(64, (rows + 255) / 256);
    hipLaunchKernelGGL(layernorm_param_grad_kernel, dim3((H + 255) / 256, (unsigned)((rows + slab - 1) / slab)), dim3(256), 0, s,
                       dOut, g_gs, g_idx, Y, stats, n, T, H, dgamma, dbeta, slab, det_shadow(dgamma), det_shadow(dbeta));
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// row-scaled dense backward helper: G = dZ.W already computed for the SCALED input (rs_t * x_t):
//   dX[xi][t][:] += rs_t * G[g][t][:]   (atomic),   drs[ri][t] += sum_c G[g][t][c] * X[xi][t][c]
__global__ void rowscale_bwd_kernel(const float *G, const float *X, int64_t x_gs, const int32_t *x_idx, const float *rs,
                                    int64_t rs_gs, const int32_t *rs_idx, float *dX, float *drs, int n, int T, int H, const int32_t *gx_idx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= (int64_t)n * T) return;
    const int g = (int)(row / T), t = (int)(row - (int64_t)g * T);
    const int64_t xo = (int64_t)idx_or_id(x_idx, g) * x_gs + (int64_t)t * H;
    const int64_t ro = (int64_t)idx_or_id(rs_idx, g) * rs_gs + t;
    const float r = rs[ro];
    const float *gr = G + row * H;
    const int64_t go = gx_idx ? (int64_t)gx_idx[g] * x_gs + (int64_t)t * H : xo;      // where the gradient of the tile goes (fan-in staging)
    float d = 0.f;
    for (int c = lane; c < H; c += 64) {
        const float gv = gr[c];
        d += gv * X[xo + c];
        if (dX) unsafeAtomicAdd(dX + go + c, r * gv);
    }
    d = wave_sum(d);
    if (lane == 0 && drs) unsafeAtomicAdd(drs + ro, d);
}
int launch_rowscale_bwd(const float *G, const float *X, int64_t x_gs, const int32_t *x_idx, const float *rs, int64_t rs_gs,
                        const int32_t *rs_idx, float *dX, float *drs, int n, int T, int H, hipStream_t s, const int32_t *gx_idx) {
    if (n == 0) return 0;
    STAIR_ACCT("rowscale_bwd_kernel", (2ll * n * T * H + (dX ? (int64_t)n * T * H : 0) + (int64_t)n * T) * 4);
    const int64_t rows = (int64_t)n * T;
    hipLaunchKernelGGL(rowscale_bwd_kernel, dim3((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, s,
                       G, X, x_gs, x_idx, rs, rs_gs, rs_idx, dX, drs, n, T, H, gx_idx);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// out[g][t] = sigmoid(X[g][t].w + b + extra[g]) backward:
//   dpre = dout * a (1-a);  dXdst[g][t][:] (+)= dpre * w;  dextra[g] += sum_t dpre;  dpre_out[g][t] = dpre
// add_mode: 0 store into dXdst, 1 add (non-atomic read-modify-write: rows are private to this launch)
__global__ void rowdot_sigmoid_bwd_kernel(const float *dOut, int64_t o_gs, const int32_t *o_idx, const float *A, int64_t a_gs,
                                          const int32_t *a_idx, const float *w, float *dXdst, int add_mode, float *dpre_out,
                                          float *dextra, int n, int T, int H, float keep) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= (int64_t)n * T) return;
    const int g = (int)(row / T), t = (int)(row - (int64_t)g * T);
    // keep < 1: A holds sigmoid * mask / keep (HasItem's Sigmoid -> Dropout, modules.py:129-131): sigmoid = A * keep where
    // the element was kept and the factor 1/keep multiplies the incoming gradient; a dropped element (A == 0) gets none
    const float a = A[(int64_t)idx_or_id(a_idx, g) * a_gs + t] * keep;
    const float dpre = dOut[(int64_t)idx_or_id(o_idx, g) * o_gs + t] / keep * a * (1.f - a);
    float *dx = dXdst + row * H;
    if ((H & 3) == 0) {                     // 16-byte accesses (rows start on multiples of H floats of a 256-byte aligned workspace)
        float4 *dx4 = reinterpret_cast<float4 *>(dx);
        const float4 *w4 = reinterpret_cast<const float4 *>(w);
        for (int c = lane; c < H / 4; c += 64) {
            const float4 wv = w4[c];
            float4 o = add_mode ? dx4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            o.x += dpre * wv.x; o.y += dpre * wv.y; o.z += dpre * wv.z; o.w += dpre * wv.w;
            dx4[c] = o;
        }
    } else
        for (int c = lane; c < H; c += 64) dx[c] = (add_mode ? dx[c] : 0.f) + dpre * w[c];
    if (lane == 0) {
        dpre_out[row] = dpre;
        if (dextra) unsafeAtomicAdd(dextra + g, dpre);
    }
}
// out[g] = sum_t X[g][t] in frame order (the per-instance sum of FilterFrame's pre-sigmoid gradients: one thread per instance, no atomics)
__global__ void rowsum_small_kernel(const float *X, float *out, int n, int T) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += X[(int64_t)g * T + t];
    out[g] = acc;
}
int launch_rowsum_small(const float *X, float *out, int n, int T, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(rowsum_small_kernel, dim3((n + 127) / 128), dim3(128), 0, s, X, out, n, T);
    STAIR_LAUNCH_CHECK();
    return 0;
}
int launch_rowdot_sigmoid_bwd(const float *dOut, int64_t o_gs, const int32_t *o_idx, const float *A, int64_t a_gs,
                              const int32_t *a_idx, const float *w, float *dXdst, int add_mode, float *dpre_out,
                              float *dextra, int n, int T, int H, hipStream_t s, float keep) {
    if (n == 0) return 0;
    STAIR_ACCT("rowdot_sigmoid_bwd_kernel", (2ll * n * T * H + 2ll * n * T) * 4);
    const int64_t rows = (int64_t)n * T;
    hipLaunchKernelGGL(rowdot_sigmoid_bwd_kernel, dim3((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock),
                       0, s, dOut, o_gs, o_idx, A, a_gs, a_idx, w, dXdst, add_mode, dpre_out, dextra, n, T, H, keep);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// out[c] += sum_rows scale[row] * X[row][c]   (gradient of a [H] weight used in a row dot; X rows optionally gathered)
__global__ void weighted_colsum_kernel(const float *X, int64_t ld, const int32_t *x_idx, const float *scale, float *out,
                                       int rows, int H, int slab, long long *out64) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    const int rbeg = blockIdx.y * slab, rend = min(rows, rbeg + slab);
    float acc = 0.f;
    int r = rbeg;
    for (; r + 8 <= rend; r += 8) {          // eight independent loads in flight per thread (one at a time: 64 round trips per block)
        float sc[8], xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = scale[r + j]; xv[j] = X[(int64_t)idx_or_id(x_idx, r + j) * ld + c]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += sc[j] * xv[j];
    }
    for (; r < rend; ++r) acc += scale[r] * X[(int64_t)idx_or_id(x_idx, r) * ld + c];
    grad_add(out, out64, c, acc);
}
int launch_weighted_colsum(const float *X, int64_t ld, const int32_t *x_idx, const float *scale, float *out, int rows, int H,
                           hipStream_t s) {
    if (rows == 0) return 0;
    STAIR_ACCT("weighted_colsum_kernel", ((int64_t)rows * H + rows + H) * 4);
    const int slab = std::max(64, (rows + 255) / 256);
    hipLaunchKernelGGL(weighted_colsum_kernel, dim3((H + 255) / 256, (rows + slab - 1) / slab), dim3(256), 0, s, X, ld, x_idx,
                       scale, out, rows, H, slab, det_shadow(out));
    STAIR_LAUNCH_CHECK();
    return 0;
}

// dV[idx[i]][:] += scale[i] * w[:]    (vecdot backward w.r.t. the gathered vectors)
__global__ void axpy_rows_kernel(float *dV, const int32_t *idx, const float *scale, const float *w, int n, int H) {
    const int i = blockIdx.x;
    const float sc = scale[i];
    float *d = dV + (int64_t)idx_or_id(idx, i) * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) unsafeAtomicAdd(d + c, sc * w[c]);
}
int launch_axpy_rows(float *dV, const int32_t *idx, const float *scale, const float *w, int n, int H, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpy_rows_kernel, dim3(n), dim3(128), 0, s, dV, idx, scale, w, n, H);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// sum of a float array into out[0] (bias of a 1-output Linear)
__global__ void sum_all_kernel(const float *x, float *out, int n, long long *out64) {
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += x[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) grad_add(out, out64, 0, acc);
}
int launch_sum_all(const float *x, float *out, int n, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(sum_all_kernel, dim3(std::min((n + 255) / 256, 256)), dim3(256), 0, s, x, out, n, det_shadow(out));
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Relate backward: y = softmax(x + sign*beta): dx = y * (dy - sum(dy*y)); datt[in] += dx; dbeta += sign*dx
__global__ void relate_softmax_bwd_kernel(const float *att, float *datt, const int32_t *in_idx, const int32_t *out_idx,
                                          float *dbeta, float sign, int n, int T, const int32_t *len, const int32_t *gin_idx, long long *dbeta64) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float *y = att + (int64_t)out_idx[i] * T;
    const float *dy = datt + (int64_t)out_idx[i] * T;
    const int L = len ? len[i] : T;
    float dot = 0.f;
    for (int t = lane; t < L; t += 64) dot += dy[t] * y[t];
    dot = wave_sum(dot);
    for (int t = lane; t < L; t += 64) {
        const float dx = y[t] * (dy[t] - dot);
        unsafeAtomicAdd(datt + (int64_t)(gin_idx ? gin_idx[i] : in_idx[i]) * T + t, dx);
        grad_add(dbeta, dbeta64, t, sign * dx);
    }
}
int launch_relate_softmax_bwd(const float *att, float *datt, const int32_t *in_idx, const int32_t *out_idx, float *dbeta,
                              float sign, int n, int T, hipStream_t s, const int32_t *len, const int32_t *gin_idx) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(relate_softmax_bwd_kernel, dim3((n + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, s, att, datt,
                       in_idx, out_idx, dbeta, sign, n, T, len, gin_idx, det_shadow(dbeta));
    STAIR_LAUNCH_CHECK();
    return 0;
}

// min / |a-b| backward (torch.minimum splits a tie evenly; sign(0) = 0 for abs)
__global__ void eltwise_bwd_kernel(int mode, const float *base, float *dbase, const int32_t *ia, const int32_t *ib,
                                   const int32_t *io, int n, int len, const int32_t *gia, const int32_t *gib) {
    const int i = blockIdx.x;
    const int64_t oa = (int64_t)ia[i] * len, ob = (int64_t)ib[i] * len, oo = (int64_t)io[i] * len;
    for (int c = threadIdx.x; c < len; c += blockDim.x) {
        const float a = base[oa + c], b = base[ob + c], g = dbase[oo + c];
        float da, db;
        if (mode == 0) {
            da = a < b ? g : (a == b ? 0.5f * g : 0.f);
            db = b < a ? g : (a == b ? 0.5f * g : 0.f);
        } else {
            const float sg = sgn(a - b);
            da = sg * g; db = -sg * g;
        }
        unsafeAtomicAdd(dbase + (gia ? (int64_t)gia[i] * len : oa) + c, da);
        unsafeAtomicAdd(dbase + (gib ? (int64_t)gib[i] * len : ob) + c, db);
    }
}
int launch_eltwise_bwd(int mode, const float *base, float *dbase, const int32_t *ia, const int32_t *ib, const int32_t *io,
                       int n, int len, hipStream_t s, const int32_t *gia, const int32_t *gib) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(eltwise_bwd_kernel, dim3(n), dim3(128), 0, s, mode, base, dbase, ia, ib, io, n, len, gia, gib);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// AttnVideo backward: dmap[in][t][:] += att[a][t]*dmap[out][t][:];  datt[a][t] += dmap[out][t].map[in][t]
__global__ void attnvideo_bwd_kernel(const float *map, float *dmap, const float *att, float *datt, const int32_t *in_idx,
                                     const int32_t *att_idx, const int32_t *out_idx, int n, int T, int H, const int32_t *gin_idx,
                                     const int32_t *gatt_idx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= (int64_t)n * T) return;
    const int i = (int)(row / T), t = (int)(row - (int64_t)i * T);
    const int64_t xi = ((int64_t)in_idx[i] * T + t) * H, xo = ((int64_t)out_idx[i] * T + t) * H;
    const int64_t gi = gin_idx ? ((int64_t)gin_idx[i] * T + t) * H : xi;
    const float a = att[(int64_t)att_idx[i] * T + t];
    float d = 0.f;
    for (int c = lane; c < H; c += 64) {
        const float g = dmap[xo + c];
        d += g * map[xi + c];
        unsafeAtomicAdd(dmap + gi + c, a * g);
    }
    d = wave_sum(d);
    if (lane == 0) unsafeAtomicAdd(datt + (int64_t)(gatt_idx ? gatt_idx[i] : att_idx[i]) * T + t, d);
}
int launch_attnvideo_bwd(const float *map, float *dmap, const float *att, float *datt, const int32_t *in_idx,
                         const int32_t *att_idx, const int32_t *out_idx, int n, int T, int H, hipStream_t s, const int32_t *gin_idx,
                         const int32_t *gatt_idx) {
    if (n == 0) return 0;
    STAIR_ACCT("attnvideo_bwd_kernel", (4ll * n * T * H + 2ll * n * T) * 4);
    const int64_t rows = (int64_t)n * T;
    hipLaunchKernelGGL(attnvideo_bwd_kernel, dim3((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, s,
                       map, dmap, att, datt, in_idx, att_idx, out_idx, n, T, H, gin_idx, gatt_idx);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// Choose backward: the output IS one of the two keywords -> route its gradient there (selection recomputed)
__global__ void choose_bwd_kernel(const float *vec, float *dvec, const int32_t *k1, const int32_t *k2, const int32_t *q,
                                  const int32_t *out, int n, int H, const int32_t *gk1, const int32_t *gk2) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float *a = vec + (int64_t)k1[i] * H, *b = vec + (int64_t)k2[i] * H, *c = vec + (int64_t)q[i] * H;
    float na = 0.f, nb = 0.f, nc = 0.f;
    for (int e = lane; e < H; e += 64) { na += a[e] * a[e]; nb += b[e] * b[e]; nc += c[e] * c[e]; }
    na = fmaxf(sqrtf(wave_sum(na)), 1e-8f); nb = fmaxf(sqrtf(wave_sum(nb)), 1e-8f); nc = fmaxf(sqrtf(wave_sum(nc)), 1e-8f);
    float da = 0.f, db = 0.f;
    for (int e = lane; e < H; e += 64) { const float cn = c[e] / nc; da += (a[e] / na) * cn; db += (b[e] / nb) * cn; }
    da = wave_sum(da); db = wave_sum(db);
    const int64_t dst = (int64_t)(da > db ? (gk1 ? gk1[i] : k1[i]) : (gk2 ? gk2[i] : k2[i])) * H, src = (int64_t)out[i] * H;
    for (int e = lane; e < H; e += 64) unsafeAtomicAdd(dvec + dst + e, dvec[src + e]);
}
int launch_choose_bwd(const float *vec, float *dvec, const int32_t *k1, const int32_t *k2, const int32_t *q,
                      const int32_t *out, int n, int H, hipStream_t s, const int32_t *gk1, const int32_t *gk2) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(choose_bwd_kernel, dim3((n + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, s, vec, dvec, k1, k2,
                       q, out, n, H, gk1, gk2);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// Superlative pooling backward.  pre[i] = sum_a w'_a rows[a]; w' = w or 1-w, w = softmax_a(sum_t S[a][t]).
//   drows[row_id[a]] += w'_a dpre;   dS[a][t] = w_a (dw_a - sum_b dw_b w_b),  dw_a = +-(dpre . rows[a])
__global__ void superlative_pool_bwd_kernel(const float *S, const float *rowbase, float *drowbase, const int32_t *row_id,
                                            const int32_t *row_start, const int32_t *row_cnt, int is_min, const float *dpre,
                                            float *dS, int n, int T, int H, const int32_t *len, const int32_t *g_row_id) {
    extern __shared__ float sm[];    // w[Ka], dw[Ka], row ids [Ka]
    const int i = blockIdx.x;
    const int r0 = row_start[i], Ka = row_cnt[i];
    const int L = len ? len[i] : T;
    float *w = sm, *dw = sm + Ka;
    int *rid = reinterpret_cast<int *>(sm + 2 * Ka);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *dp = dpre + (int64_t)i * H;
    for (int a = threadIdx.x; a < Ka; a += blockDim.x) rid[a] = row_id[r0 + a];
    __syncthreads();
    // four rows per wave and pass: their loads are independent and issued together (one row at a time, each row's two
    // reductions waited for its own loads: 16 round trips per wave, ~40 us of a 137 us launch)
    for (int a0 = wave; a0 < Ka; a0 += 4 * kWavesPerBlock) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = a0 + j * kWavesPerBlock;
            if (a < Ka) {
                const float *sr = S + (int64_t)(r0 + a) * T;
                for (int t = lane; t < L; t += 64) acc[j] += sr[t];
                const float *row = rowbase + (int64_t)rid[a] * H;
                for (int c = lane; c < H; c += 64) d[j] += dp[c] * row[c];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = a0 + j * kWavesPerBlock;
            const float as = wave_sum(acc[j]), ds = wave_sum(d[j]);
            if (a < Ka && lane == 0) { w[a] = as; dw[a] = is_min ? -ds : ds; }
        }
    }
    __syncthreads();
    __shared__ float s_dot;
    if (wave == 0) {
        float m = -INFINITY;
        for (int a = lane; a < Ka; a += 64) m = fmaxf(m, w[a]);
        m = wave_max(m);
        float sum = 0.f;
        for (int a = lane; a < Ka; a += 64) sum += expf(w[a] - m);
        sum = wave_sum(sum);
        float dot = 0.f;
        for (int a = lane; a < Ka; a += 64) { w[a] = expf(w[a] - m) / sum; dot += dw[a] * w[a]; }
        dot = wave_sum(dot);
        if (lane == 0) s_dot = dot;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < Ka * T; e += blockDim.x) {
        const int a = e / T;
        dS[(int64_t)(r0 + a) * T + (e - a * T)] = (e - a * T) < L ? w[a] * (dw[a] - s_dot) : 0.f;
    }
    if (g_row_id) {                 // the action rows' GRADIENT rows (fan-in staging, csrc/plan.hip build_grad_fanin)
        __syncthreads();
        for (int a = threadIdx.x; a < Ka; a += blockDim.x) rid[a] = g_row_id[r0 + a];
        __syncthreads();
    }
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float g = dp[c];
        for (int a = 0; a < Ka; ++a)
            unsafeAtomicAdd(drowbase + (int64_t)rid[a] * H + c, (is_min ? 1.f - w[a] : w[a]) * g);
    }
}
int launch_superlative_pool_bwd(const float *S, const float *rowbase, float *drowbase, const int32_t *row_id,
                                const int32_t *row_start, const int32_t *row_cnt, int is_min, const float *dpre, float *dS,
                                int n, int T, int H, hipStream_t s, const int32_t *len, const int32_t *g_row_id) {
    if (n == 0) return 0;
    // per instance: scores S [Ka, T] in, dS [Ka, T] out, the Ka action rows [Ka, H] in and their gradient rows out (read-modify-
    // write), the pooled vector's gradient [H] in; Ka = T for the map-valued action lists of the AGQA programs
    STAIR_ACCT("superlative_pool_bwd_kernel", (int64_t)n * ((int64_t)2 * T * T + (int64_t)3 * T * H + H) * 4);
    hipLaunchKernelGGL(superlative_pool_bwd_kernel, dim3(n), dim3(kBlock), (size_t)3 * std::max(T, 2) * sizeof(float), s, S, rowbase,
                       drowbase, row_id, row_start, row_cnt, is_min, dpre, dS, n, T, H, len, g_row_id);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// decoder cross entropy (train_module.py:193-194, nn.CrossEntropyLoss on one row):
//   loss[i] = logsumexp(logits[i]) - logits[i][ans[i]];  dlogits[i] = scale * (softmax - onehot)
__global__ void ce_loss_kernel(const float *logits, const int32_t *answers, float scale, float *loss, float *dlogits, int n,
                               int A) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float *x = logits + (int64_t)i * A;
    float m = -INFINITY;
    for (int c = lane; c < A; c += 64) m = fmaxf(m, x[c]);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane; c < A; c += 64) sum += expf(x[c] - m);
    sum = wave_sum(sum);
    const int ans = answers[i];
    if (ans < 0) {                 // question without a decoder loss (train_module.py:376: global_steps <= train_decoder_after_iters)
        if (lane == 0 && loss) loss[i] = 0.f;
        if (dlogits)
            for (int c = lane; c < A; c += 64) dlogits[(int64_t)i * A + c] = 0.f;
        return;
    }
    if (ans >= A) {                // not a class of this vocabulary: never read out of bounds; the NaN loss flags the caller's bug
        if (lane == 0 && loss) loss[i] = __builtin_nanf("");
        if (dlogits)
            for (int c = lane; c < A; c += 64) dlogits[(int64_t)i * A + c] = 0.f;
        return;
    }
    if (lane == 0 && loss) loss[i] = logf(sum) + m - x[ans];
    if (dlogits)
        for (int c = lane; c < A; c += 64) dlogits[(int64_t)i * A + c] = scale * (expf(x[c] - m) / sum - (c == ans ? 1.f : 0.f));
}
int launch_ce_loss(const float *logits, const int32_t *answers, float scale, float *loss, float *dlogits, int n, int A,
                   hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(ce_loss_kernel, dim3((n + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, s, logits, answers, scale,
                       loss, dlogits, n, A);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults, train_module.py:326: betas (0.9, 0.999), eps 1e-8, weight_decay 0 unless given)
// over a flat parameter buffer.  `touched[seg]` != 0 marks parameter tensors that received a gradient in
// this window; untouched tensors are skipped entirely, which is what torch does for grad == None
// (modules that no program of the window used) -- their moments and step count do not advance.
// one workgroup of 64 threads per 256-float block (= one segment granule), 16 bytes per thread
__global__ void adam_kernel(float *p, const float *g, float *m, float *v, const int32_t *seg_of_block, const int32_t *touched,
                            const float *step_of_seg, float lr, float b1, float b2, float eps, float wd, int64_t n,
                            const uint32_t *guard) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (guard && *guard) return;                      // the pass that produced g reported a failure: leave p, m, v untouched
    const int seg = seg_of_block[blockIdx.x];
    if (!touched[seg]) return;
    const float t = step_of_seg[seg];                 // already incremented for this step
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float a = lr / bc1, rb2 = 1.0f / sqrtf(bc2);
    const v4f gv = *reinterpret_cast<const v4f *>(g + i);
    v4f pv = *reinterpret_cast<v4f *>(p + i), mv = *reinterpret_cast<v4f *>(m + i), vv = *reinterpret_cast<v4f *>(v + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float grad = gv[j];
        if (wd != 0.f) grad += wd * pv[j];
        mv[j] = b1 * mv[j] + (1.f - b1) * grad;
        vv[j] = b2 * vv[j] + (1.f - b2) * grad * grad;
        pv[j] -= a * mv[j] / (sqrtf(vv[j]) * rb2 + eps);
    }
    *reinterpret_cast<v4f *>(p + i) = pv; *reinterpret_cast<v4f *>(m + i) = mv; *reinterpret_cast<v4f *>(v + i) = vv;
}
int launch_adam(float *p, const float *g, float *m, float *v, const int32_t *seg_of_block, const int32_t *touched,
                const float *step_of_seg, float lr, float b1, float b2, float eps, float wd, int64_t n, const uint32_t *guard,
                hipStream_t s) {
    if (n == 0) return 0;
    STAIR_ACCT("adam_kernel", 7ll * n * 4);
    STAIR_CHECK(n % 256 == 0, "the flat parameter buffer is made of whole 256-float blocks");
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(n / 256)), dim3(64), 0, s, p, g, m, v, seg_of_block, touched,
                       step_of_seg, lr, b1, b2, eps, wd, n, guard);
    STAIR_LAUNCH_CHECK();
    return 0;
}

}  // namespace stair

namespace stair {
// G[g][t][:] *= rs[g][t]   (rows of H floats; rs contiguous [n*T])
__global__ void scale_rows_kernel(float *G, const float *rs, int64_t rows, int H) {
    const int64_t total = rows * H;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
        G[e] *= rs[e / H];
}
int launch_scale_rows(float *G, const float *rs, int64_t rows, int H, hipStream_t s) {
    if (rows == 0) return 0;
    STAIR_ACCT("scale_rows_kernel", (2ll * rows * H + rows) * 4);
    const int64_t total = rows * H;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, G, rs, rows, H);
    STAIR_LAUNCH_CHECK();
    return 0;
}
}  // namespace stair

// ---- C ABI: the backward halves of the building-block families exported in forward form (include/stair_hip.h) ----
extern "C" int stair_cosine_attn_bwd(const float *F, int64_t f_gstride, const int32_t *f_idx, const float *Kmat, const int32_t *k_idx,
                                     const float *d_att, const int32_t *out_idx, float *dF, float *dK, int32_t npairs, int32_t T,
                                     int32_t H, stair_stream stream) {
    STAIR_CHECK(F && Kmat && d_att && dF && dK && npairs >= 0 && T > 0 && H > 0, "bad argument");
    return stair::launch_cosine_attn_bwd(F, f_gstride, f_idx, Kmat, k_idx, d_att, out_idx, dF, dK, npairs, T, H,
                                         static_cast<hipStream_t>(stream));
}
extern "C" int stair_temporal_relate_bwd(const float *att, const int32_t *att_idx, const int32_t *att_k, const float *d_out,
                                         const int32_t *out_idx, float *d_att, int32_t n, int32_t T, int32_t mode, int32_t conv,
                                         int32_t ksize, const float *const w[6], float *const dw[6], stair_stream stream) {
    STAIR_CHECK(att && att_idx && att_k && d_out && out_idx && d_att && n >= 0 && T > 0, "bad argument");
    STAIR_CHECK(mode >= 0 && mode <= 3 && (mode == 0 || (w && dw)), "mode 1..3 needs the relate net's weights and gradient buffers");
    return stair::launch_temporal_relate_bwd(att, att_idx, att_k, d_out, out_idx, d_att, n, T, mode, conv, ksize, w, dw,
                                             static_cast<hipStream_t>(stream), nullptr);
}

