// Per-module intermediate-supervision losses of /root/reference/train_module.py:33-194 (CriterionByModule),
// forward value + gradient w.r.t. the module result, injected straight into the plan's gradient arenas
// (so that stair_plan_backward continues from them).  One wave or one small block per loss item; items are
// few (a handful per question) and rows short, so these kernels are launch/latency bound by nature.
#include "ops.h"

namespace stair {
// stair_loss_groups: the NEXT stair_loss_* launch of this thread evaluates its items group by group (items that add into the same
// gradient slot, sorted by the caller): one workgroup (one wave for the head criterion) walks a group's items in order and adds their
// gradients with plain read - add - write, so the arena sums do not depend on the order in which float atomics land.  Consumed by
// that launch.
struct LossGroups { const int32_t *order = nullptr, *grp_off = nullptr; int n_groups = 0; };
static thread_local LossGroups tl_groups;
static LossGroups take_groups(int n_items) {
    LossGroups g = tl_groups;
    tl_groups = LossGroups();
    if (g.n_groups <= 0 || g.n_groups > n_items) g = LossGroups();
    return g;
}
namespace {
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;
// add into a gradient arena element: plain when the launch is grouped (one workgroup owns the slot), else a float atomic
__device__ __forceinline__ void arena_add(float *p, float v, bool owned) {
    if (owned) *p += v;
    else unsafeAtomicAdd(p, v);
}

// span_to_attention, train_module.py:67-81, evaluated per frame t (double arithmetic like the reference)
__device__ __forceinline__ float span_gold(double g0, double g1, int L, int t) {
    const double start = fmin((double)L - 0.002, fmax(0.001, g0));
    const double end = fmin((double)L - 0.001, g1);
    const int si = (int)ceil(start), ei = (int)floor(end);
    double v = 0.0;
    if (si < ei && t >= si && t < ei) v += 1.0;
    if (si <= ei) {
        if (t == si - 1) v += (double)si - start;
        if (t == ei) v += end - (double)ei;
    } else if (t == ei) {
        v += end - start;
    }
    return (float)v;
}
}  // namespace

// attention_score_criterion (:83-90) for Localize [K,T] (:173-182), Temporal / ExistsFrame [T] (:157-164,184-191)
// `len` (optional): frames of item i's clip; the attention rows keep the stride T, criterion and gold mask see L frames
__global__ void loss_attention_kernel(const float *att, float *d_att, const int32_t *slot, const int32_t *K,
                                      const int32_t *iv_off, const double *intervals, int n, int T, float scale, float *loss,
                                      const int32_t *len, const int32_t *order, const int32_t *grp_off) {
  const int q0 = grp_off ? grp_off[blockIdx.x] : blockIdx.x, q1 = grp_off ? grp_off[blockIdx.x + 1] : blockIdx.x + 1;
  for (int q = q0; q < q1; ++q) {            // (a group's items share the slot, hence K and the clip: element e is the same thread's every time)
    const int i = order ? order[q] : q;
    const int k = K[i];
    const int L = len ? len[i] : T;
    const float inv = 1.0f / (float)(k * L);
    float acc = 0.f;
    for (int e = threadIdx.x; e < k * L; e += blockDim.x) {
        const int r = e / L, t = e - r * L;
        const double *iv = intervals + 2 * (int64_t)(iv_off[i] + r);
        const float g = span_gold(iv[0], iv[1], L, t);
        const int64_t o = ((int64_t)slot[i] + r) * T + t;
        const float p = att[o];
        acc += -(g * logf(p) + (1.f - g) * logf(1.f - p));
        if (d_att) arena_add(d_att + o, scale * inv * (-g / p + (1.f - g) / (1.f - p)), grp_off != nullptr);
    }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) loss[i] = acc * inv;       // blockDim == 64: one wave
  }
}
int launch_loss_attention(const float *att, float *d_att, const int32_t *slot, const int32_t *K, const int32_t *iv_off,
                          const double *intervals, int n, int T, float scale, float *loss, hipStream_t s, const int32_t *len = nullptr) {
    if (n == 0) return 0;
    const LossGroups G = take_groups(n);
    hipLaunchKernelGGL(loss_attention_kernel, dim3(G.n_groups ? G.n_groups : n), dim3(64), 0, s, att, d_att, slot, K, iv_off, intervals, n, T, scale, loss, len,
                       G.order, G.grp_off);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// Linear pretrain head [NOUT,H] + loss.  NOUT == 2: CrossEntropy against a bool (Exists / Xor, :92-99);
// NOUT == 1: mean squared error against 0/1 (Equals, :101-107).
template <int NOUT>
__global__ void loss_head_kernel(const float *vec, float *d_vec, const int32_t *slot, const int32_t *label, const float *W,
                                 const float *b, float *dW, float *db, int n, int H, float scale, float *loss,
                                 const int32_t *order, const int32_t *grp_off, int n_groups, long long *dW64, long long *db64) {
    const int lane = threadIdx.x & 63;
    const int gq = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);          // ungrouped: the item; grouped: the group (a wave walks it)
    if (gq >= (grp_off ? n_groups : n)) return;
  const int q0 = grp_off ? grp_off[gq] : gq, q1 = grp_off ? grp_off[gq + 1] : gq + 1;
  for (int q = q0; q < q1; ++q) {
    const int i = order ? order[q] : q;
    const float *x = vec + (int64_t)slot[i] * H;
    float z[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        float d = 0.f;
        for (int c = lane; c < H; c += 64) d += x[c] * W[(int64_t)j * H + c];
        z[j] = wave_sum(d) + b[j];
    }
    float dz[NOUT];
    float l;
    if (NOUT == 2) {
        const float m = fmaxf(z[0], z[NOUT - 1]);
        const float e0 = expf(z[0] - m), e1 = expf(z[NOUT - 1] - m), sum = e0 + e1;
        const int y = label[i] ? 1 : 0;
        l = logf(sum) + m - z[y];
        dz[0] = scale * (e0 / sum - (y == 0 ? 1.f : 0.f));
        dz[NOUT - 1] = scale * (e1 / sum - (y == 1 ? 1.f : 0.f));
    } else {
        const float diff = z[0] - (label[i] ? 1.f : 0.f);
        l = diff * diff;
        dz[0] = scale * 2.f * diff;
    }
    for (int c = lane; c < H; c += 64) {
        float dx = 0.f;
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
            dx += dz[j] * W[(int64_t)j * H + c];
            if (dW) grad_add(dW, dW64, (int64_t)j * H + c, dz[j] * x[c]);
        }
        if (d_vec) arena_add(d_vec + (int64_t)slot[i] * H + c, dx, grp_off != nullptr);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NOUT; ++j)
            if (db) grad_add(db, db64, j, dz[j]);
        loss[i] = l;
    }
  }
}
int launch_loss_head(int nout, const float *vec, float *d_vec, const int32_t *slot, const int32_t *label, const float *W,
                     const float *b, float *dW, float *db, int n, int H, float scale, float *loss, hipStream_t s) {
    if (n == 0) return 0;
    STAIR_CHECK(nout == 1 || nout == 2, "head width must be 1 (Equals) or 2 (Exists/Xor)");
    const LossGroups G = take_groups(n);
    const int units = G.n_groups ? G.n_groups : n;
    const dim3 grid((units + kWavesPerBlock - 1) / kWavesPerBlock), block(kBlock);
    long long *dW64 = dW ? det_shadow(dW) : nullptr, *db64 = db ? det_shadow(db) : nullptr;       // inside a fixed-point scope (stair_grad_shadows_begin)
    if (nout == 2) hipLaunchKernelGGL(loss_head_kernel<2>, grid, block, 0, s, vec, d_vec, slot, label, W, b, dW, db, n, H, scale, loss, G.order, G.grp_off, G.n_groups, dW64, db64);
    else hipLaunchKernelGGL(loss_head_kernel<1>, grid, block, 0, s, vec, d_vec, slot, label, W, b, dW, db, n, H, scale, loss, G.order, G.grp_off, G.n_groups, dW64, db64);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// Contrastive CE of Filter / ToAction / Superlative (:113-125 with the window pooling of :388-406):
//   pred = L2Normalize(x) (module_net.py:211-216); logits_c = G[c] . pred over the classes of the item's window;
//   loss = -log softmax(logits)[positive].   Gradient goes back through the normalisation into d_vec.
// Table form (presence != NULL): G is the table of ALL class representations [n_cls, H], the item's pool is the set of
// classes c with presence[win_row[i]][c] > 0 (a window's pool summed over the data-parallel ranks, so no rank needs the
// others' class lists on the host), pos[i] the positive CLASS id; absent classes take no part in the softmax.
__global__ void loss_contrastive_kernel(const float *vec, float *d_vec, const int32_t *slot, const int32_t *pos,
                                        const int32_t *win_start, const int32_t *win_cnt, const float *G, int n, int H,
                                        float scale, float *loss, const float *presence, const int32_t *win_row, int n_cls,
                                        const int32_t *order, const int32_t *grp_off) {
    extern __shared__ float sm[];      // [C] logits -> softmax probs, [H] dpred, [C] classes with a non-zero gradient term, [C] their coefficients
    __shared__ float s_nrm, s_part[kWavesPerBlock];
    __shared__ int s_nact;
  const int q0 = grp_off ? grp_off[blockIdx.x] : blockIdx.x, q1 = grp_off ? grp_off[blockIdx.x + 1] : blockIdx.x + 1;
  for (int q = q0; q < q1; ++q) {
    const int i = order ? order[q] : q;
    const int c0 = presence ? 0 : win_start[i], C = presence ? n_cls : win_cnt[i];
    const float *mask = presence ? presence + (int64_t)win_row[i] * n_cls : nullptr;
    float *prob = sm, *dpred = sm + C, *acoef = sm + C + H + C;
    int *act = reinterpret_cast<int *>(sm + C + H);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *x = vec + (int64_t)slot[i] * H;
    if (wave == 0) {
        float ss = 0.f;
        for (int h = lane; h < H; h += 64) ss += x[h] * x[h];
        ss = wave_sum(ss);
        if (lane == 0) s_nrm = fmaxf(sqrtf(ss), 1e-12f);
    }
    __syncthreads();
    const float inv = 1.0f / s_nrm;
    float xr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xr[j] = 64 * j + lane < H ? x[64 * j + lane] : 0.f;
    // (a class that is absent from the item's window is never multiplied: its logit is -inf whatever the product, its softmax term
    // and its gradient term are exactly zero -- with a table of 214 classes and windows of 32 questions four classes in five are absent)
    for (int c = wave; c < C; c += kWavesPerBlock) {
        if (mask && !(mask[c] > 0.f)) {             // wave-uniform
            if (lane == 0) prob[c] = -INFINITY;
            continue;
        }
        const float *g = G + (int64_t)(c0 + c) * H;
        float gr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) gr[j] = 64 * j + lane < H ? g[64 * j + lane] : 0.f;       // the row's loads together (H <= 512)
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (64 * j + lane < H) d += gr[j] * xr[j];
        d = wave_sum(d);
        if (lane == 0) prob[c] = d * inv;
    }
    __syncthreads();
    if (wave == 0) {
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, prob[c]);
        m = wave_max(m);
        float sum = 0.f;
        for (int c = lane; c < C; c += 64) sum += expf(prob[c] - m);
        sum = wave_sum(sum);
        const int p = pos[i] - c0;
        if (lane == 0) loss[i] = logf(sum) + m - prob[p];
        // dlogits; the classes whose term is not an exact zero, compacted in class order (skipping a zero term leaves every sum below
        // bit-identical; a list instead of a test inside the loop lets the loads of several classes be in flight together)
        int base = 0;
        for (int cb = 0; cb < C; cb += 64) {
            const int c = cb + lane;
            const float dl = c < C ? scale * (expf(prob[c] - m) / sum - (c == p ? 1.f : 0.f)) : 0.f;
            const bool on = dl != 0.f;
            const unsigned long long bal = __ballot(on);
            if (on) {
                const int at = base + __popcll(bal & ((1ull << lane) - 1ull));
                act[at] = c; acoef[at] = dl;
            }
            base += __popcll(bal);
        }
        if (lane == 0) s_nact = base;
    }
    __syncthreads();
    const int nact = s_nact;
    float part = 0.f;
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
        const float *gh = G + (int64_t)c0 * H + h;
        float d = 0.f;
        int k = 0;
        for (; k + 4 <= nact; k += 4) {
            const float g0 = gh[(int64_t)act[k] * H], g1 = gh[(int64_t)act[k + 1] * H], g2 = gh[(int64_t)act[k + 2] * H], g3 = gh[(int64_t)act[k + 3] * H];
            d += acoef[k] * g0; d += acoef[k + 1] * g1; d += acoef[k + 2] * g2; d += acoef[k + 3] * g3;
        }
        for (; k < nact; ++k) d += acoef[k] * gh[(int64_t)act[k] * H];
        dpred[h] = d;
        part += d * x[h] * inv;
    }
    part = wave_sum(part);
    if (lane == 0) s_part[wave] = part;
    __syncthreads();
    float dot = 0.f;               // pred . dpred, the waves' parts in wave order (an LDS float atomic summed them in arrival order)
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) dot += s_part[w];
    if (d_vec)
        for (int h = threadIdx.x; h < H; h += blockDim.x)
            arena_add(d_vec + (int64_t)slot[i] * H + h, (dpred[h] - x[h] * inv * dot) * inv, grp_off != nullptr);
    __syncthreads();               // the next item of the group reuses the LDS
  }
}
int launch_loss_contrastive(const float *vec, float *d_vec, const int32_t *slot, const int32_t *pos, const int32_t *win_start,
                            const int32_t *win_cnt, const float *G, int n, int H, int max_classes, float scale, float *loss,
                            hipStream_t s) {
    if (n == 0) return 0;
    STAIR_CHECK(H <= 512, "hidden size above 512");
    const LossGroups Gr = take_groups(n);
    hipLaunchKernelGGL(loss_contrastive_kernel, dim3(Gr.n_groups ? Gr.n_groups : n), dim3(kBlock), (size_t)(3 * max_classes + H) * sizeof(float), s, vec, d_vec,
                       slot, pos, win_start, win_cnt, G, n, H, scale, loss, (const float *)nullptr, (const int32_t *)nullptr, 0, Gr.order, Gr.grp_off);
    STAIR_LAUNCH_CHECK();
    return 0;
}
int launch_loss_contrastive_table(const float *vec, float *d_vec, const int32_t *slot, const int32_t *pos_class, const int32_t *win_row,
                                  const float *presence, const float *reps, int n, int n_cls, int H, float scale, float *loss, hipStream_t s) {
    if (n == 0) return 0;
    STAIR_CHECK(H <= 512, "hidden size above 512");
    STAIR_CHECK(n_cls > 0 && (size_t)(3 * n_cls + H) * sizeof(float) <= 60 * 1024, "class table too large for the loss kernel's LDS (3 n_cls + H floats)");
    const LossGroups Gr = take_groups(n);
    hipLaunchKernelGGL(loss_contrastive_kernel, dim3(Gr.n_groups ? Gr.n_groups : n), dim3(kBlock), (size_t)(3 * n_cls + H) * sizeof(float), s, vec, d_vec,
                       slot, pos_class, (const int32_t *)nullptr, (const int32_t *)nullptr, reps, n, H, scale, loss, presence, win_row, n_cls, Gr.order, Gr.grp_off);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// FilterFrame criterion (:141-155): pretrain head Lin(H -> O) on every frame of map tile slot[i] (modules.py:381-414),
// softmax over the O object classes, BCELoss against the row-normalised interval masks gold [n][T][O] (built by the host
// from the gold dict: one span_to_attention column per entity), mean over T*O.  One workgroup (4 waves) per item:
//   phase 1  wave per frame: x[t][:] in registers, one butterfly reduction per class -> z[t][o] in LDS
//   phase 2  thread per frame: softmax, loss terms, dz = softmax' . dBCE (torch's backward: (p-g)/max(p(1-p),1e-12))
//   phase 3  dx[t][:] += dz[t][:] W  (wave per frame),  dW[o][:] += dz[:,o]^T x  (wave per class),  db += colsum(dz)
// len (optional): frames of item i's clip; the tile keeps the stride T, the criterion sees its first L frames (pred.size(0) of
// train_module.py:146 is the clip's own length), gold rows past L are ignored
__global__ __launch_bounds__(256) void loss_filterframe_kernel(const float *map, float *d_map, const int32_t *slot,
                                                               const float *gold, const float *W, const float *b, float *dW,
                                                               float *db, int T, int H, int O, float scale, float *loss,
                                                               const int32_t *len, const int32_t *order, const int32_t *grp_off,
                                                               long long *dW64, long long *db64) {
    extern __shared__ float z[];                 // [T][O]
    __shared__ float s_part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q0 = grp_off ? grp_off[blockIdx.x] : blockIdx.x, q1 = grp_off ? grp_off[blockIdx.x + 1] : blockIdx.x + 1;
  for (int q = q0; q < q1; ++q) {
    const int i = order ? order[q] : q;
    const float *x = map + (int64_t)slot[i] * T * H;
    const float *g = gold + (int64_t)i * T * O;
    const int nh = H >> 6;                       // floats per lane (H % 64 == 0, H <= 512)
    const int L = len ? len[i] : T;
    for (int e = threadIdx.x + L * O; e < T * O; e += blockDim.x) z[e] = 0.f;       // frames past the clip: no gradient
    for (int t = wave; t < L; t += 4) {
        float xr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[j] = j < nh ? x[(int64_t)t * H + j * 64 + lane] : 0.f;
        for (int o = 0; o < O; ++o) {
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nh) d += xr[j] * W[(int64_t)o * H + j * 64 + lane];
            d = wave_sum(d);
            if (lane == 0) z[t * O + o] = d + b[o];
        }
    }
    __syncthreads();
    const float inv = 1.0f / (float)(L * O);
    float part = 0.f;
    for (int t = threadIdx.x; t < L; t += blockDim.x) {
        float m = -INFINITY;
        for (int o = 0; o < O; ++o) m = fmaxf(m, z[t * O + o]);
        float sum = 0.f;
        for (int o = 0; o < O; ++o) sum += expf(z[t * O + o] - m);
        float dot = 0.f;                          // sum_j p_j dL/dp_j
        for (int o = 0; o < O; ++o) {
            const float p = expf(z[t * O + o] - m) / sum, gg = g[t * O + o];
            part += -(gg * fmaxf(logf(p), -100.f) + (1.f - gg) * fmaxf(logf(1.f - p), -100.f));
            dot += p * (p - gg) / fmaxf(p * (1.f - p), 1e-12f);
        }
        for (int o = 0; o < O; ++o) {
            const float p = expf(z[t * O + o] - m) / sum, gg = g[t * O + o];
            z[t * O + o] = scale * inv * p * ((p - gg) / fmaxf(p * (1.f - p), 1e-12f) - dot);     // d loss / d logit
        }
    }
    part = wave_sum(part);
    if (lane == 0) s_part[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0) loss[i] = (s_part[0] + s_part[1] + s_part[2] + s_part[3]) * inv;      // in wave order
    if (d_map) {
        float *dx = d_map + (int64_t)slot[i] * T * H;
        for (int t = wave; t < T; t += 4) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int o = 0; o < O; ++o) {
                const float dz = z[t * O + o];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < nh) acc[j] += dz * W[(int64_t)o * H + j * 64 + lane];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nh) arena_add(dx + (int64_t)t * H + j * 64 + lane, acc[j], grp_off != nullptr);
        }
    }
    if (dW) {
        for (int o = wave; o < O; o += 4) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < T; ++t) {
                const float dz = z[t * O + o];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < nh) acc[j] += dz * x[(int64_t)t * H + j * 64 + lane];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nh) grad_add(dW, dW64, (int64_t)o * H + j * 64 + lane, acc[j]);
        }
    }
    if (db)
        for (int o = threadIdx.x; o < O; o += blockDim.x) {
            float sdz = 0.f;
            for (int t = 0; t < T; ++t) sdz += z[t * O + o];
            grad_add(db, db64, o, sdz);
        }
    __syncthreads();               // the next item of the group reuses z
  }
}
int launch_loss_filterframe(const float *map, float *d_map, const int32_t *slot, const float *gold, const float *W, const float *b,
                            float *dW, float *db, int n, int T, int H, int O, float scale, float *loss, hipStream_t s, const int32_t *len) {
    STAIR_CHECK(H % 64 == 0 && H <= 512, "hidden size must be a multiple of 64, at most 512");
    STAIR_CHECK(T > 0 && O > 0 && (int64_t)T * O * 4 <= 60 * 1024, "T * object_types too large for the LDS tile");
    if (n == 0) return 0;
    const LossGroups G = take_groups(n);
    hipLaunchKernelGGL(loss_filterframe_kernel, dim3(G.n_groups ? G.n_groups : n), dim3(256), (size_t)T * O * sizeof(float), s, map, d_map, slot, gold, W, b,
                       dW, db, T, H, O, scale, loss, len, G.order, G.grp_off, dW ? det_shadow(dW) : nullptr, db ? det_shadow(db) : nullptr);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// 'cont-valid' score of the validation loop (train_module.py:127-132): cosine between a contrastive module's output
// vec[slot[i]] and the MEAN of the question's own gold class representations reps[seg_off[i] .. seg_off[i+1]) (L2-normalised
// rows), nn.CosineSimilarity(dim=0, eps=1e-8); an empty gold list scores 0.  One wave per item.
__global__ void cosine_to_mean_kernel(const float *vec, const int32_t *slot, const float *reps, const int32_t *seg_off, float *out,
                                      int n, int H) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const int r0 = seg_off[i], r1 = seg_off[i + 1];
    if (r1 <= r0) {
        if (lane == 0) out[i] = 0.0f;
        return;
    }
    const float *x = vec + (int64_t)slot[i] * H;
    const float inv = 1.0f / (float)(r1 - r0);
    float dot = 0.f, nx = 0.f, nm = 0.f;
    for (int c = lane; c < H; c += 64) {
        float m = 0.f;
        for (int r = r0; r < r1; ++r) m += reps[(int64_t)r * H + c];
        m *= inv;
        const float v = x[c];
        dot += v * m; nx += v * v; nm += m * m;
    }
    dot = wave_sum(dot); nx = wave_sum(nx); nm = wave_sum(nm);
    if (lane == 0) out[i] = dot / (fmaxf(sqrtf(nx), 1e-8f) * fmaxf(sqrtf(nm), 1e-8f));
}

int launch_cosine_to_mean(const float *vec, const int32_t *slot, const float *reps, const int32_t *seg_off, float *out, int n, int H,
                          hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(cosine_to_mean_kernel, dim3((n + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, s, vec, slot, reps, seg_off,
                       out, n, H);
    STAIR_LAUNCH_CHECK();
    return 0;
}

}  // namespace stair

extern "C" int stair_loss_groups(const int32_t *order, const int32_t *grp_off, int32_t n_groups) {
    STAIR_CHECK((order && grp_off && n_groups > 0) || n_groups == 0, "order, grp_off and n_groups > 0 (or n_groups == 0: no grouping)");
    stair::tl_groups.order = n_groups ? order : nullptr; stair::tl_groups.grp_off = n_groups ? grp_off : nullptr; stair::tl_groups.n_groups = n_groups;
    return 0;
}
extern "C" int stair_loss_attention(const float *att, float *d_att, const int32_t *slot, const int32_t *K,
                                    const int32_t *iv_off, const double *intervals, int32_t n, int32_t T, float scale,
                                    float *loss, stair_stream stream) {
    return stair::launch_loss_attention(att, d_att, slot, K, iv_off, intervals, n, T, scale, loss, static_cast<hipStream_t>(stream));
}
extern "C" int stair_loss_attention_len(const float *att, float *d_att, const int32_t *slot, const int32_t *K,
                                        const int32_t *iv_off, const double *intervals, const int32_t *len, int32_t n, int32_t T,
                                        float scale, float *loss, stair_stream stream) {
    return stair::launch_loss_attention(att, d_att, slot, K, iv_off, intervals, n, T, scale, loss, static_cast<hipStream_t>(stream), len);
}
extern "C" int stair_loss_head(int32_t nout, const float *vec, float *d_vec, const int32_t *slot, const int32_t *label,
                               const float *W, const float *b, float *dW, float *db, int32_t n, int32_t H, float scale,
                               float *loss, stair_stream stream) {
    return stair::launch_loss_head(nout, vec, d_vec, slot, label, W, b, dW, db, n, H, scale, loss, static_cast<hipStream_t>(stream));
}
extern "C" int stair_loss_contrastive(const float *vec, float *d_vec, const int32_t *slot, const int32_t *pos,
                                      const int32_t *win_start, const int32_t *win_cnt, const float *G, int32_t n, int32_t H,
                                      int32_t max_classes, float scale, float *loss, stair_stream stream) {
    return stair::launch_loss_contrastive(vec, d_vec, slot, pos, win_start, win_cnt, G, n, H, max_classes, scale, loss,
                                          static_cast<hipStream_t>(stream));
}
extern "C" int stair_loss_contrastive_table(const float *vec, float *d_vec, const int32_t *slot, const int32_t *pos_class,
                                            const int32_t *win_row, const float *presence, const float *reps, int32_t n,
                                            int32_t n_cls, int32_t H, float scale, float *loss, stair_stream stream) {
    STAIR_CHECK(vec && slot && pos_class && win_row && presence && reps && loss, "null argument");
    return stair::launch_loss_contrastive_table(vec, d_vec, slot, pos_class, win_row, presence, reps, n, n_cls, H, scale, loss,
                                                static_cast<hipStream_t>(stream));
}
extern "C" int stair_loss_filterframe(const float *map, float *d_map, const int32_t *slot, const float *gold, const float *W,
                                      const float *b, float *dW, float *db, int32_t n, int32_t T, int32_t H, int32_t O,
                                      float scale, float *loss, stair_stream stream) {
    return stair::launch_loss_filterframe(map, d_map, slot, gold, W, b, dW, db, n, T, H, O, scale, loss,
                                          static_cast<hipStream_t>(stream), nullptr);
}
extern "C" int stair_loss_filterframe_len(const float *map, float *d_map, const int32_t *slot, const float *gold, const float *W,
                                          const float *b, float *dW, float *db, const int32_t *len, int32_t n, int32_t T, int32_t H,
                                          int32_t O, float scale, float *loss, stair_stream stream) {
    return stair::launch_loss_filterframe(map, d_map, slot, gold, W, b, dW, db, n, T, H, O, scale, loss,
                                          static_cast<hipStream_t>(stream), len);
}
extern "C" int stair_loss_decoder_ce(const float *logits, const int32_t *answers, float *loss, int32_t n, int32_t A, stair_stream stream) {
    STAIR_CHECK(logits && answers && loss && n >= 0 && A > 0, "bad argument");
    return stair::launch_ce_loss(logits, answers, 0.0f, loss, nullptr, n, A, static_cast<hipStream_t>(stream));
}
extern "C" int stair_score_cosine_to_mean(const float *vec, const int32_t *slot, const float *reps, const int32_t *seg_off, float *out,
                                          int32_t n, int32_t H, stair_stream stream) {
    STAIR_CHECK(vec && slot && seg_off && out && n >= 0 && H > 0, "bad argument");
    return stair::launch_cosine_to_mean(vec, slot, reps, seg_off, out, n, H, static_cast<hipStream_t>(stream));
}
