// Context, program-plan builder and plan runner: the batched form of the stack interpreter in
// /root/reference/video_nmn/module_net.py:94-138.
//
// The reference walks ONE question's prefix program token by token and launches 50-240 tiny ATen
// kernels per question.  Here n questions are compiled together: every token becomes a node with a
// value kind and an arena slot, nodes are levelled exactly as utils/program_parser.py:307-321
// (leaf 0, module 1 + max(children)), and all nodes sharing (level, module, keyword variant) are
// packed into ONE launch sequence whose operands are gathered through int32 slot arrays.  Launch
// count per batch is O(levels x module kinds), independent of n.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <unordered_map>
#include <vector>

#include <chrono>

#include "ops.h"

namespace stair {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

bool g_acct_on = false;
thread_local const Policy *tl_policy = nullptr;
struct AcctEntry { int64_t launches = 0, bytes = 0, flops = 0; };
static std::map<std::string, AcctEntry> g_acct;       // kernel -> (launches, algorithmic bytes, algorithmic flops)
void acct_add(const char *kernel, int64_t bytes, int64_t flops) {
    auto &e = g_acct[kernel];
    e.launches += 1;
    e.bytes += bytes;
    e.flops += flops;
}

}  // namespace stair

using namespace stair;

// =============================================================================================
// context
// =============================================================================================
struct stair_ctx {
    stair_config cfg;
    bool conv;
    int ksize;
    std::vector<std::string> names;
    std::vector<int64_t> numel;
    std::vector<const float *> ptr;
    std::vector<float *> gptr;           // gradient buffers (training), same ids
    std::unordered_map<std::string, int> by_name;
    // backward pass: the per-weight gradient products are leaves of the graph, they run on a second stream beside BPTT
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_fork_enc = nullptr, ev_join_enc = nullptr;     // the text encoder's backward beside the video encoder's
    // 64-bit fixed-point shadows of the weight gradients (common.h det_shadow): owned by the context, all zero between two backward
    // passes (the flush empties what it adds), so a pass does not start by clearing 8 bytes per parameter; `dirty`: a pass that did not
    // reach its end left something behind -> the next one clears first
    void *gshadow = nullptr;
    int64_t gshadow_elems = 0;
    bool gshadow_dirty = false;
    Policy policy;                       // stair_ctx_set_option: this context's overrides of the process-wide settings
    ~stair_ctx() {
        if (gshadow) (void)hipFree(gshadow);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (ev_fork_enc) (void)hipEventDestroy(ev_fork_enc);
        if (ev_join_enc) (void)hipEventDestroy(ev_join_enc);
        if (side) (void)hipStreamDestroy(side);
    }

    void add(const std::string &name, int64_t n) {
        by_name[name] = (int)names.size();
        names.push_back(name);
        numel.push_back(n);
        ptr.push_back(nullptr);
        gptr.push_back(nullptr);
    }
    void lin(const std::string &prefix, int64_t out, int64_t in) {
        add(prefix + ".weight", out * in);
        add(prefix + ".bias", out);
    }
    const float *find(const std::string &name) const {
        auto it = by_name.find(name);
        return it == by_name.end() ? nullptr : ptr[it->second];
    }
    float *findg(const std::string &name) const {
        auto it = by_name.find(name);
        return it == by_name.end() ? nullptr : gptr[it->second];
    }
};

// python's round(): round half to even
static int py_round(double x) {
    double f = std::floor(x);
    double d = x - f;
    if (d > 0.5) return (int)f + 1;
    if (d < 0.5) return (int)f;
    return ((long long)f % 2 == 0) ? (int)f : (int)f + 1;
}

// mirrors stair_amd/spec.py::weight_table (reference state_dict order, aliases excluded)
static void build_weight_table(stair_ctx *c) {
    const stair_config &g = c->cfg;
    const int64_t H = g.hidden_size, V = g.video_size, E = g.text_size, A = g.answer_vocab_length;
    const int64_t L = g.max_video_length, O = g.object_types, Hh = H / 2;
    const bool heads = g.have_pretrain_head != 0;
    const std::string p = "submodules.";
    c->lin(p + "Compare.param.0", H, 2 * H);
    c->lin(p + "Equals.param.0", H, 2 * H);
    if (heads) c->lin(p + "Equals.pretrain_head", 1, H);
    c->lin(p + "Exists.param.0", H, 3 * H);
    c->lin(p + "Exists.param.3", H, H);
    if (heads) c->lin(p + "Exists.pretrain_head", 2, H);
    for (const char *kw : {"representation", "actions", "objects", "relations"}) {
        c->lin(p + "Filter.param." + kw + ".0", H, H);
        c->lin(p + "Filter.param." + kw + ".3", H, H);
    }
    c->lin(p + "Filter.attention.0", 1, 2 * H);
    c->lin(p + "Filter.dense.0", H, H);
    for (const char *kw : {"representation", "relations", "actions"}) {
        c->lin(p + "FilterFrame.param." + kw + ".0", H, H);
        c->lin(p + "FilterFrame.param." + kw + ".3", H, H);
    }
    c->lin(p + "FilterFrame.attention.0", 1, 2 * H);
    c->lin(p + "FilterFrame.dense.0", H, H);
    if (heads) c->lin(p + "FilterFrame.pretrain_head", O, H);
    c->lin(p + "HasItem.param.0", H, H);
    c->lin(p + "HasItem.param.3", 1, H);
    c->lin(p + "Localize.video_linear.0", H, H);
    c->lin(p + "Localize.video_linear.3", H, H);
    c->lin(p + "Localize.keyword_linear.0", H, H);
    c->add(p + "Relate.beta", L);
    c->lin(p + "Superlative.dense.0", H, H);
    for (const char *mode : {"before", "after", "between"}) {
        for (int layer : {0, 2, 4}) {
            const std::string pre = p + "Temporal.relate." + mode + "." + std::to_string(layer);
            if (c->conv) {
                const int64_t ks = layer == 4 ? 2 * c->ksize + 1 : c->ksize;
                c->add(pre + ".weight", ks);
                c->add(pre + ".bias", 1);
            } else {
                c->lin(pre, L, L);
            }
        }
    }
    c->lin(p + "Temporal.dense.0", H, H);
    c->add(p + "Temporal.layer_norm.weight", H);
    c->add(p + "Temporal.layer_norm.bias", H);
    c->lin(p + "ToAction.param.0", H, 2 * H);
    c->lin(p + "ToAction.param.3", H, H);
    c->lin(p + "Xor.param.0", H, 3 * H);
    if (heads) c->lin(p + "Xor.pretrain_head", 2, H);
    for (int enc = 0; enc < 2; ++enc) {
        const std::string e = p + (enc == 0 ? "video_encoder" : "text_encoder");
        const int64_t in = enc == 0 ? V : E;
        for (const char *sfx : {"", "_reverse"}) {
            c->add(e + ".weight_ih_l0" + sfx, 4 * Hh * in);
            c->add(e + ".weight_hh_l0" + sfx, 4 * Hh * Hh);
            c->add(e + ".bias_ih_l0" + sfx, 4 * Hh);
            c->add(e + ".bias_hh_l0" + sfx, 4 * Hh);
        }
    }
    c->lin(p + "decoder.0", 2 * H, 2 * H);
    c->lin(p + "decoder.3", A, 2 * H);
}

extern "C" int stair_abi_version(void) { return STAIR_ABI_VERSION; }
extern "C" void stair_acct_enable(int32_t on) { g_acct_on = on != 0; if (on) g_acct.clear(); }
extern "C" int stair_acct_dump(char *buf, int32_t cap) {        // "kernel launches bytes flops\n" lines; returns the length needed
    std::string out;
    for (auto &kv : g_acct)
        out += kv.first + " " + std::to_string(kv.second.launches) + " " + std::to_string(kv.second.bytes) + " " + std::to_string(kv.second.flops) + "\n";
    if (buf && cap > 0) { const size_t n = std::min<size_t>(out.size(), (size_t)cap - 1); memcpy(buf, out.data(), n); buf[n] = 0; }
    return (int)out.size() + 1;
}
extern "C" const char *stair_last_error(void) { return g_err.c_str(); }

extern "C" int stair_ctx_create(const stair_config *cfg, stair_ctx **out) {
    STAIR_CHECK(cfg && out, "null argument");
    STAIR_CHECK(cfg->hidden_size >= 64 && cfg->hidden_size % 64 == 0 && cfg->hidden_size <= 512,
                "hidden_size must be a multiple of 64, at most 512 (LSTM recurrence tiles Hh = H/2 in blocks of 32)");
    STAIR_CHECK(cfg->video_size > 0 && cfg->video_size % 4 == 0, "video_size must be a multiple of 4");
    STAIR_CHECK(cfg->text_size > 0 && cfg->text_size % 4 == 0, "text_size must be a multiple of 4");
    STAIR_CHECK(cfg->answer_vocab_length > 0 && cfg->max_video_length > 0 && cfg->object_types > 0, "bad config");
    auto c = std::make_unique<stair_ctx>();
    c->cfg = *cfg;
    c->conv = cfg->max_video_length > 32;                       // modules.py:255
    c->ksize = py_round(cfg->max_video_length / 4.0);           // modules.py:258
    build_weight_table(c.get());
    *out = c.release();
    return 0;
}
extern "C" void stair_ctx_destroy(stair_ctx *ctx) { delete ctx; }
extern "C" int stair_weight_count(const stair_ctx *ctx) { return ctx ? (int)ctx->names.size() : 0; }
extern "C" const char *stair_weight_name(const stair_ctx *ctx, int id) {
    return (ctx && id >= 0 && id < (int)ctx->names.size()) ? ctx->names[id].c_str() : nullptr;
}
extern "C" int64_t stair_weight_numel(const stair_ctx *ctx, int id) {
    return (ctx && id >= 0 && id < (int)ctx->names.size()) ? ctx->numel[id] : -1;
}
extern "C" int stair_ctx_set_weight(stair_ctx *ctx, int id, const float *dev_ptr, int64_t numel) {
    STAIR_CHECK(ctx, "null ctx");
    STAIR_CHECK(id >= 0 && id < (int)ctx->names.size(), "weight id out of range");
    STAIR_CHECK(numel == ctx->numel[id], "numel mismatch for " + ctx->names[id]);
    STAIR_CHECK(dev_ptr && (reinterpret_cast<uintptr_t>(dev_ptr) & 15) == 0, "weight pointer must be 16-byte aligned: " + ctx->names[id]);
    ctx->ptr[id] = dev_ptr;
    return 0;
}

extern "C" int stair_ctx_set_grad(stair_ctx *ctx, int id, float *dev_ptr, int64_t numel) {
    STAIR_CHECK(ctx, "null ctx");
    STAIR_CHECK(id >= 0 && id < (int)ctx->names.size(), "weight id out of range");
    STAIR_CHECK(numel == ctx->numel[id], "numel mismatch for " + ctx->names[id]);
    STAIR_CHECK(dev_ptr && (reinterpret_cast<uintptr_t>(dev_ptr) & 15) == 0, "gradient pointer must be 16-byte aligned: " + ctx->names[id]);
    ctx->gptr[id] = dev_ptr;
    return 0;
}

// =============================================================================================
// plan
// =============================================================================================
namespace {

constexpr int64_t kSplitKFloats = 16ll * 64 * 128 * 128;     // split-K scratch: 16 K pieces x 64 output tiles of 128 x 128

// how a fused tile launch deals its tiles out: the self-resetting atomic queue (default) or a static round robin
// (stair_set_tile_queue(0) / STAIR_TILE_QUEUE=0); both run eagerly and inside a stream capture
int g_tile_queue = -1;
bool tile_queue_on() {
    static const bool env_on = [] { const char *e = getenv("STAIR_TILE_QUEUE"); return !(e && e[0] == '0'); }();
    return policy_or(STAIR_OPT_TILE_QUEUE, g_tile_queue >= 0 ? g_tile_queue : (env_on ? 1 : 0)) != 0;
}

// Diagnostic twin of the tile operator's work queue (stair_debug_queue_probe): the same ticket protocol, no tile work, every
// ticket a workgroup sees is RECORDED instead of used as an index -- safe whatever the words hold.
__global__ void queue_probe_kernel(unsigned *counter, unsigned *seen, int total, int self_reset) {
    unsigned first = 0xffffffffu, taken = 0;
    if (threadIdx.x == 0) {
        for (;;) {
            const unsigned w = atomicAdd(counter, 1u);
            if (first == 0xffffffffu) first = w;
            if (w >= (unsigned)total) break;
            ++taken;
        }
        seen[2 * blockIdx.x] = first;
        seen[2 * blockIdx.x + 1] = taken;
        if (self_reset && atomicAdd(counter + 1, 1u) == gridDim.x - 1) {
            atomicExch(counter, 0u);
            atomicExch(counter + 1, 0u);
        }
    }
}


constexpr int OP_SPAN = 50;      // pseudo op: span mean (level 0)
const int kArity[STAIR_OP_COUNT] = {2, 2, 3, 2, 2, 2, 2, 2, 2, 1, 2, 2, 3, 3, 2, 2, 2, 2};
const char *kOpName[STAIR_OP_COUNT] = {"And", "AttnVideo", "Choose", "Compare", "Equals", "Exists", "ExistsFrame",
                                       "Filter", "FilterFrame", "HasItem", "Localize", "Relate", "Superlative",
                                       "Temporal", "ToAction", "Xor", "XorFrame", "Array2"};

struct Node {
    int kind = -1, slot = -1, aux = -1, level = 0, rel = -1;
    int bop = -1, bvariant = 0, bsub = 0, inst = -1;      // the bucket (level, bop, bvariant, bsub) that computes it and its instance there
};

// The module-level Linear layers that act on [T, H] tiles, as one table (fused tile operators pack their planes by these
// ids; the backward pass groups the weight-gradient products by them).
enum { WF_F0 = 0, WF_F3 = 4, WF_FF0 = 8, WF_FF3 = 11, WF_FFD = 14, WF_HI0 = 15, WF_LV0 = 16, WF_LV3 = 17, WF_TD = 18, WF_COUNT = 19 };
// plane images of the vector-level modules' weights (forward only): one [H, H] image per H-wide column block of a first layer
enum { WV_CMP = 19, WV_EQ = 21, WV_XOR = 23, WV_TA0 = 26, WV_TA3 = 28, WV_EX0 = 29, WV_EX3 = 32,
       WV_FD = 33, WV_LK = 34, WV_DEC0 = 35 /* four [512 x 512] blocks: (output block, input segment) */, WV_END = 39 };

struct Bucket {
    int64_t dzA = -1, dzB = -1;   // training: this bucket's blocks inside the per-WEIGHT dZ regions (first / second layer of its tile MLP)
    int64_t dzC = -1, gRow = -1;  // training, fused backward: FilterFrame's third dZ tile set; Filter's per-instance gradient row [c, H]
    int64_t dzV0 = -1, dzV3 = -1; // training: this bucket's rows inside the per-weight dZ regions of its vector-level weights
    int level, op, variant, sub;
    int cnt = 0;        // instances
    int nrows = 0;      // secondary count (Localize pairs / Superlative action rows)
    std::vector<int32_t> col[8];
    int64_t off[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // offsets into the device idx buffer
    // training: where the GRADIENT of operand column c goes (same slot numbering; an operand slot that has several consumers at one
    // level sends all but the first to private staging slots behind the gradient arena, see build_grad_fanin); goff[c] = off[c]
    // where a column has no redirected copy
    std::vector<int32_t> gcol[8];
    int64_t goff[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // float offsets of this bucket's intermediates.  Inference: all buckets share one scratch set;
    // training: private regions, kept until stair_plan_backward has consumed them.
    int64_t svA = 0, svB = 0, svK = 0, svCat = 0, svHid = 0, svRs = 0, svSup = 0, svExtra = 0;
    // training, fused tile operators: relu' of the first / second / third layer's activation as one bit per element ([cnt][512] 64-bit
    // words, written by the forward launch, read by the backward chain instead of the 128 KB activation; csrc/tile_mlp.hip save_bits)
    int64_t bitA = -1, bitB = -1, bitC = -1;
};

// Vector-level weights whose gradient is ONE product per weight over all buckets that use it (training plans): the dZ rows
// and the saved inputs of every bucket lie back to back in per-weight regions, like the map level's.  K = columns of the input.
enum { VD_CMP = 0, VD_EQ, VD_XOR, VD_TA0, VD_TA3, VD_EX0, VD_EX3, VD_FD, VD_COUNT };
inline int vd_cols(int w, int H) { return (w == VD_XOR || w == VD_EX0) ? 3 * H : (w == VD_CMP || w == VD_EQ || w == VD_TA0) ? 2 * H : H; }
// first-layer (reads the packed operands; Filter: the pooled rows) and second-layer (reads the saved hidden rows) weight of a bucket
inline void bucket_vec_weights(int op, int &v0, int &v3) {
    v0 = v3 = -1;
    switch (op) {
        case STAIR_OP_COMPARE: v0 = VD_CMP; break;
        case STAIR_OP_EQUALS: v0 = VD_EQ; break;
        case STAIR_OP_XOR: v0 = VD_XOR; break;
        case STAIR_OP_TOACTION: v0 = VD_TA0; v3 = VD_TA3; break;
        case STAIR_OP_EXISTS: v0 = VD_EX0; v3 = VD_EX3; break;
        case STAIR_OP_FILTER: v0 = VD_FD; break;
        default: break;
    }
}

// first-layer weight (reads the module's input tile) and second-layer weight (reads the saved first activation) of a bucket's
// tile MLP, as WF_* ids; -1 = none
void bucket_weights(const Bucket &b, int &w0, int &w3) {
    w0 = w3 = -1;
    switch (b.op) {
        case STAIR_OP_FILTER: w0 = WF_F0 + b.variant; w3 = WF_F3 + b.variant; break;
        case STAIR_OP_FILTERFRAME: w0 = WF_FF0 + b.variant; w3 = WF_FF3 + b.variant; break;
        case STAIR_OP_HASITEM: w0 = WF_HI0; break;
        case STAIR_OP_LOCALIZE: case STAIR_OP_SUPERLATIVE: w0 = WF_LV0; w3 = WF_LV3; break;
        case STAIR_OP_TEMPORAL: w0 = WF_TD; break;
        default: break;
    }
}

}  // namespace

// Pinned staging of a plan's index image.  The image used to be copied from the plan's pageable std::vector: a pageable
// hipMemcpyAsync stalls the host on the stream and leaves the source's lifetime to the runtime's own staging.  Now the
// copy reads page-locked memory and an event marks its completion; stair_plan_destroy parks the buffer until that event has
// completed, then it goes back to a small process-wide pool (so a step does not pay a hipHostMalloc).
namespace {
struct PinnedBuf { int32_t *p = nullptr; size_t cap = 0; };
std::mutex g_pin_mu;
std::vector<PinnedBuf> g_pin_pool;
// Regions of the encoders' input projections, as offsets in floats from a base: [xproj video | xproj text | bias sums | video W_ih planes |
// text W_ih planes | text row planes].  A function of the batch SHAPE only (clips, frames, token rows): the same layout inside a plan's
// workspace and in a caller-owned buffer filled by stair_encoders_project before the plan is built.
struct ProjLayout { int64_t xpv, xpt, bias, wplanes, wplanes_t, xplanes_t, total; };
static ProjLayout proj_layout(const stair_config &g, int64_t n_vid, int64_t T, int64_t rows_q) {
    const int64_t H = g.hidden_size, Ep = (g.text_size + 31) / 32 * 32;
    ProjLayout L;
    int64_t at = 0;
    auto take = [&](int64_t len) { const int64_t o = at; at += (len + 63) / 64 * 64; return o; };
    L.xpv = take(n_vid * T * 4 * H);
    L.xpt = take(rows_q * 4 * H);
    L.bias = take(2 * 4 * H);
    L.wplanes = g.video_size % 32 == 0 ? take(4 * H * g.video_size) : 0;
    L.wplanes_t = take(4 * H * Ep);
    L.xplanes_t = take(std::max<int64_t>(rows_q, 1) * Ep);
    L.total = at;
    return L;
}

PinnedBuf pinned_take(size_t ints) {
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (size_t i = 0; i < g_pin_pool.size(); ++i)
            if (g_pin_pool[i].cap >= ints) {
                PinnedBuf b = g_pin_pool[i];
                g_pin_pool.erase(g_pin_pool.begin() + i);
                return b;
            }
    }
    PinnedBuf b;
    b.cap = std::max<size_t>(ints * 5 / 4, 1 << 14);
    if (hipHostMalloc(reinterpret_cast<void **>(&b.p), b.cap * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) b.p = nullptr;
    return b;
}
void pinned_give(PinnedBuf b) {
    if (!b.p) return;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (g_pin_pool.size() < 8) { g_pin_pool.push_back(b); return; }
    // full: keep the larger buffers (a process that moves on to bigger batches would otherwise free and re-pin a buffer of the
    // new size on every step, milliseconds each, while the pool holds eight that are too small)
    size_t k = 0;
    for (size_t i = 1; i < g_pin_pool.size(); ++i)
        if (g_pin_pool[i].cap < g_pin_pool[k].cap) k = i;
    if (g_pin_pool[k].cap < b.cap) std::swap(g_pin_pool[k], b);
    (void)hipHostFree(b.p);
}
// A destroyed plan's buffer may still be the source of an upload the GPU has not reached (the host runs ahead of the
// stream): it is parked with its event and goes back to the pool once the event has completed -- checked, never waited for,
// so that destroying a plan does not tie the host to the GPU's progress.
struct ParkedBuf { PinnedBuf b; hipEvent_t ev; };
std::vector<ParkedBuf> g_parked;
void pinned_reclaim() {                 // g_pin_mu NOT held
    std::vector<ParkedBuf> done;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (size_t i = 0; i < g_parked.size();)
            if (hipEventQuery(g_parked[i].ev) != hipErrorNotReady) { done.push_back(g_parked[i]); g_parked.erase(g_parked.begin() + i); }
            else ++i;
    }
    for (const ParkedBuf &d : done) { (void)hipEventDestroy(d.ev); pinned_give(d.b); }
}
void pinned_park(PinnedBuf b, hipEvent_t ev) {
    if (!ev) { pinned_give(b); return; }
    if (hipEventQuery(ev) != hipErrorNotReady) { (void)hipEventDestroy(ev); pinned_give(b); return; }
    ParkedBuf oldest{};
    bool wait_oldest = false;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        g_parked.push_back({b, ev});
        if (g_parked.size() > 4) {          // the host is more than 4 plans ahead of the stream: let it wait for the oldest one
            oldest = g_parked.front();      // (bounds the page-locked memory in flight; no allocation churn in a long run)
            g_parked.erase(g_parked.begin());
            wait_oldest = true;
        }
    }
    if (wait_oldest) {
        (void)hipEventSynchronize(oldest.ev);
        (void)hipEventDestroy(oldest.ev);
        pinned_give(oldest.b);
    }
}
}  // namespace

struct stair_plan {
    stair_config cfg;
    PinnedBuf pin;                  // page-locked copy of idx, made at the first upload
    hipEvent_t pin_ev = nullptr;    // recorded after every upload from `pin`
    ~stair_plan() { pinned_park(pin, pin_ev); }
    int n = 0, n_vid = 0, T = 0, rows_q = 0, max_q = 0;   // n_vid distinct videos (== n unless questions share them)
    std::vector<Node> nodes;
    std::vector<Bucket> buckets;
    std::vector<int32_t> roots;
    std::vector<int32_t> idx;       // host image of the device index buffer
    int64_t off_seqv = 0, off_seqt = 0, off_roots = 0, off_lenv = 0;
    bool ragged = false;            // clips of different frame counts in this batch (padded to T; per-instance lengths in col[6])
    bool bits_written = false;      // the last forward run of this (training) plan went through the fused tile operators: Bucket::bit* are valid
    std::vector<int32_t> vlen;      // frames per clip [n_vid]
    int n_vec = 0, n_map = 0, n_att = 0, n_aliased = 0;     // n_aliased: nodes that share another node's value (common subexpressions)
    int maxI = 0, maxV = 0, maxK = 0, maxSupRows = 0, n_levels = 0;
    // deterministic gradient fan-in (training plans): staging slots behind the gradient arenas and, per producer level, the table of
    // (kind, destination, first staging slot, count, rows) that stair_plan_backward sums in order before it walks that level
    int n_vec_stage = 0, n_map_stage = 0, n_att_stage = 0;
    std::vector<int32_t> groots;                 // gradient rows of the program roots (decoder)
    int64_t off_groots = 0;
    std::vector<int32_t> fanin;                  // 5 ints per entry
    std::vector<int> fanin_first, fanin_count;   // per level
    int64_t off_fanin = 0;
    int64_t o_gshadow = 0;
    std::vector<int32_t> tok_ptr, tok_span;      // training: for every question-token row the span means it belongs to (CSR over the span bucket's instances)
    int64_t off_tok_ptr = 0, off_tok_span = 0;                       // 64-bit fixed-point shadows of every weight gradient (two floats per element)
    int64_t coop_bytes = 0;
    // training: weight-gradient products grouped per WEIGHT (WF_* ids).  Every bucket that uses a weight writes its dZ into its
    // block of wg_dz[w]; the matching X operand is the input tiles gathered through wg_off_idx[w] (first-layer weights) or the
    // saved first activations, which lie in the same order in wg_sx[w] (second-layer weights).  ONE long-reduction TN GEMM per
    // weight at the end of stair_plan_backward instead of one per bucket.
    int64_t vd_rows[VD_COUNT] = {}, vd_dz[VD_COUNT] = {}, vd_x[VD_COUNT] = {};
    int64_t o_tnring = 0, tnring_floats = 0, o_tnenc[2] = {0, 0}, tnenc_floats[2] = {0, 0};
    int64_t wg_rows[WF_COUNT] = {}, wg_dz[WF_COUNT] = {}, wg_sx[WF_COUNT] = {}, wg_off_idx[WF_COUNT] = {}, wg_off_rs[WF_COUNT] = {}, wg_part[WF_COUNT] = {};
    // workspace layout (float offsets)
    // STAIR_PLAN_EXT_PROJECTION: the encoders' input-projection regions live in a caller-owned buffer (stair_plan_set_projection) whose
    // layout depends on the batch shape alone (proj_layout), so that the projections can be enqueued BEFORE the plan exists
    bool ext_proj = false;
    float *proj = nullptr;
    int64_t o_idx = 0, o_vec = 0, o_map = 0, o_att = 0, o_tok = 0, o_qfeat = 0, o_vhn = 0, o_xpv = 0, o_xpt = 0,
            o_bias = 0, o_wpack = 0, o_wplanes = 0, o_wplanes_t = 0, o_xplanes_t = 0, o_coop = 0, o_coop2 = 0, o_splitk = 0, o_tmpA = 0, o_tmpB = 0, o_kbuf = 0, o_cat = 0, o_hid = 0, o_rs = 0, o_sup = 0, o_extra = 0,
            o_logits = 0, o_status = 0, o_wfrag = 0, total = 0;
    // training only
    bool train = false;
    void *bwd_event = nullptr;      // stair_plan_set_backward_event
    float drop_p = 0.0f;            // training-mode dropout (stair_plan_set_dropout); 0 = off
    uint64_t drop_seed = 0;
    int64_t o_cv = 0, o_ct = 0, o_hprev = 0, o_gblock = 0, o_gatt = 0, o_gtok = 0, o_gqfeat = 0, o_gA = 0, o_gB = 0,
            o_gK = 0, o_gV0 = 0, o_gV1 = 0, o_gCat = 0, o_gS = 0, o_gRs = 0, o_gRs2 = 0, o_gExtra = 0, o_gStats = 0,
            o_wt = 0, o_dlogits = 0, o_loss = 0, o_zero_beg = 0, o_zero_end = 0, o_wfragT = 0;
};

namespace {

// floats needed for the transposed weight images used by the dX products (all 2-D weights)
int64_t ctx_weight_floats(const stair_ctx *ctx) {
    int64_t t = 0;
    for (int64_t v : ctx->numel) t += align_up(v, 64);
    return t;
}

// Deterministic fan-in of the gradient arenas.  A value slot read by several nodes receives one gradient contribution per reader; the
// readers add with float atomics, so the order inside ONE launch is not defined.  Launches follow each other on the stream, hence an
// address that gets at most one contribution per launch has a defined sum.  Readers of a slot at different program levels are in
// different launches; readers at the SAME level may share one (all tile chains of a level are one launch, so are the grouped
// vector-level problems, and two instances of one bucket share every kernel of the bucket).  So: per (slot, level) the first reader
// adds into the slot itself, the j-th (j >= 1) into staging slot j - 1 of that slot (staging slots are shared across levels: different
// launches again), and before the backward pass walks the level that PRODUCED the slot the staging slots are added to it in index
// order by one small kernel (grad_fanin_kernel).  The columns below replace the operand columns wherever a gradient is scattered.
struct GEdge { int32_t slot; int level; int bidx, col, pos; int width; };
enum { FAN_VEC = 0, FAN_MAP = 1, FAN_ATT = 2 };

// vec_fix: where a vec-arena staging index was written with the provisional numbering n_vec + j; the layout pass moves the vec staging
// rows behind the map staging tiles (the gradient block mirrors [vec arena | map arena]) and adds the difference (fix_vec_staging)
int build_grad_fanin(stair_plan *pl, int T, std::vector<int32_t *> &vec_fix) {
    std::vector<GEdge> ev, em, ea;
    vec_fix.clear();
    const int top = pl->n_levels;                                         // the decoder reads the roots "above" every level
    for (int bi = 0; bi < (int)pl->buckets.size(); ++bi) {
        Bucket &b = pl->buckets[bi];
        for (int c = 0; c < 8; ++c) b.gcol[c].clear();
        auto all = [&](std::vector<GEdge> &dst, int c, int width = 1, const std::vector<int32_t> *w = nullptr) {
            b.gcol[c] = b.col[c];
            for (int p = 0; p < (int)b.col[c].size(); ++p) dst.push_back({b.col[c][p], b.level, bi, c, p, w ? (*w)[p] : width});
        };
        switch (b.op) {
            case STAIR_OP_AND: case STAIR_OP_XORFRAME:
                if (b.sub == STAIR_VAL_VEC) { all(ev, 0); all(ev, 1); } else { all(ea, 0); all(ea, 1); }
                break;
            case STAIR_OP_ATTNVIDEO: all(em, 0); all(ea, 1); break;
            case STAIR_OP_CHOOSE: all(ev, 0); all(ev, 1); break;
            case STAIR_OP_COMPARE: case STAIR_OP_EQUALS: case STAIR_OP_XOR: case STAIR_OP_TOACTION: case STAIR_OP_EXISTS:
                all(ev, 0); all(ev, 1); break;
            case STAIR_OP_EXISTSFRAME: all(ev, 0); all(em, 1); break;
            case STAIR_OP_FILTER: all(em, 0); break;
            case STAIR_OP_FILTERFRAME: all(em, 0); if (b.variant == 0) all(ev, 1); break;
            case STAIR_OP_HASITEM: all(em, 0); break;
            case STAIR_OP_LOCALIZE: all(em, 0); all(ev, 2); break;
            case STAIR_OP_RELATE: all(ea, 0); break;
            case STAIR_OP_TEMPORAL: all(em, 0); all(ea, 1, 1, &b.col[2]); break;
            case STAIR_OP_SUPERLATIVE: {
                all(em, 0);
                // action rows (before they are resolved to global row ids): >= 0 a vec row, < 0 row a of map tile -(rid + 1) / T;
                // a map tile is ONE operand of its instance (edge at its first row)
                b.gcol[4] = b.col[4];
                for (int p = 0; p < (int)b.col[4].size(); ++p) {
                    const int rid = b.col[4][p];
                    if (rid >= 0) ev.push_back({rid, b.level, bi, 4, p, 1});
                    else if ((-rid - 1) % T == 0) em.push_back({(-rid - 1) / T, b.level, bi, 4, p, 1});
                }
                break;
            }
            default: break;
        }
    }
    pl->groots = pl->roots;
    for (int q = 0; q < (int)pl->roots.size(); ++q) ev.push_back({pl->roots[q], top, -1, 0, q, 1});

    // producer level of every slot (clip tiles and anything not produced by a bucket: 0)
    std::vector<int> lv(pl->n_vec, 0), lm(pl->n_map, 0), la(std::max(pl->n_att, 1), 0);
    for (const Node &nd : pl->nodes) {
        if (nd.bop < 0) continue;
        if (nd.kind == STAIR_VAL_VEC) lv[nd.slot] = nd.level;
        else if (nd.kind == STAIR_VAL_MAP) lm[nd.slot] = nd.level;
        else if (nd.kind == STAIR_VAL_ATT || nd.kind == STAIR_VAL_FRAME) la[nd.slot] = nd.level;
    }
    pl->fanin.clear();
    std::vector<std::vector<int32_t>> per_level(pl->n_levels + 1);
    auto assign = [&](std::vector<GEdge> &E, int kind, int n_slots, int &n_stage, const std::vector<int> &plevel) {
        std::sort(E.begin(), E.end(), [](const GEdge &a, const GEdge &b) {
            return std::tie(a.slot, a.level, a.bidx, a.col, a.pos) < std::tie(b.slot, b.level, b.bidx, b.col, b.pos);
        });
        n_stage = 0;
        for (size_t i = 0; i < E.size();) {
            size_t e = i;
            while (e < E.size() && E[e].slot == E[i].slot) ++e;
            const int width = E[i].width;                                // rows per operand (Localize output read by Temporal: K)
            int most = 0;
            for (size_t a = i; a < e;) {                                  // per level: reader j -> staging j - 1
                size_t z = a;
                while (z < e && E[z].level == E[a].level) ++z;
                most = std::max(most, (int)(z - a) - 1);
                a = z;
            }
            const int base = n_slots + n_stage;                          // first staging slot (row, for the att arena) of this slot
            for (size_t a = i; a < e;) {
                size_t z = a;
                while (z < e && E[z].level == E[a].level) ++z;
                for (size_t k = a + 1; k < z; ++k) {
                    const GEdge &g = E[k];
                    const int32_t target = base + (int)(k - a - 1) * width;
                    if (g.bidx < 0) { pl->groots[g.pos] = target; vec_fix.push_back(&pl->groots[g.pos]); }
                    else if (g.col == 4 && pl->buckets[g.bidx].op == STAIR_OP_SUPERLATIVE) {
                        std::vector<int32_t> &gc = pl->buckets[g.bidx].gcol[4];
                        const std::vector<int32_t> &vc = pl->buckets[g.bidx].col[4];
                        if (kind == FAN_VEC) { gc[g.pos] = target; vec_fix.push_back(&gc[g.pos]); }
                        else for (int t = 0; t < T && g.pos + t < (int)vc.size() && vc[g.pos + t] == -(g.slot * T + t) - 1; ++t)
                            gc[g.pos + t] = -(target * T + t) - 1;                                  // the instance's rows of the tile follow each other
                    } else {
                        pl->buckets[g.bidx].gcol[g.col][g.pos] = target;
                        if (kind == FAN_VEC) vec_fix.push_back(&pl->buckets[g.bidx].gcol[g.col][g.pos]);
                    }
                }
                a = z;
            }
            if (most > 0) {
                const int L = std::min(std::max(plevel[E[i].slot], 0), pl->n_levels);
                per_level[L].insert(per_level[L].end(), {kind, E[i].slot, base, most, width});
                n_stage += most * width;
            }
            i = e;
        }
    };
    assign(ev, FAN_VEC, pl->n_vec, pl->n_vec_stage, lv);
    assign(em, FAN_MAP, pl->n_map, pl->n_map_stage, lm);
    assign(ea, FAN_ATT, pl->n_att, pl->n_att_stage, la);
    pl->fanin_first.assign(pl->n_levels + 1, 0);
    pl->fanin_count.assign(pl->n_levels + 1, 0);
    for (int L = 0; L <= pl->n_levels; ++L) {
        pl->fanin_first[L] = (int)pl->fanin.size() / 5;
        pl->fanin_count[L] = (int)per_level[L].size() / 5;
        pl->fanin.insert(pl->fanin.end(), per_level[L].begin(), per_level[L].end());
    }
    return 0;
}

struct Builder {
    stair_plan *pl;
    std::map<std::tuple<int, int, int, int>, int> index;
    Bucket &bucket(int level, int op, int variant, int sub) {
        auto key = std::make_tuple(level, op, variant, sub);
        auto it = index.find(key);
        if (it == index.end()) {
            Bucket b;
            b.level = level; b.op = op; b.variant = variant; b.sub = sub;
            pl->buckets.push_back(b);
            it = index.emplace(key, (int)pl->buckets.size() - 1).first;
        }
        return pl->buckets[it->second];
    }
};

std::string where(int q, int i, int tok) {
    std::string s = "question " + std::to_string(q) + ", token " + std::to_string(i);
    if (tok >= 0 && tok < STAIR_OP_COUNT) s += std::string(" (") + kOpName[tok] + ")";
    return s;
}

}  // namespace

extern "C" int stair_plan_build(stair_ctx *ctx, int32_t n, const int32_t *prog_off, const int32_t *tokens,
                                const int32_t *span_lo, const int32_t *span_hi, const int32_t *q_off, int32_t T,
                                int32_t flags, stair_plan **out) {
    return stair_plan_build_shared(ctx, n, prog_off, tokens, span_lo, span_hi, q_off, n, nullptr, T, flags, out);
}

extern "C" int stair_plan_build_shared(stair_ctx *ctx, int32_t n, const int32_t *prog_off, const int32_t *tokens,
                                       const int32_t *span_lo, const int32_t *span_hi, const int32_t *q_off,
                                       int32_t n_videos, const int32_t *video_of_question, int32_t T,
                                       int32_t flags, stair_plan **out) {
    return stair_plan_build_ragged(ctx, n, prog_off, tokens, span_lo, span_hi, q_off, n_videos, video_of_question, nullptr, T, flags, out);
}

extern "C" int stair_plan_build_ragged(stair_ctx *ctx, int32_t n, const int32_t *prog_off, const int32_t *tokens,
                                       const int32_t *span_lo, const int32_t *span_hi, const int32_t *q_off,
                                       int32_t n_videos, const int32_t *video_of_question, const int32_t *video_len, int32_t T,
                                       int32_t flags, stair_plan **out) {
    STAIR_CHECK(ctx && prog_off && tokens && span_lo && span_hi && q_off && out, "null argument");
    STAIR_CHECK(n > 0 && T > 0, "n and T must be positive");
    STAIR_CHECK(n_videos > 0 && (video_of_question || n_videos == n), "video_of_question is required when n_videos != n");
    if (video_of_question)
        for (int q = 0; q < n; ++q)
            STAIR_CHECK(video_of_question[q] >= 0 && video_of_question[q] < n_videos,
                        "video_of_question[" + std::to_string(q) + "] out of range");
    STAIR_CHECK(ctx->conv || T == ctx->cfg.max_video_length,
                "Linear(T,T) Temporal nets need T == max_video_length (modules.py:266-277)");
    auto plp = std::make_unique<stair_plan>();
    stair_plan *pl = plp.get();
    pl->cfg = ctx->cfg;
    pl->n = n;
    pl->T = T;
    pl->train = (flags & STAIR_PLAN_TRAIN) != 0;
    pl->ext_proj = (flags & STAIR_PLAN_EXT_PROJECTION) != 0;
    const int ntok = prog_off[n];
    pl->nodes.assign(ntok, Node());
    pl->roots.assign(n, -1);
    pl->n_vid = n_videos;
    pl->n_map = n_videos;   // map slot v = encoded video v
    pl->vlen.assign(n_videos, T);
    if (video_len)
        for (int v = 0; v < n_videos; ++v) {
            STAIR_CHECK(video_len[v] >= 1 && video_len[v] <= T, "video_len[" + std::to_string(v) + "] must be in 1..T");
            pl->vlen[v] = video_len[v];
            if (video_len[v] != T) pl->ragged = true;
        }
    STAIR_CHECK(!pl->ragged || ctx->conv, "clips of different lengths need the Conv1d Temporal nets (Linear(T,T) fixes T, modules.py:266-277)");
    Builder B{pl};
    std::vector<int> stack;
    const auto t_begin = std::chrono::steady_clock::now();
    // Common-subexpression sharing across the batch (module_net.py:100-106 evaluates every node of every question; a node
    // whose operands are the encoded clip, keyword strings, identical question spans or other such nodes has the SAME value
    // wherever it occurs -- in another question about the same clip, or twice in one program).  key[i] names the computation
    // of token i ("" = not shareable); the first node with a key is computed, later ones alias its slot.  Filter ignores its
    // tensor keyword (its attention is identically 1, modules.py:354,373), so that operand does not enter the key.
    static const bool cse_env = [] { const char *e = getenv("STAIR_PLAN_CSE"); return !(e && e[0] == '0'); }();
    const bool cse_on = cse_env && !(flags & STAIR_PLAN_NO_CSE);
    // keys are interned: a computation is (tag, operand key ids) -> a small integer id; id 0 = not shareable
    struct CseKey {
        int32_t v[4];
        bool operator==(const CseKey &o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2] && v[3] == o.v[3]; }
    };
    struct CseHash {
        size_t operator()(const CseKey &k) const {
            uint64_t h = 0x9e3779b97f4a7c15ull;
            for (int j = 0; j < 4; ++j) { h ^= (uint32_t)k.v[j] + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); h *= 0xff51afd7ed558ccdull; }
            return (size_t)(h ^ (h >> 32));
        }
    };
    std::vector<int32_t> key(cse_on ? ntok : 0, 0);            // key id of token i
    std::vector<int32_t> first_node(1, -1);                    // key id -> token that computes it (-1: a leaf, nothing to alias)
    // open-addressing table of key ids (4 bytes per slot: the whole table stays in the host's L2) over the keys stored by id;
    // kept per thread and invalidated by a generation stamp, so a build neither allocates nor clears it (the builder runs once
    // per batch on the training loop's critical host path)
    size_t cap = 64;
    while (cse_on && cap < 2 * (size_t)ntok) cap <<= 1;
    static thread_local std::vector<uint64_t> table;            // (generation << 32) | key id
    static thread_local std::vector<CseKey> keys;
    static thread_local uint32_t generation = 0;
    if (cse_on) {
        if (table.size() < cap) table.assign(cap, 0);
        cap = table.size();
        if (++generation == 0) { std::fill(table.begin(), table.end(), 0); generation = 1; }
        keys.clear();
        keys.push_back(CseKey{{0, 0, 0, 0}});
        first_node.reserve((size_t)ntok + 1);
    }
    const uint64_t gen_tag = (uint64_t)generation << 32;
    long cse_probe_steps = 0, cse_probe_inserts = 0;
    auto intern = [&](const CseKey &k, int node, bool &fresh) -> int32_t {
        size_t h = CseHash()(k) & (cap - 1);
        while ((table[h] >> 32) == generation) {
            const int32_t id = (int32_t)(table[h] & 0xffffffffu);
            if (keys[id] == k) { fresh = false; return id; }
            h = (h + 1) & (cap - 1);
            ++cse_probe_steps;
        }
        const int32_t id = (int32_t)first_node.size();
        first_node.push_back(node);
        keys.push_back(k);
        table[h] = gen_tag | (uint32_t)id;
        ++cse_probe_inserts;
        fresh = true;
        return id;
    };
    pl->n_aliased = 0;

    for (int q = 0; q < n; ++q) {
        const int Q = q_off[q + 1] - q_off[q];
        STAIR_CHECK(Q > 0, "empty question " + std::to_string(q));
        pl->max_q = std::max(pl->max_q, Q);
        stack.clear();
        const int Lq = pl->vlen[video_of_question ? video_of_question[q] : q];      // frames of this question's clip
        STAIR_CHECK(prog_off[q + 1] > prog_off[q], "empty program, question " + std::to_string(q));
        for (int i = prog_off[q + 1] - 1; i >= prog_off[q]; --i) {
            const int tok = tokens[i];
            Node &nd = pl->nodes[i];
            if (tok >= 0 && tok < STAIR_OP_COUNT) {
                const int ar = kArity[tok];
                STAIR_CHECK((int)stack.size() >= ar, "invalid program (stack underflow) at " + where(q, i, tok));
                int ch[3] = {-1, -1, -1};
                int lvl = 0;
                for (int k = 0; k < ar; ++k) {
                    ch[k] = stack.back();
                    stack.pop_back();
                    lvl = std::max(lvl, pl->nodes[ch[k]].level);
                }
                nd.level = lvl + 1;
                if (cse_on) {
                    CseKey k = {{tok, -1, -1, -1}};
                    bool ok = true;
                    for (int j = 0; j < ar && ok; ++j) {
                        const bool ignored = tok == STAIR_OP_FILTER && j == 1 && pl->nodes[ch[1]].kind == STAIR_VAL_VEC;
                        if (ignored) { k.v[1 + j] = -2; continue; }
                        if (key[ch[j]] == 0) ok = false;
                        else k.v[1 + j] = key[ch[j]];
                    }
                    if (ok) {
                        bool fresh;
                        const int32_t id = intern(k, i, fresh);
                        key[i] = id;
                        if (!fresh) {                       // computed already: alias it (same value, same level by construction)
                            const Node &o = pl->nodes[first_node[id]];
                            nd.kind = o.kind; nd.slot = o.slot; nd.aux = o.aux; nd.rel = o.rel;
                            nd.bop = o.bop; nd.bvariant = o.bvariant; nd.bsub = o.bsub; nd.inst = o.inst;
                            ++pl->n_aliased;
                            stack.push_back(i);
                            continue;
                        }
                    }
                }
                const Node &c0 = pl->nodes[ch[0]];
                const Node &c1 = ar > 1 ? pl->nodes[ch[1]] : c0;
                const Node &c2 = ar > 2 ? pl->nodes[ch[2]] : c0;
                auto bad = [&](const char *what) {
                    set_error(std::string("stair_plan_build: operand kind mismatch (") + what + ") at " + where(q, i, tok));
                    return 1;
                };
                switch (tok) {
                    case STAIR_OP_AND:
                    case STAIR_OP_XORFRAME: {
                        if (!(c0.kind == c1.kind && (c0.kind == STAIR_VAL_VEC || c0.kind == STAIR_VAL_FRAME)))
                            return bad("needs two [H] vectors or two [T] frame attentions");
                        nd.kind = c0.kind;
                        nd.slot = c0.kind == STAIR_VAL_VEC ? pl->n_vec++ : pl->n_att++;
                        Bucket &b = B.bucket(nd.level, tok, 0, c0.kind);
                        b.col[0].push_back(c0.slot); b.col[1].push_back(c1.slot); b.col[2].push_back(nd.slot);
                        b.cnt++;
                        break;
                    }
                    case STAIR_OP_ATTNVIDEO: {
                        if (c0.kind != STAIR_VAL_MAP || c1.kind != STAIR_VAL_FRAME) return bad("AttnVideo(feat [T,H], attn [T])");
                        nd.kind = STAIR_VAL_MAP; nd.slot = pl->n_map++;
                        Bucket &b = B.bucket(nd.level, tok, 0, 0);
                        b.col[0].push_back(c0.slot); b.col[1].push_back(c1.slot); b.col[2].push_back(nd.slot);
                        b.cnt++;
                        break;
                    }
                    case STAIR_OP_CHOOSE: {
                        if (c0.kind != STAIR_VAL_VEC || c1.kind != STAIR_VAL_VEC || c2.kind != STAIR_VAL_VEC)
                            return bad("Choose(kw1 [H], kw2 [H], query [H])");
                        nd.kind = STAIR_VAL_VEC; nd.slot = pl->n_vec++;
                        Bucket &b = B.bucket(nd.level, tok, 0, 0);
                        b.col[0].push_back(c0.slot); b.col[1].push_back(c1.slot); b.col[2].push_back(c2.slot);
                        b.col[3].push_back(nd.slot);
                        b.cnt++;
                        break;
                    }
                    case STAIR_OP_COMPARE:
                    case STAIR_OP_EQUALS:
                    case STAIR_OP_XOR:
                    case STAIR_OP_TOACTION:
                    case STAIR_OP_EXISTS: {
                        if (c0.kind != STAIR_VAL_VEC || c1.kind != STAIR_VAL_VEC) return bad("needs two [H] vectors");
                        nd.kind = STAIR_VAL_VEC; nd.slot = pl->n_vec++;
                        Bucket &b = B.bucket(nd.level, tok, 0, 0);
                        b.col[0].push_back(c0.slot); b.col[1].push_back(c1.slot); b.col[2].push_back(nd.slot);
                        b.cnt++;
                        break;
                    }
                    case STAIR_OP_EXISTSFRAME: {
                        if (c0.kind != STAIR_VAL_VEC || c1.kind != STAIR_VAL_MAP) return bad("ExistsFrame(keyword [H], feat [T,H])");
                        nd.kind = STAIR_VAL_FRAME; nd.slot = pl->n_att++;
                        Bucket &b = B.bucket(nd.level, tok, 0, 0);
                        b.col[0].push_back(c0.slot); b.col[1].push_back(c1.slot); b.col[2].push_back(nd.slot);
                        b.cnt++;
                        break;
                    }
                    case STAIR_OP_FILTER: {
                        if (c0.kind != STAIR_VAL_MAP) return bad("Filter(feat [T,H], keyword)");
                        int variant;
                        if (c1.kind == STAIR_VAL_VEC) variant = 0;
                        else if (c1.kind == STAIR_VAL_STR && c1.aux == STAIR_KW_ACTIONS) variant = 1;
                        else if (c1.kind == STAIR_VAL_STR && c1.aux == STAIR_KW_OBJECTS) variant = 2;
                        else if (c1.kind == STAIR_VAL_STR && c1.aux == STAIR_KW_RELATIONS) variant = 3;
                        else return bad("Filter keyword must be a [H] vector or actions/objects/relations (modules.py:346-351)");
                        nd.kind = STAIR_VAL_VEC; nd.slot = pl->n_vec++;
                        Bucket &b = B.bucket(nd.level, tok, variant, 0);
                        b.col[0].push_back(c0.slot); b.col[1].push_back(nd.slot); b.col[6].push_back(Lq);
                        b.cnt++;
                        break;
                    }
                    case STAIR_OP_FILTERFRAME: {
                        if (c0.kind != STAIR_VAL_MAP) return bad("FilterFrame(feat [T,H], keyword)");
                        int variant;
                        if (c1.kind == STAIR_VAL_VEC) variant = 0;
                        else if (c1.kind == STAIR_VAL_STR && c1.aux == STAIR_KW_RELATIONS) variant = 1;
                        else if (c1.kind == STAIR_VAL_STR && c1.aux == STAIR_KW_ACTIONS) variant = 2;
                        else return bad("FilterFrame keyword must be a [H] vector or relations/actions (modules.py:384-388)");
                        nd.kind = STAIR_VAL_MAP; nd.slot = pl->n_map++;
                        Bucket &b = B.bucket(nd.level, tok, variant, 0);
                        b.col[0].push_back(c0.slot); b.col[1].push_back(variant == 0 ? c1.slot : 0); b.col[2].push_back(nd.slot);
                        b.cnt++;
                        break;
                    }
                    case STAIR_OP_HASITEM: {
                        if (c0.kind != STAIR_VAL_MAP) return bad("HasItem(feat [T,H])");
                        nd.kind = STAIR_VAL_FRAME; nd.slot = pl->n_att++;
                        Bucket &b = B.bucket(nd.level, tok, 0, 0);
                        b.col[0].push_back(c0.slot); b.col[1].push_back(nd.slot);
                        b.cnt++;
                        break;
                    }
                    case STAIR_OP_LOCALIZE: {
                        if (c0.kind != STAIR_VAL_MAP || !(c1.kind == STAIR_VAL_VEC || c1.kind == STAIR_VAL_PAIR))
                            return bad("Localize(feat [T,H], keyword [H] or [2,H])");
                        const int K = c1.kind == STAIR_VAL_PAIR ? 2 : 1;
                        nd.kind = STAIR_VAL_ATT; nd.slot = pl->n_att; nd.aux = K;
                        pl->n_att += K;
                        Bucket &b = B.bucket(nd.level, tok, 0, 0);
                        b.col[0].push_back(c0.slot);
                        b.col[4].push_back(b.nrows);      // first pair of this instance
                        b.col[5].push_back(K);            // number of pairs
                        for (int k = 0; k < K; ++k) {
                            b.col[1].push_back(b.cnt);                       // pair -> instance
                            b.col[2].push_back(k == 0 ? c1.slot : c1.aux);   // pair -> keyword vec row
                            b.col[3].push_back(nd.slot + k);                 // pair -> att row
                            b.nrows++;
                        }
                        b.cnt++;
                        break;
                    }
                    case STAIR_OP_RELATE: {
                        if (c0.kind != STAIR_VAL_STR || c1.kind != STAIR_VAL_FRAME) return bad("Relate(mode, attn [T])");
                        nd.kind = STAIR_VAL_FRAME; nd.slot = pl->n_att++;
                        Bucket &b = B.bucket(nd.level, tok, c0.aux == STAIR_KW_FORWARD ? 0 : 1, 0);   // modules.py:429
                        b.col[0].push_back(c1.slot); b.col[1].push_back(nd.slot); b.col[6].push_back(Lq);
                        b.cnt++;
                        break;
                    }
                    case STAIR_OP_SUPERLATIVE: {
                        if (c0.kind != STAIR_VAL_STR || c2.kind != STAIR_VAL_MAP ||
                            !(c1.kind == STAIR_VAL_MAP || c1.kind == STAIR_VAL_PAIR || c1.kind == STAIR_VAL_VEC))
                            return bad("Superlative(mode, actions [Ka,H], feat [T,H])");
                        nd.kind = STAIR_VAL_VEC; nd.slot = pl->n_vec++;
                        Bucket &b = B.bucket(nd.level, tok, c0.aux == STAIR_KW_MIN ? 1 : 0, 0);       // modules.py:245
                        b.col[0].push_back(c2.slot);          // feat map
                        b.col[1].push_back(b.nrows);          // first action row of this instance
                        const int Ka = c1.kind == STAIR_VAL_MAP ? Lq : (c1.kind == STAIR_VAL_PAIR ? 2 : 1);   // a [T,H] map = one action per frame of the clip
                        b.col[2].push_back(Ka);
                        b.col[3].push_back(nd.slot); b.col[6].push_back(Lq);
                        for (int a = 0; a < Ka; ++a) {
                            // action row id: >= 0 -> vec arena row; < 0 -> -(map row + 1) (resolved at finalise)
                            int rid;
                            if (c1.kind == STAIR_VAL_MAP) rid = -(c1.slot * T + a) - 1;
                            else rid = a == 0 ? c1.slot : c1.aux;
                            b.col[4].push_back(rid);
                            b.col[5].push_back(b.cnt);        // row -> instance
                            b.nrows++;
                        }
                        b.cnt++;
                        break;
                    }
                    case STAIR_OP_TEMPORAL: {
                        if (c0.kind != STAIR_VAL_STR || c1.kind != STAIR_VAL_MAP || c2.kind != STAIR_VAL_ATT)
                            return bad("Temporal(mode, feat [T,H], attention [K,T])");
                        int mode;
                        if (c0.aux == STAIR_KW_WHILE) mode = 0;
                        else if (c0.aux == STAIR_KW_BEFORE) mode = 1;
                        else if (c0.aux == STAIR_KW_AFTER) mode = 2;
                        else if (c0.aux == STAIR_KW_BETWEEN) mode = 3;
                        else return bad("Temporal mode must be while/before/after/between (modules.py:263,279)");
                        nd.kind = STAIR_VAL_MAP; nd.slot = pl->n_map++; nd.rel = pl->n_att++;
                        Bucket &b = B.bucket(nd.level, tok, mode, 0);
                        b.col[0].push_back(c1.slot); b.col[1].push_back(c2.slot); b.col[2].push_back(c2.aux);
                        b.col[3].push_back(nd.rel); b.col[4].push_back(nd.slot); b.col[6].push_back(Lq);
                        b.cnt++;
                        break;
                    }
                    case STAIR_OP_ARRAY2: {
                        if (c0.kind != STAIR_VAL_VEC || c1.kind != STAIR_VAL_VEC) return bad("Array2(a [H], b [H])");
                        nd.kind = STAIR_VAL_PAIR; nd.slot = c0.slot; nd.aux = c1.slot;   // alias, no launch
                        break;
                    }
                    default:
                        STAIR_FAIL("unknown module token at " + where(q, i, tok));
                }
                if (tok != STAIR_OP_ARRAY2) {         // which bucket computes this node, and as which of its instances
                    int variant = 0, sub = 0;
                    switch (tok) {
                        case STAIR_OP_AND: case STAIR_OP_XORFRAME: sub = c0.kind; break;
                        case STAIR_OP_FILTER:
                            variant = c1.kind == STAIR_VAL_VEC ? 0 : (c1.aux == STAIR_KW_ACTIONS ? 1 : (c1.aux == STAIR_KW_OBJECTS ? 2 : 3)); break;
                        case STAIR_OP_FILTERFRAME: variant = c1.kind == STAIR_VAL_VEC ? 0 : (c1.aux == STAIR_KW_RELATIONS ? 1 : 2); break;
                        case STAIR_OP_RELATE: variant = c0.aux == STAIR_KW_FORWARD ? 0 : 1; break;
                        case STAIR_OP_SUPERLATIVE: variant = c0.aux == STAIR_KW_MIN ? 1 : 0; break;
                        case STAIR_OP_TEMPORAL:
                            variant = c0.aux == STAIR_KW_WHILE ? 0 : (c0.aux == STAIR_KW_BEFORE ? 1 : (c0.aux == STAIR_KW_AFTER ? 2 : 3)); break;
                        default: break;
                    }
                    nd.bop = tok; nd.bvariant = variant; nd.bsub = sub;
                    nd.inst = B.bucket(nd.level, tok, variant, sub).cnt - 1;
                }
            } else if (tok >= STAIR_KW_FORWARD && tok <= STAIR_KW_RELATIONS) {
                if (tok == STAIR_KW_VIDEO) {        // module_net.py:103-104
                    nd.kind = STAIR_VAL_MAP; nd.slot = video_of_question ? video_of_question[q] : q;
                    if (cse_on) { bool fresh; key[i] = intern(CseKey{{1000, nd.slot, -1, -1}}, -1, fresh); }
                } else {
                    nd.kind = STAIR_VAL_STR; nd.aux = tok;
                    if (cse_on) { bool fresh; key[i] = intern(CseKey{{1001, tok, -1, -1}}, -1, fresh); }
                }
            } else if (tok == STAIR_TOK_SPAN) {     // module_net.py:126-129
                const int lo = span_lo[i], hi = std::min(span_hi[i], Q);
                STAIR_CHECK(lo >= 0 && lo < hi, "empty or out-of-range question span at " + where(q, i, tok));
                if (cse_on) {                       // the same words of the same question: one span mean
                    bool fresh;
                    const int32_t id = intern(CseKey{{1002, q, lo, hi}}, i, fresh);
                    key[i] = id;
                    if (!fresh) {
                        const Node &o = pl->nodes[first_node[id]];
                        nd.kind = o.kind; nd.slot = o.slot;
                        ++pl->n_aliased;
                        stack.push_back(i);
                        continue;
                    }
                }
                nd.kind = STAIR_VAL_VEC; nd.slot = pl->n_vec++;
                Bucket &b = B.bucket(0, OP_SPAN, 0, 0);
                b.col[0].push_back(q_off[q] + lo); b.col[1].push_back(hi - lo); b.col[2].push_back(nd.slot);
                b.cnt++;
            } else {
                STAIR_FAIL("unknown token code " + std::to_string(tok) + " at " + where(q, i, tok));
            }
            stack.push_back(i);
        }
        STAIR_CHECK(stack.size() == 1, "invalid program: stack holds " + std::to_string(stack.size()) +
                                           " values at the end (module_net.py:135), question " + std::to_string(q));
        const Node &root = pl->nodes[stack[0]];
        STAIR_CHECK(root.kind == STAIR_VAL_VEC, "program root must produce a [H] vector for the decoder (module_net.py:136), question " + std::to_string(q));
        pl->roots[q] = root.slot;
    }
    pl->rows_q = q_off[n];
    if (getenv("STAIR_PLAN_DEBUG"))
        fprintf(stderr, "cse: %ld inserts, %ld extra probes, cap %zu; scan %.3f ms\n", cse_probe_inserts, cse_probe_steps, cap,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());

    std::sort(pl->buckets.begin(), pl->buckets.end(), [](const Bucket &a, const Bucket &b) {
        return std::tie(a.level, a.op, a.variant, a.sub) < std::tie(b.level, b.op, b.variant, b.sub);
    });

    // ---- sizes & the device index buffer image --------------------------------------------
    const int64_t H = ctx->cfg.hidden_size, A = ctx->cfg.answer_vocab_length;
    pl->maxV = n;
    for (Bucket &b : pl->buckets) {
        pl->n_levels = std::max(pl->n_levels, b.level + 1);
        switch (b.op) {
            case STAIR_OP_FILTER: case STAIR_OP_FILTERFRAME: case STAIR_OP_HASITEM: case STAIR_OP_LOCALIZE:
            case STAIR_OP_SUPERLATIVE: case STAIR_OP_TEMPORAL:
                pl->maxI = std::max(pl->maxI, b.cnt);
                break;
            default: break;
        }
        pl->maxV = std::max(pl->maxV, b.cnt);
        if (b.op == STAIR_OP_LOCALIZE) pl->maxK = std::max(pl->maxK, b.nrows);
        if (b.op == STAIR_OP_SUPERLATIVE) {
            pl->maxK = std::max(pl->maxK, b.nrows);
            pl->maxSupRows = std::max(pl->maxSupRows, b.nrows);
        }
    }
    auto push = [&](const std::vector<int32_t> &v) {
        const int64_t off = (int64_t)pl->idx.size();
        pl->idx.insert(pl->idx.end(), v.begin(), v.end());
        while (pl->idx.size() % 4) pl->idx.push_back(0);
        return off;
    };
    {
        std::vector<int32_t> sv(pl->n_vid + 1), st(q_off, q_off + n + 1);
        for (int v = 0; v <= pl->n_vid; ++v) sv[v] = v * T;
        pl->off_seqv = push(sv);
        pl->off_seqt = push(st);
        pl->off_roots = push(pl->roots);
        pl->off_lenv = push(pl->vlen);
    }
    // layout (float offsets; vec and map arenas start on multiples of H so that any [H] row of
    // either arena has a global row id relative to the workspace base)
    int64_t o = 0;
    auto take = [&](int64_t nfloats, int64_t align) {
        o = align_up(o, align);
        const int64_t r = o;
        o += nfloats;
        return r;
    };
    std::vector<int32_t *> vec_fix;
    if (pl->train) {
        if (int rc = build_grad_fanin(pl, T, vec_fix)) return rc;
    }
    // idx buffer placed first: its size is known only after bucket columns are pushed -> reserve now
    int64_t idx_ints = (int64_t)pl->idx.size();
    for (Bucket &b : pl->buckets)
        for (int c = 0; c < 8; ++c) idx_ints += align_up((int64_t)b.col[c].size(), 4) + align_up((int64_t)b.gcol[c].size(), 4);
    if (pl->train) {
        // token row -> the spans that average over it (their gradient reaches the row by a gather in span order: no atomics)
        pl->tok_ptr.assign(pl->rows_q + 1, 0);
        pl->tok_span.clear();
        for (const Bucket &b : pl->buckets)
            if (b.op == OP_SPAN) {
                for (int i = 0; i < b.cnt; ++i)
                    for (int r = 0; r < b.col[1][i]; ++r) ++pl->tok_ptr[b.col[0][i] + r + 1];
                for (int r = 0; r < pl->rows_q; ++r) pl->tok_ptr[r + 1] += pl->tok_ptr[r];
                pl->tok_span.assign(pl->tok_ptr[pl->rows_q], 0);
                std::vector<int32_t> at(pl->tok_ptr.begin(), pl->tok_ptr.end() - 1);
                for (int i = 0; i < b.cnt; ++i)
                    for (int r = 0; r < b.col[1][i]; ++r) pl->tok_span[at[b.col[0][i] + r]++] = i;
            }
        idx_ints += align_up((int64_t)pl->groots.size(), 4) + align_up((int64_t)pl->fanin.size(), 4) +
                    align_up((int64_t)pl->tok_ptr.size(), 4) + align_up((int64_t)pl->tok_span.size(), 4);
    }
    // per-weight gather columns of the deferred weight-gradient products: the input tile of every instance that uses the weight
    // as a first layer, bucket after bucket (Temporal: also the rows of its per-frame scale)
    std::vector<int32_t> wg_idx[WF_COUNT], wg_rs[WF_COUNT];
    if (pl->train) {
        for (const Bucket &b : pl->buckets) {
            int w0, w3;
            bucket_weights(b, w0, w3);
            if (w0 < 0 || b.cnt == 0) continue;
            wg_idx[w0].insert(wg_idx[w0].end(), b.col[0].begin(), b.col[0].end());
            if (b.op == STAIR_OP_TEMPORAL) wg_rs[w0].insert(wg_rs[w0].end(), b.col[3].begin(), b.col[3].end());
        }
        for (int w = 0; w < WF_COUNT; ++w) idx_ints += align_up((int64_t)wg_idx[w].size(), 4) + align_up((int64_t)wg_rs[w].size(), 4);
    }
    pl->o_idx = take(idx_ints, 64);
    pl->o_vec = take((int64_t)pl->n_vec * H, H);
    pl->o_map = take((int64_t)pl->n_map * T * H, H);
    // resolve Superlative action row ids into global row ids (units of H floats from workspace base)
    if (pl->train && !vec_fix.empty()) {
        // vec staging rows live behind the map staging tiles: as vec-row indices relative to the gradient block's vec base
        const int64_t first = ((pl->o_map - pl->o_vec) + (int64_t)(pl->n_map + pl->n_map_stage) * T * H) / H;
        STAIR_CHECK(first + pl->n_vec_stage < (1ll << 31), "batch too large for 32-bit row ids");
        const int32_t delta = (int32_t)(first - pl->n_vec);
        for (int32_t *v : vec_fix) *v += delta;
        for (size_t e = 0; e + 4 < pl->fanin.size(); e += 5)
            if (pl->fanin[e] == FAN_VEC) pl->fanin[e + 2] += delta;
    }
    for (Bucket &b : pl->buckets)
        if (b.op == STAIR_OP_SUPERLATIVE) {
            for (int32_t &rid : b.col[4]) rid = rid >= 0 ? (int32_t)(pl->o_vec / H + rid) : (int32_t)(pl->o_map / H + (-rid - 1));
            for (int32_t &rid : b.gcol[4]) rid = rid >= 0 ? (int32_t)(pl->o_vec / H + rid) : (int32_t)(pl->o_map / H + (-rid - 1));
        }
    STAIR_CHECK((pl->o_map + (int64_t)(pl->n_map + pl->n_map_stage) * T * H) / H + pl->n_vec_stage < (1ll << 31), "batch too large for 32-bit row ids");
    for (Bucket &b : pl->buckets)
        for (int c = 0; c < 8; ++c) {
            b.off[c] = push(b.col[c]);
            b.goff[c] = b.gcol[c].empty() ? b.off[c] : push(b.gcol[c]);
        }
    if (pl->train) {
        pl->off_groots = push(pl->groots); pl->off_fanin = push(pl->fanin);
        pl->off_tok_ptr = push(pl->tok_ptr); pl->off_tok_span = push(pl->tok_span);
    }
    if (pl->train)
        for (int w = 0; w < WF_COUNT; ++w) { pl->wg_off_idx[w] = push(wg_idx[w]); pl->wg_off_rs[w] = push(wg_rs[w]); }
    STAIR_CHECK((int64_t)pl->idx.size() == idx_ints, "internal: idx size");
    pl->o_att = take((int64_t)std::max(pl->n_att, 1) * T, 64);
    pl->o_tok = take((int64_t)pl->rows_q * H, 64);
    pl->o_qfeat = take((int64_t)n * H, 64);
    pl->o_vhn = take((int64_t)pl->n_vid * H, 64);
    {   // input projections of the two encoders: xproj (the gates of a training plan), bias sums, W_ih planes of the video encoder (bf16
        // features: 2 x [4H, V] bf16), and the text encoder's projection as a plane GEMM (W_ih and the token rows as zero-padded hi / lo
        // planes, counted in floats) -- one block with the layout of proj_layout, here or in the caller's buffer (ext_proj)
        const ProjLayout L = proj_layout(ctx->cfg, pl->n_vid, T, pl->rows_q);
        const int64_t base = pl->ext_proj ? 0 : take(L.total, 64);
        pl->o_xpv = base + L.xpv; pl->o_xpt = base + L.xpt; pl->o_bias = base + L.bias;
        pl->o_wplanes = ctx->cfg.video_size % 32 == 0 ? base + L.wplanes : 0;
        pl->o_wplanes_t = base + L.wplanes_t; pl->o_xplanes_t = base + L.xplanes_t;
    }
    pl->o_wpack = take(2 * 2 * H * H, 64);       // 8*Hh*Hh floats per encoder
    pl->coop_bytes = std::max(lstm_coop_ws_bytes(pl->n_vid), lstm_coop_ws_bytes(n));
    if (pl->train) pl->coop_bytes = std::max(pl->coop_bytes, std::max(lstm_coop_bwd_ws_bytes(pl->n_vid), lstm_coop_bwd_ws_bytes(n)));
    pl->o_coop = take((pl->coop_bytes + 3) / 4, 64);   // exchange slabs + flags of the cooperative recurrence (video encoder)
    pl->o_coop2 = take((pl->coop_bytes + 3) / 4, 64);  // the text encoder's: the two recurrences share a launch when both fit on the chip
    pl->o_splitk = take(kSplitKFloats, 64);      // partial sums of split-K launches (<= 64 output tiles x 16 pieces)
    pl->o_tmpA = take((int64_t)std::max(pl->maxI, 1) * T * H, 64);
    pl->o_tmpB = take((int64_t)std::max(pl->maxI, 1) * T * H, 64);
    pl->o_kbuf = take((int64_t)std::max(pl->maxK, 1) * H, 64);
    pl->o_cat = take((int64_t)pl->maxV * 3 * H, 64);
    pl->o_hid = take((int64_t)pl->maxV * 2 * H, 64);
    pl->o_rs = take((int64_t)std::max(pl->maxI, 1) * T, 64);
    pl->o_sup = take((int64_t)std::max(pl->maxSupRows, 1) * T, 64);
    pl->o_extra = take(std::max(pl->maxI, 1), 64);
    pl->o_logits = take((int64_t)n * A, 64);
    pl->o_wfrag = (H == 512 && T <= 64) ? take((int64_t)WV_END * H * H, 64) : 0;     // bf16 hi/lo fragment-order planes of the fused tile operators' weights
    pl->o_status = take(128, 64);                // word 0: sticky "a cooperative hand-off timed out" flag of this plan's passes; words 16, 17
                                                 // and 18, 19: the self-resetting work queues of the fused forward / backward tile launches
    for (Bucket &b : pl->buckets) {
        b.svA = pl->o_tmpA; b.svB = pl->o_tmpB; b.svK = pl->o_kbuf; b.svCat = pl->o_cat; b.svHid = pl->o_hid;
        b.svRs = pl->o_rs; b.svSup = pl->o_sup; b.svExtra = pl->o_extra;
        if (!pl->train) {
            // inference: the tile operators of one level run in ONE launch (fused path), so what a bucket's tile operator
            // writes for its own later launches must not be shared with the level's other buckets
            if (pl->o_wfrag > 0 && b.cnt > 0) {
                if (b.op == STAIR_OP_FILTER) b.svCat = take((int64_t)b.cnt * H, 64);
                if (b.op == STAIR_OP_SUPERLATIVE) b.svB = take((int64_t)b.cnt * T * H, 64);
                // the grouped vector-level launches: a two-layer module's hidden rows live from the level's first launch to its second
                if (b.op == STAIR_OP_EXISTS || b.op == STAIR_OP_TOACTION) b.svHid = take((int64_t)b.cnt * H, 64);
            }
            continue;
        }
        const int64_t c = b.cnt;
        switch (b.op) {
            // (svA of Filter / FilterFrame / Localize / Superlative lives in the per-weight region wg_sx, assigned below)
            case STAIR_OP_FILTER:            // (svCat, the pooled rows = the dense layer's input, lives in the per-weight region vd_x)
                b.svB = take(c * T * H, 64); break;
            case STAIR_OP_FILTERFRAME:
                b.svB = take(c * T * H, 64); b.svRs = take(c * T, 64); b.svExtra = take(c, 64); break;
            case STAIR_OP_HASITEM:
            case STAIR_OP_TEMPORAL:
                b.svA = take(c * T * H, 64); break;
            case STAIR_OP_LOCALIZE:
                b.svB = take(c * T * H, 64); b.svK = take((int64_t)b.nrows * H, 64); break;
            case STAIR_OP_SUPERLATIVE:
                b.svB = take(c * T * H, 64); b.svK = take((int64_t)b.nrows * H, 64);
                b.svSup = take((int64_t)b.nrows * T, 64); b.svCat = take(c * H, 64); break;
            default: break;                  // (the vector-level modules' packed inputs and hidden rows: per-weight regions vd_x)
        }
    }
    if (pl->train) {
        // per-weight regions: dZ blocks (and, for second-layer weights, the saved first activations) of all buckets of a weight
        // lie back to back in bucket order
        int64_t rows0[WF_COUNT] = {}, rows3[WF_COUNT] = {};
        for (const Bucket &b : pl->buckets) {
            int w0, w3;
            bucket_weights(b, w0, w3);
            if (w0 >= 0) rows0[w0] += b.cnt;
            if (w3 >= 0) rows3[w3] += b.cnt;
        }
        for (int w = 0; w < WF_COUNT; ++w) {
            STAIR_CHECK(!(rows0[w] && rows3[w]), "internal: a weight is either a first or a second layer");
            pl->wg_rows[w] = rows0[w] + rows3[w];
            if (pl->wg_rows[w]) pl->wg_dz[w] = take(pl->wg_rows[w] * T * H, 64);
            if (rows3[w]) pl->wg_sx[w] = take(rows3[w] * T * H, 64);
            // slab partials of the weight's one long weight-gradient reduction (csrc/gemm_tn_x3tr.hip: stored, then added in fixed order)
            if (pl->wg_rows[w] && H % 256 == 0 && T % 32 == 0)
                pl->wg_part[w] = take(tn_x3tr_scratch_floats(pl->wg_rows[w] * T, H, H), 64);
        }
        if (H % 256 == 0) {
            // scratch of the other slab-reduced products: a ring for the map-level layers outside the per-weight regions (two of the
            // largest in flight), and one piece per encoder weight (dW_ih on fp32 rows, dW_hh; two directions)
            const int64_t most_rows = std::max<int64_t>((int64_t)std::max(pl->maxI, 1) * T, 64);
            pl->tnring_floats = 2 * align_up(tn_x3tr_scratch_floats(most_rows & ~31ll, H, 3 * H), 64);
            pl->o_tnring = take(pl->tnring_floats, 64);
            const int64_t rv = ((int64_t)pl->n_vid * T) & ~31ll, rq = (int64_t)pl->rows_q & ~31ll;
            const int E4 = ctx->cfg.text_size, V4 = ctx->cfg.video_size;
            pl->tnenc_floats[0] = 2 * (align_up(tn_x3tr_scratch_floats(std::max<int64_t>(rv, 64), 2 * H, H / 2), 64) +
                                       align_up(tn_x3tr_scratch_floats(std::max<int64_t>(rv, 64), 2 * H, V4), 64));
            pl->tnenc_floats[1] = 2 * (align_up(tn_x3tr_scratch_floats(std::max<int64_t>(rq, 64), 2 * H, H / 2), 64) +
                                       align_up(tn_x3tr_scratch_floats(std::max<int64_t>(rq, 64), 2 * H, E4), 64));
            pl->o_tnenc[0] = take(pl->tnenc_floats[0], 64);
            pl->o_tnenc[1] = take(pl->tnenc_floats[1], 64);
        }
        {   // vector-level weights: per-weight dZ and input regions, the buckets' rows back to back in bucket order
            for (const Bucket &b : pl->buckets) {
                int v0, v3;
                bucket_vec_weights(b.op, v0, v3);
                if (v0 >= 0) pl->vd_rows[v0] += b.cnt;
                if (v3 >= 0) pl->vd_rows[v3] += b.cnt;
            }
            for (int w = 0; w < VD_COUNT; ++w)
                if (pl->vd_rows[w]) { pl->vd_dz[w] = take(pl->vd_rows[w] * H, 64); pl->vd_x[w] = take(pl->vd_rows[w] * vd_cols(w, H), 64); }
            int64_t vat[VD_COUNT] = {};
            for (Bucket &b : pl->buckets) {
                int v0, v3;
                bucket_vec_weights(b.op, v0, v3);
                if (v0 >= 0) { b.dzV0 = pl->vd_dz[v0] + vat[v0] * H; b.svCat = pl->vd_x[v0] + vat[v0] * vd_cols(v0, H); vat[v0] += b.cnt; }
                if (v3 >= 0) { b.dzV3 = pl->vd_dz[v3] + vat[v3] * H; b.svHid = pl->vd_x[v3] + vat[v3] * H; vat[v3] += b.cnt; }
            }
        }
        int64_t at[WF_COUNT] = {};
        for (Bucket &b : pl->buckets) {
            int w0, w3;
            bucket_weights(b, w0, w3);
            if (pl->o_wfrag > 0 && b.cnt > 0) {          // the level's backward chains share a launch: no scratch in common
                if (b.op == STAIR_OP_FILTER) b.gRow = take((int64_t)b.cnt * H, 64);
                if (b.op == STAIR_OP_FILTERFRAME && b.variant != 0) b.dzC = take((int64_t)b.cnt * T * H, 64);
                const int64_t words = (int64_t)b.cnt * 512 * 2;          // 64-bit words, counted in floats
                switch (b.op) {
                    case STAIR_OP_FILTER: b.bitA = take(words, 64); b.bitB = take(words, 64); break;
                    case STAIR_OP_FILTERFRAME: b.bitA = take(words, 64); b.bitB = take(words, 64); b.bitC = take(words, 64); break;
                    case STAIR_OP_HASITEM: case STAIR_OP_LOCALIZE: case STAIR_OP_SUPERLATIVE: b.bitA = take(words, 64); break;
                    default: break;
                }
            }
            if (w0 >= 0) { b.dzA = pl->wg_dz[w0] + at[w0] * T * H; at[w0] += b.cnt; }
            if (w3 >= 0) { b.dzB = pl->wg_dz[w3] + at[w3] * T * H; b.svA = pl->wg_sx[w3] + at[w3] * T * H; at[w3] += b.cnt; }
        }
        const int64_t I = std::max(pl->maxI, 1), Vv = pl->maxV;
        pl->o_cv = take((int64_t)pl->n_vid * T * H, 64);
        pl->o_ct = take((int64_t)pl->rows_q * H, 64);
        pl->o_hprev = take(((int64_t)pl->n_vid * T + pl->rows_q) * H, 64);        // video rows, then text rows: the two backward passes may run side by side
        pl->o_wt = take(ctx_weight_floats(ctx), 64);
        pl->o_gA = take(I * T * H, 64);
        pl->o_gB = take(I * T * H, 64);
        if (pl->o_wfrag > 0) pl->o_wfragT = take((int64_t)WV_END * H * H, 64);     // backward chains of the fused tile operators and the grouped
                                                                                   // vector-level launches: planes of the transposed weights
        pl->o_gV0 = take(Vv * 2 * H, 64);
        pl->o_gV1 = take(Vv * 2 * H, 64);
        pl->o_gCat = take(Vv * 3 * H, 64);
        pl->o_gStats = take(I * T * 2, 64);
        pl->o_gRs2 = take(I * T, 64);
        pl->o_dlogits = take((int64_t)n * A, 64);
        pl->o_loss = take(n, 64);
        // everything from here to o_zero_end is cleared at the start of every backward pass
        pl->o_zero_beg = align_up(o, 64);
        o = pl->o_zero_beg;
        // mirrors [vec arena .. map arena]; behind it the staging tiles and rows of the deterministic fan-in (build_grad_fanin)
        pl->o_gblock = take(pl->o_map + (int64_t)(pl->n_map + pl->n_map_stage) * T * H - pl->o_vec + (int64_t)pl->n_vec_stage * H, H);
        pl->o_gatt = take((int64_t)std::max(pl->n_att + pl->n_att_stage, 1) * T, 64);
        pl->o_gtok = take((int64_t)pl->rows_q * H, 64);
        pl->o_gqfeat = take((int64_t)n * H, 64);
        pl->o_zero_end = align_up(o, 64);
        o = pl->o_zero_end;
        pl->o_gshadow = 1;            // (training plans use the context's fixed-point gradient shadows, stair_ctx::gshadow)
        // scratch that individual buckets clear themselves before accumulating into it
        pl->o_gK = take((int64_t)std::max(pl->maxK, 1) * H, 64);
        pl->o_gS = take((int64_t)std::max(pl->maxSupRows, 1) * T, 64);
        pl->o_gRs = take(I * T, 64);
        pl->o_gExtra = take(I, 64);
    }
    pl->total = align_up(o, 64);
    if (getenv("STAIR_PLAN_DEBUG") && pl->train) {
        fprintf(stderr, "weight-gradient regions (instances of T rows):");
        for (int w = 0; w < WF_COUNT; ++w) fprintf(stderr, " %ld", (long)pl->wg_rows[w]);
        fprintf(stderr, "\n");
    }
    if (getenv("STAIR_PLAN_DEBUG"))
        fprintf(stderr, "plan build total %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    *out = plp.release();
    return 0;
}

extern "C" void stair_plan_destroy(stair_plan *plan) { delete plan; }

extern "C" int stair_plan_get_info(const stair_plan *pl, stair_plan_info *info) {
    STAIR_CHECK(pl && info, "null argument");
    info->workspace_bytes = pl->total * (int64_t)sizeof(float);
    info->vec_off = pl->o_vec; info->map_off = pl->o_map; info->att_off = pl->o_att;
    info->tok_off = pl->o_tok; info->qfeat_off = pl->o_qfeat; info->logits_off = pl->o_logits;
    info->gvec_off = pl->train ? pl->o_gblock : -1;
    info->gmap_off = pl->train ? pl->o_gblock + (pl->o_map - pl->o_vec) : -1;
    info->gatt_off = pl->train ? pl->o_gatt : -1;
    info->status_off = pl->o_status;
    info->n_vec = pl->n_vec; info->n_map = pl->n_map; info->n_att = pl->n_att; info->n_tok_rows = pl->rows_q;
    info->n_nodes = (int)pl->nodes.size();
    int launches = 0;
    for (const Bucket &b : pl->buckets) launches += b.cnt > 0;
    info->n_launches = launches; info->n_levels = pl->n_levels; info->n_questions = pl->n; info->T = pl->T;
    info->n_aliased = pl->n_aliased;
    info->n_vec_stage = pl->n_vec_stage; info->n_map_stage = pl->n_map_stage; info->n_att_stage = pl->n_att_stage;
    return 0;
}

extern "C" int stair_plan_status(const stair_plan *pl, const void *workspace, stair_stream stream) {
    STAIR_CHECK(pl && workspace, "null argument");
    uint32_t word = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    STAIR_HIP(hipMemcpyAsync(&word, static_cast<const float *>(workspace) + pl->o_status, sizeof(word), hipMemcpyDeviceToHost, s));
    STAIR_HIP(hipStreamSynchronize(s));
    STAIR_CHECK(word == 0, "a cooperative LSTM hand-off timed out (its workgroups were not co-resident: another queue on this GPU, or a "
                           "partitioned device); the results of this run contain NaN.  Set STAIR_LSTM_COOP=0 to use the one-workgroup kernels");
    return 0;
}

// Where a training plan keeps the activations its backward pass reads -- for tests that compare the ReLU masks of two
// implementations.  which = 0 / 1: first / second saved activation of the node's tile MLP ([T, H]: Filter, FilterFrame,
// Localize, Superlative; first only: HasItem, Temporal) or the hidden row [H] of Exists / ToAction (which = 0);
// tok = -1 - q: the decoder's hidden row [2H] of question q.  *off = float offset into the workspace, -1 = no such buffer.
extern "C" int stair_plan_saved_offset(const stair_plan *pl, int32_t tok, int32_t which, int64_t *off) {
    STAIR_CHECK(pl && off, "null argument");
    STAIR_CHECK(pl->train, "plan was not built with STAIR_PLAN_TRAIN");
    *off = -1;
    const int64_t H = pl->cfg.hidden_size, T = pl->T;
    if (tok < 0) {
        const int q = -1 - tok;
        STAIR_CHECK(q < pl->n, "question index out of range");
        if (which == 0) *off = pl->o_hid + (int64_t)q * 2 * H;
        return 0;
    }
    STAIR_CHECK(tok < (int)pl->nodes.size(), "token index out of range");
    const Node &nd = pl->nodes[tok];
    if (nd.inst < 0) return 0;
    for (const Bucket &b : pl->buckets) {
        if (b.level != nd.level || b.op != nd.bop || b.variant != nd.bvariant || b.sub != nd.bsub) continue;
        switch (b.op) {
            case STAIR_OP_FILTER: case STAIR_OP_FILTERFRAME: case STAIR_OP_LOCALIZE: case STAIR_OP_SUPERLATIVE:
                if (which == 0) *off = b.svA + nd.inst * T * H;
                else if (which == 1) *off = b.svB + nd.inst * T * H;
                break;
            case STAIR_OP_HASITEM: case STAIR_OP_TEMPORAL:
                if (which == 0) *off = b.svA + nd.inst * T * H;
                break;
            case STAIR_OP_EXISTS: case STAIR_OP_TOACTION:
                if (which == 0) *off = b.svHid + nd.inst * H;
                break;
            default: break;
        }
        return 0;
    }
    return 0;
}

extern "C" int stair_plan_node(const stair_plan *pl, int32_t tok, int32_t *kind, int32_t *slot, int32_t *aux,
                               int32_t *level, int32_t *rel_slot) {
    STAIR_CHECK(pl, "null plan");
    STAIR_CHECK(tok >= 0 && tok < (int)pl->nodes.size(), "token index out of range");
    const Node &nd = pl->nodes[tok];
    if (kind) *kind = nd.kind;
    if (slot) *slot = nd.slot;
    if (aux) *aux = nd.aux;
    if (level) *level = nd.level;
    if (rel_slot) *rel_slot = nd.rel;
    return 0;
}

// the whole node table in one call (host arrays of n_nodes int32 each; any of them may be NULL)
extern "C" int stair_plan_nodes(const stair_plan *pl, int32_t *kind, int32_t *slot, int32_t *aux, int32_t *level, int32_t *rel_slot,
                                int32_t count) {
    STAIR_CHECK(pl, "null plan");
    STAIR_CHECK(count == (int32_t)pl->nodes.size(), "count must equal stair_plan_info.n_nodes");
    for (int32_t i = 0; i < count; ++i) {
        const Node &nd = pl->nodes[i];
        if (kind) kind[i] = nd.kind;
        if (slot) slot[i] = nd.slot;
        if (aux) aux[i] = nd.aux;
        if (level) level[i] = nd.level;
        if (rel_slot) rel_slot[i] = nd.rel;
    }
    return 0;
}

// =============================================================================================
// runner
// =============================================================================================
namespace {

struct Lin { const float *w = nullptr, *b = nullptr; float *dw = nullptr, *db = nullptr; int id = -1; };
struct Weights {
    Lin compare, equals, exists0, exists3, f0[4], f3[4], fdense, ff0[3], ff3[3], ffatt, ffdense, hi0, hi3, lv0, lv3,
        lk, supdense, tdense, ta0, ta3, xorl, dec0, dec3;
    const float *beta = nullptr, *ln_w = nullptr, *ln_b = nullptr;
    float *dbeta = nullptr, *dln_w = nullptr, *dln_b = nullptr;
    const float *relate[3][6] = {};
    float *drelate[3][6] = {};
    const float *enc[2][8] = {};   // [video|text][w_ih, w_hh, b_ih, b_hh, then reverse]
    float *denc[2][8] = {};
};

int resolve(const stair_ctx *ctx, Weights &W, bool grads) {
    const std::string p = "submodules.";
    auto get = [&](const std::string &name, const float *&dst, float **gdst) {
        dst = ctx->find(name);
        if (!dst) {
            set_error("weight not set: " + name);
            return 1;
        }
        if (grads) {
            *gdst = ctx->findg(name);
            if (!*gdst) {
                set_error("gradient buffer not set: " + name);
                return 1;
            }
        }
        return 0;
    };
    auto lin = [&](const std::string &prefix, Lin &l) {
        l.id = ctx->by_name.at(prefix + ".weight");
        return get(prefix + ".weight", l.w, &l.dw) || get(prefix + ".bias", l.b, &l.db);
    };
#define R(x) if (x) return 1
    R(lin(p + "Compare.param.0", W.compare));
    R(lin(p + "Equals.param.0", W.equals));
    R(lin(p + "Exists.param.0", W.exists0));
    R(lin(p + "Exists.param.3", W.exists3));
    const char *fk[4] = {"representation", "actions", "objects", "relations"};
    for (int i = 0; i < 4; ++i) {
        R(lin(p + "Filter.param." + fk[i] + ".0", W.f0[i]));
        R(lin(p + "Filter.param." + fk[i] + ".3", W.f3[i]));
    }
    R(lin(p + "Filter.dense.0", W.fdense));
    const char *ffk[3] = {"representation", "relations", "actions"};
    for (int i = 0; i < 3; ++i) {
        R(lin(p + "FilterFrame.param." + ffk[i] + ".0", W.ff0[i]));
        R(lin(p + "FilterFrame.param." + ffk[i] + ".3", W.ff3[i]));
    }
    R(lin(p + "FilterFrame.attention.0", W.ffatt));
    R(lin(p + "FilterFrame.dense.0", W.ffdense));
    R(lin(p + "HasItem.param.0", W.hi0));
    R(lin(p + "HasItem.param.3", W.hi3));
    R(lin(p + "Localize.video_linear.0", W.lv0));
    R(lin(p + "Localize.video_linear.3", W.lv3));
    R(lin(p + "Localize.keyword_linear.0", W.lk));
    R(get(p + "Relate.beta", W.beta, &W.dbeta));
    R(lin(p + "Superlative.dense.0", W.supdense));
    const char *modes[3] = {"before", "after", "between"};
    for (int m = 0; m < 3; ++m)
        for (int l = 0; l < 3; ++l) {
            const std::string pre = p + "Temporal.relate." + modes[m] + "." + std::to_string(2 * l);
            R(get(pre + ".weight", W.relate[m][2 * l], &W.drelate[m][2 * l]));
            R(get(pre + ".bias", W.relate[m][2 * l + 1], &W.drelate[m][2 * l + 1]));
        }
    R(lin(p + "Temporal.dense.0", W.tdense));
    R(get(p + "Temporal.layer_norm.weight", W.ln_w, &W.dln_w));
    R(get(p + "Temporal.layer_norm.bias", W.ln_b, &W.dln_b));
    R(lin(p + "ToAction.param.0", W.ta0));
    R(lin(p + "ToAction.param.3", W.ta3));
    R(lin(p + "Xor.param.0", W.xorl));
    for (int e = 0; e < 2; ++e) {
        const std::string enc = p + (e == 0 ? "video_encoder" : "text_encoder");
        const char *sfx[2] = {"", "_reverse"};
        for (int d = 0; d < 2; ++d) {
            R(get(enc + ".weight_ih_l0" + sfx[d], W.enc[e][4 * d + 0], &W.denc[e][4 * d + 0]));
            R(get(enc + ".weight_hh_l0" + sfx[d], W.enc[e][4 * d + 1], &W.denc[e][4 * d + 1]));
            R(get(enc + ".bias_ih_l0" + sfx[d], W.enc[e][4 * d + 2], &W.denc[e][4 * d + 2]));
            R(get(enc + ".bias_hh_l0" + sfx[d], W.enc[e][4 * d + 3], &W.denc[e][4 * d + 3]));
        }
    }
    R(lin(p + "decoder.0", W.dec0));
    R(lin(p + "decoder.3", W.dec3));
#undef R
    return 0;
}

// Where the fused tile operators pay (measured on one box per pair, same build, STAIR_TILE_MLP / STAIR_TILE_TRAIN = 0 against 1;
// profiles/r03_*):
//   inference       32 / 128 / 512 / 2048 questions per batch: 1.36 / 1.49 / 2.19 / 5.6-6.1 ms fused, 1.58 / 1.91 / 2.59 / 6.1 ms sequenced
//   training step   32 / 128 / 512 / 1024 / 2048: 4.47 / 5.34 / 8.01 / 12.08 / 18.50 ms fused, 4.60 / 5.71 / 8.33 / 12.18 / 18.85 ms sequenced
// (the training figures after two fixes that were worth 2.2 ms at 2048 questions: the backward chain's accumulation as
// 256-byte wave-instructions instead of 16-byte-strided lanes -- float atomics run at full rate only in that shape -- and
// every mask load / activation save between layers row-wise through the fp32 staging instead of 32-byte pieces from the
// accumulator layout).  Fused is the default for both kinds of plan; STAIR_TILE_TRAIN=0 / STAIR_TILE_TRAIN_BWD=0 put a
// training plan's forward / backward back on the GEMM sequences.
bool tile_policy(const stair_plan *pl, bool backward = false) {
    if (!pl->train) return true;
    static const int force = [] { const char *e = getenv("STAIR_TILE_TRAIN"); return e ? atoi(e) : -1; }();
    static const int force_bwd = [] { const char *e = getenv("STAIR_TILE_TRAIN_BWD"); return e ? atoi(e) : -1; }();
    if (backward && force_bwd >= 0) return force_bwd != 0;
    if (force >= 0) return force != 0;
    return true;
}

// dense helper: C[g][r] = act(rs * A[g][r] W^T + b)
thread_local float *g_splitk_ws = nullptr;      // set by stair_plan_run for the duration of the call, on the calling thread

int dense(hipStream_t s, const float *A, int64_t lda, int64_t a_gs, const int32_t *a_gidx, const Lin &l, int64_t ldw,
          float *C, int64_t ldc, int64_t c_gs, const int32_t *c_gidx, int groups, int R, int N, int K, int act,
          const float *rs = nullptr, int64_t rs_gs = 0, const int32_t *rs_gidx = nullptr) {
    stair_gemm_args g = {};
    g.splitk_ws = g_splitk_ws; g.splitk_ws_floats = g_splitk_ws ? kSplitKFloats : 0;
    g.A = A; g.lda = lda; g.a_gstride = a_gs; g.a_gidx = a_gidx;
    g.W = l.w; g.ldw = ldw; g.bias = l.b;
    g.C = C; g.ldc = ldc; g.c_gstride = c_gs; g.c_gidx = c_gidx;
    g.row_scale = rs; g.rs_gstride = rs_gs; g.rs_gidx = rs_gidx;
    g.groups = groups; g.rows_per_group = R; g.N = N; g.K = K; g.act = act;
    return launch_gemm(g, s);
}

// a forward-shaped problem of the grouped vector-level launch (csrc/vec_group.hip): out[io[i]] = act(in(a_i, b_i) W^T + bias)
VgProblem vg_fwd(int rows, const float *a, const int32_t *ia, int64_t lda, const float *b, const int32_t *ib, int64_t ldb, int pack,
                 const float *Wm, int64_t ldw, const float *bias, int N, int act, float *out, const int32_t *io, int64_t ldo) {
    VgProblem p = {};
    p.kind = VG_FWD; p.rows = rows; p.a = a; p.ia = ia; p.lda = lda; p.b = b; p.ib = ib; p.ldb = ldb; p.pack = pack; p.in_scale = 1.0f;
    p.kred = 512; p.W = Wm; p.ldw = ldw; p.bias = bias; p.N = N; p.act = act; p.out = out; p.io = io; p.ldo = ldo;
    return p;
}
// STAIR_VEC_GROUP=0 keeps the per-module pack -> GEMM -> reduction sequences (and the tile form for large buckets)
static bool tile_dropout_on() {
    static const bool on = [] { const char *e = getenv("STAIR_TILE_DROPOUT"); return !(e && e[0] == '0'); }();
    return on;
}
bool vec_group_on() {
    static const bool on = [] { const char *e = getenv("STAIR_VEC_GROUP"); return !(e && e[0] == '0'); }();
    return policy_or(STAIR_OPT_VEC_GROUP, on ? 1 : 0) != 0;
}

struct Ptrs {     // workspace views shared by forward and backward
    float *ws, *vec, *map, *att, *tok, *qfeat;
    int32_t *didx;
};

}  // namespace

extern "C" int stair_plan_set_dropout(stair_plan *pl, float p, uint64_t seed) {
    STAIR_CHECK(pl, "null plan");
    STAIR_CHECK(p >= 0.0f && p < 1.0f, "dropout probability must be in [0, 1)");
    STAIR_CHECK(p == 0.0f || pl->train, "dropout needs a STAIR_PLAN_TRAIN plan (model.eval() has none, modules.py)");
    // module_net.py:100-106 evaluates every node of every question, so under model.train() every occurrence draws its OWN mask;
    // a node shared by several questions would be dropped once for all of them
    STAIR_CHECK(p == 0.0f || pl->n_aliased == 0, "dropout needs a plan without shared subexpressions: build it with STAIR_PLAN_NO_CSE");
    pl->drop_p = p;
    pl->drop_seed = seed;
    return 0;
}

extern "C" int stair_plan_set_backward_event(stair_plan *pl, void *event) {
    STAIR_CHECK(pl, "null plan");
    STAIR_CHECK(!event || pl->train, "the backward event belongs to a STAIR_PLAN_TRAIN plan");
    pl->bwd_event = event;
    return 0;
}

static int upload_index_image(stair_plan *pl, int32_t *didx, hipStream_t s) {
    if (!pl->pin.p) {
        pinned_reclaim();
        pl->pin = pinned_take(pl->idx.size());
        STAIR_CHECK(pl->pin.p != nullptr, "hipHostMalloc of the index staging buffer failed");
        memcpy(pl->pin.p, pl->idx.data(), pl->idx.size() * sizeof(int32_t));
        STAIR_HIP(hipEventCreateWithFlags(&pl->pin_ev, hipEventDisableTiming));
    }
    STAIR_HIP(hipMemcpyAsync(didx, pl->pin.p, pl->idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    STAIR_HIP(hipEventRecord(pl->pin_ev, s));
    return 0;
}

extern "C" int stair_plan_upload(stair_plan *pl, void *workspace, int64_t workspace_bytes, stair_stream stream) {
    STAIR_CHECK(pl && workspace, "null argument");
    STAIR_CHECK(workspace_bytes >= pl->total * (int64_t)sizeof(float), "workspace too small");
    STAIR_CHECK((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");
    int32_t *didx = reinterpret_cast<int32_t *>(static_cast<float *>(workspace) + pl->o_idx);
    return upload_index_image(pl, didx, static_cast<hipStream_t>(stream));
}

extern "C" int64_t stair_projection_floats(const stair_ctx *ctx, int32_t n_videos, int32_t T, int64_t question_rows) {
    if (!ctx || n_videos < 0 || T < 0 || question_rows < 0) return -1;
    return proj_layout(ctx->cfg, n_videos, T, question_rows).total;
}

// The input projections of both encoders (x W_ih^T + b_ih + b_hh for every clip frame and every token row) into a caller-owned buffer,
// from the batch's inputs alone -- no plan needed, so the ~2 ms of GPU work they are at 2048 questions can be enqueued before the host
// packs the programs and builds the plan (3-4 ms during which the GPU would otherwise idle whenever it is not already a step behind).
extern "C" int stair_encoders_project(stair_ctx *ctx, const void *video, int32_t video_is_bf16, int32_t n_videos, int32_t T,
                                      const float *question, int64_t question_rows, float *buf, int64_t buf_floats,
                                      stair_stream stream) {
    STAIR_CHECK(ctx && video && question && buf, "null argument");
    STAIR_CHECK(n_videos >= 0 && T >= 1 && question_rows >= 0, "bad shape");
    STAIR_CHECK((reinterpret_cast<uintptr_t>(buf) & 255) == 0, "projection buffer must be 256-byte aligned");
    PolicyScope policy_scope(&ctx->policy);
    const stair_config &g = ctx->cfg;
    const ProjLayout L = proj_layout(g, n_videos, T, question_rows);
    STAIR_CHECK(buf_floats >= L.total, "projection buffer too small (stair_projection_floats)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Weights W;
    if (resolve(ctx, W, false)) return 1;
    const int H = g.hidden_size, Hh = H / 2, V = g.video_size, E = g.text_size;
    stair_lstm_args a = {}, t = {};
    a.x = static_cast<const float *>(video); a.ldx = V; a.rows = n_videos * T; a.n = n_videos; a.max_len = T; a.I = V; a.Hh = Hh;
    if (video_is_bf16) {
        STAIR_CHECK(V % 32 == 0, "bf16 clip features need video_size % 32 == 0");
        a.x = nullptr; a.x_bf16 = video; a.wih_planes_ws = buf + L.wplanes;
    }
    for (int d = 0; d < 2; ++d) {
        a.w_ih[d] = W.enc[0][4 * d]; a.b_ih[d] = W.enc[0][4 * d + 2]; a.b_hh[d] = W.enc[0][4 * d + 3];
        t.w_ih[d] = W.enc[1][4 * d]; t.b_ih[d] = W.enc[1][4 * d + 2]; t.b_hh[d] = W.enc[1][4 * d + 3];
    }
    a.xproj_ws = buf + L.xpv; a.bias_ws = buf + L.bias;
    t.x = question; t.ldx = E; t.rows = (int32_t)question_rows; t.n = 1; t.max_len = (int32_t)question_rows; t.I = E; t.Hh = Hh;
    static const bool text_planes = [] { const char *e = getenv("STAIR_TEXT_PLANES"); return !(e && e[0] == '0'); }();
    if (text_planes) { t.wih_planes_ws = buf + L.wplanes_t; t.x_planes_ws = buf + L.xplanes_t; }
    t.xproj_ws = buf + L.xpt; t.bias_ws = buf + L.bias + 4 * H;
    if (int rc = launch_lstm_project(a, s)) return rc;
    return launch_lstm_project(t, s);
}

// The buffer of a STAIR_PLAN_EXT_PROJECTION plan (>= stair_projection_floats of the plan's batch shape); it must stay alive and untouched
// until the last pass of the plan that reads it has run (a training plan's backward pass reads the gates there).
extern "C" int stair_plan_set_projection(const stair_ctx *ctx, stair_plan *pl, float *buf, int64_t buf_floats) {
    STAIR_CHECK(ctx && pl && buf, "null argument");
    STAIR_CHECK(pl->ext_proj, "plan was not built with STAIR_PLAN_EXT_PROJECTION");
    STAIR_CHECK((reinterpret_cast<uintptr_t>(buf) & 255) == 0, "projection buffer must be 256-byte aligned");
    STAIR_CHECK(buf_floats >= proj_layout(pl->cfg, pl->n_vid, pl->T, pl->rows_q).total, "projection buffer too small (stair_projection_floats)");
    pl->proj = buf;
    return 0;
}

extern "C" int stair_plan_run(stair_ctx *ctx, stair_plan *pl, const float *video, const float *question,
                              void *workspace, int64_t workspace_bytes, float *logits, int32_t *argmax,
                              stair_stream stream) {
    return stair_plan_run_flags(ctx, pl, video, question, workspace, workspace_bytes, logits, argmax, 0, stream);
}

extern "C" int stair_plan_run_flags(stair_ctx *ctx, stair_plan *pl, const float *video, const float *question,
                                    void *workspace, int64_t workspace_bytes, float *logits, int32_t *argmax,
                                    int32_t flags, stair_stream stream) {
    STAIR_CHECK(ctx && pl && video && question && workspace, "null argument");
    PolicyScope policy_scope(&ctx->policy);
    STAIR_CHECK(workspace_bytes >= pl->total * (int64_t)sizeof(float), "workspace too small");
    STAIR_CHECK((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");
    STAIR_CHECK(memcmp(&ctx->cfg, &pl->cfg, sizeof(stair_config)) == 0, "plan was built for another configuration");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Weights W;
    if (resolve(ctx, W, false)) return 1;

    const stair_config &g = ctx->cfg;
    const int H = g.hidden_size, Hh = H / 2, V = g.video_size, E = g.text_size, A = g.answer_vocab_length;
    const int n = pl->n, T = pl->T;
    const int64_t TH = (int64_t)T * H;
    float *ws = static_cast<float *>(workspace);
    int32_t *didx = reinterpret_cast<int32_t *>(ws + pl->o_idx);
    float *vec = ws + pl->o_vec, *map = ws + pl->o_map, *att = ws + pl->o_att, *tok = ws + pl->o_tok;
    float *qfeat = ws + pl->o_qfeat;
    if (pl->train) logits = ws + pl->o_logits;       // backward reads them from the workspace
    else if (!logits) logits = ws + pl->o_logits;

    if (!(flags & STAIR_RUN_INDEX_RESIDENT))
        if (int rc_ = upload_index_image(pl, didx, s)) return rc_;
    // A new run of this plan starts clean (the backward pass keeps the word): status word + the work-queue words of the fused tile
    // launches.  By a KERNEL, not hipMemsetAsync: inside a stream capture this becomes an ordinary kernel node of the chain like
    // every other launch of the pass ("the replay abort").  The queue words reset themselves after every launch
    // (csrc/tile_mlp.hip); zeroing them here only covers a workspace that has never been used.
    uint32_t *status = reinterpret_cast<uint32_t *>(ws + pl->o_status);
    if (int rcz_ = launch_zero(status, 64 * sizeof(uint32_t), s)) return rcz_;

    struct SplitKScope {            // forward products of this call may stage split-K partials in the workspace
        explicit SplitKScope(float *p) { g_splitk_ws = p; }
        ~SplitKScope() { g_splitk_ws = nullptr; }
    } splitk_scope(ws + pl->o_splitk);

#define RUN(x) do { if (int rc_ = (x)) return rc_; } while (0)
    // nn.Dropout at the `D` positions of modules.py, training plans only; site = bucket * 8 + position (decoder: 0xffff)
    const float dp = pl->train ? pl->drop_p : 0.0f;
    int bucket_no = -1;
    auto drop = [&](float *X, int64_t gs, const int32_t *gidx, int groups, int64_t rowlen, int pos) -> int {
        if (dp <= 0.0f) return 0;
        return launch_dropout_rows(X, gs, gidx, groups, rowlen, dp, pl->drop_seed, (uint32_t)(bucket_no * 8 + pos), s);
    };
    // ---- encoders (module_net.py:74-75) ------------------------------------------------------
    {
        stair_lstm_args a = {}, t = {};
        // the projections' regions: in the workspace, or in the caller's buffer (STAIR_PLAN_EXT_PROJECTION), where STAIR_RUN_PROJECTED
        // says they already hold this batch's projections (stair_encoders_project, enqueued before the plan was built)
        STAIR_CHECK(!pl->ext_proj || pl->proj, "plan built with STAIR_PLAN_EXT_PROJECTION: call stair_plan_set_projection first");
        STAIR_CHECK(!(flags & STAIR_RUN_PROJECTED) || pl->ext_proj, "STAIR_RUN_PROJECTED needs a STAIR_PLAN_EXT_PROJECTION plan");
        float *pb = pl->ext_proj ? pl->proj : ws;
        a.x = video; a.ldx = V; a.rows = pl->n_vid * T; a.n = pl->n_vid; a.max_len = T; a.I = V; a.Hh = Hh;
        if (flags & STAIR_RUN_VIDEO_BF16) {
            STAIR_CHECK(V % 32 == 0, "bf16 clip features need video_size % 32 == 0");
            a.x = nullptr; a.x_bf16 = video; a.wih_planes_ws = pb + pl->o_wplanes;
        }
        a.seq_off = didx + pl->off_seqv;
        if (pl->ragged) a.seq_len = didx + pl->off_lenv;
        for (int d = 0; d < 2; ++d) {
            a.w_ih[d] = W.enc[0][4 * d]; a.w_hh[d] = W.enc[0][4 * d + 1];
            a.b_ih[d] = W.enc[0][4 * d + 2]; a.b_hh[d] = W.enc[0][4 * d + 3];
        }
        a.xproj_ws = pb + pl->o_xpv; a.bias_ws = pb + pl->o_bias; a.whh_pack_ws = ws + pl->o_wpack;
        a.out = map; a.ldo = H; a.h_n = ws + pl->o_vhn;
        a.coop_ws = ws + pl->o_coop; a.coop_ws_bytes = pl->coop_bytes; a.status = status;
        a.cbuf = pl->train ? ws + pl->o_cv : nullptr;

        t.x = question; t.ldx = E; t.rows = pl->rows_q; t.n = n; t.max_len = pl->max_q; t.I = E; t.Hh = Hh;
        t.seq_off = didx + pl->off_seqt;
        for (int d = 0; d < 2; ++d) {
            t.w_ih[d] = W.enc[1][4 * d]; t.w_hh[d] = W.enc[1][4 * d + 1];
            t.b_ih[d] = W.enc[1][4 * d + 2]; t.b_hh[d] = W.enc[1][4 * d + 3];
        }
        static const bool text_planes = [] { const char *e = getenv("STAIR_TEXT_PLANES"); return !(e && e[0] == '0'); }();
        if (text_planes) { t.wih_planes_ws = pb + pl->o_wplanes_t; t.x_planes_ws = pb + pl->o_xplanes_t; }
        t.xproj_ws = pb + pl->o_xpt; t.bias_ws = pb + pl->o_bias + 4 * H; t.whh_pack_ws = ws + pl->o_wpack + 2 * (int64_t)H * H;
        t.out = tok; t.ldo = H; t.h_n = qfeat;
        t.coop_ws = ws + pl->o_coop2; t.coop_ws_bytes = pl->coop_bytes; t.status = status;
        t.cbuf = pl->train ? ws + pl->o_ct : nullptr;

        // both input projections, then the two recurrences -- in ONE launch while all their workgroups fit on the chip
        // (csrc/lstm_coop.hip, lstm_rec_coop_pair_kernel), else one after the other
        if (flags & STAIR_RUN_PROJECTED) {
            RUN(launch_lstm_zero_tail(a, s));        // (the part of the projection step that writes the plan's own arena)
        } else {
            RUN(launch_lstm_project(a, s));
            RUN(launch_lstm_project(t, s));
        }
        const int rc_pair = launch_lstm_rec_coop_pair(a, t, s);
        if (rc_pair > 0) return rc_pair;
        if (rc_pair < 0) {
            RUN(launch_lstm_recur(a, s));
            RUN(launch_lstm_recur(t, s));
        }
    }

    // ---- fused per-clip tile operators (csrc/tile_mlp.hip): weights of the buckets that run fused, as fragment-order planes ----
    // (under dropout the tile operators and the grouped vector-level launches stay on: their kernels draw stair_dropout_fwd's bits in their
    // epilogues; STAIR_TILE_DROPOUT=0: launch-per-layer forms everywhere, as before ABI 6)
    const bool fused = pl->o_wfrag > 0 && tile_mlp_usable(H, T) && (dp <= 0.0f || tile_dropout_on()) && tile_policy(pl);
    pl->bits_written = fused && pl->train;
    auto WF = [&](int slot) { return static_cast<const void *>(ws + pl->o_wfrag + (int64_t)slot * H * H); };
    if (fused) {
        const Lin *lin_of[WF_COUNT] = {&W.f0[0], &W.f0[1], &W.f0[2], &W.f0[3], &W.f3[0], &W.f3[1], &W.f3[2], &W.f3[3], &W.ff0[0], &W.ff0[1], &W.ff0[2],
                                       &W.ff3[0], &W.ff3[1], &W.ff3[2], &W.ffdense, &W.hi0, &W.lv0, &W.lv3, &W.tdense};
        bool need[WF_COUNT] = {};
        for (const Bucket &b : pl->buckets) {
            if (b.cnt == 0) continue;
            switch (b.op) {
                case STAIR_OP_FILTER: need[WF_F0 + b.variant] = need[WF_F3 + b.variant] = true; break;
                case STAIR_OP_FILTERFRAME: need[WF_FF0 + b.variant] = need[WF_FF3 + b.variant] = need[WF_FFD] = true; break;
                case STAIR_OP_HASITEM: need[WF_HI0] = true; break;
                case STAIR_OP_LOCALIZE: case STAIR_OP_SUPERLATIVE: need[WF_LV0] = need[WF_LV3] = true; break;
                case STAIR_OP_TEMPORAL: need[WF_TD] = true; break;
                default: break;
            }
        }
        const float *src[WF_COUNT];
        void *dst[WF_COUNT];
        int cnt_w = 0;
        for (int i = 0; i < WF_COUNT; ++i)
            if (need[i]) { src[cnt_w] = lin_of[i]->w; dst[cnt_w] = const_cast<void *>(WF(i)); ++cnt_w; }
        if (cnt_w) RUN(launch_pack_wfrag_many(src, dst, cnt_w, H, H, s));
    }
    // vector-level modules on the tile operator (64 instances per tile; STAIR_TILE_VEC=0: the pack -> GEMM -> reduction sequences)
    // Policy (measured, ms per batch, fused / unfused): 2048 questions, buckets of 205..1044 instances: training 17.91 / 17.96,
    // inference 5.61 / 5.73; 128 questions, 13..66 instances: training 5.06 / 5.01 -- one workgroup carries a tile's whole
    // [64 x 1536 x 512] first layer (3 x 17 us) where the split-K GEMM spreads it over 64 workgroups (13 us), so the tile form
    // pays only when a bucket fills several tiles.  STAIR_TILE_VEC = minimum instances per bucket (0: never).
    const int vec_min = [] { const char *e = getenv("STAIR_TILE_VEC"); return e ? atoi(e) : 128; }();      // read per pass: tests switch it
    // The row-wise Linear layers of a level -- vector-level modules, Filter's dense layer, Localize's keyword rows, the decoder -- as
    // problems of ONE launch before the level's tile operators (first layers, keyword rows) and ONE after them (second layers, the
    // dense layers on the pooled rows): csrc/vec_group.hip.  Takes precedence over the tile form of the vector-level modules.
    const bool grouped = fused && vec_group_usable(H) && vec_group_on();
    std::vector<VgProblem> vg1, vg2;
    // [512 x 512] blocks of a row-wise weight as planes: slot + (output block) * nseg + (input segment); blocks of a [N, nseg * 512] matrix
    struct VgW { int slot, nblk, nseg; const Lin *l; int op; };
    const VgW vgw[10] = {{WV_CMP, 1, 2, &W.compare, STAIR_OP_COMPARE}, {WV_EQ, 1, 2, &W.equals, STAIR_OP_EQUALS}, {WV_XOR, 1, 3, &W.xorl, STAIR_OP_XOR},
                         {WV_TA0, 1, 2, &W.ta0, STAIR_OP_TOACTION}, {WV_TA3, 1, 1, &W.ta3, STAIR_OP_TOACTION}, {WV_EX0, 1, 3, &W.exists0, STAIR_OP_EXISTS},
                         {WV_EX3, 1, 1, &W.exists3, STAIR_OP_EXISTS}, {WV_FD, 1, 1, &W.fdense, STAIR_OP_FILTER}, {WV_LK, 1, 1, &W.lk, STAIR_OP_LOCALIZE},
                         {WV_DEC0, 2, 2, &W.dec0, -1}};
    if (grouped) {
        bool has[32] = {};
        for (const Bucket &b : pl->buckets) if (b.cnt > 0 && b.op >= 0 && b.op < 32) has[b.op] = true;
        const float *src[32];
        void *dst[32];
        int ld[32], cnt_w = 0;
        for (const VgW &v : vgw)
            if (v.op < 0 || has[v.op])
                for (int jb = 0; jb < v.nblk; ++jb)
                    for (int sg = 0; sg < v.nseg; ++sg) {
                        src[cnt_w] = v.l->w + ((int64_t)jb * H * v.nseg + sg) * H; dst[cnt_w] = const_cast<void *>(WF(v.slot + jb * v.nseg + sg));
                        ld[cnt_w] = v.nseg * H; ++cnt_w;
                    }
        if (cnt_w) RUN(launch_pack_wfrag_many(src, dst, cnt_w, H, H, s, false, ld));
    }
    auto fused_vec_for = [&](const Bucket &b) { return !grouped && fused && dp <= 0.0f && vec_min > 0 && b.cnt >= vec_min; };
    const bool fused_vec = !grouped && fused && dp <= 0.0f && vec_min > 0;
    if (fused_vec) {
        struct { int slot, nseg; const Lin *l; int op; } vw[7] = {{WV_CMP, 2, &W.compare, STAIR_OP_COMPARE}, {WV_EQ, 2, &W.equals, STAIR_OP_EQUALS},
            {WV_XOR, 3, &W.xorl, STAIR_OP_XOR}, {WV_TA0, 2, &W.ta0, STAIR_OP_TOACTION}, {WV_TA3, 1, &W.ta3, STAIR_OP_TOACTION},
            {WV_EX0, 3, &W.exists0, STAIR_OP_EXISTS}, {WV_EX3, 1, &W.exists3, STAIR_OP_EXISTS}};
        bool has[32] = {};
        for (const Bucket &b : pl->buckets) if (fused_vec_for(b) && b.op >= 0 && b.op < 32) has[b.op] = true;
        const float *src[16];
        void *dst[16];
        int ld[16], cnt_w = 0;
        for (const auto &v : vw)
            if (has[v.op])
                for (int j = 0; j < v.nseg; ++j) { src[cnt_w] = v.l->w + (int64_t)j * H; dst[cnt_w] = const_cast<void *>(WF(v.slot + j)); ld[cnt_w] = v.nseg * H; ++cnt_w; }
        if (cnt_w) RUN(launch_pack_wfrag_many(src, dst, cnt_w, H, H, s, false, ld));
    }
    // a vector-level module as queued tile work: first layer over the never-materialised concatenation of the operand rows
    // a = vec[ia[i]], b = vec[ib[i]], optional second layer, row i -> vec[io[i]]
    auto vec_module = [&](const Bucket &b, int pack, const int32_t *ia, const int32_t *ib, const int32_t *io, int slot0, const Lin &l0,
                          int slot3, const Lin *l3, float *cat_sv, float *hid_sv) {
        stair_tile_mlp_args a = {};
        a.vec_pack = pack; a.vec_cnt = b.cnt; a.cnt = (b.cnt + 63) / 64; a.T = 64; a.H = H; a.ln_eps = 1e-5f;
        a.pk_a = vec; a.pk_a_idx = ia; a.pk_b = vec; a.pk_b_idx = ib;
        a.W[0] = WF(slot0); a.bias[0] = l0.b; a.act[0] = 1; a.n_layers = 1;
        if (l3) { a.W[1] = WF(slot3); a.bias[1] = l3->b; a.act[1] = 1; a.n_layers = 2; }
        if (pl->train) { a.cat_save = cat_sv; if (l3) a.save[0] = hid_sv; }
        a.tail = STAIR_TILE_STORE_ROWS; a.out = vec; a.out_gstride = H; a.out_row_idx = io;
        return a;
    };
    auto tile_args = [&](const int32_t *x_idx, int cnt_) {
        stair_tile_mlp_args a = {};
        a.X = map; a.x_gstride = TH; a.x_idx = x_idx; a.cnt = cnt_; a.T = T; a.H = H; a.ln_eps = 1e-5f;
        return a;
    };
    // drop_pos: the layer's activation is followed by nn.Dropout, position `drop_pos` of the bucket (the `pos` of drop() below)
    auto tile_layer = [&](stair_tile_mlp_args &a, int slot, const Lin &l, int act, float *save, int64_t bits = -1, int drop_pos = -1) {
        const int i = a.n_layers++;
        a.W[i] = WF(slot); a.bias[i] = l.b; a.act[i] = act; a.save[i] = pl->train ? save : nullptr;
        a.save_bits[i] = pl->train && bits >= 0 && act == 1 ? reinterpret_cast<unsigned long long *>(ws + bits) : nullptr;
        if (dp > 0.0f && drop_pos >= 0) { a.drop_site[i] = (uint32_t)(bucket_no * 8 + drop_pos) + 1u; a.drop_p = dp; a.drop_seed = pl->drop_seed; }
    };

    // nn.Dropout behind a grouped forward problem's ReLU (position `pos` of the current bucket): the bits of drop() on the problem's [rows, N] output
    auto vg_drop = [&](VgProblem &q, int pos) {
        if (dp > 0.0f) { q.drop_site = (uint32_t)(bucket_no * 8 + pos) + 1u; q.drop_p = dp; q.drop_seed = pl->drop_seed; }
    };

    // ---- program levels ----------------------------------------------------------------------
    // One bucket's launches.  phase 0: everything (the GEMM / row-kernel sequences).  With the fused tile operators a level runs
    // in three phases: phase 1 = every bucket's work up to its tile operator, whose arguments are QUEUED; then ONE launch carries
    // the tiles of all buckets of the level (they are independent: a workgroup that finishes a tile takes the next one, whatever
    // bucket it belongs to, so a bucket's last partial round is filled by its neighbours); phase 2 = what follows the tile
    // operator (Filter's dense layer, Superlative's scores and pooling).
    std::vector<stair_tile_mlp_args> tile_queue;
    auto run_bucket = [&](const Bucket &b, const int bno, const int phase) -> int {
        bucket_no = bno;
        if (b.cnt == 0) return 0;
        const int c = b.cnt;
        const int32_t *I0 = didx + b.off[0], *I1 = didx + b.off[1], *I2 = didx + b.off[2], *I3 = didx + b.off[3],
                      *I4 = didx + b.off[4], *I5 = didx + b.off[5];
        const int32_t *LEN = pl->ragged ? didx + b.off[6] : nullptr;       // frames of each instance's clip (T-mixing operators)
        float *tmpA = ws + b.svA, *tmpB = ws + b.svB, *kbuf = ws + b.svK, *cat = ws + b.svCat, *hid = ws + b.svHid;
        float *rsb = ws + b.svRs, *sup = ws + b.svSup, *extra = ws + b.svExtra;
        const bool tile_op = b.op == STAIR_OP_FILTER || b.op == STAIR_OP_FILTERFRAME || b.op == STAIR_OP_HASITEM ||
                             b.op == STAIR_OP_LOCALIZE || b.op == STAIR_OP_SUPERLATIVE || b.op == STAIR_OP_TEMPORAL;
        if (phase == 2 && !(fused && tile_op)) return 0;              // everything else ran in phase 1
        switch (b.op) {
            case OP_SPAN:
                RUN(launch_span_mean(tok, H, I0, I1, vec, I2, c, H, s));
                break;
            case STAIR_OP_AND:
            case STAIR_OP_XORFRAME: {
                const bool isvec = b.sub == STAIR_VAL_VEC;
                RUN(launch_eltwise(b.op == STAIR_OP_AND ? 0 : 1, isvec ? vec : att, I0, I1, I2, c, isvec ? H : T, s));
                break;
            }
            case STAIR_OP_ATTNVIDEO:
                RUN(launch_attnvideo(map, I0, att, I1, I2, c, T, H, s));
                break;
            case STAIR_OP_CHOOSE:
                RUN(launch_choose(vec, I0, I1, I2, I3, c, H, s));
                break;
            case STAIR_OP_COMPARE:      // modules.py:15-21
            case STAIR_OP_EQUALS: {     // modules.py:24-37
                if (grouped) {
                    if (phase != 1) break;
                    const Lin &l = b.op == STAIR_OP_COMPARE ? W.compare : W.equals;
                    VgProblem q = vg_fwd(c, vec, I0, H, vec, I1, H, VG_IN_CAT2, l.w, 2 * H, l.b, H, 1, vec, I2, H);
                    q.wplanes = WF(b.op == STAIR_OP_COMPARE ? WV_CMP : WV_EQ);
                    if (pl->train) { q.in_save = cat; q.ld_save = 2 * H; }
                    vg1.push_back(q);
                    break;
                }
                if (fused_vec_for(b)) {
                    if (phase == 1) tile_queue.push_back(vec_module(b, 1, I0, I1, I2, b.op == STAIR_OP_COMPARE ? WV_CMP : WV_EQ,
                                                                    b.op == STAIR_OP_COMPARE ? W.compare : W.equals, 0, nullptr, cat, nullptr));
                    break;
                }
                RUN(launch_pack(PACK_CAT2, vec, I0, vec, I1, cat, c, H, s));
                RUN(dense(s, cat, 2 * H, 2 * H, nullptr, b.op == STAIR_OP_COMPARE ? W.compare : W.equals, 2 * H, vec, H, H,
                          I2, c, 1, H, 2 * H, 1));
                break;
            }
            case STAIR_OP_XOR:          // modules.py:59-72: cat[|a-b|, a, b]
                if (grouped) {
                    if (phase != 1) break;
                    VgProblem q = vg_fwd(c, vec, I0, H, vec, I1, H, VG_IN_XOR, W.xorl.w, 3 * H, W.xorl.b, H, 1, vec, I2, H);
                    q.wplanes = WF(WV_XOR);
                    if (pl->train) { q.in_save = cat; q.ld_save = 3 * H; }
                    vg1.push_back(q);
                    break;
                }
                if (fused_vec_for(b)) {
                    if (phase == 1) tile_queue.push_back(vec_module(b, 2, I0, I1, I2, WV_XOR, W.xorl, 0, nullptr, cat, nullptr));
                    break;
                }
                RUN(launch_pack(PACK_XOR, vec, I0, vec, I1, cat, c, H, s));
                RUN(dense(s, cat, 3 * H, 3 * H, nullptr, W.xorl, 3 * H, vec, H, H, I2, c, 1, H, 3 * H, 1));
                break;
            case STAIR_OP_TOACTION:     // modules.py:102-120: cat[action, keyword]
                if (grouped) {              // first layer before the level's tile operators, second layer after them
                    if (phase != 1) break;
                    VgProblem q = vg_fwd(c, vec, I0, H, vec, I1, H, VG_IN_CAT2, W.ta0.w, 2 * H, W.ta0.b, H, 1, hid, nullptr, H);
                    q.wplanes = WF(WV_TA0);
                    if (pl->train) { q.in_save = cat; q.ld_save = 2 * H; }
                    vg_drop(q, 0);
                    vg1.push_back(q);
                    vg2.push_back(vg_fwd(c, hid, nullptr, H, nullptr, nullptr, 0, VG_IN_A, W.ta3.w, H, W.ta3.b, H, 1, vec, I2, H));
                    vg2.back().wplanes = WF(WV_TA3);
                    break;
                }
                if (fused_vec_for(b)) {
                    if (phase == 1) tile_queue.push_back(vec_module(b, 1, I0, I1, I2, WV_TA0, W.ta0, WV_TA3, &W.ta3, cat, hid));
                    break;
                }
                RUN(launch_pack(PACK_CAT2, vec, I0, vec, I1, cat, c, H, s));
                RUN(dense(s, cat, 2 * H, 2 * H, nullptr, W.ta0, 2 * H, hid, H, H, nullptr, c, 1, H, 2 * H, 1));
                RUN(drop(hid, H, nullptr, c, H, 0));
                RUN(dense(s, hid, H, H, nullptr, W.ta3, H, vec, H, H, I2, c, 1, H, H, 1));
                break;
            case STAIR_OP_EXISTS:       // modules.py:141-159: Exists(keyword, feat) -> cat[feat, keyword, feat*keyword]
                if (grouped) {
                    if (phase != 1) break;
                    VgProblem q = vg_fwd(c, vec, I1, H, vec, I0, H, VG_IN_EXISTS, W.exists0.w, 3 * H, W.exists0.b, H, 1, hid, nullptr, H);
                    q.wplanes = WF(WV_EX0);
                    if (pl->train) { q.in_save = cat; q.ld_save = 3 * H; }
                    vg_drop(q, 0);
                    vg1.push_back(q);
                    vg2.push_back(vg_fwd(c, hid, nullptr, H, nullptr, nullptr, 0, VG_IN_A, W.exists3.w, H, W.exists3.b, H, 1, vec, I2, H));
                    vg2.back().wplanes = WF(WV_EX3);
                    vg_drop(vg2.back(), 1);
                    break;
                }
                if (fused_vec_for(b)) {
                    if (phase == 1) tile_queue.push_back(vec_module(b, 3, I1, I0, I2, WV_EX0, W.exists0, WV_EX3, &W.exists3, cat, hid));
                    break;
                }
                RUN(launch_pack(PACK_EXISTS, vec, I1, vec, I0, cat, c, H, s));
                RUN(dense(s, cat, 3 * H, 3 * H, nullptr, W.exists0, 3 * H, hid, H, H, nullptr, c, 1, H, 3 * H, 1));
                RUN(drop(hid, H, nullptr, c, H, 0));
                RUN(dense(s, hid, H, H, nullptr, W.exists3, H, vec, H, H, I2, c, 1, H, H, 1));
                RUN(drop(vec, H, I2, c, H, 1));
                break;
            case STAIR_OP_EXISTSFRAME:  // modules.py:162-178
                RUN(launch_cosine_attn(map, TH, I1, vec, I0, att, I2, c, T, H, s));
                break;
            case STAIR_OP_FILTER: {     // modules.py:343-378 (attention == 1 exactly, see oracle op_filter)
                const int v = b.variant;
                if (fused) {            // both layers and the sum over frames on the tile
                    if (phase == 1) {
                        stair_tile_mlp_args a = tile_args(I0, c);
                        tile_layer(a, WF_F0 + v, W.f0[v], 1, tmpA, b.bitA, 0);
                        tile_layer(a, WF_F3 + v, W.f3[v], 1, tmpB, b.bitB, 1);
                        a.tail = STAIR_TILE_SUM_ROWS; a.out = cat; a.out_gstride = H; a.len = LEN;
                        tile_queue.push_back(a);
                        if (grouped) {
                            vg2.push_back(vg_fwd(c, cat, nullptr, H, nullptr, nullptr, 0, VG_IN_A, W.fdense.w, H, W.fdense.b, H, 1, vec, I1, H));
                            vg2.back().wplanes = WF(WV_FD);
                        }
                    } else if (!grouped) {
                        RUN(dense(s, cat, H, H, nullptr, W.fdense, H, vec, H, H, I1, c, 1, H, H, 1));
                    }
                    break;
                }
                RUN(dense(s, map, H, TH, I0, W.f0[v], H, tmpA, H, TH, nullptr, c, T, H, H, 1));
                RUN(drop(tmpA, TH, nullptr, c, TH, 0));
                RUN(dense(s, tmpA, H, TH, nullptr, W.f3[v], H, tmpB, H, TH, nullptr, c, T, H, H, 1));
                RUN(drop(tmpB, TH, nullptr, c, TH, 1));
                RUN(launch_sum_rows(tmpB, cat, c, T, H, s, LEN));
                RUN(dense(s, cat, H, H, nullptr, W.fdense, H, vec, H, H, I1, c, 1, H, H, 1));
                break;
            }
            case STAIR_OP_FILTERFRAME: {   // modules.py:381-414
                const int v = b.variant;
                if (fused) {            // three layers with the sigmoid attention in between
                    if (phase != 1) break;
                    if (v == 0) RUN(launch_vecdot(vec, I1, W.ffatt.w + H, extra, c, H, s));
                    stair_tile_mlp_args a = tile_args(I0, c);
                    tile_layer(a, WF_FF0 + v, W.ff0[v], 1, tmpA, b.bitA, 0);
                    tile_layer(a, WF_FF3 + v, W.ff3[v], 1, tmpB, b.bitB, 1);
                    tile_layer(a, WF_FFD, W.ffdense, 1, nullptr, b.bitC, 2);
                    if (v == 0) { a.mid_rowdot = 1; a.vw = W.ffatt.w; a.vb = W.ffatt.b; a.extra = extra; a.rs_out = rsb; }
                    a.tail = STAIR_TILE_STORE; a.out = map; a.out_gstride = TH; a.out_idx = I2;
                    tile_queue.push_back(a);
                    break;
                }
                RUN(dense(s, map, H, TH, I0, W.ff0[v], H, tmpA, H, TH, nullptr, c, T, H, H, 1));
                RUN(drop(tmpA, TH, nullptr, c, TH, 0));
                RUN(dense(s, tmpA, H, TH, nullptr, W.ff3[v], H, tmpB, H, TH, nullptr, c, T, H, H, 1));
                RUN(drop(tmpB, TH, nullptr, c, TH, 1));
                if (v == 0) {
                    // sigmoid(Lin(2H->1)(cat[f_t, kw])) = sigmoid(w[:H].f_t + w[H:].kw + b)
                    RUN(launch_vecdot(vec, I1, W.ffatt.w + H, extra, c, H, s));
                    RUN(launch_rowdot_sigmoid(tmpB, c, T, H, W.ffatt.w, W.ffatt.b, extra, rsb, nullptr, T, s));
                    RUN(dense(s, tmpB, H, TH, nullptr, W.ffdense, H, map, H, TH, I2, c, T, H, H, 1, rsb, T, nullptr));
                } else {
                    RUN(dense(s, tmpB, H, TH, nullptr, W.ffdense, H, map, H, TH, I2, c, T, H, H, 1));
                }
                RUN(drop(map, TH, I2, c, TH, 2));
                break;
            }
            case STAIR_OP_HASITEM:      // modules.py:123-138
                if (fused) {
                    if (phase == 2) RUN(drop(att, T, I1, c, T, 1));        // (the D behind HasItem's sigmoid: after the level's tile launch)
                    if (phase != 1) break;
                    stair_tile_mlp_args a = tile_args(I0, c);
                    tile_layer(a, WF_HI0, W.hi0, 1, tmpA, b.bitA, 0);
                    a.tail = STAIR_TILE_ROWDOT_SIGMOID; a.vw = W.hi3.w; a.vb = W.hi3.b; a.out = att; a.out_gstride = T; a.out_idx = I1;
                    tile_queue.push_back(a);
                    break;
                }
                RUN(dense(s, map, H, TH, I0, W.hi0, H, tmpA, H, TH, nullptr, c, T, H, H, 1));
                RUN(drop(tmpA, TH, nullptr, c, TH, 0));
                RUN(launch_rowdot_sigmoid(tmpA, c, T, H, W.hi3.w, W.hi3.b, nullptr, att, I1, T, s));
                RUN(drop(att, T, I1, c, T, 1));
                break;
            case STAIR_OP_LOCALIZE:     // modules.py:181-217
                if (fused) {            // keyword rows first (a vector-level product), then both layers + the cosine on the tile
                    if (phase != 1) break;
                    if (grouped) {
                        vg1.push_back(vg_fwd(b.nrows, vec, I2, H, nullptr, nullptr, 0, VG_IN_A, W.lk.w, H, W.lk.b, H, 0, kbuf, nullptr, H));
                        vg1.back().wplanes = WF(WV_LK);
                    } else RUN(dense(s, vec, H, H, I2, W.lk, H, kbuf, H, H, nullptr, b.nrows, 1, H, H, 0));
                    stair_tile_mlp_args a = tile_args(I0, c);
                    tile_layer(a, WF_LV0, W.lv0, 1, tmpA, b.bitA, 0);
                    tile_layer(a, WF_LV3, W.lv3, 0, tmpB);
                    a.tail = STAIR_TILE_COSINE; a.kb = kbuf; a.pair_first = I4; a.pair_cnt = I5; a.att_idx = I3; a.att = att;
                    tile_queue.push_back(a);
                    break;
                }
                RUN(dense(s, map, H, TH, I0, W.lv0, H, tmpA, H, TH, nullptr, c, T, H, H, 1));
                RUN(drop(tmpA, TH, nullptr, c, TH, 0));
                RUN(dense(s, tmpA, H, TH, nullptr, W.lv3, H, tmpB, H, TH, nullptr, c, T, H, H, 0));
                RUN(dense(s, vec, H, H, I2, W.lk, H, kbuf, H, H, nullptr, b.nrows, 1, H, H, 0));
                RUN(launch_cosine_attn(tmpB, TH, I1, kbuf, nullptr, att, I3, b.nrows, T, H, s));
                break;
            case STAIR_OP_RELATE:       // modules.py:417-435
                RUN(launch_relate_softmax(att, I0, I1, W.beta, b.variant == 0 ? 1.0f : -1.0f, c, T, s, LEN));
                break;
            case STAIR_OP_SUPERLATIVE:  // modules.py:220-248 (shares Localize's weights, module_net.py:31-32)
                if (fused && phase == 1) {
                    stair_tile_mlp_args a = tile_args(I0, c);
                    tile_layer(a, WF_LV0, W.lv0, 1, tmpA, b.bitA, 0);
                    tile_layer(a, WF_LV3, W.lv3, 0, nullptr);
                    a.tail = STAIR_TILE_STORE; a.out = tmpB; a.out_gstride = TH;
                    tile_queue.push_back(a);
                    break;
                } else if (fused) {
                    // phase 2: the tile operator has written the clip's two-layer features to tmpB
                } else {
                    RUN(dense(s, map, H, TH, I0, W.lv0, H, tmpA, H, TH, nullptr, c, T, H, H, 1));
                    RUN(drop(tmpA, TH, nullptr, c, TH, 0));
                    RUN(dense(s, tmpA, H, TH, nullptr, W.lv3, H, tmpB, H, TH, nullptr, c, T, H, H, 0));
                }
                RUN(dense(s, ws, H, H, I4, W.lk, H, kbuf, H, H, nullptr, b.nrows, 1, H, H, 0));
                if (T <= 128 && H % 64 == 0)      // one block per instance: its pairs share the tile (csrc/rowops.hip)
                    RUN(launch_cosine_attn_grouped(tmpB, TH, kbuf, I1, I2, sup, c, b.nrows, T, H, T, s));
                else
                    RUN(launch_cosine_attn(tmpB, TH, I5, kbuf, nullptr, sup, nullptr, b.nrows, T, H, s));
                RUN(launch_superlative_pool(sup, ws, I4, I1, I2, b.variant, cat, c, T, H, s, LEN));
                RUN(dense(s, cat, H, H, nullptr, W.supdense, H, vec, H, H, I3, c, 1, H, H, 1));
                break;
            case STAIR_OP_TEMPORAL: {   // modules.py:310-327
                const int mode = b.variant;
                if (fused && phase == 2) break;
                RUN(launch_temporal_relate(att, I1, I2, att, I3, c, T, mode, ctx->conv ? 1 : 0, ctx->ksize,
                                           mode ? W.relate[mode - 1] : nullptr, s, LEN));
                if (fused) {            // r_t feat_t -> Lin . ReLU -> LayerNorm on the tile
                    if (phase != 1) break;
                    stair_tile_mlp_args a = tile_args(I0, c);
                    tile_layer(a, WF_TD, W.tdense, 1, tmpA, -1, 0);
                    a.row_scale = att; a.rs_idx = I3;
                    a.tail = STAIR_TILE_LAYERNORM; a.gamma = W.ln_w; a.beta = W.ln_b; a.out = map; a.out_gstride = TH; a.out_idx = I4;
                    tile_queue.push_back(a);
                    break;
                }
                RUN(dense(s, map, H, TH, I0, W.tdense, H, tmpA, H, TH, nullptr, c, T, H, H, 1, att, T, I3));
                RUN(drop(tmpA, TH, nullptr, c, TH, 0));
                RUN(launch_layernorm(tmpA, map, TH, I4, c, T, H, W.ln_w, W.ln_b, 1e-5f, s));
                break;
            }
            default:
                STAIR_FAIL("internal: unhandled bucket op " + std::to_string(b.op));
        }
        return 0;
    };
    if (!fused) {
        int bno = 0;
        for (const Bucket &b : pl->buckets) RUN(run_bucket(b, bno++, 0));
    } else {
        // the work queue of the forward tile launches: (head, sign-off count) at status words 16, 17.  Every launch leaves both at
        // zero and the launches of a pass follow each other on one stream, so ONE pair serves them all: no per-launch heads, no limit
        // on the number of fused launches of a plan (deep programs, levels with many buckets)
        unsigned *tile_ctr = reinterpret_cast<unsigned *>(ws + pl->o_status) + 16;
        for (size_t lo_ = 0; lo_ < pl->buckets.size();) {
            size_t hi_ = lo_;
            while (hi_ < pl->buckets.size() && pl->buckets[hi_].level == pl->buckets[lo_].level) ++hi_;
            tile_queue.clear();
            vg1.clear(); vg2.clear();
            for (size_t k = lo_; k < hi_; ++k) RUN(run_bucket(pl->buckets[k], (int)k, 1));
            if (!vg1.empty()) RUN(launch_vec_group(vg1.data(), (int)vg1.size(), s));
            static const int merge_max = [] { const char *e = getenv("STAIR_TILE_MERGE"); return e ? std::max(1, std::min(8, atoi(e))) : 8; }();
            const bool use_queue = tile_queue_on();
            for (size_t q0 = 0; q0 < tile_queue.size(); q0 += merge_max) {
                const int nq = (int)std::min<size_t>(merge_max, tile_queue.size() - q0);
                RUN(launch_tile_mlp_batch(tile_queue.data() + q0, nq, use_queue ? tile_ctr : nullptr, s));
            }
            if (!vg2.empty()) RUN(launch_vec_group(vg2.data(), (int)vg2.size(), s));
            for (size_t k = lo_; k < hi_; ++k) RUN(run_bucket(pl->buckets[k], (int)k, 2));
            lo_ = hi_;
        }
    }

    // ---- decoder (module_net.py:136-138) -----------------------------------------------------
    float *cat = ws + pl->o_cat, *hid = ws + pl->o_hid;
    if (grouped) {          // cat[root, question] never materialised for the product (kept for the weight gradient in training)
        VgProblem q = vg_fwd(n, vec, didx + pl->off_roots, H, qfeat, nullptr, H, VG_IN_CAT2, W.dec0.w, 2 * H, W.dec0.b, 2 * H, 1, hid, nullptr, 2 * H);
        q.wplanes = WF(WV_DEC0);
        if (pl->train) { q.in_save = cat; q.ld_save = 2 * H; }
        bucket_no = 0x1fff;
        vg_drop(q, 7);
        RUN(launch_vec_group(&q, 1, s));
        VgProblem q3 = vg_fwd(n, hid, nullptr, 2 * H, hid + H, nullptr, 2 * H, VG_IN_CAT2, W.dec3.w, 2 * H, W.dec3.b, A, 0, logits, nullptr, A);
        RUN(launch_vec_group(&q3, 1, s));
        if (argmax) RUN(launch_argmax(logits, argmax, n, A, s));
        return 0;
    }
    RUN(launch_pack(PACK_CAT2, vec, didx + pl->off_roots, qfeat, nullptr, cat, n, H, s));
    RUN(dense(s, cat, 2 * H, 2 * H, nullptr, W.dec0, 2 * H, hid, 2 * H, 2 * H, nullptr, n, 1, 2 * H, 2 * H, 1));
    bucket_no = 0x1fff;
    RUN(drop(hid, 2 * H, nullptr, n, 2 * H, 7));
    RUN(dense(s, hid, 2 * H, 2 * H, nullptr, W.dec3, 2 * H, logits, A, A, nullptr, n, 1, A, 2 * H, 0));
    if (argmax) RUN(launch_argmax(logits, argmax, n, A, s));
    return 0;
}

// =============================================================================================
// backward
// =============================================================================================
namespace {

// ---- deterministic weight-gradient accumulation (common.h det_shadow) ----
struct DetRange { const float *beg, *end; int id; };
struct DetState {
    bool active = false;
    const stair_ctx *early_ctx = nullptr;       // the scope was opened by stair_grad_shadows_begin for this context (the next backward pass keeps it)
    long long *base = nullptr;
    std::vector<DetRange> ranges;                // the bound gradient tensors, sorted by address
    std::vector<int64_t> off;                    // per weight id: offset of its shadow
    std::vector<char> touched;
};
thread_local DetState g_det;

bool det_enabled() {
    static const bool on = [] { const char *e = getenv("STAIR_DETERMINISTIC"); return !(e && e[0] == '0'); }();
    return on;
}

struct FxFlushBatch { float *dst[128]; long long *src[128]; int count[128]; int n; };
__global__ void fx_flush_kernel(FxFlushBatch b) {
    const int t = blockIdx.y;
    float *d = b.dst[t];
    long long *sh = b.src[t];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < b.count[t]; i += gridDim.x * blockDim.x)
        if (sh[i] != 0) { d[i] += (float)((double)sh[i] * (1.0 / 17592186044416.0)); sh[i] = 0; }     // (emptied: a second flush adds nothing twice)
}

// the shadows of every tensor touched since the last flush -> the fp32 gradients, on stream s
int det_flush(const stair_ctx *ctx, hipStream_t s) {
    DetState &d = g_det;
    if (!d.active) return 0;
    FxFlushBatch fb;
    fb.n = 0;
    int most = 0;
    auto launch = [&]() -> int {
        if (fb.n == 0) return 0;
        for (int i = fb.n; i < 128; ++i) { fb.dst[i] = nullptr; fb.src[i] = nullptr; fb.count[i] = 0; }
        hipLaunchKernelGGL(fx_flush_kernel, dim3(std::max(1, std::min((most + 255) / 256, 512)), fb.n), dim3(256), 0, s, fb);
        STAIR_LAUNCH_CHECK();
        fb.n = 0; most = 0;
        return 0;
    };
    for (size_t i = 0; i < ctx->names.size(); ++i) {
        if (!d.touched[i]) continue;
        d.touched[i] = 0;
        fb.dst[fb.n] = ctx->gptr[i]; fb.src[fb.n] = d.base + d.off[i]; fb.count[fb.n] = (int)ctx->numel[i];
        most = std::max(most, (int)ctx->numel[i]);
        if (++fb.n == 128) if (int rc = launch()) return rc;
    }
    return launch();
}

// opens the fixed-point accumulation scope on this thread: shadow layout of the context's gradient tensors, shadows zeroed if a pass
// left them dirty
// which gradient tensors are bound where (they may have been bound since the scope was opened): offsets depend on the sizes only
static int64_t det_layout(const stair_ctx *ctx) {
    DetState &d = g_det;
    d.off.assign(ctx->names.size(), 0);
    d.ranges.clear();
    int64_t o64 = 0;
    for (size_t i = 0; i < ctx->names.size(); ++i) {
        d.off[i] = o64; o64 += align_up(ctx->numel[i], 64);
        if (ctx->gptr[i]) d.ranges.push_back({ctx->gptr[i], ctx->gptr[i] + ctx->numel[i], (int)i});
    }
    std::sort(d.ranges.begin(), d.ranges.end(), [](const DetRange &a, const DetRange &b) { return a.beg < b.beg; });
    return o64;
}
int det_begin(stair_ctx *ctx, hipStream_t s) {
    DetState &d = g_det;
    d.touched.assign(ctx->names.size(), 0);
    const int64_t o64 = det_layout(ctx);
    if (!ctx->gshadow || ctx->gshadow_elems < o64) {
        if (ctx->gshadow) { STAIR_HIP(hipStreamSynchronize(s)); STAIR_HIP(hipFree(ctx->gshadow)); ctx->gshadow = nullptr; }
        STAIR_HIP(hipMalloc(&ctx->gshadow, (size_t)o64 * sizeof(long long)));
        ctx->gshadow_elems = o64;
        ctx->gshadow_dirty = true;
    }
    d.base = static_cast<long long *>(ctx->gshadow);
    if (ctx->gshadow_dirty)
        if (int rcz_ = launch_zero(d.base, o64 * (int64_t)sizeof(long long), s)) return rcz_;
    ctx->gshadow_dirty = true;                   // until a pass has flushed everything it touched
    d.active = true;
    return 0;
}

struct BwdCtx {
    hipStream_t s;
    float *wt;                       // transposed weight images
    float *splitk = nullptr;         // split-K scratch (plan workspace): dX products that overwrite their target stage partials there
    std::vector<int64_t> wt_off;     // per weight id
    std::vector<char> deferred;      // per weight id: its weight-gradient product runs once, after all buckets (per-weight regions)
    // scratch ring of the slab-reduced weight-gradient products of the remaining map-level layers (csrc/gemm_tn_x3tr.hip): a product
    // takes the next piece; when the ring wraps, the pending sums are added first (tn_x3tr_flush)
    float *tn_ring = nullptr; int64_t tn_ring_floats = 0;
    mutable int64_t tn_ring_at = 0;
};

// rows from which a per-bucket weight-gradient product takes the slab-reduced kernel (below it the atomic kernel's single pass is
// shorter); stair_set_tn_slab_min_rows lowers it so that small test batches exercise the same path
int g_tn_slab_min_rows = 4096;

// Backward of Y = act(rs * X W^T + b) given dZ (already multiplied by act'):
//   dW += dZ^T (rs * X);  db += colsum(dZ);  dX (+)= dZ W      (dX is w.r.t. the scaled input rs*X)
int dense_bwd(const BwdCtx &B, const float *dZ, int groups, int R, int N, int K, const float *X, int64_t ldx, int64_t x_gs,
              const int32_t *x_gidx, const Lin &l, float *dX, int64_t ldd, int64_t d_gs, const int32_t *d_gidx, int accumulate,
              const float *rs = nullptr, int64_t rs_gs = 0, const int32_t *rs_gidx = nullptr) {
    const int M = groups * R;
    if (M == 0) return 0;
    stair_gemm_tn_args t = {};
    t.A = dZ; t.lda = N; t.B = X; t.ldb = ldx; t.b_gstride = x_gs; t.b_gidx = x_gidx;
    t.row_scale = rs; t.rs_gstride = rs_gs; t.rs_gidx = rs_gidx;
    t.C = l.dw; t.ldc = K; t.M = M; t.rows_per_group = R; t.N = N; t.K = K;
    t.colsum = l.db;                 // db += colsum(dZ), summed while the TN kernel stages dZ
    if (!(l.id >= 0 && l.id < (int)B.deferred.size() && B.deferred[l.id])) {
        const int64_t need = M >= policy_or(STAIR_OPT_TN_SLAB_MIN_ROWS, g_tn_slab_min_rows) && B.tn_ring && tn_x3tr_takes(t) ? align_up(tn_x3tr_scratch_floats(M, N, K), 64) : 0;
        if (need && need <= B.tn_ring_floats) {
            if (B.tn_ring_at + need > B.tn_ring_floats) {
                if (int rc = tn_x3tr_flush(B.s)) return rc;
                B.tn_ring_at = 0;
            }
            if (int rc = launch_gemm_tn_x3tr(t, B.tn_ring + B.tn_ring_at, B.s)) return rc;
            B.tn_ring_at += need;
        } else if (int rc = launch_gemm_tn(t, B.s)) return rc;
    }
    if (dX) {
        stair_gemm_args g = {};
        g.A = dZ; g.lda = N; g.a_gstride = (int64_t)R * N;
        g.W = B.wt + B.wt_off[l.id]; g.ldw = N;                 // W^T [K][N]
        g.C = dX; g.ldc = ldd; g.c_gstride = d_gs; g.c_gidx = d_gidx;
        g.groups = groups; g.rows_per_group = R; g.N = K; g.K = N; g.act = 0; g.accumulate = accumulate;
        g.splitk_ws = B.splitk; g.splitk_ws_floats = B.splitk ? kSplitKFloats : 0;
        if (int rc = launch_gemm(g, B.s)) return rc;
    }
    return 0;
}

}  // namespace

extern "C" int stair_set_tn_slab_min_rows(int32_t rows) {
    STAIR_CHECK(rows >= 64, "rows must be >= 64");
    g_tn_slab_min_rows = rows;
    return 0;
}

long long *stair::det_shadow(const float *g) {
    DetState &d = g_det;
    if (!d.active || !g) return nullptr;
    size_t lo = 0, hi = d.ranges.size();
    while (lo < hi) {                            // the last range that begins at or before g
        const size_t mid = (lo + hi) / 2;
        if (d.ranges[mid].beg <= g) lo = mid + 1; else hi = mid;
    }
    if (lo == 0) return nullptr;
    const DetRange &r = d.ranges[lo - 1];
    if (g >= r.end) return nullptr;
    d.touched[r.id] = 1;
    return d.base + d.off[r.id] + (g - r.beg);
}

extern "C" int stair_plan_backward(stair_ctx *ctx, stair_plan *pl, const float *video, const float *question,
                                   void *workspace, int64_t workspace_bytes, const int32_t *answers, float loss_scale,
                                   float *loss_out, int32_t flags, stair_stream stream) {
    STAIR_CHECK(ctx && pl && video && question && workspace && answers, "null argument");
    PolicyScope policy_scope(&ctx->policy);
    STAIR_CHECK(pl->train, "plan was not built with STAIR_PLAN_TRAIN");
    // slab products queued by a pass that failed half-way must not be added into THIS pass's buffers: the queue starts empty and
    // is emptied again however this function returns
    struct PendingScope { PendingScope() { tn_x3tr_discard(); } ~PendingScope() { tn_x3tr_discard(); } } pending_scope;
    STAIR_CHECK(workspace_bytes >= pl->total * (int64_t)sizeof(float), "workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Weights W;
    if (resolve(ctx, W, true)) return 1;

    const stair_config &g = ctx->cfg;
    const int H = g.hidden_size, Hh = H / 2, V = g.video_size, E = g.text_size, A = g.answer_vocab_length;
    const float inv_keep = 1.0f / (1.0f - pl->drop_p);      // gradient factor of every ReLU -> Dropout pair (1 when dropout is off)
    const int n = pl->n, T = pl->T;
    const int64_t TH = (int64_t)T * H;
    float *ws = static_cast<float *>(workspace);
    int32_t *didx = reinterpret_cast<int32_t *>(ws + pl->o_idx);
    float *vec = ws + pl->o_vec, *map = ws + pl->o_map, *att = ws + pl->o_att;
    float *qfeat = ws + pl->o_qfeat, *logits = ws + pl->o_logits;
    float *gws = ws + pl->o_gblock - pl->o_vec;          // gradient of workspace row r lives at gws + r*H
    float *g_vec = ws + pl->o_gblock, *g_map = ws + pl->o_gblock + (pl->o_map - pl->o_vec), *g_att = ws + pl->o_gatt;
    float *g_tok = ws + pl->o_gtok, *g_qfeat = ws + pl->o_gqfeat;
    float *scrA = ws + pl->o_gA, *scrB = ws + pl->o_gB, *gK = ws + pl->o_gK, *gV0 = ws + pl->o_gV0, *gV1 = ws + pl->o_gV1;
    float *gCat = ws + pl->o_gCat, *gS = ws + pl->o_gS, *gRs = ws + pl->o_gRs, *gRs2 = ws + pl->o_gRs2;
    float *gExtra = ws + pl->o_gExtra, *gStats = ws + pl->o_gStats, *dlogits = ws + pl->o_dlogits;
    float *loss = loss_out ? loss_out : ws + pl->o_loss;

#define RUN(x) do { if (int rc_ = (x)) return rc_; } while (0)
    if (!(flags & STAIR_BWD_KEEP_ARENAS))
        if (int rcz_ = launch_zero(ws + pl->o_zero_beg, (pl->o_zero_end - pl->o_zero_beg) * sizeof(float), s)) return rcz_;

    // transposed images of every 2-D weight that needs a dX product
    BwdCtx B;
    B.s = s; B.wt = ws + pl->o_wt; B.splitk = ws + pl->o_splitk;
    B.tn_ring = pl->o_tnring ? ws + pl->o_tnring : nullptr; B.tn_ring_floats = pl->tnring_floats;
    B.wt_off.assign(ctx->names.size(), 0);
    // run-to-run reproducible weight gradients: kernels that add into a gradient tensor from several workgroups add into its
    // fixed-point shadow while this scope is open (common.h det_shadow); the shadows reach the fp32 gradients at the end of the pass.
    // stair_grad_shadows_begin may have opened the scope already (the criteria of csrc/losses.hip add their head-weight gradients
    // through it before this pass starts): then it is kept as it is, with what it has recorded as touched.
    struct DetScope {
        ~DetScope() { g_det.active = false; g_det.early_ctx = nullptr; }
    } det_scope;
    if (det_enabled() && pl->o_gshadow > 0) {
        if (!(g_det.active && g_det.early_ctx == ctx)) {
            if (int rc_ = det_begin(ctx, s)) return rc_;
        } else {
            det_layout(ctx);                 // (gradient buffers bound since then are found now; what was touched stays touched)
        }
        g_det.early_ctx = nullptr;
    } else {
        g_det.active = false;
    }
    // weight-gradient products of the tile-level layers run ONCE per weight, after all buckets (FilterFrame's dense layer keeps
    // its per-bucket product: its X operand carries the attention scale only in the tensor-keyword variant)
    const Lin *lin_of[WF_COUNT] = {&W.f0[0], &W.f0[1], &W.f0[2], &W.f0[3], &W.f3[0], &W.f3[1], &W.f3[2], &W.f3[3], &W.ff0[0], &W.ff0[1], &W.ff0[2],
                                   &W.ff3[0], &W.ff3[1], &W.ff3[2], &W.ffdense, &W.hi0, &W.lv0, &W.lv3, &W.tdense};
    B.deferred.assign(ctx->names.size(), 0);
    for (int w = 0; w < WF_COUNT; ++w)
        if (w != WF_FFD) B.deferred[lin_of[w]->id] = 1;
    const Lin *vlin_of[VD_COUNT] = {&W.compare, &W.equals, &W.xorl, &W.ta0, &W.ta3, &W.exists0, &W.exists3, &W.fdense};
    for (int w = 0; w < VD_COUNT; ++w) B.deferred[vlin_of[w]->id] = 1;
    {
        int64_t o = 0;
        for (size_t i = 0; i < ctx->names.size(); ++i) { B.wt_off[i] = o; o += align_up(ctx->numel[i], 64); }
        const Lin *lins[] = {&W.compare, &W.equals, &W.exists0, &W.exists3, &W.f0[0], &W.f0[1], &W.f0[2], &W.f0[3], &W.f3[0],
                             &W.f3[1], &W.f3[2], &W.f3[3], &W.fdense, &W.ff0[0], &W.ff0[1], &W.ff0[2], &W.ff3[0], &W.ff3[1],
                             &W.ff3[2], &W.ffdense, &W.hi0, &W.lv0, &W.lv3, &W.lk, &W.supdense, &W.tdense, &W.ta0, &W.ta3,
                             &W.xorl, &W.dec0, &W.dec3};
        TransposeBatch tb;
        tb.count = 0;
        int tiles = 0;
        static_assert(sizeof(lins) / sizeof(lins[0]) <= 32, "TransposeBatch holds 32 matrices");
        for (const Lin *l : lins) {
            const int64_t rows = ctx->numel[l->id + 1];                 // bias length = out features
            const int64_t cols = ctx->numel[l->id] / rows;
            const int m = tb.count++;
            tb.in[m] = l->w; tb.out[m] = B.wt + B.wt_off[l->id];
            tb.rows[m] = (int)rows; tb.cols[m] = (int)cols; tb.first_tile[m] = tiles;
            tiles += (int)(((rows + 31) / 32) * ((cols + 31) / 32));
        }
        RUN(launch_transpose_many(tb, tiles, s));       // one launch for all 31 images
    }
    // ---- backward chains of the fused tile operators (csrc/tile_mlp.hip): dX = (dZ2 W2 * relu'(Z1)) W1 stays on the tile ----
    const bool fused = pl->o_wfragT > 0 && tile_mlp_usable(H, T) && (pl->drop_p <= 0.0f || tile_dropout_on()) && tile_policy(pl, true);
    auto WFT = [&](int slot) { return static_cast<const void *>(ws + pl->o_wfragT + (int64_t)slot * H * H); };
    // the row-wise layers' backward as grouped launches (csrc/vec_group.hip): per level one launch for everything that starts from a
    // gradient row of the arena (relu' on load, kept as dZ), one for the second stage of the two-layer modules
    const bool grouped = fused && vec_group_usable(H) && vec_group_on();
    std::vector<VgProblem> bvg1, bvg2;
    if (grouped) {      // planes of the transposed images (already there as fp32): block (j, s) of W^T = slot + j * nin + s
        struct VgWT { int slot, nblk, nin; const Lin *l; int op; };        // W^T is [nblk * 512, nin * 512]
        const VgWT vgw[10] = {{WV_CMP, 2, 1, &W.compare, STAIR_OP_COMPARE}, {WV_EQ, 2, 1, &W.equals, STAIR_OP_EQUALS}, {WV_XOR, 3, 1, &W.xorl, STAIR_OP_XOR},
                              {WV_TA0, 2, 1, &W.ta0, STAIR_OP_TOACTION}, {WV_TA3, 1, 1, &W.ta3, STAIR_OP_TOACTION}, {WV_EX0, 3, 1, &W.exists0, STAIR_OP_EXISTS},
                              {WV_EX3, 1, 1, &W.exists3, STAIR_OP_EXISTS}, {WV_FD, 1, 1, &W.fdense, STAIR_OP_FILTER}, {WV_LK, 1, 1, &W.lk, STAIR_OP_LOCALIZE},
                              {WV_DEC0, 2, 2, &W.dec0, -1}};
        bool has[32] = {};
        for (const Bucket &b : pl->buckets) if (b.cnt > 0 && b.op >= 0 && b.op < 32) has[b.op] = true;
        const float *src[32];
        void *dst[32];
        int ld[32], cnt_w = 0;
        for (const VgWT &v : vgw)
            if (v.op < 0 || has[v.op])
                for (int jb = 0; jb < v.nblk; ++jb)
                    for (int sg = 0; sg < v.nin; ++sg) {
                        src[cnt_w] = B.wt + B.wt_off[v.l->id] + ((int64_t)jb * H * v.nin + sg) * H; dst[cnt_w] = const_cast<void *>(WFT(v.slot + jb * v.nin + sg));
                        ld[cnt_w] = v.nin * H; ++cnt_w;
                    }
        if (cnt_w) RUN(launch_pack_wfrag_many(src, dst, cnt_w, H, H, s, false, ld));
    }
    auto vslot = [&](const Lin &l) {
        const Lin *ls[10] = {&W.compare, &W.equals, &W.xorl, &W.ta0, &W.ta3, &W.exists0, &W.exists3, &W.fdense, &W.lk, &W.dec0};
        const int slots[10] = {WV_CMP, WV_EQ, WV_XOR, WV_TA0, WV_TA3, WV_EX0, WV_EX3, WV_FD, WV_LK, WV_DEC0};
        for (int i = 0; i < 10; ++i) if (ls[i]->id == l.id) return WFT(slots[i]);
        return static_cast<const void *>(nullptr);
    };
    auto vg_adj = [&](int rows, const float *a, const int32_t *ia, const float *bmask, int pack, float in_scale, float *in_save, const Lin &l, int nseg,
                      int adj, const int32_t *fia, const int32_t *fib, const int32_t *gia, const int32_t *gib) {
        VgProblem q = {};
        q.kind = VG_ADJ; q.rows = rows; q.a = a; q.ia = ia; q.lda = H; q.b = bmask; q.ib = ia; q.ldb = H; q.pack = pack; q.in_scale = in_scale; q.kred = 512;
        q.in_save = in_save; q.ld_save = H;
        q.W = B.wt + B.wt_off[l.id]; q.ldw = H; q.N = nseg * H; q.adj = adj; q.wplanes = vslot(l);
        q.fa = vec; q.fb = vec; q.fia = fia; q.fib = fib; q.ldfa = H; q.ldfb = H; q.ga = g_vec; q.gb = g_vec; q.gia = gia; q.gib = gib;
        return q;
    };
    if (fused) {
        bool need[WF_COUNT] = {};
        for (const Bucket &b : pl->buckets) {
            if (b.cnt == 0) continue;
            switch (b.op) {
                case STAIR_OP_FILTER: need[WF_F0 + b.variant] = need[WF_F3 + b.variant] = true; break;
                case STAIR_OP_FILTERFRAME: if (b.variant) need[WF_FF0 + b.variant] = need[WF_FF3 + b.variant] = need[WF_FFD] = true; break;
                case STAIR_OP_HASITEM: need[WF_HI0] = true; break;
                case STAIR_OP_LOCALIZE: case STAIR_OP_SUPERLATIVE: need[WF_LV0] = need[WF_LV3] = true; break;
                case STAIR_OP_TEMPORAL: need[WF_TD] = true; break;
                default: break;
            }
        }
        const float *src[WF_COUNT];
        void *dst[WF_COUNT];
        int cnt_w = 0;
        for (int i = 0; i < WF_COUNT; ++i)          // the transposed fp32 images are there already: their planes are those of W^T
            if (need[i]) { src[cnt_w] = B.wt + B.wt_off[lin_of[i]->id]; dst[cnt_w] = const_cast<void *>(WFT(i)); ++cnt_w; }
        if (cnt_w) RUN(launch_pack_wfrag_many(src, dst, cnt_w, H, H, s));
    }

    // ---- loss + decoder ------------------------------------------------------------------------
    RUN(launch_ce_loss(logits, answers, loss_scale, loss, dlogits, n, A, s));
    {
        const float *cat = ws + pl->o_cat, *hid = ws + pl->o_hid;     // decoder buffers are never reused by buckets in training
        if (grouped) {
            // d(hidden) = (dlogits W3) * relu'(hidden) in one launch, d(cat[root, question]) = d(hidden) W0 with the adjoint of the
            // concatenation in the epilogue in a second one; the two weight gradients stay reductions over the batch (TN products)
            RUN(dense_bwd(B, dlogits, n, 1, A, 2 * H, hid, 2 * H, 2 * H, nullptr, W.dec3, nullptr, 2 * H, 2 * H, nullptr, 0));
            VgProblem q3 = vg_fwd(n, dlogits, nullptr, A, nullptr, nullptr, 0, VG_IN_A, B.wt + B.wt_off[W.dec3.id], A, nullptr, 2 * H, 2, gV0, nullptr, 2 * H);
            q3.kred = A; q3.emask = hid; q3.ldm = 2 * H; q3.escale = inv_keep;
            RUN(launch_vec_group(&q3, 1, s));
            RUN(dense_bwd(B, gV0, n, 1, 2 * H, 2 * H, cat, 2 * H, 2 * H, nullptr, W.dec0, nullptr, 2 * H, 2 * H, nullptr, 0));
            VgProblem q0 = {};
            q0.kind = VG_ADJ; q0.rows = n; q0.a = gV0; q0.lda = 2 * H; q0.b = gV0 + H; q0.ldb = 2 * H; q0.pack = VG_IN_CAT2; q0.in_scale = 1.0f; q0.kred = 512;
            q0.W = B.wt + B.wt_off[W.dec0.id]; q0.ldw = 2 * H; q0.N = 2 * H; q0.adj = VG_IN_CAT2; q0.wplanes = WFT(WV_DEC0);
            q0.fia = didx + pl->off_roots; q0.ldfa = H; q0.ldfb = H; q0.ga = g_vec; q0.gb = g_qfeat; q0.gia = didx + pl->off_groots;
            RUN(launch_vec_group(&q0, 1, s));
        } else {
        RUN(dense_bwd(B, dlogits, n, 1, A, 2 * H, hid, 2 * H, 2 * H, nullptr, W.dec3, gV0, 2 * H, 2 * H, nullptr, 0));
        RUN(launch_mask_relu(gV0, gV0, 2 * H, nullptr, hid, 2 * H, nullptr, n, 2 * H, s, inv_keep));
        RUN(dense_bwd(B, gV0, n, 1, 2 * H, 2 * H, cat, 2 * H, 2 * H, nullptr, W.dec0, gCat, 2 * H, 2 * H, nullptr, 0));
        RUN(launch_pack_bwd(PACK_CAT2, vec, didx + pl->off_roots, qfeat, nullptr, gCat, g_vec, g_qfeat, n, H, s));
        }
    }

    // ---- program levels in reverse ---------------------------------------------------------------
    // phase 0: a bucket's whole adjoint.  With the fused chains a level runs as: phase 1 = everything up to the tile chain (its
    // arguments are queued), ONE launch for the chains of all buckets of the level, phase 2 = what needs the chain's outputs.
    std::vector<stair_tile_mlp_args> chain_queue;
    // every same-level reader of a gradient slot has a target of its own (build_grad_fanin) and the chains of a level are ONE launch
    // per kernel form: no two instances of a launch add into the same tile, so the chains add with plain read - add - write
    // (STAIR_TILE_RMW=0: float atomics)
    static const int chain_rmw = [] { const char *e = getenv("STAIR_TILE_RMW"); return (e && e[0] == '0') ? 0 : 1; }();
    // relu' masks as bits (written by the fused forward launches of THIS plan's last run; STAIR_TILE_BITS=0: the fp32 activations)
    static const bool bits_on = [] { const char *e = getenv("STAIR_TILE_BITS"); return !(e && e[0] == '0'); }();
    const bool use_bits = bits_on && pl->bits_written;
    auto BITS = [&](int64_t off) { return reinterpret_cast<const unsigned long long *>(ws + off); };
    auto bwd_bucket = [&](const Bucket &b, const int phase) -> int {
        if (b.cnt == 0) return 0;
        const int c = b.cnt;
        const int32_t *I0 = didx + b.off[0], *I1 = didx + b.off[1], *I2 = didx + b.off[2], *I3 = didx + b.off[3],
                      *I4 = didx + b.off[4], *I5 = didx + b.off[5];
        const int32_t *LEN = pl->ragged ? didx + b.off[6] : nullptr;
        // where the gradients of the operand columns go: the operand slots themselves, or staging slots (build_grad_fanin)
        const int32_t *G0 = didx + b.goff[0], *G1 = didx + b.goff[1], *G2 = didx + b.goff[2], *G4 = didx + b.goff[4];
        const float *svA = ws + b.svA, *svB = ws + b.svB, *svK = ws + b.svK, *svCat = ws + b.svCat, *svHid = ws + b.svHid;
        const float *svRs = ws + b.svRs, *svSup = ws + b.svSup;
        // dZ of the bucket's first / second tile layer: its block of the weight's region (the product with X is deferred), else scratch
        float *gA = b.dzA >= 0 ? ws + b.dzA : scrA, *gB = b.dzB >= 0 ? ws + b.dzB : scrB;
        // tail shared by Filter / FilterFrame / Localize / Superlative: gB = d(second linear output)
        // the same on the tile: the chain's two dX products, the ReLU mask between them and the accumulation into the input's
        // gradient tile in ONE launch; the two weight-gradient products (reductions over all instances) stay TN GEMMs.
        // in_bcast: Filter -- the incoming gradient is ONE row per instance (the sum over frames), broadcast and masked on load.
        auto mlp_tail_fused = [&](const Lin &l3, const Lin &l0, int slot3, int slot0, bool relu_second, const float *bcast_row) -> int {
            (void)l3; (void)l0;             // their weight-gradient products run per weight, after all buckets
            stair_tile_mlp_args a = {};
            a.cnt = c; a.T = T; a.H = H; a.len = LEN;
            if (bcast_row) { a.X = bcast_row; a.x_gstride = H; a.x_broadcast = 1; }
            else { a.X = gB; a.x_gstride = TH; }
            if (bcast_row || relu_second) {
                if (use_bits && b.bitB >= 0) a.in_bits = BITS(b.bitB); else { a.in_mask = svB; a.in_mask_gstride = TH; }
                a.in_scale = inv_keep; a.save_in = gB;
            }
            a.n_layers = 2;
            a.W[0] = WFT(slot3); a.act[0] = 3; a.act_scale = inv_keep; a.save[0] = gA;
            if (use_bits && b.bitA >= 0) a.act_bits[0] = BITS(b.bitA); else a.act_mask[0] = svA;
            a.W[1] = WFT(slot0); a.act[1] = 0;
            a.tail = STAIR_TILE_ACCUMULATE; a.out = g_map; a.out_gstride = TH; a.out_idx = G0; a.acc_exclusive = chain_rmw;
            chain_queue.push_back(a);
            return 0;
        };
        // Temporal's backward as a chain of its own kernel form (LayerNorm adjoint in, row-scale adjoint out): STAIR_TILE_TEMPORAL_BWD=0
        // keeps the four-launch sequence per bucket
        static const bool temporal_chain_on = [] { const char *e = getenv("STAIR_TILE_TEMPORAL_BWD"); return !(e && e[0] == '0'); }();
        const bool temporal_chain = fused && temporal_chain_on && b.op == STAIR_OP_TEMPORAL && b.dzA >= 0;
        const bool chain_op = fused && (b.op == STAIR_OP_FILTER || (b.op == STAIR_OP_FILTERFRAME && b.variant != 0) || b.op == STAIR_OP_HASITEM ||
                                        b.op == STAIR_OP_LOCALIZE || b.op == STAIR_OP_SUPERLATIVE || temporal_chain);
        if (phase == 2 && !chain_op) return 0;
        auto mlp_tail = [&](const Lin &l3, const Lin &l0, bool relu_second) -> int {
            if (relu_second) RUN(launch_mask_relu(gB, gB, TH, nullptr, svB, TH, nullptr, c, (int)TH, s, inv_keep));
            RUN(dense_bwd(B, gB, c, T, H, H, svA, H, TH, nullptr, l3, gA, H, TH, nullptr, 0));
            RUN(launch_mask_relu(gA, gA, TH, nullptr, svA, TH, nullptr, c, (int)TH, s, inv_keep));
            RUN(dense_bwd(B, gA, c, T, H, H, map, H, TH, I0, l0, g_map, H, TH, G0, 1));
            return 0;
        };
        switch (b.op) {
            case OP_SPAN:
                RUN(launch_span_mean_bwd_rows(g_tok, H, pl->rows_q, didx + pl->off_tok_ptr, didx + pl->off_tok_span, I1, g_vec, I2, H, s));
                break;
            case STAIR_OP_AND:
            case STAIR_OP_XORFRAME: {
                const bool isvec = b.sub == STAIR_VAL_VEC;
                RUN(launch_eltwise_bwd(b.op == STAIR_OP_AND ? 0 : 1, isvec ? vec : att, isvec ? g_vec : g_att, I0, I1, I2, c,
                                       isvec ? H : T, s, G0, G1));
                break;
            }
            case STAIR_OP_ATTNVIDEO:
                RUN(launch_attnvideo_bwd(map, g_map, att, g_att, I0, I1, I2, c, T, H, s, G0, G1));
                break;
            case STAIR_OP_CHOOSE:
                RUN(launch_choose_bwd(vec, g_vec, I0, I1, I2, I3, c, H, s, G0, G1));
                break;
            case STAIR_OP_COMPARE:
            case STAIR_OP_EQUALS:
            case STAIR_OP_XOR: {
                const bool isx = b.op == STAIR_OP_XOR;
                const int K = isx ? 3 * H : 2 * H;
                const Lin &l = isx ? W.xorl : (b.op == STAIR_OP_COMPARE ? W.compare : W.equals);
                float *dz0 = ws + b.dzV0;                 // this bucket's rows of the weight's dZ region (its product runs once, at the end)
                if (grouped) {          // dZ = g[out] * relu'(out) on load (kept), dZ W, the concatenation's adjoint: one work list entry
                    if (phase == 1) bvg1.push_back(vg_adj(c, g_vec, I2, vec, VG_IN_MASK, 1.0f, dz0, l, isx ? 3 : 2, isx ? VG_IN_XOR : VG_IN_CAT2, I0, I1, G0, G1));
                    break;
                }
                RUN(launch_mask_relu(dz0, g_vec, H, I2, vec, H, I2, c, H, s));
                RUN(dense_bwd(B, dz0, c, 1, H, K, svCat, K, K, nullptr, l, gCat, K, K, nullptr, 0));
                RUN(launch_pack_bwd(isx ? PACK_XOR : PACK_CAT2, vec, I0, vec, I1, gCat, g_vec, g_vec, c, H, s));
                break;
            }
            case STAIR_OP_TOACTION:
            case STAIR_OP_EXISTS: {
                const bool ex = b.op == STAIR_OP_EXISTS;
                const int K = ex ? 3 * H : 2 * H;
                float *dz3 = ws + b.dzV3, *dz0 = ws + b.dzV0;
                if (grouped) {
                    if (phase != 1) break;
                    // stage 1: dZ3 = g[out] * relu'(out) (kept), dZ0 = (dZ3 W3) * relu'(hidden) written to the first layer's dZ region
                    VgProblem q = vg_fwd(c, g_vec, I2, H, vec, I2, H, VG_IN_MASK, B.wt + B.wt_off[(ex ? W.exists3 : W.ta3).id], H, nullptr, H, 2, dz0, nullptr, H);
                    q.in_scale = ex ? inv_keep : 1.0f; q.in_save = dz3; q.ld_save = H; q.emask = svHid; q.ldm = H; q.escale = inv_keep;
                    q.wplanes = vslot(ex ? W.exists3 : W.ta3);
                    bvg1.push_back(q);
                    // stage 2: dZ0 W0 and the concatenation's adjoint (Exists packs [feat, keyword, feat * keyword] = rows I1, I0)
                    bvg2.push_back(vg_adj(c, dz0, nullptr, nullptr, VG_IN_A, 1.0f, nullptr, ex ? W.exists0 : W.ta0, ex ? 3 : 2, ex ? VG_IN_EXISTS : VG_IN_CAT2,
                                          ex ? I1 : I0, ex ? I0 : I1, ex ? G1 : G0, ex ? G0 : G1));
                    break;
                }
                RUN(launch_mask_relu(dz3, g_vec, H, I2, vec, H, I2, c, H, s, ex ? inv_keep : 1.0f));     // only Exists ends in ReLU . Dropout
                RUN(dense_bwd(B, dz3, c, 1, H, H, svHid, H, H, nullptr, ex ? W.exists3 : W.ta3, gV1, H, H, nullptr, 0));
                RUN(launch_mask_relu(dz0, gV1, H, nullptr, svHid, H, nullptr, c, H, s, inv_keep));
                RUN(dense_bwd(B, dz0, c, 1, H, K, svCat, K, K, nullptr, ex ? W.exists0 : W.ta0, gCat, K, K, nullptr, 0));
                if (ex) RUN(launch_pack_bwd(PACK_EXISTS, vec, I1, vec, I0, gCat, g_vec, g_vec, c, H, s));
                else RUN(launch_pack_bwd(PACK_CAT2, vec, I0, vec, I1, gCat, g_vec, g_vec, c, H, s));
                break;
            }
            case STAIR_OP_EXISTSFRAME:
                RUN(launch_cosine_attn_bwd(map, TH, I1, vec, I0, g_att, I2, g_map, g_vec, c, T, H, s, G1, G0));
                break;
            case STAIR_OP_FILTER: {
                const int v = b.variant;
                if (phase == 2) break;
                float *grow = fused ? ws + b.gRow : gV1;          // the gradient of the sum over frames: one row per instance
                float *dzf = ws + b.dzV0;
                if (grouped) {          // (grouped implies fused: the chain below reads `grow` after the level's grouped launch)
                    VgProblem q = vg_fwd(c, g_vec, I1, H, vec, I1, H, VG_IN_MASK, B.wt + B.wt_off[W.fdense.id], H, nullptr, H, 0, grow, nullptr, H);
                    q.in_save = dzf; q.ld_save = H; q.wplanes = vslot(W.fdense);
                    bvg1.push_back(q);
                } else {
                RUN(launch_mask_relu(dzf, g_vec, H, I1, vec, H, I1, c, H, s));
                RUN(dense_bwd(B, dzf, c, 1, H, H, svCat, H, H, nullptr, W.fdense, grow, H, H, nullptr, 0));
                }
                if (fused) { RUN(mlp_tail_fused(W.f3[v], W.f0[v], WF_F3 + v, WF_F0 + v, false, grow)); break; }
                RUN(launch_bcast_mask_relu(gB, gV1, svB, c, T, H, s, inv_keep, LEN));
                RUN(mlp_tail(W.f3[v], W.f0[v], false));
                break;
            }
            case STAIR_OP_FILTERFRAME: {
                const int v = b.variant;
                if (fused && v != 0) {        // three-layer chain on the tile: dZ3 -> (ffdense) -> dZ2 -> (ff3) -> dZ1 -> (ff0) -> += g_map[I0]
                    float *gC = ws + b.dzC;
                    if (phase == 2) {             // FilterFrame's dense layer keeps its per-bucket weight-gradient product (dZ3 is there now)
                        RUN(dense_bwd(B, gC, c, T, H, H, svB, H, TH, nullptr, W.ffdense, nullptr, H, TH, nullptr, 0));
                        break;
                    }
                    stair_tile_mlp_args a = {};
                    a.cnt = c; a.T = T; a.H = H;
                    a.X = g_map; a.x_gstride = TH; a.x_idx = I2;
                    if (use_bits && b.bitC >= 0) a.in_bits = BITS(b.bitC); else { a.in_mask = map; a.in_mask_gstride = TH; a.in_mask_idx = I2; }
                    a.in_scale = inv_keep; a.save_in = gC;
                    a.n_layers = 3; a.act_scale = inv_keep;
                    a.W[0] = WFT(WF_FFD); a.act[0] = 3; a.save[0] = gB;
                    a.W[1] = WFT(WF_FF3 + v); a.act[1] = 3; a.save[1] = gA;
                    if (use_bits && b.bitB >= 0) a.act_bits[0] = BITS(b.bitB); else a.act_mask[0] = svB;
                    if (use_bits && b.bitA >= 0) a.act_bits[1] = BITS(b.bitA); else a.act_mask[1] = svA;
                    a.W[2] = WFT(WF_FF0 + v); a.act[2] = 0;
                    a.tail = STAIR_TILE_ACCUMULATE; a.out = g_map; a.out_gstride = TH; a.out_idx = G0; a.acc_exclusive = chain_rmw;
                    chain_queue.push_back(a);
                    break;
                }
                RUN(launch_mask_relu(gA, g_map, TH, I2, map, TH, I2, c, (int)TH, s, inv_keep));        // dZ of the dense layer
                if (v == 0) {
                    // dense input is a_t * f_t: weight grads see the scaled input, G = dZ.W is d(a*f)
                    RUN(dense_bwd(B, gA, c, T, H, H, svB, H, TH, nullptr, W.ffdense, gB, H, TH, nullptr, 0, svRs, T, nullptr));
                    if (int rcz_ = launch_zero(gRs, (size_t)c * T * sizeof(float), s)) return rcz_;
                    if (int rcz_ = launch_zero(gExtra, (size_t)c * sizeof(float), s)) return rcz_;
                    RUN(launch_rowscale_bwd(gB, svB, TH, nullptr, svRs, T, nullptr, nullptr, gRs, c, T, H, s));   // da_t = G_t . f_t
                    RUN(launch_scale_rows(gB, svRs, (int64_t)c * T, H, s));                                        // df  = a_t * G_t
                    RUN(launch_rowdot_sigmoid_bwd(gRs, T, nullptr, svRs, T, nullptr, W.ffatt.w, gB, 1, gRs2, nullptr, c, T, H, s));
                    RUN(launch_rowsum_small(gRs2, gExtra, c, T, s));          // d(keyword term) = sum_t d(pre-sigmoid), in frame order
                    RUN(launch_weighted_colsum(svB, H, nullptr, gRs2, W.ffatt.dw, c * T, H, s));                  // d w[:H]
                    RUN(launch_weighted_colsum(vec, H, I1, gExtra, W.ffatt.dw + H, c, H, s));                     // d w[H:]
                    RUN(launch_sum_all(gRs2, W.ffatt.db, c * T, s));
                    RUN(launch_axpy_rows(g_vec, G1, gExtra, W.ffatt.w + H, c, H, s));                             // d keyword
                } else {
                    RUN(dense_bwd(B, gA, c, T, H, H, svB, H, TH, nullptr, W.ffdense, gB, H, TH, nullptr, 0));
                }
                RUN(mlp_tail(W.ff3[v], W.ff0[v], true));
                break;
            }
            case STAIR_OP_HASITEM:
                if (phase == 2) break;
                RUN(launch_rowdot_sigmoid_bwd(g_att, T, I1, att, T, I1, W.hi3.w, gA, 0, gRs2, nullptr, c, T, H, s, 1.0f / inv_keep));
                RUN(launch_weighted_colsum(svA, H, nullptr, gRs2, W.hi3.dw, c * T, H, s));
                RUN(launch_sum_all(gRs2, W.hi3.db, c * T, s));
                if (fused) {
                    stair_tile_mlp_args a = {};
                    a.cnt = c; a.T = T; a.H = H;
                    a.X = gA; a.x_gstride = TH; a.in_scale = inv_keep; a.save_in = gA;
                    if (use_bits && b.bitA >= 0) a.in_bits = BITS(b.bitA); else { a.in_mask = svA; a.in_mask_gstride = TH; }
                    a.n_layers = 1; a.W[0] = WFT(WF_HI0); a.act[0] = 0;
                    a.tail = STAIR_TILE_ACCUMULATE; a.out = g_map; a.out_gstride = TH; a.out_idx = G0; a.acc_exclusive = chain_rmw;
                    chain_queue.push_back(a);
                    break;
                }
                RUN(launch_mask_relu(gA, gA, TH, nullptr, svA, TH, nullptr, c, (int)TH, s, inv_keep));
                RUN(dense_bwd(B, gA, c, T, H, H, map, H, TH, I0, W.hi0, g_map, H, TH, G0, 1));
                break;
            case STAIR_OP_LOCALIZE:
                if (phase == 2) break;
                RUN(launch_cosine_attn_bwd_grouped(svB, svK, att, I3, g_att, I3, I4, I5, gB, gK, gRs2, gStats, c, b.nrows, T, H, 2, s));
                if (grouped) {          // dW_lk stays a reduction over the keyword rows; d(keyword rows) joins the level's grouped launch
                    RUN(dense_bwd(B, gK, b.nrows, 1, H, H, vec, H, H, I2, W.lk, nullptr, H, H, I2, 1));
                    VgProblem q = vg_fwd(b.nrows, gK, nullptr, H, nullptr, nullptr, 0, VG_IN_A, B.wt + B.wt_off[W.lk.id], H, nullptr, H, 0, g_vec, G2, H);
                    q.accumulate = 1; q.wplanes = vslot(W.lk);
                    bvg1.push_back(q);
                } else
                RUN(dense_bwd(B, gK, b.nrows, 1, H, H, vec, H, H, I2, W.lk, g_vec, H, H, G2, 1));
                if (fused) RUN(mlp_tail_fused(W.lv3, W.lv0, WF_LV3, WF_LV0, false, nullptr));
                else RUN(mlp_tail(W.lv3, W.lv0, false));
                break;
            case STAIR_OP_RELATE:
                RUN(launch_relate_softmax_bwd(att, g_att, I0, I1, W.dbeta, b.variant == 0 ? 1.0f : -1.0f, c, T, s, LEN, G0));
                break;
            case STAIR_OP_SUPERLATIVE:
                if (phase == 2) break;
                RUN(launch_mask_relu(gV0, g_vec, H, I3, vec, H, I3, c, H, s));
                RUN(dense_bwd(B, gV0, c, 1, H, H, svCat, H, H, nullptr, W.supdense, gV1, H, H, nullptr, 0));
                RUN(launch_superlative_pool_bwd(svSup, ws, gws, I4, I1, I2, b.variant, gV1, gS, c, T, H, s, LEN, G4));
                RUN(launch_cosine_attn_bwd_grouped(svB, svK, svSup, nullptr, gS, nullptr, I1, I2, gB, gK, gRs2, gStats, c, b.nrows, T, H, T, s));
                RUN(dense_bwd(B, gK, b.nrows, 1, H, H, ws, H, H, I4, W.lk, gws, H, H, G4, 1));
                if (fused) RUN(mlp_tail_fused(W.lv3, W.lv0, WF_LV3, WF_LV0, false, nullptr));
                else RUN(mlp_tail(W.lv3, W.lv0, false));
                break;
            case STAIR_OP_TEMPORAL: {
                const int mode = b.variant;
                if (temporal_chain) {
                    if (phase == 2) {       // the chain has left d(related attention) in g_att[I3]
                        RUN(launch_temporal_relate_bwd(att, I1, I2, g_att, I3, g_att, c, T, mode, ctx->conv ? 1 : 0, ctx->ksize,
                                                       mode ? W.relate[mode - 1] : nullptr, mode ? W.drelate[mode - 1] : nullptr, s, LEN, G1));
                        break;
                    }
                    // d(output tile) -> LayerNorm adjoint * relu' = dZ (kept in the weight's region) -> dZ W -> += r_t . into the
                    // input's gradient tile, d r_t = (dZ W)_t . feat_t: one chain per tile, all Temporal buckets of the level in one launch
                    stair_tile_mlp_args a = {};
                    a.cnt = c; a.T = T; a.H = H; a.ln_eps = 1e-5f;
                    a.X = g_map; a.x_gstride = TH; a.x_idx = I4;
                    a.ln_bwd = 1; a.in_mask = svA; a.in_mask_gstride = TH; a.in_scale = inv_keep; a.save_in = gA;
                    a.gamma = W.ln_w; a.dgamma = W.dln_w; a.dbeta = W.dln_b;
                    a.n_layers = 1; a.W[0] = WFT(WF_TD); a.act[0] = 0;
                    a.tail = STAIR_TILE_ROWSCALE_ADJ; a.out = g_map; a.out_gstride = TH; a.out_idx = G0; a.acc_exclusive = chain_rmw;
                    a.adj_feat = map; a.adj_feat_gstride = TH; a.adj_feat_idx = I0;
                    a.adj_rs = att; a.adj_rs_idx = I3; a.adj_drs = g_att;
                    chain_queue.push_back(a);
                    break;
                }
                RUN(launch_layernorm_bwd(g_map, TH, I4, svA, c, T, H, W.ln_w, 1e-5f, gA, gStats, W.dln_w, W.dln_b, s, inv_keep));
                RUN(dense_bwd(B, gA, c, T, H, H, map, H, TH, I0, W.tdense, gB, H, TH, nullptr, 0, att, T, I3));
                RUN(launch_rowscale_bwd(gB, map, TH, I0, att, T, I3, g_map, g_att, c, T, H, s, G0));
                RUN(launch_temporal_relate_bwd(att, I1, I2, g_att, I3, g_att, c, T, mode, ctx->conv ? 1 : 0, ctx->ksize,
                                               mode ? W.relate[mode - 1] : nullptr, mode ? W.drelate[mode - 1] : nullptr, s, LEN, G1));
                break;
            }
            default:
                STAIR_FAIL("internal: unhandled bucket op " + std::to_string(b.op));
        }
        return 0;
    };
    // deterministic fan-in: before a level's nodes use the gradients of their outputs, the contributions that same-level readers
    // parked in staging slots are added to those outputs' gradient slots in a fixed order (build_grad_fanin)
    std::vector<char> fanin_done(pl->n_levels + 2, 0);
    auto fanin_level = [&](int L) -> int {
        if (L < 0 || L >= (int)pl->fanin_count.size() || fanin_done[L]) return 0;
        fanin_done[L] = 1;
        return launch_grad_fanin(didx + pl->off_fanin + 5 * pl->fanin_first[L], pl->fanin_count[L], g_vec, g_map, g_att, H, T, s);
    };
    if (!fused) {
        for (auto it = pl->buckets.rbegin(); it != pl->buckets.rend(); ++it) {
            RUN(fanin_level(it->level));
            RUN(bwd_bucket(*it, 0));
        }
    } else {
        // queue words of the backward chains: status words 18, 19 (zeroed by the forward run, left at zero by every launch)
        unsigned *chain_ctr = reinterpret_cast<unsigned *>(ws + pl->o_status) + 18;
        const bool use_queue = tile_queue_on();
        for (int64_t hi_ = (int64_t)pl->buckets.size(); hi_ > 0;) {
            int64_t lo_ = hi_;
            while (lo_ > 0 && pl->buckets[lo_ - 1].level == pl->buckets[hi_ - 1].level) --lo_;
            chain_queue.clear();
            bvg1.clear(); bvg2.clear();
            RUN(fanin_level(pl->buckets[hi_ - 1].level));
            for (int64_t k = hi_ - 1; k >= lo_; --k) RUN(bwd_bucket(pl->buckets[k], 1));
            if (!bvg1.empty()) RUN(launch_vec_group(bvg1.data(), (int)bvg1.size(), s));
            if (!bvg2.empty()) RUN(launch_vec_group(bvg2.data(), (int)bvg2.size(), s));
            for (size_t q0 = 0; q0 < chain_queue.size(); q0 += 8) {
                const int nq = (int)std::min<size_t>(8, chain_queue.size() - q0);
                RUN(launch_tile_mlp_batch(chain_queue.data() + q0, nq, use_queue ? chain_ctr : nullptr, s));
            }
            for (int64_t k = hi_ - 1; k >= lo_; --k) RUN(bwd_bucket(pl->buckets[k], 2));
            hi_ = lo_;
        }
    }

    for (int L = pl->n_levels; L >= 0; --L) RUN(fanin_level(L));        // levels without a bucket of their own (clip tiles: before BPTT)

    // ---- the deferred weight-gradient products: one long reduction per weight -------------------------------
    // They are leaves (nothing downstream reads dW before the optimizer), and what follows on `s` -- BPTT through both encoders,
    // HBM-bound on its saved state -- touches none of their operands, so they CAN run on a second stream beside it
    // (STAIR_BWD_OVERLAP=1).  Measured (profiles/r03_*): 19.11 ms per 2048-question step with the overlap, 18.72 ms without,
    // 5.53 against 5.44 ms at 128 questions -- the two streams contend for the same LDS / L2 / HBM paths; off by default.
    static const bool overlap_tn = [] { const char *e = getenv("STAIR_BWD_OVERLAP"); return e && e[0] == '1'; }();
    hipStream_t s_tn = s;
    if (overlap_tn) {
        RUN(tn_x3tr_flush(s));               // sums queued on `s` by the buckets stay on `s`
        if (!ctx->side) STAIR_HIP(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        if (!ctx->ev_fork) {
            STAIR_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            STAIR_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        }
        STAIR_HIP(hipEventRecord(ctx->ev_fork, s));
        STAIR_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        s_tn = ctx->side;
    }
    for (int w = 0; w < WF_COUNT; ++w) {
        if (w == WF_FFD || pl->wg_rows[w] == 0) continue;
        const Lin &l = *lin_of[w];
        stair_gemm_tn_args t = {};
        t.A = ws + pl->wg_dz[w]; t.lda = H;
        t.C = l.dw; t.ldc = H; t.colsum = l.db;
        t.M = (int)(pl->wg_rows[w] * T); t.rows_per_group = T; t.N = H; t.K = H;
        STAIR_CHECK(pl->wg_rows[w] * T < (1ll << 31), "batch too large for one weight-gradient product");
        if (pl->wg_sx[w]) {                   // second layer: X = the saved first activations, in the same order
            t.B = ws + pl->wg_sx[w]; t.ldb = H; t.b_gstride = TH;
        } else {                              // first layer: X = the instances' input tiles
            t.B = map; t.ldb = H; t.b_gstride = TH; t.b_gidx = didx + pl->wg_off_idx[w];
            if (w == WF_TD) { t.row_scale = att; t.rs_gstride = T; t.rs_gidx = didx + pl->wg_off_rs[w]; }
        }
        if (pl->wg_part[w] && tn_x3tr_takes(t)) RUN(launch_gemm_tn_x3tr(t, ws + pl->wg_part[w], s_tn));      // deterministic: no atomics
        else RUN(launch_gemm_tn(t, s_tn));
    }
    // the vector-level weights (and Filter's dense layer on the pooled rows): one product per weight over the rows of all its
    // buckets; from 2048 rows on through the slab kernel (whole 32-row stages; the < 32 rows left over by the atomic kernel, one
    // add per element), below that the atomic kernel's single pass is shorter
    std::vector<stair_gemm_tn_args> small_tn;
    for (int w = 0; w < VD_COUNT; ++w) {
        if (pl->vd_rows[w] == 0) continue;
        const Lin &l = *vlin_of[w];
        const int Kw = vd_cols(w, H);
        stair_gemm_tn_args t = {};
        t.A = ws + pl->vd_dz[w]; t.lda = H;
        t.B = ws + pl->vd_x[w]; t.ldb = Kw; t.b_gstride = Kw; t.rows_per_group = 1;
        t.C = l.dw; t.ldc = Kw; t.colsum = l.db;
        t.M = (int)pl->vd_rows[w]; t.N = H; t.K = Kw;
        stair_gemm_tn_args h = t;
        h.M = t.M & ~31;
        const int64_t need = h.M >= 2048 && B.tn_ring && tn_x3tr_takes(h) ? align_up(tn_x3tr_scratch_floats(h.M, H, Kw), 64) : 0;
        if (need && need <= B.tn_ring_floats) {
            if (B.tn_ring_at + need > B.tn_ring_floats) { RUN(tn_x3tr_flush(s_tn)); B.tn_ring_at = 0; }
            RUN(launch_gemm_tn_x3tr(h, B.tn_ring + B.tn_ring_at, s_tn));
            B.tn_ring_at += need;
            if (t.M == h.M) continue;
            t.A += (int64_t)h.M * t.lda; t.B += (int64_t)h.M * t.ldb; t.M -= h.M;
        }
        small_tn.push_back(t);
    }
    if (!small_tn.empty()) RUN(launch_gemm_tn_batch(small_tn.data(), (int)small_tn.size(), s_tn));
    RUN(tn_x3tr_flush(s_tn));                 // dW, db += the slabs of every product above, in slab order: one launch
    if (overlap_tn) STAIR_HIP(hipEventRecord(ctx->ev_join, ctx->side));
    // Every gradient except the two encoders' is final here (decoder, all module levels, their weight-gradient products): a
    // data-parallel trainer starts reducing that part of its bucket now, beside the BPTT below (stair_plan_set_backward_event)
    if (overlap_tn && (pl->bwd_event || g_det.active)) STAIR_HIP(hipStreamWaitEvent(s, ctx->ev_join, 0));
    RUN(det_flush(ctx, s));                   // ... including what went through the fixed-point shadows (module levels, decoder)
    if (pl->bwd_event) STAIR_HIP(hipEventRecord(static_cast<hipEvent_t>(pl->bwd_event), s));

    // ---- encoders ------------------------------------------------------------------------------------
    {
        stair_lstm_bwd_args enc[2] = {};
        for (int e = 0; e < 2; ++e) {
            stair_lstm_bwd_args &a = enc[e];
            if (e == 0) {
                a.x = video; a.ldx = V; a.rows = pl->n_vid * T; a.max_len = T; a.I = V; a.seq_off = didx + pl->off_seqv;
                if (pl->ragged) a.seq_len = didx + pl->off_lenv;
                if (flags & STAIR_RUN_VIDEO_BF16) { a.x = nullptr; a.x_bf16 = video; }
                a.gates = (pl->ext_proj ? pl->proj : ws) + pl->o_xpv; a.cbuf = ws + pl->o_cv; a.out = map; a.d_out = g_map; a.d_hn = nullptr;
                a.whh_pack_ws = ws + pl->o_wpack;
                a.coop_ws = ws + pl->o_coop;
            } else {
                a.x = question; a.ldx = E; a.rows = pl->rows_q; a.max_len = pl->max_q; a.I = E; a.seq_off = didx + pl->off_seqt;
                a.gates = (pl->ext_proj ? pl->proj : ws) + pl->o_xpt; a.cbuf = ws + pl->o_ct; a.out = ws + pl->o_tok; a.d_out = g_tok; a.d_hn = g_qfeat;
                a.whh_pack_ws = ws + pl->o_wpack + 2 * (int64_t)H * H;
                a.coop_ws = ws + pl->o_coop2;
            }
            a.n = e == 0 ? pl->n_vid : n; a.Hh = Hh; a.ldo = H; a.ldd = H;
            a.hprev_ws = ws + pl->o_hprev + (e == 0 ? 0 : (int64_t)pl->n_vid * T * H);
            a.coop_ws_bytes = pl->coop_bytes;
            a.status = reinterpret_cast<uint32_t *>(ws + pl->o_status);
            if (pl->o_tnenc[e]) { a.tn_ws = ws + pl->o_tnenc[e]; a.tn_ws_floats = pl->tnenc_floats[e]; }
            for (int d = 0; d < 2; ++d) {
                a.w_hh[d] = W.enc[e][4 * d + 1];
                a.dw_ih[d] = W.denc[e][4 * d]; a.dw_hh[d] = W.denc[e][4 * d + 1];
                a.db_ih[d] = W.denc[e][4 * d + 2]; a.db_hh[d] = W.denc[e][4 * d + 3];
            }
        }
        // the two reverse-time recurrences in one launch while both fit on the chip, then each layer's weight gradients
        int rc_pair = -1;
        if (lstm_bwd_takes_coop(enc[0]) && lstm_bwd_takes_coop(enc[1])) rc_pair = launch_lstm_bwd_coop_pair(enc[1], enc[0], s);
        if (rc_pair > 0) return rc_pair;
        if (rc_pair < 0) {
            // Two independent passes (disjoint inputs, workspaces and gradient buffers; their slab partials are added after the join), so the
            // text encoder's backward CAN run beside the video encoder's on a second stream (STAIR_ENC_OVERLAP=1).  Measured: nothing to gain
            // at 512 or 2048 questions (6.13 vs 6.12 ms; 16.4-16.7 vs 15.9-16.8) -- the one-workgroup reverse recurrence takes 256 registers x
            // 512 threads, so from 1024 sequences on either pass fills every CU by itself.  Off by default.
            static const bool enc_overlap = [] { const char *e = getenv("STAIR_ENC_OVERLAP"); return e && e[0] == '1'; }();
            if (enc_overlap) {
                if (!ctx->side) STAIR_HIP(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
                if (!ctx->ev_fork_enc) {
                    STAIR_HIP(hipEventCreateWithFlags(&ctx->ev_fork_enc, hipEventDisableTiming));
                    STAIR_HIP(hipEventCreateWithFlags(&ctx->ev_join_enc, hipEventDisableTiming));
                }
                STAIR_HIP(hipEventRecord(ctx->ev_fork_enc, s));
                STAIR_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_fork_enc, 0));
                const int rc_t = launch_lstm_bwd(enc[1], ctx->side);
                const int rc_v = rc_t ? 0 : launch_lstm_bwd(enc[0], s);
                STAIR_HIP(hipEventRecord(ctx->ev_join_enc, ctx->side));       // joined on every path: nothing of this pass is left running on the side stream
                STAIR_HIP(hipStreamWaitEvent(s, ctx->ev_join_enc, 0));
                if (rc_t) return rc_t;
                if (rc_v) return rc_v;
            } else {
                RUN(launch_lstm_bwd(enc[1], s));
                RUN(launch_lstm_bwd(enc[0], s));
            }
        } else {
            RUN(launch_lstm_bwd_weights(enc[1], s));
            RUN(launch_lstm_bwd_weights(enc[0], s));
        }
    }
    RUN(tn_x3tr_flush(s));                   // the encoders' slab-reduced weight gradients
    if (overlap_tn) STAIR_HIP(hipStreamWaitEvent(s, ctx->ev_join, 0));      // the optimizer (next on `s`) sees every dW
    RUN(det_flush(ctx, s));                  // the encoders' fixed-point shadows -> the fp32 gradients
    if (g_det.active) ctx->gshadow_dirty = false;    // every touched shadow has been emptied again
    g_det.active = false;
#undef RUN
    return 0;
}

// Which weights receive a gradient from this plan (1) and which do not (0).  torch leaves .grad = None
// for parameters of modules no program in the window used, and Adam then skips them entirely
// (train_module.py:408-410); the optimizer kernel reproduces that with this mask.
extern "C" int stair_plan_touched(const stair_ctx *ctx, const stair_plan *pl, int32_t *touched, int32_t count) {
    STAIR_CHECK(ctx && pl && touched, "null argument");
    STAIR_CHECK(count == (int)ctx->names.size(), "count must equal stair_weight_count");
    for (int i = 0; i < count; ++i) touched[i] = 0;
    auto mark = [&](const std::string &name) {
        auto it = ctx->by_name.find("submodules." + name);
        if (it != ctx->by_name.end()) touched[it->second] = 1;
    };
    auto lin = [&](const std::string &prefix) { mark(prefix + ".weight"); mark(prefix + ".bias"); };
    for (const char *enc : {"video_encoder", "text_encoder"})
        for (const char *sfx : {"", "_reverse"})
            for (const char *w : {"weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"}) mark(std::string(enc) + "." + w + sfx);
    lin("decoder.0"); lin("decoder.3");
    const char *fk[4] = {"representation", "actions", "objects", "relations"};
    const char *ffk[3] = {"representation", "relations", "actions"};
    const char *modes[4] = {"", "before", "after", "between"};
    for (const Bucket &b : pl->buckets) {
        if (b.cnt == 0) continue;
        switch (b.op) {
            case STAIR_OP_COMPARE: lin("Compare.param.0"); break;
            case STAIR_OP_EQUALS: lin("Equals.param.0"); break;
            case STAIR_OP_XOR: lin("Xor.param.0"); break;
            case STAIR_OP_TOACTION: lin("ToAction.param.0"); lin("ToAction.param.3"); break;
            case STAIR_OP_EXISTS: lin("Exists.param.0"); lin("Exists.param.3"); break;
            case STAIR_OP_FILTER:
                lin(std::string("Filter.param.") + fk[b.variant] + ".0"); lin(std::string("Filter.param.") + fk[b.variant] + ".3");
                lin("Filter.dense.0");
                if (b.variant == 0) lin("Filter.attention.0");     // receives an all-zero gradient (softmax over one element)
                break;
            case STAIR_OP_FILTERFRAME:
                lin(std::string("FilterFrame.param.") + ffk[b.variant] + ".0"); lin(std::string("FilterFrame.param.") + ffk[b.variant] + ".3");
                lin("FilterFrame.dense.0");
                if (b.variant == 0) lin("FilterFrame.attention.0");
                break;
            case STAIR_OP_HASITEM: lin("HasItem.param.0"); lin("HasItem.param.3"); break;
            case STAIR_OP_LOCALIZE: lin("Localize.video_linear.0"); lin("Localize.video_linear.3"); lin("Localize.keyword_linear.0"); break;
            case STAIR_OP_SUPERLATIVE:
                lin("Localize.video_linear.0"); lin("Localize.video_linear.3"); lin("Localize.keyword_linear.0");
                lin("Superlative.dense.0");
                break;
            case STAIR_OP_RELATE: mark("Relate.beta"); break;
            case STAIR_OP_TEMPORAL:
                lin("Temporal.dense.0"); mark("Temporal.layer_norm.weight"); mark("Temporal.layer_norm.bias");
                if (b.variant)
                    for (int l : {0, 2, 4}) lin(std::string("Temporal.relate.") + modes[b.variant] + "." + std::to_string(l));
                break;
            default: break;
        }
    }
    return 0;
}

extern "C" int stair_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq,
                               const int32_t *seg_of_block, const int32_t *touched, const float *step_of_seg, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int64_t n, const uint32_t *guard,
                               stair_stream stream) {
    STAIR_CHECK(params && grads && exp_avg && exp_avg_sq && seg_of_block && touched && step_of_seg, "null argument");
    return launch_adam(params, grads, exp_avg, exp_avg_sq, seg_of_block, touched, step_of_seg, lr, beta1, beta2, eps, weight_decay, n,
                       guard, static_cast<hipStream_t>(stream));
}

// Layout introspection for tests: every workspace region as (name, begin, end) in floats.
// Returns the number of regions; fills up to `cap` entries.  `names` receives pointers to static strings
// or to strings owned by the plan (valid until the plan is destroyed).
extern "C" int stair_plan_regions(stair_plan *pl, const stair_ctx *ctx, const char **names, int64_t *beg, int64_t *end, int32_t cap) {
    if (!pl || !ctx) return -1;
    static thread_local std::vector<std::string> store;
    store.clear();
    std::vector<std::tuple<std::string, int64_t, int64_t>> r;
    const int64_t H = ctx->cfg.hidden_size, A = ctx->cfg.answer_vocab_length, T = pl->T, n = pl->n;
    auto add = [&](const std::string &nm, int64_t b, int64_t len) { if (len > 0) r.emplace_back(nm, b, b + len); };
    add("idx", pl->o_idx, (int64_t)pl->idx.size());
    add("vec", pl->o_vec, (int64_t)pl->n_vec * H);
    add("map", pl->o_map, (int64_t)pl->n_map * T * H);
    add("att", pl->o_att, (int64_t)std::max(pl->n_att, 1) * T);
    add("tok", pl->o_tok, (int64_t)pl->rows_q * H);
    add("qfeat", pl->o_qfeat, n * H);
    add("vhn", pl->o_vhn, (int64_t)pl->n_vid * H);
    if (!pl->ext_proj) {
    add("xpv", pl->o_xpv, (int64_t)pl->n_vid * T * 4 * H);
    add("xpt", pl->o_xpt, (int64_t)pl->rows_q * 4 * H);
    }
    if (!pl->ext_proj) add("bias", pl->o_bias, 8 * H);
    add("wpack", pl->o_wpack, 4 * H * H);
    if (!pl->ext_proj) {
    if (ctx->cfg.video_size % 32 == 0) add("wplanes", pl->o_wplanes, 4 * H * ctx->cfg.video_size);
    add("wplanes_t", pl->o_wplanes_t, 4 * H * ((ctx->cfg.text_size + 31) / 32 * 32));
    add("xplanes_t", pl->o_xplanes_t, (int64_t)std::max(pl->rows_q, 1) * ((ctx->cfg.text_size + 31) / 32 * 32));
    }
    add("coop", pl->o_coop, (pl->coop_bytes + 3) / 4);
    add("coop2", pl->o_coop2, (pl->coop_bytes + 3) / 4);
    add("splitk", pl->o_splitk, kSplitKFloats);
    add("tmpA", pl->o_tmpA, (int64_t)std::max(pl->maxI, 1) * T * H);
    add("tmpB", pl->o_tmpB, (int64_t)std::max(pl->maxI, 1) * T * H);
    add("kbuf", pl->o_kbuf, (int64_t)std::max(pl->maxK, 1) * H);
    add("cat", pl->o_cat, (int64_t)pl->maxV * 3 * H);
    add("hid", pl->o_hid, (int64_t)pl->maxV * 2 * H);
    add("rs", pl->o_rs, (int64_t)std::max(pl->maxI, 1) * T);
    add("sup", pl->o_sup, (int64_t)std::max(pl->maxSupRows, 1) * T);
    add("extra", pl->o_extra, std::max(pl->maxI, 1));
    add("logits", pl->o_logits, n * A);
    add("status", pl->o_status, 128);
    if (H == 512 && T <= 64) add("wfrag", pl->o_wfrag, 19 * H * H);
    if (pl->train) {
        int bi = 0;
        for (const Bucket &b : pl->buckets) {
            const std::string p = "b" + std::to_string(bi++) + "(op" + std::to_string(b.op) + "v" + std::to_string(b.variant) + ").";
            const int64_t c = b.cnt;
            if (b.svA != pl->o_tmpA) add(p + "svA", b.svA, c * T * H);
            if (b.svB != pl->o_tmpB) add(p + "svB", b.svB, c * T * H);
            if (b.svK != pl->o_kbuf) add(p + "svK", b.svK, (int64_t)b.nrows * H);
            int v0, v3;
            bucket_vec_weights(b.op, v0, v3);
            if (b.svCat != pl->o_cat) add(p + "svCat", b.svCat, v0 >= 0 ? c * vd_cols(v0, H) : c * H);
            if (b.dzV0 >= 0) add(p + "dzV0", b.dzV0, c * H);
            if (b.dzV3 >= 0) add(p + "dzV3", b.dzV3, c * H);
            if (b.svHid != pl->o_hid) add(p + "svHid", b.svHid, c * H);
            if (b.svRs != pl->o_rs) add(p + "svRs", b.svRs, c * T);
            if (b.svSup != pl->o_sup) add(p + "svSup", b.svSup, (int64_t)b.nrows * T);
            if (b.svExtra != pl->o_extra) add(p + "svExtra", b.svExtra, c);
        }
        const int64_t I = std::max(pl->maxI, 1), Vv = pl->maxV;
        add("cv", pl->o_cv, (int64_t)pl->n_vid * T * H);
        add("ct", pl->o_ct, (int64_t)pl->rows_q * H);
        add("hprev", pl->o_hprev, ((int64_t)pl->n_vid * T + pl->rows_q) * H);
        add("wt", pl->o_wt, ctx_weight_floats(ctx));
        add("gA", pl->o_gA, I * T * H);
        add("gB", pl->o_gB, I * T * H);
        if (pl->o_wfragT > 0) add("wfragT", pl->o_wfragT, (int64_t)WV_END * H * H);
        for (const Bucket &b : pl->buckets) {
            if (b.dzC >= 0) add("dzC", b.dzC, (int64_t)b.cnt * T * H);
            if (b.gRow >= 0) add("gRow", b.gRow, (int64_t)b.cnt * H);
        }
        for (int w = 0; w < WF_COUNT; ++w) {
            if (pl->wg_rows[w]) add("wg_dz" + std::to_string(w), pl->wg_dz[w], pl->wg_rows[w] * T * H);
            if (pl->wg_part[w]) add("wg_part" + std::to_string(w), pl->wg_part[w], tn_x3tr_scratch_floats(pl->wg_rows[w] * T, H, H));
        }
        add("gV0", pl->o_gV0, Vv * 2 * H);
        add("gV1", pl->o_gV1, Vv * 2 * H);
        add("gCat", pl->o_gCat, Vv * 3 * H);
        add("gStats", pl->o_gStats, I * T * 2);
        add("gRs2", pl->o_gRs2, I * T);
        add("dlogits", pl->o_dlogits, n * A);
        add("loss", pl->o_loss, n);
        add("gblock", pl->o_gblock, pl->o_map + (int64_t)pl->n_map * T * H - pl->o_vec);
        add("gatt", pl->o_gatt, (int64_t)std::max(pl->n_att, 1) * T);
        add("gtok", pl->o_gtok, (int64_t)pl->rows_q * H);
        add("gqfeat", pl->o_gqfeat, n * H);
        add("gK", pl->o_gK, (int64_t)std::max(pl->maxK, 1) * H);
        add("gS", pl->o_gS, (int64_t)std::max(pl->maxSupRows, 1) * T);
        add("gRs", pl->o_gRs, I * T);
        add("gExtra", pl->o_gExtra, I);
    }
    add("END", pl->total, 1);
    for (auto &t : r) store.push_back(std::get<0>(t));
    for (int i = 0; i < (int)r.size() && i < cap; ++i) {
        names[i] = store[i].c_str();
        beg[i] = std::get<1>(r[i]);
        end[i] = std::get<2>(r[i]);
    }
    return (int)r.size();
}

extern "C" int stair_plan_zero_grads(stair_plan *pl, void *workspace, stair_stream stream) {
    STAIR_CHECK(pl && workspace, "null argument");
    STAIR_CHECK(pl->train, "plan was not built with STAIR_PLAN_TRAIN");
    float *ws = static_cast<float *>(workspace);
    if (int rcz_ = launch_zero(ws + pl->o_zero_beg, (pl->o_zero_end - pl->o_zero_beg) * sizeof(float), static_cast<hipStream_t>(stream))) return rcz_;
    return 0;
}

// Diagnostics for the hipGraph question of DESIGN.md section 2: enqueue, on `stream` (which may be capturing), the reset + `launches`
// queue kernels exactly as stair_plan_run lays them out -- reset_mode 0: hipMemsetAsync of 256 bytes (round 3's reset), 1: the
// zeroing kernel, 2: no reset at all (self-resetting queue only) -- each launch on its own head (per_launch_heads = 1, round 3) or
// all on one (0).  seen [launches][grid][2] receives (first ticket, tiles taken) of every workgroup.  No tile work: safe to replay.
extern "C" int stair_debug_queue_probe(void *words, uint32_t *seen, int32_t launches, int32_t grid, int32_t total, int32_t reset_mode,
                                       int32_t per_launch_heads, int32_t self_reset, stair_stream stream) {
    STAIR_CHECK(words && seen && launches >= 1 && launches <= 24 && grid >= 1 && total >= 0, "bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t *w = static_cast<uint32_t *>(words);
    if (reset_mode == 0) STAIR_HIP(hipMemsetAsync(w, 0, 64 * sizeof(float), s));
    else if (reset_mode == 1) { if (int rcz_ = launch_zero(w, 64 * sizeof(uint32_t), s)) return rcz_; }
    for (int l = 0; l < launches; ++l)
        hipLaunchKernelGGL(queue_probe_kernel, dim3(grid), dim3(64), 0, s, w + 16 + (per_launch_heads ? 2 * l : 0), seen + (int64_t)l * grid * 2,
                           total, self_reset);
    STAIR_LAUNCH_CHECK();
    return 0;
}

extern "C" int stair_set_tile_queue(int32_t on) { g_tile_queue = on; return 0; }

extern "C" int stair_grad_shadows_begin(stair_ctx *ctx, stair_stream stream) {
    STAIR_CHECK(ctx, "null context");
    if (!det_enabled()) return 0;                // STAIR_DETERMINISTIC=0: float atomics everywhere
    if (int rc = det_begin(ctx, static_cast<hipStream_t>(stream))) return rc;
    g_det.early_ctx = ctx;
    return 0;
}

extern "C" int stair_ctx_set_option(stair_ctx *ctx, int32_t option, int32_t value) {
    STAIR_CHECK(ctx, "null context");
    STAIR_CHECK(option >= 0 && option < STAIR_OPT_COUNT, "unknown option");
    STAIR_CHECK(option != STAIR_OPT_MATMUL_MODE || value < 0 || value == STAIR_MATMUL_F32 || value == STAIR_MATMUL_BF16X3 || value == STAIR_MATMUL_BF16,
                "matmul mode must be STAIR_MATMUL_F32, STAIR_MATMUL_BF16X3 or STAIR_MATMUL_BF16");
    ctx->policy.v[option] = value < 0 ? -1 : value;
    return 0;
}
extern "C" int stair_ctx_get_option(const stair_ctx *ctx, int32_t option, int32_t *value) {
    STAIR_CHECK(ctx && value, "null argument");
    STAIR_CHECK(option >= 0 && option < STAIR_OPT_COUNT, "unknown option");
    *value = ctx->policy.v[option];
    return 0;
}

// reset_mode 0 of the probe on its own: a hipMemsetAsync of `bytes` zero bytes on `stream` (which may be capturing) -- which sizes
// of a captured memset survive a replay?  mode 1: the library's zero-fill kernel instead.
extern "C" int stair_debug_memset(void *ptr, int64_t bytes, int32_t mode, stair_stream stream) {
    STAIR_CHECK(ptr && bytes > 0, "bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == 0) STAIR_HIP(hipMemsetAsync(ptr, 0, (size_t)bytes, s));
    else if (int rcz_ = launch_zero(ptr, bytes, s)) return rcz_;
    return 0;
}
