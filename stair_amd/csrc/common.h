// Shared host/device helpers for libstair_hip.so (gfx950 only).
#pragma once
#include <cstdlib>
#include <cstdio>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/stair_hip.h"

namespace stair {

void set_error(const std::string &msg);
#define STAIR_FAIL(msg)                                                                     \
    do {                                                                                    \
        ::stair::set_error(std::string(__func__) + ": " + (msg));                           \
        return 1;                                                                           \
    } while (0)
#define STAIR_CHECK(cond, msg)                                                              \
    do {                                                                                    \
        if (!(cond)) STAIR_FAIL(msg);                                                       \
    } while (0)
#define STAIR_HIP(call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) STAIR_FAIL(std::string(#call) + " -> " + hipGetErrorString(e_)); \
    } while (0)
#define STAIR_LAUNCH_CHECK()                                                                \
    do {                                                                                    \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess) STAIR_FAIL(std::string("kernel launch -> ") + hipGetErrorString(e_)); \
    } while (0)

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
// fast forms for the LSTM cell (absolute error ~1e-7, far inside the 1e-4 logit budget)
// (v_exp_f32 and v_rcp_f32 are 1-ulp instructions; __frcp_rn would expand to a 12-instruction IEEE division)
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// ---- run-to-run reproducible weight gradients -----------------------------------------------------------------------------------
// Inside stair_plan_backward every kernel that ADDS into a weight-gradient tensor from several workgroups (slab partials of the
// TN products, the bias sums riding along, LayerNorm / Relate / Conv1d / 1-output-Linear gradients) adds into a 64-bit fixed-point
// shadow of that tensor instead: integer addition is associative, so the sum does not depend on the order in which the atomics
// land.  Scale 2^44: a contribution is kept to 5.7e-14 absolute (the sums of a step stay far below 2^19); the shadows are added to
// the fp32 gradients by one launch at the end of the pass (csrc/plan.hip det_*).  det_shadow(g): the shadow of the gradient
// element g points at, or NULL outside a backward pass / for a pointer that is not inside a bound gradient tensor / STAIR_DETERMINISTIC=0.
long long *det_shadow(const float *gptr);
constexpr float kFxScale = 17592186044416.0f;            // 2^44
__device__ __forceinline__ void fx_add(long long *p, float v) {
    atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__float2ll_rn(v * kFxScale));
}
// add into the shadow when there is one, else the plain float atomic
__device__ __forceinline__ void grad_add(float *g, long long *g64, int64_t off, float v) {
    if (g64) fx_add(g64 + off, v);
    else unsafeAtomicAdd(g + off, v);
}

// ---- algorithmic-byte accounting of the HBM-bound row kernels (measurement aid, off unless stair_acct_enable(1)) --------
// Each row-kernel launcher reports the bytes its launch MUST move (inputs read once + outputs written once, the figure of
// SURVEY.md section 8d); tools/row_kernels.py divides the per-kernel sums by the rocprofv3 kernel durations of the same run.
void acct_add(const char *kernel, int64_t bytes, int64_t flops = 0);
extern bool g_acct_on;
#define STAIR_ACCT(name, bytes) do { if (::stair::g_acct_on) ::stair::acct_add(name, (int64_t)(bytes)); } while (0)
// the MFMA-bound kernels: which variant a launcher selected, with its algorithmic flops (2MNK) and bytes (operands once + result once)
#define STAIR_ACCT_MFMA(name, bytes, flops) do { if (::stair::g_acct_on) ::stair::acct_add(name, (int64_t)(bytes), (int64_t)(flops)); } while (0)

// ---- per-context options (stair_ctx_set_option).  The process-wide setters (stair_set_matmul_mode, stair_set_tile_mlp,
// stair_set_tile_queue, stair_set_tn_slab_min_rows) and the environment give the DEFAULTS; a context may override each of them, and
// its values are in force on the calling thread for the duration of stair_plan_run / stair_plan_backward on that context (PolicyScope):
// two contexts of one process -- two GPUs driven from two threads, or two configurations on one GPU -- do not see each other's settings.
struct Policy { int v[STAIR_OPT_COUNT]; Policy() { for (int &x : v) x = -1; } };       // -1: inherit the process default
extern thread_local const Policy *tl_policy;
inline int policy_or(int opt, int dflt) { return tl_policy && tl_policy->v[opt] >= 0 ? tl_policy->v[opt] : dflt; }
struct PolicyScope {
    const Policy *prev;
    explicit PolicyScope(const Policy *p) : prev(tl_policy) { tl_policy = p; }
    ~PolicyScope() { tl_policy = prev; }
};

// ---- internal launchers shared between the C ABI and the plan runner ----------------------
int launch_gemm(const stair_gemm_args &a, hipStream_t s);
int launch_lstm(const stair_lstm_args &a, hipStream_t s);
int launch_lstm_bwd(const stair_lstm_bwd_args &a, hipStream_t s);
// STAIR_GEMM_TRACE=1: every GEMM launcher prints its shape to stderr (tools/gemm_shapes.py joins the lines with a rocprofv3 kernel trace)
inline bool gemm_trace_on() { static const bool on = [] { const char *e = getenv("STAIR_GEMM_TRACE"); return e && e[0] == '1'; }(); return on; }

struct TransposeBatch {          // up to 32 matrices transposed by one launch (launch_transpose_many)
    const float *in[32]; float *out[32];
    int rows[32], cols[32], first_tile[32];
    int count;
};
int launch_transpose_many(const TransposeBatch &tb, int total_tiles, hipStream_t s);
int launch_gemm_tn_tr(const stair_gemm_tn_args &a, hipStream_t s);      // -1: not this kernel's shape
int launch_gemm_tn_tr_slabs(const stair_gemm_tn_args &a, float *scratch, hipStream_t s);    // slab partials into scratch [8 N K] + queued reduction
int tn_x3tr_queue(const float *P, float *dst, int nslab, int count4, hipStream_t s);
// csrc/gemm_tn_x3tr.hip: slab partials of a weight-gradient product into scratch; tn_x3tr_flush adds every queued product's slabs
// to its destination in fixed order (one launch), on the same stream and thread
bool tn_x3tr_takes(const stair_gemm_tn_args &a);
int tn_x3tr_slabs(int64_t M);
int64_t tn_x3tr_scratch_floats(int64_t M, int64_t N, int64_t K);
int launch_gemm_tn_x3tr(const stair_gemm_tn_args &a, float *scratch, hipStream_t s);
int tn_x3tr_flush(hipStream_t s);
void tn_x3tr_discard();          // forget queued sums (a pass that failed half-way must not leak them into the next one)
int launch_lstm_rec_coop(const stair_lstm_args &a, hipStream_t s);
bool lstm_coop_usable(int Hh);
int64_t lstm_coop_ws_bytes(int n);
int launch_lstm_bwd_coop(const stair_lstm_bwd_args &a, hipStream_t s);
int launch_lstm_rec_coop_pair(const stair_lstm_args &a, const stair_lstm_args &b, hipStream_t s);          // -1: not applicable
int launch_lstm_bwd_coop_pair(const stair_lstm_bwd_args &a, const stair_lstm_bwd_args &b, hipStream_t s);  // -1: not applicable
// Dropout mask bits: a counter-based hash of (seed, site, element / 4) gives 64 mixed bits, 16 per element of an aligned group of four
// consecutive elements; an element is DROPPED when its 16 bits are below p * 2^16 (csrc/rowops.hip dropout_rows_kernel, and the
// epilogue of the fused tile operator, csrc/tile_mlp.hip, whose lanes hold such groups: one hash per four elements)
__device__ __forceinline__ uint64_t drop_hash4(uint64_t seed, uint32_t site, uint64_t e4) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (e4 + 1) + ((uint64_t)site << 40);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ bool drop_keep(uint64_t bits4, int i, uint32_t thresh16) { return (uint32_t)((bits4 >> (16 * i)) & 0xffffu) >= thresh16; }
int launch_lstm_project(const stair_lstm_args &a, hipStream_t s);
int launch_lstm_recur(const stair_lstm_args &a, hipStream_t s);
int launch_lstm_zero_tail(const stair_lstm_args &a, hipStream_t s);
int launch_lstm_bwd_recur(const stair_lstm_bwd_args &a, hipStream_t s);
int launch_lstm_bwd_weights(const stair_lstm_bwd_args &a, hipStream_t s);
bool lstm_bwd_takes_coop(const stair_lstm_bwd_args &a);
int64_t lstm_coop_bwd_ws_bytes(int n);
int launch_gemm_bf16x3(const stair_gemm_args &a, hipStream_t s);
int launch_gemm_tn_bf16x3(const stair_gemm_tn_args &a, hipStream_t s);
int matmul_mode();
int launch_split_planes(const float *x, void *hi, void *lo, int64_t n, hipStream_t s);
int launch_split_planes_tiled(const float *x, void *hi, void *lo, int rows, int cols, hipStream_t s, int row_off = 0, int total_rows = 0);
int launch_split_planes_pad(const float *x, int64_t ldx, void *hi, void *lo, int rows, int cols, int Kp, bool tiled, hipStream_t s, int row_off = 0, int total_rows = 0);
int launch_gemm_planes(const stair_gemm_planes_args &a, hipStream_t s);
bool gemm_planes_supported(int64_t M, int N, int K);
int launch_gemm_tn(const stair_gemm_tn_args &a, hipStream_t s);
int launch_gemm_tn_batch(const stair_gemm_tn_args *a, int n, hipStream_t s);
int launch_tile_mlp(const stair_tile_mlp_args &a, hipStream_t s);          // csrc/tile_mlp.hip
int launch_tile_mlp_batch(const stair_tile_mlp_args *args, int n, unsigned *counter, hipStream_t s);   // <= 8 buckets, one launch
bool tile_mlp_usable(int H, int T);
// csrc/vec_group.hip: the row-wise Linear layers of a program level as one launch
using VgProblem = stair_vec_problem;
constexpr int VG_FWD = STAIR_VEC_FWD, VG_ADJ = STAIR_VEC_ADJ;
constexpr int VG_IN_A = STAIR_VEC_IN_A, VG_IN_CAT2 = STAIR_VEC_IN_CAT2, VG_IN_XOR = STAIR_VEC_IN_XOR, VG_IN_EXISTS = STAIR_VEC_IN_EXISTS,
              VG_IN_MASK = STAIR_VEC_IN_MASK;
int launch_vec_group(const VgProblem *probs, int n, hipStream_t s);
bool vec_group_usable(int H);
int launch_pack_wfrag_many(const float *const *W, void *const *out, int count, int N, int K, hipStream_t s, bool transpose = false,
                           const int *ld = nullptr);     // ld: row stride per matrix (default: K, or N when transposed)
int launch_colsum(const float *A, int64_t lda, float *out, int M, int N, hipStream_t s, float *out2 = nullptr);
int launch_transpose(const float *in, float *out, int rows, int cols, hipStream_t s);
// bytes % 4 == 0 zero bytes at a 4-byte aligned address, as a kernel (never a memset node of a captured graph; csrc/rowops.hip)
int launch_zero(void *ptr, int64_t bytes, hipStream_t s);

}  // namespace stair
