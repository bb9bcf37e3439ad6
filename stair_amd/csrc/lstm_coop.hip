// Cooperative LSTM recurrence: hidden units split across a group of co-resident workgroups, W_hh resident in REGISTERS.
//
// Round 1's recurrent kernel (csrc/lstm.hip, lstm_rec_x3_kernel) gives one workgroup 16 sequences and ALL 4*Hh gate rows,
// so every workgroup re-streams the whole 1 MB bf16 hi/lo image of W_hh from L2 on each of the serial steps
// (256 WG x 1 MB x 64 steps = 16.8 GB of L2 reads per launch against 2 MB of weights) and a batch of 8 questions still
// pays 64 x 14 us.  Here (nn.LSTM of /root/reference/video_nmn/module_net.py:39-47,151-163, Hh = 256):
//
//   * a GROUP of 4 workgroups (4 CUs, placed on one XCD) owns up to 64 sequences of one direction for all steps;
//     workgroup j owns hidden units [64 j, 64 j + 64), i.e. 256 of the 1024 gate rows;
//   * wave w of a workgroup owns 8 units = one 32-row MFMA tile (row 8 q + u <-> gate q, unit u), and keeps those rows of
//     W_hh as bf16 hi + lo fragments in 128 VGPRs for the whole launch: nothing of W_hh is read again after the prologue;
//   * gates^T[32 rows, 32 sequences] = W_tile[32, 256] x h^T[256, 32] on v_mfma_f32_32x32x16_bf16 (hi*hi + lo*hi + hi*lo,
//     fp32 accumulate; 48 MFMAs per wave, sequence tile and step); the four gates of a unit land in ONE lane, so the cell
//     update is lane-local and c never leaves registers;
//   * the new h (bf16 hi + lo) is exchanged between the 4 workgroups through a double-buffered global slab with the
//     write-through (sc1) store / flag / sc1 load protocol of cdna_hip_programming.md Guideline 16 (R1, table row 1 of
//     MI355X_MICROARCH.md "Valid forms"): every storing wave drains vmcnt, the workgroup barrier, ONE lane stores the
//     epoch flag; a consumer polls the four flags relaxed, then every load of the slab is an sc1 load.  Two sequence
//     tiles (A, B) are interleaved so that the hand-off of one tile hides behind the MFMAs of the other.
// Every spin is bounded; on a timeout an error word is set, the launch still drains, and the wave that gave up writes NaN
// from then on (hidden states forward, gate gradients backward), so the failure reaches the logits / the gradients instead of
// passing as a plausible result.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace stair {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using v4f = __attribute__((ext_vector_type(4))) float;
using v4u = __attribute__((ext_vector_type(4))) unsigned;
using v2u = __attribute__((ext_vector_type(2))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int CH = 256;                 // Hh this kernel is built for
constexpr int CP = 4;                   // workgroups per group (64 units each)
constexpr int TILE_BYTES = 2 * 32 * CH * 2;      // one sequence tile of h: 2 planes x 32 sequences x 256 bf16 = 32 KB
constexpr int FLAG_STRIDE = 16;         // uint32 words between flags (64 B: no two flags share a line)
constexpr unsigned SPIN_LIMIT = 1u << 20;     // ~1 s of polling

struct CoopParams {
    float *xproj;             // [rows, 8 Hh]: gate pre-activations from the input projection; training: overwritten with activated gates
    const float *w_hh[2];     // [4 Hh, Hh] per direction
    const int32_t *seq_off;   // [n + 1]
    const int32_t *seq_len;   // optional [n]: rows of sequence s when the storage is padded (else off[s+1] - off[s])
    float *out; int64_t ldo;  // [rows, ldo]
    float *h_n;               // [n, 2 Hh]
    float *cbuf;              // [rows, 2 Hh] or null
    char *xh;                 // exchange slabs [2 parity][groups][NT tiles][TILE_BYTES]
    unsigned *flags;          // [groups][NT tiles][CP] epochs, FLAG_STRIDE words apart; zeroed before the launch
    unsigned *err;            // one word: set to 1 when a spin timed out (inside coop_ws, cleared before every launch)
    unsigned *status;         // optional caller-owned STICKY word (stair_lstm_args.status): also set on a timeout, never cleared here
    int n, gpd;               // sequences, groups per direction
};

__device__ __forceinline__ int lds_unit(int seq, int chunk) { return (seq * 32 + (chunk ^ (seq & 15))) * 16; }

}  // namespace

// NT: sequence tiles (of 32) a group carries through the steps together, 1..3.  The hand-off of a tile is spread over the
// computes of the other tiles: h(t) of tile x is PUBLISHED (sc1 stores) at the end of its own compute, FLAGGED in the
// middle of the next compute (by then the stores have drained: the wait costs nothing), its slab loads are ISSUED at the
// start of the compute after that and WRITTEN to LDS at its end -- with NT = 3 that is exactly when tile x is next, so no
// phase waits for memory; NT = 2 and 1 expose part of the latency (small batches).
template <bool TRAIN, int NT, bool PROF>
__device__ __forceinline__ void lstm_rec_coop_body(const CoopParams &p, const int b, char *hl) {
    // hl: [NT tiles][2 planes][32 seq][256] bf16, swizzled
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // group placement: blocks with equal b % 8 share an XCD (speed only); the 4 workgroups of a group do
    const int q8 = b >> 3;
    const int j = q8 & 3, g = (b & 7) + 8 * (q8 >> 2);
    const int G = 2 * p.gpd;
    if (g >= G) return;                                            // whole groups only: nobody waits for this workgroup
    const int dir = g / p.gpd, gl = g - dir * p.gpd;
    constexpr int Hh = CH;
    constexpr int SPG = 32 * NT;                                   // sequences per group and chunk
    const int U0 = 64 * j + 8 * wave + 4 * hh;                     // this lane's 4 hidden units (accumulator rows 4 q + i)
    const int64_t ldx = 8 * (int64_t)Hh;
    const int xcol = dir * 4 * Hh + U0;

    // ---- W_hh fragments: tile row rho = lane & 31 <-> gate rho >> 3, unit 64 j + 8 wave + (rho & 7) ----
    bf16x8 whi[16], wlo[16];
    {
        const float *wrow = p.w_hh[dir] + (int64_t)((r >> 3) * Hh + 64 * j + 8 * wave + (r & 7)) * Hh + 8 * hh;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const v4f a = *reinterpret_cast<const v4f *>(wrow + 16 * s), c = *reinterpret_cast<const v4f *>(wrow + 16 * s + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                whi[s][e] = (__bf16)a[e]; wlo[s][e] = (__bf16)(a[e] - (float)whi[s][e]);
                whi[s][4 + e] = (__bf16)c[e]; wlo[s][4 + e] = (__bf16)(c[e] - (float)whi[s][4 + e]);
            }
        }
    }
    __amdgpu_buffer_rsrc_t xh_rs = __builtin_amdgcn_make_buffer_rsrc(p.xh, 0, 0x7fffffff, 0x00020000);
    // slab of (parity, tile): [2][G][NT][TILE_BYTES]
    auto slab_off = [&](unsigned ep, int x) { return (int)(((int64_t)((ep & 1) * G + g) * NT + x) * TILE_BYTES); };
    auto flag_of = [&](int x, int wg) { return (gu32 *)(p.flags + ((g * NT + x) * CP + wg) * FLAG_STRIDE); };

    // diagnostic build only (PROF): cycles per phase of wave 0 of workgroup 0, summed over the launch, into err[8..]
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    auto stamp = [&](int slot) {
        if (PROF) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long tnow;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (slot >= 0) prof[slot] += tnow - tlast;
            tlast = tnow;
        }
    };
    unsigned ebase = 0;                                            // epochs published before this chunk (same on all 4 workgroups)
    bool dead = false;                                             // a spin timed out: stop waiting, drain
    for (int chunk = gl; chunk * SPG < p.n; chunk += p.gpd) {
        const int s_base = chunk * SPG;
        int off[NT], len[NT], lmax = 0;
#pragma unroll
        for (int x = 0; x < NT; ++x) {
            const int s = s_base + 32 * x + r;
            off[x] = 0; len[x] = 0;
            if (s < p.n) { off[x] = p.seq_off[s]; len[x] = p.seq_len ? p.seq_len[s] : p.seq_off[s + 1] - off[x]; }
        }
        for (int s = s_base; s < min(s_base + SPG, p.n); ++s) lmax = max(lmax, p.seq_len ? p.seq_len[s] : p.seq_off[s + 1] - p.seq_off[s]);
        for (int i = tid; i < NT * TILE_BYTES / 16 + 1; i += 512) reinterpret_cast<v4u *>(hl)[i] = v4u{0, 0, 0, 0};     // h(-1) = 0, arrival counters = 0
        float creg[NT][4];
#pragma unroll
        for (int x = 0; x < NT; ++x)
#pragma unroll
            for (int i = 0; i < 4; ++i) creg[x][i] = 0.0f;
        __syncthreads();

        v4u inflight[4];                                           // slab of the tile being fetched (16 B x 4 per thread)
        // Every wave checks the 4 flags of tile x for epoch ep itself (lanes 0..3; `seen` is a value loaded earlier, behind
        // the previous compute, so in the steady state nothing is waited for), then issues the sc1 loads of its share of
        // the slab: no workgroup barrier between the poll and the loads (MI355X_MICROARCH.md "Valid forms": the wave that
        // polled loads after its poll has matched).
        auto fetch_issue = [&](const int x, const unsigned ep, unsigned seen) {
            stamp(-1);
            if (!dead) {
                const gu32 *fl = flag_of(x, lane & 3);
                unsigned spins = 0;
                while (!__all((int)(seen - ep) >= 0)) {
                    if (++spins > SPIN_LIMIT) {
                        if (lane == 0) {
                            __hip_atomic_store((gu32 *)p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (p.status) __hip_atomic_store((gu32 *)p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        dead = true;                               // this wave polls no more; the results are void, the launch drains
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    seen = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");              // no instruction: keeps the loads below the poll
            const int so = slab_off(ep, x);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                inflight[i] = __builtin_amdgcn_raw_buffer_load_b128(xh_rs, so + (tid + 512 * i) * 16, 0, 16);     // aux 16 = sc1
            stamp(7);                                              // 7: poll + slab load issue
        };
        auto fetch_land = [&](const int x) {
            stamp(-1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int u = tid + 512 * i;                       // 16-byte unit: plane = u >> 10, seq = (u >> 5) & 31, chunk = u & 31
                *reinterpret_cast<v4u *>(hl + x * TILE_BYTES + (u >> 10) * (32 * CH * 2) + lds_unit((u >> 5) & 31, u & 31)) = inflight[i];
            }
            stamp(5);                                              // 5: slab wait + ds_write
            __syncthreads();
            stamp(6);                                              // 6: barrier
        };
        // one tile-step: gates = W h(t-1) + xproj(t); [flag the tile published one compute ago]; cell update; publish h(t)
        unsigned peek = 0;                                         // flag value of the tile fetched next, loaded behind the cell update
        auto compute = [&](const int x, const int t, const int flag_x, const unsigned flag_ep, const int peek_x) {
            stamp(-1);
            const bool active = t < len[x];
            const int tt = dir == 0 ? t : len[x] - 1 - t;
            const int64_t row = off[x] + (active ? tt : 0);
            // xproj of this step: 4 gates x 4 units, issued before the MFMA chain that hides their latency
            v4f xp[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                xp[q] = active ? *(const __attribute__((address_space(1))) v4f *)(p.xproj + row * ldx + xcol + q * Hh)
                               : v4f{0.f, 0.f, 0.f, 0.f};
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
            const char *hx = hl + x * TILE_BYTES;
            // h(t-1) of my own 4 units (bf16 hi / lo), re-published unchanged when the sequence has ended
            const int own_ = lds_unit(r, 8 * j + wave) + 8 * hh;
            v2u ph_ = *reinterpret_cast<const v2u *>(hx + own_), pl_ = *reinterpret_cast<const v2u *>(hx + 32 * CH * 2 + own_);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(hx + lds_unit(r, 2 * s + hh));
                const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(hx + 32 * CH * 2 + lds_unit(r, 2 * s + hh));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[s], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[s], bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[s], bh, acc, 0, 0, 0);
            }
            stamp(0);                                              // 0: xproj issue + MFMA chain
            // every older store of this wave has drained by now (issued one MFMA chain ago), the xproj loads are due anyway
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(1);                                              // 1: vmcnt(0)
            // No workgroup barrier here: each wave, after ITS drain, adds to a counter in LDS and the wave whose add comes
            // last stores the flag (MI355X_MICROARCH.md "Valid forms", condition 3).  A wave that is through with its MFMAs
            // goes on to its cell update (VALU) while its SIMD partner still multiplies.
            if (flag_x >= 0 && lane == 0) {
                unsigned *cnt = reinterpret_cast<unsigned *>(hl + NT * TILE_BYTES) + flag_x;
                const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (old == 7u) {
                    __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(flag_of(flag_x, j), flag_ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            stamp(2);                                              // 2: arrival counter / flag
            if (peek_x >= 0) peek = __hip_atomic_load(flag_of(peek_x, lane & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (active) {
                v4f si, sf, tg, so, cn, hv;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    si[i] = sigmoid_fast(acc[i] + xp[0][i]);
                    sf[i] = sigmoid_fast(acc[4 + i] + xp[1][i]);
                    tg[i] = tanh_fast(acc[8 + i] + xp[2][i]);
                    so[i] = sigmoid_fast(acc[12 + i] + xp[3][i]);
                    cn[i] = sf[i] * creg[x][i] + si[i] * tg[i];
                    hv[i] = so[i] * tanh_fast(cn[i]);
                    creg[x][i] = cn[i];
                }
                if (dead) hv = v4f{NAN, NAN, NAN, NAN};           // a hand-off timed out: fail LOUDLY -- NaN through out / h_n into the logits
                bf16x4 ph, pl;
#pragma unroll
                for (int i = 0; i < 4; ++i) { ph[i] = (__bf16)hv[i]; pl[i] = (__bf16)(hv[i] - (float)ph[i]); }
                ph_ = __builtin_bit_cast(v2u, ph); pl_ = __builtin_bit_cast(v2u, pl);
                *(__attribute__((address_space(1))) v4f *)(p.out + row * p.ldo + dir * Hh + U0) = hv;
                if (t == len[x] - 1)                               // h_n = h after the last step of the sequence
                    *(__attribute__((address_space(1))) v4f *)(p.h_n + (int64_t)(s_base + 32 * x + r) * 2 * Hh + dir * Hh + U0) = hv;
                if (TRAIN) {        // activated gates replace this row's xproj (already consumed), c saved
                    float *gs = p.xproj + row * ldx + xcol;
                    *(__attribute__((address_space(1))) v4f *)(gs) = si;
                    *(__attribute__((address_space(1))) v4f *)(gs + Hh) = sf;
                    *(__attribute__((address_space(1))) v4f *)(gs + 2 * Hh) = tg;
                    *(__attribute__((address_space(1))) v4f *)(gs + 3 * Hh) = so;
                    *(__attribute__((address_space(1))) v4f *)(p.cbuf + row * 2 * Hh + dir * Hh + U0) = cn;
                }
            }
            stamp(3);                                              // 3: flag, peek, cell update, out stores
            // publish (finished sequences re-publish their last h): 8 bytes per plane, write-through
            char *slab = p.xh + slab_off(ebase + t + 1, x);
            const int so_ = r * (CH * 2) + U0 * 2;
            __hip_atomic_store((gu64 *)(slab + so_), ((unsigned long long)ph_[1] << 32) | ph_[0], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((gu64 *)(slab + 32 * CH * 2 + so_), ((unsigned long long)pl_[1] << 32) | pl_[0], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            stamp(4);                                              // 4: publish
        };
        // publish -> flag lags one compute; flag -> loads issued lags one more; landed one compute later.
        // Schedule of compute k = t * NT + x (tile x, step t): before it, issue the fetch of tile (x + 1) % NT [its epoch:
        // the last one it published]; inside it, flag tile (x - 1) % NT; after it, land the fetch.
        const int total = lmax * NT;
        for (int t = 0; t < lmax; ++t) {
#pragma unroll
            for (int x = 0; x < NT; ++x) {                         // unrolled: x indexes register arrays (a runtime index would put them in scratch)
                const int k = t * NT + x;
                if (NT > 1) {
                    // compute k + 1 (tile xf) consumes what compute k + 1 - NT published (step tf)
                    const int xf = (x + 1) % NT;
                    const int tf = x + 1 == NT ? t : t - 1;
                    const bool do_fetch = tf >= 0 && k + 1 < total;
                    // flagged inside this compute: the tile computed just before (k - 1)
                    const int xq = (x + NT - 1) % NT;
                    const int tq = x == 0 ? t - 1 : t;
                    if (NT >= 3) {
                        // tile xf was flagged one compute ago: issue its loads now, land them behind this compute.  The flags
                        // of the tile fetched before the NEXT compute (xf + 1, flagged inside this one) are peeked at here.
                        if (do_fetch) fetch_issue(xf, ebase + tf + 1, peek);
                        compute(x, t, tq >= 0 ? xq : -1, ebase + tq + 1, (xf + 1) % NT);
                        if (do_fetch) fetch_land(xf);
                    } else {
                        // two tiles: tile xf is flagged inside THIS compute, so its fetch follows it (latency partly exposed)
                        compute(x, t, tq >= 0 ? xq : -1, ebase + tq + 1, -1);
                        if (do_fetch) {
                            fetch_issue(xf, ebase + tf + 1, 0u);
                            fetch_land(xf);
                        }
                    }
                } else {
                    compute(0, t, -1, 0, -1);
                    // one tile: nothing to hide behind; drain, flag, fetch right away
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store(flag_of(0, j), ebase + t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (t + 1 < lmax) { fetch_issue(0, ebase + t + 1, 0u); fetch_land(0); }
                }
            }
        }
        if (NT > 1) {
            // the last compute's tile was never flagged; later chunks poll absolute epochs, so close the books
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0 && total > 0)
                __hip_atomic_store(flag_of((total - 1) % NT, j), ebase + (total - 1) / NT + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // a further chunk's first publish (epoch ebase' + 1) must not land in the slab a slower workgroup of the group is still
        // reading -- that of the last epoch consumed, ebase + lmax - 1 -- so one epoch is skipped: the parities then differ
        ebase += lmax + 1;
        __syncthreads();                                           // LDS is re-zeroed for the next chunk
    }
    if (PROF && b == 0 && tid == 0)
        for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long *>(p.err + 8)[i] = prof[i];
}

template <bool TRAIN, int NT, bool PROF = false>
__global__ __launch_bounds__(512, 1) void lstm_rec_coop_kernel(CoopParams p) {
    extern __shared__ __attribute__((aligned(16))) char hl_dyn[];
    lstm_rec_coop_body<TRAIN, NT, PROF>(p, (int)blockIdx.x, hl_dyn);
}

// Two recurrences in ONE launch (the video and the text encoder of a batch small enough that both sets of groups are
// co-resident: nb_a + nb_b workgroups <= CUs): blocks [0, nb_a) run `a`, the rest run `b`.  At 128 questions the two
// encoders occupy 32 workgroups each and their per-step latency (4 exchange hops) is what a recurrence costs, so running
// them one after the other left seven eighths of the chip idle twice.
template <bool TRAIN>
__global__ __launch_bounds__(512, 1) void lstm_rec_coop_pair_kernel(CoopParams a, CoopParams b, int nb_a) {
    extern __shared__ __attribute__((aligned(16))) char hl_dyn[];
    if ((int)blockIdx.x < nb_a) lstm_rec_coop_body<TRAIN, 1, false>(a, (int)blockIdx.x, hl_dyn);
    else lstm_rec_coop_body<TRAIN, 1, false>(b, (int)blockIdx.x - nb_a, hl_dyn);
}

// tiles per group: 3 once the batch fills most of the chip that way, else as many as it takes to use all groups
// Per-DEVICE launch state (a process may drive several GPUs): CU count, "dynamic LDS attribute set" and the occupancy the
// runtime reports for each kernel variant.  STAIR_LSTM_COOP_MAX_BLOCKS caps the number of workgroups a cooperative launch may
// use (tests: forces the smaller geometries and the fallback to the one-workgroup kernels).
constexpr int MAX_DEVICES = 64;
struct CoopDevice {
    int cus = 0;
    bool rec_attrs = false, bwd_attrs = false;
    int occ[16] = {};          // max co-resident workgroups of variant v on this device (0 = not asked yet)
};
static CoopDevice &coop_device() {
    static CoopDevice devs[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
    CoopDevice &d = devs[dev];
    if (d.cus == 0) {
        int v = 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        d.cus = v;
    }
    return d;
}
static int g_coop_cap = -1;          // stair_lstm_coop_limit; -1 = STAIR_LSTM_COOP_MAX_BLOCKS or no cap
static int coop_block_cap() {
    static const int env_cap = [] { const char *e = getenv("STAIR_LSTM_COOP_MAX_BLOCKS"); return e ? std::max(0, atoi(e)) : 1 << 30; }();
    return std::min(g_coop_cap >= 0 ? g_coop_cap : env_cap, coop_device().cus);
}
static int coop_cu_count() { return coop_block_cap(); }

// All workgroups of a cooperative launch wait for each other, so the whole grid must be resident at once.  The contract is
// checked against the runtime's own occupancy figure for THIS kernel on THIS device (registers, LDS, workgroup size), not
// assumed from __launch_bounds__: false -> the caller falls back to the one-workgroup-per-16-sequences kernels.
template <typename K>
static bool coop_fits(K kernel, int variant, int blocks, size_t shmem) {
    CoopDevice &d = coop_device();
    if (d.occ[variant] == 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kernel), 512, shmem) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            per_cu = 0;
        }
        d.occ[variant] = per_cu > 0 ? per_cu * d.cus : -1;
    }
    return d.occ[variant] > 0 && blocks <= std::min(d.occ[variant], coop_block_cap());
}

static void coop_geometry(int n, int &nt, int &gpd) {
    static const int force = [] { const char *e = getenv("STAIR_LSTM_COOP_TILES"); return e ? atoi(e) : 0; }();
    // measured (profiles/r02_b_lstm_coop.txt): one tile per group while every tile gets a group of its own (n <= 1024), then two;
    // three tiles need more registers than a wave has (spills) and lose
    nt = force >= 1 && force <= 3 ? force : (n > 32 * 32 ? 2 : 1);
    // every workgroup of the launch must be resident at once (they wait for each other): groups per direction <= CUs / 8
    // (2 directions x 4 workgroups), i.e. 32 on the 256-CU part, fewer on a partitioned device; the cap keeps whole XCD rounds
    const int cap = std::max(1, std::min(32, coop_cu_count() / 8));
    gpd = std::max(1, std::min((n + 32 * nt - 1) / (32 * nt), cap));
}

// workgroups of a launch with `gpd` groups per direction: group g = (b & 7) + 8 * (b >> 5); workgroup (b >> 3) & 3
static int coop_blocks(int gpd) { return 32 * ((2 * gpd + 7) / 8); }

int64_t lstm_coop_ws_bytes(int n) {
    // sized for the largest geometry (3 tiles, 32 groups per direction): the choice above may change with n
    const int G = 64;
    (void)n;
    return (int64_t)2 * G * 3 * TILE_BYTES + (int64_t)(G * 3 * CP * FLAG_STRIDE + 64) * 4;
}

bool lstm_coop_usable(int Hh) {
    static const bool on = [] { const char *e = getenv("STAIR_LSTM_COOP"); return !(e && e[0] == '0'); }();
    // even the smallest launch (one group per direction) needs 32 co-resident workgroups
    return on && Hh == CH && matmul_mode() != STAIR_MATMUL_F32 && coop_block_cap() >= 32;
}

// fills the kernel parameters of one recurrence and enqueues its per-launch memsets (flags, h_n)
static int coop_prepare(const stair_lstm_args &a, hipStream_t s, CoopParams &p, int &nt, int &blocks) {
    STAIR_CHECK(a.Hh == CH, "cooperative recurrence is built for Hh = 256");
    STAIR_CHECK(a.coop_ws && a.coop_ws_bytes >= lstm_coop_ws_bytes(a.n), "coop_ws missing or too small");
    STAIR_CHECK((reinterpret_cast<uintptr_t>(a.coop_ws) & 255) == 0, "coop_ws must be 256-byte aligned");
    STAIR_CHECK(a.ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.h_n) & 15) == 0,
                "out / h_n must be 16-byte aligned with ldo % 4 == 0");
    p.xproj = a.xproj_ws; p.w_hh[0] = a.w_hh[0]; p.w_hh[1] = a.w_hh[1]; p.seq_off = a.seq_off; p.seq_len = a.seq_len;
    p.out = a.out; p.ldo = a.ldo; p.h_n = a.h_n; p.cbuf = a.cbuf; p.n = a.n; p.status = a.status;
    nt = 1;
    coop_geometry(a.n, nt, p.gpd);
    const int G = 2 * p.gpd;
    char *base = static_cast<char *>(a.coop_ws);
    const int64_t slab_bytes = (int64_t)2 * G * nt * TILE_BYTES;
    const int flag_words = G * nt * CP * FLAG_STRIDE + 64;
    p.xh = base;
    p.flags = reinterpret_cast<unsigned *>(base + slab_bytes);
    p.err = p.flags + G * nt * CP * FLAG_STRIDE;
    // flags and the error word are one block, zeroed before every launch (epochs restart at 1)
    if (int rcz_ = launch_zero(p.flags, (size_t)flag_words * 4, s)) return rcz_;
    // h_n of empty sequences is zero and the kernel only writes it at a sequence's last step
    if (int rcz_ = launch_zero(a.h_n, (size_t)a.n * 2 * CH * sizeof(float), s)) return rcz_;
    blocks = coop_blocks(p.gpd);
    return 0;
}

static void coop_attrs() {
    bool &attr_set = coop_device().rec_attrs;
    if (attr_set) return;
#define C_ATTR(TR_, NT_) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_rec_coop_kernel<TR_, NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, NT_ * TILE_BYTES + 64);
    C_ATTR(false, 1) C_ATTR(false, 2) C_ATTR(false, 3) C_ATTR(true, 1) C_ATTR(true, 2) C_ATTR(true, 3)
#undef C_ATTR
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_rec_coop_kernel<false, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 1 * TILE_BYTES + 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_rec_coop_kernel<false, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE_BYTES + 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_rec_coop_kernel<false, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * TILE_BYTES + 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_rec_coop_pair_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 1 * TILE_BYTES + 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_rec_coop_pair_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 1 * TILE_BYTES + 64);
    attr_set = true;
}

// -1: the grid would not be co-resident on this device (the caller runs the one-workgroup kernel)
int launch_lstm_rec_coop(const stair_lstm_args &a, hipStream_t s) {
    CoopParams p;
    int nt = 1, blocks = 0;
    coop_attrs();
    {
        int gpd = 0;
        coop_geometry(a.n, nt, gpd);
        const int nb = coop_blocks(gpd);
        const size_t sh = (size_t)nt * TILE_BYTES + 64;
        const bool tr = a.cbuf != nullptr;
        const bool ok = nt == 1 ? (tr ? coop_fits(&lstm_rec_coop_kernel<true, 1>, 0, nb, sh) : coop_fits(&lstm_rec_coop_kernel<false, 1>, 1, nb, sh))
                      : nt == 2 ? (tr ? coop_fits(&lstm_rec_coop_kernel<true, 2>, 2, nb, sh) : coop_fits(&lstm_rec_coop_kernel<false, 2>, 3, nb, sh))
                                : (tr ? coop_fits(&lstm_rec_coop_kernel<true, 3>, 4, nb, sh) : coop_fits(&lstm_rec_coop_kernel<false, 3>, 5, nb, sh));
        if (!ok) return -1;
    }
    if (int rc = coop_prepare(a, s, p, nt, blocks)) return rc;
    static const bool prof = [] { const char *e = getenv("STAIR_LSTM_COOP_PROF"); return e && e[0] == '1'; }();   // diagnostic build, never the product
    if (prof && !a.cbuf) {
        if (nt == 1) hipLaunchKernelGGL((lstm_rec_coop_kernel<false, 1, true>), dim3(blocks), dim3(512), 1 * TILE_BYTES + 64, s, p);
        else if (nt == 2) hipLaunchKernelGGL((lstm_rec_coop_kernel<false, 2, true>), dim3(blocks), dim3(512), 2 * TILE_BYTES + 64, s, p);
        else hipLaunchKernelGGL((lstm_rec_coop_kernel<false, 3, true>), dim3(blocks), dim3(512), 3 * TILE_BYTES + 64, s, p);
        STAIR_LAUNCH_CHECK();
        return 0;
    }
    STAIR_ACCT_MFMA("lstm_rec_coop", 0, 2ll * 2 * a.rows * 4 * CH * CH);
#define C_LAUNCH(NT_)                                                                                                              \
    if (a.cbuf) hipLaunchKernelGGL((lstm_rec_coop_kernel<true, NT_>), dim3(blocks), dim3(512), NT_ * TILE_BYTES + 64, s, p);        \
    else hipLaunchKernelGGL((lstm_rec_coop_kernel<false, NT_>), dim3(blocks), dim3(512), NT_ * TILE_BYTES + 64, s, p);
    if (nt == 1) { C_LAUNCH(1) } else if (nt == 2) { C_LAUNCH(2) } else { C_LAUNCH(3) }
#undef C_LAUNCH
    STAIR_LAUNCH_CHECK();
    return 0;
}

// Both recurrences in one launch when each has one sequence tile per group and all their workgroups are co-resident;
// -1: not applicable (the caller runs them one after the other).  a and b need DISJOINT coop_ws regions.
int launch_lstm_rec_coop_pair(const stair_lstm_args &a, const stair_lstm_args &b, hipStream_t s) {
    static const bool on = [] { const char *e = getenv("STAIR_LSTM_COOP_PAIR"); return !(e && e[0] == '0'); }();
    if (!on || !a.coop_ws || !b.coop_ws || a.coop_ws == b.coop_ws || !lstm_coop_usable(a.Hh) || !lstm_coop_usable(b.Hh)) return -1;
    if ((a.cbuf != nullptr) != (b.cbuf != nullptr) || a.n <= 0 || b.n <= 0) return -1;
    int nta = 1, ntb = 1, ga = 0, gb = 0;
    coop_geometry(a.n, nta, ga);
    coop_geometry(b.n, ntb, gb);
    if (nta != 1 || ntb != 1) return -1;
    if (coop_blocks(ga) + coop_blocks(gb) > coop_cu_count()) return -1;
    static const bool prof = [] { const char *e = getenv("STAIR_LSTM_COOP_PROF"); return e && e[0] == '1'; }();
    if (prof) return -1;
    coop_attrs();
    if (a.cbuf ? !coop_fits(&lstm_rec_coop_pair_kernel<true>, 6, coop_blocks(ga) + coop_blocks(gb), TILE_BYTES + 64)
               : !coop_fits(&lstm_rec_coop_pair_kernel<false>, 7, coop_blocks(ga) + coop_blocks(gb), TILE_BYTES + 64)) return -1;
    CoopParams pa, pb;
    int blocks_a = 0, blocks_b = 0;
    if (int rc = coop_prepare(a, s, pa, nta, blocks_a)) return rc;
    if (int rc = coop_prepare(b, s, pb, ntb, blocks_b)) return rc;
    STAIR_ACCT_MFMA("lstm_rec_coop_pair", 0, 2ll * 2 * (a.rows + b.rows) * 4 * CH * CH);
    if (a.cbuf) hipLaunchKernelGGL(lstm_rec_coop_pair_kernel<true>, dim3(blocks_a + blocks_b), dim3(512), 1 * TILE_BYTES + 64, s, pa, pb, blocks_a);
    else hipLaunchKernelGGL(lstm_rec_coop_pair_kernel<false>, dim3(blocks_a + blocks_b), dim3(512), 1 * TILE_BYTES + 64, s, pa, pb, blocks_a);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// Cooperative BPTT: the reverse-time recurrence with the same ownership as the forward kernel above.
//
// Workgroup j of a group owns hidden units [64 j, 64 j + 64) of up to 32 NT sequences of one direction: the cell
// backward of those units is lane-local (the lane that holds (sequence r, units U0..U0+3) reads their four saved gates,
// c(t), c(t-1) and d_out(t), and keeps dh / dc in registers).  What crosses units is
//     dh(t-1)[k] = sum over the 4 Hh gate rows rho of dgates(t)[rho] * W_hh[rho][k]:
// each workgroup multiplies ITS 256 gate rows (gate q, unit 64 j + 8 w' + u' <-> local row 32 w' + 8 q + u') into ALL 256
// hidden columns -- the transposed quarter of W_hh lives in 128 VGPRs as bf16 hi + lo fragments for the whole launch, the
// gate gradients are the B operand out of LDS (bf16 hi + lo, written by the cell lanes) -- and publishes the fp32 partial
// sums, each 64-column block to the workgroup that owns those units: slab [dst][src][32 seq][64 units], write-through
// 16-byte stores, epoch flags and sc1 loads exactly as in the forward kernel (Guideline 16 R1, "Valid forms" row 1).
// The consumer lane loads the four partials of its own 4 units (4 x 16 B) straight into registers and adds them in source
// order, so the result does not depend on timing.  (lstm_bwd_x3_kernel, which this replaces for Hh = 256, re-streams the
// 1 MB image of W_hh^T from L2 on each of the serial steps: 12 us per step whatever the batch.)
// =============================================================================================
namespace {

constexpr int BSLAB_BYTES = CP * CP * 32 * 64 * 4;      // one tile and parity of partial sums: 128 KB

struct CoopBwdParams {
    float *G;                 // [rows, 8 Hh]: activated gates in, gate pre-activation gradients out (in place)
    const float *cbuf;        // [rows, 2 Hh]
    const float *d_out;       // [rows, ldd]
    int64_t ldd;
    const float *d_hn;        // [n, 2 Hh] or null
    const float *w_hh[2];
    const int32_t *seq_off, *seq_len;
    char *xh;                 // partial-sum slabs [2 parity][groups][NT][BSLAB_BYTES]
    unsigned *flags, *err, *status;
    int n, gpd;
};

}  // namespace

template <int NT>
__device__ __forceinline__ void lstm_bwd_coop_body(const CoopBwdParams &p, const int b, char *hl) {
    // hl: [NT tiles][2 planes][32 seq][256 local gate rows] bf16, swizzled; counters
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int q8 = b >> 3;
    const int j = q8 & 3, g = (b & 7) + 8 * (q8 >> 2);
    const int G = 2 * p.gpd;
    if (g >= G) return;
    const int dir = g / p.gpd, gl = g - dir * p.gpd;
    constexpr int Hh = CH;
    constexpr int SPG = 32 * NT;
    const int U0 = 64 * j + 8 * wave + 4 * hh;                     // this lane's 4 hidden units in the cell backward
    const int64_t ldx = 8 * (int64_t)Hh;
    const int xcol = dir * 4 * Hh + U0;

    // ---- W_hh^T fragments: A[m][rho] = W_hh[row(rho)][32 wave + m], rho = 16 s + 8 hh + e <-> gate 2 (s & 1) + hh, unit 64 j + 8 (s >> 1) + e ----
    bf16x8 whi[16], wlo[16];
    {
        const float *wcol = p.w_hh[dir] + 32 * wave + r;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float *w0 = wcol + (int64_t)((2 * (s & 1) + hh) * Hh + 64 * j + 8 * (s >> 1)) * Hh;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = w0[(int64_t)e * Hh];
                whi[s][e] = (__bf16)v; wlo[s][e] = (__bf16)(v - (float)whi[s][e]);
            }
        }
    }
    __amdgpu_buffer_rsrc_t xh_rs = __builtin_amdgcn_make_buffer_rsrc(p.xh, 0, 0x7fffffff, 0x00020000);
    auto slab_off = [&](unsigned ep, int x) { return (int)(((int64_t)((ep & 1) * G + g) * NT + x) * BSLAB_BYTES); };
    auto flag_of = [&](int x, int wg) { return (gu32 *)(p.flags + ((g * NT + x) * CP + wg) * FLAG_STRIDE); };
    unsigned *counters = reinterpret_cast<unsigned *>(hl + NT * TILE_BYTES);

    unsigned ebase = 0;
    bool dead = false;
    for (int chunk = gl; chunk * SPG < p.n; chunk += p.gpd) {
        const int s_base = chunk * SPG;
        int off[NT], len[NT], lmax = 0;
        float dh[NT][4], dc[NT][4];
#pragma unroll
        for (int x = 0; x < NT; ++x) {
            const int s = s_base + 32 * x + r;
            off[x] = 0; len[x] = 0;
            if (s < p.n) { off[x] = p.seq_off[s]; len[x] = p.seq_len ? p.seq_len[s] : p.seq_off[s + 1] - off[x]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dc[x][i] = 0.0f;
                dh[x][i] = (p.d_hn && s < p.n) ? p.d_hn[(int64_t)s * 2 * Hh + dir * Hh + U0 + i] : 0.0f;
            }
        }
        for (int s = s_base; s < min(s_base + SPG, p.n); ++s) lmax = max(lmax, p.seq_len ? p.seq_len[s] : p.seq_off[s + 1] - p.seq_off[s]);
        for (int i = tid; i < NT * TILE_BYTES / 16 + 1; i += 512) reinterpret_cast<v4u *>(hl)[i] = v4u{0, 0, 0, 0};
        __syncthreads();

        int pend_x = -1; unsigned pend_ep = 0;                     // a published tile whose flag is still to be stored
        auto flush_pending = [&]() {
            if (pend_x < 0) return;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its write-through stores ...
            if (lane == 0) {                                       // ... then the wave whose add comes last stores the flag
                const unsigned old = __hip_atomic_fetch_add(counters + pend_x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (old == 7u) {
                    __hip_atomic_store(counters + pend_x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(flag_of(pend_x, j), pend_ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            pend_x = -1;
        };

        for (int k = 0; k < lmax; ++k) {
            const int tau = lmax - 1 - k;
#pragma unroll
            for (int x = 0; x < NT; ++x) {
                char *hx = hl + x * TILE_BYTES;
                const bool active = tau < len[x];
                const int tt = dir == 0 ? tau : len[x] - 1 - tau;
                const int64_t row = off[x] + (active ? tt : 0);
                const int64_t rowp = row + (dir == 0 ? -1 : 1);
                // ---- the saved forward state of this step: independent of the recurrence, in flight during the poll ----
                v4f gi, gf, gg, go, cc, cp, dy;
                gi = gf = gg = go = cc = cp = dy = v4f{0.f, 0.f, 0.f, 0.f};
                if (active) {
                    const float *gs = p.G + row * ldx + xcol;
                    gi = *(const __attribute__((address_space(1))) v4f *)(gs);
                    gf = *(const __attribute__((address_space(1))) v4f *)(gs + Hh);
                    gg = *(const __attribute__((address_space(1))) v4f *)(gs + 2 * Hh);
                    go = *(const __attribute__((address_space(1))) v4f *)(gs + 3 * Hh);
                    cc = *(const __attribute__((address_space(1))) v4f *)(p.cbuf + row * 2 * Hh + dir * Hh + U0);
                    if (tau > 0) cp = *(const __attribute__((address_space(1))) v4f *)(p.cbuf + rowp * 2 * Hh + dir * Hh + U0);
                    dy = *(const __attribute__((address_space(1))) v4f *)(p.d_out + row * p.ldd + dir * Hh + U0);
                }
                // ---- dh(t) of my units: the four workgroups' partial sums of the previous step ----
                if (k > 0) {
                    const unsigned ep = ebase + k;
                    if (!dead) {
                        const gu32 *fl = flag_of(x, lane & 3);
                        unsigned spins = 0, seen = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        while (!__all((int)(seen - ep) >= 0)) {
                            if (++spins > SPIN_LIMIT) {
                                if (lane == 0) {
                            __hip_atomic_store((gu32 *)p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (p.status) __hip_atomic_store((gu32 *)p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                                dead = true;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                            seen = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // no instruction: keeps the loads below the poll
                    const int so = slab_off(ep, x) + ((j * CP * 32 + r) * 64 + 8 * wave + 4 * hh) * 4;
                    v4u part[CP];
#pragma unroll
                    for (int src = 0; src < CP; ++src)
                        part[src] = __builtin_amdgcn_raw_buffer_load_b128(xh_rs, so + src * (32 * 64 * 4), 0, 16);     // aux 16 = sc1
                    if (tau + 1 < len[x]) {                        // the sequence took part in the previous step
                        v4f sum = __builtin_bit_cast(v4f, part[0]);
#pragma unroll
                        for (int src = 1; src < CP; ++src) sum += __builtin_bit_cast(v4f, part[src]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) dh[x][i] = sum[i];
                    }
                }
                // ---- cell backward (lstm_bwd_x3_kernel's arithmetic, lane-local) ----
                v4f di = v4f{0.f, 0.f, 0.f, 0.f}, df = di, dg = di, dob = di;
                if (active) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float dht = dh[x][i] + dy[i];
                        const float tc = tanh_fast(cc[i]);
                        dob[i] = dht * tc * go[i] * (1.0f - go[i]);
                        const float dct = dc[x][i] + dht * go[i] * (1.0f - tc * tc);
                        di[i] = dct * gg[i] * gi[i] * (1.0f - gi[i]);
                        df[i] = dct * cp[i] * gf[i] * (1.0f - gf[i]);
                        dg[i] = dct * gi[i] * (1.0f - gg[i] * gg[i]);
                        dc[x][i] = dct * gf[i];
                    }
                    if (dead) di = v4f{NAN, NAN, NAN, NAN};       // a hand-off timed out: fail loudly (NaN into every weight gradient)
                    float *gs = p.G + row * ldx + xcol;
                    *(__attribute__((address_space(1))) v4f *)(gs) = di;
                    *(__attribute__((address_space(1))) v4f *)(gs + Hh) = df;
                    *(__attribute__((address_space(1))) v4f *)(gs + 2 * Hh) = dg;
                    *(__attribute__((address_space(1))) v4f *)(gs + 3 * Hh) = dob;
                }
                {
                    const v4f vq[4] = {di, df, dg, dob};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bf16x4 ph, pl;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { ph[i] = (__bf16)vq[q][i]; pl[i] = (__bf16)(vq[q][i] - (float)ph[i]); }
                        const int o = lds_unit(r, 4 * wave + q) + 8 * hh;
                        *reinterpret_cast<v2u *>(hx + o) = __builtin_bit_cast(v2u, ph);
                        *reinterpret_cast<v2u *>(hx + 32 * CH * 2 + o) = __builtin_bit_cast(v2u, pl);
                    }
                }
                __syncthreads();
                if (k + 1 < lmax) {                                // the last step's dh(t-1) has no consumer
                    f32x16 acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(hx + lds_unit(r, 2 * s + hh));
                        const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(hx + 32 * CH * 2 + lds_unit(r, 2 * s + hh));
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[s], bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[s], bl, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[s], bh, acc, 0, 0, 0);
                    }
                    if (NT > 1) flush_pending();                   // the other tile's stores were issued one chain ago
                    // publish: rows 32 wave + 8 q + 4 hh + i of dh(t-1) go to workgroup wave >> 1, 16 bytes per gate block
                    const int so = slab_off(ebase + k + 1, x) + ((((wave >> 1) * CP + j) * 32 + r) * 64 + 32 * (wave & 1) + 4 * hh) * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v4f v = v4f{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), xh_rs, so + 8 * q * 4, 0, 16);   // aux 16 = sc1: write-through
                    }
                    pend_x = x; pend_ep = ebase + k + 1;
                    if (NT == 1) flush_pending();
                } else if (NT > 1) {
                    flush_pending();
                }
            }
        }
        flush_pending();
        // the next chunk's first publish must not land in the slab a slower workgroup is still reading (parity of the
        // last epoch consumed): skip one epoch
        ebase += lmax + 1;
        __syncthreads();
    }
}

template <int NT>
__global__ __launch_bounds__(512, 1) void lstm_bwd_coop_kernel(CoopBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char hl_dyn[];
    lstm_bwd_coop_body<NT>(p, (int)blockIdx.x, hl_dyn);
}

// BPTT of two encoders in one launch (see lstm_rec_coop_pair_kernel)
__global__ __launch_bounds__(512, 1) void lstm_bwd_coop_pair_kernel(CoopBwdParams a, CoopBwdParams b, int nb_a) {
    extern __shared__ __attribute__((aligned(16))) char hl_dyn[];
    if ((int)blockIdx.x < nb_a) lstm_bwd_coop_body<1>(a, (int)blockIdx.x, hl_dyn);
    else lstm_bwd_coop_body<1>(b, (int)blockIdx.x - nb_a, hl_dyn);
}

int64_t lstm_coop_bwd_ws_bytes(int n) {
    (void)n;
    const int G = 64;
    return (int64_t)2 * G * 2 * BSLAB_BYTES + (int64_t)(G * 2 * CP * FLAG_STRIDE + 64) * 4;
}

static void coop_bwd_geometry(int n, int &nt, int &gpd) {
    nt = 1;
    coop_geometry(n, nt, gpd);
    if (nt > 2) nt = 2;
    gpd = std::max(1, std::min((n + 32 * nt - 1) / (32 * nt), std::max(1, std::min(32, coop_cu_count() / 8))));
}

static int coop_bwd_prepare(const stair_lstm_bwd_args &a, hipStream_t s, CoopBwdParams &p, int &nt, int &blocks) {
    STAIR_CHECK(a.Hh == CH, "cooperative BPTT is built for Hh = 256");
    STAIR_CHECK(a.coop_ws && a.coop_ws_bytes >= lstm_coop_bwd_ws_bytes(a.n), "coop_ws missing or too small");
    STAIR_CHECK((reinterpret_cast<uintptr_t>(a.coop_ws) & 255) == 0, "coop_ws must be 256-byte aligned");
    STAIR_CHECK(a.ldd % 4 == 0 && (reinterpret_cast<uintptr_t>(a.d_out) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.gates) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(a.cbuf) & 15) == 0, "gates / cbuf / d_out must be 16-byte aligned with ldd % 4 == 0");
    p.G = a.gates; p.cbuf = a.cbuf; p.d_out = a.d_out; p.ldd = a.ldd; p.d_hn = a.d_hn;
    p.w_hh[0] = a.w_hh[0]; p.w_hh[1] = a.w_hh[1]; p.seq_off = a.seq_off; p.seq_len = a.seq_len; p.n = a.n; p.status = a.status;
    coop_bwd_geometry(a.n, nt, p.gpd);
    const int G = 2 * p.gpd;
    char *base = static_cast<char *>(a.coop_ws);
    const int64_t slab_bytes = (int64_t)2 * G * nt * BSLAB_BYTES;
    const int flag_words = G * nt * CP * FLAG_STRIDE + 64;
    p.xh = base;
    p.flags = reinterpret_cast<unsigned *>(base + slab_bytes);
    p.err = p.flags + G * nt * CP * FLAG_STRIDE;
    if (int rcz_ = launch_zero(p.flags, (size_t)flag_words * 4, s)) return rcz_;
    blocks = coop_blocks(p.gpd);
    return 0;
}

static void coop_bwd_attrs() {
    bool &attr_set = coop_device().bwd_attrs;
    if (attr_set) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_bwd_coop_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 1 * TILE_BYTES + 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_bwd_coop_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE_BYTES + 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_bwd_coop_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 1 * TILE_BYTES + 64);
    attr_set = true;
}

// -1: the grid would not be co-resident on this device (the caller runs the one-workgroup BPTT kernel)
int launch_lstm_bwd_coop(const stair_lstm_bwd_args &a, hipStream_t s) {
    CoopBwdParams p;
    int nt = 1, blocks = 0;
    coop_bwd_attrs();
    {
        int gpd = 0;
        coop_bwd_geometry(a.n, nt, gpd);
        const size_t sh = (size_t)nt * TILE_BYTES + 64;
        if (nt == 1 ? !coop_fits(&lstm_bwd_coop_kernel<1>, 8, coop_blocks(gpd), sh) : !coop_fits(&lstm_bwd_coop_kernel<2>, 9, coop_blocks(gpd), sh)) return -1;
    }
    if (int rc = coop_bwd_prepare(a, s, p, nt, blocks)) return rc;
    STAIR_ACCT_MFMA("lstm_bwd_coop", 0, 2ll * 2 * a.rows * 4 * CH * CH);
    if (nt == 1) hipLaunchKernelGGL((lstm_bwd_coop_kernel<1>), dim3(blocks), dim3(512), 1 * TILE_BYTES + 64, s, p);
    else hipLaunchKernelGGL((lstm_bwd_coop_kernel<2>), dim3(blocks), dim3(512), 2 * TILE_BYTES + 64, s, p);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// BPTT of two encoders in one launch; -1: not applicable.  a and b need DISJOINT coop_ws regions.
int launch_lstm_bwd_coop_pair(const stair_lstm_bwd_args &a, const stair_lstm_bwd_args &b, hipStream_t s) {
    static const bool on = [] { const char *e = getenv("STAIR_LSTM_COOP_PAIR"); return !(e && e[0] == '0'); }();
    if (!on || !a.coop_ws || !b.coop_ws || a.coop_ws == b.coop_ws || !lstm_coop_usable(a.Hh) || !lstm_coop_usable(b.Hh)) return -1;
    if (a.n <= 0 || b.n <= 0) return -1;
    int nta = 1, ntb = 1, ga = 0, gb = 0;
    coop_geometry(a.n, nta, ga);
    coop_geometry(b.n, ntb, gb);
    if (nta != 1 || ntb != 1) return -1;
    if (coop_blocks(ga) + coop_blocks(gb) > coop_cu_count()) return -1;
    coop_bwd_attrs();
    if (!coop_fits(&lstm_bwd_coop_pair_kernel, 10, coop_blocks(ga) + coop_blocks(gb), TILE_BYTES + 64)) return -1;
    CoopBwdParams pa, pb;
    int blocks_a = 0, blocks_b = 0;
    if (int rc = coop_bwd_prepare(a, s, pa, nta, blocks_a)) return rc;
    if (int rc = coop_bwd_prepare(b, s, pb, ntb, blocks_b)) return rc;
    STAIR_ACCT_MFMA("lstm_bwd_coop_pair", 0, 2ll * 2 * (a.rows + b.rows) * 4 * CH * CH);
    hipLaunchKernelGGL(lstm_bwd_coop_pair_kernel, dim3(blocks_a + blocks_b), dim3(512), 1 * TILE_BYTES + 64, s, pa, pb, blocks_a);
    STAIR_LAUNCH_CHECK();
    return 0;
}

}  // namespace stair

extern "C" int stair_lstm_coop_limit(int32_t max_blocks) { stair::g_coop_cap = max_blocks; return 0; }
extern "C" int64_t stair_lstm_coop_ws_bytes(int32_t n) { return stair::lstm_coop_ws_bytes(n); }
extern "C" int64_t stair_lstm_coop_bwd_ws_bytes(int32_t n) { return stair::lstm_coop_bwd_ws_bytes(n); }
