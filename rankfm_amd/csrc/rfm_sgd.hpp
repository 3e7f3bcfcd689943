// rfm_sgd.hpp -- the BPR/WARP SGD wavefront kernels of the MI355X RankFM engine (gfx950 only).
//
// Replaces the reference's sequential row loop rankfm/_rankfm.pyx:230-326 (one SGD step per shuffled interaction:
// score positive, draw/score negative(s), sigmoid gradient, in-place update of w_i, w_if, v_u, v_i, v_uf, v_if)
// together with compute_ui_utility (:48-89), the rejection sampler (:250-253 + lsearch :20-27) and the WARP
// early-exit loop (:244-270).
//
// Mapping onto CDNA4:
//   * A "row group" of G lanes (G = 16 for 16 <= F <= 128) owns one interaction; lane s of the group owns factor
//     dwords s, s+G, s+2G, ...  Every load / atomic instruction of a group therefore covers ONE contiguous 64-byte
//     segment of a factor row, and a wavefront carries 64/G interactions at once.  (The L2 atomic path is
//     request-bound: 16-byte-per-lane chunks gave 4 dwords per 64-byte request and ran 3.4x slower, profiles/.)
//   * The k-dimension dot products are xor-butterfly reductions inside the group (DPP / ds_bpermute); every lane
//     ends with the bit-identical sum, so the WARP control flow is group-uniform.
//   * Negative draws are counter based (include/rfm_rng.h): no shared RNG state, any lane can draw.
//   * No MFMA anywhere: ~2 flops per 4-byte factor element; this is an HBM/L2 gather-scatter.
//
// Two kernels share the step (RowStep):
//   sgd_rows_kernel      one interaction per group per iteration, positions of the epoch's shuffled order striding the
//                        grid.  SERIAL instantiation: ONE wavefront, ONE group, plain read-modify-write in the reference's
//                        exact order -- the reference's sequential semantics (golden-vector parity, debugging).  The
//                        Hogwild instantiation is used when the caller dictates the visiting order (`perms`).
//   sgd_segments_kernel  production Hogwild.  The unit of work is a user SEGMENT (<= 32 consecutive CSR rows of one
//                        user): the group keeps v_u[u] in registers for the whole segment (read once, one atomic
//                        delta write-back), walks the user's rows straight out of the CSR arrays (no interaction /
//                        permutation / sample-weight gathers) and updates the two item rows with fp32 hardware
//                        atomics (global_atomic_add_f32).  Segments are visited in a keyed pseudo-random order and rows
//                        inside a segment in a keyed order (rfm_rng.h), both reproducible on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rfm_rng.h"

namespace rfm {

struct SgdArgs {
    const int32_t *__restrict__ interactions;   // [N,2]                         (rows kernel)
    const float *__restrict__ sample_weight;    // [N]                           (rows kernel)
    const int64_t *__restrict__ csr_off;        // [U+1]
    const int32_t *__restrict__ csr_items;      // [nnz] positive items by CSR position, sorted within each user
    const float *__restrict__ x_uf;             // [U,P]
    const float *__restrict__ x_if;             // [I,Q]
    float *w_i, *w_if, *v_u, *v_i, *v_uf, *v_if;
    // Hogwild kernels address the item biases as w_i[i * w_stride]: with w_stride = 16 every bias has a 64-byte line of
    // its own (a padded copy in the workspace).  Sixteen biases per line means the two bias atomics of every in-flight
    // update collide on ~3000 lines and retire serially memory-side: they cost as much as the four atomics of an item
    // row (measured on config 2, uniform items: 2.88 ms with, 2.37 ms without the bias atomics).
    int32_t w_stride;
    int32_t scale_in_pad;                       // 1: dword 1 of an item's padded bias line holds pos_scale[item] (one request for both)
    const int32_t *__restrict__ perm;           // this epoch's visiting order [N] or nullptr        (rows kernel)
    const float *__restrict__ sw_csr;           // [N] sample weight by CSR position                 (segments kernel)
    const int4 *__restrict__ seg_desc;          // [S] {user, first CSR position, length, 0}         (segments kernel)
    const float *__restrict__ multiplier;       // [max_samples+1]: log((I-1)/s)/log(I), s = 1..max_samples (host, double)
    uint32_t *mt_state;                         // [625] MT19937 words + index (serial + MT only)
    double *ll;                                 // this epoch's log-likelihood accumulator
    unsigned long long *draws;                  // this epoch's accepted-draw counter
    unsigned int *error_flags;                  // bit 0: rejection sampler gave up
    int64_t pos_begin, pos_end;                 // positions (rows kernel) / segment-order positions (segments kernel)
    int64_t n_rows;                             // N
    int64_t n_segments;                         // S
    int32_t n_items, n_uf, n_if, n_factors;     // I, P, Q, F
    int32_t has_uf, has_if;
    int32_t max_samples;
    int32_t rng;                                // RFM_RNG_*
    uint32_t epoch_key, perm_bits, seg_bits;
    float eta, reg_a, reg_b;                    // learning rate of the epoch, 2*alpha, 2*beta
    // Hogwild step damping (DESIGN.md "staleness"): a row that n in-flight updates touch at once receives n steps computed
    // from the same stale value; above ~M of them the combined step overshoots.  The step on such a row is scaled by
    // min(1, M / n), with n = in-flight rows x the row's share of the data.  All 1 / null in serial mode.
    const float *__restrict__ pos_scale;        // [I] scale for the positive item's row (by item popularity), or nullptr
    float user_cap;                             // a user of degree d gets min(1, user_cap / d)
    float feat_scale;                           // scale for the dense feature tables (every row touches them)
    int32_t single_group;                       // debug: only group 0 of wavefront 0 works (sequential Hogwild kernel)
    int64_t max_groups;                         // row groups allowed to work (the concurrency cap can be below one workgroup)
    int32_t block_threads;                      // workgroup size of the features row-loop kernel
    int32_t table_threads;                      // workgroup size of the tables kernel (trainer + producers)
    // hot positive items (segments kernel, HOT instantiation): pos_scale[i] >= 2 encodes slot = int(v / 2) - 1 and
    // scale = v - 2 (slot + 1).  A workgroup accumulates its updates of slot s in LDS and publishes them with one set of
    // atomics every hot_period[s] touches (DESIGN.md "hot rows").
    const int32_t *__restrict__ hot_item;       // [n_hot] item index of each slot
    const int32_t *__restrict__ hot_period;     // [n_hot] touches per workgroup between publications
    int32_t n_hot;
    // Publications do not go to the hot rows themselves: memory-side atomics on ONE address retire serially, and 256
    // workgroups publishing into the same 64 rows cost 0.55 ms of a 3.6 ms epoch (measured by publishing to private
    // addresses instead).  A workgroup adds its pending sums into bin (workgroup % kHotBins) of these arrays; every
    // 64-byte line of the bins has an owner workgroup that sweeps it every few rows (exchange with zero over the bins,
    // one atomic add of the total into v_i / w_i), and hot_reduce_kernel drains what is left when the launch ends.
    // With few workgroups (fewer than a quarter of the lines) there is little contention and a sweeping turn would take
    // long: hot_direct = 1 publishes straight into the rows.
    float *hot_bins_v;                          // [kHotBins, n_hot, F]
    float *hot_bins_w;                          // [kHotBins, n_hot]
    int32_t hot_direct;
    const unsigned int *sw_max_bits;            // bits of max |sample_weight| (plan): range of the fixed-point hot sums
    // negative stripes (segments kernel, STRIPE instantiation; include/rfm_rng.h "negative stripes"): the workgroup draws the
    // negatives of a window of `stripe_window` rows per group from `stripe_rows` items whose rows it holds in LDS
    int32_t stripe_rows, stripe_window;
    uint32_t item_bits, launch_index;
    float stripe_cover;                         // share of the catalogue that sits in some workgroup's stripe at any time, <= 1
    // features kernel: the step producers hand their batches to the table trainer through `feat_ring` ([2 * n_producers] slots of
    // one staged step per row group of a workgroup), synchronised by the counters in `feat_flags` (sgd_features_kernel)
    float *feat_ring;
    unsigned int *feat_flags;
    int32_t n_producers;
    int32_t feat_frozen;                        // debug: the feature tables are not trained (no trainer, no producers)
    // dynamic segment order (segments kernel without stripes, pipelined feature row loop): a row group takes its next segment from
    // a ticket counter instead of striding the order with the number of groups (SegmentTickets below); nullptr = static stride
    unsigned int *tickets;                      // the launch's counter of order positions handed out, zero at launch
    int32_t damp_positive_only;                 // experiments: the round-3 rule (an item's scale applies to its step as the POSITIVE item only)
    int32_t reserved_i32;
    // features: the table trainer applies EXACTLY table_quota staged steps per launch (rounded up to whole batches) -- a number the host
    // derives from the launch's rows and geometry, not from when the row loops happen to finish (feat_tables_kernel)
    int64_t table_quota;
    unsigned long long *feat_clock;             // [4] wall-clock ticks: tables kernel begin | end | row-loop kernel begin | end (diagnostics)
    unsigned long long *sclk;                   // [4] workgroup 0 of the row-loop kernel: wall clock (100 MHz) at its start | end, shader cycle counter at its start | end
};
constexpr int kTicketWords = 16;                // one counter per launch, on a 64-byte line of its own
constexpr int kHotBins = 16;

constexpr size_t kLdsBytes = 160 * 1024;        // per workgroup on gfx950
// LDS floats of a negative stripe of R rows: [R] items | [R, F+1] snapshot | [R, F+1] pending sums | [F+1] their column sums
inline size_t stripe_lds_floats(int rows, int n_factors) { return (size_t)rows * (1 + 2 * ((size_t)n_factors + 1)) + (size_t)n_factors + 1; }

constexpr float kMargin = 1.0f;                 // rankfm/_rankfm.pyx:149
constexpr uint32_t kMaxAttempts = 1u << 22;     // safety net; the host rejects saturated users up front
constexpr int kSegmentRows = 32;                // longest user segment (host planner uses the same constant)

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
// All-reduce over the G lanes of a row group; every lane ends with the bit-identical sum (the WARP control flow relies on
// it).  Inside a 16-lane row the partner values come through DPP row rotations -- plain VALU operands, no LDS round trip
// (a ds_bpermute-based butterfly is four dependent ~100-clock LDS accesses per dot product, and a WARP row computes ~20 of them:
// the candidate scoring loop was bound by exactly that latency chain).  Rotation by 8, 4, 2, 1 pairs the same lanes as the xor
// butterfly (after the first step the partial sums have period 8, and so on), so the result is the butterfly's, bit for bit.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int32_t dpp_movi(int32_t x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, false);
}
// The launch's arguments re-read from the kernel-argument segment.  A kernel keeps every field of its by-value SgdArgs it ever uses
// in scalar registers from its first instruction on; the WARP kernel, say, needs ~150 and has 102, and what does not fit is parked in
// the lanes of a vector register -- one v_readlane (a VECTOR instruction, in a kernel bound by those) per use.  The rarely
// executed parts of a row loop (a segment's start, a row's update, the sweeping duty) instead take a copy of the arguments through
// a pointer the compiler cannot see through: the fields such a part uses are scalar loads when it is entered (the scalar cache
// holds the 456-byte segment) and occupy registers only inside it.  (SgdArgs is the kernel's first and only parameter: offset 0.)
__device__ __forceinline__ SgdArgs cold_args() {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) SgdArgs *KernelArgPtr;
    KernelArgPtr p = (KernelArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *p;                         // (only the fields the caller goes on to use are loaded)
#else
    return SgdArgs();                  // (the host pass of the compiler only parses device code)
#endif
}
// (the frozen stripe instantiations keep the code they were measured with: with COLD = false, `c` IS the kernel's parameter)
#define RFM_COLD_ARGS(c, COLD)                                            \
    const SgdArgs c##_reread_ = (COLD) ? cold_args() : SgdArgs();         \
    const SgdArgs &c = (COLD) ? c##_reread_ : a;

template <int G>
__device__ __forceinline__ float group_sum(float x) {
    if constexpr (G >= 16) {
        x += dpp_mov<0x128>(x);      // row_ror:8
        x += dpp_mov<0x124>(x);      // row_ror:4
        x += dpp_mov<0x122>(x);      // row_ror:2
        x += dpp_mov<0x121>(x);      // row_ror:1
#pragma unroll
        for (int m = 16; m < G; m <<= 1) x += __shfl_xor(x, m);
    } else {
        static_assert(G == 4, "row groups are 4, 16 or 64 lanes");
        x += dpp_mov<0x4E>(x);       // quad_perm:[2,3,0,1]
        x += dpp_mov<0xB1>(x);       // quad_perm:[1,0,3,2]
    }
    return x;
}

// fp32 hardware atomic add, no return value (global_atomic_add_f32)
__device__ __forceinline__ void atomic_add_f32(float *p, float v) { unsafeAtomicAdd(p, v); }

// LDS-resident tables are addressed through address_space(3) pointers so that the compiler emits ds_read / ds_add_f32.
// Through generic (flat) pointers every access would be a FLAT instruction, which has to wait on BOTH memory counters
// and serialises the step's outstanding global loads (measured: the feature kernel ran 180 us per row that way).
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ void atomic_add_f32(lds_float *p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <bool LDS> struct TablePtr { typedef float *type; };
template <> struct TablePtr<true> { typedef lds_float *type; };

// FRESH loads bypass the per-CU L1 (global_load_dword sc1): another CU's atomics are then visible as soon as they
// have been performed, instead of whenever the L1 line happens to be evicted
template <bool FRESH>
__device__ __forceinline__ float load_f32(const float *p) {
    if constexpr (FRESH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

template <bool PLAIN, class Ptr>
__device__ __forceinline__ void apply_f32(Ptr p, float oldv, float delta) {
    if (PLAIN) *p = oldv + delta;
    else atomic_add_f32(p, delta);
}

// membership of `item` in the user's sorted list: the predicate of lsearch (rankfm/_rankfm.pyx:20-27),
// evaluated by binary search
__device__ __forceinline__ bool is_member(const int32_t *__restrict__ items, int64_t lo, int64_t hi, int32_t item) {
    while (lo < hi) {
        const int64_t md = lo + ((hi - lo) >> 1);
        const int32_t v = items[md];
        if (v == item) return true;
        if (v < item) lo = md + 1; else hi = md;
    }
    return false;
}

// The same predicate evaluated by all G lanes of a row group together: one or two memory round trips instead of
// ~log2(degree) dependent ones.  Lists of up to 4G items are scanned outright (4 strided loads per lane, all in flight at
// once); longer lists are first narrowed by G-ary search steps (G evenly spaced pivots per step).  Arguments are
// group-uniform; every lane of the group must call it.
template <int G>
__device__ __forceinline__ unsigned group_ballot(bool pred) {
    const unsigned long long m = __ballot(pred);
    if constexpr (G == 64) return (unsigned)(m != 0ull);          // only "any" is needed for a full-wave group (see callers)
    else return (unsigned)((m >> (((threadIdx.x & 63) / G) * G)) & ((1ull << G) - 1ull));
}

template <int G>
__device__ __forceinline__ bool is_member_group(const int32_t *__restrict__ items, int64_t lo, int64_t hi, int32_t item, int sub) {
    while (hi - lo > 4 * G) {
        const int64_t n = hi - lo, step = (n + G - 1) / G;
        const int64_t p = lo + (int64_t)sub * step;
        const int32_t v = p < hi ? items[p] : 0x7fffffff;
        // lanes whose pivot is <= item form a prefix of the group (the list is sorted): its length picks the sub-range
        int c;
        if constexpr (G == 64) c = __popcll(__ballot(v <= item));
        else c = __popc(group_ballot<G>(v <= item));
        if (c == 0) return false;                                  // item below the first element
        lo = lo + (int64_t)(c - 1) * step;
        hi = lo + step < hi ? lo + step : hi;
    }
    bool found = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t idx = lo + sub + (int64_t)G * k;
        if (idx < hi) found |= (items[idx] == item);
    }
    if constexpr (G == 64) return __ballot(found) != 0ull;
    else return group_ballot<G>(found) != 0u;
}

// four candidates against one user's list: the list is read once
template <int G>
__device__ __forceinline__ void members4_group(const int32_t *__restrict__ items, int64_t lo, int64_t hi, const int32_t (&c)[4],
                                               bool (&m)[4], int sub) {
    if (hi - lo <= 4 * G) {
        bool f[4] = {false, false, false, false};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t idx = lo + sub + (int64_t)G * k;
            if (idx < hi) {
                const int32_t v = items[idx];
#pragma unroll
                for (int q = 0; q < 4; ++q) f[q] |= (v == c[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (G == 64) m[q] = __ballot(f[q]) != 0ull;
            else m[q] = group_ballot<G>(f[q]) != 0u;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = is_member_group<G>(items, lo, hi, c[q], sub);
    }
}

// MT19937 step on a state kept in global memory (serial mode, one lane).  Published algorithm of
// Matsumoto & Nishimura; the reference vendors it as rankfm/mt19937ar/mt19937ar.c:105-140.
__device__ inline uint32_t mt_next_global(uint32_t *st) {
    uint32_t idx = st[624];
    if (idx >= 624u) {
        for (int k = 0; k < 624; ++k) {
            const uint32_t y = (st[k] & 0x80000000u) | (st[(k + 1) % 624] & 0x7fffffffu);
            st[k] = st[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        idx = 0;
    }
    uint32_t y = st[idx];
    st[624] = idx + 1;
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// The reference evaluates exp / log in double and narrows (rankfm/_rankfm.pyx:269-276; rankfm/_rankfm.c:4780, 5247).  Here: the
// correctly-rounded-to-1-ulp fp32 library functions (expf, log1pf), not the fast intrinsics (__expf is exp2 of a rounded
// product: ~2 ulp and worse at large arguments).  Double-precision exp / log1p per row and lane were measured too: they cost the
// config-2 kernel 2.6 -> 4.55 ms (fp64 exp is ~100 instructions for all 64 lanes of a wavefront) for a difference below 1e-7 in
// d_outer -- far inside the 2e-5 the serial-mode tests allow against the reference's own numbers.
// log(1 / (1 + exp(-x))) (:270) and 1 / (1 + exp(x)) (:276) from ONE exponential, e = exp(-|x|) in (0, 1]: overflow-free, the
// 1-ulp library functions (expf, log1pf) and an IEEE division.  (The hardware's log2 / reciprocal instead of log1pf / the division
// were measured on config 2: ~130 fewer instructions per row and under 1 % of the kernel time -- the row loop is not bound by
// its arithmetic -- so the accurate forms stay.)
__device__ __forceinline__ void sigmoid_terms(float x, float &log_sig, float &sig_neg) {
    const float e = expf(-fabsf(x));
    const float r = 1.0f / (1.0f + e);
    log_sig = fminf(x, 0.0f) - log1pf(e);
    sig_neg = x >= 0.0f ? e * r : r;
}

// ---------------------------------------------------------------------------------------------
// one SGD step for one interaction, executed by the G lanes of a row group
//   SERIAL   plain read-modify-write everywhere, MT stream allowed
//   VU_REGS  v_u lives in the caller's registers: the step updates them in place and does not touch v_u memory
//   FRESH    item-row loads bypass L1
// ---------------------------------------------------------------------------------------------
//   LDSF     the dense feature tables (v_uf, v_if, w_if) are read from this workgroup's LDS copy (see TMODE)
//   HOT      updates of hot positive items are accumulated in the workgroup's LDS and published every few touches
//   WARPB    compile the batched WARP draw loop (max_samples > 1); the BPR instantiation stays at ~76 VGPRs without it
//   STRIPE   negatives come from the workgroup's LDS stripe: candidate rows are read from, and the negative's update is added
//            to, LDS (snapshot + fixed-point pending delta); the user's item list is tested from registers when it is short
//   TMODE    with LDSF: what the step updates (sgd_features_kernel).  0 = the rows only (v_u, v_i, w_i; the tables are a read-only
//            copy), 1 = the tables only (the table trainer: plain read-modify-write on the master copy, groups of a wavefront one
//            after the other), 2 = both (one group alone: the reference's sequential step)
template <int G, int KPL, bool SERIAL, bool FEAT, bool VU_REGS, bool FRESH, bool LDSF = false, bool HOT = false, bool WARPB = true,
          bool STRIPE = false, int TMODE = 0>
struct RowStep {
    const SgdArgs &a;
    const int sub;                   // lane index inside the group
    const int F;
    typedef typename TablePtr<LDSF>::type TabPtr;
    TabPtr t_v_uf, t_v_if, t_w_if;   // feature tables: global memory, or the workgroup's LDS copy (LDSF)
    static constexpr bool UPD_ROWS = !(FEAT && LDSF) || TMODE != 1;
    static constexpr bool UPD_TAB = FEAT && (!LDSF || TMODE == 2);
    // table trainer (TMODE 1): the step's table update is STAGED here -- [0] g * d_outer | [1, F] updated v_u | [F] updated
    // v_i - v_j | [P] x_uf[u] | [Q] x_if[i] - x_if[j] -- and applied by sgd_features_kernel, table row by table row
    lds_float *stage = nullptr;
    // LDS [n_hot, F] pending factor deltas, [n_hot] pending bias deltas, [n_hot] touch counters.  The pending sums are
    // 32-bit FIXED POINT: ds_add_u32 takes ~5 clocks per wave instruction where ds_add_f32 takes ~3 clocks per active
    // lane (tools/microbench/lds_atomic.hip), and every cross-lane shuffle of the workgroup queues behind them in the
    // same LDS pipeline.  The unit is 2^-24 * max(1, 10 * eta * max |sample_weight|), the range +-128 times that: a
    // pending sum is at most 64 touches of steps eta * sample_weight * |v|, i.e. < 0.3 at eta = 0.1 and unit weights.
    lds_int *hot_acc = nullptr;
    lds_int *hot_accw = nullptr;
    lds_int *hot_cnt = nullptr;
    float kHotScale = 16777216.0f, kHotUnit = 1.0f / 16777216.0f;
    __device__ __forceinline__ void hot_add(lds_int *p, float v) const {
        __hip_atomic_fetch_add(p, __float2int_rn(v * kHotScale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ float hot_take(lds_int *p) const {
        return (float)__hip_atomic_exchange(p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) * kHotUnit;
    }

    // negative stripe (STRIPE): [R] item of each row | [R, F+1] fp32 snapshot of v_i[item] and (last column) w_i[item] at window
    // start | [R, F+1] pending updates in the same 32-bit fixed point as the hot sums.  A row's value is snapshot + pending.
    lds_int *sn_item = nullptr;
    lds_float *sn_snap = nullptr;
    lds_int *sn_delta = nullptr;
    // [F+1] column sums of sn_delta.  The positive item of a step sits, with probability stripe_cover, in some other
    // workgroup's stripe and then carries pending pushes this workgroup cannot see; the workgroups run their windows in step
    // and stripes are uniform samples of the items, so the MEAN pending sum of this workgroup's own stripe rows (x the cover)
    // is what such an item is expected to carry.  Negative: published + own pending sum (exact, sequential inside the
    // workgroup); positive: published + expected pending sum.  Without the correction every pairwise utility is
    // overestimated by the positive's unseen downward pushes (log-likelihood -6 % against the sequential oracle at a 32-row
    // window on config 2; with it +0.1 %, profiles/r02_notes.md).
    // The BIAS column of the sums receives every push of the window with the same sign (about -eta * sample weight * d_outer each,
    // groups x window of them: 2048 at 64 groups x 32 rows, 6144 with 4-lane groups), which would wrap the +-128-unit range of the
    // pending sums; it is therefore kept in a unit kSumCoarse times coarser (range +-8192 x the step scale; the factor columns are
    // sums of signed terms ~100 times smaller and keep the fine unit).
    static constexpr float kSumCoarse = 64.0f;
    lds_int *sn_sum = nullptr;
    float sn_inv_rows = 0.0f;
    int sn_rows = 0;
    // the user's sorted item list, held across the lanes when it has at most 4 G entries (lane s: entries s, s+G, ...; -1 pads):
    // the membership test of a draw is then four compares and a ballot instead of a memory round trip
    int32_t ulist[4] = {-1, -1, -1, -1};
    bool ulist_ok = false;
    float user_scale = 1.0f;          // damping of this user's step (SgdArgs::user_cap), constant over a segment
    __device__ __forceinline__ void load_ulist(int64_t lo, int64_t hi) {
        user_scale = fminf(1.0f, a.user_cap / (float)(hi - lo));
        ulist_ok = (hi - lo) <= 4 * G;
        if (ulist_ok) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t idx = lo + sub + (int64_t)G * k;
                ulist[k] = idx < hi ? a.csr_items[idx] : -1;
            }
        }
    }
    __device__ __forceinline__ bool member(int64_t lo, int64_t hi, int32_t item) const {
        if (ulist_ok) {
            const bool f = (ulist[0] == item) | (ulist[1] == item) | (ulist[2] == item) | (ulist[3] == item);
            if constexpr (G == 64) return __ballot(f) != 0ull;
            else return group_ballot<G>(f) != 0u;
        }
        return is_member_group<G>(a.csr_items, lo, hi, item, sub);
    }
    __device__ __forceinline__ void members4(int64_t lo, int64_t hi, const int32_t (&c)[4], bool (&m)[4]) const {
        if (ulist_ok) {
#pragma unroll
            for (int q = 0; q < 4; ++q) m[q] = member(lo, hi, c[q]);
        } else {
            members4_group<G>(a.csr_items, lo, hi, c, m, sub);
        }
    }
    // factor row + bias of item `it`.  An item of the workgroup's stripe (row `srow` >= 0) has two views:
    //   screening (fresh = false): the snapshot taken when the window started -- an LDS read; WARP examines ~20 candidates per
    //       update this way;
    //   published (fresh = true): memory as of now (L1 bypassed) -- everything every workgroup has published, like the view
    //       every step has of its POSITIVE item.  Used for the negative that is actually stepped.
    // Both include the workgroup's own pending sum of the row.  The positive item's pending pushes sit unseen in some other
    // workgroup's LDS, and a step that saw its negative's pending pushes but not its positive's would overestimate every
    // pairwise utility (measured: log-likelihood -6 % against the sequential oracle at a 32-row window, profiles/r02_notes.md):
    // the positive's view therefore carries the stripe's MEAN pending sum (sn_sum, operator()).  The alternatives that were
    // measured (snapshot views, no own sums, biases published at once, reads through the atomic unit) are in the notes; the
    // kernel compiles the chosen one only.
    __device__ __forceinline__ void fetch_item(int32_t it, int srow, float (&v)[KPL], float &w, bool fresh = true) const {
        if constexpr (STRIPE) {
            if (srow >= 0) {
                const int base = srow * (F + 1);
                const float own = kHotUnit;
                if (!fresh) {
#pragma unroll
                    for (int k = 0; k < KPL; ++k)
                        v[k] = dword_ok(k) ? sn_snap[base + dword_f(k)] + (float)sn_delta[base + dword_f(k)] * own : 0.0f;
                    w = sn_snap[base + F] + (float)sn_delta[base + F] * own;
                } else {
#pragma unroll
                    for (int k = 0; k < KPL; ++k) v[k] = dword_ok(k) ? load_f32<true>(a.v_i + (size_t)it * F + dword_f(k)) : 0.0f;
                    w = load_f32<true>(a.w_i + (size_t)it * a.w_stride);
#pragma unroll
                    for (int k = 0; k < KPL; ++k)
                        if (dword_ok(k)) v[k] += (float)sn_delta[base + dword_f(k)] * own;
                    w += (float)sn_delta[base + F] * own;
                }
                return;
            }
        }
        load_row<FRESH>(a.v_i + (size_t)it * F, v);
        w = load_f32<FRESH>(a.w_i + (size_t)it * a.w_stride);
    }
    // raw draw -> candidate item (and its stripe row)
    __device__ __forceinline__ int32_t draw_item(uint32_t raw, int &srow, uint32_t attempt) const {
        if (STRIPE && sn_rows > 0 && attempt < RFM_STRIPE_ATTEMPTS) {
            srow = (int)rfm_draw_to_item(raw, (uint32_t)sn_rows);
            return sn_item[srow];
        } else {
            srow = -1;
            return (int32_t)rfm_draw_to_item(raw, (uint32_t)a.n_items);
        }
    }

    __device__ __forceinline__ RowStep(const SgdArgs &args, int sub_, TabPtr v_uf, TabPtr v_if, TabPtr w_if)
        : a(args), sub(sub_), F(STRIPE ? G * KPL : args.n_factors), t_v_uf(v_uf), t_v_if(v_if), t_w_if(w_if) {}
    // (stripe launches are planned for full factor rows only: the factor count is the compile-time constant G * KPL there)

    __device__ __forceinline__ int dword_f(int k) const { return sub + G * k; }
    // (stripe launches are planned for FULL factor rows only, F == G * KPL: no per-dword predicate -- a v_cmp, an exec-mask
    //  save and a branch around every load, atomic and LDS update of the row loop otherwise)
    __device__ __forceinline__ bool dword_ok(int k) const { return STRIPE || dword_f(k) < F; }

    template <bool FR>
    __device__ __forceinline__ void load_row(const float *base, float (&r)[KPL]) const {
#pragma unroll
        for (int k = 0; k < KPL; ++k) r[k] = dword_ok(k) ? load_f32<FR>(base + dword_f(k)) : 0.0f;
    }

    __device__ __forceinline__ void zero(float (&r)[KPL]) const {
#pragma unroll
        for (int k = 0; k < KPL; ++k) r[k] = 0.0f;
    }

    // A dense feature vector of one user / item, held across the G lanes of the group (lane s keeps entries s, s+G, ...,
    // at most MAXR of them) so that the loops over features read registers through shuffles instead of re-loading the
    // vector from memory five times per step.  Vectors longer than G*MAXR are read from memory (`mem`).
    static constexpr int MAXR = 4;
    struct XV { float r[MAXR]; const float *mem; int n; };

    __device__ __forceinline__ void xload(const float *x, int n, XV &v) const {
        v.mem = x; v.n = n;
#pragma unroll
        for (int k = 0; k < MAXR; ++k) v.r[k] = (sub + G * k < n) ? x[sub + G * k] : 0.0f;
    }

    // fn(p, x[p]) for every p with x[p] != 0, in index order; x[p] is group-uniform.  The non-zero positions of each
    // register slot come from one ballot, so the loop runs once per NON-ZERO entry (dense 0/1 tag vectors are mostly zero)
    // and the slot index stays a compile-time constant (the vector stays in registers).
    template <class Fn>
    __device__ __forceinline__ void xfor(const XV &v, Fn &&fn) const {
        if (v.n <= G * MAXR) {
            const int lane = threadIdx.x & 63;
            const int base = lane - sub;
#pragma unroll
            for (int k = 0; k < MAXR; ++k) {
                unsigned long long m = __ballot(v.r[k] != 0.0f);
                if constexpr (G < 64) m = (m >> base) & ((1ull << G) - 1ull);
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    fn(k * G + b, __shfl(v.r[k], base + b));
                }
            }
        } else {
            for (int p = 0; p < v.n; ++p) {
                const float x = v.mem[p];
                if (x != 0.0f) fn(p, x);
            }
        }
    }

    // fn(q, xa[q], xb[q]) for the entries q = sub, sub+G, ... this lane owns (two vectors of the same length)
    template <class Fn>
    __device__ __forceinline__ void xown2(const XV &va, const XV &vb, Fn &&fn) const {
        if (va.n <= G * MAXR) {
#pragma unroll
            for (int k = 0; k < MAXR; ++k)
                if (sub + G * k < va.n) fn(sub + G * k, va.r[k], vb.r[k]);
        } else {
            for (int q = sub; q < va.n; q += G) fn(q, va.mem[q], vb.mem[q]);
        }
    }

    // acc[f] = sum_r x[r] * table[r, f]   (feature projection into factor space, this lane's dwords)
    __device__ __forceinline__ void project(const XV &x, TabPtr table, float (&acc)[KPL]) const {
        zero(acc);
        xfor(x, [&](int r, float xr) {
            if (xr == 0.0f) return;       // zero entries contribute nothing (and are skipped by the reference, :73,:81)
            TabPtr row = table + r * F;
#pragma unroll
            for (int k = 0; k < KPL; ++k)
                if (dword_ok(k)) acc[k] += xr * row[dword_f(k)];
        });
    }

    // compute_ui_utility (rankfm/_rankfm.pyx:48-89) for item `it` given the user-side registers:
    //   w_i[it] + sum_q x_if[it,q] w_if[q] + sum_f [ (vu_f + A_f) * vi_f + B_f(it) * vu_f ]
    // A = x_uf[u] . v_uf  (user-feature projection), B(it) = x_if[it] . v_if  (item-feature projection)
    __device__ __forceinline__ float utility(const float (&vu)[KPL], const float (&A)[KPL], int32_t it, float (&vi)[KPL],
                                             float (&B)[KPL], float &wi, int slot = -1, const XV *xit = nullptr, int srow = -1,
                                             bool fresh = true) const {
        fetch_item(it, srow, vi, wi, fresh);
        if constexpr (HOT) {
            if (slot >= 0) {      // the workgroup's own pending updates of a hot row are part of its view of the row
#pragma unroll
                for (int k = 0; k < KPL; ++k)
                    if (dword_ok(k)) vi[k] += (float)hot_acc[slot * F + dword_f(k)] * kHotUnit;
                wi += (float)hot_accw[slot] * kHotUnit;
            }
        }
        float part = 0.0f, scalar = 0.0f;
        if constexpr (FEAT) {
            if (a.has_if) {
                project(*xit, t_v_if, B);
                // sum_q x_if[it,q] * w_if[q]: lanes split q, one more group reduction (the reference adds term by term)
                float sc = 0.0f;
                xown2(*xit, *xit, [&](int q, float x, float) { sc += x * t_w_if[q]; });
                scalar = group_sum<G>(sc);
            } else {
                zero(B);
            }
#pragma unroll
            for (int k = 0; k < KPL; ++k) part += (vu[k] + A[k]) * vi[k] + B[k] * vu[k];
        } else {
#pragma unroll
            for (int k = 0; k < KPL; ++k) part += vu[k] * vi[k];
        }
        return wi + scalar + group_sum<G>(part);
    }

    // draw the next unobserved item for the user (rankfm/_rankfm.pyx:250-253)
    __device__ __forceinline__ int32_t next_negative(int64_t lo, int64_t hi, uint32_t row_key, uint32_t &attempt, int &srow) const {
        int32_t j = 0;
        srow = -1;
        if (SERIAL && a.rng == 0 /* RFM_RNG_MT19937 */) {
            if (sub == 0) {
                do { j = (int32_t)(mt_next_global(a.mt_state) % (uint32_t)a.n_items); } while (is_member(a.csr_items, lo, hi, j));
            }
            j = __shfl(j, (threadIdx.x & 63) - sub);
        } else {
            for (;;) {
                j = draw_item(rfm_draw(row_key, attempt), srow, attempt);
                ++attempt;
                if (!member(lo, hi, j)) break;
                if (attempt >= kMaxAttempts) { if (sub == 0) atomicOr(a.error_flags, 1u); break; }
            }
        }
        return j;
    }

    // the positive item's row, bias and step scale fetched ahead of the row's turn (segments kernel, STRIPE): rows of a segment
    // depend on each other only through v_u, which lives in registers, so the next row's gathers overlap the current row
    struct PosRow { float v[KPL]; float w, scale; };
    __device__ __forceinline__ void prefetch_pos(int32_t it, PosRow &p) const {
        load_row<FRESH>(a.v_i + (size_t)it * F, p.v);
        if (STRIPE || a.scale_in_pad) {
            // lanes 0 / 1 of the group read dwords 0 / 1 of the item's line: bias and step scale in ONE request
            const float x = load_f32<FRESH>(a.w_i + (size_t)it * a.w_stride + (sub & 1));
            const int base = (threadIdx.x & 63) - sub;
            p.w = __shfl(x, base);
            p.scale = __shfl(x, base + 1);
        } else {
            p.w = load_f32<FRESH>(a.w_i + (size_t)it * a.w_stride);
            p.scale = a.pos_scale ? a.pos_scale[it] : 1.0f;
        }
    }

    // `vu` holds v_u[u] on entry; with VU_REGS it holds the updated row on exit
    __device__ __forceinline__ void operator()(uint32_t row_key, int32_t u, int32_t i, float sw, int64_t lo, int64_t hi,
                                               float (&vu)[KPL], double &ll_acc, unsigned &draw_acc, const PosRow *pre = nullptr) const {
        uint32_t attempt = 0;
        float A[KPL];
        XV xu, xi, xj, xc;
        if constexpr (FEAT) {
            if (a.has_uf) { xload(a.x_uf + (size_t)u * a.n_uf, a.n_uf, xu); project(xu, t_v_uf, A); }
            else zero(A);
            if (a.has_if) xload(a.x_if + (size_t)i * a.n_if, a.n_if, xi);
        }

        int slot = -1;
        float pos_scale_i = 1.0f;
        if constexpr (!SERIAL) {
            if (STRIPE || a.pos_scale) {
                // (stripe launches always carry bias and step scale in the item's padded line: scale 1 when nothing is damped)
                pos_scale_i = (STRIPE || pre) ? pre->scale : a.pos_scale[i];
                // a hot item's entry carries its accumulator slot above the scale (SgdArgs::hot_item).  EVERY instantiation decodes
                // it: the plan of a launch is shared by kernels with and without accumulators (the step producers of the features
                // kernel score hot items through this generic step -- undecoded, their staged steps were up to ~130 x too long)
                if (pos_scale_i >= 2.0f) {
                    const int sl = (int)(pos_scale_i * 0.5f) - 1;
                    pos_scale_i -= 2.0f * (float)(sl + 1);
                    if constexpr (HOT) slot = sl;
                }
            }
        }
        float vi[KPL], Bi[KPL], wi;
        float vj[KPL], Bj[KPL], wj = 0.0f;
        float min_pu = 1e6f;
        int32_t j = -1;
        int jrow = -1;                // stripe row of the chosen negative (STRIPE)
        int sampled = 0;
        float ut_ui = 0.0f;
        int s = 1;
        bool done = false;
        constexpr bool BATCH_WARP = !SERIAL && !FEAT && WARPB;
        // BPR instantiation of the feature kernel: the one negative does not depend on any score, and both the pairwise
        // utility and the gradients need the item-feature terms only as DIFFERENCES, so x_if[i] - x_if[j] is projected
        // once instead of x_if[i] and x_if[j] separately:
        //   pu = (w_i - w_j) + (x_i - x_j).w_if + <v_u + A, v_i - v_j> + <(x_i - x_j).v_if, v_u>
        // (the reference's ut_ui - ut_uj, :239 and :256-257, regrouped; Bi then holds B(i) - B(j) and Bj zero)
        constexpr bool BPRF = FEAT && LDSF && !WARPB;
        if constexpr (BPRF) {
            int srow_unused;
            j = next_negative(lo, hi, row_key, attempt, srow_unused);
            sampled = 1;
            load_row<FRESH>(a.v_i + (size_t)i * F, vi);
            load_row<FRESH>(a.v_i + (size_t)j * F, vj);
            wi = load_f32<FRESH>(a.w_i + (size_t)i * a.w_stride);
            wj = load_f32<FRESH>(a.w_i + (size_t)j * a.w_stride);
            float scalar = 0.0f;
            zero(Bi);
            zero(Bj);
            if (a.has_if) {
                xload(a.x_if + (size_t)j * a.n_if, a.n_if, xj);
                float sc = 0.0f;
                if (a.n_if <= G * MAXR) {
                    XV dxv = xi;
#pragma unroll
                    for (int k = 0; k < MAXR; ++k) dxv.r[k] = xi.r[k] - xj.r[k];
                    project(dxv, t_v_if, Bi);
#pragma unroll
                    for (int k = 0; k < MAXR; ++k)
                        if (sub + G * k < a.n_if) sc += dxv.r[k] * t_w_if[sub + G * k];
                } else {
                    float Bn[KPL];
                    project(xi, t_v_if, Bi);
                    project(xj, t_v_if, Bn);
#pragma unroll
                    for (int k = 0; k < KPL; ++k) Bi[k] -= Bn[k];
                    xown2(xi, xj, [&](int q, float xa, float xb) { sc += (xa - xb) * t_w_if[q]; });
                }
                scalar = group_sum<G>(sc);
            }
            float part = 0.0f;
#pragma unroll
            for (int k = 0; k < KPL; ++k) part += (vu[k] + A[k]) * (vi[k] - vj[k]) + Bi[k] * vu[k];
            min_pu = (wi - wj) + scalar + group_sum<G>(part);
        } else {
        if (STRIPE && pre) {
            // (STRIPE has no features: the utility is bias + dot product, on the prefetched row)
#pragma unroll
            for (int k = 0; k < KPL; ++k) vi[k] = pre->v[k];
            wi = pre->w;
            if constexpr (HOT) {
                if (slot >= 0) {
#pragma unroll
                    for (int k = 0; k < KPL; ++k)
                        if (dword_ok(k)) vi[k] += (float)hot_acc[slot * F + dword_f(k)] * kHotUnit;
                    wi += (float)hot_accw[slot] * kHotUnit;
                }
            }
            if (sn_rows > 0) {
                const float c = kHotUnit * sn_inv_rows;
#pragma unroll
                for (int k = 0; k < KPL; ++k)
                    if (dword_ok(k)) vi[k] += (float)sn_sum[dword_f(k)] * c;
                wi += (float)sn_sum[F] * (c * kSumCoarse);
            }
            float part = 0.0f;
#pragma unroll
            for (int k = 0; k < KPL; ++k) part += vu[k] * vi[k];
            ut_ui = wi + group_sum<G>(part);
        } else
        ut_ui = utility(vu, A, i, vi, Bi, wi, slot, &xi);    // :239

        // WARP sampling loop (:244-264); BPR is max_samples == 1
        // first draw (all of BPR): one candidate at a time
        for (; s <= ((BATCH_WARP || (!SERIAL && !FEAT && !WARPB)) ? 1 : a.max_samples); ++s) {
            int crow;
            const int32_t cand = next_negative(lo, hi, row_key, attempt, crow);
            float vc[KPL], Bc[KPL], wc;
            if constexpr (FEAT) { if (a.has_if) xload(a.x_if + (size_t)cand * a.n_if, a.n_if, xc); }
            // (stripes: BPR steps its one candidate -> exact view; WARP screens candidates on the snapshot)
            const float pu = ut_ui - utility(vu, A, cand, vc, Bc, wc, -1, &xc, crow, !(STRIPE && WARPB));   // :256-257
            sampled = s;
            if (pu < min_pu || j < 0) {                                   // :259-261 (j < 0: keep a valid index under NaN)
                if (pu < min_pu) min_pu = pu;
                j = cand; wj = wc; jrow = crow;
                if constexpr (FEAT) xj = xc;
#pragma unroll
                for (int k = 0; k < KPL; ++k) { vj[k] = vc[k]; if constexpr (FEAT) Bj[k] = Bc[k]; }
            }
            if (pu < kMargin) { done = true; break; }                     // :263-264
        }
        }
        if constexpr (BATCH_WARP) {
            // Later draws four at a time: the draw stream is keyed by (row, attempt), so looking ahead is free.  Four raw
            // draws are checked against the user's list in one pass, the survivors' rows are fetched together and then
            // examined IN DRAW ORDER with the reference's rule (first violator stops; draws after it are discarded).
            s = 2;
            while (!done && s <= a.max_samples) {
                int32_t c[4];
                int crow[4];
                bool mem[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) c[q] = draw_item(rfm_draw(row_key, attempt + q), crow[q], attempt + q);
                attempt += 4;
                members4(lo, hi, c, mem);
                // rows are fetched NB at a time: four at KPL <= 6; two at KPL >= 8, where four rows of registers spill and two
                // rows are as many requests in flight as four rows at KPL = 4
                constexpr int NB = KPL >= 8 ? 2 : 4;
#pragma unroll
                for (int q0 = 0; q0 < 4; q0 += NB) {
                    if (done) break;
                    float vc[NB][KPL], wc[NB], part[NB];
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        part[q] = 0.0f;
                        wc[q] = 0.0f;
                        if (!mem[q0 + q]) fetch_item(c[q0 + q], crow[q0 + q], vc[q], wc[q], false);
                    }
#pragma unroll
                    for (int q = 0; q < NB; ++q)
                        if (!mem[q0 + q]) {
#pragma unroll
                            for (int k = 0; k < KPL; ++k) part[q] += vu[k] * vc[q][k];
                        }
#pragma unroll
                    for (int q = 0; q < NB; ++q) part[q] = group_sum<G>(part[q]);
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        if (done || mem[q0 + q] || s > a.max_samples) continue;
                        const float pu = ut_ui - (wc[q] + part[q]);
                        sampled = s;
                        ++s;
                        if (pu < min_pu) {
                            min_pu = pu; j = c[q0 + q]; wj = wc[q]; jrow = crow[q0 + q];
#pragma unroll
                            for (int k = 0; k < KPL; ++k) vj[k] = vc[q][k];
                        }
                        if (pu < kMargin) done = true;
                    }
                }
                if (attempt >= kMaxAttempts) { if (sub == 0) atomicOr(a.error_flags, 1u); break; }
            }
        }
        if constexpr (STRIPE && WARPB) {
            // the negative that was chosen on the snapshot is stepped on its exact view: row, bias and pairwise utility again
            if (jrow >= 0) {
                fetch_item(j, jrow, vj, wj, true);
                float part = 0.0f;
#pragma unroll
                for (int k = 0; k < KPL; ++k) part += vu[k] * vj[k];
                min_pu = ut_ui - (wj + group_sum<G>(part));
            }
        }
        const float pu = min_pu;                                          // :267-268
        // Hogwild step damping is a property of the ITEM, whichever side of the pair it is on: the chosen negative's step takes its
        // item's scale too.  (Rounds 1-3 scaled the positive's step only: that moves the fixed point of a hot item's bias -- its upward
        // pushes weigh less than its downward ones -- and alone accounted for the whole +1.9 % log-likelihood / +2.7 % |w_i| deviation
        // of config 3 from the reference algorithm; with both sides scaled the sequential stand-in sits within 0.01 % / 0.1 %,
        // profiles/r04_notes.md.)  Same line as the bias just read (padded table) or the plan's scale array.
        float neg_scale_j = 1.0f;
        if constexpr (!SERIAL) {
            if (STRIPE || a.pos_scale) {
                float sc = (STRIPE || a.scale_in_pad) ? a.w_i[(size_t)j * a.w_stride + 1] : a.pos_scale[j];
                if (sc >= 2.0f) sc -= 2.0f * floorf(sc * 0.5f);           // (a hot item's entry carries its slot above the scale)
                neg_scale_j = sc;
            }
        }
        const float multiplier = a.multiplier[sampled];                   // :269 (integer division inside the log)
        float log_sig, d_outer;
        sigmoid_terms(pu, log_sig, d_outer);                              // :270, :276
        if (UPD_ROWS && sub == 0) { ll_acc += (double)log_sig; draw_acc += (unsigned)sampled; }
        const float g = sw * multiplier;
        const float eta = a.eta, reg_a = a.reg_a, reg_b = a.reg_b;
        float eta_u = eta, eta_i = eta, eta_f = eta;
        const float eta_j = a.damp_positive_only ? eta : eta * neg_scale_j;
        if constexpr (!SERIAL) {
            if constexpr (STRIPE) eta_u = eta * user_scale;          // (per segment: load_ulist)
            else eta_u = eta * fminf(1.0f, a.user_cap / (float)(hi - lo));
            eta_i = eta * pos_scale_i;
            if constexpr (!LDSF) eta_f = eta * a.feat_scale;
        }
        float nvu[KPL], dij[KPL];     // updated v_u, updated (v_i - v_j) (feature paths)
        if constexpr (STRIPE && !SERIAL && !FEAT && VU_REGS) {
            // The same arithmetic as the generic code below, arranged for the stripe instantiations: every delta first, then ONE
            // branch per publication target (hot slot or atomics for the positive, stripe row or atomics for the negative)
            // instead of one per dword.
            float d_i[KPL], d_j[KPL];
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                const float g_u = vi[k] - vj[k];                                     // :292
                const float g_i = vu[k];                                             // :293-294 (d_v_j = -d_v_i)
                const float d_u = eta_u * (g * (d_outer * g_u) - reg_a * vu[k]);     // :308
                d_i[k] = eta_i * (g * (d_outer * g_i) - reg_a * vi[k]);              // :309
                d_j[k] = eta_j * (g * (d_outer * -g_i) - reg_a * vj[k]);             // :310
                vu[k] += d_u;
            }
            const float dwi = eta_i * (g * (d_outer * 1.0f) - reg_a * wi);           // :279
            const float dwj = eta_j * (g * (d_outer * -1.0f) - reg_a * wj);         // :280
            if (HOT && slot >= 0) {
#pragma unroll
                for (int k = 0; k < KPL; ++k) hot_add(hot_acc + slot * F + dword_f(k), d_i[k]);
                if (sub == 0) hot_add(hot_accw + slot, dwi);
            } else {
                float *pv = a.v_i + (size_t)i * F + sub;
#pragma unroll
                for (int k = 0; k < KPL; ++k) atomic_add_f32(pv + G * k, d_i[k]);
                if (sub == 0) atomic_add_f32(a.w_i + (size_t)i * a.w_stride, dwi);
            }
            if (jrow >= 0) {
                lds_int *pd = sn_delta + jrow * (F + 1) + sub, *ps = sn_sum + sub;
#pragma unroll
                for (int k = 0; k < KPL; ++k) {
                    const int q = __float2int_rn(d_j[k] * kHotScale);
                    __hip_atomic_fetch_add(pd + G * k, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(ps + G * k, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (sub == 0) {
                    const int q = __float2int_rn(dwj * kHotScale);
                    __hip_atomic_fetch_add(sn_delta + jrow * (F + 1) + F, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(sn_sum + F, __float2int_rn(dwj * (kHotScale * (1.0f / kSumCoarse))), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                float *pv = a.v_i + (size_t)j * F + sub;
#pragma unroll
                for (int k = 0; k < KPL; ++k) atomic_add_f32(pv + G * k, d_j[k]);
                if (sub == 0) atomic_add_f32(a.w_i + (size_t)j * a.w_stride, dwj);
            }
        } else {
        // item biases (:279-280) -- one lane per group
        if (UPD_ROWS && sub == 0) {
            const float dwi = eta_i * (g * (d_outer * 1.0f) - reg_a * wi);
            const float dwj = eta_j * (g * (d_outer * -1.0f) - reg_a * wj);
            if (HOT && slot >= 0) hot_add(hot_accw + slot, dwi);
            else apply_f32<SERIAL>(a.w_i + (size_t)i * a.w_stride, wi, dwi);
            if (STRIPE && jrow >= 0) {
                hot_add(sn_delta + jrow * (F + 1) + F, dwj);
                hot_add(sn_sum + F, dwj * (1.0f / kSumCoarse));
            } else apply_f32<SERIAL>(a.w_i + (size_t)j * a.w_stride, wj, dwj);
        }

        // factor updates (:289-326), this lane's dwords
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
            float g_u = vi[k] - vj[k];                                    // :292
            float g_i = vu[k];                                            // :293-294 (d_v_j = -d_v_i)
            if constexpr (FEAT) { g_i += A[k]; g_u += Bi[k] - Bj[k]; }   // :297-305
            const float d_u = eta_u * (g * (d_outer * g_u) - reg_a * vu[k]);   // :308
            const float d_i = eta_i * (g * (d_outer * g_i) - reg_a * vi[k]);   // :309
            const float d_j = eta_j * (g * (d_outer * -g_i) - reg_a * vj[k]);  // :310
            nvu[k] = vu[k] + d_u;
            dij[k] = (vi[k] + d_i) - (vj[k] + d_j);
            if (UPD_ROWS && dword_ok(k)) {
                const int f = dword_f(k);
                if constexpr (!VU_REGS) apply_f32<SERIAL>(a.v_u + (size_t)u * F + f, vu[k], d_u);
                if (HOT && slot >= 0) hot_add(hot_acc + slot * F + f, d_i);
                else apply_f32<SERIAL>(a.v_i + (size_t)i * F + f, vi[k], d_i);
                if (STRIPE && jrow >= 0) { hot_add(sn_delta + jrow * (F + 1) + f, d_j); hot_add(sn_sum + f, d_j); }
                else apply_f32<SERIAL>(a.v_i + (size_t)j * F + f, vj[k], d_j);
            }
        }
        if constexpr (VU_REGS && UPD_ROWS) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) vu[k] = nvu[k];
        }
        }
        if constexpr (HOT) {
            if (slot >= 0) {
                // every hot_period-th toucher of the slot publishes what the workgroup has accumulated for it
                // (a keyed coin with probability 1 / period instead of a shared counter: no LDS round trip on the row's path)
                if (__umulhi(rfm_mix32(row_key ^ 0x7A5C3B1DU), (uint32_t)a.hot_period[slot]) == 0u) {     // probability 1 / period
                    RFM_COLD_ARGS(c, !STRIPE)
#pragma unroll
                    for (int k = 0; k < KPL; ++k) {
                        if (!dword_ok(k)) continue;
                        const float d = hot_take(hot_acc + slot * F + dword_f(k));
                        if (d != 0.0f)
                            atomic_add_f32(c.hot_direct ? a.v_i + (size_t)i * F + dword_f(k)
                                                        : c.hot_bins_v + ((size_t)(blockIdx.x % kHotBins) * c.n_hot + slot) * F + dword_f(k), d);
                    }
                    if (sub == 0) {
                        const float d = hot_take(hot_accw + slot);
                        if (d != 0.0f) atomic_add_f32(c.hot_direct ? a.w_i + (size_t)i * a.w_stride : c.hot_bins_w + (size_t)(blockIdx.x % kHotBins) * c.n_hot + slot, d);
                    }
                }
            }
        }

        if constexpr (FEAT && LDSF && TMODE == 1) {
            if (sub == 0) stage[0] = g * d_outer;
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                if (!dword_ok(k)) continue;
                stage[1 + dword_f(k)] = nvu[k];
                stage[1 + F + dword_f(k)] = dij[k];
            }
            lds_float *sx = stage + 1 + 2 * F;
            if (a.has_uf) {
                if (a.n_uf <= G * MAXR) {
#pragma unroll
                    for (int k = 0; k < MAXR; ++k)
                        if (sub + G * k < a.n_uf) sx[sub + G * k] = xu.r[k];
                } else for (int q = sub; q < a.n_uf; q += G) sx[q] = xu.mem[q];
            }
            sx += a.n_uf;
            if (a.has_if) {
                if (a.n_if <= G * MAXR) {
#pragma unroll
                    for (int k = 0; k < MAXR; ++k)
                        if (sub + G * k < a.n_if) sx[sub + G * k] = xi.r[k] - xj.r[k];
                } else for (int q = sub; q < a.n_if; q += G) sx[q] = xi.mem[q] - xj.mem[q];
            }
        }
        if constexpr (UPD_TAB) {
          {
            // item-feature weights (:283-286): every q shrinks, lanes split the q range.  (The reference updates them before the
            // factor loop; within one interaction the three tables do not read each other, so the order is immaterial.)
            if (a.has_if) {
                xown2(xi, xj, [&](int q, float xa, float xb) {
                    const float w = t_w_if[q];
                    apply_f32<SERIAL || LDSF>(t_w_if + q, w, eta_f * (g * (d_outer * (xa - xb)) - reg_b * w));
                });
            }
            // user-feature factors (:313-318): rows p with x_uf[u,p] != 0, using the UPDATED v_i[i]-v_i[j]
            if (a.has_uf) {
                xfor(xu, [&](int p, float xp) {
                    if (xp == 0.0f) return;
                    TabPtr trow = t_v_uf + p * F;
#pragma unroll
                    for (int k = 0; k < KPL; ++k) {
                        if (!dword_ok(k)) continue;
                        const float t = trow[dword_f(k)];
                        apply_f32<SERIAL || LDSF>(trow + dword_f(k), t, eta_f * (g * (d_outer * (xp * dij[k])) - reg_b * t));
                    }
                });
            }
            // item-feature factors (:321-326): rows q with x_if[i,q] != x_if[j,q], using the UPDATED v_u[u]
            if (a.has_if) {
                XV dxv = xi;                      // x_if[i] - x_if[j], same distribution over the lanes
#pragma unroll
                for (int k = 0; k < MAXR; ++k) dxv.r[k] = xi.r[k] - xj.r[k];
                auto body = [&](int q, float dx) {
                    if (dx == 0.0f) return;
                    TabPtr trow = t_v_if + q * F;
#pragma unroll
                    for (int k = 0; k < KPL; ++k) {
                        if (!dword_ok(k)) continue;
                        const float t = trow[dword_f(k)];
                        apply_f32<SERIAL || LDSF>(trow + dword_f(k), t, eta_f * (g * (d_outer * (dx * nvu[k])) - reg_b * t));
                    }
                };
                if (dxv.n <= G * MAXR) xfor(dxv, body);
                else for (int q = 0; q < dxv.n; ++q) body(q, xi.mem[q] - xj.mem[q]);
            }
          }
        }
    }
};

// The shader clock a launch actually ran at (rfm_fit_report.shader_mhz): thread 0 of workgroup 0 -- resident from the launch's first
// microsecond to (nearly) its last -- stamps the constant 100 MHz wall clock and the shader cycle counter when it starts and when it
// leaves.  The same binary runs 2.9 ... 3.9 ms on different boxes of the pool (profiles/r03_notes.md): without the clock next to a
// timing, round-to-round comparisons inside that spread are noise.
__device__ __forceinline__ void stamp_clock(const SgdArgs &a, int which) {
    if (a.sclk && blockIdx.x == 0 && threadIdx.x == 0) {
        a.sclk[which] = wall_clock64();
        a.sclk[2 + which] = (unsigned long long)clock64();
    }
}

// wavefront reduction of the log-likelihood / draw counters, one atomic each per wavefront
__device__ __forceinline__ void flush_counters(const SgdArgs &a, double ll_acc, unsigned draw_acc) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        ll_acc += __shfl_xor(ll_acc, m);
        draw_acc += __shfl_xor(draw_acc, m);
    }
    if ((threadIdx.x & 63) == 0) {
        if (ll_acc != 0.0) unsafeAtomicAdd(a.ll, ll_acc);
        if (draw_acc) atomicAdd(a.draws, (unsigned long long)draw_acc);
    }
}

// Dynamic segment order.  With a static stride (group g walks order positions g, g + n_groups, ...) a group's share of an epoch is
// ~12 segments of 8 ... 32 rows: on config 2 the busiest group has 1.20 x the mean rows and the launch waits for it (utilisation
// 0.84 if every row cost the same; a greedy hand-out reaches 0.94).  So a group that finishes a segment takes the NEXT position of
// the epoch's keyed order from a counter.  The hand-out has two levels: a WORKGROUP draws chunks of kTicketChunk consecutive order
// positions from the launch's counter in memory (one returning atomic per chunk), its row groups take single positions out of the
// chunk through a counter in LDS.  (A returning memory-side atomic per SEGMENT was measured first and made config 2 7 % SLOWER:
// loads and returning atomics come back in order, so every vector load the wavefront issues behind the ticket request -- the rows of
// all four of its groups -- waits out the atomic's fabric round trip.  LDS atomics are counted separately and hold nothing up.)
// The chunk AFTER the current one is requested by whoever draws the first ticket of a chunk, so nobody waits for a chunk in the
// steady state.  The draws are keyed by CSR position and the segment order by position in the epoch's order, so which group works
// on a segment changes neither; the realised interleaving is closer to the order's own sequence than the static stride's.
constexpr int kTicketChunk = 16;
// chunks whose {base, number} a workgroup keeps in LDS at a time.  A row group holds at most one ticket it has not finished looking
// up, so the tickets "in the air" of a workgroup span at most (row groups per workgroup) + kTicketChunk positions -- 272 with 4-lane
// row groups -- and a ring of 32 chunks (512 tickets) can never be lapped.  (A ring of 4 was: when all 64 groups of a workgroup draw
// at once -- the launch's first tickets -- the opener of chunk 3 published chunk 4 over chunk 0's slot while the opener of chunk 0
// was still away fetching chunk 1, and that lane then waited for a chunk number that was gone: a hang, one run in three.)
constexpr int kTicketRing = 32;
constexpr int kTicketLdsWords = 1 + 2 * kTicketRing;
struct SegmentTickets {
    lds_int *q;               // LDS: [0] tickets handed out by this workgroup | [1, 1 + ring) chunk bases | [1 + ring, 1 + 2 ring) chunk numbers + 1
    // (thread 0, before the workgroup's first barrier)
    __device__ __forceinline__ void init_block(const SgdArgs &a) {
        q[0] = 0;
        for (int k = 0; k < kTicketRing; ++k) q[1 + kTicketRing + k] = 0;
        q[1] = (int)__hip_atomic_fetch_add(a.tickets, (unsigned)kTicketChunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q[1 + kTicketRing] = 1;
    }
    // the next order position of the launch for this group, or -1 when none is left (lane 0 of the group only; the caller broadcasts)
    __device__ __forceinline__ int64_t take(const SgdArgs &a) {
        const unsigned t = (unsigned)__hip_atomic_fetch_add(q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned c = t / kTicketChunk, o = t % kTicketChunk;
        // this ticket's own chunk FIRST (it was requested when chunk c - 1 was opened, tens of microseconds ago in the steady state) ...
        lds_int *tag = q + 1 + kTicketRing + (c % kTicketRing);
        unsigned spin = 0;
        while ((unsigned)__hip_atomic_load(tag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != c + 1u) {
            if (++spin > (1u << 22)) { atomicOr(a.error_flags, 16u); return -1; }      // (a hang guard, never observed)
            __builtin_amdgcn_s_sleep(1);
        }
        const int64_t p = a.pos_begin + (int64_t)(unsigned)__hip_atomic_load(q + 1 + (c % kTicketRing), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + (int64_t)o;
        // ... then the duty of a chunk's first ticket: request chunk c + 1 for those who come next
        if (o == 0) {
            const unsigned b = __hip_atomic_fetch_add(a.tickets, (unsigned)kTicketChunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(q + 1 + ((c + 1) % kTicketRing), (int)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(q + 1 + kTicketRing + ((c + 1) % kTicketRing), (int)(c + 2), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return p < a.pos_end ? p : -1;
    }
};

// ---------------------------------------------------------------------------------------------
// rows kernel: every wavefront walks the epoch's positions with a grid stride of (waves * rows-per-wave)
// ---------------------------------------------------------------------------------------------
template <int G, int KPL, bool SERIAL, bool FEAT>
__global__ void __launch_bounds__(256) sgd_rows_kernel(const SgdArgs a) {
    constexpr int RPW = SERIAL ? 1 : 64 / G;                    // interactions carried by one wavefront at a time
    const int lane = threadIdx.x & 63;
    const int grp = lane / G, sub = lane % G;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    int64_t n_groups = n_waves * RPW;                                        // groups that work (concurrency cap)
    if (!SERIAL && a.max_groups > 0 && a.max_groups < n_groups) n_groups = a.max_groups;
    const int64_t group = SERIAL ? 0 : wave * RPW + grp;
    const RowStep<G, KPL, SERIAL, FEAT, false, false> step(a, sub, a.v_uf, a.v_if, a.w_if);
    const int F = a.n_factors;

    double ll_acc = 0.0;
    unsigned draw_acc = 0;
    const bool works = SERIAL ? (grp == 0) : (group < n_groups);
    for (int64_t pos = a.pos_begin + group; __any(works && pos < a.pos_end); pos += n_groups) {
        const bool active = works && pos < a.pos_end;
        if (active) {
            const int64_t row = a.perm ? (int64_t)a.perm[pos]
                                       : (int64_t)rfm_perm((uint32_t)pos, (uint32_t)a.n_rows, a.perm_bits, a.epoch_key);
            const int32_t u = a.interactions[2 * row];                       // :233-235
            const int32_t i = a.interactions[2 * row + 1];
            const float sw = a.sample_weight[row];                           // :236
            const int64_t lo = a.csr_off[u], hi = a.csr_off[u + 1];
            float vu[KPL];
#pragma unroll
            for (int k = 0; k < KPL; ++k) vu[k] = (sub + G * k < F) ? a.v_u[(size_t)u * F + sub + G * k] : 0.0f;
            step(rfm_row_key(a.epoch_key, (uint32_t)row), u, i, sw, lo, hi, vu, ll_acc, draw_acc);
        }
        if constexpr (SERIAL) __threadfence_block();   // row r+1 must observe row r (cross-lane w_i / w_if reads)
    }
    flush_counters(a, ll_acc, draw_acc);
}

// One 64-byte line of the hot-row bins (see SgdArgs::hot_bins_v), swept by one wavefront: lanes = 4 bins x 16 dwords at a
// time, exchange with zero, sum over the bins, one atomic add of the total into the hot row.  Lines 0 .. n_hot*LPR-1 are
// 16-factor pieces of the hot rows (LPR = lines per row), the rest are 16 slots' biases each.
__device__ __forceinline__ int hot_lines(const SgdArgs &a) { return a.n_hot * ((a.n_factors + 15) / 16) + (a.n_hot + 15) / 16; }

__device__ __forceinline__ void hot_sweep_line(const SgdArgs &a, int line) {
    const int lane = threadIdx.x & 63, d16 = lane & 15, quad = lane >> 4;
    const int F = a.n_factors, lpr = (F + 15) / 16;
    const bool bias = line >= a.n_hot * lpr;
    float *src, *dst;
    size_t bin_stride;
    bool ok;
    if (!bias) {
        const int slot = line / lpr, f = (line % lpr) * 16 + d16;
        ok = f < F;
        src = a.hot_bins_v + (size_t)slot * F + f;
        bin_stride = (size_t)a.n_hot * F;
        dst = a.v_i + (size_t)a.hot_item[slot] * F + f;
    } else {
        const int slot = (line - a.n_hot * lpr) * 16 + d16;
        ok = slot < a.n_hot;
        src = a.hot_bins_w + slot;
        bin_stride = (size_t)a.n_hot;
        dst = a.w_i + (size_t)a.hot_item[ok ? slot : 0] * a.w_stride;
    }
    float acc = 0.0f;
    if (ok) {
#pragma unroll
        for (int b = 0; b < kHotBins; b += 4)
            acc += __hip_atomic_exchange(src + (size_t)(b + quad) * bin_stride, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if (quad == 0 && ok && acc != 0.0f) atomic_add_f32(dst, acc);
}

// drains the bins after a launch of the HOT kernel (one wavefront per line)
static __global__ void __launch_bounds__(256) hot_reduce_kernel(const SgdArgs a) {
    const int line = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (line < hot_lines(a)) hot_sweep_line(a, line);
}

// ---------------------------------------------------------------------------------------------
// segments kernel (production Hogwild): see the header comment.  Each group is a little state machine
//   [fetch segment + v_u] -> row, row, ... -> [write back v_u delta] -> next segment
// so the four groups of a wavefront stay busy although their segments differ in length.
// ---------------------------------------------------------------------------------------------
// With features the workgroup is 1024 threads (16 wavefronts): the feature tables are per-WORKGROUP replicas, and fewer,
// larger workgroups mean fewer replicas for the same number of interactions in flight.
// The HOT instantiation (no features) also uses 1024 threads: the hot-row accumulators are per workgroup, and fewer,
// larger workgroups combine more touches per publication at the same amount of unpublished work.
//
// The STRIPE instantiation (production, no features) also uses 1024 threads, one workgroup per CU, and most of the CU's LDS:
// the workgroup draws the negatives of a WINDOW of rows (stripe_window per group) from a STRIPE of stripe_rows items
// (include/rfm_rng.h) whose factor rows and biases it snapshots into LDS when the window starts.  Candidate rows are then LDS
// reads (WARP examines ~20 per update), the chosen negative's update is an LDS add into a fixed-point pending sum, and when
// the window ends every stripe row is published with ONE set of atomics however many updates it received -- with
// 64 groups x 32 rows on 256 stripe rows about eight.  That takes the negative item's 4 + 1 memory-side atomic requests per
// update (of ~10, the kernel's bound: DESIGN.md section 7) down to ~0.6, and the negative's row reads from 5 to ~0.6.
template <int G, int KPL, bool FRESH, bool HOT = false, bool WARPB = true, bool STRIPE = false>
__global__ void __launch_bounds__((HOT || STRIPE) ? 1024 : 256) sgd_segments_kernel(const SgdArgs a) {
    constexpr bool FEAT = false;        // (models with features run sgd_features_kernel)
    const int lane = threadIdx.x & 63;
    const int sub = lane % G;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    int64_t n_groups = ((int64_t)gridDim.x * blockDim.x) / G;
    if (a.max_groups > 0 && a.max_groups < n_groups) n_groups = a.max_groups;
    const int F = STRIPE ? G * KPL : a.n_factors;
    extern __shared__ __attribute__((aligned(16))) float lds_tables[];
    lds_float *lds = (lds_float *)lds_tables;
    typedef RowStep<G, KPL, false, false, true, FRESH, false, HOT, WARPB, STRIPE> Step;
    Step step(a, sub, a.v_uf, a.v_if, a.w_if);
    if constexpr (HOT) {
        // LDS: [n_hot * F] pending factor deltas | [n_hot] pending bias deltas | [n_hot] touch counters
        const int n_acc = a.n_hot * (F + 2);
        for (int k = threadIdx.x; k < n_acc; k += blockDim.x) lds_tables[k] = 0.0f;
        __syncthreads();
        step.hot_acc = (lds_int *)lds;
        step.hot_accw = (lds_int *)(lds + a.n_hot * F);
        step.hot_cnt = (lds_int *)(lds + a.n_hot * (F + 1));
    }
    if constexpr (HOT || STRIPE) {
        // steps scale with learning rate x sample weight: unit 2^-24 at the defaults (eta 0.1, weights <= 1)
        const float range = fmaxf(1.0f, __uint_as_float(*a.sw_max_bits) * a.eta * 10.0f);
        step.kHotScale = 16777216.0f / range;
        step.kHotUnit = range / 16777216.0f;
    }
    // negative stripe: [R] items | [R, F+1] snapshot | [R, F+1] pending, behind the hot-row accumulators
    const int R = STRIPE ? a.stripe_rows : 0, FS = F + 1;
    if constexpr (STRIPE) {
        lds_float *base = lds + (HOT ? a.n_hot * (F + 2) : 0);
        step.sn_item = (lds_int *)base;
        step.sn_snap = base + R;
        step.sn_delta = (lds_int *)(base + R + R * FS);
        step.sn_sum = (lds_int *)(base + R + 2 * R * FS);
        // mean pending sum of a random ITEM = mean over this stripe's rows x the chance that the item is in a stripe at all
        step.sn_inv_rows = a.stripe_cover / (float)R;
        step.sn_rows = R;
    }
    // window turn-over: every stripe row is published (one atomic per touched 64-byte segment, whatever the number of
    // updates it received) and, when work remains, replaced by the same row of the next stripe.  Row `slot` is handled by
    // one 16-lane group; loads bypass L1 (other workgroups' atomics must be seen).
    auto stripe_turn = [&](bool flush, bool load, uint32_t window) {
        const int gw = threadIdx.x / G, ngw = blockDim.x / G;
        const uint32_t start = load ? rfm_stripe_start(a.epoch_key, a.launch_index, blockIdx.x, gridDim.x, window, (uint32_t)R, (uint32_t)a.n_items) : 0u;
        if (load) for (int k = threadIdx.x; k < FS; k += blockDim.x) step.sn_sum[k] = 0;
        for (int slot = gw; slot < R; slot += ngw) {
            const int base = slot * FS;
            if (flush) {
                const int32_t it = step.sn_item[slot];
#pragma unroll
                for (int k = 0; k < KPL; ++k) {
                    const int f = sub + G * k;
                    if (f < F) {
                        const int d = step.sn_delta[base + f];
                        if (d != 0) atomic_add_f32(a.v_i + (size_t)it * F + f, (float)d * step.kHotUnit);
                    }
                }
                if (sub == 0) {
                    const int d = step.sn_delta[base + F];
                    if (d != 0) atomic_add_f32(a.w_i + (size_t)it * a.w_stride, (float)d * step.kHotUnit);
                }
            }
            if (load) {
                const int32_t it = (int32_t)rfm_stripe_item(a.epoch_key, start, (uint32_t)slot, (uint32_t)a.n_items, a.item_bits);
#pragma unroll
                for (int k = 0; k < KPL; ++k) {
                    const int f = sub + G * k;
                    if (f < F) {
                        if (WARPB) step.sn_snap[base + f] = load_f32<true>(a.v_i + (size_t)it * F + f);   // screening view
                        step.sn_delta[base + f] = 0;
                    }
                }
                if (sub == 0) {
                    if (WARPB) step.sn_snap[base + F] = load_f32<true>(a.w_i + (size_t)it * a.w_stride);
                    step.sn_delta[base + F] = 0;
                    step.sn_item[slot] = it;
                }
            }
        }
    };
    uint32_t window = 0;
    if constexpr (STRIPE) {
        if (R > 0) stripe_turn(false, true, 0);
        __syncthreads();
    }

    double ll_acc = 0.0;
    unsigned draw_acc = 0;
    const int lane_base = lane - sub;
    stamp_clock(a, 0);
    int64_t sp = a.pos_begin + (a.single_group ? 0 : group);      // position in the epoch's segment order
    const int64_t stride = a.single_group ? 1 : n_groups;
    bool active = sp < a.pos_end && (a.single_group ? group == 0 : group < n_groups);
    // dynamic segment order (SegmentTickets): stripe launches keep the static stride, their window schedule is a function of it
    const bool dynamic = !STRIPE && a.tickets != nullptr && !a.single_group;
    SegmentTickets tickets;
    __shared__ int s_ticket_q[kTicketLdsWords];
    tickets.q = (lds_int *)s_ticket_q;
    if constexpr (!STRIPE) {
        if (dynamic) {
            if (threadIdx.x == 0) tickets.init_block(a);
            __syncthreads();
            active = group < n_groups && a.pos_begin < a.pos_end;
            int64_t first = -1;
            if (active && sub == 0) first = tickets.take(a);
            sp = __shfl(first, lane_base);
            active = active && sp >= 0;
        }
    }
    bool have = false;
    int32_t u = 0, begin = 0, len = 0, t = 0, len_bits = 0;
    uint32_t seg_key = 0;
    int64_t lo = 0, hi = 0;
    float vu[KPL], vu0[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) vu[k] = vu0[k] = 0.0f;
    constexpr int SEGR = (kSegmentRows + G - 1) / G;     // rows of a segment per lane
    int32_t seg_item[SEGR], seg_pos[SEGR];
    float seg_sw[SEGR];
    typename Step::PosRow cur_pos, next_pos;
    // row t of the segment: register t / G of lane t % G (the register index is selected, not indexed: registers stay registers).
    // 16-lane groups are DPP rows: the registers are ROTATED one lane per processed row (seg_rotate), so the current row is
    // always in lane 0 and the next one in lane 1 of the selected register -- a row_share move, no LDS shuffle and no index math.
    auto seg_pick = [&](const int32_t (&r)[SEGR], int tt) {
        int32_t x = r[0];
#pragma unroll
        for (int k = 1; k < SEGR; ++k) x = ((unsigned)tt / G == (unsigned)k) ? r[k] : x;
        return x;
    };
    auto seg_get = [&](const int32_t (&r)[SEGR], int tt, bool next = false) {
        const int32_t x = seg_pick(r, tt);
        if constexpr (G == 16 && STRIPE) return next ? dpp_movi<0x151>(x) : dpp_movi<0x150>(x);      // row_share:1 / row_share:0
        else return __shfl(x, lane_base + (int)((unsigned)tt % G));
    };
    auto seg_getf = [&](const float (&r)[SEGR], int tt) {
        float x = r[0];
#pragma unroll
        for (int k = 1; k < SEGR; ++k) x = ((unsigned)tt / G == (unsigned)k) ? r[k] : x;
        if constexpr (G == 16 && STRIPE) return dpp_mov<0x150>(x);
        else return __shfl(x, lane_base + (int)((unsigned)tt % G));
    };
    auto seg_rotate = [&]() {
        if constexpr (G == 16 && STRIPE) {
#pragma unroll
            for (int k = 0; k < SEGR; ++k) {
                seg_item[k] = dpp_movi<0x12F>(seg_item[k]);       // row_ror:15: lane s takes lane s + 1
                seg_pos[k] = dpp_movi<0x12F>(seg_pos[k]);
                seg_sw[k] = dpp_mov<0x12F>(seg_sw[k]);
            }
        }
    };

    for (int iter = 0;; ++iter) {
        if constexpr (!FEAT && !STRIPE) { if (!__any(active)) break; }
        if constexpr (STRIPE) {
            // (stripe_rows = 0: the pipelined row loop alone -- draws over the whole catalogue, atomics per negative)
            if (R == 0) { if (!__any(active)) break; }
            // window boundary (workgroup-uniform): all of the window's LDS adds are done behind the barrier
            else if (iter > 0 && iter % a.stripe_window == 0) {
                const bool more = __syncthreads_or(active) != 0;
                // (one group alone is a sequential program: the previous row's atomics must have been performed before this
                // row reads the same addresses again)
                if (a.single_group) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                stripe_turn(true, false, 0);
                if (a.single_group) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                stripe_turn(false, more, ++window);
                __syncthreads();
                if (!more) break;
            }
        }
        if constexpr (HOT) {
            // bin sweeping duty (see SgdArgs::hot_bins_v): the wavefronts of a workgroup take turns, one turn per row; a turn
            // sweeps the workgroup's lines (at most four, else the host chose hot_direct).
            // (Round 4 tried the sweep in two halves -- the exchanges issued at the top of the row, their sum added at its bottom, one
            // line per turn -- to take the fabric round trip out of the sweeping wavefront's path: no faster once the segments are
            // handed out dynamically (2.89 against 2.79 ms on config 2), and the first epoch's log-likelihood moved from +0.6 % to
            // +3.3 % against the oracle, the hottest biases' bins being swept half as often; and a sweep WITHOUT returning atomics --
            // system-scope loads, then subtracting what was read -- diverged: a bin that reads as zero is not written by its sweeper,
            // its line stays in the sweeper's L2, and the memory-side atomics of the publishers in the other XCDs never invalidate
            // it.  profiles/r04_notes.md.)
            const int n_waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
            if (!a.hot_direct && iter % n_waves == wave) {
                RFM_COLD_ARGS(c, !STRIPE)                        // (the rarely executed parts read their arguments afresh: cold_args)
                for (int line = blockIdx.x; line < hot_lines(c); line += gridDim.x) hot_sweep_line(c, line);
            }
        }
        if (active && !have) {
            RFM_COLD_ARGS(c, !STRIPE)
            const uint32_t seg = rfm_perm((uint32_t)sp, (uint32_t)c.n_segments, c.seg_bits, c.epoch_key ^ 0x5bd1e995u);
            const int4 d = c.seg_desc[seg];
            u = d.x; begin = d.y; len = d.z;
            lo = c.csr_off[u]; hi = c.csr_off[u + 1];
            len_bits = (int32_t)rfm_perm_bits((uint32_t)len);
            seg_key = rfm_mix32(c.epoch_key ^ (seg * 0x9E3779B9u + 0x7F4A7C15u));
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                vu0[k] = (STRIPE || sub + G * k < F) ? load_f32<FRESH>(c.v_u + (size_t)u * F + sub + G * k) : 0.0f;
                vu[k] = vu0[k];
            }
            t = 0;
            have = true;
            step.load_ulist(lo, hi);
            if constexpr (STRIPE) {
                // the segment's rows in visiting order, held across the lanes (row t in lane t % G, register t / G): item,
                // sample weight and CSR position come out of registers for the rest of the segment
#pragma unroll
                for (int k = 0; k < SEGR; ++k) {
                    const int tt = sub + G * k;
                    seg_pos[k] = tt < len ? begin + (int32_t)rfm_perm((uint32_t)tt, (uint32_t)len, (uint32_t)len_bits, seg_key) : begin;
                    seg_item[k] = a.csr_items[seg_pos[k]];
                    seg_sw[k] = a.sw_csr[seg_pos[k]];
                }
                step.prefetch_pos(seg_get(seg_item, 0), next_pos);
            }
        }
        if (active) {
            int32_t pos, i;
            float sw;
            if constexpr (STRIPE) {
                pos = seg_get(seg_pos, t); i = seg_get(seg_item, t); sw = seg_getf(seg_sw, t);
                cur_pos = next_pos;
                // (one group alone is a sequential program: a repeated (user, item) row must see the previous row's update of
                // the same item, so nothing is fetched ahead there)
                if (a.single_group) step.prefetch_pos(i, cur_pos);
                else if (t + 1 < len) step.prefetch_pos(seg_get(seg_item, t + 1, true), next_pos);     // overlaps this row
                seg_rotate();
                step(rfm_row_key(a.epoch_key, (uint32_t)pos), u, i, sw, lo, hi, vu, ll_acc, draw_acc, &cur_pos);
            } else {
                pos = begin + (int32_t)rfm_perm((uint32_t)t, (uint32_t)len, (uint32_t)len_bits, seg_key);
                i = a.csr_items[pos];
                sw = a.sw_csr[pos];
                step(rfm_row_key(a.epoch_key, (uint32_t)pos), u, i, sw, lo, hi, vu, ll_acc, draw_acc);
            }
            if (++t == len) {
                RFM_COLD_ARGS(c, !STRIPE)
                // one write-back per segment; other segments of a heavy user may be in flight, so add the delta
#pragma unroll
                for (int k = 0; k < KPL; ++k)
                    if (STRIPE || sub + G * k < F) atomic_add_f32(c.v_u + (size_t)u * F + sub + G * k, vu[k] - vu0[k]);
                have = false;
                bool stepped = false;
                if constexpr (!STRIPE) {
                    if (dynamic) {
                        int64_t nxt = -1;
                        if (sub == 0) nxt = tickets.take(c);
                        sp = __shfl(nxt, lane_base);
                        active = sp >= 0;
                        stepped = true;
                    }
                }
                if (!stepped) {
                    sp += stride;
                    active = sp < a.pos_end;
                }
            }
        }
    }
    if constexpr (HOT) {          // publish whatever is still pending
        __syncthreads();
        for (int k = threadIdx.x; k < a.n_hot * F; k += blockDim.x) {
            const float d = (float)step.hot_acc[k] * step.kHotUnit;
            if (d != 0.0f) atomic_add_f32(a.hot_direct ? a.v_i + (size_t)a.hot_item[k / F] * F + (k % F)
                                                       : a.hot_bins_v + (size_t)(blockIdx.x % kHotBins) * a.n_hot * F + k, d);
        }
        for (int k = threadIdx.x; k < a.n_hot; k += blockDim.x) {
            const float d = (float)step.hot_accw[k] * step.kHotUnit;
            if (d != 0.0f) atomic_add_f32(a.hot_direct ? a.w_i + (size_t)a.hot_item[k] * a.w_stride : a.hot_bins_w + (size_t)(blockIdx.x % kHotBins) * a.n_hot + k, d);
        }
    }
    flush_counters(a, ll_acc, draw_acc);
    stamp_clock(a, 1);
}

// ---------------------------------------------------------------------------------------------
// WARP kernel (production Hogwild, no features, 16-lane row groups): the candidate loop as a per-GROUP state machine.
//
// In sgd_segments_kernel a WARP row is one call of RowStep: its candidate loop (rankfm/_rankfm.pyx:244-264) runs to the row's end
// before the wavefront moves on, and the four row groups of a wavefront wait for the SLOWEST of their four rows.  The number of
// draws per row is anything but uniform -- on config 3 after a few epochs 33 % of the rows stop at their first draw and 36 % run to
// the cap of 50 (mean 22.4) -- so a wavefront spends the time of 11.6 candidate batches per row step where its rows need 5.8 on
// average (tools/warp_draw_histogram.py, profiles/r04_notes.md): half of the candidate phase is groups idling beside a long row.
// Here one loop iteration is ONE batch of item rows for every group, whatever it is doing: a group that starts a row gathers its
// positive item and the first NC - 1 candidates, a group in the middle of a row its next NC candidates; all rows of the wavefront's
// gather are in flight together, then every group examines what it fetched IN DRAW ORDER with the reference's rule (first
// violator stops, `min_index` tracking, `sampled` semantics, :247-264) and, when its row is finished, applies the update (:267-326,
// the arithmetic of RowStep) and moves on.  Rows of different lengths no longer hold each other up.
// Same draws (keyed by CSR position and attempt), same order inside a row, same update: the one-group mode is the sequential
// algorithm like sgd_segments_kernel's, and the Hogwild tests of configs 3 and 5 are the parity check.
// ---------------------------------------------------------------------------------------------
// FULL: the factor rows fill the lanes (F == G * KPL: 64 or 128 factors, ...): no per-dword bounds predicate anywhere -- the kernel is
// bound by its vector instructions (PMC: the SIMDs' vector ALUs are ~80 % busy on config 3), and every predicate is a compare, an
// exec-mask save and a branch around a load.
template <int G, int KPL, bool FRESH, bool HOT, bool FULL>
__global__ void __launch_bounds__(HOT ? 1024 : 256) sgd_warp_kernel(const SgdArgs a) {
    static_assert(G == 16, "the WARP state machine is written for 16-lane row groups");
    constexpr int NC = KPL >= 8 ? 2 : 4;                    // item rows a group gathers per iteration
    constexpr int kSweepEvery = 4;                          // (an iteration is a fraction of a row step: sweep the bins every 4th turn)
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, lane_base = lane - sub;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    int64_t n_groups = ((int64_t)gridDim.x * blockDim.x) / G;
    if (a.max_groups > 0 && a.max_groups < n_groups) n_groups = a.max_groups;
    const int F = FULL ? G * KPL : a.n_factors;
    auto ok = [&](int kk) { return FULL || sub + G * kk < F; };
    extern __shared__ __attribute__((aligned(16))) float lds_tables[];
    lds_float *lds = (lds_float *)lds_tables;
    typedef RowStep<G, KPL, false, false, true, FRESH, false, HOT, true, false> Step;       // draws, membership test, fixed-point hot sums
    Step step(a, sub, a.v_uf, a.v_if, a.w_if);
    // LDS: [n_hot * F] pending factor deltas | [n_hot] pending bias deltas | [n_hot] (unused) -- like the HOT segments kernel -- then
    // the WARP multipliers (:269) and the hot slots' publication periods, read on every update
    const int n_acc = HOT ? a.n_hot * (F + 2) : 0;
    const int n_mult = a.max_samples + 1 <= 256 ? a.max_samples + 1 : 0;
    lds_float *l_mult = lds + n_acc;
    lds_int *l_period = (lds_int *)(lds + n_acc + n_mult);
    for (int k = threadIdx.x; k < n_acc; k += blockDim.x) lds_tables[k] = 0.0f;
    for (int k = threadIdx.x; k < n_mult; k += blockDim.x) l_mult[k] = a.multiplier[k];
    if constexpr (HOT) {
        for (int k = threadIdx.x; k < a.n_hot; k += blockDim.x) l_period[k] = a.hot_period[k];
        step.hot_acc = (lds_int *)lds;
        step.hot_accw = (lds_int *)(lds + a.n_hot * F);
        const float range = fmaxf(1.0f, __uint_as_float(*a.sw_max_bits) * a.eta * 10.0f);
        step.kHotScale = 16777216.0f / range;
        step.kHotUnit = range / 16777216.0f;
    }
    stamp_clock(a, 0);
    // dynamic segment order (SegmentTickets)
    const bool dynamic = a.tickets != nullptr && !a.single_group;
    SegmentTickets tickets;
    __shared__ int s_ticket_q[kTicketLdsWords];
    tickets.q = (lds_int *)s_ticket_q;
    int64_t sp = a.pos_begin + (a.single_group ? 0 : group);
    const int64_t stride = a.single_group ? 1 : n_groups;
    bool active = sp < a.pos_end && (a.single_group ? group == 0 : group < n_groups);
    if (dynamic && threadIdx.x == 0) tickets.init_block(a);
    __syncthreads();
    if (dynamic) {
        active = group < n_groups && a.pos_begin < a.pos_end;
        int64_t first = -1;
        if (active && sub == 0) first = tickets.take(a);
        sp = __shfl(first, lane_base);
        active = active && sp >= 0;
    }

    double ll_acc = 0.0;
    unsigned draw_acc = 0;
    // segment state
    bool have = false;
    int32_t u = 0, len = 0, t = 0;
    int64_t lo = 0, hi = 0;
    float vu[KPL], vu0[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) vu[k] = vu0[k] = 0.0f;
    constexpr int SEGR = (kSegmentRows + G - 1) / G;      // the segment's rows in visiting order, held across the lanes (row t in lane t % G)
    int32_t seg_item[SEGR], seg_pos[SEGR];
    float seg_sw[SEGR];
    auto pick = [&](const int32_t (&r)[SEGR], int tt) {
        int32_t x = r[0];
#pragma unroll
        for (int k = 1; k < SEGR; ++k) x = ((unsigned)tt / G == (unsigned)k) ? r[k] : x;
        return __shfl(x, lane_base + (int)((unsigned)tt % G));
    };
    auto pickf = [&](const float (&r)[SEGR], int tt) {
        float x = r[0];
#pragma unroll
        for (int k = 1; k < SEGR; ++k) x = ((unsigned)tt / G == (unsigned)k) ? r[k] : x;
        return __shfl(x, lane_base + (int)((unsigned)tt % G));
    };
    // row state
    bool in_row = false;
    int32_t i = 0, j = -1;
    uint32_t row_key = 0, attempt = 0;
    float sw = 0.0f, ut_ui = 0.0f, min_pu = 1e6f, wi = 0.0f, wj = 0.0f, pos_scale_i = 1.0f, neg_scale_j = 1.0f;
    int s = 1, sampled = 0, slot = -1;
    float vi[KPL], vj[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) vi[k] = vj[k] = 0.0f;

    for (int iter = 0;; ++iter) {
        if (!__any(active)) break;
        if constexpr (HOT) {      // bin sweeping duty (SgdArgs::hot_bins_v), as in sgd_segments_kernel
            const int n_waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
            if (!a.hot_direct && iter % (n_waves * kSweepEvery) == wave * kSweepEvery) {
                const SgdArgs c = cold_args();
                for (int line = blockIdx.x; line < hot_lines(c); line += gridDim.x) hot_sweep_line(c, line);
            }
        }
        if (active && !have) {
            const SgdArgs c = cold_args();
            const uint32_t seg = rfm_perm((uint32_t)sp, (uint32_t)c.n_segments, c.seg_bits, c.epoch_key ^ 0x5bd1e995u);
            const int4 d = c.seg_desc[seg];
            u = d.x; len = d.z;
            const int32_t begin = d.y;
            lo = c.csr_off[u]; hi = c.csr_off[u + 1];
            const uint32_t len_bits = rfm_perm_bits((uint32_t)len);
            const uint32_t seg_key = rfm_mix32(c.epoch_key ^ (seg * 0x9E3779B9u + 0x7F4A7C15u));
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                vu0[k] = ok(k) ? load_f32<FRESH>(c.v_u + (size_t)u * F + sub + G * k) : 0.0f;
                vu[k] = vu0[k];
            }
#pragma unroll
            for (int k = 0; k < SEGR; ++k) {
                const int tt = sub + G * k;
                seg_pos[k] = tt < len ? begin + (int32_t)rfm_perm((uint32_t)tt, (uint32_t)len, len_bits, seg_key) : begin;
                seg_item[k] = c.csr_items[seg_pos[k]];
                seg_sw[k] = c.sw_csr[seg_pos[k]];
            }
            t = 0;
            have = true;
            in_row = false;
            step.load_ulist(lo, hi);
        }
        // ---- gather: the item rows this group looks at in this iteration (slot 0 = the positive item when a row starts) ----------
        const bool starts = active && !in_row;
        if (starts) {
            const int32_t pos = pick(seg_pos, t);
            i = pick(seg_item, t);
            sw = pickf(seg_sw, t);
            row_key = rfm_row_key(a.epoch_key, (uint32_t)pos);
            attempt = 0; s = 1; sampled = 0; j = -1; min_pu = 1e6f; slot = -1; pos_scale_i = 1.0f;
        }
        int32_t c[NC];
        bool skip[NC];                                     // slot holds nothing to examine (own item / beyond the cap / group idle)
#pragma unroll
        for (int q = 0; q < NC; ++q) { c[q] = 0; skip[q] = true; }
        if (active) {
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                if (q == 0 && starts) { c[0] = i; skip[0] = false; continue; }
                int srow_unused;
                c[q] = step.draw_item(rfm_draw(row_key, attempt), srow_unused, attempt);
                ++attempt;
                skip[q] = step.member(lo, hi, c[q]);      // (rankfm/_rankfm.pyx:250-253: a drawn item of the user's own is drawn again)
            }
        }
        // (a slot with nothing to examine still gathers a row -- item 0's -- and ignores it: an unconditional load is cheaper than
        //  the branch around a conditional one, and the kernel is not bound by its requests)
        float vc[NC][KPL];
        float wsc = 0.0f, ssc = 1.0f;                       // lane q of the group: bias and step scale of slot q's item
        int32_t mine = 0;                                   // ... and the item itself
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int32_t cq = skip[q] ? 0 : c[q];
            const float *row = a.v_i + (size_t)cq * F + sub;
#pragma unroll
            for (int k = 0; k < KPL; ++k) vc[q][k] = ok(k) ? load_f32<FRESH>(row + G * k) : 0.0f;
            mine = sub == q ? cq : mine;
        }
        wsc = load_f32<FRESH>(a.w_i + (size_t)mine * a.w_stride);
        if (a.pos_scale) ssc = a.scale_in_pad ? a.w_i[(size_t)mine * a.w_stride + 1] : a.pos_scale[mine];
        // ---- examine, in draw order ------------------------------------------------------------------------------------------
        float part[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            part[q] = 0.0f;
#pragma unroll
            for (int k = 0; k < KPL; ++k) part[q] += vu[k] * vc[q][k];
        }
        bool done = false;
        if (active) {
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const float wq = __shfl(wsc, lane_base + q), sq = __shfl(ssc, lane_base + q);
                if (q == 0 && starts) {
                    // the positive item: its row, bias and step scale; a hot item's pending updates in this workgroup's LDS are part
                    // of the view (RowStep, HOT); :239
                    float sc = sq;
                    if (sc >= 2.0f) {
                        const int sl = (int)(sc * 0.5f) - 1;
                        sc -= 2.0f * (float)(sl + 1);
                        if constexpr (HOT) slot = sl;
                    }
                    pos_scale_i = a.pos_scale ? sc : 1.0f;
                    wi = wq;
#pragma unroll
                    for (int k = 0; k < KPL; ++k) vi[k] = vc[0][k];
                    float pp = part[0];
                    if constexpr (HOT) {
                        if (slot >= 0) {
                            pp = 0.0f;
#pragma unroll
                            for (int k = 0; k < KPL; ++k) {
                                if (ok(k)) vi[k] += (float)step.hot_acc[slot * F + sub + G * k] * step.kHotUnit;
                                pp += vu[k] * vi[k];
                            }
                            wi += (float)step.hot_accw[slot] * step.kHotUnit;
                        }
                    }
                    ut_ui = wi + group_sum<G>(pp);
                    continue;
                }
                const float dot = group_sum<G>(part[q]);
                if (done || skip[q] || s > a.max_samples) continue;
                const float pu = ut_ui - (wq + dot);                               // :256-257
                sampled = s;
                ++s;
                if (pu < min_pu || j < 0) {                                        // :259-261 (j < 0: keep a valid index under NaN)
                    if (pu < min_pu) min_pu = pu;
                    j = c[q]; wj = wq;
                    neg_scale_j = sq;                                              // (raw: decoded when the row is finished)
#pragma unroll
                    for (int k = 0; k < KPL; ++k) vj[k] = vc[q][k];
                }
                if (pu < kMargin) done = true;                                     // :263-264
            }
            if (s > a.max_samples) done = true;                                    // the loop's range is exhausted (:247)
            if (attempt >= kMaxAttempts) { if (sub == 0) atomicOr(a.error_flags, 1u); done = true; }      // (a safety net: the host rejects saturated users)
            in_row = !done;
        }
        // ---- the row is finished: the update (:267-326; the arithmetic and operand order of RowStep) -------------------------------
        if (active && done && j < 0) {           // (the sampler gave up before it found a single unobserved item: the row is skipped)
            j = i;
#pragma unroll
            for (int k = 0; k < KPL; ++k) vj[k] = vi[k];
            wj = wi; min_pu = 1e6f; sampled = 1; sw = 0.0f;
        }
        if (active && done) {
            const SgdArgs c = cold_args();
            const float pu = min_pu;                                               // :267-268
            const float multiplier = n_mult ? l_mult[sampled] : c.multiplier[sampled];   // :269 (integer division inside the log)
            float log_sig, d_outer;
            sigmoid_terms(pu, log_sig, d_outer);                                   // :270, :276
            if (sub == 0) { ll_acc += (double)log_sig; draw_acc += (unsigned)sampled; }
            const float g = sw * multiplier;
            const float eta = c.eta, reg_a = c.reg_a;
            const float eta_u = eta * step.user_scale, eta_i = eta * pos_scale_i;
            if (neg_scale_j >= 2.0f) neg_scale_j -= 2.0f * floorf(neg_scale_j * 0.5f);      // (a hot item's entry carries its slot above the scale)
            const float eta_j = (c.damp_positive_only || !a.pos_scale) ? eta : eta * neg_scale_j;
            float d_i[KPL], d_j[KPL];
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                const float g_u = vi[k] - vj[k];                                   // :292
                const float g_i = vu[k];                                           // :293-294 (d_v_j = -d_v_i)
                const float d_u = eta_u * (g * (d_outer * g_u) - reg_a * vu[k]);   // :308
                d_i[k] = eta_i * (g * (d_outer * g_i) - reg_a * vi[k]);            // :309
                d_j[k] = eta_j * (g * (d_outer * -g_i) - reg_a * vj[k]);           // :310
                vu[k] += d_u;
            }
            const float dwi = eta_i * (g * (d_outer * 1.0f) - reg_a * wi);         // :279
            const float dwj = eta_j * (g * (d_outer * -1.0f) - reg_a * wj);        // :280
            bool hot_done = false;
            if constexpr (HOT) {
                if (slot >= 0) {
#pragma unroll
                    for (int k = 0; k < KPL; ++k)
                        if (ok(k)) step.hot_add(step.hot_acc + slot * F + sub + G * k, d_i[k]);
                    if (sub == 0) step.hot_add(step.hot_accw + slot, dwi);
                    hot_done = true;
                    // every hot_period-th toucher of the slot publishes what the workgroup has accumulated for it (a keyed coin)
                    if (__umulhi(rfm_mix32(row_key ^ 0x7A5C3B1DU), (uint32_t)l_period[slot]) == 0u) {
#pragma unroll
                        for (int k = 0; k < KPL; ++k) {
                            if (!ok(k)) continue;
                            const float d = step.hot_take(step.hot_acc + slot * F + sub + G * k);
                            if (d != 0.0f)
                                atomic_add_f32(c.hot_direct ? a.v_i + (size_t)i * F + sub + G * k
                                                            : c.hot_bins_v + ((size_t)(blockIdx.x % kHotBins) * c.n_hot + slot) * F + sub + G * k, d);
                        }
                        if (sub == 0) {
                            const float d = step.hot_take(step.hot_accw + slot);
                            if (d != 0.0f) atomic_add_f32(c.hot_direct ? a.w_i + (size_t)i * a.w_stride : c.hot_bins_w + (size_t)(blockIdx.x % kHotBins) * c.n_hot + slot, d);
                        }
                    }
                }
            }
            if (!hot_done) {
#pragma unroll
                for (int k = 0; k < KPL; ++k)
                    if (ok(k)) atomic_add_f32(a.v_i + (size_t)i * F + sub + G * k, d_i[k]);
                if (sub == 0) atomic_add_f32(a.w_i + (size_t)i * a.w_stride, dwi);
            }
#pragma unroll
            for (int k = 0; k < KPL; ++k)
                if (ok(k)) atomic_add_f32(a.v_i + (size_t)j * F + sub + G * k, d_j[k]);
            if (sub == 0) atomic_add_f32(a.w_i + (size_t)j * a.w_stride, dwj);
            // (one group alone is a sequential program: the next row must read what this one wrote)
            if (c.single_group) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            if (++t == len) {
                // one write-back per segment; other segments of a heavy user may be in flight, so add the delta
#pragma unroll
                for (int k = 0; k < KPL; ++k)
                    if (ok(k)) atomic_add_f32(c.v_u + (size_t)u * F + sub + G * k, vu[k] - vu0[k]);
                if (c.single_group) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                have = false;
                if (dynamic) {
                    int64_t nxt = -1;
                    if (sub == 0) nxt = tickets.take(c);
                    sp = __shfl(nxt, lane_base);
                    active = sp >= 0;
                } else {
                    sp += stride;
                    active = sp < c.pos_end;
                }
            }
        }
    }
    if constexpr (HOT) {          // publish whatever is still pending
        __syncthreads();
        for (int k = threadIdx.x; k < a.n_hot * F; k += blockDim.x) {
            const float d = (float)step.hot_acc[k] * step.kHotUnit;
            if (d != 0.0f) atomic_add_f32(a.hot_direct ? a.v_i + (size_t)a.hot_item[k / F] * F + (k % F)
                                                       : a.hot_bins_v + (size_t)(blockIdx.x % kHotBins) * a.n_hot * F + k, d);
        }
        for (int k = threadIdx.x; k < a.n_hot; k += blockDim.x) {
            const float d = (float)step.hot_accw[k] * step.kHotUnit;
            if (d != 0.0f) atomic_add_f32(a.hot_direct ? a.w_i + (size_t)a.hot_item[k] * a.w_stride : a.hot_bins_w + (size_t)(blockIdx.x % kHotBins) * a.n_hot + k, d);
        }
    }
    flush_counters(a, ll_acc, draw_acc);
    stamp_clock(a, 1);
}

// ---------------------------------------------------------------------------------------------
// features kernel (production Hogwild for models with user / item features)
//
// The dense feature tables v_uf [P,F], v_if [Q,F], w_if [Q] are touched by EVERY update (rankfm/_rankfm.pyx:283-286, 313-326),
// and each touch shrinks the touched rows by 2 beta eta: in the sequential algorithm they are an exponential moving average of
// the last ~1 / (2 beta eta) = 50-170 updates' gradients, i.e. they forget within a tiny fraction of an epoch.  16 k
// interactions in flight cannot share such rows Hogwild-style (thousands of stale shrinks diverge), and per-workgroup replicas
// that evolve independently and are merged now and then drift apart (measured: profiles/r02_notes.md).  So the tables are trained
// by ONE sequential stream and read, coherently, by everybody.  Roles by workgroup index:
//   * 0: the TABLE TRAINER.  It applies the reference's table updates (:283-286, :313-326) of a stream of interactions in order
//     on a master copy in its LDS -- all table rows in parallel, one row group per table row with the row in registers (the rows
//     of the tables do not read each other), walking only the interactions that touch the row -- and publishes the copy to the
//     weight arrays after every batch.  The interactions' steps come to it ready-made:
//   * 1 .. n_producers: STEP PRODUCERS.  Each row group samples an interaction of the rank's data at random, scores it exactly
//     like a regular step and stages the step's g * d_outer, updated v_u, updated v_i - v_j, x_uf[u] and x_if[i] - x_if[j]
//     WITHOUT storing any row (the rows are trained when their own turn comes); a batch of one staged step per row group goes to
//     the trainer through a double-buffered slot in memory.  A step is a chain of ~5 dependent gathers (~15 us), applying 64 of
//     them takes ~2 us: round 2's trainer produced its own steps and so managed every ~150th row of the stream, which showed in
//     the first epoch from random weights (the item biases picked up what the tables carry in the reference); with the steps
//     produced beside it the trainer's rate is its apply rate -- a sequential SGD stream on a uniform sample of every ~20th-40th
//     row, the same process that drives the tables in the reference, with the same memory and the same noise level.
//   * the rest: the asynchronous ROW LOOP (user segments, v_u in registers, atomics for v_i / w_i) with the tables as a READ-ONLY
//     copy in the workgroup's LDS that its wavefronts keep refreshing, a slice per wavefront and row (system-scope loads: the
//     per-XCD L2s are not coherent, and a 16 KB table that is re-read all the time would otherwise never leave them).  No
//     lock-step, no barrier in the loop.  BPR models with at most 32 + 32 features on 16-lane row groups take the pipelined form
//     (FeatFast below), everything else the generic RowStep.
// One group alone (debug_flags bit 0) does everything in the reference's order -- the sequential form the parity tests pin; with
// debug_flags bit 5 (tables frozen: no trainer, no producers) one group alone runs the pipelined row loop sequentially, which
// pins THAT code to the oracle as well.
// ---------------------------------------------------------------------------------------------
#define RFM_REP8(X, O) X(O + 0) X(O + 1) X(O + 2) X(O + 3) X(O + 4) X(O + 5) X(O + 6) X(O + 7)
typedef float rfm_f4 __attribute__((ext_vector_type(4)));
typedef float rfm_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) rfm_f4 lds_f4;
typedef __attribute__((address_space(3))) rfm_f2 lds_f2;

// flags of one launch of the features kernel (SgdArgs::feat_flags, zero between launches)
constexpr int kFeatMaxProducers = 16;
constexpr int kFeatReady = 0;                                  // [2 * producers] batches written into each slot
constexpr int kFeatConsumed = 2 * kFeatMaxProducers;           // [2 * producers] batches the trainer has taken out of each slot
constexpr int kFeatStop = 4 * kFeatMaxProducers;               // the regular workgroups are done
constexpr int kFeatExited = kFeatStop + 1;                     // producers that have left
constexpr int kFeatDone = kFeatStop + 2;                       // (unused since the trainer works to a fixed quota)
constexpr int kFeatFlagWords = kFeatStop + 4;
constexpr unsigned kFeatSpinLimit = 1u << 23;                  // polls (~0.5 us each) before a waiting workgroup gives up: seconds

// (The trainer and the producers are non-inlined functions that receive SgdArgs by value: the compiler no longer knows that its
// pointers are global memory, and a FLAT load counts against the LDS counter as well -- every LDS wait of the trainer's apply loop
// waited for the batch in flight (measured: 17 us per batch instead of 6).  Their hot pointers are therefore cast to the global
// address space explicitly.)
typedef __attribute__((address_space(1))) float g_float;
typedef __attribute__((address_space(1))) unsigned int g_uint;
__device__ __forceinline__ unsigned flag_load(const g_uint *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void flag_store(g_uint *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// The pipelined row loop keeps its LDS copy of v_uf / v_if LANE-MAJOR: the KPL factor dwords lane s of a row group owns (s, s + 16,
// ...) are consecutive, so a table row costs the lane one 16-byte LDS read instead of KPL 4-byte ones.  Row stride 16 * KPL.
template <int KPL>
__device__ __forceinline__ void lds_row_load(const lds_float *p, float (&t)[KPL]) {
    if constexpr (KPL % 4 == 0) {
#pragma unroll
        for (int q = 0; q < KPL / 4; ++q) {
            const rfm_f4 v = *(const lds_f4 *)(p + 4 * q);
            t[4 * q] = v.x; t[4 * q + 1] = v.y; t[4 * q + 2] = v.z; t[4 * q + 3] = v.w;
        }
    } else if constexpr (KPL % 2 == 0) {
#pragma unroll
        for (int q = 0; q < KPL / 2; ++q) {
            const rfm_f2 v = *(const lds_f2 *)(p + 2 * q);
            t[2 * q] = v.x; t[2 * q + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < KPL; ++k) t[k] = p[k];
    }
}

// acc[k] += sum_t x[t] * table[t][this lane's dwords], t = 0 .. n-1 (n <= 32): x is held across the 16 lanes of the group (lane s:
// x[s] in xr0, x[s + 16] in xr1; entries >= n are zero) and reaches all lanes through a DPP row_share -- one VALU move per tag, no
// ballot / shuffle walk over the non-zero entries (zero entries add an exact zero; the reference skips them, :73, :81).  The table
// is padded with zero rows to a multiple of 8.
template <int KPL>
__device__ __forceinline__ void project_dense(float xr0, float xr1, int n, const lds_float *tab_lane, float (&acc)[KPL]) {
    constexpr int FS = 16 * KPL;
#define RFM_PSTEP(B)                                                                      \
    {                                                                                     \
        const float x = dpp_mov<0x150 + ((B) & 15)>(((B) < 16) ? xr0 : xr1);              \
        float t[KPL];                                                                     \
        lds_row_load<KPL>(tab_lane + (B) * FS, t);                                        \
        _Pragma("unroll") for (int k = 0; k < KPL; ++k) acc[k] = __builtin_fmaf(x, t[k], acc[k]); \
    }
    if (n > 0) { RFM_REP8(RFM_PSTEP, 0) }
    if (n > 8) { RFM_REP8(RFM_PSTEP, 8) }
    if (n > 16) { RFM_REP8(RFM_PSTEP, 16) }
    if (n > 24) { RFM_REP8(RFM_PSTEP, 24) }
#undef RFM_PSTEP
}

// locals every role of the features kernel derives from the launch (trainer, producers, row loops)
#define RFM_FEAT_LOCALS                                                                                                                   \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;                                                \
    const int sub = lane % G;                                                                                                             \
    const int F = a.n_factors;                                                                                                            \
    lds_float *lds_tables = lds;                                                                                                          \
    const int n_uf_f = a.n_uf * F, n_if_f = a.n_if * F, n_tab = n_uf_f + n_if_f + a.n_if;                                                  \
    auto table_ptr = [&](int k) { return (g_float *)(k < n_uf_f ? a.v_uf + k : (k < n_uf_f + n_if_f ? a.v_if + (k - n_uf_f) : a.w_if + (k - n_uf_f - n_if_f))); }; \
    const int NP = a.single_group ? 0 : a.n_producers;               /* (one group alone: no trainer, no producers) */                    \
    const bool trains = !a.single_group && !a.feat_frozen;                                                                                \
    g_uint *flags = (g_uint *)a.feat_flags;                                                                                               \
    const int gid = threadIdx.x / G, gpb = blockDim.x / G;                                                                                \
    const int n_slot = 1 + 2 * F + a.n_uf + a.n_if;                  /* staged step of one interaction (RowStep::stage) */                \
    const size_t batch_floats = (size_t)gpb * n_slot;                                                                                     \
    const int n_regular = (int)gridDim.x;                            /* (row-loop kernels: every workgroup walks rows) */                 \
    (void)lane; (void)wave; (void)n_waves; (void)sub; (void)gid; (void)flags; (void)batch_floats; (void)n_regular; (void)lds_tables; (void)table_ptr; \
    (void)trains; (void)NP; (void)n_tab;

// The roles of the features kernel other than the pipelined row loop are separate (non-inlined) functions: each gets a register
// allocation of its own, so that the trainer's batch in flight or the generic step's feature vectors do not cost the row loop spills.
template <int G, int KPL>
__device__ __forceinline__ void feat_table_trainer(const SgdArgs &a, lds_float *lds, lds_int *s_stop_p) {
    RFM_FEAT_LOCALS
    // natural layout of the tables: [P, F] | [Q, F] | [Q]
    for (int k = threadIdx.x; k < n_tab; k += blockDim.x) lds_tables[k] = *table_ptr(k);
    // The staging area holds a batch in the TRAINER'S layout: per staged step [updated v_u | updated v_i - v_j | g * d_outer | x_uf[u] |
    // x_if[i] - x_if[j]], the two vectors lane-major and padded to the group width (lane s of a row group reads its KPL dwords with
    // one 16-byte LDS read, no per-dword bounds predicate), slots padded to a multiple of four floats.
    constexpr int FS = G * KPL;
    const int NSL = (2 * FS + 1 + a.n_uf + a.n_if + 3) & ~3;
    lds_float *stage = lds + ((n_tab + 3) & ~3);
    __syncthreads();
    // ---- the table trainer ------------------------------------------------------------------------------------------
    const float eta_f = a.eta, reg_b = a.reg_b;
    // rho^n, n = 0 .. gpb: w_if shrinks on EVERY interaction (:283-286), also those whose tag difference is zero, which the
    // row walk below skips
    lds_float *rho_pow = stage + (size_t)gpb * NSL;
    if (threadIdx.x <= (unsigned)gpb) rho_pow[threadIdx.x] = powf(1.0f - eta_f * reg_b, (float)threadIdx.x);
    for (size_t k = threadIdx.x; k < (size_t)gpb * NSL; k += blockDim.x) stage[k] = 0.0f;     // (the padding is read, never written)
    // The loop is software-pipelined: while batch q - 1 is being applied out of LDS, batch q is on its way from memory into
    // registers and the ready flag of batch q + 1 is being polled, so that a batch costs the trainer its apply time and two
    // barriers instead of three dependent memory round trips (flag, data, publication).
    constexpr int kPre = 14;                                  // dwords of a batch a thread keeps in flight
    float pre[kPre];
    const unsigned n_batch = (unsigned)batch_floats, n_threads = blockDim.x;
    // where dword threadIdx.x + j * blockDim.x of a batch (the producers' layout: RowStep::stage, [g d_outer | v_u | v_i - v_j | x_uf |
    // x_if diff] per step) goes in the staging area; the same for every batch, so computed once
    auto stage_index = [&](size_t k) {
        const int s2 = (int)(k / (size_t)n_slot), off = (int)(k - (size_t)s2 * n_slot);
        int dst;
        if (off == 0) dst = 2 * FS;
        else if (off < 1 + 2 * F) {
            const int v = off - 1 < F ? 0 : 1, f = off - 1 - v * F;
            dst = v * FS + (f % G) * KPL + f / G;
        } else dst = 2 * FS + 1 + (off - 1 - 2 * F);
        return s2 * NSL + dst;
    };
    // (two 16-bit staging indexes per register; 0xFFFF = none.  Register pressure matters here: a spilled value reloaded between two
    // of the batch's loads waits for every load issued before it -- the system-scope loads return in order)
    unsigned pre_dst[(kPre + 1) / 2];
#pragma unroll
    for (int j = 0; j < kPre; j += 2) {
        const unsigned k0 = threadIdx.x + (unsigned)j * n_threads, k1 = k0 + n_threads;
        const unsigned d0 = k0 < n_batch ? (unsigned)stage_index(k0) : 0xFFFFu, d1 = (j + 1 < kPre && k1 < n_batch) ? (unsigned)stage_index(k1) : 0xFFFFu;
        pre_dst[j / 2] = d0 | (d1 << 16);
    }
    const bool prefetch_ok = (size_t)gpb * NSL < 0xFFFFu;    // (else every dword of a batch takes the direct path below)
    unsigned applied = 0;
    // The trainer applies a FIXED number of staged steps per launch: the host's quota (rows of the launch / the pace a trainer keeps
    // beside that many row-loop workgroups, rfm_api.hip), in whole batches, at least one.  It does not look at the row loops: the
    // step count -- and with it the tables a given launch geometry produces -- no longer depends on timing.
    const unsigned quota = (unsigned)(((a.table_quota > (int64_t)gpb ? a.table_quota : (int64_t)gpb) + gpb - 1) / gpb) * (unsigned)gpb;
    bool staged = false;                                      // LDS holds a batch that has not been applied yet
    unsigned flag_next = 0;                                   // (thread 0) the ready counter of the next batch, loaded ahead
    auto slot_of = [&](unsigned q, int &p, unsigned &par, unsigned &m) {
        p = NP > 0 ? (int)(q % (unsigned)NP) : 0;
        // (ONE slot per producer: a staged step is scored on the tables of its time, and every batch that waits in a slot is a batch
        // of stale steps -- with two slots each, three producers cost the 3000 x 2000 feature fixture 0.6 point of hit_rate@10 against
        // one: profiles/r03_notes.md.  The second slot of the ring stays unused.)
        const unsigned n_p = NP > 0 ? q / (unsigned)NP : q;
        par = 0u; m = n_p;
    };
    if (threadIdx.x == 0 && NP > 0) flag_next = __hip_atomic_load(flags + kFeatReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // (diagnostics, rfm_fit_report.feat_diag: time this workgroup waited for a batch / ran in all, in 100 MHz ticks)
    const unsigned long long t_begin = wall_clock64();
    unsigned long long t_wait = 0, t_seg[4] = {0, 0, 0, 0};      // (apply | publish | batch into LDS | slot release)
    for (unsigned q = 0;; ++q) {
        int p;
        unsigned par, m;
        slot_of(q, p, par, m);
        if (threadIdx.x == 0) {
            int stop = applied >= quota ? 1 : 0;
            const unsigned long long t0 = wall_clock64();
            for (unsigned spin = 0; !stop && (NP == 0 || flag_next < m + 1u); ++spin) {
                if (spin > kFeatSpinLimit) { atomicOr(a.error_flags, 8u); stop = 1; break; }      // (never observed: a hang guard)
                __builtin_amdgcn_s_sleep(4);
                if (NP > 0) flag_next = __hip_atomic_load(flags + kFeatReady + 2 * p + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            t_wait += wall_clock64() - t0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");       // the batch's data is read after its flag
            *s_stop_p = stop;
        }
        __syncthreads();
        if (*s_stop_p) break;
        const unsigned long long tA = wall_clock64();
        // batch q: on its way into registers (the part beyond kPre dwords per thread goes straight to LDS below)
        const g_float *src = (const g_float *)a.feat_ring + (size_t)(2 * p + par) * batch_floats;
        {
            const g_float *pp = src + threadIdx.x;
            unsigned k = threadIdx.x;
#pragma unroll
            for (int j = 0; j < kPre; ++j) {
                pre[j] = (prefetch_ok && k < n_batch) ? __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0f;
                pp += n_threads;
                k += n_threads;
            }
        }
        if (threadIdx.x == 0 && NP > 0) {                     // ... and the flag of batch q + 1
            int p1;
            unsigned par1, m1;
            slot_of(q + 1, p1, par1, m1);
            flag_next = __hip_atomic_load(flags + kFeatReady + 2 * p1 + par1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (staged) {
            // Apply the staged steps.  Within one interaction the table rows do not read each other, so the reference's
            // sequential update of the tables over the batch (rankfm/_rankfm.pyx:283-286, 313-326) is, for each table ROW, a walk
            // over the interactions that touch it -- all rows at once, one row group per row with the row in registers, plain
            // read and write.  The interactions that touch the row are found by the group's lanes together (one ballot per G
            // staged steps); the row of v_if for tag q also carries w_if[q] (lane 0).
            for (int r = gid; r < a.n_uf + a.n_if; r += gpb) {
                const bool uf = r < a.n_uf;
                if (uf ? !a.has_uf : !a.has_if) continue;
                lds_float *row = lds + (size_t)r * F;                                  // v_uf rows, then v_if rows
                const int xoff = 2 * FS + 1 + r;                                        // the step's coefficient of this table row
                const lds_float *vec = stage + (uf ? FS : 0) + sub * KPL;               // updated v_i - v_j | updated v_u, this lane's dwords
                float tr[KPL];
#pragma unroll
                for (int k = 0; k < KPL; ++k) tr[k] = (sub + G * k < F) ? row[sub + G * k] : 0.0f;
                float wq = uf ? 0.0f : lds[n_uf_f + n_if_f + (r - a.n_uf)];
                int last = -1;                                                          // last staged step applied to w_if[q]
                // one touching step: tr <- tr + eta (c v - reg_b tr), and the shrink of w_if over the untouched steps before it
                auto one = [&](int s2, float c, const float (&v)[KPL], float rho_gap) {
#pragma unroll
                    for (int k = 0; k < KPL; ++k) tr[k] += eta_f * (c * v[k] - reg_b * tr[k]);
                    if (!uf) {
                        wq = wq * rho_gap;
                        wq += eta_f * (c - reg_b * wq);
                        last = s2;
                    }
                };
                for (int c64 = 0; c64 < gpb; c64 += 64) {
                    // the steps that touch this row, found by the group's lanes together (one ballot per G staged steps, their LDS reads
                    // in flight together)
                    unsigned long long act = 0;
                    for (int c0 = c64; c0 < gpb && c0 < c64 + 64; c0 += G) {
                        const int mine = c0 + sub;
                        const bool on = mine < gpb && stage[(size_t)mine * NSL + xoff] != 0.0f;
                        unsigned long long bits;
                        if constexpr (G == 64) bits = __ballot(on);
                        else bits = (unsigned long long)group_ballot<G>(on);
                        act |= bits << (c0 - c64);
                    }
                    // two touching steps per round: their coefficients, vectors and shrink powers are read together (one LDS round
                    // trip), then applied one after the other
                    while (act) {
                        const int sA = c64 + __ffsll((long long)act) - 1;
                        act &= act - 1;
                        const bool two = act != 0;
                        const int sB = two ? c64 + __ffsll((long long)act) - 1 : sA;
                        if (two) act &= act - 1;
                        const lds_float *stA = stage + (size_t)sA * NSL, *stB = stage + (size_t)sB * NSL;
                        float vA[KPL], vB[KPL];
                        lds_row_load<KPL>(vec + (size_t)sA * NSL, vA);
                        lds_row_load<KPL>(vec + (size_t)sB * NSL, vB);
                        const float cA = stA[2 * FS] * stA[xoff], cB = stB[2 * FS] * stB[xoff];
                        const float rA = uf ? 1.0f : rho_pow[sA - last - 1], rB = uf ? 1.0f : rho_pow[sB - sA - (two ? 1 : 0)];
                        one(sA, cA, vA, rA);
                        if (two) one(sB, cB, vB, rB);
                    }
                }
#pragma unroll
                for (int k = 0; k < KPL; ++k)
                    if (sub + G * k < F) row[sub + G * k] = tr[k];
                if (!uf && sub == 0) lds[n_uf_f + n_if_f + (r - a.n_uf)] = wq * rho_pow[gpb - 1 - last];
            }
            __syncthreads();
            const unsigned long long tB = wall_clock64();
            // publish the master copy (write-through to memory)
            for (int k = threadIdx.x; k < n_tab; k += blockDim.x)
                __hip_atomic_store(table_ptr(k), lds_tables[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            applied += (unsigned)gpb;
            t_seg[0] += tB - tA;
            t_seg[1] += wall_clock64() - tB;
        }
        const unsigned long long tC = wall_clock64();
        // batch q into the staging area (everybody has finished reading batch q - 1: the barrier above / the first round)
#pragma unroll
        for (int j = 0; j < kPre; ++j) {
            const unsigned d = (pre_dst[j / 2] >> (16 * (j & 1))) & 0xFFFFu;
            if (prefetch_ok && d != 0xFFFFu) stage[d] = pre[j];
        }
        for (size_t k = threadIdx.x + (prefetch_ok ? (size_t)kPre * blockDim.x : 0); k < batch_floats; k += blockDim.x)
            stage[stage_index(k)] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        staged = true;
        __syncthreads();
        const unsigned long long tD = wall_clock64();
        if (threadIdx.x == 0) flag_store(flags + kFeatConsumed + 2 * p + par, m + 1u);      // (the loads have returned: the slot is free)
        t_seg[2] += tD - tC;
        t_seg[3] += wall_clock64() - tD;
    }
    if (threadIdx.x == 0) {
        a.feat_clock[0] = t_begin;
        a.feat_clock[1] = wall_clock64();
        if (applied) atomicAdd(a.error_flags + 2, applied);      // staged steps applied (rfm_fit_report.table_steps)
        atomicAdd(a.error_flags + 4, (unsigned)(t_wait / 100));
        atomicAdd(a.error_flags + 5, (unsigned)((wall_clock64() - t_begin) / 100));
        for (int k = 0; k < 4; ++k) atomicAdd(a.error_flags + 8 + k, (unsigned)(t_seg[k] / 100));
    }
    // the launch is over: release the producers, wait until they have left, and leave the flags zero for the next launch
    if (threadIdx.x == 0) {
        flag_store(flags + kFeatStop, 1u);
        for (unsigned spin = 0; flag_load(flags + kFeatExited) < (unsigned)NP && spin <= kFeatSpinLimit; ++spin) __builtin_amdgcn_s_sleep(8);
        for (int k = 0; k < kFeatFlagWords; ++k) flag_store(flags + k, 0u);
    }
    return;
}

template <int G, int KPL, bool WARPB>
__device__ __forceinline__ void feat_step_producer(const SgdArgs &a, lds_float *lds, lds_int *s_stop_p) {
    RFM_FEAT_LOCALS
    for (int k = threadIdx.x; k < n_tab; k += blockDim.x) lds_tables[k] = *table_ptr(k);
    lds_float *stage = lds + n_tab;
    __syncthreads();
    // ---- a step producer ----------------------------------------------------------------------------------------------------
    typedef RowStep<G, KPL, false, true, true, true, true, false, WARPB, false, 1> Train;
    Train step(a, sub, lds, lds + n_uf_f, lds + n_uf_f + n_if_f);
    const int p = (int)blockIdx.x - 1;
    double ll_unused = 0.0;
    unsigned draws_unused = 0;
    const unsigned long long t_begin = wall_clock64();
    unsigned long long t_wait = 0;
    for (unsigned n = 0;; ++n) {
        const unsigned par = 0u, m = n;                  // (one slot per producer: see the trainer)
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            int stop = 0;
            for (unsigned spin = 0;; ++spin) {           // the slot must have been emptied m times
                if (flag_load(flags + kFeatStop)) { stop = 1; break; }
                if (flag_load(flags + kFeatConsumed + 2 * p + par) >= m) break;
                if (spin > kFeatSpinLimit) { atomicOr(a.error_flags, 8u); stop = 1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            t_wait += wall_clock64() - t0;
            *s_stop_p = stop;
        }
        __syncthreads();
        if (*s_stop_p) break;
        // this batch is scored on the tables as published now
        for (int k = threadIdx.x; k < n_tab; k += blockDim.x)
            lds_tables[k] = __hip_atomic_load(table_ptr(k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
        // a uniformly random row: a random segment (accepted with probability length / 32) and a random row of it
        uint32_t h = rfm_mix32(a.epoch_key ^ rfm_mix32(((n * (unsigned)NP + (unsigned)p) * (unsigned)gpb + (unsigned)gid) * 0x9E3779B9U + 0x3C6EF372U + a.launch_index));
        int4 d;
        for (;;) {
            d = a.seg_desc[rfm_draw_to_item(h, (uint32_t)a.n_segments)];
            h = rfm_mix32(h + 0x632BE5ABU);
            if ((int)rfm_draw_to_item(h, (uint32_t)kSegmentRows) < d.z) break;
            h = rfm_mix32(h + 0x7F4A7C15U);
        }
        h = rfm_mix32(h ^ 0x85EBCA6BU);
        const int32_t u = d.x, pos = d.y + (int32_t)rfm_draw_to_item(h, (uint32_t)d.z);
        const int32_t i = a.csr_items[pos];
        const float sw = a.sw_csr[pos];
        const int64_t lo = a.csr_off[u], hi = a.csr_off[u + 1];
        float vu[KPL];
#pragma unroll
        for (int k = 0; k < KPL; ++k) vu[k] = (sub + G * k < F) ? load_f32<true>(a.v_u + (size_t)u * F + sub + G * k) : 0.0f;
        step.stage = stage + (size_t)gid * n_slot;
        step(rfm_mix32(h ^ 0xC2B2AE35U), u, i, sw, lo, hi, vu, ll_unused, draws_unused);
        __syncthreads();
        g_float *dst = (g_float *)a.feat_ring + (size_t)(2 * p + par) * batch_floats;
        for (size_t k = threadIdx.x; k < batch_floats; k += blockDim.x)
            __hip_atomic_store(dst + k, stage[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // this wavefront's stores have been performed ...
        __syncthreads();                                     // ... and everybody's, before the slot is announced
        if (threadIdx.x == 0) flag_store(flags + kFeatReady + 2 * p + par, m + 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(a.error_flags + 6, (unsigned)(t_wait / 100));
        atomicAdd(a.error_flags + 7, (unsigned)((wall_clock64() - t_begin) / 100));
        __hip_atomic_fetch_add(flags + kFeatExited, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
}

// (inlined into the kernel: as a separate function its pointers would lose their global address space -- the struct is passed by
// value -- and every access of the generic step would become a FLAT instruction)
template <int G, int KPL, bool FRESH, bool WARPB>
__device__ __forceinline__ void feat_generic_rows(const SgdArgs &a, lds_float *lds) {
    RFM_FEAT_LOCALS
    const int64_t group = a.single_group ? (int64_t)threadIdx.x / G : (int64_t)blockIdx.x * gpb + threadIdx.x / G;
    int64_t n_groups = a.single_group ? gpb : (int64_t)n_regular * gpb;
    if (a.max_groups > 0 && a.max_groups < n_groups) n_groups = a.max_groups;
    double ll_acc = 0.0;
    unsigned draw_acc = 0;
    int64_t sp = a.pos_begin + (a.single_group ? 0 : group);
    const int64_t stride = a.single_group ? 1 : n_groups;
    bool active = sp < a.pos_end && (a.single_group ? group == 0 : group < n_groups);
    bool have = false;
    int32_t u = 0, begin = 0, len = 0, t = 0, len_bits = 0;
    uint32_t seg_key = 0;
    int64_t lo = 0, hi = 0;
    float vu[KPL], vu0[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) vu[k] = vu0[k] = 0.0f;
    // ---- generic row loop (WARP, wide feature vectors, other row-group shapes; one group alone: both rows and tables) -----------
    if (threadIdx.x == 0 && blockIdx.x == 0) a.feat_clock[2] = wall_clock64();
    for (int k = threadIdx.x; k < n_tab; k += blockDim.x) lds_tables[k] = *table_ptr(k);
    __syncthreads();
    typedef RowStep<G, KPL, false, true, true, FRESH, true, false, WARPB, false, 0> Reg;
    typedef RowStep<G, KPL, false, true, true, FRESH, true, false, WARPB, false, 2> Both;
    Reg step(a, sub, lds, lds + n_uf_f, lds + n_uf_f + n_if_f);
    Both both(a, sub, lds, lds + n_uf_f, lds + n_uf_f + n_if_f);
    const bool train_here = a.single_group && !a.feat_frozen;          // one group alone trains the tables in its LDS
    while (__any(active)) {
        if (trains) {
            const int per = (n_tab + n_waves - 1) / n_waves, e0 = wave * per, e1 = e0 + per < n_tab ? e0 + per : n_tab;
            for (int k = e0 + lane; k < e1; k += 64)
                lds_tables[k] = __hip_atomic_load(table_ptr(k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (active && !have) {
            const uint32_t seg = rfm_perm((uint32_t)sp, (uint32_t)a.n_segments, a.seg_bits, a.epoch_key ^ 0x5bd1e995u);
            const int4 d = a.seg_desc[seg];
            u = d.x; begin = d.y; len = d.z;
            lo = a.csr_off[u]; hi = a.csr_off[u + 1];
            len_bits = (int32_t)rfm_perm_bits((uint32_t)len);
            seg_key = rfm_mix32(a.epoch_key ^ (seg * 0x9E3779B9u + 0x7F4A7C15u));
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                vu0[k] = (sub + G * k < F) ? load_f32<FRESH>(a.v_u + (size_t)u * F + sub + G * k) : 0.0f;
                vu[k] = vu0[k];
            }
            t = 0;
            have = true;
        }
        if (active) {
            const int32_t pos = begin + (int32_t)rfm_perm((uint32_t)t, (uint32_t)len, (uint32_t)len_bits, seg_key);
            const int32_t i = a.csr_items[pos];
            const float sw = a.sw_csr[pos];
            if (train_here) both(rfm_row_key(a.epoch_key, (uint32_t)pos), u, i, sw, lo, hi, vu, ll_acc, draw_acc);
            else step(rfm_row_key(a.epoch_key, (uint32_t)pos), u, i, sw, lo, hi, vu, ll_acc, draw_acc);
            if (++t == len) {
#pragma unroll
                for (int k = 0; k < KPL; ++k)
                    if (sub + G * k < F) atomic_add_f32(a.v_u + (size_t)u * F + sub + G * k, vu[k] - vu0[k]);
                have = false;
                sp += stride;
                active = sp < a.pos_end;
            }
        }
    }
    if (train_here) {             // the one group trained the tables in its LDS: store them
        for (int k = threadIdx.x; k < n_tab; k += blockDim.x) *table_ptr(k) = lds_tables[k];
    }
    flush_counters(a, ll_acc, draw_acc);
    if (threadIdx.x == 0) atomicMax(a.feat_clock + 3, wall_clock64());
}

// The table trainer and its step producers: a kernel of their own (1 + n_producers workgroups, launched on a second stream beside the
// row loops -- launch_segments in rfm_sgd_inst.inc).  Rounds 2-3 ran them as roles inside the row-loop kernel: non-inlined functions
// whose register ceiling (128 at 1024 threads) and 1 KB of stack the row loop shared -- it compiled with 36 spilled VGPRs -- and whose
// step count depended on when the row loops finished.  Here they have their own allocation, the row-loop kernels compile alone, and
// the trainer applies a fixed quota of steps (SgdArgs::table_quota).
template <int G, int KPL, bool WARPB>
__global__ void __launch_bounds__(1024) feat_tables_kernel(const SgdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_dynamic[];
    lds_float *lds = (lds_float *)lds_dynamic;
    __shared__ int s_stop;
    if (blockIdx.x == 0) feat_table_trainer<G, KPL>(a, lds, (lds_int *)&s_stop);
    else feat_step_producer<G, KPL, WARPB>(a, lds, (lds_int *)&s_stop);
}

// the generic row loop (WARP with features, wide feature vectors, 4- / 64-lane row groups; one group alone: rows AND tables)
template <int G, int KPL, bool FRESH, bool WARPB>
__global__ void __launch_bounds__(1024) sgd_features_kernel(const SgdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_dynamic[];
    feat_generic_rows<G, KPL, FRESH, WARPB>(a, (lds_float *)lds_dynamic);
}

// the pipelined row loop (BPR, <= 32 + 32 features, 16-lane row groups): every workgroup walks rows, the tables are a read-only
// lane-major copy in LDS that its wavefronts keep refreshing
// (THREADS: the largest workgroup the instantiation is launched with.  The loop wants ~176 VGPRs: at 1024 threads -- 128 registers --
//  it spills ~50 of them, at 768 -- three wavefronts per SIMD, 168 registers -- two.)
template <int G, int KPL, bool FRESH, int THREADS>
__global__ void __launch_bounds__(THREADS) sgd_features_fast_kernel(const SgdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_dynamic[];
    lds_float *lds = (lds_float *)lds_dynamic;
    RFM_FEAT_LOCALS
    const int first_regular = 0;
    const int64_t group = a.single_group ? (int64_t)threadIdx.x / G : (int64_t)blockIdx.x * gpb + threadIdx.x / G;
    int64_t n_groups = a.single_group ? gpb : (int64_t)n_regular * gpb;
    if (a.max_groups > 0 && a.max_groups < n_groups) n_groups = a.max_groups;
    double ll_acc = 0.0;
    unsigned draw_acc = 0;
    int64_t sp = a.pos_begin + (a.single_group ? 0 : group);
    const int64_t stride = a.single_group ? 1 : n_groups;
    bool active = sp < a.pos_end && (a.single_group ? group == 0 : group < n_groups);
    bool have = false;
    int32_t u = 0, begin = 0, len = 0, t = 0, len_bits = 0;
    uint32_t seg_key = 0;
    int64_t lo = 0, hi = 0;
    float vu[KPL], vu0[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) vu[k] = vu0[k] = 0.0f;
    if (threadIdx.x == 0 && blockIdx.x == 0) a.feat_clock[2] = wall_clock64();
    stamp_clock(a, 0);

    static_assert(G == 16, "the pipelined feature row loop is written for 16-lane row groups");
    {
      {
        // ---- the pipelined row loop (BPR, <= 32 + 32 features, 16-lane row groups) -------------------------------------------
        constexpr int FS = G * KPL;                                     // LDS row stride (rows are padded to full width)
        const int P8 = (a.n_uf + 7) & ~7, Q8 = (a.n_if + 7) & ~7;     // tables padded with zero rows to a multiple of 8
        lds_float *t_uf = lds, *t_if = lds + (size_t)P8 * FS, *t_wif = lds + (size_t)(P8 + Q8) * FS;
        const int n_fast = (P8 + Q8) * FS + a.n_if;
        // LDS element e of the lane-major copy <- table element (global), or zero padding
        auto table_elem = [&](int e) {
            float v = 0.0f;
            if (e < (P8 + Q8) * FS) {
                const int r = e / FS, w = e % FS, f = (w % KPL) * G + w / KPL;     // lane w / KPL, its dword w % KPL
                if (f < F) {
                    if (r < P8) { if (r < a.n_uf) v = __hip_atomic_load(a.v_uf + (size_t)r * F + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
                    else if (r - P8 < a.n_if) v = __hip_atomic_load(a.v_if + (size_t)(r - P8) * F + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            } else v = __hip_atomic_load(a.w_if + (e - (P8 + Q8) * FS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return v;
        };
        for (int e = threadIdx.x; e < n_fast; e += blockDim.x) lds_tables[e] = table_elem(e);
        // hot positive items (SgdArgs::hot_item): pending updates of this workgroup in 32-bit fixed point, behind the tables
        const int hot_off = (n_fast + 3) & ~3, n_hot = a.n_hot;
        lds_int *hot_acc = (lds_int *)(lds + hot_off), *hot_accw = hot_acc + n_hot * F;
        for (int k = threadIdx.x; k < n_hot * (F + 1); k += blockDim.x) hot_acc[k] = 0;
        float hot_scale = 16777216.0f, hot_unit = 1.0f / 16777216.0f;
        if (n_hot > 0) {
            const float range = fmaxf(1.0f, __uint_as_float(*a.sw_max_bits) * a.eta * 10.0f);      // (see RowStep::kHotScale)
            hot_scale = 16777216.0f / range;
            hot_unit = range / 16777216.0f;
        }
        __syncthreads();
        typedef RowStep<G, KPL, false, true, true, FRESH, true, false, false, false, 0> Reg;
        Reg step(a, sub, lds, lds, lds);                               // (draws, membership test, user damping; its tables are unused)
        const lds_float *uf_lane = t_uf + sub * KPL, *if_lane = t_if + sub * KPL;
        const int lane_base = lane - sub;
        const float multiplier = a.multiplier[1];                       // :269 with sampled == 1
        const float eta = a.eta, reg_a = a.reg_a;
        constexpr int SEGR = (kSegmentRows + G - 1) / G;
        int32_t seg_item[SEGR], seg_pos[SEGR];
        float seg_sw[SEGR];
        float xu0 = 0.0f, xu1 = 0.0f;
        // the positive item's row, bias + step scale (one padded line), tags: fetched one row ahead
        struct Pos { float v[KPL]; float w, scale, x0, x1; } cur, nxt;
        auto fetch_pos = [&](int32_t it, Pos &p) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) p.v[k] = (sub + G * k < F) ? load_f32<FRESH>(a.v_i + (size_t)it * F + sub + G * k) : 0.0f;
            if (a.scale_in_pad) {
                const float x = load_f32<FRESH>(a.w_i + (size_t)it * a.w_stride + (sub & 1));
                p.w = __shfl(x, lane_base);
                p.scale = __shfl(x, lane_base + 1);
            } else {
                p.w = load_f32<FRESH>(a.w_i + (size_t)it * a.w_stride);
                p.scale = a.pos_scale ? a.pos_scale[it] : 1.0f;
            }
            p.x0 = p.x1 = 0.0f;
            if (a.has_if) {
                const float *x = a.x_if + (size_t)it * a.n_if;
                if (sub < a.n_if) p.x0 = x[sub];
                if (sub + G < a.n_if) p.x1 = x[sub + G];
            }
        };
        auto pick = [&](const int32_t (&r)[SEGR], int tt) {
            int32_t x = r[0];
#pragma unroll
            for (int k = 1; k < SEGR; ++k) x = ((unsigned)tt / G == (unsigned)k) ? r[k] : x;
            return x;
        };
        auto pickf = [&](const float (&r)[SEGR], int tt) {
            float x = r[0];
#pragma unroll
            for (int k = 1; k < SEGR; ++k) x = ((unsigned)tt / G == (unsigned)k) ? r[k] : x;
            return x;
        };
        // dynamic segment order (SegmentTickets)
        const bool dynamic = a.tickets != nullptr && !a.single_group;
        SegmentTickets tickets;
        __shared__ int s_ticket_q[kTicketLdsWords];
        tickets.q = (lds_int *)s_ticket_q;
        if (dynamic) {
            if (threadIdx.x == 0) tickets.init_block(a);
            __syncthreads();
            active = group < n_groups && a.pos_begin < a.pos_end;
            int64_t first = -1;
            if (active && sub == 0) first = tickets.take(a);
            sp = __shfl(first, lane_base);
            active = active && sp >= 0;
        }
        for (int iter = 0; __any(active); ++iter) {
            // Every wavefront keeps a slice of the workgroup's copy fresh, a part of it per row: the loads are issued here and land in
            // LDS at the END of the row, so that their latency (system-scope loads go to memory) is the row's, not an extra round
            // trip.  (Readers may see a row half old, half new: both are tables the trainer published.)
            constexpr int RF = 2;
            float rf_val[RF];
            int rf_e[RF];
            if (trains) {
                const int per = (n_fast + n_waves - 1) / n_waves, e0 = wave * per, e1 = e0 + per < n_fast ? e0 + per : n_fast;
                const int parts = (per + 64 * RF - 1) / (64 * RF);
#pragma unroll
                for (int k = 0; k < RF; ++k) {
                    rf_e[k] = e0 + lane + 64 * ((iter % parts) * RF + k);
                    rf_val[k] = rf_e[k] < e1 ? table_elem(rf_e[k]) : 0.0f;
                    if (rf_e[k] >= e1) rf_e[k] = -1;
                }
            }
            // bin sweeping duty (SgdArgs::hot_bins_v).  The lines are owned by the ROW-LOOP workgroups only: the trainer and the producers
            // never come here, and a line nobody sweeps -- the first lines are the hottest items' -- would stay unpublished all launch
            if (n_hot > 0 && !a.hot_direct && iter % n_waves == wave) {
                const SgdArgs c = cold_args();                           // (the rarely executed parts read their arguments afresh: cold_args)
                for (int line = (int)blockIdx.x - first_regular; line < hot_lines(c); line += n_regular) hot_sweep_line(c, line);
            }
            if (active && !have) {
                const SgdArgs c = cold_args();
                const uint32_t seg = rfm_perm((uint32_t)sp, (uint32_t)c.n_segments, c.seg_bits, c.epoch_key ^ 0x5bd1e995u);
                const int4 d = c.seg_desc[seg];
                u = d.x; begin = d.y; len = d.z;
                lo = c.csr_off[u]; hi = c.csr_off[u + 1];
                len_bits = (int32_t)rfm_perm_bits((uint32_t)len);
                seg_key = rfm_mix32(c.epoch_key ^ (seg * 0x9E3779B9u + 0x7F4A7C15u));
#pragma unroll
                for (int k = 0; k < KPL; ++k) {
                    vu0[k] = (sub + G * k < F) ? load_f32<FRESH>(c.v_u + (size_t)u * F + sub + G * k) : 0.0f;
                    vu[k] = vu0[k];
                }
                t = 0;
                have = true;
                step.load_ulist(lo, hi);
                xu0 = xu1 = 0.0f;
                if (c.has_uf) {
                    const float *x = c.x_uf + (size_t)u * c.n_uf;
                    if (sub < c.n_uf) xu0 = x[sub];
                    if (sub + G < c.n_uf) xu1 = x[sub + G];
                }
                // the segment's rows in visiting order, held across the lanes (row t in lane t % G, register t / G)
#pragma unroll
                for (int k = 0; k < SEGR; ++k) {
                    const int tt = sub + G * k;
                    seg_pos[k] = tt < len ? begin + (int32_t)rfm_perm((uint32_t)tt, (uint32_t)len, (uint32_t)len_bits, seg_key) : begin;
                    seg_item[k] = c.csr_items[seg_pos[k]];
                    seg_sw[k] = c.sw_csr[seg_pos[k]];
                }
                fetch_pos(__shfl(pick(seg_item, 0), lane_base), nxt);
            }
            if (active) {
                const int src = lane_base + (int)((unsigned)t % G);
                const int32_t pos = __shfl(pick(seg_pos, t), src), i = __shfl(pick(seg_item, t), src);
                const float sw = __shfl(pickf(seg_sw, t), src);
                cur = nxt;
                // (one group alone is a sequential program: a repeated (user, item) row must see the previous row's update)
                if (a.single_group) fetch_pos(i, cur);
                else if (t + 1 < len) fetch_pos(__shfl(pick(seg_item, t + 1), lane_base + (int)((unsigned)(t + 1) % G)), nxt);   // overlaps this row
                const uint32_t row_key = rfm_row_key(a.epoch_key, (uint32_t)pos);
                // a hot positive item: the workgroup's own pending updates of its row are part of the view (RowStep, HOT)
                int slot = -1;
                if (n_hot > 0 && cur.scale >= 2.0f) {
                    slot = (int)(cur.scale * 0.5f) - 1;
                    cur.scale -= 2.0f * (float)(slot + 1);
#pragma unroll
                    for (int k = 0; k < KPL; ++k)
                        if (sub + G * k < F) cur.v[k] += (float)hot_acc[slot * F + sub + G * k] * hot_unit;
                    cur.w += (float)hot_accw[slot] * hot_unit;
                }
                // the negative (:250-253) and its gathers
                uint32_t attempt = 0;
                int srow_unused;
                const int32_t j = step.next_negative(lo, hi, row_key, attempt, srow_unused);
                float vj[KPL], wj, xj0 = 0.0f, xj1 = 0.0f;
#pragma unroll
                for (int k = 0; k < KPL; ++k) vj[k] = (sub + G * k < F) ? load_f32<FRESH>(a.v_i + (size_t)j * F + sub + G * k) : 0.0f;
                // (bias and the item's step scale -- the damping scales an item's step on either side of the pair, RowStep -- in one request)
                float neg_scale_j = 1.0f;
                if (a.scale_in_pad) {
                    const float x = load_f32<FRESH>(a.w_i + (size_t)j * a.w_stride + (sub & 1));
                    wj = __shfl(x, lane_base);
                    neg_scale_j = __shfl(x, lane_base + 1);
                } else {
                    wj = load_f32<FRESH>(a.w_i + (size_t)j * a.w_stride);
                    if (a.pos_scale) neg_scale_j = a.pos_scale[j];
                }
                if (neg_scale_j >= 2.0f) neg_scale_j -= 2.0f * floorf(neg_scale_j * 0.5f);
                if (a.has_if) {
                    const float *x = a.x_if + (size_t)j * a.n_if;
                    if (sub < a.n_if) xj0 = x[sub];
                    if (sub + G < a.n_if) xj1 = x[sub + G];
                }
                // A = x_uf[u] . v_uf (:297-300), while the negative's row is on its way
                float A[KPL], Bd[KPL];
#pragma unroll
                for (int k = 0; k < KPL; ++k) A[k] = Bd[k] = 0.0f;
                if (a.has_uf) project_dense<KPL>(xu0, xu1, a.n_uf, uf_lane, A);
                // pairwise utility (:239, :256-257 regrouped: both the utility and the gradients need the item-feature terms only as
                // differences):  pu = (w_i - w_j) + (x_i - x_j).w_if + <v_u + A, v_i - v_j> + <(x_i - x_j).v_if, v_u>
                float part = 0.0f;
                const float dx0 = cur.x0 - xj0, dx1 = cur.x1 - xj1;
                if (a.has_if) {
                    project_dense<KPL>(dx0, dx1, a.n_if, if_lane, Bd);
                    if (sub < a.n_if) part = dx0 * t_wif[sub];
                    if (sub + G < a.n_if) part += dx1 * t_wif[sub + G];
                }
#pragma unroll
                for (int k = 0; k < KPL; ++k) part += (vu[k] + A[k]) * (cur.v[k] - vj[k]) + Bd[k] * vu[k];
                const float pu = (cur.w - wj) + group_sum<G>(part);
                float log_sig, d_outer;
                sigmoid_terms(pu, log_sig, d_outer);                              // :270, :276
                if (sub == 0) { ll_acc += (double)log_sig; draw_acc += 1u; }
                const float g = sw * multiplier;
                const float eta_u = eta * step.user_scale, eta_i = eta * cur.scale, eta_j = a.damp_positive_only ? eta : eta * neg_scale_j;
                float *pi = a.v_i + (size_t)i * F + sub, *pj = a.v_i + (size_t)j * F + sub;
#pragma unroll
                for (int k = 0; k < KPL; ++k) {
                    const float g_u = (cur.v[k] - vj[k]) + Bd[k];                                     // :292, :303-305
                    const float g_i = vu[k] + A[k];                                                   // :293-294, :297-300
                    const float d_u = eta_u * (g * (d_outer * g_u) - reg_a * vu[k]);                  // :308
                    const float d_i = eta_i * (g * (d_outer * g_i) - reg_a * cur.v[k]);               // :309
                    const float d_j = eta_j * (g * (d_outer * -g_i) - reg_a * vj[k]);                 // :310
                    vu[k] += d_u;
                    if (sub + G * k < F) {
                        if (slot >= 0) __hip_atomic_fetch_add(hot_acc + slot * F + sub + G * k, __float2int_rn(d_i * hot_scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else atomic_add_f32(pi + G * k, d_i);
                        atomic_add_f32(pj + G * k, d_j);
                    }
                }
                if (sub == 0) {
                    const float dwi = eta_i * (g * (d_outer * 1.0f) - reg_a * cur.w);                                  // :279
                    if (slot >= 0) __hip_atomic_fetch_add(hot_accw + slot, __float2int_rn(dwi * hot_scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else atomic_add_f32(a.w_i + (size_t)i * a.w_stride, dwi);
                    atomic_add_f32(a.w_i + (size_t)j * a.w_stride, eta_j * (g * (d_outer * -1.0f) - reg_a * wj));      // :280
                }
                // every hot_period-th toucher of a slot publishes what the workgroup has accumulated for it (a keyed coin, RowStep)
                if (slot >= 0 && __umulhi(rfm_mix32(row_key ^ 0x7A5C3B1DU), (uint32_t)a.hot_period[slot]) == 0u) {
                    const SgdArgs c = cold_args();
#pragma unroll
                    for (int k = 0; k < KPL; ++k) {
                        if (sub + G * k >= F) continue;
                        const float d = (float)__hip_atomic_exchange(hot_acc + slot * F + sub + G * k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) * hot_unit;
                        if (d != 0.0f)
                            atomic_add_f32(c.hot_direct ? a.v_i + (size_t)i * F + sub + G * k
                                                        : c.hot_bins_v + ((size_t)(blockIdx.x % kHotBins) * n_hot + slot) * F + sub + G * k, d);
                    }
                    if (sub == 0) {
                        const float d = (float)__hip_atomic_exchange(hot_accw + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) * hot_unit;
                        if (d != 0.0f) atomic_add_f32(c.hot_direct ? a.w_i + (size_t)i * a.w_stride : c.hot_bins_w + (size_t)(blockIdx.x % kHotBins) * n_hot + slot, d);
                    }
                }
                if (a.single_group) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");      // the next row reads what this one wrote
                if (++t == len) {
                    const SgdArgs c = cold_args();
#pragma unroll
                    for (int k = 0; k < KPL; ++k)
                        if (sub + G * k < F) atomic_add_f32(c.v_u + (size_t)u * F + sub + G * k, vu[k] - vu0[k]);
                    if (c.single_group) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                    have = false;
                    if (dynamic) {
                        int64_t nx = -1;
                        if (sub == 0) nx = tickets.take(c);
                        sp = __shfl(nx, lane_base);
                        active = sp >= 0;
                    } else {
                        sp += stride;
                        active = sp < c.pos_end;
                    }
                }
            }
            if (trains) {
#pragma unroll
                for (int k = 0; k < RF; ++k)
                    if (rf_e[k] >= 0) lds_tables[rf_e[k]] = rf_val[k];
            }
        }
        if (n_hot > 0) {          // publish whatever is still pending
            __syncthreads();
            for (int k = threadIdx.x; k < n_hot * F; k += blockDim.x) {
                const float d = (float)hot_acc[k] * hot_unit;
                if (d != 0.0f) atomic_add_f32(a.hot_direct ? a.v_i + (size_t)a.hot_item[k / F] * F + (k % F)
                                                           : a.hot_bins_v + (size_t)(blockIdx.x % kHotBins) * n_hot * F + k, d);
            }
            for (int k = threadIdx.x; k < n_hot; k += blockDim.x) {
                const float d = (float)hot_accw[k] * hot_unit;
                if (d != 0.0f) atomic_add_f32(a.hot_direct ? a.w_i + (size_t)a.hot_item[k] * a.w_stride : a.hot_bins_w + (size_t)(blockIdx.x % kHotBins) * n_hot + k, d);
            }
        }
        flush_counters(a, ll_acc, draw_acc);
        if (threadIdx.x == 0) atomicMax(a.feat_clock + 3, wall_clock64());
        stamp_clock(a, 1);
      }
    }
}

// host-side launcher table (rfm_sgd_inst_*.hip): [0..3] rows kernel {hogwild, hogwild+feat, serial, serial+feat},
// [4..7] segments kernel {plain, features kernel, fresh, features kernel fresh}, [8..9] segments kernel with hot-row accumulators {plain, fresh},
// [10..13] segments kernel with negative stripes {plain, fresh, hot, hot+fresh}
typedef void (*sgd_launch_fn)(const SgdArgs &, int grid, hipStream_t);

// second stream of the features path (rfm_api.hip): the tables kernel forks off the caller's stream and joins it again
struct FeatSide { hipStream_t stream; hipEvent_t fork, join; };
FeatSide *feat_side();

}  // namespace rfm
