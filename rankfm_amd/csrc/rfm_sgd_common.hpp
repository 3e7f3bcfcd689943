// rfm_sgd_common.hpp -- what every SGD kernel of the engine shares: the launch arguments (SgdArgs), lane / row-group helpers,
// the membership tests, the device MT19937, the clock stamps and counters of a launch, the dynamic segment order (SegmentTickets)
// and the hot-row bins' sweep.  See rfm_sgd.hpp for the map of the kernels.
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
    // one atomic add of the total into v_i / w_i); what is left when a launch ends is drained by the epoch tail or the export (rfm_api.hip).
    // With few workgroups (fewer than a quarter of the lines) there is little contention and a sweeping turn would take
    // long: hot_direct = 1 publishes straight into the rows.
    float *hot_bins_v;                          // [kHotBins, n_hot, F]   (hot_bin_v)
    float *hot_bins_w;                          // [kHotBins, n_hot]      (hot_bin_w)
    int32_t hot_direct;
    int32_t hot_sweep_every;                    // a wavefront's sweeping turn comes every (wavefronts x this many) iterations (BPR segments kernel)
    const unsigned int *sw_max_bits;            // bits of max |sample_weight| (plan): range of the fixed-point hot sums
    uint32_t launch_index;                      // which launch of the epoch this is (keys the step producers' row sample)
    // features kernel: the step producers hand their batches to the table trainer through `feat_ring` ([2 * n_producers] slots of
    // one staged step per row group of a workgroup), synchronised by the counters in `feat_flags` (sgd_features_kernel)
    float *feat_ring;
    unsigned int *feat_flags;
    int32_t n_producers;
    int32_t feat_frozen;                        // debug: the feature tables are not trained (no trainer, no producers)
    // dynamic segment order (segments kernel, WARP kernel, pipelined feature row loop): a row group takes its next segment from
    // a ticket counter instead of striding the order with the number of groups (SegmentTickets below); nullptr = static stride
    unsigned int *tickets;                      // the launch's counter of order positions handed out, zero at launch
    int32_t damp_positive_only;                 // experiments: the round-3 rule (an item's scale applies to its step as the POSITIVE item only)
    // Item factor rows SEGMENT-MAJOR (round 5): dword f of item i at ((f / 16) * I + i) * 16 + f % 16 of `v_i` -- the four 64-byte
    // segments of a row are lines 64 I bytes apart instead of 256 contiguous bytes.  The memory-side atomic path is slower on skewed
    // targets (a warm row's in-flight updates queue at its memory channel: profiles/r05_notes.md), and a contiguous row puts all
    // four of its segments on one channel; spread out, config 2's address mix retires 18.1 - 18.6 G requests/s instead of 16.6 -
    // 16.8 (tools/microbench/atomic_skew.hip).  The engine works on such a copy in its workspace for the length of a call (the
    // caller's v_i keeps the reference's layout before and after); full factor rows of 16-lane groups only.
    int32_t vi_split;
    // features: the table trainer applies EXACTLY table_quota staged steps per launch (rounded up to whole batches) -- a number the host
    // derives from the launch's rows and geometry, not from when the row loops happen to finish (feat_tables_kernel)
    float table_step;                           // the trainer's step length relative to eta (tune_table_step_pct; 1 = the reference's)
    float table_quiet_from;                     // the trainer stops once this share of the launch's segments is handed out (0: never; see feat_table_trainer)
    float table_pace;                           // the producers spread the quota's batches over this share of the launch's segments (0: as fast as they can; feat_step_producer)
    int64_t table_quota;
    unsigned long long *feat_clock;             // [4] wall-clock ticks: tables kernel begin | end | row-loop kernel begin | end (diagnostics)
    unsigned long long *sclk;                   // [4] workgroup 0 of the row-loop kernel: wall clock (100 MHz) at its start | end, shader cycle counter at its start | end
};
// Hot-row bins: element index of dword f of slot `slot` in bin `bin` (hot_bin_v) / of the slot's bias (hot_bin_w).
// (Round 5 measured a staggered layout -- segment-major, an odd number of lines between bins, so that the sixteen bins of a slot and the
// four segments of a bin row do not sit at power-of-two strides -- against this dense one: 2.647 / 2.694 against 2.727 / 2.662 ms on
// config 2, nothing; profiles/r05_notes.md.)
__device__ __forceinline__ size_t hot_bin_v(const SgdArgs &a, int bin, int slot, int f) { return ((size_t)bin * a.n_hot + slot) * a.n_factors + f; }
__device__ __forceinline__ size_t hot_bin_w(const SgdArgs &a, int bin, int slot) { return (size_t)bin * a.n_hot + slot; }
// element index of dword f of item `item` in SgdArgs::v_i (see vi_split)
__device__ __forceinline__ size_t vi_index(const SgdArgs &a, int32_t item, int f) {
    return a.vi_split ? ((size_t)(f >> 4) * (size_t)a.n_items + (size_t)item) * 16 + (size_t)(f & 15) : (size_t)item * a.n_factors + f;
}
constexpr int kTicketWords = 16;                // one counter per launch, on a 64-byte line of its own
constexpr int kHotBins = 16;

constexpr size_t kLdsBytes = 160 * 1024;        // per workgroup on gfx950

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
#define RFM_COLD_ARGS(c) const SgdArgs c = cold_args();

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

// One 64-byte line of the hot-row bins (see SgdArgs::hot_bins_v), swept by one wavefront: lanes = 4 bins x 16 dwords at a
// time, exchange with zero, sum over the bins, one atomic add of the total into the hot row.  Lines 0 .. n_hot*LPR-1 are
// 16-factor pieces of the hot rows (LPR = lines per row), the rest are 16 slots' biases each.
__device__ __forceinline__ int hot_lines(const SgdArgs &a) { return a.n_hot * ((a.n_factors + 15) / 16) + (a.n_hot + 15) / 16; }

__device__ __forceinline__ void hot_sweep_line(const SgdArgs &a, int line) {
    const int lane = threadIdx.x & 63, d16 = lane & 15, quad = lane >> 4;
    const int F = a.n_factors, lpr = (F + 15) / 16;
    const bool bias = line >= a.n_hot * lpr;
    float *dst;
    bool ok;
    int slot, f = 0;
    if (!bias) {
        slot = line / lpr;
        f = (line % lpr) * 16 + d16;
        ok = f < F;
        dst = a.v_i + vi_index(a, a.hot_item[slot], ok ? f : 0);
    } else {
        slot = (line - a.n_hot * lpr) * 16 + d16;
        ok = slot < a.n_hot;
        if (!ok) slot = 0;
        dst = a.w_i + (size_t)a.hot_item[slot] * a.w_stride;
    }
    float acc = 0.0f;
    if (ok) {
#pragma unroll
        for (int b = 0; b < kHotBins; b += 4) {
            float *src = bias ? a.hot_bins_w + hot_bin_w(a, b + quad, slot) : a.hot_bins_v + hot_bin_v(a, b + quad, slot, f);
            acc += __hip_atomic_exchange(src, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
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

// host-side launcher table (rfm_sgd_inst_*.hip): [0..3] rows kernel {hogwild, hogwild+feat, serial, serial+feat},
// [4..7] segments kernel {plain, features kernel, fresh, features kernel fresh}, [8..9] segments kernel with hot-row accumulators {plain, fresh}
typedef void (*sgd_launch_fn)(const SgdArgs &, int grid, hipStream_t);

// second stream of the features path (rfm_api.hip): the tables kernel forks off the caller's stream and joins it again
struct FeatSide { hipStream_t stream; hipEvent_t fork, join; };
FeatSide *feat_side();

}  // namespace rfm
