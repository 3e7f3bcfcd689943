// rfm_api.hip -- C ABI of librankfm_hip.so: host-side orchestration of `_fit` on one MI355X.
//
// Replaces the epoch wrapper of the reference's `_fit` (rankfm/_rankfm.pyx:182-228, 329-342): MT seeding,
// shape inference, learning-rate schedule, the epoch loop, the epoch-end finiteness assertion
// (assert_finite, :95-103) and the verbose penalty (reg_penalty, :106-116).  The row loop itself is the
// wavefront kernel in rfm_sgd.hpp.  See include/rankfm_hip.h for the boundary contract.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/rankfm_hip.h"
#include "rfm_sgd.hpp"

namespace rfm {
#define RFM_DECLARE_SHAPE(name) const sgd_launch_fn *sgd_table_##name();
RFM_DECLARE_SHAPE(g4_k1) RFM_DECLARE_SHAPE(g16_k1) RFM_DECLARE_SHAPE(g16_k2) RFM_DECLARE_SHAPE(g16_k3)
RFM_DECLARE_SHAPE(g16_k4) RFM_DECLARE_SHAPE(g16_k6) RFM_DECLARE_SHAPE(g16_k8) RFM_DECLARE_SHAPE(g64_k3)
RFM_DECLARE_SHAPE(g64_k4) RFM_DECLARE_SHAPE(g64_k8)

// Row-group shapes: G lanes per interaction, lane s owns factor dwords s, s+G, s+2G, ... (KPL of them).  With
// G = 16 every load / atomic instruction of a group covers one contiguous 64-byte segment of the row, which is
// what the L2 atomic path wants: the first version of this kernel used 16-byte-per-lane chunks (4 dwords per
// 64-byte segment per instruction) and ran its atomics 3.4x slower (profiles/r01_notes.md).
struct ShapeEntry { int group, kpl, max_f; const sgd_launch_fn *(*table)(); };
static const ShapeEntry kShapes[] = {
    {4, 1, 4, sgd_table_g4_k1},     {16, 1, 16, sgd_table_g16_k1},  {16, 2, 32, sgd_table_g16_k2},
    {16, 3, 48, sgd_table_g16_k3},  {16, 4, 64, sgd_table_g16_k4},  {16, 6, 96, sgd_table_g16_k6},
    {16, 8, 128, sgd_table_g16_k8}, {64, 3, 192, sgd_table_g64_k3}, {64, 4, 256, sgd_table_g64_k4},
    {64, 8, 512, sgd_table_g64_k8},
};

// smallest row-group shape that holds F factors
static const ShapeEntry *pick_shape(int F) {
    for (const ShapeEntry &s : kShapes)
        if (F <= s.max_f) return &s;
    return nullptr;
}

static thread_local std::string g_last_error;

static int hip_fail(hipError_t e, const char *what) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    g_last_error = buf;
    return RFM_ERR_HIP;
}
#define RFM_HIP(call)                                            \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return hip_fail(e_, #call);        \
    } while (0)

// ---------------------------------------------------------------------------------------------
// epoch tail: finiteness of the six weight arrays (assert_finite) and, optionally, their squared norms
// ---------------------------------------------------------------------------------------------
struct TailArgs {
    const float *ptr[6];
    unsigned long long len[6];
    double *sumsq;            // [6] or nullptr
    unsigned int *nonfinite;  // bit k set when array k holds a non-finite value
    int w_stride;             // array 0 (the item biases) is read at ptr[0][i * w_stride]: the engine's padded copy (SgdArgs::w_stride)
    int drain_hot;            // 1: the launch first folds what the SGD launches left in the hot-row bins into the rows (hot_sweep_line)
};

// `hot`: the epoch's SgdArgs (read only when t.drain_hot; the fields hot_sweep_line uses).  The drain replaces a kernel launch of its
// own behind every SGD launch (hot_reduce_kernel, ~4 us + a launch gap per epoch): what a launch's workgroups publish when they
// leave stays in the bins until the epoch's tail -- or, without a tail, until the next launch's sweeping turns or the export.
// Finiteness is not affected by where a pending sum sits (the sums are finite by construction: fixed point), the penalty's norms
// would be: a call that wants them drains with hot_reduce_kernel BEFORE this kernel.
template <bool PENALTY>
__global__ void __launch_bounds__(256) tail_kernel(const TailArgs t, const SgdArgs hot) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (!PENALTY && t.drain_hot) {
        const int n_lines = hot_lines(hot);
        for (int line = blockIdx.x * 4 + wid; line < n_lines; line += gridDim.x * 4) hot_sweep_line(hot, line);
    }
    unsigned bad = 0;
    for (int k = 0; k < 6; ++k) {
        const float *p = t.ptr[k];
        const unsigned long long n = t.len[k];
        if (k == 0 && t.w_stride > 1) {                     // (the padded biases: one dword per 64-byte line)
            float acc = 0.0f;
            unsigned b = 0;
            for (unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (unsigned long long)gridDim.x * blockDim.x) {
                const float v = p[idx * (unsigned long long)t.w_stride];
                b |= ((__float_as_uint(v) & 0x7f800000u) == 0x7f800000u);
                if (PENALTY) acc += v * v;
            }
            if (b) bad |= 1u;
            if (PENALTY) {
                double d = (double)acc;
#pragma unroll
                for (int m = 32; m > 0; m >>= 1) d += __shfl_xor(d, m);
                if (lane == 0) red[wid] = d;
                __syncthreads();
                if (threadIdx.x == 0) {
                    const double tot = red[0] + red[1] + red[2] + red[3];
                    if (tot != 0.0) unsafeAtomicAdd(t.sumsq + k, tot);
                }
                __syncthreads();
            }
            continue;
        }
        const unsigned long long n4 = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? (n >> 2) : 0;
        float acc = 0.0f;
        unsigned b = 0;
        const float4 *p4 = reinterpret_cast<const float4 *>(p);
        for (unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n4;
             idx += (unsigned long long)gridDim.x * blockDim.x) {
            const float4 v = p4[idx];
            const float s = v.x + v.y + v.z + v.w;
            // a sum of four finite floats can only be non-finite by overflow near 3.4e38, which assert_finite's
            // np.sum would flag as well
            b |= ((__float_as_uint(s) & 0x7f800000u) == 0x7f800000u);
            if (PENALTY) acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (unsigned long long idx = (n4 << 2) + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
             idx += (unsigned long long)gridDim.x * blockDim.x) {
            const float v = p[idx];
            b |= ((__float_as_uint(v) & 0x7f800000u) == 0x7f800000u);
            if (PENALTY) acc += v * v;
        }
        if (b) bad |= (1u << k);
        if (PENALTY) {
            double d = (double)acc;
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) d += __shfl_xor(d, m);
            if (lane == 0) red[wid] = d;
            __syncthreads();
            if (threadIdx.x == 0) {
                const double tot = red[0] + red[1] + red[2] + red[3];
                if (tot != 0.0) unsafeAtomicAdd(t.sumsq + k, tot);
            }
            __syncthreads();
        }
    }
    if (bad) atomicOr(t.nonfinite, bad);
}

// a user whose list holds every item would make the rejection sampler spin forever (rankfm/_rankfm.pyx:250-253)
// The lists may hold duplicates (the reference keeps repeated (user, item) rows, rankfm/rankfm.py:174, and trains them
// fine): what matters is the number of DISTINCT items, counted in the sorted list -- only for the rare list that is long
// enough to matter.
__global__ void degree_check_kernel(const int64_t *__restrict__ off, const int32_t *__restrict__ items, int n_users, int n_items,
                                    unsigned int *flag) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_users) return;
    const int64_t lo = off[u], hi = off[u + 1];
    if (hi - lo < (int64_t)n_items) return;
    int64_t distinct = 1;
    for (int64_t k = lo + 1; k < hi; ++k) distinct += (items[k] != items[k - 1]);
    if (distinct >= (int64_t)n_items) atomicOr(flag, 2u);
}

// ---------------------------------------------------------------------------------------------
// Hogwild plan: per-item step scale from item popularity (see SgdArgs::pos_scale)
// ---------------------------------------------------------------------------------------------
constexpr int kBiasStride = 16;
// item biases <-> their padded copy (SgdArgs::w_stride); PAD: w_pad[i * 16] = w_i[i], else the reverse
// (dword 1 of an item's line carries its step scale pos_scale[i], so that bias and scale cost the SGD kernel one request)
template <bool PAD>
__global__ void bias_pad_kernel(float *__restrict__ w_i, float *__restrict__ w_pad, const float *__restrict__ pos_scale, int n_items) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    if (PAD) {
        w_pad[(size_t)i * kBiasStride] = w_i[i];
        w_pad[(size_t)i * kBiasStride + 1] = pos_scale ? pos_scale[i] : 1.0f;
    } else w_i[i] = w_pad[(size_t)i * kBiasStride];
}

// item factor rows row-major <-> segment-major (SgdArgs::vi_split): dword f of item i at ((f / 16) * I + i) * 16 + f % 16.  One thread
// per 16-byte quarter of a 64-byte segment; both sides are 16-byte aligned (F is a multiple of 16).
template <bool TO_SPLIT>
__global__ void __launch_bounds__(256) vi_split_kernel(float4 *__restrict__ row_major, float4 *__restrict__ seg_major, int n_items, int n_segs) {
    const size_t total = (size_t)n_items * n_segs * 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x) {
        // q walks the segment-major copy: (seg, item, quarter)
        const size_t seg = q / ((size_t)n_items * 4), rest = q % ((size_t)n_items * 4);
        const size_t item = rest >> 2, quarter = rest & 3;
        const size_t r = (item * n_segs + seg) * 4 + quarter;
        if (TO_SPLIT) seg_major[q] = row_major[r]; else row_major[r] = seg_major[q];
    }
}

__global__ void item_count_kernel(const int32_t *__restrict__ interactions, long long n, int *__restrict__ count) {
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x)
        atomicAdd(count + interactions[2 * r + 1], 1);
}

// sample weights re-ordered to CSR positions: row r = (u, i) takes the first free slot among the positions of user u that
// hold item i (duplicates of a pair occupy consecutive positions).  `sw_csr` is pre-filled with the sentinel 0xFFFFFFFF.
__global__ void sw_to_csr_kernel(const int32_t *__restrict__ interactions, const float *__restrict__ sw, long long n,
                                 const int64_t *__restrict__ off, const int32_t *__restrict__ items, unsigned int *sw_csr,
                                 unsigned int *error_flags, unsigned int *sw_max_bits) {
    float sw_max = 0.0f;          // largest |sample weight|: scales the fixed-point hot-row accumulators (SgdArgs::sw_max_bits)
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
        const int32_t u = interactions[2 * r], i = interactions[2 * r + 1];
        int64_t lo = off[u];
        const int64_t end = off[u + 1];
        int64_t hi = end;
        while (lo < hi) {                                   // lower bound of i in the user's sorted list
            const int64_t md = lo + ((hi - lo) >> 1);
            if (items[md] < i) lo = md + 1; else hi = md;
        }
        bool placed = false;
        for (int64_t slot = lo; slot < end && items[slot] == i; ++slot)
            if (atomicCAS(sw_csr + slot, 0xFFFFFFFFu, __float_as_uint(sw[r])) == 0xFFFFFFFFu) { placed = true; break; }
        if (!placed) atomicOr(error_flags, 4u);             // the CSR lists do not describe these interactions
        sw_max = fmaxf(sw_max, fabsf(sw[r]));
    }
    for (int m = 32; m > 0; m >>= 1) sw_max = fmaxf(sw_max, __shfl_xor(sw_max, m));
    if ((threadIdx.x & 63) == 0 && sw_max > 0.0f) atomicMax(sw_max_bits, __float_as_uint(sw_max));   // non-negative floats order like their bits
}

// ---------------------------------------------------------------------------------------------
// workspace layout (device)
// ---------------------------------------------------------------------------------------------
struct Workspace {
    float *pos_scale;             // [I]     persistent across calls (plan_is_cached)
    float *sw_csr;                // [N]     persistent
    int4 *seg_desc;               // [<= U + N / min_segment_rows]  persistent
    int32_t *hot_item;            // [kMaxHot] persistent
    int32_t *hot_period;          // [kMaxHot] persistent
    unsigned int *sw_max_bits;    // bits of max |sample_weight| (persistent, written with the plan)
    float *feat_ring;             // features kernel: [2 * kFeatMaxProducers] batches of staged steps (never read before written)
    double *ll;                   // [epochs]
    unsigned long long *draws;    // [epochs]
    double *sumsq;                // [epochs][6]
    unsigned int *nonfinite;      // [epochs]
    unsigned int *error_flags;    // [1] (+pad)
    uint32_t *mt_state;           // [625] (+pad)
    float *multiplier;            // [max_samples + 1]
    // the ENGINE LAYOUT of the item-side weights, in front of everything that is zeroed per call: a call imports the caller's v_i / w_i into
    // it and exports them again when it returns, unless the caller keeps the layout between calls (rfm_fit_config.keep_layout)
    size_t layout_offset;
    float *w_pad;                     // [n_items * kBiasStride] item biases, one 64-byte line each (SgdArgs::w_stride)
    float *hot_bins_v, *hot_bins_w;   // [kHotBins, n_hot, F], [kHotBins, n_hot]: pending hot-row sums (zeroed at import, drained by the epoch tail / the export)
    float *vi_split;                  // [I, F] segment-major working copy of the item factor rows (SgdArgs::vi_split), or nullptr
    size_t volatile_offset;       // everything from here on is zeroed at the start of every call
    unsigned int *feat_flags;     // [kFeatFlagWords] producer / trainer hand-shake of the features kernel (zero between launches)
    unsigned long long *feat_clock;   // [4] wall-clock ticks of the last launch's tables kernel (begin, end) and row-loop kernel (begin, end)
    unsigned long long *sclk;         // [4] SgdArgs::sclk of the last launch
    unsigned int *tickets;        // [epochs, windows_per_epoch, kTicketWords] segment ticket heads, one set per launch
    int64_t windows_per_epoch;
    size_t bytes;
};

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// the experiments' overrides of a call (rfm_fit_config.tuning; NULL = production = all zero)
static rfm_fit_tuning tuning_of(const rfm_fit_config *c) {
    rfm_fit_tuning t;
    memset(&t, 0, sizeof t);
    if (c && c->tuning) t = *c->tuning;
    return t;
}

constexpr int kMaxHot = 128;                     // most hot-row accumulator slots per workgroup a plan can have (LDS: slots * (F + 2) floats)
constexpr int kHotSlots = 64;                    // ... and what a plan takes unless told otherwise (rfm_fit_tuning.hot_slots)
// Publications of a hot row per epoch and workgroup (rfm_fit_tuning.hot_publications).  A publication is not only five atomic requests
// into the bins: with 48 of them every seventh row of config 2 publishes, i.e. nearly every second wavefront row step runs the publication
// block for one of its four rows.  Measured at full size against the sequential oracle (tools/pub_margin.py; epoch 1 / 2 log-likelihood,
// |w_i|, kernel) -- round 5: 48: +0.40 % / -0.04 %, +0.25 %, 2.67 ms; 32: +0.55 % / -0.05 %, +0.36 %, 2.58 ms; 24: +0.73 % / -0.07 %,
// +0.50 %, 2.52 ms; 16: +1.08 % / -0.10 %, +0.75 %, 2.49 ms.  Round 6 (the bins drained by the epoch tail instead of behind every launch;
// two runs each, profiles/r06_notes.md): 32: +0.52 % / -0.05 %, +0.35 %, 2.59 ms; 24: +0.70 % / -0.06 %, +0.49 %, 2.555 ms; 20: +0.84 % /
// -0.08 %, +0.59 %, 2.545 ms.  24 since round 6: it keeps 0.3 of the parity bound (1 %) as margin; below it a
// publication saved buys less and less time (the requests it saves are ~0.2 of 9.4 per update) for the same step in the log-likelihood.
constexpr double kHotPublications = 24.0;
// Row steps between two sweeps of a workgroup's bin lines (BPR segments kernel; SgdArgs::hot_sweep_every).  A sweep is sixteen returning
// exchanges and an add per line: at one sweep per row step 1.8 M of a config-2 launch's 47 M memory-side requests.  What a publication
// waits in a bin (~4 us at every step) is small beside what it waited in LDS before (~50 us at 24 publications per epoch), so sweeping
// less often buys more time per unit of log-likelihood than publishing less often.  Measured (round 6, one box, interleaved sessions;
// first-epoch log-likelihood against the oracle at full size, tools/pub_margin.py): every step 2.551 ms / +0.70 %; every 2nd 2.513;
// every 4th 2.505 / +0.81 % (with 32 publications 2.532 / +0.65 %); every 8th 2.495 / +1.04 % -- over the 1 % bound.
constexpr int kHotSweepEvery = 4;
// Table trainer quota on chip-filling launches: every (kTableQuotaFactor x row groups / 64)-th row -- the 446th on a full chip.  Round 4
// found ranking quality a HUMP in it (denser: the trainer ran to the launch's end and cost the rows their quiet period, -3.8 points of
// hit_rate@10 at every 250th row; sparser: -1.2 at the 600th).  Since round 5 a quota denser than the default makes the trainer stop by itself once 80 % of the
// launch's segments are handed out (kTableQuietFrom, rfm_sgd_features.hpp) and the dense side has no cliff -- config-2 shape with tags, three seeds x
// two runs against the oracle's 0.3792: every 123rd / 223rd / 246th / 300th row -0.34 / -0.29 / -0.10 / -0.29 point, 446th -0.77 ...
// -0.17 (two runs of the sweep), 491st -0.17, 650th -1.2, 892nd -1.4 (profiles/r05_notes.md section 8).  A denser default (1.3 x, every
// 246th row) was measured too: flat from x 0.5 to x 2 at that shape, but on config 4's share -- 32 + 32 tags without signal, learning
// rate 0.03 -- the tables' norms leave the oracle's (first epoch |w_if| +47 %, |v_if| +18 %, |w_i| +2.7 % against -9 %, +4.5 %, -1.1 %
// at 2.4 x), so 2.4 x stays: the tables track the oracle's to 10 % there, and the sparse side (x 2) remains the open end of row a6.
constexpr double kTableQuotaFactor = 2.4;
// share of a launch's segments over which the trainer's quota is spread (SgdArgs::table_pace): where the default quota ends by itself
constexpr float kTablePace = 0.65f;

// (a user of degree d is cut into ceil(d / rows) <= d / rows + 1 segments)
static size_t max_segments(int64_t n_rows, int n_users, int seg_rows) { return (size_t)n_users + (size_t)(n_rows / seg_rows) + 1; }
// shortest segment length a plan of this call may use: kSegmentRows, or the caller's override when that is shorter
static int min_segment_rows(const rfm_fit_config *c) {
    const int rows = tuning_of(c).segment_rows;
    return rows > 0 && rows < kSegmentRows ? rows : kSegmentRows;
}

// floats of the features kernel's step ring: 2 slots per producer, one staged step (1 + 2F + P + Q floats) per row group of a
// 1024-thread workgroup
static size_t feat_ring_floats(const rfm_fit_config *c) {
    if (!c->has_user_features && !c->has_item_features) return 0;
    const ShapeEntry *sh = pick_shape(c->n_factors);
    const size_t gpb = 1024 / (size_t)(sh ? sh->group : 16);
    return 2 * (size_t)kFeatMaxProducers * gpb * (1 + 2 * (size_t)c->n_factors + (size_t)c->n_user_features + (size_t)c->n_item_features);
}

// launches an epoch can be cut into (rows_per_launch) + the opening launch of a fit with features: every launch has ticket heads of its own
static int64_t ticket_windows(const rfm_fit_config *c) {
    const int rpl = tuning_of(c).rows_per_launch;
    const int64_t w = rpl > 0 ? c->n_interactions / rpl + 3 : 2;
    return w < 4096 ? w : 4096;        // (beyond that the launches fall back to the static segment stride)
}

static Workspace carve(void *base, int epochs, int max_samples, int n_items, int n_users, int64_t n_rows, size_t n_ring, int n_factors,
                       int seg_rows_min, int64_t windows_per_epoch, bool vi_split) {
    Workspace w;
    char *p = (char *)base;
    size_t o = 0;
    // ---- the plan: persistent across calls (plan_token)
    w.pos_scale = (float *)(p + o);              o += align_up(sizeof(float) * (size_t)n_items);
    w.sw_csr = (float *)(p + o);                 o += align_up(sizeof(float) * (size_t)(n_rows > 0 ? n_rows : 1));
    w.seg_desc = (int4 *)(p + o);                o += align_up(sizeof(int4) * max_segments(n_rows, n_users, seg_rows_min));
    w.hot_item = (int32_t *)(p + o);             o += align_up(sizeof(int32_t) * kMaxHot);
    w.hot_period = (int32_t *)(p + o);           o += align_up(sizeof(int32_t) * kMaxHot);
    w.sw_max_bits = (unsigned int *)(p + o);     o += align_up(sizeof(unsigned int));
    w.feat_ring = (float *)(p + o);              o += align_up(sizeof(float) * n_ring);
    // ---- the engine layout of the item-side weights: persistent across calls that keep it (layout_token).  Its place must not depend
    //      on the number of epochs of a call (the arrays behind it do): a kept layout is found again by a call of another length.
    w.layout_offset = o;
    w.w_pad = (float *)(p + o);                  o += align_up(sizeof(float) * (size_t)kBiasStride * (size_t)n_items);
    w.hot_bins_v = (float *)(p + o);             o += align_up(sizeof(float) * (size_t)kHotBins * kMaxHot * (size_t)n_factors);
    w.hot_bins_w = (float *)(p + o);             o += align_up(sizeof(float) * (size_t)kHotBins * kMaxHot);
    w.vi_split = vi_split ? (float *)(p + o) : nullptr;
    if (vi_split) o += align_up(sizeof(float) * (size_t)n_items * (size_t)n_factors);
    // ---- everything from here on is zeroed at the start of every call
    w.volatile_offset = o;
    w.ll = (double *)(p + o);                    o += align_up(sizeof(double) * epochs);
    w.draws = (unsigned long long *)(p + o);     o += align_up(sizeof(unsigned long long) * epochs);
    w.sumsq = (double *)(p + o);                 o += align_up(sizeof(double) * 6 * epochs);
    w.nonfinite = (unsigned int *)(p + o);       o += align_up(sizeof(unsigned int) * epochs);
    w.error_flags = (unsigned int *)(p + o);     o += align_up(sizeof(unsigned int) * 16);
    w.feat_clock = (unsigned long long *)(p + o);    o += align_up(sizeof(unsigned long long) * 4);      // (read back with the results: keep behind error_flags)
    w.sclk = (unsigned long long *)(p + o);          o += align_up(sizeof(unsigned long long) * 4);      // (likewise)
    w.mt_state = (uint32_t *)(p + o);            o += align_up(sizeof(uint32_t) * 640);
    w.multiplier = (float *)(p + o);             o += align_up(sizeof(float) * ((size_t)max_samples + 1));
    w.feat_flags = (unsigned int *)(p + o);      o += align_up(sizeof(unsigned int) * kFeatFlagWords);
    w.windows_per_epoch = windows_per_epoch;
    w.tickets = (unsigned int *)(p + o);         o += align_up(sizeof(unsigned int) * kTicketWords * (size_t)windows_per_epoch * (size_t)epochs);
    w.bytes = o;
    return w;
}

// calls that MAY run on segment-major item rows (SgdArgs::vi_split; the workspace then holds the copy): BPR without features, Hogwild,
// full factor rows of 16-lane row groups (k = 16, 32, 48, 64, 96); debug_flags bit 9 keeps the rows row-major (experiments)
static bool vi_split_eligible(const rfm_fit_config *c) {
    const ShapeEntry *sh = pick_shape(c->n_factors);
    // (Models with features stay row-major: measured on config 4's share, the pipelined feature row loop on segment-major rows runs
    //  4.12 against 3.82 - 3.85 ms -- it is bound by its 168 registers and its latency chain, not by the atomic path, and the
    //  per-segment addresses cost it two spills: profiles/r05_notes.md.  WARP reads ~23 candidate rows per update: row-major too.)
    // (k = 128: the segment-major instantiation needs more than the 128 registers a 16-wavefront workgroup has -- 31 spilled -- and stays row-major)
    return sh && sh->group == 16 && sh->kpl <= 6 && c->n_factors == sh->group * sh->kpl && c->max_samples == 1 && !c->has_user_features && !c->has_item_features &&
           c->mode == RFM_MODE_HOGWILD && !(tuning_of(c).debug_flags & 512);
}

static size_t feat_table_floats(const rfm_fit_config *c) {
    return (size_t)(c->n_user_features + c->n_item_features) * (size_t)c->n_factors + (size_t)c->n_item_features;
}
constexpr size_t kMaxLdsTableFloats = 16384;      // 64 KiB of LDS per workgroup for the feature-table replica

static int validate(const rfm_fit_config *c) {
    if (!c) return RFM_ERR_BAD_ARG;
    if (c->n_interactions < 0 || c->n_interactions > 0x7fffffffLL) return RFM_ERR_BAD_ARG;   // int32 row ids, like the reference
    if (c->n_users < 1 || c->n_items < 2 || c->n_user_features < 1 || c->n_item_features < 1 || c->n_factors < 1)
        return RFM_ERR_BAD_ARG;
    if (c->max_samples < 1 || c->epochs < 1 || c->epoch_begin < 0 || c->rng_epoch_offset < 0) return RFM_ERR_BAD_ARG;
    if (c->epoch_parts > 1 && (c->epoch_part_index < 0 || c->epoch_part_index >= c->epoch_parts)) return RFM_ERR_BAD_ARG;
    if (c->learning_schedule != RFM_SCHEDULE_CONSTANT && c->learning_schedule != RFM_SCHEDULE_INVSCALING)
        return RFM_ERR_UNKNOWN_SCHEDULE;
    if (c->mode != RFM_MODE_HOGWILD && c->mode != RFM_MODE_SERIAL) return RFM_ERR_BAD_ARG;
    const rfm_fit_tuning t = tuning_of(c);
    if (t.segment_rows < 0 || t.segment_rows > kSegmentRows || t.hot_publications < 0 || t.hot_publications > 65536 ||
        t.feature_waves < 0 || t.feature_waves > 16 || t.table_producers < 0 || t.table_producers > kFeatMaxProducers ||
        t.table_every < 0 || t.table_step_pct < 0 || t.table_step_pct > 400 || t.table_batch < 0 || t.table_batch > 256 || (t.table_batch & 3) ||
        t.n_workgroups < 0 || t.rows_per_launch < 0 || t.debug_shape < 0 || t.table_pace_pct < -1 || t.table_pace_pct > 100 || t.hot_sweep_every < 0 || t.hot_sweep_every > 64 || t.hot_slots < 0 || t.hot_slots > 128)
        return RFM_ERR_BAD_ARG;
    if (c->keep_layout != 0 && c->keep_layout != 1) return RFM_ERR_BAD_ARG;
    if (c->layout_token < 0 || (c->layout_token != 0 && c->plan_token <= 0)) return RFM_ERR_BAD_ARG;      // (a kept layout lives with its plan)
    if (c->rng != RFM_RNG_MT19937 && c->rng != RFM_RNG_COUNTER) return RFM_ERR_BAD_ARG;
    if (c->rng == RFM_RNG_MT19937 && c->mode != RFM_MODE_SERIAL) return RFM_ERR_BAD_ARG;   // one serial stream
    if (!pick_shape(c->n_factors)) return RFM_ERR_UNSUPPORTED;
    return RFM_OK;
}

static int g_sm_count = 0;

// The second stream of the features path: the table trainer's kernel runs beside the row-loop kernel of the caller's stream
// (launch_segments, rfm_sgd_inst.inc).  One per device, created on first use, never destroyed (process lifetime).
FeatSide *feat_side() {
    static FeatSide sides[64];
    static bool made[64];
    static std::mutex mu;                       // (creation only; the launches that USE the side stream are serialised per device: g_fit_mutex)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!made[dev]) {
        if (hipStreamCreateWithFlags(&sides[dev].stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&sides[dev].fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&sides[dev].join, hipEventDisableTiming) != hipSuccess) return nullptr;
        made[dev] = true;
    }
    return &sides[dev];
}

static int device_ok() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RFM_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return RFM_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_last_error = std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only";
        return RFM_ERR_NO_DEVICE;
    }
    g_sm_count = prop.multiProcessorCount;
    return RFM_OK;
}

// ---------------------------------------------------------------------------------------------
// HBM stream probe (rfm_hbm_probe): what this box's memory system delivers to a plain streaming kernel, measured next to the
// SGD kernel so that the roofline fraction can also be read against an ACHIEVABLE peak, not only the 8 TB/s of the data sheet
// ---------------------------------------------------------------------------------------------
template <bool COPY>
__global__ void __launch_bounds__(256) stream_probe_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4, float *sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.0f;
    // four independent 16-byte accesses per lane in flight
    for (; idx + 3 * stride < n4; idx += 4 * stride) {
        const float4 a = src[idx], b = src[idx + stride], c = src[idx + 2 * stride], d = src[idx + 3 * stride];
        if (COPY) { dst[idx] = a; dst[idx + stride] = b; dst[idx + 2 * stride] = c; dst[idx + 3 * stride] = d; }
        else acc += (a.x + b.y) + (c.z + d.w);
    }
    for (; idx < n4; idx += stride) {
        const float4 a = src[idx];
        if (COPY) dst[idx] = a; else acc += a.x;
    }
    if (!COPY && acc == 123456.789f) *sink = acc;        // keeps the loads alive
}

// ---------------------------------------------------------------------------------------------
// multi-GPU exchange (rankfm_amd/distributed.py): one pass over the flat bucket of item-side tables on each side of the all-reduce
//   begin:  flat <- flat - start                      (this rank's deltas of the epoch)
//   finish: flat <- start + scale .* flat             (start + damped sum of all ranks' deltas; scale per element or uniform)
// ---------------------------------------------------------------------------------------------
template <bool FINISH>
__global__ void __launch_bounds__(256) delta_kernel(float *__restrict__ flat, const float *__restrict__ start, const float *__restrict__ scale,
                                                    float uniform, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n4 = n >> 2;
    const bool vec = ((reinterpret_cast<uintptr_t>(flat) | reinterpret_cast<uintptr_t>(start) | reinterpret_cast<uintptr_t>(scale)) & 15) == 0;
    size_t done = 0;
    if (vec) {
        float4 *f4 = reinterpret_cast<float4 *>(flat);
        const float4 *s4 = reinterpret_cast<const float4 *>(start), *c4 = reinterpret_cast<const float4 *>(scale);
        for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += stride) {
            float4 f = f4[k];
            const float4 s = s4[k];
            if (FINISH) {
                float4 c = make_float4(uniform, uniform, uniform, uniform);
                if (scale) c = c4[k];
                f.x = s.x + c.x * f.x; f.y = s.y + c.y * f.y; f.z = s.z + c.z * f.z; f.w = s.w + c.w * f.w;
            } else { f.x -= s.x; f.y -= s.y; f.z -= s.z; f.w -= s.w; }
            f4[k] = f;
        }
        done = n4 << 2;
    }
    for (size_t k = done + (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride)
        flat[k] = FINISH ? start[k] + (scale ? scale[k] : uniform) * flat[k] : flat[k] - start[k];
}

}  // namespace rfm

using namespace rfm;

extern "C" {

int rfm_abi_version(void) { return RFM_ABI_VERSION; }

const char *rfm_last_error(void) { return g_last_error.c_str(); }

const char *rfm_status_string(int s) {
    switch (s) {
        case RFM_OK: return "ok";
        case RFM_ERR_BAD_ARG: return "bad argument";
        case RFM_ERR_UNKNOWN_SCHEDULE: return "unknown [learning_schedule]";
        case RFM_ERR_NO_DEVICE: return "no MI355X (gfx950) device available - the engine has no CPU fallback";
        case RFM_ERR_HIP: return "HIP runtime error";
        case RFM_ERR_UNSUPPORTED: return "unsupported shape";
        case RFM_ERR_USER_SATURATED: return "a user has interacted with every item - negative sampling cannot terminate";
        case RFM_ERR_WORKSPACE: return "workspace missing or too small";
        case RFM_ERR_NONFINITE + 0: return "item weights [w_i] are not finite - try decreasing feature/sample_weight magnitudes";
        case RFM_ERR_NONFINITE + 1: return "item feature weights [w_if] are not finite - try decreasing feature/sample_weight magnitudes";
        case RFM_ERR_NONFINITE + 2: return "user factors [v_u] are not finite - try decreasing feature/sample_weight magnitudes";
        case RFM_ERR_NONFINITE + 3: return "item factors [v_i] are not finite - try decreasing feature/sample_weight magnitudes";
        case RFM_ERR_NONFINITE + 4: return "user-feature factors [v_uf] are not finite - try decreasing feature/sample_weight magnitudes";
        case RFM_ERR_NONFINITE + 5: return "item-feature factors [v_if] are not finite - try decreasing feature/sample_weight magnitudes";
        default: return "unknown status";
    }
}

int rfm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

int rfm_fit_supported(const rfm_fit_config *cfg) { return validate(cfg); }

int rfm_delta_begin(float *dev_flat, const float *dev_start, size_t n, void *hip_stream) {
    if (!dev_flat || !dev_start) return RFM_ERR_BAD_ARG;
    if (n == 0) return RFM_OK;
    const int grid = (int)std::min<size_t>(2048, (n / 4 + 255) / 256 + 1);
    delta_kernel<false><<<dim3(grid), dim3(256), 0, (hipStream_t)hip_stream>>>(dev_flat, dev_start, nullptr, 1.0f, n);
    RFM_HIP(hipGetLastError());
    return RFM_OK;
}

int rfm_delta_finish(float *dev_flat, const float *dev_start, const float *dev_scale, float uniform_scale, size_t n, void *hip_stream) {
    if (!dev_flat || !dev_start) return RFM_ERR_BAD_ARG;
    if (n == 0) return RFM_OK;
    const int grid = (int)std::min<size_t>(2048, (n / 4 + 255) / 256 + 1);
    delta_kernel<true><<<dim3(grid), dim3(256), 0, (hipStream_t)hip_stream>>>(dev_flat, dev_start, dev_scale, uniform_scale, n);
    RFM_HIP(hipGetLastError());
    return RFM_OK;
}

int rfm_hbm_probe(size_t bytes, int iters, double *read_gbps, double *copy_gbps) {
    int rc = device_ok();
    if (rc != RFM_OK) return rc;
    if (bytes < (1u << 20) || iters < 1) return RFM_ERR_BAD_ARG;
    const size_t n4 = bytes / sizeof(float4);
    float4 *src = nullptr, *dst = nullptr;
    float *sink = nullptr;
    struct Free { float4 *&a, *&b; float *&c; ~Free() { (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); } } guard{src, dst, sink};
    RFM_HIP(hipMalloc((void **)&src, n4 * sizeof(float4)));
    RFM_HIP(hipMalloc((void **)&dst, n4 * sizeof(float4)));
    RFM_HIP(hipMalloc((void **)&sink, sizeof(float)));
    RFM_HIP(hipMemset(src, 0, n4 * sizeof(float4)));
    RFM_HIP(hipMemset(dst, 0, n4 * sizeof(float4)));
    hipEvent_t e0, e1;
    RFM_HIP(hipEventCreate(&e0));
    RFM_HIP(hipEventCreate(&e1));
    const int grid = (g_sm_count > 0 ? g_sm_count : 256) * 8;
    double best[2] = {0.0, 0.0};
    for (int mode = 0; mode < 2; ++mode) {
        for (int it = 0; it < iters + 1; ++it) {            // (first pass untimed)
            RFM_HIP(hipEventRecord(e0, nullptr));
            if (mode == 0) stream_probe_kernel<false><<<dim3(grid), dim3(256), 0, nullptr>>>(src, dst, n4, sink);
            else stream_probe_kernel<true><<<dim3(grid), dim3(256), 0, nullptr>>>(src, dst, n4, sink);
            RFM_HIP(hipEventRecord(e1, nullptr));
            RFM_HIP(hipEventSynchronize(e1));
            float ms = 0.0f;
            RFM_HIP(hipEventElapsedTime(&ms, e0, e1));
            const double gbps = (double)(n4 * sizeof(float4)) * (mode == 0 ? 1.0 : 2.0) / ((double)ms * 1e-3) / 1e9;
            if (it > 0 && gbps > best[mode]) best[mode] = gbps;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (read_gbps) *read_gbps = best[0];
    if (copy_gbps) *copy_gbps = best[1];
    return RFM_OK;
}

size_t rfm_fit_workspace_bytes(const rfm_fit_config *cfg) {
    if (validate(cfg) != RFM_OK) return 0;
    return carve(nullptr, cfg->epochs, cfg->max_samples, cfg->n_items, cfg->n_users, cfg->n_interactions, feat_ring_floats(cfg), cfg->n_factors, min_segment_rows(cfg), ticket_windows(cfg), vi_split_eligible(cfg)).bytes;
}

// `host_offsets`: the caller's HOST copy of the CSR offsets, when it has one (rfm_fit_host): the planner then cuts the user segments
// from it instead of reading the device copy back (one 8 (U + 1)-byte transfer and one stream synchronisation less per planned call)
static int fit_device_impl(const rfm_fit_config *cfg, const rfm_fit_buffers *b, void *hip_stream, rfm_fit_report *rep, const int64_t *host_offsets);

static std::mutex &fit_mutex(int device);

int rfm_fit_device(const rfm_fit_config *cfg, const rfm_fit_buffers *b, void *hip_stream, rfm_fit_report *rep) {
    // models with features fork their tables kernel onto the engine's ONE side stream per device: such calls take the device's lock
    // (rfm_fit_host holds it for every call); everything else runs on the caller's stream and buffers only
    if (cfg && (cfg->has_user_features || cfg->has_item_features)) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(fit_mutex(dev));
        return fit_device_impl(cfg, b, hip_stream, rep, nullptr);
    }
    return fit_device_impl(cfg, b, hip_stream, rep, nullptr);
}

int rfm_fit_export_weights(const rfm_fit_config *cfg, const rfm_fit_buffers *b, void *hip_stream) {
    int rc = validate(cfg);
    if (rc != RFM_OK) return rc;
    if (!b || !b->w_i || !b->v_i || !b->workspace) return RFM_ERR_BAD_ARG;
    if (cfg->layout_token == 0) return RFM_OK;
    const Workspace ws = carve(b->workspace, cfg->epochs, cfg->max_samples, cfg->n_items, cfg->n_users, cfg->n_interactions, feat_ring_floats(cfg), cfg->n_factors,
                               min_segment_rows(cfg), ticket_windows(cfg), vi_split_eligible(cfg));
    if (b->workspace_bytes < ws.volatile_offset) return RFM_ERR_WORKSPACE;      // (the plan and the layout: what this call touches)
    hipStream_t stream = (hipStream_t)hip_stream;
    const bool pad = (cfg->layout_token & 2) != 0, split = (cfg->layout_token & 4) != 0;
    const int n_hot = (int)((cfg->layout_token >> 8) & 0xFF);
    if (!(cfg->layout_token & 1) || (split && !ws.vi_split) || n_hot > kMaxHot) return RFM_ERR_BAD_ARG;
    if (n_hot > 0) {
        SgdArgs a;
        memset(&a, 0, sizeof a);
        a.v_i = split ? ws.vi_split : b->v_i; a.w_i = pad ? ws.w_pad : b->w_i; a.w_stride = pad ? kBiasStride : 1;
        a.n_items = cfg->n_items; a.n_factors = cfg->n_factors; a.vi_split = split ? 1 : 0;
        a.hot_item = ws.hot_item; a.n_hot = n_hot; a.hot_bins_v = ws.hot_bins_v; a.hot_bins_w = ws.hot_bins_w;
        const int lines = n_hot * ((cfg->n_factors + 15) / 16) + (n_hot + 15) / 16;
        hipLaunchKernelGGL(hot_reduce_kernel, dim3((lines + 3) / 4), dim3(256), 0, stream, a);
    }
    if (pad) bias_pad_kernel<false><<<dim3((cfg->n_items + 255) / 256), dim3(256), 0, stream>>>(b->w_i, ws.w_pad, nullptr, cfg->n_items);
    if (split) {
        const int vi_segs = cfg->n_factors / 16;
        const int vi_grid = (int)std::min<size_t>(4096, ((size_t)cfg->n_items * vi_segs * 4 + 255) / 256);
        vi_split_kernel<false><<<dim3(vi_grid), dim3(256), 0, stream>>>((float4 *)b->v_i, (float4 *)ws.vi_split, cfg->n_items, vi_segs);
    }
    RFM_HIP(hipGetLastError());
    return RFM_OK;
}

static int fit_device_impl(const rfm_fit_config *cfg, const rfm_fit_buffers *b, void *hip_stream, rfm_fit_report *rep, const int64_t *host_offsets) {
    int rc = validate(cfg);
    if (rc != RFM_OK) return rc;
    const rfm_fit_tuning T = tuning_of(cfg);
    if (!b || !b->interactions || !b->sample_weight || !b->csr_offsets || !b->csr_items || !b->x_uf || !b->x_if ||
        !b->w_i || !b->w_if || !b->v_u || !b->v_i || !b->v_uf || !b->v_if)
        return RFM_ERR_BAD_ARG;
    if (cfg->rng == RFM_RNG_MT19937 && !b->perms && cfg->n_interactions > 0) return RFM_ERR_BAD_ARG;
    if ((rc = device_ok()) != RFM_OK) return rc;
    const int E = cfg->epochs;
    const int64_t N = cfg->n_interactions;
    const Workspace ws = carve(b->workspace, E, cfg->max_samples, cfg->n_items, cfg->n_users, N, feat_ring_floats(cfg), cfg->n_factors, min_segment_rows(cfg), ticket_windows(cfg), vi_split_eligible(cfg));
    if (!b->workspace || b->workspace_bytes < ws.bytes) return RFM_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)hip_stream;
    const ShapeEntry *shape = pick_shape(cfg->n_factors);
    if (T.debug_shape > 0 && T.debug_shape <= (int)(sizeof(kShapes) / sizeof(kShapes[0]))) shape = &kShapes[T.debug_shape - 1];
    if (shape->max_f < cfg->n_factors) return RFM_ERR_BAD_ARG;          // (a debug_shape too narrow for the factor rows)
    const bool serial = cfg->mode == RFM_MODE_SERIAL;
    const bool feat = cfg->has_user_features || cfg->has_item_features;
    // production Hogwild walks user segments of the CSR lists; that needs the lists to BE the interactions.  When they are
    // not (fit_partial keeps earlier items in the lists, rankfm/rankfm.py:170-172) or the caller dictates the order, the
    // rows kernel runs instead.  kRowsPlan marks such a plan in plan_token.
    constexpr int64_t kRowsPlan = (int64_t)1 << 62;
    bool use_segments = !serial && !b->perms && N > 0 && cfg->plan_token != kRowsPlan &&
                        (!feat || feat_table_floats(cfg) <= kMaxLdsTableFloats);

    // ---- host-side constants: WARP multipliers in double like the reference (integer division inside the log,
    //      rankfm/_rankfm.pyx:269 under cdivision=True), MT19937 seeding (mt19937ar.c:60-73)
    std::vector<float> mult((size_t)cfg->max_samples + 1, 0.0f);
    for (int s = 1; s <= cfg->max_samples; ++s)
        mult[s] = (float)(log((double)((cfg->n_items - 1) / s)) / log((double)cfg->n_items));
    // (the engine layout of the item-side weights in front of it is imported below, or kept from the previous call)
    RFM_HIP(hipMemsetAsync((char *)b->workspace + ws.volatile_offset, 0, ws.bytes - ws.volatile_offset, stream));
    RFM_HIP(hipMemcpyAsync(ws.multiplier, mult.data(), mult.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    std::vector<uint32_t> mt(625);
    if (cfg->rng == RFM_RNG_MT19937) {
        mt[0] = cfg->seed;
        for (int k = 1; k < 624; ++k) mt[k] = 1812433253u * (mt[k - 1] ^ (mt[k - 1] >> 30)) + (uint32_t)k;
        mt[624] = 624;
        RFM_HIP(hipMemcpyAsync(ws.mt_state, mt.data(), 625 * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    }
    // (skipped when the caller vouches for a cached plan: the lists were checked when it was built)
    if (cfg->plan_token <= 0)      // (a cached plan of either kind: the lists were checked when it was built)
        degree_check_kernel<<<dim3((cfg->n_users + 255) / 256), dim3(256), 0, stream>>>(b->csr_offsets, b->csr_items, cfg->n_users,
                                                                                          cfg->n_items, ws.error_flags);

    // ---- plan, part 1 (segments kernel): user segments.  A user of degree d is cut into ceil(d / 32) near-equal runs of
    //      consecutive CSR positions; descriptors {user, first position, length} are built on the host from the offsets.
    //      The plan lives in the persistent head of the workspace; `plan_token` (= segment count) says it is still valid.
    const float damp_m = cfg->hogwild_damping == 0.0f ? 128.0f : cfg->hogwild_damping;
    const bool one_group_flag = (T.debug_flags & 1) != 0;
    int seg_rows = kSegmentRows;
    if (T.segment_rows > 0) seg_rows = std::min(kSegmentRows, T.segment_rows);
    // plan_token = segment count | hot-slot count << 40 | segment length << 48; a plan cut for another segment length is rebuilt
    const bool token_plan = cfg->plan_token > 0 && cfg->plan_token != kRowsPlan;
    const bool have_plan = token_plan && (int)((cfg->plan_token >> 48) & 0xFF) == seg_rows;
    int64_t n_segments = have_plan ? (cfg->plan_token & (((int64_t)1 << 40) - 1)) : 0;
    int n_hot = have_plan ? (int)((cfg->plan_token >> 40) & 0xFF) : 0;
    const bool build_plan = !serial && (cfg->plan_token <= 0 || (token_plan && !have_plan));
    std::vector<int64_t> off;
    std::vector<int> item_count;
    if (use_segments && build_plan) {
        off.resize((size_t)cfg->n_users + 1);
        if (host_offsets) memcpy(off.data(), host_offsets, sizeof(int64_t) * off.size());
        else {
            RFM_HIP(hipMemcpyAsync(off.data(), b->csr_offsets, sizeof(int64_t) * off.size(), hipMemcpyDeviceToHost, stream));
            RFM_HIP(hipStreamSynchronize(stream));
        }
        if (off[cfg->n_users] != N) use_segments = false;            // lists hold more than this call's interactions
    }
    if (use_segments && build_plan) {
        std::vector<int4> desc;
        desc.reserve(max_segments(N, cfg->n_users, seg_rows));
        for (int u = 0; u < cfg->n_users; ++u) {
            const int64_t d = off[u + 1] - off[u];
            if (d <= 0) continue;
            const int64_t parts = (d + seg_rows - 1) / seg_rows;
            for (int64_t p = 0; p < parts; ++p) {
                const int64_t s0 = off[u] + d * p / parts, s1 = off[u] + d * (p + 1) / parts;
                desc.push_back(make_int4(u, (int)s0, (int)(s1 - s0), 0));
            }
        }
        n_segments = (int64_t)desc.size();
        if (desc.size() > max_segments(N, cfg->n_users, min_segment_rows(cfg))) return RFM_ERR_WORKSPACE;       // (cannot happen: see max_segments)
        RFM_HIP(hipMemcpyAsync(ws.seg_desc, desc.data(), sizeof(int4) * desc.size(), hipMemcpyHostToDevice, stream));
        RFM_HIP(hipMemsetAsync(ws.sw_csr, 0xFF, sizeof(float) * (size_t)N, stream));
        RFM_HIP(hipMemsetAsync(ws.sw_max_bits, 0, sizeof(unsigned int), stream));
        sw_to_csr_kernel<<<dim3(1024), dim3(256), 0, stream>>>(b->interactions, b->sample_weight, (long long)N, b->csr_offsets,
                                                                 b->csr_items, (unsigned int *)ws.sw_csr, ws.error_flags, ws.sw_max_bits);
        // (the item histogram of plan part 2 rides on the same synchronisation: one read-back round trip less per planned call)
        if (!serial && damp_m > 0.0f && !one_group_flag) {
            item_count.resize((size_t)cfg->n_items);
            RFM_HIP(hipMemsetAsync(ws.pos_scale, 0, sizeof(float) * (size_t)cfg->n_items, stream));
            item_count_kernel<<<dim3(1024), dim3(256), 0, stream>>>(b->interactions, (long long)N, (int *)ws.pos_scale);
            RFM_HIP(hipMemcpyAsync(item_count.data(), ws.pos_scale, sizeof(int) * item_count.size(), hipMemcpyDeviceToHost, stream));
        }
        unsigned int flags = 0;
        RFM_HIP(hipMemcpyAsync(&flags, ws.error_flags, sizeof flags, hipMemcpyDeviceToHost, stream));
        RFM_HIP(hipStreamSynchronize(stream));                        // also: `desc` is pageable host memory
        if (flags & 4u) {                                             // same length, different content: not this call's rows
            use_segments = false;
            RFM_HIP(hipMemsetAsync(ws.error_flags, 0, sizeof(unsigned int) * 4, stream));
            degree_check_kernel<<<dim3((cfg->n_users + 255) / 256), dim3(256), 0, stream>>>(b->csr_offsets, b->csr_items, cfg->n_users,
                                                                                              cfg->n_items, ws.error_flags);
        }
    }
    // A saturated user (every item in the list) would make the rejection sampler of EVERY one of its rows run to its attempt limit
    // before the error comes back -- minutes on a 70-row user -- so the verdict of the degree check is read before anything is
    // launched (one 4-byte read-back per call that checked; the reference spins forever at rankfm/_rankfm.pyx:250-253).
    // (only when a list CAN hold every item: some degree >= I -- known exactly when the offsets are on the host for the plan, else
    // bounded by N >= I; every other call skips the synchronisation)
    // (the lists may hold MORE than this call's interactions -- fit_partial keeps earlier items -- so N bounds a degree only when the
    //  offsets are not at hand; with the offsets on the host every degree is looked at, whatever N)
    const int64_t *deg_off = !off.empty() ? off.data() : host_offsets;
    bool may_saturate = N >= (int64_t)cfg->n_items;
    if (deg_off) {
        may_saturate = false;
        for (int u = 0; u < cfg->n_users && !may_saturate; ++u) may_saturate = deg_off[(size_t)u + 1] - deg_off[u] >= (int64_t)cfg->n_items;
    } else if (!may_saturate && cfg->plan_token <= 0 && !use_segments && N > 0) {
        // device-resident lists of unknown length (rows kernel): the lists' total length bounds every degree -- one 8-byte read-back
        int64_t nnz_dev = 0;
        RFM_HIP(hipMemcpyAsync(&nnz_dev, b->csr_offsets + cfg->n_users, sizeof nnz_dev, hipMemcpyDeviceToHost, stream));
        RFM_HIP(hipStreamSynchronize(stream));
        may_saturate = nnz_dev >= (int64_t)cfg->n_items;
    }
    if (cfg->plan_token <= 0 && may_saturate) {
        unsigned int flags = 0;
        RFM_HIP(hipMemcpyAsync(&flags, ws.error_flags, sizeof flags, hipMemcpyDeviceToHost, stream));
        RFM_HIP(hipStreamSynchronize(stream));
        if (flags & 2u) {
            if (rep) {            // (nothing ran: every scalar field of the report says so, the caller's arrays stay untouched)
                rfm_fit_report z;
                memset(&z, 0, sizeof z);
                z.log_likelihood = rep->log_likelihood; z.reg_penalty = rep->reg_penalty; z.sgd_kernel_ms = rep->sgd_kernel_ms; z.n_draws = rep->n_draws;
                z.nonfinite_array = -1;
                *rep = z;
            }
            return RFM_ERR_USER_SATURATED;
        }
    }
    const bool single_group = use_segments && one_group_flag, fresh = (T.debug_flags & 2) != 0;
    const bool damp = !serial && !single_group && damp_m > 0.0f && N > 0;

    // ---- plan, part 2: item popularity (positive occurrences per item), needed by the damping and by the hot-row choice
    if (damp && build_plan && item_count.empty()) {
        item_count.resize((size_t)cfg->n_items);
        RFM_HIP(hipMemsetAsync(ws.pos_scale, 0, sizeof(float) * (size_t)cfg->n_items, stream));
        item_count_kernel<<<dim3(1024), dim3(256), 0, stream>>>(b->interactions, (long long)N, (int *)ws.pos_scale);
        RFM_HIP(hipMemcpyAsync(item_count.data(), ws.pos_scale, sizeof(int) * item_count.size(), hipMemcpyDeviceToHost, stream));
        RFM_HIP(hipStreamSynchronize(stream));
    }
    // Hot rows: items that >= kHotMin in-flight updates would touch at once (estimated with the default geometry).  Their
    // atomics serialise on one or two cache lines -- on BASELINE config 2 the ten hottest items cost half the epoch -- so
    // the HOT kernel accumulates them per workgroup in LDS.  (bit 2 of debug_flags switches this off.)
    // (models with features: only the pipelined row loop of sgd_features_kernel carries the accumulators -- BPR, 16-lane row groups,
    //  at most 32 + 32 features)
    const bool feat_fast = feat && shape->group == 16 && cfg->max_samples == 1 && cfg->n_user_features <= 32 && cfg->n_item_features <= 32;
    std::vector<int> hot_order;
    if (damp && build_plan && use_segments && (!feat || feat_fast) && !(T.debug_flags & 4)) {
        // (interactions in flight: the full-chip geometry, or what the concurrency caps of the geometry below leave of it -- on a
        // small problem a popular item is touched by a handful of concurrent updates at most, and accumulating it would only
        // delay its updates: measured on the 3000 x 2000 feature-model fixture, |w_i| -4.6 % with, -0.8 % without)
        const double g0 = std::min((double)(g_sm_count > 0 ? g_sm_count : 256) * 16.0 * (64 / shape->group),
                                   (double)std::max<long long>(1, std::min<long long>(N / 128, (long long)std::min(cfg->n_users, cfg->n_items) / 3)));
        const double kHotMin = 16.0;
        for (int i = 0; i < cfg->n_items; ++i)
            if ((double)item_count[i] * g0 / (double)N >= kHotMin) hot_order.push_back(i);
        std::sort(hot_order.begin(), hot_order.end(), [&](int x, int y) { return item_count[x] > item_count[y] || (item_count[x] == item_count[y] && x < y); });
        // LDS budget of the accumulators: 48 KiB per workgroup
        const int max_hot = std::min(T.hot_slots > 0 ? std::min(T.hot_slots, kMaxHot) : kHotSlots, 12288 / (cfg->n_factors + 2));
        if ((int)hot_order.size() > max_hot) hot_order.resize(max_hot > 0 ? max_hot : 0);
        n_hot = (int)hot_order.size();
    }
    const bool use_hot = use_segments && (!feat || feat_fast) && !single_group && n_hot > 0;
    const sgd_launch_fn launch = (use_hot && !feat) ? shape->table()[8 + (fresh ? 1 : 0)]
                                 : use_segments ? shape->table()[4 + (feat ? 1 : 0) + (fresh ? 2 : 0)]
                                                : shape->table()[(serial ? 2 : 0) + (feat ? 1 : 0)];

    // ---- launch geometry.  unit of work = one interaction (rows kernel) or one user segment (segments kernel)
    const int64_t units = use_segments ? n_segments : N;
    const int groups_per_wave = serial ? 1 : 64 / shape->group;
    // features kernel: 16 wavefronts per workgroup share one LDS copy of the tables (<= 64 KB, checked above)
    int feat_waves = 16;
    if (T.feature_waves > 0) feat_waves = std::max(2, std::min(16, T.feature_waves));
    // (the table trainer also stages one step per row group: 1 + 2F + P + Q floats each; wide tables take smaller workgroups)
    while (feat_waves > 2 && sizeof(float) * (feat_table_floats(cfg) + 8 + (size_t)feat_waves * (64 / shape->group) *
                                              (5 + 2 * (size_t)shape->group * shape->kpl + cfg->n_user_features + cfg->n_item_features)) > kLdsBytes)
        feat_waves /= 2;
    // (the tables kernel keeps `feat_waves`; the pipelined row loop runs 12 wavefronts per workgroup -- three per SIMD, 168 registers:
    //  at 16 it has 128 and spills a third of its working set, rfm_sgd.hpp -- unless the caller overrides)
    // (a batch of the table trainer = one staged step per row group of the tables kernel's workgroup: `table_batch` sizes that workgroup)
    int table_waves = feat_waves;
    if (T.table_batch > 0) table_waves = std::max(1, std::min(feat_waves, T.table_batch * shape->group / 64));
    if (use_segments && feat && feat_fast && T.feature_waves == 0 && !single_group) feat_waves = std::min(feat_waves, 12);
    const int waves_per_block = serial ? 1 : (use_segments && feat ? feat_waves : (use_hot ? 16 : 4));   // see sgd_segments_kernel
    int grid = 1, n_producers = 0;
    const bool feat_frozen = (T.debug_flags & 32) != 0;
    int64_t max_groups = 0;
    int64_t units_per_launch = units > 0 ? units : 1;
    if (!serial) {
        if (T.rows_per_launch > 0 && T.rows_per_launch < N) {
            units_per_launch = use_segments ? (int64_t)((double)T.rows_per_launch * (double)units / (double)N) : T.rows_per_launch;
            if (units_per_launch < 1) units_per_launch = 1;
        }
        const int64_t groups_per_block = (int64_t)groups_per_wave * waves_per_block;
        int64_t need = (units_per_launch + groups_per_block - 1) / groups_per_block;
        // Default concurrency: 4 workgroups of 4 wavefronts per CU (1024 workgroups, 16 k interactions in flight on
        // MI355X).  Measured on BASELINE config 2 the update rate saturates there (profiles/); more wavefronts only add
        // staleness.  All workgroups are resident, so they sweep the epoch's order together and the realised order stays
        // close to the sequential one.  Never keep more than 1/128 of an epoch in flight: every in-flight update reads
        // weights that are stale by up to that many steps, and Hogwild only tracks sequential SGD while that window is
        // a small fraction of the data (DESIGN.md "staleness").
        int64_t cap = (int64_t)(g_sm_count > 0 ? g_sm_count : 256) * 16 / waves_per_block;
        // BPR with hot-row accumulators (config 2's kernel): three quarters of the CUs.  The kernel sits at the memory-side atomic
        // path's capacity, not at the CUs': measured in round 5 (three interleaved sessions each, profiles/r05_notes.md) 192 workgroups
        // run 2.61 / 2.63 / 2.60 ms against 2.63 / 2.66 / 2.66 at 256 (224: 2.64 / 2.66 / 2.61; 160 and fewer: slower), with a quarter
        // fewer rows in flight -- the asynchrony term of the ranking quality scales with those (-1.9 point at 16 k in
        // flight against the oracle in the engine's order, -1.2 at 12 k) -- and a quarter fewer hot-row publications.
        if (use_hot && !feat && cfg->max_samples == 1 && T.n_workgroups <= 0) cap = cap * 3 / 4;
        // (feature launches: ONE workgroup per CU whatever its size -- the 12-wavefront row loop takes three wavefronts per SIMD and no
        //  second workgroup fits beside it; the trainer, its producers and the row loops must all be resident)
        if (use_segments && feat) cap = std::min<int64_t>(cap, g_sm_count > 0 ? g_sm_count : 256);
        const int64_t window = (N / 128 + groups_per_block - 1) / groups_per_block;
        if (window < cap) cap = window;
        // ... and keep conflicts sparse: with g interactions in flight an update meets ~2g/I concurrent updates of its two
        // item rows and ~g/U of its user row.  Ranking quality tracks the sequential reference while g <= min(U, I) / 3
        // and collapses beyond ~1 (measured on the MovieLens-1M-shaped surrogate, profiles/r01_notes.md).
        const int64_t sparse = ((int64_t)(cfg->n_users < cfg->n_items ? cfg->n_users : cfg->n_items) / 3 + groups_per_block - 1) / groups_per_block;
        if (sparse < cap) cap = sparse;
        if (cap < 1) cap = 1;
        // both limits are in interactions (row groups); below one workgroup's worth the kernel idles the surplus groups
        max_groups = N / 128 < (int64_t)(cfg->n_users < cfg->n_items ? cfg->n_users : cfg->n_items) / 3
                         ? N / 128 : (int64_t)(cfg->n_users < cfg->n_items ? cfg->n_users : cfg->n_items) / 3;
        if (max_groups < 1) max_groups = 1;
        if (T.n_workgroups > 0) max_groups = 0;          // explicit geometry: no cap
        if (T.n_workgroups > 0) cap = T.n_workgroups;
        grid = (int)(need < cap ? need : cap);
        if (grid < 1 || single_group) grid = 1;
        // features kernel: workgroup 0 is the table trainer and workgroups 1 .. n_producers stage the steps it applies
        // (sgd_features_kernel) -- resident workgroups like the others, not extra ones (a workgroup that had to wait for a free
        // CU would run its share after everybody else).  Measured on config 4's share (rfm_fit_report.feat_diag, profiles/r03_notes.md):
        // a producer stages a batch of 64 steps in ~33 us, the trainer applies one in ~12 us (its apply walk is bound by the LDS
        // pipe of its CU): three producers keep it busy.  More only make the steps STALER -- a step is scored on the tables of its
        // time, and on the 3000 x 2000 feature fixture two producers (one slot each) rank 0.9 point of hit_rate@10 better than four.
        if (use_segments && feat && !single_group && !feat_frozen) {
            const int64_t room = std::max<int64_t>(cap, 3);
            n_producers = grid >= 64 ? 3 : (grid >= 4 ? 2 : 1);
            if (T.table_producers > 0) n_producers = std::min(kFeatMaxProducers, T.table_producers);
            // the trainer, its producers and at least one row loop must all be RESIDENT (they hand-shake by spinning): never more
            // producers than the launch's room leaves beside one trainer and one row-loop workgroup
            n_producers = (int)std::max<int64_t>(1, std::min<int64_t>(n_producers, room - 2));
            if (grid + 1 + n_producers > room) grid = (int)std::max<int64_t>(1, room - 1 - n_producers);
            // The tables kernel and the row-loop kernel are two launches, and the hardware deals the workgroups of EACH launch round
            // the XCDs (workgroup b of a launch -> XCD b % 8, observed; MI355X_MICROARCH.md): with 4 + 252 workgroups on 256 CUs the
            // XCDs 0 - 3 are handed 33 one-per-CU workgroups for their 32 CUs, the trainer's own among them -- measured, the two
            // kernels then ran one AFTER the other (overlap 0.1 of 4.5 ms, the tables trained before the rows).  A chip-filling row
            // loop therefore leaves every XCD as many CUs free as the tables kernel puts there (248 = 8 x 31 beside 1 + 3).
            // Placement is not contractual: if it changes, the kernels overlap less -- slower, still correct (fixed quota).
            const int sms = g_sm_count > 0 ? g_sm_count : 256;
            constexpr int kXcds = 8;
            if (sms % kXcds == 0 && grid + 1 + n_producers > sms - kXcds) {
                const int per_xcd = sms / kXcds - (1 + n_producers + kXcds - 1) / kXcds;
                grid = std::max(1, std::min(grid, per_xcd * kXcds));
            }
            grid += 1 + n_producers;
        }
    }
    const int launches = (int)((units + units_per_launch - 1) / units_per_launch);
    // ---- plan, part 3: Hogwild damping.  n(row) = interactions in flight x the row's share of the data (+ what the other
    //      workgroups hold unpublished for a hot row); scale = min(1, M / n)
    long long in_flight = single_group ? 1 : (long long)(grid - (n_producers > 0 ? 1 + n_producers : 0)) * waves_per_block * groups_per_wave;
    if (!single_group && max_groups > 0 && max_groups < in_flight) in_flight = max_groups;
    const float damp_cap = damp ? damp_m * (float)N / (float)in_flight : 0.0f;
    // a user's in-flight SEGMENT publishes its accumulated steps only when it ends: count a concurrent segment as its length
    const float avg_seg = use_segments && n_segments > 0 ? (float)N / (float)n_segments : 1.0f;
    if (damp && build_plan) {
        std::vector<float> scale((size_t)cfg->n_items);
        for (int i = 0; i < cfg->n_items; ++i) {
            const double n = (double)in_flight * (double)item_count[i] / (double)N;
            scale[i] = n > damp_m ? (float)(damp_m / n) : 1.0f;
        }
        std::vector<int32_t> h_item(kMaxHot, 0), h_period(kMaxHot, 1);
        // (BPR, with or without features: 24, measured -- config 4's share 3.74 -> 3.63 ms; WARP keeps 48: its kernel does not notice
        //  them, 8.74 / 8.77 / 8.76 ms at 32 / 24 / 16 on config 3: profiles/r06_notes.md)
        const double hot_pubs = T.hot_publications > 0 ? (double)T.hot_publications : (cfg->max_samples == 1 ? kHotPublications : 48.0);
        for (int s = 0; s < (use_hot ? n_hot : 0); ++s) {
            const int i = hot_order[s];
            // publish about kHotPublications times per epoch and workgroup: ~3 % of the row's updates are pending chip-wide at any time
            int period = (int)((double)item_count[i] / ((double)grid * hot_pubs) + 0.5);
            if (period < 1) period = 1;
            if (period > 64) period = 64;
            const double n = (double)in_flight * (double)item_count[i] / (double)N + 0.5 * (double)grid * (double)period;
            const float sc = n > damp_m ? (float)(damp_m / n) : 1.0f;
            scale[i] = sc + 2.0f * (float)(s + 1);              // slot encoded above the scale (see SgdArgs::hot_item)
            h_item[s] = i;
            h_period[s] = period;
        }
        RFM_HIP(hipMemcpyAsync(ws.pos_scale, scale.data(), sizeof(float) * scale.size(), hipMemcpyHostToDevice, stream));
        RFM_HIP(hipMemcpyAsync(ws.hot_item, h_item.data(), sizeof(int32_t) * kMaxHot, hipMemcpyHostToDevice, stream));
        RFM_HIP(hipMemcpyAsync(ws.hot_period, h_period.data(), sizeof(int32_t) * kMaxHot, hipMemcpyHostToDevice, stream));
        RFM_HIP(hipStreamSynchronize(stream));              // pageable host vectors
    }

    // Hogwild launches work on a padded copy of the item biases (one 64-byte line each; the epoch tail reads it in place)
    // (WARP reads a bias per candidate, ~20 per update: there the 16x larger table costs more in read misses than the
    // atomics gain -- config 3: 318 M updates/s unpadded, 306 M padded -- so only BPR-like sampling pads)
    const bool pad_bias = !serial && cfg->max_samples <= 4;
    // Chip-filling BPR launches with hot-row accumulators (the launches that sit at the memory-side atomic path's capacity) work on a
    // segment-major copy of the item factor rows (SgdArgs::vi_split).  The epoch tail reads the copy: its sums and its finiteness
    // check do not depend on the order of the elements.
    const bool vi_split = ws.vi_split && use_hot && !feat && !single_group && vi_split_eligible(cfg);
    const int vi_segs = cfg->n_factors / 16;
    const int vi_grid = (int)std::min<size_t>(4096, ((size_t)cfg->n_items * vi_segs * 4 + 255) / 256);
    // ---- the engine layout of the item-side weights (padded biases, segment-major rows, hot-row bins): imported from the caller's
    //      arrays now, unless the previous call on this workspace kept it (layout_token) -- then the workspace holds the current
    //      weights and the caller's v_i / w_i are stale.  layout token = 1 | padded biases << 1 | segment-major rows << 2 | hot slots << 8.
    const int64_t layout_kind = 1 | (pad_bias ? 2 : 0) | (vi_split ? 4 : 0) | ((int64_t)(use_hot ? n_hot : 0) << 8);
    const bool layout_live = cfg->layout_token != 0;
    if (layout_live && (build_plan || cfg->layout_token != layout_kind)) {
        g_last_error = "layout_token: the workspace does not hold this call's engine layout (another plan, geometry or model kind)";
        return RFM_ERR_BAD_ARG;
    }
    if (!layout_live) {
        if (use_hot) RFM_HIP(hipMemsetAsync(ws.hot_bins_v, 0, (size_t)((const char *)(ws.hot_bins_w + (size_t)kHotBins * kMaxHot) - (const char *)ws.hot_bins_v), stream));
        if (pad_bias) bias_pad_kernel<true><<<dim3((cfg->n_items + 255) / 256), dim3(256), 0, stream>>>(b->w_i, ws.w_pad, damp ? ws.pos_scale : nullptr, cfg->n_items);
        if (vi_split) vi_split_kernel<true><<<dim3(vi_grid), dim3(256), 0, stream>>>((float4 *)b->v_i, (float4 *)ws.vi_split, cfg->n_items, vi_segs);
    }
    // the way back (the end of a call that does not keep the layout; rfm_fit_export_weights): pending hot-row sums into the rows, then
    // the two copies into the caller's arrays
    SgdArgs hot_args;                 // (the last epoch's arguments: what hot_sweep_line reads)
    memset(&hot_args, 0, sizeof hot_args);
    auto export_layout = [&]() {
        if (use_hot) hipLaunchKernelGGL(hot_reduce_kernel, dim3((n_hot * ((cfg->n_factors + 15) / 16) + (n_hot + 15) / 16 + 3) / 4), dim3(256), 0, stream, hot_args);
        if (pad_bias) bias_pad_kernel<false><<<dim3((cfg->n_items + 255) / 256), dim3(256), 0, stream>>>(b->w_i, ws.w_pad, nullptr, cfg->n_items);
        if (vi_split) vi_split_kernel<false><<<dim3(vi_grid), dim3(256), 0, stream>>>((float4 *)b->v_i, (float4 *)ws.vi_split, cfg->n_items, vi_segs);
    };
    // timing events: destroyed on every exit path
    struct Events {
        std::vector<hipEvent_t> ev;
        ~Events() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
    } events;
    std::vector<hipEvent_t> &ev = events.ev;
    ev.assign((size_t)2 * E + 1, nullptr);
    const bool timing = rep && rep->sgd_kernel_ms;
    if (timing)
        for (size_t k = 0; k < (size_t)2 * E; ++k) RFM_HIP(hipEventCreate(&ev[k]));
    // The reference stops at the first epoch that ends non-finite (assert_finite, rankfm/_rankfm.pyx:329).  Epochs are
    // enqueued without waiting; every kCheckEvery epochs the flags the tail kernels have written so far are copied to pinned host
    // memory behind the work already queued, and the copy issued kCheckEvery epochs EARLIER -- long complete unless the host runs that
    // far ahead of the device -- is looked at: launching stops once a flag is set, instead of training on NaN for the rest of the
    // call, and the device never waits for the host (a blocking read-back here idled it for a launch latency every eighth epoch).
    constexpr int kCheckEvery = 8;
    int epochs_launched = 0;
    // (the pinned buffer is kept per host thread and grown on demand: allocating and freeing pinned memory costs a call more than all of
    //  its launches' gaps together)
    struct Probe { unsigned int *host = nullptr; size_t words = 0; };
    static thread_local Probe probe_cache;
    Probe probe;
    hipEvent_t &probe_ev = ev[(size_t)2 * E];
    int probed = 0;                   // epochs covered by the copy in flight (0: none)
    if (cfg->check_finite && E > kCheckEvery) {
        if (probe_cache.words < (size_t)E + 16) {
            if (probe_cache.host) (void)hipHostFree(probe_cache.host);
            probe_cache.host = nullptr; probe_cache.words = 0;
            const size_t words = std::max<size_t>(1024, (size_t)E + 16);
            RFM_HIP(hipHostMalloc((void **)&probe_cache.host, sizeof(unsigned int) * words, hipHostMallocPortable));
            probe_cache.words = words;
        }
        probe = probe_cache;
        RFM_HIP(hipEventCreateWithFlags(&probe_ev, hipEventDisableTiming));
    }

    for (int e = 0; e < E; ++e) {
        if (probe.host && e > 0 && e % kCheckEvery == 0) {
            if (probed > 0) {
                RFM_HIP(hipEventSynchronize(probe_ev));
                bool stop = (probe.host[E] & 3u) != 0;
                for (int k = 0; k < probed; ++k) stop |= probe.host[k] != 0;
                if (stop) break;
            }
            RFM_HIP(hipMemcpyAsync(probe.host, ws.nonfinite, sizeof(unsigned int) * e, hipMemcpyDeviceToHost, stream));
            RFM_HIP(hipMemcpyAsync(probe.host + E, ws.error_flags, sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
            RFM_HIP(hipEventRecord(probe_ev, stream));
            probed = e;
        }
        epochs_launched = e + 1;
        const int epoch = cfg->epoch_begin + e;
        SgdArgs a;
        a.interactions = b->interactions; a.sample_weight = b->sample_weight;
        a.csr_off = b->csr_offsets; a.csr_items = b->csr_items; a.x_uf = b->x_uf; a.x_if = b->x_if;
        a.w_i = pad_bias ? ws.w_pad : b->w_i; a.w_stride = pad_bias ? kBiasStride : 1; a.scale_in_pad = pad_bias ? 1 : 0; a.w_if = b->w_if; a.v_u = b->v_u; a.v_i = vi_split ? ws.vi_split : b->v_i; a.v_uf = b->v_uf; a.v_if = b->v_if;
        a.perm = b->perms ? b->perms + (size_t)e * N : nullptr;
        a.multiplier = ws.multiplier; a.mt_state = ws.mt_state;
        a.ll = ws.ll + e; a.draws = ws.draws + e; a.error_flags = ws.error_flags;
        a.n_rows = N; a.n_items = cfg->n_items; a.n_uf = cfg->n_user_features; a.n_if = cfg->n_item_features;
        a.n_factors = cfg->n_factors; a.has_uf = cfg->has_user_features; a.has_if = cfg->has_item_features;
        a.max_samples = cfg->max_samples; a.rng = cfg->rng;
        a.epoch_key = rfm_epoch_key(cfg->seed, (uint32_t)(epoch + cfg->rng_epoch_offset));
        a.perm_bits = rfm_perm_bits((uint32_t)N);
        a.sw_csr = ws.sw_csr; a.seg_desc = ws.seg_desc; a.n_segments = n_segments;
        a.seg_bits = rfm_perm_bits((uint32_t)(n_segments > 0 ? n_segments : 1));
        a.single_group = single_group ? 1 : 0;
        a.max_groups = max_groups;
        a.hot_item = ws.hot_item; a.hot_period = ws.hot_period; a.n_hot = use_hot ? n_hot : 0;
        a.hot_bins_v = ws.hot_bins_v; a.hot_bins_w = ws.hot_bins_w; a.sw_max_bits = ws.sw_max_bits;
        // (sweeping workgroups: all of them, or the row-loop workgroups of the features kernel)
        a.hot_sweep_every = T.hot_sweep_every > 0 ? T.hot_sweep_every : kHotSweepEvery;
        a.hot_direct = n_hot * ((cfg->n_factors + 15) / 16) + (n_hot + 15) / 16 > 4 * (grid - (n_producers > 0 ? 1 + n_producers : 0)) ? 1 : 0;   // see SgdArgs::hot_bins_v
        a.launch_index = 0;
        a.block_threads = waves_per_block * 64;
        a.table_threads = table_waves * 64;
        a.feat_ring = ws.feat_ring; a.feat_flags = ws.feat_flags; a.n_producers = n_producers; a.feat_frozen = feat_frozen ? 1 : 0;
        a.tickets = nullptr;
        a.damp_positive_only = (T.debug_flags & 256) ? 1 : 0;
        a.vi_split = vi_split ? 1 : 0;
        a.feat_clock = ws.feat_clock;
        a.sclk = ws.sclk;
        a.table_quota = 0;
        a.table_step = T.table_step_pct > 0 ? (float)T.table_step_pct * 0.01f : 1.0f;
        a.table_quiet_from = 0.0f;
        a.table_pace = 0.0f;
        // table trainer: steps to apply in a launch of `n_units` segments beside `rowloop_wgs` row-loop workgroups.  A row-loop workgroup
        // walks about as many rows per second as the trainer applies steps (profiles/r03_notes.md section 7), so a trainer that works
        // flat out for the length of the launch gets through rows / workgroups of them; measured on config 4's share with the split
        // kernels: 5.8 steps/us against 1450 rows/us of 252 x 48 row groups, i.e. every 250th row in equal time.  The quota is that
        // pace with a wide margin: every (kTableQuotaFactor x row groups / 64)-th row -- 446 on the full chip; the opening launch keeps
        // 1.8 x: every 22nd -- or the caller's `tune_table_every`.  The rows of a launch's last part, which train against tables that have
        // STOPPED moving, are what brings the engine's tables (64 staged steps scored on one table state) to the reference's ranking
        // quality (profiles/r04_notes.md section 11); the quota's margin secures that quiet period, and since round 5 the trainer also
        // stops by itself once 80 % of the launch's segments are handed out when the CALLER asks for a denser quota than this
        // (`tune_table_every`; kTableQuietFrom), which can therefore no longer run to the launch's end and beyond.  Launches that do
        // not fill a good part of the chip (fewer than 4096 row groups) keep 1.8: their row loops are latency-bound and slow per row, the
        // trainer is nowhere near their length, and the 3000 x 2000 feature fixture sits within 0.3 point of the REFERENCE there (2.4
        // ranks it a full point ABOVE the reference -- outside the bar from the other side).
        auto every_default_of = [&](int rowloop_wgs, double factor) -> double {
            // (in units of 64 row groups -- sixteen wavefronts -- which is what the measurement was made with)
            double rowloop_groups = (double)rowloop_wgs * (double)(waves_per_block * groups_per_wave);
            if (max_groups > 0) rowloop_groups = std::min(rowloop_groups, (double)max_groups);
            if (rowloop_groups < 4096.0) factor = std::min(factor, 1.8);
            return std::max(1.0, factor * rowloop_groups / 64.0);
        };
        auto quota_of = [&](int64_t n_units_launch, int rowloop_wgs, double factor, float *quiet_from) -> int64_t {
            const double rows = (double)N * (double)n_units_launch / (double)std::max<int64_t>(1, units);
            const double every_default = every_default_of(rowloop_wgs, factor);
            const double every = T.table_every > 0 ? (double)T.table_every : every_default;
            // a caller's quota DENSER than the default gets the trainer's own stop (kTableQuietFrom); the default and anything sparser
            // are done long before it and keep their exact, repeatable step count
            if (quiet_from) *quiet_from = (T.table_every > 0 && every < every_default) ? kTableQuietFrom : 0.0f;
            return (int64_t)(rows / every);
        };
        // ticket heads of launch `w` of this epoch (dynamic segment order; debug_flags bit 7 keeps the static stride)
        const bool use_tickets = use_segments && !single_group && !(T.debug_flags & 128);
        auto tickets_of = [&](int w) -> unsigned int * {
            return use_tickets && w < ws.windows_per_epoch ? ws.tickets + ((size_t)e * ws.windows_per_epoch + w) * kTicketWords : nullptr;
        };
        // rankfm/_rankfm.pyx:220-223: pow() in double, narrowed to the float `eta`
        a.eta = cfg->learning_schedule == RFM_SCHEDULE_CONSTANT
                    ? cfg->learning_rate
                    : (float)((double)cfg->learning_rate / pow((double)(epoch + 1), (double)cfg->learning_exponent));
        a.reg_a = 2.0f * cfg->alpha;      // :171
        a.reg_b = 2.0f * cfg->beta;       // :172
        a.pos_scale = damp ? ws.pos_scale : nullptr;
        a.user_cap = damp ? damp_cap / avg_seg : INFINITY;
        // the dense feature tables are touched by EVERY in-flight row and shrink by 2*beta*eta per touch: keep the summed
        // stale shrink of one in-flight window below 1/2 as well
        a.feat_scale = damp ? fminf(1.0f, fminf(damp_m / (float)in_flight,
                                                0.5f / ((float)in_flight * a.eta * fmaxf(a.reg_b, 1e-6f)))) : 1.0f;

        if (timing) RFM_HIP(hipEventRecord(ev[2 * e], stream));
        int window = 0;
        // a caller may ask for one part of the epoch's order only (several delta exchanges per epoch on multi-GPU jobs)
        int64_t u_begin = 0, u_end = units;
        if (cfg->epoch_parts > 1) {
            u_begin = units * cfg->epoch_part_index / cfg->epoch_parts;
            u_end = units * (cfg->epoch_part_index + 1) / cfg->epoch_parts;
        }
        // The opening of a fit with features (absolute epoch 0, its first rows): the dense tables start at their initial values and
        // the first few tens of table-memories decide what the item biases pick up in the tables' place -- measured on config 4's
        // share, the whole first-epoch deviation of the asynchronous trainer (log-likelihood +7 %, |w_i| +20 % against the oracle)
        // comes from the first ~1 % of the rows: the sequential stand-in with the tables trained on every 240th row is at +2.1 % /
        // +6.8 %, with every 8th-15th row for the first 1-2 % of the rows and every 240th afterwards at +0.0 ... +0.2 % / -0.0 ...
        // +0.5 % (profiles/r03_notes.md section 7).  So those rows run as a launch of their own with a handful of row-loop
        // workgroups beside the trainer (a row-loop workgroup walks about as many rows per second as the trainer applies steps:
        // sixteen of them let it see every ~20th row): 500 table-memories of rows (a memory = 1 / (2 beta eta) table steps), at
        // most 1 / 16 of the epoch.  Measured on config 4's share: first epoch -0.5 ... +0.1 % / |w_i| -2 ... -4 % against the
        // oracle with 8 or 16 workgroups and 250 or 500 memories alike; it costs that epoch ~1 ms, every other epoch is untouched.
        int64_t head_units = 0;
        // what keys the step producers' row sample in a launch: the launch's place in the EPOCH -- a caller's part of the epoch is a launch
        // sequence of its own, and keyed by the window alone every part of an epoch trained the tables on the SAME sampled rows (eight parts
        // per epoch on one GPU: -3.0 points of hit_rate@10 at config 2's shape with tags, tables +7 ... +23 %; tools/merge_tags_scan.py)
        auto launch_key = [&](int w) -> uint32_t { return (uint32_t)w + 4099u * (uint32_t)(cfg->epoch_parts > 1 ? cfg->epoch_part_index : 0); };
        // (small launches: an eighth of their row-loop workgroups, at least one)
        const int n_rowloops = grid - 1 - n_producers;
        const int head_rowloops = std::max(1, std::min(16, n_rowloops / 8));
        if (use_segments && feat && !single_group && !feat_frozen && n_producers > 0 && epoch == 0 && cfg->rng_epoch_offset == 0 && u_begin == 0 &&
            n_rowloops >= 2 * head_rowloops && !(T.debug_flags & 64)) {
            const double memory = 1.0 / std::max(1e-6, (double)a.reg_b * (double)a.eta);
            const double frac = std::min(1.0 / 16.0, 500.0 * memory / (double)N);
            head_units = std::max<int64_t>(1, (int64_t)((double)units * frac));
            if (head_units > u_end - u_begin) head_units = u_end - u_begin;
        }
        if (head_units > 0) {
            const int saved_direct = a.hot_direct;
            a.launch_index = launch_key(window++);
            a.pos_begin = u_begin;
            a.pos_end = u_begin + head_units;
            a.hot_direct = n_hot * ((cfg->n_factors + 15) / 16) + (n_hot + 15) / 16 > 4 * head_rowloops ? 1 : 0;
            a.tickets = tickets_of(window - 1);
            a.table_quota = quota_of(a.pos_end - a.pos_begin, head_rowloops, 1.8, nullptr);
            a.table_quiet_from = 0.0f;           // (the opening launch's trainer is the slower side by design: no stop of its own)
            launch(a, 1 + n_producers + head_rowloops, stream);
            a.hot_direct = saved_direct;
        }
        for (int64_t p0 = u_begin + head_units; p0 < u_end; p0 += units_per_launch, ++window) {
            a.launch_index = launch_key(window);
            float part_pace = 0.0f;
            a.pos_begin = p0;
            a.pos_end = p0 + units_per_launch < u_end ? p0 + units_per_launch : u_end;
            a.tickets = tickets_of(window);
            if (n_producers > 0) a.table_quota = quota_of(a.pos_end - a.pos_begin, grid - 1 - n_producers, kTableQuotaFactor, &a.table_quiet_from);
            // A caller's PART of an epoch (multi-GPU: one exchange window): the tables keep the schedule of the EPOCH, not one of their own
            // per launch.  An epoch in one launch spreads its quota over the first kTablePace of its segments and leaves the rest of the
            // rows to settle on tables that have stopped moving; eight launches with that schedule each leave the model, at the end of the
            // fit, with a quiet period an eighth as long -- too short for most users' and items' rows to be walked once -- and cost 5.0
            // points of hit_rate@10 at config 2's shape with tags with every norm in place (one GPU, eight parts per epoch:
            // tools/merge_tags_scan.py, profiles/r06_notes.md section 8).  So a part's launch gets the share of the EPOCH's quota that
            // falls into its stretch of the epoch's first kTablePace, spread over that stretch; launches behind it run without a trainer.
            bool quiet_launch = false;
            if (n_producers > 0 && cfg->epoch_parts > 1 && T.table_pace_pct >= 0) {
                const double P = T.table_pace_pct > 0 ? 0.01 * (double)T.table_pace_pct : (double)kTablePace;
                const double f0 = (double)a.pos_begin / (double)units, f1 = (double)a.pos_end / (double)units;
                const double inside = std::max(0.0, std::min(f1, P) - std::min(f0, P));
                const int64_t epoch_quota = quota_of(units, grid - 1 - n_producers, kTableQuotaFactor, &a.table_quiet_from);
                a.table_quota = (int64_t)((double)epoch_quota * inside / P);
                quiet_launch = a.table_quota <= 0;
                part_pace = (float)std::min(1.0, inside / std::max(f1 - f0, 1e-12));
                a.table_quiet_from = 0.0f;       // (the epoch's schedule is the quiet period: no stop of the trainer's own per launch)
            }
            // (the quota's batches spread over the first kTablePace of the launch: feat_step_producer; the opening launch above is unpaced --
            //  its trainer is the slower side by design)
            // (only beside the pipelined row loop: the generic one strides the order statically and never touches the ticket counter)
            // (and not in a fit's FIRST epoch: there the tables are leaving their initial values and every early step counts -- unpaced, the
            //  quota is front-loaded; config 4's share, first epoch against the oracle: log-likelihood +0.55 % unpaced, +1.0 % paced)
            // (a caller's quota SPARSER than the default is paced in the first epoch as well: left to itself it is done within the launch's
            //  first third -- config 2's shape with tags, twice the spacing, 24 runs: hit_rate@10 -1.34 point against the oracle with the
            //  first epoch unpaced, -0.75 paced; the default -0.65 either way: profiles/r06_notes.md section 4)
            const bool first_epoch = epoch == 0 && cfg->rng_epoch_offset == 0;
            const bool sparser = T.table_every > 0 && (double)T.table_every > every_default_of(grid - 1 - n_producers, kTableQuotaFactor);
            if (n_producers > 0 && feat_fast && a.tickets && (!first_epoch || T.table_pace_pct > 0 || sparser))
                a.table_pace = T.table_pace_pct < 0 ? 0.0f : (T.table_pace_pct > 0 ? 0.01f * (float)T.table_pace_pct : kTablePace);
            if (part_pace > 0.0f) a.table_pace = (n_producers > 0 && feat_fast && a.tickets) ? part_pace : 0.0f;
            a.feat_frozen = (feat_frozen || quiet_launch) ? 1 : 0;
            launch(a, grid, stream);
        }
        if (timing) RFM_HIP(hipEventRecord(ev[2 * e + 1], stream));
        hot_args = a;

        if (cfg->check_finite || cfg->want_penalty) {
            // the epoch tail works on the engine's layout in place: the padded biases (strided), the segment-major rows (its sums do not
            // depend on the order of the elements), and -- in the same launch -- folds the pending hot-row sums into the rows
            TailArgs t;
            const float *ptrs[6] = {pad_bias ? ws.w_pad : b->w_i, b->w_if, b->v_u, vi_split ? ws.vi_split : b->v_i, b->v_uf, b->v_if};
            const unsigned long long lens[6] = {
                (unsigned long long)cfg->n_items, (unsigned long long)cfg->n_item_features,
                (unsigned long long)cfg->n_users * cfg->n_factors, (unsigned long long)cfg->n_items * cfg->n_factors,
                (unsigned long long)cfg->n_user_features * cfg->n_factors,
                (unsigned long long)cfg->n_item_features * cfg->n_factors};
            unsigned long long total = 0;
            for (int k = 0; k < 6; ++k) { t.ptr[k] = ptrs[k]; t.len[k] = lens[k]; total += lens[k]; }
            t.sumsq = ws.sumsq + 6 * e;
            t.nonfinite = ws.nonfinite + e;
            t.w_stride = pad_bias ? kBiasStride : 1;
            t.drain_hot = use_hot ? 1 : 0;
            int tgrid = (int)((total / 4 + 255) / 256);
            if (tgrid > 512) tgrid = 512;
            if (tgrid < 1) tgrid = 1;
            if (cfg->want_penalty) {      // (the norms want the pending sums in the rows BEFORE they are read)
                if (use_hot) hipLaunchKernelGGL(hot_reduce_kernel, dim3((n_hot * ((cfg->n_factors + 15) / 16) + (n_hot + 15) / 16 + 3) / 4), dim3(256), 0, stream, a);
                tail_kernel<true><<<dim3(tgrid), dim3(256), 0, stream>>>(t, a);
            } else tail_kernel<false><<<dim3(tgrid), dim3(256), 0, stream>>>(t, a);
        }
    }
    // A caller that keeps the engine layout (keep_layout) gets its weights back through rfm_fit_export_weights or a later call; everybody
    // else now.  (After a failed epoch -- the verdict is only known behind the synchronisation below -- the layout is exported there.)
    const bool keep = cfg->keep_layout && (pad_bias || vi_split) && epochs_launched == E;
    if (!keep) export_layout();
    RFM_HIP(hipGetLastError());

    // ---- one synchronisation: bring the per-epoch results back
    std::vector<double> h_ll(E), h_sumsq((size_t)6 * E);
    std::vector<unsigned long long> h_draws(E);
    std::vector<unsigned int> h_nonfinite(E);
    unsigned int h_err[16] = {0};
    // ll | draws | sumsq | nonfinite | error_flags | feat_clock | sclk are laid out back to back: one copy, one synchronisation
    const size_t res_bytes = (size_t)((const char *)(ws.sclk + 4) - (const char *)ws.ll);
    std::vector<char> h_res(res_bytes);
    RFM_HIP(hipMemcpyAsync(h_res.data(), ws.ll, res_bytes, hipMemcpyDeviceToHost, stream));
    RFM_HIP(hipStreamSynchronize(stream));
    auto at = [&](const void *dev) { return h_res.data() + ((const char *)dev - (const char *)ws.ll); };
    memcpy(h_ll.data(), at(ws.ll), sizeof(double) * E);
    memcpy(h_draws.data(), at(ws.draws), sizeof(unsigned long long) * E);
    memcpy(h_sumsq.data(), at(ws.sumsq), sizeof(double) * 6 * E);
    memcpy(h_nonfinite.data(), at(ws.nonfinite), sizeof(unsigned int) * E);
    memcpy(h_err, at(ws.error_flags), sizeof(h_err));

    int status = RFM_OK;
    int epochs_done = E, bad_array = -1;
    if (h_err[0] & 3u) status = RFM_ERR_USER_SATURATED;
    if (h_err[0] & 8u) { g_last_error = "features kernel: a workgroup gave up waiting for the table trainer / a step producer"; status = RFM_ERR_HIP; }
    if (h_err[0] & 16u) { g_last_error = "segment tickets: a row group gave up waiting for its workgroup's next chunk of the epoch's order"; status = RFM_ERR_HIP; }
    for (int e = 0; e < epochs_launched && status == RFM_OK; ++e) {
        if (cfg->check_finite && h_nonfinite[e]) {
            for (int k = 0; k < 6; ++k)
                if (h_nonfinite[e] & (1u << k)) { bad_array = k; break; }    // first in assert_finite order
            status = RFM_ERR_NONFINITE + bad_array;
            epochs_done = e;
        }
    }
    bool kept = keep;
    if (keep && status != RFM_OK) {       // the reference has mutated its weights in place when it raises (rankfm/_rankfm.pyx:329): so has this call
        export_layout();
        RFM_HIP(hipStreamSynchronize(stream));
        kept = false;
    }
    if (rep) {
        rep->layout_token = kept ? layout_kind : 0;
        for (int e = 0; e < E; ++e) {
            if (rep->log_likelihood) rep->log_likelihood[e] = h_ll[e];
            if (rep->n_draws) rep->n_draws[e] = (int64_t)h_draws[e];
            if (rep->reg_penalty) {
                const double *s = &h_sumsq[(size_t)6 * e];
                rep->reg_penalty[e] = (double)cfg->alpha * (s[0] + s[2] + s[3]) + (double)cfg->beta * (s[1] + s[4] + s[5]);
            }
            if (timing) {
                float ms = 0.0f;
                if (e < epochs_launched) (void)hipEventElapsedTime(&ms, ev[2 * e], ev[2 * e + 1]);
                rep->sgd_kernel_ms[e] = ms;
            }
        }
        rep->epochs_done = epochs_done;
        rep->nonfinite_array = bad_array;
        rep->launches_per_epoch = launches;
        rep->waves_per_launch = single_group ? 1 : grid * waves_per_block;
        rep->workgroups = grid;
        rep->groups_per_workgroup = single_group ? 1 : waves_per_block * groups_per_wave;
        // (row-loop groups: the trainer and the producers of the features kernel do not train rows)
        const int64_t row_groups = (int64_t)(grid - (n_producers > 0 ? 1 + n_producers : 0)) * waves_per_block * groups_per_wave;
        rep->working_groups = single_group ? 1 : ((max_groups > 0 && max_groups < row_groups) ? max_groups : row_groups);
        rep->units_per_launch = units_per_launch;
        rep->n_units = units;
        rep->segment_rows = use_segments ? seg_rows : 0;
        rep->table_producers = n_producers;
        rep->table_steps = (int64_t)h_err[2];
        for (int k = 0; k < 8; ++k) rep->feat_diag[k] = (int64_t)h_err[4 + k];
        {   // shader clock of the last launch: cycle counter against the 100 MHz wall clock, both stamped by workgroup 0
            unsigned long long c[4];
            memcpy(c, at(ws.sclk), sizeof c);
            rep->shader_mhz = (c[1] > c[0] && c[3] > c[2]) ? (float)((double)(c[3] - c[2]) / (double)(c[1] - c[0]) * 100.0) : 0.0f;
        }
        rep->table_overlap_us = -1;
        rep->table_span_us[0] = rep->table_span_us[1] = 0;
        if (n_producers > 0) {      // (wall_clock64 counts at 100 MHz)
            unsigned long long c[4];
            memcpy(c, at(ws.feat_clock), sizeof c);
            const long long lo = (long long)std::max(c[0], c[2]), hi = (long long)std::min(c[1], c[3]);
            rep->table_overlap_us = c[0] && c[2] ? std::max<long long>(0, hi - lo) / 100 : 0;
            rep->table_span_us[0] = (int64_t)(c[1] - c[0]) / 100;
            rep->table_span_us[1] = (int64_t)(c[3] - c[2]) / 100;
        }
        rep->plan_token = serial || b->perms ? 0 : (use_segments ? (n_segments | ((int64_t)(use_hot ? n_hot : 0) << 40) | ((int64_t)seg_rows << 48)) : kRowsPlan);
    }
    return status;
}

struct HostArena { char *ptr; size_t bytes; };
static HostArena &host_arena(int device) {
    static HostArena arenas[64];
    return arenas[device >= 0 && device < 64 ? device : 0];
}
// rfm_fit_host is serialised PER DEVICE for the whole call: the staging arena is one allocation per device that a larger call
// re-allocates, and models with features fork their tables kernel onto one side stream per device.  ctypes releases the GIL around
// the call, so two Python threads can arrive here together; the second simply waits.  (rfm_fit_device on caller-owned
// buffers and streams takes the same lock only when the model has features, for the side stream.)
static std::mutex &fit_mutex(int device) {
    static std::mutex mus[64];
    return mus[device >= 0 && device < 64 ? device : 0];
}
// A staging arena above this size is not kept between calls (a multi-GB buffer of one large fit would otherwise stay resident for
// the life of the process and starve later torch / predict allocations).
constexpr size_t kArenaKeepBytes = (size_t)1 << 30;

__attribute__((visibility("hidden"))) void rfm_serve_release_cache(void);      // rfm_infer.hip: the serving arenas of rfm_predict_host / rfm_recommend_host

void rfm_release_cache(void) {
    rfm_serve_release_cache();
    int n = 0, cur = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < n && d < 64; ++d) {
        std::lock_guard<std::mutex> lock(fit_mutex(d));
        HostArena &c = host_arena(d);
        if (c.ptr) { (void)hipSetDevice(d); (void)hipFree(c.ptr); c.ptr = nullptr; c.bytes = 0; }
    }
    (void)hipSetDevice(cur);
}

int rfm_fit_host(const rfm_fit_config *cfg, const rfm_fit_buffers *h, int device, rfm_fit_report *rep) {
    int rc = validate(cfg);
    if (rc != RFM_OK) return rc;
    if (!h || !h->interactions || !h->sample_weight || !h->csr_offsets || !h->csr_items || !h->x_uf || !h->x_if ||
        !h->w_i || !h->w_if || !h->v_u || !h->v_i || !h->v_uf || !h->v_if)
        return RFM_ERR_BAD_ARG;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1 || device < 0 || device >= n_dev) {
        g_last_error = "no HIP device visible";
        return RFM_ERR_NO_DEVICE;
    }
    std::lock_guard<std::mutex> fit_lock(fit_mutex(device));
    RFM_HIP(hipSetDevice(device));
    const int64_t N = cfg->n_interactions;
    const size_t U = cfg->n_users, I = cfg->n_items, P = cfg->n_user_features, Q = cfg->n_item_features, F = cfg->n_factors;
    const size_t nnz = (size_t)h->csr_offsets[U];
    struct Item { const void *src; void **dst; size_t bytes; bool out; };
    rfm_fit_buffers d;
    memset(&d, 0, sizeof d);
    const size_t ws_bytes = rfm_fit_workspace_bytes(cfg);
    Item items[] = {
        {h->interactions, (void **)&d.interactions, sizeof(int32_t) * 2 * (size_t)N, false},
        {h->sample_weight, (void **)&d.sample_weight, sizeof(float) * (size_t)N, false},
        {h->csr_offsets, (void **)&d.csr_offsets, sizeof(int64_t) * (U + 1), false},
        {h->csr_items, (void **)&d.csr_items, sizeof(int32_t) * nnz, false},
        {h->x_uf, (void **)&d.x_uf, sizeof(float) * U * P, false},
        {h->x_if, (void **)&d.x_if, sizeof(float) * I * Q, false},
        {h->w_i, (void **)&d.w_i, sizeof(float) * I, true},
        {h->w_if, (void **)&d.w_if, sizeof(float) * Q, true},
        {h->v_u, (void **)&d.v_u, sizeof(float) * U * F, true},
        {h->v_i, (void **)&d.v_i, sizeof(float) * I * F, true},
        {h->v_uf, (void **)&d.v_uf, sizeof(float) * P * F, true},
        {h->v_if, (void **)&d.v_if, sizeof(float) * Q * F, true},
        {h->perms, (void **)&d.perms, h->perms ? sizeof(int32_t) * (size_t)N * cfg->epochs : 0, false},
        {nullptr, &d.workspace, ws_bytes, false},
    };
    // ONE device allocation for everything (14 hipMalloc / hipFree pairs cost more than a millisecond of a 2.6 ms epoch), carved
    // on 256-byte boundaries -- and KEPT between calls (per device, grown when a call needs more, released by rfm_release_cache or
    // at process exit; arenas above 1 GiB are freed when the call returns): the reference's call site calls `_fit` once per fit, but
    // epoch-by-epoch callers (fit_partial loops) pay the allocation and the implicit synchronisation of hipFree every time otherwise.
    // Calls on one device are serialised (fit_mutex), like the reference's `_fit` under the GIL.
    size_t total = 0;
    for (Item &it : items) total += align_up(it.bytes ? it.bytes : 16);
    HostArena &cache = host_arena(device);
    if (cache.bytes < total) {
        if (cache.ptr) (void)hipFree(cache.ptr);
        cache.ptr = nullptr; cache.bytes = 0;
        hipError_t e = hipMalloc((void **)&cache.ptr, total);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc");
        cache.bytes = total;
    }
    char *arena = cache.ptr;
    auto cleanup = [&]() {
        if (cache.bytes > kArenaKeepBytes) { (void)hipFree(cache.ptr); cache.ptr = nullptr; cache.bytes = 0; }
    };
    // (Pinning the caller's weight arrays for the call -- hipHostRegister, so that both of their copies run as direct DMA -- was
    //  measured in round 4: 8.8 ms against 8.7 - 9.4 ms for a one-epoch config-2 call: registering 39 MB costs what it saves.)
    size_t at = 0;
    for (Item &it : items) {
        const bool keep_null = it.bytes == 0 && it.dst != (void **)&d.csr_items && it.dst != (void **)&d.interactions &&
                               it.dst != (void **)&d.sample_weight;
        void *p = arena + at;
        at += align_up(it.bytes ? it.bytes : 16);
        if (keep_null) { *it.dst = nullptr; continue; }
        *it.dst = p;
        if (it.src && it.bytes) {
            hipError_t e = hipMemcpy(p, it.src, it.bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) { cleanup(); return hip_fail(e, "hipMemcpy H2D"); }
        }
    }
    d.workspace_bytes = ws_bytes;
    rc = fit_device_impl(cfg, &d, nullptr, rep, h->csr_offsets);
    // the reference mutates the weights in place; on a non-finite epoch it has done so too (rankfm/_rankfm.pyx:329)
    if (rc == RFM_OK || rc >= RFM_ERR_NONFINITE) {
        for (Item &it : items) {
            if (!it.out) continue;
            hipError_t e = hipMemcpy(const_cast<void *>(it.src), *it.dst, it.bytes, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { cleanup(); return hip_fail(e, "hipMemcpy D2H"); }
        }
    }
    cleanup();
    return rc;
}

}  // extern "C"
