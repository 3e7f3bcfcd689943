// rfm_sgd_segments.hpp -- sgd_rows_kernel (reference order) and sgd_segments_kernel (production Hogwild, BPR).
#pragma once
#include "rfm_rowstep.hpp"

namespace rfm {

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

// ---------------------------------------------------------------------------------------------
// segments kernel (production Hogwild): see the header comment.  Each group is a little state machine
//   [fetch segment + v_u] -> row, row, ... -> [write back v_u delta] -> next segment
// so the four groups of a wavefront stay busy although their segments differ in length.
// ---------------------------------------------------------------------------------------------
// The HOT instantiation uses 1024 threads: the hot-row accumulators are per workgroup, and fewer, larger workgroups combine more
// touches per publication at the same amount of unpublished work.  (Models with features run sgd_features_kernel.)
template <int G, int KPL, bool FRESH, bool HOT = false, bool WARPB = true, bool VISPLIT = false>
__global__ void __launch_bounds__(HOT ? 1024 : 256) sgd_segments_kernel(const SgdArgs a) {
    const int lane = threadIdx.x & 63;
    const int sub = lane % G;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    int64_t n_groups = ((int64_t)gridDim.x * blockDim.x) / G;
    if (a.max_groups > 0 && a.max_groups < n_groups) n_groups = a.max_groups;
    const int F = a.n_factors;
    extern __shared__ __attribute__((aligned(16))) float lds_tables[];
    lds_float *lds = (lds_float *)lds_tables;
    typedef RowStep<G, KPL, false, false, true, FRESH, false, HOT, WARPB, 0, VISPLIT> Step;
    Step step(a, sub, a.v_uf, a.v_if, a.w_if);
    if constexpr (HOT) {
        // LDS: [n_hot * F] pending factor deltas | [n_hot] pending bias deltas | [n_hot] touch counters
        const int n_acc = a.n_hot * (F + 2);
        for (int k = threadIdx.x; k < n_acc; k += blockDim.x) lds_tables[k] = 0.0f;
        __syncthreads();
        step.hot_acc = (lds_int *)lds;
        step.hot_accw = (lds_int *)(lds + a.n_hot * F);
        step.hot_cnt = (lds_int *)(lds + a.n_hot * (F + 1));
        // steps scale with learning rate x sample weight: unit 2^-24 at the defaults (eta 0.1, weights <= 1)
        const float range = fmaxf(1.0f, __uint_as_float(*a.sw_max_bits) * a.eta * 10.0f);
        step.kHotScale = 16777216.0f / range;
        step.kHotUnit = range / 16777216.0f;
    }

    double ll_acc = 0.0;
    unsigned draw_acc = 0;
    const int lane_base = lane - sub;
    stamp_clock(a, 0);
    int64_t sp = a.pos_begin + (a.single_group ? 0 : group);      // position in the epoch's segment order
    const int64_t stride = a.single_group ? 1 : n_groups;
    bool active = sp < a.pos_end && (a.single_group ? group == 0 : group < n_groups);
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
    bool have = false;
    int32_t u = 0, begin = 0, len = 0, t = 0, len_bits = 0;
    uint32_t seg_key = 0;
    int64_t lo = 0, hi = 0;
    float vu[KPL], vu0[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) vu[k] = vu0[k] = 0.0f;

    // (sweeping turns, HOT: wavefront w of the workgroup sweeps in iterations w x every, w x every + wavefronts x every, ...)
    const int sweep_mod = (int)(blockDim.x >> 6) * (a.hot_sweep_every > 0 ? a.hot_sweep_every : 1);
    const int sweep_at = (int)(threadIdx.x >> 6) * (a.hot_sweep_every > 0 ? a.hot_sweep_every : 1);
    for (int iter = 0;; ++iter) {
        if (!__any(active)) break;
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
            if (!a.hot_direct && iter % sweep_mod == sweep_at) {
                RFM_COLD_ARGS(c)                                 // (the rarely executed parts read their arguments afresh: cold_args)
                for (int line = blockIdx.x; line < hot_lines(c); line += gridDim.x) hot_sweep_line(c, line);
            }
        }
        if (active && !have) {
            RFM_COLD_ARGS(c)
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
        }
        if (active) {
            const int32_t pos = begin + (int32_t)rfm_perm((uint32_t)t, (uint32_t)len, (uint32_t)len_bits, seg_key);
            const int32_t i = a.csr_items[pos];
            const float sw = a.sw_csr[pos];
            step(rfm_row_key(a.epoch_key, (uint32_t)pos), u, i, sw, lo, hi, vu, ll_acc, draw_acc);
            if (++t == len) {
                RFM_COLD_ARGS(c)
                // one write-back per segment; other segments of a heavy user may be in flight, so add the delta
#pragma unroll
                for (int k = 0; k < KPL; ++k)
                    if (sub + G * k < F) atomic_add_f32(c.v_u + (size_t)u * F + sub + G * k, vu[k] - vu0[k]);
                have = false;
                if (dynamic) {
                    int64_t nxt = -1;
                    if (sub == 0) nxt = tickets.take(c);
                    sp = __shfl(nxt, lane_base);
                    active = sp >= 0;
                } else {
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
            if (d != 0.0f) atomic_add_f32(a.hot_direct ? a.v_i + vi_index(a, a.hot_item[k / F], k % F)
                                                       : a.hot_bins_v + hot_bin_v(a, blockIdx.x % kHotBins, k / F, k % F), d);
        }
        for (int k = threadIdx.x; k < a.n_hot; k += blockDim.x) {
            const float d = (float)step.hot_accw[k] * step.kHotUnit;
            if (d != 0.0f) atomic_add_f32(a.hot_direct ? a.w_i + (size_t)a.hot_item[k] * a.w_stride : a.hot_bins_w + hot_bin_w(a, blockIdx.x % kHotBins, k), d);
        }
    }
    flush_counters(a, ll_acc, draw_acc);
    stamp_clock(a, 1);
}

}  // namespace rfm
