// rfm_sgd_segments.hpp -- sgd_rows_kernel (reference order) and sgd_segments_kernel (production Hogwild, BPR; the frozen stripes).
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
template <int G, int KPL, bool FRESH, bool HOT = false, bool WARPB = true, bool STRIPE = false, bool VISPLIT = false>
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
    typedef RowStep<G, KPL, false, false, true, FRESH, false, HOT, WARPB, STRIPE, 0, VISPLIT> Step;
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
