// rfm_sgd_warp.hpp -- sgd_warp_kernel: WARP's candidate loop as a per-group state machine (configs 3 and 5).
#pragma once
#include "rfm_rowstep.hpp"

namespace rfm {

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
//
// DEFER (full rows, plain loads): a finished row's atomics are issued BEHIND the next iteration's gathers instead of in front of them.
// A wavefront has ONE in-order counter for its vector-memory operations (vmcnt: gfx9 has no separate store counter), so a wait for
// the gathered rows is also a wait for every atomic issued before them -- and the round trip of a memory-side fp32 atomic is the
// longest there is: with the row updates left out the kernel runs 27 % faster on config 3 (profiles/r05_notes.md section 13), which is
// what waiting for them costs.  Issued behind the gathers they have a whole iteration to retire in, and the wait for the gathers
// becomes `s_waitcnt vmcnt(<the atomics just issued>)`.  The compiler cannot write that wait: the atomics are conditional (a group
// has a finished row or not, its positive item is an LDS-accumulated hot row or not), and across a branch that may skip them its
// counter model falls back to vmcnt(0).  So in this form the gathers are issued from inline assembly (the compiler does not know they
// are loads and waits for nothing), the pending atomics follow as ordinary code on one of three wavefront-uniform paths -- none
// pending / negative rows only / positive and negative rows -- on which the number of atomic INSTRUCTIONS issued is fixed (KPL + 1 per
// row kind: every `if` around them has at least one lane inside, by the path's own condition), and each path ends in its own counted
// wait, after which the gathered registers are handed to the compiler (`settle`).  Anything the compiler adds in between can only
// make the wait longer than needed, never shorter: younger operations it does not know of are waited for as well.
// One group alone (the sequential program) keeps the old order: the next row must read what this one wrote.
template <int OFF>
__device__ __forceinline__ float gather_f32(const float *p) {
    float x;
    asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(x) : "v"(p), "n"(OFF) : "memory");
    return x;
}
template <int N>
__device__ __forceinline__ void wait_gathers_behind() {      // at most N younger vector-memory operations still in flight
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
__device__ __forceinline__ void settle(float &x) { asm volatile("" : "+v"(x)); }
// lane q (0 ... 3; a constant once the caller's loop is unrolled) of every 16-lane row, in all of the row's lanes: DPP row_newbcast
__device__ __forceinline__ float row_bcast(float x, int q) {
    switch (q) {
    case 0: return dpp_mov<0x150>(x);
    case 1: return dpp_mov<0x151>(x);
    case 2: return dpp_mov<0x152>(x);
    default: return dpp_mov<0x153>(x);
    }
}

template <int G, int KPL, bool FRESH, bool HOT, bool FULL>
__global__ void __launch_bounds__(HOT ? 1024 : 256) sgd_warp_kernel(const SgdArgs a) {
    static_assert(G == 16, "the WARP state machine is written for 16-lane row groups");
    constexpr int NC = KPL >= 8 ? 2 : 4;                    // item rows a group gathers per iteration
    constexpr bool DEFER = FULL && !FRESH && KPL <= 4;      // (wider rows: the pending deltas do not fit beside two gathered rows in 128 registers)
    constexpr int kSweepEvery = 4;                          // (an iteration is a fraction of a row step: sweep the bins every 4th turn)
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, lane_base = lane - sub;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    int64_t n_groups = ((int64_t)gridDim.x * blockDim.x) / G;
    if (a.max_groups > 0 && a.max_groups < n_groups) n_groups = a.max_groups;
    const int F = FULL ? G * KPL : a.n_factors;
    auto ok = [&](int kk) { return FULL || sub + G * kk < F; };
    auto vi_at = [&](int32_t item, int k) { return a.v_i + (size_t)item * F + sub + G * k; };      // (row-major: see rfm_api.hip, vi_split_eligible)
    extern __shared__ __attribute__((aligned(16))) float lds_tables[];
    lds_float *lds = (lds_float *)lds_tables;
    typedef RowStep<G, KPL, false, false, true, FRESH, false, HOT, true> Step;       // draws, membership test, fixed-point hot sums
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
    // DEFER: a finished row's deltas wait in vi / vj / wi / wj (dead until the group's next row is examined) for the next gathers
    bool pend = false;
    int32_t pi = -1, pj = 0;                                // (pi < 0: the positive item's delta went to the LDS sums)
    auto issue_pending = [&](bool with_pos) {
        if (with_pos && pend && pi >= 0) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) atomic_add_f32(vi_at(pi, k), vi[k]);
            if (sub == 0) atomic_add_f32(a.w_i + (size_t)pi * a.w_stride, wi);
        }
        if (pend) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) atomic_add_f32(vi_at(pj, k), vj[k]);
            if (sub == 0) atomic_add_f32(a.w_i + (size_t)pj * a.w_stride, wj);
        }
    };

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
            // (DEFER: everything the segment start loaded has arrived before the row loop goes on -- the compiler would otherwise wait
            //  for it with vmcnt(0) at every row start, where its model merges this path in, and with it for the atomics in flight)
            if constexpr (DEFER) __builtin_amdgcn_s_waitcnt(0x0F70);
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
                c[q] = step.draw_item(rfm_draw(row_key, attempt));
                ++attempt;
                skip[q] = step.member(lo, hi, c[q]);      // (rankfm/_rankfm.pyx:250-253: a drawn item of the user's own is drawn again)
            }
        }
        // (a slot with nothing to examine still gathers a row -- item 0's -- and ignores it: an unconditional load is cheaper than
        //  the branch around a conditional one, and the kernel is not bound by its requests)
        float vc[NC][KPL];
        float wsc = 0.0f, ssc = 1.0f;                       // lane q of the group: bias and step scale of slot q's item
        int32_t mine = 0;                                   // ... and the item itself
        if constexpr (DEFER) {
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int32_t cq = skip[q] ? 0 : c[q];
                const float *row = vi_at(cq, 0);
                vc[q][0] = gather_f32<0>(row);
                if constexpr (KPL > 1) vc[q][1] = gather_f32<4 * G>(row);
                if constexpr (KPL > 2) vc[q][2] = gather_f32<8 * G>(row);
                if constexpr (KPL > 3) vc[q][3] = gather_f32<12 * G>(row);
                static_assert(!DEFER || KPL <= 4, "the gathers spell out the row's dwords");
                mine = sub == q ? cq : mine;
            }
            wsc = gather_f32<0>(a.w_i + (size_t)mine * a.w_stride);
            if (a.pos_scale) ssc = gather_f32<0>(a.scale_in_pad ? a.w_i + (size_t)mine * a.w_stride + 1 : a.pos_scale + mine);
            // the finished rows' atomics, behind the gathers; then the wait that leaves exactly them in flight
            if (__any(pend)) {
                if (__any(pend && pi >= 0)) {
                    issue_pending(true);
                    wait_gathers_behind<2 * (KPL + 1)>();
                } else {
                    issue_pending(false);
                    wait_gathers_behind<KPL + 1>();
                }
                pend = false;
            } else {
                wait_gathers_behind<0>();
            }
#pragma unroll
            for (int q = 0; q < NC; ++q) {
#pragma unroll
                for (int k = 0; k < KPL; ++k) settle(vc[q][k]);
            }
            settle(wsc);
            settle(ssc);
        } else {
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int32_t cq = skip[q] ? 0 : c[q];
#pragma unroll
                for (int k = 0; k < KPL; ++k) vc[q][k] = ok(k) ? load_f32<FRESH>(vi_at(cq, k)) : 0.0f;
                mine = sub == q ? cq : mine;
            }
            wsc = load_f32<FRESH>(a.w_i + (size_t)mine * a.w_stride);
            if (a.pos_scale) ssc = a.scale_in_pad ? a.w_i[(size_t)mine * a.w_stride + 1] : a.pos_scale[mine];
        }
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
            // (straight-line: every slot's verdict is a handful of selects -- as branches, the four slots were a dozen exec-mask
            //  regions with the register copies that come with them; slot q's bias and step scale come from lane q of the group
            //  through a DPP row broadcast instead of an LDS permute)
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const float wq = row_bcast(wsc, q), sq = row_bcast(ssc, q);
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
                }
                const float dot = group_sum<G>(part[q]);
                const bool take = !(q == 0 && starts) && !done && !skip[q] && s <= a.max_samples;
                const float pu = ut_ui - (wq + dot);                               // :256-257
                const bool lower = pu < min_pu;
                const bool better = take && (lower || j < 0);                      // :259-261 (j < 0: keep a valid index under NaN)
                sampled = take ? s : sampled;
                s += take ? 1 : 0;
                min_pu = (take && lower) ? pu : min_pu;
                j = better ? c[q] : j;
                wj = better ? wq : wj;
                neg_scale_j = better ? sq : neg_scale_j;                           // (raw: decoded when the row is finished)
#pragma unroll
                for (int k = 0; k < KPL; ++k) vj[k] = better ? vc[q][k] : vj[k];
                done = done || (take && pu < kMargin);                             // :263-264
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
            float multiplier;                                                      // :269 (integer division inside the log)
            if (n_mult) multiplier = l_mult[sampled];
            else if constexpr (DEFER) {        // (max_samples > 255: a load of its own, waited for inside this branch)
                multiplier = gather_f32<0>(c.multiplier + sampled);
                wait_gathers_behind<0>();
                settle(multiplier);
            } else multiplier = c.multiplier[sampled];
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
                                atomic_add_f32(c.hot_direct ? vi_at(i, k)
                                                            : c.hot_bins_v + hot_bin_v(c, blockIdx.x % kHotBins, slot, sub + G * k), d);
                        }
                        if (sub == 0) {
                            const float d = step.hot_take(step.hot_accw + slot);
                            if (d != 0.0f) atomic_add_f32(c.hot_direct ? a.w_i + (size_t)i * a.w_stride : c.hot_bins_w + hot_bin_w(c, blockIdx.x % kHotBins, slot), d);
                        }
                    }
                }
            }
            if (DEFER && !c.single_group) {
                pend = true;
                pi = hot_done ? -1 : i;
                pj = j;
#pragma unroll
                for (int k = 0; k < KPL; ++k) { vi[k] = d_i[k]; vj[k] = d_j[k]; }
                wi = dwi;
                wj = dwj;
            } else {
                if (!hot_done) {
#pragma unroll
                    for (int k = 0; k < KPL; ++k)
                        if (ok(k)) atomic_add_f32(vi_at(i, k), d_i[k]);
                    if (sub == 0) atomic_add_f32(a.w_i + (size_t)i * a.w_stride, dwi);
                }
#pragma unroll
                for (int k = 0; k < KPL; ++k)
                    if (ok(k)) atomic_add_f32(vi_at(j, k), d_j[k]);
                if (sub == 0) atomic_add_f32(a.w_i + (size_t)j * a.w_stride, dwj);
            }
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
    if constexpr (DEFER) issue_pending(true);        // (the last finished rows)
    if constexpr (HOT) {          // publish whatever is still pending
        __syncthreads();
        for (int k = threadIdx.x; k < a.n_hot * F; k += blockDim.x) {
            const float d = (float)step.hot_acc[k] * step.kHotUnit;
            if (d != 0.0f) atomic_add_f32(a.hot_direct ? a.v_i + (size_t)a.hot_item[k / F] * F + (k % F)
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
