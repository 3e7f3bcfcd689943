// rfm_rowstep.hpp -- RowStep: one SGD step (rankfm/_rankfm.pyx:230-326) by one row group, shared by the row loops.
#pragma once
#include "rfm_sgd_common.hpp"

namespace rfm {

// ---------------------------------------------------------------------------------------------
// one SGD step for one interaction, executed by the G lanes of a row group
//   SERIAL   plain read-modify-write everywhere, MT stream allowed
//   VU_REGS  v_u lives in the caller's registers: the step updates them in place and does not touch v_u memory
//   FRESH    item-row loads bypass L1
// ---------------------------------------------------------------------------------------------
//   LDSF     the dense feature tables (v_uf, v_if, w_if) are read from this workgroup's LDS copy (see TMODE)
//   HOT      updates of hot positive items are accumulated in the workgroup's LDS and published every few touches
//   WARPB    compile the batched WARP draw loop (max_samples > 1); the BPR instantiation stays at ~76 VGPRs without it
//   TMODE    with LDSF: what the step updates (sgd_features_kernel).  0 = the rows only (v_u, v_i, w_i; the tables are a read-only
//            copy), 1 = the tables only (the table trainer: plain read-modify-write on the master copy, groups of a wavefront one
//            after the other), 2 = both (one group alone: the reference's sequential step)
//   VISPLIT  the item factor rows are segment-major (SgdArgs::vi_split): full rows of 16-lane groups, no features
template <int G, int KPL, bool SERIAL, bool FEAT, bool VU_REGS, bool FRESH, bool LDSF = false, bool HOT = false, bool WARPB = true,
          int TMODE = 0, bool VISPLIT = false>
struct RowStep {
    static_assert(!VISPLIT || (G == 16 && !FEAT && !SERIAL), "segment-major item rows: the plain Hogwild row loops of 16-lane groups");
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

    // the user's sorted item list, held across the lanes when it has at most UL x G entries (lane s: entries s, s+G, ...; -1 pads):
    // the membership test of a draw is then UL compares and a ballot instead of a memory round trip
    // (WARP's state machine tests up to four candidates per iteration and its configurations have the long lists -- config 5's users hold
    //  100 items on average --, where the memory path is two dependent round trips PER candidate: its kernels keep up to 8 G = 128
    //  entries.  Round 5, same box, alternating builds: config 5's share 243.3 -> 180.8 ms per epoch, config 3 9.2 -> 8.94 ms; the BPR kernel of
    //  config 2 -- 2 % of its rows belong to users with more than 64 items, and it is bound by its atomics -- gains nothing (2.626 against 2.628
    //  ms) and loses nothing: BPR rows of up to 64 factors keep 8 G too, for data whose users are heavier than config 2's; wider BPR rows --
    //  whose segment-major instantiations are short of registers as it is -- keep 4 G.)
    //  Factor rows of at most 32 dwords (k <= 32: BASELINE config 1's k = 20, where MovieLens-like users hold 100 - 200 items) have the
    //  registers to spare for 16 G = 256 entries, BPR and WARP alike.  The pipelined feature row loop (FEAT && LDSF) keeps 8 G as well since
    //  the end of round 6: the compiler fits them into the 168 registers of its 768-thread workgroups without a spill; config 4's users hold
    //  ~50 items and gain nothing measurable (3.58 against 3.61 ms), data with 65 - 128 items per user skip the memory path.
    static constexpr int UL = (!FEAT && !SERIAL && G == 16) ? (KPL <= 2 ? 16 : ((WARPB || KPL <= 4) ? 8 : 4)) : ((FEAT && LDSF && !SERIAL && G == 16) ? 8 : 4);
    int32_t ulist[UL];
    bool ulist_ok = false;
    float user_scale = 1.0f;          // damping of this user's step (SgdArgs::user_cap), constant over a segment
    __device__ __forceinline__ void load_ulist(int64_t lo, int64_t hi) {
        user_scale = fminf(1.0f, a.user_cap / (float)(hi - lo));
        ulist_ok = (hi - lo) <= UL * G;
        if (ulist_ok) {
#pragma unroll
            for (int k = 0; k < UL; ++k) {
                const int64_t idx = lo + sub + (int64_t)G * k;
                ulist[k] = idx < hi ? a.csr_items[idx] : -1;
            }
        }
    }
    __device__ __forceinline__ bool member(int64_t lo, int64_t hi, int32_t item) const {
        if (ulist_ok) {
            bool f = false;
#pragma unroll
            for (int k = 0; k < UL; ++k) f |= (ulist[k] == item);
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
    // factor row + bias of item `it`
    __device__ __forceinline__ void fetch_item(int32_t it, float (&v)[KPL], float &w) const {
        if constexpr (VISPLIT) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) v[k] = load_f32<FRESH>(a.v_i + vi_off(it, k));
        } else load_row<FRESH>(a.v_i + (size_t)it * F, v);
        w = load_f32<FRESH>(a.w_i + (size_t)it * a.w_stride);
    }
    // raw draw -> candidate item
    __device__ __forceinline__ int32_t draw_item(uint32_t raw) const { return (int32_t)rfm_draw_to_item(raw, (uint32_t)a.n_items); }

    __device__ __forceinline__ RowStep(const SgdArgs &args, int sub_, TabPtr v_uf, TabPtr v_if, TabPtr w_if)
        : a(args), sub(sub_), F(args.n_factors), t_v_uf(v_uf), t_v_if(v_if), t_w_if(w_if) {}

    __device__ __forceinline__ int dword_f(int k) const { return sub + G * k; }
    // element index in `v_i` of this lane's dword k of item `it`
    __device__ __forceinline__ size_t vi_off(int32_t it, int k) const {
        if constexpr (VISPLIT) return ((size_t)k * (size_t)a.n_items + (size_t)it) * G + sub;
        else return (size_t)it * F + dword_f(k);
    }
    __device__ __forceinline__ bool dword_ok(int k) const { return dword_f(k) < F; }

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
                                             float (&B)[KPL], float &wi, int slot = -1, const XV *xit = nullptr) const {
        fetch_item(it, vi, wi);
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
    __device__ __forceinline__ int32_t next_negative(int64_t lo, int64_t hi, uint32_t row_key, uint32_t &attempt) const {
        int32_t j = 0;
        if (SERIAL && a.rng == 0 /* RFM_RNG_MT19937 */) {
            if (sub == 0) {
                do { j = (int32_t)(mt_next_global(a.mt_state) % (uint32_t)a.n_items); } while (is_member(a.csr_items, lo, hi, j));
            }
            j = __shfl(j, (threadIdx.x & 63) - sub);
        } else {
            for (;;) {
                j = draw_item(rfm_draw(row_key, attempt));
                ++attempt;
                if (!member(lo, hi, j)) break;
                if (attempt >= kMaxAttempts) { if (sub == 0) atomicOr(a.error_flags, 1u); break; }
            }
        }
        return j;
    }

    // `vu` holds v_u[u] on entry; with VU_REGS it holds the updated row on exit
    __device__ __forceinline__ void operator()(uint32_t row_key, int32_t u, int32_t i, float sw, int64_t lo, int64_t hi,
                                               float (&vu)[KPL], double &ll_acc, unsigned &draw_acc) const {
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
            if (a.pos_scale) {
                pos_scale_i = a.pos_scale[i];
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
            j = next_negative(lo, hi, row_key, attempt);
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
        ut_ui = utility(vu, A, i, vi, Bi, wi, slot, &xi);    // :239

        // WARP sampling loop (:244-264); BPR is max_samples == 1
        // first draw (all of BPR): one candidate at a time
        for (; s <= ((BATCH_WARP || (!SERIAL && !FEAT && !WARPB)) ? 1 : a.max_samples); ++s) {
            const int32_t cand = next_negative(lo, hi, row_key, attempt);
            float vc[KPL], Bc[KPL], wc;
            if constexpr (FEAT) { if (a.has_if) xload(a.x_if + (size_t)cand * a.n_if, a.n_if, xc); }
            const float pu = ut_ui - utility(vu, A, cand, vc, Bc, wc, -1, &xc);   // :256-257
            sampled = s;
            if (pu < min_pu || j < 0) {                                   // :259-261 (j < 0: keep a valid index under NaN)
                if (pu < min_pu) min_pu = pu;
                j = cand; wj = wc;
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
                bool mem[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) c[q] = draw_item(rfm_draw(row_key, attempt + q));
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
                        if (!mem[q0 + q]) fetch_item(c[q0 + q], vc[q], wc[q]);
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
                            min_pu = pu; j = c[q0 + q]; wj = wc[q];
#pragma unroll
                            for (int k = 0; k < KPL; ++k) vj[k] = vc[q][k];
                        }
                        if (pu < kMargin) done = true;
                    }
                }
                if (attempt >= kMaxAttempts) { if (sub == 0) atomicOr(a.error_flags, 1u); break; }
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
            if (a.pos_scale) {
                float sc = a.scale_in_pad ? a.w_i[(size_t)j * a.w_stride + 1] : a.pos_scale[j];
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
            eta_u = eta * fminf(1.0f, a.user_cap / (float)(hi - lo));
            eta_i = eta * pos_scale_i;
            if constexpr (!LDSF) eta_f = eta * a.feat_scale;
        }
        float nvu[KPL], dij[KPL];     // updated v_u, updated (v_i - v_j) (feature paths)
        // item biases (:279-280) -- one lane per group
        if (UPD_ROWS && sub == 0) {
            const float dwi = eta_i * (g * (d_outer * 1.0f) - reg_a * wi);
            const float dwj = eta_j * (g * (d_outer * -1.0f) - reg_a * wj);
            if (HOT && slot >= 0) hot_add(hot_accw + slot, dwi);
            else apply_f32<SERIAL>(a.w_i + (size_t)i * a.w_stride, wi, dwi);
            apply_f32<SERIAL>(a.w_i + (size_t)j * a.w_stride, wj, dwj);
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
                else apply_f32<SERIAL>(a.v_i + vi_off(i, k), vi[k], d_i);
                apply_f32<SERIAL>(a.v_i + vi_off(j, k), vj[k], d_j);
            }
        }
        if constexpr (VU_REGS && UPD_ROWS) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) vu[k] = nvu[k];
        }
        if constexpr (HOT) {
            if (slot >= 0) {
                // every hot_period-th toucher of the slot publishes what the workgroup has accumulated for it
                // (a keyed coin with probability 1 / period instead of a shared counter: no LDS round trip on the row's path)
                if (__umulhi(rfm_mix32(row_key ^ 0x7A5C3B1DU), (uint32_t)a.hot_period[slot]) == 0u) {     // probability 1 / period
                    RFM_COLD_ARGS(c)
#pragma unroll
                    for (int k = 0; k < KPL; ++k) {
                        if (!dword_ok(k)) continue;
                        const float d = hot_take(hot_acc + slot * F + dword_f(k));
                        if (d != 0.0f)
                            atomic_add_f32(c.hot_direct ? a.v_i + vi_off(i, k)
                                                        : c.hot_bins_v + hot_bin_v(c, blockIdx.x % kHotBins, slot, dword_f(k)), d);
                    }
                    if (sub == 0) {
                        const float d = hot_take(hot_accw + slot);
                        if (d != 0.0f) atomic_add_f32(c.hot_direct ? a.w_i + (size_t)i * a.w_stride : c.hot_bins_w + hot_bin_w(c, blockIdx.x % kHotBins, slot), d);
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

}  // namespace rfm
