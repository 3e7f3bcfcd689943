/* rfm_async_sim.c -- a CPU MODEL of how the stripe kernel executes an epoch.  TEST / ANALYSIS INFRASTRUCTURE (like rfm_oracle.c:
 * only tests/ and tools/ may build or call it; nothing under rankfm_amd/ does).
 *
 * Not the reference's algorithm and not the product: a sequential program that replays one BPR epoch (no features) in the
 * order and with the visibility rules of rankfm_amd/csrc/rfm_sgd.hpp's sgd_segments_kernel<STRIPE>, so that questions about
 * parity ("what does a 24-row window cost", "what would another view of the positive item do") can be answered on a CPU in
 * seconds instead of on GPU minutes.  The model:
 *   - `n_groups` row groups work in lock-step ROUNDS: in round r every group processes the r-th row of its walk (the caller
 *     passes the rows sorted by round; rankfm_amd/order.py knows each row's group and iteration).  What a row adds to memory with
 *     atomics -- the positive item's step, a negative outside the stripe, a finished segment's user delta, publications -- becomes
 *     visible to everybody at the END of the round ("in flight": the real kernel's interleaving is finer, its staleness the same);
 *   - v_u lives in the group's registers over a user segment (loaded at its first row, written back as a delta after its last);
 *   - a workgroup (`gpb` consecutive groups) draws a window's negatives from a stripe of R items (include/rfm_rng.h), keeps their
 *     pending steps in "LDS" -- visible at once, to the workgroup only -- and publishes them when the window ends;
 *   - views: negative = memory + the workgroup's pending sum of its stripe row; positive = memory + the workgroup's pending sum
 *     of a hot slot + `mean_view` x cover x the mean pending sum of the workgroup's stripe rows;
 *   - hot items accumulate per workgroup and are published on the row key's coin (1 / period);
 *   - step damping: pos_step[item], user_step[user] (rfm_oracle_fit_ex).
 * Left out: fixed-point rounding of the LDS sums, the one-row-ahead prefetch, L2 staleness, the hot-row bins.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rfm_rng.h"

typedef struct {
    int64_t n_rows;               /* rows of the epoch, in processing order */
    int32_t U, I, F;
    float alpha, eta;
    uint32_t epoch_key;
    int32_t n_groups, gpb, grid;  /* row groups, groups per workgroup, workgroups */
    int32_t stripe_rows, stripe_window;   /* 0 rows: draws over the whole catalogue, atomics per negative */
    float cover;                  /* share of the catalogue that sits in some stripe */
    float mean_view;              /* 1: the committed mean-field view of the positive; 0: none; anything between / beyond: experiments */
    int32_t n_hot;
    int32_t defer;                /* 1: atomics become visible at the end of the round; 0: at once (sequential limit) */
    int32_t publish_now;          /* 1: a stripe row's step goes to memory at once (with defer = 0: the sequential limit on the same draws) */
    int32_t phases;               /* experiment: stripes of window w lie in part w % phases of the item permutation (1: the committed schedule) */
} rfm_sim_params;

typedef struct { int32_t idx; int32_t kind; } sim_ref;       /* kind 0: item row (F + 1 floats), 1: user row (F floats) */

static int member_binary(int item, const int32_t *items, int64_t n) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t md = lo + ((hi - lo) >> 1); if (items[md] == item) return 1; if (items[md] < item) lo = md + 1; else hi = md; }
    return 0;
}

/* item in row r of the stripe starting at position `start` of the epoch's item permutation; with `phases` > 1 the stripe wraps inside
 * the part of the permutation that `start` lies in (parts = equal consecutive ranges) */
static uint32_t sim_stripe_item(const rfm_sim_params *p, uint32_t start, uint32_t r, uint32_t item_bits) {
    const uint32_t I = (uint32_t)p->I;
    if (p->phases <= 1) return rfm_stripe_item(p->epoch_key, start, r, I, item_bits);
    const uint32_t part = (uint32_t)(((uint64_t)start * (uint32_t)p->phases) / I);
    const uint32_t lo = (uint32_t)(((uint64_t)part * I + p->phases - 1) / (uint32_t)p->phases);
    const uint32_t hi = (uint32_t)(((uint64_t)(part + 1) * I + p->phases - 1) / (uint32_t)p->phases);
    uint32_t q = start + r;
    if (q >= hi) q -= hi - lo;
    return rfm_perm(q, I, item_bits, p->epoch_key ^ 0x2545F491U);
}

/* rows: per processed row (sorted by round, then group): CSR position, group, round, flags (1 = first row of its segment,
 * 2 = last), the start of the stripe it draws from.  interactions / sample_weight are indexed by CSR position.
 * hot_slot [I]: slot of a hot item or -1; hot_period [n_hot]. */
int rfm_async_sim_epoch(const rfm_sim_params *p, const int32_t *interactions, const float *sample_weight,
                        const int64_t *csr_off, const int32_t *csr_items,
                        const int32_t *row_pos, const int32_t *row_group, const int32_t *row_round, const int32_t *row_flags,
                        const int32_t *row_stripe, const float *pos_step, const float *user_step,
                        const int32_t *hot_slot, const int32_t *hot_period,
                        float *w_i, float *v_u, float *v_i, double *ll_out, int64_t *fallback_out) {
    const int F = p->F, FS = p->F + 1, I = p->I, R = p->stripe_rows;
    const uint32_t item_bits = rfm_perm_bits((uint32_t)I);
    const float reg = 2.0f * p->alpha, eta = p->eta;
    const double multiplier = log((double)(I - 1)) / log((double)I);      /* max_samples 1: sampled = 1 */
    float *vu_reg = (float *)calloc((size_t)p->n_groups * 2 * F, sizeof(float));          /* [group][vu | vu0] */
    float *pend = R > 0 ? (float *)calloc((size_t)p->grid * R * FS, sizeof(float)) : NULL;  /* [wg][row][F + 1] */
    float *psum = (float *)calloc((size_t)p->grid * FS, sizeof(float));
    int32_t *stripe_of = (int32_t *)malloc(sizeof(int32_t) * (size_t)p->grid);            /* current stripe start per workgroup, -1 none */
    int32_t *sitem = R > 0 ? (int32_t *)malloc(sizeof(int32_t) * (size_t)p->grid * R) : NULL;
    float *hot = p->n_hot > 0 ? (float *)calloc((size_t)p->grid * p->n_hot * FS, sizeof(float)) : NULL;
    /* deferred atomics of the current round */
    size_t cap = 1 << 16, n_def = 0;
    sim_ref *dref = (sim_ref *)malloc(sizeof(sim_ref) * cap);
    float *dval = (float *)malloc(sizeof(float) * cap * FS);
    if (!vu_reg || !psum || !stripe_of || !dref || !dval || (R > 0 && (!pend || !sitem))) return -1;
    for (int w = 0; w < p->grid; ++w) stripe_of[w] = -1;
    double ll = 0.0;
    int64_t fallbacks = 0;

#define DEFER(KIND, IDX, PTR, LEN)                                                                      \
    do {                                                                                                \
        if (!p->defer) {                                                                                \
            if ((KIND) == 0) { float *row_ = v_i + (size_t)(IDX) * F; for (int f_ = 0; f_ < F; ++f_) row_[f_] += (PTR)[f_]; w_i[IDX] += (PTR)[F]; } \
            else { float *row_ = v_u + (size_t)(IDX) * F; for (int f_ = 0; f_ < F; ++f_) row_[f_] += (PTR)[f_]; } \
        } else {                                                                                        \
            if (n_def == cap) { cap *= 2; dref = (sim_ref *)realloc(dref, sizeof(sim_ref) * cap); dval = (float *)realloc(dval, sizeof(float) * cap * FS); } \
            dref[n_def].idx = (IDX); dref[n_def].kind = (KIND);                                         \
            memcpy(dval + n_def * FS, (PTR), sizeof(float) * (LEN));                                    \
            ++n_def;                                                                                    \
        }                                                                                               \
    } while (0)
#define APPLY_DEFERRED()                                                                                \
    do {                                                                                                \
        for (size_t k_ = 0; k_ < n_def; ++k_) {                                                         \
            const float *d_ = dval + k_ * FS;                                                           \
            if (dref[k_].kind == 0) { float *row_ = v_i + (size_t)dref[k_].idx * F; for (int f_ = 0; f_ < F; ++f_) row_[f_] += d_[f_]; w_i[dref[k_].idx] += d_[F]; } \
            else { float *row_ = v_u + (size_t)dref[k_].idx * F; for (int f_ = 0; f_ < F; ++f_) row_[f_] += d_[f_]; } \
        }                                                                                               \
        n_def = 0;                                                                                      \
    } while (0)

    float *tmp = (float *)malloc(sizeof(float) * 4 * FS);
    float *vi = tmp, *vj = tmp + FS, *d_i = tmp + 2 * FS, *d_j = tmp + 3 * FS;
    int32_t cur_round = -1;
    for (int64_t n = 0; n < p->n_rows; ++n) {
        if (row_round[n] != cur_round) { APPLY_DEFERRED(); cur_round = row_round[n]; }
        const int pos = row_pos[n], g = row_group[n], wg = g / p->gpb;
        const int u = interactions[2 * (size_t)pos], i = interactions[2 * (size_t)pos + 1];
        const float sw = sample_weight[pos];
        float *vu = vu_reg + (size_t)g * 2 * F, *vu0 = vu + F;
        if (row_flags[n] & 1) { memcpy(vu, v_u + (size_t)u * F, sizeof(float) * F); memcpy(vu0, vu, sizeof(float) * F); }
        /* window turn-over of the workgroup: publish the old stripe (visible from this round on), load the new one */
        if (R > 0 && row_stripe[n] != stripe_of[wg]) {
            float *pw = pend + (size_t)wg * R * FS;
            if (stripe_of[wg] >= 0)
                for (int r = 0; r < R; ++r) {
                    float *d = pw + (size_t)r * FS;
                    const int it = sitem[(size_t)wg * R + r];
                    float *row = v_i + (size_t)it * F;
                    for (int f = 0; f < F; ++f) row[f] += d[f];
                    w_i[it] += d[F];
                }
            memset(pw, 0, sizeof(float) * (size_t)R * FS);
            memset(psum + (size_t)wg * FS, 0, sizeof(float) * FS);
            stripe_of[wg] = row_stripe[n];
            for (int r = 0; r < R; ++r)
                sitem[(size_t)wg * R + r] = (int32_t)sim_stripe_item(p, (uint32_t)row_stripe[n], (uint32_t)r, item_bits);
        }
        /* positive item's view */
        const int slot = hot_slot ? hot_slot[i] : -1;
        memcpy(vi, v_i + (size_t)i * F, sizeof(float) * F);
        vi[F] = w_i[i];
        if (slot >= 0) { const float *h = hot + ((size_t)wg * p->n_hot + slot) * FS; for (int f = 0; f < FS; ++f) vi[f] += h[f]; }
        if (R > 0 && p->mean_view != 0.0f) {
            const float c = p->mean_view * p->cover / (float)R;
            const float *s = psum + (size_t)wg * FS;
            for (int f = 0; f < FS; ++f) vi[f] += s[f] * c;
        }
        /* the negative: rejection sampling inside the stripe (whole catalogue after RFM_STRIPE_ATTEMPTS) */
        const uint32_t rkey = rfm_row_key(p->epoch_key, (uint32_t)pos);
        const int32_t *items_u = csr_items + csr_off[u];
        const int64_t n_u = csr_off[u + 1] - csr_off[u];
        uint32_t attempt = 0;
        int j, jrow;
        for (;;) {
            jrow = -1;
            if (R > 0 && attempt < RFM_STRIPE_ATTEMPTS) {
                jrow = (int)rfm_draw_to_item(rfm_draw(rkey, attempt++), (uint32_t)R);
                j = sitem[(size_t)wg * R + jrow];
            } else j = (int)rfm_draw_to_item(rfm_draw(rkey, attempt++), (uint32_t)I);
            if (!member_binary(j, items_u, n_u)) break;
        }
        if (R > 0 && jrow < 0) ++fallbacks;
        memcpy(vj, v_i + (size_t)j * F, sizeof(float) * F);
        vj[F] = w_i[j];
        if (jrow >= 0) { const float *d = pend + ((size_t)wg * R + jrow) * FS; for (int f = 0; f < FS; ++f) vj[f] += d[f]; }
        float ui = vi[F], uj = vj[F];
        for (int f = 0; f < F; ++f) { ui += vu[f] * vi[f]; uj += vu[f] * vj[f]; }
        const double pu = (double)(ui - uj);
        ll += log(1.0 / (1.0 + exp(-pu)));
        const float d_outer = (float)(1.0 / (exp(pu) + 1.0));
        const float gm = sw * (float)multiplier;
        const float eta_i = eta * (pos_step ? pos_step[i] : 1.0f), eta_u = eta * (user_step ? user_step[u] : 1.0f);
        for (int f = 0; f < F; ++f) {
            const float g_u = vi[f] - vj[f], g_i = vu[f];
            const float d_u = eta_u * (gm * (d_outer * g_u) - reg * vu[f]);
            d_i[f] = eta_i * (gm * (d_outer * g_i) - reg * vi[f]);
            d_j[f] = eta * (gm * (d_outer * -g_i) - reg * vj[f]);
            vu[f] += d_u;
        }
        d_i[F] = eta_i * (gm * d_outer - reg * vi[F]);
        d_j[F] = eta * (gm * -d_outer - reg * vj[F]);
        if (slot >= 0) {
            float *h = hot + ((size_t)wg * p->n_hot + slot) * FS;
            for (int f = 0; f < FS; ++f) h[f] += d_i[f];
            if ((uint32_t)(((uint64_t)rfm_mix32(rkey ^ 0x7A5C3B1DU) * (uint64_t)(uint32_t)hot_period[slot]) >> 32) == 0u) {
                DEFER(0, i, h, FS);
                memset(h, 0, sizeof(float) * FS);
            }
        } else DEFER(0, i, d_i, FS);
        if (jrow >= 0 && p->publish_now) DEFER(0, j, d_j, FS);
        else if (jrow >= 0) {
            float *d = pend + ((size_t)wg * R + jrow) * FS, *s = psum + (size_t)wg * FS;
            for (int f = 0; f < FS; ++f) { d[f] += d_j[f]; s[f] += d_j[f]; }
        } else DEFER(0, j, d_j, FS);
        if (row_flags[n] & 2) {
            for (int f = 0; f < F; ++f) d_i[f] = vu[f] - vu0[f];
            DEFER(1, u, d_i, F);
        }
    }
    APPLY_DEFERRED();
    /* publish what is still pending */
    if (R > 0)
        for (int w = 0; w < p->grid; ++w) {
            if (stripe_of[w] < 0) continue;
            for (int r = 0; r < R; ++r) {
                const float *d = pend + ((size_t)w * R + r) * FS;
                const int it = sitem[(size_t)w * R + r];
                float *row = v_i + (size_t)it * F;
                for (int f = 0; f < F; ++f) row[f] += d[f];
                w_i[it] += d[F];
            }
        }
    if (hot)
        for (int w = 0; w < p->grid; ++w)
            for (int i = 0; i < I; ++i) {
                const int s = hot_slot[i];
                if (s < 0) continue;
                const float *h = hot + ((size_t)w * p->n_hot + s) * FS;
                float *row = v_i + (size_t)i * F;
                for (int f = 0; f < F; ++f) row[f] += h[f];
                w_i[i] += h[F];
            }
    if (ll_out) *ll_out = ll;
    if (fallback_out) *fallback_out = fallbacks;
    free(vu_reg); free(pend); free(psum); free(stripe_of); free(sitem); free(hot); free(dref); free(dval); free(tmp);
    return 0;
}
