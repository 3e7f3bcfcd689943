/* rfm_oracle.c -- CPU restatement of the reference's `_fit` hot path.  TEST INFRASTRUCTURE.
 *
 * This file is the parity checker for the MI355X engine.  Only tests/, the smoke() entry of
 * __graft_entry__.py and the `cpu_baseline` leg of bench.py may build, load or call it; the
 * product (rankfm_amd/) never does and fails loudly when its HIP library is missing.
 *
 * What it restates (all paths relative to /root/reference):
 *   rankfm/_rankfm.pyx:122-342   _fit: sequential BPR/WARP SGD over shuffled interactions
 *   rankfm/_rankfm.pyx:48-89     compute_ui_utility: pointwise FM utility
 *   rankfm/_rankfm.pyx:20-27     lsearch: membership of j in the user's sorted item list
 *   rankfm/mt19937ar/mt19937ar.c:60-73,105-140   MT19937 init_genrand / genrand_int32
 *                                (the published Matsumoto-Nishimura algorithm, re-typed here)
 *   rankfm/_rankfm.pyx:345-390   _predict        (rfm_oracle_predict)
 *   rankfm/_rankfm.pyx:393-460   _recommend      (rfm_oracle_recommend)
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against golden vectors
 * produced by the reference's own compiled `_fit` (tests/golden/make_golden.py imports the
 * reference in the build container; tolerance 2e-6 abs on every weight array, the residue is
 * the reference's -ffast-math).  The MT stream is additionally pinned to the known-answer
 * outputs of init_genrand(1492) (= numpy RandomState(1492) raw stream).
 *
 * Two RNG modes:
 *   RFM_RNG_MT19937 (0)  the reference's behaviour: one serial MT stream seeded 1492 per call,
 *                        `% I`, rejection by membership.  Needs an explicit permutation array
 *                        (the reference's np.random.shuffle is host state we capture).
 *   RFM_RNG_COUNTER (1)  the engine's counter-based draws / permutation (include/rfm_rng.h), so
 *                        the sequential CPU run and the GPU run see identical negatives.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffast-math, the reference's own flags, setup.py:25-26).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rfm_rng.h"

#define RFM_RNG_MT19937 0
#define RFM_RNG_COUNTER 1

/* ------------------------------------------------------------------------------------------
 * MT19937 (rankfm/mt19937ar/mt19937ar.c:44-57 constants and state, :60-73 seeding,
 * :105-140 generation + tempering).  State is per-oracle-call here, not a process global.
 * ------------------------------------------------------------------------------------------ */
#define MT_N 624
#define MT_M 397
typedef struct { uint32_t mt[MT_N]; int mti; } mt_state;

static void mt_seed(mt_state *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int k = 1; k < MT_N; ++k)
        s->mt[k] = 1812433253U * (s->mt[k - 1] ^ (s->mt[k - 1] >> 30)) + (uint32_t)k;
    s->mti = MT_N;
}

static uint32_t mt_next(mt_state *s) {
    if (s->mti >= MT_N) {
        int k;
        for (k = 0; k < MT_N; ++k) {
            uint32_t y = (s->mt[k] & 0x80000000U) | (s->mt[(k + 1) % MT_N] & 0x7fffffffU);
            s->mt[k] = s->mt[(k + MT_M) % MT_N] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        }
        s->mti = 0;
    }
    uint32_t y = s->mt[s->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
}

/* exported for the known-answer test */
void rfm_oracle_mt_stream(uint32_t seed, int n, uint32_t *out) {
    mt_state s;
    mt_seed(&s, seed);
    for (int k = 0; k < n; ++k) out[k] = mt_next(&s);
}

/* ------------------------------------------------------------------------------------------
 * problem description shared by fit / predict / recommend
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t U, I, P, Q, F;
    int32_t has_uf, has_if;          /* the reference's x_uf.any() / x_if.any()  (_rankfm.pyx:193-194) */
    const float *x_uf, *x_if;        /* [U,P], [I,Q] */
    float *w_i, *w_if;               /* [I], [Q] */
    float *v_u, *v_i, *v_uf, *v_if;  /* [U,F], [I,F], [P,F], [Q,F] */
} rfm_model;

/* _rankfm.pyx:48-89.  fp32 accumulator, sequential over f; zero feature entries are skipped. */
static float utility(const rfm_model *m, int u, int i) {
    const int F = m->F;
    const float *vu = m->v_u + (size_t)u * F, *vi = m->v_i + (size_t)i * F;
    float res = m->w_i[i];
    for (int f = 0; f < F; ++f) res += vu[f] * vi[f];
    if (m->has_uf) {
        const float *xu = m->x_uf + (size_t)u * m->P;
        for (int p = 0; p < m->P; ++p) {
            if (xu[p] == 0.0f) continue;
            const float *vuf = m->v_uf + (size_t)p * F;
            for (int f = 0; f < F; ++f) res += xu[p] * (vuf[f] * vi[f]);
        }
    }
    if (m->has_if) {
        const float *xi = m->x_if + (size_t)i * m->Q;
        for (int q = 0; q < m->Q; ++q) {
            if (xi[q] == 0.0f) continue;
            res += xi[q] * m->w_if[q];
            const float *vif = m->v_if + (size_t)q * F;
            for (int f = 0; f < F; ++f) res += xi[q] * (vif[f] * vu[f]);
        }
    }
    return res;
}

/* _rankfm.pyx:20-27 (linear scan: what the reference calls) */
static int member_linear(int item, const int32_t *items, int64_t n) {
    for (int64_t k = 0; k < n; ++k)
        if (items[k] == item) return 1;
    return 0;
}

/* same predicate, O(log n): used in counter mode where only the result matters */
static int member_binary(int item, const int32_t *items, int64_t n) {
    int64_t lo = 0, hi = n - 1;
    while (lo <= hi) {
        int64_t md = lo + (hi - lo) / 2;
        if (items[md] == item) return 1;
        if (items[md] < item) lo = md + 1; else hi = md - 1;
    }
    return 0;
}

typedef struct {
    int64_t N;
    int32_t U, I, P, Q, F;
    int32_t has_uf, has_if;
    float alpha, beta, learning_rate;
    int32_t schedule;                /* 0 = 'constant', 1 = 'invscaling' (_rankfm.pyx:220-225) */
    float learning_exponent;
    int32_t max_samples;             /* 1 for BPR (rankfm.py:294-295) */
    int32_t epochs, epoch_begin;     /* eta / counter keys use the absolute epoch = epoch_begin + e */
    int32_t rng_mode;                /* RFM_RNG_MT19937 / RFM_RNG_COUNTER */
    uint32_t seed;                   /* MT seed (reference: 1492) or counter seed */
    int32_t membership;              /* 0 linear (lsearch), 1 binary */
    /* analysis only (NOT the reference's algorithm; 0 / 0 = the reference): the dense feature tables are updated on every
     * `table_every`-th visited row only (never when it is negative), with their step scaled by `table_step` -- a sequential stand-in for the engine's table
     * trainer, which trains the tables on a sample of the rows while every row reads them (rfm_sgd.hpp, sgd_features_kernel) */
    int32_t table_every;
    float table_step;
    /* analysis only: the first `table_head_rows` visited rows of the FIRST epoch of the call use `table_head_every` instead of
     * `table_every` (0 rows = no head) -- a sequential stand-in for a slow, table-friendly opening of the first epoch */
    int32_t table_head_every;
    int64_t table_head_rows;
    /* analysis only: after every epoch, `table_tail` further visits of randomly chosen rows (keyed by epoch and count) that update the
     * dense feature tables ONLY -- biases and factor rows frozen -- a sequential stand-in for a table trainer that goes on after the row
     * loops of its launch are done (tools/table_quota_standin.py).  0 = none (the reference). */
    int64_t table_tail;
    /* analysis only: the last `table_quiet_rows` visited rows of every epoch do not train the tables (they still read them) -- a
     * sequential stand-in for a table trainer that finishes its quota before the row loops are done.  0 = the reference. */
    int64_t table_quiet_rows;
    /* analysis only: the visits that train the tables are scored, for the TABLES' update, against a snapshot of the tables taken every
     * `table_batch` such visits (the rows' own update keeps the current tables) -- a sequential stand-in for the engine's trainer,
     * which applies batches of 64 staged steps that were all scored on the tables as published when the batch was produced.
     * 0 / 1 = every visit scored on the current tables (the reference). */
    int32_t table_batch;
} rfm_oracle_params;

/* return codes */
#define RFM_ORACLE_OK 0
#define RFM_ORACLE_BAD_ARG -1
#define RFM_ORACLE_NONFINITE_BASE 100   /* + index of the first offending array, order of assert_finite */

/* bit-level classification: this file is compiled with -ffast-math, under which isfinite()/isnan()
 * fold to constants */
static int f32_is_finite(float x) { uint32_t b; memcpy(&b, &x, 4); return (b & 0x7f800000U) != 0x7f800000U; }
static int f32_is_nan(float x) { uint32_t b; memcpy(&b, &x, 4); return (b & 0x7fffffffU) > 0x7f800000U; }

static int all_finite_sum(const float *x, size_t n) {
    /* _rankfm.pyx:95-103 tests np.isfinite(np.sum(x)); element-wise finiteness is equivalent
     * unless finite values overflow the sum, which the engine treats the same way */
    for (size_t k = 0; k < n; ++k)
        if (!f32_is_finite(x[k])) return 0;
    return 1;
}

/* One call = the reference's `_fit` body for `epochs` epochs.
 *   perms      int32 [epochs, N] explicit shuffled order per epoch, or NULL (counter permutation)
 *   ll_out     double [epochs]   raw (un-penalised) log-likelihood as the fp32 accumulator ends up
 *   neg_out    int32 [epochs, N] chosen negative per visited position (optional, may be NULL)
 *   nsamp_out  int32 [epochs, N] `sampled` per visited position (optional, may be NULL)
 */
static int fit_impl(const rfm_oracle_params *p,
                    const int32_t *interactions, const float *sample_weight,
                    const int64_t *csr_off, const int32_t *csr_items,
                    const float *x_uf, const float *x_if,
                    float *w_i, float *w_if, float *v_u, float *v_i, float *v_uf, float *v_if,
                    const int32_t *perms, double *ll_out, int32_t *neg_out, int32_t *nsamp_out,
                    const float *pos_step, const float *user_step, const float *pos_step_bias, const float *neg_step, double *ll64_out) {
    if (!p || p->N < 0 || p->I < 2 || p->F < 1 || p->max_samples < 1) return RFM_ORACLE_BAD_ARG;
    if (p->rng_mode == RFM_RNG_MT19937 && !perms && p->N > 0) return RFM_ORACLE_BAD_ARG;
    const int64_t N = p->N;
    const int I = p->I, P = p->P, Q = p->Q, F = p->F;
    rfm_model m = { p->U, I, P, Q, F, p->has_uf, p->has_if, x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if };

    /* (analysis option table_batch: the tables as they were when the current batch of table-training visits began) */
    rfm_model msnap = m;
    float *snap = NULL;
    int64_t tab_visits = 0;
    if (p->table_batch > 1) {
        snap = (float *)malloc(sizeof(float) * ((size_t)P * F + (size_t)Q * F + (size_t)Q + 1));
        if (!snap) return RFM_ORACLE_BAD_ARG;
        msnap.v_uf = snap; msnap.v_if = snap + (size_t)P * F; msnap.w_if = snap + (size_t)P * F + (size_t)Q * F;
    }
    const float MARGIN = 1.0f;                                  /* :149 */
    const float d_reg_a = 2.0f * p->alpha, d_reg_b = 2.0f * p->beta;   /* :171-172 */
    mt_state mt;
    mt_seed(&mt, p->seed);                                      /* :182 (reference passes 1492) */
    const uint32_t perm_bits = rfm_perm_bits((uint32_t)N);

    for (int e = 0; e < p->epochs; ++e) {
        const int epoch = p->epoch_begin + e;
        float eta;
        if (p->schedule == 0) eta = p->learning_rate;                              /* :220-221 */
        else if (p->schedule == 1) eta = (float)(p->learning_rate / pow((double)(epoch + 1), (double)p->learning_exponent)); /* :222-223 */
        else { free(snap); return RFM_ORACLE_BAD_ARG; }                                            /* :224-225 ValueError */
        const uint32_t ekey = rfm_epoch_key(p->seed, (uint32_t)epoch);
        float log_likelihood = 0.0f;                                               /* :228 */
        double log_likelihood64 = 0.0;     /* the same sum without the float accumulator's rounding (ll64_out) */

        const int64_t n_visits = N + (p->table_tail > 0 ? p->table_tail : 0);      /* (analysis option: table-only visits behind the epoch) */
        for (int64_t r = 0; r < n_visits; ++r) {                                   /* :230 */
            const int tail = r >= N;
            const int64_t row = tail ? (int64_t)rfm_draw_to_item(rfm_mix32(ekey ^ (0x51ED270BU + (uint32_t)(r - N))), (uint32_t)N)
                              : perms ? perms[(size_t)e * N + r]
                                      : (int64_t)rfm_perm((uint32_t)r, (uint32_t)N, perm_bits, ekey);
            const int u = interactions[2 * row], i = interactions[2 * row + 1];    /* :233-235 */
            const float sw = sample_weight[row];                                   /* :236 */
            const int32_t *items_u = csr_items + csr_off[u];
            const int64_t n_u = csr_off[u + 1] - csr_off[u];
            const uint32_t rkey = tail ? rfm_row_key(ekey ^ 0x3C6EF372U, (uint32_t)r) : rfm_row_key(ekey, (uint32_t)row);   /* (tail visits draw negatives of their own) */
            uint32_t attempt = 0;

            const float ut_ui = utility(&m, u, i);                                 /* :239 */
            int min_index = -1, sampled = 0, j = 0;
            float min_pu = 1e6f, pu = 0.0f;                                        /* :244-245 */
            for (sampled = 1; sampled <= p->max_samples; ++sampled) {              /* :247 */
                for (;;) {                                                         /* :250-253 */
                    if (p->rng_mode == RFM_RNG_MT19937) j = (int)(mt_next(&mt) % (uint32_t)I);
                    else j = (int)rfm_draw_to_item(rfm_draw(rkey, attempt++), (uint32_t)I);
                    if (!(p->membership ? member_binary(j, items_u, n_u) : member_linear(j, items_u, n_u))) break;
                }
                pu = ut_ui - utility(&m, u, j);                                    /* :256-257 */
                if (pu < min_pu) { min_index = j; min_pu = pu; }                   /* :259-261 */
                if (pu < MARGIN) break;                                            /* :263-264 */
            }
            if (sampled > p->max_samples) sampled = p->max_samples;  /* Cython for-range leaves the last value */
            /* :267-268.  With a NaN utility no draw ever compares below min_pu and the reference indexes
             * w_i[-1] (boundscheck=False); keep the last draw instead so the non-finite weights are
             * reported by assert_finite rather than corrupting memory. */
            if (min_index >= 0) { j = min_index; pu = min_pu; }
            /* :269 -- C integer division inside the log (cdivision=True) */
            const float multiplier = (float)(log((double)((I - 1) / sampled)) / log((double)I));
            const double log_sig = log(1.0 / (1.0 + exp(-(double)pu)));
            if (!tail) {
                log_likelihood = (float)((double)log_likelihood + log_sig);        /* :270 */
                log_likelihood64 += log_sig;
            }
            const float d_outer = (float)(1.0 / (exp((double)pu) + 1.0));         /* :276 */
            if (neg_out && !tail) neg_out[(size_t)e * N + r] = j;
            if (nsamp_out && !tail) nsamp_out[(size_t)e * N + r] = sampled;

            /* (rfm_oracle_fit_ex: the engine's Hogwild step damping applied sequentially -- the positive item's step and
             *  the user's step are scaled, nothing else; both scales are 1 in rfm_oracle_fit) */
            const float eta_row = tail ? 0.0f : eta;                               /* (a tail visit leaves biases and factor rows alone) */
            const float eta_i = pos_step ? eta_row * pos_step[i] : eta_row, eta_u = user_step ? eta_row * user_step[u] : eta_row;
            const float eta_iw = pos_step_bias ? eta_row * pos_step_bias[i] : eta_i;   /* (the bias of the positive item may be damped on its own) */
            const float eta_j = neg_step ? eta_row * neg_step[j] : eta_row;            /* (the engine scales an item's step whichever side it is on) */
            w_i[i] += eta_iw * (sw * multiplier * (d_outer * 1.0f) - (d_reg_a * w_i[i]));  /* :279 */
            w_i[j] += eta_j * (sw * multiplier * (d_outer * -1.0f) - (d_reg_a * w_i[j]));   /* :280 */

            const float *xi = x_if + (size_t)i * Q, *xj = x_if + (size_t)j * Q, *xu = x_uf + (size_t)u * P;
            /* (analysis option; always 1 for the reference.  table_every < 0: the tables are frozen -- what the engine's row loop does
             *  when its table trainer is switched off, debug_flags bit 5) */
            const int tab_every = (e == 0 && r < p->table_head_rows) ? p->table_head_every : p->table_every;
            const int do_tab = tail ? 1 : (tab_every < 0 || r >= N - p->table_quiet_rows) ? 0 : (tab_every <= 1 || r % tab_every == 0);
            const float eta_t = p->table_step > 0.0f ? eta * p->table_step : eta;
            float d_outer_t = d_outer;                     /* what the TABLES' update is scored with (analysis option table_batch) */
            if (snap && do_tab) {
                if (tab_visits % p->table_batch == 0) {
                    memcpy(msnap.v_uf, v_uf, sizeof(float) * (size_t)P * F);
                    memcpy(msnap.v_if, v_if, sizeof(float) * (size_t)Q * F);
                    memcpy(msnap.w_if, w_if, sizeof(float) * (size_t)Q);
                }
                ++tab_visits;
                const float pu_t = utility(&msnap, u, i) - utility(&msnap, u, j);
                d_outer_t = (float)(1.0 / (exp((double)pu_t) + 1.0));
            }
            if (p->has_if && do_tab)                                               /* :283-286 */
                for (int q = 0; q < Q; ++q) {
                    const float d_w_if = xi[q] - xj[q];
                    w_if[q] += eta_t * (sw * multiplier * (d_outer_t * d_w_if) - (d_reg_b * w_if[q]));
                }

            float *vu = v_u + (size_t)u * F, *vi = v_i + (size_t)i * F, *vj = v_i + (size_t)j * F;
            for (int f = 0; f < F; ++f) {                                          /* :289 */
                float d_v_u = vi[f] - vj[f];                                       /* :292 */
                float d_v_i = vu[f], d_v_j = -vu[f];                               /* :293-294 */
                if (p->has_uf)                                                     /* :297-300 */
                    for (int pp = 0; pp < P; ++pp) {
                        d_v_i += v_uf[(size_t)pp * F + f] * xu[pp];
                        d_v_j -= v_uf[(size_t)pp * F + f] * xu[pp];
                    }
                if (p->has_if)                                                     /* :303-305 */
                    for (int q = 0; q < Q; ++q) d_v_u += v_if[(size_t)q * F + f] * (xi[q] - xj[q]);

                vu[f] += eta_u * (sw * multiplier * (d_outer * d_v_u) - (d_reg_a * vu[f])); /* :308 */
                vi[f] += eta_i * (sw * multiplier * (d_outer * d_v_i) - (d_reg_a * vi[f])); /* :309 */
                vj[f] += eta_j * (sw * multiplier * (d_outer * d_v_j) - (d_reg_a * vj[f])); /* :310 */

                if (p->has_uf && do_tab)                                           /* :313-318 (post-update v_i) */
                    for (int pp = 0; pp < P; ++pp) {
                        if (xu[pp] == 0.0f) continue;
                        const float d_v_uf = xu[pp] * (vi[f] - vj[f]);
                        float *t = v_uf + (size_t)pp * F + f;
                        *t += eta_t * (sw * multiplier * (d_outer_t * d_v_uf) - (d_reg_b * *t));
                    }
                if (p->has_if && do_tab)                                           /* :321-326 (post-update v_u) */
                    for (int q = 0; q < Q; ++q) {
                        if (xi[q] - xj[q] == 0.0f) continue;
                        const float d_v_if = (xi[q] - xj[q]) * vu[f];
                        float *t = v_if + (size_t)q * F + f;
                        *t += eta_t * (sw * multiplier * (d_outer_t * d_v_if) - (d_reg_b * *t));
                    }
            }
        }
        if (ll_out) ll_out[e] = (double)log_likelihood;
        if (ll64_out) ll64_out[e] = log_likelihood64;
        /* :329 assert_finite, same array order as :98-103 */
        if (!all_finite_sum(w_i, (size_t)I)) { free(snap); return RFM_ORACLE_NONFINITE_BASE + 0; }
        if (!all_finite_sum(w_if, (size_t)Q)) { free(snap); return RFM_ORACLE_NONFINITE_BASE + 1; }
        if (!all_finite_sum(v_u, (size_t)p->U * F)) { free(snap); return RFM_ORACLE_NONFINITE_BASE + 2; }
        if (!all_finite_sum(v_i, (size_t)I * F)) { free(snap); return RFM_ORACLE_NONFINITE_BASE + 3; }
        if (!all_finite_sum(v_uf, (size_t)P * F)) { free(snap); return RFM_ORACLE_NONFINITE_BASE + 4; }
        if (!all_finite_sum(v_if, (size_t)Q * F)) { free(snap); return RFM_ORACLE_NONFINITE_BASE + 5; }
    }
    { free(snap); return RFM_ORACLE_OK; }
}

int rfm_oracle_fit(const rfm_oracle_params *p,
                   const int32_t *interactions, const float *sample_weight,
                   const int64_t *csr_off, const int32_t *csr_items,
                   const float *x_uf, const float *x_if,
                   float *w_i, float *w_if, float *v_u, float *v_i, float *v_uf, float *v_if,
                   const int32_t *perms, double *ll_out, int32_t *neg_out, int32_t *nsamp_out) {
    return fit_impl(p, interactions, sample_weight, csr_off, csr_items, x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if,
                    perms, ll_out, neg_out, nsamp_out, NULL, NULL, NULL, NULL, NULL);
}

/* The extended entry point of the checker:
 *   pos_step, user_step (both or neither)  NOT the reference's algorithm any more: the same sequential loop with the engine's
 *              Hogwild step damping applied -- pos_step float [I] scales the POSITIVE item's step (bias and
 *              factor row; the engine's pos_scale), user_step float [U] the user's step (the engine's min(1, user_cap / degree)),
 *              pos_step_bias float [I] or NULL: a scale of its own for the positive item's BIAS step (NULL: pos_step's),
 *              neg_step float [I] or NULL: the scale of the NEGATIVE item's step (bias and factor row; NULL: 1) --
 *              so that a test can separate what the damping changes (this against rfm_oracle_fit: a deliberate, documented change
 *              of the optimiser) from what asynchronous execution changes (the engine against this);
 *   ll64_out   double [epochs] or NULL: the epoch's log-likelihood summed in DOUBLE.  The reference accumulates it in a float
 *              (`cdef float log_likelihood`, :228, :270): once the running sum passes 2^20 every term below 1/16 is rounded away
 *              (well-classified pairs contribute ~0.01), so at 5 M rows its printed value is ~0.5 % too small in magnitude and at
 *              50 M rows it is meaningless.  ll_out keeps the reference's arithmetic (the golden vectors pin it); statistical
 *              comparisons at scale should use this one. */
int rfm_oracle_fit_ex(const rfm_oracle_params *p,
                      const int32_t *interactions, const float *sample_weight,
                      const int64_t *csr_off, const int32_t *csr_items,
                      const float *x_uf, const float *x_if,
                      float *w_i, float *w_if, float *v_u, float *v_i, float *v_uf, float *v_if,
                      const int32_t *perms, double *ll_out, int32_t *neg_out, int32_t *nsamp_out,
                      const float *pos_step, const float *user_step, const float *pos_step_bias, const float *neg_step, double *ll64_out) {
    return fit_impl(p, interactions, sample_weight, csr_off, csr_items, x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if,
                    perms, ll_out, neg_out, nsamp_out, pos_step, user_step, pos_step_bias, neg_step, ll64_out);
}

/* _rankfm.pyx:106-116  (double accumulation like numpy's float64 `penalty`) */
double rfm_oracle_reg_penalty(float alpha, float beta, int32_t U, int32_t I, int32_t P, int32_t Q, int32_t F,
                              const float *w_i, const float *w_if, const float *v_u, const float *v_i,
                              const float *v_uf, const float *v_if) {
    double pa = 0.0, pb = 0.0;
    for (size_t k = 0; k < (size_t)I; ++k) pa += (double)w_i[k] * w_i[k];
    for (size_t k = 0; k < (size_t)U * F; ++k) pa += (double)v_u[k] * v_u[k];
    for (size_t k = 0; k < (size_t)I * F; ++k) pa += (double)v_i[k] * v_i[k];
    for (size_t k = 0; k < (size_t)Q; ++k) pb += (double)w_if[k] * w_if[k];
    for (size_t k = 0; k < (size_t)P * F; ++k) pb += (double)v_uf[k] * v_uf[k];
    for (size_t k = 0; k < (size_t)Q * F; ++k) pb += (double)v_if[k] * v_if[k];
    return (double)alpha * pa + (double)beta * pb;
}

/* _rankfm.pyx:345-390: pairs are float32 indexes with NaN for unknown ids */
void rfm_oracle_predict(int64_t n_pairs, const float *pairs, int32_t U, int32_t I, int32_t P, int32_t Q, int32_t F,
                        int32_t has_uf, int32_t has_if, const float *x_uf, const float *x_if,
                        const float *w_i, const float *w_if, const float *v_u, const float *v_i,
                        const float *v_uf, const float *v_if, float *scores) {
    rfm_model m = { U, I, P, Q, F, has_uf, has_if, x_uf, x_if, (float *)w_i, (float *)w_if,
                    (float *)v_u, (float *)v_i, (float *)v_uf, (float *)v_if };
    for (int64_t r = 0; r < n_pairs; ++r) {
        const float uf = pairs[2 * r], itf = pairs[2 * r + 1];
        scores[r] = (f32_is_nan(uf) || f32_is_nan(itf)) ? nanf("") : utility(&m, (int)uf, (int)itf);   /* :380-388 */
    }
}

/* all-item scores of one user (the inner loop of _rankfm.pyx:440-441); ranking is done by the caller */
void rfm_oracle_user_scores(int32_t u, int32_t U, int32_t I, int32_t P, int32_t Q, int32_t F,
                            int32_t has_uf, int32_t has_if, const float *x_uf, const float *x_if,
                            const float *w_i, const float *w_if, const float *v_u, const float *v_i,
                            const float *v_uf, const float *v_if, float *scores) {
    rfm_model m = { U, I, P, Q, F, has_uf, has_if, x_uf, x_if, (float *)w_i, (float *)w_if,
                    (float *)v_u, (float *)v_i, (float *)v_uf, (float *)v_if };
    for (int i = 0; i < I; ++i) scores[i] = utility(&m, u, i);
}
