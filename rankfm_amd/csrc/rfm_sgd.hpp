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
//
// The code lives in: rfm_sgd_common.hpp (arguments, helpers, segment tickets, hot-row sweeps), rfm_rowstep.hpp (RowStep),
// rfm_sgd_segments.hpp (sgd_rows_kernel, sgd_segments_kernel), rfm_sgd_warp.hpp (sgd_warp_kernel), rfm_sgd_features.hpp (feature models).
#pragma once
#include "rfm_sgd_common.hpp"
#include "rfm_rowstep.hpp"
#include "rfm_sgd_segments.hpp"
#include "rfm_sgd_warp.hpp"
#include "rfm_sgd_features.hpp"
