// Reference-stream proposals (host): the perturbation of ABC::sample_mvn_predictive_priors / sample_predictive_priors
// consuming the shared taus2 stream EXACTLY as the reference does -- sequentially, with a data-dependent number of draws:
//   /root/reference/src/AbcUtil.cpp:122-143   gsl_ran_trunc_mv_normal: z_i = gsl_ran_ugaussian in order (polar Box-Muller on
//                                             gsl_rng_uniform_pos, second variate discarded), x = mu + L z (dtrmv, lower),
//                                             recast / valid coordinate by coordinate, short-circuit, whole vector redrawn
//   /root/reference/src/AbcUtil.cpp:145-158,  gsl_ran_trunc_normal / Prior::noise: per coordinate recast(gaussian(sigma) + mu),
//     include/AbcSmc/Priors.h:19-43           up to 1000 tries, then the prior mean
//   /root/reference/src/AbcSmc.cpp:535        one gsl_rng_get per new particle AFTER all proposals (the simulator seeds)
// Selected by abc_ctx_set_noise_mode(ctx, ABC_NOISE_REFERENCE_STREAM); the default is the counter-based device stream, which
// is distributed identically but cannot reproduce the reference's numbers.  Inherently serial (every draw depends on how
// many the previous rows consumed): meant for the sizes at which bit-for-bit comparison with a CPU run of the reference is
// wanted (<= 1e5 rows; ~50 ns per normal), not for the 1e6-row generations.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/abcsmc_hip.h"

// theta: K x P column-major posterior, parent[n]: rows drawn by the alias sampler, out: n x P column-major (ld = n).
// rng: state after the n resampling draws; advanced past everything consumed.  Returns the number of rows given up on
// (MVN: max_tries whole-vector rejections -> the parent itself; independent: coordinates that fell back to the prior mean).
size_t abc_ref_perturb_mvn(abc_rng* rng, size_t n, size_t K, size_t P, const double* theta, const uint64_t* parent,
                           const double* L /* P x P column-major, lower */, const abc_prior* priors, size_t max_tries,
                           double* out);
size_t abc_ref_perturb_indep(abc_rng* rng, size_t n, size_t K, size_t P, const double* theta, const uint64_t* parent,
                             const double* dv, const abc_prior* priors, double* out);
void abc_ref_seeds(abc_rng* rng, size_t n, uint64_t* seeds);
