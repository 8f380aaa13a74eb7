// One SMC set turn-over written against the reference's own call sequence (AbcSmc.cpp:634-664, 1041-1066,
// 490-518) but through AbcUtilHip.hpp.   g++ -std=c++17 facade_demo.cpp -L.. -labcsmc_hip -Wl,-rpath,..
#include <cstdio>

#include "AbcUtilHip.hpp"

int main() {
    using namespace ABC;
    const size_t N = 2000, M = 6, P = 3, K = 200;
    Mat2D X(N, M), Y(N, P);
    RNG gen(42);
    auto u = [&] { return rng_get(&gen) / 4294967296.0; };
    for (size_t i = 0; i < N; i++) {
        for (size_t p = 0; p < P; p++) Y(i, p) = u() * 10.0;
        for (size_t m = 0; m < M; m++) X(i, m) = Y(i, m % P) * (1.0 + m) + u();
    }
    Row obs(M);
    for (size_t m = 0; m < M; m++) obs[m] = 5.0 * (1.0 + m) + 0.5;
    ContinuousUniformPrior p0(0, 10), p1(0, 10);
    GaussianPrior p2(5, 10);
    std::vector<const Parameter*> pars = {&p0, &p1, &p2};
    try {
        std::vector<size_t> rank = particle_ranking_PLS(X, Y, obs, 0.5);      // AbcSmc.cpp:635-637
        rank.resize(K);                                                       // :645-646
        Mat2D post(K, P);
        for (size_t i = 0; i < K; i++) for (size_t p = 0; p < P; p++) post(i, p) = Y(rank[i], p);
        Row dv = calculate_doubled_variance(post);                            // :1043-1047
        Row w = weight_predictive_prior(pars, post);                          // set 0, :1049-1054
        Mat2D L = setup_mvn_sampler(post);                                    // :492-494
        RNG rng(1234);
        Mat2D next = sample_mvn_predictive_priors(&rng, N, w, post, pars, L); // :495-501
        Row w1 = weight_predictive_prior(pars, next /* pretend posterior of set 1 */, post, w, dv);
        double mean0 = 0;
        for (size_t i = 0; i < K; i++) mean0 += post(i, 0) / K;
        std::printf("facade ok: best particle %zu, posterior mean[0]=%.3f, dv[0]=%.3f, next(0,0)=%.3f, w1[0]=%.3e\n",
                    rank[0], mean0, dv[0], next(0, 0), w1[0]);
        if (!(mean0 > 3.5 && mean0 < 6.5)) return 2;   // observed metrics correspond to parameters near 5
        // the component rule of particle_ranking_PLS (AbcUtil.cpp:447-449): default = upstream's Wilcoxon reduction, switchable
        const int dflt = component_rule();
        set_component_rule(ABC_RULE_MIN_PRESS);
        const std::vector<size_t> r_press = particle_ranking_PLS(X, Y, obs, 0.5);
        set_component_rule(ABC_RULE_WILCOXON);
        const std::vector<size_t> r_wx = particle_ranking_PLS(X, Y, obs, 0.5);
        bool rejected = false;
        try { set_component_rule(7); } catch (const HipError&) { rejected = true; }
        std::printf("rules: default %d invalid_rejected %d same_as_default %d\n", dflt, (int)rejected, (int)(r_wx == std::vector<size_t>(particle_ranking_PLS(X, Y, obs, 0.5))));
        std::printf("press:");
        for (size_t i = 0; i < 12; i++) std::printf(" %zu", r_press[i]);
        std::printf("\nwilcoxon:");
        for (size_t i = 0; i < 12; i++) std::printf(" %zu", r_wx[i]);
        std::printf("\n");
        if (dflt != ABC_RULE_WILCOXON || !rejected) return 3;
        // AbcUtil.h:80-91: the per-row samplers on the reference's own stream (one uniform of the resampling draw in front, as
        // sample_predictive_priors / sample_mvn_predictive_priors consume it for a one-row posterior)
        DiscreteUniformPrior q0("a", "a", 1, 1000);
        ContinuousUniformPrior q1("b", "b", -2.0, 3.0);
        GaussianPrior q2("c", "c", 5.0, 10.0);
        std::vector<const Parameter*> qp = {&q0, &q1, &q2};
        const Row mu = {500.2, 2.9, 4.0}, s2 = {2500.0, 4.0, 1.5};
        RNG r1(777);
        (void)rng_get(&r1);
        const Row tn = gsl_ran_trunc_normal(&r1, qp, mu, s2);
        std::printf("trunc_normal: %.17g %.17g %.17g state %lu\n", tn[0], tn[1], tn[2], rng_get(&r1));
        Mat2D Lq(3, 3);
        Lq(0, 0) = 40.0; Lq(1, 0) = 0.7; Lq(1, 1) = 1.9; Lq(2, 0) = -0.3; Lq(2, 1) = 0.4; Lq(2, 2) = 1.1;
        RNG r2(778);
        (void)rng_get(&r2);
        const Row tm = gsl_ran_trunc_mv_normal(&r2, qp, mu, Lq);
        std::printf("trunc_mv_normal: %.17g %.17g %.17g state %lu\n", tm[0], tm[1], tm[2], rng_get(&r2));
    } catch (const HipError& e) {
        std::printf("HipError %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
