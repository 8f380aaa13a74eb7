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
    } catch (const HipError& e) {
        std::printf("HipError %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
