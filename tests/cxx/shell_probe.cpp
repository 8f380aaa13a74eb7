// Test helper: prints what AbcSmc::parse_config understood from a configuration file, one `key value...` line per
// item, and (with --sample N) the first N rows sample_priors draws for set 0.  No GPU work.
#include "../../abcsmc_amd/cxx/AbcSmcHip.hpp"

// --gauss s1 s2 s3 sigma n: n draws of the facade's ran_gaussian from the given taus2 state, as hex floats, then the state
static int gauss_mode(char** a) {
    ABC::RNG r(1);
    r.state.s1 = (uint32_t)strtoul(a[0], nullptr, 10);
    r.state.s2 = (uint32_t)strtoul(a[1], nullptr, 10);
    r.state.s3 = (uint32_t)strtoul(a[2], nullptr, 10);
    const double sigma = atof(a[3]);
    const int n = atoi(a[4]);
    for (int i = 0; i < n; i++) printf("%a\n", ABC::ran_gaussian(&r, sigma));
    printf("state %u %u %u\n", r.state.s1, r.state.s2, r.state.s3);
    return 0;
}

// --filter-report cfg K: AbcLog::filtering_report (AbcLog.cpp:79-123) on K hand-made posterior rows, to stdout
static int report_mode(const char* cfg, int K) {
    AbcSmc abc;
    abc.parse_config(cfg);
    Mat2D ppars((size_t)K, abc.npar()), pmets((size_t)K, abc.nmet());
    for (int i = 0; i < K; i++) {
        for (size_t j = 0; j < abc.npar(); j++) ppars(i, j) = (double)((i * 37 + 11 * (int)j) % 101) + 0.25 * (double)j;
        for (size_t j = 0; j < abc.nmet(); j++) pmets(i, j) = 40.0 + (double)((i * 13 + 7 * (int)j) % 17) * (j ? 0.125 : 1.0) - 3.0 * (double)j * 10.0;
    }
    AbcLog::filtering_report(&abc, 3, ppars, pmets, std::cout);
    return 0;
}

// --describe cfg: everything parse_config took from the file, one line per item (tests/test_shell.py feeds it the content of the
// reference's own examples/reference.json merged with examples/shared/partial.json)
static int describe_mode(const char* cfg) {
    AbcSmc abc;
    abc.parse_config(cfg);
    std::cout << "iterations " << abc.get_smc_iterations() << "\n";
    std::cout << "noise " << (abc.noise_type() == ABC::NOISE::MULTIVARIATE ? "MULTIVARIATE" : "INDEPENDENT") << "\n";
    std::cout << "filtering " << (abc.filtering_type() == ABC::FILTER::PLS ? "PLS" : "SIMPLE") << "\n";
    std::cout << "component_rule " << (abc.component_rule() == ABC_RULE_WILCOXON ? "wilcoxon" : "min_press") << "\n";
    std::cout << "set_sizes";
    for (size_t t = 0; t < abc.get_smc_iterations(); t++) std::cout << " " << abc.get_smc_size_at(t);
    std::cout << "\npred_prior_sizes";
    for (size_t t = 0; t < abc.get_smc_iterations(); t++) std::cout << " " << abc.get_pred_prior_size_at(t);
    std::cout << "\n";
    for (const ABC::Parameter* p : abc.parameters()) {
        const abc_prior pod = p->pod();
        std::cout << "parameter " << p->get_short_name() << " kind " << pod.kind << " a " << pod.a << " b " << pod.b << " name " << p->get_name() << "\n";
    }
    for (size_t j = 0; j < abc.metrics().size(); j++)
        std::cout << "metric " << abc.metrics()[j]->get_short_name() << " obs " << std::setprecision(17) << abc.observed_metrics()[j] << "\n";
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 3 && std::string(argv[1]) == "--describe") return describe_mode(argv[2]);
    if (argc >= 7 && std::string(argv[1]) == "--gauss") return gauss_mode(argv + 2);
    if (argc >= 4 && std::string(argv[1]) == "--filter-report") return report_mode(argv[2], atoi(argv[3]));
    if (argc < 2) return 2;
    AbcSmc abc;
    abc.parse_config(argv[1]);
    std::cout << "iterations " << abc.get_smc_iterations() << "\n";
    std::cout << "npar " << abc.npar() << " nmet " << abc.nmet() << "\n";
    std::cout << "set_sizes";
    for (size_t t = 0; t < abc.get_smc_iterations(); t++) std::cout << " " << abc.get_smc_size_at(t);
    std::cout << "\n";
    if (abc.get_smc_iterations() > 1 || argc > 2) {
        std::cout << "pred_prior_sizes";
        for (size_t t = 0; t < abc.get_smc_iterations(); t++) {
            try { std::cout << " " << abc.get_pred_prior_size_at(t); } catch (const std::exception&) { std::cout << " -"; }
        }
        std::cout << "\n";
    }
    return 0;
}
