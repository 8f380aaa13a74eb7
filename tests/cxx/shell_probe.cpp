// Test helper: prints what AbcSmc::parse_config understood from a configuration file, one `key value...` line per
// item, and (with --sample N) the first N rows sample_priors draws for set 0.  No GPU work.
#include "../../abcsmc_amd/cxx/AbcSmcHip.hpp"

int main(int argc, char** argv) {
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
