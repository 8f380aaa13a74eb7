// abc_dice -- the reference's dice-game fit (examples/integral/main.cpp + examples/include/{dice,examples}.h) as a
// driver over AbcSmcHip.hpp: same command line, same configuration file, the ranking / weighting / resampling of
// every --process step on the MI355X.
//
//   g++ -std=c++17 -O2 examples/abc_dice.cpp -o abc_dice -Labcsmc_amd -labcsmc_hip -ldl -Wl,-rpath,$PWD/abcsmc_amd
//   ./abc_dice config.json --process                      build the database / rank the finished set / propose the next
//   ./abc_dice config.json --simulate -n 500              run up to 500 queued particles
//   ./abc_dice config.json --process --simulate --all     every set in turn, then the final posterior
#include <unistd.h>

#include <cstring>
#include <ctime>

#include "../abcsmc_amd/cxx/AbcSmcHip.hpp"

static const ABC::RNG* RNG = new ABC::RNG();

// sum and standard deviation of `ndice` rolls of a `sides`-sided die
static std::vector<double> simulator(std::vector<double> parameters, const unsigned long int rng_seed, const unsigned long int /* serial */) {
    ABC::RNG dice(rng_seed);
    const size_t ndice = (size_t)parameters[0], sides = (size_t)parameters[1];
    double sum = 0, sumsq = 0;
    std::vector<double> rolls(ndice);
    for (size_t i = 0; i < ndice; i++) { rolls[i] = (double)(ABC::rng_uniform_int(&dice, sides) + 1); sum += rolls[i]; }
    const double mean = ndice ? sum / (double)ndice : 0.0;
    for (double r : rolls) sumsq += (r - mean) * (r - mean);
    return {sum, ndice > 1 ? std::sqrt(sumsq / (double)(ndice - 1)) : 0.0};
}

static void usage() {
    std::cerr << "\n\tUsage: ./abc_dice config.json --process\n\n"
              << "\t       ./abc_dice config.json --simulate [-n <simulations per database write>]\n\n"
              << "\t       ./abc_dice config.json --process --simulate [-n <...>] [--all] [--seed <s>] [--configured-simulator] [--devices 0,1,...] [--reference-stream] [--component-rule wilcoxon|press]\n\n";
}

int main(int argc, char* argv[]) {
    if (argc < 3) { usage(); return 100; }
    bool process_db = false, simulate_db = false, do_all = false, seeded = false, configured = false;
    unsigned long seed = 0;
    int buffer_size = 1;
    std::vector<int> devices;
    bool reference_stream = false;
    int component_rule = -1;                 // -1: the configuration's / the facade's default (ABC_RULE_WILCOXON)
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--process")) process_db = true;
        else if (!strcmp(argv[i], "--simulate")) simulate_db = true;
        else if (!strcmp(argv[i], "--all")) do_all = true;
        else if (!strcmp(argv[i], "--configured-simulator")) configured = true;   // use the "shared" / "executable" of the configuration
        else if (!strcmp(argv[i], "-n") && i + 1 < argc) buffer_size = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--reference-stream")) reference_stream = true;   // proposals from the reference's own taus2 stream
        else if (!strcmp(argv[i], "--component-rule") && i + 1 < argc) {       // PLS component rule (ABC::set_component_rule)
            const char* r = argv[++i];
            if (!strcmp(r, "wilcoxon")) component_rule = ABC_RULE_WILCOXON;
            else if (!strcmp(r, "press") || !strcmp(r, "min_press")) component_rule = ABC_RULE_MIN_PRESS;
            else { usage(); return 101; }
        }
        else if (!strcmp(argv[i], "--devices") && i + 1 < argc) {              // e.g. --devices 0,1,2,3: rows sharded over these GPUs
            for (char* tok = strtok(argv[++i], ","); tok; tok = strtok(nullptr, ",")) devices.push_back(atoi(tok));
        }
        else if (!strcmp(argv[i], "--seed") && i + 1 < argc) { seed = strtoul(argv[++i], nullptr, 10); seeded = true; }   // reproducible runs (tests)
        else { usage(); return 101; }
    }
    AbcSmc* abc = new AbcSmc();
    abc->parse_config(argv[1]);
    if (component_rule >= 0) abc->set_component_rule(component_rule);          // (the command line overrides the configuration; per object)
    try {
        if (!devices.empty()) ABC::use_devices(devices);
        if (reference_stream && process_db) ABC::set_reference_stream(true);
    } catch (const ABC::HipError& e) {
        std::cerr << "abc_dice: " << e.what() << std::endl;
        return 3;
    }
    if (!configured) abc->set_simulator(simulator);     // examples/integral: compiled in; examples/shared, executable: from the file
    try {
        auto turn = [&](int n, size_t step) {
            if (process_db) {
                // the reference reseeds from the clock and the pid before every --process (examples.h:61)
                ABC::rng_set(RNG, seeded ? seed + step : (unsigned long)time(NULL) * (unsigned long)getpid());
                abc->process_database(RNG);
            }
            if (simulate_db) abc->simulate_next_particles(n);
        };
        if (do_all && process_db && simulate_db) {
            // the whole fit: AbcSmc::run() = every set in turn, then the final posterior
            if (seeded) abc->run(RNG, [&](size_t step) { return seed + step; });
            else abc->run(RNG);
        } else if (do_all) {
            const size_t sets = abc->get_smc_iterations();
            for (size_t t = 0; t < sets; t++) turn((int)abc->get_smc_size_at(t), t);
            if (process_db) {
                ABC::rng_set(RNG, seeded ? seed + sets : (unsigned long)time(NULL) * (unsigned long)getpid());
                abc->process_database(RNG);      // once more, for the final posterior
            }
        } else {
            turn(buffer_size, 0);
        }
    } catch (const ABC::HipError& e) {
        std::cerr << "abc_dice: HIP path failed (" << e.code << "): " << e.what() << std::endl;
        return 3;
    }
    return 0;
}
