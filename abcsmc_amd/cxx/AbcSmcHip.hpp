// AbcSmcHip.hpp -- the AbcSmc orchestrator over the MI355X hot path.
//
// Same public surface as the reference class (/root/reference/include/AbcSmc/AbcSmc.h:36-118): parse_config,
// build_database, process_database, simulate_next_particles (+ by serial / posterior index), set_simulator
// (function pointer or shared object), set_executable, the set-size getters and the manual add_next_* builders, so
// that examples/*/main.cpp-style drivers compile against it with `#include "AbcSmcHip.hpp"` and
// `const ABC::RNG* RNG = new ABC::RNG()` in place of the GSL generator.  The storage schema, the Q/R/D job status
// machine, the 6-significant-digit text round trip of parameters and metrics, the `upar` table and the process exit
// codes follow src/AbcSmc.cpp (cited per function).  Everything numerical per generation (ranking, weights,
// covariance, resampling, perturbation) is the HIP library behind AbcUtilHip.hpp: there is no CPU fallback, and
// process_database on a machine without the GPU fails with ABC::HipError.
//
// Not carried over: the MPI scheduler/worker pair (AbcSmc.h:192-194, no MPI in this image) and AbcSimMPI.
#ifndef ABCSMC_AMD_ABCSMCHIP_HPP
#define ABCSMC_AMD_ABCSMCHIP_HPP

#include <dlfcn.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <map>
#include <optional>
#include <sstream>

#include "AbcUtilHip.hpp"
#include "MiniJson.hpp"
#include "SqliteDyn.hpp"

using ABC::Col;
using ABC::float_type;
using ABC::Mat2D;
using ABC::Row;

// ---- simulators (AbcSim.h:30-157) ---------------------------------------------------------------------------
struct AbcSimFun {
    virtual ~AbcSimFun() {}
    virtual std::vector<float_type> operator()(std::vector<float_type> pars, const unsigned long int seed,
                                               const unsigned long int serial) const = 0;
};
struct AbcSimUnset : AbcSimFun {     // AbcSim.h:45-52
    std::vector<float_type> operator()(std::vector<float_type>, const unsigned long int, const unsigned long int) const override {
        std::cerr << "ERROR: A pointer to a simulator function (prefered) or an external simulator executable must be defined." << std::endl;
        exit(100);
    }
};
typedef std::vector<float_type> AbcSimBase(std::vector<float_type>, const unsigned long int, const unsigned long int);
inline AbcSimBase* loadSO(const char* target) {      // AbcSim.h:62-76
    void* handle = dlopen(target, RTLD_LAZY);
    if (!handle) { std::cerr << "Failed to open simulator object: " << target << " ; " << dlerror() << std::endl; exit(101); }
    auto simf = (AbcSimBase*)dlsym(handle, "simulator");
    if (!simf) {
        std::cerr << "Failed to find 'simulator' function in " << target << " ; " << dlerror() << std::endl;
        dlclose(handle);
        exit(102);
    }
    return simf;
}
struct AbcFPtrBase : AbcSimFun {     // AbcSim.h:106-117
    AbcSimBase* fptr;
    AbcFPtrBase(AbcSimBase* f) : fptr(f) {}
    AbcFPtrBase(const char* target) : fptr(loadSO(target)) {}
    std::vector<float_type> operator()(std::vector<float_type> pars, const unsigned long int seed, const unsigned long int serial) const override {
        return fptr(pars, seed, serial);
    }
};
struct AbcExec : AbcSimFun {         // AbcSim.h:122-157: parameters as command-line arguments, metrics on stdout
    const std::string command;
    AbcExec(std::string c) : command(std::move(c)) {}
    // the command line: the configured command, then every parameter as the reference's ostream prints a double (6 significant digits)
    std::string command_line(const std::vector<float_type>& pars) const {
        std::ostringstream line;
        line << command;
        for (size_t i = 0; i < pars.size(); i++) line << ' ' << pars[i];
        return line.str();
    }
    // everything the child wrote to its standard output (RAII on the pipe: closed on every path)
    static bool slurp(const std::string& line, std::string* out) {
        struct Pipe {
            FILE* f;
            explicit Pipe(const std::string& l) : f(popen(l.c_str(), "r")) {}
            ~Pipe() { if (f) pclose(f); }
        } pipe(line);
        if (!pipe.f) return false;
        char chunk[4096];
        for (size_t got; (got = fread(chunk, 1, sizeof chunk, pipe.f)) > 0;) out->append(chunk, got);
        return true;
    }
    std::vector<float_type> operator()(std::vector<float_type> pars, const unsigned long int, const unsigned long int) const override {
        const std::string line = command_line(pars);
        std::string reply;
        if (!slurp(line, &reply)) { std::cerr << "ERROR: Unable to create pipe to " << line << std::endl; exit(103); }
        std::vector<float_type> mets;
        if (reply.empty() || reply == "ERROR") {                            // (the caller counts the metrics: AbcSmc.cpp:998, exit -211)
            std::cerr << command << " does not exist or appears to be an invalid simulator." << std::endl;
            std::cerr << "Attempted: " << line << std::endl;
            return mets;
        }
        std::istringstream numbers(reply);
        for (float_type m; numbers >> m;) mets.push_back(m);
        return mets;
    }
};

namespace ABC {
enum FILTER { PLS, SIMPLE };
enum NOISE { INDEPENDENT, MULTIVARIATE };

struct Metric {                       // Metric.h:8-35
    Metric(std::string s, std::string ss, double val, bool integral) : name(s), short_name(ss), obs_val(val), integral_(integral) {}
    std::string get_name() const { return name; }
    std::string get_short_name() const { return short_name == "" ? name : short_name; }
    bool is_integral() const { return integral_; }
    double get_obs_val() const { return obs_val; }

   private:
    std::string name, short_name;
    double obs_val;
    bool integral_;
};

// ParXform.h:12-68: fitting space -> model space
typedef float_type transformer(const float_type&);
struct ParXform {
    ParXform(transformer* u, const std::vector<size_t>& tadd = {}, const std::vector<size_t>& tmul = {},
             const std::vector<size_t>& uadd = {}, const std::vector<size_t>& umul = {})
        : u_(u), tadd_(tadd), tmul_(tmul), uadd_(uadd), umul_(umul) {}
    template <typename T> float_type transform(const float_type& pval, const T& fitting_space_values) const {
        float_type tplus = 0.0, ttimes = 1.0, uplus = 0.0, utimes = 1.0;
        for (size_t i : tadd_) tplus += fitting_space_values[i];
        for (size_t i : tmul_) ttimes *= fitting_space_values[i];
        for (size_t i : uadd_) uplus += fitting_space_values[i];
        for (size_t i : umul_) utimes *= fitting_space_values[i];
        return (u_((pval + tplus) * ttimes) + uplus) * utimes;
    }

   private:
    transformer* u_;
    std::vector<size_t> tadd_, tmul_, uadd_, umul_;
};
struct ParRescale {
    ParRescale(float_type p1 = 0.0, float_type p2 = 1.0) : par1(p1), par2(p2) {}
    const float_type par1, par2;
    float_type rescale(const float_type& pval) const { return (par2 - par1) * pval + par1; }
};
}  // namespace ABC

class AbcSmc;

// ---- stderr reports (AbcLog.h:20-42, AbcLog.cpp:6-123) -------------------------------------------------------
struct AbcLog {
    static constexpr int WIDTH = 12;
    static const char* double_bar() { return "========================================================================================="; }
    static void print_stats(const std::string& str1, const std::string& str2, double val1, double val2, double delta,
                            double pct_chg, const std::string& tail, std::ostream& os) {
        os << "    " + str1 + ", " + str2 + "  ( delta, % ): " << std::setw(WIDTH) << val1 << ", " << std::setw(WIDTH) << val2
           << " ( " << std::setw(WIDTH) << delta << ", " << std::setw(WIDTH) << pct_chg << "% )\n" + tail;
    }
    static void report_convergence_data(AbcSmc* abc, size_t set_t, std::ostream& os = std::cerr);
    static void filtering_report(AbcSmc* abc, size_t t, const Mat2D& posterior_pars, const Mat2D& posterior_mets,
                                 std::ostream& os = std::cerr);

   private:
    static void table_header(AbcSmc* abc, std::ostream& os);
};

class AbcSmc {
   public:
    AbcSmc() {}

    // ---- AbcSmc.h:41-107 ------------------------------------------------------------------------------------
    size_t get_smc_iterations() { return n_sets_; }
    size_t get_smc_size_at(const size_t set_num) {
        if (set_num >= n_sets_) throw std::out_of_range("set_num out of range");
        return (set_num < particles_per_set_.size()) ? particles_per_set_[set_num] : particles_per_set_.back();
    }
    size_t get_pred_prior_size_at(const size_t set_num) {
        if (set_num >= n_sets_ || kept_per_set_.empty()) throw std::out_of_range("set_num out of range");
        return (set_num < kept_per_set_.size()) ? kept_per_set_[set_num] : kept_per_set_.back();
    }
    void set_smc_iterations(const size_t n) { n_sets_ = n; }
    void set_smc_set_sizes(const std::vector<size_t>& v) { particles_per_set_ = v; }                 // manual configuration
    void set_predictive_prior_sizes(const std::vector<size_t>& v) { kept_per_set_ = v; }
    void set_pls_validation_training_fraction(const float_type f) {
        if (!((0 < f) && (f <= 1))) throw std::invalid_argument("pls training fraction must be in (0, 1]");
        train_frac_ = f;
    }
    void set_simulation(AbcSimFun* abcsf) { sim_ = abcsf; }
    void set_executable(std::string cmd) { set_simulation(new AbcExec(cmd)); }
    void set_simulator(AbcSimBase* simulator) { set_simulation(new AbcFPtrBase(simulator)); }
    void set_simulator(std::string soname) { set_simulation(new AbcFPtrBase(soname.c_str())); }
    void set_database_filename(std::string name) { db_path_ = name; }
    void set_retain_posterior_rank(const bool retain_rank) { keep_posterior_rank_ = retain_rank; }
    void set_filtering_type(const ABC::FILTER& ft) { ranking_kind_ = ft; }
    // the PLS component rule of THIS object ("pls_component_rule" / "pls_max_components" of its configuration; -1: whatever
    // ABC::set_component_rule / set_max_components say process-wide at the time of the ranking).  Kept per object: two AbcSmc
    // objects with different configurations do not change each other's rule
    void set_component_rule(int rule) {
        if (rule != ABC_RULE_MIN_PRESS && rule != ABC_RULE_WILCOXON) { std::cerr << "Unknown PLS component rule. Aborting." << std::endl; exit(-210); }
        component_rule_ = rule;
    }
    int component_rule() const { return component_rule_ < 0 ? ABC::component_rule() : component_rule_; }
    void set_max_components(int a) { max_components_ = a < 0 ? 0 : a; }
    int max_components() const { return max_components_ < 0 ? ABC::max_components_ref() : max_components_; }
    void set_noise_type(const ABC::NOISE& nt) { noise_kind_ = nt; }
    const ABC::Metric* add_next_metric(const ABC::Metric* m) {
        mets_.push_back(m);
        observed_.push_back(m->get_obs_val());
        return m;
    }
    const ABC::Parameter* add_next_parameter(const ABC::Parameter* p) { pars_.push_back(p); return p; }
    void add_modification_map(const ABC::Parameter* par, const ABC::ParXform* xform) { xform_of_par_[par] = xform; }
    void add_par_rescale(const ABC::Parameter* par, const ABC::ParRescale* r) { rescale_of_par_[par] = r; }

    bool parse_config(const std::string& conf_filename);
    bool build_database(const ABC::RNG* RNG);
    bool process_database(const ABC::RNG* RNG, const bool verbose = false);
    bool read_SMC_sets_from_database(sqdyn::Db& db, std::vector<std::vector<int>>& serials);
    bool simulate_next_particles(const int n = 1, const int serial_req = -1, const int posterior_req = -1);
    bool simulate_particle_by_serial(const int serial_req) { return simulate_next_particles(1, serial_req, -1); }
    bool simulate_particle_by_posterior_idx(const int posterior_req) { return simulate_next_particles(1, -1, posterior_req); }
    // AbcSmc::run() is named by the stale drivers (examples/direct/main.cpp:32, examples/scratch/main_mpi_*.cpp) and absent from
    // AbcSmc.h of this reference version; here it is the `--process --simulate --all` sequence of abc_loop
    // (examples/include/examples.h:57-93): for every set, reseed + process_database, then simulate that set's particles;
    // process_database once more for the final posterior.  reseed(step) gives the seed before the step-th --process
    // (default: time(NULL) * getpid(), examples.h:64).
    bool run(const ABC::RNG* RNG, const std::function<unsigned long(size_t)>& reseed = nullptr);
    bool run(const std::string& executable, const ABC::RNG* RNG) { set_executable(executable); return run(RNG); }

    size_t npar() { return pars_.size(); }
    size_t nmet() { return mets_.size(); }
    std::vector<Mat2D> get_particle_parameters() { return set_params_; }
    std::vector<Mat2D> get_particle_metrics() { return set_metrics_; }
    // extras for tests / drivers
    const std::vector<const ABC::Parameter*>& parameters() const { return pars_; }
    const std::vector<const ABC::Metric*>& metrics() const { return mets_; }
    const Row& observed_metrics() const { return observed_; }
    ABC::NOISE noise_type() const { return noise_kind_; }
    ABC::FILTER filtering_type() const { return ranking_kind_; }
    const std::vector<std::vector<size_t>>& get_predictive_priors() const { return kept_rows_; }
    const std::vector<Col>& get_weights() const { return set_weights_; }
    std::ostream* log_stream = &std::cerr;      // the reference writes its reports to std::cerr

   private:
    friend struct AbcLog;
    static constexpr const char* JOB_TABLE = "job";
    static constexpr const char* MET_TABLE = "met";
    static constexpr const char* PAR_TABLE = "par";
    static constexpr const char* UPAR_TABLE = "upar";

    std::vector<const ABC::Parameter*> pars_;
    std::map<const ABC::Parameter*, const ABC::ParXform*> xform_of_par_;
    std::map<const ABC::Parameter*, const ABC::ParRescale*> rescale_of_par_;
    Mat2D posterior_path_;
    AbcSimFun* sim_ = new AbcSimUnset();
    std::vector<const ABC::Metric*> mets_;
    Row observed_;
    bool keep_posterior_rank_ = false;
    ABC::FILTER ranking_kind_ = ABC::FILTER::PLS;
    int component_rule_ = -1, max_components_ = -1;      // -1: the process-wide setting (ABC::component_rule(), default Wilcoxon)
    ABC::NOISE noise_kind_ = ABC::NOISE::INDEPENDENT;
    size_t n_sets_ = 0;
    std::vector<size_t> particles_per_set_, kept_per_set_;
    float_type train_frac_ = 0.5;
    std::vector<std::vector<size_t>> kept_rows_;
    std::vector<Mat2D> set_metrics_, set_params_;
    std::vector<Col> set_weights_;
    std::vector<Row> set_dv_;
    std::string db_path_;
    std::optional<std::string> resume_dir_;

    Row _to_model_space(const Row& pars);
    bool run_sim_(Row& par, Row& met, const size_t rng_seed, const size_t serial);
    void calculate_predictive_prior_weights(const size_t set_num);
    std::string _column_list(const char* prefix, bool pars, const char* suffix);
    void _insert_particles(sqdyn::Db& db, size_t set_num, long long first_serial, const Mat2D& pars,
                           const std::vector<unsigned long>& seeds, const std::vector<long long>& posterior_ranks);
    template <typename F> bool _transaction(sqdyn::Db& db, const char* what, F&& body);
};

// =============================================================================================================
// configuration (AbcSmc.cpp:44-430)
// =============================================================================================================
namespace abcsmc_detail {
[[noreturn]] inline void die(int code, const std::string& msg) { std::cerr << msg << std::endl; exit(code); }

template <typename T> std::vector<T> as_vector(const mjson::Value& val) {       // scalar or array (AbcSmc.cpp:42-52)
    std::vector<T> out;
    if (val.isArray()) { for (const mjson::Value& jv : val) out.push_back(jv.as<T>()); }
    else out.push_back(val.as<T>());
    return out;
}
inline float_type identity_xf(const float_type& t) { return t; }
inline float_type pow10_xf(const float_type& t) { return std::pow(10.0, t); }
inline float_type logistic_xf(const float_type& t) { return ABC::logistic(t); }
inline std::string slurp(const std::string& fn) {
    std::ifstream in(fn);
    std::stringstream ss;
    ss << in.rdbuf();
    return ss.str();
}

// AbcSmc.cpp:54-134: set sizes, predictive prior sizes (as fractions or counts) and the iteration count
inline void parse_iterations(const mjson::Value& par, const size_t pseudosize, size_t* iterations, float_type* training_frac,
                             std::vector<size_t>* set_sizes, std::vector<size_t>* pred_prior_sizes) {
    if (pseudosize != 0) {   // projection mode: a single set enumerating the PSEUDO / POSTERIOR grid
        if (par.get("smc_iterations", 1).asInt() != 1) die(-202, "Cannot use smc_iterations > 1 with ONLY PSEUDO or POSTERIOR parameters.  Aborting.");
        if (par.isMember("num_samples")) {
            const size_t checksize = as_vector<size_t>(par["num_samples"])[0];
            if (checksize != pseudosize)
                die(-201, "ERROR: `num_samples` (" + std::to_string(checksize) + ") does not match imputed combinations of PSEUDO and/or POSTERIOR parameters (" +
                              std::to_string(pseudosize) + ").");
            std::cerr << "WARNING: specified `num_samples` for all PSEUDO and/or POSTERIOR parameters." << std::endl;
        }
        if (par.isMember("predictive_prior_fraction") || par.isMember("predictive_prior_size"))
            std::cerr << "WARNING: ignoring `predictive_prior_*` options in projection mode." << std::endl;
        *iterations = 1;
        *set_sizes = {pseudosize};
        return;
    }
    const bool has_frac = par.isMember("predictive_prior_fraction"), has_size = par.isMember("predictive_prior_size");
    if (has_frac == has_size)
        die(1, "Error: exactly one of `predictive_prior_fraction` or `predictive_prior_size` must be specified in configuration file.");
    *training_frac = par.get("pls_training_fraction", 0.5).asDouble();
    if (*training_frac <= 0 || 1 <= *training_frac) die(1, "Error: pls_training_fraction must be in (0, 1).");
    *set_sizes = as_vector<size_t>(par["num_samples"]);
    if (has_frac) {
        std::vector<float_type> ppfs = as_vector<float_type>(par["predictive_prior_fraction"]);
        for (float_type f : ppfs) if (!((0 < f) && (f <= 1))) die(1, "Error: `predictive_prior_fraction`s must be in (0, 1]");
        std::vector<size_t> sizes = *set_sizes;
        const size_t max_set = std::max(ppfs.size(), sizes.size());
        ppfs.resize(max_set, ppfs.back());
        sizes.resize(max_set, sizes.back());
        pred_prior_sizes->resize(max_set);
        for (size_t i = 0; i < max_set; i++) (*pred_prior_sizes)[i] = (size_t)std::round(ppfs[i] * sizes[i]);
    } else {
        *pred_prior_sizes = as_vector<size_t>(par["predictive_prior_size"]);
        const size_t np = pred_prior_sizes->size(), ns = set_sizes->size();
        for (size_t i = 0; i < std::max(np, ns); i++) {      // the shorter list continues with its last entry
            const size_t pp = (*pred_prior_sizes)[std::min(i, np - 1)], ss = (*set_sizes)[std::min(i, ns - 1)];
            if (pp > ss) die(1, "Error: requested predictive prior size > SMC set size at: " + std::to_string(i));
        }
    }
    *iterations = (size_t)par.get("smc_iterations", (double)std::max(set_sizes->size(), pred_prior_sizes->size())).asUInt64();
}

inline ABC::Metric* parse_metric(const mjson::Value& mmet) {                    // AbcSmc.cpp:136-152
    const std::string name = mmet["name"].asString();
    const std::string short_name = mmet.get("short_name", name).asString();
    const float_type val = mmet["value"].asDouble();
    const std::string ntype = mmet["num_type"].asString();
    if (ntype == "INT") return new ABC::Metric(name, short_name, val, true);
    if (ntype == "FLOAT") return new ABC::Metric(name, short_name, val, false);
    die(-209, "Unknown metric numeric type: " + ntype + ".  Aborting.");
}

inline ABC::Parameter* parse_parameter(const mjson::Value& mpar) {              // AbcSmc.cpp:212-274
    const std::string name = mpar["name"].asString();
    const std::string short_name = mpar.get("short_name", name).asString();
    const std::string ptype = mpar["dist_type"].asString();
    const std::string ntype = mpar["num_type"].asString();
    if (!(ntype == "INT" || ntype == "FLOAT")) die(-206, "Unknown parameter numeric type: " + ntype + ".  Aborting.");
    if (ptype == "UNIFORM") {
        if (ntype == "INT") return new ABC::DiscreteUniformPrior(name, short_name, (long)mpar["par1"].asInt64(), (long)mpar["par2"].asInt64());
        return new ABC::ContinuousUniformPrior(name, short_name, mpar["par1"].asDouble(), mpar["par2"].asDouble());
    }
    if (ptype == "NORMAL" || ptype == "GAUSSIAN") {
        if (ntype == "INT") die(-206, "Parameter numeric " + ntype + " not supported for parameter type " + ptype + ".  Aborting.");
        return new ABC::GaussianPrior(name, short_name, mpar["par1"].asDouble(), mpar["par2"].asDouble());
    }
    if (ptype == "PSEUDO") {
        std::vector<float_type> states;
        if (mpar.isMember("vals")) {
            states = as_vector<float_type>(mpar["vals"]);
        } else {
            const float_type smax = mpar["par2"].asDouble();
            const float_type step = mpar.get("step", 1.0).asDouble();
            if (step != 0) {
                const double EPSILON = 0.0001;
                for (float_type s = mpar["par1"].asDouble(); s <= smax + EPSILON * step; s += step) states.push_back(s);
            } else {
                states.push_back(mpar["par1"].asDouble());
            }
        }
        return new ABC::PseudoPar(name, short_name, states);
    }
    if (ptype == "POSTERIOR") {
        const size_t size = (size_t)(mpar["par2"].asUInt64() - mpar["par1"].asUInt64() + 1);
        return new ABC::PosteriorPar(name, short_name, size);
    }
    die(-205, "Unknown parameter distribution type: " + ptype + ".  Aborting.");
}

// AbcSmc.cpp:154-210: "untransform" as a string (NONE / POW_10 / LOGISTIC) or as a LOGISTIC object with bounds
// and lists of parameter names that enter as addends / factors before or after the back-transformation
inline void parse_transform(const mjson::Value& mparu, ABC::ParRescale** pscale, ABC::ParXform** pxform,
                            const std::map<std::string, size_t>& par_name_idx) {
    if (mparu.isString()) {
        const std::string ttype = mparu.asString();
        ABC::transformer* u = nullptr;
        if (ttype == "NONE") u = identity_xf;
        else if (ttype == "POW_10") u = pow10_xf;
        else if (ttype == "LOGISTIC") u = logistic_xf;
        else die(-206, "Unknown parameter transformation type: " + ttype + ".  Aborting.");
        *pscale = new ABC::ParRescale();
        *pxform = new ABC::ParXform(u);
    } else if (mparu.isObject()) {
        if (mparu["type"].asString() != "LOGISTIC")
            die(-207, "Only type: LOGISTIC is currently supported for untransformation objects.  (NONE and POW_10 supported as untransformation strings.)");
        *pscale = new ABC::ParRescale(mparu["min"].asDouble(), mparu["max"].asDouble());
        std::vector<size_t> idx[4];
        const char* keys[4] = {"transformed_addend", "transformed_factor", "untransformed_addend", "untransformed_factor"};
        for (int k = 0; k < 4; k++)
            if (mparu.isMember(keys[k]))
                for (const mjson::Value& jv : mparu[keys[k]]) idx[k].push_back(par_name_idx.at(jv.asString()));
        *pxform = new ABC::ParXform(logistic_xf, idx[0], idx[1], idx[2], idx[3]);
    } else {
        die(-208, "Unsupported JSON data type associated with 'untransform' parameter key.");
    }
}

// default ostream formatting (6 significant digits): what the reference's `ss << double` stores in the database
inline std::string num(double v) { std::ostringstream ss; ss << v; return ss.str(); }
}  // namespace abcsmc_detail

// AbcSmc.cpp:293-335: the posterior rows of an earlier fit (upar when present, else par), in job order
inline Mat2D abcsmc_slurp_posterior(const std::string& filename, const std::vector<const ABC::Parameter*>& pars) {
    sqdyn::Db post_db(filename);
    size_t posterior_size = 0;
    { sqdyn::Stmt q = post_db.query("select count(*) from job where posterior > -1;"); q.next(); posterior_size = (size_t)q.i64(0); }
    std::string cols;
    size_t ncol = 0;
    for (const ABC::Parameter* p : pars)
        if (p->isPosterior()) { cols += (ncol++ ? ", " : "") + p->get_short_name(); }
    Mat2D posterior(posterior_size, ncol);
    const std::string table = post_db.table_exists("upar") ? "upar" : "par";
    sqdyn::Stmt q = post_db.query("select " + cols + " from " + table + " P, job J where P.serial = J.serial and posterior > -1;");
    size_t r = 0;
    while (q.next() && r < posterior_size) {
        for (size_t c = 0; c < ncol; c++) posterior(r, c) = q.f64((int)c);
        r++;
    }
    return posterior;
}

inline bool AbcSmc::parse_config(const std::string& conf_filename) {          // AbcSmc.cpp:337-430
    using namespace abcsmc_detail;
    { std::ifstream probe(conf_filename); if (!probe.good()) die(1, "File does not exist: " + conf_filename); }
    mjson::Value par;
    try { par = mjson::parse(slurp(conf_filename)); }
    catch (const mjson::ParseError& e) { die(1, std::string("Failed to parse configuration\n") + e.what()); }

    set_retain_posterior_rank(par.get("retain_posterior_rank", false).asBool());

    const mjson::Value& model_par = par["parameters"];
    std::map<std::string, size_t> par_name_idx;
    for (size_t i = 0; i < model_par.size(); i++) {
        const std::string name = model_par[i]["name"].asString();
        if (par_name_idx.count(name)) die(-206, "Duplicate parameter name: " + name);
        par_name_idx.emplace(name, i);
    }
    bool any_posterior = false;
    size_t pseudosize = 1, posterior_size = 0;
    for (const mjson::Value& mpar : model_par) {
        ABC::Parameter* p = parse_parameter(mpar);
        if (p->isPosterior()) {
            if (posterior_size == 0) { posterior_size = p->state_size(); any_posterior = true; }
            else if (p->state_size() != posterior_size) die(-204, "POSTERIOR parameters must all span the same rank range.");
        } else {
            pseudosize *= p->state_size();      // 0 for a prior: any prior makes this a fit, not a projection
        }
        add_next_parameter(p);
        if (mpar.isMember("untransform")) {
            ABC::ParRescale* rescale = nullptr;
            ABC::ParXform* xform = nullptr;
            parse_transform(mpar["untransform"], &rescale, &xform, par_name_idx);
            add_modification_map(p, xform);
            add_par_rescale(p, rescale);
        }
    }
    if (any_posterior) {
        pseudosize *= posterior_size;
        if (!par.isMember("posterior_database_filename"))
            die(-204, "Parameter specfied as type POSTERIOR, without previously specifying a posterior_database_filename.  Aborting.");
        if (n_sets_ > 1) die(-203, "Cannot use posterior parameters with multiple SMC sets.  Aborting.");
        posterior_path_ = abcsmc_slurp_posterior(par["posterior_database_filename"].asString(), pars_);
    }
    for (const mjson::Value& mmet : par["metrics"]) add_next_metric(parse_metric(mmet));

    parse_iterations(par, pseudosize, &n_sets_, &train_frac_, &particles_per_set_, &kept_per_set_);

    const std::string executable = par.get("executable", "").asString();
    if (executable != "") set_executable(executable);
    const std::string sharedobj = par.get("shared", "").asString();
    if (sharedobj != "") set_simulator(sharedobj);
    const std::string resume_dir = par.get("resume_directory", "").asString();
    if (resume_dir != "") { std::cerr << "Resuming in directory: " << resume_dir << std::endl; resume_dir_.emplace(resume_dir); }
    set_database_filename(par["database_filename"].asString());

    const std::string noise = par.get("noise", "INDEPENDENT").asString();
    if (noise == "INDEPENDENT") noise_kind_ = ABC::NOISE::INDEPENDENT;
    else if (noise == "MULTIVARIATE") noise_kind_ = ABC::NOISE::MULTIVARIATE;
    else die(-210, "Unknown parameter noise type specified: " + noise + ". Aborting.");
    // not in the reference's JSON: lets a configuration ask for the simple (non-PLS) ranking that upstream only
    // reaches through set_filtering_type()
    const std::string filtering = par.get("filtering", "PLS").asString();
    if (filtering == "PLS") ranking_kind_ = ABC::FILTER::PLS;
    else if (filtering == "SIMPLE") ranking_kind_ = ABC::FILTER::SIMPLE;
    else die(-210, "Unknown filtering type specified: " + filtering + ". Aborting.");
    // not in the reference's JSON either: the PLS component rule (ABC::set_component_rule; default = upstream's Wilcoxon
    // reduction as SURVEY A.2 describes it) and a cap on the number of components
    if (par.isMember("pls_component_rule")) {
        const std::string rule = par["pls_component_rule"].asString();
        if (rule == "wilcoxon" || rule == "WILCOXON") component_rule_ = ABC_RULE_WILCOXON;
        else if (rule == "min_press" || rule == "MIN_PRESS" || rule == "press") component_rule_ = ABC_RULE_MIN_PRESS;
        else die(-210, "Unknown pls_component_rule specified: " + rule + ". Aborting.");
    }
    if (par.isMember("pls_max_components")) { max_components_ = par["pls_max_components"].asInt(); if (max_components_ < 0) max_components_ = 0; }
    return true;
}

inline Row AbcSmc::_to_model_space(const Row& fitting) {                       // AbcSmc.cpp:432-447
    Row model = fitting;
    for (size_t j = 0; j < fitting.size(); j++) {
        const ABC::Parameter* mpar = pars_[j];
        auto it = xform_of_par_.find(mpar);
        if (it != xform_of_par_.end()) model[j] = rescale_of_par_[mpar]->rescale(it->second->transform(fitting[j], fitting));
    }
    return model;
}

// =============================================================================================================
// storage (AbcSmc.cpp:688-873)
// =============================================================================================================
inline std::string AbcSmc::_column_list(const char* prefix, bool pars, const char* suffix) {
    std::string out;
    const size_t n = pars ? npar() : nmet();
    for (size_t i = 0; i < n; i++) {
        out += prefix + (pars ? pars_[i]->get_short_name() : mets_[i]->get_short_name()) + suffix;
        out += (i + 1 < n) ? ", " : " ";
    }
    return out;
}

template <typename F> bool AbcSmc::_transaction(sqdyn::Db& db, const char* what, F&& body) {
    try {
        db.begin_exclusive();
        body();
        db.commit();
        return true;
    } catch (const std::exception& e) {        // AbcSmc.cpp:728-742: report, roll back, carry on
        db.rollback();
        std::cerr << "CAUGHT e: " << e.what() << std::endl << "Failed while " << what << std::endl;
        return false;
    }
}

// one job / par / (upar) / met row per new particle: AbcSmc.cpp:522-552 and :846-871
inline void AbcSmc::_insert_particles(sqdyn::Db& db, size_t set_num, long long first_serial, const Mat2D& pars,
                                      const std::vector<unsigned long>& seeds, const std::vector<long long>& posterior_ranks) {
    using abcsmc_detail::num;
    const bool upar = !xform_of_par_.empty();
    std::string null_mets;
    for (size_t j = 0; j < nmet(); j++) null_mets += ", NULL";
    for (size_t i = 0; i < pars.rows(); i++) {
        const long long serial = first_serial + (long long)i;
        const long long rank = posterior_ranks.empty() ? -1 : posterior_ranks[i];
        std::ostringstream ss;
        ss << "insert into " << JOB_TABLE << " values ( " << serial << ", " << set_num << ", " << i << ", " << time(NULL)
           << ", NULL, 'Q', " << rank << ", 0 );";
        db.exec(ss.str());
        Row fitting(npar());
        std::string vals;
        for (size_t j = 0; j < npar(); j++) { fitting[j] = pars(i, j); vals += ", " + num(fitting[j]); }
        db.exec("insert into " + std::string(PAR_TABLE) + " values ( " + std::to_string(serial) + ", '" + std::to_string(seeds[i]) + "'" + vals + " );");
        if (upar) {
            const Row u = _to_model_space(fitting);
            std::string uvals;
            for (size_t j = 0; j < npar(); j++) uvals += ", " + num(u[j]);
            db.exec("insert into " + std::string(UPAR_TABLE) + " values ( " + std::to_string(serial) + ", '" + std::to_string(seeds[i]) + "'" + uvals + " );");
        }
        db.exec("insert into " + std::string(MET_TABLE) + " values ( " + std::to_string(serial) + null_mets + " );");
    }
}

inline bool AbcSmc::build_database(const ABC::RNG* RNG) {                      // AbcSmc.cpp:810-873
    sqdyn::Db db(db_path_);
    if (db.table_exists(JOB_TABLE) || db.table_exists(PAR_TABLE) || db.table_exists(MET_TABLE)) return false;
    // the tables first; the priors and seeds are drawn (and the RNG stream consumed) only once they exist
    const bool created = _transaction(db, "creating tables", [&] {
        db.exec(std::string("create table ") + JOB_TABLE + " ( serial int primary key asc, smcSet int, particleIdx int, startTime int, duration real, status text, posterior int, attempts int );");
        db.exec(std::string("create index idx1 on ") + JOB_TABLE + " (status, attempts);");
        db.exec(std::string("create table ") + PAR_TABLE + " ( serial int primary key, seed blob, " + _column_list("", true, " real") + ");");
        if (!xform_of_par_.empty())
            db.exec(std::string("create table ") + UPAR_TABLE + " ( serial int primary key, seed blob, " + _column_list("", true, " real") + ");");
        db.exec(std::string("create table ") + MET_TABLE + " ( serial int primary key, " + _column_list("", false, " real") + ");");
    });
    if (!created) { std::cerr << "ERROR: could not create the tables of " << db_path_ << std::endl; return false; }
    const size_t num_particles = get_smc_size_at(0);
    std::vector<size_t> posterior_ranks;
    const Mat2D pars = ABC::sample_priors(RNG, num_particles, posterior_path_, pars_, posterior_ranks);
    std::vector<unsigned long> seeds(num_particles);
    for (size_t i = 0; i < num_particles; i++) seeds[i] = ABC::rng_get(RNG);     // after all samples (:843, 859)
    std::vector<long long> ranks;
    if (keep_posterior_rank_) ranks.assign(posterior_ranks.begin(), posterior_ranks.end());
    if (!_transaction(db, "inserting the first set", [&] { _insert_particles(db, 0, 0, pars, seeds, ranks); })) {
        std::cerr << "ERROR: could not store the first set in " << db_path_ << std::endl;
        return false;
    }
    return true;
}

inline bool AbcSmc::read_SMC_sets_from_database(sqdyn::Db& db, std::vector<std::vector<int>>& serials) {   // AbcSmc.cpp:562-679
    for (const char* t : {JOB_TABLE, PAR_TABLE, MET_TABLE})
        if (!db.table_exists(t)) {
            std::cerr << "Table " << t << " does not exist in database.\n"
                      << "ERROR: Failed to read SMC set from database because one or more tables are missing.\n";
            return false;
        }
    struct SetRow { int t, size, done; };
    std::vector<SetRow> sets;
    {
        sqdyn::Stmt s = db.query(std::string("select smcSet, count(*), COUNT(case status when 'D' then 1 else null end) from ") + JOB_TABLE +
                                 " group by smcSet order by smcSet;");
        while (s.next()) sets.push_back({(int)s.i64(0), (int)s.i64(1), (int)s.i64(2)});
    }
    serials.clear();
    set_params_.clear();
    set_metrics_.clear();
    for (const SetRow& sr : sets) {
        const int t = sr.t;
        if (sr.size != sr.done) {
            std::cerr << "ERROR: Failed to read SMC set from database because not all particles are complete in set " << t << "\n";
            return false;
        }
        if ((size_t)t >= n_sets_ || sr.size != (int)get_smc_size_at(t)) {
            std::cerr << "ERROR:\tSet size for one or more sets does not agree between configuration file and database:" << std::endl
                      << "\tSet " << t << " in configuration file has size " << ((size_t)t < n_sets_ ? (long)get_smc_size_at(t) : -1L)
                      << " vs size " << sr.size << " in database." << std::endl
                      << "\tNB: You may want to edit the configuration file to have an array of set sizes that reflect what is already in the database." << std::endl
                      << "\t    Sizes of sets currently in database: [";
            for (size_t k = 0; k < sets.size(); k++) std::cerr << (k ? ", " : "") << sets[k].size;
            std::cerr << "]" << std::endl;
            exit(1);
        }
        const size_t n = (size_t)sr.size;
        set_params_.push_back(Mat2D(n, npar()));
        set_metrics_.push_back(Mat2D(n, nmet()));
        serials.push_back(std::vector<int>(n));
        std::vector<std::pair<int, int>> posterior_pairs;
        {
            sqdyn::Stmt s2 = db.query("select J.serial, J.particleIdx, J.posterior, " + _column_list("P.", true, "") + ", " + _column_list("M.", false, "") +
                                      "from " + JOB_TABLE + " J, " + MET_TABLE + " M, " + PAR_TABLE + " P where J.serial = M.serial and J.serial = P.serial " +
                                      "and J.smcSet = " + std::to_string(t) + " order by J.particleIdx;");
            size_t counter = 0;
            while (s2.next()) {
                const int serial = (int)s2.i64(0), particle_idx = (int)s2.i64(1), rank = (int)s2.i64(2);
                if (counter != (size_t)particle_idx || counter >= n) {
                    std::cerr << "ERROR: particle_counter != particle_idx (" << counter << " != " << particle_idx << ")\n";
                    return false;
                }
                serials[t][counter] = serial;
                if (rank > -1) posterior_pairs.push_back({rank, particle_idx});
                for (size_t j = 0; j < npar(); j++) set_params_[t](counter, j) = s2.f64(3 + (int)j);
                for (size_t j = 0; j < nmet(); j++) set_metrics_[t](counter, j) = s2.f64(3 + (int)npar() + (int)j);
                counter++;
            }
        }
        if (!posterior_pairs.empty()) {          // already filtered and ranked
            kept_rows_.push_back(std::vector<size_t>(posterior_pairs.size()));
            for (const auto& pr : posterior_pairs) kept_rows_.back()[pr.first] = (size_t)pr.second;
        } else if (ABC::multi_device() && ranking_kind_ == ABC::FILTER::PLS) {
            // several GPUs (ABC::use_devices): rank + truncate + doubled variance + weights of this set in one row-sharded call
            const size_t K = get_pred_prior_size_at(t);
            Mat2D prev_post;
            if (t > 0) prev_post = ABC::select_rows(set_params_[t - 1], kept_rows_[t - 1]);
            ABC::RankedSet rs = ABC::rank_and_weight(set_metrics_[t], set_params_[t], observed_, train_frac_, K,
                                                     pars_, t > 0 ? &prev_post : nullptr, t > 0 ? &set_weights_[t - 1] : nullptr,
                                                     t > 0 ? &set_dv_[t - 1] : nullptr, component_rule(), max_components());
            kept_rows_.push_back(rs.idx);
            AbcLog::filtering_report(this, t, rs.theta, ABC::select_rows(set_metrics_[t], kept_rows_[t]), *log_stream);
            _transaction(db, "recording posterior ranks", [&] {
                for (size_t i = 0; i < K; i++)
                    db.exec(std::string("update ") + JOB_TABLE + " set posterior = " + std::to_string(i) + " where serial = " +
                            std::to_string(serials[t][kept_rows_[t][i]]) + ";");
            });
            set_dv_.push_back(rs.doubled_variance);
            set_weights_.push_back(rs.weights);
            continue;
        } else {                                 // rank on the GPU, keep the best K, record the ranks
            switch (ranking_kind_) {
                case ABC::FILTER::PLS:
                    kept_rows_.push_back(ABC::particle_ranking_PLS(set_metrics_[t], set_params_[t], observed_, train_frac_, component_rule(),
                                                                   max_components()));
                    break;
                case ABC::FILTER::SIMPLE:
                    kept_rows_.push_back(ABC::particle_ranking_simple(set_metrics_[t], set_params_[t], observed_));
                    break;
                default: std::cerr << "ERROR: Unsupported filtering method: " << ranking_kind_ << std::endl; return false;
            }
            const size_t K = get_pred_prior_size_at(t);
            kept_rows_.back().resize(K);
            AbcLog::filtering_report(this, t, ABC::select_rows(set_params_[t], kept_rows_[t]),
                                     ABC::select_rows(set_metrics_[t], kept_rows_[t]), *log_stream);
            _transaction(db, "recording posterior ranks", [&] {
                for (size_t i = 0; i < K; i++)
                    db.exec(std::string("update ") + JOB_TABLE + " set posterior = " + std::to_string(i) + " where serial = " +
                            std::to_string(serials[t][kept_rows_[t][i]]) + ";");
            });
        }
        calculate_predictive_prior_weights(t);
    }
    return true;
}

inline void AbcSmc::calculate_predictive_prior_weights(const size_t t) {      // AbcSmc.cpp:1041-1066
    const Mat2D post = ABC::select_rows(set_params_[t], kept_rows_[t]);
    set_dv_.push_back(ABC::calculate_doubled_variance(post));
    if (t == 0) {
        set_weights_.push_back(ABC::weight_predictive_prior(pars_, post));
    } else {
        set_weights_.push_back(ABC::weight_predictive_prior(pars_, post, ABC::select_rows(set_params_[t - 1], kept_rows_[t - 1]),
                                                        set_weights_[t - 1], set_dv_[t - 1]));
    }
}

inline bool AbcSmc::run(const ABC::RNG* RNG, const std::function<unsigned long(size_t)>& reseed) {
    auto seed_for = [&](size_t step) { return reseed ? reseed(step) : (unsigned long)time(NULL) * (unsigned long)getpid(); };
    const size_t sets = get_smc_iterations();
    for (size_t t = 0; t < sets; t++) {
        ABC::rng_set(RNG, seed_for(t));
        process_database(RNG);                                 // abc_inner ignores both results (examples.h:63-70)
        simulate_next_particles((int)get_smc_size_at(t));
    }
    ABC::rng_set(RNG, seed_for(sets));
    return process_database(RNG);                              // one last time, to get the posterior (examples.h:91-93)
}

inline bool AbcSmc::process_database(const ABC::RNG* RNG, const bool verbose) {   // AbcSmc.cpp:452-559
    if (build_database(RNG)) return true;
    sqdyn::Db db(db_path_);
    set_params_.clear();
    set_metrics_.clear();
    set_weights_.clear();
    kept_rows_.clear();
    set_dv_.clear();
    *log_stream << std::setprecision(5);
    std::vector<std::vector<int>> serials;
    if (!read_SMC_sets_from_database(db, serials)) return false;
    const size_t next_set = serials.size();
    if (next_set == 0) return false;
    AbcLog::report_convergence_data(this, next_set - 1, *log_stream);
    *log_stream << std::endl << std::endl;
    if (n_sets_ > next_set) {
        const size_t num_particles = get_smc_size_at(next_set);
        const Mat2D prior = ABC::select_rows(set_params_[next_set - 1], kept_rows_[next_set - 1]);
        std::vector<unsigned long> seeds;
        Mat2D noised;
        if (noise_kind_ == ABC::NOISE::MULTIVARIATE) {
            const Mat2D L = ABC::setup_mvn_sampler(prior);
            noised = ABC::sample_mvn_predictive_priors(RNG, num_particles, set_weights_[next_set - 1], prior, pars_, L, &seeds);
            if (verbose) std::cerr << "Populating next set using MULTIVARIATE noising of parameters." << std::endl;
        } else {
            noised = ABC::sample_predictive_priors(RNG, num_particles, set_weights_[next_set - 1], prior, pars_, set_dv_[next_set - 1], &seeds);
            if (verbose) std::cerr << "Populating next set using INDEPENDENT noising of parameters." << std::endl;
        }
        const long long last_serial = serials.back().back();
        _transaction(db, "inserting the next set", [&] { _insert_particles(db, next_set, last_serial + 1, noised, seeds, {}); });
    } else {
        std::cerr << "Database already contains " << n_sets_ << " complete sets.\n";
    }
    return true;
}

// =============================================================================================================
// simulation (AbcSmc.cpp:681-689, 876-1039)
// =============================================================================================================
inline bool AbcSmc::run_sim_(Row& par, Row& met, const size_t rng_seed, const size_t serial) {
    std::vector<float_type> met_vec = (*sim_)(par, rng_seed, serial);
    const bool ok = (met_vec.size() == nmet());
    if (!ok) std::cerr << "ERROR: simulator function returned the wrong number of metrics: expected " << nmet() << ", received " << met_vec.size() << std::endl;
    met = met_vec;
    return ok;
}

inline bool AbcSmc::simulate_next_particles(const int n, const int serial_req, const int posterior_req) {
    using abcsmc_detail::num;
    using namespace std::chrono;
    const bool verbose = (n == 1);
    if (!(n == 1 || (serial_req == -1 && posterior_req == -1)) || !(serial_req == -1 || posterior_req == -1))
        throw std::invalid_argument("simulate_next_particles: a serial or posterior request is for exactly one particle");
    sqdyn::Db db(db_path_);
    const std::string model_par_table = db.table_exists(UPAR_TABLE) ? UPAR_TABLE : PAR_TABLE;
    std::ostringstream select_ss;
    select_ss << "select J.serial, P.seed, " << _column_list("P.", true, "") << "from " << model_par_table << " P, " << JOB_TABLE
              << " J where P.serial = J.serial ";
    if (serial_req > -1) select_ss << "and J.serial = " << serial_req << ";";
    else if (posterior_req > -1) select_ss << "and smcSet = (select max(smcSet) from job where posterior > -1) and posterior = " << posterior_req << ";";
    else select_ss << "and (J.status = 'Q' or J.status = 'R') order by J.status, J.attempts " << (n == -1 ? std::string() : "limit " + std::to_string(n)) << ";";

    const long long overall_start = duration_cast<seconds>(system_clock::now().time_since_epoch()).count();
    std::vector<int> serials;
    std::vector<unsigned long> seeds;
    std::vector<Row> par_mat;
    const bool fetched = _transaction(db, "fetching particle parameters", [&] {       // AbcSmc.cpp:876-927
        if (verbose) std::cerr << "Attempting: " << select_ss.str() << std::endl;
        {
            sqdyn::Stmt s = db.query(select_ss.str());
            while (s.next()) {
                serials.push_back((int)s.i64(0));
                seeds.push_back((unsigned long)s.i64(1));
                Row pars(npar());
                for (size_t j = 0; j < npar(); j++) pars[j] = s.f64(2 + (int)j);
                par_mat.push_back(pars);
            }
        }
        for (int serial : serials)
            db.exec(std::string("update ") + JOB_TABLE + " set startTime = " + std::to_string(overall_start) +
                    ", status = 'R', attempts = attempts + 1 where serial = " + std::to_string(serial) + ";");
    });
    if (!fetched) { std::cerr << "Parameter selection from database failed.\n"; return true; }

    std::vector<std::string> met_updates, job_updates;
    for (size_t i = 0; i < par_mat.size(); i++) {
        const auto start = system_clock::now();
        Row met(nmet());
        if (!run_sim_(par_mat[i], met, seeds[i], (size_t)serials[i])) exit(-211);
        std::string sets;
        for (size_t j = 0; j < nmet(); j++) sets += mets_[j]->get_short_name() + "=" + num(met[j]) + (j + 1 < nmet() ? ", " : " ");
        // only while the job is still running, queued or paused (AbcSmc.cpp:1006-1008)
        met_updates.push_back(std::string("update ") + MET_TABLE + " set " + sets + "where serial = " + std::to_string(serials[i]) +
                              " and (select (status is 'R' or status is 'Q' or status is 'P') from " + JOB_TABLE + " J where J.serial=" +
                              std::to_string(serials[i]) + ");");
        const double span = duration_cast<duration<double>>(system_clock::now() - start).count();
        job_updates.push_back(std::string("update ") + JOB_TABLE + " set startTime = " +
                              std::to_string(duration_cast<seconds>(start.time_since_epoch()).count()) + ", duration = " + num(span) +
                              ", status = 'D' where serial = " + std::to_string(serials[i]) + " and (status = 'R' or status = 'Q' or status = 'P');");
    }
    _transaction(db, "updating metrics", [&] {
        for (size_t i = 0; i < met_updates.size(); i++) { db.exec(met_updates[i]); db.exec(job_updates[i]); }
    });
    return true;
}

// =============================================================================================================
// reports
// =============================================================================================================
inline void AbcLog::table_header(AbcSmc* abc, std::ostream& os) {
    for (size_t i = 0; i < abc->npar(); i++) os << std::setw(WIDTH) << abc->pars_[i]->get_short_name();
    os << " | ";
    for (size_t i = 0; i < abc->nmet(); i++) os << std::setw(WIDTH) << abc->mets_[i]->get_short_name();
    os << std::endl;
}

inline void AbcLog::report_convergence_data(AbcSmc* abc, const size_t set_t, std::ostream& os) {   // AbcLog.cpp:24-77
    if (abc->kept_rows_.size() <= set_t) {
        os << "ERROR: attempting to report stats for set " << set_t << ", but data aren't available. " << std::endl
           << "       This can happen if --process is called on a database that is not ready to be processed." << std::endl;
        exit(-214);
    }
    const Row current_means = ABC::col_means(ABC::select_rows(abc->set_params_[set_t], abc->kept_rows_[set_t]));
    Row last_means;
    if (set_t > 0) last_means = ABC::col_means(ABC::select_rows(abc->set_params_[set_t - 1], abc->kept_rows_[set_t - 1]));
    os << double_bar() << std::endl << (set_t == 0 ? "Predictive prior summary statistics:\n" : "Convergence data for predictive priors:\n");
    auto pct = [](double delta, double base) { return base != 0 ? 100 * delta / base : INFINITY; };
    for (size_t j = 0; j < abc->pars_.size(); j++) {
        const ABC::Parameter* par = abc->pars_[j];
        const double cur_sd = std::sqrt(abc->set_dv_[set_t][j] / 2.0);
        const double pm = par->get_mean(), ps = par->get_sd();
        os << "  Par " << j << ": \"" << par->get_name() << "\"\n" << "  Means:\n";
        print_stats("Prior", "current", pm, current_means[j], current_means[j] - pm, pct(current_means[j] - pm, pm), "", os);
        if (set_t != 0) print_stats("Last", " current", last_means[j], current_means[j], current_means[j] - last_means[j],
                                    pct(current_means[j] - last_means[j], last_means[j]), "\n", os);
        os << "  Standard deviations:\n";
        print_stats("Prior", "current", ps, cur_sd, cur_sd - ps, pct(cur_sd - ps, ps), "\n", os);
        if (set_t != 0) {
            const double last_sd = std::sqrt(abc->set_dv_[set_t - 1][j] / 2.0);
            print_stats("Last", " current", last_sd, cur_sd, cur_sd - last_sd, pct(cur_sd - last_sd, last_sd), "\n", os);
        }
    }
}

inline void AbcLog::filtering_report(AbcSmc* abc, const size_t t, const Mat2D& ppars, const Mat2D& pmets, std::ostream& os) {   // AbcLog.cpp:79-123
    os << double_bar() << std::endl << "Set " << t << std::endl << double_bar() << std::endl;
    os << "Observed:" << std::endl;
    table_header(abc, os);
    for (size_t i = 0; i < ppars.cols(); i++) os << std::setw(WIDTH) << "---";
    os << " | ";
    for (auto m : abc->mets_) os << std::setw(WIDTH) << m->get_obs_val();
    os << std::endl;
    os << "Normalized RMSE for metric means (lower is better):  " << ABC::calculate_nrmse(pmets, abc->observed_) << std::endl;
    auto medians = [](const Mat2D& m) {
        Row out(m.cols());
        for (size_t j = 0; j < m.cols(); j++) out[j] = ABC::median(Col(m.data() + j * m.rows(), m.data() + (j + 1) * m.rows()));
        return out;
    };
    auto row_of = [&](const char* title, const Row& a, const Row& b) {
        os << title << std::endl;
        table_header(abc, os);
        for (double v : a) os << std::setw(WIDTH) << v;
        os << " | ";
        for (double v : b) os << std::setw(WIDTH) << v;
        os << std::endl;
    };
    row_of("Posterior means:", ABC::col_means(ppars), ABC::col_means(pmets));
    row_of("Posterior medians:", medians(ppars), medians(pmets));
    auto rows = [&](const char* title, size_t first, size_t count) {
        os << title << std::endl;
        table_header(abc, os);
        for (size_t q = first; q < first + count; q++) {
            for (size_t j = 0; j < ppars.cols(); j++) os << std::setw(WIDTH) << ppars(q, j);
            os << " | ";
            for (size_t j = 0; j < pmets.cols(); j++) os << std::setw(WIDTH) << pmets(q, j);
            os << std::endl;
        }
    };
    const size_t five = std::min<size_t>(5, ppars.rows());      // the reference assumes >= 5 posterior rows
    rows("Best five:", 0, five);
    rows("Worst five:", ppars.rows() - five, five);
}

#endif
