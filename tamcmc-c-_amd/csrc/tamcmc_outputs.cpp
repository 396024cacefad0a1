// tamcmc_outputs.cpp -- result / restore files in the reference's formats and the single-process phase driver
// (include/tamcmc_outputs.h).  Host-only C++; uses the public C API of the setup and of the
// sampler, nothing else.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "tamcmc_outputs.h"

namespace {

typedef std::vector<double> dvec;

struct Fail { int code; std::string msg; };

std::string cfg_get(const tamcmc_setup *s, const char *group, const char *key, const std::string &dflt = "")
{
    char buf[4096];
    if (tamcmc_setup_get(s, group, key, buf, (int32_t)sizeof(buf)) != TAMCMC_IO_OK) return dflt;
    return buf;
}

std::string trim(const std::string &s)
{
    const size_t b = s.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) return "";
    return s.substr(b, s.find_last_not_of(" \t\r\n") - b + 1);
}

long cfg_long(const tamcmc_setup *s, const char *group, const char *key, long dflt)
{
    const std::string v = trim(cfg_get(s, group, key));
    if (v.empty()) return dflt;
    long out = dflt;
    std::istringstream(v) >> out;
    return out;
}

// Eigen's default stream format for one row (IOFormat(): coefficients separated by one space, every coefficient
// padded to the width of the widest one; precision = the stream's).
template <class T> std::string eigen_row(const T *v, size_t n, int precision)
{
    std::vector<std::string> cells(n);
    size_t width = 0;
    for (size_t i = 0; i < n; i++) {
        std::ostringstream o;
        o.precision(precision);
        o << v[i];
        cells[i] = o.str();
        if (cells[i].size() > width) width = cells[i].size();
    }
    std::string out;
    for (size_t i = 0; i < n; i++) {
        if (i) out += " ";
        out += std::string(width - cells[i].size(), ' ') + cells[i];
    }
    return out;
}

dvec parse_row(const std::string &line)
{
    dvec out;
    std::istringstream is(line);
    double v;
    while (is >> v) out.push_back(v);
    return out;
}

dvec sampler_get(const tamcmc_sampler *s, int which, size_t count)
{
    dvec v(count);
    if (tamcmc_sampler_get(s, which, v.data(), (int64_t)count) != TAMCMC_OK)
        throw Fail{TAMCMC_IO_E_INVALID, "tamcmc_sampler_get failed (sampler / outputs size mismatch)"};
    return v;
}

} // namespace

struct tamcmc_outputs {
    // configuration (Outputs::Outputs, outputs.cpp:24-146)
    long Nbuffer = 1, Nsamples = 0, sofar = 0;
    int Nchains = 0, Nvars = 0, Ncons = 0, Nparams = 0;
    bool erase_old_files = true, get_params = true, get_stat = true, get_pt = true;
    bool want_bin = true, want_txt = false, dbg = false;
    std::string file_ext = "bin";
    std::string f_params, f_pt, f_stat, f_acc, f_restore;
    std::vector<std::string> vars_names, cons_names;
    dvec cons, Tcoefs;
    std::vector<int32_t> relax;
    int32_t plength[11];
    int restore_precision = 6;
    // buffers: `counts` samples of the current block
    long counts = 0, Ncopy = 0, written = 0;
    dvec b_vars, b_stat, b_Pswitch;               // [Nbuffer][Nchains][Nvars], [Nbuffer][3 Nchains], [Nbuffer]
    std::vector<uint8_t> b_attempt, b_switched, b_moved;   // [Nbuffer], [Nbuffer], [Nbuffer][Nchains]
    std::vector<int32_t> b_chain0;
    dvec sum_sigma, sum_mu, sum_covar, sum_vars;  // running sums of the block, for the *_mean entries
    std::string error;
};

namespace {

void open_or_fail(std::ofstream &f, const std::string &name, bool append, bool binary)
{
    std::ios_base::openmode mode = std::ios_base::out;
    if (append) mode |= std::ios_base::app;
    if (binary) mode |= std::ios_base::binary;
    f.open(name.c_str(), mode);
    if (!f.is_open()) throw Fail{TAMCMC_IO_E_OPEN, "Unable to open file " + name + " (check that the full path exists)"};
}

std::string names_line(const std::vector<std::string> &v)
{
    std::string out;
    for (const std::string &n : v) out += n + "   ";
    return out;
}

void write_params_header(const tamcmc_outputs &o, std::ostream &f, bool binary, int chain)
{
    if (binary) {
        f << "# This is the header file of the BINARY output file for the model parameters \n";
        f << "# This file contains values for vars[0:Nchains-1][ 0:Nvars-1]. Each matrix is in a different file, indexed by the chain number\n";
    } else {
        f << "# This is an output file for the model parameters \n";
        f << "# This file contains values for vars[0:Nchains-1][ 0:Nvars-1]. Each matrix is in a different file, indexed by the chain number\n";
    }
    f << "! Nsamples= " << o.Nsamples << "\n";
    f << "! Nchains= " << o.Nchains << "\n";
    if (binary) f << "! Nsamples_done=" << o.written + o.counts + o.sofar << "\n";
    f << "! Nvars= " << o.Nvars << "\n";
    f << "! Ncons= " << o.Ncons << "\n";
    if (!binary) f << "! chain= " << chain << "\n";
    f << "! relax= " << eigen_row(o.relax.data(), o.relax.size(), 6) << "\n";
    f << "! plength= " << eigen_row(o.plength, 11, 6) << "\n";
    f << "! constant_names= " << names_line(o.cons_names) << "\n";
    f << "! constant_values= ";
    if (o.cons_names[0] == "None") f << "-1\n";
    else f << eigen_row(o.cons.data(), o.cons.size(), 6) << "\n";
    f << "! variable_names=" << names_line(o.vars_names) << "\n";
}

// Flush the `counts` buffered samples to every requested file (Outputs::write_bin_* / write_txt_*).
void flush_block(tamcmc_outputs &o)
{
    const long n = o.counts;
    if (n <= 0) return;
    const bool first = o.Ncopy == 0;
    const bool truncate = o.erase_old_files && first;
    const bool need_header = o.erase_old_files;
    const size_t nc = (size_t)o.Nchains, nv = (size_t)o.Nvars;
    const std::string txt_ext = o.dbg ? o.file_ext + ".txt" : o.file_ext;

    if (o.get_params) {
        if (o.want_bin) {
            if (need_header) { std::ofstream h; open_or_fail(h, o.f_params + ".hdr", false, false); write_params_header(o, h, true, 0); }
            for (size_t c = 0; c < nc; c++) {
                std::ofstream f;
                open_or_fail(f, o.f_params + "_chain-" + std::to_string(c) + "." + o.file_ext, !truncate, true);
                for (long i = 0; i < n; i++)
                    f.write(reinterpret_cast<const char *>(&o.b_vars[((size_t)i * nc + c) * nv]), (std::streamsize)(nv * sizeof(double)));
            }
        }
        if (o.want_txt)
            for (size_t c = 0; c < nc; c++) {
                std::ofstream f;
                open_or_fail(f, o.f_params + "_chain-" + std::to_string(c) + "." + txt_ext, !truncate, false);
                if (first && need_header) write_params_header(o, f, false, (int)c);
                for (long i = 0; i < n; i++) f << eigen_row(&o.b_vars[((size_t)i * nc + c) * nv], nv, 6) << "\n";
            }
    }
    if (o.get_stat) {
        const char *labels[3] = {"logLikelihood", "logPrior", "logPosteriors"};
        auto label_line = [&](std::ostream &f) {
            f << "! labels= ";
            for (int k = 0; k < 3; k++) for (size_t c = 0; c < nc; c++) f << labels[k] << "[" << c << "]   ";
            f << "\n";
        };
        if (o.want_bin) {
            if (need_header) {
                std::ofstream h; open_or_fail(h, o.f_stat + ".hdr", false, false);
                h << "# This is the header of the BINARY output file for the statistical information.\n";
                h << "# This file contains values for the logLikelihood (columns 0:Nchains-1), logPrior (columns Nchains:2*Nchains-1) and logPosterior (columns 2*Nchains:3*Nchains-1),  \n";
                h << "! Nsamples_done=" << o.written + o.counts + o.sofar << "\n";
                h << "! Nchains= " << o.Nchains << "\n";
                label_line(h);
            }
            std::ofstream f;
            open_or_fail(f, o.f_stat + "." + o.file_ext, !truncate, true);
            f.write(reinterpret_cast<const char *>(o.b_stat.data()), (std::streamsize)((size_t)n * 3 * nc * sizeof(double)));
        }
        if (o.want_txt) {
            std::ofstream f;
            open_or_fail(f, o.f_stat + "." + txt_ext, !truncate, false);
            if (first && need_header) {
                f << "# This is an output file for the statistical information.\n";
                f << "# This file contains values for the logLikelihood (columns 0:Nchains-1), logPrior (columns Nchains:2*Nchains-1) and logPosterior (columns 2*Nchains:3*Nchains-1),  \n";
                f << "! Nchains= " << o.Nchains << "\n";
                label_line(f);
            }
            for (long i = 0; i < n; i++) {
                const double *r = &o.b_stat[(size_t)i * 3 * nc];
                f << eigen_row(r, nc, 6) << "     " << eigen_row(r + nc, nc, 6) << "     " << eigen_row(r + 2 * nc, nc, 6) << "\n";
            }
        }
    }
    if (o.get_pt) {
        if (o.want_bin) {
            if (need_header) {
                std::ofstream h; open_or_fail(h, o.f_pt + ".hdr", false, false);
                h << "# This is the header of the BINARY output file for the parameters of the parallel tempering.\n";
                h << "# This file contains values for \n";
                h << "# Correspondance between chain0=[0:Nchains-1] and temperature Tcoefs[chain] \n";
                h << "! Nsamples_done=" << o.written + o.counts + o.sofar << "\n";
                h << "! Tcoefs = " << eigen_row(o.Tcoefs.data(), nc, 6) << "\n";
                h << "! labels= attempt_mixing    chain0    Pswitch    switched \n";
            }
            std::ofstream f;
            open_or_fail(f, o.f_pt + "." + o.file_ext, !truncate, true);
            for (long i = 0; i < n; i++) {                      // bool, int, double, bool: 14 bytes, outputs.cpp:1383-1393
                f.write(reinterpret_cast<const char *>(&o.b_attempt[i]), 1);
                f.write(reinterpret_cast<const char *>(&o.b_chain0[i]), sizeof(int32_t));
                f.write(reinterpret_cast<const char *>(&o.b_Pswitch[i]), sizeof(double));
                f.write(reinterpret_cast<const char *>(&o.b_switched[i]), 1);
            }
        }
        if (o.want_txt) {
            std::ofstream f;
            open_or_fail(f, o.f_pt + "." + txt_ext, !truncate, false);
            if (first && need_header) {
                f << "# This is an output file for the parameters of the parallel tempering.\n";
                f << "# This file contains values for \n";
                f << "# Correspondance between chain0=[0:Nchains-1] and temperature Tcoefs[chain] \n";
                f << "! Tcoefs = " << eigen_row(o.Tcoefs.data(), nc, 6) << "\n";
                f << "! labels= attempt_mixing    chain0    Pswitch    switched \n";
            }
            for (long i = 0; i < n; i++)
                f << (int)o.b_attempt[i] << "   " << o.b_chain0[i] << "   " << o.b_Pswitch[i] << "   " << (int)o.b_switched[i] << "\n";
        }
    }
    {   // acceptance.txt: one line per block (Outputs::write_txt_acceptance, outputs.cpp:747-789)
        const bool exists = (bool)std::ifstream((o.f_acc + ".txt").c_str());
        const bool hdr = !(exists && !o.erase_old_files);
        std::ofstream f;
        open_or_fail(f, o.f_acc + ".txt", !truncate, false);
        if (first && hdr) {
            f << "# This is an output file for the acceptance rate. \n";
            f << "# This file contains values for the acceptance_rate[0:Nchains-1] in function of the average sample position\n";
            f << "# Averaging is done over Nbuffer \n";
            f << "! Nchains= " << o.Nchains << "\n";
        }
        dvec rate(nc, 0.0);
        for (size_t c = 0; c < nc; c++) {
            long acc = 0;
            for (long i = 0; i < n; i++) acc += o.b_moved[(size_t)i * nc + c];
            rate[c] = (double)acc / (double)n;
        }
        std::ostringstream x;
        x << ((double)o.Ncopy + 0.5) * (double)o.Nbuffer + (double)o.sofar;                 // reject_rate, outputs.cpp:1836
        f << x.str() << " " << eigen_row(rate.data(), nc, 6) << "\n";
    }
    o.written += n;
    o.counts = 0;
    o.Ncopy++;
}

void write_restore_arrays(tamcmc_outputs &o, const dvec &vars, const dvec &sigma, const dvec &mu, const dvec &covar, long block_count)
{
    const size_t nc = (size_t)o.Nchains, nv = (size_t)o.Nvars;
    const int P = o.restore_precision;
    const double inv = block_count > 0 ? 1.0 / (double)block_count : 0.0;
    auto mean_of = [&](const dvec &sum, const dvec &last) {
        dvec m(sum.size());
        for (size_t i = 0; i < sum.size(); i++) m[i] = block_count > 0 ? sum[i] * inv : last[i];
        return m;
    };
    const dvec vars_m = mean_of(o.sum_vars, vars), sigma_m = mean_of(o.sum_sigma, sigma), mu_m = mean_of(o.sum_mu, mu),
               covar_m = mean_of(o.sum_covar, covar);
    const long iteration = o.written + o.counts + o.sofar - 1 < 0 ? 0 : o.written + o.counts + o.sofar - 1;   // index of the last sample
    auto head = [&](std::ostream &f, int number, const char *what) {
        const char *x = number == 1 ? "do_restore_[X]=1" : "do_restore=1";
        f << "# This is an output file containing what is required to restore a run to its last saved position \n";
        f << "# File number: " << number << " \n";
        f << what;
        f << "# Use this if you wish to: \n";
        f << "#       (1) complete a finished job that requires more samples ==> set erase_old_file=0 and " << x << " \n";
        f << "#       (2) restart a finished job by ignoring old samples (e.g. ignoring a Burn-in) ==> set erase_old_file=1 and "
          << (number == 1 ? "do_restore_proposal=1" : "do_restore=1") << " \n";
        f << "#       (3) terminate an unfinished job which failed to finished (e.g. due to computer unexpected shutdown) ==> set erase_old_file=0 and " << x << " \n";
        f << "! Nchains= " << o.Nchains << "\n";
        f << "! Nvars= " << o.Nvars << "\n";
        f << "! iteration=" << iteration << "\n";
        f << "! variable_names=" << names_line(o.vars_names) << "\n";
    };
    auto rows = [&](std::ostream &f, const dvec &m, size_t nrows, size_t ncols, size_t offset = 0) {
        for (size_t r = 0; r < nrows; r++) f << eigen_row(&m[offset + r * ncols], ncols, P) << "\n";
    };
    {
        std::ofstream f; open_or_fail(f, o.f_restore + "1.dat", false, false);
        head(f, 1, "# Contains the last values for the variables vars[0:Nchain-1]. vars_mean denotes averaged values of Nbuffer \n");
        f << "! vars= \n"; rows(f, vars, nc, nv);
        f << "! vars_mean= \n"; rows(f, vars_m, nc, nv);
    }
    {
        std::ofstream f; open_or_fail(f, o.f_restore + "2.dat", false, false);
        head(f, 2, "# Contains the last values of (a) sigmas[0:Nchains-1] and (b) mus[0:Nchains-1, 0:Nvars-1].  sigmas_mean and mus_mean denotes averaged values of Nbuffer\n");
        f << "! sigmas= " << eigen_row(sigma.data(), nc, P) << "\n";
        f << "! mus= \n"; rows(f, mu, nc, nv);
        f << "! sigmas_mean= " << eigen_row(sigma_m.data(), nc, P) << "\n";
        f << "! mus_mean= \n"; rows(f, mu_m, nc, nv);
    }
    {
        std::ofstream f; open_or_fail(f, o.f_restore + "3.dat", false, false);
        head(f, 3, "# Contains the last value of the covariance matrix covarmats[0:Nchains-1, 0:Nvars-1, 0:Nvars-1]. covarmats_mean denotes the averaged values over Nbuffer\n");
        f << "! covarmats= \n";
        for (size_t c = 0; c < nc; c++) { f << "*" << c << "\n"; rows(f, covar, nv, nv, c * nv * nv); }
        f << "! covarmats_mean= \n";
        for (size_t c = 0; c < nc; c++) { f << "*" << c << "\n"; rows(f, covar_m, nv, nv, c * nv * nv); }
    }
}

void write_restore(tamcmc_outputs &o, const tamcmc_sampler *s, long block_count)
{
    const size_t nc = (size_t)o.Nchains, nv = (size_t)o.Nvars;
    write_restore_arrays(o, sampler_get(s, 0, nc * nv), sampler_get(s, 6, nc), sampler_get(s, 7, nc * nv), sampler_get(s, 8, nc * nv * nv),
                         block_count);
}

void reset_sums(tamcmc_outputs &o)
{
    std::fill(o.sum_sigma.begin(), o.sum_sigma.end(), 0.0);
    std::fill(o.sum_mu.begin(), o.sum_mu.end(), 0.0);
    std::fill(o.sum_covar.begin(), o.sum_covar.end(), 0.0);
    std::fill(o.sum_vars.begin(), o.sum_vars.end(), 0.0);
}

template <class F> int guarded(std::string *err, F &&fn)
{
    try { fn(); return TAMCMC_IO_OK; }
    catch (const Fail &e) { if (err) *err = e.msg; return e.code; }
    catch (const std::exception &e) { if (err) *err = e.what(); return TAMCMC_IO_E_SYNTAX; }
}

void copy_err(const std::string &s, char *buf, int32_t cap)
{
    if (!buf || cap <= 0) return;
    const size_t n = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1;
    memcpy(buf, s.data(), n);
    buf[n] = 0;
}

} // namespace

extern "C" int tamcmc_outputs_create(tamcmc_outputs **out, const tamcmc_setup *setup, int32_t Nchains, const double *Tcoefs,
                                     int64_t iteration0, int32_t restore_precision)
{
    if (!out || !setup || Nchains < 1 || !Tcoefs || iteration0 < 0) return TAMCMC_IO_E_INVALID;
    *out = nullptr;
    tamcmc_outputs *o = new tamcmc_outputs();
    const int rc = guarded(&o->error, [&] {
        int32_t Nparams = 0, Nvars = 0;
        if (tamcmc_setup_sizes(setup, &Nparams, &Nvars, nullptr, o->plength, nullptr, nullptr, nullptr, nullptr) != TAMCMC_IO_OK)
            throw Fail{TAMCMC_IO_E_INVALID, "the setup holds no loaded model"};
        o->Nparams = Nparams; o->Nvars = Nvars; o->Nchains = Nchains;
        o->Tcoefs.assign(Tcoefs, Tcoefs + Nchains);
        o->Nbuffer = cfg_long(setup, "Outputs", "Nbuffer", 1000);
        o->Nsamples = cfg_long(setup, "Outputs", "Nsamples", 0);
        if (o->Nbuffer < 1 || o->Nsamples < 1) throw Fail{TAMCMC_IO_E_SYNTAX, "Outputs.Nbuffer and Outputs.Nsamples must be positive"};
        o->sofar = (long)iteration0;
        o->erase_old_files = cfg_long(setup, "Outputs", "erase_old_files", 1) != 0;
        o->get_params = cfg_long(setup, "Outputs", "get_params", 1) != 0;
        o->get_stat = cfg_long(setup, "Outputs", "get_statcriteria", 1) != 0;
        o->get_pt = cfg_long(setup, "Outputs", "get_parallel_tempering", 1) != 0;
        const std::string fmt = trim(cfg_get(setup, "Outputs", "file_format", "binary"));
        if (fmt == "binary") { o->want_bin = true; o->want_txt = false; o->file_ext = "bin"; }
        else if (fmt == "text") { o->want_bin = false; o->want_txt = true; o->file_ext = "txt"; }
        else if (fmt == "debug") { o->want_bin = true; o->want_txt = true; o->dbg = true; o->file_ext = "dbg"; }
        else throw Fail{TAMCMC_IO_E_SYNTAX, "file_format must be 'text', 'binary' or 'debug' (lower case)"};
        const std::string dir = trim(cfg_get(setup, "Outputs", "output_dir")), root = trim(cfg_get(setup, "Outputs", "output_root_name"));
        o->f_params = dir + root + trim(cfg_get(setup, "Outputs", "params_txt_fileout", "params"));
        o->f_pt = dir + root + trim(cfg_get(setup, "Outputs", "parallel_tempering_txt_fileout", "parallel_tempering"));
        o->f_stat = dir + root + trim(cfg_get(setup, "Outputs", "stat_txt_fileout", "stat_criteria"));
        o->f_acc = dir + root + trim(cfg_get(setup, "Outputs", "acceptance_txt_fileout", "acceptance"));
        o->f_restore = trim(cfg_get(setup, "Outputs", "restore_dir")) + trim(cfg_get(setup, "Outputs", "restore_file_out", "restore_"));
        o->restore_precision = restore_precision > 0 ? restore_precision : 6;

        dvec inputs(Nparams);
        o->relax.resize(Nparams);
        tamcmc_setup_inputs(setup, inputs.data(), o->relax.data(), nullptr, nullptr, nullptr, nullptr);
        char nm[512];
        for (int i = 0; i < Nparams; i++) {                                  // outputs.cpp:85-112
            tamcmc_setup_name(setup, 0, i, nm, sizeof(nm));
            if (o->relax[i] == 1) o->vars_names.push_back(nm);
            else { o->cons.push_back(inputs[i]); o->cons_names.push_back(nm); }
        }
        o->Ncons = (int)o->cons.size();
        if (o->Ncons == 0) { o->cons.assign(1, -1.0); o->cons_names.assign(1, "None"); }
        const size_t nb = (size_t)(o->Nbuffer < o->Nsamples ? o->Nbuffer : o->Nsamples), nc = (size_t)Nchains, nv = (size_t)Nvars;
        o->b_vars.resize(nb * nc * nv); o->b_stat.resize(nb * 3 * nc); o->b_Pswitch.resize(nb);
        o->b_attempt.resize(nb); o->b_switched.resize(nb); o->b_chain0.resize(nb); o->b_moved.resize(nb * nc);
        o->sum_sigma.assign(nc, 0.0); o->sum_mu.assign(nc * nv, 0.0); o->sum_covar.assign(nc * nv * nv, 0.0); o->sum_vars.assign(nc * nv, 0.0);
    });
    if (rc != TAMCMC_IO_OK) { fprintf(stderr, "tamcmc_outputs_create: %s\n", o->error.c_str()); delete o; return rc; }
    *out = o;
    return TAMCMC_IO_OK;
}

extern "C" int tamcmc_outputs_destroy(tamcmc_outputs *o) { delete o; return TAMCMC_IO_OK; }
extern "C" const char *tamcmc_outputs_error(const tamcmc_outputs *o) { return o ? o->error.c_str() : "null outputs"; }

extern "C" int tamcmc_outputs_record(tamcmc_outputs *o, const tamcmc_sampler *s, int32_t attempted, int32_t chain_A, double Pswap,
                                     int32_t swapped)
{
    if (!o || !s) return TAMCMC_IO_E_INVALID;
    return guarded(&o->error, [&] {
        const size_t nc = (size_t)o->Nchains, nv = (size_t)o->Nvars;
        const size_t cap = o->b_attempt.size();
        if ((size_t)o->counts >= cap) flush_block(*o);                      // cannot happen (flushed when full), kept as a guard
        const size_t i = (size_t)o->counts;
        dvec vars = sampler_get(s, 0, nc * nv), logL = sampler_get(s, 2, nc), logP = sampler_get(s, 3, nc), logPost = sampler_get(s, 4, nc),
             moved = sampler_get(s, 10, nc), sigma = sampler_get(s, 6, nc), mu = sampler_get(s, 7, nc * nv), covar = sampler_get(s, 8, nc * nv * nv);
        memcpy(&o->b_vars[i * nc * nv], vars.data(), nc * nv * sizeof(double));
        memcpy(&o->b_stat[i * 3 * nc], logL.data(), nc * sizeof(double));
        memcpy(&o->b_stat[i * 3 * nc + nc], logP.data(), nc * sizeof(double));
        memcpy(&o->b_stat[i * 3 * nc + 2 * nc], logPost.data(), nc * sizeof(double));
        for (size_t c = 0; c < nc; c++) o->b_moved[i * nc + c] = moved[c] != 0.0;
        o->b_attempt[i] = attempted ? 1 : 0; o->b_chain0[i] = chain_A; o->b_Pswitch[i] = Pswap; o->b_switched[i] = swapped ? 1 : 0;
        for (size_t k = 0; k < nc; k++) o->sum_sigma[k] += sigma[k];
        for (size_t k = 0; k < nc * nv; k++) { o->sum_mu[k] += mu[k]; o->sum_vars[k] += vars[k]; }
        for (size_t k = 0; k < nc * nv * nv; k++) o->sum_covar[k] += covar[k];
        o->counts++;
        const bool last = o->written + o->counts + o->sofar >= o->Nsamples;
        if ((size_t)o->counts == cap || o->counts == o->Nbuffer || last) {
            const long block = o->counts;
            write_restore(*o, s, block);
            flush_block(*o);
            reset_sums(*o);
        }
    });
}

extern "C" int tamcmc_outputs_finish(tamcmc_outputs *o, const tamcmc_sampler *s)
{
    if (!o || !s) return TAMCMC_IO_E_INVALID;
    return guarded(&o->error, [&] {
        if (o->counts > 0) {
            const long block = o->counts;
            write_restore(*o, s, block);
            flush_block(*o);
            reset_sums(*o);
        }
    });
}

// A whole block at once, from arrays that cover ALL chains (sharded runs: rank 0 gathers the ranks' blocks).
extern "C" int tamcmc_outputs_push_block(tamcmc_outputs *o, int64_t n, const double *vars, const double *stat, const uint8_t *moved,
                                         const uint8_t *attempted, const int32_t *chain0, const double *Pswitch, const uint8_t *switched,
                                         const double *last_vars, const double *last_sigma, const double *last_mu, const double *last_covar,
                                         const double *sum_vars, const double *sum_sigma, const double *sum_mu, const double *sum_covar)
{
    if (!o || n < 1 || !vars || !stat || !moved || !attempted || !chain0 || !Pswitch || !switched || !last_vars || !last_sigma ||
        !last_mu || !last_covar || !sum_vars || !sum_sigma || !sum_mu || !sum_covar) return TAMCMC_IO_E_INVALID;
    return guarded(&o->error, [&] {
        const size_t nc = (size_t)o->Nchains, nv = (size_t)o->Nvars;
        if (o->counts != 0 || (size_t)n > o->b_attempt.size()) throw Fail{TAMCMC_IO_E_INVALID, "push_block: block larger than Nbuffer, or a partial block is pending"};
        memcpy(o->b_vars.data(), vars, (size_t)n * nc * nv * sizeof(double));
        memcpy(o->b_stat.data(), stat, (size_t)n * 3 * nc * sizeof(double));
        memcpy(o->b_moved.data(), moved, (size_t)n * nc);
        memcpy(o->b_attempt.data(), attempted, (size_t)n);
        memcpy(o->b_chain0.data(), chain0, (size_t)n * sizeof(int32_t));
        memcpy(o->b_Pswitch.data(), Pswitch, (size_t)n * sizeof(double));
        memcpy(o->b_switched.data(), switched, (size_t)n);
        o->sum_vars.assign(sum_vars, sum_vars + nc * nv); o->sum_sigma.assign(sum_sigma, sum_sigma + nc);
        o->sum_mu.assign(sum_mu, sum_mu + nc * nv); o->sum_covar.assign(sum_covar, sum_covar + nc * nv * nv);
        o->counts = (long)n;
        write_restore_arrays(*o, dvec(last_vars, last_vars + nc * nv), dvec(last_sigma, last_sigma + nc), dvec(last_mu, last_mu + nc * nv),
                             dvec(last_covar, last_covar + nc * nv * nv), (long)n);
        flush_block(*o);
        reset_sums(*o);
    });
}

// ---------------------------------------------------------------- Config::read_restore_files, config.cpp:1322-1577
namespace {

struct Restored {
    int Nchains = 0, Nvars = 0;
    long iteration = 0;
    dvec vars, vars_mean, sigma, sigma_mean, mu, mu_mean, covar, covar_mean;
};

std::string after_eq(const std::string &line)
{
    const size_t p = line.find('=');
    return p == std::string::npos ? "" : line.substr(p + 1);
}

std::string key_of(const std::string &line)
{
    const size_t p = line.find('=');
    return trim(p == std::string::npos ? line : line.substr(0, p));
}

void read_rows(std::ifstream &f, dvec &dst, size_t nrows, size_t ncols, const std::string &what)
{
    dst.clear();
    std::string line;
    for (size_t r = 0; r < nrows; r++) {
        if (!std::getline(f, line)) throw Fail{TAMCMC_IO_E_SYNTAX, "restore file ends inside " + what};
        dvec row = parse_row(line);
        if (row.size() != ncols) throw Fail{TAMCMC_IO_E_SYNTAX, "restore file: a row of " + what + " does not have Nvars entries"};
        dst.insert(dst.end(), row.begin(), row.end());
    }
}

void read_mats(std::ifstream &f, dvec &dst, size_t nchains, size_t nv, const std::string &what)
{
    dst.clear();
    std::string line;
    for (size_t c = 0; c < nchains; c++) {
        if (!std::getline(f, line) || trim(line) != "*" + std::to_string(c))
            throw Fail{TAMCMC_IO_E_SYNTAX, "Syntax error while reading the covarmat: indicator of matrix " + std::to_string(c) + " not found in " + what};
        dvec m;
        read_rows(f, m, nv, nv, what);
        dst.insert(dst.end(), m.begin(), m.end());
    }
}

Restored read_restore(const std::string &root, bool want_proposal)
{
    Restored R;
    {
        std::ifstream f((root + "1.dat").c_str());
        if (!f.is_open()) throw Fail{TAMCMC_IO_E_OPEN, "restore file missing: " + root + "1.dat"};
        std::string line;
        int found = 0;
        while (std::getline(f, line)) {
            line = trim(line);
            if (line.empty() || line[0] != '!') continue;
            const std::string k = key_of(line);
            if (k == "! Nvars") { R.Nvars = (int)parse_row(after_eq(line)).at(0); found++; }
            else if (k == "! Nchains") { R.Nchains = (int)parse_row(after_eq(line)).at(0); found++; }
            else if (k == "! iteration") { R.iteration = (long)parse_row(after_eq(line)).at(0); found++; }
            else if (k == "! variable_names") found++;
            else if (k == "! vars") { read_rows(f, R.vars, R.Nchains, R.Nvars, "vars"); found++; }
            else if (k == "! vars_mean") { read_rows(f, R.vars_mean, R.Nchains, R.Nvars, "vars_mean"); found++; }
        }
        if (found != 6) throw Fail{TAMCMC_IO_E_SYNTAX, "Syntax error: At least one of the expected keywords were not found in " + root + "1.dat"};
    }
    if (!want_proposal) return R;
    {
        std::ifstream f((root + "2.dat").c_str());
        if (!f.is_open()) throw Fail{TAMCMC_IO_E_OPEN, "restore file missing: " + root + "2.dat"};
        std::string line;
        int found = 0;
        while (std::getline(f, line)) {
            line = trim(line);
            if (line.empty() || line[0] != '!') continue;
            const std::string k = key_of(line);
            if (k == "! sigmas") { R.sigma = parse_row(after_eq(line)); found++; }
            else if (k == "! mus") { read_rows(f, R.mu, R.Nchains, R.Nvars, "mus"); found++; }
            else if (k == "! sigmas_mean") { R.sigma_mean = parse_row(after_eq(line)); found++; }
            else if (k == "! mus_mean") { read_rows(f, R.mu_mean, R.Nchains, R.Nvars, "mus_mean"); found++; }
        }
        if (found != 4 || (int)R.sigma.size() != R.Nchains || (int)R.sigma_mean.size() != R.Nchains)
            throw Fail{TAMCMC_IO_E_SYNTAX, "Syntax error: At least one of the expected keywords were not found in " + root + "2.dat"};
    }
    {
        std::ifstream f((root + "3.dat").c_str());
        if (!f.is_open()) throw Fail{TAMCMC_IO_E_OPEN, "restore file missing: " + root + "3.dat"};
        std::string line;
        int found = 0;
        while (std::getline(f, line)) {
            line = trim(line);
            if (line.empty() || line[0] != '!') continue;
            const std::string k = key_of(line);
            if (k == "! covarmats") { read_mats(f, R.covar, R.Nchains, R.Nvars, "covarmats"); found++; }
            else if (k == "! covarmats_mean") { read_mats(f, R.covar_mean, R.Nchains, R.Nvars, "covarmats_mean"); found++; }
        }
        if (found != 2) throw Fail{TAMCMC_IO_E_SYNTAX, "Syntax error: At least one of the expected keywords were not found in " + root + "3.dat"};
    }
    return R;
}

} // namespace

extern "C" int tamcmc_restore_apply(const tamcmc_setup *setup, tamcmc_sampler *s, int64_t *iteration, char *errbuf, int32_t errcap)
{
    if (!setup || !s) return TAMCMC_IO_E_INVALID;
    if (iteration) *iteration = 0;
    std::string err;
    const int rc = guarded(&err, [&] {
        const bool r_vars = cfg_long(setup, "Outputs", "do_restore_variables", 0) != 0;
        const bool r_prop = cfg_long(setup, "Outputs", "do_restore_proposal", 0) != 0;
        const bool r_mean = cfg_long(setup, "Outputs", "do_restore_proposal_mean", 0) != 0;
        const bool r_index = cfg_long(setup, "Outputs", "do_restore_last_index", 0) != 0;
        if (r_index && !r_prop)                                                  // config.cpp:176-182
            throw Fail{TAMCMC_IO_E_SYNTAX, "If do_restore_last_index is 1, do_restore_proposal must also be 1"};
        if (!r_vars && !r_prop) return;
        const std::string root = trim(cfg_get(setup, "Outputs", "restore_dir")) + trim(cfg_get(setup, "Outputs", "restore_file_in", "restore_"));
        Restored R = read_restore(root, r_prop);
        const int nv = tamcmc_sampler_nvars(s);
        int32_t Nglobal = 0, off = 0, nloc = 0;
        tamcmc_sampler_layout(s, &Nglobal, &off, &nloc);
        if (R.Nvars != nv) throw Fail{TAMCMC_IO_E_SYNTAX, "Inconsistency in the number of variables between the model and the restore files"};
        if (R.Nchains != Nglobal) throw Fail{TAMCMC_IO_E_SYNTAX, "Inconsistency in the number of chains between the configuration and the restore files"};
        // a sharded process restores its own block of chains [off, off + nloc)
        const size_t o1 = (size_t)off * nv, n1 = (size_t)nloc * nv, o2 = (size_t)off * nv * nv, n2 = (size_t)nloc * nv * nv;
        if (r_vars && tamcmc_sampler_set(s, 0, R.vars.data() + o1, (int64_t)n1) != TAMCMC_OK)
            throw Fail{TAMCMC_IO_E_INVALID, "could not set the restored variables"};
        if (r_prop) {
            const dvec &sg = r_mean ? R.sigma_mean : R.sigma, &mu = r_mean ? R.mu_mean : R.mu, &cv = r_mean ? R.covar_mean : R.covar;
            if (tamcmc_sampler_set(s, 6, sg.data() + off, (int64_t)nloc) != TAMCMC_OK || tamcmc_sampler_set(s, 7, mu.data() + o1, (int64_t)n1) != TAMCMC_OK ||
                tamcmc_sampler_set(s, 8, cv.data() + o2, (int64_t)n2) != TAMCMC_OK)
                throw Fail{TAMCMC_IO_E_INVALID, "could not set the restored proposal"};
        }
        if (r_index) {
            tamcmc_sampler_set_iteration(s, R.iteration);
            if (iteration) *iteration = R.iteration;
        }
    });
    if (rc != TAMCMC_IO_OK) copy_err(err, errbuf, errcap);
    return rc;
}

// ---------------------------------------------------------------- MALA::execute, MALA.cpp:608-720 (single process)
extern "C" int tamcmc_run_phase(const tamcmc_setup *setup, tamcmc_sampler *s, tamcmc_progress_fn progress, void *user,
                                int32_t restore_precision, char *errbuf, int32_t errcap)
{
    if (!setup || !s) return TAMCMC_IO_E_INVALID;
    int64_t it0 = 0;
    int rc = tamcmc_restore_apply(setup, s, &it0, errbuf, errcap);
    if (rc != TAMCMC_IO_OK) return rc;
    rc = tamcmc_sampler_init(s);
    if (rc != TAMCMC_OK) { copy_err(std::string("tamcmc_sampler_init: ") + tamcmc_strerror(rc), errbuf, errcap); return TAMCMC_IO_E_INVALID; }
    const int nchains = tamcmc_sampler_nlocal(s);
    dvec T((size_t)nchains);
    tamcmc_sampler_get(s, 9, T.data(), nchains);
    tamcmc_outputs *o = nullptr;
    rc = tamcmc_outputs_create(&o, setup, nchains, T.data(), it0, restore_precision);
    if (rc != TAMCMC_IO_OK) { copy_err("could not set up the output files (see stderr)", errbuf, errcap); return rc; }
    const int64_t Nsamples = o->Nsamples;
    double Pswap = 0.0;
    int32_t swapped = 0;
    for (int64_t i = tamcmc_sampler_iteration(s); i < Nsamples; i++) {
        if (progress && i % o->Nbuffer == 0) progress(i, Nsamples, user);
        int src = tamcmc_sampler_mh_step(s);
        if (src != TAMCMC_OK) { copy_err(std::string("tamcmc_sampler_mh_step: ") + tamcmc_strerror(src), errbuf, errcap); tamcmc_outputs_destroy(o); return TAMCMC_IO_E_INVALID; }
        int32_t attempted = 0, A = -1;
        if (tamcmc_sampler_pt_due(s)) {
            double u = 0, r = 0;
            tamcmc_sampler_pt_draw(s, &A, &u);
            src = tamcmc_sampler_pt_local(s, A, u, &swapped, &r);
            if (src != TAMCMC_OK) { copy_err("tamcmc_sampler_pt_local failed (the phase driver is single-process)", errbuf, errcap); tamcmc_outputs_destroy(o); return TAMCMC_IO_E_INVALID; }
            Pswap = r;                     // Model_def::Pswap / ::swaped keep their last values between attempts
            attempted = 1;
        }
        rc = tamcmc_outputs_record(o, s, attempted, A, Pswap, swapped);
        if (rc != TAMCMC_IO_OK) { copy_err(o->error, errbuf, errcap); tamcmc_outputs_destroy(o); return rc; }
        tamcmc_sampler_end_iteration(s);
    }
    rc = tamcmc_outputs_finish(o, s);
    if (rc != TAMCMC_IO_OK) copy_err(o->error, errbuf, errcap);
    if (progress) progress(Nsamples, Nsamples, user);
    tamcmc_outputs_destroy(o);
    return rc;
}
