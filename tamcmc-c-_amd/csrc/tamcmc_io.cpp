// tamcmc_io.cpp -- readers of the reference's input files (include/tamcmc_io.h).
// Host-only C++.  Every block cites the reference file:line whose behaviour it restates; quirks that change
// a number or a name downstream are kept and marked "quirk".
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "tamcmc_io.h"

namespace {

typedef std::vector<double> dvec;
const double EMPTY = -9999.;
const long double PI_L = 3.141592653589793238L;

struct Fail {                       // replaces the reference's "print + exit(EXIT_FAILURE)"
    int code;
    std::string msg;
};

// ---------------------------------------------------------------- string_handler.cpp
std::string trim(const std::string &s)                                   // strtrim, string_handler.cpp:23-39
{
    const size_t b = s.find_first_not_of(" \t");
    if (b == std::string::npos) return "";
    const size_t e = s.find_last_not_of(" \t");
    return s.substr(b, e - b + 1);
}

std::string trim_eol(std::string s)                                      // files written on other systems
{
    while (!s.empty() && (s.back() == '\r' || s.back() == '\n')) s.pop_back();
    return s;
}

std::vector<std::string> split(const std::string &str, const std::string &delims)   // strsplit, :41-59
{
    std::string s = trim(str);
    std::vector<std::string> out;
    size_t pos = 0;
    while ((pos = s.find_first_of(delims)) != std::string::npos) {
        if (pos != 0) out.push_back(trim(s.substr(0, pos)));
        s.erase(0, pos + 1);
        s = trim(s);
    }
    if (pos != 0) out.push_back(s);     // the reference pushes the tail even when it is empty
    return out;
}

double to_dbl(const std::string &s)                                      // str_to_dbl, :391-396
{
    double v = 0.0;
    std::istringstream(trim(s)) >> v;
    return v;
}

long to_long(const std::string &s)
{
    long v = 0;
    std::istringstream(trim(s)) >> v;
    return v;
}

bool to_bool(const std::string &s)                                       // str_to_bool: operator>>(bool&), :412-417
{
    bool v = false;
    std::istringstream(trim(s)) >> v;
    return v;
}

dvec to_dvec(const std::string &line, const std::string &delims)         // str_to_Xdarr, :281-295
{
    std::vector<std::string> w = split(line, delims);
    dvec out(w.size(), 0.0);
    for (size_t i = 0; i < w.size(); i++)
        if (trim(w[i]) != "") std::istringstream(w[i]) >> out[i];
    return out;
}

std::string first_char(const std::string &line) { return trim(line.substr(0, 1)); }

// ---------------------------------------------------------------- function_rot.cpp:20-106 (host copy)
long factorial(int n) { long f = 1; for (int i = 2; i <= n; i++) f *= i; return f; }
double combi(int n, int r) { return (double)(factorial(n) / factorial(n - r) / factorial(r)); }

double dmm(int l, int m1, int m2, double beta)                           // function_rot.cpp:81-93
{
    double sum = 0.0;
    for (int s = 0; s <= l - m1; s++) {
        double var = combi(l + m2, l - m1 - s) * combi(l - m2, s) * std::pow(-1., l - m1 - s);
        var = var * std::pow(std::cos(beta / 2.), 2 * s + m1 + m2) * std::pow(std::sin(beta / 2.), 2 * l - 2 * s - m1 - m2);
        sum = sum + var;
    }
    sum = sum * std::sqrt((double)(factorial(l + m1) * factorial(l - m1)));
    sum = sum / std::sqrt((double)(factorial(l + m2) * factorial(l - m2)));
    return sum;
}

dvec amplitude_ratio(int l, double beta_deg)                             // function_rot.cpp:20-47
{
    const double PI = 3.141592653589793238462643;
    const double angle = PI * beta_deg / 180.;
    dvec out(2 * l + 1);
    for (int m = 0; m <= l; m++) {
        const double v = dmm(l, m, 0, angle);
        out[l + m] = v * v;
        out[l - m] = v * v;
    }
    return out;
}

// harvey_like on a list of frequencies (noise_models.cpp:17-41), used by the local reader for 2 points
dvec harvey_like(const dvec &noise, const dvec &x, int Nharvey)
{
    dvec y(x.size(), 0.0);
    int cpt = 0;
    for (int k = 0; k < Nharvey; k++) {
        if (noise[cpt + 1] != 0)
            for (size_t i = 0; i < x.size(); i++)
                y[i] += noise[cpt] / (1. + std::pow(1e-3 * noise[cpt + 1] * x[i], noise[cpt + 2]));
        cpt += 3;
    }
    for (size_t i = 0; i < x.size(); i++) y[i] += noise[cpt];
    return y;
}

// ---------------------------------------------------------------- data.h:85-106 MCMC_files
struct ModelFile {
    std::string ID;
    double Dnu = 0, numax = EMPTY, C_l = 0;
    std::vector<int> els;
    double fmin = 0, fmax = 0;
    std::vector<std::string> param_type;
    dvec freqs_ref;
    std::vector<bool> relax_freq, relax_gamma, relax_H;
    dvec hyper_priors;
    std::vector<dvec> eigen_params;      // rows of 6: l, nu, win_min, win_max, Gamma, H
    dvec noise_params;                   // 10
    std::vector<dvec> noise_s2;          // 10 x 3
    std::vector<std::string> common_names, common_priors;
    std::vector<dvec> modes_common;      // rows of 5, -9999 padded
};

// read_MCMC_file_MS_Global (io_ms_global.cpp:25-306) and read_MCMC_file_local (io_local.cpp:26-302): the same
// section walker; slice_ind < 0 selects the global behaviour (a second `*` line is an error).
ModelFile read_model_file(const std::string &path, int slice_ind)
{
    std::ifstream f(path.c_str());
    if (!f.is_open()) throw Fail{TAMCMC_IO_E_OPEN, "Unable to open the file: " + path};
    ModelFile mf;
    std::string line;
    int out = 0, range_counter = 0;
    bool range_done = false;

    std::getline(f, line);
    while (out < 3 && !f.eof()) {                                                       // header + mode list
        line = trim(trim_eol(line));
        const std::string c0 = first_char(line), c1 = line.size() > 1 ? line.substr(1, 1) : "";
        if (line.empty()) { std::getline(f, line); continue; }                          // (the reference has UB here)
        if (c0 == "#" && c1 == "K") {
            std::vector<std::string> w = split(line, "= \t");
            if (w.size() > 1) mf.ID = trim(w[1]);
        }
        if (c0 == "!" && c1 == "n") { std::vector<std::string> w = split(line, " "); if (w.size() > 1) mf.numax = to_dbl(w[1]); }
        if (c0 == "!" && c1 != "!" && c1 != "n") { std::vector<std::string> w = split(line, " "); if (w.size() > 1) mf.Dnu = to_dbl(w[1]); }
        if (c0 == "!" && c1 == "!") { std::vector<std::string> w = split(line, " "); if (w.size() > 1) mf.C_l = to_dbl(w[1]); }
        if (c0 == "*") {
            std::vector<std::string> w = split(line, " ");
            if (w.size() < 3) throw Fail{TAMCMC_IO_E_SYNTAX, "frequency range line needs two values: " + line};
            if (slice_ind < 0) {
                if (range_done)
                    throw Fail{TAMCMC_IO_E_SYNTAX, "Multiple range detected. This is not allowed with io_ms_global models and priors"};
                mf.fmin = to_dbl(w[1]); mf.fmax = to_dbl(w[2]); range_done = true;
            } else {
                if (range_counter == slice_ind) { mf.fmin = to_dbl(w[1]); mf.fmax = to_dbl(w[2]); }
                range_counter++;
            }
        }
        if (c0 != "#" && c0 != "!" && c0 != "*") {
            std::vector<std::string> w = split(line, " ");
            if (w.size() < 3 || !(w[0] == "p" || w[0] == "g" || w[0] == "co"))
                throw Fail{TAMCMC_IO_E_SYNTAX, "The type of mode should be either 'p', 'g' or 'co': " + line};
            mf.param_type.push_back(w[0]);
            mf.els.push_back((int)to_long(w[1]));
            mf.freqs_ref.push_back(to_dbl(w[2]));
            mf.relax_freq.push_back(w.size() >= 4 ? to_bool(w[3]) : true);
            mf.relax_H.push_back(w.size() >= 5 ? to_bool(w[4]) : true);       // column 5 = relax_H, column 6 = relax_W
            mf.relax_gamma.push_back(w.size() >= 6 ? to_bool(w[5]) : true);
        }
        if (c0 == "#") out++;
        std::getline(f, line);
    }
    // hyper priors: the line left in `line` by the loop above is dropped (io_ms_global.cpp:160-175)
    while (out < 4 && !f.eof()) {
        std::getline(f, line);
        line = trim(trim_eol(line));
        if (first_char(line) != "#") { if (!line.empty()) mf.hyper_priors.push_back(to_dbl(line)); }
        else out++;
    }
    std::getline(f, line);
    while (out < 5 && !f.eof()) {                                                       // eigen solution
        line = trim(trim_eol(line));
        if (first_char(line) != "#") {
            if (!line.empty()) {
                dvec r = to_dvec(line, " \t");
                if (r.size() != 6) throw Fail{TAMCMC_IO_E_SYNTAX, "eigen solution rows need 6 columns (l nu win_min win_max Gamma H): " + line};
                mf.eigen_params.push_back(r);
            }
        } else out++;
        std::getline(f, line);
    }
    dvec noise;
    while (out < 6 && !f.eof()) {                                                       // noise parameters
        line = trim(trim_eol(line));
        if (first_char(line) != "#") { dvec r = to_dvec(line, " \t"); noise.insert(noise.end(), r.begin(), r.end()); }
        else out++;
        std::getline(f, line);
    }
    if (noise.size() > 10) throw Fail{TAMCMC_IO_E_SYNTAX, "more than 10 noise parameters"};
    mf.noise_params.assign(10, -1.);                                                    // right-aligned, io_ms_global.cpp:226-230
    if (noise.size() == 10) mf.noise_params = noise;
    else for (size_t k = 0; k < noise.size(); k++) mf.noise_params[10 - noise.size() + k] = noise[k];
    std::vector<dvec> s2;
    int nlines = 0;
    while (out < 7 && !f.eof()) {                                                       // noise_s2
        line = trim(trim_eol(line));
        if (first_char(line) != "#") {
            dvec r = to_dvec(line, " \t");
            if (r.size() != 3) throw Fail{TAMCMC_IO_E_SYNTAX, "noise_s2 rows need 3 columns: " + line};
            s2.push_back(r);
        } else out++;
        nlines++;
        std::getline(f, line);
    }
    if (s2.size() > 10) throw Fail{TAMCMC_IO_E_SYNTAX, "more than 10 noise_s2 rows"};
    mf.noise_s2.assign(10, dvec(3, -1.));
    if (nlines < 11) for (size_t k = 0; k < s2.size(); k++) mf.noise_s2[10 - s2.size() + k] = s2[k];   // :254-258
    else for (size_t k = 0; k < s2.size(); k++) mf.noise_s2[k] = s2[k];
    while (out < 9 && !f.eof()) {                                                       // common parameters
        line = trim(trim_eol(line));
        if (first_char(line) != "#") {
            std::vector<std::string> w = split(line, " \t");
            if (w.size() >= 2) {
                mf.common_names.push_back(trim(w[0]));
                mf.common_priors.push_back(trim(w[1]));
                dvec row(5, EMPTY);
                for (size_t k = 2; k < w.size() && k - 2 < 5; k++) std::istringstream(w[k]) >> row[k - 2];
                mf.modes_common.push_back(row);
            }
        } else out++;
        std::getline(f, line);      // quirk: a last line without a trailing newline sets eof and is never interpreted
    }
    return mf;
}

// ---------------------------------------------------------------- data.h:38-48 Input_Data + io_models.cpp
struct Block {
    std::vector<std::string> names, prior_names;
    dvec inputs;
    std::vector<int> relax;
    std::vector<dvec> priors;            // priors[k][i], k < 4
    void init(int n)                     // IO_models::initialise_param, io_models.cpp:232-268
    {
        names.assign(n, "Empty"); prior_names.assign(n, "Fix");
        inputs.assign(n, 0.0); relax.assign(n, 0);
        priors.assign(4, dvec(n, EMPTY));
    }
    int size() const { return (int)inputs.size(); }
    dvec prior_col(int i) const { dvec c(4); for (int k = 0; k < 4; k++) c[k] = priors[k][i]; return c; }
    // IO_models::fill_param, io_models.cpp:24-59
    void fill(const std::string &name, const std::string &prior, double val, const dvec &pv, int pos, int i0)
    {
        if (pos < 0 || pos >= size()) throw Fail{TAMCMC_IO_E_SYNTAX, "parameter slot out of range while filling " + name};
        names[pos] = name; prior_names[pos] = prior; inputs[pos] = val;
        if (prior == "Fix") {
            relax[pos] = 0;
            for (int k = 0; k < 4; k++) priors[k][pos] = EMPTY;
        } else {
            relax[pos] = 1;
            for (int k = 0; k < 4; k++) priors[k][pos] = (size_t)(k + i0) < pv.size() ? pv[k + i0] : EMPTY;
        }
    }
    // IO_models::fill_param_vect, io_models.cpp:61-78
    void fill_vect(const dvec &vals, const std::vector<bool> &rel, const std::string &name, const std::string &prior,
                   const dvec &pv, int pos, int i0_if, int i0_else)
    {
        for (size_t i = 0; i < vals.size(); i++) {
            if (rel[i]) fill(name, prior, vals[i], pv, (int)i + pos, i0_if);
            else fill(name, "Fix", vals[i], pv, (int)i + pos, i0_else);
        }
    }
    void add(const Block &b, int pos)    // IO_models::add_param, io_models.cpp:103-127
    {
        for (int i = 0; i < b.size(); i++) {
            names[pos + i] = b.names[i]; prior_names[pos + i] = b.prior_names[i];
            inputs[pos + i] = b.inputs[i]; relax[pos + i] = b.relax[i];
            for (int k = 0; k < 4; k++) priors[k][pos + i] = b.priors[k][i];
        }
    }
};

struct InputData {
    std::string model_fullname;
    Block all;
    int plength[11] = {0};
    double extra_priors[4] = {0, 0, 0, 0};
};

std::vector<int> where_dbl(const dvec &v, double value, double tol)      // string_handler.cpp:81-108
{
    std::vector<int> idx;
    for (size_t i = 0; i < v.size(); i++)
        if (v[i] > value - tol && v[i] < value + tol) idx.push_back((int)i);
    return idx;
}

void need_prior(const std::string &key, const std::string &prior, const std::string &expected)
{
    if (prior != expected)
        throw Fail{TAMCMC_IO_E_SYNTAX, key + " should always be defined as '" + expected + "'"};
}

void no_fix_auto(const std::string &key, const std::string &prior)
{
    if (prior == "Fix_Auto")
        throw Fail{TAMCMC_IO_E_SYNTAX, "Fix_Auto is not implemented for " + key + ": you must choose the prior yourself"};
}

double rho_from_dnu(double Dnu)                                          // io_ms_global.cpp:310-323
{
    const double Dnu_sun = 135.1, R_sun = 6.96342e5, M_sun = 1.98855e30;
    const double rho_sun = (double)(M_sun * 1e3 / (4 * PI_L * std::pow(R_sun * 1e5, 3) / 3));
    return std::pow(Dnu / Dnu_sun, 2.) * rho_sun;
}

double centrifugal_eta(double a1, double rho)                            // io_ms_global.cpp:740
{
    const double G = 6.667e-8, Dnl = 0.75;
    return (double)((4. / 3.) * PI_L * Dnl * std::pow(a1 * 1e-6, 2.) / (rho * G));
}

// The part of the keyword loop that both readers share word for word: splitting block, inclination and the
// projected-splitting keywords (io_ms_global.cpp:704-777,823-863 == io_local.cpp:826-937).
struct SplitState {
    bool a1cosi = false, a1sini = false;
};

void splitting_keyword(const std::string &key, const std::string &prior, const dvec &row, Block &Snlm, Block &Inc,
                       bool a11_eq_a12, bool avg_a1n, const int Nf_el[4], double Dnu, bool local, SplitState &st,
                       std::string &log)
{
    if (key == "splitting_a1" || key == "Splitting_a1") {
        no_fix_auto("splitting_a1", prior);
        Snlm.fill("Splitting_a1", prior, row[0], row, 0, 1);
        if (!a11_eq_a12 && avg_a1n) Snlm.fill(Snlm.names[0], Snlm.prior_names[0], Snlm.inputs[0], row, 6, 1);
        if (a11_eq_a12 && !avg_a1n) {
            for (int kk = 0; kk < Nf_el[1]; kk++) Snlm.fill(Snlm.names[0], Snlm.prior_names[0], Snlm.inputs[0], row, 6 + kk, 1);
            Snlm.fill("Empty", "Fix", 0, row, 0, 1);
        }
        if (!a11_eq_a12 && !avg_a1n) {
            for (int kk = 0; kk < Nf_el[1] + Nf_el[2]; kk++) Snlm.fill(Snlm.names[0], Snlm.prior_names[0], Snlm.inputs[0], row, 6 + kk, 1);
            Snlm.fill("Empty", "Fix", 0, row, 0, 1);
        }
    }
    if (key == "asphericity_eta" || key == "Asphericity_eta") {
        Snlm.names[1] = "Asphericity_eta";
        if (prior == "Fix_Auto") {                        // centrifugal distortion, fixed
            Snlm.prior_names[1] = "Fix";
            Snlm.relax[1] = 0;
            if (row[0] == 1) {
                if (Snlm.inputs[0] == EMPTY)
                    throw Fail{TAMCMC_IO_E_SYNTAX, "the keyword 'asphericity_eta' must appear after the keyword splitting_a1"};
                if (!local || Dnu > 0) Snlm.inputs[1] = centrifugal_eta(Snlm.inputs[0], rho_from_dnu(Dnu));
                else { Snlm.inputs[1] = 0; log += "Centrifugal force set to 0 because the model file gives no Dnu\n"; }
            } else Snlm.inputs[1] = 0;
        } else Snlm.fill("Asphericity_eta", prior, row[0], row, 1, 1);
    }
    if (key == "splitting_a3" || key == "Splitting_a3") { no_fix_auto("splitting_a3", prior); Snlm.fill("Splitting_a3", prior, row[0], row, 2, 1); }
    if (key == "asymetry" || key == "Asymetry") { no_fix_auto("asymetry", prior); Snlm.fill("Lorentzian_asymetry", prior, row[0], row, 5, 1); }
    if (key == "inclination" || key == "Inclination") {
        no_fix_auto("inclination", prior);
        const double v = row[0] >= 90 ? 89.99999 : row[0];
        Inc.fill("Inclination", prior, v, row, 0, 1);
    }
    for (int which = 0; which < 2; which++) {
        const std::string name = which == 0 ? "sqrt(splitting_a1).cosi" : "sqrt(splitting_a1).sini";
        if (key != name) continue;
        no_fix_auto(name, prior);
        Snlm.fill(name, prior, row[0], row, 3 + which, 1);
        if (!a11_eq_a12 || !avg_a1n)
            throw Fail{TAMCMC_IO_E_SYNTAX, "models *_a1n* / *_a1l* are not available when fitting sqrt(a1).cosi and sqrt(a1).sini"};
        (which == 0 ? st.a1cosi : st.a1sini) = true;
    }
}

// (Splitting_a1, Inclination) -> (sqrt(a1) cos i, sqrt(a1) sin i): io_ms_global.cpp:881-909 == io_local.cpp:956-986
void project_splitting(Block &Snlm, Block &Inc, std::string &log)
{
    const long double ang = Inc.inputs[0] * PI_L / 180.;
    const double vc = (double)(std::sqrt(Snlm.inputs[0]) * std::cos(ang));
    const double vs = (double)(std::sqrt(Snlm.inputs[0]) * std::sin(ang));
    if (Inc.prior_names[0] == "Fix" && Snlm.prior_names[0] == "Fix") {
        Snlm.fill("sqrt(splitting_a1).cosi", "Fix", vc, Snlm.prior_col(0), 3, 0);
        Snlm.fill("sqrt(splitting_a1).sini", "Fix", vs, Snlm.prior_col(0), 4, 0);
    } else {
        log += "Splitting_a1 / Inclination replaced by sqrt(splitting_a1).cos(i), sqrt(splitting_a1).sin(i) with prior " +
               Snlm.prior_names[0] + "\n";
        Snlm.priors[1][0] = std::sqrt(Snlm.priors[1][0]);      // max of a1 -> max of sqrt(a1)
        Snlm.fill("sqrt(splitting_a1).cosi", Snlm.prior_names[0], vc, Snlm.prior_col(0), 3, 0);
        Snlm.fill("sqrt(splitting_a1).sini", Snlm.prior_names[0], vs, Snlm.prior_col(0), 4, 0);
    }
    if (Snlm.inputs[3] < 1e-2) Snlm.inputs[3] = 1e-2;          // keep away from the edge of uniform priors
    if (Snlm.inputs[4] < 1e-2) Snlm.inputs[4] = 1e-2;
    Inc.fill("Empty", "Fix", 0, Inc.prior_col(0), 0, 1);
    Snlm.fill("Empty", "Fix", 0, Snlm.prior_col(0), 0, 1);
}

void init_snlm(Block &Snlm, bool a11_eq_a12, bool avg_a1n, const int Nf_el[4])   // io_ms_global.cpp:581-601
{
    if (a11_eq_a12 && avg_a1n) Snlm.init(6);
    if (!a11_eq_a12 && avg_a1n) Snlm.init(7);
    if (a11_eq_a12 && !avg_a1n) {
        if (Nf_el[1] != Nf_el[2])
            throw Fail{TAMCMC_IO_E_SYNTAX, "When considering a11=a22 you must have as many l=1 than l=2"};
        Snlm.init(6 + Nf_el[1]);
    }
    if (!a11_eq_a12 && !avg_a1n) Snlm.init(6 + Nf_el[1] + Nf_el[2]);
}

int lmax_of(const ModelFile &mf)
{
    if (mf.els.empty()) throw Fail{TAMCMC_IO_E_SYNTAX, "the .model file lists no mode"};
    int m = mf.els[0];
    for (int e : mf.els) m = e > m ? e : m;
    if (m > 3) throw Fail{TAMCMC_IO_E_SYNTAX, "mode degrees above l=3 are not supported"};
    return m;
}

// Match an eigen-solution frequency to the relax table of its degree (io_ms_global.cpp:470-497)
int unique_relax_slot(const dvec &f_el, double f, double tol)
{
    std::vector<int> p = where_dbl(f_el, f, tol);
    if (p.size() != 1) {
        char b[160];
        snprintf(b, sizeof(b), "The uniqueness of the frequency %g is not respected (%zu matches in the relax list)", f, p.size());
        throw Fail{TAMCMC_IO_E_SYNTAX, b};
    }
    return p[0];
}

void finish_all(InputData &in, double trunc_c, bool do_amp, std::string &log)      // io_ms_global.cpp:1041-1051
{
    int p0 = 0;
    for (int k = 0; k < 10; k++) p0 += in.plength[k];
    in.all.fill("Truncation parameter", "Fix", trunc_c, dvec(5, EMPTY), p0, 1);
    if (in.all.inputs[p0] <= 0) {
        log += "Warning: trunc_c <= 0. This is forbidden. Setting default to 10000. (No truncation)\n";
        in.all.inputs[p0] = 10000.;
    }
    in.all.fill("Switch for fit of Amplitudes or Heights", "Fix", do_amp ? 1. : 0., dvec(5, EMPTY), p0 + 1, 1);
}

// set_noise_params, io_ms_global.cpp:1091-1184
void set_noise_params(Block &N, const std::vector<dvec> &s2, const dvec &noise_params, std::string &log)
{
    for (int k = 0; k < 3; k++) {
        N.names[3 * k] = "Harvey-Noise_H"; N.names[3 * k + 1] = "Harvey-Noise_tc"; N.names[3 * k + 2] = "Harvey-Noise_p";
    }
    N.names[9] = "White_Noise_N0";
    for (int i = 0; i < 6; i++) { N.prior_names[i] = "Fix"; N.relax[i] = 0; }
    for (int i = 6; i < 10; i++) { N.prior_names[i] = "Gaussian"; N.relax[i] = 1; }
    N.inputs = noise_params;
    for (int k = 0; k < 3; k++)
        if (N.inputs[3 * k] <= 0 || N.inputs[3 * k + 1] <= 0 || N.inputs[3 * k + 2] <= 0) {   // no Harvey profile k
            for (int j = 0; j < 3; j++) { N.prior_names[3 * k + j] = "Fix"; N.relax[3 * k + j] = 0; }
            N.inputs[3 * k] = 0; N.inputs[3 * k + 1] = 0; N.inputs[3 * k + 2] = 1;
        }
    for (int i = 6; i < 10; i++) N.priors[0][i] = s2[i][0];
    N.priors[1][6] = (s2[6][1] + s2[6][2]) * 3. / 2;
    N.priors[1][7] = (s2[7][1] + s2[7][2]) * 3. / 2;
    if (s2[8][1] != 0) N.priors[1][8] = (s2[8][1] + s2[8][2]) * 3. / 2;
    else N.priors[1][8] = N.priors[0][8] * 0.1;
    if (N.prior_names[9] == "Uniform") N.priors[1][9] = s2[9][1] + s2[9][2];
    else N.priors[1][9] = N.priors[0][9] * 0.1;
    const double floor_rel[4] = {0.05, 0.005, 0.05, 0.0005};
    for (int i = 6; i < 10; i++)
        if (N.priors[1][i] / N.priors[0][i] <= floor_rel[i - 6] && N.prior_names[i] != "Fix") {
            N.priors[1][i] = N.priors[0][i] * floor_rel[i - 6];
            log += "Warning: relative uncertainty of " + N.names[i] + " too small in the .model file: floor applied\n";
        }
}

// set_width_App2016_params_v1 / _v2, io_ms_global.cpp:1227-1321
void set_width_appourchaux(Block &W, double numax, int version)
{
    dvec out;
    if (version == 2) out.push_back(numax);
    out.push_back(numax);                                                   // nudip
    out.push_back(4. / 2150. * numax + (1. - 1000. * 4. / 2150.));         // alpha
    out.push_back(0.8 / 2150. * numax + (4.5 - 1000. * 0.8 / 2150.));      // Gamma_alpha
    out.push_back(3400. / 2150. * numax + (1000. - 1000. * 3400. / 2150.)); // Wdip
    out.push_back(2.8 / 2200. * numax + (1. - 2.8 / 2200. * 1.));          // DeltaGammadip
    const char *n1[5] = {"nudip", "alpha", "Gamma_alpha", "Wdip", "DeltaGammadip"};
    const char *n2[6] = {"numax", "nudip", "alpha", "Gamma_alpha", "Wdip", "DeltaGammadip"};
    const double f1[5] = {0.1, 0.2, 0.2, 0.2, 0.4}, f2[6] = {0.1, 0.1, 0.2, 0.2, 0.2, 0.4};
    for (size_t i = 0; i < out.size(); i++) {
        dvec pr(4, EMPTY);
        pr[0] = out[i];
        pr[1] = out[i] * (version == 1 ? f1[i] : f2[i]);
        const std::string nm = std::string("width:Appourchaux_v") + (version == 1 ? "1:" : "2:") + (version == 1 ? n1[i] : n2[i]);
        W.fill(nm, "Gaussian", out[i], pr, (int)i, 0);
    }
}

// ---------------------------------------------------------------- build_init_MS_Global, io_ms_global.cpp:308-1088
InputData build_init_ms_global(const ModelFile &mf, double resol, std::string &log)
{
    const double Hmin = 1, Hmax = 10000, tol = 1e-2;
    InputData in;
    bool a11_eq_a12 = true, avg_a1n = true, do_amp = false;
    int width_app = 0;
    double trunc_c = -1, numax = mf.numax;
    const int lmax = lmax_of(mf);
    const size_t NC = mf.common_names.size();

    in.model_fullname = " ";
    for (size_t i = 0; i < NC; i++) {
        if (mf.common_names[i] == "model_fullname") {
            in.model_fullname = mf.common_priors[i];
            const std::string &m = in.model_fullname;
            if (m == "model_MS_Global_a1l_etaa3_HarveyLike") { a11_eq_a12 = false; avg_a1n = true; }
            else if (m == "model_MS_Global_a1n_etaa3_HarveyLike") { a11_eq_a12 = true; avg_a1n = false; }
            else if (m == "model_MS_Global_a1nl_etaa3_HarveyLike") { a11_eq_a12 = false; avg_a1n = false; }
            else if (m == "model_MS_Global_a1etaa3_HarveyLike_Classic" || m == "model_MS_Global_a1etaa3_HarveyLike_Classic_v2" ||
                     m == "model_MS_Global_a1etaa3_HarveyLike_Classic_v3" || m == "model_MS_Global_a1etaa3_HarveyLike" ||
                     m == "model_MS_Global_a1etaa3_Harvey1985" || m == "model_MS_Global_a1acta3_HarveyLike" ||
                     m == "model_MS_Global_a1acta3_Harvey1985") { a11_eq_a12 = true; avg_a1n = true; }
            if (m == "model_MS_Global_a1etaa3_AppWidth_HarveyLike_v1" || m == "model_MS_Global_a1etaa3_AppWidth_HarveyLike_v2") {
                a11_eq_a12 = true; avg_a1n = true;
                width_app = m == "model_MS_Global_a1etaa3_AppWidth_HarveyLike_v1" ? 1 : 2;
                if (numax != EMPTY && numax <= 0)
                    throw Fail{TAMCMC_IO_E_SYNTAX, "The model " + m + " takes an optional numax: give a positive one or -9999"};
            }
        }
        if (mf.common_names[i] == "fit_squareAmplitude_instead_Height") {
            need_prior("fit_squareAmplitude_instead_Height", mf.common_priors[i], "bool");
            do_amp = mf.modes_common[i][0] != 0;
        }
    }
    if (in.model_fullname == " ")
        throw Fail{TAMCMC_IO_E_SYNTAX, "Model name empty. Check that the .model file contains the model_fullname variable."};

    Block Vis, Inc, Snlm, Noise, freq, height, width;
    Vis.init(lmax);
    Inc.init(1);

    // ---- frequencies / widths / heights per degree (io_ms_global.cpp:437-508)
    int Nf_el[4] = {0, 0, 0, 0};
    dvec f_inputs, h_inputs, w_inputs, f_pmin, f_pmax;
    std::vector<bool> f_relax, h_relax, w_relax;
    for (int el = 0; el <= lmax; el++) {
        dvec f_el; std::vector<bool> rf, rw, rh;
        for (size_t i = 0; i < mf.els.size(); i++)
            if (mf.els[i] == el) { f_el.push_back(mf.freqs_ref[i]); rf.push_back(mf.relax_freq[i]); rw.push_back(mf.relax_gamma[i]); rh.push_back(mf.relax_H[i]); }
        for (size_t r = 0; r < mf.eigen_params.size(); r++) {
            const dvec &e = mf.eigen_params[r];
            if ((int)e[0] != el) continue;
            Nf_el[el]++;
            f_inputs.push_back(e[1]); f_pmin.push_back(e[2]); f_pmax.push_back(e[3]);
            if (el == 0) { w_inputs.push_back(e[4]); h_inputs.push_back(e[5]); }      // l>0 columns are ignored
            const int p = unique_relax_slot(f_el, e[1], tol);
            f_relax.push_back(rf[p]);
            if (el == 0) { w_relax.push_back(rw[p]); h_relax.push_back(rh[p]); }
        }
    }

    std::string name_h;
    if (do_amp) {
        name_h = "Amplitude_l0";
        for (size_t i = 0; i < h_inputs.size(); i++) h_inputs[i] = (double)(PI_L * w_inputs[i] * h_inputs[i]);
    } else name_h = "Height_l0";

    height.init((int)h_relax.size());
    if (width_app == 0) width.init((int)w_relax.size());
    if (width_app == 1) width.init(5);
    if (width_app == 2) width.init(6);
    freq.init((int)f_relax.size());

    dvec pv(4);
    pv = {Hmin, Hmax, EMPTY, EMPTY};                                                  // default heights
    for (size_t i = 0; i < h_inputs.size(); i++) height.fill(name_h, h_relax[i] ? "Jeffreys" : "Fix", h_inputs[i], pv, (int)i, 0);
    pv = {resol, mf.Dnu / 3., EMPTY, EMPTY};                                          // default widths
    // quirk: the default width slots receive the HEIGHT values (io_ms_global.cpp:553-559); a `Width` keyword repairs it
    for (size_t i = 0; i < w_inputs.size() && (int)i < width.size(); i++)
        width.fill("Width_l0", w_relax[i] ? "Jeffreys" : "Fix", h_inputs[i], pv, (int)i, 0);
    for (size_t i = 0; i < f_inputs.size(); i++) {                                    // default frequencies: GUG
        if (f_relax[i]) pv = {f_pmin[i], f_pmax[i], 0.01 * mf.Dnu, 0.01 * mf.Dnu};
        freq.fill("Frequency_l", f_relax[i] ? "GUG" : "Fix", f_inputs[i], pv, (int)i, 0);
    }

    if (numax <= 0) {                                                                 // io_ms_global.cpp:572-607
        const double vis_default[4] = {1., 1.5, 0.53, 0.08};
        double num = 0, Htot = 0;
        size_t c = 0;
        for (int el = 0; el <= 3; el++)
            for (int k = 0; k < Nf_el[el]; k++, c++) {
                const double H = (size_t)k < (size_t)height.size() ? height.inputs[k] * vis_default[el] : 0.0;
                num += f_inputs[c] * H; Htot += H;
            }
        numax = num / Htot;
        char b[96]; snprintf(b, sizeof(b), "numax not provided: height-weighted mean frequency used, numax = %.10g\n", numax);
        log += b;
    }

    init_snlm(Snlm, a11_eq_a12, avg_a1n, Nf_el);

    in.extra_priors[0] = 1; in.extra_priors[1] = 2.; in.extra_priors[2] = 0.2; in.extra_priors[3] = 0;

    SplitState st;
    for (size_t i = 0; i < NC; i++) {
        const std::string &key = mf.common_names[i], &prior = mf.common_priors[i];
        const dvec &row = mf.modes_common[i];
        if (key == "freq_smoothness" || key == "Freq_smoothness") {
            need_prior("freq_smoothness", prior, "bool");
            in.extra_priors[0] = row[0]; in.extra_priors[1] = row[1];
        }
        if (key == "trunc_c") { need_prior("trunc_c", prior, "Fix"); trunc_c = row[0]; }
        if (key == "Frequency" || key == "frequency") {
            if (prior != "GUG" && prior != "Uniform") throw Fail{TAMCMC_IO_E_SYNTAX, key + " should always be defined as 'GUG or Uniform'"};
            for (size_t p = 0; p < f_inputs.size(); p++) {
                if (f_relax[p]) {
                    if (prior == "GUG") pv = {f_pmin[p], f_pmax[p], row[3], row[4]};
                    else pv = {f_pmin[p], f_pmax[p], EMPTY, EMPTY};
                    freq.fill("Frequency_l", prior, f_inputs[p], pv, (int)p, 0);
                } else freq.fill("Frequency_l", "Fix", f_inputs[p], pv, (int)p, 0);
            }
        }
        if (key == "height" || key == "Height" || key == "amplitude" || key == "Amplitude") {
            no_fix_auto(key, prior);
            for (size_t p = 0; p < h_inputs.size(); p++)
                height.fill(name_h, h_relax[p] ? prior : "Fix", h_inputs[p], row, (int)p, h_relax[p] ? 0 : 1);
        }
        if ((key == "width" || key == "Width") && width_app == 0) {
            std::string pr = prior;
            dvec wv = row;
            if (prior == "Fix_Auto") { pr = "Jeffreys"; wv = {resol, mf.Dnu / 3., EMPTY, EMPTY}; }
            for (size_t p = 0; p < w_inputs.size(); p++) {
                if (w_relax[p]) width.fill("Width_l", pr, w_inputs[p], wv, (int)p, 0);
                else width.fill("Width_l", "Fix", w_inputs[p], row, (int)p, 1);
            }
        }
        if ((key == "width" || key == "Width") && width_app == 1)
            log += "Width keyword ignored: the model fits the Appourchaux+2016 width relation\n";
        splitting_keyword(key, prior, row, Snlm, Inc, a11_eq_a12, avg_a1n, Nf_el, mf.Dnu, false, st, log);
        for (int v = 1; v <= 3; v++) {
            const std::string lo = "visibility_l" + std::to_string(v), up = "Visibility_l" + std::to_string(v);
            if (key != lo && key != up) continue;
            no_fix_auto(lo, prior);
            if (lmax >= v) Vis.fill(up, prior, row[0], row, v - 1, 1);
            else log += "Warning: lmax < " + std::to_string(v) + " but keyword '" + lo + "' detected: ignored\n";
        }
    }

    if (st.a1cosi != st.a1sini)
        throw Fail{TAMCMC_IO_E_SYNTAX, "Both 'sqrt(splitting_a1).sini' and 'sqrt(splitting_a1).cosi' keywords must appear"};
    const std::string &m = in.model_fullname;
    if (!st.a1cosi) {
        if (m == "model_MS_Global_a1etaa3_HarveyLike" || m == "model_MS_Global_a1etaa3_Harvey1985" ||
            m == "model_MS_Global_a1etaa3_AppWidth_HarveyLike_v1" || m == "model_MS_Global_a1etaa3_AppWidth_HarveyLike_v2")
            project_splitting(Snlm, Inc, log);
        if (m == "model_MS_Global_a1etaa3_HarveyLike_Classic_v2") {                 // io_ms_global.cpp:926-940
            const double inc0 = Inc.inputs[0];
            Inc.init(9);
            int ind = 0;
            pv = {0, 1, EMPTY, EMPTY};
            for (int el = 1; el <= lmax; el++) {
                dvec r = amplitude_ratio(el, inc0);
                for (int em = 0; em <= el; em++)
                    Inc.fill("Inc:H" + std::to_string(el) + "," + std::to_string(em), "Uniform", r[el + em], pv, ind++, 0);
            }
            in.extra_priors[3] = 1;
        }
        if (m == "model_MS_Global_a1etaa3_HarveyLike_Classic_v3") {                 // io_ms_global.cpp:941-973
            log += "WARNING: model_MS_Global_a1etaa3_HarveyLike_Classic_v3 may have too many height parameters for a global fit\n";
            const double inc0 = Inc.inputs[0];
            const dvec vis0 = Vis.inputs;
            pv = {EMPTY, EMPTY, EMPTY, EMPTY};
            for (int el = 1; el < lmax; el++) Vis.fill("Empty", "Fix", 0, pv, el - 1, 0);   // quirk: `el < lmax`, the last one stays
            Inc.init(Nf_el[1] * 2 + Nf_el[2] * 3 + Nf_el[3] * 4);
            int ind = 0;
            pv = {Hmin, Hmax, EMPTY, EMPTY};
            for (int el = 1; el <= lmax; el++) {
                dvec r = amplitude_ratio(el, inc0);
                for (int en = 0; en < Nf_el[el]; en++)
                    for (int em = 0; em <= el; em++) {
                        if (en >= height.size()) throw Fail{TAMCMC_IO_E_SYNTAX, "Classic_v3 needs at least as many l=0 as l>0 modes"};
                        Inc.fill("Inc: H" + std::to_string(en) + "," + std::to_string(el) + "," + std::to_string(em), "Jeffreys",
                                 height.inputs[en] * vis0[el - 1] * r[el + em], pv, ind++, 0);
                    }
            }
            in.extra_priors[3] = 2;
        }
    } else {
        if (m == "model_MS_Global_a1etaa3_HarveyLike_Classic")
            throw Fail{TAMCMC_IO_E_SYNTAX, "We cannot use " + m + " with variables sqrt(splitting_a1).cosi and sqrt(splitting_a1).sini"};
        Inc.fill("Empty", "Fix", 0, Inc.prior_col(0), 0, 1);
        Snlm.fill("Empty", "Fix", 0, Snlm.prior_col(0), 0, 1);
    }

    Noise.init(10);
    set_noise_params(Noise, mf.noise_s2, mf.noise_params, log);

    int *pl = in.plength;
    pl[0] = (int)h_inputs.size(); pl[1] = lmax; pl[2] = Nf_el[0]; pl[3] = Nf_el[1]; pl[4] = Nf_el[2]; pl[5] = Nf_el[3];
    pl[6] = Snlm.size(); pl[7] = (int)w_inputs.size(); pl[8] = Noise.size(); pl[9] = Inc.size(); pl[10] = 2;
    if (width_app == 1) set_width_appourchaux(width, numax, 1);
    if (width_app == 2) set_width_appourchaux(width, numax, 2);
    // quirk: plength[7] stays the number of l=0 widths even for the Appourchaux models, whose 5 (v1) or 6 (v2)
    // parameters occupy the first slots of that block; the rest stay "Empty"/Fix/0 (io_ms_global.cpp:993-1031).
    // With fewer l=0 modes than that the reference writes past the block: refused here.
    if (width_app != 0 && width.size() > pl[7])
        throw Fail{TAMCMC_IO_E_SYNTAX, "Appourchaux width models need at least " + std::to_string(width.size()) +
                                       " l=0 modes (plength[7] is the number of l=0 widths)"};
    int total = 0;
    for (int k = 0; k < 11; k++) total += pl[k];
    in.all.init(total);
    int p0 = 0;
    in.all.add(height, p0); p0 += pl[0];
    in.all.add(Vis, p0); p0 += pl[1];
    in.all.add(freq, p0); p0 += pl[2] + pl[3] + pl[4] + pl[5];
    in.all.add(Snlm, p0); p0 += pl[6];
    in.all.add(width, p0); p0 += pl[7];
    in.all.add(Noise, p0); p0 += pl[8];
    in.all.add(Inc, p0);
    finish_all(in, trunc_c, do_amp, log);
    return in;
}

// ---------------------------------------------------------------- build_init_local, io_local.cpp:304-1158
InputData build_init_local(const ModelFile &mf, double resol, std::string &log)
{
    const double Hmin = 1, Hmax = 10000, tol = 1e-2;
    InputData in;
    const bool a11_eq_a12 = true, avg_a1n = true;
    bool do_amp = false;
    double trunc_c = -1;
    int pos_prior_height = -1;
    const int lmax = lmax_of(mf);
    const size_t NC = mf.common_names.size();

    in.model_fullname = " ";
    for (size_t i = 0; i < NC; i++) {
        if (mf.common_names[i] == "model_fullname") in.model_fullname = mf.common_priors[i];
        if (mf.common_names[i] == "fit_squareAmplitude_instead_Height") {
            need_prior("fit_squareAmplitude_instead_Height", mf.common_priors[i], "bool");
            do_amp = mf.modes_common[i][0] != 0;
        }
    }
    if (in.model_fullname == " ")
        throw Fail{TAMCMC_IO_E_SYNTAX, "Model name empty. Check that the .model file contains the model_fullname variable."};
    const bool hnlm = in.model_fullname == "model_MS_local_Hnlm";

    Block Inc, Snlm, Noise, freq, height, width;
    Inc.init(1);

    // ---- per-degree lists (io_local.cpp:386-478), then keep only the modes strictly inside the slice (:483-540)
    dvec f[4], h[4], w[4], fmin[4], fmax[4];
    std::vector<bool> fr[4], hr[4], wr[4];
    for (int el = 0; el <= lmax; el++) {
        dvec f_el; std::vector<bool> rf, rw, rh;
        for (size_t i = 0; i < mf.els.size(); i++)
            if (mf.els[i] == el) { f_el.push_back(mf.freqs_ref[i]); rf.push_back(mf.relax_freq[i]); rw.push_back(mf.relax_gamma[i]); rh.push_back(mf.relax_H[i]); }
        if (f_el.empty()) continue;
        for (size_t r = 0; r < mf.eigen_params.size(); r++) {
            const dvec &e = mf.eigen_params[r];
            if ((int)e[0] != el) continue;
            const int p = unique_relax_slot(f_el, e[1], tol);
            if (!(e[1] > mf.fmin && e[1] < mf.fmax)) continue;                        // filter_range(..., strict)
            f[el].push_back(e[1]); fmin[el].push_back(e[2]); fmax[el].push_back(e[3]); w[el].push_back(e[4]); h[el].push_back(e[5]);
            fr[el].push_back(rf[p]); wr[el].push_back(rw[p]); hr[el].push_back(rh[p]);
        }
    }
    int Nf_el[4];
    int Ntot = 0;
    for (int el = 0; el < 4; el++) { Nf_el[el] = (int)f[el].size(); Ntot += Nf_el[el]; }
    if (Ntot == 0)
        throw Fail{TAMCMC_IO_E_RANGE, "No parameters found in the specified frequency range!"};

    std::string name_h;
    if (do_amp) {
        name_h = "Amplitude_l";
        for (int el = 0; el < 4; el++)
            for (size_t i = 0; i < h[el].size(); i++) h[el][i] = (double)(PI_L * w[el][i] * h[el][i]);
    } else name_h = "Height_l";

    dvec pv = {Hmin, Hmax, EMPTY, EMPTY};
    int p0 = 0;
    if (hnlm) {
        height.init(Nf_el[0] + Nf_el[1] * 2 + Nf_el[2] * 3 + Nf_el[3] * 4);
        height.fill_vect(h[0], hr[0], name_h, "Jeffreys", pv, 0, 0, 0);               // l>0 wait for the inclination
    } else {
        height.init(Ntot);
        for (int el = 0, p = 0; el < 4; p += Nf_el[el], el++) { p0 = p; height.fill_vect(h[el], hr[el], name_h, "Jeffreys", pv, p, 0, 0); }
    }
    width.init(Ntot);
    freq.init(Ntot);
    if (mf.Dnu > 0) pv = {resol, mf.Dnu / 3., EMPTY, EMPTY};
    else { pv = {resol, 20., EMPTY, EMPTY}; log += "Warning: No large separation provided. The default maximum width will be set to 20 microHz\n"; }
    for (int el = 0, p = 0; el < 4; p += Nf_el[el], el++) { p0 = p; width.fill_vect(w[el], wr[el], "Width_l", "Jeffreys", pv, p, 0, 0); }
    for (int el = 0, p = 0; el < 4; p += Nf_el[el], el++) {                            // frequencies: GUG inside the window
        p0 = p;
        for (size_t i = 0; i < f[el].size(); i++) {
            if (fr[el][i]) {
                const double s = 0.01 * std::fabs(fmax[el][i] - fmin[el][i]);
                pv = {fmin[el][i], fmax[el][i], s, s};
                freq.fill("Frequency_l", "GUG", f[el][i], pv, (int)i + p, 0);
            } else freq.fill("Frequency_l", "Fix", f[el][i], pv, (int)i + p, 0);
        }
    }
    init_snlm(Snlm, a11_eq_a12, avg_a1n, Nf_el);

    in.extra_priors[0] = 0; in.extra_priors[1] = 0; in.extra_priors[2] = 0.2; in.extra_priors[3] = 0;

    SplitState st;
    for (size_t i = 0; i < NC; i++) {
        const std::string &key = mf.common_names[i], &prior = mf.common_priors[i];
        const dvec &row = mf.modes_common[i];
        if (key == "freq_smoothness" || key == "Freq_smoothness") log += "freq_smoothness is irrelevant for a local fit: skipped\n";
        if (key == "Visibility_l1" || key == "Visibility_l2" || key == "Visibility_l3") log += key + " is irrelevant for a local fit (heights are fitted directly): skipped\n";
        if (key == "trunc_c") { need_prior("trunc_c", prior, "Fix"); trunc_c = row[0]; }
        if (key == "height" || key == "Height" || key == "amplitude" || key == "Amplitude") {
            const bool amp = key == "amplitude" || key == "Amplitude";
            if (prior == "Fix_Auto") {                                               // io_local.cpp:728-781
                auto bounds = [&](const dvec &hh, size_t k) {
                    dvec b(4, EMPTY);
                    if (amp) { b[0] = (double)(PI_L * mf.Dnu / 3. * hh[k] / row[0]); b[1] = (double)(PI_L * mf.Dnu / 3. * hh[k] * row[1]); }
                    else { b[0] = hh[k] / row[0]; b[1] = hh[k] * row[1]; }
                    return b;
                };
                if (hnlm) {
                    pos_prior_height = (int)i;
                    for (size_t k = 0; k < h[0].size(); k++) height.fill(name_h, hr[0][k] ? "Jeffreys" : "Fix", h[0][k], bounds(h[0], k), (int)k + p0, 0);
                } else {
                    for (int el = 0, p = 0; el < 4; p += Nf_el[el], el++) {
                        p0 = p;
                        for (size_t k = 0; k < h[el].size(); k++) height.fill(name_h, hr[el][k] ? "Jeffreys" : "Fix", h[el][k], bounds(h[el], k), (int)k + p, 0);
                    }
                }
            } else if (hnlm) {
                pos_prior_height = (int)i;
                height.fill_vect(h[0], hr[0], name_h, prior, row, p0, 0, 0);           // quirk: p0 is whatever was set last
            } else {
                for (int el = 0, p = 0; el < 4; p += Nf_el[el], el++) { p0 = p; height.fill_vect(h[el], hr[el], name_h, prior, row, p, 1, 1); }
            }
        }
        if (key == "width" || key == "Width") {
            std::string pr = prior;
            dvec wv = row;
            if (prior == "Fix_Auto") { pr = "Jeffreys"; wv = {resol, mf.Dnu > 0 ? mf.Dnu / 3 : 20., EMPTY, EMPTY}; }
            for (int el = 0, p = 0; el < 4; p += Nf_el[el], el++) { p0 = p; width.fill_vect(w[el], wr[el], "Width_l", pr, wv, p, 0, 1); }
        }
        // the shared keywords assign p0 too (io_local.cpp:826-905); the values matter only for the Hnlm quirk above
        if (key == "splitting_a1" || key == "Splitting_a1") p0 = 0;
        if ((key == "asphericity_eta" || key == "Asphericity_eta") && prior != "Fix_Auto") p0 = 1;
        if (key == "splitting_a3" || key == "Splitting_a3") p0 = 2;
        if (key == "asymetry" || key == "Asymetry") p0 = 5;
        if (key == "inclination" || key == "Inclination") p0 = 0;
        if (key == "sqrt(splitting_a1).cosi") p0 = 3;
        if (key == "sqrt(splitting_a1).sini") p0 = 4;
        splitting_keyword(key, prior, row, Snlm, Inc, a11_eq_a12, avg_a1n, Nf_el, mf.Dnu, true, st, log);
    }

    if (st.a1cosi != st.a1sini)
        throw Fail{TAMCMC_IO_E_SYNTAX, "Both 'sqrt(splitting_a1).sini' and 'sqrt(splitting_a1).cosi' keywords must appear"};
    if (!st.a1cosi) {
        if (in.model_fullname == "model_MS_local_basic") project_splitting(Snlm, Inc, log);
        if (hnlm) {                                                                   // io_local.cpp:995-1034
            const double inc0 = Inc.inputs[0];
            Inc.fill("Empty", "Fix", 0, Inc.prior_col(0), 0, 1);
            int ind = (int)h[0].size();
            std::string pr = "Jeffreys";
            pv = {Hmin, Hmax, EMPTY, EMPTY};
            if (pos_prior_height > 0) { pv = mf.modes_common[pos_prior_height]; pr = mf.common_priors[pos_prior_height]; }
            for (int el = 1; el <= lmax; el++) {
                dvec r = amplitude_ratio(el, inc0);
                for (int en = 0; en < Nf_el[el]; en++)
                    for (int em = 0; em <= el; em++)
                        height.fill("H(" + std::to_string(en) + "," + std::to_string(el) + "," + std::to_string(em) + ")", pr,
                                    h[el][en] * r[el + em], pv, ind++, 0);
            }
            in.extra_priors[3] = 2;
        }
    } else {
        Inc.fill("Empty", "Fix", 0, Inc.prior_col(0), 0, 1);
        Snlm.fill("Empty", "Fix", 0, Snlm.prior_col(0), 0, 1);
    }

    // ---- noise: one flat level over the slice (set_noise_params_local, io_local.cpp:1160-1220)
    Noise.init(1);
    {
        Noise.names[0] = "White_Noise_N0"; Noise.prior_names[0] = "Uniform"; Noise.relax[0] = 1;
        size_t n_skip = 0;
        for (double v : mf.noise_params) if (v > -2 - 1e-6 && v < -2 + 1e-6) n_skip++;
        dvec np(mf.noise_params.size() - n_skip, EMPTY);
        size_t c = 0;
        for (double v : mf.noise_params) {
            if (c >= np.size()) break;
            if (v == -1) np[c++] = 0;
            else if (v >= 0) np[c++] = v;
        }
        const int Nh = ((int)np.size() - 1) / 3;
        dvec vals = harvey_like(np, dvec{mf.fmin, mf.fmax}, Nh);
        Noise.inputs[0] = (vals[0] + vals[1]) / 2.;
        Noise.priors[0][0] = (vals[0] < vals[1] ? vals[0] : vals[1]) * 0.5;
        Noise.priors[1][0] = (vals[0] > vals[1] ? vals[0] : vals[1]) * 1.5;
    }

    int *pl = in.plength;
    pl[0] = hnlm ? Nf_el[0] + 2 * Nf_el[1] + 3 * Nf_el[2] + 4 * Nf_el[3] : Ntot;
    pl[1] = 0; pl[2] = Nf_el[0]; pl[3] = Nf_el[1]; pl[4] = Nf_el[2]; pl[5] = Nf_el[3];
    pl[6] = Snlm.size(); pl[7] = Ntot; pl[8] = Noise.size(); pl[9] = Inc.size(); pl[10] = 2;
    int total = 0;
    for (int k = 0; k < 11; k++) total += pl[k];
    in.all.init(total);
    p0 = 0;
    in.all.add(height, p0); p0 += pl[0] + pl[1];
    in.all.add(freq, p0); p0 += pl[2] + pl[3] + pl[4] + pl[5];
    in.all.add(Snlm, p0); p0 += pl[6];
    in.all.add(width, p0); p0 += pl[7];
    in.all.add(Noise, p0); p0 += pl[8];
    in.all.add(Inc, p0);
    finish_all(in, trunc_c, do_amp, log);
    return in;
}

// ---------------------------------------------------------------- Config::read_data_ascii_Ncols, config.cpp:519-647
struct DataFile {
    std::vector<std::string> header, labels, units;
    std::vector<dvec> rows;
    size_t ncols = 0;
};

DataFile read_data_file(const std::string &path)
{
    std::ifstream f(path.c_str());
    if (!f.is_open()) throw Fail{TAMCMC_IO_E_OPEN, "Could not open the data file: " + path};
    DataFile d;
    std::string line;
    std::getline(f, line);
    line = trim(trim_eol(line));
    if (first_char(line) == "#") {
        while (first_char(line) == "#" && !f.eof()) {
            d.header.push_back(trim(line.substr(1)));
            std::getline(f, line);
            line = trim(trim_eol(line));
        }
    } else d.header.push_back("");
    if (first_char(line) == "!") {
        d.labels = split(trim(line.substr(1)), " \t");
        std::getline(f, line);
        line = trim(trim_eol(line));
    } else d.labels.assign(5, "");
    bool have_units = false;
    if (first_char(line) == "*") { d.units = split(trim(line.substr(1)), " \t"); have_units = true; }
    else d.units.assign(5, "");
    if ((!d.labels.empty() && d.labels[0] != "") || have_units) std::getline(f, line);
    const size_t max_rows = 1000000;                                                  // data_Maxsize, config.cpp:531
    while (!f.eof()) {          // quirk kept: a last line without a trailing newline is never parsed
        std::vector<std::string> w = split(trim(trim_eol(line)), " \t");
        if (d.rows.size() >= max_rows) throw Fail{TAMCMC_IO_E_RANGE, "data file has more than 1000000 rows"};
        dvec r(w.size());
        for (size_t i = 0; i < w.size(); i++) {
            long double v;
            if (!(std::istringstream(w[i]) >> v)) v = std::nan("");
            r[i] = (double)v;
        }
        if (!r.empty()) { d.rows.push_back(r); d.ncols = r.size(); }
        std::getline(f, line);
    }
    return d;
}

// ---------------------------------------------------------------- Config::read_listfiles, config.cpp:1763-1813
struct CtrlList {
    std::vector<int> ids;
    std::vector<std::string> names;
    int find(const std::string &n) const
    {
        int out = -9999;                         // the last match wins (config.cpp:419-424)
        for (size_t i = 0; i < names.size(); i++) if (names[i] == n) out = ids[i];
        return out;
    }
};

CtrlList read_list_file(const std::string &path)
{
    std::ifstream f(path.c_str());
    if (!f.is_open()) throw Fail{TAMCMC_IO_E_OPEN, "Could not open the list file: " + path};
    CtrlList L;
    std::string line;
    while (std::getline(f, line)) {
        line = trim(trim_eol(line));
        if (line.empty() || line[0] == '#') continue;
        std::vector<std::string> w = split(line, " \t");
        if (w.size() < 2) continue;
        L.ids.push_back((int)to_long(w[0]));
        L.names.push_back(w[1]);
    }
    return L;
}

// ---------------------------------------------------------------- Config::read_cfg_file, config.cpp:810-1320
struct KeySpec { const char *group, *key; };
const KeySpec KNOWN_KEYS[] = {
    {"MALA", "target_acceptance"}, {"MALA", "c0"}, {"MALA", "epsilon1"}, {"MALA", "epsilon2"}, {"MALA", "A1"},
    {"MALA", "Nt_learn"}, {"MALA", "periods_learn"}, {"MALA", "use_drift"}, {"MALA", "delta"}, {"MALA", "delta_x"},
    {"MALA", "Nchains"}, {"MALA", "dN_mixing"}, {"MALA", "lambda_temp"}, {"MALA", "proposal_type"},
    {"Modeling", "model_fct_name"}, {"Modeling", "prior_fct_name"}, {"Modeling", "likelihood_fct_name"},
    {"Modeling", "likelihood_params"}, {"Modeling", "cfg_model_file"},
    {"Data", "verbose_data"}, {"Data", "file_data"}, {"Data", "x_col"}, {"Data", "y_col"}, {"Data", "ysig_col"},
    {"Outputs", "Nsamples"}, {"Outputs", "Nbuffer"}, {"Outputs", "erase_old_files"}, {"Outputs", "file_format"},
    {"Outputs", "get_params"}, {"Outputs", "get_statcriteria"}, {"Outputs", "get_proposal_params"},
    {"Outputs", "get_parallel_tempering"}, {"Outputs", "get_models"}, {"Outputs", "output_root_name"},
    {"Outputs", "output_dir"}, {"Outputs", "params_txt_fileout"}, {"Outputs", "proposal_txt_fileout"},
    {"Outputs", "parallel_tempering_txt_fileout"}, {"Outputs", "model_txt_fileout"}, {"Outputs", "stat_txt_fileout"},
    {"Outputs", "acceptance_txt_fileout"}, {"Outputs", "do_restore_proposal"}, {"Outputs", "do_restore_proposal_mean"},
    {"Outputs", "do_restore_variables"}, {"Outputs", "do_restore_last_index"}, {"Outputs", "restore_dir"},
    {"Outputs", "restore_file_in"}, {"Outputs", "restore_file_out"}, {"Outputs", "do_backup_input_file"},
    {"Outputs", "do_backup_cfg_files"},
    {"Diagnostics", "chains_diags"}, {"Diagnostics", "pdfs_diags"}, {"Diagnostics", "evidence_diags"},
    {"Diagnostics", "output_root_name"}, {"Diagnostics", "output_dir"}, {"Diagnostics", "file_chains_diags"},
    {"Diagnostics", "file_evidence_diags"}, {"Diagnostics", "file_pdfs_diags"}, {"Diagnostics", "model_initial_diags"},
    {"Diagnostics", "model_buffer_diags"}, {"Diagnostics", "model_final_diags"}, {"Diagnostics", "file_model_init_diags"},
    {"Diagnostics", "file_model_buffer_diags"}, {"Diagnostics", "file_model_final_diags"}, {"Diagnostics", "data_scoef1"},
    {"Diagnostics", "data_scoef2"}, {"Diagnostics", "show_original_data"}, {"Diagnostics", "Nclasses"},
    {"Diagnostics", "evidence_interpolation_factor"},
};

bool known_key(const std::string &group, const std::string &key)
{
    for (const KeySpec &k : KNOWN_KEYS) if (group == k.group && key == k.key) return true;
    return false;
}

typedef std::map<std::string, std::string> KeyMap;       // "Group.key" -> value string

void read_cfg_file(const std::string &path, KeyMap &cfg)
{
    std::ifstream f(path.c_str());
    if (!f.is_open()) throw Fail{TAMCMC_IO_E_OPEN, "Could not open the configuration file: " + path};
    std::string line, group;
    while (std::getline(f, line)) {
        line = trim(trim_eol(line));
        if (line.empty() || line[0] == '#') continue;
        if (line[0] == '!') {                                                           // group indicator "!Name:"
            const size_t p = line.find(':');
            if (p == std::string::npos) throw Fail{TAMCMC_IO_E_SYNTAX, "Terminator not found! Each group name must finish by a : symbol: " + line};
            group = trim(line.substr(1, p - 1));
            continue;
        }
        const size_t p = line.find(';');                                                // format_line, config.cpp:649-698
        if (p == std::string::npos) throw Fail{TAMCMC_IO_E_SYNTAX, "Terminator not found! Each uncommented line must finish by a ; symbol: " + line};
        line = trim(line.substr(0, p));
        if (line == "/END") break;
        std::vector<std::string> w = split(line, "=");
        if (w.empty()) continue;
        const std::string key = w[0], val = w.size() > 1 ? w[1] : "";
        if (!known_key(group, key)) throw Fail{TAMCMC_IO_E_NAME, "The keyword " + key + " is not a known keyword of group " + group};
        cfg[group + "." + key] = val;
    }
}

struct ErrorTable { std::vector<std::string> names; dvec frac, offset; };

ErrorTable read_errors_file(const std::string &path)                     // Config::read_defautlerrors, config.cpp:1608-1667
{
    std::ifstream f(path.c_str());
    if (!f.is_open()) throw Fail{TAMCMC_IO_E_OPEN, "Could not open the default-errors file: " + path};
    ErrorTable t;
    std::string line;
    while (std::getline(f, line)) {
        line = trim(trim_eol(line));
        if (line.empty() || line[0] == '#') continue;
        std::vector<std::string> w = split(line, " \t");
        if (w.size() < 3) continue;
        t.names.push_back(trim(w[0])); t.frac.push_back(to_dbl(w[1])); t.offset.push_back(to_dbl(w[2]));
    }
    return t;
}

std::vector<long> to_longs(const std::string &s, const std::string &delims)       // str_to_arrint
{
    std::vector<long> out;
    for (const std::string &w : split(s, delims)) if (trim(w) != "") out.push_back(to_long(w));
    return out;
}

void copy_str(const std::string &s, char *buf, int32_t cap)
{
    if (!buf || cap <= 0) return;
    const size_t n = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1;
    memcpy(buf, s.data(), n);
    buf[n] = 0;
}

} // namespace

// ---------------------------------------------------------------- the handle
struct tamcmc_setup {
    KeyMap cfg;
    ErrorTable errors;
    CtrlList models, priors, likelihoods, primepriors;
    // after load
    bool loaded = false;
    ModelFile mf;
    InputData in;
    std::vector<int> prior_switch;
    dvec x, y, sig;
    std::string xlabel, ylabel, xunit, yunit;
    double resol = 0;
    int model_case = -9999, likelihood_case = -9999, prior_case = -9999;
    mutable std::string error, log;

    std::string get(const std::string &k, const std::string &dflt = "") const
    {
        KeyMap::const_iterator it = cfg.find(k);
        return it == cfg.end() ? dflt : it->second;
    }
};

template <class F> static int guarded(const tamcmc_setup *s, F &&fn)
{
    try { fn(); return TAMCMC_IO_OK; }
    catch (const Fail &e) { if (s) s->error = e.msg; return e.code; }
    catch (const std::exception &e) { if (s) s->error = e.what(); return TAMCMC_IO_E_SYNTAX; }
}

extern "C" int tamcmc_setup_create_files(tamcmc_setup **out, const char *cfg_file, const char *errors_file,
                                         const char *models_list, const char *priors_list, const char *likelihoods_list,
                                         const char *primepriors_list)
{
    if (!out || !cfg_file || !errors_file || !models_list || !priors_list || !likelihoods_list || !primepriors_list)
        return TAMCMC_IO_E_INVALID;
    *out = nullptr;
    tamcmc_setup *s = new tamcmc_setup();
    const int rc = guarded(s, [&] {
        read_cfg_file(cfg_file, s->cfg);
        s->errors = read_errors_file(errors_file);
        s->models = read_list_file(models_list);
        s->priors = read_list_file(priors_list);
        s->likelihoods = read_list_file(likelihoods_list);
        s->primepriors = read_list_file(primepriors_list);
    });
    if (rc != TAMCMC_IO_OK) { fprintf(stderr, "tamcmc_setup_create: %s\n", s->error.c_str()); delete s; return rc; }
    *out = s;
    return TAMCMC_IO_OK;
}

extern "C" int tamcmc_setup_create(tamcmc_setup **out, const char *config_dir)
{
    if (!out || !config_dir) return TAMCMC_IO_E_INVALID;
    const std::string d = std::string(config_dir) + "/";
    return tamcmc_setup_create_files(out, (d + "config_default.cfg").c_str(), (d + "errors_default.cfg").c_str(),
                                     (d + "models_ctrl.list").c_str(), (d + "priors_ctrl.list").c_str(),
                                     (d + "likelihoods_ctrl.list").c_str(), (d + "primepriors_ctrl.list").c_str());
}

extern "C" int tamcmc_setup_destroy(tamcmc_setup *s) { delete s; return TAMCMC_IO_OK; }
extern "C" const char *tamcmc_setup_error(const tamcmc_setup *s) { return s ? s->error.c_str() : "null setup"; }
extern "C" const char *tamcmc_setup_log(const tamcmc_setup *s) { return s ? s->log.c_str() : ""; }

extern "C" int tamcmc_setup_set(tamcmc_setup *s, const char *group, const char *key, const char *value)
{
    if (!s || !group || !key || !value) return TAMCMC_IO_E_INVALID;
    if (!known_key(group, key)) { s->error = std::string("unknown keyword ") + group + "." + key; return TAMCMC_IO_E_NAME; }
    s->cfg[std::string(group) + "." + key] = value;
    return TAMCMC_IO_OK;
}

extern "C" int tamcmc_setup_get(const tamcmc_setup *s, const char *group, const char *key, char *buf, int32_t cap)
{
    if (!s || !group || !key || !buf || cap <= 0) return TAMCMC_IO_E_INVALID;
    KeyMap::const_iterator it = s->cfg.find(std::string(group) + "." + key);
    if (it == s->cfg.end()) { s->error = std::string("keyword not set: ") + group + "." + key; return TAMCMC_IO_E_NAME; }
    if ((int32_t)it->second.size() + 1 > cap) return TAMCMC_IO_E_CAPACITY;
    copy_str(it->second, buf, cap);
    return TAMCMC_IO_OK;
}

extern "C" int tamcmc_setup_apply_phase(tamcmc_setup *s, const char *phase, int64_t Nsamples, double c0)
{
    if (!s || !phase || Nsamples < 1) return TAMCMC_IO_E_INVALID;
    return guarded(s, [&] {
        std::vector<long> Nt = to_longs(s->get("MALA.Nt_learn"), ",");
        if (Nt.empty()) throw Fail{TAMCMC_IO_E_SYNTAX, "MALA.Nt_learn is empty"};
        const std::string p = phase;
        std::string dN = s->get("MALA.dN_mixing", "1");
        if (p == "Burn-in") Nt.back() = (long)Nsamples + 1;                              // never stop learning
        else if (p == "Learning") { Nt.back() = (long)Nsamples + 1; dN = std::to_string(Nsamples + 1); }   // never mix
        else if (p == "Acquire") for (size_t i = 0; i < Nt.size(); i++) Nt[i] = (long)Nsamples + 1 + (long)i;   // never learn
        else throw Fail{TAMCMC_IO_E_NAME, "phase must be Burn-in, Learning or Acquire"};
        std::string v;
        for (size_t i = 0; i < Nt.size(); i++) v += (i ? ", " : "") + std::to_string(Nt[i]);
        s->cfg["MALA.Nt_learn"] = v;
        s->cfg["MALA.dN_mixing"] = dN;
        s->cfg["Outputs.Nsamples"] = std::to_string(Nsamples);
        char b[64]; snprintf(b, sizeof(b), "%.17g", c0);
        s->cfg["MALA.c0"] = b;
    });
}

extern "C" int tamcmc_data_file_read(const char *data_file, double **data, int64_t *nrows, int32_t *ncols)
{
    if (!data_file || !data || !nrows || !ncols) return TAMCMC_IO_E_INVALID;
    *data = nullptr; *nrows = 0; *ncols = 0;
    try {
        DataFile d = read_data_file(data_file);
        if (d.rows.empty()) return TAMCMC_IO_E_SYNTAX;
        double *m = (double *)malloc(sizeof(double) * d.rows.size() * d.ncols);
        if (!m) return TAMCMC_IO_E_CAPACITY;
        for (size_t r = 0; r < d.rows.size(); r++)
            for (size_t c = 0; c < d.ncols; c++) m[r * d.ncols + c] = c < d.rows[r].size() ? d.rows[r][c] : std::nan("");
        *data = m; *nrows = (int64_t)d.rows.size(); *ncols = (int32_t)d.ncols;
        return TAMCMC_IO_OK;
    } catch (const Fail &e) { fprintf(stderr, "tamcmc_data_file_read: %s\n", e.msg.c_str()); return e.code; }
}

extern "C" void tamcmc_buffer_free(void *p) { free(p); }

extern "C" int tamcmc_list_file_lookup(const char *list_file, const char *name, int32_t *id)
{
    if (!list_file || !name || !id) return TAMCMC_IO_E_INVALID;
    try {
        const int v = read_list_file(list_file).find(name);
        if (v == -9999) return TAMCMC_IO_E_NAME;
        *id = v;
        return TAMCMC_IO_OK;
    } catch (const Fail &e) { return e.code; }
}

extern "C" int tamcmc_model_file_slices(const char *model_file, double *ranges, int32_t cap_rows, int32_t *n)
{
    if (!model_file || !n) return TAMCMC_IO_E_INVALID;
    std::ifstream f(model_file);
    if (!f.is_open()) return TAMCMC_IO_E_OPEN;
    std::string line;
    int32_t rows = 0;
    bool started = false;
    while (std::getline(f, line)) {                                                     // main.cpp:404-428
        line = trim(trim_eol(line));
        if (first_char(line) == "*") {
            started = true;
            std::vector<std::string> w = split(line, "= \t");
            if (w.size() < 3) return TAMCMC_IO_E_SYNTAX;
            if (ranges && rows < cap_rows) { ranges[2 * rows] = to_dbl(w[1]); ranges[2 * rows + 1] = to_dbl(w[2]); }
            rows++;
        } else if (started) break;
    }
    *n = rows;
    return (ranges && rows > cap_rows) ? TAMCMC_IO_E_CAPACITY : TAMCMC_IO_OK;
}

extern "C" int tamcmc_setup_load(tamcmc_setup *s, const char *model_file, const char *data_file, int32_t slice_ind)
{
    if (!s || !model_file || !data_file || slice_ind < 0) return TAMCMC_IO_E_INVALID;
    s->loaded = false;
    s->log.clear();
    return guarded(s, [&] {
        const int x_col = (int)to_long(s->get("Data.x_col", "0")), y_col = (int)to_long(s->get("Data.y_col", "1"));
        const int ysig_col = (int)to_long(s->get("Data.ysig_col", "-1"));
        DataFile d = read_data_file(data_file);
        if (x_col < 0 || (size_t)x_col >= d.ncols) throw Fail{TAMCMC_IO_E_SYNTAX, "x_col outside the columns of the data file"};
        if (d.rows.size() < 3) throw Fail{TAMCMC_IO_E_SYNTAX, "the data file needs at least 3 rows"};
        s->resol = d.rows[2][x_col] - d.rows[1][x_col];                                  // config.cpp:333,354

        const std::string reader = trim(s->get("Modeling.prior_fct_name"));            // Config::read_inputs_files
        if (reader == "io_MS_Global") { s->mf = read_model_file(model_file, -1); s->in = build_init_ms_global(s->mf, s->resol, s->log); }
        else if (reader == "io_local") { s->mf = read_model_file(model_file, slice_ind); s->in = build_init_local(s->mf, s->resol, s->log); }
        else throw Fail{TAMCMC_IO_E_NAME, "prior_fct_name must be io_MS_Global or io_local (got '" + reader + "')"};

        s->prior_switch.resize(s->in.all.size());
        for (int i = 0; i < s->in.all.size(); i++) {                                     // convert_priors_names_to_switch
            const int id = s->primepriors.find(s->in.all.prior_names[i]);
            if (id == -9999) throw Fail{TAMCMC_IO_E_NAME, "Unknown prior name detected: " + s->in.all.prior_names[i]};
            s->prior_switch[i] = id;
        }
        s->model_case = s->models.find(s->in.model_fullname);
        if (s->model_case == -9999) throw Fail{TAMCMC_IO_E_NAME, "Unknown model name detected: " + s->in.model_fullname};
        s->likelihood_case = s->likelihoods.find(trim(s->get("Modeling.likelihood_fct_name")));
        if (s->likelihood_case == -9999) throw Fail{TAMCMC_IO_E_NAME, "Unknown likelihood name detected: " + s->get("Modeling.likelihood_fct_name")};
        s->prior_case = s->priors.find(reader);
        if (s->prior_case == -9999) throw Fail{TAMCMC_IO_E_NAME, "Unknown prior function name detected: " + reader};

        // crop to the model's range (config.cpp:101-129): first row with x >= fmin up to the first with x >= fmax
        const size_t N = d.rows.size();
        size_t imin = 0;
        while (imin < N && d.rows[imin][x_col] < s->mf.fmin) imin++;
        if (imin >= N) throw Fail{TAMCMC_IO_E_RANGE, "Found that xmin > max(data.x): the requested range is inconsistent with the data"};
        size_t imax = imin;
        while (imax < N && d.rows[imax][x_col] < s->mf.fmax) imax++;
        if (y_col < 0) throw Fail{TAMCMC_IO_E_SYNTAX, "You need to specify a column for y-data"};
        if ((size_t)y_col >= d.ncols) throw Fail{TAMCMC_IO_E_SYNTAX, "y_col outside the columns of the data file"};
        if (ysig_col >= 0 && (size_t)ysig_col >= d.ncols) throw Fail{TAMCMC_IO_E_SYNTAX, "ysig_col outside the columns of the data file"};
        const size_t n = imax - imin;
        s->x.resize(n); s->y.resize(n); s->sig.assign(n, 1.0);
        for (size_t i = 0; i < n; i++) {
            const dvec &r = d.rows[imin + i];
            s->x[i] = (size_t)x_col < r.size() ? r[x_col] : std::nan("");
            s->y[i] = (size_t)y_col < r.size() ? r[y_col] : std::nan("");
            if (ysig_col >= 0) s->sig[i] = (size_t)ysig_col < r.size() ? r[ysig_col] : std::nan("");
        }
        auto pick = [](const std::vector<std::string> &v, int i) { return (size_t)i < v.size() ? v[i] : std::string(); };
        s->xlabel = pick(d.labels, x_col); s->ylabel = pick(d.labels, y_col);
        s->xunit = pick(d.units, x_col); s->yunit = pick(d.units, y_col);
        s->loaded = true;
    });
}

extern "C" int tamcmc_setup_sizes(const tamcmc_setup *s, int32_t *Nparams, int32_t *Nvars, int64_t *Nx, int32_t plength[11],
                                  int32_t *model_case, int32_t *likelihood_case, int32_t *prior_case, double *likelihood_p)
{
    if (!s || !s->loaded) return TAMCMC_IO_E_INVALID;
    int nv = 0;
    for (int r : s->in.all.relax) nv += r == 1;
    if (Nparams) *Nparams = s->in.all.size();
    if (Nvars) *Nvars = nv;
    if (Nx) *Nx = (int64_t)s->x.size();
    if (plength) for (int k = 0; k < 11; k++) plength[k] = s->in.plength[k];
    if (model_case) *model_case = s->model_case;
    if (likelihood_case) *likelihood_case = s->likelihood_case;
    if (prior_case) *prior_case = s->prior_case;
    if (likelihood_p) *likelihood_p = to_dbl(s->get("Modeling.likelihood_params", "1"));
    return TAMCMC_IO_OK;
}

extern "C" int tamcmc_setup_inputs(const tamcmc_setup *s, double *inputs, int32_t *relax, int32_t *priors_names_switch,
                                   double *priors, double extra_priors[4], double *err)
{
    if (!s || !s->loaded) return TAMCMC_IO_E_INVALID;
    const Block &a = s->in.all;
    const int n = a.size();
    for (int i = 0; i < n; i++) {
        if (inputs) inputs[i] = a.inputs[i];
        if (relax) relax[i] = a.relax[i];
        if (priors_names_switch) priors_names_switch[i] = s->prior_switch[i];
        if (priors) for (int k = 0; k < 4; k++) priors[(size_t)k * n + i] = a.priors[k][i];
    }
    if (extra_priors) for (int k = 0; k < 4; k++) extra_priors[k] = s->in.extra_priors[k];
    if (err) {                                                                          // MALA::init_proposal, MALA.cpp:246-257
        int v = 0;
        for (int i = 0; i < n; i++) {
            if (a.relax[i] != 1) continue;
            double e = 1.0;                                                             // names without an entry keep 1
            for (size_t j = 0; j < s->errors.names.size(); j++)
                if (a.names[i] == s->errors.names[j]) e = a.inputs[i] * s->errors.frac[j] + s->errors.offset[j];
            err[v++] = e;
        }
    }
    return TAMCMC_IO_OK;
}

extern "C" int tamcmc_setup_data(const tamcmc_setup *s, double *x, double *y, double *sigma_y)
{
    if (!s || !s->loaded) return TAMCMC_IO_E_INVALID;
    const size_t n = s->x.size();
    if (x) memcpy(x, s->x.data(), n * sizeof(double));
    if (y) memcpy(y, s->y.data(), n * sizeof(double));
    if (sigma_y) memcpy(sigma_y, s->sig.data(), n * sizeof(double));
    return TAMCMC_IO_OK;
}

extern "C" int tamcmc_setup_name(const tamcmc_setup *s, int32_t which, int32_t i, char *buf, int32_t cap)
{
    if (!s || !s->loaded || !buf || cap <= 0) return TAMCMC_IO_E_INVALID;
    std::string v;
    switch (which) {
    case 0: case 1:
        if (i < 0 || i >= s->in.all.size()) return TAMCMC_IO_E_INVALID;
        v = which == 0 ? s->in.all.names[i] : s->in.all.prior_names[i];
        break;
    case 2: v = s->in.model_fullname; break;
    case 3: v = s->mf.ID; break;
    case 4: v = s->xlabel; break;
    case 5: v = s->ylabel; break;
    case 6: v = s->xunit; break;
    case 7: v = s->yunit; break;
    default: return TAMCMC_IO_E_INVALID;
    }
    if ((int32_t)v.size() + 1 > cap) return TAMCMC_IO_E_CAPACITY;
    copy_str(v, buf, cap);
    return TAMCMC_IO_OK;
}

extern "C" double tamcmc_setup_scalar(const tamcmc_setup *s, int32_t which)
{
    if (!s || !s->loaded) return std::nan("");
    switch (which) {
    case 0: return s->mf.Dnu;
    case 1: return s->mf.numax;
    case 2: return s->mf.C_l;
    case 3: return s->mf.fmin;
    case 4: return s->mf.fmax;
    case 5: return s->resol;
    default: return std::nan("");
    }
}

extern "C" int tamcmc_setup_sampler_cfg(const tamcmc_setup *s, tamcmc_sampler_cfg *cfg)
{
    if (!s || !cfg) return TAMCMC_IO_E_INVALID;
    return guarded(s, [&] {
        memset(cfg, 0, sizeof(*cfg));
        cfg->Nchains = (int32_t)to_long(s->get("MALA.Nchains", "1"));
        cfg->chain_offset = 0;
        cfg->Nchains_local = cfg->Nchains;
        cfg->lambda_temp = to_dbl(s->get("MALA.lambda_temp", "1"));
        cfg->target_acceptance = to_dbl(s->get("MALA.target_acceptance", "0.234"));
        cfg->c0 = to_dbl(s->get("MALA.c0", "1"));
        cfg->epsilon1 = to_dbl(s->get("MALA.epsilon1", "1e-12"));
        cfg->epsilon2 = to_dbl(s->get("MALA.epsilon2", "1e-12"));
        cfg->A1 = to_dbl(s->get("MALA.A1", "1e14"));
        cfg->dN_mixing = (int64_t)to_long(s->get("MALA.dN_mixing", "1"));               // str_to_int: "1." -> 1
        std::vector<long> Nt = to_longs(s->get("MALA.Nt_learn"), ","), per = to_longs(s->get("MALA.periods_learn"), " ,");
        if (Nt.empty() || Nt.size() > TAMCMC_MAX_LEARN) throw Fail{TAMCMC_IO_E_SYNTAX, "MALA.Nt_learn needs 1..8 entries"};
        if (per.size() + 1 < Nt.size()) throw Fail{TAMCMC_IO_E_SYNTAX, "MALA.periods_learn must have Nt_learn.size()-1 entries"};
        cfg->n_learn = (int32_t)Nt.size();
        for (size_t i = 0; i < Nt.size(); i++) cfg->Nt_learn[i] = Nt[i];
        for (size_t i = 0; i + 1 < Nt.size(); i++) cfg->periods_learn[i] = per[i];
        cfg->seed = 0;
        cfg->prior_fct_switch = s->loaded ? s->prior_case : s->priors.find(trim(s->get("Modeling.prior_fct_name")));
    });
}
