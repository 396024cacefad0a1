// cpptamcmc_hip -- command-line driver with the reference program's call sequence and directory conventions
// (main.cpp:29-199, config_presets.cpp:39-200,255-390), built on the C ABI of libtamcmc_accel.so:
//
//   cpptamcmc_hip execute 1 <first_object> <last_object> [<first_slice> [<last_slice>]]   (1-based, like main.cpp:44-61)
//   cpptamcmc_hip execute 0                                   read the configuration and stop
//   cpptamcmc_hip version
//   options (after the positional arguments): --root DIR (where Config/ lives; default: current directory),
//            --seed N (default: time(NULL), MALA.cpp:62-63), --device D, --restore-precision P (default 17), --quiet
//
// For every object of config_presets.cfg's table, every `* fmin fmax` slice of its .model file and every phase
// (Burn-in / Learning / Acquire ...): apply the presets, read <models_dir>/<ID>.model and .data, run the phase on the
// GPU, write <out_dir>/<ID>/outputs/* and <out_dir>/<ID>/restore/*.  Not reproduced: the zip backups of the inputs
// (main.cpp:201-244) and every gnuplot diagnostic (Diagnostics, SURVEY.md 2: out of scope).
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "tamcmc_accel.h"
#include "tamcmc_io.h"
#include "tamcmc_outputs.h"
#include "tamcmc_sampler.h"

namespace {

std::string trim(const std::string &s)
{
    const size_t b = s.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) return "";
    return s.substr(b, s.find_last_not_of(" \t\r\n") - b + 1);
}

std::vector<std::string> split(const std::string &s, char sep)
{
    std::vector<std::string> out;
    std::string cur;
    std::istringstream is(s);
    while (std::getline(is, cur, sep)) out.push_back(trim(cur));
    return out;
}

[[noreturn]] void die(const std::string &msg)
{
    fprintf(stderr, "cpptamcmc_hip: %s\nThe program will exit now\n", msg.c_str());
    exit(EXIT_FAILURE);
}

// Config_presets (config_presets.h; read_cfg_file_presets, config_presets.cpp:255-390)
struct Presets {
    bool force_manual_config = false;
    std::string manual_config_file, cfg_models_dir, cfg_out_dir;
    std::vector<std::string> processing, core_out, core_in;
    std::vector<long> Nsamples, restore;
    std::vector<double> c0;
    long first_process = 0, last_process = 0;
    std::vector<std::vector<std::string>> table_ids;
};

Presets read_presets(const std::string &path)
{
    std::ifstream f(path.c_str());
    if (!f.is_open()) die("Could not open the master configuration file " + path);
    Presets P;
    int found = 0;
    std::string line;
    while (std::getline(f, line)) {
        line = trim(line);
        if (line.empty() || line[0] == '#') continue;
        const size_t semi = line.find(';');
        if (semi == std::string::npos) die("config_presets.cfg: terminator ';' not found in: " + line);
        line = trim(line.substr(0, semi));
        if (line == "/END") break;
        const size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        const std::string key = trim(line.substr(0, eq)), val = trim(line.substr(eq + 1));
        found++;
        if (key == "force_manual_config") P.force_manual_config = atoi(val.c_str()) != 0;
        else if (key == "manual_config_file") P.manual_config_file = val;
        else if (key == "cfg_models_dir") P.cfg_models_dir = val;
        else if (key == "cfg_out_dir") P.cfg_out_dir = val;
        else if (key == "processing") P.processing = split(val, ',');
        else if (key == "Nsamples") for (const std::string &v : split(val, ',')) P.Nsamples.push_back(atol(v.c_str()));
        else if (key == "c0") for (const std::string &v : split(val, ',')) P.c0.push_back(atof(v.c_str()));
        else if (key == "restore") for (const std::string &v : split(val, ',')) P.restore.push_back(atol(v.c_str()));
        else if (key == "core_out") P.core_out = split(val, ',');
        else if (key == "core_in") P.core_in = split(val, ',');
        else if (key == "start_index_processing") P.first_process = (long)atof(val.c_str());
        else if (key == "last_index_processing") P.last_process = (long)atof(val.c_str());
        else if (key == "table_ids") {
            const std::vector<std::string> sz = split(val, ',');
            if (sz.size() < 2) die("config_presets.cfg: table_ids needs 'rows, columns'");
            const long rows = atol(sz[0].c_str()), cols = atol(sz[1].c_str());
            for (long r = 0; r < rows; r++) {
                if (!std::getline(f, line)) die("config_presets.cfg: table_ids announces more rows than the file holds");
                line = trim(line);
                const size_t s2 = line.find(';');
                if (s2 != std::string::npos) line = line.substr(0, s2);
                std::vector<std::string> cells;
                std::istringstream is(line);
                std::string c;
                while (is >> c) cells.push_back(c);
                if ((long)cells.size() < cols) die("config_presets.cfg: a table_ids row has too few columns: " + line);
                P.table_ids.push_back(cells);
            }
        } else found--;
    }
    if (found != 13) die("config_presets.cfg: incorrect number of keywords (expected 13): check the syntax");
    const size_t np = P.processing.size();
    if (P.Nsamples.size() < np || P.c0.size() < np || P.restore.size() < np || P.core_out.size() < np || P.core_in.size() < np)
        die("config_presets.cfg: Nsamples / c0 / restore / core_out / core_in need one entry per processing phase");
    return P;
}

void make_dir(const std::string &p)
{
    struct stat sb;
    if (stat(p.c_str(), &sb) == 0 && S_ISDIR(sb.st_mode)) return;
    if (mkdir(p.c_str(), 0777) != 0 && errno != EEXIST) die("cannot create directory " + p + ": " + strerror(errno));
}

void set_key(tamcmc_setup *s, const char *group, const char *key, const std::string &v)
{
    if (tamcmc_setup_set(s, group, key, v.c_str()) != TAMCMC_IO_OK) die(std::string("cannot set ") + group + "." + key);
}

void progress(int64_t i, int64_t n, void *)
{
    printf("[%lld]  of %lld\n", (long long)i, (long long)n);
    fflush(stdout);
}

} // namespace

int main(int argc, char *argv[])
{
    std::vector<std::string> pos;
    std::string root = ".";
    unsigned seed = (unsigned)time(NULL);
    int device = 0, restore_precision = 17;
    bool quiet = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](const char *what) { if (i + 1 >= argc) die(std::string(what) + " needs a value"); return std::string(argv[++i]); };
        if (a == "--root") root = need("--root");
        else if (a == "--seed") seed = (unsigned)strtoul(need("--seed").c_str(), nullptr, 10);
        else if (a == "--device") device = atoi(need("--device").c_str());
        else if (a == "--restore-precision") restore_precision = atoi(need("--restore-precision").c_str());
        else if (a == "--quiet") quiet = true;
        else pos.push_back(a);
    }
    if (pos.size() == 1 && pos[0] == "version") { printf("cpptamcmc_hip (%s)\n", tamcmc_version()); return 0; }
    const bool read_only = pos.size() == 2 && pos[0] == "execute" && pos[1] == "0";
    if (!read_only && !(pos.size() >= 4 && pos.size() <= 6 && pos[0] == "execute")) {
        fprintf(stderr, "Unrecognized argument\n"
                        "     - To execute: %s execute 1 <start_idx_object> <last_idx_object>  <first_idx_slice> <last_idx_slice>\n"
                        "       <first_idx_slice> and <last_idx_slice> are optional. If none provided, then the program does all slices.\n"
                        "     - To stop after reading the configuration: %s execute 0\n"
                        "     - To show version: %s version\n"
                        "     - options: --root DIR  --seed N  --device D  --restore-precision P  --quiet\n", argv[0], argv[0], argv[0]);
        return EXIT_FAILURE;
    }
    printf(" --------- TAMCMC (HIP) ----------\n- Configuration root: %s\n", root.c_str());
    const std::string cfg_dir = root + "/Config/default";
    const Presets P = read_presets(root + "/Config/config_presets.cfg");
    {
        tamcmc_setup *probe = nullptr;
        if (tamcmc_setup_create(&probe, cfg_dir.c_str()) != TAMCMC_IO_OK) die("cannot read the default configuration in " + cfg_dir);
        tamcmc_setup_destroy(probe);
    }
    if (read_only) { printf("Detection of execute=0... The code will exit now\n"); return 0; }
    if (P.force_manual_config) die("force_manual_config=1 is not supported by this driver: edit Config/default/config_default.cfg instead");
    if (atoi(pos[1].c_str()) == 0) { printf("Detection of execute=0... The code will exit now\n"); return 0; }

    long first_id = atol(pos[2].c_str()) - 1, last_id = atol(pos[3].c_str()) - 1;
    long start_slice = pos.size() > 4 ? atol(pos[4].c_str()) - 1 : -1, last_slice = pos.size() > 5 ? atol(pos[5].c_str()) - 1 : -1;
    if (first_id < 0) first_id = 0;
    if (last_id >= (long)P.table_ids.size()) last_id = (long)P.table_ids.size() - 1;
    long last_process = P.last_process;
    if (last_process >= (long)P.processing.size()) last_process = (long)P.processing.size() - 1;

    for (long id = first_id; id <= last_id; id++) {
        const std::string name = P.table_ids[id][0];
        const std::string model_file = P.cfg_models_dir + name + ".model", data_file = P.cfg_models_dir + name + ".data";
        int32_t nslices = 0;
        if (tamcmc_model_file_slices(model_file.c_str(), nullptr, 0, &nslices) != TAMCMC_IO_OK || nslices < 1)
            die("Unable to read the frequency range(s) of " + model_file);
        const long s0 = start_slice < 0 ? 0 : start_slice;
        const long s1 = (last_slice < 0 || (last_slice < start_slice && last_slice > 0)) ? nslices : last_slice;   // main.cpp:115-125
        for (long sl = s0; sl < s1; sl++)
            for (long ph = P.first_process; ph <= last_process; ph++) {
                printf("---------------------------------------------------------------------------------------\n");
                printf("   Processing Object %ld/%zu: %s   Frequency Slice %ld/%d   Phase %ld/%zu: %s\n", id + 1, P.table_ids.size(),
                       name.c_str(), sl + 1, nslices, ph + 1, P.processing.size(), P.processing[ph].c_str());
                tamcmc_setup *S = nullptr;
                if (tamcmc_setup_create(&S, cfg_dir.c_str()) != TAMCMC_IO_OK) die("cannot read the default configuration in " + cfg_dir);
                // ---- Config_presets::apply_presets, config_presets.cpp:39-192
                const std::string obj = P.cfg_out_dir + "/" + name;
                make_dir(P.cfg_out_dir); make_dir(obj); make_dir(obj + "/diags"); make_dir(obj + "/diags/pdfs");
                make_dir(obj + "/restore"); make_dir(obj + "/outputs");
                set_key(S, "Outputs", "output_dir", obj + "/outputs/");
                set_key(S, "Outputs", "restore_dir", obj + "/restore/");
                set_key(S, "Diagnostics", "output_dir", obj + "/diags/");
                if (tamcmc_setup_apply_phase(S, P.processing[ph].c_str(), P.Nsamples[ph], P.c0[ph]) != TAMCMC_IO_OK)
                    die(std::string("phase '") + P.processing[ph] + "': " + tamcmc_setup_error(S));
                const std::string tag = nslices == 1 ? name + "_" : name + "_" + std::to_string(sl + 1) + "_";
                set_key(S, "Outputs", "output_root_name", tag + P.core_out[ph] + "_");
                set_key(S, "Diagnostics", "output_root_name", tag + P.core_out[ph] + "_");
                set_key(S, "Outputs", "restore_file_in", tag + "restore_" + P.core_in[ph] + "_");
                set_key(S, "Outputs", "restore_file_out", tag + "restore_" + P.core_out[ph] + "_");
                const long r = P.restore[ph];
                if (r < 0 || r > 3) die("restore[i] must be a number not greater than 3");
                set_key(S, "Outputs", "do_restore_proposal", r >= 2 ? "1" : "0");
                set_key(S, "Outputs", "do_restore_variables", r >= 1 ? "1" : "0");
                set_key(S, "Outputs", "do_restore_last_index", r == 3 ? "1" : "0");
                set_key(S, "Outputs", "erase_old_files", r == 3 ? "0" : "1");
                // ---- Config::setup(slice)
                if (tamcmc_setup_load(S, model_file.c_str(), data_file.c_str(), (int32_t)sl) != TAMCMC_IO_OK)
                    die(std::string("reading ") + model_file + " / " + data_file + ": " + tamcmc_setup_error(S));
                if (!quiet) printf("%s", tamcmc_setup_log(S));
                int32_t Nparams = 0, Nvars = 0, plength[11], model_case = 0, like_case = 0, prior_case = 0;
                int64_t Nx = 0;
                double like_p = 1.0;
                tamcmc_setup_sizes(S, &Nparams, &Nvars, &Nx, plength, &model_case, &like_case, &prior_case, &like_p);
                std::vector<double> x(Nx), y(Nx), sig(Nx), inputs(Nparams), priors(4 * (size_t)Nparams), err(Nvars);
                std::vector<int32_t> relax(Nparams), sw(Nparams);
                double extra[4];
                tamcmc_setup_data(S, x.data(), y.data(), sig.data());
                tamcmc_setup_inputs(S, inputs.data(), relax.data(), sw.data(), priors.data(), extra, err.data());
                printf("   model %d (%d parameters, %d free), %lld bins in [%g, %g]\n", model_case, Nparams, Nvars, (long long)Nx,
                       tamcmc_setup_scalar(S, 3), tamcmc_setup_scalar(S, 4));
                // ---- the hot path on the GPU + sampler + outputs
                tamcmc_ctx *ctx = nullptr;
                int rc = tamcmc_ctx_create(&ctx, device, model_case, like_case, like_p, plength, Nx, x.data(), y.data(), sig.data());
                if (rc != TAMCMC_OK) die(std::string("tamcmc_ctx_create: ") + tamcmc_strerror(rc) + " " + tamcmc_last_hip_error());
                tamcmc_sampler_cfg cfg;
                if (tamcmc_setup_sampler_cfg(S, &cfg) != TAMCMC_IO_OK) die(std::string("MALA configuration: ") + tamcmc_setup_error(S));
                cfg.seed = seed;
                tamcmc_sampler *smp = nullptr;
                rc = tamcmc_sampler_create_hip(&smp, &cfg, ctx, Nparams, plength, inputs.data(), relax.data(), sw.data(), priors.data(), 4,
                                               extra, err.data());
                if (rc != TAMCMC_OK) die(std::string("tamcmc_sampler_create_hip: ") + tamcmc_strerror(rc));
                char ebuf[1024] = "";
                const time_t t0 = time(NULL);
                rc = tamcmc_run_phase(S, smp, quiet ? nullptr : progress, nullptr, restore_precision, ebuf, sizeof(ebuf));
                if (rc != TAMCMC_IO_OK) die(std::string("phase failed: ") + ebuf);
                printf("    Calculation finished in: %.2f min\n", difftime(time(NULL), t0) / 60.);
                tamcmc_sampler_destroy(smp);
                tamcmc_ctx_destroy(ctx);
                tamcmc_setup_destroy(S);
            }
    }
    return 0;
}
