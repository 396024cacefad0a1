// getmodel_hip -- tools/getmodel.cpp of the reference on the GPU path: build the model spectrum of up to 5 parameter
// rows on the grid of a .data file and write data + models as one ASCII matrix.
//
//   getmodel_hip <data file> <parameters file> <model name> [<output file>]      (tools/getmodel.cpp:47-62)
//
// parameters file: `#` comments, then plength (11 integers), then one parameter row per line (at most 5).
// models_ctrl.list is looked for next to the executable's working directory (getmodel.cpp:81) or in $TAMCMC_MODELS_LIST.
// The pre-1.3.0 retro-compatibility path (10-entry plength, getmodel.cpp:100-111,279-305) is not provided.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "tamcmc_accel.h"
#include "tamcmc_io.h"

static std::string trim(const std::string &s)
{
    const size_t b = s.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) return "";
    return s.substr(b, s.find_last_not_of(" \t\r\n") - b + 1);
}

static std::string cell(double v)
{
    std::ostringstream o;
    o.precision(12);                         // getmodel.cpp:133
    o << v;
    return o.str();
}

int main(int argc, char *argv[])
{
    if (argc == 2 && std::string(argv[1]) == "version") { printf("getmodel_hip (%s)\n", tamcmc_version()); return 0; }
    if (argc != 4 && argc != 5) {
        fprintf(stderr, " You need to provide at least three argument to that function. The available arguments are: \n"
                        "     [1] The data filename. Should be in the same format as those provided to TAMCMC (*.data file)\n"
                        "     [2] The filename for the parameters to read. After comments ('#'), this files contains on row(1) plength and on row(2:2+Nmaxlines) the model parameters. Maximum number of models is Nmaxlines=5\n"
                        "     [3] The model name among the family of MS_Global models (e.g. model_MS_Global_a1etaa3_HarveyLike)\n"
                        "     [4] [Optional] the output filename. If not given, then the output file 'output_model.ascii'\n");
        return EXIT_FAILURE;
    }
    const std::string data_file = argv[1], params_file = argv[2], model_name = argv[3];
    const std::string out_file = argc == 5 ? argv[4] : "output_model.ascii";
    const int Nmaxlines = 5;

    double *data = nullptr;
    int64_t nrows = 0;
    int32_t ncols = 0;
    if (tamcmc_data_file_read(data_file.c_str(), &data, &nrows, &ncols) != TAMCMC_IO_OK || ncols < 2) {
        fprintf(stderr, "Unable to read the data file: %s\n", data_file.c_str());
        return EXIT_FAILURE;
    }
    const char *lenv = getenv("TAMCMC_MODELS_LIST");
    const std::string list = lenv ? lenv : "models_ctrl.list";
    int32_t model_case = -1;
    if (tamcmc_list_file_lookup(list.c_str(), model_name.c_str(), &model_case) != TAMCMC_IO_OK) {
        fprintf(stderr, "Unknown model name '%s' (list file: %s)\n", model_name.c_str(), list.c_str());
        return EXIT_FAILURE;
    }
    std::ifstream f(params_file.c_str());
    if (!f.is_open()) { fprintf(stderr, "Unable to open the file: %s\n", params_file.c_str()); return EXIT_FAILURE; }
    std::string line;
    std::vector<int32_t> plength;
    std::vector<std::vector<double>> rows;
    while (std::getline(f, line)) {
        line = trim(line);
        if (line.empty() || line[0] == '#') continue;
        std::istringstream is(line);
        if (plength.empty()) { int v; while (is >> v) plength.push_back(v); continue; }
        if ((int)rows.size() >= Nmaxlines) break;
        std::vector<double> r; double v;
        while (is >> v) r.push_back(v);
        rows.push_back(r);
    }
    if (plength.size() != 11) { fprintf(stderr, "plength must have 11 entries (found %zu)\n", plength.size()); return EXIT_FAILURE; }
    int np = 0;
    for (int v : plength) np += v;
    std::vector<double> x(nrows), y(nrows), sig(nrows, 1.0);
    for (int64_t i = 0; i < nrows; i++) { x[i] = data[i * ncols]; y[i] = data[i * ncols + 1]; if (ncols == 3) sig[i] = data[i * ncols + 2]; }
    tamcmc_ctx *ctx = nullptr;
    int rc = tamcmc_ctx_create(&ctx, 0, model_case, 0, 1.0, plength.data(), nrows, x.data(), y.data(), sig.data());
    if (rc != TAMCMC_OK) { fprintf(stderr, "tamcmc_ctx_create: %s %s\n", tamcmc_strerror(rc), tamcmc_last_hip_error()); return EXIT_FAILURE; }
    std::vector<std::vector<double>> models;
    for (const std::vector<double> &r : rows) {
        if ((int)r.size() != np) { fprintf(stderr, "a parameter row has %zu entries, plength sums to %d\n", r.size(), np); return EXIT_FAILURE; }
        std::vector<double> m(nrows);
        int32_t st = 0;
        rc = tamcmc_model_explicit(ctx, np, r.data(), m.data(), &st);
        if (rc != TAMCMC_OK || st == TAMCMC_CHAIN_EMPTY_WINDOW) { fprintf(stderr, "model evaluation failed (%s, status %d)\n", tamcmc_strerror(rc), st); return EXIT_FAILURE; }
        models.push_back(m);
    }
    tamcmc_ctx_destroy(ctx);
    // Eigen's default matrix format: every coefficient padded to the widest one of the whole matrix, one space between
    const size_t ctot = (size_t)ncols + models.size();
    std::vector<std::string> cells((size_t)nrows * ctot);
    size_t width = 0;
    for (int64_t i = 0; i < nrows; i++)
        for (size_t c = 0; c < ctot; c++) {
            const double v = c < (size_t)ncols ? data[i * ncols + c] : models[c - ncols][i];
            cells[i * ctot + c] = cell(v);
            if (cells[i * ctot + c].size() > width) width = cells[i * ctot + c].size();
        }
    std::ofstream o(out_file.c_str());
    if (!o.is_open()) { fprintf(stderr, " Unable to open the output file %s\n", out_file.c_str()); return EXIT_FAILURE; }
    for (int64_t i = 0; i < nrows; i++) {
        for (size_t c = 0; c < ctot; c++) o << (c ? " " : "") << std::string(width - cells[i * ctot + c].size(), ' ') << cells[i * ctot + c];
        o << "\n";
    }
    tamcmc_buffer_free(data);
    printf("Output model file successfully written\n");
    return 0;
}
