// solve_mps_file -- command-line driver: read an .mps/.mps.gz file, solve, print a summary.
// Same flags and flow as the reference driver (reference src/solve_mps_file.cpp:14-32,120-131);
// links only against the C boundary in include/HPRLP.h.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include <sys/stat.h>

#include "HPRLP.h"

static void usage(const char *prog) {
    std::cout << "Usage: " << prog << " -i <input.mps|input.mps.gz> [options]\n\nOptions:\n"
              << "  -i, --input <path>         input .mps or .mps.gz file (required)\n"
              << "      --device <id>          GPU device id (default: 0)\n"
              << "      --max-iter <N>         max iterations (default: INT32_MAX)\n"
              << "      --tol <eps>            stopping tolerance (default: 1e-4)\n"
              << "      --time-limit <sec>     time limit in seconds (default: 3600)\n"
              << "      --check-iter <N>       check interval (default: 150)\n"
              << "      --cusparse-spmv <true/false>    accepted for compatibility, ignored\n"
              << "      --autotune-verbose <true/false> accepted for compatibility, ignored\n"
              << "      --cr <true/false>      Curtis-Reid prescaling (default: true)\n"
              << "      --ruiz <true/false>    Ruiz scaling (default: true)\n"
              << "      --pock <true/false>    Pock-Chambolle scaling (default: true)\n"
              << "      --bc <true/false>      bounds/cost scaling (default: true)\n"
              << "      --presolve <true/false>  enable/disable the host presolve (default: true)\n"
              << "  -h, --help                 show this help and exit\n";
}

static bool truthy(const char *s) { return std::strcmp(s, "true") == 0 || std::strcmp(s, "1") == 0; }

int main(int argc, char **argv) {
    std::string input;
    HPRLP_parameters param;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-h" || a == "--help") { usage(argv[0]); return 0; }
        if (i + 1 >= argc) { std::cerr << "Missing value for option: " << a << "\n"; usage(argv[0]); return 1; }
        const char *v = argv[++i];
        if (a == "-i" || a == "--input") input = v;
        else if (a == "--device") param.device_number = std::atoi(v);
        else if (a == "--max-iter") param.max_iter = std::atoi(v);
        else if (a == "--tol") param.stop_tol = std::atof(v);
        else if (a == "--time-limit") param.time_limit = std::atof(v);
        else if (a == "--check-iter") param.check_iter = std::atoi(v);
        else if (a == "--cusparse-spmv") param.CUSPARSE_spmv = truthy(v);
        else if (a == "--autotune-verbose") param.autotune_verbose = truthy(v);
        else if (a == "--cr") param.use_CR_scaling = truthy(v);
        else if (a == "--ruiz") param.use_Ruiz_scaling = truthy(v);
        else if (a == "--pock") param.use_Pock_Chambolle_scaling = truthy(v);
        else if (a == "--bc") param.use_bc_scaling = truthy(v);
        else if (a == "--presolve") param.use_presolve = truthy(v);
        else { std::cerr << "Unknown option: " << a << "\n"; usage(argv[0]); return 1; }
    }
    if (input.empty()) { std::cerr << "Error: Input file is required. Use -i or --input option.\n"; usage(argv[0]); return 1; }
    struct stat st;
    if (stat(input.c_str(), &st) != 0) { std::cerr << "Input file does not exist: " << input << "\n"; return 1; }
    LP_info_cpu *model = create_model_from_mps(input.c_str());
    if (!model) { std::cerr << "Failed to create model from " << input << "\n"; return 1; }
    HPRLP_results r = solve(model, &param);
    std::cout << "status = " << r.status << ", iter = " << r.iter << ", primal_obj = " << r.primal_obj
              << ", residual = " << r.residuals << ", time = " << r.time << " s\n";
    const bool ok = std::strcmp(r.status, "ERROR") != 0;
    std::free(r.x); std::free(r.y); std::free(r.z);
    free_model(model);
    return ok ? 0 : 2;
}
