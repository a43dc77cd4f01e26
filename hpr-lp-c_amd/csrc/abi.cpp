// abi.cpp -- extern "C" boundary: the reference's solve entry points (include/HPRLP.h) and the
// step-level extension (include/hprlp_amd.h).  Every entry point catches exceptions.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <iomanip>
#include <iostream>

#include "hprlp_amd.h"
#include "env.h"
#include "dist.h"
#include "presolve.h"
#include "reorder.h"
#include "solver.h"
#include "version.h"

using namespace hprlp;

struct hprlp_solver {
    Solver s;
    Comm *comm = nullptr;   // owned; destroyed after the solver's device state
    Comm *xcomm = nullptr;  // owned; the exchange stream's own communicator (second unique id), may be null
    ~hprlp_solver() {}
};

static HPRLP_results make_error_result(const char *status) {
    HPRLP_results r;
    std::memset(r.status, 0, sizeof(r.status));
    std::strncpy(r.status, status, sizeof(r.status) - 1);
    r.residuals = r.primal_obj = r.gap = 0.0;
    return r;
}

static void print_banner_and_parameters(const HPRLP_parameters *p) {
    std::cout << "\n==================================================================\n"
              << "                 HPR-LP Solver  (MI355X / gfx950 HIP build)       \n"
              << "     Halpern Peaceman-Rachford Linear Programming Solver          \n"
              << "  Version: " << HPRLP_VERSION_STRING << "   backend: " << HPRLP_BACKEND_STRING << "\n"
              << "==================================================================\n\n";
    std::cout << "Solver Parameters:\n"
              << "  Device:              GPU " << p->device_number << "\n"
              << "  Max Iterations:      " << p->max_iter << "\n"
              << "  Stopping Tolerance:  " << std::scientific << std::setprecision(1) << p->stop_tol << "\n"
              << std::defaultfloat << "  Time Limit:          " << std::fixed << std::setprecision(1) << p->time_limit
              << " seconds\n" << std::defaultfloat << "  Check Interval:      " << p->check_iter << " iterations\n"
              << "  PSLP Presolve:       " << (p->use_presolve ? "Enabled" : "Disabled") << "\n"
              << "  Scaling:  CR=" << p->use_CR_scaling << " Ruiz=" << p->use_Ruiz_scaling
              << " Pock-Chambolle=" << p->use_Pock_Chambolle_scaling << " b/c=" << p->use_bc_scaling << "\n\n";
}

extern "C" const char *hprlp_last_error(void) { return last_error_cstr(); }
extern "C" const char *hprlp_backend(void) { return HPRLP_BACKEND_STRING; }

// ---- warm-up -------------------------------------------------------------------------------------------------------
// The first solve of a process pays for the HIP runtime's start-up, the device context, the first stream and the library's code
// objects (loaded as their first kernels are launched): 0.10 s on a Netlib-scale LP whose solve takes 0.04 s
// (profiles/r04_cold_start.txt).  None of it depends on the LP.  hprlp_warmup() does it on request -- a serving process calls
// it once at start-up, the first solve() then costs what the later ones do plus the first stream (0.03 s).  Measured and NOT
// done: starting it implicitly from the model constructors on a background thread -- a context first touched by another
// thread cost the solving thread MORE set-up time (0.18 instead of 0.09 s), and loading all six code objects up front is
// 0.05 s of which a small solve needs 0.01.
namespace hprlp {
void warm_kernels_tu();
void warm_small_tu();
void warm_batched_tu();
void warm_transpose_tu();
void warm_tiled_build_tu();
void warm_reorder_tu();
}  // namespace hprlp

static double g_warm_seconds[4] = {0, 0, 0, 0};

extern "C" int hprlp_warmup(int device) {
    const auto t0 = time_now();
    int count = 0;
    if (hipInit(0) != hipSuccess || hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        set_last_error("hprlp_warmup: no usable GPU");
        return -1;
    }
    if (device < 0 || device >= count) {
        set_last_error("hprlp_warmup: device " + std::to_string(device) + " is outside [0, " + std::to_string(count) + ")");
        return -1;
    }
    const auto t1 = time_now();
    hipStream_t s = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipFree(nullptr) != hipSuccess || hipStreamCreate(&s) != hipSuccess) {
        (void)hipGetLastError();
        set_last_error("hprlp_warmup: the device context could not be created");
        return -1;
    }
    (void)hipStreamDestroy(s);
    const auto t2 = time_now();
    // one attribute query per translation unit loads that unit's code object (deferred loading: otherwise at its first launch)
    int rc = 0;
    try {
        warm_kernels_tu();
        warm_small_tu();
        warm_batched_tu();
        warm_transpose_tu();
        warm_tiled_build_tu();
        warm_reorder_tu();
    } catch (const std::exception &e) {
        set_last_error(std::string("hprlp_warmup: a code object did not load: ") + e.what());
        rc = -2;
    } catch (...) {
        set_last_error("hprlp_warmup: a code object did not load");
        rc = -2;
    }
    (void)hipGetLastError();
    const auto t3 = time_now();
    g_warm_seconds[0] = std::chrono::duration<double>(t1 - t0).count();
    g_warm_seconds[1] = std::chrono::duration<double>(t2 - t1).count();
    g_warm_seconds[2] = std::chrono::duration<double>(t3 - t2).count();
    g_warm_seconds[3] = std::chrono::duration<double>(t3 - t0).count();
    if (env_get("HPRLP_TIMING"))
        std::cerr << "[timing] warm-up: runtime start-up " << g_warm_seconds[0] << " s, device context + first stream " << g_warm_seconds[1]
                  << " s, code objects " << g_warm_seconds[2] << " s" << std::endl;
    return rc;
}

// The drop-in path (round 5).  A caller that comes through the reference's unchanged bindings cannot call hprlp_warmup(): the
// model constructors do the process-wide part on the CALLING thread, once -- runtime start-up, device 0's context and first
// stream, and the two code objects every solve launches from (kernels.hip, small.hip; the others load with their first kernel:
// a Netlib-scale solve never pays for the tiled builders).  Quiet: a host without a GPU builds models as before (solves fail
// loudly later).  The cost moves from the first solve() to the first create_model_* (profiles/r05_cold_start.txt); the
// background-thread form of round 4 stays rejected (a context first touched by another thread cost the solver more).
namespace hprlp {
void warm_for_first_solve() {
    if (env_on("HPRLP_NO_WARM_MODEL")) return;  // (a process that forks after building its models: INTEGRATION.md)
    static std::once_flag once;
    std::call_once(once, []() {
        const auto t0 = time_now();
        int count = 0;
        if (hipInit(0) != hipSuccess || hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
            (void)hipGetLastError();
            return;
        }
        hipStream_t s = nullptr;
        if (hipSetDevice(0) == hipSuccess && hipFree(nullptr) == hipSuccess && hipStreamCreate(&s) == hipSuccess) (void)hipStreamDestroy(s);
        try {
            warm_kernels_tu();
            warm_small_tu();
        } catch (...) {
        }
        (void)hipGetLastError();
        if (env_get("HPRLP_TIMING")) std::cerr << "[timing] warm-up at model creation: " << time_since(t0) << " s" << std::endl;
    });
}
}  // namespace hprlp

extern "C" int hprlp_warmup_seconds(double out[4]) {
    if (!out) return -1;
    for (int i = 0; i < 4; ++i) out[i] = g_warm_seconds[i];
    return 0;
}

// phases of the calling thread's last HPRLP_main_solve (hprlp_last_solve_phases): device set-up (upload, transpose, tiled
// copies, ordering), scaling, power iteration, loop, solution's way back, teardown of the device state, whole call
static thread_local double g_phases[8] = {0, 0, 0, 0, 0, 0, 0, 0};

extern "C" int hprlp_last_solve_phases(double out[8]) {
    if (!out) return -1;
    for (int i = 0; i < 8; ++i) out[i] = g_phases[i];
    return 0;
}

// reference src/HPRLP.cu:116-311
extern "C" HPRLP_results HPRLP_main_solve(const LP_info_cpu *model, const HPRLP_parameters *param) {
    if (!model || !param) {
        std::cerr << "[error] Null model or parameter pointer" << std::endl;
        return make_error_result("ERROR");
    }
    const auto t_call = time_now();
    try {
        print_banner_and_parameters(param);
        HPRLP_results out;
        auto t_down = time_now();
        {
        Solver s;
        s.setup(model, param);
        std::cout << "Setup (copy and allocation) time = " << std::fixed << std::setprecision(2) << s.setup_time
                  << " seconds" << std::endl;
        s.scale();
        std::cout << "Scaling time = " << std::fixed << std::setprecision(2) << s.scaling_time << " seconds" << std::endl;
        s.lambda_max = s.power_iteration(5000, 1e-4, nullptr) * 1.01;  // src/HPRLP.cu:81-97
        if (s.power_iters >= 5000)
            std::cout << "Power iteration did not converge within the specified tolerance.\n";
        std::cout << "ESTIMATING MAXIMUM EIGENVALUE time = " << std::fixed << std::setprecision(2) << s.power_time
                  << " seconds" << std::endl << std::defaultfloat;
        s.init_iteration_state();
        const auto t_loop = time_now();
        s.solve_loop(&out);
        const double loop_s = time_since(t_loop);
        const auto t_col = time_now();
        s.collect_solution(&out);
        g_phases[0] = s.setup_time; g_phases[1] = s.scaling_time; g_phases[2] = s.power_time; g_phases[3] = loop_s;
        g_phases[4] = time_since(t_col);
        t_down = time_now();
        }  // (the solver's device state goes here: part of the call's wall time)
        g_phases[5] = time_since(t_down);
        g_phases[6] = time_since(t_call);
        std::cout << "\n=== Solution Summary ===\n"
                  << "Status: " << out.status << "\nIterations: " << out.iter << "\nTime: " << out.time
                  << " seconds\nPrimal Objective: " << std::scientific << std::setprecision(12) << out.primal_obj
                  << "\nResidual: " << out.residuals << "\n\n" << std::defaultfloat;
        return out;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        std::cerr << "[error] HPRLP_main_solve failed: " << e.what() << std::endl;
        return make_error_result("ERROR");
    }
}

// NULL if the CSR arrays of `model` are consistent, else what is wrong with them.
static const char *invalid_model_reason(const LP_info_cpu *model) {
    const sparseMatrix *A = model->A;
    if (model->m <= 0 || model->n <= 0) return "model dimensions must be positive";
    if (!A || !A->rowPtr || !A->colIndex || !A->value) return "model matrix arrays missing";
    if (!model->AL || !model->AU || !model->l || !model->u || !model->c) return "model vectors missing";
    if (A->row != model->m || A->col != model->n) return "model matrix dimensions inconsistent";
    if (A->rowPtr[0] != 0) return "row pointer array does not start at 0";
    for (int i = 0; i < model->m; ++i)
        if (A->rowPtr[i + 1] < A->rowPtr[i]) return "row pointer array is not monotone";
    if (A->rowPtr[model->m] != A->numElements) return "row pointer array does not end at numElements";
    for (int k = 0; k < A->numElements; ++k)
        if (A->colIndex[k] < 0 || A->colIndex[k] >= model->n) return "column index out of range";
    return nullptr;
}

// reference src/HPRLP.cu:493-524.  With use_presolve (the default) the model is first reduced on the
// host (presolve.h -- our own in-process presolver where the reference forks a PSLP worker,
// src/pslp_integration.cpp:628-713), the reduced model goes through HPRLP_main_solve and the result
// is mapped back to the original dimensions and checked against the original model
// (src/pslp_integration.cpp:715-787).  If the presolver declines, the original model is solved.
extern "C" HPRLP_results solve(const LP_info_cpu *model, const HPRLP_parameters *param) {
    if (!model) {
        std::cerr << "[error] Null model pointer" << std::endl;
        return make_error_result("ERROR");
    }
    HPRLP_parameters dflt;
    const HPRLP_parameters *p = param ? param : &dflt;
    if (!p->use_presolve) return HPRLP_main_solve(model, p);
    const auto t_entry = time_now();  // (the fallback below charges everything since here against the caller's time limit)

    // The presolver indexes its work arrays by the model's column indices and trusts rowPtr: a hand-built
    // LP_info_cpu (create_model_from_arrays validates, a caller-filled struct does not) must be refused here, before
    // any host array is touched -- without presolve DeviceMatrix::upload refuses the same models.
    if (const char *why = invalid_model_reason(model)) {
        set_last_error(why);
        std::cerr << "[error] invalid model: " << why << std::endl;
        return make_error_result("ERROR");
    }

    Presolve pre;
    bool reduced = false;
    try {
        std::cout << "Doing presolve..." << std::endl;
        reduced = pre.run(model);
        std::cout << "Presolve time: " << pre.stats().seconds << " seconds" << std::endl;
    } catch (const std::exception &e) {
        std::cerr << "[warn] presolve failed (" << e.what() << "); solving original model" << std::endl;
        reduced = false;
    }
    if (!reduced && pre.solved()) {
        // the reductions removed every row and column: the undo sequence alone produces a primal-dual optimum
        std::cout << "Presolve solved the model (no rows or columns left)" << std::endl;
        HPRLP_results r;
        std::memset(r.status, 0, sizeof(r.status));
        r.x = static_cast<double *>(std::malloc(sizeof(double) * std::max(model->n, 1)));
        r.y = static_cast<double *>(std::malloc(sizeof(double) * std::max(model->m, 1)));
        r.z = static_cast<double *>(std::malloc(sizeof(double) * std::max(model->n, 1)));
        if (!r.x || !r.y || !r.z) {
            std::free(r.x); std::free(r.y); std::free(r.z);
            return make_error_result("ERROR");
        }
        pre.postsolve(nullptr, nullptr, nullptr, r.x, r.y, r.z);
        const OriginalKkt k = original_kkt(model, r.x, r.y, r.z);
        r.primal_obj = k.primal_obj;
        r.gap = k.gap;
        r.residuals = std::max(k.primal_feas, std::max(k.dual_feas, k.gap));
        r.time = r.time4 = r.time6 = r.time8 = pre.stats().seconds;
        r.iter = r.iter4 = r.iter6 = r.iter8 = 0;
        std::strncpy(r.status, r.residuals <= p->stop_tol ? "OPTIMAL" : "ERROR", sizeof(r.status) - 1);
        if (r.residuals <= p->stop_tol) return r;
        // (cannot happen with exact arithmetic; with rounding trouble fall back to the iteration)
        std::free(r.x); std::free(r.y); std::free(r.z);
        std::cout << "Postsolve-only solution failed the KKT check; solving original model" << std::endl;
        return HPRLP_main_solve(model, p);
    }
    if (!reduced) {
        std::cout << "Presolve left the model unchanged; solving original model" << std::endl;
        return HPRLP_main_solve(model, p);
    }
    const Presolve::Stats &st = pre.stats();
    std::cout << "Presolve reduced problem: (" << model->m << ", " << model->n << ") -> (" << pre.reduced()->m << ", "
              << pre.reduced()->n << ")  [fixed cols " << st.fixed_cols << ", empty cols " << st.empty_cols
              << ", dual-fixed cols " << st.dual_fixed_cols << ", slack cols " << st.slack_cols << ", parallel rows " << st.parallel_rows << ", parallel cols " << st.parallel_cols << ", forcing rows " << st.forcing_rows << ", singleton rows " << st.singleton_rows << ", empty rows " << st.empty_rows << ", redundant rows "
              << st.redundant_rows << ", doubleton rows " << st.doubleton_rows << ", tightened bounds " << st.tightened_bounds << "; " << st.rounds
              << " rounds]" << std::endl;
    // The stopping test is relative to 1 + |b| and 1 + |c| of the model it runs on, and slack substitution moves cost
    // between columns (|c| of the reduced model can be several times the original's): the same absolute residuals would
    // then pass on the reduced model and fail the original-model check below.  Hand the reduced solve the tolerance
    // that corresponds to stop_tol on the ORIGINAL norms.
    auto norm_bc = [](const LP_info_cpu *mod, double *nb, double *nc) {
        double sb = 0.0, sc = 0.0;
        for (int i = 0; i < mod->m; ++i) {
            const double a = std::isinf(mod->AL[i]) ? 0.0 : std::abs(mod->AL[i]), b = std::isinf(mod->AU[i]) ? 0.0 : std::abs(mod->AU[i]);
            const double v = std::max(a, b);
            sb += v * v;
        }
        for (int j = 0; j < mod->n; ++j) sc += mod->c[j] * mod->c[j];
        *nb = std::sqrt(sb);
        *nc = std::sqrt(sc);
    };
    double nb0, nc0, nb1, nc1;
    norm_bc(model, &nb0, &nc0);
    norm_bc(pre.reduced(), &nb1, &nc1);
    HPRLP_parameters pr = *p;
    const double shrink = std::min(1.0, std::min((1.0 + nb0) / (1.0 + nb1), (1.0 + nc0) / (1.0 + nc1)));
    if (shrink < 1.0) {
        pr.stop_tol = p->stop_tol * shrink;
        std::cout << "Reduced-model tolerance " << pr.stop_tol << " (original norms |b| " << nb0 << ", |c| " << nc0 << "; reduced "
                  << nb1 << ", " << nc1 << ")" << std::endl;
    }
    HPRLP_results r = HPRLP_main_solve(pre.reduced(), &pr);
    if (!(r.x && r.y && r.z)) return r;
    double *x = static_cast<double *>(std::malloc(sizeof(double) * std::max(model->n, 1)));
    double *y = static_cast<double *>(std::malloc(sizeof(double) * std::max(model->m, 1)));
    double *z = static_cast<double *>(std::malloc(sizeof(double) * std::max(model->n, 1)));
    if (!x || !y || !z) {
        std::free(x); std::free(y); std::free(z);
        std::free(r.x); std::free(r.y); std::free(r.z);
        std::cerr << "[error] out of memory in postsolve" << std::endl;
        return make_error_result("ERROR");
    }
    pre.postsolve(r.x, r.y, r.z, x, y, z);
    std::free(r.x); std::free(r.y); std::free(r.z);
    r.x = x; r.y = y; r.z = z;
    if (std::strcmp(r.status, "OPTIMAL") == 0) {
        const OriginalKkt k = original_kkt(model, x, y, z);
        const double err = std::max(k.primal_feas, std::max(k.dual_feas, k.gap));
        if (err <= p->stop_tol) {
            std::cout << "Postsolve original KKT check passed" << std::endl;
        } else {
            std::cout << "Warning: postsolve original KKT check failed (the primal solution and objective are reliable)\n"
                      << "  Primal Residual: " << k.primal_feas << "  Dual Residual: " << k.dual_feas
                      << "  Relative Gap: " << k.gap << "  (tolerance " << p->stop_tol << ")" << std::endl;
            // Safety net: a reduced model that is solved to tolerance but misses the original model by orders of magnitude
            // means a reduction went wrong on this input -- solve the model as given rather than hand back a wrong answer.
            if (err > 100.0 * p->stop_tol && err > 1e-3) {
                std::cout << "Postsolved solution is far from the original model's KKT conditions; solving the original model" << std::endl;
                std::free(r.x); std::free(r.y); std::free(r.z);
                // the caller's time limit and iteration limit cover presolve + the reduced solve (its set-up and scaling
                // included: wall clock since entry) + this one; so do the reported times and counts
                const double spent = time_since(t_entry);
                const int it_first = r.iter;
                HPRLP_parameters p2 = *p;
                p2.time_limit = std::max(p->time_limit - spent, 0.0);
                p2.max_iter = std::max(p->max_iter - it_first, 0);
                HPRLP_results r2 = HPRLP_main_solve(model, &p2);
                if (std::strcmp(r2.status, "ERROR") != 0) {
                    // iter4/6/8 of a tolerance the second solve never reached are back-filled with its final count
                    // (reference src/HPRLP.cu:248-253): offset like `iter`, so they stay "<= iter"
                    r2.time += spent; r2.time4 += spent; r2.time6 += spent; r2.time8 += spent;
                    r2.iter += it_first; r2.iter4 += it_first; r2.iter6 += it_first; r2.iter8 += it_first;
                }
                std::cout << "Fallback solve: reported time and iterations include presolve and the reduced solve (" << spent
                          << " s, " << it_first << " iterations)" << std::endl;
                return r2;
            }
        }
    } else {
        std::cout << "Skipping postsolve original KKT check since the reduced solution is not optimal" << std::endl;
    }
    return r;
}

// presolve as separate steps (host only; used by the CPU tests and by callers that want the maps)
struct hprlp_presolve {
    Presolve p;
};
extern "C" hprlp_presolve *hprlp_presolve_run(const LP_info_cpu *model) {
    hprlp_presolve *h = nullptr;
    try {
        h = new hprlp_presolve();
        if (h->p.run(model)) return h;
        set_last_error("presolve left the model unchanged");
    } catch (const std::exception &e) {
        set_last_error(e.what());
    }
    delete h;
    return nullptr;
}
extern "C" const LP_info_cpu *hprlp_presolve_reduced(const hprlp_presolve *h) { return h ? h->p.reduced() : nullptr; }
extern "C" int hprlp_presolve_stats(const hprlp_presolve *h, int out[16]) {
    if (!h || !out) return -1;
    const Presolve::Stats &s = h->p.stats();
    out[0] = h->p.reduced()->m; out[1] = h->p.reduced()->n; out[2] = s.fixed_cols; out[3] = s.empty_cols;
    out[4] = s.singleton_rows; out[5] = s.empty_rows; out[6] = s.redundant_rows; out[7] = s.passes;
    out[8] = s.dual_fixed_cols; out[9] = s.slack_cols; out[10] = s.parallel_rows; out[11] = s.parallel_cols;
    out[12] = s.forcing_rows; out[13] = s.doubleton_rows; out[14] = s.tightened_bounds; out[15] = s.rounds;
    return 0;
}
extern "C" int hprlp_presolve_postsolve(const hprlp_presolve *h, const double *xr, const double *yr, const double *zr,
                                        double *x, double *y, double *z) {
    if (!h || !xr || !yr || !zr || !x || !y || !z) return -1;
    h->p.postsolve(xr, yr, zr, x, y, z);
    return 0;
}
extern "C" void hprlp_presolve_free(hprlp_presolve *h) { delete h; }
extern "C" int hprlp_original_kkt(const LP_info_cpu *model, const double *x, const double *y, const double *z,
                                  double out[5]) {
    if (!model || !model->A || !x || !y || !z || !out) return -1;
    const OriginalKkt k = original_kkt(model, x, y, z);
    out[0] = k.primal_feas; out[1] = k.dual_feas; out[2] = k.gap; out[3] = k.primal_obj; out[4] = k.dual_obj;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// step-level extension
// ------------------------------------------------------------------------------------------------
#define GUARD_BEGIN try {
#define GUARD_END(failval)                    \
    }                                         \
    catch (const std::exception &e) {         \
        set_last_error(e.what());             \
        return failval;                       \
    }

extern "C" hprlp_solver *hprlp_solver_create(const LP_info_cpu *model, const HPRLP_parameters *param) {
    if (!model) {
        set_last_error("null model");
        return nullptr;
    }
    hprlp_solver *h = nullptr;
    try {
        HPRLP_parameters dflt;
        h = new hprlp_solver();
        h->s.verbose = false;
        h->s.setup(model, param ? param : &dflt);
        return h;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        delete h;
        return nullptr;
    }
}

struct hprlp_local_group {
    LocalGroup *g = nullptr;
};
extern "C" hprlp_local_group *hprlp_local_group_create(int size) {
    try {
        auto *h = new hprlp_local_group();
        h->g = make_local_group(size);
        return h;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return nullptr;
    }
}
extern "C" void hprlp_local_group_destroy(hprlp_local_group *h) {
    if (!h) return;
    free_local_group(h->g);
    delete h;
}

// A launcher that hands over TWO unique ids (256 bytes, hprlp_dist_unique_id with a 256-byte buffer) gets a second
// communicator for the exchange stream: the exchanges that overlap the local part of a half-step are enqueued on another
// stream than the collectives of the check iterations, and one RCCL communicator must not be driven from two streams.
static void make_exchange_comm(hprlp_solver *h, int rank, int size, const void *unique_id, int id_bytes, int device) {
    if (!unique_id || id_bytes < 256) return;
    // a launcher that padded ONE id to 256 bytes: bytes 128..255 hold no id -- single communicator (the documented fallback)
    bool any = false;
    for (int k = 128; k < 256; ++k) any = any || static_cast<const unsigned char *>(unique_id)[k] != 0;
    if (!any) return;
    h->xcomm = make_rccl_comm(rank, size, static_cast<const char *>(unique_id) + 128, 128, device);
    h->s.xcomm = h->xcomm;
}

// The transport a (rank, size, id) names: a shared-memory group of processes (id made under HPRLP_DIST_TRANSPORT=shm; staging
// room = one full-length vector per rank) or RCCL communicators (one, or two when the launcher handed over two ids).
static void make_transport(hprlp_solver *h, int m, int n, int rank, int size, const void *unique_id, int id_bytes, int device) {
    if (is_shm_unique_id(unique_id, static_cast<size_t>(id_bytes))) {
        const size_t area = (static_cast<size_t>(std::max(m, n)) + 64u * static_cast<size_t>(size)) * sizeof(double) + 4096u * static_cast<size_t>(size + 1);
        h->comm = make_shm_comm(rank, size, unique_id, static_cast<size_t>(id_bytes), area, device);
        return;
    }
    h->comm = make_rccl_comm(rank, size, unique_id, static_cast<size_t>(id_bytes), device);
    make_exchange_comm(h, rank, size, unique_id, id_bytes, device);
}

static void destroy_handle(hprlp_solver *h) {
    if (!h) return;
    Comm *c = h->comm, *x = h->xcomm;
    delete h;  // the solver first: its device state (and a background tiling job) goes before the transport
    delete x;
    delete c;
}

// rank `rank` of `size`: RCCL communicator from unique_id, or (group != NULL) the in-process group
static hprlp_solver *create_sharded(const LP_info_cpu *model, const HPRLP_parameters *param, int rank, int size,
                                    const void *unique_id, int id_bytes, hprlp_local_group *group) {
    if (!model) {
        set_last_error("null model");
        return nullptr;
    }
    hprlp_solver *h = nullptr;
    hprlp_shard sh;
    std::memset(&sh, 0, sizeof(sh));
    try {
        HPRLP_parameters dflt;
        const HPRLP_parameters *p = param ? param : &dflt;
        if (hprlp_extract_shard(model, rank, size, &sh) != 0) throw std::runtime_error(last_error_cstr());
        h = new hprlp_solver();
        h->s.verbose = false;
        HIP_CHECK(hipSetDevice(p->device_number));
        if (group) {
            h->comm = make_local_comm(group->g, rank);
        } else if (size > 1 || (unique_id && id_bytes >= 128)) {
            // a unique id given with size 1 builds a one-rank RCCL communicator (exercises the collective path)
            make_transport(h, sh.m, sh.n, rank, size, unique_id, id_bytes, p->device_number);
        }
        h->s.setup_shard(sh.m, sh.n, sh.row_off, sh.m_loc, sh.col_off, sh.n_loc, sh.A_rowptr, sh.A_col, sh.A_val,
                         sh.AT_rowptr, sh.AT_col, sh.AT_val, sh.AL, sh.AU, sh.l, sh.u, sh.c, sh.obj_constant, p, h->comm);
        hprlp_free_shard(&sh);
        return h;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        destroy_handle(h);  // the solver first: a background tiling job may still read the shard arrays (its destructor joins it)
        hprlp_free_shard(&sh);
        return nullptr;
    }
}

extern "C" hprlp_solver *hprlp_solver_create_dist(const LP_info_cpu *model, const HPRLP_parameters *param, int rank,
                                                  int size, const void *unique_id, int id_bytes) {
    return create_sharded(model, param, rank, size, unique_id, id_bytes, nullptr);
}

// The same from a shard the caller built itself (no rank ever holds the whole matrix: SURVEY.md 8d, config 5 "generated per
// shard"); the shard stays the caller's.
static hprlp_solver *create_from_shard(const hprlp_shard *sh, const HPRLP_parameters *param, int rank, int size, const void *unique_id,
                                       int id_bytes, hprlp_local_group *group) {
    hprlp_solver *h = nullptr;
    try {
        if (!sh || !sh->A_rowptr || !sh->AT_rowptr || (sh->m_loc > 0 && !(sh->AL && sh->AU)) || (sh->n_loc > 0 && !(sh->l && sh->u && sh->c)))
            throw std::runtime_error("incomplete shard");
        if (sh->A_rowptr[sh->m_loc] > 0 && !(sh->A_col && sh->A_val)) throw std::runtime_error("incomplete shard (A)");
        if (sh->AT_rowptr[sh->n_loc] > 0 && !(sh->AT_col && sh->AT_val)) throw std::runtime_error("incomplete shard (A^T)");
        HPRLP_parameters dflt;
        const HPRLP_parameters *p = param ? param : &dflt;
        h = new hprlp_solver();
        h->s.verbose = false;
        HIP_CHECK(hipSetDevice(p->device_number));
        if (group) h->comm = make_local_comm(group->g, rank);
        else if (size > 1 || (unique_id && id_bytes >= 128)) {
            make_transport(h, sh->m, sh->n, rank, size, unique_id, id_bytes, p->device_number);
        }
        h->s.setup_shard(sh->m, sh->n, sh->row_off, sh->m_loc, sh->col_off, sh->n_loc, sh->A_rowptr, sh->A_col, sh->A_val, sh->AT_rowptr,
                         sh->AT_col, sh->AT_val, sh->AL, sh->AU, sh->l, sh->u, sh->c, sh->obj_constant, p, h->comm);
        return h;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        destroy_handle(h);
        return nullptr;
    }
}

extern "C" hprlp_solver *hprlp_solver_create_dist_from_shard(const hprlp_shard *shard, const HPRLP_parameters *param, int rank, int size,
                                                             const void *unique_id, int id_bytes) {
    return create_from_shard(shard, param, rank, size, unique_id, id_bytes, nullptr);
}

extern "C" hprlp_solver *hprlp_solver_create_local_from_shard(const hprlp_shard *shard, const HPRLP_parameters *param, int rank, int size,
                                                              hprlp_local_group *group) {
    if (!group) {
        set_last_error("null local group");
        return nullptr;
    }
    return create_from_shard(shard, param, rank, size, nullptr, 0, group);
}

extern "C" hprlp_solver *hprlp_solver_create_local(const LP_info_cpu *model, const HPRLP_parameters *param, int rank,
                                                   int size, hprlp_local_group *group) {
    if (!group) {
        set_last_error("null local group");
        return nullptr;
    }
    return create_sharded(model, param, rank, size, nullptr, 0, group);
}

// The shared-memory transport's protocol on HOST buffers (no GPU, no HIP call): rank `rank` of `size` attaches to the group the
// id names and runs `rounds` rounds of all-gather, scalar all-reduce and a neighbour exchange whose payloads every receiver
// checks.  hang_rank >= 0: that rank leaves before round `rounds / 2` without a word (the others must end with an error once
// HPRLP_DIST_TIMEOUT_S has passed, not wait for ever).  Returns 0, or -1 + hprlp_last_error().
extern "C" int hprlp_shm_transport_selftest(const void *unique_id, int id_bytes, int rank, int size, int rounds, int hang_rank) {
    try {
        const int chunk = 1000 + 7;
        std::unique_ptr<Comm> c(make_shm_comm(rank, size, unique_id, static_cast<size_t>(id_bytes), sizeof(double) * chunk * static_cast<size_t>(size), -1));
        std::vector<double> g(static_cast<size_t>(chunk) * size), sc(3);
        std::vector<std::vector<double>> snd(size), rcv(size);
        for (int it = 0; it < rounds; ++it) {
            if (rank == hang_rank && it == rounds / 2) return 0;
            std::fill(g.begin(), g.end(), -1.0);
            for (int j = 0; j < chunk; ++j) g[static_cast<size_t>(rank) * chunk + j] = 1e6 * it + 1e3 * rank + j;
            c->allgather_inplace(g.data(), chunk, nullptr);
            for (int p = 0; p < size; ++p)
                for (int j = 0; j < chunk; ++j)
                    if (g[static_cast<size_t>(p) * chunk + j] != 1e6 * it + 1e3 * p + j) throw std::runtime_error("all-gather delivered a wrong entry");
            sc[0] = rank + 1.0; sc[1] = it; sc[2] = 0.1 * (rank + 1);
            c->allreduce_sum(sc.data(), 3, nullptr);
            double want2 = 0.0;
            for (int p = 0; p < size; ++p) want2 += 0.1 * (p + 1);
            if (sc[0] != size * (size + 1) / 2.0 || sc[1] != static_cast<double>(it) * size || sc[2] != want2) throw std::runtime_error("all-reduce gave a wrong sum");
            // rank a sends (a + 2 b + it) % 5 + (b > a) doubles to rank b: ragged, some empty
            std::vector<P2P> ops;
            for (int p = 0; p < size; ++p) {
                if (p == rank) continue;
                const size_t ns = static_cast<size_t>((rank + 2 * p + it) % 5 + (p > rank)), nr = static_cast<size_t>((p + 2 * rank + it) % 5 + (rank > p));
                snd[p].assign(ns, 0.0);
                for (size_t j = 0; j < ns; ++j) snd[p][j] = 100.0 * rank + p + 0.001 * static_cast<double>(j) + it;
                rcv[p].assign(nr, -7.0);
                if (ns || nr) ops.push_back(P2P{p, snd[p].data(), ns * sizeof(double), rcv[p].data(), nr * sizeof(double)});
            }
            c->exchange(ops.data(), static_cast<int>(ops.size()), nullptr);
            for (int p = 0; p < size; ++p)
                for (size_t j = 0; p != rank && j < rcv[p].size(); ++j)
                    if (rcv[p][j] != 100.0 * p + rank + 0.001 * static_cast<double>(j) + it) throw std::runtime_error("exchange delivered a wrong entry");
        }
        return 0;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return -1;
    }
}

// out = {halo_m sparse?, entries sent, entries received, halo_n sparse?, sent, received, requests m, requests n}
extern "C" int hprlp_solver_dist_info(hprlp_solver *h, long out[8]) {
    if (!h || !out) return -1;
    const Solver &s = h->s;
    out[0] = s.halo_m.sparse; out[1] = s.halo_m.nsend; out[2] = s.halo_m.nrecv;
    out[3] = s.halo_n.sparse; out[4] = s.halo_n.nsend; out[5] = s.halo_n.nrecv;
    out[6] = s.halo_m.total_requests; out[7] = s.halo_n.total_requests;
    return 0;
}

// out = {ranks / this rank / device as the main communicator reports them (RCCL: ncclCommCount, ncclCommUserRank,
//        ncclCommCuDevice), the same three of the exchange stream's communicator (0, -1, -1 if there is none), the
//        solver's HIP device, 1 if the exchange overlaps the local part of the half-steps}
extern "C" int hprlp_solver_dist_comm_info(hprlp_solver *h, long out[8]) {
    GUARD_BEGIN
    if (!h || !out) throw std::runtime_error("null solver / output");
    const Solver &s = h->s;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    out[1] = out[2] = out[4] = out[5] = -1;
    if (s.comm) { out[0] = s.comm->reported_ranks(); out[1] = s.comm->reported_rank(); out[2] = s.comm->reported_device(); }
    if (s.xcomm) { out[3] = s.xcomm->reported_ranks(); out[4] = s.xcomm->reported_rank(); out[5] = s.xcomm->reported_device(); }
    out[6] = s.prm.device_number;
    out[7] = s.overlap_enabled ? 1 : 0;
    return 0;
    GUARD_END(-1)
}

// One grouped send/recv of `count` doubles from this rank to ITSELF through the solver's communicator (pattern
// i -> 3 i + 1), checked on the host: exercises the point-to-point entry points of the transport (RCCL: ncclGroupStart /
// ncclSend / ncclRecv / ncclGroupEnd) on a box where no second rank can exist.  0 = delivered intact.
extern "C" int hprlp_solver_dist_loopback(hprlp_solver *h, int count) {
    GUARD_BEGIN
    Solver &s = h->s;
    if (!s.comm) throw std::runtime_error("solver has no communicator");
    if (count <= 0) throw std::runtime_error("count must be positive");
    std::vector<double> src(static_cast<size_t>(count)), dst(static_cast<size_t>(count), -1.0);
    for (int i = 0; i < count; ++i) src[i] = 3.0 * i + 1.0;
    DBuf<double> a(src.size()), b(dst.size());
    a.upload(src.data(), src.size());
    b.upload(dst.data(), dst.size());
    const P2P op{s.comm->rank, a.p, src.size() * sizeof(double), b.p, dst.size() * sizeof(double)};
    s.comm->exchange(&op, 1, s.stream);
    HIP_CHECK(hipStreamSynchronize(s.stream));
    b.download(dst.data(), dst.size());
    for (int i = 0; i < count; ++i)
        if (dst[i] != src[i]) throw std::runtime_error("loopback send/recv delivered wrong data at element " + std::to_string(i));
    return 0;
    GUARD_END(-1)
}

extern "C" void hprlp_solver_destroy(hprlp_solver *h) {
    try {
        destroy_handle(h);
    } catch (...) {
    }
}

extern "C" void hprlp_solver_set_verbose(hprlp_solver *h, int verbose) {
    if (h) h->s.verbose = verbose != 0;
}

extern "C" int hprlp_solver_scale(hprlp_solver *h) {
    GUARD_BEGIN
    h->s.scale();
    return 0;
    GUARD_END(-1)
}

extern "C" double hprlp_solver_power_iteration(hprlp_solver *h, int max_iter, double tol, int *iters_out) {
    GUARD_BEGIN
    return h->s.power_iteration(max_iter, tol, iters_out);
    GUARD_END(-1.0)
}

extern "C" int hprlp_solver_init(hprlp_solver *h, double sigma, double lambda_max) {
    GUARD_BEGIN
    h->s.lambda_max = lambda_max;
    if (sigma > 0) h->s.set_sigma_lambda(sigma, lambda_max, true);
    else h->s.init_iteration_state();
    HIP_CHECK(hipStreamSynchronize(h->s.stream));
    return 0;
    GUARD_END(-1)
}

extern "C" int hprlp_solver_reset_iterates(hprlp_solver *h) {
    GUARD_BEGIN
    if (!h) throw std::runtime_error("null solver");
    h->s.reset_iterates();
    return 0;
    GUARD_END(-1)
}

extern "C" int hprlp_solver_iterate(hprlp_solver *h, int normal, int then_check) {
    GUARD_BEGIN
    if (then_check) h->s.run_normal_then_check(normal);
    else h->s.run_normal(normal);
    HIP_CHECK(hipStreamSynchronize(h->s.stream));
    return 0;
    GUARD_END(-1)
}

extern "C" int hprlp_solver_residuals(hprlp_solver *h, int iter, int compute_gap, double out[8]) {
    GUARD_BEGIN
    Residuals r;
    RestartState rs;
    h->s.compute_residuals(iter, compute_gap != 0, &r, &rs);
    out[0] = r.err_Rp; out[1] = r.err_Rd; out[2] = r.primal_obj; out[3] = r.dual_obj;
    out[4] = r.rel_gap; out[5] = r.kkt; out[6] = rs.current_gap; out[7] = h->s.lambda_max;
    return 0;
    GUARD_END(-1)
}

extern "C" int hprlp_solver_restart(hprlp_solver *h, const double in[6], double *sigma_out) {
    GUARD_BEGIN
    RestartState rs;
    rs.flag = 1;
    rs.first = false;
    rs.current_gap = in[0]; rs.best_gap = in[1]; rs.best_sigma = in[2];
    Residuals r;
    r.err_Rd = in[3]; r.err_Rp = in[4]; r.rel_gap = in[5];
    h->s.update_sigma_and_restart(&rs, r);
    HIP_CHECK(hipStreamSynchronize(h->s.stream));
    if (sigma_out) *sigma_out = h->s.sigma;
    return 0;
    GUARD_END(-1)
}

extern "C" double hprlp_solver_weighted_norm(hprlp_solver *h) {
    GUARD_BEGIN
    return h->s.weighted_norm_after_restart();
    GUARD_END(-1.0)
}

extern "C" int hprlp_solver_run(hprlp_solver *h, HPRLP_results *out, hprlp_trace_row *trace, int max_trace,
                                int *n_trace) {
    GUARD_BEGIN
    static_assert(sizeof(hprlp_trace_row) == sizeof(TraceRow), "trace row layout");
    h->s.trace = reinterpret_cast<TraceRow *>(trace);
    h->s.trace_cap = trace ? max_trace : 0;
    h->s.solve_loop(out);
    h->s.collect_solution(out);
    if (n_trace) *n_trace = h->s.trace_n;
    h->s.trace = nullptr;
    h->s.trace_cap = 0;
    return 0;
    GUARD_END(-1)
}

namespace {
struct VecRef {
    double *p;
    long n;
    char kind;  // 'r': per row of A, 'c': per column, '-': raw (matrix values)
};
VecRef find_vector(Solver &s, const std::string &name) {
    if (name == "x") return {s.x.p, s.n_loc, 'c'};
    if (name == "last_x") return {s.last_x.p, s.n_loc, 'c'};
    if (name == "x_hat") return {s.x_hat, s.n_loc, 'c'};
    if (name == "x_bar") return {s.x_bar, s.n_loc, 'c'};
    if (name == "z_bar") return {s.z_bar.p, s.n_loc, 'c'};
    if (name == "x_temp") return {s.x_temp, s.n_loc, 'c'};
    if (name == "y") return {s.y, s.m_loc, 'r'};
    if (name == "last_y") return {s.last_y.p, s.m_loc, 'r'};
    if (name == "y_bar") return {s.y_bar, s.m_loc, 'r'};
    if (name == "y_obj") return {s.y_obj.p, s.m_loc, 'r'};
    if (name == "y_temp") return {s.y_temp.p, s.m_loc, 'r'};
    if (name == "AL") return {s.AL.p, s.m_loc, 'r'};
    if (name == "AU") return {s.AU.p, s.m_loc, 'r'};
    if (name == "l") return {s.l.p, s.n_loc, 'c'};
    if (name == "u") return {s.u.p, s.n_loc, 'c'};
    if (name == "c") return {s.c.p, s.n_loc, 'c'};
    if (name == "row_norm") return {s.row_norm.p, s.m_loc, 'r'};
    if (name == "col_norm") return {s.col_norm.p, s.n_loc, 'c'};
    if (name == "A_val") return {s.A.val.p, s.A.view.nnz, '-'};
    if (name == "AT_val") return {s.AT.val.p, s.AT.view.nnz, '-'};
    return {nullptr, -1, '-'};
}
// the permutation (device index -> caller's index) that applies to a vector, or null
const std::vector<int> *perm_of(const Solver &s, const VecRef &v) {
    if (v.kind == 'r' && !s.perm_r.empty()) return &s.perm_r;
    if (v.kind == 'c' && !s.perm_c.empty()) return &s.perm_c;
    return nullptr;
}
}  // namespace

// Vectors travel in the caller's numbering: with a set-up time locality ordering in place (solver.h: perm_r / perm_c)
// the device order is un-permuted on the way out and permuted on the way in.  A_val / AT_val are the device arrays as
// they are (entries of the permuted matrix).
extern "C" long hprlp_solver_get_vector(hprlp_solver *h, const char *name, double *out, long cap) {
    GUARD_BEGIN
    VecRef v = find_vector(h->s, name ? name : "");
    if (v.n < 0) throw std::runtime_error(std::string("unknown vector name: ") + (name ? name : "(null)"));
    if (cap < v.n) throw std::runtime_error("output buffer too small");
    HIP_CHECK(hipStreamSynchronize(h->s.stream));
    if (v.n > 0) {
        if (const std::vector<int> *perm = perm_of(h->s, v)) {
            std::vector<double> tmp(static_cast<size_t>(v.n));
            HIP_CHECK(hipMemcpy(tmp.data(), v.p, sizeof(double) * v.n, hipMemcpyDeviceToHost));
            for (long i = 0; i < v.n; ++i) out[(*perm)[i]] = tmp[i];
        } else {
            HIP_CHECK(hipMemcpy(out, v.p, sizeof(double) * v.n, hipMemcpyDeviceToHost));
        }
    }
    return v.n;
    GUARD_END(-1)
}

extern "C" int hprlp_solver_set_vector(hprlp_solver *h, const char *name, const double *in, long len) {
    GUARD_BEGIN
    VecRef v = find_vector(h->s, name ? name : "");
    if (v.n < 0) throw std::runtime_error(std::string("unknown vector name: ") + (name ? name : "(null)"));
    if (len != v.n) throw std::runtime_error("length mismatch");
    h->s.invalidate_far();
    HIP_CHECK(hipStreamSynchronize(h->s.stream));
    if (v.n > 0) {
        if (const std::vector<int> *perm = perm_of(h->s, v)) {
            std::vector<double> tmp(static_cast<size_t>(v.n));
            for (long i = 0; i < v.n; ++i) tmp[i] = in[(*perm)[i]];
            HIP_CHECK(hipMemcpy(v.p, tmp.data(), sizeof(double) * v.n, hipMemcpyHostToDevice));
        } else {
            HIP_CHECK(hipMemcpy(v.p, in, sizeof(double) * v.n, hipMemcpyHostToDevice));
        }
    }
    if (v.p == h->s.l.p || v.p == h->s.u.p || v.p == h->s.AL.p || v.p == h->s.AU.p) {  // the bound codes follow the arrays
        h->s.refresh_bound_codes();
        HIP_CHECK(hipStreamSynchronize(h->s.stream));
    }
    return 0;
    GUARD_END(-1)
}

// Locality ordering of a CSR pattern (reorder.cpp), host only: out = {accepted (0/1), tiled share of the entries before,
// after, clusters, components, seconds}; the permutations (new -> old) are written only when accepted.
extern "C" int hprlp_locality_ordering(int m, int n, const int *rowptr, const int *col, int *row_new2old, int *col_new2old, double out[6]) {
    try {
        if (m <= 0 || n <= 0 || !rowptr || !col || !row_new2old || !col_new2old) throw std::runtime_error("bad arguments");
        std::vector<int> pr, pc;
        ReorderStats st;
        const bool ok = locality_ordering(m, n, rowptr, col, &pr, &pc, &st);
        if (ok) {
            std::copy(pr.begin(), pr.end(), row_new2old);
            std::copy(pc.begin(), pc.end(), col_new2old);
        }
        if (out) {
            out[0] = ok ? 1.0 : 0.0; out[1] = st.fraction_before; out[2] = st.fraction_after;
            out[3] = st.clusters; out[4] = st.components; out[5] = st.seconds;
        }
        return 0;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return -1;
    }
}

// Host only: the stream kernel's row blocks of a CSR pattern (split rows, chunks by column eighths), built and checked as a
// solver's set-up does.  out = {blocks, split rows, chunk slots, rows cut by column eighths, longest chunk, entries covered}.
extern "C" int hprlp_row_block_plan(int m, int n, const int *rowptr, const int *col, int with_cuts, long out[6]) {
    try {
        if (m <= 0 || n <= 0 || !rowptr || !col || !out) throw std::runtime_error("bad arguments");
        row_block_plan_host(m, n, rowptr, col, with_cuts != 0, out);
        return 0;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return -1;
    }
}

extern "C" int hprlp_solver_get_scalars(hprlp_solver *h, double out[16]) {
    GUARD_BEGIN
    Solver &s = h->s;
    Ctrl ck;
    HIP_CHECK(hipStreamSynchronize(s.stream));
    HIP_CHECK(hipMemcpy(&ck, s.ctrl.p, sizeof(Ctrl), hipMemcpyDeviceToHost));
    const double v[16] = {s.b_scale, s.c_scale, s.norm_b, s.norm_c, s.norm_b_org, s.norm_c_org, s.sigma, s.lambda_max,
                          s.setup_time, s.scaling_time, s.power_time, static_cast<double>(s.power_iters),
                          static_cast<double>(ck.kx), static_cast<double>(ck.ky), 0, 0};
    for (int i = 0; i < 16; ++i) out[i] = v[i];
    return 0;
    GUARD_END(-1)
}

extern "C" int hprlp_solver_info(hprlp_solver *h, long out[8]) {
    GUARD_BEGIN
    Solver &s = h->s;
    s.finish_tiling();  // report the kernels that will actually run
    out[0] = s.m; out[1] = s.n; out[2] = s.A.view.nnz;
    out[3] = s.A.view.nblk; out[4] = s.AT.view.nblk;
    out[5] = s.A.view.grid(); out[6] = s.AT.view.grid();
    // bit0: A tiled, bit1: A^T tiled, bit2: normal iterations run in the single-workgroup small-LP kernel
    // bit3: a set-up time locality ordering is in place (the device works on P A Q)
    out[7] = (s.A.view.tiled.valid ? 1 : 0) + (s.AT.view.tiled.valid ? 2 : 0) + (s.use_small && !s.comm ? 4 : 0) + (s.perm_r.empty() ? 0 : 8);
    return 0;
    GUARD_END(-1)
}

// One line per matrix saying which kernel form runs and on what structure (bench.py's ladder, tools/run_mps_dir.py)
extern "C" int hprlp_solver_describe(hprlp_solver *h, char *buf, int cap) {
    GUARD_BEGIN
    if (!h || !buf || cap <= 0) throw std::runtime_error("null solver / buffer");
    Solver &s = h->s;
    s.finish_tiling();
    auto one = [](const char *name, const DeviceMatrix &M) {
        const TiledDev &t = M.view.tiled;
        std::string d = std::string(name) + ": ";
        if (!t.valid) {
            d += "stream kernel (k_spmv_fused, " + std::to_string(M.view.nblk) + " row blocks, " + std::to_string(M.view.nlong) + " split rows)";
            if (M.declined_skew) d += " [tiled form not attempted: too many entries in long rows]";
            else if (M.declined_imbalance) d += " [tiled form not attempted: unbalanced row blocks]";
            else if (M.declined_coalesced) d += " [tiled form not attempted: neighbouring rows gather from the same lines]";
            else if (M.declined_l2) d += " [tiled form not attempted: the stream kernel's gathers stay in one L2]";
            else if (M.declined_shape) d += " [tiled form not attempted: shape]";
            else if (M.declined_sparse) d += " [tiled form declined: too few entries in dense tiles]";
            else if (M.declined_thin) d += " [tiled piece form declined: thin rows]";
            else if (M.declined_popular) d += " [tiled form declined: its remainder gathers from a few popular columns]";
            else if (M.declined_few_rows) d += " [tiled form not attempted: too few rows]";
            return d;
        }
        d += t.n_pieces > 0 ? "tiled, piece form (k_tiled_part + k_tiled_finish, " + std::to_string(t.n_pieces) + " pieces)"
             : t.rem_cap == kPbRemCap ? "tiled, all-remainder form (k_pb_fused, grid " + std::to_string(t.grid) + ")"
                                      : "tiled, fused (k_tiled_fused, grid " + std::to_string(t.grid) + ")";
        d += ", " + std::to_string(t.nsb) + " super-blocks, " + std::to_string(M.tiled.n_steps) + " steps";
        if (t.R != kTileRows || t.T != kTileCols) d += " (" + std::to_string(t.R) + " rows, tiles of " + std::to_string(t.T) + " columns)";
        const double all = static_cast<double>(M.tiled.dense_entries) + static_cast<double>(M.tiled.n_rem);
        if (all > 0) d += ", " + std::to_string(static_cast<int>(100.0 * M.tiled.dense_entries / all + 0.5)) + " % of the entries in staged tiles";
        if (t.f_rk) d += ", source-side run tables";
        if (t.f_work) d += ", pre-pass work list of " + std::to_string(t.n_work) + " workgroups for " + std::to_string(t.n_groups) + " source groups";
        if (t.side_nblk > 0) d += ", long rows aside (" + std::to_string(t.side_nblk) + " blocks through the stream kernel)";
        return d;
    };
    std::string d = one("A", s.A) + "; " + one("A^T", s.AT);
    if (s.use_small && !s.comm) d += "; normal iterations in the single-workgroup kernel (k_small_iterations)";
    if (!s.perm_r.empty()) d += "; locality ordering applied at set-up";
    // (anything but the default path says so: the switches this solver was set up under, and hooks that were set but not honoured)
    if (!s.env_at_setup.empty()) d += "; switches: " + s.env_at_setup;
    if (!s.env_ignored_at_setup.empty()) d += "; ignored without HPRLP_TEST_HOOKS=1: " + s.env_ignored_at_setup;
    std::snprintf(buf, static_cast<size_t>(cap), "%s", d.c_str());
    return static_cast<int>(d.size());
    GUARD_END(-1)
}

// Host only: the column-tiled structure of a CSR pattern (host builder, tiled.cpp) with super-blocks of R rows and tiles of T
// columns, verified entry by entry (every entry once, codes name their entries, one chunk per accumulator inside a step, layers).
// out = {tile entries incl. padding, remainder entries, steps, padding, most consecutive steps of one tile, staged share x 1e6}.
extern "C" int hprlp_tiled_host_check(int m, int n, const int *rowptr, const int *col, int R, int T, double min_dense, long out[6]) {
    try {
        if (m <= 0 || n <= 0 || !rowptr || !col || !out) throw std::runtime_error("bad arguments");
        tiled_host_check(m, n, rowptr, col, R, T, min_dense, out);
        return 0;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return -1;
    }
}

// The table of environment switches (env.h) as text, one per line: "<name>\t<integrator|hook>\t<what>"; returns the length.
extern "C" int hprlp_env_switches(char *buf, int cap) {
    int count = 0;
    const EnvEntry *t = env_table(&count);
    std::string d;
    for (int i = 0; i < count; ++i) d += std::string(t[i].name) + "\t" + (t[i].kind == EnvKind::Integrator ? "integrator" : "hook") + "\t" + t[i].what + "\n";
    if (buf && cap > 0) std::snprintf(buf, static_cast<size_t>(cap), "%s", d.c_str());
    return static_cast<int>(d.size());
}

extern "C" int hprlp_solver_time_iterations(hprlp_solver *h, int warmup, int steps, int mode, double *total_ms,
                                            double *xhalf_ms, double *yhalf_ms) {
    GUARD_BEGIN
    Solver &s = h->s;
    if (steps <= 0) throw std::runtime_error("steps must be positive");
    s.run_normal(warmup);
    HIP_CHECK(hipStreamSynchronize(s.stream));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    double tx = 0, ty = 0;
    float ms = 0;
    if (mode == 0) {
        HIP_CHECK(hipEventRecord(e0, s.stream));
        s.run_normal(steps);
        HIP_CHECK(hipEventRecord(e1, s.stream));
        HIP_CHECK(hipEventSynchronize(e1));
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    } else if (mode == 2) {
        s.invalidate_far();
        // the bare SpMVs of the path (A^T y into scratch, A x_hat into scratch): no half-step update, no state change
        std::vector<hipEvent_t> ev(static_cast<size_t>(steps) * 3);
        for (auto &e : ev) HIP_CHECK(hipEventCreate(&e));
        HIP_CHECK(hipEventRecord(e0, s.stream));
        for (int i = 0; i < steps; ++i) {
            HIP_CHECK(hipEventRecord(ev[3 * i], s.stream));
            launch_spmv_plain(s.AT.view, s.gy.p, s.sn1.p, nullptr, false, nullptr, 0, s.stream);
            HIP_CHECK(hipEventRecord(ev[3 * i + 1], s.stream));
            launch_spmv_plain(s.A.view, s.gxh.p, s.sm1.p, nullptr, false, nullptr, 0, s.stream);
            HIP_CHECK(hipEventRecord(ev[3 * i + 2], s.stream));
        }
        HIP_CHECK(hipEventRecord(e1, s.stream));
        HIP_CHECK(hipEventSynchronize(e1));
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        for (int i = 0; i < steps; ++i) {
            float a = 0, b = 0;
            HIP_CHECK(hipEventElapsedTime(&a, ev[3 * i], ev[3 * i + 1]));
            HIP_CHECK(hipEventElapsedTime(&b, ev[3 * i + 1], ev[3 * i + 2]));
            tx += a;
            ty += b;
        }
        for (auto &e : ev) (void)hipEventDestroy(e);
    } else {
        std::vector<hipEvent_t> ev(static_cast<size_t>(steps) * 3);
        for (auto &e : ev) HIP_CHECK(hipEventCreate(&e));
        // the solver's own normal pair (with the multi-GPU overlap when it is on): events before / between / after the halves
        HIP_CHECK(hipEventRecord(e0, s.stream));
        for (int i = 0; i < steps; ++i) s.launch_normal_pair(i + 1 < steps, &ev[3 * static_cast<size_t>(i)], s.x_mode_of(i, steps));
        HIP_CHECK(hipEventRecord(e1, s.stream));
        HIP_CHECK(hipEventSynchronize(e1));
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        for (int i = 0; i < steps; ++i) {
            float a = 0, b = 0;
            HIP_CHECK(hipEventElapsedTime(&a, ev[3 * i], ev[3 * i + 1]));
            HIP_CHECK(hipEventElapsedTime(&b, ev[3 * i + 1], ev[3 * i + 2]));
            tx += a;
            ty += b;
        }
        for (auto &e : ev) (void)hipEventDestroy(e);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (total_ms) *total_ms = ms;
    if (xhalf_ms) *xhalf_ms = tx;
    if (yhalf_ms) *yhalf_ms = ty;
    return 0;
    GUARD_END(-1)
}
