// env.cpp -- the table behind env.h: every environment switch the library reads.
#include "env.h"

#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace hprlp {

static const EnvEntry kTable[] = {
    {"HPRLP_TEST_HOOKS", EnvKind::Integrator, "1: honour the test hooks below (tests/conftest.py and the tools/ scripts set it)"},
    {"HPRLP_TIMING", EnvKind::Integrator, "wall time of the set-up and solve phases on stderr"},
    {"HPRLP_NO_GRAPH", EnvKind::Integrator, "launch normal iterations eagerly instead of replaying hipGraphs (fall-back)"},
    {"HPRLP_NO_TILED", EnvKind::Integrator, "never build the column-tiled matrix copies: stream kernel everywhere (fall-back)"},
    {"HPRLP_NO_REORDER", EnvKind::Integrator, "no set-up time locality ordering of a large matrix"},
    {"HPRLP_NO_ALLOC_CACHE", EnvKind::Integrator, "do not keep freed device blocks for the next solver of the process"},
    {"HPRLP_DIST_TRANSPORT", EnvKind::Integrator, "shm: hprlp_dist_unique_id names a shared-memory segment (host-staged group of processes on one node) instead of RCCL ids"},
    {"HPRLP_NO_WARM_MODEL", EnvKind::Integrator, "the model constructors do NOT start the HIP runtime / device context / code objects (for a process that forks workers after building its models: a HIP context does not survive fork); the first solve pays for them as in the reference"},
    {"HPRLP_DIST_EXCHANGE", EnvKind::Hook, "sparse | allgather: force the multi-GPU exchange form (same value on every rank)"},
    {"HPRLP_NO_OVERLAP", EnvKind::Integrator, "multi-GPU: shards unsplit, exchange in line on the solver stream"},
    {"HPRLP_DIST_TIMEOUT_S", EnvKind::Integrator, "seconds a rank of the shared-memory transport waits for its peers before it fails (default 120)"},
    {"HPRLP_BATCH_CHUNK", EnvKind::Hook, "solve_batched with 64 or more problems: chunk width of the panels (default 64; narrower chunks map one chunk to each XCD -- measured slower, profiles/r03_tiled_decomposition.md section 6b)"},
    {"HPRLP_BATCH_GRID", EnvKind::Hook, "solve_batched: workgroups of the half-step kernels"},
    {"HPRLP_COPY_PAUSE_US", EnvKind::Hook, "piece size and pause of the threaded copies of set-up (64 MB, 100 us)"},
    {"HPRLP_COPY_PIECE_MB", EnvKind::Hook, "piece size and pause of the threaded copies of set-up (64 MB, 100 us)"},
    {"HPRLP_DEVICE_TRANSPOSE_MIN", EnvKind::Hook, "build Aᵀ on the host / nonzero threshold of the device transpose (default 4 M)"},
    {"HPRLP_DIST_SELFTEST_FAIL", EnvKind::Hook, "tests only: the set-up self-test of the neighbour exchange reports a failure once (exercises the fall-back to the all-gather)"},
    {"HPRLP_DTON_MAX", EnvKind::Hook, "presolve: cap of the doubleton stage"},
    {"HPRLP_DTON_TRACE", EnvKind::Hook, "presolve: trace the doubleton stage on stderr"},
    {"HPRLP_GEN_FAR", EnvKind::Hook, "benchmark generator: share of a row's entries that fall anywhere (default 0.05)"},
    {"HPRLP_HOST_POWER_START", EnvKind::Hook, "power iteration's start vector made on the host also above 1e6 rows (default there: on the device)"},
    {"HPRLP_HOST_TILING", EnvKind::Hook, "build the tiled copies with the host builder (background threads, default min(16, cores)) instead of on the device"},
    {"HPRLP_HOST_TRANSPOSE", EnvKind::Hook, "build Aᵀ on the host / nonzero threshold of the device transpose (default 4 M)"},
    {"HPRLP_NO_BOUND_CODES", EnvKind::Hook, "the x-half always reads l[j] and u[j], the y-half AL[i] and AU[i] (default: one code byte per column / row says which of them is not a constant; an equality row reads one value)"},
    {"HPRLP_NO_FAR_PUSH", EnvKind::Hook, "always run the remainder pre-pass (k_far_products) instead of the hand-off from the producing half-step's epilogue"},
    {"HPRLP_NO_FAR_WORK", EnvKind::Hook, "the remainder pre-pass with one workgroup per source group also where one group holds several times the mean (default: a work list cuts heavy groups into chunks, tiled.h f_work)"},
    {"HPRLP_NO_PB_LONG_ROWS", EnvKind::Hook, "a matrix kept off the tiled forms for its long rows keeps the stream kernel also where its columns are not popular (default: all-remainder form, Solver::pb_fallback_wanted)"},
    {"HPRLP_NO_FUSED_NORMS", EnvKind::Hook, "the Ruiz row norms by their own passes instead of as a by-product of the preceding scaling pass (same bits)"},
    {"HPRLP_NO_LONG_SIDE", EnvKind::Hook, "a matrix with rows over 1024 entries keeps the stream kernel instead of being tiled with those rows kept aside"},
    {"HPRLP_NO_PB_FALLBACK", EnvKind::Hook, "unstructured large matrices keep the stream kernel instead of the tiled form without dense-tile requirement"},
    {"HPRLP_NO_PB_KERNEL", EnvKind::Hook, "run an all-remainder copy through k_tiled_fused's remainder steps instead of k_pb_fused (A/B)"},
    {"HPRLP_NO_REM2", EnvKind::Hook, "tiled build: no second remainder level"},
    {"HPRLP_NO_SETUP_OVERLAP", EnvKind::Hook, "set-up of a large model in line: no helper threads / copy streams for the value upload, the row blocks, A^T's row pointers and the model vectors (same bits either way)"},
    {"HPRLP_NO_SLAB_CUTS", EnvKind::Hook, "dense rows of a large matrix are not cut at the XCD eighths of the gathered vector (stream kernel)"},
    {"HPRLP_NO_SMALL", EnvKind::Hook, "do not use the single-workgroup kernel for Netlib-scale LPs"},
    {"HPRLP_NO_SMALL_POWER", EnvKind::Hook, "Netlib-scale LPs: power iteration by the regular kernels with the host test every 10th iteration (default: one launch of the single-workgroup kernel)"},
    {"HPRLP_NO_TILED_CR", EnvKind::Hook, "Curtis-Reid passes through the stream kernel also for matrices with a tiled copy (default: tiled kernel on a second value array of -log|a|)"},
    {"HPRLP_NT", EnvKind::Hook, "force default / nontemporal loads of the matrix in the stream kernel"},
    {"HPRLP_OVERLAP_COMM_FIRST", EnvKind::Hook, "tests only: with the in-process rank group, use RCCL's launch order (exchange enqueued before the local SpMV)"},
    {"HPRLP_PB_MIN_COLS", EnvKind::Hook, "gathered-vector length from which an unstructured matrix takes the all-remainder form (default 800 k columns, the crossover measured with k_pb_fused)"},
    {"HPRLP_PB_MIN_NNZ", EnvKind::Hook, "fewest entries for the all-remainder form"},
    {"HPRLP_PB_STAMPS", EnvKind::Hook, "developer build -DHPRLP_PB_PHASE_STAMPS=1: print k_pb_fused's phase times"},
    {"HPRLP_PIECES_ANYWAY", EnvKind::Hook, "attempt the tiled form although neighbouring rows gather from the same lines (<= 0.25 lines per entry) / keep the piece form although an XCD's rows gather from under 3 MB"},
    {"HPRLP_PRESOLVE_MAX_LINKS", EnvKind::Hook, "run one presolve stage alone / cut the chain after n links (tests, debugging)"},
    {"HPRLP_PRESOLVE_OFF", EnvKind::Hook, "switch presolve reductions / stages off (any subset; diagnostics)"},
    {"HPRLP_PRESOLVE_ONLY", EnvKind::Hook, "run one presolve stage alone / cut the chain after n links (tests, debugging)"},
    {"HPRLP_REORDER_CLUSTER", EnvKind::Hook, "knobs of the HOST reference path of the ordering (hprlp_locality_ordering): nodes per cluster, refinement sweeps, trimming percentage"},
    {"HPRLP_REORDER_HOST", EnvKind::Hook, "run the clustering of the locality ordering on the host (the reference form; default: on the device)"},
    {"HPRLP_REORDER_SWEEPS", EnvKind::Hook, "knobs of the HOST reference path of the ordering (hprlp_locality_ordering): nodes per cluster, refinement sweeps, trimming percentage"},
    {"HPRLP_REORDER_TRIM", EnvKind::Hook, "knobs of the HOST reference path of the ordering (hprlp_locality_ordering): nodes per cluster, refinement sweeps, trimming percentage"},
    {"HPRLP_SLACK_PIVOT", EnvKind::Hook, "pivot threshold of the costed slack-column substitution (default 0.5 of the row's largest entry)"},
    {"HPRLP_STORE_X", EnvKind::Hook, "every normal x-half reads and stores x (default for matrices with a tiled copy: inside a run of normal iterations x is rebuilt from x_hat and last_x by the next launch instead of travelling through memory)"},
    {"HPRLP_STREAM_NNZ", EnvKind::Hook, "row-block shape of the stream kernel (≤ 64 rows, ≤ 512 nonzeros per wave)"},
    {"HPRLP_STREAM_ROWS", EnvKind::Hook, "row-block shape of the stream kernel (≤ 64 rows, ≤ 512 nonzeros per wave)"},
    {"HPRLP_TILED_ANYWAY", EnvKind::Hook, "attempt the tiled form although neighbouring rows gather from the same lines (<= 0.25 lines per entry) / keep the piece form although an XCD's rows gather from under 3 MB"},
    {"HPRLP_TILED_MIN_COLS", EnvKind::Hook, "fewest columns for the tiled form (default 800 k, 2^19 for lowered super-blocks; 0 when HPRLP_TILED_MIN_ROWS is given)"},
    {"HPRLP_TILED_MIN_DENSE", EnvKind::Hook, "thresholds of the tiled copy (default 32·8192 rows, 0.5 of the entries in staged tiles)"},
    {"HPRLP_TILED_MIN_ROWS", EnvKind::Hook, "thresholds of the tiled copy (default 32·8192 rows, 0.5 of the entries in staged tiles)"},
    {"HPRLP_TILE_COLS", EnvKind::Hook, "force the tile width / super-block height of the tiled copies (default: Solver::choose_sb_rows)"},
    {"HPRLP_TILE_PIECES", EnvKind::Hook, "piece count of the tiled kernel's piece form (default: the chip's workgroup slots when the matrix has at most that many super-blocks; 0 = off)"},
    {"HPRLP_TILE_ROT", EnvKind::Hook, "alignment period of the rotated tile sweeps (default: mean window width of the matrix; 0 = ascending sweeps, i.e. CSR summation order)"},
    {"HPRLP_TILE_ROWS", EnvKind::Hook, "force the tile width / super-block height of the tiled copies (default: Solver::choose_sb_rows)"},
    {"HPRLP_TILE_STAMPS", EnvKind::Hook, "diagnostic instantiation of the piece kernel: shader-clock time per phase of a step on stderr when the solver is destroyed"},
    {"HPRLP_TILE_THREADS", EnvKind::Hook, "build the tiled copies with the host builder (background threads, default min(16, cores)) instead of on the device"},
    {"HPRLP_TILING_CHECK", EnvKind::Hook, "build them with both builders and fail on the first differing array element"},
    {"HPRLP_WG_TIMES", EnvKind::Hook, "diagnostic: per-workgroup wall-clock stamps of the fused tiled kernel on stderr"},
    {"HPRLP_WG_TIMES_DUMP", EnvKind::Hook, "per-workgroup stamps of the normal half-step kernels only, with the CU each ran on; raw table for tools/wgtimes_analyze.py"},
};

const EnvEntry *env_table(int *count) {
    if (count) *count = static_cast<int>(sizeof(kTable) / sizeof(kTable[0]));
    return kTable;
}

static bool hooks_on() {
    const char *h = std::getenv("HPRLP_TEST_HOOKS");
    return h && h[0] == '1';
}

const char *env_get(const char *name) {
    for (const EnvEntry &e : kTable) {
        if (std::strcmp(e.name, name) != 0) continue;
        if (e.kind == EnvKind::Hook && !hooks_on()) return nullptr;
        return std::getenv(name);
    }
    throw std::logic_error(std::string("environment switch missing from env.cpp's table: ") + name);
}

std::string env_in_effect(std::string *ignored_out) {
    std::string on, off;
    const bool hooks = hooks_on();
    for (const EnvEntry &e : kTable) {
        const char *v = std::getenv(e.name);
        if (!v || std::strcmp(e.name, "HPRLP_TEST_HOOKS") == 0) continue;
        std::string &dst = (e.kind == EnvKind::Hook && !hooks) ? off : on;
        if (!dst.empty()) dst += ' ';
        dst += std::string(e.name) + "=" + v;
    }
    if (ignored_out) *ignored_out = off;
    return on;
}

}  // namespace hprlp
