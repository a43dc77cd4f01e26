#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include <vector>
#include "HPRLP.h"
#include "hprlp_amd.h"
#include "common.h"
#include "tiled.h"
#include "presolve.h"
#include "reorder.h"
using namespace hprlp;
// device-side members of DeviceTiled (tiled_build.hip) that tiled.cpp's upload path references: never called here
namespace hprlp {
void DeviceTiled::pack_indices(hipStream_t) {}
void DeviceTiled::finish_schedule(hipStream_t) {}
}  // namespace hprlp

int main() {
    std::mt19937_64 rng(7);
    for (int rep = 0; rep < 6; ++rep) {
        const int m = 500 + rep * 3001, n = 700 + rep * 2503;
        std::vector<int> rp(m + 1, 0), ci; std::vector<double> v;
        for (int i = 0; i < m; ++i) {
            int len = rng() % 9; if (i % 97 == 0) len = 0; if (i % 501 == 1) len = 300;
            std::vector<int> cols;
            for (int k = 0; k < len; ++k) cols.push_back(rng() % n);
            if (i >= m - 40) cols = {(i * 7) % n, (i * 7 + 7) % n};  // chained doubleton rows (made equalities below): the matrix-changing presolve stage
            std::sort(cols.begin(), cols.end()); cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
            if (i % 50 == 4 && rp[i] - rp[i - 1] >= 2) {  // parallel to the previous row: -2 x its entries
                for (int k = rp[i - 1]; k < rp[i]; ++k) { ci.push_back(ci[k]); v.push_back(-2.0 * v[k]); }
                rp[i + 1] = (int)ci.size();
                continue;
            }
            for (int c : cols) { ci.push_back(c); v.push_back((double)(rng() % 1000) / 100.0 - 5.0); }
            rp[i + 1] = (int)ci.size();
        }
        const long nnz = rp[m];
        std::vector<int> trp, tci; std::vector<double> tv;
        csr_transpose_host(m, n, nnz, rp.data(), ci.data(), v.data(), trp, tci, tv);
        std::vector<int> trp2, tci2; std::vector<double> tv2;
        csr_transpose_range_host(m, n / 3, 2 * n / 3, rp.data(), ci.data(), v.data(), trp2, tci2, tv2);
        TiledHost th;
        bool ok = build_tiled(m, n, rp.data(), ci.data(), &th, 1, 0.0);
        std::vector<double> AL(m, -1.0), AU(m, 1.0), l(n, 0.0), u(n, 2.0), c(n, 1.0);
        for (int j = 0; j < n; j += 13) u[j] = l[j];          // fixed columns
        for (int i = 0; i < m; i += 7) AU[i] = INFINITY;
        for (int i = 2; i < m; i += 5) AL[i] = AU[i] = 0.0;   // equality rows: slack-column substitution
        for (int j = 0; j < n; j += 3) c[j] = -0.5;           // both cost signs: dual fixing either way
        for (int i = m - 40; i < m; ++i) AL[i] = AU[i] = 0.0;   // the doubleton rows (x = 0 is feasible)
        for (int j = 5; j < n; j += 11) u[j] = INFINITY;        // columns whose bounds the rows imply (bound propagation)
        LP_info_cpu *model = create_model_from_arrays(m, n, (int)nnz, rp.data(), ci.data(), v.data(), AL.data(), AU.data(), l.data(), u.data(), c.data(), false);
        if (!model) { printf("model null\n"); return 1; }
        {
            Presolve pre;
            bool red = pre.run(model);
            if (red) {
                const LP_info_cpu *r = pre.reduced();
                std::vector<double> xr(r->n, 0.5), yr(r->m, 0.1), zr(r->n, 0.0), x(n), y(m), z(n);
                pre.postsolve(xr.data(), yr.data(), zr.data(), x.data(), y.data(), z.data());
                OriginalKkt k = original_kkt(model, x.data(), y.data(), z.data());
                printf("rep %d: tiled %d presolve (%d,%d)->(%d,%d) kkt %.3g, doubleton rows %d, tightened bounds %d\n", rep, (int)ok, m, n, r->m, r->n, k.primal_feas, pre.stats().doubleton_rows, pre.stats().tightened_bounds);
            } else printf("rep %d: tiled %d presolve declined\n", rep, (int)ok);
        }
        {   // the host form of the locality ordering on this pattern (declines or orders: either way every loop runs)
            std::vector<int> pr, pc;
            ReorderStats st;
            const bool ord = locality_ordering(m, n, rp.data(), ci.data(), &pr, &pc, &st, 2.0);  // (2.0: no order is good enough, so nothing is skipped)
            printf("rep %d: locality ordering %s (tiled share %.3f -> %.3f, %d clusters)\n", rep, ord ? "accepted" : "declined", st.fraction_before,
                   st.fraction_after, st.clusters);
        }
        hprlp_shard sh;
        if (hprlp_extract_shard(model, 1, 3, &sh) == 0) hprlp_free_shard(&sh);
        free_model(model);
    }
    LP_info_cpu *mm = create_model_from_mps("tests/data/lp_features.mps");
    if (mm) free_model(mm);
    printf("done\n");
    return 0;
}
