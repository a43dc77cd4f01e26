// mps_reader.cpp -- MPS / MPS.gz -> LP model (create_model_from_mps).
//
// Host-side text parsing; not part of the accelerated path, provided so that the boundary is
// complete.  Written from the behaviour of reference src/mps_reader.cpp:
//   * cards are always tokenised as FREE format (whitespace separated), as the reference does even
//     for fixed-format files (reference :1517);
//   * ROWS: the first N row is the objective, later N rows are ignored ("rim" objectives, :604-616);
//   * COLUMNS: one or two (row,value) pairs per card; 'MARKER' cards toggle an integer section whose
//     only effect is the default upper bound 1 for variables without bounds (:1139-1160);
//   * RHS: a value on the objective row sets the objective constant to MINUS that value (:765-767);
//     only the first RHS / RANGES / BOUNDS set name is honoured (:750-757);
//   * RANGES (:813-836): E rows: R>=0 -> [b, b+R], R<0 -> [b+R, b]; L rows: [b-|R|, b]; G rows: [b, b+|R|];
//   * BOUNDS (:860-927): FR MI PL BV LO UP FX LI UI; a variable with only an upper bound u<0 gets
//     lower bound -inf (:1150-1156); no bounds -> [0, inf);
//   * OBJSENSE is parsed and, like the reference (:577-585, no consumer), NOT applied;
//   * duplicate (row,col) entries are summed (:1325-1336).  The reference builds its row pointers
//     from the un-merged entry list (:1342-1351), which is wrong when duplicates exist; here they
//     are built from the merged list.
//   * .gz input is inflated with zlib (:24-58).
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <iomanip>
#include <iostream>
#include <limits>
#include <sstream>
#include <unordered_map>

#include "HPRLP.h"
#include "common.h"

namespace hprlp {
LP_info_cpu *model_from_csr(int m, int n, long nnz, const int *rp, const int *ci, const double *v, const double *AL,
                            const double *AU, const double *l, const double *u, const double *c, double obj_constant);
}

namespace {

using hprlp::model_from_csr;
const double INF = std::numeric_limits<double>::infinity();
const double NANV = std::numeric_limits<double>::quiet_NaN();

struct LineSource {
    FILE *fp = nullptr;
    gzFile gz = nullptr;
    bool open(const std::string &path) {
        const bool is_gz = path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
        if (is_gz) gz = gzopen(path.c_str(), "rb");
        else fp = std::fopen(path.c_str(), "r");
        return fp || gz;
    }
    bool getline(std::string &out) {
        char buf[4096];
        out.clear();
        while (true) {
            char *r = gz ? gzgets(gz, buf, sizeof(buf)) : std::fgets(buf, sizeof(buf), fp);
            if (!r) return !out.empty();
            out += buf;
            if (!out.empty() && out.back() == '\n') break;
        }
        while (!out.empty() && (out.back() == '\n' || out.back() == '\r')) out.pop_back();
        return true;
    }
    ~LineSource() {
        if (fp) std::fclose(fp);
        if (gz) gzclose(gz);
    }
};

enum class Sec { None, Name, ObjSense, Rows, Columns, Rhs, Bounds, Ranges, Other, Endata };

Sec section_of(const std::string &h) {
    if (h == "NAME") return Sec::Name;
    if (h == "OBJSENSE") return Sec::ObjSense;
    if (h == "ROWS") return Sec::Rows;
    if (h == "COLUMNS") return Sec::Columns;
    if (h == "RHS") return Sec::Rhs;
    if (h == "BOUNDS") return Sec::Bounds;
    if (h == "RANGES") return Sec::Ranges;
    if (h == "ENDATA") return Sec::Endata;
    return Sec::Other;  // QUADOBJ, QMATRIX, OBJECT BOUND, ...: skipped (LP only)
}

struct Parsed {
    std::unordered_map<std::string, int> row_index;  // 0 objective, -1 rim objective, k>0 constraint k-1
    std::unordered_map<std::string, int> col_index;
    std::vector<char> row_type;
    std::vector<double> lcon, ucon, c, lvar, uvar;
    std::vector<char> marked;
    std::vector<int> er, ec;
    std::vector<double> ev;
    double c0 = 0.0;
    bool have_obj = false, maximize = false;
    std::string rhs_name, rng_name, bnd_name;
};

void apply_pair_columns(Parsed &P, int col, const std::string &row, double val, int line) {
    auto it = P.row_index.find(row);
    if (it == P.row_index.end()) {
        std::cerr << "Error: Unknown row " << row << " at line " << line << "\n";
        return;
    }
    if (it->second == 0) P.c[col] = val;
    else if (it->second > 0) { P.er.push_back(it->second - 1); P.ec.push_back(col); P.ev.push_back(val); }
}

void apply_rhs(Parsed &P, const std::string &row, double val) {
    auto it = P.row_index.find(row);
    if (it == P.row_index.end()) { std::cerr << "Error: Unknown row " << row << "\n"; return; }
    if (it->second == 0) { P.c0 = -val; return; }
    if (it->second < 0) { std::cerr << "Error: Ignoring RHS for rim objective " << row << "\n"; return; }
    const int i = it->second - 1;
    if (P.row_type[i] == 'E') P.lcon[i] = P.ucon[i] = val;
    else if (P.row_type[i] == 'L') P.ucon[i] = val;
    else if (P.row_type[i] == 'G') P.lcon[i] = val;
}

void apply_range(Parsed &P, const std::string &row, double val, int line) {
    auto it = P.row_index.find(row);
    if (it == P.row_index.end()) { std::cerr << "Error: Unknown row " << row << " in RANGES section (l. " << line << ")\n"; return; }
    if (it->second <= 0) { std::cerr << "Error: Encountered objective row " << row << " in RANGES section (l. " << line << ")\n"; return; }
    const int i = it->second - 1;
    if (P.row_type[i] == 'E') { if (val >= 0.0) P.ucon[i] += val; else P.lcon[i] += val; }
    else if (P.row_type[i] == 'L') P.lcon[i] = P.ucon[i] - std::fabs(val);
    else if (P.row_type[i] == 'G') P.ucon[i] = P.lcon[i] + std::fabs(val);
}

bool parse(LineSource &src, Parsed &P) {
    Sec cur = Sec::None;
    bool seen[16] = {false};
    bool integer_section = false, endata = false;
    std::string line;
    int nline = 0;
    std::vector<std::string> f;
    while (src.getline(line)) {
        ++nline;
        if (line.empty() || line[0] == '*' || line[0] == '&') continue;
        f.clear();
        {
            std::istringstream is(line);
            std::string tok;
            while (f.size() < 6 && (is >> tok)) f.push_back(tok);
        }
        if (f.empty()) continue;
        if (!std::isspace(static_cast<unsigned char>(line[0]))) {  // section header
            const Sec s = section_of(f[0]);
            if (s == Sec::Endata) { endata = true; break; }
            if (s != Sec::Other) {
                if (seen[static_cast<int>(s)]) { std::cerr << "Error: More than one " << f[0] << " section\n"; return false; }
                seen[static_cast<int>(s)] = true;
                const bool rows_ok = seen[static_cast<int>(Sec::Rows)], cols_ok = seen[static_cast<int>(Sec::Columns)];
                if (s == Sec::Columns && !rows_ok) { std::cerr << "Error: ROWS section must come before COLUMNS\n"; return false; }
                if ((s == Sec::Rhs || s == Sec::Ranges) && !(rows_ok && cols_ok)) { std::cerr << "Error: " << f[0] << " section must come after ROWS and COLUMNS\n"; return false; }
                if (s == Sec::Bounds && !cols_ok) { std::cerr << "Error: BOUNDS section must come after COLUMNS\n"; return false; }
            }
            cur = (s == Sec::Name) ? Sec::None : s;
            continue;
        }
        const size_t nf = f.size();
        switch (cur) {
            case Sec::ObjSense:
                if (f[0] == "MAX") P.maximize = true;
                else if (f[0] != "MIN") std::cerr << "Warning: Unrecognized objective sense: " << f[0] << "\n";
                break;
            case Sec::Rows: {
                if (nf < 2) { std::cerr << "Error: Line " << nline << " contains only " << nf << " fields\n"; break; }
                const std::string &t = f[0], &name = f[1];
                const bool is_con = (t == "E" || t == "L" || t == "G");
                if (!is_con) {  // N (or anything else): objective
                    if (!P.have_obj) { P.have_obj = true; P.row_index[name] = 0; }
                    else { std::cerr << "Warning: Detected rim objective row " << name << " at line " << nline << "\n"; P.row_index[name] = -1; }
                    break;
                }
                P.row_index[name] = static_cast<int>(P.row_type.size()) + 1;
                P.row_type.push_back(t[0]);
                P.lcon.push_back(t == "L" ? -INF : 0.0);
                P.ucon.push_back(t == "G" ? INF : 0.0);
                break;
            }
            case Sec::Columns: {
                if (nf >= 3 && f[1] == "'MARKER'") {
                    if (f[2] == "'INTORG'") integer_section = true;
                    else if (f[2] == "'INTEND'") integer_section = false;
                    else std::cerr << "Error: Ignoring marker " << f[2] << " at line " << nline << "\n";
                    break;
                }
                if (nf < 3) { std::cerr << "Error: Line " << nline << " contains only " << nf << " fields\n"; break; }
                auto ins = P.col_index.emplace(f[0], static_cast<int>(P.c.size()));
                if (ins.second) { P.c.push_back(0.0); P.lvar.push_back(NANV); P.uvar.push_back(NANV); P.marked.push_back(integer_section); }
                const int col = ins.first->second;
                apply_pair_columns(P, col, f[1], std::atof(f[2].c_str()), nline);
                if (nf >= 5) apply_pair_columns(P, col, f[3], std::atof(f[4].c_str()), nline);
                break;
            }
            case Sec::Rhs: {
                if (nf < 3) { std::cerr << "Error: Line " << nline << " contains only " << nf << " fields\n"; break; }
                if (P.rhs_name.empty()) P.rhs_name = f[0];
                else if (P.rhs_name != f[0]) { std::cerr << "Error: Skipping line " << nline << " with rim RHS " << f[0] << "\n"; break; }
                apply_rhs(P, f[1], std::atof(f[2].c_str()));
                if (nf >= 5) apply_rhs(P, f[3], std::atof(f[4].c_str()));
                break;
            }
            case Sec::Ranges: {
                if (nf < 3) { std::cerr << "Error: Line " << nline << " contains only " << nf << " fields\n"; break; }
                if (P.rng_name.empty()) P.rng_name = f[0];
                else if (P.rng_name != f[0]) { std::cerr << "Error: Skipping line " << nline << " with rim RANGES " << f[0] << "\n"; break; }
                apply_range(P, f[1], std::atof(f[2].c_str()), nline);
                if (nf >= 5) apply_range(P, f[3], std::atof(f[4].c_str()), nline);
                break;
            }
            case Sec::Bounds: {
                if (nf < 3) { std::cerr << "Error: Line " << nline << " contains only " << nf << " fields\n"; break; }
                if (P.bnd_name.empty()) P.bnd_name = f[1];
                else if (P.bnd_name != f[1]) { std::cerr << "Error: Skipping line " << nline << " with rim bound " << f[1] << "\n"; break; }
                auto it = P.col_index.find(f[2]);
                if (it == P.col_index.end()) { std::cerr << "Error: Unknown column " << f[2] << "\n"; break; }
                const int col = it->second;
                const std::string &b = f[0];
                if (b == "FR") { P.lvar[col] = -INF; P.uvar[col] = INF; break; }
                if (b == "MI") { P.lvar[col] = -INF; break; }
                if (b == "PL") { P.uvar[col] = INF; break; }
                if (b == "BV") { P.lvar[col] = 0.0; P.uvar[col] = 1.0; break; }
                if (nf < 4) { std::cerr << "Error: At least 4 fields required for " << b << " bounds\n"; break; }
                const double val = std::atof(f[3].c_str());
                if (b == "LO" || b == "LI") P.lvar[col] = val;
                else if (b == "UP" || b == "UI") P.uvar[col] = val;
                else if (b == "FX") P.lvar[col] = P.uvar[col] = val;
                else std::cerr << "Warning: Unknown bound type " << b << "\n";
                break;
            }
            default:
                break;
        }
    }
    if (!endata) std::cerr << "Warning: Reached end of file before ENDATA section\n";
    for (size_t j = 0; j < P.c.size(); ++j) {  // bound defaults
        const bool nl = std::isnan(P.lvar[j]), nu = std::isnan(P.uvar[j]);
        if (nl && nu) { P.lvar[j] = 0.0; P.uvar[j] = P.marked[j] ? 1.0 : INF; }
        else if (nl) P.lvar[j] = (P.uvar[j] < 0) ? -INF : 0.0;
        else if (nu) P.uvar[j] = INF;
    }
    return true;
}

}  // namespace

extern "C" LP_info_cpu *create_model_from_mps(const char *mps_file_path) {
    if (!mps_file_path) {
        std::cerr << "[error] Null MPS file path pointer" << std::endl;
        return nullptr;
    }
    try {
        std::cout << "Start reading file....\n";
        const auto t0 = hprlp::time_now();
        LineSource src;
        if (!src.open(mps_file_path)) {
            std::cerr << "Error: Cannot open file " << mps_file_path << "\n[error] Invalid model from MPS file" << std::endl;
            return nullptr;
        }
        Parsed P;
        if (!parse(src, P)) {
            std::cerr << "Error: Failed to read MPS file\n[error] Invalid model from MPS file" << std::endl;
            return nullptr;
        }
        std::cout << "File reading time: " << std::fixed << std::setprecision(4) << hprlp::time_since(t0) << " seconds\n"
                  << std::defaultfloat;
        if (P.maximize)
            std::cerr << "Warning: OBJSENSE MAX is parsed but not applied (same as the reference): the model is minimised\n";
        const int m = static_cast<int>(P.row_type.size()), n = static_cast<int>(P.c.size());
        // COO -> CSR, sorted by (row, col), duplicates summed
        std::vector<size_t> order(P.er.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
            return P.er[a] != P.er[b] ? P.er[a] < P.er[b] : P.ec[a] < P.ec[b];
        });
        std::vector<int> rp(static_cast<size_t>(m) + 1, 0), ci;
        std::vector<double> v;
        int prev_r = -1, prev_c = -1;
        for (size_t k : order) {
            if (P.er[k] == prev_r && P.ec[k] == prev_c) { v.back() += P.ev[k]; continue; }
            ci.push_back(P.ec[k]); v.push_back(P.ev[k]);
            rp[P.er[k] + 1]++;
            prev_r = P.er[k]; prev_c = P.ec[k];
        }
        for (int i = 0; i < m; ++i) rp[i + 1] += rp[i];
        const long nnz = static_cast<long>(v.size());
        if (m <= 0 || n <= 0 || nnz <= 0) {
            std::cerr << "Error: Invalid dimensions in MPS model: m=" << m << ", n=" << n << ", nnz=" << nnz
                      << "\n[error] Invalid model from MPS file" << std::endl;
            return nullptr;
        }
        std::cout << "problem information: nRow = " << m << ", nCol = " << n << ", nnz A = " << nnz << std::endl << std::endl;
        LP_info_cpu *mo = model_from_csr(m, n, nnz, rp.data(), ci.data(), v.data(), P.lcon.data(), P.ucon.data(), P.lvar.data(),
                                         P.uvar.data(), P.c.data(), P.c0);
        hprlp::warm_for_first_solve();  // (abi.cpp: the process-wide part of the first solve, on the calling thread, once)
        return mo;
    } catch (const std::exception &e) {
        std::cerr << "[error] Failed to read MPS file: " << e.what() << std::endl;
        return nullptr;
    }
}
