// env.h -- the library's environment switches, all of them, in one table (env.cpp).
//
// The reference has no environment switches (its configuration is HPRLP_parameters, include/structs.h:25-40).  This library
// reads two kinds: a handful meant for an INTEGRATOR (diagnostics, safety fall-backs, the multi-process transport) that are
// always honoured, and TEST HOOKS (forcing a kernel form, thresholds, A/B knobs of the measurements in profiles/) that are
// ignored unless HPRLP_TEST_HOOKS=1 is set as well -- a production process that inherited a stray HPRLP_TILE_ROWS from
// somebody's shell runs the default path.  Every read goes through env_get(): a name missing from the table is a programming
// error (std::logic_error, caught by tests/test_abi.py's scan of the sources).  Switches are read when a solver (or model) is
// built, not cached at load: the parity tests build many forms in one process.
#pragma once

#include <string>

namespace hprlp {

enum class EnvKind { Integrator, Hook };
struct EnvEntry {
    const char *name;
    EnvKind kind;
    const char *what;
};
const EnvEntry *env_table(int *count);
// the variable's value if it is set and honoured (hooks: only under HPRLP_TEST_HOOKS=1), else nullptr
const char *env_get(const char *name);
inline bool env_on(const char *name) {
    const char *e = env_get(name);
    return e && e[0] == '1';
}
// "HPRLP_X=1 HPRLP_Y=64" -- the switches set AND honoured right now ("" if none); ignored_out: hooks that are set but ignored
std::string env_in_effect(std::string *ignored_out = nullptr);

}  // namespace hprlp
