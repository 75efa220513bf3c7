// Engine options: every switch that changes what an engine does, in ONE documented table (options.cpp), settable through the
// C-ABI (tkspmv_set_option, include/tkspmv.h) and, for shell-driven A/B runs, through the environment (TKSPMV_<NAME>; a value set
// through the API wins). Options are read when an engine (or a packed matrix, or a communicator) is CREATED; changing one
// afterwards affects later creations only. No other library source reads the environment for an engine option (tests/test_capi.py checks).
#pragma once
#include <cstdlib>

namespace tkspmv {

struct OptionDef {
    const char *name;   // without the TKSPMV_ prefix
    const char *kind;   // "behaviour" | "layout" | "tuning" | "diagnostic"
    const char *values; // accepted values and the default
    const char *doc;
};

// The current value of a DOCUMENTED option (nullptr: unset -> the engine's own default). An undocumented name is a programming
// error: it aborts, so that no switch can exist outside the table.
const char *opt(const char *name);
inline bool opt_set(const char *name) { return opt(name) != nullptr; }
inline long opt_int(const char *name, long dflt) {
    const char *v = opt(name);
    return v ? atol(v) : dflt;
}

int option_count();
const OptionDef *option_def(int i);
// 0 = stored; -1 = no such option. value == nullptr removes the API-side setting (the environment, if set, shows through again).
int set_option(const char *name, const char *value);

}  // namespace tkspmv
