// hy_options.hpp -- the library's process-wide options (include/hyrise_amd.h HY_OPT_*): which of several equivalent paths / launch
// shapes an operator takes.  An array of atomics filled with the defaults; hy_set_option stores, the operators load.  No getenv in a
// release build: the debug aids (traces, host timelines, kernels with parts switched off -- "results are wrong then") exist only
// under -DHY_DEBUG_SWITCHES (tools/build_variant.sh debug -DHY_DEBUG_SWITCHES), where HY_DEBUG_ENV reads the environment.
#pragma once

#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "../../include/hyrise_amd.h"

namespace hy {

extern std::atomic<int64_t> g_options[HY_OPT_COUNT];   // runtime.hip
inline int64_t option(uint32_t id) { return g_options[id].load(std::memory_order_relaxed); }

}  // namespace hy

#ifdef HY_DEBUG_SWITCHES
#define HY_DEBUG_ENV(name) getenv(name)
#else
#define HY_DEBUG_ENV(name) (static_cast<const char*>(nullptr))
#endif
