// options.cpp -- the library's behaviour switches (include/rvsgpu.h:
// rvs_option_set / rvs_option_get).
//
// The entry points are driven from several host threads at once (vel_fit.process
// runs its halves on two threads with the GIL released), so a switch is never read
// from the environment at launch time -- getenv beside a setenv of the interpreter
// is a data race in glibc, and two halves of one batch could take different kernels
// in the middle of a run.  The table is filled ONCE, from the environment variables
// of the same (upper-case, RVS_-prefixed) names, the first time any entry point
// looks; after that only rvs_option_set changes it (relaxed atomics: a switch takes
// effect at the next call that reads it).
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "options.h"

namespace {
struct OptDef {
  const char *name;   // rvs_option_set's name; the variable is RVS_<NAME>
  const char *env;
  int dflt;
  bool env_presence;  // the variable's presence alone means 1
};
const OptDef kDefs[RVS_OPT_COUNT] = {
    {"xc_ws", "RVS_XC_WS", 1, false},
    {"xc_ws1", "RVS_XC_WS1", 0, true},
    {"nm_glue", "RVS_NM_GLUE", 1, false},
    {"nm_bucket", "RVS_NM_BUCKET", 0, true},
    {"obj_inblk_max", "RVS_OBJ_INBLK_MAX", 768, false},
    {"obj_sort", "RVS_OBJ_SORT", 1, false},
    {"nn_pipe", "RVS_NN_PIPE", 1, false},
    {"nm_split_min", "RVS_NM_SPLIT_MIN", 1024, false},
    {"nm_spec_max", "RVS_NM_SPEC_MAX", 21, false},
    {"nm_tail_window", "RVS_NM_TAIL_WINDOW", 16, false},
};
std::atomic<int> g_val[RVS_OPT_COUNT];
std::once_flag g_once;

void init_table() {
  for (int i = 0; i < RVS_OPT_COUNT; i++) {
    int v = kDefs[i].dflt;
    if (const char *ev = std::getenv(kDefs[i].env))
      v = kDefs[i].env_presence ? 1 : std::atoi(ev);
    g_val[i].store(v, std::memory_order_relaxed);
  }
}
int find(const char *name) {
  if (!name) return -1;
  for (int i = 0; i < RVS_OPT_COUNT; i++)
    if (!std::strcmp(name, kDefs[i].name)) return i;
  return -1;
}
}  // namespace

int rvs_opt(int id) {
  std::call_once(g_once, init_table);
  return g_val[id].load(std::memory_order_relaxed);
}

extern "C" int rvs_option_set(const char *name, int value) {
  const int i = find(name);
  if (i < 0) return RVS_E_ARG;
  std::call_once(g_once, init_table);
  g_val[i].store(value, std::memory_order_relaxed);
  return 0;
}

extern "C" int rvs_option_get(const char *name, int *value) {
  const int i = find(name);
  if (i < 0 || !value) return RVS_E_ARG;
  *value = rvs_opt(i);
  return 0;
}
