// Registry of measured launch plans (see tune.hpp).  Written before the first launch by the host binding and
// read-only afterwards; a mutex keeps concurrent registration / lookup well defined.
#include "common.hpp"
#include "tune.hpp"
#include <mutex>
#include <vector>

namespace {
struct Entry { int kind; int64_t a, b, c, d; TunePlan plan; };
std::vector<Entry>& table() { static std::vector<Entry> t; return t; }
std::mutex& mu() { static std::mutex m; return m; }
}  // namespace

bool creid_tune_lookup(int kind, int64_t a, int64_t b, int64_t c, int64_t d, TunePlan& out) {
  std::lock_guard<std::mutex> g(mu());
  for (const Entry& e : table())
    if (e.kind == kind && e.a == a && e.b == b && e.c == c && e.d == d) { out = e.plan; return true; }
  return false;
}

extern "C" {

int creid_tune_set(int32_t kind, int64_t a, int64_t b, int64_t c, int64_t d, int32_t p0, int32_t p1, int32_t p2) {
  if (kind != CREID_TUNE_WGRAD && kind != CREID_TUNE_IGEMM) return CREID_E_ARG;
  std::lock_guard<std::mutex> g(mu());
  for (Entry& e : table())
    if (e.kind == kind && e.a == a && e.b == b && e.c == c && e.d == d) { e.plan = TunePlan{p0, p1, p2}; return 0; }
  table().push_back(Entry{kind, a, b, c, d, TunePlan{p0, p1, p2}});
  return 0;
}

int creid_tune_clear(void) {
  std::lock_guard<std::mutex> g(mu());
  table().clear();
  return 0;
}

}  // extern "C"
