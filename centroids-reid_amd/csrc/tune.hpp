// Per-shape launch plans measured on the target (tools/tune_plans.py -> centroids-reid_amd/tuned_plans.json,
// registered through creid_tune_set when the Python binding loads the library).  The kernels' own heuristics
// (tile / split / ring-depth rules in conv_igemm.hip and conv_wgrad.hip) stay the fallback for every shape that has
// no entry; an entry only selects among variants those files already implement.
#pragma once
#include <stdint.h>

enum { CREID_TUNE_WGRAD = 0, CREID_TUNE_IGEMM = 1 };

struct TunePlan { int p0, p1, p2; };

// returns true and fills `out` when (kind, a, b, c, d) has a registered plan
bool creid_tune_lookup(int kind, int64_t a, int64_t b, int64_t c, int64_t d, TunePlan& out);
