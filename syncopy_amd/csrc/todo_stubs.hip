// Entry points of include/spyhip.h whose kernels are not written yet: they fail
// loudly (there is no CPU fallback).  Replaced one by one as the kernels land.
#include "spy_common.h"

extern "C" int spyhip_cwt_plan_create(spyhip_ctx*, int, int, int, const double*, double, double, int, int, int, int,
                                      int, spyhip_cwt_plan**) {
    spy::set_error("spyhip_cwt_plan_create: Morlet CWT kernel not implemented in this build");
    return -5;
}
extern "C" int spyhip_cwt_plan_destroy(spyhip_cwt_plan*) { return 0; }
extern "C" int spyhip_cwt_exec(spyhip_cwt_plan*, const float*, int64_t, const int32_t*, const int64_t*, int, void*,
                               int) {
    spy::set_error("spyhip_cwt_exec: Morlet CWT kernel not implemented in this build");
    return -5;
}
extern "C" int spyhip_granger(spyhip_ctx*, const void*, int, int, double, int, double, double, void*, void*, void*,
                              double*) {
    spy::set_error("spyhip_granger: Wilson/Granger kernels not implemented in this build");
    return -5;
}
