// Entry points of include/spyhip.h whose kernels are not written yet: they fail
// loudly (there is no CPU fallback).  Replaced one by one as the kernels land.
#include "spy_common.h"

extern "C" int spyhip_granger(spyhip_ctx*, const void*, int, int, double, int, double, double, void*, void*, void*,
                              double*) {
    spy::set_error("spyhip_granger: Wilson/Granger kernels not implemented in this build");
    return -5;
}
