// Internal helpers shared by the .hip translation units.
#ifndef PMC_INTERNAL_H
#define PMC_INTERNAL_H

#include <hip/hip_runtime.h>
#include "../../include/pocomc_amd.h"

int pmc_fail(const char* msg);
int pmc_fail_hip(hipError_t e, const char* what);
int pmc_check_launch(const char* what);

#endif
