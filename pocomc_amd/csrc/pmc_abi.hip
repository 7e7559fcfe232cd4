// Error reporting and version of the C ABI (include/pocomc_amd.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "pmc_internal.h"

static thread_local char g_err[512] = "";

int pmc_fail(const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return 1;
}

int pmc_fail_hip(hipError_t e, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return 2;
}

int pmc_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pmc_fail_hip(e, what);
    return 0;
}

extern "C" const char* pmc_last_error(void) { return g_err; }
extern "C" int pmc_abi_version(void) { return PMC_ABI_VERSION; }

// the hash of the sources this library was built from (csrc/Makefile: BUILD_ID); pocomc_amd/_lib.py compares it with the
// hash of the sources next to it and refuses a stale library
#ifndef PMC_BUILD_ID
#define PMC_BUILD_ID "unknown"
#endif
extern "C" const char* pmc_build_id(void) { return PMC_BUILD_ID; }
