// C-ABI plumbing shared by all entry points: error string, version/arch query.
#include "common.hpp"
#include "../../include/gsplat_amd.h"

#include <cstdarg>
#include <cstdio>

namespace gsx {

static thread_local char g_last_error[512] = {0};

void set_last_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

int check_launch(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return GSX_ERR_LAUNCH;
    }
    return GSX_OK;
}

} // namespace gsx

extern "C" const char *gsx_last_error(void) { return gsx::g_last_error; }
extern "C" int gsx_version(void) { return GSX_ABI_VERSION; }
extern "C" const char *gsx_arch(void) { return "gfx950"; }
