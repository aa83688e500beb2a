// cm_api.hip - error plumbing + version of the C ABI (include/commarl.h)
#include "cm_internal.h"

namespace cm {

static thread_local std::string g_last_error;

int set_error(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return CM_ERR_HIP;
}

}  // namespace cm

extern "C" int cm_abi_version(void) { return CM_ABI_VERSION; }
extern "C" const char *cm_last_error(void) { return cm::g_last_error.c_str(); }
