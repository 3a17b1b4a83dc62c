// ABI bookkeeping for libpwc_hip.so: version and the thread-local error string.
// Error convention mirrors the reference's "launcher returns a status, binding raises"
// (correlation_cuda_kernel.cu:417-426 -> correlation_cuda.cc:81-83), minus the printf.
#include "pwc_common.h"

namespace pwc {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace pwc

extern "C" int pwc_abi_version(void) { return PWC_ABI_VERSION; }

extern "C" const char *pwc_last_error(void) { return pwc::g_err; }
