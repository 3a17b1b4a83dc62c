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

// last MFMA convolution variant launched by this thread (what the tile cost model picked): name + template arguments
static thread_local const char *g_kname = "";
static thread_local int g_kargs[6] = {0, 0, 0, 0, 0, 0};
static thread_local char g_kbuf[128] = "";

void note_kernel(const char *name, int a, int b, int c, int d, int e, int f) {
    g_kname = name;
    g_kargs[0] = a; g_kargs[1] = b; g_kargs[2] = c; g_kargs[3] = d; g_kargs[4] = e; g_kargs[5] = f;
}

}  // namespace pwc

extern "C" const char *pwc_last_conv_kernel(void) {
    snprintf(pwc::g_kbuf, sizeof(pwc::g_kbuf), "%s<%d, %d, %d, %d, %d, %d>", pwc::g_kname, pwc::g_kargs[0], pwc::g_kargs[1],
             pwc::g_kargs[2], pwc::g_kargs[3], pwc::g_kargs[4], pwc::g_kargs[5]);
    return pwc::g_kbuf;
}

extern "C" int pwc_abi_version(void) { return PWC_ABI_VERSION; }

extern "C" const char *pwc_last_error(void) { return pwc::g_err; }
