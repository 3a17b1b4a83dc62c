// ABI bookkeeping for libpwc_hip.so: version and the thread-local error string.
// Error convention mirrors the reference's "launcher returns a status, binding raises"
// (correlation_cuda_kernel.cu:417-426 -> correlation_cuda.cc:81-83), minus the printf.
#include <stdlib.h>
#include <string.h>

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

// ---- run-time options ---------------------------------------------------------------------------------------
// One table for the switches that used to be function-local `static const int knob = getenv(...)` (read
// once at first use, so a test could only flip one if it ran before that use).  The environment variable still gives the
// DEFAULT -- read the first time the option is looked at -- and pwc_set_option() overrides it at any time, for every thread.
struct OptDesc { const char *name, *env; int dflt; };
static constexpr OptDesc kOpts[OPT_COUNT] = {
    {"conv_wino4", "PWC_CONV_WINO4", 1},                  // F(4x4) route allowed (pwc_conv3x3_wino4_preferred)
    {"w4_tailsplit", "PWC_W4_TAILSPLIT", 1},              // partial last round of an F(4x4) launch cut along Cin
    {"w4_smallsplit", "PWC_W4_SMALLSPLIT", 1},            // launches that do not fill the chip cut along Cin (small batches)
    {"w4_small_min_wgs", "PWC_W4_SMALL_MIN_WGS", 160},     // workgroups (tiles x cout groups x Cin slices) a small F(4x4) launch must reach to be preferred
    {"corr_pipe", "PWC_CORR_PIPE", 0},                    // PLAIN correlation on the round-4 pipelined / rolling kernels (pwc_corr_pipe.hip): parity with
                                                          // the round-2 kernel at level 2, slower at level 3 -- opt-in (profiles/r04_corr_notes.md)
    {"corr_pipe_min_tiles", "PWC_CORR_PIPE_MIN_TILES", 1024},
    {"corr_roll", "PWC_CORR_ROLL", 1},                    // C <= 32: the rolling form (a workgroup walks down a column of tiles and keeps its in2 rows)
    {"corr_small_tiles", "PWC_CORR_SMALL_TILES", 48},     // launches of at most this many 8x32 tiles: the small-map correlation (and warp, then correlation, instead of the fused kernel)
    {"head10", "PWC_HEAD10", 1},                          // small levels: flow head + upfeat as one 10-channel convolution + pwc_upsample_entry_f32 (read by the Python engine)
    {"f16_level_corr", "PWC_F16_LEVEL_CORR", 0},          // half-precision plans: level entry + warp + correlation as one kernel (read by the Python engine);
                                                          // bit-identical, but 97 vs 81 us at level 2 (2.5x halo gathers): opt-in
    {"warpcorr_window", "PWC_WARPCORR_WINDOW", 2},        // fused warp+correlation on the LDS-window kernel: 2 = C <= 32 and C <= 64 (levels 2, 3), 1 = C <= 32 only, 0 = round-2 kernel
    {"stream_slice_wgs", "PWC_STREAM_SLICE_WGS", 512},    // streaming flow head (+ upfeat): launches whose 4-row tiles are under 3/4 of this many workgroups are cut along Cin
                                                          // into slices (fixed-order reduction, needs the caller's workspace); 0 = off
    {"c1_in_arena", "PWC_C1_IN_ARENA", 1},                // fp32 plans: the pyramid's last convolution of levels 2-5 writes the first image's features straight into
                                                          // their slot of the decoder arena (batch-strided output) instead of a copy per level (read by the Python engine)
    {"head_sliced_min_tiles", "PWC_HEAD_SLICED_MIN_TILES", 14},   // fp32 plans: levels of at least this many 8-row x 128-column tiles run predict_flowL + upfeatL through
                                                          // pwc_head_upfeat_ws_fwd (Cin slices below 64 tiles) instead of the 10-channel convolution; 64 = only where the one-pass
                                                          // kernel runs (read by the Python engine)
};
// the table is indexed by enum Opt (pwc_common.h): a row out of order would silently give one switch another's value
constexpr bool opt_is(Opt o, const char *name) {
    const char *a = kOpts[o].name;
    while (*a && *a == *name) { ++a; ++name; }
    return *a == *name;
}
static_assert(opt_is(OPT_CONV_WINO4, "conv_wino4") && opt_is(OPT_W4_TAILSPLIT, "w4_tailsplit") && opt_is(OPT_W4_SMALLSPLIT, "w4_smallsplit") &&
              opt_is(OPT_W4_SMALL_MIN_WGS, "w4_small_min_wgs") && opt_is(OPT_CORR_PIPE, "corr_pipe") &&
              opt_is(OPT_CORR_PIPE_MIN_TILES, "corr_pipe_min_tiles") && opt_is(OPT_CORR_ROLL, "corr_roll") &&
              opt_is(OPT_CORR_SMALL_TILES, "corr_small_tiles") && opt_is(OPT_HEAD10, "head10") && opt_is(OPT_F16_LEVEL_CORR, "f16_level_corr") &&
              opt_is(OPT_WARPCORR_WINDOW, "warpcorr_window") && opt_is(OPT_STREAM_SLICE_WGS, "stream_slice_wgs") &&
              opt_is(OPT_C1_IN_ARENA, "c1_in_arena") && opt_is(OPT_HEAD_SLICED_MIN_TILES, "head_sliced_min_tiles"), "kOpts rows follow enum Opt");
static std::atomic<int> g_opt_val[OPT_COUNT];
static std::atomic<unsigned char> g_opt_set[OPT_COUNT];

int option(Opt o) {
    if (!g_opt_set[o].load(std::memory_order_acquire)) {
        const char *e = getenv(kOpts[o].env);
        g_opt_val[o].store((e && *e) ? atoi(e) : kOpts[o].dflt, std::memory_order_relaxed);
        g_opt_set[o].store(1, std::memory_order_release);
    }
    return g_opt_val[o].load(std::memory_order_relaxed);
}

}  // namespace pwc

extern "C" int pwc_set_option(const char *name, int value) {
    if (!name) PWC_FAIL(PWC_EINVAL, "pwc_set_option: null name");
    for (int o = 0; o < pwc::OPT_COUNT; ++o)
        if (!strcmp(name, pwc::kOpts[o].name)) {
            pwc::g_opt_val[o].store(value, std::memory_order_relaxed);
            pwc::g_opt_set[o].store(1, std::memory_order_release);
            return PWC_OK;
        }
    PWC_FAIL(PWC_EINVAL, "pwc_set_option: unknown option '%s'", name);
}

extern "C" int pwc_get_option(const char *name, int *value) {
    if (!name || !value) PWC_FAIL(PWC_EINVAL, "pwc_get_option: null argument");
    for (int o = 0; o < pwc::OPT_COUNT; ++o)
        if (!strcmp(name, pwc::kOpts[o].name)) { *value = pwc::option((pwc::Opt)o); return PWC_OK; }
    PWC_FAIL(PWC_EINVAL, "pwc_get_option: unknown option '%s'", name);
}

extern "C" const char *pwc_last_conv_kernel(void) {
    snprintf(pwc::g_kbuf, sizeof(pwc::g_kbuf), "%s<%d, %d, %d, %d, %d, %d>", pwc::g_kname, pwc::g_kargs[0], pwc::g_kargs[1],
             pwc::g_kargs[2], pwc::g_kargs[3], pwc::g_kargs[4], pwc::g_kargs[5]);
    return pwc::g_kbuf;
}

extern "C" int pwc_abi_version(void) { return PWC_ABI_VERSION; }

// Non-zero when a kernel source was compiled with a timing-experiment switch (-DPWC_*_EXP=mask: parts of the work are skipped, the
// results are INVALID).  Such builds are made next to the product by tools/variant_build.sh and selected with PWC_HIP_LIB; the Python
// binding refuses one as the default library.
namespace pwc { int exp_mask_wino4(); int exp_mask_corr(); int exp_mask_corr_pipe(); int exp_mask_stream3x3(); }
extern "C" int pwc_experiment_mask(void) {
    return (pwc::exp_mask_wino4() ? 1 : 0) | (pwc::exp_mask_corr() ? 2 : 0) | (pwc::exp_mask_corr_pipe() ? 4 : 0) | (pwc::exp_mask_stream3x3() ? 8 : 0);
}

extern "C" const char *pwc_last_error(void) { return pwc::g_err; }
