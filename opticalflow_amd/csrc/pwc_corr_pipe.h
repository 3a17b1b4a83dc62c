// Internal launchers of pwc_corr_pipe.hip (round-4 correlation kernels: one workgroup per CU, deep LDS-DMA ring, output drained
// through LDS by its own wave).  Called by the C-ABI entry points in pwc_corr.hip, which keep the older kernels for the shapes
// these refuse.
#pragma once
#include "pwc_common.h"

namespace pwc {

// true when the plain / fused pipelined kernel takes this geometry (fp32, W % 4 == 0 and 16-byte alignment are the caller's check)
bool corr81_pipe_fits(int B, int C, int H, int W);
// option "corr_pipe" = 0 (pwc_set_option / PWC_CORR_PIPE) keeps the round-2 kernels for every shape (A/B runs)
bool corr81_pipe_enabled();

int launch_corr81_pipe(const float *in1, const float *in2, float *out, int B, int C, int H, int W,
                       int64_t bs1, int64_t bs2, int64_t bso, float scale, float slope, int do_leaky, hipStream_t st);

// fused warp + correlation on the LDS-window kernel: geometry rule + option "warpcorr_window"
bool warp_corr81_pipe_fits(int B, int C, int H, int W);
int launch_warp_corr81_pipe(const float *in1, const float *x2, const float *flo, float *out, int B, int C, int H, int W,
                            int64_t bs1, int64_t bs2, int64_t bsf, int64_t bso, float flow_scale, int align_corners, float thr,
                            float scale, float slope, int do_leaky, hipStream_t st);

}  // namespace pwc
