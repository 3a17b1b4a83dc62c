"""``correlation_cuda`` -- module-level stand-in for the reference's pybind extension.

Same two entry points and argument order as ``PYBIND11_MODULE`` in the reference
(models/correlation_package/correlation_cuda.cc:169-172):

    forward (input1, input2, rbot1, rbot2, output,
             pad_size, kernel_size, max_displacement, stride1, stride2, corr_type_multiply) -> int
    backward(input1, input2, rbot1, rbot2, gradOutput, gradInput1, gradInput2, <same six ints>) -> int

so the reference's own ``correlation.py`` (which does ``import correlation_cuda`` at :4) can run on
top of libpwc_hip.so unchanged.  As in the reference, the caller passes EMPTY tensors for
``output`` / ``gradInput*`` and the callee resizes them (correlation_cuda.cc:36-42); ``rbot1`` /
``rbot2`` (the NHWC scratch of the CUDA design) are accepted and left untouched -- the HIP kernels
read NCHW directly.  Normalisation follows the reference's native kernel: ``/ (kernel_size**2 * C)``
(correlation_cuda_kernel.cu:104,143).  Returns 1 on success like the reference (.cc:85); failures
raise ``RuntimeError``.
"""
from opticalflow_amd import ops as _ops


def forward(input1, input2, rbot1, rbot2, output, pad_size, kernel_size, max_displacement, stride1, stride2,
            corr_type_multiply):
    B, C, H, W = input1.shape
    nch, oh, ow = _ops.corr_output_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    output.resize_((B, nch, oh, ow))
    _ops.correlation(input1.contiguous(), input2.contiguous(), pad_size, kernel_size, max_displacement,
                     stride1, stride2, corr_type_multiply, normalize=True, out=output)
    return 1


def backward(input1, input2, rbot1, rbot2, gradOutput, gradInput1, gradInput2, pad_size, kernel_size,
             max_displacement, stride1, stride2, corr_type_multiply):
    g1, g2 = _ops.correlation_backward(input1.contiguous(), input2.contiguous(), gradOutput.contiguous(),
                                       pad_size, kernel_size, max_displacement, stride1, stride2,
                                       corr_type_multiply, normalize=True)
    gradInput1.resize_(g1.shape).copy_(g1)
    gradInput2.resize_(g2.shape).copy_(g2)
    return 1
