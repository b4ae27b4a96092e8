"""Drop-in for the reference's pybind11 extension module ``iou3d_nms_cuda``
(``utils/iou3d_nms/src/iou3d_nms_api.cpp:11-17``): the same five functions with
the same tensor arguments and return values, forwarded to the C ABI of
libmodest_hip.so.  The reference prints and ``exit(-1)``s on bad input
(``src/iou3d_nms.cpp:14-26``); here a ``RuntimeError`` is raised instead.

To make the MODEST / OpenPCDet sources pick this module up unchanged, put
``modest_amd/utils/iou3d_nms`` on ``sys.path``,
see INTEGRATION.md.
"""
import ctypes as _C

import torch as _torch

from ... import _lib


def _check(t, name, cuda):
    if not isinstance(t, _torch.Tensor):
        raise RuntimeError(f"{name} must be a tensor")
    if cuda and not t.is_cuda:
        raise RuntimeError(f"{name} must be CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous tensor")


def _stream():
    return _torch.cuda.current_stream().cuda_stream


def _pair(fn_name, boxes_a, boxes_b, out):
    for t, n in ((boxes_a, "boxes_a"), (boxes_b, "boxes_b"), (out, "ans")):
        _check(t, n, True)
        if t.dtype != _torch.float32:
            raise RuntimeError(f"{n} must be float32")
    lib = _lib.load()
    _lib.check(getattr(lib, fn_name)(boxes_a.data_ptr(), boxes_a.size(0), boxes_b.data_ptr(), boxes_b.size(0),
                                     out.data_ptr(), _stream()), fn_name)
    return 1


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    return _pair("modest_boxes_overlap_bev", boxes_a, boxes_b, ans_overlap)


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    return _pair("modest_boxes_iou_bev", boxes_a, boxes_b, ans_iou)


def _nms(fn_name, boxes, keep, thresh):
    _check(boxes, "boxes", True)
    _check(keep, "keep", False)
    if keep.dtype != _torch.int64 or keep.is_cuda:
        raise RuntimeError("keep must be a CPU LongTensor")
    lib = _lib.load()
    ctx = _lib.default_context(boxes.device.index or 0)
    num = _C.c_int(0)
    _lib.check(getattr(lib, fn_name)(ctx.handle, boxes.data_ptr(), boxes.size(0), float(thresh), keep.data_ptr(),
                                     _C.byref(num), _stream()), fn_name)
    return num.value


def nms_gpu(boxes, keep, nms_overlap_thresh):
    return _nms("modest_nms_bev", boxes, keep, nms_overlap_thresh)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    return _nms("modest_nms_normal", boxes, keep, nms_overlap_thresh)


def boxes_iou_bev_cpu(boxes_a_tensor, boxes_b_tensor, ans_iou_tensor):
    for t, n in ((boxes_a_tensor, "boxes_a"), (boxes_b_tensor, "boxes_b"), (ans_iou_tensor, "ans_iou")):
        _check(t, n, False)
        if t.is_cuda or t.dtype != _torch.float32:
            raise RuntimeError(f"{n} must be a CPU float32 tensor")
    lib = _lib.load()
    ctx = _lib.default_context(0)
    _lib.check(lib.modest_boxes_iou_bev_host(ctx.handle, boxes_a_tensor.data_ptr(), boxes_a_tensor.size(0),
                                             boxes_b_tensor.data_ptr(), boxes_b_tensor.size(0),
                                             ans_iou_tensor.data_ptr(), None), "modest_boxes_iou_bev_host")
    return 1
