"""BEV IoU / NMS front-end with the call signatures of the reference's
``utils/iou3d_nms/iou3d_nms_utils.py`` (same file in OpenPCDet:
``pcdet/ops/iou3d_nms/iou3d_nms_utils.py``), backed by libmodest_hip.so through
the ``iou3d_nms_cuda``-compatible shim next to this file.

Boxes are (n,7) float32 ``[x, y, z, dx, dy, dz, heading]`` everywhere.
"""
import numpy as np
import torch

from . import iou3d_nms_cuda as _ext


def check_numpy_to_torch(x):
    return (torch.from_numpy(x).float(), True) if isinstance(x, np.ndarray) else (x, False)


def _need_boxes(*tensors):
    for t in tensors:
        assert t.shape[1] == 7


def _pairwise(kernel, a, b):
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    kernel(a.contiguous(), b.contiguous(), out)
    return out


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """Host boxes -> (N,M) IoU (reference :18-34).  The library has no CPU
    arithmetic path: the HIP kernel runs through a staging copy."""
    boxes_a, was_numpy = check_numpy_to_torch(boxes_a)
    boxes_b, _ = check_numpy_to_torch(boxes_b)
    assert not (boxes_a.is_cuda or boxes_b.is_cuda), 'Only support CPU tensors'
    _need_boxes(boxes_a, boxes_b)
    iou = _pairwise(_ext.boxes_iou_bev_cpu, boxes_a, boxes_b)
    return iou.numpy() if was_numpy else iou


def boxes_iou_bev(boxes_a, boxes_b):
    """Device boxes -> device (N,M) rotated BEV IoU (reference :37-51)."""
    _need_boxes(boxes_a, boxes_b)
    return _pairwise(_ext.boxes_iou_bev_gpu, boxes_a, boxes_b)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """3-D IoU = BEV overlap x height overlap / union volume (reference :54-87)."""
    _need_boxes(boxes_a, boxes_b)
    top = torch.min((boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1), (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1))
    bot = torch.max((boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1), (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1))
    inter = _pairwise(_ext.boxes_overlap_bev_gpu, boxes_a, boxes_b) * torch.clamp(top - bot, min=0)
    vol = (boxes_a[:, 3:6].prod(dim=1)).view(-1, 1) + (boxes_b[:, 3:6].prod(dim=1)).view(1, -1)
    return inter / torch.clamp(vol - inter, min=1e-6)


def _score_sorted_nms(kernel, boxes, scores, thresh, pre_maxsize=None):
    _need_boxes(boxes)
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    ranked = boxes[order].contiguous()
    keep = torch.LongTensor(ranked.size(0))
    n_kept = kernel(ranked, keep, thresh)
    return order[keep[:n_kept].to(order.device)].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """Rotated NMS; returns (indices into `boxes`, None) (reference :90-106)."""
    return _score_sorted_nms(_ext.nms_gpu, boxes, scores, thresh, pre_maxsize)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """Axis-aligned NMS (reference :109-122)."""
    return _score_sorted_nms(_ext.nms_normal_gpu, boxes, scores, thresh)
