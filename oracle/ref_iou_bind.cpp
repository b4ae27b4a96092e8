// ORACLE BUILD GLUE — test infrastructure only.
// Exposes the reference's own CPU BEV-IoU routine (boxes_iou_bev_cpu, declared in
// generate_cluster_mask/utils/iou3d_nms/src/iou3d_cpu.h:9 of the reference
// checkout) as a tiny Python module.  The reference source file is compiled
// where it lies under /root/reference by oracle/build_ref.py; nothing of it is
// copied into this repository.
#include <torch/extension.h>
#include "iou3d_cpu.h"

PYBIND11_MODULE(iou3d_ref, m) {
    m.def("boxes_iou_bev_cpu", &boxes_iou_bev_cpu, "reference rotated BEV IoU (CPU)");
}
