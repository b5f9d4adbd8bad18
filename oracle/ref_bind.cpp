// ORACLE — test infrastructure.  pybind shim (ours) over the reference's own CPU BEV-IoU,
// compiled from /root/reference/mmdet3d/ops/pcdet_nms/src/iou3d_cpu.cpp where it lies
// (see oracle/Makefile).  Declares only the symbol that file defines (iou3d_cpu.cpp:232).
#include <torch/extension.h>
int boxes_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor, at::Tensor ans_iou_tensor);
PYBIND11_MODULE(pcdet_iou3d_cpu, m) {
  m.def("boxes_iou_bev_cpu", &boxes_iou_bev_cpu, "reference oriented BEV IoU (CPU)");
}
