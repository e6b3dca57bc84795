// ORACLE — TEST INFRASTRUCTURE ONLY.
// pybind11 front door for the reference's own CPU ROIAlign / ROIAlignRotated / rotated NMS, which
// build_ref.py compiles from the sources WHERE THEY LIE under /root/reference
// (detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp,
//  detectron2/layers/csrc/ROIAlignRotated/ROIAlignRotated_cpu.cpp,
//  detectron2/layers/csrc/nms_rotated/nms_rotated_cpu.cpp).
// Nothing of the reference is copied here: this file only names the entry points those translation units
// define (ROIAlign.h:7-27, ROIAlignRotated.h:7-27, nms_rotated.h:7-10) and exports them the way
// detectron2/layers/csrc/vision.cpp:96-112 does.
#include <torch/extension.h>
#include "ROIAlign/ROIAlign.h"
#include "ROIAlignRotated/ROIAlignRotated.h"
#include "nms_rotated/nms_rotated.h"

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("roi_align_forward", &detectron2::ROIAlign_forward_cpu);
  m.def("roi_align_backward", &detectron2::ROIAlign_backward_cpu);
  m.def("roi_align_rotated_forward", &detectron2::ROIAlignRotated_forward_cpu);
  m.def("roi_align_rotated_backward", &detectron2::ROIAlignRotated_backward_cpu);
  m.def("nms_rotated", &detectron2::nms_rotated_cpu);
}
