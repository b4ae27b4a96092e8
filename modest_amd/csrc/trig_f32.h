// float32 sin / cos / atan2 for the BEV-IoU kernels.
//
// The reference evaluates cosf/sinf/atan2f (CUDA libdevice on its GPU path,
// glibc on its CPU twin, src/iou3d_cpu.cpp:69,162-163,124); neither is
// correctly rounded, so no implementation can match both bit for bit.  These
// evaluate in float64 and round once to float32 (error < 0.5000001 ulp), which
// is within 1 ulp of either reference flavour.
#pragma once
#include <hip/hip_runtime.h>

namespace modest {
__device__ __forceinline__ float sin_f32(float x) { return (float)sin((double)x); }
__device__ __forceinline__ float cos_f32(float x) { return (float)cos((double)x); }
__device__ __forceinline__ float atan2_f32(float y, float x) { return (float)atan2((double)y, (double)x); }
}  // namespace modest
