// rpl_fuse.hip — clouds of several sensors into one frame (first step).
//
// The reference broadcasts one static identity transform base_link -> frame_id per node
// (src/rplidar_node.cpp:183-197) and leaves fusion to downstream consumers; BASELINE config 5
// (8 sensors -> one fused PointCloud2) needs the per-sensor rigid transform applied to the
// clouds before they are handed on as one message.  In place, one pose per scan, 16-byte loads
// and stores; the arithmetic order is part of the spec (fp32, products then sums left to right,
// no FMA: the library is built with -ffp-contract=off):
//   x' = ((r00*x + r01*y) + r02*z) + t0,  y' and z' likewise, intensity untouched.
#include <hip/hip_runtime.h>

#include "rpl_launch.hpp"

namespace rpl {
namespace {

constexpr uint32_t kFuseThreads = 256;
constexpr uint32_t kFuseChunk = 4096;  // points per workgroup per trip

__global__ __launch_bounds__(kFuseThreads) void k_transform_clouds(
    float4 *__restrict__ xyzi, uint32_t out_stride, const unsigned long long *__restrict__ scan_start,
    const uint32_t *__restrict__ n_points, const float *__restrict__ pose) {
  const uint32_t b = blockIdx.y;
  const uint32_t np = n_points[b];
  if (blockIdx.x * kFuseChunk >= np) return;
  const float *m = pose + (size_t)b * 12;
  const float r00 = m[0], r01 = m[1], r02 = m[2], t0 = m[3];
  const float r10 = m[4], r11 = m[5], r12 = m[6], t1 = m[7];
  const float r20 = m[8], r21 = m[9], r22 = m[10], t2 = m[11];
  float4 *pts = xyzi + (scan_start ? (size_t)scan_start[b] : (size_t)b * out_stride);
  for (uint32_t first = blockIdx.x * kFuseChunk; first < np; first += gridDim.x * kFuseChunk) {
    const uint32_t last = min(first + kFuseChunk, np);
    for (uint32_t i = first + threadIdx.x; i < last; i += kFuseThreads) {
      const float4 p = pts[i];
      float4 q;
      q.x = ((r00 * p.x + r01 * p.y) + r02 * p.z) + t0;
      q.y = ((r10 * p.x + r11 * p.y) + r12 * p.z) + t1;
      q.z = ((r20 * p.x + r21 * p.y) + r22 * p.z) + t2;
      q.w = p.w;
      pts[i] = q;
    }
  }
}

}  // namespace

hipError_t launch_transform_clouds(hipStream_t s, float *xyzi, uint32_t out_stride,
                                   uint32_t max_points, const unsigned long long *scan_start,
                                   const uint32_t *n_points, uint32_t B, const float *pose) {
  if (B == 0) return hipSuccess;
  const uint32_t gx = max(1u, min((max_points + kFuseChunk - 1) / kFuseChunk, 2u));
  for (uint32_t b0 = 0; b0 < B; b0 += 65535u) {  // gridDim.y limit
    const uint32_t nb = min(B - b0, 65535u);
    hipLaunchKernelGGL(k_transform_clouds, dim3(gx, nb), dim3(kFuseThreads), 0, s,
                       reinterpret_cast<float4 *>(xyzi) + (scan_start ? 0 : (size_t)b0 * out_stride),
                       out_stride, scan_start ? scan_start + b0 : nullptr, n_points + b0,
                       pose + (size_t)b0 * 12);
  }
  return hipGetLastError();
}

}  // namespace rpl
