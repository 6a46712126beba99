// rpl_project.hip — k_laserscan_to_cloud: the binned LaserScan as a PointCloud2 (E7, SURVEY.md
// §8(f) row 3: the `laser_geometry`-style alternative cloud source).
//
// Producer of the input: publish_scan's ranges[] / intensities[] / beam count and the scalars
// angle_min = 0, angle_increment (/root/reference src/rplidar_node.cpp:618-662 Mode A,
// :663-680 Mode B) — here the outputs of rplgpu_laserscan_batch_dev, still resident in HBM.
// Spec (not in the reference; oracle: oracle.cpp orc_laserscan_to_cloud):
//   count  = beam_count[b];   inc = Mode A float(2*pi / double(count))          (:635)
//                                   Mode B float(2*pi / double(max(count-1, 1))) (:666-668)
//   beam i is kept iff ranges[i] is finite (Mode A's untouched bins are +inf, :640) and, with
//   clip_enable, range_min <= ranges[i] <= range_max;
//   theta = angle_min + float(i) * inc   (float32: one multiply, angle_min = 0)
//   x = ranges[i] * (float)cos((double)theta),  y = ranges[i] * (float)sin((double)theta),
//   z = 0, intensity = intensities[i]; points in beam order (stable compaction), E3 layout.
// One 1024-thread workgroup per scan; a beam costs 8 B read + 16 B written when kept.
#include "rpl_device.hpp"
#include "rpl_launch.hpp"

namespace rpl {

__global__ __launch_bounds__(kBlock) void k_laserscan_to_cloud(
    const float *__restrict__ ranges, const float *__restrict__ intens, uint32_t n_stride,
    const uint32_t *__restrict__ beam_count, KParams p, float4 *__restrict__ xyzi,
    uint32_t out_stride, uint32_t *__restrict__ n_points, uint32_t *__restrict__ status) {
  __shared__ uint32_t s_wave[kWaves + 1];
  const uint32_t b = blockIdx.x;
  const uint32_t count = min(beam_count[b], n_stride);  // never past the scan's slot
  const float *r_in = ranges + (size_t)b * n_stride;
  const float *i_in = intens + (size_t)b * n_stride;
  float4 *out = xyzi + (size_t)b * out_stride;
  const double den = p.scan_processing ? (double)count : (double)(count > 1u ? count - 1u : 1u);
  const float inc = (float)(kTwoPi / den);  // IEEE fp64 divide, one rounding to float
  uint32_t base_out = 0;
  for (uint32_t base = 0; base < count; base += kBlock) {
    const uint32_t i = base + threadIdx.x;
    const float r = (i < count) ? r_in[i] : __builtin_inff();
    bool keep = (i < count) && (r < __builtin_inff()) && (r > -__builtin_inff());  // NaN: dropped
    if (p.clip_enable) keep = keep && (r >= p.range_min) && (r <= p.range_max);
    const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
    if (lane_id() == 0) s_wave[wave_id()] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = base_out, total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const uint32_t c = s_wave[w];
      before += (w < (int)wave_id()) ? c : 0u;
      total += c;
    }
    const uint32_t pos = before + (uint32_t)__popcll(m & lanemask_lt());
    if (keep && pos < out_stride) {
      const float theta = 0.0f + (float)i * inc;
      double sn, cn;
      sincos((double)theta, &sn, &cn);
      out[pos] = make_float4(r * (float)cn, r * (float)sn, 0.0f, i_in[i]);
    }
    base_out += total;
    __syncthreads();  // s_wave is rewritten by the next chunk
  }
  if (threadIdx.x == 0) {
    n_points[b] = min(base_out, out_stride);
    if (status) status[b] = (base_out > out_stride) ? RPLGPU_SCAN_OUT_TRUNCATED : 0u;
  }
}

hipError_t launch_laserscan_to_cloud(hipStream_t s, const float *ranges, const float *intens,
                                     uint32_t n_stride, const uint32_t *beam_count, uint32_t B,
                                     const KParams &p, float *xyzi, uint32_t out_stride,
                                     uint32_t *n_points, uint32_t *status) {
  if (B == 0) return hipSuccess;
  hipLaunchKernelGGL(k_laserscan_to_cloud, dim3(B), dim3(kBlock), 0, s, ranges, intens, n_stride,
                     beam_count, p, (float4 *)xyzi, out_stride, n_points, status);
  return hipGetLastError();
}

}  // namespace rpl
