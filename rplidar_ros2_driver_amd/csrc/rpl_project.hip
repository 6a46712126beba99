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

// sin and cos of an angle in [0, ~2 pi] (theta = float(i) * inc <= 2 pi * count / (count - 1)) in
// fp64, to well under one fp64 ulp of error: quadrant by two-constant Cody-Waite reduction (exact
// products inside the FMAs; q <= 4), then the fdlibm kernel polynomials on |r| <= pi/4.  The
// spec's (float)cos((double)theta) is then reproduced except where the fp64 value lies within
// ~1e-16 of a float rounding boundary (about three in 1e9 beams) — exactly the relation the
// library sincos has to the host's; the library routine carries the large-argument
// (Payne-Hanek) path and made this kernel compute-bound: 1.12 ms against the 0.5 ms its 3.1 GB
// of traffic take.
__device__ __forceinline__ void sincos_0_2pi(double x, double *sn, double *cn) {
  const double q = __builtin_rint(x * 6.36619772367581382433e-01);  // x * 2/pi
  double r = __builtin_fma(-q, 1.57079632679489655800e+00, x);      // pi/2, high part
  r = __builtin_fma(-q, 6.12323399573676603587e-17, r);             // pi/2, low part
  const double z = r * r;
  // __kernel_sin(r, 0)
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  const double s = __builtin_fma(z * r, __builtin_fma(z, ps, -1.66666666666666324348e-01), r);
  // __kernel_cos(r, 0)
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double c = w + (((1.0 - w) - hz) + z * (z * pc));
  const int n = (int)q & 3;
  const double ss = (n & 1) ? c : s, cc = (n & 1) ? s : c;
  *sn = (n & 2) ? -ss : ss;
  *cn = ((n + 1) & 2) ? -cc : cc;
}

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
      sincos_0_2pi((double)theta, &sn, &cn);
      typedef float nt_f4 __attribute__((ext_vector_type(4)));
      const nt_f4 v = {r * (float)cn, r * (float)sn, 0.0f, i_in[i]};
      __builtin_nontemporal_store(v, reinterpret_cast<nt_f4 *>(&out[pos]));  // (streamed once)
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
