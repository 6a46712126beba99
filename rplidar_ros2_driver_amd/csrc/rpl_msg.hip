// rpl_msg.hip — assembly of serialised (CDR) LaserScan / PointCloud2 messages in HBM
// (wire format in include/rplgpu_msg.h).
//
// Pure data movement: per message a prefix of < 0.5 KB (patched template) and the bulk arrays
// copied to their final offsets.  The arrays start at 4-byte (not 16-byte) aligned offsets
// inside a message — the frame id decides — so the copies are dword-granular: a wave moves
// 256 contiguous bytes per instruction on either side, which is what coalescing needs.
// One workgroup column per scan (blockIdx.y); the few workgroups of a column (blockIdx.x)
// walk the scan's payload in kChunk-dword pieces.  Few, because voxelised clouds are ~20 KB:
// a column sized for the worst case would be mostly workgroups that start only to leave.
#include <hip/hip_runtime.h>

#include <math.h>

#include "rpl_launch.hpp"
#include "rpl_msg.hpp"

namespace rpl {
namespace {

constexpr uint32_t kMsgThreads = 256;
constexpr uint32_t kChunk = 16384;  // dwords per workgroup

__device__ inline void put_prefix(uint32_t *msg, const rplmsg::Prefix &P) {
  for (uint32_t i = threadIdx.x; i < P.len / 4; i += kMsgThreads) msg[i] = P.words[i];
}

// LaserScan scalars, the reference's expressions (src/rplidar_node.cpp:623-627 and
// :634-638 / :665-669): fp64 divides, results rounded to float once.
__global__ __launch_bounds__(kMsgThreads) void k_msg_laserscan(
    const float *__restrict__ ranges, const float *__restrict__ intens, uint32_t n_stride,
    const uint32_t *__restrict__ beam_count, int scan_processing,
    const rplgpu_stamp_t *__restrict__ stamps, const double *__restrict__ scan_duration,
    rplmsg::Prefix P, uint8_t *__restrict__ msgs, uint32_t msg_stride,
    uint32_t *__restrict__ msg_len, uint32_t *__restrict__ status) {
  const uint32_t b = blockIdx.y;
  const uint32_t bc = beam_count[b];
  const uint64_t total = (uint64_t)P.len + 8ull * bc + 4ull;
  const bool fits = total <= msg_stride;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    msg_len[b] = (bc && fits) ? (uint32_t)total : 0u;
    if (status && bc && !fits) atomicOr(&status[b], RPLGPU_SCAN_OUT_TRUNCATED);
  }
  if (bc == 0 || !fits) return;
  uint32_t *msg = reinterpret_cast<uint32_t *>(msgs + (size_t)b * msg_stride);
  if (blockIdx.x * kChunk >= 2u * bc) return;
  if (blockIdx.x == 0) {
    put_prefix(msg, P);
    __syncthreads();  // the patches below overwrite template words
    if (threadIdx.x == 0) {
      msg[P.stamp_off / 4] = (uint32_t)stamps[b].sec;
      msg[P.stamp_off / 4 + 1] = stamps[b].nanosec;
      const double dur = scan_duration[b];
      const double denom = scan_processing ? (double)bc : (double)(bc > 1 ? bc - 1 : 1);
      float *f = reinterpret_cast<float *>(msg + P.a_off / 4);
      // f[0] angle_min, f[1] angle_max, f[5] range_min, f[6] range_max come with the template
      f[2] = (float)((2.0 * M_PI) / denom);
      f[3] = (float)(dur / denom);
      f[4] = (float)dur;
      msg[P.b_off / 4] = bc;
      msg[P.len / 4 + bc] = bc;  // intensities length word, right after ranges
    }
  }
  const float *r = ranges + (size_t)b * n_stride;
  const float *q = intens + (size_t)b * n_stride;
  uint32_t *out = msg + P.len / 4;
  for (uint32_t first = blockIdx.x * kChunk; first < 2u * bc; first += gridDim.x * kChunk) {
    const uint32_t last = min(first + kChunk, 2u * bc);
    for (uint32_t j = first + threadIdx.x; j < last; j += kMsgThreads) {
      if (j < bc)
        out[j] = __float_as_uint(r[j]);
      else
        out[j + 1] = __float_as_uint(q[j - bc]);
    }
  }
}

__global__ __launch_bounds__(kMsgThreads) void k_msg_cloud(
    const uint32_t *__restrict__ xyzi, uint32_t out_stride,
    const unsigned long long *__restrict__ scan_start, const uint32_t *__restrict__ n_points,
    const rplgpu_stamp_t *__restrict__ stamps, rplmsg::Prefix P, uint8_t *__restrict__ msgs,
    uint32_t msg_stride, uint32_t *__restrict__ msg_len, uint32_t *__restrict__ status) {
  const uint32_t b = blockIdx.y;
  const uint32_t np = n_points[b];
  const uint64_t total = (uint64_t)P.len + 16ull * np + 1ull;
  const bool fits = total <= msg_stride;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    msg_len[b] = fits ? (uint32_t)total : 0u;
    if (status && !fits) atomicOr(&status[b], RPLGPU_SCAN_OUT_TRUNCATED);
  }
  if (!fits) return;
  uint8_t *msg8 = msgs + (size_t)b * msg_stride;
  uint32_t *msg = reinterpret_cast<uint32_t *>(msg8);
  if (blockIdx.x != 0 && blockIdx.x * kChunk >= 4u * np) return;
  if (blockIdx.x == 0) {
    put_prefix(msg, P);
    __syncthreads();
    if (threadIdx.x == 0) {
      msg[P.stamp_off / 4] = (uint32_t)stamps[b].sec;
      msg[P.stamp_off / 4 + 1] = stamps[b].nanosec;
      msg[P.a_off / 4] = np;        // width
      msg[P.b_off / 4] = 16u * np;  // row_step
      msg[P.c_off / 4] = 16u * np;  // data.size()
      msg8[(size_t)P.len + 16ull * np] = 1;  // is_dense, the last byte
    }
  }
  const size_t src0 = scan_start ? (size_t)scan_start[b] * 4 : (size_t)b * out_stride * 4;
  const uint32_t *src = xyzi + src0;
  uint32_t *out = msg + P.len / 4;
  for (uint32_t first = blockIdx.x * kChunk; first < 4u * np; first += gridDim.x * kChunk) {
    const uint32_t last = min(first + kChunk, 4u * np);
    for (uint32_t j = first + threadIdx.x; j < last; j += kMsgThreads) out[j] = src[j];
  }
}

// The whole arena (clouds of every scan, e.g. of several sensors after rpl_fuse.hip) as ONE
// PointCloud2: width = *total points (a device word: the arena cursor), no host round trip.
__global__ __launch_bounds__(kMsgThreads) void k_msg_fused(
    const uint32_t *__restrict__ arena, const unsigned long long *__restrict__ total_points,
    unsigned long long arena_capacity, rplgpu_stamp_t stamp, rplmsg::Prefix P,
    uint8_t *__restrict__ msg8, unsigned long long msg_capacity,
    unsigned long long *__restrict__ msg_len, uint32_t *__restrict__ status) {
  const unsigned long long np = min(*total_points, arena_capacity);  // the cursor may overshoot
  const unsigned long long total = (unsigned long long)P.len + 16ull * np + 1ull;
  const bool fits = total <= msg_capacity && np < (1ull << 28);  // 32-bit lengths on the wire
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *msg_len = fits ? total : 0ull;
    if (status && !fits) atomicOr(status, RPLGPU_SCAN_OUT_TRUNCATED);
  }
  if (!fits) return;
  uint32_t *msg = reinterpret_cast<uint32_t *>(msg8);
  if (blockIdx.x == 0) {
    put_prefix(msg, P);
    __syncthreads();
    if (threadIdx.x == 0) {
      msg[P.stamp_off / 4] = (uint32_t)stamp.sec;
      msg[P.stamp_off / 4 + 1] = stamp.nanosec;
      msg[P.a_off / 4] = (uint32_t)np;
      msg[P.b_off / 4] = 16u * (uint32_t)np;
      msg[P.c_off / 4] = 16u * (uint32_t)np;
      msg8[(size_t)P.len + 16ull * np] = 1;  // is_dense
    }
  }
  uint32_t *out = msg + P.len / 4;
  const unsigned long long nw = 4ull * np;
  for (unsigned long long j = (unsigned long long)blockIdx.x * kMsgThreads + threadIdx.x; j < nw;
       j += (unsigned long long)gridDim.x * kMsgThreads)
    out[j] = arena[j];
}

}  // namespace

hipError_t launch_msg_fused(hipStream_t s, const float *arena,
                            const unsigned long long *total_points,
                            unsigned long long arena_capacity, rplgpu_stamp_t stamp,
                            const rplmsg::Prefix &P, uint8_t *msg, unsigned long long msg_capacity,
                            unsigned long long *msg_len, uint32_t *status) {
  hipLaunchKernelGGL(k_msg_fused, dim3(2048), dim3(kMsgThreads), 0, s,
                     reinterpret_cast<const uint32_t *>(arena), total_points, arena_capacity, stamp,
                     P, msg, msg_capacity, msg_len, status);
  return hipGetLastError();
}

hipError_t launch_msg_laserscan(hipStream_t s, const float *ranges, const float *intens,
                                uint32_t n_stride, const uint32_t *beam_count, uint32_t B,
                                int scan_processing, const rplgpu_stamp_t *stamps,
                                const double *scan_duration, const rplmsg::Prefix &P,
                                uint8_t *msgs, uint32_t msg_stride, uint32_t *msg_len,
                                uint32_t *status) {
  if (B == 0) return hipSuccess;
  const uint32_t gx = min((2u * n_stride + kChunk - 1) / kChunk, 4u);
  for (uint32_t b0 = 0; b0 < B; b0 += 65535u) {  // gridDim.y limit
    const uint32_t nb = min(B - b0, 65535u);
    hipLaunchKernelGGL(k_msg_laserscan, dim3(gx ? gx : 1, nb), dim3(kMsgThreads), 0, s,
                       ranges + (size_t)b0 * n_stride, intens + (size_t)b0 * n_stride, n_stride,
                       beam_count + b0, scan_processing, stamps + b0, scan_duration + b0, P,
                       msgs + (size_t)b0 * msg_stride, msg_stride, msg_len + b0,
                       status ? status + b0 : nullptr);
  }
  return hipGetLastError();
}

hipError_t launch_msg_cloud(hipStream_t s, const float *xyzi, uint32_t out_stride,
                            uint32_t max_points, const unsigned long long *scan_start,
                            const uint32_t *n_points, uint32_t B, const rplgpu_stamp_t *stamps,
                            const rplmsg::Prefix &P, uint8_t *msgs, uint32_t msg_stride,
                            uint32_t *msg_len, uint32_t *status) {
  if (B == 0) return hipSuccess;
  const uint32_t gx = min((4u * max_points + kChunk - 1) / kChunk, 2u);
  for (uint32_t b0 = 0; b0 < B; b0 += 65535u) {
    const uint32_t nb = min(B - b0, 65535u);
    hipLaunchKernelGGL(k_msg_cloud, dim3(gx ? gx : 1, nb), dim3(kMsgThreads), 0, s,
                       reinterpret_cast<const uint32_t *>(xyzi) +
                           (scan_start ? 0 : (size_t)b0 * out_stride * 4),
                       out_stride, scan_start ? scan_start + b0 : nullptr, n_points + b0,
                       stamps + b0, P, msgs + (size_t)b0 * msg_stride, msg_stride, msg_len + b0,
                       status ? status + b0 : nullptr);
  }
  return hipGetLastError();
}

}  // namespace rpl
