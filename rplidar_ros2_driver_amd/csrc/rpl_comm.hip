// rpl_comm.hip — device side of include/rplgpu_comm.h: the META block of a rank's arena and the
// turning of gathered slots into one contiguous cloud.  (The RCCL calls themselves are host code,
// rplgpu_api.hip.)  SURVEY.md §8(e): no collective on the data path, one all-gather to assemble.
#include <algorithm>

#include "rpl_device.hpp"
#include "rpl_launch.hpp"
#include "rpl_comm_layout.hpp"

namespace rpl {

// meta = [count lo, count hi, B, flags, B x {start lo, start hi, n_points}] (words): the rules live in
// rpl_comm_layout.hpp, shared with the host entry points the CPU tests drive
__global__ __launch_bounds__(256) void k_pack_meta(const unsigned long long *__restrict__ cursor,
                                                   const unsigned long long *__restrict__ scan_start,
                                                   const uint32_t *__restrict__ n_points, uint32_t B,
                                                   unsigned long long slot_points, uint32_t max_scans,
                                                   uint32_t *__restrict__ meta) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) layout::meta_head(*cursor, B, slot_points, meta);
  if (t < max_scans) layout::meta_scan(t, B, scan_start, n_points, slot_points, meta);
}

// The exchanged point: 12 bytes (x, y, intensity).  z is 0.0 for every point this path makes (a
// planar sensor, planar de-skew and poses), so it does not travel: a quarter less on the links.
// Compaction of a rank's arena (16-byte points) into its 12-byte slot; count = min(cursor, slot).
__global__ __launch_bounds__(256) void k_pack_xyi(const float4 *__restrict__ arena,
                                                  const unsigned long long *__restrict__ cursor,
                                                  unsigned long long slot_points,
                                                  float *__restrict__ slot) {
  const unsigned long long c = *cursor, n = c < slot_points ? c : slot_points;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const float4 p = arena[i];
    slot[3 * i] = p.x;
    slot[3 * i + 1] = p.y;
    slot[3 * i + 2] = p.w;
  }
}

// one block per (rank, piece): copies the valid part of rank r's slot to its place in the
// contiguous cloud and writes rank r's rows of the per-scan tables.  XYI: the slots hold 12-byte
// points, z = 0 is put back.
template <bool XYI>
__global__ __launch_bounds__(256) void k_unpack_gathered(
    const float *__restrict__ points_all, unsigned long long slot_points,
    const uint32_t *__restrict__ meta_all, uint32_t meta_words, uint32_t world, uint32_t max_scans,
    float4 *__restrict__ packed, unsigned long long *__restrict__ total,
    unsigned long long *__restrict__ scan_start_all, uint32_t *__restrict__ n_points_all,
    uint32_t *__restrict__ status) {
  const uint32_t r = blockIdx.y;
  unsigned long long off, mine, all;  // (world is small: every block adds the counts up)
  layout::rank_extent(meta_all, meta_words, world, slot_points, r, &off, &mine, &all);
  const uint32_t *m = meta_all + (size_t)r * meta_words;
  if (blockIdx.x == 0) {
    if (r == 0 && threadIdx.x == 0) *total = all;
    if (status && threadIdx.x == 0) status[r] = (m[3] & 1u) ? RPLGPU_SCAN_OUT_TRUNCATED : 0u;
    for (uint32_t s = threadIdx.x; s < max_scans; s += blockDim.x)
      layout::scan_row(m, s, max_scans, off, &scan_start_all[(size_t)r * max_scans + s],
                       &n_points_all[(size_t)r * max_scans + s]);
  }
  const float *src = points_all + (size_t)r * slot_points * (XYI ? 3u : 4u);
  float4 *dst = packed + off;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < mine;
       i += (unsigned long long)gridDim.x * blockDim.x)
    dst[i] = XYI ? make_float4(src[3 * i], src[3 * i + 1], 0.0f, src[3 * i + 2])
                 : reinterpret_cast<const float4 *>(src)[i];
}

// completion flag of a single-scan call (rplgpu_api.hip wait_scan): everything queued before this
// kernel on the stream is complete when it runs; the store is a system-scope release
__global__ void k_signal(uint32_t *flag, uint32_t seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_signal(hipStream_t s, uint32_t *flag, uint32_t seq) {
  hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s, flag, seq);
  return hipGetLastError();
}

// A staged scan (pinned host memory, read over the bus by the kernel itself) into HBM, for the
// single-scan paths that read a scan more than once (E5).  One kernel in the stream instead of a
// copy-engine transfer: no engine hand-over in front of the first kernel of the call.
__global__ __launch_bounds__(256) void k_stage_in(const uint2 *__restrict__ src, uint2 *__restrict__ dst,
                                                  uint32_t n_words2) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n_words2) dst[i] = src[i];
}
hipError_t launch_stage_in(hipStream_t s, const void *src, void *dst, uint32_t n_words2) {
  if (n_words2 == 0) return hipSuccess;
  hipLaunchKernelGGL(k_stage_in, dim3((n_words2 + 255u) / 256u), dim3(256), 0, s,
                     (const uint2 *)src, (uint2 *)dst, n_words2);
  return hipGetLastError();
}

hipError_t launch_pack_meta(hipStream_t s, const unsigned long long *cursor,
                            const unsigned long long *scan_start, const uint32_t *n_points,
                            uint32_t B, unsigned long long slot_points, uint32_t max_scans,
                            uint32_t *meta) {
  const uint32_t threads = max_scans > 1u ? max_scans : 1u;
  hipLaunchKernelGGL(k_pack_meta, dim3((threads + 255u) / 256u), dim3(256), 0, s, cursor, scan_start,
                     n_points, B, slot_points, max_scans, meta);
  return hipGetLastError();
}

hipError_t launch_pack_xyi(hipStream_t s, const float *arena, const unsigned long long *cursor,
                           unsigned long long slot_points, float *slot, uint32_t n_cu) {
  hipLaunchKernelGGL(k_pack_xyi, dim3(8u * (n_cu ? n_cu : 256u)), dim3(256), 0, s,
                     (const float4 *)arena, cursor, slot_points, slot);
  return hipGetLastError();
}

hipError_t launch_unpack_gathered(hipStream_t s, const float *points_all,
                                  unsigned long long slot_points, const uint32_t *meta_all,
                                  uint32_t meta_words, uint32_t world, uint32_t max_scans,
                                  float *packed, unsigned long long *total,
                                  unsigned long long *scan_start_all, uint32_t *n_points_all,
                                  uint32_t *status, uint32_t n_cu, bool xyi) {
  if (world == 0) return hipSuccess;
  const uint32_t per_rank = std::max<uint32_t>(1u, (4u * (n_cu ? n_cu : 256u) + world - 1u) / world);
  if (xyi)
    hipLaunchKernelGGL(k_unpack_gathered<true>, dim3(per_rank, world), dim3(256), 0, s, points_all,
                       slot_points, meta_all, meta_words, world, max_scans, (float4 *)packed, total,
                       scan_start_all, n_points_all, status);
  else
    hipLaunchKernelGGL(k_unpack_gathered<false>, dim3(per_rank, world), dim3(256), 0, s, points_all,
                       slot_points, meta_all, meta_words, world, max_scans, (float4 *)packed, total,
                       scan_start_all, n_points_all, status);
  return hipGetLastError();
}

}  // namespace rpl
