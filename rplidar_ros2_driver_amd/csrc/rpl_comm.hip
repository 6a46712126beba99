// rpl_comm.hip — device side of include/rplgpu_comm.h: the META block of a rank's arena and the
// turning of gathered slots into one contiguous cloud.  (The RCCL calls themselves are host code,
// rplgpu_api.hip.)  SURVEY.md §8(e): no collective on the data path, one all-gather to assemble.
#include <algorithm>

#include "rpl_device.hpp"
#include "rpl_launch.hpp"

namespace rpl {

// meta = [count lo, count hi, B, flags, B x {start lo, start hi, n_points}] (words)
__global__ __launch_bounds__(256) void k_pack_meta(const unsigned long long *__restrict__ cursor,
                                                   const unsigned long long *__restrict__ scan_start,
                                                   const uint32_t *__restrict__ n_points, uint32_t B,
                                                   unsigned long long slot_points, uint32_t max_scans,
                                                   uint32_t *__restrict__ meta) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) {
    const unsigned long long c = *cursor;
    const unsigned long long k = c < slot_points ? c : slot_points;
    meta[0] = (uint32_t)k;
    meta[1] = (uint32_t)(k >> 32);
    meta[2] = B;
    meta[3] = c > slot_points ? 1u : 0u;
  }
  if (t < max_scans) {
    unsigned long long st = 0ull;
    uint32_t np = 0u;
    if (t < B) {
      st = scan_start[t];
      np = n_points[t];
      // a scan (partly) beyond the slot is cut like the slot is
      if (st >= slot_points) { np = 0u; st = 0ull; }
      else if (st + np > slot_points) np = (uint32_t)(slot_points - st);
    }
    meta[4 + 3 * t] = (uint32_t)st;
    meta[5 + 3 * t] = (uint32_t)(st >> 32);
    meta[6 + 3 * t] = np;
  }
}

// one block per (rank, piece): copies the valid part of rank r's slot to its place in the
// contiguous cloud and writes rank r's rows of the per-scan tables
__global__ __launch_bounds__(256) void k_unpack_gathered(
    const float4 *__restrict__ points_all, unsigned long long slot_points,
    const uint32_t *__restrict__ meta_all, uint32_t meta_words, uint32_t world, uint32_t max_scans,
    float4 *__restrict__ packed, unsigned long long *__restrict__ total,
    unsigned long long *__restrict__ scan_start_all, uint32_t *__restrict__ n_points_all,
    uint32_t *__restrict__ status) {
  const uint32_t r = blockIdx.y;
  // offset of rank r = sum of the counts before it (world is small: every block adds them up)
  unsigned long long off = 0ull, mine = 0ull, all = 0ull;
  for (uint32_t q = 0; q < world; ++q) {
    const uint32_t *m = meta_all + (size_t)q * meta_words;
    unsigned long long c = ((unsigned long long)m[1] << 32) | m[0];
    if (c > slot_points) c = slot_points;
    if (q < r) off += c;
    if (q == r) mine = c;
    all += c;
  }
  const uint32_t *m = meta_all + (size_t)r * meta_words;
  if (blockIdx.x == 0) {
    if (r == 0 && threadIdx.x == 0) *total = all;
    if (status && threadIdx.x == 0) status[r] = (m[3] & 1u) ? RPLGPU_SCAN_OUT_TRUNCATED : 0u;
    const uint32_t B = min(m[2], max_scans);
    for (uint32_t s = threadIdx.x; s < max_scans; s += blockDim.x) {
      unsigned long long st = 0ull;
      uint32_t np = 0u;
      if (s < B) {
        st = (((unsigned long long)m[5 + 3 * s] << 32) | m[4 + 3 * s]) + off;
        np = m[6 + 3 * s];
      }
      scan_start_all[(size_t)r * max_scans + s] = np ? st : 0ull;
      n_points_all[(size_t)r * max_scans + s] = np;
    }
  }
  const float4 *src = points_all + (size_t)r * slot_points;
  float4 *dst = packed + off;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < mine;
       i += (unsigned long long)gridDim.x * blockDim.x)
    dst[i] = src[i];
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

hipError_t launch_pack_meta(hipStream_t s, const unsigned long long *cursor,
                            const unsigned long long *scan_start, const uint32_t *n_points,
                            uint32_t B, unsigned long long slot_points, uint32_t max_scans,
                            uint32_t *meta) {
  const uint32_t threads = max_scans > 1u ? max_scans : 1u;
  hipLaunchKernelGGL(k_pack_meta, dim3((threads + 255u) / 256u), dim3(256), 0, s, cursor, scan_start,
                     n_points, B, slot_points, max_scans, meta);
  return hipGetLastError();
}

hipError_t launch_unpack_gathered(hipStream_t s, const float *points_all,
                                  unsigned long long slot_points, const uint32_t *meta_all,
                                  uint32_t meta_words, uint32_t world, uint32_t max_scans,
                                  float *packed, unsigned long long *total,
                                  unsigned long long *scan_start_all, uint32_t *n_points_all,
                                  uint32_t *status, uint32_t n_cu) {
  if (world == 0) return hipSuccess;
  const uint32_t per_rank = std::max<uint32_t>(1u, (4u * (n_cu ? n_cu : 256u) + world - 1u) / world);
  hipLaunchKernelGGL(k_unpack_gathered, dim3(per_rank, world), dim3(256), 0, s,
                     (const float4 *)points_all, slot_points, meta_all, meta_words, world, max_scans,
                     (float4 *)packed, total, scan_start_all, n_points_all, status);
  return hipGetLastError();
}

}  // namespace rpl
