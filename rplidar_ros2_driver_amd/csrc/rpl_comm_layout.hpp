// rpl_comm_layout.hpp — the layout rules of the multi-GPU exchange (include/rplgpu_comm.h), written
// once for the device kernels (rpl_comm.hip) and for the host entry points
// rplgpu_pack_cloud_meta_host / rplgpu_unpack_gathered_host (rplgpu_api.hip): the world-size-2
// gloo tests on CPU drive the SAME code the kernels run, not a restatement of it.
//   meta = [count lo, count hi, B, flags, B x {start lo, start hi, n_points}]   (32-bit words)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RPL_HD __host__ __device__
#else
#define RPL_HD
#endif

namespace rpl {
namespace layout {

typedef unsigned long long u64;

RPL_HD inline void meta_head(u64 cursor, uint32_t B, u64 slot_points, uint32_t *meta) {
  const u64 k = cursor < slot_points ? cursor : slot_points;
  meta[0] = (uint32_t)k;
  meta[1] = (uint32_t)(k >> 32);
  meta[2] = B;
  meta[3] = cursor > slot_points ? 1u : 0u;
}

// entry of scan slot t (t < max_scans; slots >= B are empty)
RPL_HD inline void meta_scan(uint32_t t, uint32_t B, const u64 *scan_start, const uint32_t *n_points,
                             u64 slot_points, uint32_t *meta) {
  u64 st = 0ull;
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

// where rank r's points go in the contiguous cloud: offset = sum of the counts before it
RPL_HD inline void rank_extent(const uint32_t *meta_all, uint32_t meta_words, uint32_t world,
                               u64 slot_points, uint32_t r, u64 *off, u64 *mine, u64 *all) {
  u64 o = 0ull, m = 0ull, a = 0ull;
  for (uint32_t q = 0; q < world; ++q) {
    const uint32_t *mq = meta_all + (size_t)q * meta_words;
    u64 c = ((u64)mq[1] << 32) | mq[0];
    if (c > slot_points) c = slot_points;
    if (q < r) o += c;
    if (q == r) m = c;
    a += c;
  }
  *off = o;
  *mine = m;
  *all = a;
}

// row s of rank r's per-scan tables in the contiguous cloud
RPL_HD inline void scan_row(const uint32_t *m, uint32_t s, uint32_t max_scans, u64 off, u64 *start,
                            uint32_t *n_points) {
  const uint32_t B = m[2] < max_scans ? m[2] : max_scans;
  u64 st = 0ull;
  uint32_t np = 0u;
  if (s < B) {
    st = (((u64)m[5 + 3 * s] << 32) | m[4 + 3 * s]) + off;
    np = m[6 + 3 * s];
  }
  *start = np ? st : 0ull;
  *n_points = np;
}

}  // namespace layout
}  // namespace rpl
