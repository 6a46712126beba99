// rpl_ror.hip — k_ror_mask: radius-outlier-removal keep mask (extension E5 of SURVEY.md §8
// a-ext): a kept point survives iff at least `k` OTHER kept points of the same scan lie
// within `r` (fp32: dx*dx + dy*dy <= r*r, products then sum, no FMA — the oracle's
// orc_ror_mask expression, oracle/oracle.cpp).  One 1024-thread workgroup owns one scan and
// keeps it in registers (same geometry as rpl_kernels.hip).
//
// The oracle is O(n^2).  Here the kept samples are binned by y into rows of height
// h >= 1.001 r (h grows with the scan's y extent so that 2048 rows always suffice), a
// counting sort groups their sample indices by row in LDS, and a point only visits the
// candidates of its own row and the two neighbouring rows — one contiguous slice of the index
// list — stopping as soon as k neighbours are found.  Candidate coordinates are recomputed
// from the raw node and the (cos, sin) table, exactly as the cloud kernels compute them, so
// the distance test sees bit-identical operands and the mask is bit-exact.
//
// Output: one bit per input sample (bit i of word i/32), 1 = the sample passed E1 AND E5.
#include "rpl_device.hpp"
#include "rpl_launch.hpp"

namespace rpl {

constexpr uint32_t kRorRows = 2048;

struct RorLds {
  uint16_t idx[kMaxN];          // sample indices grouped by row
  uint32_t rowstart[kRorRows];  // first slot of a row
  uint32_t rowfill[kRorRows];   // one past its last slot (after the scatter)
  uint32_t misc[8];             // 0/1 ymin/ymax (order-preserving uint encoding)
  uint32_t tmp[32];
};

// order-preserving float <-> uint map (for LDS atomicMin / atomicMax on floats)
__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__device__ __forceinline__ float2 node_xy(uint2 nd, const float2 *__restrict__ cs) {
  const float dm = nd_dist_m(nd_dist(nd));  // :590
  const float2 c = cs[nd_q14(nd)];
  return make_float2(dm * c.x, dm * c.y);  // E2
}

__global__ __launch_bounds__(kBlock) void k_ror_mask(const uint2 *__restrict__ nodes,
                                                     uint32_t n_stride,
                                                     const uint32_t *__restrict__ n_per_scan,
                                                     KParams p, Tables T,
                                                     uint32_t *__restrict__ mask_out,
                                                     uint32_t mask_stride) {
  __shared__ RorLds L;
  const uint32_t b = blockIdx.x;
  const uint32_t n = min(n_per_scan[b], kMaxN);
  const uint2 *scan = nodes + (size_t)b * n_stride;
  uint32_t *mask = mask_out + (size_t)b * mask_stride;
  const float2 *cs = p.inverted ? T.cs_inv : T.cs;
  const uint32_t wave = wave_id(), lane = lane_id();

  if (threadIdx.x == 0) {
    L.misc[0] = 0xFFFFFFFFu;
    L.misc[1] = 0u;
  }
  for (uint32_t t = threadIdx.x; t < kRorRows; t += kBlock) L.rowstart[t] = 0u;
  __syncthreads();

  // ---- E1 keep bits and the y extent of the kept points --------------------------------
  uint32_t kept = 0;
  float ymin = __uint_as_float(0x7F800000u), ymax = __uint_as_float(0xFF800000u);
  for (int j = 0; j < kIters; ++j) {
    const uint32_t i = ((uint32_t)(j * kWaves) + wave) * 64u + lane;
    if (i < n) {
      const uint2 nd = scan[i];
      if (nd_keep(nd_dist(nd), nd_quality(nd), p)) {
        kept |= 1u << j;
        const float y = node_xy(nd, cs).y;
        ymin = fminf(ymin, y);
        ymax = fmaxf(ymax, y);
      }
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    ymin = fminf(ymin, __shfl_xor(ymin, d, 64));
    ymax = fmaxf(ymax, __shfl_xor(ymax, d, 64));
  }
  if (lane == 0 && ymin <= ymax) {
    atomicMin(&L.misc[0], f2ord(ymin));
    atomicMax(&L.misc[1], f2ord(ymax));
  }
  __syncthreads();
  const bool any = L.misc[0] != 0xFFFFFFFFu;  // block-uniform
  uint32_t keep = 0;
  if (any) {
    const float y0 = ord2f(L.misc[0]), y1 = ord2f(L.misc[1]);
    const float r = sqrtf(p.ror_r2);
    // rows at least 1.001 r high (points within r in y are at most one row apart) and few
    // enough to fit the table whatever the extent
    const float h = fmaxf(r * 1.001f, (y1 - y0) / (float)(kRorRows - 2));
    const float inv_h = 1.0f / h;
    auto row_of = [&](float y) -> uint32_t {
      const float t = (y - y0) * inv_h;
      return min((uint32_t)fmaxf(t, 0.0f), kRorRows - 1u);
    };
    // ---- counting sort of the kept sample indices by row -------------------------------
    for (int j = 0; j < kIters; ++j) {
      if ((kept >> j) & 1u) {
        const uint32_t i = ((uint32_t)(j * kWaves) + wave) * 64u + lane;
        atomicAdd(&L.rowstart[row_of(node_xy(scan[i], cs).y)], 1u);
      }
    }
    __syncthreads();
    {
      const uint32_t r0 = L.rowstart[2 * threadIdx.x], r1 = L.rowstart[2 * threadIdx.x + 1];
      uint32_t tot;
      const uint32_t ex = block_excl_scan(r0 + r1, L.tmp, &tot);
      L.rowstart[2 * threadIdx.x] = ex;
      L.rowstart[2 * threadIdx.x + 1] = ex + r0;
      L.rowfill[2 * threadIdx.x] = ex;
      L.rowfill[2 * threadIdx.x + 1] = ex + r0;
    }
    __syncthreads();
    for (int j = 0; j < kIters; ++j) {
      if ((kept >> j) & 1u) {
        const uint32_t i = ((uint32_t)(j * kWaves) + wave) * 64u + lane;
        const uint32_t pos = atomicAdd(&L.rowfill[row_of(node_xy(scan[i], cs).y)], 1u);
        L.idx[pos] = (uint16_t)i;
      }
    }
    __syncthreads();
    // ---- neighbour count: own row and the two rows next to it --------------------------
    const float r2 = p.ror_r2;
    const uint32_t need = p.ror_k;
    for (int j = 0; j < kIters; ++j) {
      if ((kept >> j) & 1u) {
        const uint32_t i = ((uint32_t)(j * kWaves) + wave) * 64u + lane;
        const float2 me = node_xy(scan[i], cs);
        const uint32_t row = row_of(me.y);
        const uint32_t a = L.rowstart[row > 0u ? row - 1u : 0u];
        const uint32_t e = L.rowfill[min(row + 1u, kRorRows - 1u)];
        uint32_t cnt = 0;
        for (uint32_t q = a; q < e && cnt < need; q += 2u) {
          const uint32_t j0 = L.idx[q], j1 = (q + 1u < e) ? L.idx[q + 1u] : i;
          const float2 p0 = node_xy(scan[j0], cs), p1 = node_xy(scan[j1], cs);
          const float dx0 = me.x - p0.x, dy0 = me.y - p0.y;
          const float dx1 = me.x - p1.x, dy1 = me.y - p1.y;
          const float d0 = dx0 * dx0 + dy0 * dy0;  // products then sum (-ffp-contract=off)
          const float d1 = dx1 * dx1 + dy1 * dy1;
          cnt += (j0 != i && d0 <= r2) ? 1u : 0u;
          cnt += (j1 != i && d1 <= r2) ? 1u : 0u;
        }
        if (cnt >= need) keep |= 1u << j;
      }
    }
  }
  // ---- one bit per sample: chunk c = j*16 + wave covers samples [64c, 64c+64) -------------
  for (int j = 0; j < kIters; ++j) {
    const uint64_t m = __ballot((keep >> j) & 1u);
    const uint32_t c = (uint32_t)(j * kWaves) + wave;
    if (lane == 0 && 2u * c < mask_stride) mask[2u * c] = (uint32_t)m;
    if (lane == 1 && 2u * c + 1u < mask_stride) mask[2u * c + 1u] = (uint32_t)(m >> 32);
  }
}

hipError_t launch_ror_mask(hipStream_t s, const void *nodes, uint32_t n_stride,
                           const uint32_t *n_per_scan, uint32_t B, const KParams &p,
                           const Tables &T, uint32_t *mask, uint32_t mask_stride) {
  if (B == 0) return hipSuccess;
  hipLaunchKernelGGL(k_ror_mask, dim3(B), dim3(kBlock), 0, s, (const uint2 *)nodes, n_stride,
                     n_per_scan, p, T, mask, mask_stride);
  return hipGetLastError();
}

}  // namespace rpl
