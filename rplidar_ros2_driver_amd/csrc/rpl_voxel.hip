// rpl_voxel.hip — k_cloud_voxel: raw scan -> clipped, voxel-downsampled PointCloud2
// (extensions E1 + E2 + E4 of SURVEY.md §8 a-ext) in ONE streaming pass over the
// packed 8-byte nodes.  One 1024-thread workgroup owns one scan.
//
// Phase S (streaming, straight-line code, no data-dependent loops):
//   8-byte node -> keep mask, dist_m, (cos,sin) LUT -> x, y -> cell (iy, ix) and the
//   fixed-point offsets inside the cell.  A smooth ring visits a 5 cm cell ~10-200
//   samples in a row, so consecutive lanes mostly share a cell: runs are detected with
//   one DPP lane shift + ballots, their partial sums come from three DPP prefix scans,
//   and only the LAST lane of a run writes one 16-byte run record to an LDS queue
//   (each wave owns a private queue segment: no atomics, deterministic order).
// Phase R (per scan, regular data-parallel passes over the <= 7168 run records):
//   counting sort by row + rank inside the row -> records in (iy, ix) order ->
//   segmented integer sums over equal keys -> one output point per cell.
//
// Fixed point: offset = (x - ix*leaf) * 2^K + 2^15 with 2^-K = ulp(leaf) (K = 28 for
// 5 cm).  One fp32 FMA yields x - ix*leaf EXACTLY whenever x is a multiple of 2^-K
// (|x| >= 3 cm at K = 28), so the integer sums are exact and sum/count reproduces the
// spec's fp64 running sum bit for bit; closer to the axes the per-point error is
// <= 2^-(K+1) m (1.9e-9 m), far below the 1e-6 m bar.  Integer sums are order
// independent, so the kernel is run-to-run deterministic.
//
// A scan whose records do not fit (or that spans > 2048 rows) is processed in key
// bands: the key range is bisected until a band fits, each band re-streaming the scan
// (from L2 / Infinity Cache).
#include "rpl_device.hpp"
#include "rpl_launch.hpp"

namespace rpl {

constexpr uint32_t kRecPerWave = 448;                 // run records per wave segment
constexpr uint32_t kRecCap = kRecPerWave * kWaves;    // 7168 records (112 KiB)
constexpr uint32_t kRecPerThread = kRecCap / kBlock;  // 7
constexpr uint32_t kRowCap = 2048;                    // rows the counting sort handles
constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;

template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, false);
}
// inclusive wave64 prefix sum entirely in the VALU (DPP row shifts + row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
  v = dpp_add<0x111, 0xF>(v);  // row_shr:1
  v = dpp_add<0x112, 0xF>(v);  // row_shr:2
  v = dpp_add<0x114, 0xF>(v);  // row_shr:4
  v = dpp_add<0x118, 0xF>(v);  // row_shr:8
  v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 -> rows 1,3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 -> rows 2,3
  return v;
}

struct VoxelLds {
  uint4 rec[kRecCap];          // {key, sum_x, sum_y, count<<16 | intensity_sum}
  uint32_t rowstart[kRowCap];
  uint32_t rowfill[kRowCap];
  alignas(16) uint32_t bucket[kRecCap];  // (ix << 16 | record index) grouped by row
  uint32_t wcount[kWaves];
  uint32_t band_lo[34], band_hi[34];
  uint32_t misc[16];  // 1 status, 2 overflow, 3 sp, 4 rowmin, 5 rowmax, 7 out_base
  uint32_t tmp[32];
};

// One wave-pass of phase S: turn 64 (key, qx, qy, inten) items into run records.
// `key == kEmptyKey` marks a lane that contributes nothing.  Returns the new record
// count of this wave's queue segment (wave-uniform).
__device__ __forceinline__ uint32_t voxel_wave_pass(VoxelLds &L, uint32_t wcnt, uint32_t key,
                                                    uint32_t qx, uint32_t qy, uint32_t inten) {
  const bool kept = key != kEmptyKey;
  const uint64_t keptmask = __ballot(kept);
  // previous lane's key (wave_shr:1); lane 0 sees "no key"
  const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)kEmptyKey, (int)key, 0x138,
                                                              0xF, 0xF, false);
  const uint64_t headmask = __ballot(kept && (key != prev));
  const uint64_t cont = keptmask & ~headmask;         // lanes continuing the previous lane's run
  const uint64_t tailmask = keptmask & ~(cont >> 1);  // kept lanes whose successor starts anew
  const uint32_t lane = lane_id();

  const uint32_t ci = kept ? ((1u << 16) | inten) : 0u;  // [count : 16 | intensity sum : 16]
  const uint32_t Px = wave_incl_scan_dpp(qx);
  const uint32_t Py = wave_incl_scan_dpp(qy);
  const uint32_t Pc = wave_incl_scan_dpp(ci);
  // exclusive prefix at the head lane of my run
  const uint64_t below = headmask & ((2ull << lane) - 1ull);
  const int h = 63 - __builtin_clzll(below | 1ull);
  const uint32_t Ex = (uint32_t)__shfl((int)(Px - qx), h, 64);
  const uint32_t Ey = (uint32_t)__shfl((int)(Py - qy), h, 64);
  const uint32_t Ec = (uint32_t)__shfl((int)(Pc - ci), h, 64);

  const uint32_t ntails = (uint32_t)__popcll(tailmask);
  if (wcnt + ntails > kRecPerWave) {  // wave-uniform: queue segment full -> bisect the band
    if (lane == 0) L.misc[2] = 1u;
    return wcnt;
  }
  if ((tailmask >> lane) & 1ull) {
    const uint32_t pos = wcnt + (uint32_t)__popcll(tailmask & lanemask_lt());
    L.rec[wave_id() * kRecPerWave + pos] = make_uint4(key, Px - Ex, Py - Ey, Pc - Ec);
  }
  return wcnt + ntails;
}

// a / d without v_div_scale / v_rcp / v_div_fmas / v_div_fixup: `rd` = RN(1/d), one
// multiply, the exact FMA remainder and one FMA correction (Markstein: a faithful first
// quotient plus the correctly rounded reciprocal give the correctly rounded quotient).
// The claim is not taken on faith: k_validate_div below compares it bit for bit with the
// IEEE divide over the whole operand range for the divisor in use, on this device, and
// the kernels only take this path after that check passed.  (The sign of a zero quotient
// may differ; every user takes floor() -> int of it, where -0 and +0 coincide.)
__device__ __forceinline__ float div_by(float a, float d, float rd) {
  float q = a * rd;
  float e = fmaf(-q, d, a);
  return fmaf(e, rd, q);
}

template <bool FAST_DIV>
__global__ __launch_bounds__(kBlock) void k_cloud_voxel(
    const uint2 *__restrict__ nodes, uint32_t n_stride, const uint32_t *__restrict__ n_per_scan,
    KParams p, Tables T, float4 *__restrict__ xyzi, uint32_t out_stride,
    uint32_t *__restrict__ n_points, uint32_t *__restrict__ status) {
  __shared__ VoxelLds L;

  const uint32_t b = blockIdx.x;
  const uint32_t n = min(n_per_scan[b], kMaxN);
  const uint2 *scan = nodes + (size_t)b * n_stride;
  float4 *out = xyzi + (size_t)b * out_stride;

  if (threadIdx.x < 16) L.misc[threadIdx.x] = 0u;
  if (threadIdx.x == 0) {
    L.band_lo[0] = 0u;
    L.band_hi[0] = 0xFFFFFFFEu;
    L.misc[3] = 1u;  // stack pointer
  }
  __syncthreads();

  const float leaf = p.voxel_leaf;
  const float rleaf = p.inv_leaf;
  const float vscale = p.vox_scale_f;
  const int vbias = p.vox_bias;
  const float2 *cs = p.inverted ? T.cs_inv : T.cs;
  const uint32_t ishift = p.is_new_protocol ? 0u : 2u;
  uint32_t flags = 0;
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_mark = clock64();
#define RPL_MARK(i)                      \
  {                                      \
    unsigned long long now_ = clock64(); \
    tacc[i] += now_ - t_mark;            \
    t_mark = now_;                       \
  }

  while (true) {
    // ---- pop a key band -----------------------------------------------------------
    const uint32_t sp = L.misc[3];
    if (sp == 0) break;
    const uint32_t klo = L.band_lo[sp - 1], khi = L.band_hi[sp - 1];
    __syncthreads();
    if (threadIdx.x == 0) {
      L.misc[2] = 0u;
      L.misc[3] = sp - 1;
      L.misc[4] = 0xFFFFFFFFu;
      L.misc[5] = 0u;
    }
    __syncthreads();

    // ---- phase S: stream the scan; the next 4 x 8 B per thread are already in flight
    //      while the current ones are processed (software prefetch across iterations)
    uint32_t wcnt = 0;
    constexpr int UNR = 4;
    uint2 nxt[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      uint32_t i = (uint32_t)u * kBlock + threadIdx.x;
      nxt[u] = (i < n) ? scan[i] : make_uint2(0u, 0u);
    }
    for (uint32_t base = 0; base < n; base += kBlock * UNR) {
      uint2 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        v[u] = nxt[u];
        uint32_t i = base + kBlock * UNR + (uint32_t)u * kBlock + threadIdx.x;
        nxt[u] = (i < n) ? scan[i] : make_uint2(0u, 0u);
      }
      if (*(volatile uint32_t *)&L.misc[2]) break;  // band already known not to fit
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const uint32_t d = nd_dist(v[u]);
        const uint32_t qual = nd_quality(v[u]);
        const float df = __uint2float_rn(d);
        const float dm = FAST_DIV ? div_by(df, 4000.0f, 0.00025f) : df / 4000.0f;  // :590
        bool kept = nd_keep(d, qual, dm, p);                                         // E1
        const float2 c = cs[nd_q14(v[u])];
        const float x = dm * c.x, y = dm * c.y;                                      // E2
        const float tx = FAST_DIV ? div_by(x, leaf, rleaf) : x / leaf;               // E4 cell
        const float ty = FAST_DIV ? div_by(y, leaf, rleaf) : y / leaf;
        const float fx = floorf(tx), fy = floorf(ty);
        const bool inrange = (fabsf(fx) < 32767.0f) && (fabsf(fy) < 32767.0f);
        if (kept && !inrange) flags |= RPLGPU_SCAN_CELL_RANGE;
        uint32_t key = ((uint32_t)((int)fy + 32768) << 16) | (uint32_t)((int)fx + 32768);
        kept = kept && inrange && (key >= klo) && (key <= khi);
        const int ox = (int)rintf(fmaf(-fx, leaf, x) * vscale) + vbias;
        const int oy = (int)rintf(fmaf(-fy, leaf, y) * vscale) + vbias;
        const uint32_t qx = kept ? (uint32_t)min(max(ox, 0), 0x1FFFFFF) : 0u;
        const uint32_t qy = kept ? (uint32_t)min(max(oy, 0), 0x1FFFFFF) : 0u;
        const uint32_t inten = kept ? (qual >> ishift) : 0u;
        key = kept ? key : kEmptyKey;
        wcnt = voxel_wave_pass(L, wcnt, key, qx, qy, inten);
      }
    }
    if (lane_id() == 0) L.wcount[wave_id()] = wcnt;
    __syncthreads();
    RPL_MARK(0)

    auto bisect = [&]() {  // block-uniform: replace the band by its two halves
      if (threadIdx.x == 0) {
        uint32_t s = L.misc[3];
        if (klo == khi || s + 2 > 33) {
          L.misc[1] |= RPLGPU_SCAN_TABLE_FULL;  // cannot happen: one key is one cell/row
        } else {
          uint32_t mid = klo + (khi - klo) / 2;
          L.band_lo[s] = mid + 1;  // upper half is processed after the lower half
          L.band_hi[s] = khi;
          L.band_lo[s + 1] = klo;
          L.band_hi[s + 1] = mid;
          L.misc[3] = s + 2;
        }
      }
      __syncthreads();
    };
    if (L.misc[2]) {
      bisect();
      continue;
    }

    // ---- phase R ---------------------------------------------------------------------
    // my (up to 7) records: linear index -> (wave segment, slot)
    uint4 mine[kRecPerThread];
    uint32_t rmin = 0xFFFFFFFFu, rmax = 0u;
#pragma unroll
    for (int k = 0; k < (int)kRecPerThread; ++k) {
      uint32_t idx = threadIdx.x + (uint32_t)k * kBlock;
      uint32_t w = idx / kRecPerWave, j = idx - w * kRecPerWave;
      bool ok = j < L.wcount[w];
      mine[k] = ok ? L.rec[idx] : make_uint4(kEmptyKey, 0u, 0u, 0u);
      if (ok) {
        rmin = min(rmin, mine[k].x >> 16);
        rmax = max(rmax, mine[k].x >> 16);
      }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      rmin = min(rmin, (uint32_t)__shfl_xor((int)rmin, d, 64));
      rmax = max(rmax, (uint32_t)__shfl_xor((int)rmax, d, 64));
    }
    if (lane_id() == 0 && rmin != 0xFFFFFFFFu) {
      atomicMin(&L.misc[4], rmin);
      atomicMax(&L.misc[5], rmax);
    }
    for (uint32_t t = threadIdx.x; t < kRowCap; t += kBlock) L.rowstart[t] = 0u;
    __syncthreads();
    RPL_MARK(1)
    rmin = L.misc[4];
    const uint32_t out_base = L.misc[7];
    uint32_t ncell = 0;
    if (rmin != 0xFFFFFFFFu) {  // at least one record (block-uniform)
      if (L.misc[5] - rmin + 1u > kRowCap) {
        bisect();
        continue;
      }
      // counting sort over rows
#pragma unroll
      for (int k = 0; k < (int)kRecPerThread; ++k)
        if (mine[k].x != kEmptyKey) atomicAdd(&L.rowstart[(mine[k].x >> 16) - rmin], 1u);
      __syncthreads();
      RPL_MARK(2)
      uint32_t nrec;
      {  // exclusive scan over kRowCap = 2 rows per thread
        uint32_t r0 = L.rowstart[2 * threadIdx.x], r1 = L.rowstart[2 * threadIdx.x + 1];
        uint32_t ex = block_excl_scan(r0 + r1, L.tmp, &nrec);
        L.rowstart[2 * threadIdx.x] = ex;
        L.rowstart[2 * threadIdx.x + 1] = ex + r0;
        L.rowfill[2 * threadIdx.x] = ex;
        L.rowfill[2 * threadIdx.x + 1] = ex + r0;
      }
      __syncthreads();
      RPL_MARK(3)
#pragma unroll
      for (int k = 0; k < (int)kRecPerThread; ++k) {
        if (mine[k].x != kEmptyKey) {
          uint32_t idx = threadIdx.x + (uint32_t)k * kBlock;
          uint32_t pos = atomicAdd(&L.rowfill[(mine[k].x >> 16) - rmin], 1u);
          L.bucket[pos] = (mine[k].x << 16) | idx;  // (ix, record index): unique
        }
      }
      __syncthreads();
      RPL_MARK(4)
      // rank inside the row, then permute the records in place (they are all in registers)
#pragma unroll
      for (int k = 0; k < (int)kRecPerThread; ++k) {
        if (mine[k].x != kEmptyKey) {
          uint32_t idx = threadIdx.x + (uint32_t)k * kBlock;
          uint32_t row = (mine[k].x >> 16) - rmin;
          uint32_t me = (mine[k].x << 16) | idx;
          uint32_t s0 = L.rowstart[row], s1 = L.rowfill[row];
          uint32_t rank = s0;
          for (uint32_t m = s0; m < s1; ++m) rank += (L.bucket[m] < me);
          L.rec[rank] = mine[k];
        }
      }
      __syncthreads();
      RPL_MARK(5)
      // heads of equal-key groups -> cell index; thread t owns sorted records [7t, 7t+7)
      const uint32_t r_lo = threadIdx.x * kRecPerThread;
      uint32_t headbits = 0, nheads = 0;
      uint32_t prevkey = (r_lo > 0 && r_lo <= nrec) ? L.rec[r_lo - 1].x : kEmptyKey;
#pragma unroll
      for (int k = 0; k < (int)kRecPerThread; ++k) {
        uint32_t r = r_lo + k;
        uint32_t key = (r < nrec) ? L.rec[r].x : kEmptyKey;
        if (r < nrec && key != prevkey) {
          headbits |= 1u << k;
          ++nheads;
        }
        prevkey = key;
      }
      uint32_t cell = block_excl_scan(nheads, L.tmp, &ncell);
      RPL_MARK(6)
      // position of every cell's first record (the bucket array is free again)
#pragma unroll
      for (int k = 0; k < (int)kRecPerThread; ++k)
        if ((headbits >> k) & 1u) L.bucket[cell++] = r_lo + k;
      __syncthreads();
      // one cell per thread, coalesced 16-byte output rows
      const double inv_scale = 1.0 / p.vox_scale;  // exact power of two
      const uint32_t nemit = min(ncell, out_stride > out_base ? out_stride - out_base : 0u);
      for (uint32_t c = threadIdx.x; c < nemit; c += kBlock) {
        uint32_t r = L.bucket[c];
        const uint32_t key = L.rec[r].x;
        uint64_t sx = 0, sy = 0;
        uint32_t cnt = 0, isum = 0;
        for (; r < nrec; ++r) {  // segmented sum over the records of this cell
          uint4 q = L.rec[r];
          if (q.x != key) break;
          sx += q.y;
          sy += q.z;
          cnt += q.w >> 16;
          isum += q.w & 0xFFFFu;
        }
        int ix = (int)(key & 0xFFFFu) - 32768, iy = (int)(key >> 16) - 32768;
        // exact integer coordinate sums in units of 2^-K m
        int64_t Sx = (int64_t)sx + (int64_t)cnt * ((int64_t)ix * p.vox_L - (int64_t)vbias);
        int64_t Sy = (int64_t)sy + (int64_t)cnt * ((int64_t)iy * p.vox_L - (int64_t)vbias);
        double dc = (double)cnt;
        double cx = ((double)Sx * inv_scale) / dc;  // == (fp64 sum of x) / count of the spec
        double cy = ((double)Sy * inv_scale) / dc;
        out[out_base + c] = make_float4((float)cx, (float)cy, 0.0f, (float)((double)isum / dc));
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) L.misc[7] = out_base + ncell;
    __syncthreads();
    RPL_MARK(7)
  }
  if (p.dbg && threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) p.dbg[8 * b + i] = tacc[i];
  }
#undef RPL_MARK

  if (flags) atomicOr(&L.misc[1], flags);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t total = L.misc[7];
    n_points[b] = min(total, out_stride);
    if (status) status[b] = L.misc[1] | ((total > out_stride) ? RPLGPU_SCAN_OUT_TRUNCATED : 0u);
  }
}

// ------------------------------------------------------------------------------
// Divisor validation: div_by(a, d, RN(1/d)) must equal the IEEE quotient a / d for
// every fp32 `a` with biased exponent in [e_lo, e_hi] (both signs); +-0 must give a zero.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_validate_div(float d, float rd, uint32_t e_lo,
                                                      uint32_t e_hi, uint32_t *mismatches) {
  const uint64_t per_exp = 1ull << 23;
  const uint64_t total = (uint64_t)(e_hi - e_lo + 1) * per_exp;
  uint32_t bad = 0;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t bits = ((uint32_t)(e_lo + (uint32_t)(t >> 23)) << 23) | (uint32_t)(t & (per_exp - 1));
    float a = __uint_as_float(bits);
    float na = __uint_as_float(bits | 0x80000000u);
    bad += (__float_as_uint(div_by(a, d, rd)) != __float_as_uint(a / d));
    bad += (__float_as_uint(div_by(na, d, rd)) != __float_as_uint(na / d));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    bad += (div_by(0.0f, d, rd) != 0.0f);
    bad += (div_by(-0.0f, d, rd) != 0.0f);
  }
  if (bad) atomicAdd(mismatches, bad);
}

hipError_t launch_validate_div(hipStream_t s, float d, float rd, uint32_t e_lo, uint32_t e_hi,
                               uint32_t *d_mismatches) {
  hipLaunchKernelGGL(k_validate_div, dim3(256 * 16), dim3(256), 0, s, d, rd, e_lo, e_hi,
                     d_mismatches);
  return hipGetLastError();
}

hipError_t launch_cloud_voxel(hipStream_t s, const void *nodes, uint32_t n_stride,
                              const uint32_t *n_per_scan, uint32_t B, const KParams &p,
                              const Tables &T, float *xyzi, uint32_t out_stride,
                              uint32_t *n_points, uint32_t *status) {
  if (B == 0) return hipSuccess;
  if (p.fast_div) {
    hipLaunchKernelGGL(k_cloud_voxel<true>, dim3(B), dim3(kBlock), 0, s, (const uint2 *)nodes,
                       n_stride, n_per_scan, p, T, (float4 *)xyzi, out_stride, n_points, status);
  } else {
    hipLaunchKernelGGL(k_cloud_voxel<false>, dim3(B), dim3(kBlock), 0, s, (const uint2 *)nodes,
                       n_stride, n_per_scan, p, T, (float4 *)xyzi, out_stride, n_points, status);
  }
  return hipGetLastError();
}

}  // namespace rpl
