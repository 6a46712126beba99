// rpl_voxel.hip — k_cloud_voxel: raw scan -> clipped, voxel-downsampled PointCloud2
// (extensions E1 + E2 + E4 of SURVEY.md §8 a-ext) in ONE streaming pass over the
// packed 8-byte nodes.  One 1024-thread workgroup owns one scan.
//
// Data flow per wave-pass (64 consecutive samples, one per lane):
//   8-byte node  ->  keep mask, dist_m, (cos,sin) LUT  ->  x, y  ->  cell (iy, ix)
//   -> run detection along the lanes (a smooth ring visits a 5 cm cell ~10-200 times
//      in a row) -> DPP prefix sums give every run's partial sums -> only the LAST
//      lane of a run touches LDS: one hash-table probe + two u64 atomic adds.
// Cells live in an LDS open-addressing table (double hashing) keyed by the packed
// (iy, ix) pair.  Per cell two u64 words are accumulated with LDS integer atomics
// (order independent => run-to-run deterministic):
//   A = [ sum of x offsets : 40 | sum of integer intensities : 24 ]
//   B = [ sum of y offsets : 40 | point count               : 24 ]
// offset = (x - ix*leaf) * 2^K + bias with 2^-K = ulp(leaf) (K = 28 for 5 cm).  One
// fp32 FMA yields x - ix*leaf EXACTLY whenever x is a multiple of 2^-K (|x| >= 3 cm at
// K = 28), so the integer sums are exact and sum/count reproduces the spec's fp64
// running sum bit for bit; closer to the axes the per-point error is <= 2^-(K+1) m
// (1.9e-9 m), far below the 1e-6 m bar.
// Output order (iy, ix): cells are ranked by a counting sort over rows plus an
// in-row rank (keys are unique), no comparison sort on the common path.
// A scan with more cells than the table holds is processed in key bands (the key
// space is bisected until a band fits), each band re-streaming the scan from L2.
#include "rpl_device.hpp"
#include "rpl_launch.hpp"

namespace rpl {

constexpr uint32_t kCellCap = 6144;    // table slots: 24 KiB keys + 96 KiB sums
constexpr uint32_t kCellLimit = 5600;  // bisect the band beyond this many cells (load 0.91)
constexpr uint32_t kRowCap = 2048;     // rows handled by the counting-sort ranker
constexpr uint32_t kAuxWords = 8192;   // 32 KiB: row tables + buckets, or bitonic buffer
constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t cell_hash(uint32_t key) {
  uint32_t h = key * 0x9E3779B1u;
  return (uint32_t)(((uint64_t)h * kCellCap) >> 32);
}
// probe step: odd and == 1 (mod 3)  =>  coprime with 6144 = 2^11 * 3
__device__ __forceinline__ uint32_t cell_step(uint32_t key) {
  return 1u + 6u * ((key * 0x85EBCA6Bu) >> 22);
}

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
  uint64_t A[kCellCap];
  uint64_t B[kCellCap];
  uint32_t key[kCellCap];
  uint32_t aux[kAuxWords];
  uint32_t band_lo[34], band_hi[34];
  uint32_t misc[16];  // 0 ncell, 1 status, 2 overflow, 3 sp, 4 rowmin, 5 rowmax, 6 cursor, 7 out_base
  uint32_t tmp[32];
};

// Accumulate one wave-pass. `kept` lanes carry (key, qx, qy, inten).
__device__ __forceinline__ void voxel_wave_pass(VoxelLds &L, bool kept, uint32_t key, uint32_t qx,
                                                uint32_t qy, uint32_t inten) {
  const uint64_t keptmask = __ballot(kept);
  if (keptmask == 0ull) return;
  if (!kept) {
    key = kEmptyKey;
    qx = qy = inten = 0u;
  }
  const uint32_t lane = lane_id();
  // previous lane's key (wave_shr:1); lane 0 sees "no key"
  uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)kEmptyKey, (int)key, 0x138, 0xF, 0xF,
                                                        false);
  const bool head = kept && (key != prev);
  const uint64_t headmask = __ballot(head);
  const uint64_t cont = keptmask & ~headmask;         // lanes continuing the previous lane's run
  const uint64_t tailmask = keptmask & ~(cont >> 1);  // kept lanes whose successor starts anew
  const bool tail = (tailmask >> lane) & 1ull;

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
  if (!tail) return;

  const uint64_t sx = (uint64_t)(Px - Ex), sy = (uint64_t)(Py - Ey);
  const uint32_t sc = Pc - Ec;
  uint32_t slot = cell_hash(key);
  const uint32_t step = cell_step(key);
  bool placed = false;
  for (uint32_t probe = 0; probe < kCellCap; ++probe) {
    uint32_t old = atomicCAS(&L.key[slot], kEmptyKey, key);
    if (old == kEmptyKey) {
      if (atomicAdd(&L.misc[0], 1u) >= kCellLimit) L.misc[2] = 1u;  // band too dense: bisect
      placed = true;
      break;
    }
    if (old == key) {
      placed = true;
      break;
    }
    slot += step;
    if (slot >= kCellCap) slot -= kCellCap;
  }
  if (!placed) {
    L.misc[2] = 1u;
    return;
  }
  atomicAdd((unsigned long long *)&L.A[slot], (unsigned long long)((sx << 24) | (sc & 0xFFFFu)));
  atomicAdd((unsigned long long *)&L.B[slot], (unsigned long long)((sy << 24) | (sc >> 16)));
}

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
  const float vscale = p.vox_scale_f;
  const int vbias = p.vox_bias;
  const float2 *cs = p.inverted ? T.cs_inv : T.cs;
  uint32_t flags = 0;

  while (true) {
    // ---- pop a key band -----------------------------------------------------------
    const uint32_t sp = L.misc[3];
    if (sp == 0) break;
    const uint32_t klo = L.band_lo[sp - 1], khi = L.band_hi[sp - 1];
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < kCellCap; t += kBlock) {
      L.key[t] = kEmptyKey;
      L.A[t] = 0ull;
      L.B[t] = 0ull;
    }
    if (threadIdx.x == 0) {
      L.misc[0] = 0u;
      L.misc[2] = 0u;
      L.misc[3] = sp - 1;
      L.misc[4] = 0xFFFFFFFFu;
      L.misc[5] = 0u;
      L.misc[6] = 0u;
    }
    __syncthreads();

    // ---- stream the scan: 4 x 8 B per thread in flight ------------------------------
    constexpr int UNR = 4;
    for (uint32_t base = 0; base < n; base += kBlock * UNR) {
      uint2 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        uint32_t i = base + (uint32_t)u * kBlock + threadIdx.x;
        v[u] = (i < n) ? scan[i] : make_uint2(0u, 0u);
      }
      if (*(volatile uint32_t *)&L.misc[2]) break;  // band already known to be too dense
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        uint32_t d = nd_dist(v[u]);
        float dm = nd_dist_m(d);
        uint32_t qual = nd_quality(v[u]);
        bool kept = nd_keep(d, qual, dm, p);
        uint32_t key = kEmptyKey, qx = 0, qy = 0;
        if (kept) {
          float2 c = cs[nd_q14(v[u])];
          float x = dm * c.x, y = dm * c.y;                   // E2
          float fx = floorf(x / leaf), fy = floorf(y / leaf);  // E4 cell, IEEE fp32 divide
          if (!(fabsf(fx) < 32767.0f) || !(fabsf(fy) < 32767.0f)) {
            flags |= RPLGPU_SCAN_CELL_RANGE;
            kept = false;
          } else {
            key = ((uint32_t)((int)fy + 32768) << 16) | (uint32_t)((int)fx + 32768);
            if (key < klo || key > khi) {
              kept = false;
            } else {
              int ox = (int)rintf(fmaf(-fx, leaf, x) * vscale) + vbias;
              int oy = (int)rintf(fmaf(-fy, leaf, y) * vscale) + vbias;
              qx = (uint32_t)min(max(ox, 0), 0x1FFFFFF);
              qy = (uint32_t)min(max(oy, 0), 0x1FFFFFF);
            }
          }
        }
        uint32_t inten = p.is_new_protocol ? qual : (qual >> 2);
        voxel_wave_pass(L, kept, key, qx, qy, inten);
      }
    }
    __syncthreads();

    if (L.misc[2]) {  // too many cells in this band: bisect the key range and retry
      if (threadIdx.x == 0) {
        uint32_t s = L.misc[3];
        if (klo == khi || s + 2 > 33) {
          L.misc[1] |= RPLGPU_SCAN_TABLE_FULL;  // cannot happen: one key is one cell
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
      continue;
    }

    // ---- rank the occupied cells in (iy, ix) order and emit them ------------------------
    const uint32_t ncell = L.misc[0];
    const uint32_t out_base = L.misc[7];
    if (ncell > 0) {
      uint32_t rmin = 0xFFFFFFFFu, rmax = 0u;
      for (uint32_t t = threadIdx.x; t < kCellCap; t += kBlock) {
        uint32_t k = L.key[t];
        if (k != kEmptyKey) {
          rmin = min(rmin, k >> 16);
          rmax = max(rmax, k >> 16);
        }
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        rmin = min(rmin, (uint32_t)__shfl_xor((int)rmin, d, 64));
        rmax = max(rmax, (uint32_t)__shfl_xor((int)rmax, d, 64));
      }
      if (lane_id() == 0) {
        atomicMin(&L.misc[4], rmin);
        atomicMax(&L.misc[5], rmax);
      }
      __syncthreads();
      rmin = L.misc[4];
      const uint32_t nrows = L.misc[5] - rmin + 1u;
      const double inv_scale = 1.0 / p.vox_scale;  // exact power of two

      auto emit = [&](uint32_t slot, uint32_t rank) {
        if (out_base + rank >= out_stride) return;
        uint32_t key = L.key[slot];
        uint64_t A = L.A[slot], B = L.B[slot];
        int64_t cnt = (int64_t)(B & 0xFFFFFFull);
        int64_t isum = (int64_t)(A & 0xFFFFFFull);
        int ix = (int)(key & 0xFFFFu) - 32768, iy = (int)(key >> 16) - 32768;
        // exact integer coordinate sums in units of 2^-K m
        int64_t Sx = (int64_t)(A >> 24) + cnt * ((int64_t)ix * p.vox_L - (int64_t)vbias);
        int64_t Sy = (int64_t)(B >> 24) + cnt * ((int64_t)iy * p.vox_L - (int64_t)vbias);
        double dc = (double)cnt;
        double cx = ((double)Sx * inv_scale) / dc;  // == (fp64 sum of x) / count of the spec
        double cy = ((double)Sy * inv_scale) / dc;
        out[out_base + rank] = make_float4((float)cx, (float)cy, 0.0f, (float)((double)isum / dc));
      };

      if (nrows <= kRowCap) {
        // counting sort over rows + rank inside the row (keys are unique)
        uint32_t *rowstart = L.aux;            // [kRowCap]
        uint32_t *rowfill = L.aux + kRowCap;   // [kRowCap]
        uint16_t *bucket = reinterpret_cast<uint16_t *>(L.aux + 2 * kRowCap);  // [<= 8192]
        for (uint32_t t = threadIdx.x; t < kRowCap; t += kBlock) rowstart[t] = 0u;
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < kCellCap; t += kBlock) {
          uint32_t k = L.key[t];
          if (k != kEmptyKey) atomicAdd(&rowstart[(k >> 16) - rmin], 1u);
        }
        __syncthreads();
        {  // exclusive scan over kRowCap = 2 rows per thread
          uint32_t r0 = rowstart[2 * threadIdx.x], r1 = rowstart[2 * threadIdx.x + 1];
          uint32_t total;
          uint32_t ex = block_excl_scan(r0 + r1, L.tmp, &total);
          rowstart[2 * threadIdx.x] = ex;
          rowstart[2 * threadIdx.x + 1] = ex + r0;
          rowfill[2 * threadIdx.x] = ex;
          rowfill[2 * threadIdx.x + 1] = ex + r0;
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < kCellCap; t += kBlock) {
          uint32_t k = L.key[t];
          if (k != kEmptyKey) bucket[atomicAdd(&rowfill[(k >> 16) - rmin], 1u)] = (uint16_t)(k & 0xFFFFu);
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < kCellCap; t += kBlock) {
          uint32_t k = L.key[t];
          if (k != kEmptyKey) {
            uint32_t row = (k >> 16) - rmin, ixb = k & 0xFFFFu;
            uint32_t s0 = rowstart[row], s1 = rowfill[row];
            uint32_t rank = s0;
            for (uint32_t m = s0; m < s1; ++m) rank += (bucket[m] < ixb);
            emit(t, rank);
          }
        }
      } else {
        // generic path: bitonic sort of the occupied keys, then look each one up
        uint32_t *sortbuf = L.aux;
        for (uint32_t t = threadIdx.x; t < kCellCap; t += kBlock) {
          uint32_t k = L.key[t];
          if (k != kEmptyKey) sortbuf[atomicAdd(&L.misc[6], 1u)] = k;
        }
        __syncthreads();
        const uint32_t N = next_pow2(ncell);
        for (uint32_t t = ncell + threadIdx.x; t < N; t += kBlock) sortbuf[t] = kEmptyKey;
        block_bitonic_sort(sortbuf, N);
        for (uint32_t r = threadIdx.x; r < ncell; r += kBlock) {
          uint32_t key = sortbuf[r];
          uint32_t slot = cell_hash(key);
          const uint32_t step = cell_step(key);
          while (L.key[slot] != key) {
            slot += step;
            if (slot >= kCellCap) slot -= kCellCap;
          }
          emit(slot, r);
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) L.misc[7] = out_base + ncell;
    __syncthreads();
  }

  if (flags) atomicOr(&L.misc[1], flags);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t total = L.misc[7];
    n_points[b] = min(total, out_stride);
    if (status) status[b] = L.misc[1] | ((total > out_stride) ? RPLGPU_SCAN_OUT_TRUNCATED : 0u);
  }
}

hipError_t launch_cloud_voxel(hipStream_t s, const void *nodes, uint32_t n_stride,
                              const uint32_t *n_per_scan, uint32_t B, const KParams &p,
                              const Tables &T, float *xyzi, uint32_t out_stride,
                              uint32_t *n_points, uint32_t *status) {
  if (B == 0) return hipSuccess;
  hipLaunchKernelGGL(k_cloud_voxel, dim3(B), dim3(kBlock), 0, s, (const uint2 *)nodes, n_stride,
                     n_per_scan, p, T, (float4 *)xyzi, out_stride, n_points, status);
  return hipGetLastError();
}

}  // namespace rpl
