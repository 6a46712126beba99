// rpl_device.hpp — shared device-side building blocks for the gfx950 scan kernels.
//
// Geometry used by every "resident" kernel: one 1024-thread workgroup (16 wave64)
// owns one scan and keeps it in registers: wave w, iteration j, lane l holds sample
//   i = (j*16 + w)*64 + l          (chunk c = j*16 + w covers 64 consecutive samples)
// so each wave-level load is one contiguous 512-byte segment and consecutive waves
// touch consecutive segments (coalesced HBM streaming of the packed 8-byte nodes,
// reference layout src/sdk/include/sl_lidar_cmd.h:272-278).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rplgpu.h"

namespace rpl {

constexpr int kBlock = 1024;
constexpr int kWaves = kBlock / 64;
constexpr int kIters = 32;                       // 32768 samples / 1024 threads
constexpr uint32_t kMaxN = 32768;                // == RPLGPU_MAX_SAMPLES_PER_SCAN
constexpr int kChunks = kIters * kWaves;         // 512 chunks of 64 samples
constexpr double kTwoPi = 6.283185307179586476925286766559;  // 2.0 * M_PI

// Host-built lookup tables indexed by angle_z_q14 (bit-exact by construction: the
// host evaluates the reference's own float/double expressions once per u16 value).
struct Tables {
  const float *angle;      // angle_rad of src/rplidar_node.cpp:588-599
  const float *angle_inv;  // after the invert rule of :646-651
  const float2 *cs;        // (float)cos((double)angle), (float)sin((double)angle)
  const float2 *cs_inv;    // same for the inverted angle
  const double *rcp;       // rcp[c] = RN(1.0 / c), c = 1..32768 (voxel centroids), rcp[0] = 0
  uint32_t *work_ctr;      // dynamic scan queue of k_cloud_voxel (one word, cleared by every launch)
  uint32_t n_cu;           // compute units of the handle's device (persistent-workgroup grids)
  void *voxel_store;       // k_cloud_voxel's record stores: voxel_store_wgs x voxel_store_recs x 16 B
  uint32_t voxel_store_wgs;
  uint32_t voxel_store_recs;  // records per workgroup (>= kMaxN; group x kMaxN for an E8 group)
  unsigned long long *voxel_stats;  // [0] queue entries, [1] work items of the launch (device; may be null)
  unsigned long long *voxel_stats_host;  // pinned copy of voxel_stats, refreshed behind every launch (or null)
  int32_t voxel_split;     // host decision from the previous launch's statistics: the noisy-batch instance
  const float *scan_t0;       // per scan: time of its first sample relative to the fused instant (E6), or null
  uint32_t *redo;             // E5 inside the voxel kernel: [0] number of work items it left to the two-kernel
                              // path, [4 ...] their numbers (cleared in front of the launch); or null
};

struct KParams {
  int32_t is_new_protocol;
  int32_t inverted;
  int32_t scan_processing;
  int32_t clip_enable;
  uint32_t q_min;
  float range_min;
  float range_max;
  float voxel_leaf;
  float ror_r2;
  uint32_t ror_k;
  // voxel fixed point (host-derived from voxel_leaf): coordinates are accumulated in
  // units of 2^-K m relative to the integer cell origin ix*vox_L, 2^-K = ulp(leaf)
  double vox_scale;   // 2^K, K = 23 - ilogb(leaf)  (28 for 0.05f)
  float vox_scale_f;  // same, fp32
  int32_t vox_L;      // leaf * 2^K: the leaf's 24-bit significand, exact
  int32_t vox_bias;   // 0 since round 3 (offsets are summed as wrapping two's complement)
  float inv_leaf;     // RN(1 / voxel_leaf)
  float leaf_s;       // voxel_leaf * 2^K (= vox_L) and RN(1 / voxel_leaf) * 2^-K: the cell divide on
  float inv_leaf_s;   // coordinates in units of 2^-K m (both exact scalings)
  // E1 keep mask as an integer interval on dist_mm_q2 (host-derived, see make_keep_interval in
  // rplgpu_api.hip): keep <=> (dist_q2 - d_lo) <= d_span (unsigned) && quality >= q_min.
  // dist_m = RN(u32->f32(d)) / 4000 is monotone in d, so the float compares against range_min /
  // range_max select exactly one interval of d; d_lo >= 1 also covers the d != 0 test of :584.
  uint32_t d_lo;
  uint32_t d_span;
  int32_t cell_range_safe;  // 1: host proved |cell index| < 32767 for every kept sample
  unsigned long long *dbg;  // optional per-block phase cycle counters (developer aid), or null
  uint32_t *cell_keys;      // optional: one word per output cell, (iy + 32768) << 16 | (ix + 32768), at
                            // the cell's index in the output buffer (rplgpu_set_cell_key_output)
  int32_t fast_div;   // 1: the mul+2*FMA divides by 4000 and by leaf were validated on this
                      //    device to be bit-identical to the IEEE divide (see k_validate_div)
  int32_t fast_d4000; // 1: ... the divide by 4000 alone (kernels that do not divide by the leaf)
};

// ---- packed node decode (uint2 = the 8 raw bytes, little endian) ----------------
__device__ __forceinline__ uint32_t nd_q14(uint2 v) { return v.x & 0xFFFFu; }
__device__ __forceinline__ uint32_t nd_dist(uint2 v) {
  return __builtin_amdgcn_alignbit(v.y, v.x, 16);  // unaligned u32 at byte offset 2
}
__device__ __forceinline__ uint32_t nd_quality(uint2 v) { return (v.y >> 16) & 0xFFu; }

// dist_m = dist_mm_q2 / 4000.0f  (src/rplidar_node.cpp:590): u32->f32 RNE, IEEE divide.
__device__ __forceinline__ float nd_dist_m(uint32_t dist_q2) {
  return __uint2float_rn(dist_q2) / 4000.0f;
}
// intensity (src/rplidar_node.cpp:591-592)
__device__ __forceinline__ float nd_intensity(uint32_t quality, int is_new_protocol) {
  return is_new_protocol ? (float)quality : (float)(quality >> 2);
}
// keep mask: :584 plus the optional E1 clip, as one unsigned interval test on dist_mm_q2
// (KParams::d_lo / d_span; quality threshold only when clipping)
__device__ __forceinline__ bool nd_keep(uint32_t dist_q2, uint32_t quality, const KParams &p) {
  bool k = (dist_q2 - p.d_lo) <= p.d_span;
  if (p.clip_enable) k = k && (quality >= p.q_min);
  return k;
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ uint64_t lanemask_lt() {
  return (1ull << lane_id()) - 1ull;
}

// Inclusive wave scan (64 lanes) with shuffles (ds_bpermute: ~24 issue cycles a step).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if ((int)lane_id() >= d) v += t;
  }
  return v;
}
// The same scan with DPP row shifts / row broadcasts: six VALU instructions, no LDS crossbar.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp_add_u32(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan_fast(uint32_t v) {
  v = dpp_add_u32<0x111, 0xF>(v);  // row_shr:1
  v = dpp_add_u32<0x112, 0xF>(v);  // row_shr:2
  v = dpp_add_u32<0x114, 0xF>(v);  // row_shr:4
  v = dpp_add_u32<0x118, 0xF>(v);  // row_shr:8
  v = dpp_add_u32<0x142, 0xA>(v);  // row_bcast:15 -> rows 1,3
  v = dpp_add_u32<0x143, 0xC>(v);  // row_bcast:31 -> rows 2,3
  return v;
}

// Wave-wide min / max with the same DPP pattern (lanes without a source keep their own value);
// the result is valid in lane 63.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp_min_u32(uint32_t v) {
  return min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xF, false));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp_max_u32(uint32_t v) {
  return max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xF, false));
}
__device__ __forceinline__ uint32_t wave_min_lane63(uint32_t v) {
  v = dpp_min_u32<0x111, 0xF>(v);
  v = dpp_min_u32<0x112, 0xF>(v);
  v = dpp_min_u32<0x114, 0xF>(v);
  v = dpp_min_u32<0x118, 0xF>(v);
  v = dpp_min_u32<0x142, 0xA>(v);
  v = dpp_min_u32<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ uint32_t wave_max_lane63(uint32_t v) {
  v = dpp_max_u32<0x111, 0xF>(v);
  v = dpp_max_u32<0x112, 0xF>(v);
  v = dpp_max_u32<0x114, 0xF>(v);
  v = dpp_max_u32<0x118, 0xF>(v);
  v = dpp_max_u32<0x142, 0xA>(v);
  v = dpp_max_u32<0x143, 0xC>(v);
  return v;
}

// Exclusive block scan of one value per thread (1024 threads). `tmp` = 17 u32 in LDS.
// Returns the exclusive prefix; *total receives the block sum. Contains barriers.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *tmp, uint32_t *total) {
  uint32_t inc = wave_incl_scan_fast(v);
  if (lane_id() == 63) tmp[wave_id()] = inc;
  __syncthreads();
  if (threadIdx.x < 64) {
    uint32_t w = (threadIdx.x < kWaves) ? tmp[threadIdx.x] : 0u;
    uint32_t ws = wave_incl_scan_fast(w);
    if (threadIdx.x < kWaves) tmp[threadIdx.x] = ws - w;  // exclusive wave base
    if (threadIdx.x == kWaves - 1) tmp[kWaves] = ws;
  }
  __syncthreads();
  uint32_t res = tmp[wave_id()] + inc - v;
  *total = tmp[kWaves];
  return res;
}

// In-LDS bitonic sort of N (power of two) u32 keys, ascending, by the whole block.
__device__ __forceinline__ void block_bitonic_sort(uint32_t *keys, uint32_t N) {
  for (uint32_t k = 2; k <= N; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (uint32_t t = threadIdx.x; t < (N >> 1); t += kBlock) {
        // t-th compare-exchange pair of this stage
        uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        uint32_t l = i | j;
        uint32_t a = keys[i], b = keys[l];
        bool up = (i & k) == 0;
        if ((a > b) == up) {
          keys[i] = b;
          keys[l] = a;
        }
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t v) {
  if (v <= 2) return 2;
  return 1u << (32 - __builtin_clz(v - 1));
}

}  // namespace rpl
