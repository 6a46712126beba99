// rpl_launch.hpp — host-callable launchers of the gfx950 kernels (rpl_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rplgpu.h"
#include "rplgpu_msg.h"
#include "rplgpu_comm.h"
#include "rpl_device.hpp"
#include "rpl_msg.hpp"

namespace rplmsg {
struct Prefix;
}

namespace rpl {

// `need_sort`: B + 1 words of scratch (how many / which scans the second, sorting kernel has to redo)
// status bit of a scan that launch_ascend(..., defer_sort) left for launch_ascend_sort (internal: the
// sorting kernel rewrites the status word without it)
constexpr uint32_t kAscendUnsorted = 0x80000000u;
hipError_t launch_ascend(hipStream_t s, void *nodes, uint32_t n_stride, const uint32_t *n_per_scan,
                         uint32_t B, uint32_t *status, uint32_t *need_sort, bool defer_sort = false,
                         uint32_t *sort_stat = nullptr);
hipError_t launch_ascend_sort(hipStream_t s, void *nodes, uint32_t n_stride, const uint32_t *n_per_scan,
                              uint32_t B, uint32_t *status, uint32_t *need_sort,
                              uint32_t *sort_stat = nullptr);
// publish_scan Mode A (rpl_laserscan.hip); `fast`: the mul+2*FMA divides were validated
// A single scan straight into its serialised sensor_msgs/LaserScan (Mode A, validated fast
// divides): the kernel writes the message prefix, patches stamp / scalars / array lengths and
// flushes its bins to the two arrays' places inside the message — no ranges / intensities in HBM,
// no second kernel.  msg: device view of a (pinned, 4-byte aligned) buffer that holds the worst
// case; msg_len: device word receiving the length (0: nothing published).
struct LsMsgOut {
  rplmsg::Prefix P;
  uint32_t *msg;
  uint32_t *msg_len;
  int32_t sec;
  uint32_t nanosec;
  double scan_duration;
};
hipError_t launch_laserscan_a(hipStream_t s, const void *nodes, uint32_t n_stride,
                              const uint32_t *n_per_scan, uint32_t B, const KParams &p,
                              const Tables &T, const float *inc_table, const float *rinc_table,
                              bool fast, float *ranges, float *intens, uint32_t *beam_count,
                              uint32_t n_given = 0xFFFFFFFFu, const LsMsgOut *msg_out = nullptr);
hipError_t launch_validate_idx(hipStream_t s, const Tables &T, const float *inc_table,
                               const float *rinc_table, uint32_t max_count,
                               uint32_t *d_mismatches);
// publish_scan Mode B (rpl_kernels.hip)
hipError_t launch_laserscan_raw(hipStream_t s, const void *nodes, uint32_t n_stride,
                                const uint32_t *n_per_scan, uint32_t B, const KParams &p,
                                float *ranges, float *intens, uint32_t *beam_count);
// status of a work item the voxel kernel's ROR instance left to the two kernels (internal: the listed launches
// overwrite it; the single-scan entry points look for it instead of launching them blind)
constexpr uint32_t kRorListedBit = 0x80000000u;
// `keepmask` (optional): one bit per sample from launch_ror_mask, `mask_stride` words per scan.
hipError_t launch_cloud(hipStream_t s, const void *nodes, uint32_t n_stride,
                        const uint32_t *n_per_scan, uint32_t B, const KParams &p, const Tables &T,
                        bool voxel, const uint32_t *keepmask, uint32_t mask_stride, float *xyzi,
                        uint32_t out_stride, uint32_t *n_points, uint32_t *status,
                        const float *motion = nullptr);  // E6 de-skew (plain cloud only)
hipError_t launch_cloud_voxel(hipStream_t s, const void *nodes, uint32_t n_stride,
                              const uint32_t *n_per_scan, uint32_t B, const KParams &p,
                              const Tables &T, const uint32_t *keepmask, uint32_t mask_stride,
                              float *xyzi, uint32_t out_stride, uint32_t *n_points,
                              uint32_t *status, float *arena = nullptr,
                              unsigned long long arena_capacity = 0,
                              unsigned long long *arena_cursor = nullptr,
                              unsigned long long *scan_start = nullptr,
                              // E8: `group` consecutive scans share one grid; per-scan motion
                              // (vx, vy, wz, dt) and planar pose (r00 r01 tx r10 r11 ty), optional
                              uint32_t group = 1, const float *motion = nullptr,
                              const float *pose2d = nullptr,
                              // the arena holds 12-byte points (x, y, intensity): the exchange payload
                              bool arena_xyi = false,
                              // E5 (round 6): 1 = inside the pass (arena launches, validated divides; items it
                              // cannot settle go on T.redo), 2 = the items of T.redo with `keepmask`
                              int ror_mode = 0);
// record stores k_cloud_voxel needs: one per resident workgroup (two per CU) ...
uint32_t voxel_max_workgroups(uint32_t n_cu);
// ... of this many 16-byte entries for work items of `group` scans of `n_stride` samples: every
// sample can end a run record, and every block of 128 samples adds one marker entry
// (two per block: the noisy-batch instance aggregates a block in two classes, each behind its own marker)
// (round 6: the ROR instance's blocks own 124 samples, and a sample E5 settles late brings its own marker:
// at most 2 x (256 + 8) entries per scan)
inline uint64_t voxel_store_need(uint32_t group, uint32_t n_stride) {
  const uint64_t s = n_stride < kMaxN ? n_stride : kMaxN;
  return (uint64_t)(group ? group : 1u) * (s + 2u * ((s + 123u) / 124u) + 1u + 528u);
}
// `listed`: the scans of the work items on T.redo (count word, then item numbers; an item = `group`
// consecutive scans) instead of all B scans: a persistent grid that reads the count on the device
hipError_t launch_ror_mask(hipStream_t s, const void *nodes, uint32_t n_stride,
                           const uint32_t *n_per_scan, uint32_t B, const KParams &p,
                           const Tables &T, uint32_t *mask, uint32_t mask_stride,
                           bool listed = false, uint32_t group = 1);
hipError_t launch_validate_div(hipStream_t s, float d, float rd, uint32_t e_lo, uint32_t e_hi,
                               uint32_t *d_mismatches);
hipError_t launch_pack(hipStream_t s, const float *xyzi, uint32_t out_stride,
                       const uint32_t *n_points, uint32_t B, float *packed, uint64_t *offsets);

// E7: the binned LaserScan as a cloud (rpl_project.hip)
hipError_t launch_laserscan_to_cloud(hipStream_t s, const float *ranges, const float *intens,
                                     uint32_t n_stride, const uint32_t *beam_count, uint32_t B,
                                     const KParams &p, float *xyzi, uint32_t out_stride,
                                     uint32_t *n_points, uint32_t *status);

// multi-GPU exchange, device side (rpl_comm.hip)
hipError_t launch_signal(hipStream_t s, uint32_t *flag, uint32_t seq);
hipError_t launch_stage_in(hipStream_t s, const void *src, void *dst, uint32_t n_words2);  // 8-byte words
hipError_t launch_pack_meta(hipStream_t s, const unsigned long long *cursor,
                            const unsigned long long *scan_start, const uint32_t *n_points,
                            uint32_t B, unsigned long long slot_points, uint32_t max_scans,
                            uint32_t *meta);
hipError_t launch_unpack_gathered(hipStream_t s, const float *points_all,
                                  unsigned long long slot_points, const uint32_t *meta_all,
                                  uint32_t meta_words, uint32_t world, uint32_t max_scans,
                                  float *packed, unsigned long long *total,
                                  unsigned long long *scan_start_all, uint32_t *n_points_all,
                                  uint32_t *status, uint32_t n_cu, bool xyi = false);
hipError_t launch_pack_xyi(hipStream_t s, const float *arena, const unsigned long long *cursor,
                           unsigned long long slot_points, float *slot, uint32_t n_cu);

// decode stage (rpl_decode.hip)
hipError_t launch_decode(hipStream_t s, int ans, const uint8_t *bytes, uint64_t stream_stride,
                         const uint32_t *frame_off, const uint8_t *gap, const uint32_t *n_frames,
                         uint32_t max_frames, uint32_t B, uint32_t sample_duration_us,
                         const int32_t *state_in, int32_t *state_out, void *nodes,
                         uint32_t node_stride, uint32_t *n_nodes, uint32_t *reset_at,
                         uint32_t reset_stride, uint32_t *n_reset, uint32_t *n_errors,
                         uint32_t *status, uint32_t *sync_at = nullptr, uint32_t sync_stride = 0,
                         uint32_t *n_sync = nullptr, const uint32_t *only = nullptr,
                         // capsule streams that fit: the LDS-staged instance (rpl_decode.hip); with frame
                         // offsets it needs B words of scratch for the streams it leaves to the plain kernel
                         bool staged_ok = true, uint32_t *stage_todo = nullptr);
bool decode_fusable(int ans);
hipError_t launch_decode_fused(hipStream_t s, int ans, const uint8_t *bytes, uint64_t stream_stride,
                               const uint32_t *frame_off, const uint8_t *gap,
                               const uint32_t *n_frames, uint32_t max_frames, uint32_t B,
                               uint32_t sample_duration_us, const int32_t *state_in,
                               int32_t *state_out, uint32_t *n_errors, uint32_t *status,
                               uint32_t max_count, void *batch, uint32_t n_stride,
                               uint32_t scan_cap, uint32_t *n_per_scan, uint32_t *n_scans,
                               uint32_t *todo, bool staged_ok = true);
hipError_t launch_segment(hipStream_t s, const void *nodes, uint32_t node_stride,
                          const uint32_t *n_nodes, const uint32_t *reset_at, uint32_t reset_stride,
                          const uint32_t *n_reset, uint32_t B, uint32_t max_count, void *out_nodes,
                          uint32_t out_stride, uint32_t *scan_off, uint32_t scan_cap,
                          uint32_t *n_scans, uint32_t *status, const uint32_t *sync_at = nullptr,
                          uint32_t sync_stride = 0, const uint32_t *n_sync = nullptr);
// decoder's sync list + node stream -> completed scans in batch slots b*scan_cap + s
hipError_t launch_assemble(hipStream_t s, const void *nodes, uint32_t node_stride,
                           const uint32_t *n_nodes, const uint32_t *sync_at, uint32_t sync_stride,
                           const uint32_t *n_sync, const uint32_t *reset_at, uint32_t reset_stride,
                           const uint32_t *n_reset, uint32_t B, uint32_t max_count, void *batch,
                           uint32_t n_stride, uint32_t scan_cap, uint32_t *n_per_scan,
                           uint32_t *n_scans, uint32_t *status, const uint32_t *only = nullptr,
                           // the scan open at a call boundary (rplgpu_decode_scans_carry_dev), or nulls
                           const void *carry_in = nullptr, void *carry_out = nullptr,
                           const uint32_t *carry_len_in = nullptr, uint32_t *carry_len_out = nullptr,
                           uint32_t carry_stride = 0);
uint32_t decode_sync_stride();
hipError_t launch_scans_to_batch(hipStream_t s, const void *seg_nodes, uint32_t seg_stride,
                                 const uint32_t *scan_off, uint32_t scan_cap,
                                 const uint32_t *n_scans, uint32_t B, uint32_t *scan_base,
                                 void *batch, uint32_t n_stride, uint32_t max_scans,
                                 uint32_t *n_per_scan);
uint32_t decode_max_frames(int ans);
uint32_t decode_staged_frames(int ans);  // largest max_frames of a back-to-back call the LDS-staged decoder takes

// serialised-message assembly (rpl_msg.hip)
hipError_t launch_msg_laserscan(hipStream_t s, const float *ranges, const float *intens,
                                uint32_t n_stride, const uint32_t *beam_count, uint32_t B,
                                int scan_processing, const rplgpu_stamp_t *stamps,
                                const double *scan_duration, const rplmsg::Prefix &P,
                                uint8_t *msgs, uint32_t msg_stride, uint32_t *msg_len,
                                uint32_t *status);
hipError_t launch_msg_cloud(hipStream_t s, const float *xyzi, uint32_t out_stride,
                            uint32_t max_points, const unsigned long long *scan_start,
                            const uint32_t *n_points, uint32_t B, const rplgpu_stamp_t *stamps,
                            const rplmsg::Prefix &P, uint8_t *msgs, uint32_t msg_stride,
                            uint32_t *msg_len, uint32_t *status);

hipError_t launch_msg_fused(hipStream_t s, const float *arena,
                            const unsigned long long *total_points,
                            unsigned long long arena_capacity, rplgpu_stamp_t stamp,
                            const rplmsg::Prefix &P, uint8_t *msg, unsigned long long msg_capacity,
                            unsigned long long *msg_len, uint32_t *status);
// several sensors into one frame (rpl_fuse.hip)
hipError_t launch_transform_clouds(hipStream_t s, float *xyzi, uint32_t out_stride,
                                   uint32_t max_points, const unsigned long long *scan_start,
                                   const uint32_t *n_points, uint32_t B, const float *pose);

}  // namespace rpl
