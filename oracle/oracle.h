/*
 * oracle.h — CPU oracle for the RPLIDAR scan-preprocessing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load
 * liboracle.so / oracle/_ref/ libraries.  The product (librplgpu.so) never links,
 * loads or falls back to anything in this directory.
 *
 * Parity status
 *   - ascendScanData (orc_ascend)      : PINNED — checked bit-for-bit against the
 *       reference SDK itself compiled from /root/reference/src/sdk
 *       (oracle/_ref/libslref.so, see oracle/Makefile) and against the KAT in
 *       SURVEY.md §8(c); golden vectors from the real SDK live in tests/golden/.
 *   - publish_scan (orc_publish_scan)  : PINNED — checked bit-for-bit against the
 *       genuine RPlidarNode::publish_scan compiled from /root/reference/src
 *       against stub ROS headers (oracle/_ref/libnoderef.so); golden vectors
 *       generated from it live in tests/golden/.
 *   - extensions E1..E5 (clip, polar->XYZ, PointCloud2 layout, voxel grid,
 *       radius outlier removal) are NOT in the reference: "parity unpinned".
 *       Their oracle is written to the spec in SURVEY.md §8(a-ext).
 *
 * All citations are relative to /root/reference/.
 */
#ifndef RPL_ORACLE_H_
#define RPL_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Wire/record format of one sample: src/sdk/include/sl_lidar_cmd.h:272-278
 * (packed, 8 bytes, u32 at offset 2). */
typedef struct __attribute__((packed)) orc_node {
  uint16_t angle_z_q14; /* deg = v * 90 / 16384 */
  uint32_t dist_mm_q2;  /* mm = v / 4 ; 0 => invalid */
  uint8_t quality;
  uint8_t flag;
} orc_node_t;

/* Scalar knobs of the path.  Reference-pinned ones cite their origin. */
typedef struct orc_params {
  int32_t is_new_protocol; /* src/rplidar_node.cpp:577-581 */
  int32_t inverted;        /* params_.inverted, src/rplidar_node.cpp:646,676 */
  int32_t scan_processing; /* params_.scan_processing, src/rplidar_node.cpp:632 */
  int32_t clip_enable;     /* E1 (extension); 0 => reference behaviour */
  uint32_t q_min;          /* E1: keep quality >= q_min (raw quality byte) */
  float range_min;         /* E1: keep dist_m >= range_min */
  float range_max;         /* E1: keep dist_m <= range_max; also LaserScan.range_max
                              (cached_current_max_range_, src/rplidar_node.cpp:626) */
  float voxel_leaf;        /* E4: leaf size in metres (0.05) */
  float ror_radius;        /* E5: radius (0.10) */
  uint32_t ror_min_neighbors; /* E5: k (2) */
  int32_t ror_enable;      /* E5 on/off */
  int32_t voxel_enable;    /* E4 on/off */
} orc_params_t;

/* LaserScan metadata, POD stand-in for sensor_msgs::msg::LaserScan
 * (src/rplidar_node.cpp:618-627,634-638,665-669). */
typedef struct orc_scan_meta {
  float angle_min, angle_max, angle_increment, time_increment;
  float scan_time, range_min, range_max;
  uint32_t count;     /* ranges.size() == intensities.size() */
  int32_t published;  /* 0 when publish_scan returned before publishing */
} orc_scan_meta_t;

/* ---- a3: SDK ascendScanData restatement (src/sdk/src/sl_lidar_driver.cpp:102-184).
 * In place. Returns the SDK's sl_result (0 OK, 0x80008001 when all invalid). */
uint32_t orc_ascend(orc_node_t *nodes, size_t count);

/* ---- a4..a8: publish_scan restatement (src/rplidar_node.cpp:558-683).
 * ranges/intensities must hold n floats each. With clip_enable!=0 the E1 mask
 * replaces the plain validity mask of :584. */
void orc_publish_scan(const orc_node_t *nodes, size_t n, const orc_params_t *p,
                      double scan_duration, float *ranges, float *intensities,
                      orc_scan_meta_t *meta);

/* ---- a9: range-limit scalar (src/rplidar_node.cpp:391-396). */
float orc_effective_max_range(float max_distance_param, float hw_limit);

/* ---- a11: DummyLidarDriver::grab_scan_data generator
 * (src/lidar_driver_wrapper.cpp:441-471). `scan_index` = how many scans were
 * produced before this one (the reference keeps a static phase += 0.1f). Writes
 * 360 nodes. */
void orc_gen_dummy(uint32_t scan_index, orc_node_t *nodes /*[360]*/);

/* ---- a-ext E1+E2(+E5): raw nodes -> xyzi points in input order.
 * out must hold 4*n floats. Returns number of points. */
size_t orc_scan_to_cloud(const orc_node_t *nodes, size_t n, const orc_params_t *p,
                         float *xyzi);

/* ---- E7 ("laser_geometry-style LaserScan -> cloud"; not in the
 * reference, parity unpinned): beam i of a published LaserScan (angle_min = 0, angle_increment
 * per src/rplidar_node.cpp:635 / :666-668) -> (r cos, r sin, 0, intensity); non-finite ranges
 * dropped, clip_enable applies range_min/max.  xyzi must hold 4*count floats. */
size_t orc_laserscan_to_cloud(const float *ranges, const float *intensities, uint32_t count,
                              const orc_params_t *p, float *xyzi);

/* ---- a-ext E5 alone on an xyzi array (O(n^2)). keep[] receives 0/1. */
void orc_ror_mask(const float *xyzi, size_t n, float radius, uint32_t k,
                  uint8_t *keep);

/* ---- a-ext E4: voxel grid over an xyzi array. out must hold 4*n floats;
 * cells (optional, may be NULL) receives (ix,iy) int32 pairs; counts (optional)
 * receives the per-cell point count. Returns number of occupied cells. */
size_t orc_voxel_grid(const float *xyzi, size_t n, float leaf, float *out,
                      int32_t *cells, uint32_t *counts);

/* ---- full E1..E5 pipeline for one scan (what rplgpu_scan_to_cloud does). */
size_t orc_cloud_pipeline(const orc_node_t *nodes, size_t n, const orc_params_t *p,
                          float *out, int32_t *cells, uint32_t *counts);

/* ---- batched, multi-threaded drivers used only for the timed CPU baseline.
 * nodes is B scans of stride n_stride nodes; n_per_scan[b] valid entries each.
 * Return the total number of output elements (beams or points) so the work
 * cannot be optimised away. */
uint64_t orc_batch_ascend(orc_node_t *nodes, size_t n_stride,
                          const uint32_t *n_per_scan, size_t B, int threads);
uint64_t orc_batch_laserscan(const orc_node_t *nodes, size_t n_stride,
                             const uint32_t *n_per_scan, size_t B,
                             const orc_params_t *p, int threads);
uint64_t orc_batch_cloud(const orc_node_t *nodes, size_t n_stride,
                         const uint32_t *n_per_scan, size_t B,
                         const orc_params_t *p, int threads);
/* whole-batch check of a device-made voxel cloud batch against orc_cloud_pipeline (see oracle.cpp) */
uint64_t orc_batch_cloud_check(const orc_node_t *nodes, size_t n_stride,
                               const uint32_t *n_per_scan, size_t B, const orc_params_t *p,
                               const float *got_xyzi, const uint64_t *got_start,
                               const uint32_t *got_npts, const uint32_t *got_keys, int threads,
                               uint32_t *res);

/* whole-batch checks of device-made ascended scans / LaserScan arrays against orc_ascend /
 * orc_publish_scan on all host threads (see oracle.cpp for what res[] holds) */
uint64_t orc_batch_ascend_check(const orc_node_t *src, const orc_node_t *got, size_t n_stride,
                                const uint32_t *n_per_scan, size_t B, int threads, uint32_t *res);
uint64_t orc_batch_laserscan_check(const orc_node_t *nodes, size_t n_stride,
                                   const uint32_t *n_per_scan, size_t B, const orc_params_t *p,
                                   const float *got_ranges, const float *got_intens,
                                   const uint32_t *got_count, size_t out_stride, int threads,
                                   uint32_t *res);


/* =====================================================================================
 * SURVEY.md §8(f) rows 1-2 — the step before the hot path: sample-data unpackers
 * (src/sdk/src/dataunpacker/unpacker/handler_*.cpp) and scan assembly
 * (ScanDataHolder, src/sdk/src/sl_lidar_driver.cpp:236-360).  Restated in
 * oracle_unpack.cpp; PINNED against the reference's own code (oracle/_ref/libunpackref.so).
 * ===================================================================================== */
/* answer types: src/sdk/include/sl_lidar_cmd.h:144-151 */
#define ORC_ANS_MEASUREMENT 0x81
#define ORC_ANS_CAPSULED 0x82
#define ORC_ANS_HQ 0x83
#define ORC_ANS_CAPSULED_ULTRA 0x84
#define ORC_ANS_DENSE_CAPSULED 0x85
#define ORC_ANS_ULTRA_DENSE_CAPSULED 0x86

typedef struct orc_unpack_state {
  int32_t last_sync_bit; /* dense: static lastNodeSyncBit; ultra-dense: _last_node_sync_bit */
  int32_t last_dist_q2;  /* ultra-dense: _last_dist_q2 */
} orc_unpack_state_t;

size_t orc_frame_size(uint8_t ans_type);
size_t orc_frame_stream(uint8_t ans_type, const uint8_t *bytes, size_t nbytes,
                        uint32_t *frame_off, uint8_t *gap, size_t cap);
size_t orc_unpack_frames(uint8_t ans_type, const uint8_t *bytes, const uint32_t *frame_off,
                         const uint8_t *gap, size_t nframes, uint32_t sample_duration_us,
                         orc_unpack_state_t *st, orc_node_t *out, size_t cap,
                         uint32_t *reset_at, size_t reset_cap, size_t *n_reset,
                         uint32_t *n_checksum_err);
size_t orc_unpack(uint8_t ans_type, const uint8_t *bytes, size_t nbytes,
                  uint32_t sample_duration_us, orc_unpack_state_t *st, orc_node_t *out,
                  size_t cap, uint32_t *reset_at, size_t reset_cap, size_t *n_reset,
                  uint32_t *n_checksum_err);
size_t orc_segment(const orc_node_t *nodes, size_t n, const uint32_t *reset_at, size_t n_reset,
                   size_t max_count, orc_node_t *out, size_t out_cap, uint32_t *scan_off,
                   size_t scan_cap);

#ifdef __cplusplus
}
#endif
#endif /* RPL_ORACLE_H_ */
