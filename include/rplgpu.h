/*
 * rplgpu.h — C ABI of librplgpu.so: the MI355X (gfx950) scan-preprocessing path
 * for RPLIDAR-class 2-D lidars.
 *
 * This is the drop-in boundary for ONE hot path of frozenreboot/rplidar_ros2_driver
 * (citations relative to the reference tree):
 *
 *   S1  sl::ILidarDriver::ascendScanData(node_hq*, size_t)
 *         src/sdk/include/sl_lidar_driver.h:477, body src/sdk/src/sl_lidar_driver.cpp:128-184,
 *         called from RealLidarDriver::grab_scan_data src/lidar_driver_wrapper.cpp:329
 *         -> rplgpu_ascend / rplgpu_ascend_batch_dev
 *   S3  RPlidarNode::publish_scan body, src/rplidar_node.cpp:568-680 (mask, Q14->rad,
 *         Q2mm->m, quality->intensity, sort, Mode A min-binning / Mode B raw mapping)
 *         -> rplgpu_scan_to_laserscan / rplgpu_laserscan_batch_dev
 *   ext polar->Cartesian PointCloud2 (x,y,z,intensity FLOAT32, point_step 16) with
 *         quality/range clip (E1), radius-outlier removal (E5, applied before the voxel
 *         grid) and voxel-grid downsample (E4)
 *         (not in the reference; spec in SURVEY.md §8 a-ext / DESIGN.md)
 *         -> rplgpu_scan_to_cloud / rplgpu_cloud_batch_dev
 *
 * Plain C: no exceptions, no STL and no torch types cross this boundary.  All
 * buffers are caller-owned.  Every call returns int32: 0 = OK, negative = error,
 * so the node can fall back to its own CPU loop instead of entering RESETTING
 * (src/rplidar_node.cpp:453-474).  The library itself has NO CPU fallback: without
 * a usable gfx950 device rplgpu_create fails with RPLGPU_ERR_NO_DEVICE.
 *
 * Threading contract: one in-flight call per handle (the node's scan thread,
 * src/rplidar_node.cpp:222); create/destroy from the lifecycle callbacks
 * (on_configure :116 / on_cleanup :244).  Handles share no state — unlike the
 * reference's process-wide static buffer (src/lidar_driver_wrapper.cpp:313).
 */
#ifndef RPLGPU_H_
#define RPLGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPLGPU_ABI_VERSION 1

/* error codes */
#define RPLGPU_OK 0
#define RPLGPU_ERR_INVALID_ARG (-1)
#define RPLGPU_ERR_NO_DEVICE (-2)   /* no HIP device / not gfx950 / runtime missing */
#define RPLGPU_ERR_HIP (-3)         /* a HIP runtime call failed; see rplgpu_last_error */
#define RPLGPU_ERR_CAPACITY (-4)    /* n or B exceeds what the handle was created for */
#define RPLGPU_ERR_ALL_INVALID (-5) /* ascend: every sample invalid == SL_RESULT_OPERATION_FAIL */
#define RPLGPU_ERR_SCAN_OVERFLOW (-6) /* a scan exceeded an on-chip table (see status words) */

/* hard limit of the per-scan kernels (one workgroup holds a scan on chip) */
#define RPLGPU_MAX_SAMPLES_PER_SCAN 32768u

/* per-scan status bits written by the batch entry points (0 == clean) */
#define RPLGPU_SCAN_ALL_INVALID 0x1u   /* ascend: SL_RESULT_OPERATION_FAIL, scan left untouched */
#define RPLGPU_SCAN_CELL_RANGE 0x2u    /* voxel: |cell index| >= 32767 (range/leaf too large) */
#define RPLGPU_SCAN_TABLE_FULL 0x4u    /* voxel: more occupied cells than the on-chip table holds */
#define RPLGPU_SCAN_OUT_TRUNCATED 0x8u /* output region (out_stride) too small; count is clamped */

/* One raw sample == sl_lidar_response_measurement_node_hq_t
 * (src/sdk/include/sl_lidar_cmd.h:272-278): packed, 8 bytes, u32 at offset 2. */
typedef struct __attribute__((packed)) rplgpu_node {
  uint16_t angle_z_q14; /* deg = v * 90 / 16384 */
  uint32_t dist_mm_q2;  /* mm = v / 4 ; 0 => invalid sample */
  uint8_t quality;
  uint8_t flag;         /* bit0 = scan-start sync */
} rplgpu_node_t;

/* Scalar inputs of the path.  The first three are read by publish_scan today;
 * the rest configure the extensions (all "off" reproduces the reference). */
typedef struct rplgpu_params {
  int32_t is_new_protocol;    /* RealLidarDriver && NEW_TYPE, src/rplidar_node.cpp:577-581 */
  int32_t inverted;           /* params_.inverted, src/rplidar_node.cpp:646,676 */
  int32_t scan_processing;    /* params_.scan_processing (Mode A / Mode B), :632 */
  int32_t clip_enable;        /* E1: apply q_min/range_min/range_max to the keep mask */
  uint32_t q_min;             /* E1: keep quality >= q_min (raw byte) */
  float range_min;            /* E1: keep dist_m >= range_min (default 0.15f, :625) */
  float range_max;            /* E1 + LaserScan.range_max (cached_current_max_range_, :626) */
  float voxel_leaf;           /* E4: leaf in metres (default 0.05f) */
  float ror_radius;           /* E5: radius in metres (default 0.10f) */
  uint32_t ror_min_neighbors; /* E5: k (default 2) */
  int32_t ror_enable;         /* E5 on/off */
  int32_t voxel_enable;       /* E4 on/off */
} rplgpu_params_t;

/* LaserScan scalars the node copies into sensor_msgs::msg::LaserScan
 * (src/rplidar_node.cpp:618-627,634-638,665-669). */
typedef struct rplgpu_scan_meta {
  float angle_min, angle_max, angle_increment, time_increment;
  float scan_time, range_min, range_max;
  uint32_t count;    /* ranges.size() == intensities.size() */
  int32_t published; /* 0: publish_scan would have returned without publishing (:561,:611) */
} rplgpu_scan_meta_t;

typedef struct rplgpu_ctx *rplgpu_handle_t;

/* ---- lifecycle ----------------------------------------------------------- */
int32_t rplgpu_abi_version(void);
/* on_configure: binds to HIP device `device_id`, builds the Q14 angle / cos / sin
 * tables on the host with the reference's own expressions and uploads them,
 * allocates staging for max_batch scans of max_samples_per_scan samples
 * (max_samples_per_scan <= RPLGPU_MAX_SAMPLES_PER_SCAN). */
int32_t rplgpu_create(int32_t device_id, uint32_t max_samples_per_scan, uint32_t max_batch,
                      rplgpu_handle_t *out);
int32_t rplgpu_destroy(rplgpu_handle_t h); /* on_cleanup */
const char *rplgpu_last_error(rplgpu_handle_t h);
void rplgpu_default_params(rplgpu_params_t *p);
/* Run this handle's kernels on a caller-owned hipStream_t (NULL = the handle's own). */
int32_t rplgpu_set_stream(rplgpu_handle_t h, void *hip_stream);
int32_t rplgpu_synchronize(rplgpu_handle_t h);

/* ---- single scan, HOST buffers (the drop-in seams) ----------------------- */
/* == S1. In place. *sl_result receives the SDK code (0 or 0x80008001); the return
 * value is RPLGPU_OK in both cases unless the call itself failed. */
int32_t rplgpu_ascend(rplgpu_handle_t h, rplgpu_node_t *nodes, size_t n, uint32_t *sl_result);
/* == body of S3. ranges/intensities: n floats each. */
int32_t rplgpu_scan_to_laserscan(rplgpu_handle_t h, const rplgpu_node_t *nodes, size_t n,
                                 const rplgpu_params_t *p, double scan_duration, float *ranges,
                                 float *intensities, rplgpu_scan_meta_t *meta);
/* ext: xyzi = n*4 floats (PointCloud2 data, point_step 16). */
int32_t rplgpu_scan_to_cloud(rplgpu_handle_t h, const rplgpu_node_t *nodes, size_t n,
                             const rplgpu_params_t *p, float *xyzi, uint32_t *n_points,
                             uint32_t *status);

/* ---- batches, DEVICE-resident buffers (no host copies, async on the stream) */
/* d_nodes: B scans, scan b at d_nodes + b*n_stride, d_n_per_scan[b] samples used. */
int32_t rplgpu_ascend_batch_dev(rplgpu_handle_t h, rplgpu_node_t *d_nodes, uint32_t n_stride,
                                const uint32_t *d_n_per_scan, uint32_t B, uint32_t *d_status);
/* d_ranges/d_intensities: B*n_stride floats; d_beam_count: B. */
int32_t rplgpu_laserscan_batch_dev(rplgpu_handle_t h, const rplgpu_node_t *d_nodes,
                                   uint32_t n_stride, const uint32_t *d_n_per_scan, uint32_t B,
                                   const rplgpu_params_t *p, float *d_ranges,
                                   float *d_intensities, uint32_t *d_beam_count);
/* d_xyzi: B*out_stride points of 4 floats; d_n_points, d_status: B. */
int32_t rplgpu_cloud_batch_dev(rplgpu_handle_t h, const rplgpu_node_t *d_nodes, uint32_t n_stride,
                               const uint32_t *d_n_per_scan, uint32_t B, const rplgpu_params_t *p,
                               float *d_xyzi, uint32_t out_stride, uint32_t *d_n_points,
                               uint32_t *d_status);
/* Pack the per-scan regions into one contiguous cloud (scan order) for the xGMI
 * all-gather: d_offsets[B+1] (points), d_packed sized for sum(n_points). */
int32_t rplgpu_pack_clouds_dev(rplgpu_handle_t h, const float *d_xyzi, uint32_t out_stride,
                               const uint32_t *d_n_points, uint32_t B, float *d_packed,
                               uint64_t *d_offsets);

/* LaserScan scalars from a beam count (host arithmetic of :623-627,:635-638,:666-669). */
void rplgpu_fill_meta(const rplgpu_params_t *p, uint32_t count, double scan_duration,
                      rplgpu_scan_meta_t *meta);

#ifdef __cplusplus
}
#endif
#endif /* RPLGPU_H_ */
