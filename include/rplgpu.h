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
 *   pre the step before S1 for recorded input: SDK sample-data unpackers + scan assembly
 *         -> rplgpu_frame_stream / rplgpu_decode_batch_dev / rplgpu_segment_batch_dev
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
/* 0x80u: reserved (rounds 4-5: RPLGPU_SCAN_NOT_PRODUCED of the pipelined two-kernel voxel form, removed) */
/* per-stream status bits of the decode stage */
#define RPLGPU_STREAM_UNFRAMED 0x10u         /* frames given back to back but a frame does not start
                                                 with its sync pattern: run rplgpu_frame_stream */
#define RPLGPU_STREAM_FRAMES_TRUNCATED 0x20u /* more frames / sync nodes than one call handles */
#define RPLGPU_STREAM_RESETS_TRUNCATED 0x40u /* reset list / scan table (caller-sized) too small */

/* measurement answer types (src/sdk/include/sl_lidar_cmd.h:144-151) */
#define RPLGPU_ANS_MEASUREMENT 0x81
#define RPLGPU_ANS_CAPSULED 0x82
#define RPLGPU_ANS_HQ 0x83
#define RPLGPU_ANS_CAPSULED_ULTRA 0x84
#define RPLGPU_ANS_DENSE_CAPSULED 0x85
#define RPLGPU_ANS_ULTRA_DENSE_CAPSULED 0x86

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
 * (max_samples_per_scan <= RPLGPU_MAX_SAMPLES_PER_SCAN, max_batch < 2^24). */
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
/* d_nodes: B scans, scan b at d_nodes + b*n_stride, d_n_per_scan[b] samples used (clamped to
 * n_stride and RPLGPU_MAX_SAMPLES_PER_SCAN: a kernel never reads past a scan's slot). */
int32_t rplgpu_ascend_batch_dev(rplgpu_handle_t h, rplgpu_node_t *d_nodes, uint32_t n_stride,
                                const uint32_t *d_n_per_scan, uint32_t B, uint32_t *d_status);
/* d_ranges/d_intensities: B*n_stride floats; d_beam_count: B. */
int32_t rplgpu_laserscan_batch_dev(rplgpu_handle_t h, const rplgpu_node_t *d_nodes,
                                   uint32_t n_stride, const uint32_t *d_n_per_scan, uint32_t B,
                                   const rplgpu_params_t *p, float *d_ranges,
                                   float *d_intensities, uint32_t *d_beam_count);
/* == S1 -> S3 as the node runs them: RealLidarDriver::grab_scan_data with
 * apply_geometric_correction (src/lidar_driver_wrapper.cpp:328-337: ascendScanData in place, result
 * ignored, nodes copied to the caller) followed by RPlidarNode::publish_scan on those nodes
 * (src/rplidar_node.cpp:568-680), for a batch, in ONE pass over the raw nodes.  publish_scan drops
 * every node with dist_mm_q2 == 0 (:584) — the only nodes whose angle ascendScanData rewrites
 * (src/sdk/src/sl_lidar_driver.cpp:171-178) — and sorts what is left by angle itself (:607-609), so
 * the LaserScan of the ascended scan IS the LaserScan of the raw scan (this library's order inside
 * a run of equal angles is the input order before and after the ascend step, which is a stable
 * sort): d_ranges / d_intensities / d_beam_count as rplgpu_laserscan_batch_dev.  write_ascended != 0
 * also leaves the ascended nodes in d_nodes (what the caller's vector holds after grab_scan_data)
 * and the per-scan ascend status in d_status (RPLGPU_SCAN_ALL_INVALID = SL_RESULT_OPERATION_FAIL,
 * buffer untouched); with write_ascended == 0 d_nodes is only read and d_status is not written. */
int32_t rplgpu_ascend_laserscan_batch_dev(rplgpu_handle_t h, rplgpu_node_t *d_nodes,
                                          uint32_t n_stride, const uint32_t *d_n_per_scan,
                                          uint32_t B, const rplgpu_params_t *p, float *d_ranges,
                                          float *d_intensities, uint32_t *d_beam_count,
                                          int32_t write_ascended, uint32_t *d_status);
/* d_xyzi: B*out_stride points of 4 floats; d_n_points, d_status: B. */
int32_t rplgpu_cloud_batch_dev(rplgpu_handle_t h, const rplgpu_node_t *d_nodes, uint32_t n_stride,
                               const uint32_t *d_n_per_scan, uint32_t B, const rplgpu_params_t *p,
                               float *d_xyzi, uint32_t out_stride, uint32_t *d_n_points,
                               uint32_t *d_status);
/* Voxelised clouds of a whole batch in ONE contiguous cloud ("arena"), no packing pass: every
 * scan reserves exactly its cells once their number is known, so the scans lie in the arena in
 * completion order; scan b is d_arena[d_scan_start[b] .. + d_n_points[b]).  *d_cursor (a device
 * word, reset by this call) ends up as the total number of points; points beyond
 * arena_capacity are dropped and flagged RPLGPU_SCAN_OUT_TRUNCATED.  Requires voxel_enable. */
int32_t rplgpu_cloud_arena_dev(rplgpu_handle_t h, const rplgpu_node_t *d_nodes, uint32_t n_stride,
                               const uint32_t *d_n_per_scan, uint32_t B, const rplgpu_params_t *p,
                               float *d_arena, uint64_t arena_capacity, uint64_t *d_cursor,
                               uint64_t *d_scan_start, uint32_t *d_n_points, uint32_t *d_status);
/* Pack the per-scan regions into one contiguous cloud (scan order) for the xGMI
 * all-gather: d_offsets[B+1] (points), d_packed sized for sum(n_points). */
int32_t rplgpu_pack_clouds_dev(rplgpu_handle_t h, const float *d_xyzi, uint32_t out_stride,
                               const uint32_t *d_n_points, uint32_t B, float *d_packed,
                               uint64_t *d_offsets);

/* ---- the step before the path: recorded answer streams -> nodes -> scans -------------
 * (SURVEY.md §8(f) rows 1-2).  Replaces, for recorded / batched input, the reference SDK's
 *   sl::internal::LIDARSampleDataUnpacker::onSampleData(ansType, buf, len)
 *     src/sdk/src/dataunpacker/dataunpacker.h:79, handlers unpacker/handler_*.cpp
 *   ScanDataHolder::pushScanNodeData / rewindCurrentScanData
 *     src/sdk/src/sl_lidar_driver.cpp:272-315 (driven from :1645-1653)
 * Decoding is split like the reference's byte state machines are not: FRAMING (which bytes
 * form frames; sequential, trivial, done on the host by rplgpu_frame_stream) and DECODING
 * (checksums, inter-capsule state, per-sample integer arithmetic; on the GPU).  Node
 * timestamps are not produced (the reference uses the decoding host's wall clock). */
size_t rplgpu_frame_size(uint8_t ans_type);      /* bytes per frame, 0 = unknown type */
size_t rplgpu_nodes_per_frame(uint8_t ans_type); /* nodes a frame can publish */
uint32_t rplgpu_decode_max_frames(uint8_t ans_type); /* frames of one stream per decode call */
/* Performance hint, no change of results: rplgpu_decode_batch_dev calls of a capsule type with
 * max_frames up to this value take the LDS-staged decoder (the workgroup copies its stream into
 * LDS once and decodes from there: 15-30 % faster on full batches) — with
 * back-to-back frames (d_frame_off == NULL) always; with frame offsets for every stream whose
 * frames lie in ascending order, at most 16383 rejected bytes behind their back-to-back place and
 * within the LDS the call reserves (rplgpu_frame_stream's output of a link that loses a few bytes
 * now and then), the other streams of the call going through the plain kernel.  Longer pieces
 * and the other types take the plain one.  0: the type has no staged decoder. */
uint32_t rplgpu_decode_staged_frames(uint8_t ans_type);
/* Host framing: the position-0/1 rules of every onData loop.  frame_off[k] = byte offset of
 * frame k; gap[k] = 1 when bytes were rejected between frame k-1 and frame k (that clears the
 * reference's previous-capsule latch).  Returns the number of frames found (may exceed cap). */
size_t rplgpu_frame_stream(uint8_t ans_type, const uint8_t *bytes, size_t nbytes,
                           uint32_t *frame_off, uint8_t *gap, size_t cap);
/* Decode B streams.  Stream b: bytes at d_bytes + b*stream_stride; d_n_frames[b] frames at byte
 * offsets d_frame_off[b*max_frames + k] (NULL: back to back, k*frame_size — then every frame
 * must start with its sync pattern or the stream gets RPLGPU_STREAM_UNFRAMED and no output);
 * d_gap optional (same shape).  sample_duration_us: SlamtecLidarTimingDesc::sample_duration_uS
 * (1..1000000; dense / ultra-dense discard threshold).  d_state_in/out (optional, 4 x int32 per
 * stream): {last sync bit, last dist_q2, flags, reserved} carried between calls on the same
 * stream (the reference keeps the first two in a function-level static / members).  flags bit 0
 * (input only): frame 0 of this call is the previous call's last frame, passed again only as
 * the predecessor of frame 1 — this is how a recording longer than rplgpu_decode_max_frames is
 * cut without losing the capsule at the cut.  Outputs per stream: nodes at
 * d_nodes + b*node_stride, d_n_nodes[b]; d_reset_at[b*reset_stride + i] = number of nodes
 * published before the i-th scan-reset request, d_n_reset[b]; d_n_errors[b] checksum / CRC
 * failures; d_status[b]. */
int32_t rplgpu_decode_batch_dev(rplgpu_handle_t h, uint8_t ans_type, uint32_t sample_duration_us,
                                const uint8_t *d_bytes, uint64_t stream_stride,
                                const uint32_t *d_frame_off, const uint8_t *d_gap,
                                const uint32_t *d_n_frames, uint32_t max_frames, uint32_t B,
                                const int32_t *d_state_in, int32_t *d_state_out,
                                rplgpu_node_t *d_nodes, uint32_t node_stride, uint32_t *d_n_nodes,
                                uint32_t *d_reset_at, uint32_t reset_stride, uint32_t *d_n_reset,
                                uint32_t *d_n_errors, uint32_t *d_status);
/* Scan assembly: completed scans of stream b back to back at d_out_nodes + b*out_stride, scan s
 * = [d_scan_off[b*(scan_cap+1)+s], ..+s+1]), d_n_scans[b].  max_count = ScanDataHolder capacity
 * (8192 in the reference; a longer scan keeps overwriting its last slot). */
int32_t rplgpu_segment_batch_dev(rplgpu_handle_t h, const rplgpu_node_t *d_nodes,
                                 uint32_t node_stride, const uint32_t *d_n_nodes,
                                 const uint32_t *d_reset_at, uint32_t reset_stride,
                                 const uint32_t *d_n_reset, uint32_t B, uint32_t max_count,
                                 rplgpu_node_t *d_out_nodes, uint32_t out_stride,
                                 uint32_t *d_scan_off, uint32_t scan_cap, uint32_t *d_n_scans,
                                 uint32_t *d_status);
/* Completed scans of all streams -> the fixed-stride scan batch of the *_batch_dev entry points
 * (scan g at d_batch + g*n_stride, stream-major order).  d_scan_base: B+1 words of scratch, on
 * return d_scan_base[B] = total number of scans. */
int32_t rplgpu_scans_to_batch_dev(rplgpu_handle_t h, const rplgpu_node_t *d_seg_nodes,
                                  uint32_t seg_stride, const uint32_t *d_scan_off,
                                  uint32_t scan_cap, const uint32_t *d_n_scans, uint32_t B,
                                  uint32_t *d_scan_base, rplgpu_node_t *d_batch, uint32_t n_stride,
                                  uint32_t max_scans, uint32_t *d_n_per_scan);
/* The whole step before the path in ONE call: recorded streams -> completed scans, written
 * straight into the fixed-stride batch the *_batch_dev entry points take.  The four capsule
 * types: scan boundaries follow from the capsule headers, so the nodes of completed scans
 * are decoded directly into their slots and the others not at all; the other types (and streams
 * with more than 256 sync nodes / 64 reset requests per call) are decoded to a node stream in
 * scratch first, with the decoder's own list of sync nodes.  Decoding as rplgpu_decode_batch_dev, assembly rules as rplgpu_segment_batch_dev
 * (ScanDataHolder, src/sdk/src/sl_lidar_driver.cpp:272-315).  Completed scan s of stream b is
 * batch slot g = b*scan_cap + s: nodes at d_batch + g*n_stride, d_n_per_scan[g] of them
 * (RPLGPU_SCAN_OUT_TRUNCATED in d_status[b] when a scan is longer than n_stride); the slots a
 * stream does not fill get d_n_per_scan[g] = 0, which every batch entry point takes as an empty
 * scan.  d_n_scans[b] = completed scans of stream b (<= scan_cap).  d_status is required.
 * RPLGPU_STREAM_RESETS_TRUNCATED: a stream completed more than scan_cap scans (the first scan_cap
 * are delivered).  The decoded node streams live in scratch owned by the handle
 * (B * max_frames * nodes_per_frame nodes; allocated on first use, grown on demand). */
int32_t rplgpu_decode_scans_dev(rplgpu_handle_t h, uint8_t ans_type, uint32_t sample_duration_us,
                                const uint8_t *d_bytes, uint64_t stream_stride,
                                const uint32_t *d_frame_off, const uint8_t *d_gap,
                                const uint32_t *d_n_frames, uint32_t max_frames, uint32_t B,
                                const int32_t *d_state_in, int32_t *d_state_out,
                                uint32_t max_count, rplgpu_node_t *d_batch, uint32_t n_stride,
                                uint32_t scan_cap, uint32_t *d_n_per_scan, uint32_t *d_n_scans,
                                uint32_t *d_n_errors, uint32_t *d_status);
/* A recording longer than one call (rplgpu_decode_max_frames frames per stream and call): the same,
 * plus the scan a stream is still building when the call ends.  ScanDataHolder keeps that scan in
 * its operational buffer (src/sdk/src/sl_lidar_driver.cpp:272-310); here it leaves the call in
 * d_carry_out (d_carry_len_out[b] nodes at d_carry_out + b*carry_stride, at most
 * min(max_count, carry_stride), the buffer's own "keep overwriting the last slot" rule applied) and
 * enters the next call as d_carry_in / d_carry_len_in (NULL, NULL for the first call of a recording):
 * it is completed by that call's first sync node — and delivered as its first scan — unless a
 * scan-reset request comes first.  Fed call after call with d_state_out -> d_state_in (for the
 * capsule types each piece after the first starts one frame early with state flags bit 0 set, as for
 * rplgpu_decode_batch_dev) and
 * carry_out -> carry_in (two buffers, swapped by the caller; in and out must differ), a recording of
 * any length becomes exactly the scans one pass of the SDK's unpacker + ScanDataHolder makes of it,
 * without the host seeing a node.  Every stream takes the decode-then-assemble path here (the
 * capsule types' fused decoder skips the nodes outside a call's completed scans). */
int32_t rplgpu_decode_scans_carry_dev(rplgpu_handle_t h, uint8_t ans_type,
                                      uint32_t sample_duration_us, const uint8_t *d_bytes,
                                      uint64_t stream_stride, const uint32_t *d_frame_off,
                                      const uint8_t *d_gap, const uint32_t *d_n_frames,
                                      uint32_t max_frames, uint32_t B, const int32_t *d_state_in,
                                      int32_t *d_state_out, uint32_t max_count,
                                      rplgpu_node_t *d_batch, uint32_t n_stride, uint32_t scan_cap,
                                      uint32_t *d_n_per_scan, uint32_t *d_n_scans,
                                      uint32_t *d_n_errors, uint32_t *d_status,
                                      const rplgpu_node_t *d_carry_in, const uint32_t *d_carry_len_in,
                                      rplgpu_node_t *d_carry_out, uint32_t *d_carry_len_out,
                                      uint32_t carry_stride);
/* One stream of any length, HOST buffers: framing on the host, decode on the GPU in pieces of
 * rplgpu_decode_max_frames frames (overlapping by one frame, see flags bit 0).  No state is kept
 * in the handle: pass state in/out explicitly ({0,0,0,0} for a fresh unpacker).  nodes: cap
 * entries; reset_at: reset_cap entries (node positions in the whole output). */
int32_t rplgpu_decode_stream(rplgpu_handle_t h, uint8_t ans_type, uint32_t sample_duration_us,
                             const uint8_t *bytes, size_t nbytes, int32_t state[4],
                             rplgpu_node_t *nodes, size_t cap, size_t *n_nodes,
                             uint32_t *reset_at, size_t reset_cap, size_t *n_reset,
                             uint32_t *n_errors);

/* LaserScan scalars from a beam count (host arithmetic of :623-627,:635-638,:666-669). */
void rplgpu_fill_meta(const rplgpu_params_t *p, uint32_t count, double scan_duration,
                      rplgpu_scan_meta_t *meta);

/* Optional second output of the voxel grid (E4): the CELL of every output point.  While a buffer is
 * set, every launch of the voxel path (rplgpu_cloud_batch_dev / rplgpu_cloud_arena_dev /
 * rplgpu_cloud_fused_voxel_dev with voxel_enable) also writes one 32-bit word per output point,
 * at the point's own index in the output buffer (per-scan region or arena):
 *     (iy + 32768) << 16 | (ix + 32768),   (ix, iy) = (floor(x / leaf), floor(y / leaf)) of SURVEY 8(a-ext) E4
 * i.e. the key the points of a scan are ordered by.  d_cell_keys needs as many words as the
 * output buffer has points; NULL switches the output off (the default).  The buffer is the
 * caller's and must stay valid until the launches that use it have completed. */
int32_t rplgpu_set_cell_key_output(rplgpu_handle_t h, uint32_t *d_cell_keys);

/* How the voxel grid (E4) aggregates a block of 128 samples before sorting.  The results are
 * identical in every mode (integer sums); only the time differs.  PLAIN makes one run record per
 * run of samples in one cell (clean rings: ~14 records per block).  TWO_CLASS lets a block that
 * would make more than 26 records be aggregated in the two colours of a checkerboard of cells
 * instead (range noise makes neighbouring samples alternate between two cells: 8 700 -> 6 200
 * records per 32 000-sample scan at 1 cm), at 1.5-2.7 % on clean data.  AUTO (the default) decides
 * per batch launch from the records per scan THIS batch made the last time the handle launched it
 * (same device buffer, stride, scan count and group size: the statistics follow every launch to
 * pinned memory); a batch the handle has not launched before runs PLAIN.  So the first launch over
 * a noisy batch is the slow one and repeated launches (a bench loop, a sensor's frames arriving in
 * one staging buffer) converge after one; no other batch's history enters.  A caller who knows its
 * sensor pins PLAIN or TWO_CLASS. */
#define RPLGPU_VOXEL_AGG_AUTO 0
#define RPLGPU_VOXEL_AGG_PLAIN 1
#define RPLGPU_VOXEL_AGG_TWO_CLASS 2
int32_t rplgpu_set_voxel_aggregation(rplgpu_handle_t h, int32_t mode);

/* How E5 (radius outlier removal) runs in front of the voxel grid in the ARENA entry points
 * (rplgpu_cloud_arena_dev, rplgpu_cloud_arena_xyi_dev, rplgpu_cloud_fused_voxel_dev).  The results are
 * identical in both modes; only the time differs.  INSIDE (the default): the voxel kernel applies E5
 * while it streams a scan — a kept sample with ror_min_neighbors neighbours among its four nearest
 * indices survives at once, the few others (islands between drop-outs, isolated returns) are settled
 * exactly behind the pass (64 indices either side, then the whole scan) — so a scan is read ONCE.  A
 * work item with more open samples than that (clutter: > 256 after the index test or > 8 after the
 * window) is redone by the two kernels of TWO_KERNELS behind the launch.  TWO_KERNELS: k_ror_mask
 * writes one keep bit per sample, the voxel kernel reads the scan again with the mask (rounds 1-5;
 * also what any launch uses whose divides were not validated on this device, and the single-scan call
 * rplgpu_scan_to_cloud below 8192 samples, where index neighbours are mostly farther apart than the
 * radius).  rplgpu_cloud_batch_dev (per-scan regions) follows the mode as well. */
#define RPLGPU_ROR_INSIDE 0
#define RPLGPU_ROR_TWO_KERNELS 1
int32_t rplgpu_set_ror_mode(rplgpu_handle_t h, int32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* RPLGPU_H_ */
