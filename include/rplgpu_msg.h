/*
 * rplgpu_msg.h — publish-ready messages: the LaserScan / PointCloud2
 * the path produces, laid out as the serialised (CDR) messages a ROS 2 publisher sends, so the
 * node hands the buffer to `publish(const rclcpp::SerializedMessage &)` without building a typed
 * message and without a further copy.
 *
 * What it replaces in the reference, per scan:
 *   sensor_msgs::msg::LaserScan scan_msg; ... scan_msg.ranges.assign / [index] = ...;
 *   scan_pub_->publish(scan_msg);                  src/rplidar_node.cpp:618-682
 * i.e. two std::vector<float> fills, then the middleware's own serialisation pass over them
 * (rosidl_typesupport_fastrtps_cpp + eProsima Fast-CDR; third-party, not in the reference tree:
 * the README names ROS 2 Jazzy, whose default rmw serialises with Fast-CDR 2.2.x as XCDR
 * version 1, PLAIN_CDR, little endian).  Here the device results are copied ONCE, by DMA,
 * straight to their final offsets inside the serialised message; for device-resident batches
 * a kernel assembles B messages in HBM.
 *
 * Wire format (OMG DDS-XTypes 1.3 §7.4.1 "plain CDR" as Fast-CDR emits it for ROS 2):
 *   4-byte encapsulation header 00 01 00 00 (CDR little endian, no options); every primitive
 *   aligned to its size counted from the byte AFTER that header; string = uint32 length
 *   including the terminating NUL, the bytes, the NUL; sequence<T> = uint32 element count, the
 *   elements; bool / uint8 = one byte; no padding after the last member.
 *   std_msgs/Header = { int32 stamp.sec, uint32 stamp.nanosec, string frame_id }.
 *   sensor_msgs/LaserScan = Header, float32 angle_min, angle_max, angle_increment,
 *     time_increment, scan_time, range_min, range_max, float32[] ranges, float32[] intensities.
 *   sensor_msgs/PointCloud2 = Header, uint32 height, width, PointField[] fields
 *     ({string name, uint32 offset, uint8 datatype, uint32 count}), bool is_bigendian,
 *     uint32 point_step, row_step, uint8[] data, bool is_dense — with the E3 layout of
 *     fields x,y,z,intensity FLOAT32 (=7) at 0/4/8/12, point_step 16, height 1,
 *     width = #points, row_step = 16*width, is_bigendian false, is_dense true.
 *
 * Parity: the reference holds no serialised vectors and Fast-CDR is not in this image, so the
 * byte layout is checked against an independent restatement of the format (oracle/cdr_oracle.py)
 * and hand-derived known-answer bytes — "parity unpinned" for the wire format itself; the float
 * payloads inside the messages are the same bit-exact results as rplgpu.h's entry points.
 */
#ifndef RPLGPU_MSG_H_
#define RPLGPU_MSG_H_

#include "rplgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rplgpu_stamp {
  int32_t sec;      /* builtin_interfaces/Time */
  uint32_t nanosec;
} rplgpu_stamp_t;

/* Byte offsets inside one serialised LaserScan (from the first byte of the encapsulation
 * header).  All are multiples of 4. */
typedef struct rplgpu_laserscan_layout {
  uint32_t scalars_off;         /* angle_min; the 7 float32 scalars follow each other */
  uint32_t ranges_len_off;      /* uint32 ranges.size() */
  uint32_t ranges_off;          /* ranges[0] */
  uint32_t intensities_len_off; /* uint32 intensities.size() */
  uint32_t intensities_off;     /* intensities[0] */
  uint32_t total_len;           /* serialised size */
} rplgpu_laserscan_layout_t;

/* Byte offsets inside one serialised PointCloud2. */
typedef struct rplgpu_cloud_layout {
  uint32_t width_off;    /* uint32 width (height precedes it) */
  uint32_t row_step_off; /* uint32 row_step (point_step precedes it) */
  uint32_t data_len_off; /* uint32 data.size() = 16 * n_points */
  uint32_t data_off;     /* first point; a multiple of 4 */
  uint32_t is_dense_off; /* the last byte */
  uint32_t total_len;
} rplgpu_cloud_layout_t;

/* ---- host-only: layout and everything but the bulk arrays (no device needed) ------------- */
/* frame_id_len = strlen(frame_id).  Returns RPLGPU_ERR_INVALID_ARG if the message would not
 * fit 32-bit offsets. */
int32_t rplgpu_msg_laserscan_layout(size_t frame_id_len, uint32_t count,
                                    rplgpu_laserscan_layout_t *out);
int32_t rplgpu_msg_cloud_layout(size_t frame_id_len, uint32_t n_points,
                                rplgpu_cloud_layout_t *out);
/* Write encapsulation header, Header, the 7 scalars of `meta` and both sequence lengths
 * (meta->count) into msg[0..cap); the float arrays at ranges_off / intensities_off are left
 * for the caller (or the DMA) to fill.  RPLGPU_ERR_CAPACITY if total_len > cap. */
int32_t rplgpu_msg_laserscan_header(const char *frame_id, rplgpu_stamp_t stamp,
                                    const rplgpu_scan_meta_t *meta, uint8_t *msg, size_t cap,
                                    rplgpu_laserscan_layout_t *layout);
/* Write everything of a PointCloud2 but the points (16 * n_points bytes at data_off). */
int32_t rplgpu_msg_cloud_header(const char *frame_id, rplgpu_stamp_t stamp, uint32_t n_points,
                                uint8_t *msg, size_t cap, rplgpu_cloud_layout_t *layout);

/* ---- pinned message buffers ---------------------------------------------------------------- */
/* Page-locked host memory: a message buffer obtained here receives the device results by DMA
 * with no staging copy (any other host pointer works too, through the runtime's staging). */
int32_t rplgpu_host_alloc(rplgpu_handle_t h, size_t bytes, void **out);
int32_t rplgpu_host_free(rplgpu_handle_t h, void *p);

/* ---- single scan: raw nodes -> one serialised message in a HOST buffer --------------------- */
/* == the body of publish_scan (src/rplidar_node.cpp:568-680) plus the serialisation of the
 * result.  *msg_len = 0 and meta->published = 0 when publish_scan would have returned without
 * publishing (:561,:611).  cap must hold the message for the worst case count == n
 * (rplgpu_msg_laserscan_layout(strlen(frame_id), n)). */
int32_t rplgpu_scan_to_laserscan_msg(rplgpu_handle_t h, const rplgpu_node_t *nodes, size_t n,
                                     const rplgpu_params_t *p, double scan_duration,
                                     const char *frame_id, rplgpu_stamp_t stamp, uint8_t *msg,
                                     size_t cap, size_t *msg_len, rplgpu_scan_meta_t *meta);
/* ext E1-E5 -> serialised PointCloud2 (an empty cloud is a valid message: width 0). */
int32_t rplgpu_scan_to_cloud_msg(rplgpu_handle_t h, const rplgpu_node_t *nodes, size_t n,
                                 const rplgpu_params_t *p, const char *frame_id,
                                 rplgpu_stamp_t stamp, uint8_t *msg, size_t cap, size_t *msg_len,
                                 uint32_t *n_points, uint32_t *status);

/* ---- batches: B serialised messages assembled in DEVICE memory ----------------------------- */
/* From the outputs of rplgpu_laserscan_batch_dev (same d_ranges / d_intensities / n_stride /
 * d_beam_count): message b at d_msgs + b*msg_stride (msg_stride a multiple of 4), its length
 * in d_msg_len[b]; 0 = not published (beam count 0) or msg_stride too small (then d_status[b]
 * gets RPLGPU_SCAN_OUT_TRUNCATED).  The scalars are computed on the device with the
 * reference's own expressions (:623-627,:634-638,:665-669; IEEE fp64 divides).
 * d_stamps: B stamps; d_scan_duration: B doubles; both device pointers. */
int32_t rplgpu_laserscan_msgs_dev(rplgpu_handle_t h, const float *d_ranges,
                                  const float *d_intensities, uint32_t n_stride,
                                  const uint32_t *d_beam_count, uint32_t B,
                                  const rplgpu_params_t *p, const char *frame_id,
                                  const rplgpu_stamp_t *d_stamps, const double *d_scan_duration,
                                  uint8_t *d_msgs, uint32_t msg_stride, uint32_t *d_msg_len,
                                  uint32_t *d_status);
/* From the outputs of rplgpu_cloud_batch_dev (d_scan_start = NULL: scan b at
 * d_xyzi + b*out_stride points) or rplgpu_cloud_arena_dev (d_xyzi = the arena, d_scan_start =
 * its per-scan starts).  Same message slots as above. */
int32_t rplgpu_cloud_msgs_dev(rplgpu_handle_t h, const float *d_xyzi, uint32_t out_stride,
                              const uint64_t *d_scan_start, const uint32_t *d_n_points,
                              uint32_t B, const char *frame_id, const rplgpu_stamp_t *d_stamps,
                              uint8_t *d_msgs, uint32_t msg_stride, uint32_t *d_msg_len,
                              uint32_t *d_status);

/* ---- LaserScan -> PointCloud2 projection (E7; row 3's `laser_geometry`-style cloud source) ---
 * Input: what publish_scan produces (src/rplidar_node.cpp:618-662 Mode A, :663-680 Mode B) —
 * the outputs of rplgpu_laserscan_batch_dev, still in HBM.  For scan b with count =
 * d_beam_count[b] and angle_increment as the reference computes it (Mode A :635, Mode B
 * :666-668; p->scan_processing selects), beam i is kept iff ranges[i] is finite (Mode A's empty
 * bins are +inf, :640) and, with p->clip_enable, range_min <= ranges[i] <= range_max; then
 *   theta = angle_min + float(i) * angle_increment            (float32; angle_min = 0, :623)
 *   x = ranges[i] * (float)cos((double)theta);  y = ranges[i] * (float)sin((double)theta);  z = 0
 *   intensity = intensities[i]
 * in beam order, E3 layout (16-byte points), d_n_points[b] points at d_xyzi + b*out_stride.
 * The serialised message of such a cloud is rplgpu_cloud_msgs_dev on the result. */
int32_t rplgpu_laserscan_to_cloud_batch_dev(rplgpu_handle_t h, const float *d_ranges,
                                            const float *d_intensities, uint32_t n_stride,
                                            const uint32_t *d_beam_count, uint32_t B,
                                            const rplgpu_params_t *p, float *d_xyzi,
                                            uint32_t out_stride, uint32_t *d_n_points,
                                            uint32_t *d_status);
/* One scan, HOST buffers: ranges / intensities as a published LaserScan holds them (count
 * beams, count <= max_samples_per_scan); xyzi: count * 4 floats. */
int32_t rplgpu_laserscan_to_cloud(rplgpu_handle_t h, const float *ranges,
                                  const float *intensities, uint32_t count,
                                  const rplgpu_params_t *p, float *xyzi, uint32_t *n_points);

/* ---- several sensors -> one fused cloud (first step) ------------------ */
/* Rigid transform of the clouds of B scans, in place: d_pose holds B row-major 3x4 matrices
 * [R | t] (tf2: target frame <- frame_id of scan b; the reference itself only ever broadcasts
 * the identity, src/rplidar_node.cpp:183-197).  Same addressing as rplgpu_cloud_msgs_dev
 * (d_scan_start = NULL: per-scan regions of out_stride points; else the arena).  float32,
 *   x' = ((r00*x + r01*y) + r02*z) + t0   (products, then sums left to right, no FMA),
 * y', z' likewise, intensity untouched. */
int32_t rplgpu_transform_clouds_dev(rplgpu_handle_t h, float *d_xyzi, uint32_t out_stride,
                                    const uint64_t *d_scan_start, const uint32_t *d_n_points,
                                    uint32_t B, const float *d_pose);
/* Motion de-skew of the plain cloud (E1 clip, E2 polar -> XYZ, optionally the E5 mask; not
 * with voxel_enable): d_motion holds per scan (vx, vy, wz, time_increment) — the sensor's planar
 * twist in its own frame at the first sample [m/s, m/s, rad/s] and the time between samples [s]
 * (LaserScan.time_increment, src/rplidar_node.cpp:637,668).  Sample i (its index in the scan as
 * handed in, i.e. acquisition order) was taken at tau = float(i) * time_increment; its point
 * moves to the sensor pose at the first sample:
 *   a = wz * tau;  a2 = a * a
 *   s = a * (1 + a2 * (-1/6 + a2 * (1/120)))            (Horner, every operation rounded to
 *   c = 1 + a2 * (-1/2 + a2 * (1/24 + a2 * (-1/720)))    float32 once, no FMA)
 *   x' = (c*x - s*y) + vx*tau;   y' = (s*x + c*y) + vy*tau
 * stated for |a| <= 0.5 rad.  These ARE the definition of E6 (the oracle evaluates the same
 * polynomials, so parity does not depend on their accuracy); against the true rotation s is short by
 * a^7 / 5040 and c by a^8 / 40320: 2e-8 / 7e-10 at 0.27 rad (a 10 Hz scan at 2.7 rad/s), 1.5e-6 / 1e-7
 * at 0.5 rad (round 6 corrects the "< 2e-8 at 0.5 rad" this comment used to claim).  Output as
 * rplgpu_cloud_batch_dev. */
int32_t rplgpu_cloud_deskew_batch_dev(rplgpu_handle_t h, const rplgpu_node_t *d_nodes,
                                      uint32_t n_stride, const uint32_t *d_n_per_scan, uint32_t B,
                                      const rplgpu_params_t *p, const float *d_motion,
                                      float *d_xyzi, uint32_t out_stride, uint32_t *d_n_points,
                                      uint32_t *d_status);
/* E8 — ONE voxel grid for a GROUP of scans (row 4: de-skew in front of the voxel grid, and the
 * cross-sensor voxel grid).  Scans [g*group, (g+1)*group) of the batch — e.g. the 8 sensors of one
 * time step, or a single scan (group = 1) — are voxelised together: every kept sample (E1 clip,
 * E5 mask when ror_enable) gives (x, y) by E2; then, in float32, one rounding per operation:
 *   E6 de-skew with the scan's (vx, vy, wz, time_increment) from d_motion (NULL: none) —
 *     tau = float(i) * time_increment, formulas of rplgpu_cloud_deskew_batch_dev above;
 *   the scan's planar pose from d_pose2d (NULL: identity), 6 floats per scan
 *     (r00 r01 tx r10 r11 ty):  x'' = (r00*x' + r01*y') + tx;  y'' = (r10*x' + r11*y') + ty;
 * and the E4 voxel grid (leaf p->voxel_leaf) runs over ALL points of the group: one output point
 * per occupied cell = centroid (fp64 sum / count, rounded to float), mean intensity, z = 0, cells
 * in (iy, ix) order.  Outputs as rplgpu_cloud_arena_dev, indexed by GROUP: d_group_start[g],
 * d_n_points[g], d_status[g] for g < ceil(B / group); *d_cursor = all points.
 * The raw scans are streamed once; a group's run records beyond the on-chip queue go through the
 * handle's record store (L2).  The reference publishes time_increment / scan_time
 * (src/rplidar_node.cpp:627,637-638) and an identity transform (:183-197); nothing there
 * consumes them — this is the consumer. */
int32_t rplgpu_cloud_fused_voxel_dev(rplgpu_handle_t h, const rplgpu_node_t *d_nodes,
                                     uint32_t n_stride, const uint32_t *d_n_per_scan, uint32_t B,
                                     uint32_t group, const rplgpu_params_t *p,
                                     const float *d_motion, const float *d_pose2d, float *d_arena,
                                     uint64_t arena_capacity, uint64_t *d_cursor,
                                     uint64_t *d_group_start, uint32_t *d_n_points,
                                     uint32_t *d_status);
/* Time alignment inside a group (row 4, the temporal side): the scans fused into one grid do not
 * start at the same instant — every sensor free-runs, and a node stamps its LaserScan with its own
 * start time (src/rplidar_node.cpp:620) and publishes time_increment / scan_time (:627,637-638).  d_t0 holds
 * per scan of the batch the time [s] of its FIRST sample relative to the instant the group is fused
 * at (scan stamp - fused stamp; negative: the scan started earlier).  While set, E6 — in
 * rplgpu_cloud_deskew_batch_dev and rplgpu_cloud_fused_voxel_dev, which then require d_motion —
 * moves every point to the sensor pose at the FUSED instant instead of the scan's first sample:
 *   tau = t0 + float(i) * time_increment      (product rounded to float32, then the sum)
 * and the formulas of rplgpu_cloud_deskew_batch_dev apply unchanged.  NULL (the default) switches it
 * off: tau = float(i) * time_increment, bit for bit as before.  The buffer is the caller's and must
 * stay valid until the launches that use it have completed.
 * DOMAIN: the sin / cos polynomials of E6 are stated for |a| = |wz * tau| <= 0.5 rad, and with an
 * offset tau covers [t0, t0 + n * time_increment]: the caller must keep |wz| * max(|t0|, |t0 + n *
 * time_increment|) <= 0.5 rad (a few scan periods of offset at a realistic yaw rate leave it: 1 rad/s
 * and t0 = 0.6 s already do).  Nothing checks this — beyond the domain the rotation is still the
 * degree-5 / -6 Taylor value (s short by 5.5e-6 at 0.6 rad, 2e-4 at 1 rad), and the oracle computes the same.
 * NOT consulted by: rplgpu_cloud_arena_dev, rplgpu_cloud_arena_xyi_dev, rplgpu_cloud_batch_dev,
 * rplgpu_scan_to_cloud[_msg] and the LaserScan entry points (no E6 there), nor by
 * rplgpu_cloud_fused_voxel_dev when d_motion is NULL together with NULL offsets (with offsets set and
 * d_motion NULL that call is refused: RPLGPU_ERR_INVALID_ARG). */
int32_t rplgpu_set_scan_time_offsets_dev(rplgpu_handle_t h, const float *d_t0);
/* The whole arena as ONE serialised PointCloud2 (the fused cloud of BASELINE config 5):
 * width = min(*d_total_points, arena_capacity) with d_total_points the arena cursor of
 * rplgpu_cloud_arena_dev — no host round trip.  *d_msg_len (device) = serialised size, or 0 +
 * RPLGPU_SCAN_OUT_TRUNCATED in *d_status when msg_capacity is too small. */
int32_t rplgpu_fused_cloud_msg_dev(rplgpu_handle_t h, const float *d_arena,
                                   const uint64_t *d_total_points, uint64_t arena_capacity,
                                   const char *frame_id, rplgpu_stamp_t stamp, uint8_t *d_msg,
                                   uint64_t msg_capacity, uint64_t *d_msg_len, uint32_t *d_status);

#ifdef __cplusplus
}
#endif
#endif /* RPLGPU_MSG_H_ */
