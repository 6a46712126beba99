/*
 * rplgpu_comm.h — the multi-GPU exchange of the path behind the C ABI.
 *
 * Every scan is independent (no cross-scan state in ascendScanData / publish_scan,
 * /root/reference src/sdk/src/sl_lidar_driver.cpp:128-184, src/rplidar_node.cpp:568-680, nor in
 * E1-E5), so a batch shards by scan index (BASELINE config 4) or by sensor (config 5: one GPU
 * per sensor) with NO collective on the data path.  One exchange assembles the result: an
 * all-gather of the ranks' filtered / voxelised clouds, RCCL over xGMI.  xGMI is point to point
 * (7 links x ~153 GB/s per GPU): in a direct all-gather every link carries one rank's slot once,
 * so the slot size — not the number of ranks — sets the time; nothing here is a ring all-reduce.
 *
 * One process per GPU, one handle per process.  The library loads RCCL at run time
 * (librccl.so, the one already mapped into the process if any), so a node that never calls
 * rplgpu_comm_init does not need it.
 *
 * Layout of an exchange (all device memory, no host round trip, sizes travel on the device):
 *   every rank contributes a SLOT of `slot_points` points (16 B each: x, y, z, intensity) — its
 *   arena, of which the first `count` points are valid — and a META block of
 *   rplgpu_cloud_meta_words(max_scans) 32-bit words:
 *       [0..1] count (u64)   [2] number of local scans B   [3] flags (bit 0: count was clamped
 *       to slot_points)   then B x { scan_start lo, scan_start hi, n_points }
 *   written by rplgpu_pack_cloud_meta_dev from the outputs of rplgpu_cloud_arena_dev.
 *   After the gather rank r's slot is d_points_all + r*slot_points*4 floats and its meta block
 *   d_meta_all + r*meta_words; rplgpu_unpack_gathered_dev turns both into ONE contiguous cloud
 *   (rank order, scans in each rank's arena order) with per-scan start / count tables, the
 *   input of rplgpu_transform_clouds_dev / rplgpu_fused_cloud_msg_dev / rplgpu_cloud_msgs_dev.
 * A slot too small for a rank's cloud truncates that cloud (flag bit 0, status
 * RPLGPU_SCAN_OUT_TRUNCATED): size slots from the previous step's counts plus head room.
 *
 * Streams: the collective runs on the handle's EXCHANGE stream (owned by the handle), ordered
 * after everything queued on the handle's main stream at the time of the call; later work on the
 * main stream does not wait for it — that is the overlap: voxelise chunk k + 1 into the other
 * arena while chunk k is on the links.  rplgpu_comm_fence makes the main stream wait for the
 * last exchange (no host synchronisation); rplgpu_synchronize waits for both.
 */
#ifndef RPLGPU_COMM_H_
#define RPLGPU_COMM_H_

#include "rplgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RPLGPU_COMM_ID_BYTES 128

/* rank 0: a fresh RCCL unique id (ncclGetUniqueId); the caller carries it to the other ranks
 * (any channel: a file, MPI, a ROS parameter, torch.distributed ...). */
int32_t rplgpu_comm_unique_id(uint8_t id[RPLGPU_COMM_ID_BYTES]);
/* Collective: every rank calls it with the same id.  RPLGPU_ERR_NO_DEVICE when RCCL cannot be
 * loaded.  world == 1 is allowed (the exchange degenerates to a copy through RCCL). */
int32_t rplgpu_comm_init(rplgpu_handle_t h, int32_t rank, int32_t world,
                         const uint8_t id[RPLGPU_COMM_ID_BYTES]);
int32_t rplgpu_comm_destroy(rplgpu_handle_t h); /* also done by rplgpu_destroy */
/* The communicator as RCCL sees it (ncclCommCount / ncclCommUserRank): *world = 0 when the handle
 * has none.  A launcher checks it against the number of ranks it meant to start. */
int32_t rplgpu_comm_size(rplgpu_handle_t h, int32_t *world, int32_t *rank);

uint32_t rplgpu_cloud_meta_words(uint32_t max_scans); /* 4 + 3 * max_scans */
/* Device-side META block of this rank's arena (see above), asynchronous on the main stream.
 * d_cursor / d_scan_start / d_n_points: outputs of rplgpu_cloud_arena_dev for B scans;
 * slot_points: what the rank will contribute (count is clamped to it). */
int32_t rplgpu_pack_cloud_meta_dev(rplgpu_handle_t h, const uint64_t *d_cursor,
                                   const uint64_t *d_scan_start, const uint32_t *d_n_points,
                                   uint32_t B, uint64_t slot_points, uint32_t max_scans,
                                   uint32_t *d_meta);
/* All-gather of slots and meta blocks (two RCCL all-gathers in one group) on the exchange stream.
 * d_points_all: world * slot_points * 4 floats; d_meta_all: world * meta_words words. */
int32_t rplgpu_allgather_clouds_dev(rplgpu_handle_t h, const float *d_points_local,
                                    uint64_t slot_points, const uint32_t *d_meta_local,
                                    uint32_t meta_words, float *d_points_all,
                                    uint32_t *d_meta_all);
/* Gather to ONE rank ("for C5 a gather-to-root + per-sensor rigid transform
 * suffices" — the fused message of the eight sensors is built on one GPU; the node only ever
 * publishes the identity base_link -> frame_id, /root/reference src/rplidar_node.cpp:183-197):
 * grouped ncclSend / ncclRecv on the exchange stream.  Rank `root` receives every rank's slot and
 * META block into the layout of the all-gather (rank r's slot at d_points_all +
 * r * slot_points * point_floats floats, its META block at d_meta_all + r * meta_words), its own
 * by a device copy (none when d_points_local already IS its slot of the receive buffer); the other
 * ranks only send, d_points_all / d_meta_all may be NULL there.  A link then carries ONE slot, into
 * the root, where the all-gather puts one slot on every link in every direction and the whole
 * cloud in every rank's memory.  point_floats: 4 (16-byte points) or 3 (the compact slots below).
 * rplgpu_unpack_gathered_dev / _xyi_dev on the root afterwards, as behind the all-gather.
 * Ordering and fences exactly as rplgpu_allgather_clouds_dev. */
int32_t rplgpu_gather_clouds_dev(rplgpu_handle_t h, int32_t root, const float *d_points_local,
                                 uint64_t slot_points, uint32_t point_floats,
                                 const uint32_t *d_meta_local, uint32_t meta_words,
                                 float *d_points_all, uint32_t *d_meta_all);
/* main stream waits (on the device, no host synchronisation) for the last exchange enqueued ... */
int32_t rplgpu_comm_fence(rplgpu_handle_t h);
/* ... or for the one `lag` exchanges before it (0 <= lag <= 3): with two arenas used in turn,
 * `rplgpu_comm_fence_lag(h, 1)` ahead of the kernel that refills an arena waits exactly for the
 * gather that still reads it, while the gather enqueued last keeps overlapping that kernel. */
int32_t rplgpu_comm_fence_lag(rplgpu_handle_t h, uint32_t lag);
/* Gathered slots -> one contiguous cloud.  d_packed: room for the sum of the counts (at most
 * world * slot_points points); d_total: 1 x u64 (total points, device); d_scan_start_all /
 * d_n_points_all: world * max_scans entries, rank-major (entries of missing scans: 0 / 0);
 * d_status (optional, world words): RPLGPU_SCAN_OUT_TRUNCATED for a rank whose cloud was cut.
 * Runs on the main stream (call rplgpu_comm_fence first when the gather was just enqueued).
 * No communicator needed: `world` is a parameter (the gloo twin of the tests uses that). */
int32_t rplgpu_unpack_gathered_dev(rplgpu_handle_t h, const float *d_points_all,
                                   uint64_t slot_points, const uint32_t *d_meta_all,
                                   uint32_t meta_words, uint32_t world, uint32_t max_scans,
                                   float *d_packed, uint64_t *d_total,
                                   uint64_t *d_scan_start_all, uint32_t *d_n_points_all,
                                   uint32_t *d_status);

/* ---- the compact exchange: 12-byte points (x, y, intensity) --------------------------------
 * z is 0.0 for every point this path produces (a planar sensor: E2 sets z = 0, the E6 / E8
 * transforms are planar), so it need not travel: a quarter fewer bytes per link.
 *   rplgpu_pack_cloud_xyi_dev        arena (16-byte points) -> this rank's 12-byte slot; the first
 *                                    min(*d_cursor, slot_points) points (device-side count)
 *   rplgpu_allgather_clouds_xyi_dev  as rplgpu_allgather_clouds_dev for slots of slot_points * 3
 *                                    floats (META blocks unchanged)
 *   rplgpu_unpack_gathered_xyi_dev   gathered 12-byte slots -> ONE contiguous cloud of 16-byte
 *                                    points (z = 0 put back) + the per-scan tables */
int32_t rplgpu_pack_cloud_xyi_dev(rplgpu_handle_t h, const float *d_arena, const uint64_t *d_cursor,
                                  uint64_t slot_points, float *d_slot);
/* rplgpu_cloud_arena_dev (include/rplgpu.h) writing the 12-byte points ITSELF: d_slot is this rank's
 * slot of the receive buffer (slot_points x 3 floats), filled by the voxel kernel directly — no
 * 16-byte arena, no compaction pass in front of the all-gather.  d_cursor / d_scan_start /
 * d_n_points / d_status as rplgpu_cloud_arena_dev, in points; a cloud that outgrows the slot is cut
 * and flagged (RPLGPU_SCAN_OUT_TRUNCATED).  The optional cell-key output still has one word per point. */
int32_t rplgpu_cloud_arena_xyi_dev(rplgpu_handle_t h, const rplgpu_node_t *d_nodes,
                                   uint32_t n_stride, const uint32_t *d_n_per_scan, uint32_t B,
                                   const rplgpu_params_t *p, float *d_slot, uint64_t slot_points,
                                   uint64_t *d_cursor, uint64_t *d_scan_start, uint32_t *d_n_points,
                                   uint32_t *d_status);
int32_t rplgpu_allgather_clouds_xyi_dev(rplgpu_handle_t h, const float *d_slot_local,
                                        uint64_t slot_points, const uint32_t *d_meta_local,
                                        uint32_t meta_words, float *d_slots_all,
                                        uint32_t *d_meta_all);
int32_t rplgpu_unpack_gathered_xyi_dev(rplgpu_handle_t h, const float *d_slots_all,
                                       uint64_t slot_points, const uint32_t *d_meta_all,
                                       uint32_t meta_words, uint32_t world, uint32_t max_scans,
                                       float *d_packed, uint64_t *d_total,
                                       uint64_t *d_scan_start_all, uint32_t *d_n_points_all,
                                       uint32_t *d_status);

/* ---- host twins of the layout functions ------------------------------------------------------
 * The same rules as the device kernels (one source: csrc/rpl_comm_layout.hpp), plain host loops,
 * no device and no handle: for transports other than RCCL and for world-size > 1 tests without
 * GPUs (tests/test_sharding_gloo.py moves the slots with gloo and calls these).
 * point_floats: 4 (x, y, z, intensity slots) or 3 (the compact slots above). */
int32_t rplgpu_pack_cloud_meta_host(uint64_t cursor, const uint64_t *scan_start,
                                    const uint32_t *n_points, uint32_t B, uint64_t slot_points,
                                    uint32_t max_scans, uint32_t *meta);
int32_t rplgpu_pack_cloud_xyi_host(const float *arena, uint64_t cursor, uint64_t slot_points,
                                   float *slot);
int32_t rplgpu_unpack_gathered_host(const float *points_all, uint64_t slot_points,
                                    uint32_t point_floats, const uint32_t *meta_all,
                                    uint32_t meta_words, uint32_t world, uint32_t max_scans,
                                    float *packed, uint64_t *total, uint64_t *scan_start_all,
                                    uint32_t *n_points_all, uint32_t *status);

#ifdef __cplusplus
}
#endif
#endif /* RPLGPU_COMM_H_ */
