/*
 * ref_unpack_shim.cpp — thin extern "C" door into the REAL reference SDK's sample-data
 * unpackers and scan assembler.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  This file contains no algorithm.  It
 *   - instantiates sl::internal::LIDARSampleDataUnpacker exactly as the reference driver does
 *     (src/sdk/src/sl_lidar_driver.cpp: CreateInstance(listener) + updateUnpackerContext(TIMING)
 *     + enable(), interface src/sdk/src/dataunpacker/dataunpacker.h:48-88), feeds it a recorded
 *     byte stream through onSampleData(ansType, buf, len) and records what it publishes through
 *     the listener interface (onHQNodeDecoded / onHQNodeScanResetReq / onDecodingError);
 *   - instantiates the reference's own ScanDataHolder<node_hq> (src/sdk/src/sl_lidar_driver.cpp:
 *     236-360) — the class is private to that translation unit, so this shim #includes the
 *     translation unit itself, unmodified, where it lies — and replays a decoded node stream
 *     into pushScanNodeData / rewindCurrentScanData the way SlamtecLidarDriver::onHQNodeDecoded
 *     and ::onHQNodeScanResetReq do (:1645-1653), harvesting every completed scan.
 * Compiled by oracle/Makefile into oracle/_ref/libunpackref.so together with the other SDK
 * sources (git-ignored, never copied into the repo).
 */
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include "sl_lidar_driver.cpp"  // the reference translation unit itself (for ScanDataHolder)

namespace {

typedef sl_lidar_response_measurement_node_hq_t node_t;

struct Collector : public sl::internal::LIDARSampleDataListener {
  std::vector<node_t> nodes;
  std::vector<uint32_t> reset_at;  // number of nodes published before each reset request
  uint32_t n_err = 0;
  uint32_t n_encoder_reset = 0;
  void onHQNodeScanResetReq() override { reset_at.push_back((uint32_t)nodes.size()); }
  void onHQNodeDecoded(_u64, const rplidar_response_measurement_node_hq_t *node) override {
    nodes.push_back(*node);
  }
  void onDecodingError(int errMsg, _u8, const void *, size_t) override {
    if (errMsg == sl::internal::LIDARSampleDataUnpacker::ERR_EVENT_ON_EXP_CHECKSUM_ERR) ++n_err;
    if (errMsg == sl::internal::LIDARSampleDataUnpacker::ERR_EVENT_ON_EXP_ENCODER_RESET)
      ++n_encoder_reset;
  }
};

}  // namespace

/* Feed `nbytes` of a recorded answer stream of type `ans_type` to a FRESH unpacker in pieces of
 * `chunk` bytes (0 = all at once).  Returns the number of nodes published (may exceed `cap`;
 * only the first `cap` are stored).  reset_at[i] = number of nodes published before the i-th
 * scan-reset request. */
extern "C" size_t ref_unpack(uint8_t ans_type, const uint8_t *bytes, size_t nbytes, size_t chunk,
                             uint32_t sample_duration_us, void *out_nodes, size_t cap,
                             uint32_t *reset_at, size_t reset_cap, size_t *n_reset,
                             uint32_t *n_checksum_err, uint32_t *n_encoder_reset) {
  Collector col;
  sl::internal::LIDARSampleDataUnpacker *u =
      sl::internal::LIDARSampleDataUnpacker::CreateInstance(col);
  sl::SlamtecLidarTimingDesc timing;
  memset(&timing, 0, sizeof(timing));
  timing.sample_duration_uS = sample_duration_us;
  timing.native_baudrate = 0;
  timing.linkage_delay_uS = 0;
  timing.native_interface_type = sl::LIDAR_INTERFACE_UART;
  timing.native_timestamp_support = false;
  u->updateUnpackerContext(sl::internal::LIDARSampleDataUnpacker::UNPACKER_CONTEXT_TYPE_LIDAR_TIMING,
                           &timing, sizeof(timing));
  u->enable();
  if (chunk == 0) chunk = nbytes ? nbytes : 1;
  for (size_t pos = 0; pos < nbytes; pos += chunk) {
    size_t len = nbytes - pos < chunk ? nbytes - pos : chunk;
    u->onSampleData(ans_type, bytes + pos, len);
  }
  sl::internal::LIDARSampleDataUnpacker::ReleaseInstance(u);
  size_t n = col.nodes.size();
  if (out_nodes && cap) memcpy(out_nodes, col.nodes.data(), (n < cap ? n : cap) * sizeof(node_t));
  if (n_reset) *n_reset = col.reset_at.size();
  for (size_t i = 0; i < col.reset_at.size() && i < reset_cap; ++i) reset_at[i] = col.reset_at[i];
  if (n_checksum_err) *n_checksum_err = col.n_err;
  if (n_encoder_reset) *n_encoder_reset = col.n_encoder_reset;
  return n;
}

/* Replay a decoded node stream (with the scan-reset requests at node indices reset_at[],
 * ascending, "before node i") into the reference's ScanDataHolder and harvest every scan it
 * completes.  out_nodes receives the completed scans back to back, scan_off[s] .. scan_off[s+1]
 * delimit scan s.  Returns the number of completed scans. */
extern "C" size_t ref_segment(const void *nodes_in, size_t n, const uint32_t *reset_at,
                              size_t n_reset, size_t max_count, void *out_nodes, size_t out_cap,
                              uint32_t *scan_off, size_t scan_cap) {
  const node_t *nodes = reinterpret_cast<const node_t *>(nodes_in);
  node_t *out = reinterpret_cast<node_t *>(out_nodes);
  sl::ScanDataHolder<node_t> holder(max_count);
  size_t nscans = 0, wr = 0, r = 0;
  if (scan_cap) scan_off[0] = 0;
  auto harvest = [&]() {
    if (!holder.checkNewScanSignalAndReset()) return;
    std::vector<node_t> *scan = holder.waitAndLockAvailableScan(0);
    if (!scan) return;
    for (size_t i = 0; i < scan->size(); ++i) {
      if (wr < out_cap) out[wr] = (*scan)[i];
      ++wr;
    }
    holder.unlockScan(scan);
    ++nscans;
    if (nscans < scan_cap) scan_off[nscans] = (uint32_t)wr;
  };
  for (size_t i = 0; i <= n; ++i) {
    while (r < n_reset && reset_at[r] == i) {
      holder.rewindCurrentScanData();
      ++r;
    }
    if (i == n) break;
    holder.pushScanNodeData(0, &nodes[i]);
    harvest();
  }
  return nscans;
}

extern "C" uint32_t ref_unpack_node_size(void) { return (uint32_t)sizeof(node_t); }
