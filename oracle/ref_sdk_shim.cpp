/*
 * ref_sdk_shim.cpp — thin extern "C" door into the REAL reference SDK.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  This file contains no algorithm:
 * it instantiates the vendored Slamtec driver object exactly as the reference
 * wrapper does (src/lidar_driver_wrapper.cpp:97-105 uses sl::createLidarDriver)
 * and forwards to the virtual ILidarDriver::ascendScanData
 * (src/sdk/include/sl_lidar_driver.h:477, body src/sdk/src/sl_lidar_driver.cpp:957-960),
 * which needs no connected device.  It is compiled together with the SDK
 * sources *where they lie* under /root/reference/src/sdk by oracle/Makefile
 * into oracle/_ref/libslref.so (git-ignored, never copied into the repo).
 */
#include <cstddef>
#include <cstdint>

#include "sl_lidar.h"
#include "sl_lidar_driver.h"

static sl::ILidarDriver *g_drv = nullptr;

extern "C" uint32_t ref_ascend(void *nodes, size_t count) {
  if (!g_drv) {
    auto r = sl::createLidarDriver();
    g_drv = *r;
  }
  static_assert(sizeof(sl_lidar_response_measurement_node_hq_t) == 8, "packed 8-byte node");
  return (uint32_t)g_drv->ascendScanData(
      reinterpret_cast<sl_lidar_response_measurement_node_hq_t *>(nodes), count);
}

extern "C" uint32_t ref_node_size(void) {
  return (uint32_t)sizeof(sl_lidar_response_measurement_node_hq_t);
}
