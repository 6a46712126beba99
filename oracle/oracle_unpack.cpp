/*
 * oracle_unpack.cpp — CPU restatement of the reference SDK's sample-data unpackers and scan
 * assembler (the step just before the hot path).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PINNED: every function here is checked bit for bit
 * against the reference's own code compiled from /root/reference (oracle/_ref/libunpackref.so,
 * oracle/ref_unpack_shim.cpp) and against golden vectors generated from it (tests/golden/).
 *
 * The reference decodes with one byte-at-a-time state machine per answer type
 * (src/sdk/src/dataunpacker/unpacker/handler_*.cpp, `onData`).  The restatement separates the two
 * things those state machines do — exactly the split the product makes:
 *   1. FRAMING  (orc_frame_stream): which byte ranges become frames, and whether bytes were
 *      rejected between two frames (that clears the "previous capsule ready" latch);
 *   2. DECODING (orc_unpack_frames): checksum / CRC per frame, the inter-capsule latch, the
 *      per-sample integer arithmetic, the sync-bit filter and distance smoothing recurrences.
 * Timestamps are not restated: the reference takes them from the wall clock (getus(),
 * dataunpacker.cpp:164-166), they are not a function of the input bytes.
 *
 * All citations are relative to /root/reference/src/sdk/.
 */
#include <cstdlib>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

/* frame sizes: include/sl_lidar_cmd.h:189-286 (packed structs) */
size_t frame_size(uint8_t ans) {
  switch (ans) {
    case ORC_ANS_MEASUREMENT: return 5;        /* :189-194 */
    case ORC_ANS_CAPSULED: return 84;          /* :215-221  2 + 2 + 16*5 */
    case ORC_ANS_HQ: return 781;               /* :280-286  1 + 8 + 96*8 + 4 */
    case ORC_ANS_CAPSULED_ULTRA: return 132;   /* :264-270  2 + 2 + 32*4 */
    case ORC_ANS_DENSE_CAPSULED: return 84;    /* :228-234  2 + 2 + 40*2 */
    case ORC_ANS_ULTRA_DENSE_CAPSULED: return 170; /* :242-249  2 + 4 + 2 + 2 + 32*5 */
  }
  return 0;
}

inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* sl_crc.cpp:36-101: reflected CRC-32 (poly 0x04C11DB7), init 0xFFFFFFFF, the input is zero
 * padded by 4 - (len & 3) bytes (so a multiple of four gets FOUR pad bytes), final xor. */
uint32_t crc32_padded(const uint8_t *p, size_t len) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int j = 0; j < 8; ++j) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
      table[i] = c;
    }
    init = true;
  }
  uint32_t crc = 0xFFFFFFFFu;
  for (size_t i = 0; i < len; ++i) crc = (crc >> 8) ^ table[(uint8_t)(crc ^ p[i])];
  const size_t pad = 4 - (len & 3);
  for (size_t i = 0; i < pad; ++i) crc = (crc >> 8) ^ table[(uint8_t)crc];
  return crc ^ 0xFFFFFFFFu;
}

/* handler_capsules.cpp:422-458 _varbitscale_decode */
uint32_t varbitscale(uint32_t scaled, uint32_t &lvl) {
  static const uint32_t base[5] = {3328, 1792, 1280, 512, 0};        /* *_DEST_VAL */
  static const uint32_t level[5] = {4, 3, 2, 1, 0};
  static const uint32_t target[5] = {1u << 14, 1u << 12, 1u << 11, 1u << 9, 0}; /* *_SRC_BIT */
  for (int i = 0; i < 5; ++i) {
    int remain = (int)scaled - (int)base[i];
    if (remain >= 0) {
      lvl = level[i];
      return target[i] + ((uint32_t)remain << lvl);
    }
  }
  return 0;
}

orc_node_t mk(int angle_q6, uint32_t dist_q2, uint8_t quality, int sync) {
  /* the common tail of every capsule decoder, e.g. handler_capsules.cpp:246-257 */
  if (angle_q6 < 0) angle_q6 += (360 << 6);
  if (angle_q6 >= (360 << 6)) angle_q6 -= (360 << 6);
  orc_node_t n;
  n.flag = (uint8_t)(sync | ((!sync) << 1));
  n.quality = quality;
  n.angle_z_q14 = (uint16_t)((angle_q6 << 8) / 90);
  n.dist_mm_q2 = dist_q2;
  return n;
}

}  // namespace

extern "C" size_t orc_frame_size(uint8_t ans_type) { return frame_size(ans_type); }

/* ---- 1. framing: the position-0 / position-1 rules of every onData loop
 * (handler_capsules.cpp:107-135 and its three siblings :326-354, :641-669, :853-881;
 * handler_hqnode.cpp:99-113; handler_normalnode.cpp:88-112).  A byte rejected at position 0 is
 * skipped; a byte rejected at position 1 is skipped too and the machine returns to position 0
 * (the rejected byte is NOT re-examined).  For the capsule types either rejection clears the
 * previous-capsule latch: gap[k] = 1 when that happened after frame k-1 was completed and
 * before frame k started.  A trailing partial frame is dropped (it stays in the reference's
 * cache). */
extern "C" size_t orc_frame_stream(uint8_t ans, const uint8_t *bytes, size_t nbytes,
                                   uint32_t *frame_off, uint8_t *gap, size_t cap) {
  const size_t S = frame_size(ans);
  if (!S) return 0;
  size_t nf = 0, pos = 0, start = 0;
  uint8_t g = 0;
  for (size_t i = 0; i < nbytes; ++i) {
    const uint8_t b = bytes[i];
    if (pos == 0) {
      bool ok;
      if (ans == ORC_ANS_HQ) ok = (b == 0xA5);
      else if (ans == ORC_ANS_MEASUREMENT) ok = (((b >> 1) ^ b) & 1) != 0;
      else ok = (b >> 4) == 0xA;
      if (!ok) { g = 1; continue; }
      start = i;
      pos = 1;
      continue;
    }
    if (pos == 1 && ans != ORC_ANS_HQ) {
      const bool ok = (ans == ORC_ANS_MEASUREMENT) ? ((b & 1) != 0) : ((b >> 4) == 0x5);
      if (!ok) { pos = 0; g = 1; continue; }
    }
    if (++pos == S) {
      if (nf < cap) {
        frame_off[nf] = (uint32_t)start;
        if (gap) gap[nf] = g;
      }
      ++nf;
      pos = 0;
      g = 0;
    }
  }
  return nf;
}

/* ---- 2. decoding framed data.  `st` carries the state that outlives a capsule pair:
 * last_sync_bit  — dense: the function-level `static int lastNodeSyncBit`
 *                  (handler_capsules.cpp:738), ultra-dense: _last_node_sync_bit (:1044);
 * last_dist_q2   — ultra-dense _last_dist_q2 (:999-1003, :1020). */
extern "C" size_t orc_unpack_frames(uint8_t ans, const uint8_t *bytes, const uint32_t *frame_off,
                                    const uint8_t *gap, size_t nframes,
                                    uint32_t sample_duration_us, orc_unpack_state_t *st,
                                    orc_node_t *out, size_t cap, uint32_t *reset_at,
                                    size_t reset_cap, size_t *n_reset_out,
                                    uint32_t *n_checksum_err) {
  const size_t S = frame_size(ans);
  size_t n = 0, n_reset = 0;
  uint32_t n_err = 0;
  auto emit = [&](const orc_node_t &nd) {
    if (n < cap) out[n] = nd;
    ++n;
  };
  bool prev_rdy = false;
  const uint8_t *prev = nullptr;
  int last_sync = st ? st->last_sync_bit : 0;
  int last_dist = st ? st->last_dist_q2 : 0;

  for (size_t k = 0; k < nframes; ++k) {
    const uint8_t *f = bytes + frame_off[k];
    if (ans == ORC_ANS_MEASUREMENT) { /* handler_normalnode.cpp:113-136 */
      const uint16_t angle_q6_checkbit = rd16(f + 1);
      orc_node_t nd;
      nd.angle_z_q14 = (uint16_t)((((int)angle_q6_checkbit >> 1) << 8) / 90);
      nd.dist_mm_q2 = rd16(f + 3);
      nd.flag = (uint8_t)(f[0] & 1);
      nd.quality = (uint8_t)((f[0] >> 2) << 2);
      emit(nd);
      continue;
    }
    if (ans == ORC_ANS_HQ) { /* handler_hqnode.cpp:120-172 (CONF_NO_BOOST_CRC_SUPPORT path) */
      if (crc32_padded(f, S - 4) == rd32(f + S - 4)) {
        for (int i = 0; i < 96; ++i) {
          orc_node_t nd;
          memcpy(&nd, f + 9 + 8 * i, 8);
          emit(nd);
        }
      } else {
        ++n_err;
      }
      continue;
    }
    /* the four capsule types share the frame epilogue: handler_capsules.cpp:137-194 etc. */
    if (gap && gap[k]) prev_rdy = false;
    uint8_t x = 0;
    for (size_t i = 2; i < S; ++i) x ^= f[i];
    const uint8_t recv = (uint8_t)((f[0] & 0xF) | (f[1] << 4));
    if (recv != x) {
      prev_rdy = false;
      ++n_err;
      continue;
    }
    const size_t sa_off = (ans == ORC_ANS_ULTRA_DENSE_CAPSULED) ? 8 : 2;
    const uint16_t sa = rd16(f + sa_off);
    if (sa & 0x8000) { /* first capsule of a revolution: drop the cached one, request a reset */
      prev_rdy = false;
      if (n_reset < reset_cap) reset_at[n_reset] = (uint32_t)n;
      ++n_reset;
    }
    if (prev_rdy) {
      const uint16_t psa = rd16(prev + sa_off);
      const int cur_q8 = (sa & 0x7FFF) << 2, prev_q8 = (psa & 0x7FFF) << 2;
      int diff_q8 = cur_q8 - prev_q8;
      if (prev_q8 > cur_q8) diff_q8 += (360 << 8);
      int ang_q16 = prev_q8 << 8;

      if (ans == ORC_ANS_CAPSULED) { /* :206-260 */
        const int inc_q16 = diff_q8 << 3;
        for (int pos = 0; pos < 16; ++pos) {
          const uint8_t *c = prev + 4 + 5 * pos;
          const uint16_t da1 = rd16(c), da2 = rd16(c + 2);
          const uint8_t offs = c[4];
          const int dist[2] = {da1 & 0xFFFC, da2 & 0xFFFC};
          const int aoff[2] = {(offs & 0xF) | ((da1 & 0x3) << 4), (offs >> 4) | ((da2 & 0x3) << 4)};
          for (int j = 0; j < 2; ++j) {
            const int angle_q6 = (ang_q16 - (aoff[j] << 13)) >> 10;
            const int sync = (((ang_q16 + inc_q16) % (360 << 16)) < inc_q16) ? 1 : 0;
            ang_q16 += inc_q16;
            emit(mk(angle_q6, (uint32_t)dist[j], dist[j] ? (0x2F << 2) : 0, sync));
          }
        }
      } else if (ans == ORC_ANS_CAPSULED_ULTRA) { /* :460-577 */
        const int inc_q16 = (diff_q8 << 3) / 3;
        for (int pos = 0; pos < 32; ++pos) {
          const uint32_t cx = rd32(prev + 4 + 4 * pos);
          int major = (int)(cx & 0xFFF);
          int pred1 = ((int)(cx << 10)) >> 22;  /* sign-extended bits 12..21 */
          int pred2 = ((int)cx) >> 22;          /* sign-extended bits 22..31 */
          const uint32_t nx = (pos == 31) ? rd32(f + 4) : rd32(prev + 4 + 4 * (pos + 1));
          int major2 = (int)(nx & 0xFFF);
          uint32_t lvl1 = 0, lvl2 = 0;
          major = (int)varbitscale((uint32_t)major, lvl1);
          major2 = (int)varbitscale((uint32_t)major2, lvl2);
          int base1 = major, base2 = major2;
          if (!major && major2) {
            base1 = major2;
            lvl1 = lvl2;
          }
          int dist[3];
          dist[0] = major << 2;
          if ((uint32_t)pred1 == 0xFFFFFE00u || (uint32_t)pred1 == 0x1FFu) {
            dist[1] = 0;
          } else {
            pred1 = (int)((uint32_t)pred1 << lvl1);
            dist[1] = (int)((uint32_t)(pred1 + base1) << 2);
          }
          if ((uint32_t)pred2 == 0xFFFFFE00u || (uint32_t)pred2 == 0x1FFu) {
            dist[2] = 0;
          } else {
            pred2 = (int)((uint32_t)pred2 << lvl2);
            dist[2] = (int)((uint32_t)(pred2 + base2) << 2);
          }
          for (int j = 0; j < 3; ++j) {
            const int sync = (((ang_q16 + inc_q16) % (360 << 16)) < inc_q16) ? 1 : 0;
            int off_q16 = (int)(7.5 * 3.1415926535 * (1 << 16) / 180.0);
            if (dist[j] >= (50 * 4)) { /* triangulation angle correction :547-553 */
              const int k1 = 98361;
              const int k2 = int(k1 / dist[j]);
              off_q16 = (int)(8 * 3.1415926535 * (1 << 16) / 180) - (k2 << 6) - (k2 * k2 * k2) / 98304;
            }
            const int angle_q6 = (ang_q16 - int(off_q16 * 180 / 3.14159265)) >> 10;
            ang_q16 += inc_q16;
            emit(mk(angle_q6, (uint32_t)dist[j], dist[j] ? (0x2F << 2) : 0, sync));
          }
        }
      } else if (ans == ORC_ANS_DENSE_CAPSULED) { /* :736-791 */
        const int thr_q8 = (int)((360 * 100 * 40 / (1000000 / sample_duration_us)) << 8);
        if (diff_q8 > thr_q8) { /* discard: the latch stays set */
          prev = f;
          continue;
        }
        const int inc_q16 = (diff_q8 << 8) / 40;
        for (int pos = 0; pos < 40; ++pos) {
          const int dist_q2 = (int)rd16(prev + 4 + 2 * pos) << 2;
          const int angle_q6 = ang_q16 >> 10;
          int sync = (((ang_q16 + inc_q16) % (360 << 16)) < (inc_q16 << 1)) ? 1 : 0;
          sync = (sync ^ last_sync) & sync;
          ang_q16 += inc_q16;
          emit(mk(angle_q6, (uint32_t)dist_q2, dist_q2 ? (0x2F << 2) : 0, sync));
          last_sync = sync;
        }
      } else { /* ORC_ANS_ULTRA_DENSE_CAPSULED :951-1047 */
        const int thr_q8 = (int)((360 * 100 * 32 / (1000000 / sample_duration_us)) << 8);
        if (diff_q8 > thr_q8) {
          prev = f;
          continue;
        }
        const int inc_q16 = (diff_q8 << 8) / 64;
        for (int pos = 0; pos < 64; ++pos) {
          const uint8_t *c = prev + 10 + 5 * (pos >> 1);
          uint32_t qds;
          if (!(pos & 1)) qds = rd16(c) | ((uint32_t)(c[4] & 0x0F) << 16);
          else qds = rd16(c + 2) | ((uint32_t)(c[4] >> 4) << 16);
          const uint8_t scale = qds & 0x3;
          uint8_t quality = 0;
          int dist_q2 = 0;
          switch (scale) {
            case 0:
              quality = (uint8_t)(qds >> 12);
              dist_q2 = (int)(qds & 0xFFC) * 2;
              if (last_dist) {
                if (abs(dist_q2 - last_dist) <= 8) dist_q2 = (dist_q2 + last_dist) >> 1;
              }
              break;
            case 1:
              quality = (uint8_t)((qds >> 13) << 1);
              dist_q2 = (int)(qds & 0x1FFC) * 3 + (2046 << 2);
              break;
            case 2:
              quality = (uint8_t)((qds >> 14) << 2);
              dist_q2 = (int)(qds & 0x3FFC) * 4 + (8187 << 2);
              break;
            case 3:
              quality = (uint8_t)((qds >> 15) << 3);
              dist_q2 = (int)(qds & 0x7FFC) * 5 + (24567 << 2);
              break;
          }
          last_dist = dist_q2;
          const int angle_q6 = ang_q16 >> 10;
          int sync = (((ang_q16 + inc_q16) % (360 << 16)) < (inc_q16 << 1)) ? 1 : 0;
          sync = (sync ^ last_sync) & sync;
          ang_q16 += inc_q16;
          emit(mk(angle_q6, (uint32_t)dist_q2, quality, sync));
          last_sync = sync;
        }
      }
    }
    prev = f;
    prev_rdy = true;
  }
  if (st) {
    st->last_sync_bit = last_sync;
    st->last_dist_q2 = last_dist;
  }
  if (n_reset_out) *n_reset_out = n_reset;
  if (n_checksum_err) *n_checksum_err = n_err;
  return n;
}

/* framing + decoding of a raw byte stream = what a fresh reference unpacker publishes */
extern "C" size_t orc_unpack(uint8_t ans, const uint8_t *bytes, size_t nbytes,
                             uint32_t sample_duration_us, orc_unpack_state_t *st, orc_node_t *out,
                             size_t cap, uint32_t *reset_at, size_t reset_cap, size_t *n_reset,
                             uint32_t *n_checksum_err) {
  const size_t S = frame_size(ans);
  if (!S) return 0;
  const size_t maxf = nbytes / S + 1;
  std::vector<uint32_t> off(maxf);
  std::vector<uint8_t> gap(maxf);
  const size_t nf = orc_frame_stream(ans, bytes, nbytes, off.data(), gap.data(), maxf);
  return orc_unpack_frames(ans, bytes, off.data(), gap.data(), nf, sample_duration_us, st, out,
                           cap, reset_at, reset_cap, n_reset, n_checksum_err);
}

/* ---- scan assembly: ScanDataHolder<T>::pushScanNodeData / rewindCurrentScanData
 * (src/sl_lidar_driver.cpp:272-315) as driven by SlamtecLidarDriver::onHQNodeDecoded /
 * onHQNodeScanResetReq (:1645-1653).  A node with flag bit 0 closes the scan being built (if it
 * holds anything) and opens the next one; nodes before the first sync node, or after a rewind
 * until the next sync node, are discarded; a scan that reaches max_count nodes keeps
 * overwriting its last slot (:301-304).  Completed scans are written back to back. */
extern "C" size_t orc_segment(const orc_node_t *nodes, size_t n, const uint32_t *reset_at,
                              size_t n_reset, size_t max_count, orc_node_t *out, size_t out_cap,
                              uint32_t *scan_off, size_t scan_cap) {
  std::vector<orc_node_t> cur;
  size_t nscans = 0, wr = 0, r = 0;
  if (scan_cap) scan_off[0] = 0;
  for (size_t i = 0; i <= n; ++i) {
    while (r < n_reset && reset_at[r] == i) {
      cur.clear();
      ++r;
    }
    if (i == n) break;
    const orc_node_t &nd = nodes[i];
    if (nd.flag & 1) {
      if (!cur.empty()) {
        for (const orc_node_t &c : cur) {
          if (wr < out_cap) out[wr] = c;
          ++wr;
        }
        ++nscans;
        if (nscans < scan_cap) scan_off[nscans] = (uint32_t)wr;
        cur.clear();
      }
    } else if (cur.empty()) {
      continue;
    }
    if (cur.size() >= max_count) cur.back() = nd;
    else cur.push_back(nd);
  }
  return nscans;
}
