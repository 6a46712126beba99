// rpl_decode.hip — the step BEFORE the hot path on gfx950:
//   k_decode<ANS>   recorded answer streams (already framed) -> measurement_node_hq nodes,
//                   one workgroup per stream, bit-exact with the reference SDK's unpackers
//                   (src/sdk/src/dataunpacker/unpacker/handler_capsules.cpp / handler_hqnode.cpp /
//                   handler_normalnode.cpp; integer arithmetic only);
//   k_segment       node stream + scan-reset requests -> completed scans
//                   (ScanDataHolder::pushScanNodeData / rewindCurrentScanData,
//                   src/sdk/src/sl_lidar_driver.cpp:272-315);
//   k_scans_to_batch completed scans of all streams -> the fixed-stride scan batch the
//                   ascend / LaserScan / cloud kernels consume.
//
// The reference decodes a stream with a byte-at-a-time state machine that carries three kinds
// of state.  They are made data-parallel as follows (citations: handler_capsules.cpp):
//   * the "previous capsule ready" latch (:137-194): capsule k-1 is published while capsule k
//     is handled iff  valid(k-1) && valid(k) && !gap(k) && !revolution_start(k)
//     — a pure function of two neighbouring frames (valid = checksum ok; gap = bytes were
//     rejected between the frames, known from framing);
//   * the dense / ultra-dense sync-bit filter  s_i = r_i & ~s_{i-1}  (:770-771, :1025-1026):
//     inside a run of raw sync bits the filtered bits alternate starting with 1, so
//     s_i = r_i & (distance to the run's start is even); the raw bits go to an LDS bit set and
//     each node scans back over its run with word operations;
//   * the ultra-dense distance smoothing  d_i <- (d_i + d'_{i-1}) >> 1  (:997-1003): a genuine
//     recurrence, but a smoothed value stays within +-4 of its raw value, so the carried value
//     is one of 9 states relative to its node and every node is a 9 -> 9 map; maps compose
//     associatively, so a block scan over per-thread segment maps resolves the recurrence
//     exactly (phase P5).
// Timestamps are not produced: the reference stamps nodes with the wall clock of the decoding
// host (dataunpacker.cpp:164-166), not with anything in the stream.
#include <type_traits>

#include <mutex>
#include <set>
#include <utility>

#include "rpl_device.hpp"
#include "rpl_launch.hpp"

namespace rpl {

constexpr int kDecBlock = 256;
#ifndef RPL_DEC_MAXFRAMES
#define RPL_DEC_MAXFRAMES 2048
#endif
constexpr uint32_t kDecMaxFrames = RPL_DEC_MAXFRAMES;     // frames of one stream per call (LDS frame table)
constexpr uint32_t kUdMaxFrames = 512;       // ultra-dense: 64 nodes each -> 32768 LDS slots
// LDS is sized per answer type so that several streams share a CU (a 32 000-sample DenseBoost
// scan is 800 frames): dense 25 KiB, express / ultra 16 KiB, HQ 34 KiB, ultra-dense 24 KiB
template <int ANS>
struct DecCfg {
  static constexpr bool kFiltered =
      ANS == RPLGPU_ANS_DENSE_CAPSULED || ANS == RPLGPU_ANS_ULTRA_DENSE_CAPSULED;
  // legacy 5-byte nodes carry no inter-frame state and always publish: no frame table at all
  static constexpr bool kTable = ANS != RPLGPU_ANS_MEASUREMENT;
  static constexpr uint32_t kMaxFrames = !kTable ? 0x7FFFFFFFu
                                         : ANS == RPLGPU_ANS_ULTRA_DENSE_CAPSULED ? kUdMaxFrames
                                                                                  : kDecMaxFrames;
  static constexpr uint32_t kTableSlots = kTable ? kMaxFrames : 1u;
  // u64 words of raw sync bits: one bit per node the stream can publish in one call
  static constexpr uint32_t kRawBitWords =
      !kFiltered ? 1u : ANS == RPLGPU_ANS_DENSE_CAPSULED ? kDecMaxFrames * 40u / 64u : kUdMaxFrames;
  // ultra-dense: nodes are decoded and smoothed in chunks of kUdChunk (the smoothing pass needs the
  // raw distances of a chunk in LDS: 16 KiB instead of 64 KiB for the whole stream, i.e. six
  // instead of two workgroups per CU)
  static constexpr uint32_t kUdChunk = 8192u;  // (1024 .. 6144 measured in round 4: equal or slower)
  static constexpr uint32_t kSmoothSlots = ANS == RPLGPU_ANS_ULTRA_DENSE_CAPSULED ? kUdChunk : 1u;
  static constexpr uint32_t kCrcWords = ANS == RPLGPU_ANS_HQ ? 1024u : 1u;
  // nodes one lane decodes in one go (all of ONE frame): per-frame arithmetic once per group,
  // payload in one or two wide loads, nodes out in 16-byte stores
  static constexpr uint32_t kGroup = ANS == RPLGPU_ANS_MEASUREMENT ? 1u
                                     : ANS == RPLGPU_ANS_CAPSULED_ULTRA ? 6u
                                                                        : 4u;
  // ultra: the triangulation angle correction is a function of k2 = 98361 / dist (0..491) alone
  static constexpr uint32_t kCorrSlots = ANS == RPLGPU_ANS_CAPSULED_ULTRA ? 496u : 1u;
  // the four capsule types: the sync bit of a node follows from the capsule HEADERS alone, so
  // scan boundaries are known before the payload is decoded and the nodes of completed scans
  // can be written straight into batch slots (k_decode<..., FUSE = true>); up to kFuseSyn sync
  // nodes and kFuseRst reset requests per stream and call, else the unfused path takes over
  static constexpr bool kFusable = ANS == RPLGPU_ANS_CAPSULED || ANS == RPLGPU_ANS_CAPSULED_ULTRA ||
                                   ANS == RPLGPU_ANS_DENSE_CAPSULED ||
                                   ANS == RPLGPU_ANS_ULTRA_DENSE_CAPSULED;
  static constexpr uint32_t kFuseSyn = kFusable ? 256u : 1u, kFuseRst = kFusable ? 64u : 1u;
  static constexpr uint32_t kStageWords = ANS == RPLGPU_ANS_HQ ? (kDecBlock / 64) * 64 * 17 : 1u;
};

__device__ __forceinline__ uint32_t ld8(const uint8_t *p) { return p[0]; }
// frames start at any byte: the fields are read with one (possibly unaligned) load each -- the
// target runs in unaligned access mode, so these become single global_load_ushort / _dword
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) {
  uint16_t v;
  __builtin_memcpy(&v, p, 2);
  return v;
}
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__device__ __forceinline__ uint64_t ld64(const uint8_t *p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
__device__ __forceinline__ uint4 ld128(const uint8_t *p) {
  uint4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}

// Where a workgroup reads its stream from: global memory, or (STG) a copy of the stream's bytes
// the workgroup made in LDS before anything else — byte offset `off` of the stream in both cases.
// LDS reads are aligned dwords re-aligned with v_alignbyte (a frame starts at any byte).
template <bool STG>
struct DecSrc {
  const uint8_t *g;
  const uint32_t *l;
  __device__ __forceinline__ uint32_t u32(uint32_t off) const {
    if (!STG) return ld32(g + off);
    const uint32_t a = off >> 2;
    return __builtin_amdgcn_alignbyte(l[a + 1u], l[a], off & 3u);
  }
  __device__ __forceinline__ uint32_t u16(uint32_t off) const {
    if (!STG) return ld16(g + off);
    return u32(off) & 0xFFFFu;
  }
  __device__ __forceinline__ uint32_t u8(uint32_t off) const {
    if (!STG) return ld8(g + off);
    return (l[off >> 2] >> (8u * (off & 3u))) & 0xFFu;
  }
  __device__ __forceinline__ uint64_t u64(uint32_t off) const {
    if (!STG) return ld64(g + off);
    const uint32_t a = off >> 2, w0 = l[a], w1 = l[a + 1u], w2 = l[a + 2u];
    return (uint64_t)__builtin_amdgcn_alignbyte(w1, w0, off & 3u) |
           ((uint64_t)__builtin_amdgcn_alignbyte(w2, w1, off & 3u) << 32);
  }
};

__host__ __device__ constexpr uint32_t dec_frame_size(int ans) {
  return ans == RPLGPU_ANS_MEASUREMENT            ? 5u
         : ans == RPLGPU_ANS_CAPSULED             ? 84u
         : ans == RPLGPU_ANS_HQ                   ? 781u
         : ans == RPLGPU_ANS_CAPSULED_ULTRA       ? 132u
         : ans == RPLGPU_ANS_DENSE_CAPSULED       ? 84u
         : ans == RPLGPU_ANS_ULTRA_DENSE_CAPSULED ? 170u
                                                  : 0u;
}
__host__ __device__ constexpr uint32_t dec_nodes_per_frame(int ans) {
  return ans == RPLGPU_ANS_MEASUREMENT            ? 1u
         : ans == RPLGPU_ANS_CAPSULED             ? 32u
         : ans == RPLGPU_ANS_HQ                   ? 96u
         : ans == RPLGPU_ANS_CAPSULED_ULTRA       ? 96u
         : ans == RPLGPU_ANS_DENSE_CAPSULED       ? 40u
         : ans == RPLGPU_ANS_ULTRA_DENSE_CAPSULED ? 64u
                                                  : 0u;
}

// k_decode<..., STG>: byte offsets inside the dynamic LDS of a workgroup whose call allows `mf` frames
struct DecStageLayout {
  uint32_t raw_at, raw_words, frame_at, emit_at, total;
};
__host__ __device__ inline DecStageLayout dec_stage_layout(int ans, uint32_t mf) {
  DecStageLayout l;
  // (+16: the re-aligning reads of the last bytes look a dword or two further)
  l.raw_at = ((mf * dec_frame_size(ans) + 15u) & ~15u) + 16u;
  const bool filtered = ans == RPLGPU_ANS_DENSE_CAPSULED || ans == RPLGPU_ANS_ULTRA_DENSE_CAPSULED;
  l.raw_words = filtered ? (mf * dec_nodes_per_frame(ans) + 63u) / 64u : 0u;
  l.frame_at = l.raw_at + 8u * l.raw_words;
  l.emit_at = l.frame_at + 4u * mf;
  l.total = (l.emit_at + 2u * mf + 15u) & ~15u;
  return l;
}

// the common tail of every capsule decoder (e.g. :246-257): wrap the Q6 angle once, build the
// flag byte, convert to Q14 (the u16 store truncates exactly like the reference's assignment)
__device__ __forceinline__ uint2 make_node(int angle_q6, uint32_t dist_q2, uint32_t quality,
                                           uint32_t sync) {
  if (angle_q6 < 0) angle_q6 += (360 << 6);
  if (angle_q6 >= (360 << 6)) angle_q6 -= (360 << 6);
  const uint32_t q14 = (uint32_t)((angle_q6 << 8) / 90) & 0xFFFFu;
  const uint32_t flag = sync | ((sync ^ 1u) << 1);
  uint2 v;  // packed node: u16 angle | u32 dist (unaligned at byte 2) | u8 quality | u8 flag
  v.x = q14 | (dist_q2 << 16);
  v.y = (dist_q2 >> 16) | ((quality & 0xFFu) << 16) | (flag << 24);
  return v;
}

// _varbitscale_decode (:422-458)
__device__ __forceinline__ uint32_t varbitscale(uint32_t scaled, uint32_t &lvl) {
  if (scaled >= 3328u) { lvl = 4; return (1u << 14) + ((scaled - 3328u) << 4); }
  if (scaled >= 1792u) { lvl = 3; return (1u << 12) + ((scaled - 1792u) << 3); }
  if (scaled >= 1280u) { lvl = 2; return (1u << 11) + ((scaled - 1280u) << 2); }
  if (scaled >= 512u) { lvl = 1; return (1u << 9) + ((scaled - 512u) << 1); }
  lvl = 0;
  return scaled;
}

// CRC of handler_hqnode.cpp:124-126 / sl_crc.cpp:36-101: reflected CRC-32, the data zero padded
// by 4 - (len & 3) bytes -- so the padded length is always a multiple of four and the CRC can
// be advanced a 32-bit word at a time (slicing-by-4: four independent table look-ups per word
// instead of four dependent ones); tables (4 x 256 words) in LDS.
__device__ __forceinline__ uint32_t crc32_word(uint32_t crc, const uint32_t *t) {
  return t[768u + (crc & 0xFFu)] ^ t[512u + ((crc >> 8) & 0xFFu)] ^ t[256u + ((crc >> 16) & 0xFFu)] ^
         t[crc >> 24];
}

// LDS traffic of ONE wave is ordered by the hardware; this only keeps the compiler from moving
// the staged reads above the staging writes (or the next writes above the reads).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// HQ frames (781 bytes, CRC-32 over the first 777, handler_hqnode.cpp:99-133), 64 at a time per
// wave.  A thread walking its own frame touches a new cache line with every lane of every load
// and needs it again 31 loads later, when it is long gone (131 M nodes: 2.0 ms).  Here the
// wave fetches the frames together -- 16 lanes per frame, 16 ALIGNED dwords each, 13 pieces --
// into a transposed LDS tile, and then every lane runs the serial CRC of its own frame out of
// LDS; the frame's byte alignment is undone with one v_alignbit per word.  Aligned dwords that
// hold at least one frame byte never leave the buffer's pages.
// Returns bit 0 = CRC ok, bits 8..15 = the frame's first byte (sync byte 0xA5).
__device__ __forceinline__ uint32_t hq_crc_64frames(const uint8_t *base, uint32_t my_off,
                                                    uint32_t nlive, uint32_t *stg,
                                                    const uint32_t *t) {
  const uint32_t lane = lane_id(), c = lane & 15u;
  const uint32_t sh = ((uint32_t)(uintptr_t)(base + my_off) & 3u) * 8u;
  // lane (4*it + q, c) fetches word c of every piece of frame 4*it + q: its 16 frame addresses
  // (aligned down) are fixed for the whole call — kept as 32-bit offsets from the (uniform,
  // aligned-down) stream base: half the registers of 16 pointers, and the loads take the base
  // from scalar registers
  const uint32_t *abase = reinterpret_cast<const uint32_t *>((uintptr_t)base & ~(uintptr_t)3);
  const uint32_t bmis = (uint32_t)((uintptr_t)base & 3u);
  uint32_t so[16];  // word index of (frame start, aligned down) + c; 0xFFFFFFFF = no frame
#pragma unroll
  for (uint32_t it = 0; it < 16u; ++it) {
    const uint32_t fj = it * 4u + (lane >> 4);
    const uint32_t off_j = (uint32_t)__shfl((int)my_off, (int)fj);
    so[it] = fj < nlive ? ((bmis + off_j) >> 2) + c : 0xFFFFFFFFu;
  }
  uint32_t v[16];
  auto fetch = [&](uint32_t p) {  // 16 independent loads in flight
#pragma unroll
    for (uint32_t it = 0; it < 16u; ++it)
      v[it] = (so[it] != 0xFFFFFFFFu && p * 16u + c < 196u) ? abase[so[it] + p * 16u] : 0u;
  };
  fetch(0u);
  uint32_t crc = 0xFFFFFFFFu, prev = 0u, stored = 0u, first = 0u;
  for (uint32_t p = 0; p < 13u; ++p) {
#pragma unroll
    for (uint32_t it = 0; it < 16u; ++it) stg[(it * 4u + (lane >> 4)) * 17u + c] = v[it];
    wave_lds_sync();
    if (p + 1u < 13u) fetch(p + 1u);  // the next piece travels while this one is consumed
#pragma unroll
    for (uint32_t j = 0; j < 16u; ++j) {
      const uint32_t g = p * 16u + j;  // aligned word g holds frame bytes 4g - r .. 4g + 3 - r
      const uint32_t a = stg[lane * 17u + j];
      if (g >= 1u && g <= 194u) {  // frame word g-1 = data bytes 4(g-1) .. 4(g-1)+3
        const uint32_t u = __builtin_amdgcn_alignbit(a, prev, sh);
        if (g == 1u) first = u & 0xFFu;
        crc = crc32_word(crc ^ u, t);
      } else if (g == 195u) {  // bytes 776..779: the last data byte, then the stored CRC
        const uint32_t u = __builtin_amdgcn_alignbit(a, prev, sh);
        crc = crc32_word(crc ^ (u & 0xFFu), t);  // zero padded to 780 bytes (sl_crc.cpp:83-99)
        stored = (u >> 8) | (((a >> sh) & 0xFFu) << 24);
      }
      prev = a;
    }
    wave_lds_sync();
  }
  return (((crc ^ 0xFFFFFFFFu) == stored) ? 1u : 0u) | (first << 8);
}

template <int ANS, bool STG>  // STG: the three per-frame tables live in the dynamic part, sized by the call
struct DecodeLds {
  // per frame: bit31 valid, bits 0..15 start_angle_sync_q6
  uint32_t frame[STG ? 1u : DecCfg<ANS>::kTableSlots];
  uint16_t emit_frame[STG ? 1u : DecCfg<ANS>::kTableSlots];  // compacted list of the frames that publish nodes
  unsigned long long rawbits[STG ? 1u : DecCfg<ANS>::kRawBitWords];  // dense types: raw sync bit per node
  alignas(8) uint16_t smooth[DecCfg<ANS>::kSmoothSlots];  // ultra-dense: bit15 scale 0, bits 0..13 raw dist_q2
  uint8_t fin[DecCfg<ANS>::kSmoothSlots];      // ultra-dense: the smoothing state the walks arrive at (0..8, 4: unsmoothed)
  uint32_t crc_table[DecCfg<ANS>::kCrcWords];  // HQ: slicing-by-4 tables
  uint32_t stage[DecCfg<ANS>::kStageWords];    // HQ: per wave, 64 frames x 16 words (+1 pad)
  int32_t corr[DecCfg<ANS>::kCorrSlots];       // ultra: angle correction (Q16 -> Q6 units) by k2
  uint32_t syn[DecCfg<ANS>::kFuseSyn];         // FUSE: sync nodes from the header pre-pass (unordered)
  uint32_t spos[DecCfg<ANS>::kFuseSyn];        // FUSE: ... in order
  uint16_t sslot[DecCfg<ANS>::kFuseSyn];       // FUSE: batch slot of the scan that starts at spos[j] (0xFFFF: none)
  uint32_t rst[DecCfg<ANS>::kFuseRst];         // FUSE: reset requests (node positions, ascending)
  alignas(8) uint32_t tmp[40];
  uint32_t misc[8];  // 0 status, 3 last sync out, 4 last dist out, 5 error count, 6 sync nodes
};

template <int NT = kDecBlock>
__device__ __forceinline__ uint32_t dec_block_scan(uint32_t v, uint32_t *tmp, uint32_t *total) {
  // exclusive scan over the workgroup's NT threads (NT / 64 waves)
  uint32_t inc = wave_incl_scan(v);
  if (lane_id() == 63) tmp[wave_id()] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const uint32_t t = tmp[w];
    if (w < (int)wave_id()) base += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// k_decode<..., FUSE = true>: where the completed scans go (the batch layout of k_assemble), and
// the per-stream switches between the fused kernel, the plain kernel and k_assemble
struct DecFuse {
  uint2 *batch;             // scan slot g = b * scan_cap + s at batch + g * n_stride
  uint32_t *n_per_scan;     // [B * scan_cap]
  uint32_t *n_scans;        // [B]
  uint32_t *todo;           // FUSE: [B], set to 1 when this stream has to take the unfused path
  const uint32_t *only;     // plain kernel: streams with only[b] == 0 are skipped (null: all)
  uint32_t n_stride, scan_cap, max_count;
  // STG with frame offsets: [B], 1 = this stream's frames do not lie the way the staged kernel
  // needs them (see its head) and the plain kernel has to take it, 0 = done here
  uint32_t *stage_todo;
};

// FRAMED: frame offsets (and gaps) given; else back to back.  STG: the workgroup (NT threads) copies
// its stream into LDS first and decodes from there (see the head of the kernel).
template <int ANS, bool FRAMED, bool FUSE, bool STG = false, int NT = kDecBlock>
__global__ __launch_bounds__(NT) void k_decode(
    const uint8_t *__restrict__ bytes, uint64_t stream_stride, const uint32_t *__restrict__ frame_off,
    const uint8_t *__restrict__ gap, const uint32_t *__restrict__ n_frames, uint32_t max_frames,
    uint32_t sample_duration_us, const int32_t *__restrict__ state_in,
    int32_t *__restrict__ state_out, uint2 *__restrict__ nodes_out, uint32_t node_stride,
    uint32_t *__restrict__ n_nodes, uint32_t *__restrict__ reset_at, uint32_t reset_stride,
    uint32_t *__restrict__ n_reset, uint32_t *__restrict__ n_errors, uint32_t *__restrict__ status,
    uint32_t *__restrict__ sync_at, uint32_t sync_stride, uint32_t *__restrict__ n_sync, DecFuse fz) {
  static_assert(!FUSE || DecCfg<ANS>::kFusable, "FUSE: the four capsule types only");
  constexpr uint32_t S = dec_frame_size(ANS);
  constexpr uint32_t NPF = dec_nodes_per_frame(ANS);
  constexpr bool CAPS = ANS == RPLGPU_ANS_CAPSULED || ANS == RPLGPU_ANS_CAPSULED_ULTRA ||
                        ANS == RPLGPU_ANS_DENSE_CAPSULED || ANS == RPLGPU_ANS_ULTRA_DENSE_CAPSULED;
  constexpr bool FILTERED = ANS == RPLGPU_ANS_DENSE_CAPSULED || ANS == RPLGPU_ANS_ULTRA_DENSE_CAPSULED;
  constexpr uint32_t SA_OFF = ANS == RPLGPU_ANS_ULTRA_DENSE_CAPSULED ? 8u : 2u;
  __shared__ DecodeLds<ANS, STG> L;

  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  if (!FUSE && fz.only && fz.only[b] == 0u) {  // (the fused / the staged kernel dealt with this stream)
    if (STG && FRAMED && tid == 0) fz.stage_todo[b] = 0u;
    return;
  }
  const uint8_t *base = bytes + (size_t)b * stream_stride;
  const uint32_t *foff = FRAMED ? frame_off + (size_t)b * max_frames : nullptr;
  const uint8_t *fgap = (FRAMED && gap) ? gap + (size_t)b * max_frames : nullptr;
  const uint32_t nf = min(n_frames[b], min(max_frames, DecCfg<ANS>::kMaxFrames));
  uint2 *out = FUSE ? nullptr : nodes_out + (size_t)b * node_stride;
  uint32_t *Lframe_ = nullptr;  // (= Lframe, declared below)
  auto frame_ptr = [&](uint32_t k) -> const uint8_t * {
    return base + (FRAMED ? (size_t)foff[k] : (size_t)k * S);
  };
  // byte offset of frame k in the stream.  STG with frame offsets: in the staged copy, which starts
  // at frame 0 — k * S plus the bytes rejected in front of frame k, kept in bits 16..29 of the frame's
  // table entry by P1 (with the gap flag in bit 30), so that nothing after P1 loads an offset again
  auto frame_at = [&](uint32_t k) -> uint32_t {
    if (STG && FRAMED) return k * S + ((Lframe_[k] >> 16) & 0x3FFFu);
    return FRAMED ? foff[k] : k * S;
  };
#ifdef RPL_DEC_DBG
  const unsigned long long dbg_start = __builtin_amdgcn_s_memtime();
#endif
  // STG: the dynamic part of the workgroup's LDS = [the stream's bytes | raw sync bits | frame table |
  // emit list], the three tables sized by the call's max_frames (dec_stage_layout, shared with the
  // launcher).  While other workgroups of the CU have node stores in flight every global load of
  // the CU queues behind them (profiles/r02/decode_study.txt section 3): P1 and P3 of the plain kernel
  // wait that queue out once per trip.  Here the stream crosses it ONCE, as one burst of 16-byte
  // loads issued before this workgroup has stored anything, and both passes read LDS.
  extern __shared__ uint4 dec_stage[];
  static_assert(!STG || (CAPS && (!FUSE || FRAMED)), "STG: capsule streams; the fused form only with frame offsets");
  static_assert(ANS != RPLGPU_ANS_HQ || NT == 256, "HQ: one CRC table entry per thread");
  const DecStageLayout lay = dec_stage_layout(ANS, STG ? min(max_frames, DecCfg<ANS>::kMaxFrames) : 0u);
  uint8_t *dyn = reinterpret_cast<uint8_t *>(dec_stage);
  const uint32_t raw_words = STG ? lay.raw_words : DecCfg<ANS>::kRawBitWords;
  unsigned long long *Lraw = STG ? reinterpret_cast<unsigned long long *>(dyn + lay.raw_at) : L.rawbits;
  uint32_t *Lframe = STG ? reinterpret_cast<uint32_t *>(dyn + lay.frame_at) : L.frame;
  Lframe_ = Lframe;
  uint16_t *Lemit = STG ? reinterpret_cast<uint16_t *>(dyn + lay.emit_at) : L.emit_frame;
  const DecSrc<STG> src{base, reinterpret_cast<const uint32_t *>(dec_stage)};
  // With frame offsets the staged range is [offset of frame 0, end of the last frame): as long as
  // it fits the LDS the call reserved (rejected bytes between the frames count) and, checked per
  // frame in P1, every frame lies inside it at most 16383 rejected bytes behind its back-to-back
  // place — offsets as rplgpu_frame_stream produces them.  Otherwise nothing is written,
  // fz.stage_todo[b] is raised and the plain kernel takes the stream.
  uint32_t stage_lo = 0, stage_hi = nf * S;
  if (STG && FRAMED && nf) {
    stage_lo = foff[0];
    stage_hi = foff[nf - 1u] + S;
    if (stage_hi < stage_lo + nf * S || stage_hi - stage_lo > lay.raw_at - 16u) {  // (block-uniform)
      if (tid == 0) (FUSE ? fz.todo : fz.stage_todo)[b] = 1u;  // (FUSE: the general path takes the stream)
      return;
    }
  }
  if (STG) {
    const uint8_t *from = base + stage_lo;
    const uint32_t nbytes = stage_hi - stage_lo, n16 = nbytes >> 4;
    constexpr uint32_t UN = 4;
    uint32_t i = tid;
    for (; i + (UN - 1u) * NT < n16; i += UN * NT) {
      uint4 v[UN];
#pragma unroll
      for (uint32_t u = 0; u < UN; ++u) v[u] = ld128(from + 16u * (size_t)(i + u * NT));
#pragma unroll
      for (uint32_t u = 0; u < UN; ++u) dec_stage[i + u * NT] = v[u];
    }
    for (; i < n16; i += NT) dec_stage[i] = ld128(from + 16u * (size_t)i);
    for (uint32_t q = (n16 << 4) + tid; q < nbytes; q += NT) dyn[q] = from[q];
  }

  // state word 2, bit 0: frame 0 is the previous call's last frame, handed over again only as
  // the predecessor of frame 1 (its errors / reset request / nodes were accounted for then)
  const bool carry0 = CAPS && state_in && (state_in[4 * b + 2] & 1);
  if (tid < 8) L.misc[tid] = 0u;
  if (ANS == RPLGPU_ANS_HQ) {
    uint32_t c = tid;  // NT == 256 table entries
#pragma unroll
    for (int j = 0; j < 8; ++j) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
    L.crc_table[tid] = c;
    __syncthreads();
#pragma unroll
    for (int k = 1; k < 4; ++k) {  // T_k[i] = T_{k-1}[i] advanced by one zero byte
      c = (c >> 8) ^ L.crc_table[c & 0xFFu];
      L.crc_table[256 * k + tid] = c;
    }
  }
  if (FILTERED) {  // raw sync bits are rare: P3 sets them with an LDS atomic, so start from zero
    const uint32_t nw = min(raw_words, (uint32_t)(((uint64_t)nf * NPF + 63u) >> 6));
    for (uint32_t w = tid; w < nw; w += NT) Lraw[w] = 0ull;
  }
  if (ANS == RPLGPU_ANS_CAPSULED_ULTRA) {
    // handler_capsules.cpp:545-556: offsetAngleMean_q16 depends on the distance only through
    // k2 = 98361 / dist (0 .. 491 for dist >= 200); entry 492 = the default for dist < 200.
    // The reference's double arithmetic, done once per value instead of once per node.
    for (uint32_t k2 = tid; k2 < 493u; k2 += NT) {
      int off_q16 = (int)(7.5 * 3.1415926535 * (1 << 16) / 180.0);
      if (k2 < 492u) {
        const int kk = (int)k2;
        off_q16 = (int)(8 * 3.1415926535 * (1 << 16) / 180) - (kk << 6) - (kk * kk * kk) / 98304;
      }
      L.corr[k2] = (int)((double)(off_q16 * 180) / 3.14159265);
    }
  }
  if (n_frames[b] > nf && tid == 0) L.misc[0] = RPLGPU_STREAM_FRAMES_TRUNCATED;
  __syncthreads();

#ifdef RPL_DEC_DBG
  unsigned long long dbg_t[7];
  dbg_t[0] = __builtin_amdgcn_s_memtime();
#endif
  // ---- P1: per frame: framing check (unframed input only), checksum, header word ---------
  uint32_t my_err = 0, unframed = 0;
  if (ANS == RPLGPU_ANS_HQ) {  // handler_hqnode.cpp:99-172, a wave per 64 frames
    uint32_t *stg = L.stage + wave_id() * (64u * 17u);
    for (uint32_t k0 = wave_id() * 64u; k0 < nf; k0 += NT) {
      const uint32_t k = k0 + lane_id();
      const bool live = k < nf;
      const uint32_t my_off = live ? (FRAMED ? foff[k] : k * S) : 0u;
      const uint32_t res = hq_crc_64frames(base, my_off, min(nf - k0, 64u), stg, L.crc_table);
      if (live) {
        if (!FRAMED && (res >> 8) != 0xA5u) unframed = 1;
        Lframe[k] = (res & 1u) ? 0x80000000u : 0u;
        my_err += (res & 1u) ? 0u : 1u;
      }
    }
  }
  if (ANS == RPLGPU_ANS_MEASUREMENT && !FRAMED) {  // handler_normalnode.cpp:88-112
    for (uint32_t k = tid; k < nf; k += NT) {
      const uint8_t *f = frame_ptr(k);
      const uint32_t b0 = ld8(f), b1 = ld8(f + 1);
      if (!((((b0 >> 1) ^ b0) & 1u) && (b1 & 1u))) unframed = 1;
    }
  }
  if (CAPS) {  // the four capsule types: handler_capsules.cpp:107-194 and siblings
    // XOR of bytes 2 .. S-1 (:137-150).  EIGHT lanes share a frame: lane j takes the (possibly
    // unaligned) dwords j, j+8, ... of it, so a wave instruction reads eight 32-byte runs of
    // eight neighbouring frames and the next instruction continues in the same lines (one lane
    // per frame touched a different line with every lane of every load).  The loads of FOUR
    // such trips (4 x 32 frames per workgroup) are issued before the first is consumed.
    // Alone this pass takes 30 k cycles per 801-frame stream; next to other workgroups' node
    // stores it takes 130 k whatever its shape — it queues behind them (decode_study.txt).
    constexpr uint32_t LPF = 8, NDW = (S + 3u) / 4u, TRIPS = (NDW + LPF - 1u) / LPF;
    constexpr uint32_t FPT = NT / LPF, DEPTH = 4;  // frames per trip, trips in flight
    const uint32_t j = tid & (LPF - 1u);
    // STG: out of LDS a lane takes a whole frame (its dwords are an odd number of banks apart from
    // the neighbour's: no conflicts, no cross-lane reduction)
    for (uint32_t k = tid; STG && k < nf; k += NT) {
      uint32_t f = k * S, place = 0;  // place: bits 16..30 of the table entry (rejected bytes, gap flag)
      if (FRAMED) {
        const uint32_t off = foff[k], rej = off - stage_lo - k * S;
        if (off < stage_lo || off + S > stage_hi || rej > 0x3FFFu) {
          L.misc[1] = 1u;  // (not where the staged kernel needs it: the plain kernel takes the stream)
          continue;
        }
        f = off - stage_lo;
        place = (rej << 16) | ((fgap && fgap[k]) ? 0x40000000u : 0u);
      }
      uint32_t xw = 0, first = 0;
#pragma unroll
      for (uint32_t d = 0; d < NDW; ++d) {
        uint32_t w = 0;
        if (4u * d + 4u <= S) {
          w = src.u32(f + 4u * d);
        } else {
#pragma unroll
          for (uint32_t q = 0; q < (S & 3u); ++q) w |= src.u8(f + 4u * d + q) << (8u * q);
        }
        if (d == 0u) first = w;
        xw ^= w;
      }
      const uint32_t b0 = first & 0xFFu, b1 = (first >> 8) & 0xFFu;
      if (!FRAMED && ((b0 >> 4) != 0xAu || (b1 >> 4) != 0x5u)) unframed = 1;
      xw ^= xw >> 16;
      const uint32_t x = ((xw ^ (xw >> 8)) ^ b0 ^ b1) & 0xFFu;
      const bool ok = (((b0 & 0xFu) | (b1 << 4)) & 0xFFu) == x;
      my_err += (ok || (carry0 && k == 0u)) ? 0u : 1u;
      const uint32_t sa = SA_OFF == 2u ? (first >> 16) : src.u16(f + SA_OFF);
      Lframe[k] = (ok ? 0x80000000u : 0u) | place | sa;
    }
    for (uint32_t k0 = 0; !STG && k0 < nf; k0 += DEPTH * FPT) {
      uint32_t v[DEPTH][TRIPS];
#pragma unroll
      for (uint32_t u = 0; u < DEPTH; ++u) {
        const uint32_t k = k0 + u * FPT + tid / LPF;
        const bool live = k < nf;
        const uint32_t f = frame_at(live ? k : 0u);
#pragma unroll
        for (uint32_t i = 0; i < TRIPS; ++i) {
          const uint32_t d = j + LPF * i;
          uint32_t w = 0;
          if (live && 4u * d + 4u <= S) {
            w = src.u32(f + 4u * d);
          } else if ((S & 3u) != 0u && live && d == NDW - 1u) {  // the frame's last, partial dword
#pragma unroll
            for (uint32_t q = 0; q < (S & 3u); ++q) w |= src.u8(f + 4u * (NDW - 1u) + q) << (8u * q);
          }
          v[u][i] = w;
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < DEPTH; ++u) {
        const uint32_t k = k0 + u * FPT + tid / LPF;
        const bool live = k < nf;
        uint32_t xw = 0;
#pragma unroll
        for (uint32_t i = 0; i < TRIPS; ++i) xw ^= v[u][i];
        const uint32_t first = v[u][0];
        xw ^= (uint32_t)__shfl_xor((int)xw, 1, 64);
        xw ^= (uint32_t)__shfl_xor((int)xw, 2, 64);
        xw ^= (uint32_t)__shfl_xor((int)xw, 4, 64);
        if (live && j == 0u) {
          const uint32_t b0 = first & 0xFFu, b1 = (first >> 8) & 0xFFu;  // XOR-ed out again below
          if (!FRAMED && ((b0 >> 4) != 0xAu || (b1 >> 4) != 0x5u)) unframed = 1;
          xw ^= xw >> 16;
          const uint32_t x = ((xw ^ (xw >> 8)) ^ b0 ^ b1) & 0xFFu;
          const bool ok = (((b0 & 0xFu) | (b1 << 4)) & 0xFFu) == x;
          my_err += (ok || (carry0 && k == 0u)) ? 0u : 1u;
          const uint32_t sa = SA_OFF == 2u ? (first >> 16) : src.u16(frame_at(k) + SA_OFF);
          Lframe[k] = (ok ? 0x80000000u : 0u) | sa;
        }
      }
    }
  }
  if (unframed) atomicOr(&L.misc[0], RPLGPU_STREAM_UNFRAMED);
  __syncthreads();
  if (STG && FRAMED) {
    const bool misplaced = L.misc[1] != 0u;  // (block-uniform; nothing has left the workgroup yet)
    if (tid == 0) {
      if (!FUSE) fz.stage_todo[b] = misplaced ? 1u : 0u;
      else if (misplaced) fz.todo[b] = 1u;  // (0 is written further down, once the scans are judged)
    }
    if (misplaced) return;
  }
  const bool bad_framing = (L.misc[0] & RPLGPU_STREAM_UNFRAMED) != 0u;

#ifdef RPL_DEC_DBG
  __syncthreads();
  dbg_t[1] = __builtin_amdgcn_s_memtime();
#endif
  // ---- P2: which frames publish, node offsets, reset requests ------------------------------
  // (loop over chunks of 256 frames with a running carry)
  uint32_t carry_nodes = 0, carry_emit = 0, carry_reset = 0;
  const int thr_q8 = FILTERED ? (int)((360u * 100u * (ANS == RPLGPU_ANS_DENSE_CAPSULED ? 40u : 32u) /
                                       (1000000u / sample_duration_us)) << 8)
                              : 0;
  if (!DecCfg<ANS>::kTable && !bad_framing) carry_emit = nf;  // every legacy node publishes
  for (uint32_t k0 = 0; k0 < nf && !bad_framing && DecCfg<ANS>::kTable; k0 += NT) {
    const uint32_t k = k0 + tid;
    uint32_t emits = 0, resets = 0;
    if (k < nf) {
      const uint32_t rec = Lframe[k];
      if (!CAPS) {
        emits = rec >> 31;
      } else if (rec >> 31) {
        resets = (rec >> 15) & 1u;  // revolution start: publishNewScanReset (:160-171)
        if (carry0 && k == 0u) resets = 0u;
        const bool gap_k = (STG && FRAMED) ? ((rec >> 30) & 1u) != 0u : (fgap && fgap[k]);
        const bool prev_ok = k > 0 && (Lframe[k - 1] >> 31) && !gap_k;
        if (prev_ok && !resets) {
          emits = 1;
          if (FILTERED) {  // :750-754 / :971-975: too large an angle step -> discard
            const int cur_q8 = (int)(rec & 0x7FFFu) << 2, prev_q8 = (int)(Lframe[k - 1] & 0x7FFFu) << 2;
            int diff = cur_q8 - prev_q8;
            if (prev_q8 > cur_q8) diff += (360 << 8);
            if (diff > thr_q8) emits = 0;
          }
        }
      }
    }
    uint32_t tot_e, tot_r;
    const uint32_t ex_e = dec_block_scan<NT>(emits, L.tmp, &tot_e);
    const uint32_t ex_r = dec_block_scan<NT>(resets, L.tmp, &tot_r);
    if (emits) Lemit[carry_emit + ex_e] = (uint16_t)k;
    if (resets) {
      const uint32_t slot = carry_reset + ex_r;
      if (reset_at && slot < reset_stride)
        reset_at[(size_t)b * reset_stride + slot] = (carry_emit + ex_e) * NPF;
      if (FUSE && slot < DecCfg<ANS>::kFuseRst) L.rst[slot] = (carry_emit + ex_e) * NPF;
    }
    carry_emit += tot_e;
    carry_reset += tot_r;
  }
  carry_nodes = carry_emit * NPF;
  __syncthreads();
  const uint32_t n_emit = carry_emit;
  const uint32_t n_out = FUSE ? carry_nodes : min(carry_nodes, node_stride);
  const int last_sync_in = state_in ? state_in[4 * b] : 0;
  const int last_dist_in = state_in ? state_in[4 * b + 1] : 0;

  // The sync-bit filter of the dense modes, s_i = r_i & ~s_{i-1} (see P4), over the raw bits in
  // LDS; on_sync(i) is called for every node whose filtered bit is set, L.misc[3] receives the
  // last node's bit (the state carried to the next call).
  auto sync_filter = [&](auto on_sync) {
    const uint32_t nwords = (carry_nodes + 63u) >> 6;
    for (uint32_t w = tid; w < nwords && w < raw_words; w += NT) {
      unsigned long long m = Lraw[w];
      while (m) {
        const uint32_t bit = (uint32_t)__builtin_ctzll(m);
        m &= m - 1ull;
        const uint32_t i = (w << 6) + bit;
        // length of the run of raw sync bits ending just before i (across words, and into the
        // carried-in state when the run reaches the start of the stream)
        uint32_t run = 0;
        int j = (int)i - 1;
        while (j >= 0 && ((Lraw[j >> 6] >> (j & 63)) & 1ull)) { ++run; --j; }
        if (j < 0 && last_sync_in) {
          // s_{-1} = 1 acts like one more raw bit in front of the run
          ++run;
        }
        const uint32_t s = (run & 1u) ? 0u : 1u;
        if (s) on_sync(i);
        if (i == carry_nodes - 1u) L.misc[3] = s;  // carried-out state
      }
    }
    if (tid == 0 && carry_nodes) {
      const uint32_t i = carry_nodes - 1u;
      if (!((Lraw[i >> 6] >> (i & 63)) & 1ull)) L.misc[3] = 0u;
    }
  };

  // ---- FUSE: scan boundaries from the capsule headers, before any payload is touched -------
  uint32_t f_nsync = 0;  // sync nodes of the stream (in order in L.spos)
  if (FUSE) {
    constexpr uint32_t kSyn = DecCfg<ANS>::kFuseSyn, kRst = DecCfg<ANS>::kFuseRst;
    // every node's raw sync bit: ((angle + step) mod 360 deg) < step (dense: 2 x step), a function
    // of the two start angles of its capsule pair (:246-257 and siblings) — one lane per frame
    for (uint32_t e = tid; e < n_emit; e += NT) {
      const uint32_t k = Lemit[e];
      const int cur_q8 = (int)(Lframe[k] & 0x7FFFu) << 2, prev_q8 = (int)(Lframe[k - 1] & 0x7FFFu) << 2;
      int diff_q8 = cur_q8 - prev_q8;
      if (prev_q8 > cur_q8) diff_q8 += (360 << 8);
      const int inc = ANS == RPLGPU_ANS_CAPSULED         ? diff_q8 << 3
                      : ANS == RPLGPU_ANS_CAPSULED_ULTRA ? (diff_q8 << 3) / 3
                      : ANS == RPLGPU_ANS_DENSE_CAPSULED ? (diff_q8 << 8) / 40
                                                         : (diff_q8 << 8) / 64;
      const int lim = FILTERED ? inc * 2 : inc;
      auto mark = [&](uint32_t pos) {
        const uint32_t i = e * NPF + pos;
        if (FILTERED) {
          atomicOr(&Lraw[i >> 6], 1ull << (i & 63u));
        } else {
          const uint32_t at = atomicAdd(&L.misc[6], 1u);
          if (at < kSyn) L.syn[at] = i;
        }
      };
      // v_j = A + j * inc, j = pos + 1 = 1 .. NPF, is tested as (v_j mod 360 deg) < lim.  With a positive
      // step, a start angle below 360 deg and no second wrap inside the capsule (any stream a sensor
      // produces) v_j rises through [0, 720 deg) and the test holds where v_j first reaches 360 deg
      // (j0; dense types: j0 and j0 + 1) and, dense types only, at j = 1 when A < inc — one division
      // per capsule instead of NPF remainders.  Anything else (corrupted-but-checksummed angles:
      // negative steps, angles above 360 deg) walks the positions as the reference does.
      constexpr int kTurn = 360 << 16;
      const int A = prev_q8 << 8;
      if (inc > 0 && A < kTurn && A + (int)NPF * inc < 2 * kTurn) {
        if (FILTERED && A < inc && A + inc < kTurn) mark(0u);
        const uint32_t t = (uint32_t)(kTurn - A);
        const uint32_t j0 = (t + (uint32_t)inc - 1u) / (uint32_t)inc;  // (>= 1)
        if (j0 <= NPF) mark(j0 - 1u);
        if (FILTERED && j0 + 1u <= NPF) mark(j0);
      } else {
        for (uint32_t pos = 0; pos < NPF; ++pos) {
          const int ang = A + (int)pos * inc;
          if (((ang + inc) % kTurn) < lim) mark(pos);
        }
      }
    }
    __syncthreads();
    if (FILTERED) {
      sync_filter([&](uint32_t i) {
        const uint32_t at = atomicAdd(&L.misc[6], 1u);
        if (at < kSyn) L.syn[at] = i;
      });
      __syncthreads();
    }
    f_nsync = L.misc[6];
    if (f_nsync > kSyn || carry_reset > kRst) {  // (block-uniform) more than the tables hold:
      if (tid == 0) fz.todo[b] = 1u;             // this stream takes the unfused path, nothing
      return;                                    // has been written for it yet
    }
    if (tid == 0) fz.todo[b] = 0u;
    // order the (few) sync nodes; scan j = [spos[j], spos[j+1]) is complete iff no reset request
    // p has spos[j] < p <= spos[j+1] (ScanDataHolder, src/sdk/src/sl_lidar_driver.cpp:272-315)
    if (tid < f_nsync) {
      const uint32_t v = L.syn[tid];
      uint32_t rank = 0;
      for (uint32_t q = 0; q < f_nsync; ++q) rank += L.syn[q] < v ? 1u : 0u;
      L.spos[rank] = v;
    }
    __syncthreads();
    uint32_t ok = 0, s0 = 0, s1 = 0;
    if (tid + 1u < f_nsync) {
      s0 = L.spos[tid];
      s1 = L.spos[tid + 1u];
      ok = 1;
      for (uint32_t r = 0; r < carry_reset; ++r) {
        const uint32_t p = L.rst[r];
        if (p > s0 && p <= s1) ok = 0;
      }
    }
    uint32_t completed;
    const uint32_t slot = dec_block_scan<NT>(ok, L.tmp, &completed);
    if (tid < kSyn) L.sslot[tid] = (ok && slot < fz.scan_cap) ? (uint16_t)slot : (uint16_t)0xFFFFu;
    uint32_t st_bits = 0;
    if (ok && slot < fz.scan_cap) {
      const uint32_t full = min(s1 - s0, fz.max_count);
      fz.n_per_scan[(size_t)b * fz.scan_cap + slot] = min(full, fz.n_stride);
      if (full > fz.n_stride) st_bits |= RPLGPU_SCAN_OUT_TRUNCATED;
    }
    if (completed > fz.scan_cap) st_bits |= RPLGPU_STREAM_RESETS_TRUNCATED;
    if (st_bits) atomicOr(&L.misc[0], st_bits);
    for (uint32_t q = min(completed, fz.scan_cap) + tid; q < fz.scan_cap; q += NT)
      fz.n_per_scan[(size_t)b * fz.scan_cap + q] = 0u;
    if (tid == 0) fz.n_scans[b] = min(completed, fz.scan_cap);
    __syncthreads();
  }

#ifdef RPL_DEC_DBG
  __syncthreads();
  dbg_t[2] = __builtin_amdgcn_s_memtime();
  unsigned long long dbg_p3 = 0, dbg_p5 = 0, dbg_mark = dbg_t[2];
  unsigned long long dbg_ud[4] = {0, 0, 0, 0}, dbg_u0 = 0;  // ultra-dense: walk, map scan, second walk, final decode
#endif
  // ---- P3: the nodes ----------------------------------------------------------------------
  // A lane decodes G consecutive nodes of ONE frame: the per-frame arithmetic (table look-ups,
  // angle step, the divisions) is paid once per group, the payload arrives in one or two wide
  // loads and the nodes leave in 16-byte stores.  (One node per lane and pass, as in round 1,
  // spent ~85 instructions per node, most of them per-frame work repeated 40 times, behind a
  // chain of four dependent loads.)  Sync nodes are rare (one or two per revolution): they are
  // reported through LDS / global atomics instead of a ballot per pass.
  constexpr uint32_t G = DecCfg<ANS>::kGroup, GPF = NPF / G;
  static_assert(GPF * G == NPF, "groups tile a frame");
  const uint32_t n_groups = carry_emit * GPF;
  uint32_t *my_sync = sync_at ? sync_at + (size_t)b * sync_stride : nullptr;
  auto report_sync = [&](uint32_t i) {  // node i carries the sync flag (and is inside the output)
    const uint32_t slot = atomicAdd(&L.misc[6], 1u);
    if (my_sync && slot < sync_stride) my_sync[slot] = i;
  };
  // What a group needs from memory (fetched for U groups before the first one is decoded: the
  // frame-table look-ups and the payload loads of the next groups travel while this one is
  // being computed — one group per trip left the kernel waiting on a chain of dependent loads)
  uint32_t chunk0 = 0;  // ultra-dense: first node of the chunk being decoded / smoothed
  struct GroupIn {
    uint32_t pos0, off_prev, off_cur;  // byte offsets of the frames k-1 and k in the stream
    uint32_t scan_j;                   // FUSE: the scan the group's first node lies in (-1: none yet)
    bool live;                         // FUSE: some node of the group belongs to a stored scan
    int prev_q8, diff_q8;
    uint64_t w;   // first 8 payload bytes of the group
    uint32_t w2;  // the bytes after them (express / ultra-dense: 2, ultra: the next cabin word)
    uint4 a, c;   // HQ: the four nodes
  };
  auto locate = [&](uint32_t t, GroupIn &g) {  // which frames, where (table look-ups only)
    const uint32_t e = t / GPF, q = t - e * GPF;
    g.pos0 = q * G;
    g.live = true;
    if (FUSE) {
      const uint32_t i = t * G;
      uint32_t j = 0xFFFFFFFFu;
      for (uint32_t s_ = 0; s_ < f_nsync && L.spos[s_] <= i; ++s_) j = s_;
      g.scan_j = j;
      const bool in_stored = j != 0xFFFFFFFFu && j + 1u < f_nsync && L.sslot[j] != 0xFFFFu;
      const bool crosses = j + 1u < f_nsync && L.spos[j + 1u] < i + G;  // (j = -1: first sync node)
      // (ultra-dense decodes every group: the smoothing chain runs through the nodes of scans that
      // are not delivered too; only their stores are skipped)
      g.live = in_stored || crosses || ANS == RPLGPU_ANS_ULTRA_DENSE_CAPSULED;
      if (!g.live) return;
    }
    const uint32_t k = DecCfg<ANS>::kTable ? (uint32_t)Lemit[e] : e;
    constexpr bool kPrev = CAPS;  // capsule types decode frame k-1 with frame k's start angle
    if (STG && FRAMED) {
      g.off_cur = frame_at(k);
      g.off_prev = kPrev ? frame_at(k - 1u) : 0u;
    } else if (FRAMED) {
      g.off_cur = foff[k];
      g.off_prev = kPrev ? foff[k - 1u] : 0u;
    } else {
      g.off_cur = k * S;
      g.off_prev = kPrev ? (k - 1u) * S : 0u;
    }
    if (CAPS) {
      // signed arithmetic throughout, as in the reference: a corrupted-but-checksummed start
      // angle above 360 deg makes the step (and everything derived from it) negative
      const int cur_q8 = (int)(Lframe[k] & 0x7FFFu) << 2;
      g.prev_q8 = (int)(Lframe[k - 1] & 0x7FFFu) << 2;
      g.diff_q8 = cur_q8 - g.prev_q8;
      if (g.prev_q8 > cur_q8) g.diff_q8 += (360 << 8);
    }
  };
  auto fetch = [&](GroupIn &g) {  // the payload loads (nothing here waits for another load)
    if (FUSE && !g.live) return;
    if (ANS == RPLGPU_ANS_MEASUREMENT) {
      const uint8_t *f = base + g.off_cur;
      g.w = (uint64_t)ld8(f) | ((uint64_t)ld16(f + 1) << 8) | ((uint64_t)ld16(f + 3) << 24);
    } else if (ANS == RPLGPU_ANS_HQ) {
      const uint8_t *f = base + g.off_cur + 9u + 8u * g.pos0;
      g.a = ld128(f);
      g.c = ld128(f + 16);
    } else {
      const uint32_t prev = g.off_prev;
      if (ANS == RPLGPU_ANS_CAPSULED) {  // two cabins of 5 bytes
        const uint32_t c = prev + 4u + 5u * (g.pos0 >> 1);
        g.w = src.u64(c);
        g.w2 = src.u16(c + 8u);
      } else if (ANS == RPLGPU_ANS_CAPSULED_ULTRA) {  // two cabins of 4 bytes + the one behind
        const uint32_t cab0 = g.pos0 / 3u;  // even
        g.w = src.u64(prev + 4u + 4u * cab0);
        g.w2 = (cab0 + 2u == 32u) ? src.u32(g.off_cur + 4u) : src.u32(prev + 4u + 4u * (cab0 + 2u));
      } else if (ANS == RPLGPU_ANS_DENSE_CAPSULED) {  // four 16-bit distances
        g.w = src.u64(prev + 4u + 2u * g.pos0);
      } else {  // ultra dense: two cabins of 5 bytes
        const uint32_t c = prev + 10u + 5u * (g.pos0 >> 1);
        g.w = src.u64(c);
        g.w2 = src.u16(c + 8u);
      }
    }
  };
  // mode 0: decode and store.  Ultra-dense runs it twice per group: mode 1 only leaves the raw
  // distances in LDS for the smoothing pass, mode 2 decodes again from the payload still in
  // registers, takes the smoothing states from LDS and stores the nodes — once.
  auto emit = [&](uint32_t t, const GroupIn &g, auto mode_c) {
    constexpr int MODE = decltype(mode_c)::value;
    if (FUSE && !g.live) return;
    const uint32_t i = t * G, pos0 = g.pos0;
    const int prev_q8 = g.prev_q8, diff_q8 = g.diff_q8;
    uint2 nd[G];
    uint32_t rs = 0;  // bit j: node j of the group has its (raw) sync bit set
    if (ANS == RPLGPU_ANS_MEASUREMENT) {  // handler_normalnode.cpp:121-130
      const uint32_t b0 = (uint32_t)g.w & 0xFFu, aq = (uint32_t)(g.w >> 8) & 0xFFFFu, d = (uint32_t)(g.w >> 24) & 0xFFFFu;
      const uint32_t q14 = (((aq >> 1) << 8) / 90u) & 0xFFFFu;
      nd[0].x = q14 | (d << 16);
      nd[0].y = (((b0 >> 2) << 2) << 16) | ((b0 & 1u) << 24);
      rs = b0 & 1u;
    } else if (ANS == RPLGPU_ANS_HQ) {  // handler_hqnode.cpp:150-160: verbatim copy
      nd[0] = make_uint2(g.a.x, g.a.y);
      nd[G > 1 ? 1 : 0] = make_uint2(g.a.z, g.a.w);
      nd[G > 2 ? 2 : 0] = make_uint2(g.c.x, g.c.y);
      nd[G > 3 ? 3 : 0] = make_uint2(g.c.z, g.c.w);
#pragma unroll
      for (uint32_t j = 0; j < G; ++j) rs |= ((nd[j].y >> 24) & 1u) << j;
    } else if (ANS == RPLGPU_ANS_CAPSULED) {  // :206-260 — two cabins of 5 bytes = 4 nodes
      const int inc = diff_q8 << 3;
      const uint64_t cab[2] = {g.w & 0xFFFFFFFFFFull, (g.w >> 40) | ((uint64_t)g.w2 << 24)};
#pragma unroll
      for (uint32_t j = 0; j < G; ++j) {
        const uint32_t pos = pos0 + j;
        const int ang = (prev_q8 << 8) + (int)pos * inc;
        const uint64_t cb = cab[j >> 1];
        const uint32_t da = (uint32_t)(cb >> (16u * (j & 1u))) & 0xFFFFu, offs = (uint32_t)(cb >> 32) & 0xFFu;
        const int dist = (int)(da & 0xFFFCu);
        const int aoff = (int)(((j & 1u) ? (offs >> 4) : (offs & 0xFu)) | ((da & 0x3u) << 4));
        const int angle_q6 = (ang - (aoff << 13)) >> 10;
        const uint32_t sync = (((ang + inc) % (360 << 16)) < inc) ? 1u : 0u;
        nd[j] = make_node(angle_q6, (uint32_t)dist, dist ? (0x2Fu << 2) : 0u, sync);
        rs |= sync << j;
      }
    } else if (ANS == RPLGPU_ANS_CAPSULED_ULTRA) {  // :460-577 — two cabins of 4 bytes = 6 nodes
      const int inc = (diff_q8 << 3) / 3;
      const uint32_t cw[3] = {(uint32_t)g.w, (uint32_t)(g.w >> 32), g.w2};
      uint32_t lvl[3];
      int major[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) major[c] = (int)varbitscale(cw[c] & 0xFFFu, lvl[c]);
#pragma unroll
      for (uint32_t j = 0; j < G; ++j) {
        const uint32_t pos = pos0 + j, cb = j / 3u, jj = j - cb * 3u;
        const int ang = (prev_q8 << 8) + (int)pos * inc;
        const uint32_t cx = cw[cb];
        const int mj = major[cb], mj2 = major[cb + 1u];
        const uint32_t lvl1 = lvl[cb], lvl2 = lvl[cb + 1u];
        int dist;
        if (jj == 0u) {
          dist = mj << 2;
        } else {
          int pred = (jj == 1u) ? (((int)(cx << 10)) >> 22) : (((int)cx) >> 22);
          if ((uint32_t)pred == 0xFFFFFE00u || (uint32_t)pred == 0x1FFu) {
            dist = 0;
          } else if (jj == 1u) {
            int base1 = mj;
            uint32_t l1 = lvl1;
            if (!mj && mj2) { base1 = mj2; l1 = lvl2; }
            pred = (int)((uint32_t)pred << l1);
            dist = (int)((uint32_t)(pred + base1) << 2);
          } else {
            pred = (int)((uint32_t)pred << lvl2);
            dist = (int)((uint32_t)(pred + mj2) << 2);
          }
        }
        const uint32_t sync = (((ang + inc) % (360 << 16)) < inc) ? 1u : 0u;
        // triangulation angle correction :547-556, from the table (see the kernel's head);
        // k2 = 98361 / dist exactly: dist < 2^24, so the float quotient is off by at most one
        uint32_t slot = 492u;
        if (dist >= (50 * 4)) {
          uint32_t k2 = (uint32_t)(98361.0f * __builtin_amdgcn_rcpf((float)dist));
          int r = 98361 - (int)(k2 * (uint32_t)dist);
          if (r < 0) { --k2; r += dist; }
          if (r >= dist) ++k2;
          slot = k2;
        }
        const int angle_q6 = (ang - L.corr[slot]) >> 10;
        nd[j] = make_node(angle_q6, (uint32_t)dist, dist ? (0x2Fu << 2) : 0u, sync);
        rs |= sync << j;
      }
    } else if (ANS == RPLGPU_ANS_DENSE_CAPSULED) {  // :756-784 — four 16-bit distances
      const int inc = (diff_q8 << 8) / 40;
#pragma unroll
      for (uint32_t j = 0; j < G; ++j) {
        const int ang = (prev_q8 << 8) + (int)(pos0 + j) * inc;
        const int dist = (int)((uint32_t)(g.w >> (16u * j)) & 0xFFFFu) << 2;
        rs |= ((((ang + inc) % (360 << 16)) < (inc * 2)) ? 1u : 0u) << j;
        nd[j] = make_node(ang >> 10, (uint32_t)dist, dist ? (0x2Fu << 2) : 0u, 0u);
      }
    } else {  // ultra dense :979-1045 — two cabins of 5 bytes = 4 nodes
      const int inc = (diff_q8 << 8) / 64;
      const uint64_t cab[2] = {g.w & 0xFFFFFFFFFFull, (g.w >> 40) | ((uint64_t)g.w2 << 24)};
      uint32_t sm[G];
#pragma unroll
      for (uint32_t j = 0; j < G; ++j) {
        const int ang = (prev_q8 << 8) + (int)(pos0 + j) * inc;
        const uint64_t cb = cab[j >> 1];
        const uint32_t q4 = (uint32_t)(cb >> 32) & 0xFFu;
        const uint32_t qds = ((uint32_t)(cb >> (16u * (j & 1u))) & 0xFFFFu) |
                             (((j & 1u) ? (q4 >> 4) : (q4 & 0xFu)) << 16);
        const uint32_t scale = qds & 3u;
        uint32_t quality, dist;
        if (scale == 0u) { quality = qds >> 12; dist = (qds & 0xFFCu) * 2u; }
        else if (scale == 1u) { quality = (qds >> 13) << 1; dist = (qds & 0x1FFCu) * 3u + (2046u << 2); }
        else if (scale == 2u) { quality = (qds >> 14) << 2; dist = (qds & 0x3FFCu) * 4u + (8187u << 2); }
        else { quality = (qds >> 15) << 3; dist = (qds & 0x7FFCu) * 5u + (24567u << 2); }
        // raw distance for the smoothing pass: scale 0 -> bit 15 + value (<= 8184);
        // other scales only matter as "last distance" of a scale-0 successor, whose rule
        // |d - last| <= 8 can hold only if last <= 8192: store min(dist, 0x3FFF)
        sm[j] = scale == 0u ? (0x8000u | dist) : min(dist, 0x3FFFu);
        if (MODE == 2) dist = (uint32_t)((int)dist + (int)L.fin[i - chunk0 + j] - 4);  // (state 4 unless scale 0)
        rs |= ((((ang + inc) % (360 << 16)) < (inc * 2)) ? 1u : 0u) << j;
        nd[j] = make_node(ang >> 10, dist, quality, 0u);
      }
      if (MODE != 2) {
        if (i - chunk0 + G <= DecCfg<ANS>::kSmoothSlots)  // (always: a chunk is kUdChunk nodes)
          *reinterpret_cast<uint2 *>(&L.smooth[i - chunk0]) =
              make_uint2(sm[0] | (sm[G > 1 ? 1 : 0] << 16), sm[G > 2 ? 2 : 0] | (sm[G > 3 ? 3 : 0] << 16));
        if (MODE == 1) return;
      }
    }
#ifdef RPL_DEC_NOSTORE
    {
      uint32_t hsh = rs;
#pragma unroll
      for (uint32_t j = 0; j < G; ++j) hsh = hsh * 31u + nd[j].x + nd[j].y;
      if (hsh != 0xDEADBEEFu) return;
    }
#endif
    if (FUSE) {
      // where the nodes go: scan j = the last sync node at or before the node; a node in front
      // of the first sync node, in the scan still open at the end, in a scan a reset broke or
      // beyond scan_cap is not stored at all
      uint32_t j = g.scan_j;  // (-1: in front of the first sync node)
      // the usual case: all G nodes inside one stored scan, none of them its (special) last
      // slot -> 16-byte streaming stores, as in the unfused path
      if (G >= 4u && j != 0xFFFFFFFFu && j + 1u < f_nsync && L.sslot[j] != 0xFFFFu &&
          L.spos[j + 1u] >= i + G) {
        const uint32_t s0 = L.spos[j], total = L.spos[j + 1u] - s0;
        const uint32_t len = min(min(total, fz.max_count), fz.n_stride), off = i - s0;
        if (off + G < len) {  // (strictly below the last slot)
          uint2 *dst = fz.batch + ((size_t)b * fz.scan_cap + L.sslot[j]) * fz.n_stride + off;
          uint2 n0 = nd[0];
          if (FILTERED && off == 0u) n0.y = (n0.y & 0x00FFFFFFu) | (1u << 24);  // flag byte 2 -> 1
          typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4), aligned(8)));
          const nt_u32x4 va = {n0.x, n0.y, nd[G > 1 ? 1 : 0].x, nd[G > 1 ? 1 : 0].y};
          const nt_u32x4 vb = {nd[G > 2 ? 2 : 0].x, nd[G > 2 ? 2 : 0].y, nd[G > 3 ? 3 : 0].x, nd[G > 3 ? 3 : 0].y};
          if (G == 4u) {
            __builtin_nontemporal_store(va, reinterpret_cast<nt_u32x4 *>(dst));
            __builtin_nontemporal_store(vb, reinterpret_cast<nt_u32x4 *>(dst + 2));
          } else {  // ultra: 48 bytes per lane (plain stores, see the unfused path)
            const nt_u32x4 vc = {nd[G > 4 ? 4 : 0].x, nd[G > 4 ? 4 : 0].y, nd[G > 5 ? 5 : 0].x, nd[G > 5 ? 5 : 0].y};
            *reinterpret_cast<nt_u32x4 *>(dst) = va;
            *reinterpret_cast<nt_u32x4 *>(dst + 2) = vb;
            *reinterpret_cast<nt_u32x4 *>(dst + 4) = vc;
          }
          return;
        }
      }
#pragma unroll
      for (uint32_t jn = 0; jn < G; ++jn) {
        const uint32_t ii = i + jn;
        while (j + 1u < f_nsync && L.spos[j + 1u] <= ii) ++j;  // (j = -1: 0 < f_nsync)
        if (j == 0xFFFFFFFFu || j + 1u >= f_nsync) continue;
        const uint32_t slot = L.sslot[j];
        if (slot == 0xFFFFu) continue;
        const uint32_t s0 = L.spos[j], total = L.spos[j + 1u] - s0;
        const uint32_t full = min(total, fz.max_count), len = min(full, fz.n_stride);
        const uint32_t off = ii - s0;
        // ScanDataHolder keeps max_count nodes: the last slot of a longer scan holds its last node
        uint32_t pos = off;
        if (off + 1u >= len) {
          const uint32_t last_src = (len == full && total > full) ? total - 1u : len - 1u;
          if (off != last_src) continue;
          pos = len - 1u;
        }
        uint2 v = nd[jn];
        if (FILTERED && off == 0u) v.y = (v.y & 0x00FFFFFFu) | (1u << 24);  // flag byte 2 -> 1
        fz.batch[((size_t)b * fz.scan_cap + slot) * fz.n_stride + pos] = v;
      }
      return;
    }
    if (i + G <= n_out) {  // dense types: the (rare) sync flags are set in P4
      if (G == 1) {
        out[i] = nd[0];
      } else {
#pragma unroll
        for (uint32_t j = 0; j + 1u < G; j += 2u) {
          // 32 bytes per lane as two streaming stores (non-temporal: the node stream is written
          // once and read by a later kernel; -5 % dense, -13 % express, -6 % HQ).  Not for ultra:
          // its 48 bytes per lane complete a cache line over three instructions, and streaming
          // stores that leave lines half written cost it +46 %.
          if (G == 4u) {
            typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4), aligned(8)));
            const nt_u32x4 v = {nd[j].x, nd[j].y, nd[j + 1u].x, nd[j + 1u].y};
            __builtin_nontemporal_store(v, reinterpret_cast<nt_u32x4 *>(out + i + j));
          } else {
            const uint4 v = make_uint4(nd[j].x, nd[j].y, nd[j + 1u].x, nd[j + 1u].y);
            __builtin_memcpy(out + i + j, &v, 16);
          }
        }
      }
    } else {
#pragma unroll
      for (uint32_t j = 0; j < G; ++j)
        if (i + j < n_out) out[i + j] = nd[j];
    }
    if (rs) {
#pragma unroll
      for (uint32_t j = 0; j < G; ++j) {
        if (!((rs >> j) & 1u)) continue;
        if (FILTERED) {
          if (((i + j) >> 6) < raw_words) atomicOr(&Lraw[(i + j) >> 6], 1ull << ((i + j) & 63u));
        } else if (i + j < n_out) {
          report_sync(i + j);
        }
      }
    }
  };
  // U groups per lane and trip: all table look-ups, then all payload loads, then the decoding —
  // the loads of a trip travel together.  (Tried and dropped, profiles/r02/decode_study.txt:
  // issuing trip n+1's loads before trip n is decoded and stored.  The vector-memory counter
  // orders loads among loads and stores among stores but not one against the other, so a wave
  // that has stores in flight can only wait for "everything"; the dense kernel is
  // (decode + loads) + (stores) = 0.19 + 0.21 ms, not the larger of the two.)
  constexpr uint32_t U = ANS == RPLGPU_ANS_HQ ? 2u : 4u;  // (STG: 1, 2, 4, 8 measured, no difference)
  constexpr bool UD = ANS == RPLGPU_ANS_ULTRA_DENSE_CAPSULED;
  constexpr uint32_t kChunk = UD ? DecCfg<ANS>::kUdChunk : 0xFFFFFFFFu;  // nodes per pass (UD only)
  int last_sync_out = last_sync_in, last_dist_out = last_dist_in;
  int chunk_last = last_dist_in;  // ultra-dense: what the node in front of the chunk left behind
  for (chunk0 = 0; chunk0 < (UD ? max(carry_nodes, 1u) : 1u); chunk0 += (UD ? kChunk : 1u)) {
  const uint32_t g_lo = UD ? chunk0 / G : 0u;
  const uint32_t g_hi = UD ? min(n_groups, (chunk0 + kChunk) / G) : n_groups;
  using mode0 = std::integral_constant<int, 0>;
  using mode1 = std::integral_constant<int, 1>;
  using mode2 = std::integral_constant<int, 2>;
  // ultra-dense: the payload of ALL the thread's groups of the chunk stays in registers across the
  // smoothing pass (kUdChunk / G / NT groups: 8 at 256 threads, 3 registers each), so that the nodes
  // are decoded with their final distances and stored once.  (Storing first and correcting the
  // smoothed nodes afterwards — a coalesced 2-byte store per node — writes the lines twice: on a
  // near ring with range noise 0.67 against 0.54 ms for the fused form, 0.86 against 0.51 for this
  // one; only input that smooths nothing is 5-10 % faster that way.)
  constexpr bool kUdTwice = UD;
  constexpr uint32_t GPT = kUdTwice ? DecCfg<ANS>::kUdChunk / G / NT : 1u;
  static_assert(!kUdTwice || GPT * G * NT == DecCfg<ANS>::kUdChunk, "a chunk is a whole number of groups per thread");
  uint64_t ud_w[GPT];   // (only the payload is kept: the frame look-ups are cheap to repeat)
  uint32_t ud_w2[GPT];
  uint32_t t0 = g_lo + tid;
  if (kUdTwice) {
    GroupIn gi[GPT];
#pragma unroll
    for (uint32_t u = 0; u < GPT; ++u)
      if (t0 + u * NT < g_hi) locate(t0 + u * NT, gi[u]);
#pragma unroll
    for (uint32_t u = 0; u < GPT; ++u)
      if (t0 + u * NT < g_hi) fetch(gi[u]);
#pragma unroll
    for (uint32_t u = 0; u < GPT; ++u) {
      if (t0 + u * NT < g_hi) emit(t0 + u * NT, gi[u], mode1{});
      ud_w[u] = gi[u].w;
      ud_w2[u] = gi[u].w2;
    }
  }
  for (; !kUdTwice && t0 + (U - 1u) * NT < g_hi; t0 += U * NT) {  // full trips: no conditions
    GroupIn gi[U];
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) locate(t0 + u * NT, gi[u]);
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) fetch(gi[u]);
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) emit(t0 + u * NT, gi[u], mode0{});
  }
  for (; !kUdTwice && t0 < g_hi; t0 += NT) {
    GroupIn g1;
    locate(t0, g1);
    fetch(g1);
    emit(t0, g1, mode0{});
  }

#ifdef RPL_DEC_DBG
  __syncthreads();
  {  // (ultra-dense: P3 / P5 alternate chunk by chunk; the clocks accumulate each)
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    dbg_p3 += now - dbg_mark;
    dbg_mark = now;
  }
#endif
  // ---- P5 (ultra-dense): distance smoothing (:997-1003, :1020) -------------------------------
  // out_i = smooth(d_i, out_{i-1}) is a recurrence, but a smoothed value stays within +-4 of its
  // raw value, so "out_{i-1} - d_{i-1} + 4" is one of 9 states and every node is a map from
  // the 9 states of its predecessor to its own 9 states (36 bits).  Maps compose associatively:
  // each thread folds a contiguous segment of the stream into one map (tracking all 9 inputs),
  // a block scan composes the segment maps, and a second walk with the now known entry state
  // writes the smoothed distances.  Exact for any input, O(n / 256) steps per thread.
  if (UD && carry_nodes) {  // (indices below are relative to the chunk)
    __syncthreads();
#ifdef RPL_DEC_DBG
    dbg_u0 = __builtin_amdgcn_s_memtime();
#endif
    auto rawd = [&](uint32_t i) -> int { return (int)(L.smooth[i] & 0x3FFFu); };
    auto is_s0 = [&](uint32_t i) -> bool { return (L.smooth[i] & 0x8000u) != 0u; };
    // state of node i given the value `last` its predecessor left behind (:997-1003):
    //   scale 0 and last != 0 and |d - last| <= 8  ->  ((d + last) >> 1) - d + 4,  else 4
    // The same rule with the per-node part taken out: with x = last - d = (pd - 4 - d) + e for entry
    // state e (pd: the predecessor's raw distance), (d + last) >> 1 = d + (x >> 1), so the new state
    // is (x >> 1) + 4 when |x| <= 8 and last != 0 (e != 4 - pd), else 4.  One LDS word per node.
    struct NodeCtx {
      int base, ez;
      bool s0;
    };
    auto node_ctx = [&](uint32_t i, int pd) -> NodeCtx {
      const uint32_t w = L.smooth[i];
      NodeCtx c;
      c.s0 = (w & 0x8000u) != 0u;
      c.base = pd - 4 - (int)(w & 0x3FFFu);
      c.ez = 4 - pd;
      return c;
    };
    auto stepc = [](const NodeCtx &c, int e) -> int {
      const int x = c.base + e;
      return (c.s0 && (uint32_t)(x + 8) <= 16u && e != c.ez) ? (x >> 1) + 4 : 4;
    };
    constexpr unsigned long long kIdent = 0x876543210ull;
    auto compose = [](unsigned long long first, unsigned long long then) -> unsigned long long {
      unsigned long long r = 0;
#pragma unroll
      for (int sidx = 0; sidx < 9; ++sidx) {
        const uint32_t mid = (uint32_t)(first >> (4 * sidx)) & 15u;
        r |= ((then >> (4u * mid)) & 15ull) << (4 * sidx);
      }
      return r;
    };
    const uint32_t N = min(kChunk, carry_nodes - chunk0);
    const bool last_chunk = chunk0 + N == carry_nodes;
    const uint32_t seg = (N + NT - 1u) / NT;
    const uint32_t i_lo = min(tid * seg, N), i_hi = min(i_lo + seg, N);
    // node i has state st (a guess in pass A, overwritten by pass C where the guess was wrong): kept in
    // LDS; the nodes themselves are corrected in one coalesced pass at the end of the chunk.  (Until
    // round 4 every walking thread read-modify-wrote its own nodes as it went: a dependent global load
    // per smoothed node, 64 cache lines per instruction — invisible on the constant-distance bench
    // payload, which smooths nothing, and 3 x the kernel's time on distances that vary.)
    auto patch = [&](uint32_t i, int st, bool = false) {
      L.fin[i] = (uint8_t)st;
      if (i == N - 1u) {
        L.misc[7] = (uint32_t)(rawd(i) + st - 4);  // what this chunk leaves behind (st = 4 unless scale 0)
        if (last_chunk && is_s0(i)) L.misc[4] = (uint32_t)(rawd(i) + st - 4) | 0x80000000u;
      }
    };
    // pass A: the segment as one map.  All 9 entry states are tracked only until the map's
    // image has shrunk to two values (smoothing halves differences: a few nodes), then only
    // those two (which entry state leads to which is fixed from there on), and once they have
    // met — the first node that is not scale 0, follows a zero or jumps by more than 12 forgets
    // its predecessor; past ~2 m every node does — the segment's states no longer depend on
    // what came before it: they are final and written right away.  (A constant distance keeps
    // two states apart for ever: (x >> 1) has the fixed points 0 and -1.)
    // While the states still depend on the entry, the nodes are written with the states that
    // follow from entry state 4 ("the node in front was not smoothed", by far the most frequent):
    // pass C then only has to walk the segments whose true entry state is another one.
    unsigned long long M = kIdent;
    uint32_t i_open = i_hi;  // [i_lo, i_open): states that depend on the entry state (pass C)
    if (i_lo < i_hi) {
      int cur[9];
#pragma unroll
      for (int c = 0; c < 9; ++c) cur[c] = c;
      uint32_t i = i_lo;
      int va = 0, vb = 0;
      uint32_t in_b = 0;  // entry states that lead to vb (the others lead to va)
      bool two = false;
      int pd = i_lo ? rawd(i_lo - 1u) : chunk_last;  // the predecessor's raw distance, carried along
      for (; i < i_hi && !two; ++i) {
        const NodeCtx nc = node_ctx(i, pd);
        if (i == 0u) {  // (pd = what the previous chunk left behind: its final value, i.e. entry state 4)
          const int st0 = stepc(nc, 4);
#pragma unroll
          for (int c = 0; c < 9; ++c) cur[c] = st0;
        } else {
#pragma unroll
          for (int c = 0; c < 9; ++c) cur[c] = stepc(nc, cur[c]);
        }
        pd = pd - 4 - nc.base;  // (= this node's raw distance: base = pd - 4 - d)
        patch(i, cur[4]);  // (the guess)
        va = cur[0];
        vb = va;
#pragma unroll
        for (int c = 1; c < 9; ++c) vb = (cur[c] != va) ? cur[c] : vb;
        two = true;
        in_b = 0;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
          two = two && (cur[c] == va || cur[c] == vb);
          in_b |= (cur[c] != va ? 1u : 0u) << c;
        }
      }
      if (two) {
        const bool guess_b = ((in_b >> 4) & 1u) != 0u;  // entry state 4 leads to vb
        for (; i < i_hi && va != vb; ++i) {
          const NodeCtx nc = node_ctx(i, pd);
          va = stepc(nc, va);
          vb = stepc(nc, vb);
          pd = pd - 4 - nc.base;
          patch(i, guess_b ? vb : va);
        }
        if (va == vb) {
          i_open = i - 1u;  // node i-1 merged the last two states: it has state va whatever the entry
          // (already written: every chain, the guessed one included, has state va there)
          for (; i < i_hi; ++i) {
            const NodeCtx nc = node_ctx(i, pd);
            va = stepc(nc, va);
            pd = pd - 4 - nc.base;
            patch(i, va);
          }
          vb = va;
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) cur[c] = ((in_b >> c) & 1u) ? vb : va;
      }
      M = 0;
#pragma unroll
      for (int c = 0; c < 9; ++c) M |= (unsigned long long)(uint32_t)cur[c] << (4 * c);
    }
#ifdef RPL_DEC_DBG
    __syncthreads();
    { const unsigned long long now = __builtin_amdgcn_s_memtime(); dbg_ud[0] += now - dbg_u0; dbg_u0 = now; }
#endif
    // pass B: exclusive scan of the maps over the block (composition, earlier segment first).
    // A segment whose states merged has a CONSTANT map, and "anything, then a constant" is that
    // constant: when every segment of the wave is constant — distances that vary merge within a
    // few nodes — the inclusive scan is the lane's own map and nothing is composed (the 6-step scan
    // is ~400 of the pass's ~600 vector instructions per thread).
    auto is_const = [](unsigned long long m) -> bool { return m == (m & 15ull) * 0x111111111ull; };
    const bool wave_const = __all(is_const(M) ? 1 : 0) != 0;
    unsigned long long inc = M;
    if (!wave_const) {
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)inc, d, 64);
        const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), d, 64);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        if ((int)lane_id() >= d) inc = compose(other, inc);
      }
    }
    unsigned long long *wmap = reinterpret_cast<unsigned long long *>(L.tmp);  // one map per wave
    if (lane_id() == 63) wmap[wave_id()] = inc;
    __syncthreads();
    unsigned long long before = kIdent;  // everything in front of this wave
    for (uint32_t w = 0; w < wave_id(); ++w) {
      const unsigned long long m = wmap[w];  // (the same for the whole wave)
      before = is_const(m) ? m : compose(before, m);
    }
    unsigned long long excl;
    {
      const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)inc, 1, 64);
      const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), 1, 64);
      excl = lane_id() == 0 ? kIdent : (((unsigned long long)hi << 32) | lo);
    }
    if (wave_const) {
      if (lane_id() == 0) excl = before;  // (lanes >= 1: their predecessor's constant)
    } else {
      excl = compose(before, excl);
    }
    // every map in front of a non-empty segment starts with node 0's constant map, so any
    // entry state (take 4) gives the state its predecessor node really has
    int sp = (int)((excl >> (4 * 4)) & 15ull);
#ifdef RPL_DEC_DBG
    __syncthreads();
    { const unsigned long long now = __builtin_amdgcn_s_memtime(); dbg_ud[1] += now - dbg_u0; dbg_u0 = now; }
#endif
    // pass C: the (usually empty) head of the segment again, if its true entry state is not the
    // guessed one
    if (sp != 4) {
      int pd = i_lo ? rawd(i_lo - 1u) : chunk_last;
      for (uint32_t i = i_lo; i < min(i_open, i_hi); ++i) {
        const NodeCtx nc = node_ctx(i, pd);
        sp = stepc(nc, i == 0u ? 4 : sp);
        pd = pd - 4 - nc.base;
        // the true chain has met the guessed one: from here on what pass A wrote is true
        if (sp == (int)L.fin[i]) break;
        patch(i, sp, true);
      }
    }
    __syncthreads();
#ifdef RPL_DEC_DBG
    { const unsigned long long now = __builtin_amdgcn_s_memtime(); dbg_ud[2] += now - dbg_u0; dbg_u0 = now; }
#endif
    chunk_last = (int)L.misc[7];
    if (!last_chunk) {
    } else if (L.misc[4] & 0x80000000u) {
      last_dist_out = (int)(L.misc[4] & 0x7FFFFFFFu);
    } else {  // last node is not scale 0: _last_dist_q2 = its own distance (:1020)
      const uint32_t i = carry_nodes - 1u;
      const uint32_t e = i / NPF, pos = i - e * NPF;
      const uint32_t c = frame_at(Lemit[e] - 1u) + 10u + 5u * (pos >> 1);
      const uint32_t q4 = src.u8(c + 4u);
      const uint32_t qds = src.u16(c + 2u * (pos & 1u)) | (((pos & 1u) ? (q4 >> 4) : (q4 & 0xFu)) << 16);
      const uint32_t scale = qds & 3u;
      last_dist_out = scale == 1u ? (int)((qds & 0x1FFCu) * 3u + (2046u << 2))
                      : scale == 2u ? (int)((qds & 0x3FFCu) * 4u + (8187u << 2))
                                    : (int)((qds & 0x7FFCu) * 5u + (24567u << 2));
    }
  }
  if (kUdTwice) {  // the chunk's nodes with their final distances (states: L.fin, complete since the barrier above)
#pragma unroll
    for (uint32_t u = 0; u < GPT; ++u) {
      if (t0 + u * NT < g_hi) {
        GroupIn g;
        locate(t0 + u * NT, g);
        g.w = ud_w[u];
        g.w2 = ud_w2[u];
        emit(t0 + u * NT, g, mode2{});
      }
    }
  }

#ifdef RPL_DEC_DBG
  __syncthreads();
  {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    dbg_p5 += now - dbg_mark;
    dbg_mark = now;
    (void)dbg_u0;
  }
#endif
  }  // chunks
#ifdef RPL_DEC_DBG
  dbg_t[3] = dbg_t[2] + dbg_p3;
  dbg_t[4] = dbg_t[3];
  dbg_t[5] = dbg_t[4] + dbg_p5;  // (slot "P4" = 0, slot "P5" = the smoothing passes; the sync filter goes uncounted)
#endif

  // ---- P4 (dense / ultra-dense): the sync-bit filter, s_i = r_i & ~s_{i-1} -------------------
  if (FILTERED) {
    __syncthreads();
    if (!FUSE) {  // (FUSE: done in front of P3, the flags were set as the nodes were stored)
      sync_filter([&](uint32_t i) {
        if (i < n_out) {  // flag byte: sync | (!sync << 1) : 2 -> 1
          uint2 v = out[i];
          v.y = (v.y & 0x00FFFFFFu) | (1u << 24);
          out[i] = v;
          report_sync(i);
        }
      });
      __syncthreads();
    }
    if (carry_nodes) last_sync_out = (int)L.misc[3];
  }

  // ---- per-stream results ---------------------------------------------------------------------
  for (int d = 32; d > 0; d >>= 1) my_err += (uint32_t)__shfl_xor((int)my_err, d, 64);
  if (lane_id() == 0 && my_err) atomicAdd(&L.misc[5], my_err);
  __syncthreads();
  if (tid == 0) {
    uint32_t st = L.misc[0];
    if (!FUSE && carry_nodes > node_stride) st |= RPLGPU_SCAN_OUT_TRUNCATED;
    if (reset_at && carry_reset > reset_stride) st |= RPLGPU_STREAM_RESETS_TRUNCATED;
    if (FILTERED && carry_nodes > raw_words * 64u) st |= RPLGPU_STREAM_FRAMES_TRUNCATED;
    if (n_nodes) n_nodes[b] = bad_framing ? 0u : n_out;
    if (n_reset) n_reset[b] = bad_framing ? 0u : min(carry_reset, reset_at ? reset_stride : carry_reset);
#ifdef RPL_DEC_DBG
    if (reset_at && reset_stride >= 8)
    {
      for (int d = 0; d < 5; ++d) reset_at[(size_t)b * reset_stride + 3 + d] = (uint32_t)(dbg_t[d + 1] - dbg_t[d]);
      reset_at[(size_t)b * reset_stride + 2] = (uint32_t)(dbg_t[0] - dbg_start);
      if (reset_stride >= 16)
        for (int d = 0; d < 4; ++d) reset_at[(size_t)b * reset_stride + 8 + d] = (uint32_t)dbg_ud[d];
    }
#endif
    if (n_errors) n_errors[b] = L.misc[5];
    if (n_sync) n_sync[b] = bad_framing ? 0u : min(L.misc[6], sync_stride);
    if (my_sync && L.misc[6] > sync_stride) st |= RPLGPU_STREAM_FRAMES_TRUNCATED;
    if (status) status[b] = st;
    if (state_out) {
      state_out[4 * b] = bad_framing ? last_sync_in : last_sync_out;
      state_out[4 * b + 1] = bad_framing ? last_dist_in : last_dist_out;
      state_out[4 * b + 2] = 0;
      state_out[4 * b + 3] = 0;
    }
  }
  (void)n_emit;
}

// ------------------------------------------------------------------------------------------
// Scan assembly.  ScanDataHolder rules (src/sdk/src/sl_lidar_driver.cpp:272-315): a node with
// flag bit 0 closes the scan being built (if it holds anything) and opens the next; nodes before
// the first sync node are discarded; a rewind (scan-reset request, "before node p") empties the
// scan being built and everything up to the next sync node is discarded; a scan that reached
// max_count nodes keeps overwriting its last slot.  Hence scan j = [s_j, s_{j+1}) between two
// consecutive sync nodes is completed iff no reset position p satisfies s_j < p <= s_{j+1}, and
// it is nodes s_j .. s_j + max_count - 2 followed by node s_{j+1} - 1 when it is longer than
// max_count.  One workgroup per stream: compact the sync positions, judge the scans, copy.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kSegMaxSync = 2047;  // sync nodes of one stream per call (16 KiB of LDS)

__global__ __launch_bounds__(kDecBlock) void k_segment(
    const uint2 *__restrict__ nodes, uint32_t node_stride, const uint32_t *__restrict__ n_nodes,
    const uint32_t *__restrict__ reset_at, uint32_t reset_stride, const uint32_t *__restrict__ n_reset,
    uint32_t max_count, uint2 *__restrict__ out_nodes, uint32_t out_stride,
    uint32_t *__restrict__ scan_off, uint32_t scan_cap, uint32_t *__restrict__ n_scans,
    uint32_t *__restrict__ status, const uint32_t *__restrict__ sync_at, uint32_t sync_stride,
    const uint32_t *__restrict__ n_sync) {
  __shared__ uint32_t sync_pos[kSegMaxSync + 1];
  __shared__ uint32_t scan_len[kSegMaxSync];  // 0 = not completed
  __shared__ uint32_t tmp[40];
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  const uint2 *in = nodes + (size_t)b * node_stride;
  uint2 *out = out_nodes + (size_t)b * out_stride;
  const uint32_t n = min(n_nodes[b], node_stride);
  const uint32_t nr = (reset_at && n_reset) ? min(n_reset[b], reset_stride) : 0u;
  const uint32_t *rs = reset_at ? reset_at + (size_t)b * reset_stride : nullptr;
  uint32_t st = 0;

  // sync positions: sync nodes are rare (one per revolution), so the pass over the stream is
  // barrier-free — a ballot per wave, the few hits appended through an LDS counter — and the
  // short list is put in order afterwards by counting
  __shared__ uint32_t unsorted[kSegMaxSync + 1];
  if (tid == 0) tmp[32] = 0u;
  __syncthreads();
  if (sync_at) {  // the decoder's (unordered) list of sync nodes: no pass over the stream
    const uint32_t ns = min(n_sync[b], sync_stride);
    for (uint32_t j = tid; j < ns; j += kDecBlock) {
      const uint32_t i = sync_at[(size_t)b * sync_stride + j];
      if (i < n) {
        const uint32_t slot = atomicAdd(&tmp[32], 1u);
        if (slot <= kSegMaxSync) unsorted[slot] = i;
      }
    }
  } else {
    for (uint32_t i0 = 0; i0 < n; i0 += kDecBlock) {
      const uint32_t i = i0 + tid;
      const bool is = i < n && ((in[i].y >> 24) & 1u);
      if (__builtin_amdgcn_ballot_w64(is) != 0ull && is) {
        const uint32_t slot = atomicAdd(&tmp[32], 1u);
        if (slot <= kSegMaxSync) unsorted[slot] = i;
      }
    }
  }
  __syncthreads();
  uint32_t nsync = tmp[32];
  if (nsync > kSegMaxSync + 1u) {
    st |= RPLGPU_STREAM_FRAMES_TRUNCATED;  // (which of the sync nodes were kept is arbitrary)
    nsync = kSegMaxSync + 1u;
  }
  for (uint32_t j = tid; j < nsync; j += kDecBlock) {
    const uint32_t v = unsorted[j];
    uint32_t rank = 0;
    for (uint32_t q = 0; q < nsync; ++q) rank += unsorted[q] < v ? 1u : 0u;
    sync_pos[rank] = v;
  }
  __syncthreads();
  const uint32_t ncand = nsync ? nsync - 1u : 0u;  // scans closed by a following sync node
  // which of them are complete, and how long
  for (uint32_t j = tid; j < ncand; j += kDecBlock) {
    const uint32_t s0 = sync_pos[j], s1 = sync_pos[j + 1];
    // number of reset positions <= x (ascending list): lower bound by bisection
    auto count_le = [&](uint32_t x) -> uint32_t {
      uint32_t lo = 0, hi = nr;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (rs[mid] <= x) lo = mid + 1; else hi = mid;
      }
      return lo;
    };
    const bool broken = nr && (count_le(s1) != count_le(s0));
    scan_len[j] = broken ? 0u : min(s1 - s0, max_count);
  }
  __syncthreads();
  // copy pass: the workgroup copies one completed scan after the other (there is one
  // candidate per revolution, so the serial loop is short) and records the running offsets
  uint32_t off = 0, stored = 0, completed = 0;
  for (uint32_t j = 0; j < ncand; ++j) {
    const uint32_t len = scan_len[j];
    if (!len) continue;
    ++completed;
    if (stored >= scan_cap) continue;
    const uint32_t s0 = sync_pos[j], s1 = sync_pos[j + 1];
    for (uint32_t t = tid; t < len; t += kDecBlock) {
      const uint32_t src = (t == len - 1u && (s1 - s0) > len) ? s1 - 1u : s0 + t;
      if (off + t < out_stride) out[off + t] = in[src];
    }
    if (tid == 0) scan_off[(size_t)b * (scan_cap + 1u) + stored] = off;
    off += len;
    ++stored;
  }
  const uint32_t total_scans = stored;
  if (completed > scan_cap) st |= RPLGPU_STREAM_RESETS_TRUNCATED;
  if (off > out_stride) st |= RPLGPU_SCAN_OUT_TRUNCATED;
  if (tid == 0) {
    scan_off[(size_t)b * (scan_cap + 1u) + total_scans] = off;
    n_scans[b] = total_scans;
    if (status) status[b] = st;
  }
}

// The scan a stream was still building when a call ended (ScanDataHolder's operational buffer,
// src/sdk/src/sl_lidar_driver.cpp:272-310), carried to the next call (rplgpu_decode_scans_carry_dev):
// `in` holds it on entry — len_in[b] nodes at in + b * stride —, `out` receives the one open at
// the end of this call (two buffers: other workgroups of the stream still read `in`).  All null:
// no carry, a scan across the end of a call belongs to neither.
struct AssembleCarry {
  const uint2 *in;
  uint2 *out;
  const uint32_t *len_in;
  uint32_t *len_out;
  uint32_t stride;
};

// Scan assembly straight into the batch layout: the same rules as k_segment, but the sync nodes
// come from the decoder's list and completed scan s of stream b is written to batch slot
// g = b * scan_cap + s (n_per_scan[g] = its length, 0 for the slots a stream does not fill) —
// no pass over the node stream to find the sync nodes, no intermediate copy, no prefix over
// streams.  One workgroup per (slot, stream): each redoes the (tiny) judgement and copies its
// own scan.
__global__ __launch_bounds__(kDecBlock) void k_assemble(
    const uint2 *__restrict__ nodes, uint32_t node_stride, const uint32_t *__restrict__ n_nodes,
    const uint32_t *__restrict__ sync_at, uint32_t sync_stride, const uint32_t *__restrict__ n_sync,
    const uint32_t *__restrict__ reset_at, uint32_t reset_stride, const uint32_t *__restrict__ n_reset,
    uint32_t max_count, uint2 *__restrict__ batch, uint32_t n_stride, uint32_t scan_cap,
    uint32_t *__restrict__ n_per_scan, uint32_t *__restrict__ n_scans, uint32_t *__restrict__ status,
    const uint32_t *__restrict__ only, AssembleCarry cy) {
  __shared__ uint32_t sync_pos[kSegMaxSync + 1];
  __shared__ uint32_t unsorted[kSegMaxSync + 1];
  __shared__ uint32_t tmp[8];
  // (stream index fastest: consecutive workgroup ids go round the XCDs, and the workgroups that
  // have a scan to copy are mostly those of slot 0 — with the slot fastest and scan_cap = 4 all
  // of them landed on two of the eight XCDs: 0.50 ms instead of 0.2 ms per 0.5 GB)
  const uint32_t b = blockIdx.x, slot = blockIdx.y, tid = threadIdx.x;
  if (only && only[b] == 0u) return;  // (the fused decoder delivered this stream's scans itself)
  const uint2 *in = nodes + (size_t)b * node_stride;
  const uint32_t n = min(n_nodes[b], node_stride);
  const uint32_t nr = (reset_at && n_reset) ? min(n_reset[b], reset_stride) : 0u;
  const uint32_t *rs = reset_at ? reset_at + (size_t)b * reset_stride : nullptr;
  if (tid == 0) tmp[0] = 0u;
  __syncthreads();
  const uint32_t ns_in = min(n_sync[b], sync_stride);
  for (uint32_t j = tid; j < ns_in; j += kDecBlock) {
    const uint32_t i = sync_at[(size_t)b * sync_stride + j];
    if (i < n) {
      const uint32_t at = atomicAdd(&tmp[0], 1u);
      if (at <= kSegMaxSync) unsorted[at] = i;
    }
  }
  __syncthreads();
  uint32_t st = 0;
  uint32_t nsync = tmp[0];
  if (nsync > kSegMaxSync + 1u) {
    st |= RPLGPU_STREAM_FRAMES_TRUNCATED;
    nsync = kSegMaxSync + 1u;
  }
  for (uint32_t j = tid; j < nsync; j += kDecBlock) {
    const uint32_t v = unsorted[j];
    uint32_t rank = 0;
    for (uint32_t q = 0; q < nsync; ++q) rank += unsorted[q] < v ? 1u : 0u;
    sync_pos[rank] = v;
  }
  __syncthreads();
  const uint32_t ncand = nsync ? nsync - 1u : 0u;
  // the carried scan: it completes at this call's first sync node unless a reset request came
  // first (a request at position p clears the buffer before node p is pushed: any p <= s0)
  constexpr uint32_t kHead = 0xFFFFFFFEu;
  const uint32_t ccap = min(max_count, cy.stride);
  const uint32_t C = cy.in ? min(cy.len_in[b], ccap) : 0u;
  const uint2 *cin = cy.in ? cy.in + (size_t)b * cy.stride : nullptr;
  const bool head = C > 0u && nsync >= 1u && !(nr > 0u && rs[0] <= sync_pos[0]);
  // the slot-th completed scan: candidates in order (a stream holds a handful; one lane walks)
  if (tid == 0) {
    uint32_t completed = 0, mine = 0xFFFFFFFFu;
    if (head) {
      if (slot == 0u) mine = kHead;
      completed = 1u;
    }
    uint32_t ri = 0;  // reset positions <= s0 so far (both lists ascend)
    for (uint32_t j = 0; j < ncand; ++j) {
      const uint32_t s0 = sync_pos[j], s1 = sync_pos[j + 1];
      while (ri < nr && rs[ri] <= s0) ++ri;
      const bool broken = ri < nr && rs[ri] <= s1;  // a reset position p with s0 < p <= s1
      if (!broken) {
        if (completed == slot) mine = j;
        ++completed;
      }
    }
    tmp[1] = mine;
    tmp[2] = completed;
  }
  __syncthreads();
  const uint32_t mine = tmp[1], completed = tmp[2];
  const uint32_t g = b * scan_cap + slot;
  if (slot == 0 && tid == 0) {
    n_scans[b] = min(completed, scan_cap);
    if (completed > scan_cap) st |= RPLGPU_STREAM_RESETS_TRUNCATED;
  }
  uint32_t len = 0;
  if (mine == kHead) {
    // carried nodes, then this call's nodes in front of its first sync node; the buffer keeps
    // max_count nodes and goes on overwriting its last slot (:301-304)
    const uint32_t s0 = sync_pos[0], total = C + s0;
    const uint32_t full = min(total, max_count);
    len = min(full, n_stride);
    if (full > n_stride) st |= RPLGPU_SCAN_OUT_TRUNCATED;
    uint2 *dst = batch + (size_t)g * n_stride;
    const uint32_t body = len ? len - 1u : 0u;
    for (uint32_t e = tid; e < body; e += kDecBlock) dst[e] = e < C ? cin[e] : in[e - C];
    if (tid == 0 && len) {
      const uint32_t e = len - 1u;
      if (len == full && total > full) dst[e] = s0 ? in[s0 - 1u] : cin[C - 1u];
      else dst[e] = e < C ? cin[e] : in[e - C];
    }
  } else if (mine != 0xFFFFFFFFu) {
    const uint32_t s0 = sync_pos[mine], s1 = sync_pos[mine + 1];
    const uint32_t full = min(s1 - s0, max_count);  // ScanDataHolder keeps max_count nodes
    len = min(full, n_stride);
    if (full > n_stride) st |= RPLGPU_SCAN_OUT_TRUNCATED;
    uint2 *dst = batch + (size_t)g * n_stride;
    const uint2 *src = in + s0;
    // 16-byte copies, four in flight; a scan starts at any node, so the two sides are only
    // 8-byte aligned (the target runs in unaligned access mode: a 1 KiB wave access that starts
    // 8 bytes into a line touches nine lines instead of eight)
    const uint32_t body = len ? len - 1u : 0u;  // the last slot may come from the scan's end
    const uint32_t n4 = body >> 1;
    auto ld = [&](uint32_t t) { uint4 v; __builtin_memcpy(&v, src + 2u * t, 16); return v; };
    typedef uint32_t as_u32x4 __attribute__((ext_vector_type(4), aligned(8)));
    auto st16 = [&](uint32_t t, const uint4 &v) {  // (streaming store)
      const as_u32x4 w = {v.x, v.y, v.z, v.w};
      __builtin_nontemporal_store(w, reinterpret_cast<as_u32x4 *>(dst + 2u * t));
    };
    uint32_t t = tid;
    for (; t + 3u * kDecBlock < n4; t += 4u * kDecBlock) {
      const uint4 a = ld(t), c = ld(t + kDecBlock), d = ld(t + 2u * kDecBlock), e = ld(t + 3u * kDecBlock);
      st16(t, a);
      st16(t + kDecBlock, c);
      st16(t + 2u * kDecBlock, d);
      st16(t + 3u * kDecBlock, e);
    }
    for (; t < n4; t += kDecBlock) st16(t, ld(t));
    if (tid == 0 && (body & 1u)) dst[body - 1u] = src[body - 1u];
    if (tid == 0 && len) {
      // a scan longer than max_count kept overwriting its last slot: that slot holds the
      // scan's last node (src/sdk/src/sl_lidar_driver.cpp:286-292)
      const uint32_t last_src = (len == full && (s1 - s0) > full) ? s1 - 1u : s0 + len - 1u;
      dst[len - 1u] = in[last_src];
    }
  }
  if (cy.out && slot == 0u) {
    // the scan still open at the end of this call: from the last sync node on, unless a reset
    // request followed it (then nothing is open until the next sync node); without any sync node
    // the carried scan goes on growing — or was cleared by a reset request
    uint32_t from_c = 0u, from = n;  // nodes taken from the carried scan / first node taken from `in`
    bool open = false;
    if (nsync >= 1u) {
      const uint32_t s_last = sync_pos[nsync - 1u];
      open = !(nr > 0u && rs[nr - 1u] > s_last);
      from = s_last;
    } else if (C > 0u && nr == 0u) {
      open = true;
      from_c = C;
      from = 0u;
    }
    uint2 *cout = cy.out + (size_t)b * cy.stride;
    const uint32_t total = open ? from_c + (n - from) : 0u;
    const uint32_t full = min(total, ccap);
    const uint32_t body = full ? full - 1u : 0u;
    for (uint32_t e = tid; e < body; e += kDecBlock) cout[e] = e < from_c ? cin[e] : in[from + (e - from_c)];
    if (tid == 0) {
      if (full) {
        const uint32_t e = full - 1u;
        if (total > full) cout[e] = n > from ? in[n - 1u] : cin[C - 1u];
        else cout[e] = e < from_c ? cin[e] : in[from + (e - from_c)];
      }
      cy.len_out[b] = full;
      // (the open scan did not fit the carry buffer — only possible when carry_stride is smaller than
      // the scan cap: the next call's first scan will be shorter than ScanDataHolder's, and says so)
      if (total > full && ccap < max_count && status) atomicOr(&status[b], (uint32_t)RPLGPU_SCAN_OUT_TRUNCATED);
    }
  }
  if (tid == 0) {
    n_per_scan[g] = len;
    if (status && st) atomicOr(&status[b], st);  // (on top of what the decoder stored there)
  }
}

// Completed scans of every stream -> the fixed-stride batch (scan g at batch + g*n_stride) that
// rplgpu_ascend_batch_dev / rplgpu_laserscan_batch_dev / rplgpu_cloud_batch_dev take.  The
// global scan index is the stream-major running count (d_scan_base from a prefix sum over
// d_n_scans, computed by k_scan_base).  One workgroup per output scan slot.
__global__ void k_scan_base(const uint32_t *__restrict__ n_scans, uint32_t B,
                            uint32_t *__restrict__ scan_base) {
  // B is small (streams): one thread does the running sum
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t b = 0; b < B; ++b) {
      scan_base[b] = acc;
      acc += n_scans[b];
    }
    scan_base[B] = acc;
  }
}

__global__ __launch_bounds__(kDecBlock) void k_scans_to_batch(
    const uint2 *__restrict__ seg_nodes, uint32_t seg_stride, const uint32_t *__restrict__ scan_off,
    uint32_t scan_cap, uint32_t B, const uint32_t *__restrict__ scan_base,
    uint2 *__restrict__ batch, uint32_t n_stride, uint32_t max_scans,
    uint32_t *__restrict__ n_per_scan) {
  const uint32_t g = blockIdx.x;  // global scan index
  if (g >= min(scan_base[B], max_scans)) return;
  uint32_t lo = 0, hi = B;  // the stream b with scan_base[b] <= g < scan_base[b+1]
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (scan_base[mid] <= g) lo = mid; else hi = mid;
  }
  const uint32_t b = lo, s = g - scan_base[b];
  const uint32_t *so = scan_off + (size_t)b * (scan_cap + 1u);
  const uint32_t o0 = so[s], len = min(so[s + 1] - o0, n_stride);
  const uint2 *src = seg_nodes + (size_t)b * seg_stride + o0;
  uint2 *dst = batch + (size_t)g * n_stride;
  for (uint32_t t = threadIdx.x; t < len; t += kDecBlock) dst[t] = src[t];
  if (threadIdx.x == 0) n_per_scan[g] = len;
}

// ---- launchers ------------------------------------------------------------------------------
// The LDS-staged instances (capsule streams, nodes out; back to back or with frame offsets):
// 512-thread workgroups, two per CU — for calls whose streams fit half the CU's LDS together with their tables (max_frames up to
// decode_staged_frames(ans): 830 DenseBoost frames, 876 express, 557 ultra, 339 ultra-dense).  *done stays false
// otherwise and the plain kernel takes the call.  Measured and not instantiated
// (profiles/r04/decode_staged_r04.txt): 1024-thread workgroups (one per CU), staging in windows
// for longer streams, and the fused decode->scans kernel staged — all slower than the plain kernel.
constexpr uint32_t kLdsPerCu = 160u * 1024u, kLdsGranule = 1280u;
template <int ANS>
static bool dec_stage_fits(uint32_t mf) {  // two workgroups of the staged instance on a CU?
  const uint32_t need = (uint32_t)sizeof(DecodeLds<ANS, true>) + dec_stage_layout(ANS, mf).total;
  return 2u * ((need + kLdsGranule - 1u) / kLdsGranule * kLdsGranule) <= kLdsPerCu;
}
template <int ANS>
static uint32_t dec_staged_frames() {  // the largest max_frames the staged instance takes
  uint32_t lo = 0, hi = DecCfg<ANS>::kMaxFrames;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1u) >> 1;
    if (dec_stage_fits<ANS>(mid)) lo = mid; else hi = mid - 1u;
  }
  return lo;
}
uint32_t decode_staged_frames(int ans) {
  switch (ans) {
    case RPLGPU_ANS_CAPSULED: return dec_staged_frames<RPLGPU_ANS_CAPSULED>();
    case RPLGPU_ANS_CAPSULED_ULTRA: return dec_staged_frames<RPLGPU_ANS_CAPSULED_ULTRA>();
    case RPLGPU_ANS_DENSE_CAPSULED: return dec_staged_frames<RPLGPU_ANS_DENSE_CAPSULED>();
    case RPLGPU_ANS_ULTRA_DENSE_CAPSULED: return dec_staged_frames<RPLGPU_ANS_ULTRA_DENSE_CAPSULED>();
    default: return 0u;
  }
}
template <bool FRAMED, bool FUSE = false>
static hipError_t launch_decode_staged(hipStream_t s, int ans, const uint8_t *bytes, uint64_t stream_stride,
                                       const uint32_t *frame_off, const uint8_t *gap,
                                       const uint32_t *n_frames, uint32_t max_frames, uint32_t B,
                                       uint32_t sample_duration_us, const int32_t *state_in,
                                       int32_t *state_out, uint2 *nodes, uint32_t node_stride,
                                       uint32_t *n_nodes, uint32_t *reset_at, uint32_t reset_stride,
                                       uint32_t *n_reset, uint32_t *n_errors, uint32_t *status,
                                       uint32_t *sync_at, uint32_t sync_stride, uint32_t *n_sync,
                                       const DecFuse &fz, bool *done) {
  *done = false;
  if (max_frames == 0u) return hipSuccess;
  auto go = [&](auto kfn, uint32_t nt, bool fits, uint32_t mf_cap) -> hipError_t {
    if (!fits) return hipSuccess;
    const DecStageLayout lay = dec_stage_layout(ans, max_frames < mf_cap ? max_frames : mf_cap);
    // The attribute is per kernel FUNCTION and process-wide: set with a call's own size, two handles
    // decoding on two host threads could interleave set(large) set(small) launch(large), and the
    // launch would fail.  It is set ONCE per instance and device, to the most any call of the
    // instance can ask for (the staged frame count of the answer type); a launch passes its own size.
    {
      static std::mutex mu;
      static std::set<std::pair<const void *, int>> done_for;
      // (the attribute lands on the CURRENT device: it must be the one the launch goes to — the
      // stream's.  ADVICE r5: the result of hipGetDevice was ignored and a caller that had not made
      // the handle's device current would have had the attribute recorded for the wrong device.)
      int dev = -1;
      if (const hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
      if (s) {
        hipDevice_t sdev;
        if (hipStreamGetDevice(s, &sdev) == hipSuccess) {
          if ((int)sdev != dev) return hipErrorInvalidDevice;
        } else {
          (void)hipGetLastError();
        }
      }
      std::lock_guard<std::mutex> lk(mu);
      const auto key = std::make_pair(reinterpret_cast<const void *>(kfn), dev);
      if (!done_for.count(key)) {
        const DecStageLayout most = dec_stage_layout(ans, decode_staged_frames(ans));
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)(most.total > lay.total ? most.total : lay.total));
        if (e != hipSuccess) return e;
        done_for.insert(key);
      }
    }
    hipLaunchKernelGGL(kfn, dim3(B), dim3(nt), lay.total, s, bytes, stream_stride,
                       frame_off, gap, n_frames, max_frames,
                       sample_duration_us, state_in, state_out, nodes, node_stride, n_nodes, reset_at,
                       reset_stride, n_reset, n_errors, status, sync_at, sync_stride, n_sync, fz);
    *done = true;
    return hipGetLastError();
  };
#define RPL_STAGED(A)                                                                           \
  case A:                                                                                       \
    return go(k_decode<A, FRAMED, FUSE, true, 512>, 512u,                                       \
              dec_stage_fits<A>(max_frames < DecCfg<A>::kMaxFrames ? max_frames : DecCfg<A>::kMaxFrames), \
              DecCfg<A>::kMaxFrames);
  switch (ans) {
    RPL_STAGED(RPLGPU_ANS_CAPSULED)
    RPL_STAGED(RPLGPU_ANS_CAPSULED_ULTRA)
    RPL_STAGED(RPLGPU_ANS_DENSE_CAPSULED)
    RPL_STAGED(RPLGPU_ANS_ULTRA_DENSE_CAPSULED)
    default: return hipSuccess;
  }
#undef RPL_STAGED
}

hipError_t launch_decode(hipStream_t s, int ans, const uint8_t *bytes, uint64_t stream_stride,
                         const uint32_t *frame_off, const uint8_t *gap, const uint32_t *n_frames,
                         uint32_t max_frames, uint32_t B, uint32_t sample_duration_us,
                         const int32_t *state_in, int32_t *state_out, void *nodes,
                         uint32_t node_stride, uint32_t *n_nodes, uint32_t *reset_at,
                         uint32_t reset_stride, uint32_t *n_reset, uint32_t *n_errors,
                         uint32_t *status, uint32_t *sync_at, uint32_t sync_stride,
                         uint32_t *n_sync, const uint32_t *only, bool staged_ok, uint32_t *stage_todo) {
  if (B == 0) return hipSuccess;
  DecFuse fz{};
  fz.only = only;
  fz.stage_todo = stage_todo;
#define RPL_LAUNCH_DEC3(A, F, Z)                                                                 \
  hipLaunchKernelGGL((k_decode<A, F, Z>), dim3(B), dim3(kDecBlock), 0, s, bytes, stream_stride, \
                     frame_off, gap, n_frames, max_frames, sample_duration_us, state_in,         \
                     state_out, (uint2 *)nodes, node_stride, n_nodes, reset_at, reset_stride,    \
                     n_reset, n_errors, status, sync_at, sync_stride, n_sync, fz)
#define RPL_LAUNCH_DEC(A)                          \
  do {                                             \
    if (frame_off) RPL_LAUNCH_DEC3(A, true, false); \
    else RPL_LAUNCH_DEC3(A, false, false);         \
  } while (0)
  if (staged_ok && !frame_off) {
    bool done = false;
    const hipError_t e = launch_decode_staged<false>(s, ans, bytes, stream_stride, nullptr, nullptr, n_frames,
        max_frames, B, sample_duration_us, state_in, state_out, (uint2 *)nodes, node_stride, n_nodes,
        reset_at, reset_stride, n_reset, n_errors, status, sync_at, sync_stride, n_sync, fz, &done);
    if (done) return e;
  } else if (staged_ok && stage_todo) {
    // frame offsets given: the staged kernel takes every stream whose frames lie the way it needs
    // them and lists the others in stage_todo; the plain kernel below then runs over that list
    bool done = false;
    const hipError_t e = launch_decode_staged<true>(s, ans, bytes, stream_stride, frame_off, gap, n_frames,
        max_frames, B, sample_duration_us, state_in, state_out, (uint2 *)nodes, node_stride, n_nodes,
        reset_at, reset_stride, n_reset, n_errors, status, sync_at, sync_stride, n_sync, fz, &done);
    if (e != hipSuccess) return e;
    if (done) fz.only = stage_todo;  // (0 where `only` was 0: those streams were someone else's)
  }
  switch (ans) {
    case RPLGPU_ANS_MEASUREMENT: RPL_LAUNCH_DEC(RPLGPU_ANS_MEASUREMENT); break;
    case RPLGPU_ANS_CAPSULED: RPL_LAUNCH_DEC(RPLGPU_ANS_CAPSULED); break;
    case RPLGPU_ANS_HQ: RPL_LAUNCH_DEC(RPLGPU_ANS_HQ); break;
    case RPLGPU_ANS_CAPSULED_ULTRA: RPL_LAUNCH_DEC(RPLGPU_ANS_CAPSULED_ULTRA); break;
    case RPLGPU_ANS_DENSE_CAPSULED: RPL_LAUNCH_DEC(RPLGPU_ANS_DENSE_CAPSULED); break;
    case RPLGPU_ANS_ULTRA_DENSE_CAPSULED: RPL_LAUNCH_DEC(RPLGPU_ANS_ULTRA_DENSE_CAPSULED); break;
    default: return hipErrorInvalidValue;
  }
#undef RPL_LAUNCH_DEC
  return hipGetLastError();
}

bool decode_fusable(int ans) {
  return ans == RPLGPU_ANS_CAPSULED || ans == RPLGPU_ANS_CAPSULED_ULTRA ||
         ans == RPLGPU_ANS_DENSE_CAPSULED || ans == RPLGPU_ANS_ULTRA_DENSE_CAPSULED;
}

// The fused decoder (express / ultra / dense): completed scans straight into batch slots; d_todo[b]
// = 1 for a stream it could not take (more sync nodes or reset requests than its tables hold).
hipError_t launch_decode_fused(hipStream_t s, int ans, const uint8_t *bytes, uint64_t stream_stride,
                               const uint32_t *frame_off, const uint8_t *gap,
                               const uint32_t *n_frames, uint32_t max_frames, uint32_t B,
                               uint32_t sample_duration_us, const int32_t *state_in,
                               int32_t *state_out, uint32_t *n_errors, uint32_t *status,
                               uint32_t max_count, void *batch, uint32_t n_stride,
                               uint32_t scan_cap, uint32_t *n_per_scan, uint32_t *n_scans,
                               uint32_t *todo, bool staged_ok) {
  if (B == 0) return hipSuccess;
  DecFuse fz{};
  fz.batch = (uint2 *)batch;
  fz.n_per_scan = n_per_scan;
  fz.n_scans = n_scans;
  fz.todo = todo;
  fz.n_stride = n_stride;
  fz.scan_cap = scan_cap;
  fz.max_count = max_count;
  uint2 *nodes = nullptr;
  const uint32_t node_stride = 0xFFFFFFFFu, reset_stride = 0, sync_stride = 0;
  uint32_t *n_nodes = nullptr, *reset_at = nullptr, *n_reset = nullptr, *sync_at = nullptr, *n_sync = nullptr;
  if (frame_off && staged_ok) {
    // with frame offsets the staged form wins (the plain kernel's per-group offset loads queue
    // behind the node stores: 0.33 against 0.24 ms without offsets); a stream whose frames do
    // not lie the way it needs them raises todo[b] like one with too many sync nodes
    bool done = false;
    const hipError_t e = launch_decode_staged<true, true>(s, ans, bytes, stream_stride, frame_off, gap,
        n_frames, max_frames, B, sample_duration_us, state_in, state_out, nodes, node_stride, n_nodes,
        reset_at, reset_stride, n_reset, n_errors, status, sync_at, sync_stride, n_sync, fz, &done);
    if (done || e != hipSuccess) return e;
  }
#define RPL_LAUNCH_FUSED(A)                         \
  do {                                              \
    if (frame_off) RPL_LAUNCH_DEC3(A, true, true);  \
    else RPL_LAUNCH_DEC3(A, false, true);           \
  } while (0)
  switch (ans) {
    case RPLGPU_ANS_CAPSULED: RPL_LAUNCH_FUSED(RPLGPU_ANS_CAPSULED); break;
    case RPLGPU_ANS_CAPSULED_ULTRA: RPL_LAUNCH_FUSED(RPLGPU_ANS_CAPSULED_ULTRA); break;
    case RPLGPU_ANS_DENSE_CAPSULED: RPL_LAUNCH_FUSED(RPLGPU_ANS_DENSE_CAPSULED); break;
    case RPLGPU_ANS_ULTRA_DENSE_CAPSULED: RPL_LAUNCH_FUSED(RPLGPU_ANS_ULTRA_DENSE_CAPSULED); break;
    default: return hipErrorInvalidValue;
  }
#undef RPL_LAUNCH_FUSED
#undef RPL_LAUNCH_DEC3
  return hipGetLastError();
}

hipError_t launch_segment(hipStream_t s, const void *nodes, uint32_t node_stride,
                          const uint32_t *n_nodes, const uint32_t *reset_at, uint32_t reset_stride,
                          const uint32_t *n_reset, uint32_t B, uint32_t max_count, void *out_nodes,
                          uint32_t out_stride, uint32_t *scan_off, uint32_t scan_cap,
                          uint32_t *n_scans, uint32_t *status, const uint32_t *sync_at,
                          uint32_t sync_stride, const uint32_t *n_sync) {
  if (B == 0) return hipSuccess;
  hipLaunchKernelGGL(k_segment, dim3(B), dim3(kDecBlock), 0, s, (const uint2 *)nodes, node_stride,
                     n_nodes, reset_at, reset_stride, n_reset, max_count, (uint2 *)out_nodes,
                     out_stride, scan_off, scan_cap, n_scans, status, sync_at, sync_stride, n_sync);
  return hipGetLastError();
}

hipError_t launch_assemble(hipStream_t s, const void *nodes, uint32_t node_stride,
                           const uint32_t *n_nodes, const uint32_t *sync_at, uint32_t sync_stride,
                           const uint32_t *n_sync, const uint32_t *reset_at, uint32_t reset_stride,
                           const uint32_t *n_reset, uint32_t B, uint32_t max_count, void *batch,
                           uint32_t n_stride, uint32_t scan_cap, uint32_t *n_per_scan,
                           uint32_t *n_scans, uint32_t *status, const uint32_t *only,
                           const void *carry_in, void *carry_out, const uint32_t *carry_len_in,
                           uint32_t *carry_len_out, uint32_t carry_stride) {
  if (B == 0 || scan_cap == 0) return hipSuccess;
  const AssembleCarry cy{(const uint2 *)carry_in, (uint2 *)carry_out, carry_len_in, carry_len_out,
                         carry_stride};
  hipLaunchKernelGGL(k_assemble, dim3(B, scan_cap), dim3(kDecBlock), 0, s, (const uint2 *)nodes,
                     node_stride, n_nodes, sync_at, sync_stride, n_sync, reset_at, reset_stride,
                     n_reset, max_count, (uint2 *)batch, n_stride, scan_cap, n_per_scan, n_scans,
                     status, only, cy);
  return hipGetLastError();
}
uint32_t decode_sync_stride() { return kSegMaxSync + 1u; }

hipError_t launch_scans_to_batch(hipStream_t s, const void *seg_nodes, uint32_t seg_stride,
                                 const uint32_t *scan_off, uint32_t scan_cap,
                                 const uint32_t *n_scans, uint32_t B, uint32_t *scan_base,
                                 void *batch, uint32_t n_stride, uint32_t max_scans,
                                 uint32_t *n_per_scan) {
  if (B == 0) return hipSuccess;
  hipLaunchKernelGGL(k_scan_base, dim3(1), dim3(64), 0, s, n_scans, B, scan_base);
  if (max_scans == 0) return hipGetLastError();
  hipLaunchKernelGGL(k_scans_to_batch, dim3(max_scans), dim3(kDecBlock), 0, s,
                     (const uint2 *)seg_nodes, seg_stride, scan_off, scan_cap, B, scan_base,
                     (uint2 *)batch, n_stride, max_scans, n_per_scan);
  return hipGetLastError();
}

uint32_t decode_max_frames(int ans) {
  return ans == RPLGPU_ANS_MEASUREMENT            ? 0x7FFFFFFFu
         : ans == RPLGPU_ANS_ULTRA_DENSE_CAPSULED ? kUdMaxFrames
                                                  : kDecMaxFrames;
}

}  // namespace rpl
