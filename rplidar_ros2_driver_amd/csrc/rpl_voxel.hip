// rpl_voxel.hip — k_cloud_voxel: raw scan -> clipped, voxel-downsampled PointCloud2
// (extensions E1 + E2 + E4 of SURVEY.md §8 a-ext) in ONE streaming pass over the packed
// 8-byte nodes (reference layout src/sdk/include/sl_lidar_cmd.h:272-278).
//
// Geometry: persistent workgroups of RPL_VOXEL_THREADS threads (default 1024 = 16 wave64, one per
// compute unit; 512 = two per compute unit with a 4096-record queue each), every workgroup owns
// one scan (or one E8 group of scans) at a time and draws the next from a device-side counter.
//
// Phase S (streaming, straight-line code):
//   * a wave works on BLOCKS of 128 consecutive samples; one buffer_load_dwordx4 per lane = TWO
//     consecutive samples; the (cos, sin) table entries of the next block are fetched one block
//     ahead, the raw pairs two blocks ahead;
//   * keep mask = one unsigned interval test on dist_mm_q2 (host-derived, rpl_device.hpp); a
//     dropped sample is computed with dist 0, which yields zero offsets without any masking;
//   * x/y arithmetic on packed fp32 pairs (v_pk_mul_f32 / v_pk_fma_f32): polar->XY, the two
//     validated mul+FMA divides by the leaf, the exact in-cell remainder;
//   * cell key without integer conversions: (floor + 2^23 + 32768) puts iy/ix + 32768 into
//     the low mantissa bits, one v_perm_b32 packs (iy, ix) into the 32-bit sort key;
//   * a smooth ring stays in a 5 cm cell for ~10-200 samples, so runs of equal keys are
//     aggregated before anything is stored: the samples of a lane are summed, the lane totals go
//     through three plain DPP prefix scans, and a lane writes ONE 16-byte record {key, prefix_x,
//     prefix_y, prefix(count << 16 | intensity)} wherever a run ends.  The record holds the block
//     PREFIX, not the run sum: the run sum is prefix(this record) - prefix(the queue entry before
//     it), recovered when the record is read, which removes every cross-lane gather
//     (ds_bpermute) from the hot loop.  Every block's records are preceded by one MARKER entry
//     (empty key, zero prefix), so "the entry before it" is always right and no record carries a
//     first-of-block flag (round 2: two compares, two selects and two ORs per block).
// The record queue: the first kRecCap entries of a scan live in LDS; whatever comes after them
// (noisy or random scans, very large rings) goes to a per-workgroup record store in global
// memory (it stays in L2).  A scan is therefore streamed ONCE whatever its content.
// Phase R (per key band, regular data-parallel passes over <= kRecCap queue entries):
//   prefix -> run sums, counting sort by row + rank inside the row -> records in (iy, ix)
//   order -> segmented integer sums over equal keys -> one output point per cell.
// A scan whose records all fit the LDS queue is one band and never touches the record store.
// Otherwise the LDS part joins the rest in the store and the key range is bisected until a band
// fits; every band re-reads the RECORDS (16 B per run, from L2), not the scan.
//
// E5 (radius outlier removal) INSIDE the pass — the ROR instance of the kernel (round 6; config 5 reads a scan
// once): a block owns 124 of the 128 samples it reads, a kept sample with `ror_k` of its four nearest indices
// within r survives at once (four distance tests per lane, the lane-behind operands by DPP, the counts from
// the scalar compare masks), one without is run past the wave's 128 samples, then listed for ror_resolve
// (64 indices either side, then the whole scan); a sample settled late joins the queue as a record of its
// own.  A scan with more open samples than that is handed to k_ror_mask + this kernel's masked instance
// through a list of work items (RORM 1 / 2 below; rplgpu_api.hip voxel_with_ror).
//
// Fixed point: offset = (x - ix*leaf) * 2^K with 2^-K = ulp(leaf) (K = 28 for 5 cm), summed as
// wrapping 32-bit integers (a block prefix stays below 2^32; the tiny negative offsets of a
// quotient that rounded up to the next integer are recognised by their top bits when a run sum
// is read).  One fp32 FMA yields x - ix*leaf EXACTLY whenever x is a multiple of 2^-K
// (|x| >= 3 cm at K = 28), so the integer sums are exact and sum/count reproduces the
// spec's fp64 running sum bit for bit; closer to the axes the per-point error is
// <= 2^-K m (3.7e-9 m), far below the 1e-6 m bar.  Integer sums are order
// independent, so the kernel is run-to-run deterministic although the queue order is not.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "rpl_device.hpp"
#include "rpl_launch.hpp"

namespace rpl {

#ifndef RPL_VOXEL_THREADS
#define RPL_VOXEL_THREADS 1024
#endif
#ifndef RPL_VOXEL_WGS_PER_CU  // resident workgroups per compute unit the LDS layout is sized for
#define RPL_VOXEL_WGS_PER_CU (RPL_VOXEL_THREADS == 512 ? 2 : 1)
#endif
constexpr int kVB = RPL_VOXEL_THREADS;                   // threads of a voxel workgroup
constexpr int kVW = kVB / 64;                            // its waves
constexpr int kVWG = RPL_VOXEL_WGS_PER_CU;               // workgroups per compute unit
constexpr uint32_t kRecCap = kVWG == 2 ? 4096u : 7168u;  // run records the LDS queue holds
constexpr uint32_t kRecPerThread = kRecCap / kVB;        // 7 (one workgroup per CU), 8 or 4
constexpr uint32_t kRowCap = kVWG == 2 ? 1024u : 2u * kVB;  // rows the counting sort of a band handles
constexpr uint32_t kRowsPerThread = kRowCap / kVB;       // rows a thread scans: 2, or 1 (1024 threads x 2)
constexpr uint32_t kBucketCap = kRecCap + 3u * kRowCap;  // row lists padded to multiples of 4
constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;
constexpr float kKeyMagic = 8421376.0f;                  // 2^23 + 32768
static_assert(kRowsPerThread == 1u || kRowsPerThread == 2u, "the row scan handles one or two rows per thread");
static_assert(kBucketCap * 4u <= kRecCap * 16u, "the row lists live inside the record array");
static_assert(kRecCap * 2u <= 2u * kRowCap * 4u, "the cell-head list (u16) lives in the row arrays");

typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifndef RPL_VOXEL_AHEAD
#define RPL_VOXEL_AHEAD 2
#endif
#ifndef RPL_VOXEL_GAHEAD
#define RPL_VOXEL_GAHEAD 1  // (cos, sin) gathers: blocks in front of the block being aggregated
#endif
#ifndef RPL_RAW_AUX
#define RPL_RAW_AUX 2  // cache policy of the raw-pair loads (bit 0 glc, bit 1 slc): streamed once
#endif

// Three independent inclusive wave64 prefix sums, interleaved so that every DPP read is two
// issue slots behind the write it depends on (no s_nop needed) and every step is ONE
// v_add_u32_dpp (hipcc splits the row_bcast:31 step into mov + mov_dpp + add).
__device__ __forceinline__ void wave_incl_scan3_dpp(uint32_t &a, uint32_t &b, uint32_t &c) {
#define RPL_STEP(CTRL)                                   \
  "v_add_u32_dpp %0, %0, %0 " CTRL "\n"     \
  "v_add_u32_dpp %1, %1, %1 " CTRL "\n"     \
  "v_add_u32_dpp %2, %2, %2 " CTRL "\n"
  // s_nop 1: the operands may have been written by the VALU instruction just before (a DPP
  // read needs two wait states after a VALU write; the assembler does not add them here)
  asm volatile("s_nop 1\n" RPL_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
               RPL_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
               RPL_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
               RPL_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
               RPL_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
               RPL_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
               : "+v"(a), "+v"(b), "+v"(c));
#undef RPL_STEP
}

struct VoxelLds {
  // S: {key, prefix_x, prefix_y, first-of-pass flag (bit 24) | prefix(count<<16 | intensity)}
  // R: first the (ix << 16 | record index) lists of the rows (the records sit in registers
  //    then), afterwards the records in (iy, ix) order {key, sum_x, sum_y, count<<16 | isum}
  uint4 rec[kRecCap];
  // rows[0 .. kRowCap): (first padded slot << 16) | first compact slot of a row;
  // rows[kRowCap .. 2 kRowCap): one past the last used padded slot; later the u16 cell heads
  uint32_t rows[2u * kRowCap];
  uint32_t band_lo[66], band_hi[66];
  uint32_t misc[16];  // 0 queue tail, 1 status, 2 overflow, 3 sp, 4 rowmin, 5 rowmax, 7 out_base
  uint32_t tmp[32];
  double rcp[256];  // RN(1/count) for count < 256 (copied once per workgroup from the host table)
};
static_assert(kVWG != 2 || sizeof(VoxelLds) <= 80u * 1024u, "two workgroups per compute unit");

// E5 inside the voxel kernel (the ROR instance, round 6): what the streaming pass leaves for the
// exact steps behind it — the samples whose index neighbours did not settle them.
constexpr uint32_t kRorTodoCap = 256;  // unsettled samples of a scan the kernel resolves itself
constexpr uint32_t kRorFewCap = 8;     // ... of which may still be unsettled after the +-64 window
struct RorSide {
  uint32_t n_todo, n_todo2;
  uint32_t few_cnt[kRorFewCap];
  float2 few_pt[kRorFewCap];
  uint16_t todo[kRorTodoCap];
  uint16_t todo2[kRorFewCap];
};

// Exclusive scan of one value per thread over the kVB threads.  `tmp` = kVW + 1 words of LDS.
__device__ __forceinline__ uint32_t vx_excl_scan(uint32_t v, uint32_t *tmp, uint32_t *total) {
  const uint32_t inc = wave_incl_scan_fast(v);
  if (lane_id() == 63) tmp[wave_id()] = inc;
  __syncthreads();
  if (threadIdx.x < 64) {
    const uint32_t w = (threadIdx.x < kVW) ? tmp[threadIdx.x] : 0u;
    const uint32_t ws = wave_incl_scan_fast(w);
    if (threadIdx.x < kVW) tmp[threadIdx.x] = ws - w;  // exclusive wave base
    if (threadIdx.x == kVW - 1) tmp[kVW] = ws;
  }
  __syncthreads();
  *total = tmp[kVW];
  return tmp[wave_id()] + inc - v;
}

// a / d without v_div_scale / v_rcp / v_div_fmas / v_div_fixup: `rd` = RN(1/d), one
// multiply, the exact FMA remainder and one FMA correction (Markstein: a faithful first
// quotient plus the correctly rounded reciprocal give the correctly rounded quotient).
// The claim is not taken on faith: k_validate_div below compares it bit for bit with the
// IEEE divide over the whole operand range for the divisor in use, on this device, and
// the kernels only take this path after that check passed.  (The sign of a zero quotient
// may differ; every user takes floor() of it, where -0 and +0 coincide.)
__device__ __forceinline__ float div_by(float a, float d, float rd) {
  float q = a * rd;
  float e = fmaf(-q, d, a);
  return fmaf(e, rd, q);
}
__device__ __forceinline__ f2 div_by2(f2 a, float d, float rd) {
  const f2 dd = {d, d}, rr = {rd, rd};
  f2 q = a * rr;
  f2 e = __builtin_elementwise_fma(-q, dd, a);
  return __builtin_elementwise_fma(e, rr, q);
}

// 16-byte stores the compiler must not merge or re-type (see voxel_block_pass)
__device__ __forceinline__ void lds_store128(uint32_t lds_byte_addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(lds_byte_addr), "v"(v) : "memory");
}
// (s_nop: a store of more than 8 bytes reads its data registers over several cycles, and a VALU
// instruction that overwrites one of them within the next wait state changes what is stored — a hazard
// the compiler pads for its own stores and cannot see inside inline assembly.  Round 6 hit it in an
// experimental form of this kernel, profiles/r06/voxel_keys_in_lds_r06.txt; here the instruction behind
// the store has never reused a data register, and must not start to after some unrelated change.)
__device__ __forceinline__ void glb_store128(void *p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// E8 (include/rplgpu_msg.h): what happens to a point between polar->XY and the grid when a GROUP
// of scans shares one grid — the scan's motion during its acquisition (E6 de-skew, same
// operations in the same order as k_cloud) and its sensor's planar pose.  All zeros / identity
// reproduce x, y bit for bit.
struct ScanXf {
  float vx, vy, wz, dt;             // planar twist of the sensor and time between samples
  float t0;                         // time of the first sample relative to the fused instant
  uint32_t has_t0;                  // (0: no offsets set — tau = i * dt, bit for bit as without them)
  float r00, r01, tx, r10, r11, ty;  // [R | t] of the sensor in the common frame (2-D)
};
__device__ __forceinline__ f2 apply_xf(f2 xy, uint32_t sample_index, const ScanXf &m) {
  float tau = (float)sample_index * m.dt;
  if (m.has_t0) tau = m.t0 + tau;  // (scan-uniform)
  const float a = m.wz * tau, a2 = a * a;
  float ts = a2 * (1.0f / 120.0f);
  ts = ts + (-1.0f / 6.0f);
  ts = a2 * ts;
  ts = ts + 1.0f;
  const float sn = a * ts;
  float tc = a2 * (-1.0f / 720.0f);
  tc = tc + (1.0f / 24.0f);
  tc = a2 * tc;
  tc = tc + (-0.5f);
  tc = a2 * tc;
  const float cn = tc + 1.0f;
  const float x1 = (cn * xy.x - sn * xy.y) + m.vx * tau;
  const float y1 = (sn * xy.x + cn * xy.y) + m.vy * tau;
  f2 o;
  o.x = (m.r00 * x1 + m.r01 * y1) + m.tx;
  o.y = (m.r10 * x1 + m.r11 * y1) + m.ty;
  return o;
}

// What E5 needs of a sample when it runs inside the streaming pass (see voxel_stream, HASROR)
struct RorPre {
  f2 xy;    // E2 point in the sensor frame (in front of any E6 / E8 transform)
  bool e1;  // passed E1 (whatever the cell-range test says later)
};

// Per-sample arithmetic of phase S for one 8-byte node (lo, hi) and its table entry `c`: the sort
// key and the three quantities that are summed per cell.  Straight-line: a dropped sample is
// computed with dist 0 — x = y = 0, floor 0, remainder 0 — so its offsets are zero without any
// masking (a wave issues in order: exec-mask regions and branches around an `if (kept)` cost it
// more than the few instructions they skip on the ~10 % of dropped samples); only the count |
// intensity word is selected.  The KEY of a dropped sample is whatever the zero point yields and is
// never trusted (see voxel_block_pass).  With a transform (XF) or an unproven cell range (!SAFE)
// the zero point does not stay in cell (0, 0) / the sample may be dropped late: offsets are masked.
template <bool FAST_DIV, bool SAFE, bool HASQ, bool XF = false>
__device__ __forceinline__ bool voxel_sample(uint32_t lo, uint32_t hi, float2 c, const KParams &p,
                                             uint32_t q_min16, uint32_t ibfe_off,
                                             uint32_t ibfe_w, uint32_t &key, uint32_t &qx,
                                             uint32_t &qy, uint32_t &ci, uint32_t &flags,
                                             uint32_t sample_index = 0u,
                                             const ScanXf *xf = nullptr, RorPre *ror = nullptr) {
  const uint32_t d = __builtin_amdgcn_alignbit(hi, lo, 16);  // unaligned u32 at byte 2
  bool kept = (d - p.d_lo) <= p.d_span;                      // E1 (and :584)
  if (HASQ) kept = kept & ((hi & 0x00FF0000u) >= q_min16);
  const float df = __uint2float_rn(kept ? d : 0u);
  const float dm = FAST_DIV ? div_by(df, 4000.0f, 0.00025f) : df / 4000.0f;  // :590
  const f2 cv = {c.x, c.y};
  f2 xy = cv * dm;                                                             // E2
  if (ror) {  // (compile-time: the fused E5 instance) the point E5 sees: sensor frame, E1 only
    ror->xy = xy;
    ror->e1 = kept;
  }
  if (XF) xy = apply_xf(xy, sample_index, *xf);                                // E6 + pose (E8)
  f2 t;
  if (FAST_DIV) {
    t = div_by2(xy, p.voxel_leaf, p.inv_leaf);                                 // E4 cell
  } else {
    t.x = xy.x / p.voxel_leaf;
    t.y = xy.y / p.voxel_leaf;
  }
  const f2 f = {__builtin_floorf(t.x), __builtin_floorf(t.y)};
  if (!SAFE || XF) {
    const bool inr = (fabsf(f.x) < 32767.0f) && (fabsf(f.y) < 32767.0f);
    if (kept && !inr) flags |= RPLGPU_SCAN_CELL_RANGE;
    kept = kept & inr;
  }
  // iy + 32768 | ix + 32768 from the mantissas of f + (2^23 + 32768)
  const uint32_t kx = __float_as_uint(f.x + kKeyMagic);
  const uint32_t ky = __float_as_uint(f.y + kKeyMagic);
  key = __builtin_amdgcn_perm(ky, kx, 0x05040100u);
  const f2 lf = {p.voxel_leaf, p.voxel_leaf};
  const f2 r = __builtin_elementwise_fma(-f, lf, xy);  // x - ix*leaf, exact
  const f2 o = r * p.vox_scale_f;
  qx = (uint32_t)(int)o.x;  // (wrapping two's complement: see the fixed-point note in the header)
  qy = (uint32_t)(int)o.y;
  if (!SAFE || XF) {
    qx = kept ? qx : 0u;
    qy = kept ? qy : 0u;
  }
  ci = kept ? ((1u << 16) | ((hi >> ibfe_off) & ibfe_w)) : 0u;  // count | intensity (:591-592)
  return kept;
}


// Cross-lane part of one block of 64 * NS samples: lane l holds the NS consecutive samples
// NS l .. NS l + NS - 1 (ok[j]: the sample survived the keep mask).  A record is written where a
// run of equal keys ends; the records of a block go to consecutive queue entries in sample
// order, behind one marker entry.  The queue is the LDS array while it has room and the record
// store `G` (global) after that; either way entry i of the scan sits at position i.
// FILL: a sample the quality filter or the E5 mask drops must not END the run it sits in (it
// contributes nothing to it): inside the lane it takes its neighbour's key.  Without this a scan
// filtered at q_min = 48 makes twice the run records (random drops cut every run).  The plain
// instance does not need it: its drops come in runs (dist 0), a dropped sample never writes a
// record (its own run-end bit is masked) and it ends the run in front of it by its mask bit.
// kPassSplit: when the block makes more than kSplitAbove records (range noise: neighbouring samples
// alternate between two adjacent cells and every sample or two ends a run) nothing is written and
// the function returns true; the caller then aggregates the block in two CLASSES — the cells of
// either colour of a checkerboard, (ix + iy) & 1 — each blind to the other's samples (FILL), so
// that A B A B A becomes one run of A and one of B instead of five (voxel_block_split).
#ifndef RPL_VOXEL_SPLIT_ABOVE
#define RPL_VOXEL_SPLIT_ABOVE 26  // clean rings: <= ~21 runs per 128 samples at r = 30 m
#endif
constexpr uint32_t kSplitAbove = RPL_VOXEL_SPLIT_ABOVE;
#ifndef RPL_VOXEL_BRIDGE
#define RPL_VOXEL_BRIDGE 2  // FILL passes: wholly dropped lanes a run may cross (see voxel_block_pass)
#endif
constexpr int kBridge = RPL_VOXEL_BRIDGE;
// Only the kernel instance the launcher picks for batches known to be noisy compiles kPassSplit:
// inlined next to the plain path it costs a clean batch 1.5-2.7 % (profiles/r03/voxel_split_r03.txt).
enum : int { kPassPlain = 0, kPassSplit = 2 };
// Where a block's queue entries go: the per-scan queue (LDS while it has room, the workgroup's
// record store after that; the position comes from an LDS atomic).
struct QueueSink {
  VoxelLds &L;
  uint4 *G;
};
template <bool FILL, int NS, int MODE = kPassPlain>
__device__ __forceinline__ bool voxel_block_pass(QueueSink &S, const bool (&ok)[NS], uint32_t (&key)[NS],
                                                 const uint32_t (&qx)[NS], const uint32_t (&qy)[NS],
                                                 const uint32_t (&ci)[NS]) {
  uint64_t okm[NS];
  if (FILL) {
    bool any = ok[0];
    uint32_t kfirst = key[NS - 1];  // the lane's first kept key (selects, not an indexed array)
#pragma unroll
    for (int j = NS - 2; j >= 0; --j) kfirst = ok[j] ? key[j] : kfirst;
    key[0] = kfirst;
#pragma unroll
    for (int j = 1; j < NS; ++j) {
      any = any | ok[j];
      key[j] = ok[j] ? key[j] : key[j - 1];
    }
    const uint64_t anym = __builtin_amdgcn_ballot_w64(any);
#pragma unroll
    for (int j = 0; j < NS; ++j) okm[j] = anym;
  } else {
#pragma unroll
    for (int j = 0; j < NS; ++j) okm[j] = __builtin_amdgcn_ballot_w64(ok[j]);
  }
  // lane l+1's first key (lane 63: any value, its successor counts as dropped below)
  uint32_t next = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFEu, (int)key[0], 0x130,
                                                        0xF, 0xF, false);  // wave_shl:1
  // FILL: a lane BOTH of whose samples were dropped (the other class's, a low-quality pair, a masked
  // pair) is transparent as well — the run of the lane in front of it goes on in the lane behind it
  // when the keys agree — for gaps of up to kBridge lanes.  Dropped samples contribute zeros, so the
  // prefix sums do not see them.  (1 cm noise: 5800 -> 4900 -> 4650 records per scan with 1 / 2
  // lanes bridged; every record less is one less in every step of phase R.)
  uint64_t ok_after = okm[0] >> 1;  // the lane behind has a kept sample
  if (FILL && kBridge > 0) {
    uint32_t far = next;
    uint64_t reach = ok_after;      // some lane within the bridge has a kept sample
#pragma unroll
    for (int g = 1; g <= kBridge; ++g) {
      far = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFEu, (int)far, 0x130, 0xF, 0xF, false);
      const uint64_t here = okm[0] >> (g + 1);  // lane l + g + 1 has a kept sample
      // the nearest lane with a kept sample decides: lanes nearer than g + 1 have none iff !reach
      uint32_t pick;
      asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(pick) : "v"(far), "v"(next), "s"(reach));
      next = pick;
      reach |= here;
    }
    ok_after = reach;
  }
  // run-end masks built in scalar registers: (sample kept) & (key differs from the next one, or
  // the next one is dropped).  Written with explicit compares because `ballot(a && b)` is
  // materialised by the compiler as v_cndmask + v_cmp per mask; the per-lane predicates for the
  // stores come back from the masks with inverse_ballot (one s_and_saveexec each).
  // A run also ends where the NEXT sample is dropped, whatever key the zero point gave it: if
  // that key happened to match, no record would be written for the run and its sums would leak
  // into the next record (lane 63's successor counts as dropped: its last run always ends).
  uint64_t m[NS];
  uint32_t total = 0;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const uint32_t nk = j + 1 < NS ? key[j + 1 < NS ? j + 1 : 0] : next;
    asm("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(m[j]) : "v"(key[j]), "v"(nk));
    const uint64_t ok_next = j + 1 < NS ? okm[j + 1 < NS ? j + 1 : 0] : ok_after;
    m[j] = okm[j] & (m[j] | ~ok_next);
    total += (uint32_t)__popcll(m[j]);
  }
  if (total == 0u) return false;  // wave-uniform: nothing kept in this block
  if (MODE == kPassSplit && total > kSplitAbove) return true;  // wave-uniform: the caller splits the block
  // reserve queue entries (the records + their marker): one LDS atomic by lane 0, its round trip
  // overlaps the scans below (hand-placed so that the compiler's atomic optimiser does not wait
  // for it right away)
  uint32_t base = 0u;
  if (lane_id() == 0) {
    asm volatile("ds_add_rtn_u32 %0, %1, %2"
                 : "=v"(base)
                 : "v"((uint32_t)(uintptr_t)&S.L.misc[0]), "v"(total + 1u)
                 : "memory");
  }
  // lane totals -> inclusive block prefix of the lane's LAST sample; the others by subtraction
  uint32_t Px = qx[0], Py = qy[0], Pc = ci[0];
#pragma unroll
  for (int j = 1; j < NS; ++j) {
    Px += qx[j];
    Py += qy[j];
    Pc += ci[j];
  }
  wave_incl_scan3_dpp(Px, Py, Pc);
  // records written by the lanes before this one (all masks)
  uint32_t before = 0u;
#pragma unroll
  for (int j = 0; j < NS; ++j)
    before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[j] >> 32),
                                       __builtin_amdgcn_mbcnt_lo((uint32_t)m[j], before));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(base)::"memory");
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  uint32_t pos[NS];
  pos[0] = base + 1u + before;  // entries are queued in sample order behind the marker
#pragma unroll
  for (int j = 0; j + 1 < NS; ++j) {  // pos[j+1] = pos[j] + (a record ends at sample j): the mask is the carry-in
    uint64_t carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(pos[j + 1]), "=s"(carry_out) : "v"(pos[j]), "s"(m[j]));
  }
  u32x4 rec[NS];
  rec[NS - 1] = u32x4{key[NS - 1], Px, Py, Pc};
#pragma unroll
  for (int j = NS - 2; j >= 0; --j)
    rec[j] = u32x4{key[j], rec[j + 1].y - qx[j + 1], rec[j + 1].z - qy[j + 1], rec[j + 1].w - ci[j + 1]};
  const u32x4 marker = {kEmptyKey, 0u, 0u, 0u};
  // The stores are written as ds_write / global_store instructions by hand: left to the compiler,
  // the LDS and the record-store variants of a store are sunk into ONE flat store through a
  // selected pointer, and a flat instruction anywhere in the loop makes every wait for the
  // prefetched loads a vmcnt(0) (flat accesses return out of order).  The compiler does not see
  // these stores: the stream loop ends with an explicit wait for them before its barrier.
  VoxelLds &L = S.L;
  uint4 *const G = S.G;
  const uint32_t lds_rec = (uint32_t)(uintptr_t)&L.rec[0];
  if (base + 1u + total <= kRecCap) {  // wave-uniform: the usual case, everything goes to LDS
    if (lane_id() == 0) lds_store128(lds_rec + base * 16u, marker);
#pragma unroll
    for (int j = 0; j < NS; ++j)
      if (__builtin_amdgcn_inverse_ballot_w64(m[j])) lds_store128(lds_rec + pos[j] * 16u, rec[j]);
  } else {  // past (or across) the end of the LDS queue: the record store takes the rest
    if (lane_id() == 0) {
      if (base < kRecCap) lds_store128(lds_rec + base * 16u, marker); else glb_store128(G + base, marker);
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      if (__builtin_amdgcn_inverse_ballot_w64(m[j])) {
        if (pos[j] < kRecCap) lds_store128(lds_rec + pos[j] * 16u, rec[j]); else glb_store128(G + pos[j], rec[j]);
      }
    }
  }
  return false;
}

// A noisy block in two classes (see kPassSplit): class c keeps the samples whose cell has
// (ix + iy) & 1 == c, the others count as dropped (zero contribution, neighbour's key).
template <int NS>
__device__ __forceinline__ void voxel_block_split(QueueSink &S,
                                                  const bool (&ok)[NS], const uint32_t (&key)[NS],
                                                  const uint32_t (&qx)[NS], const uint32_t (&qy)[NS],
                                                  const uint32_t (&ci)[NS]) {
#pragma unroll
  for (uint32_t cls = 0; cls < 2u; ++cls) {
    bool okc[NS];
    uint32_t keyc[NS], qxc[NS], qyc[NS], cic[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      okc[j] = ok[j] && (((key[j] >> 16) ^ key[j]) & 1u) == cls;
      keyc[j] = key[j];
      qxc[j] = okc[j] ? qx[j] : 0u;
      qyc[j] = okc[j] ? qy[j] : 0u;
      cic[j] = okc[j] ? ci[j] : 0u;
    }
    voxel_block_pass<true, NS, kPassPlain>(S, okc, keyc, qxc, qyc, cic);
  }
}

// Run sums are differences of wrapping 32-bit block prefixes: non-negative and below 2^32 - 2^28
// except for the tiny negative offsets of samples whose quotient rounded up to the next integer.
__device__ __forceinline__ double run_sum_f64(uint32_t u) {
  double s = (double)u;
  if (u >= 0xF0000000u) s -= 4294967296.0;
  return s;
}

// Where a scan's cells go.  Legacy: a fixed region per scan (xyzi + b*out_stride).  Arena: all
// scans of a batch share one contiguous cloud; a workgroup reserves exactly the cells of its scan
// with one atomic on `cursor` once their number is known, so no separate packing pass (and no
// per-scan slack) is needed; the scans then sit in the arena in completion order and
// `scan_start[b]` says where.
struct VoxelArena {
  float4 *base;                  // null: legacy per-scan regions
  unsigned long long *cursor;    // next free point
  unsigned long long capacity;   // points the arena holds
  unsigned long long *scan_start;
  // 1: the arena holds 12-byte points (x, y, intensity) — the exchange payload of
  // include/rplgpu_comm.h written by the kernel itself (z is 0 for every point of this path).
  // `base` then only serves as the origin of point indices: point j is the floats 3 j .. 3 j + 2.
  int xyi;
};
// kEmitArenaTemp (round 3): a scan of several bands writes its cells to the workgroup's temporary
// cell area (behind its record store) band after band and copies them into the arena once their
// total is known — ONE pass over the bands.  (Rounds 1-2 went through the bands twice, counting
// then writing: +70 % on every multi-band scan.)  kEmitCountOnly / kEmitArenaKnown remain for the
// optional cell-key output, whose words must land at arena indices.
enum : int { kEmitLegacy = 0, kEmitArenaFirst = 1, kEmitCountOnly = 2, kEmitArenaKnown = 3, kEmitArenaTemp = 4 };

// Developer aid: phase cycle counters (tools/voxdbg.py); empty unless the kernel is a DBG build.
template <bool DBG>
struct PhaseClock {
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mark = 0;
  __device__ __forceinline__ void start() {
    if (DBG) mark = clock64();
  }
  __device__ __forceinline__ void lap(int i) {
    if (DBG) {
      const unsigned long long now = clock64();
      acc[i] += now - mark;
      mark = now;
    }
  }
};

// ------------------------------------------------------------------------------
// Phase R for one key band: the queue of run records in LDS -> output cells in (iy, ix) order.
// `normalised`: the records already hold run sums (they came from the record store).
// Returns 0 = done (ncell written to *ncell_out), 1 = the band must be bisected (too many rows;
// the record array is untouched in that case).
// ------------------------------------------------------------------------------
template <bool DBG>
__device__ __forceinline__ uint32_t voxel_reduce(VoxelLds &L, const KParams &p,
                                                 float4 *__restrict__ out, uint32_t out_stride,
                                                 uint32_t b, uint32_t *ncell_out, int mode,
                                                 const VoxelArena &arena, bool normalised,
                                                 const float4 *out_buffer) {
  PhaseClock<DBG> pc;
  pc.start();
  auto flush_dbg = [&]() {
    if (DBG && p.dbg && threadIdx.x == 0) {
#pragma unroll
      for (int i = 1; i < 8; ++i) atomicAdd(&p.dbg[16 * b + i], pc.acc[i]);
    }
  };
  uint32_t *const rowstart = L.rows, *const rowfill = L.rows + kRowCap;
  // optional cell-key output: the word of a cell sits at the cell's own index in the output buffer
  uint32_t *cell_keys = p.cell_keys ? p.cell_keys + (out - out_buffer) : nullptr;
  const int vbias = p.vox_bias;
  const uint32_t nrec = L.misc[0];
  if (DBG && p.dbg && threadIdx.x == 0) pc.acc[7] += (unsigned long long)nrec << 40;
  // thread t owns the queue records t, t + kVB, ... ; prefix -> run sum against the record just
  // before it unless it is the first record of its wave-pass
  uint4 mine[kRecPerThread];
  uint32_t rmin = 0xFFFFFFFFu, rmax = 0u;
#pragma unroll
  for (int k = 0; k < (int)kRecPerThread; ++k) {
    const uint32_t idx = threadIdx.x + (uint32_t)k * kVB;
    uint4 m = make_uint4(kEmptyKey, 0u, 0u, 0u);
    // whole 64-record slices beyond the queue tail are skipped (wave-uniform); the previous
    // record comes from the lane to the left
    if ((idx & ~63u) < nrec) {
      const uint4 raw = L.rec[idx];  // idx < kRecCap always
      // (a marker entry carries the empty key: it is the zero prefix in front of a block's
      // records and nothing else)
      const bool ok = idx < nrec && raw.x != kEmptyKey;
      m = raw;
      if (!normalised) {
        uint4 pr;  // the queue entry before this one: the lane to the left holds it
        pr.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw.y, 0x138, 0xF, 0xF, false);
        pr.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw.z, 0x138, 0xF, 0xF, false);
        pr.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw.w, 0x138, 0xF, 0xF, false);
        if (lane_id() == 0u && idx > 0u) {
          const uint4 t = L.rec[idx - 1u];
          pr.y = t.y; pr.z = t.z; pr.w = t.w;
        }
        m.y -= pr.y;  // (entry 0 of a queue is a marker: a record always has a predecessor)
        m.z -= pr.z;
        m.w -= pr.w;
      }
      if (!ok) m = make_uint4(kEmptyKey, 0u, 0u, 0u);
      if (ok) {
        rmin = min(rmin, m.x >> 16);
        rmax = max(rmax, m.x >> 16);
        atomicAdd(&rowstart[(m.x >> 16) & (kRowCap - 1u)], 1u);  // counting sort over rows
      }
    }
    mine[k] = m;
  }
  rmin = wave_min_lane63(rmin);  // DPP reductions: the totals land in lane 63
  rmax = wave_max_lane63(rmax);
  if (lane_id() == 63 && rmin != 0xFFFFFFFFu) {
    atomicMin(&L.misc[4], rmin);
    atomicMax(&L.misc[5], rmax);
  }
  __syncthreads();  // every record is in registers: the record array is free from here on
  pc.lap(1);
  rmin = L.misc[4];
  const uint32_t out_base = L.misc[7];
  uint32_t ncell = 0;
  if (rmin != 0xFFFFFFFFu) {  // at least one record (block-uniform)
    if (L.misc[5] - rmin + 1u > kRowCap) {
      flush_dbg();
      return 1u;
    }
    uint32_t *bucket = reinterpret_cast<uint32_t *>(L.rec);  // (ix << 16 | record index) by row
    uint32_t pad_total, nreal;
    {  // exclusive scan over the kRowCap rows in row order (2 per thread); a row iy lives at
       // iy mod kRowCap, so scan position j is the physical row (j + rmin) mod kRowCap.  Two
       // sums in one word: compact positions (low half) and positions with every row padded
       // to a multiple of four entries (high half: the rank step reads whole 16-byte blocks
       // and needs no position masks)
      const uint32_t p0 = (kRowsPerThread * threadIdx.x + rmin) & (kRowCap - 1u);
      const uint32_t p1 = (kRowsPerThread * threadIdx.x + 1u + rmin) & (kRowCap - 1u);
      const uint32_t r0 = rowstart[p0], r1 = kRowsPerThread == 2u ? rowstart[p1] : 0u;
      const uint32_t w0 = r0 | (((r0 + 3u) & ~3u) << 16), w1 = r1 | (((r1 + 3u) & ~3u) << 16);
      uint32_t tot;
      const uint32_t ex = vx_excl_scan(w0 + w1, L.tmp, &tot);
      rowstart[p0] = ex;
      rowfill[p0] = ex >> 16;
      if (kRowsPerThread == 2u) {
        rowstart[p1] = ex + w0;
        rowfill[p1] = (ex + w0) >> 16;
      }
      pad_total = tot >> 16;
      nreal = tot & 0xFFFFu;  // records of the band (the queue entries minus the markers)
    }
    // padding entries compare as "not smaller than anything"
    for (uint32_t i = threadIdx.x * 4u; i < pad_total; i += kVB * 4u)
      *reinterpret_cast<uint4 *>(&bucket[i]) =
          make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    __syncthreads();
    pc.lap(3);
    // (behind the row lists: the physical row of every list slot and, later, the rank of every
    // record — both 16-bit, both inside the record array, which is free while the records sit
    // in registers)
    uint16_t *rowid = reinterpret_cast<uint16_t *>(bucket + kBucketCap);
    uint16_t *rankof = rowid + kBucketCap;
    static_assert(kBucketCap * 6u + kRecCap * 2u <= kRecCap * 16u, "scratch fits the record array");
#pragma unroll
    for (int k = 0; k < (int)kRecPerThread; ++k) {
      if (mine[k].x != kEmptyKey) {
        const uint32_t idx = threadIdx.x + (uint32_t)k * kVB;
        const uint32_t row = (mine[k].x >> 16) & (kRowCap - 1u);
        const uint32_t pos = atomicAdd(&rowfill[row], 1u);
        bucket[pos] = (mine[k].x << 16) | idx;  // (ix, record index): unique
        rowid[pos] = (uint16_t)row;
      }
    }
    __syncthreads();
    pc.lap(4);
    // rank inside the row = number of smaller entries of the same row.  The work is handed out
    // by LIST SLOT, not by record: thread t ranks the entries at slots t, t + kVB, ... — the
    // entries of a long row (a ring running along x, with range noise hundreds of records) are
    // then spread over as many threads, which read the same blocks at the same time (LDS
    // broadcast), instead of one thread walking its 7 rows alone whatever their length
    // (noisy scans: 43 k -> cycles of the v1 kernel's rank step, p99 150 k).
    for (uint32_t q = threadIdx.x; q < pad_total; q += kVB) {
      const uint32_t me = bucket[q];
      if (me != 0xFFFFFFFFu) {
        const uint32_t row = rowid[q];
        const uint32_t st = rowstart[row], s1 = rowfill[row];
        const uint32_t s0 = st >> 16;  // padded start, a multiple of 4; s1 > s0
        uint32_t rank = st & 0xFFFFu;
        // the first two blocks are read together (one LDS round trip covers rows of <= 8)
        const uint4 v0 = *reinterpret_cast<const uint4 *>(&bucket[s0]);
        const uint4 v1 = *reinterpret_cast<const uint4 *>(&bucket[min(s0 + 4u, kBucketCap - 4u)]);
        rank += (v0.x < me) + (v0.y < me) + (v0.z < me) + (v0.w < me);
        if (s0 + 4u < s1) {
          rank += (v1.x < me) + (v1.y < me) + (v1.z < me) + (v1.w < me);
          for (uint32_t m = s0 + 8u; m < s1; m += 16u) {  // long rows: four blocks per trip
            uint4 w[4];  // (blocks past the row's end belong to the next rows or are padding:
                         //  they are read — inside the array — but not counted)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              w[j] = *reinterpret_cast<const uint4 *>(&bucket[min(m + 4u * j, kBucketCap - 4u)]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (m + 4u * j < s1) rank += (w[j].x < me) + (w[j].y < me) + (w[j].z < me) + (w[j].w < me);
          }
        }
        rankof[me & 0xFFFFu] = (uint16_t)rank;
      }
    }
    __syncthreads();
    uint32_t rk[kRecPerThread];
#pragma unroll
    for (int k = 0; k < (int)kRecPerThread; ++k)
      rk[k] = rankof[threadIdx.x + (uint32_t)k * kVB];  // (garbage for empty slots: not used)
    __syncthreads();  // the sorted records overwrite the lists
#pragma unroll
    for (int k = 0; k < (int)kRecPerThread; ++k)
      if (mine[k].x != kEmptyKey) L.rec[rk[k]] = mine[k];
    __syncthreads();
    pc.lap(5);
    // heads of equal-key groups -> cell index; thread t owns sorted records [8t, 8t+8)
    const uint32_t r_lo = threadIdx.x * kRecPerThread;
    uint32_t headbits = 0, nheads = 0;
    uint32_t prevkey = (r_lo > 0 && r_lo <= nreal) ? L.rec[r_lo - 1].x : kEmptyKey;
#pragma unroll
    for (int k = 0; k < (int)kRecPerThread; ++k) {
      const uint32_t r = r_lo + k;
      const uint32_t key = (r < nreal) ? L.rec[r].x : kEmptyKey;
      if (r < nreal && key != prevkey) {
        headbits |= 1u << k;
        ++nheads;
      }
      prevkey = key;
    }
    uint32_t cell = vx_excl_scan(nheads, L.tmp, &ncell);
    pc.lap(6);
    // position of every cell's first record (16-bit entries; the row arrays are free again)
    uint16_t *heads = reinterpret_cast<uint16_t *>(L.rows);
#pragma unroll
    for (int k = 0; k < (int)kRecPerThread; ++k)
      if ((headbits >> k) & 1u) heads[cell++] = (uint16_t)(r_lo + k);
    __syncthreads();
    // one cell per thread, coalesced 16-byte output rows.  All sums are exact in fp64
    // (integers below 2^53); the quotient by the count uses the host-built correctly
    // rounded reciprocal + one FMA remainder + one FMA correction (Markstein), which is the
    // correctly rounded quotient, i.e. the spec's (fp64 sum) / count.
    const double inv_scale = 1.0 / p.vox_scale;  // exact power of two
    const double dL = (double)p.vox_L, dbias = (double)vbias;
    uint32_t nemit = min(ncell, out_stride > out_base ? out_stride - out_base : 0u);
    if (mode == kEmitArenaFirst) {  // the only band of this scan: reserve its cells now
      if (threadIdx.x == 0) {
        const unsigned long long at = atomicAdd(arena.cursor, (unsigned long long)ncell);
        L.tmp[28] = (uint32_t)at;
        L.tmp[29] = (uint32_t)(at >> 32);
      }
      __syncthreads();
      const unsigned long long at = ((unsigned long long)L.tmp[29] << 32) | L.tmp[28];
      out = arena.base + at;
      if (cell_keys) cell_keys = p.cell_keys + at;
      nemit = at >= arena.capacity ? 0u : (uint32_t)min((unsigned long long)ncell, arena.capacity - at);
    } else if (mode == kEmitCountOnly) {
      nemit = 0u;  // several bands: first learn the total, the cells are written in a second go
    }
    for (uint32_t c = threadIdx.x; c < nemit; c += kVB) {
      uint32_t r = heads[c];
      // a cell is 1.13 records on average: fetch the head and the two records behind it in one
      // round trip, continue serially only when all three belong to the cell
      const uint4 q0 = L.rec[r];
      const uint4 q1 = L.rec[min(r + 1u, kRecCap - 1u)];
      const uint4 q2 = L.rec[min(r + 2u, kRecCap - 1u)];
      const uint32_t key = q0.x;
      const bool m1 = (r + 1u < nreal) && (q1.x == key);
      const bool m2 = m1 && (r + 2u < nreal) && (q2.x == key);
      double sx = run_sum_f64(q0.y), sy = run_sum_f64(q0.z);
      // count and intensity sum are packed per RECORD (<= 128 samples: 16 bits each suffice);
      // per CELL they are summed apart -- a large or close cell collects thousands of samples
      // and its intensity sum passes 2^16 (found by tests/test_gpu_fuzz.py)
      uint32_t cnt = q0.w >> 16, isum = q0.w & 0xFFFFu;
      if (m1) { sx += run_sum_f64(q1.y); sy += run_sum_f64(q1.z); cnt += q1.w >> 16; isum += q1.w & 0xFFFFu; }
      if (m2) {
        sx += run_sum_f64(q2.y); sy += run_sum_f64(q2.z); cnt += q2.w >> 16; isum += q2.w & 0xFFFFu;
        for (r += 3u; r < nreal; ++r) {  // segmented sum over the rest of this cell's records
          const uint4 q = L.rec[r];
          if (q.x != key) break;
          sx += run_sum_f64(q.y);
          sy += run_sum_f64(q.z);
          cnt += q.w >> 16;
          isum += q.w & 0xFFFFu;
        }
      }
      const double ix = (double)((int)(key & 0xFFFFu) - 32768);
      const double iy = (double)((int)(key >> 16) - 32768);
      // RN(1/count): from the LDS copy of the host-built table for small counts, else the IEEE
      // fp64 divide (correctly rounded, i.e. the same value; ~25 instructions); a dependent
      // gather from the table in global memory (L2 latency) per cell costs more than either
      const double dc = (double)cnt;
      const double rc = cnt < 256u ? L.rcp[cnt] : 1.0 / dc;
      const double Sx = fma(dc, ix * dL - dbias, sx);  // coordinate sums in units of 2^-K m
      const double Sy = fma(dc, iy * dL - dbias, sy);
      const double si = (double)isum;
      double qx = Sx * rc, qy = Sy * rc, qi = si * rc;
      qx = fma(fma(-qx, dc, Sx), rc, qx);
      qy = fma(fma(-qy, dc, Sy), rc, qy);
      qi = fma(fma(-qi, dc, si), rc, qi);
      if (arena.xyi && (mode == kEmitArenaFirst || mode == kEmitArenaKnown)) {
        float *f = reinterpret_cast<float *>(arena.base) + 3u * (size_t)((out - arena.base) + out_base + c);
        f[0] = (float)(qx * inv_scale);
        f[1] = (float)(qy * inv_scale);
        f[2] = (float)qi;
      } else {
        out[out_base + c] = make_float4((float)(qx * inv_scale), (float)(qy * inv_scale), 0.0f,
                                        (float)qi);
      }
      if (cell_keys) cell_keys[out_base + c] = key;  // (block-uniform pointer, usually null)
    }
  }
  __syncthreads();
  pc.lap(7);
  flush_dbg();
  *ncell_out = ncell;
  return 0u;
}

// ------------------------------------------------------------------------------
// Phase S over the blocks blk0, blk0 + kVW, ... < blk_end of ONE scan (`scan_rsrc`: a bounds-checked
// buffer resource over the bytes this caller may read; beyond it a load returns zeros = dropped
// samples and costs no memory traffic): the waves of the workgroup interleave the blocks of their scan.
// block k = the samples [128 k, 128 k + 128), i.e. one contiguous KiB; lane l owns the sample pair
// 128 k + 2 l, + 1 (one buffer_load_dwordx4).  The raw pairs run kAhead blocks ahead in a ring of
// four register buffers, unrolled four times so that the ring costs no register moves; the table
// entries stay one block ahead (they come from L2 / L1).
// (Also tried in round 3: four samples per lane.  Read directly — 32 bytes per lane, table gathers
// at a stride of 8 entries — the texture addresser became the bottleneck; transposed through LDS
// each block waits for two LDS round trips on top of the loads: profiles/r03/voxel_phaseS_r03.txt.)
// ------------------------------------------------------------------------------
// HASROR (round 6, the ROR instance of the kernel): E5 runs INSIDE the pass.  A block then OWNS 124
// samples and reads 128: lanes 0 and 63 hold the pair in front of and behind the block's own samples
// (block k = the bytes [992 k - 16, 992 k + 1008) of the scan), so that every owned sample has its
// index neighbours at +-1, +-2 in the two lanes next to it.  A lane tests its two samples against
// each other and against the pair of the lane behind it (four distance tests, the oracle's
// expression: products then sum); what the lane in front found comes over as a shifted scalar mask.
// A kept sample with `ror_k` neighbours among those four is settled (it survives E5); one without is
// dropped from the aggregation here and listed in `rs` for the exact steps behind the pass
// (ror_resolve).  Dropped and unsettled samples are transparent to the runs (FILL).
template <bool FAST_DIV, bool SAFE, bool SPLIT, bool DBG, bool HASQ, bool HASMASK, bool XF, int AHEAD,
          bool HASROR = false>
__device__ __forceinline__ void voxel_stream(QueueSink &sink, const KParams &p,
                                             const float2 *__restrict__ cs,
                                             const __amdgpu_buffer_rsrc_t scan_rsrc, uint32_t blk0,
                                             uint32_t blk_end, const uint32_t *__restrict__ ror_bits,
                                             uint32_t mask_stride, const ScanXf &xf, uint32_t q_min16,
                                             uint32_t ibfe_off, uint32_t ibfe_w, uint32_t &flags,
                                             unsigned long long *dbg_slot, RorSide *rs = nullptr,
                                             uint32_t oob_off = 0u) {
  static_assert(!HASROR || !HASMASK, "the ROR instance: no mask word");
  constexpr uint32_t kBlkBytes = HASROR ? 992u : 1024u, kBlkSamples = HASROR ? 124u : 128u;
  // raw pairs run AHEAD blocks in front of the block being aggregated, the table entries GA blocks (their
  // addresses come out of the raw pair: GA < AHEAD); both live in register rings of N slots, the loop is
  // unrolled N times so that the rings cost no moves
  constexpr int GA = RPL_VOXEL_GAHEAD < AHEAD ? RPL_VOXEL_GAHEAD : AHEAD - 1, N = AHEAD <= 3 ? 4 : AHEAD + 1;
  static_assert(AHEAD >= 2 && AHEAD <= 7 && GA >= 1 && GA < AHEAD, "ring depths");
  if (blk0 >= blk_end) return;  // wave-uniform (no barrier inside the stream)
  auto load_pair = [&](uint32_t byte_off) -> uint4 {
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(scan_rsrc, (int)byte_off, 0, RPL_RAW_AUX);
    return make_uint4(t.x, t.y, t.z, t.w);
  };
  // (HASROR: lane 0 of block 0 would read the 16 bytes in front of the scan: it reads past the
  // resource instead — `oob_off`, a multiple of 16 >= the scan's bytes — and gets zeros)
  const uint32_t lane_off = HASROR ? lane_id() * 16u - 16u : lane_id() * 16u;
  const bool first_lane = HASROR && lane_id() == 0u;
  auto blk_off = [&](uint32_t blk) -> uint32_t {
    const uint32_t o = blk * kBlkBytes + lane_off;
    return (HASROR && blk == 0u) ? (first_lane ? oob_off : o) : o;
  };
  const bool owned = !HASROR || (lane_id() - 1u) < 62u;
  uint4 w[N];
  float2 cA[N], cB[N];
#pragma unroll
  for (int j = 0; j < AHEAD; ++j) w[j] = load_pair(blk_off(blk0 + (uint32_t)(j * kVW)));
  auto gat = [&](uint32_t word) -> float2 { return cs[word & 0xFFFFu]; };
#pragma unroll
  for (int j = 0; j < GA; ++j) {
    cA[j] = gat(w[j].x);
    cB[j] = gat(w[j].z);
  }
  unsigned long long sub[5] = {0, 0, 0, 0, 0}, tprev = DBG ? clock64() : 0ull;
  for (uint32_t blk4 = blk0; blk4 < blk_end; blk4 += (uint32_t)(N * kVW)) {
    if (HASROR) {  // a cluttered scan (more unsettled samples than the list holds) is given up at once
      if ((uint32_t)__builtin_amdgcn_readfirstlane((int)*(volatile uint32_t *)&rs->n_todo) > kRorTodoCap) break;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const uint32_t blk = blk4 + (uint32_t)k * kVW;
      // (the loads are issued whether or not the block exists — beyond the resource they return
      // zeros — so that every path through the loop has the same loads in flight and the
      // compiler's waits stay exact)
      unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
      if (DBG) t0 = clock64();
      cA[(k + GA) % N] = gat(w[(k + GA) % N].x);
      cB[(k + GA) % N] = gat(w[(k + GA) % N].z);
      w[(k + AHEAD) % N] = load_pair(blk_off(blk + (uint32_t)AHEAD * kVW));
      if (DBG) {  // [8] issue of the loads (incl. the wait for the raw pair the gathers need)
        asm volatile("" : "+v"(cA[(k + GA) % N].x), "+v"(cB[(k + GA) % N].x)::"memory");
        t1 = clock64();
      }
      if (blk < blk_end) {  // wave-uniform
        const uint4 w0 = w[k];
        const float2 c0[2] = {cA[k], cB[k]};
        uint32_t lo[2] = {w0.x, w0.z}, hi[2] = {w0.y, w0.w};
        if (HASMASK) {  // E5 mask (one bit per sample): a dropped sample gets dist 0
          const uint32_t word = blk * 4u + (lane_id() >> 4);  // sample 128 blk + 2 l -> bit 2 (l & 15)
          const uint32_t bits = (word < mask_stride) ? ror_bits[word] : 0u;
          const uint32_t two = bits >> ((lane_id() & 15u) * 2u);
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (!((two >> j) & 1u)) { lo[j] &= 0x0000FFFFu; hi[j] &= 0xFFFF0000u; }
        }
        // sample index inside the scan (HASROR: 124 blk - 2 + 2 lane; the halo lanes' value is never used)
        const uint32_t i0 = HASROR ? blk * kBlkSamples + lane_id() * 2u - 2u : blk * 128u + lane_id() * 2u;
        bool ok[2];
        uint32_t key[2], qx[2], qy[2], ci[2];
        RorPre pre[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
          ok[j] = voxel_sample<FAST_DIV, SAFE, HASQ, XF>(lo[j], hi[j], c0[j], p, q_min16, ibfe_off,
                                                        ibfe_w, key[j], qx[j], qy[j], ci[j], flags,
                                                        i0 + (uint32_t)j, &xf, HASROR ? &pre[j] : nullptr);
        if (HASROR) {
          // (a sample that did not pass E1 was computed with dist 0 and sits at the origin: a pair only
          // counts when BOTH its samples passed E1 — scalar masks, no selects on the coordinates)
          const float ax = pre[0].xy.x, ay = pre[0].xy.y, bx = pre[1].xy.x, by = pre[1].xy.y;
          auto behind = [&](float v) -> float {  // lane l + 1's value (lane 63: its own, masked below)
            return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(
                (int)__float_as_uint(v), (int)__float_as_uint(v), 0x130, 0xF, 0xF, false));  // wave_shl:1
          };
          const float nax = behind(ax), nay = behind(ay), nbx = behind(bx), nby = behind(by);
          const float r2 = p.ror_r2;
          auto within = [&](float x0, float y0, float x1, float y1) -> uint64_t {
            const float dx = x0 - x1, dy = y0 - y1;
            const float d2 = dx * dx + dy * dy;  // products then sum (-ffp-contract=off): the oracle's test
            return __builtin_amdgcn_ballot_w64(d2 <= r2);
          };
          const uint64_t KA = __builtin_amdgcn_ballot_w64(pre[0].e1), KB = __builtin_amdgcn_ballot_w64(pre[1].e1);
          const uint64_t t0 = within(ax, ay, bx, by) & KA & KB;            // (2l, 2l+1)
          const uint64_t t1 = within(bx, by, nax, nay) & KB & (KA >> 1);   // (2l+1, 2l+2)
          const uint64_t t2 = within(ax, ay, nax, nay) & KA & (KA >> 1);   // (2l, 2l+2)
          const uint64_t t3 = within(bx, by, nbx, nby) & KB & (KB >> 1);   // (2l+1, 2l+3)
          auto addc = [](uint32_t v, uint64_t m) -> uint32_t {
            uint32_t o;
            uint64_t co;
            asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(o), "=s"(co) : "v"(v), "s"(m));
            return o;
          };
          // sample a: 2l+1 (t0), 2l+2 (t2), 2l-1 (lane l-1's t1), 2l-2 (lane l-1's t2)
          // sample b: 2l (t0), 2l+2 (t1), 2l+3 (t3), 2l-1 (lane l-1's t3)
          const uint32_t ca = addc(addc(addc(addc(0u, t0), t2), t1 << 1), t2 << 1);
          const uint32_t cb = addc(addc(addc(addc(0u, t0), t1), t3), t3 << 1);
          bool sa = ca >= p.ror_k, sb = cb >= p.ror_k;
          bool ua = owned && pre[0].e1 && !sa, ub = owned && pre[1].e1 && !sb;
          if (__builtin_amdgcn_ballot_w64(ua || ub) != 0ull) {  // wave-uniform, rare (one block in twenty)
            // The wave holds 128 consecutive samples of the scan: an unsettled one is first run past ALL of
            // them (its point from a scalar lane read, two tests per lane, two ballots) — any kept samples
            // within r will do, and what a sample misses among its four index neighbours is nearly always
            // a few indices further on.  No memory access; only what this leaves goes on the list.
            uint64_t late[2] = {0ull, 0ull};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              uint64_t m = __builtin_amdgcn_ballot_w64(j ? ub : ua);
              while (m) {
                const uint32_t l = (uint32_t)__builtin_ctzll(m);
                m &= m - 1ull;
                const float mx = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(j ? bx : ax), (int)l));
                const float my = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(j ? by : ay), (int)l));
                uint64_t ha = within(mx, my, ax, ay) & KA, hb = within(mx, my, bx, by) & KB;
                if (j) hb &= ~(1ull << l); else ha &= ~(1ull << l);  // (not the sample itself)
                if ((uint32_t)__popcll(ha) + (uint32_t)__popcll(hb) >= p.ror_k) late[j] |= 1ull << l;
              }
            }
            const bool la = __builtin_amdgcn_inverse_ballot_w64(late[0]), lb = __builtin_amdgcn_inverse_ballot_w64(late[1]);
            sa = sa || la;
            sb = sb || lb;
            ua = ua && !la;
            ub = ub && !lb;
            if (ua) {
              const uint32_t sl = atomicAdd(&rs->n_todo, 1u);
              if (sl < kRorTodoCap) rs->todo[sl] = (uint16_t)i0;
            }
            if (ub) {
              const uint32_t sl = atomicAdd(&rs->n_todo, 1u);
              if (sl < kRorTodoCap) rs->todo[sl] = (uint16_t)(i0 + 1u);
            }
          }
          ok[0] = ok[0] && owned && sa;
          ok[1] = ok[1] && owned && sb;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            qx[j] = ok[j] ? qx[j] : 0u;
            qy[j] = ok[j] ? qy[j] : 0u;
            ci[j] = ok[j] ? ci[j] : 0u;
          }
        }
        if (DBG) {  // [9] wait for this block's table entries + sample arithmetic
          asm volatile("" : "+v"(qx[1]), "+v"(qy[0]), "+v"(ci[1]), "+v"(key[1])::"memory");
          t2 = clock64();
        }
        uint32_t key0[2] = {key[0], key[1]};  // (the FILL instance rewrites dropped samples' keys)
        // (the ROR instance without quality filter runs the PLAIN pass: its drops are E1's runs of dist 0 plus
        // a rare unsettled sample, which ends a run like any dropped sample)
        if (voxel_block_pass<HASQ || HASMASK, 2, SPLIT ? kPassSplit : kPassPlain>(sink, ok, key, qx, qy, ci))
          voxel_block_split<2>(sink, ok, key0, qx, qy, ci);
        if (DBG) {  // [10] block pass (issue only: its stores are not waited for)
          t3 = clock64();
          if (dbg_slot && threadIdx.x == 0) {
            sub[0] += t1 - t0; sub[1] += t2 - t1; sub[2] += t3 - t2; sub[3] += t0 - tprev; sub[4] += 1;
          }
          tprev = t3;
        }
      }
    }
  }
  if (DBG && dbg_slot && threadIdx.x == 0)
    for (int i = 0; i < 5; ++i) atomicAdd(&dbg_slot[8 + i], sub[i]);
}

// ------------------------------------------------------------------------------
// E5 inside the voxel kernel, the exact steps behind the streaming pass (block-uniform control flow,
// called by every thread of the workgroup behind a barrier).  The pass settled every kept sample that
// has `ror_k` neighbours among its four nearest indices; the list holds the others (islands between
// drop-outs, isolated returns: ~12 of 32 000 samples on a ring with 10 % drop-outs).
//   1b. one wave per listed sample tests the 64 samples before and the 64 after it (k_ror_mask's
//       stage 1b, rpl_ror.hip);
//   2.  what is still unsettled (at most kRorFewCap samples) is counted exhaustively: every thread
//       runs the scan's samples past those few points.
// A sample that turns out to have its neighbours is appended to the queue as a record of its own
// behind its own marker (the sums are order independent).  Returns true when the scan has more
// unsettled samples than these steps take (clutter: the work item is left to the two-kernel path).
// ------------------------------------------------------------------------------
template <bool FAST_DIV, bool SAFE, bool XF>
__device__ __forceinline__ void ror_append(QueueSink &S, uint2 nd, float2 c, uint32_t i, const KParams &p,
                                           uint32_t q_min16, uint32_t ibfe_off, uint32_t ibfe_w,
                                           const ScanXf &xf, uint32_t &flags) {
  uint32_t key, qx, qy, ci;
  const bool k = voxel_sample<FAST_DIV, SAFE, true, XF>(nd.x, nd.y, c, p, q_min16, ibfe_off,
                                                        ibfe_w, key, qx, qy, ci, flags, i, &xf);
  if (!k) return;  // (dropped late: outside the cell range — flagged)
  const uint32_t base = atomicAdd(&S.L.misc[0], 2u);
  const uint4 marker = make_uint4(kEmptyKey, 0u, 0u, 0u), rec = make_uint4(key, qx, qy, ci);
  if (base < kRecCap) S.L.rec[base] = marker; else S.G[base] = marker;
  if (base + 1u < kRecCap) S.L.rec[base + 1u] = rec; else S.G[base + 1u] = rec;
}

// E2 as voxel_sample computes it from a node and its (cos, sin) entry; 1e30 m away unless E1 keeps the node
template <bool FAST_DIV>
__device__ __forceinline__ float2 ror_point(uint2 nd, float2 c, const KParams &p, uint32_t q_min16) {
  const uint32_t d = nd_dist(nd);
  const float df = __uint2float_rn(d);
  const float dm = FAST_DIV ? div_by(df, 4000.0f, 0.00025f) : df / 4000.0f;
  const bool k = ((d - p.d_lo) <= p.d_span) && ((nd.y & 0x00FF0000u) >= q_min16);
  return make_float2(k ? dm * c.x : 1.0e30f, k ? dm * c.y : 1.0e30f);
}

// Step 2 of ror_resolve: the scan's samples past the <= kRorFewCap points still open, ONE point per pass
// over the scan (a counter and a point in registers: with all eight points tested in one unrolled pass the
// kernel spilt 58 vector and 367 scalar registers and lost 13 % on a batch that never gets here).
#ifndef RPL_ROR_KU
#define RPL_ROR_KU 4
#endif
template <bool FAST_DIV, bool SAFE, bool XF>
__device__ __forceinline__ void ror_exhaustive(QueueSink &S, RorSide &R, const KParams &p,
                                               const float2 *__restrict__ cs,
                                               const uint2 *__restrict__ scan, uint32_t n,
                                               const ScanXf &xf, uint32_t q_min16, uint32_t ibfe_off,
                                               uint32_t ibfe_w, uint32_t n2, uint32_t &flags) {
  uint32_t tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // (nothing of this step is hoisted out of the kernel's item loop)
  const float r2 = p.ror_r2;
  const uint2 none = make_uint2(0u, 0u);  // (dist 0: never kept)
  if (tid < kRorFewCap) R.few_cnt[tid] = 0u;
  __syncthreads();
  constexpr int KU = RPL_ROR_KU;
  for (uint32_t u = 0; u < n2; ++u) {  // block-uniform
    const uint32_t iu = (uint32_t)__builtin_amdgcn_readfirstlane((int)R.todo2[u]);
    const uint2 nu = scan[iu];
    const float2 me = ror_point<FAST_DIV>(nu, cs[nd_q14(nu)], p, q_min16);
    uint32_t hits = 0u;
    // KU nodes per thread and trip, their loads and their gathers issued together (the scan is in L2 / the
    // Infinity Cache: the pass just streamed it)
    for (uint32_t q0 = tid; q0 < n; q0 += (uint32_t)(KU * kVB)) {
      uint2 nd[KU];
      float2 c[KU];
#pragma unroll
      for (int j = 0; j < KU; ++j) {
        const uint32_t q = q0 + (uint32_t)(j * kVB);
        nd[j] = q < n ? scan[q] : none;
      }
#pragma unroll
      for (int j = 0; j < KU; ++j) c[j] = cs[nd_q14(nd[j])];
#pragma unroll
      for (int j = 0; j < KU; ++j) {
        const uint32_t q = q0 + (uint32_t)(j * kVB);
        const float2 pc = ror_point<FAST_DIV>(nd[j], c[j], p, q_min16);
        const float dx = me.x - pc.x, dy = me.y - pc.y;
        const float d2 = dx * dx + dy * dy;
        hits += (q != iu && d2 <= r2) ? 1u : 0u;
      }
    }
    const uint32_t tot = wave_incl_scan_fast(hits);  // (the wave's sum in lane 63)
    if ((tid & 63u) == 63u && tot) atomicAdd(&R.few_cnt[u], tot);
  }
  __syncthreads();
  if (tid < n2 && R.few_cnt[tid] >= p.ror_k) {
    const uint2 nd = scan[R.todo2[tid]];
    ror_append<FAST_DIV, SAFE, XF>(S, nd, cs[nd_q14(nd)], R.todo2[tid], p, q_min16, ibfe_off, ibfe_w, xf, flags);
  }
}

template <bool FAST_DIV, bool SAFE, bool XF>
__device__ __forceinline__ bool ror_resolve(QueueSink &S, RorSide &R, const KParams &p,
                                            const float2 *__restrict__ cs,
                                            const uint2 *__restrict__ scan, uint32_t n,
                                            const ScanXf &xf, uint32_t q_min16, uint32_t ibfe_off,
                                            uint32_t ibfe_w, uint32_t &flags) {
  const uint32_t n_todo = R.n_todo;  // (the caller's barrier is behind the pass)
  if (n_todo == 0u) return false;
  if (n_todo > kRorTodoCap) return true;
  const float r2 = p.ror_r2;
  const uint32_t need = p.ror_k;
  const uint2 none = make_uint2(0u, 0u);  // (dist 0: never kept)
  const uint32_t lane = lane_id();
  for (uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_id()); t < n_todo; t += (uint32_t)kVW) {
    const uint32_t i = (uint32_t)__builtin_amdgcn_readfirstlane((int)R.todo[t]);
    // (the three node loads, then the three gathers: two round trips per listed sample)
    const uint32_t qb = i - 1u - lane, qa = i + 1u + lane;  // (qb wraps below 0: fails q < n)
    const uint2 nm = scan[i], nb = qb < n ? scan[qb] : none, na = qa < n ? scan[qa] : none;
    const float2 cm = cs[nd_q14(nm)], cb = cs[nd_q14(nb)], ca = cs[nd_q14(na)];
    const float2 me = ror_point<FAST_DIV>(nm, cm, p, q_min16), pb = ror_point<FAST_DIV>(nb, cb, p, q_min16),
                 pa = ror_point<FAST_DIV>(na, ca, p, q_min16);
    const float dxb = me.x - pb.x, dyb = me.y - pb.y, dxa = me.x - pa.x, dya = me.y - pa.y;
    const float d2b = dxb * dxb + dyb * dyb, d2a = dxa * dxa + dya * dya;
    const uint32_t cnt = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(d2b <= r2)) +
                         (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(d2a <= r2));
    if (lane == 0u) {
      if (cnt >= need) {
        ror_append<FAST_DIV, SAFE, XF>(S, nm, cm, i, p, q_min16, ibfe_off, ibfe_w, xf, flags);
      } else {
        const uint32_t sl = atomicAdd(&R.n_todo2, 1u);
        if (sl < kRorFewCap) R.todo2[sl] = (uint16_t)i;
      }
    }
  }
  __syncthreads();
  const uint32_t n2 = R.n_todo2;
  if (n2 == 0u) return false;
  if (n2 > kRorFewCap) return true;
  ror_exhaustive<FAST_DIV, SAFE, XF>(S, R, p, cs, scan, n, xf, q_min16, ibfe_off, ibfe_w, n2, flags);
  return false;
}

// What of one scan the streaming code needs besides its nodes: the E5 keep bits and the E8 transform.
struct ScanSide {
  const uint32_t *ror_bits;
  ScanXf xf;
};
__device__ __forceinline__ ScanSide scan_side(uint32_t sc, const uint32_t *__restrict__ keepmask,
                                              uint32_t mask_stride, const float *__restrict__ motion,
                                              const float *__restrict__ pose2d,
                                              const float *__restrict__ scan_t0) {
  ScanSide s;
  s.ror_bits = keepmask ? keepmask + (size_t)sc * mask_stride : nullptr;
  s.xf = ScanXf{0.f, 0.f, 0.f, 0.f, 0.f, 0u, 1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
  if (motion) {
    s.xf.vx = motion[4 * sc]; s.xf.vy = motion[4 * sc + 1]; s.xf.wz = motion[4 * sc + 2]; s.xf.dt = motion[4 * sc + 3];
    if (scan_t0) { s.xf.t0 = scan_t0[sc]; s.xf.has_t0 = 1u; }
  }
  if (pose2d) {
    s.xf.r00 = pose2d[6 * sc]; s.xf.r01 = pose2d[6 * sc + 1]; s.xf.tx = pose2d[6 * sc + 2];
    s.xf.r10 = pose2d[6 * sc + 3]; s.xf.r11 = pose2d[6 * sc + 4]; s.xf.ty = pose2d[6 * sc + 5];
  }
  return s;
}
// (one loop instance per uniform condition, so that none of them is tested per block)
template <bool FAST_DIV, bool SAFE, bool SPLIT, bool DBG, int AHEAD, int RORM = 0>
__device__ __forceinline__ void voxel_stream_dispatch(QueueSink &sink, const KParams &p,
                                                      const float2 *__restrict__ cs,
                                                      const __amdgpu_buffer_rsrc_t rsrc, uint32_t blk0,
                                                      uint32_t blk_end, const ScanSide &sd,
                                                      uint32_t mask_stride, bool use_xf,
                                                      uint32_t q_min16, uint32_t ibfe_off,
                                                      uint32_t ibfe_w, uint32_t &flags,
                                                      unsigned long long *dbg_slot,
                                                      RorSide *rs = nullptr, uint32_t oob_off = 0u) {
  if constexpr (RORM == 1) {  // E5 inside the pass: transform (quality test always on) / quality filter / neither
    if (use_xf)
      voxel_stream<FAST_DIV, SAFE, SPLIT, false, true, false, true, AHEAD, true>(
          sink, p, cs, rsrc, blk0, blk_end, nullptr, 0u, sd.xf, q_min16, ibfe_off, ibfe_w, flags, nullptr, rs, oob_off);
    else if (q_min16)
      voxel_stream<FAST_DIV, SAFE, SPLIT, false, true, false, false, AHEAD, true>(
          sink, p, cs, rsrc, blk0, blk_end, nullptr, 0u, sd.xf, q_min16, ibfe_off, ibfe_w, flags, nullptr, rs, oob_off);
    else
      voxel_stream<FAST_DIV, SAFE, SPLIT, false, false, false, false, AHEAD, true>(
          sink, p, cs, rsrc, blk0, blk_end, nullptr, 0u, sd.xf, q_min16, ibfe_off, ibfe_w, flags, nullptr, rs, oob_off);
    return;
  }
#define RPL_VS(HQ, HM, XFB)                                                                         \
  voxel_stream<FAST_DIV, SAFE, SPLIT, DBG, HQ, HM, XFB, AHEAD>(sink, p, cs, rsrc, blk0, blk_end, \
                                                                     sd.ror_bits, mask_stride, sd.xf, \
                                                                     q_min16, ibfe_off, ibfe_w, flags, \
                                                                     dbg_slot)
  if constexpr (RORM == 2) {  // the listed items of the two-kernel E5 path: mask word always on
    if (use_xf) RPL_VS(true, true, true);
    else RPL_VS(true, true, false);
    return;
  }
  if (use_xf) {  // (E8 / de-skew: one instance, quality test and mask word always on)
    if (sd.ror_bits) RPL_VS(true, true, true);
    else RPL_VS(true, false, true);
  } else if (sd.ror_bits) RPL_VS(true, true, false);
  else if (q_min16) RPL_VS(true, false, false);
  else RPL_VS(false, false, false);
#undef RPL_VS
}

// ------------------------------------------------------------------------------
// The persistent per-item loop shared by the fused kernel and by k_voxel_cells: draw a work item,
// let `first_band(b, bl, flags)` put the item's queue entries in place — entries [0, kRecCap) in the
// LDS queue, the rest at their own index in this workgroup's record store G, the total in
// L.misc[0] — then turn key bands of them into cells (phase R) and publish the item's results.
// item0: first item of this launch (a launch may be one stage of a batch); B: its items.
// ------------------------------------------------------------------------------
// RORM (round 6): 0 = the kernel as it always was; 1 = E5 inside the pass — `first_band` returns true for
// an item whose scans hold more unsettled samples than the kernel resolves itself, the item then goes
// on T.redo (count word, then the item numbers) and nothing of it is published here; 2 = the items
// of that list (their E5 masks made by k_ror_mask in between), B read from the list's count word.
template <bool DBG, int RORM = 0, class FirstBand>
__device__ __forceinline__ void voxel_work_loop(VoxelLds &L, const KParams &p, const Tables &T,
                                                uint4 *__restrict__ G, float4 *__restrict__ xyzi,
                                                uint32_t out_stride, uint32_t *__restrict__ n_points,
                                                uint32_t *__restrict__ status, uint32_t B,
                                                uint32_t item0, const VoxelArena &arena,
                                                FirstBand &&first_band) {
  if (threadIdx.x < 256) L.rcp[threadIdx.x] = T.rcp[threadIdx.x];  // (first barrier below publishes it)
  if constexpr (RORM == 2) B = min(B, (uint32_t)__builtin_amdgcn_readfirstlane((int)T.redo[0]));
  // persistent workgroups; the first item is blockIdx.x, the next ones come from a shared counter,
  // so a workgroup that drew cheap scans simply takes more of them
  for (uint32_t bl = blockIdx.x; bl < B;) {
  const uint32_t b = RORM == 2 ? T.redo[4u + bl] : item0 + bl;
  bool abandoned = false;  // block-uniform (RORM == 1)
  float4 *out = arena.base ? arena.base : xyzi + (size_t)b * out_stride;
  int emit_mode = arena.base ? kEmitArenaFirst : kEmitLegacy;
  unsigned long long arena_at = 0ull;  // first point of this scan in the arena

  if (threadIdx.x < 16) L.misc[threadIdx.x] = 0u;
  if (threadIdx.x == 0) {
    L.band_lo[0] = 0u;
    L.band_hi[0] = 0xFFFFFFFEu;
    L.misc[3] = 1u;  // stack pointer
    L.tmp[28] = 0u;  // arena reservation of this scan (stays 0 for a scan without cells)
    L.tmp[29] = 0u;
  }
  __syncthreads();

  uint32_t flags = 0;
  PhaseClock<DBG> pc;
  pc.start();

  bool first = true;
  bool leave_single_band_mode_early = false;
  bool from_store = false;  // block-uniform: the scan's records are (all) in the record store
  uint32_t n_all = 0;       // records of the whole scan (valid once the scan was streamed)
  while (true) {
    // ---- pop a key band -----------------------------------------------------------
    const uint32_t sp = L.misc[3];
    if (sp == 0) break;
    const uint32_t klo = L.band_lo[sp - 1], khi = L.band_hi[sp - 1];
    __syncthreads();
    if (threadIdx.x == 0) {
      L.misc[0] = 0u;
      L.misc[2] = 0u;
      L.misc[3] = sp - 1;
      L.misc[4] = 0xFFFFFFFFu;
      L.misc[5] = 0u;
      L.misc[8] = 0xFFFFFFFFu;  // occupied key range of the band (known after a failed selection)
      L.misc[9] = 0u;
    }
    // the row histogram of this band is filled while the records are loaded in phase R (rows
    // are addressed modulo kRowCap, so the first row need not be known yet)
    for (uint32_t t = threadIdx.x; t < kRowCap; t += kVB) L.rows[t] = 0u;
    __syncthreads();

    if (first) {
      abandoned = first_band(b, bl, flags);
      first = false;
      // the record stores of phase S are hand-written instructions the compiler does not track
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();  // (workgroup-scope release / acquire: covers the record store too)
      if (RORM == 1 && abandoned) {  // block-uniform: left to the two-kernel path
        if (threadIdx.x == 0) {
          T.redo[4u + atomicAdd(&T.redo[0], 1u)] = b;
          // (what a caller sees who looks before the listed launches ran — the single-scan entry points
          // do, instead of launching them blind: no points and the internal "listed" bit)
          n_points[b] = 0u;
          if (status) status[b] = kRorListedBit;
          if (arena.base) arena.scan_start[b] = 0ull;
        }
        break;
      }
      pc.lap(0);
      n_all = L.misc[0];
      if (threadIdx.x == 0 && T.voxel_stats) {  // queue statistics of the launch (see SPLIT)
        atomicAdd(&T.voxel_stats[0], (unsigned long long)n_all);
        atomicAdd(&T.voxel_stats[1], 1ull);
      }
      if (n_all > kRecCap) {
        // the scan did not fit the LDS queue: its first kRecCap records join the others in the
        // record store, and the bands below are cut from the store
        for (uint32_t i = threadIdx.x; i < kRecCap; i += kVB) G[i] = L.rec[i];
        from_store = true;
        if (threadIdx.x == 0) {
          L.misc[2] = 1u;
          L.misc[10] = 0xFFFFFFFFu;
          L.misc[11] = 0u;
        }
        __syncthreads();
        // Round 5: the bands come from a HISTOGRAM of the records' rows instead of from bisecting the
        // key range (every bisection step that still held too many records cost a selection pass
        // over the whole store, and left bands half full: uniform random samples — one record per
        // sample, 28 800 per scan — took ~14 passes and 8 reduces where 1 + 5 and 5 do).  Two key-only
        // passes over the store: row range, then counts per bin of 2^shift rows; a block scan; a
        // band starts wherever the running count crosses a multiple of the fill (the fewest, equally
        // full bands that 15/16-full queues allow).  (Also built: one more pass that sets every record
        // apart by band in a third store region, the bands then copying their own records only —
        // no faster: the store lives in HBM, not in L2, and every pass over it is a chain of
        // round trips whatever it reads; removed.)  A band that
        // still holds too many records (one bin denser than the queue) is bisected as before.
        // (A scan of up to three queues' worth of records keeps the bisection: its first cut is iy = 0,
        // which halves a ring; measured on rings with 3 cm of range noise, 14-20 k records per scan:
        // 1.64 ms per batch bisected, 1.82 with histogram bands.)
        if (n_all > 3u * (kRecCap - kRecCap / 16u)) {
          uint32_t rmn = 0xFFFFFFFFu, rmx = 0u;
          constexpr int KU = 8;  // key loads in flight per thread (one at a time: 29 round trips per pass)
          for (uint32_t i0 = threadIdx.x; i0 < n_all; i0 += KU * kVB) {
            uint32_t kk[KU];
#pragma unroll
            for (int u = 0; u < KU; ++u) {
              const uint32_t i = i0 + (uint32_t)u * kVB;
              kk[u] = i < n_all ? G[i].x : kEmptyKey;
            }
#pragma unroll
            for (int u = 0; u < KU; ++u) {
              if (kk[u] != kEmptyKey) {
                rmn = min(rmn, kk[u] >> 16);
                rmx = max(rmx, kk[u] >> 16);
              }
            }
          }
          rmn = wave_min_lane63(rmn);
          rmx = wave_max_lane63(rmx);
          if (lane_id() == 63 && rmn != 0xFFFFFFFFu) {
            atomicMin(&L.misc[10], rmn);
            atomicMax(&L.misc[11], rmx);
          }
          __syncthreads();
          rmn = L.misc[10];
          rmx = L.misc[11];
          if (rmn <= rmx) {  // block-uniform
            uint32_t shift = 0u;
            while (((rmx - rmn) >> shift) >= kRowCap) ++shift;
            for (uint32_t i0 = threadIdx.x; i0 < n_all; i0 += KU * kVB) {  // (L.rows[0 .. kRowCap) is zero here)
              uint32_t kk[KU];
#pragma unroll
              for (int u = 0; u < KU; ++u) {
                const uint32_t i = i0 + (uint32_t)u * kVB;
                kk[u] = i < n_all ? G[i].x : kEmptyKey;
              }
#pragma unroll
              for (int u = 0; u < KU; ++u)
                if (kk[u] != kEmptyKey) atomicAdd(&L.rows[((kk[u] >> 16) - rmn) >> shift], 1u);
            }
            __syncthreads();
            uint32_t cnt[kRowsPerThread], mine = 0u;
#pragma unroll
            for (uint32_t r = 0; r < kRowsPerThread; ++r) {
              cnt[r] = L.rows[kRowsPerThread * threadIdx.x + r];
              mine += cnt[r];
            }
            const uint32_t before_me = threadIdx.x ? L.rows[kRowsPerThread * threadIdx.x - 1u] : 0u;
            uint32_t tot;
            uint32_t ex = vx_excl_scan(mine, L.tmp, &tot);
            // as few bands as 15/16-full queues allow, equally full
            const uint32_t want = (tot + (kRecCap - kRecCap / 16u) - 1u) / (kRecCap - kRecCap / 16u);
            const uint32_t kBandFill = max((tot + want - 1u) / max(want, 1u), 1u);
            // a band boundary in front of bin b: the running count passes a multiple of kBandFill
            uint32_t flag[kRowsPerThread], nflag = 0u, prev_cnt = before_me, e = ex;
#pragma unroll
            for (uint32_t r = 0; r < kRowsPerThread; ++r) {
              const uint32_t bin = kRowsPerThread * threadIdx.x + r;
              flag[r] = (bin > 0u && e / kBandFill != (e - prev_cnt) / kBandFill) ? 1u : 0u;
              nflag += flag[r];
              prev_cnt = cnt[r];
              e += cnt[r];
            }
            uint32_t nbound;
            uint32_t fe = vx_excl_scan(nflag, L.tmp, &nbound);
            const uint32_t nbands = nbound + 1u;
            if (nbands >= 2u && nbands <= 64u) {  // block-uniform (else: bisection as before)
#pragma unroll
              for (uint32_t r = 0; r < kRowsPerThread; ++r) {
                if (flag[r]) {
                  const uint32_t bin = kRowsPerThread * threadIdx.x + r;
                  const uint32_t j = fe + 1u;  // the band that starts here (0 = lowest keys)
                  const uint32_t key0 = (rmn + (bin << shift)) << 16;
                  L.band_lo[nbands - 1u - j] = key0;        // (the stack is popped from the top: band 0 on top)
                  L.band_hi[nbands - j] = key0 - 1u;
                  ++fe;
                }
              }
              if (threadIdx.x == 0) {
                L.band_lo[nbands - 1u] = 0u;
                L.band_hi[0] = 0xFFFFFFFEu;
                L.misc[3] = nbands;
                L.misc[2] = 0u;
              }
              __syncthreads();
              leave_single_band_mode_early = true;
            }
          }
        }
      }
    } else {
      // ---- a band of a scan that lives in the record store: select its records ----------
      // record i -> run sums against record i - 1 (unless first of its wave-pass), exactly what
      // phase R does for the LDS queue; the records of the band are appended to the LDS queue
      bool fits = true;
      uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;  // keys of this band's records seen by this lane
      constexpr int NJ = 4;  // records per thread and trip: four independent loads in flight
      for (uint32_t i0 = 0; i0 < n_all; i0 += NJ * kVB) {
        uint4 raw[NJ], prl[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const uint32_t i = i0 + (uint32_t)j * kVB + threadIdx.x;
          raw[j] = make_uint4(kEmptyKey, 0u, 0u, 0u);
          prl[j] = make_uint4(0u, 0u, 0u, 0u);
          if (i < n_all) raw[j] = G[i];
          // the record before a wave's first one lives in another wave's registers: lane 0 loads it
          if (lane_id() == 0u && i > 0u && i < n_all) prl[j] = G[i - 1u];
        }
        bool in[NJ];
        uint64_t msk[NJ];
        uint32_t cnt = 0;
        uint4 m[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const uint32_t i = i0 + (uint32_t)j * kVB + threadIdx.x;
          uint4 pr;  // record i - 1: the lane to the left holds it
          pr.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw[j].y, 0x138, 0xF, 0xF, false);
          pr.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw[j].z, 0x138, 0xF, 0xF, false);
          pr.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw[j].w, 0x138, 0xF, 0xF, false);
          if (lane_id() == 0u) { pr.y = prl[j].y; pr.z = prl[j].z; pr.w = prl[j].w; }
          m[j] = raw[j];  // (markers: empty key, outside every band; entry 0 is a marker)
          m[j].y -= pr.y;
          m[j].z -= pr.z;
          m[j].w -= pr.w;
          in[j] = (i < n_all) && ((raw[j].x - klo) <= (khi - klo));
          if (in[j]) {
            kmin = min(kmin, raw[j].x);
            kmax = max(kmax, raw[j].x);
          }
          msk[j] = __builtin_amdgcn_ballot_w64(in[j]);
          cnt += (uint32_t)__popcll(msk[j]);
        }
        if (fits && cnt) {  // wave-uniform; once the queue is full only the key range is collected
          uint32_t base = 0u;
          if (lane_id() == 0) base = atomicAdd(&L.misc[0], cnt);
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          if (base + cnt > kRecCap) {
            fits = false;  // the band holds more than the queue: it will be bisected
          } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              if (in[j]) L.rec[base + (uint32_t)__popcll(msk[j] & lanemask_lt())] = m[j];
              base += (uint32_t)__popcll(msk[j]);
            }
          }
        }
      }
      // the key range the band's records really occupy (every wave saw its share of them): a
      // bisection halves THAT, not the nominal range
      kmin = wave_min_lane63(kmin);
      kmax = wave_max_lane63(kmax);
      if (lane_id() == 63) {
        if (!fits) L.misc[2] = 1u;
        atomicMin(&L.misc[8], kmin);
        atomicMax(&L.misc[9], kmax);
      }
      __syncthreads();
      if (DBG) pc.lap(2);
    }

    auto bisect = [&]() {  // block-uniform: replace the band by its two halves
      if (threadIdx.x == 0) {
        uint32_t s = L.misc[3];
        // halve the key range the band's records really occupy when it is known (a failed
        // selection pass measured it), else the nominal range (the first split of a scan: the
        // middle of the key space is iy = 0)
        uint32_t lo = klo, hi = khi;
        if (L.misc[8] <= L.misc[9]) { lo = max(lo, L.misc[8]); hi = min(hi, L.misc[9]); }
        if (lo >= hi || s + 2 > 65) {
          L.misc[1] |= RPLGPU_SCAN_TABLE_FULL;  // cannot happen: one key is one cell/row
        } else {
          uint32_t mid = lo + (hi - lo) / 2;
          L.band_lo[s] = mid + 1;  // upper half is processed after the lower half
          L.band_hi[s] = hi;
          L.band_lo[s + 1] = lo;
          L.band_hi[s + 1] = mid;
          L.misc[3] = s + 2;
        }
      }
      __syncthreads();
    };
    auto leave_single_band_mode = [&]() {  // block-uniform
      if (emit_mode != kEmitArenaFirst) return;
      if (p.cell_keys) {
        emit_mode = kEmitCountOnly;
      } else {
        emit_mode = kEmitArenaTemp;
        out = reinterpret_cast<float4 *>(G + T.voxel_store_recs);  // the workgroup's cell area
      }
    };
    if (leave_single_band_mode_early) {  // block-uniform: the histogram above made this scan's bands
      leave_single_band_mode_early = false;
      leave_single_band_mode();
      continue;
    }
    if (L.misc[2]) {
      bisect();
      leave_single_band_mode();
      continue;
    }

    // ---- phase R (out of line) ----------------------------------------------------------
    uint32_t ncell = 0;
    uint32_t out_limit = out_stride;  // cells this scan may write
    if (emit_mode == kEmitArenaKnown) {
      const unsigned long long room = arena_at >= arena.capacity ? 0ull : arena.capacity - arena_at;
      out_limit = (uint32_t)min(room, 0xFFFFFFFFull);
    } else if (emit_mode == kEmitArenaTemp) {
      out_limit = T.voxel_store_recs;  // (cells <= queue entries <= the area's size)
    } else if (emit_mode != kEmitLegacy) {
      out_limit = 0xFFFFFFFFu;  // (first band: bounded inside, at the reservation)
    }
    if (voxel_reduce<DBG>(L, p, out, out_limit, b, &ncell, emit_mode, arena, from_store,
                          arena.base ? arena.base : xyzi)) {
      if (!from_store) {  // the LDS queue spans too many rows: cut bands from the record store
        for (uint32_t i = threadIdx.x; i < n_all; i += kVB) G[i] = L.rec[i];
        from_store = true;
        __syncthreads();
      }
      bisect();
      leave_single_band_mode();
      continue;
    }
    if (emit_mode == kEmitArenaFirst)  // (the reservation made inside voxel_reduce)
      arena_at = ((unsigned long long)L.tmp[29] << 32) | L.tmp[28];
    const uint32_t out_base = L.misc[7];
    __syncthreads();
    if (threadIdx.x == 0) L.misc[7] = out_base + ncell;
    __syncthreads();
    pc.start();
    if (emit_mode == kEmitCountOnly && L.misc[3] == 0u) {
      // every band counted: reserve the scan's cells in one piece and go through the bands
      // again (same bisections, they depend on the data only), this time writing
      if (threadIdx.x == 0) {
        const unsigned long long at = atomicAdd(arena.cursor, (unsigned long long)L.misc[7]);
        L.tmp[28] = (uint32_t)at;
        L.tmp[29] = (uint32_t)(at >> 32);
        L.band_lo[0] = 0u;
        L.band_hi[0] = 0xFFFFFFFEu;
        L.misc[3] = 1u;
        L.misc[7] = 0u;
      }
      __syncthreads();
      arena_at = ((unsigned long long)L.tmp[29] << 32) | L.tmp[28];
      out = arena.base + arena_at;
      emit_mode = kEmitArenaKnown;
    }
    if (emit_mode == kEmitArenaTemp && L.misc[3] == 0u) {
      // every band is in the cell area: reserve the scan's place in the arena, copy
      const uint32_t total = L.misc[7];
      if (threadIdx.x == 0) {
        const unsigned long long at = atomicAdd(arena.cursor, (unsigned long long)total);
        L.tmp[28] = (uint32_t)at;
        L.tmp[29] = (uint32_t)(at >> 32);
      }
      __syncthreads();  // (also: the cells written by all threads are visible workgroup-wide)
      arena_at = ((unsigned long long)L.tmp[29] << 32) | L.tmp[28];
      const unsigned long long room = arena_at >= arena.capacity ? 0ull : arena.capacity - arena_at;
      const uint32_t ncopy = (uint32_t)min((unsigned long long)total, room);
      const float4 *src = reinterpret_cast<const float4 *>(G + T.voxel_store_recs);
      if (arena.xyi) {
        float *dst = reinterpret_cast<float *>(arena.base) + 3u * (size_t)arena_at;
        for (uint32_t i = threadIdx.x; i < ncopy; i += kVB) {
          const float4 v = src[i];
          dst[3u * i] = v.x;
          dst[3u * i + 1u] = v.y;
          dst[3u * i + 2u] = v.w;
        }
      } else {
        float4 *dst = arena.base + arena_at;
        for (uint32_t i = threadIdx.x; i < ncopy; i += kVB) dst[i] = src[i];
      }
    }
  }
  if (DBG && p.dbg && threadIdx.x == 0) {
    atomicAdd(&p.dbg[16 * b], pc.acc[0]);
    atomicAdd(&p.dbg[16 * b + 2], pc.acc[2]);  // band selection passes over the record store
  }

  if (flags) atomicOr(&L.misc[1], flags);
  __syncthreads();
  if (threadIdx.x == 0 && !abandoned) {
    const uint32_t total = L.misc[7];
    if (arena.base) {
      const unsigned long long room = arena_at >= arena.capacity ? 0ull : arena.capacity - arena_at;
      const uint32_t kept = (uint32_t)min((unsigned long long)total, room);
      arena.scan_start[b] = arena_at;
      n_points[b] = kept;
      if (status) status[b] = L.misc[1] | (kept < total ? RPLGPU_SCAN_OUT_TRUNCATED : 0u);
    } else {
      n_points[b] = min(total, out_stride);
      if (status) status[b] = L.misc[1] | ((total > out_stride) ? RPLGPU_SCAN_OUT_TRUNCATED : 0u);
    }
  }
  if (RORM != 2 && B <= gridDim.x) return;  // a workgroup per item (single scans, small batches): no queue
  if (threadIdx.x == 0) L.tmp[31] = gridDim.x + atomicAdd(&T.work_ctr[0], 1u);
  __syncthreads();  // LDS is reused by the next scan
  bl = L.tmp[31];
  __syncthreads();
  }
}

// ------------------------------------------------------------------------------
// The FUSED kernel (single scans, small batches): one persistent workgroup per compute unit
// streams an item into its LDS queue (phase S) and reduces it (phase R) itself.
// SPLIT: the instance for batches known to be noisy (its blocks may be aggregated in two classes,
// voxel_block_split); the launcher picks it from the queue statistics of the handle's previous
// launch (T.voxel_stats), so a clean batch runs code without a trace of that path.
// ------------------------------------------------------------------------------
// RORM (round 6): 1 = the ROR instance — E5 inside the pass (voxel_stream HASROR, ror_resolve);
// 2 = the work items that instance left on T.redo, with the masks k_ror_mask made for them.
template <bool FAST_DIV, bool SAFE, bool DBG, bool SPLIT, int RORM = 0>
__global__ __launch_bounds__(kVB) __attribute__((amdgpu_waves_per_eu(kVB * kVWG / 256, kVB * kVWG / 256))) void k_cloud_voxel(
    const uint2 *__restrict__ nodes, uint32_t n_stride, const uint32_t *__restrict__ n_per_scan,
    KParams p, Tables T, const uint32_t *__restrict__ keepmask, uint32_t mask_stride,
    float4 *__restrict__ xyzi, uint32_t out_stride, uint32_t *__restrict__ n_points,
    uint32_t *__restrict__ status, uint32_t B, VoxelArena arena, uint4 *__restrict__ store,
    uint32_t group, uint32_t n_scans, const float *__restrict__ motion,
    const float *__restrict__ pose2d) {
  // B work items; item b = the scans [b * group, min(n_scans, (b + 1) * group)) sharing ONE grid
  // (group == 1: a scan is an item, the round-1 behaviour; E8 otherwise)
  __shared__ VoxelLds L;
  // this workgroup's record store (T.voxel_store_recs entries, voxel_store_need(): every sample
  // of the work item could end a run, plus the block markers)
  uint4 *G = store + (size_t)blockIdx.x * 2u * T.voxel_store_recs;  // records, then the cell area
  const float2 *cs = p.inverted ? T.cs_inv : T.cs;
  const uint32_t ishift = p.is_new_protocol ? 0u : 2u;  // :591-592
  const uint32_t ibfe_off = 16u + ishift, ibfe_w = 0xFFu >> ishift;  // shift, mask
  const uint32_t q_min16 = p.clip_enable ? (min(p.q_min, 256u) << 16) : 0u;
  const bool use_xf = (group > 1u) || motion || pose2d;  // block-uniform
  __shared__ RorSide Rs;  // (referenced by the ROR instance only: the others do not allocate it)
  auto phase_s = [&](uint32_t b, uint32_t, uint32_t &flags) -> bool {
    QueueSink sink{L, G};
    const uint32_t s_lo = b * group, s_hi = min(n_scans, s_lo + group);
    for (uint32_t sc = s_lo; sc < s_hi; ++sc) {
      // (readfirstlane: the value is wave-uniform, and must live in scalar registers for the
      // buffer resource below — a min3 in vector registers makes every load a waterfall loop)
      const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane(
          (int)min(n_per_scan[sc], min(n_stride, kMaxN)));  // never past the slot
      const uint2 *scan = nodes + (size_t)sc * n_stride;
      // Bounds-checked buffer resource over this scan's n*8 bytes: a node beyond the scan reads
      // as zero, i.e. dist 0, which the keep test drops.  Every lane always issues the load, so
      // the compiler's vmcnt bookkeeping is exact.
      const __amdgpu_buffer_rsrc_t scan_rsrc =
          __builtin_amdgcn_make_buffer_rsrc((void *)scan, 0, (int)(n * 8u), 0x00020000);
      const ScanSide sd = scan_side(sc, keepmask, mask_stride, motion, pose2d, T.scan_t0);
      const uint32_t blk0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_id());
      if constexpr (RORM == 1) {
        if (threadIdx.x == 0) { Rs.n_todo = 0u; Rs.n_todo2 = 0u; }
        __syncthreads();
        voxel_stream_dispatch<FAST_DIV, SAFE, SPLIT, false, RPL_VOXEL_AHEAD, 1>(
            sink, p, cs, scan_rsrc, blk0, (n + 123u) / 124u, sd, 0u, use_xf, q_min16, ibfe_off, ibfe_w, flags,
            nullptr, &Rs, (n * 8u + 15u) & ~15u);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (the pass's hand-written record stores)
        __syncthreads();
        bool clutter;
        if (use_xf)
          clutter = ror_resolve<FAST_DIV, SAFE, true>(sink, Rs, p, cs, scan, n, sd.xf, q_min16, ibfe_off, ibfe_w, flags);
        else
          clutter = ror_resolve<FAST_DIV, SAFE, false>(sink, Rs, p, cs, scan, n, sd.xf, q_min16, ibfe_off, ibfe_w, flags);
        if (clutter) return true;  // block-uniform
        __syncthreads();           // (the lists are reused by the group's next scan)
      } else {
        voxel_stream_dispatch<FAST_DIV, SAFE, SPLIT, DBG, RPL_VOXEL_AHEAD, RORM>(
            sink, p, cs, scan_rsrc, blk0, (n + 127u) >> 7, sd, mask_stride, use_xf, q_min16, ibfe_off,
            ibfe_w, flags, (DBG && p.dbg) ? p.dbg + 16 * b : nullptr);
      }
    }
    return false;
  };
  voxel_work_loop<DBG, RORM>(L, p, T, G, xyzi, out_stride, n_points, status, B, 0u, arena, phase_s);
}

// ------------------------------------------------------------------------------
// Divisor validation: div_by(a, d, RN(1/d)) must equal the IEEE quotient a / d for
// every fp32 `a` with biased exponent in [e_lo, e_hi] (both signs); +-0 must give a zero.
// The packed form (v_pk_mul_f32 / v_pk_fma_f32) is checked in the same sweep.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_validate_div(float d, float rd, uint32_t e_lo,
                                                      uint32_t e_hi, uint32_t *mismatches) {
  const uint64_t per_exp = 1ull << 23;
  const uint64_t total = (uint64_t)(e_hi - e_lo + 1) * per_exp;
  uint32_t bad = 0;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t bits = ((uint32_t)(e_lo + (uint32_t)(t >> 23)) << 23) | (uint32_t)(t & (per_exp - 1));
    float a = __uint_as_float(bits);
    float na = __uint_as_float(bits | 0x80000000u);
    const f2 q2 = div_by2(f2{a, na}, d, rd);
    bad += (__float_as_uint(q2.x) != __float_as_uint(a / d));
    bad += (__float_as_uint(q2.y) != __float_as_uint(na / d));
    bad += (__float_as_uint(div_by(a, d, rd)) != __float_as_uint(a / d));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    bad += (div_by(0.0f, d, rd) != 0.0f);
    bad += (div_by(-0.0f, d, rd) != 0.0f);
  }
  if (bad) atomicAdd(mismatches, bad);
}

hipError_t launch_validate_div(hipStream_t s, float d, float rd, uint32_t e_lo, uint32_t e_hi,
                               uint32_t *d_mismatches) {
  hipLaunchKernelGGL(k_validate_div, dim3(256 * 16), dim3(256), 0, s, d, rd, e_lo, e_hi,
                     d_mismatches);
  return hipGetLastError();
}

uint32_t voxel_max_workgroups(uint32_t n_cu) { return (uint32_t)kVWG * (n_cu ? n_cu : 256u); }


hipError_t launch_cloud_voxel(hipStream_t s, const void *nodes, uint32_t n_stride,
                              const uint32_t *n_per_scan, uint32_t B, const KParams &p,
                              const Tables &T, const uint32_t *keepmask, uint32_t mask_stride,
                              float *xyzi, uint32_t out_stride, uint32_t *n_points,
                              uint32_t *status, float *arena, unsigned long long arena_capacity,
                              unsigned long long *arena_cursor, unsigned long long *scan_start,
                              uint32_t group, const float *motion, const float *pose2d,
                              bool arena_xyi, int ror_mode) {
  if (B == 0) return hipSuccess;
  if (ror_mode && (!p.fast_div || !T.redo || p.dbg)) return hipErrorInvalidValue;
  if (ror_mode == 2 && !keepmask) return hipErrorInvalidValue;
  if (group == 0) group = 1;
  group = std::min(group, B);  // (a group larger than the batch is the whole batch)
  const uint32_t n_scans = B;
  B = (B + group - 1u) / group;  // work items
  if (!T.voxel_store || T.voxel_store_wgs == 0) return hipErrorInvalidValue;
  // (the handle allocates 2 x voxel_store_recs entries per workgroup: records, then the cell area)
  if (voxel_store_need(group, n_stride) > T.voxel_store_recs) return hipErrorInvalidValue;
  VoxelArena ar;
  ar.base = (float4 *)arena;
  ar.cursor = arena_cursor;
  ar.capacity = arena_capacity;
  ar.scan_start = scan_start;
  ar.xyi = (arena && arena_xyi) ? 1 : 0;
  if (kVWG == 2 && group > 1) return hipErrorInvalidValue;  // (fused groups: 16-wave geometry only)
  // persistent workgroups of the handle's device (no more than the handle owns record stores
  // for); the item queue is cleared by a memset ahead of every launch (an aborted launch can
  // therefore not poison the next one)
  uint32_t grid = std::min<uint32_t>(std::min<uint32_t>(B, voxel_max_workgroups(T.n_cu)),
                                     T.voxel_store_wgs);
  if (const char *e = std::getenv("RPLGPU_VOXEL_GRID")) {  // developer aid
    const long g = std::atol(e);
    if (g > 0) grid = std::min<uint32_t>(grid, (uint32_t)g);
  }
  // queue statistics of the launch (they pick the NEXT launch's instance): batches only
  const bool with_stats = T.voxel_stats && T.voxel_stats_host && B >= 64u && ror_mode != 2;
  Tables Tk = T;
  if (!with_stats) Tk.voxel_stats = nullptr;
  if (with_stats)
    if (hipError_t e = hipMemsetAsync(T.voxel_stats, 0, 16, s); e != hipSuccess) return e;
  // (a launch with a workgroup per item does not touch the queue: one command less in front of a
  // single-scan call)
  if (grid < B || ror_mode == 2)
    if (hipError_t e = hipMemsetAsync(T.work_ctr, 0, 4, s); e != hipSuccess) return e;
#define RPL_LAUNCH_VOXEL(FD, SF, DB, SP)                                                          \
  hipLaunchKernelGGL((k_cloud_voxel<FD, SF, DB, SP>), dim3(grid), dim3(kVB), 0, s, (const uint2 *)nodes, \
                     n_stride, n_per_scan, p, Tk, keepmask, mask_stride, (float4 *)xyzi, out_stride, \
                     n_points, status, B, ar, (uint4 *)T.voxel_store, group, n_scans, motion, pose2d)
#define RPL_LAUNCH_VOXEL_SF(FD, DB, SP)                                                           \
  do {                                                                                             \
    if (p.cell_range_safe) RPL_LAUNCH_VOXEL(FD, true, DB, SP); else RPL_LAUNCH_VOXEL(FD, false, DB, SP); \
  } while (0)
#define RPL_LAUNCH_VOXEL_ROR(SF, SP, RM)                                                          \
  hipLaunchKernelGGL((k_cloud_voxel<true, SF, false, SP, RM>), dim3(grid), dim3(kVB), 0, s,         \
                     (const uint2 *)nodes, n_stride, n_per_scan, p, Tk, keepmask, mask_stride,      \
                     (float4 *)xyzi, out_stride, n_points, status, B, ar, (uint4 *)T.voxel_store,   \
                     group, n_scans, motion, pose2d)
#define RPL_LAUNCH_VOXEL_ROR_SF(SP, RM)                                                           \
  do {                                                                                             \
    if (p.cell_range_safe) RPL_LAUNCH_VOXEL_ROR(true, SP, RM); else RPL_LAUNCH_VOXEL_ROR(false, SP, RM); \
  } while (0)
  if (ror_mode == 1) {  // E5 inside the pass
    if (T.voxel_split) RPL_LAUNCH_VOXEL_ROR_SF(true, 1); else RPL_LAUNCH_VOXEL_ROR_SF(false, 1);
  } else if (ror_mode == 2) {  // the items it listed, behind k_ror_mask
    if (T.voxel_split) RPL_LAUNCH_VOXEL_ROR_SF(true, 2); else RPL_LAUNCH_VOXEL_ROR_SF(false, 2);
  } else
  if (p.dbg) {  // developer aid: the instrumented build of the kernel (plain instance only)
    if (p.fast_div) RPL_LAUNCH_VOXEL_SF(true, true, false); else RPL_LAUNCH_VOXEL_SF(false, true, false);
  } else if (T.voxel_split) {  // the handle's previous launch saw a noisy batch
    if (p.fast_div) RPL_LAUNCH_VOXEL_SF(true, false, true); else RPL_LAUNCH_VOXEL_SF(false, false, true);
  } else {
    if (p.fast_div) RPL_LAUNCH_VOXEL_SF(true, false, false); else RPL_LAUNCH_VOXEL_SF(false, false, false);
  }
#undef RPL_LAUNCH_VOXEL_SF
#undef RPL_LAUNCH_VOXEL
#undef RPL_LAUNCH_VOXEL_ROR_SF
#undef RPL_LAUNCH_VOXEL_ROR
  if (with_stats) {  // the statistics follow the launch to pinned memory (no wait)
    if (hipError_t e = hipMemcpyAsync(T.voxel_stats_host, T.voxel_stats, 16, hipMemcpyDeviceToHost, s);
        e != hipSuccess)
      return e;
  }
  return hipGetLastError();
}

}  // namespace rpl
